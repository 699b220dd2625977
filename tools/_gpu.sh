cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sd3_model_gpu.py tests/test_unet_model_gpu.py tests/test_vae_gpu.py -x -q -m gpu -s -k "sampling or sample" 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -30
