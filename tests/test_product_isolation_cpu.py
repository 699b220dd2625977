"""The product has no CPU path and never reaches test infrastructure: nothing under simpletuner_amd/ imports `oracle` (the CPU restatement of the reference) or `tests`
(parity utilities, the kernel-contract emulator), and the ops wrappers refuse host tensors instead of falling back."""
import ast
import glob
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_product_module_imports_the_oracle_or_the_test_infrastructure():
    bad = []
    for path in glob.glob(os.path.join(ROOT, "simpletuner_amd", "**", "*.py"), recursive=True):
        tree = ast.parse(open(path).read(), filename=path)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom) and node.level == 0 and node.module:
                names = [node.module]
            for n in names:
                if n.split(".")[0] in ("oracle", "tests"):
                    bad.append((os.path.relpath(path, ROOT), n))
    assert not bad, bad


def test_ops_wrappers_refuse_host_tensors():
    from simpletuner_amd import ops
    from simpletuner_amd.lib import St355Error
    x = torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises(St355Error, match="no CPU path"):
        ops.gemm(x, x)
    with pytest.raises(St355Error, match="no CPU path"):
        ops.silu(x)
    with pytest.raises(St355Error, match="no CPU path"):
        ops.ln_modulate_fwd(x, x[:1], x[:1], 64)
