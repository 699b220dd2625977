#!/bin/bash
# the driver's end-of-round sequence on one box: the whole GPU suite, smoke(), the default bench line
tag=${1:-r03h}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/${tag}_pytest_gpu.log 2>&1
tail -5 gpurun_out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; tail -3 gpurun_out/${tag}_smoke.log
timeout 700 python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
grep "^\[bench" gpurun_out/${tag}_bench.err | tail -20
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_bench_line.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_stats")}, d["roofline"]["achieved"], d["cpu_baseline"]["value"], d["parity_at_config"]["pred_rel_l2"])
    print({k: (v.get("value"), v.get("ms_per_step"), v.get("error")) for k, v in d["secondary"].items()})
    print({k: (v.get("pred_rel_l2"), v.get("grad_worst_rel_l2"), v.get("seconds"), v.get("error"), v.get("skipped")) for k, v in d["parity_at_other_configs"].items()})
except Exception as e:
    print("bench FAILED", e); print(open("gpurun_out/${tag}_bench.err").read()[-1500:])
PY
