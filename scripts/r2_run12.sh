#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run12; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -n 2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["valu_issue"], d["apply_gemm"]["achieved"], d["parity"]["rel_l2_landmarks_vs_oracle"], d["cpu_baseline"]["value"])
PY
