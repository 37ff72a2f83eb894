#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_run14; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -n 2; grep -E "^FAILED|Error" $O/pytest_gpu.txt | head -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
for v in 0 1; do
SDM_NO_FUSED_UPDATE=$v timeout 900 python bench.py --no-cpu > $O/bench_$v.json 2> $O/bench_$v.err
python - <<PY
import json
d=json.load(open("$O/bench_$v.json"))
print("nofuse=$v", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["apply_gemm"]["achieved"], d["apply_gemm"]["avg_launch_ms"], d["apply_gemm"].get("standalone_update_launches"))
PY
done
