#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -s -k teacher_forced 2>&1 | grep -E "teacher-forced|passed|failed" | tee $O/teacher_forced.txt
python scripts/apply_variants.py 0 1 2>&1 | tee $O/apply_variants.txt
for v in 0 1; do
SDM_APPLY_VARIANT=$v timeout 600 python bench.py --no-cpu --rcr68-shard 0 > $O/bench_apply$v.json 2> $O/bench_apply$v.err
python - <<PY
import json
d=json.load(open("$O/bench_apply$v.json"))
print("variant $v", "faces/s %.0f ms/step %.4f hog %.4f apply %.1f TF %.4f ms" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["apply_gemm"]["achieved"], d["apply_gemm"]["avg_launch_ms"]))
PY
done
