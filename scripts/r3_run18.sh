#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run18; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -n 2; grep -E "^FAILED" $O/pytest_gpu.txt | head -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 900 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("faces/s %.0f ms/step %.4f hog %.4f apply %.1f TF" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["apply_gemm"]["achieved"]))
t=d["rcr68_train"]; s=d["rcr68_detect_shard"]
print("rcr68 train s/level %.4f %s" % (t["sec_per_cascade"], {k: round(v,2) for k,v in t["stage_ms_per_level_rank0"].items()}))
print("rcr68 detect %.0f faces/s ms/step %.3f hog %.3f ms apply %.1f TF %.4f ms" % (s["value"], s["ms_per_step"], s["hog"]["avg_launch_ms"], s["apply_gemm"]["achieved"], s["apply_gemm"]["avg_launch_ms"]))
print("train22", d["train"]["sec_per_cascade"], {k: round(v,2) for k,v in d["train"]["stage_ms_per_level_rank0"].items()})
PY
