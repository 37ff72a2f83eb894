#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3_run1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -n 2; grep -E "^FAILED|Error|teacher-forced" $O/pytest_gpu.txt | head -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -n 3 $O/bench.err
SDM_HOG_NO_SPECIALISE=1 timeout 900 python bench.py --no-cpu --rcr68-shard 0 > $O/bench_generic.json 2> $O/bench_generic.err
python - <<PY
import json
for f in ("bench","bench_generic"):
    try:
        d=json.load(open("$O/%s.json" % f))
        print(f, "faces/s %.0f ms/step %.4f hog %.4f ms frac %.4f apply %.1f TF %.4f ms" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["apply_gemm"]["achieved"], d["apply_gemm"]["avg_launch_ms"]))
        if "rcr68_train" in d:
            t=d["rcr68_train"]; s=d["rcr68_detect_shard"]
            print("  rcr68 train s/level %.4f gram %.1f TF solve %.1f ms stages %s" % (t["sec_per_cascade"], t["gram"]["achieved"], t["solve_ms"], {k: round(v,2) for k,v in t["stage_ms_per_level_rank0"].items()}))
            print("  rcr68 detect %.0f faces/s hog %.3f ms frac %.4f apply %.1f TF" % (s["value"], s["hog"]["avg_launch_ms"], s["hog"]["frac"], s["apply_gemm"]["achieved"]))
        print("  train22", d["train"]["sec_per_cascade"], {k: round(v,2) for k,v in d["train"]["stage_ms_per_level_rank0"].items()})
        if "parity" in d: print("  parity", d["parity"]["rel_l2_landmarks_vs_oracle"], d["parity"]["faces_with_different_integer_decisions"])
    except Exception as e: print(f, "ERR", e)
PY
