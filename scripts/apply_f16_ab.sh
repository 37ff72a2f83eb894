#!/bin/bash
# A/B of the regressor apply: f32 matrix-core kernel (SDM_APPLY_F32=1) against the float16-piece kernel (default), bench line figures
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in 1 0; do
  echo "SDM_APPLY_F32=$v"
  SDM_APPLY_F32=$v timeout 600 python bench.py --no-cpu --train-rows 20000 --rcr68-shard 8192 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('faces/s %.0f ms/step %.4f hog %.4f apply %.1f TF %.4f ms  parity %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['apply_gemm']['achieved'], d['apply_gemm']['avg_launch_ms'], json.dumps(d.get('parity'))[:200]))
s=d['rcr68_detect_shard']; print('rcr68 detect %.0f faces/s ms/step %.3f apply %.1f TF %.4f ms' % (s['value'], s['ms_per_step'], s['apply_gemm']['achieved'], s['apply_gemm']['avg_launch_ms']))
t=d['rcr68_train']; print('rcr68 train apply ms', t['stage_ms_per_level_rank0']['apply'])"
done
