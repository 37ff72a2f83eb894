cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "0 0" "64 4" "64 6" "64 12" "128 8" "128 4" "128 6" "0 0"; do
  set -- $cfg
  echo -n "BM=$1 SPLITS=$2: "
  SDM_APPLY_BM=$1 SDM_APPLY_SPLITS=$2 timeout 300 python bench.py --no-cpu --train-rows 2000 --rcr68-shard 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('faces/s %.0f ms/step %.4f apply %.4f ms' % (d['value'], d['ms_per_step'], d['apply_gemm']['avg_launch_ms']))"
done
