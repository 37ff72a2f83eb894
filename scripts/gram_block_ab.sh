REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in "4 8" "2 16" "8 4" "2 8" "4 4" "1 32" "16 2" "4 16" "8 8"; do
  set -- $cfg
  export SDM_GRAM_BLOCK_H=$1 SDM_GRAM_BLOCK_W=$2
  t=$(python $REPO/scripts/gram_timing.py 100000 2>&1 | tail -n 1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['gram_ms'])")
  rm -rf /tmp/pf; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pmc -- python $REPO/scripts/gram_timing.py 100000 > /dev/null 2>&1
  f=$(python $REPO/scripts/pmc_by_grid.py /tmp/pf syrk_tn_split | grep FETCH_SIZE | awk '{print $2}')
  echo "block ${1}x${2}: gram_ms $t  FETCH_SIZE(KB) $f"
done
