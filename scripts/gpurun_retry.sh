#!/bin/bash
# gpurun with retries while no GPU slot / box is free (exit code 3: nothing charged)
# usage: scripts/gpurun_retry.sh <timeout_s> <logfile> <command...>
T=$1; LOG=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $LOG; then exit $rc; fi
  sleep 45
done
exit 3
