#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> <logfile> <command...> -- retries while no GPU slot is free (exit 3)
T=$1; LOG=$2; shift 2
for i in $(seq 1 40); do
  gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $LOG; then exit $rc; fi
  sleep 45
done
exit 3
