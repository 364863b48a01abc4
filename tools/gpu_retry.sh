#!/bin/bash
# usage: tools/gpu_retry.sh TIMEOUT LOGFILE 'command'   - waits for a free GPU slot (gpurun exit 3 = nothing was charged), then runs once
T=$1; LOG=$2; CMD=$3
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$CMD" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
