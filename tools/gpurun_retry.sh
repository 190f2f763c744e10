#!/bin/bash
# gpurun with retries while no GPU slot is free (exit code 3: nothing charged): tools/gpurun_retry.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
