#!/bin/bash
# gpurun, retried while no GPU slot is free (exit code 3: nothing ran, nothing was charged)
for i in $(seq 1 15); do
  /usr/local/graft/bin/gpurun "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
