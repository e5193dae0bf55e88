#!/bin/bash
# this container only: run a command on the GPU box through gpurun, retrying while no slot is free (exit code 3)
# usage: tools/_gpu.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
