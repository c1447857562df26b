#!/usr/bin/env bash
# development aid (this container): gpurun with retries while the pod's GPU slots are busy.  usage: tools/gpu.sh TIMEOUT 'command'
T=$1; shift
for i in $(seq 1 20); do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit $rc
done
echo "$out"; exit 3
