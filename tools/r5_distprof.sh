#!/bin/bash
# kernel trace of tools/dist_layout_probe.py in both formations (N contexts on one GPU); per-kernel totals into gpurun_out/r5dist
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5dist
for n in 2 4 8; do PROBE_MODES=0,1,0,1,0,1,0,1 timeout 400 python $R/tools/dist_layout_probe.py $n 2>&1 | grep "^N=\|pairs"; done > $R/gpurun_out/r5dist/probe.txt
for m in 0 1; do
  PROBE_MODES=$m,$m,$m,$m timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp$m -o dp -- python $R/tools/dist_layout_probe.py 8 > $R/gpurun_out/r5dist/run$m.txt 2>&1
  f=$(find /tmp/dp$m -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/r5dist/kernel_stats_mode$m.csv
done
cat $R/gpurun_out/r5dist/probe.txt
