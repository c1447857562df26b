#!/usr/bin/env bash
# round-5 A/B on the GPU box: every tools/_old/wf_*.so on the HiFi-like probe (10k reads, -u 40, syncmers)
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp bella_amd/libbella_hip.so /tmp/cur.so
for rep in $(seq 1 ${REPS:-2}); do
  for f in tools/_old/wf_*.so; do cp $f bella_amd/libbella_hip.so; echo "$(basename $f): $(python tools/hifi_probe2.py 10000 ${UPPER:-40} 1 2>&1 | tail -1 | cut -c1-140)"; done
done
cp /tmp/cur.so bella_amd/libbella_hip.so
