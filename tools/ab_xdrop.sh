#!/usr/bin/env bash
# development aid (GPU box): X-drop variants with every tools/_old/wf_*.so and the current library.  usage: bash tools/ab_xdrop.sh 10000 "0 1"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp bella_amd/libbella_hip.so /tmp/cur.so
for f in $(ls tools/_old/wf_*.so 2>/dev/null) /tmp/cur.so; do cp $f bella_amd/libbella_hip.so; echo "== $(basename $f)"; python tools/xdrop_probe.py ${1:-10000} ${2:-0 1} 2>&1 | grep variant; done
cp /tmp/cur.so bella_amd/libbella_hip.so
