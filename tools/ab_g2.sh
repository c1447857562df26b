#!/usr/bin/env bash
# development aid (GPU box): k_wide_group2 variants (tools/_old/wf_*.so) against the current library
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cp $R/bella_amd/libbella_hip.so /tmp/cur.so
for f in $(ls $R/tools/_old/wf_*.so 2>/dev/null) /tmp/cur.so $(ls $R/tools/_old/wf_*.so 2>/dev/null) /tmp/cur.so; do
  cp $f $R/bella_amd/libbella_hip.so
  rm -rf /tmp/hf; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/hf -o f -- python $R/tools/hifi_probe2.py ${1:-10000} 40 1 > /tmp/hf.txt 2>&1
  echo "$(basename $f): $(python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/hf/f_kernel_trace.csv")))
for k in ("k_wide_group2","k_wide_group1","k_wide_fold_wg<512, 2048"):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if k in r["Kernel_Name"]]
    print(k, [round(x) for x in sorted(d)], end="; ")
PY
)"
  grep overlap /tmp/hf.txt | cut -c1-120
done
cp /tmp/cur.so $R/bella_amd/libbella_hip.so
