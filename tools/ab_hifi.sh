#!/usr/bin/env bash
# development aid (GPU box): the HiFi-like probe with every library build under tools/_old/wf_*.so, then the current one (each twice);
# median k_wide_fold_wg time per pass
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cp $R/bella_amd/libbella_hip.so /tmp/cur.so
for f in $(ls $R/tools/_old/wf_*.so 2>/dev/null) /tmp/cur.so $(ls $R/tools/_old/wf_*.so 2>/dev/null) /tmp/cur.so; do
  cp $f $R/bella_amd/libbella_hip.so
  rm -rf /tmp/hf; rocprofv3 --kernel-trace --output-format csv -d /tmp/hf -o f -- python $R/tools/hifi_probe2.py ${1:-3000} ${2:-40} ${3:-1} > /tmp/hf.txt 2>&1
  echo "$(basename $f): $(python - <<PY
import csv
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open("/tmp/hf/f_kernel_trace.csv")) if "k_wide_fold_wg" in r["Kernel_Name"]]
print("fold_wg us", sorted(d))
PY
)"
done
cp /tmp/cur.so $R/bella_amd/libbella_hip.so
