#!/usr/bin/env bash
# development aid (GPU box): kernel timeline of one overlap pass on the HiFi-like probe set
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/hf; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/hf -o f -- python $R/tools/hifi_probe2.py "$@" > /dev/null 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$R/gpurun_out/hf/f_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_row_flops" in r["Kernel_Name"]]
first=idx[-1]-1
t0=int(rows[first]["Start_Timestamp"])
for r in rows[first:first+14]:
    k=r["Kernel_Name"].split("(")[0]
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%-50s start %9.1f end %9.1f dur %9.1f us grid %s"%(k[-50:], (s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3, r["Grid_Size_X"]))
PY
