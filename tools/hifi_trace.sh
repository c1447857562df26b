#!/usr/bin/env bash
# development aid (GPU box): kernel totals and timeline of one overlap pass on the HiFi-like probe set
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/hf; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/hf -o f -- python $R/tools/hifi_probe2.py "$@" > /dev/null 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$R/gpurun_out/hf/f_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_row_flops" in r["Kernel_Name"]]
first=idx[-1]
t0=int(rows[first]["Start_Timestamp"])
agg=collections.defaultdict(lambda:[0,0.0,1e18,0])
for r in rows[first:]:
    k=r["Kernel_Name"].split("(")[0][-56:]
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    a=agg[k]; a[0]+=1; a[1]+=e-s; a[2]=min(a[2],s); a[3]=max(a[3],e)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:16]: print("%-58s n=%4d total %9.1f us  first start %9.1f last end %9.1f"%(k,v[0],v[1],v[2],v[3]))
print("pass span %.1f us"%((int(rows[-1]["End_Timestamp"])-t0)/1e3))
PY
