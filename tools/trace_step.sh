#!/usr/bin/env bash
# timeline of the bella kernels of the last bench step (run on the GPU box)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pf; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pf -o f -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/pf/f_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=[i for i,r in enumerate(rows) if "k_row_flops" in r["Kernel_Name"]][-2]-6
t0=int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    k=r["Kernel_Name"].split("(")[0]
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%-58s start %8.1f  end %8.1f  dur %7.1f us  grid %s lds %s vgpr %s"%(k[-58:], (s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3, r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"]))
PY
