#!/usr/bin/env bash
# instruction counts and durations of the row kernel cut after each phase (development aid; 6 = after the B' expansion)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for S in ${STOPS:-6 0 1 2 3 4 -1}; do
  OUT=$R/gpurun_out/sqp/s$S; rm -rf $OUT; mkdir -p $OUT
  BELLA_HIP_STOP_PHASE=$S rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT -o t -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2>&1
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(float)
for r in csv.DictReader(open(glob.glob("$OUT/*counter_collection.csv")[0])):
    if "k_spgemm_rows_lds" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"]) / 5
rows=[r for r in csv.DictReader(open(glob.glob("$OUT/*kernel_trace.csv")[0])) if "k_spgemm_rows_lds" in r["Kernel_Name"]]
n=len(rows)//5
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))//1000 for r in rows[-n:]]
print("stop=$S", "us/tier", d, "sum", sum(d), " ".join("%s=%.4g" % (k.replace("SQ_",""), v) for k, v in sorted(agg.items())))
PY
done
