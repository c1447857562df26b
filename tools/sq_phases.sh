#!/usr/bin/env bash
# instruction counts of the row kernel cut after each phase (development aid)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for S in 0 1 2 3 4 -1; do
  OUT=$R/gpurun_out/sqp/s$S; mkdir -p $OUT
  BELLA_HIP_STOP_PHASE=$S rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT -o t -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(float)
for r in csv.DictReader(open(glob.glob("$OUT/*counter_collection.csv")[0])):
    if "k_spgemm_rows_lds" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"]) / 5
print("stop=$S", " ".join("%s=%.4g" % kv for kv in sorted(agg.items())))
PY
done
