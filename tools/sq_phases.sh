#!/usr/bin/env bash
# development aid (GPU box, profiling build of tools/prof_build.sh): SQ instruction counts and durations of the row kernels cut after
# each phase (BELLA_DEV_STOP=n: 0 expand, 1 gather+insert, 2 count-scan, 3 scatter+singles, 4 rank/overlay, 5 parents, 6 walks,
# 7 emit = -1 whole).  Differences between consecutive cuts = the phase.  usage: BENCH_ARGS="--reads 100000" bash tools/sq_phases.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/bella_amd/libbella_hip.so /tmp/prod.so; cp $R/tools/_old/libbella_prof.so $R/bella_amd/libbella_hip.so
for S in ${STOPS:-0 1 2 3 4 5 6 -1}; do
  OUT=$R/gpurun_out/sqp/s$S; rm -rf $OUT; mkdir -p $OUT
  BELLA_DEV_STOP=$S rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-xdrop --no-10k --no-dropin --no-hifi ${BENCH_ARGS:---reads 10000} > /dev/null 2>&1
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(glob.glob("$OUT/*counter_collection.csv")[0])):
    k = r["Kernel_Name"]
    if "k_spgemm_rows_lds" in k:
        k = k[k.index("<"):k.index(">") + 1]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("$OUT/*kernel_trace.csv")[0])):
    k = r["Kernel_Name"]
    if "k_spgemm_rows_lds" in k: dur[k[k.index("<"):k.index(">") + 1]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    n = max(cnt[(k, c)] for c in v)
    print("stop=%-2s %-18s us %8.1f " % ("$S", k, sorted(dur[k])[len(dur[k]) // 2]) + " ".join("%s=%.4g" % (c.replace("SQ_", ""), x / n) for c, x in sorted(v.items())))
PY
done
cp /tmp/prod.so $R/bella_amd/libbella_hip.so
