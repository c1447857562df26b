#!/usr/bin/env bash
# development aid (GPU box): SQ counters of the wide-path kernels on the HiFi-like probe set.  usage: bash tools/sq_hifi.sh 3000 40 1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sqh; rm -rf $OUT; mkdir -p $OUT
CTRS="${CTRS:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS}"
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/p -o t -- python $R/tools/hifi_probe2.py "$@" > $OUT/probe.txt 2>&1
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(glob.glob("$OUT/p/*counter_collection.csv")[0])):
    k = r["Kernel_Name"].split("(")[0]
    if not any(t in k for t in ("k_wide", "k_order")): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    d = max(n[(k, c)] for c in v)
    print(k + " | " + " ".join("%s=%.4g" % (c.replace("SQ_", ""), x / d) for c, x in sorted(v.items())) + " | dispatches=%d" % d)
PY
tail -1 $OUT/probe.txt
