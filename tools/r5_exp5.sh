#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5g; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQ_WAIT_INST|SQ_INST_CYCLES|SQ_BUSY" | head -40 > $O/counters.txt
cat $O/counters.txt | cut -c1-160
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU"; do
  D=$O/p_$(echo $SET | cut -c1-12 | tr ' ' _); rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $D -o t -- python $R/bench.py --reads 100000 --steps 3 --warmup 1 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab > /dev/null 2>$D.err
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float))
fs = glob.glob("$D/*counter_collection.csv")
if fs:
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_spgemm_rows" in k or "k_order" in k:
            k = k[k.index("k_"):][:40]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items(): print(k, " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
else: print("no counters", open("$D.err").read()[-500:])
PY
done 2>&1 | tee $O/icache.txt
