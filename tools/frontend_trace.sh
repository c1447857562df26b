#!/usr/bin/env bash
# development aid (GPU box): per-kernel time of the front end (k-mer counting + assembly + device layout) at N reads under rocprofv3
# usage: bash tools/frontend_trace.sh 100000 > gpurun_out/frontend.txt
N=${1:-100000}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/fe_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/fe_probe.py <<PY
import sys, time
sys.path.insert(0, "$R")
from bella_amd import Engine, BellaPars
from bella_testkit import synth
rs = synth.make_reads_fast($N, read_len=10000, coverage=30.0, err=0.15, seed=1)
e = Engine(0)
e.set_reads(rs)
for it in range(2):
    nk, nt, nd = e.count_kmers(17, 2, 8)
    kc = e.timings().kcount_ms
    e.assemble_counted()
    tm = e.timings()
    print("pass %d: kcount %.2f ms, assemble %.2f ms (rows %.2f, layout %.2f)" % (it, kc, tm.assemble_ms, tm.rows_ms, tm.layout_ms), flush=True)
e.overlap(BellaPars(skipAlignment=True))
print("overlap device ms", e.timings().overlap_total_ms)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python /tmp/fe_probe.py 2>/dev/null | grep "pass\|overlap"
python $R/tools/summarize_rocprof.py $OUT/s_kernel_stats.csv 40
