#!/usr/bin/env bash
# development aid (GPU box): reservation size / counting budget of the native command line.  usage: bash tools/cli_probe2.sh 100000
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
N=${1:-100000}
D=/tmp/cli_probe; rm -rf $D; mkdir -p $D
python -c "
import sys; sys.path.insert(0, '$R')
from bella_testkit import synth
rs = synth.make_reads_fast($N, read_len=10000, coverage=30.0, err=0.15, seed=1)
synth.write_fastq('$D/reads.fastq', rs)
open('$D/in.txt', 'w').write('$D/reads.fastq\n')
" || exit 1
cd $D
run() {
  echo "== $*"
  for rep in 1 2; do
    env "$@" $R/bella_amd/bin/bella-hip -f in.txt -o out --skip-alignment 2> err.txt | tr '\n' ' '; echo
    grep -E "ContextAndReservationTime|KmerCountingTime|SparseMatrixCreationTime|OverlapTime" err.txt | sed 's/INFO:\tbella_hip_main.cpp//' | tr '\n' ';'; echo
  done
}
run X=1
run BELLA_HIP_NO_RESERVE=1
run BELLA_HIP_RESERVE_BYTES_PER_BASE=24 BELLA_HIP_KCOUNT_BUDGET=536870912
run BELLA_HIP_RESERVE_BYTES_PER_BASE=16 BELLA_HIP_KCOUNT_BUDGET=268435456
run BELLA_HIP_NO_RESERVE=1 BELLA_HIP_KCOUNT_BUDGET=268435456
