#!/usr/bin/env bash
# development aid (GPU box): the native command line on a synthetic PacBio-like FASTQ, phase by phase.  usage: bash tools/cli_probe.sh 100000 [flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
N=${1:-100000}; shift || true
D=/tmp/cli_probe; rm -rf $D; mkdir -p $D
python -c "
import sys; sys.path.insert(0, '$R')
from bella_testkit import synth
rs = synth.make_reads_fast($N, read_len=10000, coverage=30.0, err=0.15, seed=1)
synth.write_fastq('$D/reads.fastq', rs)
open('$D/in.txt', 'w').write('$D/reads.fastq\n')
" || exit 1
cd $D; ls -la
for rep in 1 2; do
  T0=$(date +%s.%N)
  $R/bella_amd/bin/bella-hip -f in.txt -o out "$@" 2> err.txt | tr '\n' ' '; echo
  T1=$(date +%s.%N)
  python -c "print(\"wall %.2f s\" % ($T1 - $T0))"
  grep -E "Time" err.txt | sed 's/INFO:\tbella_hip_main.cpp//' | tr '\n' ';'; echo
done
