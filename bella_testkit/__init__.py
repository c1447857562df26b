"""Test and bench tooling (NOT the product): synthetic read sets, FASTQ helpers, a host-side numpy mirror of the k-mer dictionary
and of the 2-bit packing.  bella_amd/ never imports this package; tests/, bench.py, tools/ and oracle/make_golden.py do."""
