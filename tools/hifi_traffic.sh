#!/usr/bin/env bash
# development aid (GPU box): FETCH_SIZE / WRITE_SIZE (KB, raw counters; gfx950: FETCH_SIZE reads 1/2 of streamed bytes) per kernel of
# the HiFi-like probe's passes.  usage: bash tools/hifi_traffic.sh 10000 40 1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ht_$C; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/ht_$C -o t -- python $R/tools/hifi_probe2.py "$@" > /dev/null 2>&1
done
python - <<PY
import csv, collections, glob
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(glob.glob("/tmp/ht_%s/*counter_collection.csv" % c)[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0][-50:]
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, v in agg.items(): res[k][c] = v[0] * 1024 / v[1]; res[k]["n"] = v[1]
for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) * 2 + kv[1].get("WRITE_SIZE", 0)))[:14]:
    print("%-52s per dispatch: fetch(x2) %8.1f MB  write %8.1f MB   (%d dispatches)" % (k, 2 * v.get("FETCH_SIZE", 0) / 1e6, v.get("WRITE_SIZE", 0) / 1e6, v["n"]))
PY
