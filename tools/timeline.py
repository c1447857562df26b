#!/usr/bin/env python3
"""Timeline of the last complete overlap pass in a rocprofv3 --kernel-trace CSV (development aid / profiles): every kernel with
start, end, duration; span of the row kernels (first start -> last end) and of the whole pass."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_row_flops" in r["Kernel_Name"]]
first = idx[-1]                                              # k_row_flops is the first kernel of a pass (it clears the control block)
t0 = int(rows[first]["Start_Timestamp"])
rs, re_, end = None, None, None
for r in rows[first:]:
    k = r["Kernel_Name"].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-52s start %9.1f end %9.1f dur %8.1f us grid %9s lds %6s vgpr %s" % (k[-52:], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3,
          r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"]))
    if "k_spgemm_rows" in k:
        rs = s if rs is None else min(rs, s)
        re_ = e if re_ is None else max(re_, e)
    if "copyBuffer" in k and rs is not None:
        end = e
        break
print("row kernels span (first start -> last end): %.1f us ; pass span (k_row_flops -> control block copy): %.1f us" % ((re_ - rs) / 1e3, (end - t0) / 1e3))
