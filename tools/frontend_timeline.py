# development aid: kernel-by-kernel timeline (start, duration, gap to the previous end) of the LAST front-end pass in a rocprofv3 kernel trace
# usage: python tools/frontend_timeline.py gpurun_out/fe_trace/s_kernel_trace.csv
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
firsts = [i for i, r in enumerate(rows) if "k_kmers_per_read" in r["Kernel_Name"] or "k_emit_codes" in r["Kernel_Name"]]
# the last pass starts at the last k_emit_codes that follows a gap of other kernels
start = firsts[-1]
for i in reversed(firsts):
    if "k_kmers_per_read" in rows[i]["Kernel_Name"]:
        start = i
        break
ends = [i for i, r in enumerate(rows) if "k_layout_compact" in r["Kernel_Name"]]
end = ends[-1]
t0 = int(rows[start]["Start_Timestamp"])
prev = t0
busy = 0.0
for r in rows[start:end + 1]:
    s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    gap = (s - prev) / 1e3
    busy += (e - s) / 1e3
    print("%9.1f  %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"].split("(")[0][-90:]))
    prev = max(prev, e)
print("span %.1f us, kernel time %.1f us" % ((prev - t0) / 1e3, busy))
