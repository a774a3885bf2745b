"""Dev helper: timeline of the LAST preparation in a rocprofv3 --kernel-trace csv (start / end relative to its first kernel, queue, name).
usage: trace_prepare_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last k_galerkin_mfma of the finest product starts the last preparation's chain
last = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_galerkin") or "k_galerkin" in r["Kernel_Name"])
i0 = last
while i0 > 0 and int(rows[last]["Start_Timestamp"]) - int(rows[i0 - 1]["Start_Timestamp"]) < 3_000_000:
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if s > 8000:
        break
    print("%9.1f %9.1f  q%-3s %s" % (s, e, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
