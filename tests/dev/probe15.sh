#!/bin/bash
# kernel-level view of the config-4 cycle and linear solve
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p15 -o x -- python $R/tests/perf_probe_ns_cycle.py > /tmp/p15.log 2>&1
tail -1 /tmp/p15.log | cut -c1-300
python - <<'PY'
import csv
rows=list(csv.DictReader(open("/tmp/p15/x_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last linear solve: take the last 34*N kernels window by time: find last 80 ms
t_end=int(rows[-1]["End_Timestamp"])
win=[r for r in rows if int(r["Start_Timestamp"])>t_end-78e6]
import collections
agg=collections.OrderedDict()
busy=0
for r in win:
    d=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); busy+=d
    k=(r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""))
    a=agg.setdefault(k,[0,0]); a[0]+=1; a[1]+=d
span=int(win[-1]["End_Timestamp"])-int(win[0]["Start_Timestamp"])
print("window: %d kernels, span %.2f ms, busy %.2f ms" % (len(win), span/1e6, busy/1e6))
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:40]:
    print("%-62s grid %8s calls %6d total %8.3f ms avg %7.2f us" % (k[0],k[1],a[0],a[1]/1e6,a[1]/a[0]/1e3))
PY
