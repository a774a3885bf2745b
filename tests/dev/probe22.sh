#!/bin/bash
# kernel-level view of the known-answer problem through the multigrid path (GMRES + ILU(0) level solvers)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- bash -c "cd $R && python tests/dev/probe20.py ilu 4" > /tmp/pk.log 2>&1
grep -v "^W2026" /tmp/pk.log | tail -6 | cut -c1-200
python - <<'PY'
import csv, glob
f=glob.glob("/tmp/pk/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f s, launches %d" % (tot/1e9, sum(int(r["Calls"]) for r in rows)))
for r in rows[:14]: print("%-60s calls %8s total %8.1f ms avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
