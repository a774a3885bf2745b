#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0 4096 86016 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p14_$d -o x -- python $R/tests/perf_probe_fused_loop.py $d 5 > /tmp/p14_$d.log 2>&1
  grep "ms per assembly" /tmp/p14_$d.log
  grep -E "k_rows_partial|k_cluster_q2hex" /tmp/p14_$d/x_kernel_stats.csv | cut -c1-60,200-400 | head -3
  python - <<PY
import csv
for r in csv.DictReader(open("/tmp/p14_$d/x_kernel_stats.csv")):
    if "rows_partial" in r["Name"] or "k_cluster_q2hex" in r["Name"]: print(r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3)
PY
done
