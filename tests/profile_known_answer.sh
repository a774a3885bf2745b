# kernel table of the known-answer run (four-level F-cycle with GMRES + ILU(0) level solvers): rocprofv3 --kernel-trace --stats, no counters
#   bash tests/profile_known_answer.sh r06   -> gpurun_out/r06_known_answer_probe.json, gpurun_out/r06_known_answer_kernel_summary.md
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/tests/perf_probe_known_answer.py all 2> /dev/null | grep '^{' > $OUT/${TAG}_known_answer_probe.json
rm -rf /tmp/pka
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pka -- python $ROOT/tests/perf_probe_known_answer.py fcycle > /tmp/pka.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/pka/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT/${TAG}_known_answer_kernel_summary.md", "w") as o:
    o.write("kernels of python tests/perf_probe_known_answer.py fcycle (testNSSteadyDD, four uniform levels, nonlinear F-cycle to convergence; rocprofv3 --kernel-trace --stats)\n\n")
    o.write("total kernel time %.1f ms in %d launches\n\n| kernel | calls | total ms | avg us | %% |\n|---|---|---|---|---|\n" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
    for r in rows[:25]:
        o.write("| %s | %s | %.2f | %.2f | %.1f |\n" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
