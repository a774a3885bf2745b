#!/bin/bash
# kernel-level view of the adaptive hierarchy's set-up (config 5)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/amrprof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/amrprof -o amr -- python $R/tests/perf_probe_amr.py 8 reference > $R/gpurun_out/amrprof/log.txt 2>&1
ls -R $R/gpurun_out/amrprof | head
python - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/amrprof/**/*kernel_stats.csv", recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:30]:
    print("%-70s calls %6s total %9.3f ms avg %9.3f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
