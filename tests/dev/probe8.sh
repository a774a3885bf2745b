cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof8
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8 -- python $R/tests/perf_probe_prepare.py 8 128 > $R/gpurun_out/r05/prep_probe.txt 2>&1
python $R/profiles/summarize.py /tmp/prof8 $R/gpurun_out/r05/prepare_kernel_summary.md > /dev/null
head -30 $R/gpurun_out/r05/prepare_kernel_summary.md; tail -5 $R/gpurun_out/r05/prep_probe.txt
