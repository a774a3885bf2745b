cd /root/repo; mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_ns.py -x -q 2>&1 | tail -3
python tests/perf_probe_ns.py 0.001 > gpurun_out/r05/ns_probe_fused.json 2>/dev/null; tail -1 gpurun_out/r05/ns_probe_fused.json
python tests/perf_probe_ns.py 0.001 vanka_fused=0 > gpurun_out/r05/ns_probe_unfused.json 2>/dev/null; tail -1 gpurun_out/r05/ns_probe_unfused.json
