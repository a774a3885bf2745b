cd /root/repo; mkdir -p gpurun_out/r05
python tests/perf_probe_ns_cycle.py 2>/dev/null | tail -1
python tests/perf_probe_ns_cycle.py vanka_fused=0 2>/dev/null | tail -1
python tests/perf_probe_ns_cycle.py gmres_device=0 2>/dev/null | tail -1
