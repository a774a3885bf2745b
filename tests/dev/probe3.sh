cd /root/repo; mkdir -p gpurun_out/r05
for d in 0 1024 256; do python tests/perf_probe_cluster_phases.py $d > gpurun_out/r05/phases3_$d.txt 2>&1; grep "ms per assembly" gpurun_out/r05/phases3_$d.txt; done
python -m pytest tests/test_gpu_fused_assembly.py -x -q > gpurun_out/r05/pytest_d.txt 2>&1; tail -3 gpurun_out/r05/pytest_d.txt
