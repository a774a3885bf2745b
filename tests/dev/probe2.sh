cd /root/repo; mkdir -p gpurun_out/r05
for d in 0 256 512 768; do python tests/perf_probe_cluster_phases.py $d > gpurun_out/r05/phases2_$d.txt 2>&1; grep "ms per assembly" gpurun_out/r05/phases2_$d.txt; done
python -m pytest tests/test_gpu_fused_assembly.py tests/test_gpu_assembly.py -x -q > gpurun_out/r05/pytest_c.txt 2>&1; tail -3 gpurun_out/r05/pytest_c.txt
