cd /root/repo; mkdir -p gpurun_out/r05
python tests/perf_probe_cluster_phases.py > gpurun_out/r05/phases1.txt 2>&1
python tests/perf_probe_cluster_phases.py 256 > gpurun_out/r05/phases1_norot.txt 2>&1
python -m pytest tests/test_gpu_fused_assembly.py -x -q > gpurun_out/r05/pytest_b.txt 2>&1; tail -3 gpurun_out/r05/pytest_b.txt
head -8 gpurun_out/r05/phases1.txt; head -4 gpurun_out/r05/phases1_norot.txt
