cd /root/repo; mkdir -p gpurun_out/r05
python tests/perf_probe_cluster_phases.py 0 0 2>&1 | grep "ms per assembly"
python bench.py --no-cpu-baseline > gpurun_out/r05/bench_1.json 2> gpurun_out/r05/bench_1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ['value','ms_per_step','assembly_ms','vcycle_ms','prepare_ms','solve_ms']}); print(d['solve']); ra=d['roofline_assembly']; print(ra['first_kernel_ms'], ra['second_pass_ms'], ra['frac'])
PY
python -m pytest tests/test_gpu_fused_assembly.py tests/test_gpu_assembly.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
