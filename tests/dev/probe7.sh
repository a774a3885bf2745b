cd /root/repo; mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_fused_assembly.py tests/test_gpu_multigrid.py tests/test_gpu_poisson_mg.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15
python bench.py --no-cpu-baseline > gpurun_out/r05/bench_3.json 2> gpurun_out/r05/bench_3.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ['value','ms_per_step','assembly_ms','vcycle_ms','prepare_ms','solve_ms']}); print(d['solve'])
PY
