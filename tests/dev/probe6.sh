cd /root/repo; mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_poisson_mg.py tests/test_gpu_dd.py tests/test_gpu_amr.py -x -q 2>&1 | tail -5
python bench.py --no-cpu-baseline > gpurun_out/r05/bench_2.json 2> gpurun_out/r05/bench_2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ['value','ms_per_step','assembly_ms','vcycle_ms','prepare_ms','solve_ms']}); print(d['solve'])
PY
