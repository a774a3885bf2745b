#!/bin/bash
# FIRST RUN ON MORE THAN ONE DEVICE -- diagnoses instead of just passing or failing.  On a box with N MI355X (one rank per GPU, RCCL over xGMI):
#   bash tests/scale_first_run.sh [N=8]         -> gpurun_out/scale_first_run/*.log + summary.txt
# 1  the RCCL preflight of every rank (ncclCommInitRank, grouped send/recv ring, the all-reduce forms, then the whole distributed path on a
#    small problem against the single-GPU solve) -- the message of each child;
# 2  BASELINE config 3 at its real size (64^3 elements per rank, box split) over RCCL against the single-GPU solve of the global mesh
#    (what tests/test_gpu_dd.py::test_config3_shape_at_full_size... checks over the host transport), N <= 2 on one GPU's memory for the reference
#    solve, larger N at nb = 4;
# 3  the general (native) partitioner with adaptive levels over RCCL against the single-GPU solve of the same hierarchy;
# 4  bench.py --gpus N: the line carries halo.by_rank (exchange / exposed / all-reduce ms per rank), vcycle_host_issue_ms, rccl_preflight.
N=${1:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/scale_first_run
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
echo "devices visible: $(python -c 'import femus_amd; print(femus_amd.device_count())')" | tee $OUT/summary.txt
# ---- 1: preflight children, one per rank ----
PORT=$((20000 + RANDOM % 20000))
for r in $(seq 0 $((N - 1))); do
  timeout 400 python -m femus_amd.rccl_preflight $r $N 127.0.0.1 $PORT $r > $OUT/preflight_rank$r.log 2>&1 &
done
wait
for r in $(seq 0 $((N - 1))); do echo "preflight rank $r: $(tail -1 $OUT/preflight_rank$r.log)" | tee -a $OUT/summary.txt; done
# ---- 2: config 3 shape over RCCL vs the single-GPU solve ----
NB=8; [ $N -gt 2 ] && NB=4
FEMUS_DD_TRANSPORT=rccl FEMUS_DD_DEVICE_PER_RANK=1 timeout 1500 python tests/perf_probe_amr_dd.py $N $NB 4 2 uniform > $OUT/config3_rccl.log 2>&1
echo "config 3 over RCCL ($N ranks, nb $NB): rc $? $(grep '^{' $OUT/config3_rccl.log | tail -1 | cut -c1-400)" | tee -a $OUT/summary.txt
# ---- 3: general partitioner + adaptive levels over RCCL ----
FEMUS_DD_TRANSPORT=rccl FEMUS_DD_DEVICE_PER_RANK=1 timeout 1500 python tests/perf_probe_amr_dd.py $N 4 4 2 adaptive general > $OUT/general_rccl.log 2>&1
echo "general partition over RCCL: rc $? $(grep '^{' $OUT/general_rccl.log | tail -1 | cut -c1-400)" | tee -a $OUT/summary.txt
# ---- 4: the bench, as the driver launches it ----
for n in 1 2 4 8; do
  [ $n -gt $N ] && break
  if [ $n -eq 1 ]; then timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
  else timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((PORT + 7 + n)) bench.py --gpus $n > $OUT/bench_n$n.json 2> $OUT/bench_n$n.err; fi
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d = json.loads([l for l in open("$OUT/bench_n$n.json") if l.startswith("{")][-1])
    h = d.get("halo") or {}
    print("bench n=$n: value %.4g %s, step %.3f ms, assembly %.3f, cycle %.3f, host issue %.3f; halo exposed %s ms, all-reduce %s ms; %s" % (
        d["value"], d["unit"][:6], d["ms_per_step"], d["assembly_ms"], d["vcycle_ms"], d["vcycle_host_issue_ms"], h.get("exposed_ms_per_cycle"),
        h.get("allreduce_ms_per_cycle"), d["config"]["parallelism"][:160]))
    for p in d.get("rccl_preflight") or []:
        print("   preflight rank %d: ok %s (%.1f s) %s" % (p["rank"], p["ok"], p["seconds"], p["message"][:120]))
except Exception as e:
    print("bench n=$n: no line (%s)" % e)
PY
done
echo "summary in $OUT/summary.txt"
