# counters of the run kernel of the natural-order sweeps (k_tri_run) over tests/perf_probe_trisweep.py; separate passes, --kernel-trace only
#   bash tests/pmc_trisweep.sh r06   -> gpurun_out/r06_trisweep_pmc_summary.md
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_tri; mkdir -p /tmp/pmc_tri
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_GATE_EN1_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_tri/p$i -- python $ROOT/tests/perf_probe_trisweep.py > /tmp/pmc_tri/log$i.txt 2>&1 || { echo "pass $i ($set) failed"; tail -2 /tmp/pmc_tri/log$i.txt; }
done
python $ROOT/profiles/summarize.py /tmp/pmc_tri /tmp/pmc_tri/all.md > /dev/null
# keep the rows of the sweeps' kernels only (the run holds the whole set-up of the problem)
{ echo "counters of tests/perf_probe_trisweep.py (rocprofv3 --pmc, separate passes, --kernel-trace only), per launch on average; k_tri_run<kind, register slots>: kind 2 = ILU lower, 3 = ILU upper"; echo; grep -E "^\| kernel|^\|---|k_tri_run|k_ilu_" /tmp/pmc_tri/all.md; } > $OUT/${TAG}_trisweep_pmc_summary.md
rm -f $OUT/${TAG}_trisweep_pmc_summary.json
grep "k_tri_run" $OUT/${TAG}_trisweep_pmc_summary.md | head -60
