# Profiles of one round on the GPU box: bench line, rocprofv3 kernel trace of the bench command, PMC passes (HBM traffic of the fine-level
# SpMV sweep and of the two assembly kernels, instruction / LDS / wait counters of the element kernel).  Counter passes use
# --kernel-trace only (never sys / hip / memory traces), one counter set per pass.
#   bash tests/profile_round.sh r03      -> gpurun_out/r03/*   (copy the summaries worth keeping into profiles/)
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
rm -rf /tmp/prof; mkdir -p /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic > $OUT/${TAG}_bench_line_under_rocprof.json 2> $OUT/rocprof_bench.err
python $ROOT/profiles/summarize.py /tmp/prof/bench $OUT/${TAG}_bench_kernel_summary.md > /dev/null
cp $(find /tmp/prof/bench -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null
# ---- SpMV: HBM traffic of the fused Jacobi sweep (fine level) ----
rm -rf /tmp/pmc_spmv; mkdir -p /tmp/pmc_spmv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_spmv/p$i -- python $ROOT/tests/perf_probe_spmv.py 4 "3,2048,32,0" 3 > /tmp/pmc_spmv/log$i.txt 2>&1 || echo "spmv pass $i failed"
done
python $ROOT/profiles/summarize.py /tmp/pmc_spmv $OUT/${TAG}_spmv_pmc_summary.md > /dev/null
# ---- assembly: element kernel + row pass ----
rm -rf /tmp/pmc; mkdir -p /tmp/pmc
i=0
for set in "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc/p$i -- python $ROOT/tests/perf_probe_kpad.py ${ASM_SPEC:-12,1} > /tmp/pmc/log$i.txt 2>&1 || echo "assembly pass $i failed"
done
python $ROOT/profiles/summarize.py /tmp/pmc $OUT/${TAG}_assembly_pmc_summary.md > /dev/null
# ---- preparation: kernels of one fh_mg_setup re-preparation ----
rm -rf /tmp/prof/prep
PYTHONPATH=$ROOT timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/prep -- python $ROOT/tests/perf_probe_prepare.py 8 128 > /tmp/prof/prep.log 2>&1 || echo "prepare trace failed"
python $ROOT/profiles/summarize.py /tmp/prof/prep $OUT/${TAG}_prepare_kernel_summary.md > /dev/null
tail -c 600 $OUT/${TAG}_bench_line.json
