# PMC passes over the fused cluster assembly (k_cluster_q2hex_sf + k_rows_partial): one counter set per pass, kernel trace only.
#   bash tests/pmc_fused.sh [asm_debug] [out.md]      (on the GPU box)
DBG=${1:-0}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${2:-$ROOT/gpurun_out/fused_pmc_summary_dbg$DBG.md}
case $OUT in /*) ;; *) OUT=$PWD/$OUT ;; esac
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcfu; mkdir -p /tmp/pmcfu
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcfu/p$i -- python $ROOT/tests/perf_probe_fused_loop.py $DBG > /tmp/pmcfu/log$i.txt 2>&1 || echo "pass $i failed"
done
python $ROOT/profiles/summarize.py /tmp/pmcfu $OUT | grep "k_cluster_q2hex_sf\|k_rows_partial"
