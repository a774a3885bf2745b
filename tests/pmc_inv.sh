# PMC passes over the preparation (dense coarse inverse): where the single-workgroup block inversion k_inv_first spends its cycles.
#   bash tests/pmc_inv.sh [out.md]      (on the GPU box)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/inv_pmc_summary.md}
case $OUT in /*) ;; *) OUT=$PWD/$OUT ;; esac
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
rm -rf /tmp/pmcinv; mkdir -p /tmp/pmcinv
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcinv/p$i -- python $ROOT/tests/perf_probe_prepare.py 8 128 > /tmp/pmcinv/log$i.txt 2>&1 || echo "pass $i failed"
done
python $ROOT/profiles/summarize.py /tmp/pmcinv $OUT | grep "k_inv_first"
