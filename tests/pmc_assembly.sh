# PMC passes over the fine-level assembly (k_elem_q2hex_mfma, k_row_assemble): one counter set per pass, kernel trace only.
#   bash tests/pmc_assembly.sh [out.md]      (on the GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc; mkdir -p /tmp/pmc
i=0
for set in "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc/p$i -- python /root/repo/tests/perf_probe_kpad.py 12,1 > /tmp/pmc/log$i.txt 2>&1 || echo "pass $i failed"
done
python /root/repo/profiles/summarize.py /tmp/pmc ${1:-/root/repo/gpurun_out/r01b_assembly_pmc_summary.md} | grep "k_elem_q2hex_mfma\|k_row_assemble<27, false>"
