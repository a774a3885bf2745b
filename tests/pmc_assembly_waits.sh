cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc2; mkdir -p /tmp/pmc2
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc2/p$i -- python /root/repo/tests/perf_probe_kpad.py 12,1 > /tmp/pmc2/log$i.txt 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pmc2/log$i.txt; }
done
python /root/repo/profiles/summarize.py /tmp/pmc2 | grep "k_elem_q2hex_mfma" | grep -v "^| .k_elem_q2hex_mfma<0, 12>. | 196608 | [0-9]* | [0-9.]* |"
