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
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --no-known-answer > $OUT/${TAG}_bench_line_under_rocprof.json 2> $OUT/rocprof_bench.err
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
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc/p$i -- python $ROOT/tests/perf_probe_fused_loop.py 0 > /tmp/pmc/log$i.txt 2>&1 || echo "assembly pass $i failed"
done
python $ROOT/profiles/summarize.py /tmp/pmc $OUT/${TAG}_assembly_pmc_summary.md > /dev/null
# the two-pass path beside it (option assemble_fused 0): traffic of k_elem_q2hex_sf + k_row_assemble2_t for the comparison in DESIGN
rm -rf /tmp/pmc2; mkdir -p /tmp/pmc2
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  FEMUS_FUSED=0 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc2/p$i -- python $ROOT/tests/perf_probe_fused_loop.py 0 > /tmp/pmc2/log$i.txt 2>&1 || echo "two-pass pass $i failed"
done
python $ROOT/profiles/summarize.py /tmp/pmc2 $OUT/${TAG}_assembly_two_pass_pmc_summary.md > /dev/null
# per-launch HBM traffic of the fine-level launches: (2 FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction, profiles/r03_fetch_calibration.md)
python - <<PY
import json
def traffic(path, names, out, src):
    d = json.load(open(path))
    res = {}
    for key, cs in d.items():
        k, g = key.rsplit("@", 1)
        if any(k.startswith(n) for n in names) and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            if k not in res or int(g) > res[k][0]:
                res[k] = (int(g), {"grid": int(g), "FETCH_SIZE_KB": cs["FETCH_SIZE"], "WRITE_SIZE_KB": cs["WRITE_SIZE"],
                                   "traffic_bytes_per_launch": (2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0})
    o = {k: v[1] for k, v in res.items()}
    o["total_bytes_per_assembly"] = sum(v["traffic_bytes_per_launch"] for v in o.values())
    o["correction"] = "gfx950: FETCH_SIZE x 2 for every access width (profiles/r03_fetch_calibration.md); WRITE_SIZE uncorrected"
    o["source"] = src
    json.dump(o, open(out, "w"), indent=1)
try:
    traffic("$OUT/${TAG}_assembly_pmc_summary.json", ["k_cluster_q2hex_sf", "k_rows_partial"], "$OUT/${TAG}_assembly_traffic.json",
            "bash tests/profile_round.sh $TAG (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tests/perf_probe_fused_loop.py 0)")
    traffic("$OUT/${TAG}_assembly_two_pass_pmc_summary.json", ["k_elem_q2hex_sf", "k_row_assemble2_t"], "$OUT/${TAG}_assembly_two_pass_traffic.json",
            "the same with FEMUS_FUSED=0 (option assemble_fused 0)")
    d = json.load(open("$OUT/${TAG}_spmv_pmc_summary.json"))
    key = max((k for k in d if k.startswith("k_spmv_lx<2048, 3") and "FETCH_SIZE" in d[k]), key=lambda k: int(k.rsplit("@", 1)[1]))
    json.dump({"kernel": key, "FETCH_SIZE_KB": d[key]["FETCH_SIZE"], "WRITE_SIZE_KB": d[key]["WRITE_SIZE"],
               "traffic_bytes_per_launch": (2 * d[key]["FETCH_SIZE"] + d[key]["WRITE_SIZE"]) * 1024.0,
               "source": "bash tests/profile_round.sh $TAG, tests/perf_probe_spmv.py"}, open("$OUT/${TAG}_spmv_traffic.json", "w"), indent=1)
except Exception as e:
    print("traffic files:", e)
PY
# ---- preparation: kernels of one fh_mg_setup re-preparation ----
rm -rf /tmp/prof/prep
PYTHONPATH=$ROOT timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/prep -- python $ROOT/tests/perf_probe_prepare.py 8 128 > /tmp/prof/prep.log 2>&1 || echo "prepare trace failed"
python $ROOT/profiles/summarize.py /tmp/prof/prep $OUT/${TAG}_prepare_kernel_summary.md > /dev/null
# ---- config 5 on one GPU (first assembly / preparation included) and the sparse exact solve ----
python $ROOT/tests/perf_probe_amr.py 8 2> /dev/null | grep '^{' | head -1 > $OUT/${TAG}_amr_probe.json
python $ROOT/tests/perf_probe_direct.py 1 2 3 2> /dev/null | grep '^n ' > $OUT/${TAG}_direct_probe.txt
tail -c 600 $OUT/${TAG}_bench_line.json
# ---- config 4 (cavity, Taylor-Hood, 80 x 80, nu = 0.001 by continuation): F-cycle Newton, cycle and linear solve; round 5: block smoother with a colour per launch;
#      round 6: the cycles stop at the 40 x 40 level (FEMUS_NS_COARSE_LEVEL=2: exact solve there), beside the full hierarchy ----
FEMUS_NS_COARSE_LEVEL=2 python $ROOT/tests/perf_probe_ns.py 0.001 2> /dev/null | grep '^{' | tail -1 > $OUT/${TAG}_ns_probe.json
FEMUS_NS_COARSE_LEVEL=0 python $ROOT/tests/perf_probe_ns.py 0.001 2> /dev/null | grep '^{' | tail -1 > $OUT/${TAG}_ns_probe_full_hierarchy.json
python $ROOT/tests/perf_probe_ns_cycle.py 2> /dev/null | grep '^{' | tail -1 > $OUT/${TAG}_ns_cycle_probe.json
python $ROOT/tests/perf_probe_ns_cycle.py vanka_fused=0 gmres_device=0 2> /dev/null | grep '^{' | tail -1 > $OUT/${TAG}_ns_cycle_probe_round4_options.json
# ---- where the waves of the fused cluster kernel and of the macro-row Galerkin kernel spend their cycles (shader-clock stamps, asm_debug bit 7) ----
python $ROOT/tests/perf_probe_cluster_phases.py 0 > $OUT/${TAG}_cluster_phase_stamps.txt 2>&1
FEMUS_CARRY=0 python $ROOT/tests/perf_probe_cluster_phases.py 0 > $OUT/${TAG}_cluster_phase_stamps_no_carried_rows.txt 2>&1
python $ROOT/tests/perf_probe_carry.py 4 2> /dev/null | tail -1 > $OUT/${TAG}_carry_probe.json
# the assembly without carried rows (assemble_carry 0: round 5's plan) beside it: traffic of the two kernels
rm -rf /tmp/pmc3; mkdir -p /tmp/pmc3
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  FEMUS_HIP_CARRY=0 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc3/p$i -- python $ROOT/tests/perf_probe_fused_loop.py 0 > /tmp/pmc3/log$i.txt 2>&1 || echo "no-carry pass $i failed"
done
python $ROOT/profiles/summarize.py /tmp/pmc3 $OUT/${TAG}_assembly_no_carried_rows_pmc_summary.md > /dev/null
bash $ROOT/tests/profile_known_answer.sh $TAG
bash $ROOT/tests/pmc_trisweep.sh $TAG > /dev/null 2>&1        # counters of the sweeps' run kernel -> ${TAG}_trisweep_pmc_summary.md
python $ROOT/tests/perf_probe_trisweep.py 2> /dev/null | grep cycle_ms > $OUT/${TAG}_trisweep_probe.json
python $ROOT/tests/dev/probe12.py 2>&1 | grep -A11 "k_galerkin_macro phase" > $OUT/${TAG}_galerkin_macro_phase_stamps.txt
# ---- set-up of the bench problem stage by stage (levels refined on the device, round 5) and the same with the host loops ----
python $ROOT/tests/perf_probe_setup.py --device 2> /dev/null | sed -n '/^{/,$p' > $OUT/${TAG}_setup_probe.json
python $ROOT/tests/perf_probe_setup.py 2> /dev/null | sed -n '/^{/,$p' > $OUT/${TAG}_setup_probe_host_refinement.json
python $ROOT/tests/perf_probe_direct_general.py > $OUT/${TAG}_direct_general_probe.txt 2>&1
python $ROOT/tests/perf_probe_shipped_inputs.py 2> /dev/null | sed -n '/^{/,$p' > $OUT/${TAG}_shipped_inputs_probe.json
