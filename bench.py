#!/usr/bin/env python3
"""bench.py -- FEMuS hot path on MI355X: fine-level Poisson assembly + one V(2,2) Jacobi cycle per step.

Workload (BASELINE.json configs[1]): 3-D Poisson, Q2 (HEX27), coarse 8^3 refined to 64^3, 4-level GMG, V(2,2)
Richardson(2/3)+Jacobi smoother, exact coarse solve, f = 1, u = 0 on the boundary; 64^3 elements PER GPU (weak scaling).

One step = KK->zero / RES->zero / batched element loop / Dirichlet residual rows (the reference's "ASSEMBLY TIME"
phase) followed by one multigrid cycle on the assembled residual (the reference's "Linear-Cycle" phase) with the
hierarchy prepared once before the timed region (Galerkin chain + SetPenalty + smoother/coarse setup = the reference's
"PREPARATION TIME", reported separately as prepare_ms).  Inputs are resident in HBM when the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task description); `roofline` is the fine-level Jacobi-sweep SpMV
kernel, `cpu_baseline` is the oracle's C restatement timed on the host cores of this box (rank 0, N=1 only).  `roofline.traffic` comes
from hardware counters of this run: two short child runs of femus_amd/traffic_probe.py under rocprofv3 --pmc after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6    # vendor FP64 vector peak = FP64 matrix peak on MI355X
# k_cluster_q2hex_sf / k_elem_q2hex_sf, per element, from the PMC passes (profiles/r04_assembly_pmc_summary.md): 361 FMA, 51 MUL and 20 (fused
# path; 2 in the two-pass kernel) ADD FP64 vector instructions on 64 lanes + 15 v_mfma_f64_4x4x4_4b of 512 flops
ELEM_EXECUTED_FLOPS = (361 * 2 + 51 + 20) * 64 + 15 * 512


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--coarse", type=int, default=8, help="coarse box elements per direction (8 -> 64^3 with 4 levels)")
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-known-answer", action="store_true", help="skip the reference's known-answer test (3 s, after the timed region)")
    ap.add_argument("--kernel-reps", type=int, default=50)
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, the environment torchrun would give
    them) and pass rank 0's line through.  More ranks than devices is an error, never a silent 1-GPU run."""
    import socket
    import subprocess
    import femus_amd
    ndev = femus_amd.device_count()
    share = os.environ.get("FEMUS_BENCH_SHARE_GPU") == "1"
    if ndev < args.gpus and not share:
        sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible\n" % (args.gpus, ndev))
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    sys.exit(0 if all(rc == 0 for rc in rcs) else 1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s)\n" % (args.gpus, world))
        sys.exit(2)

    import numpy as np
    import femus_amd
    from femus_amd import dd
    from femus_amd.poisson import PoissonMG

    # setup-time rendezvous (plans, ncclUniqueId, timing maxima) over plain TCP; the data path of a cycle is RCCL
    comm = dd.SocketComm(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")))
    device = local_rank
    if os.environ.get("FEMUS_BENCH_SHARE_GPU") == "1":       # several ranks on one device (development box; host transport only)
        device = local_rank % max(femus_amd.device_count(), 1)
    ctx = femus_amd.Context(device)
    t0 = time.time()
    parallelism = "1 rank per GPU"
    preflight = None
    dist_err = None
    pb = None
    if (world > 1 and os.environ.get("FEMUS_BENCH_DD", "1") != "0") or os.environ.get("FEMUS_BENCH_FORCE_DD") == "1":
        # transports in order of preference: "rccl" (neighbour send/recv over xGMI), then -- when the RCCL preflight fails or hangs on
        # this machine, or when asked for -- "host" (the same plans and kernels, ghost bytes staged through pinned host buffers and
        # exchanged over the setup sockets): still ONE distributed problem with ghost exchange, only slower.  Which runtime that is: femus_amd._lib
        # imports torch FIRST, on purpose (torch's wheel bundles its own HIP runtime and RCCL; two HIP runtimes in one process tear each other down
        # at exit), so the library's ncclSend / ncclRecv / ncclAllReduce bind to torch's bundled librccl.so -- same soname as the /opt/rocm copy it was
        # linked against.  The line reports the mapped paths and ncclGetVersion (`runtime_libraries`), the preflight child runs on the same copy.
        # ("gloo" = the host transport over torch.distributed stays available on request, for a launcher whose only fabric is gloo.)  Independent
        # problems are the last resort and say so.
        want = os.environ.get("FEMUS_BENCH_TRANSPORT", "rccl")
        order = {"rccl": ["rccl", "host"], "gloo": ["gloo"], "host": ["host"]}.get(want, [want])
        for transport in order:
            try:
                halo_comm = None
                if transport == "rccl" and world > 1 and os.environ.get("FEMUS_BENCH_PREFLIGHT", "1") != "0":
                    # a ring exchange + all-reduce through fh_halo_* in a child process first: a RCCL path that fails or HANGS on this
                    # machine must end in the reported fallback below, not in a hung bench
                    from femus_amd import rccl_preflight
                    t_pf = time.time()
                    good, msg = rccl_preflight.run(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                                   int(os.environ.get("MASTER_PORT", "29500")) + 100, device)
                    preflight = [{"rank": int(g[0]), "ok": bool(g[1]), "message": g[2], "seconds": g[3]}
                                 for g in comm.allgather_obj((rank, bool(good), str(msg)[:300], float(time.time() - t_pf)))]
                    if not all(comm.allgather_obj(bool(good))):
                        raise RuntimeError(msg if not good else "RCCL preflight failed on another rank")
                if transport == "gloo":
                    halo_comm = dd.TorchComm.from_env()
                pb = dd.DistributedPoisson(ctx, comm, world, rank, nb=args.coarse, nlevels=args.levels, omega=2. / 3., npre=2, npost=2,
                                           transport="rccl" if transport == "rccl" else "host", halo_comm=halo_comm)
                how = {"rccl": "RCCL neighbour send/recv", "gloo": "the host-staged transport over torch.distributed (gloo)",
                       "host": "the host-staged transport over TCP sockets"}[transport]
                parallelism = ("mesh domain decomposition, box split %dx%dx%d (METIS unavailable), one partition per GPU, ghost DOFs "
                               "exchanged by %s (fh_halo_begin/end, overlapped with the interior rows), the two coarsest levels replicated (all-reduce instead of exchanges)"
                               % (dd.GRIDS[world] + (how,)))
                parallelism += "; partitioner: " + pb.partitioner + " (arbitrary coarse meshes: fh_mesh_partition, DistributedPoisson(coarse_mesh=...))"
                if dist_err is not None:
                    parallelism += "; fell back from rccl: " + dist_err
                err_here = None
            except Exception as e:   # report, never hide: the line says what actually ran
                err_here = "%s: %s" % (type(e).__name__, str(e)[:200])
                pb = None
            # every rank must be on the same path: one failing rank moves all of them to the next transport
            oks = comm.allgather_obj(err_here is None)
            if all(oks):
                break
            if pb is not None:
                pb.destroy()       # communicator, exchange plans and the whole hierarchy of the failed attempt
                pb = None
            dist_err = err_here or "the distributed setup failed on another rank"
    ok = comm.allgather_obj(pb is not None or world == 1)
    if world > 1 and not all(ok):
        pb = None
    if pb is None:
        pb = SerialProblem(ctx, PoissonMG, args)
        if world > 1:
            parallelism = "%d independent single-GPU problems (no halo exchange)%s" % (world, "" if dist_err is None else "; distributed setup failed: " + dist_err)
    setup_s = time.time() - t0
    ndof = pb.ndof_owned
    nel = pb.nel_local
    A = pb.A[-1]

    def barrier():
        ctx.sync()
        comm.barrier()
        ctx.sync()

    def step():
        pb.assemble()                            # KK->zero, RES->zero, batched element loop (all colours)
        pb.set_penalty_top()                     # MGSetLevel: SetPenalty on the assembled operator
        pb.zero_boundary_residuals()             # ZerosBoundaryResiduals
        pb.vcycle()                              # one V(2,2) cycle on RES -> EPSC

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.marker(1)                                # phase markers (one-thread kernels) cut a kernel trace of this command: profiles/summarize.py
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = comm.allreduce_max(time.perf_counter() - t0)
    ctx.marker(2)
    ms_per_step = elapsed / args.steps * 1e3
    ndof_total = int(sum(comm.allgather_obj(int(ndof))))
    value = ndof_total * args.steps / elapsed

    # ---- sub-phase timings with HIP events on the library's compute stream --------------------------------------
    reps = max(3, args.steps)
    barrier()
    ctx.timer_start()
    for _ in range(reps):
        pb.assemble()
    asm_ms = comm.allreduce_max(ctx.timer_stop() / reps)
    asm_path = pb.asm_top.last_path()
    # "one full matrix+RHS assembly into CSR (zero -> element loop -> close -> Dirichlet rows)": with SetPenalty and
    # ZerosBoundaryResiduals -- this is what assembled_dofs_per_sec is quoted on
    barrier()
    ctx.timer_start()
    for _ in range(reps):
        pb.assemble()
        pb.set_penalty_top()
        pb.zero_boundary_residuals()
    asm_full_ms = comm.allreduce_max(ctx.timer_stop() / reps)
    # the element-matrix kernel alone (pass 1 with its stores; asm_debug bit 3 leaves out the row pass): HIP events, same stream
    ctx.set_option("asm_debug", 8)
    pb.assemble()
    barrier()
    ctx.timer_start()
    for _ in range(reps):
        pb.assemble()
    elem_ms = comm.allreduce_max(ctx.timer_stop() / reps)
    ctx.set_option("asm_debug", 0)
    pb.assemble()
    # optional fast path of the library (NOT part of `value`): affine elements through reference matrices instead of quadrature
    ctx.set_option("assemble_affine", 1)
    pb.assemble()
    barrier()
    ctx.timer_start()
    for _ in range(reps):
        pb.assemble()
    asm_affine_ms = comm.allreduce_max(ctx.timer_stop() / reps)
    ctx.set_option("assemble_affine", 0)
    pb.assemble()
    pb.set_penalty_top()
    pb.zero_boundary_residuals()
    barrier()
    ctx.marker(3)
    ctx.timer_start()
    for _ in range(reps):
        pb.vcycle()
    cyc_ms = comm.allreduce_max(ctx.timer_stop() / reps)
    ctx.marker(4)
    # host time to ISSUE one cycle (no synchronisation inside): on one GPU one hipGraphLaunch, on several ranks ~60 launches + the RCCL
    # groups one by one (a distributed cycle is not captured) -- if this exceeds vcycle_ms the host is the bound
    barrier()
    t_issue = time.perf_counter()
    for _ in range(reps):
        pb.vcycle()
    host_issue_ms = comm.allreduce_max((time.perf_counter() - t_issue) / reps * 1e3)
    barrier()
    halo_info = halo_report(ctx, comm, pb) if getattr(pb, "halos", None) else None
    if halo_info is not None:
        halo_info["host_issue_ms_per_cycle"] = host_issue_ms
    # ---- the reference's whole MGsolve (LinearImplicitSystem.cpp:288-411): assemble, prepare (Galerkin chain, SetPenalty, smoother and
    # coarse setup), then the multigrid-preconditioned GMRES down to 1e-10 -- a second headline next to the per-step value
    solve = solve_report(ctx, comm, pb)
    # dominant V-cycle kernel: fine-level fused Jacobi sweep (same kernel family as y=Ax / residual)
    n, ncols = A.m(), A.n()
    ghost_ids = np.arange(n, ncols, dtype=np.int32)
    x, y, dinv = ctx.vector(ncols, n, 0, ghost_ids), ctx.vector(ncols, n, 0, ghost_ids), ctx.vector(n)
    x.upload(np.random.default_rng(12345).uniform(-1, 1, n))
    A.get_diagonal(dinv)
    d = dinv.to_numpy()
    dinv.upload(1.0 / np.where(d == 0, 1.0, d))
    kr = args.kernel_reps
    # sweeps ping-pong between two vectors as the smoother of the cycle does (x <-> x2, fh_mg.hip): every launch reads what the previous one wrote
    for _ in range(3):
        y.jacobi_sweep(pb.RES, x, A, dinv, 2. / 3.)
        x.jacobi_sweep(pb.RES, y, A, dinv, 2. / 3.)
    # timed as the cycle issues them: the kr launches recorded into one hipGraph and replayed (eager launches of this kernel sit ~20 us apart,
    # replayed ones back to back); the eager figure is kept beside it
    ctx.marker(5)
    ctx.timer_start()
    for _ in range(kr // 2):
        y.jacobi_sweep(pb.RES, x, A, dinv, 2. / 3.)
        x.jacobi_sweep(pb.RES, y, A, dinv, 2. / 3.)
    sweep_eager_ms = ctx.timer_stop() / (2 * (kr // 2))
    ctx.marker(6)
    sweep_ms = sweep_eager_ms
    try:
        with ctx.record() as rec:
            for _ in range(kr // 2):
                y.jacobi_sweep(pb.RES, x, A, dinv, 2. / 3.)
                x.jacobi_sweep(pb.RES, y, A, dinv, 2. / 3.)
        rec.graph.launch()
        ctx.marker(7)
        ctx.timer_start()
        rec.graph.launch()
        sweep_ms = ctx.timer_stop() / (2 * (kr // 2))
        ctx.marker(8)
        rec.graph.destroy()
    except Exception as e:             # no recording on this runtime: the eager figure stands
        sys.stderr.write("bench.py: fused sweep not timed under graph replay (%s)\n" % e)
    ctx.timer_start()
    for _ in range(kr):
        y.matrix_mult(x, A)
    spmv_ms = ctx.timer_stop() / kr
    spmv_bytes = A.spmv_algorithmic_bytes()
    sweep_bytes = spmv_bytes + 3 * 8 * n       # + b, dinv, x(own row) per SURVEY 8(d) Jacobi-sweep model
    exp_lo, exp_hi = A.spmv_expected_bytes(3)
    cyc_bytes = pb.mg.cycle_algorithmic_bytes()
    ai = pb.asm_top.info(colors=False)
    fused = pb.asm_top.fused_info()

    out = {
        "metric": "assembled DOFs/sec + V-cycle SpMV GB/s (% HBM peak), 3D Poisson Q2",
        "value": value,
        "unit": "DOF/s (assembly + one V(2,2) cycle per step)",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "3D Poisson Q2 (HEX27, 64-pt Gauss) on %d^3 box per GPU, %d-level GMG V(2,2) Jacobi(2/3), Galerkin coarse "
                        "operators, dense exact coarse solve" % (args.coarse * 2 ** (args.levels - 1), args.levels),
            "dofs_total": ndof_total,
            "dofs_this_gpu": ndof,
            "elements_this_gpu": nel,
            "nnz_fine_this_gpu": A.nnz,
            "parallelism": parallelism,
        },
        "assembled_dofs_per_sec": ndof_total / (asm_full_ms * 1e-3),
        "assembly_ms": asm_full_ms,
        "assembly_element_loop_ms": asm_ms,
        "vcycle_ms": cyc_ms,
        "vcycles_per_sec": (world if "independent" in parallelism else 1.0) / (cyc_ms * 1e-3),
        "vcycle_dofs_per_sec": ndof_total / (cyc_ms * 1e-3),
        "vcycle_GBps": cyc_bytes / cyc_ms / 1e6,
        "vcycle_pct_hbm_peak": cyc_bytes / cyc_ms / 1e6 / HBM_PEAK_GBPS * 100.0,
        "vcycle_spmv_GBps": spmv_bytes / spmv_ms / 1e6,
        "vcycle_spmv_pct_hbm_peak": spmv_bytes / spmv_ms / 1e6 / HBM_PEAK_GBPS * 100.0,
        "prepare_ms": pb.prepare_ms,
        "coarse_solve": dict(zip(("dense_unknowns", "dissection_blocks", "separator", "largest_block"), pb.mg.coarse_info()),
                             what="exact solve of the coarsest level: unknowns coupled to nothing by their diagonal, the rest dense -- dissected into interior blocks "
                                  "inverted beside each other + a separator Schur complement (dissection_blocks 0: one dense inverse)"),
        "solve_ms": solve["solve_ms"],
        "solve": solve,
        "vcycle_host_issue_ms": host_issue_ms,
        "runtime_libraries": femus_amd.loaded_runtimes(),
        "prepare_first_s": pb.prepare_first_s,
        "setup_s": setup_s,
        "roofline": {
            "kernel": "k_spmv_lx<2048,3,true> (fine-level fused Jacobi sweep x+w*Dinv*(b-Ax), LDS-staged x)",
            "bound": "hbm",
            "achieved": sweep_bytes / sweep_ms / 1e6,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": sweep_bytes / sweep_ms / 1e6 / HBM_PEAK_GBPS,
            "frac_is": "replayed (no kernel trace of the cycle in this run)",
            "frac_replayed": sweep_bytes / sweep_ms / 1e6 / HBM_PEAK_GBPS,
            "frac_eager": sweep_bytes / sweep_eager_ms / 1e6 / HBM_PEAK_GBPS,
            "traffic": TRAFFIC["spmv"]["bytes"] if world == 1 else None,
            "traffic_from_profile": TRAFFIC["spmv"]["source"] if world == 1 else None,
            "algorithmic_bytes_per_launch": sweep_bytes,
            "expected_bytes": exp_lo,
            "expected_bytes_no_l2_reuse": exp_hi,
            "expected_bytes_note": "bytes of the arrays the kernel really touches (fh_spmv_expected_bytes): values 8 B + 16-bit local columns 2 B per "
                                   "non-zero, distinct-column lists 4 B each, block descriptors, row pointers, b / D^-1 / x_row / y; x counted once "
                                   "(`expected_bytes`) or once per row block (`..._no_l2_reuse`).  `traffic` (counters, FETCH_SIZE x 2 + WRITE_SIZE; "
                                   "factor calibrated per access width in profiles/r03_fetch_calibration.md) should lie between the two",
            "avg_launch_ms": sweep_ms,
            "avg_launch_ms_eager": sweep_eager_ms,
            "avg_launch_ms_replayed": sweep_ms,
            "avg_launch_note": "avg_launch_ms (and achieved / frac): the fine-level sweeps INSIDE the V(2,2) cycle -- their durations in a rocprofv3 --kernel-trace of femus_amd/traffic_probe.py "
                               "(the bench problem, the captured cycle replayed ten times between two marker kernels), child process of this run; avg_launch_ms_replayed / frac_replayed: "
                               "--kernel-reps launches of the same kernel alone, recorded into one hipGraph and replayed back to back (HIP events); avg_launch_ms_eager / frac_eager: "
                               "the same launches issued one by one from the host",
            "plain_spmv_ms": spmv_ms,
        },
        "optional_affine_fast_path": {
            "note": "fh_set_option(assemble_affine, 1): elements with affine geometry (every element of this box mesh) are assembled "
                    "from nine precomputed reference matrices instead of the 64-point quadrature loop; equal to the quadrature "
                    "result up to summation order (tests/test_gpu_assembly.py).  Off by default and NOT used for `value`, which "
                    "times the reference's algorithm (full quadrature on every element).",
            "assembly_ms": asm_affine_ms,
            "assembled_dofs_per_sec": ndof_total / (asm_affine_ms * 1e-3),
            "step_ms_estimate": ms_per_step - asm_ms + asm_affine_ms,
        },
        "roofline_assembly": {
            "kernel": ("k_cluster_q2hex_sf + k_rows_partial (fused cluster assembly: eight sibling elements per workgroup, element matrices by sum factorisation, "
                       "rows complete inside the cluster written straight into the CSR arrays, the others through the partial-row buffer and the second pass)"
                       if fused["active"] else
                       "k_elem_q2hex_sf + k_row_assemble2_t<27> (two-pass assembly: element matrices by sum factorisation into the element-row buffer, row gather)"),
            "bound": "hbm",
            "achieved": ai["algorithmic_bytes"] / asm_ms / 1e6,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": ai["algorithmic_bytes"] / asm_ms / 1e6 / HBM_PEAK_GBPS,
            "bytes_model": "SURVEY 8(d): per element 27 node ids + 27 coordinates + 27 solution values read, 27 x 27 matrix entries + 27 residual entries "
                           "written = 7 020 B, x %d elements = %d B per assembly, over the time of the assembly's kernels (HIP events)" % (nel, ai["algorithmic_bytes"]),
            "avg_launch_ms": asm_ms,
            "first_kernel_ms": elem_ms,
            "second_pass_ms": asm_ms - elem_ms,
            "fused": fused,
            "path_of_the_timed_assemblies": asm_path,
            "path_note": "assemble_fused = 1 picks per assembly: fused, unless the element-wise Galerkin product asked for the element rows of the previous assembly (then two-pass, which keeps them: the assemblies inside `solve`)",
            "executed_tflops": ELEM_EXECUTED_FLOPS * nel / elem_ms / 1e9,
            "executed_frac_fp64_peak": ELEM_EXECUTED_FLOPS * nel / elem_ms / 1e9 / FP64_VALU_PEAK_TFLOPS,
            "flops_model": "EXECUTED flops per element of the first kernel: (361 FMA x 2 + 51 MUL + 20 ADD) x 64 lanes + 15 MFMA x 512 = %d (instruction counts "
                           "from the PMC passes).  `algorithmic_tflops` prices the reference's full element loop instead (SURVEY 8d, 4.6e5 flop/element): "
                           "above the 78.6 TFLOP/s FP64 peak, i.e. the loop as the reference writes it could not run this fast on this device" % ELEM_EXECUTED_FLOPS,
            "algorithmic_tflops": ai["flops"] / elem_ms / 1e9,
            "traffic": (TRAFFIC["asm"]["elem"] + TRAFFIC["asm"]["rows"]) if world == 1 and TRAFFIC["asm"]["elem"] and TRAFFIC["asm"]["rows"] else None,
            "first_kernel_traffic": TRAFFIC["asm"]["elem"] if world == 1 else None,
            "second_pass_traffic": TRAFFIC["asm"]["rows"] if world == 1 else None,
            "traffic_from_profile": TRAFFIC["asm"]["source"] if world == 1 else None,
        },
    }

    if halo_info is not None:
        out["halo"] = halo_info
    if preflight is not None:
        out["rccl_preflight"] = preflight          # what the RCCL preflight children of every rank said (ok / message / seconds)

    # ---- the reference's own known-answer test through the device path (rank 0, N = 1; after the timed region, not part of `value`) --------
    if rank == 0 and world == 1 and not args.no_known_answer:
        try:
            from femus_amd import known_answer as ka
            out["known_answer"] = ka.run(ctx)
        except Exception as e:      # reported, never hidden
            out["known_answer"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    # ---- CPU baseline: the oracle's C restatement on the host cores (rank 0, N = 1 only) ------------------------
    if rank == 0 and not args.no_live_traffic and os.environ.get("FEMUS_BENCH_LIVE_TRAFFIC", "1") != "0":
        # roofline.traffic from hardware counters of THIS run (two short child runs under rocprofv3 --pmc); when that is not possible the
        # values of the committed counter passes stay in place and `traffic_source` says so
        # (N > 1: the probe runs on rank 0's device while the other ranks wait at the last barrier; it is the one-GPU problem of the same
        # local size -- the owned-rows operator of a rank has the same row blocks plus its ghost columns)
        live, how = live_traffic(args.coarse, args.levels, device=device)
        if live is not None:
            if world > 1:
                how += "; one-GPU problem of the same local size on rank 0's device"
            out["roofline"]["traffic"] = live["spmv"]
            out["roofline"]["traffic_source"] = how
            out["roofline"].pop("traffic_from_profile", None)            # that label names the committed passes the live value has just replaced
            out["roofline_assembly"].pop("traffic_from_profile", None)
            if live.get("sweep_in_cycle_ms"):
                t_in = live["sweep_in_cycle_ms"]
                out["roofline"].update({"avg_launch_ms": t_in, "achieved": sweep_bytes / t_in / 1e6, "frac": sweep_bytes / t_in / 1e6 / HBM_PEAK_GBPS,
                                        "frac_in_cycle": sweep_bytes / t_in / 1e6 / HBM_PEAK_GBPS, "in_cycle_launches": live["sweep_in_cycle_launches"],
                                        "frac_is": "in-cycle: the fine-level fused sweeps inside the replayed V(2,2) cycle (kernel trace of this run's child process)"})
            out["roofline_assembly"]["traffic"] = live["elem"] + live["rows"]
            out["roofline_assembly"]["first_kernel_traffic"] = live["elem"]
            out["roofline_assembly"]["second_pass_traffic"] = live["rows"]
            out["roofline_assembly"]["traffic_over_algorithmic"] = (live["elem"] + live["rows"]) / ai["algorithmic_bytes"]
            out["roofline_assembly"]["traffic_source"] = how
        else:
            out["roofline"]["traffic_source"] = "committed counter passes (%s); live measurement skipped: %s" % (TRAFFIC["spmv"]["source"], how)
    elif world == 1:
        out["roofline"]["traffic_source"] = "committed counter passes (%s)" % TRAFFIC["spmv"]["source"]
    # cpu_baseline is measured on rank 0 at N = 1 only; that run leaves its object in a file on this host and the N > 1 lines of the
    # same box (the driver runs N = 1, 2, 4, 8 back to back) carry it by reference -- per-GPU work is the same at every N
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "femus_amd_cpu_baseline.json")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pb.pb, ndof, nel)
        try:
            json.dump({"workload": out["config"]["workload"], "when": time.time(), "cpu_baseline": out["cpu_baseline"]}, open(cache, "w"))
        except OSError:
            pass
    elif rank == 0 and world > 1 and not args.no_cpu_baseline:
        try:
            c = json.load(open(cache))
            if c["workload"] == out["config"]["workload"]:
                cb = dict(c["cpu_baseline"])
                cb["by_reference"] = "measured by the N=1 run of this bench on this host %.0f s earlier (one GPU's share of the weak-scaled problem)" % (time.time() - c["when"])
                out["cpu_baseline"] = cb
        except (OSError, ValueError, KeyError):
            pass                                   # no N=1 run on this host before this one: the line carries no cpu_baseline

    if rank == 0:
        print(json.dumps(out), flush=True)
    comm.barrier()
    comm.close()


def halo_report(ctx, comm, pb):
    """ghost exchanges of one V-cycle on this rank (counters of fh_halo_stats), and -- from a few cycles run with halo_profile,
    which times every exchange with HIP events and therefore synchronises -- how long the exchanges take and how much of that the
    compute stream really waits for (exposed); the rest is hidden behind the row blocks that read no ghost"""
    for h in pb.halos:
        h.stats(reset=True)
        h.allreduce_count(reset=True)
    pb.vcycle()
    ctx.sync()
    st = [h.stats(reset=True) for h in pb.halos]
    n_allreduce = sum(h.allreduce_count(reset=True) for h in pb.halos)
    ncyc = 3
    for h in pb.halos:
        h.allreduce_ms(reset=True)
    ctx.set_option("halo_profile", 1)
    for _ in range(ncyc):
        pb.vcycle()
    ctx.sync()
    ctx.set_option("halo_profile", 0)
    pr = [h.stats(reset=True) for h in pb.halos]
    ar_ms = sum(h.allreduce_ms(reset=True) for h in pb.halos) / ncyc
    top = pb.A[-1]
    n_int, n_ifc = top.split_info(top.m())
    ex_me, xp_me = sum(p["exchange_ms"] for p in pr) / ncyc, sum(p["exposed_ms"] for p in pr) / ncyc
    ex = comm.allreduce_max(ex_me)
    xp = comm.allreduce_max(xp_me)
    # (the setup sockets carry plain values, lists and tuples only)
    got = comm.allgather_obj((float(ex_me), float(xp_me), float(ar_ms), [float(p["exposed_ms"] / ncyc) for p in pr]))
    by_rank = [{"exchange_ms": g[0], "exposed_ms": g[1], "allreduce_ms": g[2], "exposed_ms_by_level": list(g[3])} for g in got]
    return {
        "by_rank": by_rank,
        "allreduce_ms_per_cycle": max(b["allreduce_ms"] for b in by_rank),
        "exchanges_per_cycle": int(sum(s["updates"] for s in st)),
        "exchanges_per_cycle_by_level": [int(s["updates"]) for s in st],
        "bytes_sent_per_cycle_this_rank": int(sum(s["bytes_sent"] for s in st)),
        "allreduces_per_cycle": int(n_allreduce),
        "exchange_ms_per_cycle": ex,
        "exposed_ms_per_cycle": xp,
        "hidden_ms_per_cycle": max(ex - xp, 0.0),
        "fine_level_row_blocks": {"interior": n_int, "interface": n_ifc},
        "overlap": "rows without ghost columns are multiplied on the compute stream while the exchange runs on the communication "
                   "stream (fh_spmv_ghosted); times are maxima over ranks from HIP events, measured with per-exchange synchronisation",
    }


def solve_report(ctx, comm, pb):
    """assemble + prepare + GMRES(V(2,2)) to 1e-10 relative residual, wall clock with a synchronisation at both ends, and its parts"""
    def timed(fn):
        ctx.sync()
        comm.barrier()
        t = time.perf_counter()
        r = fn()
        ctx.sync()
        return comm.allreduce_max((time.perf_counter() - t) * 1e3), r

    def whole():
        pb.assemble()
        pb.prepare()
        return pb.solve(rtol=1e-10)

    whole()                                          # warm: Krylov workspace, graph
    ctx.marker(9)                                    # phase markers: profiles/summarize.py cuts a kernel trace of this command at them
    t_all, (its, rnorm) = timed(whole)
    ctx.marker(10)
    t_asm, _ = timed(pb.assemble)
    t_prep, _ = timed(pb.prepare)
    ctx.marker(11)
    t_kry, (its2, _) = timed(lambda: pb.solve(rtol=1e-10))
    ctx.marker(12)
    return {"solve_ms": t_all, "assemble_ms": t_asm, "prepare_ms": t_prep, "krylov_ms": t_kry, "gmres_iterations": int(its),
            "final_residual": float(rnorm), "rtol": 1e-10,
            "what": "one MGsolve of the reference (LinearImplicitSystem.cpp:288-411): fine-level assembly, numeric Galerkin chain + SetPenalty + "
                    "smoother / exact coarse factorisation, then GMRES preconditioned by one V(2,2) cycle per iteration from a zero guess"}


class SerialProblem:
    """N = 1 (or N independent problems): PoissonMG with the hierarchy prepared once before the timed steps"""

    def __init__(self, ctx, PoissonMG, args):
        self.ctx = ctx
        pb = PoissonMG(ctx, args.coarse, args.coarse, args.coarse, args.levels, fe="biquadratic", order="seventh",
                       omega=2. / 3., npre=2, npost=2, coarse="galerkin", source_kind=0, params=(1.0,)).init()
        pb.assemble()
        ctx.sync()
        t0 = time.time()
        pb.prepare()
        ctx.sync()
        self.prepare_first_s = time.time() - t0          # includes the one-time symbolic PtAP
        pb.assemble()
        ctx.sync()
        t0 = time.time()
        pb.prepare()                                     # numeric-only re-preparation
        ctx.sync()
        self.prepare_ms = (time.time() - t0) * 1e3
        self.pb = pb
        self.A, self.mg, self.RES = pb.A, pb.mg, pb.RES
        self.ndof_owned, self.nel_local = pb.ndof[-1], pb.meshes[-1].nel
        self.asm_top = pb.asm[-1]
        self._bdc = pb.bdc[-1]

    def assemble(self):
        self.pb.assemble()

    def set_penalty_top(self):
        self.pb.bdc_dev[-1].zero_rows(self.pb.A[-1], 1.0)

    def zero_boundary_residuals(self):
        self.pb.zero_boundary_residuals()

    def vcycle(self):
        self.pb.vcycle()

    def prepare(self):
        self.pb.prepare()

    def solve(self, rtol=1e-10):
        self.pb.EPS.zero()
        return self.pb.mgsolve(outer="gmres", rtol=rtol)


def _newest_profile(suffix):
    """the committed PMC result of the newest round: profiles/rNN*_<suffix>; (parsed json, 'file sha256:...') or (None, None)"""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_" + suffix)))
    if not files:
        return None, None
    raw = open(files[-1], "rb").read()
    return json.loads(raw), "%s sha256:%s" % (os.path.relpath(files[-1], ROOT), hashlib.sha256(raw).hexdigest()[:16])


def live_traffic(coarse, levels, timeout=150, device=0):
    """HBM bytes per launch of the roofline kernels MEASURED IN THIS RUN: femus_amd/traffic_probe.py (the bench problem, the same kernels)
    under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and again under `--pmc WRITE_SIZE` (one counter per pass, no other trace domain, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes), child processes of rank 0 after the timed region.  Units and correction as in
    profiles/README.md: the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (x 2).  Returns
    ({"spmv": bytes, "elem": bytes, "rows": bytes}, how) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None, "this process already runs under a profiler"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="femus_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "femus_amd", "traffic_probe.py"), str(coarse), str(levels)],
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", FEMUS_HIP_DEVICE=str(device)), capture_output=True, text=True,
                               timeout=timeout)
            if "TRAFFIC PROBE DONE" not in r.stdout:
                return None, "the %s pass failed (exit %d)" % (counter, r.returncode)
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    name = row["Kernel_Name"]
                    key = ("spmv" if "k_spmv_lx<2048, 3" in name else "elem" if ("k_elem_q2hex_" in name or "k_cluster_q2hex_" in name) else
                           "rows" if ("k_rows_partial" in name or ("k_row_assemble" in name and "true>" not in name.split("(")[0])) else None)
                    if key:
                        per.setdefault((key, int(row.get("Grid_Size", 0) or 0)), []).append(float(row["Counter_Value"]))
            for key in ("spmv", "elem", "rows"):
                grids = [g for (k, g) in per if k == key]
                if not grids:
                    return None, "no %s launch in the %s pass" % (key, counter)
                v = per[(key, max(grids))]                      # the fine-level launches have the largest grid
                got[(key, counter)] = sum(v) / len(v)
        except subprocess.TimeoutExpired:
            return None, "the %s pass did not finish within %d s" % (counter, timeout)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {k: got[(k, "FETCH_SIZE")] * 1024.0 * 2.0 + got[(k, "WRITE_SIZE")] * 1024.0 for k in ("spmv", "elem", "rows")}
    # third pass, no counters: kernel trace of ten replayed cycles between two marker kernels -> duration of the fine-level sweeps in the cycle
    d = tempfile.mkdtemp(prefix="femus_trace_", dir="/tmp")
    try:
        r = subprocess.run([exe, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "femus_amd", "traffic_probe.py"),
                            str(coarse), str(levels), "cycles"], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", FEMUS_HIP_DEVICE=str(device)),
                           capture_output=True, text=True, timeout=timeout)
        if "TRAFFIC PROBE DONE" in r.stdout:
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                rows += list(csv.DictReader(open(f)))
            t1 = [int(x["Start_Timestamp"]) for x in rows if "k_phase_marker<1>" in x["Kernel_Name"]]
            t2 = [int(x["Start_Timestamp"]) for x in rows if "k_phase_marker<2>" in x["Kernel_Name"]]
            if t1 and t2:
                sw = [x for x in rows if "k_spmv_lx<2048, 3" in x["Kernel_Name"] and t1[-1] < int(x["Start_Timestamp"]) < t2[-1]]
                if sw:
                    grid = lambda x: int(x.get("Grid_Size", x.get("Grid_Size_X", 0)) or 0)        # (the kernel trace names the column Grid_Size_X)
                    gmax = max(grid(x) for x in sw)
                    du = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in sw if grid(x) == gmax]
                    out["sweep_in_cycle_ms"] = sum(du) / len(du)
                    out["sweep_in_cycle_launches"] = len(du)
    except subprocess.TimeoutExpired:
        pass
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out, "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over femus_amd/traffic_probe.py (KiB counters, FETCH_SIZE x 2 on gfx950)"


def _traffic():
    """HBM bytes per launch of the roofline kernels.  Hardware counters cannot be read from inside this process (rocprofv3 wraps the
    command and serialises every dispatch), so `traffic` is NOT measured in this run: it comes from the committed PMC passes of the
    same kernels on the same problem (profiles/README.md: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, gfx950 FETCH_SIZE x 2),
    and the line says which file (with its hash) it was read from."""
    out = {"spmv": {"bytes": None, "source": None}, "asm": {"elem": None, "rows": None, "source": None}}
    d, src = _newest_profile("spmv_traffic.json")
    if d:
        out["spmv"] = {"bytes": d.get("traffic_bytes_per_launch"), "source": src}
    d, src = _newest_profile("assembly_traffic.json")
    if d:
        el = [v for k, v in d.items() if k.startswith("k_elem_q2hex_") or k.startswith("k_cluster_q2hex_")]
        rw = [v for k, v in d.items() if k.startswith("k_row_assemble") or k.startswith("k_rows_partial")]
        out["asm"] = {"elem": el[0]["traffic_bytes_per_launch"] if el else None, "rows": rw[0]["traffic_bytes_per_launch"] if rw else None, "source": src}
    return out


TRAFFIC = _traffic()


def usable_cores():
    """host cores this process may really use: min(affinity mask, cgroup CPU quota)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(pb, ndof, nel):
    """`port`: oracle/oracle_kernels.c (same element loop, same CSR, same V(2,2) Jacobi cycle).  Bounded sample:
    the element loop on a slice of elements (scaled to the full level) + full V-cycles with OpenMP over all cores."""
    import numpy as np
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)     # before libgomp starts: threads = usable cores
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    from oracle import c_kernels as ck
    from oracle import femus_oracle as fo
    import scipy.sparse as sp
    ed, xy, _ = pb.meshes[-1].arrays()
    rp, col = pb.A[-1].pattern()
    val, res = np.zeros(rp[-1]), np.zeros(ndof)
    # the WHOLE fine-level element loop (all elements, scatter included) on every usable core: contiguous element ranges per thread
    # as the reference's ranks own them, atomic adds where ranges share rows (oracle_kernels.c: oc_assemble_poisson_omp); measured,
    # not extrapolated
    ck.set_threads(cores)
    t0 = time.perf_counter()
    ck.assemble_poisson_all_cores(ed, xy, "biquadratic", "hex", (rp, col, val, res))
    t_asm_full = time.perf_counter() - t0
    A = [a.to_scipy() for a in pb.A]
    P = [None] + [p.to_scipy() for p in pb.P[1:]]
    n0 = A[0].shape[0]
    import scipy.linalg as sla
    lu = sla.lu_factor(A[0].toarray())
    cyc = ck.CVcycle(A, P, 2. / 3., 2, 2, coarse_solve=lambda b: sla.lu_solve(lu, b))
    rhs = np.ones(ndof)
    # thread count: the usable cores, unless fewer threads run the fine-level product faster (NUMA / SMT effects)
    xv, yv = np.ones(ndof), np.zeros(ndof)
    best_t, best_n = None, cores
    for nthr in sorted({cores, max(1, cores // 2), max(1, cores // 4), min(cores, 32), min(cores, 16)}, reverse=True):
        ck.set_threads(nthr)
        ck.spmv(cyc.A[-1], xv, yv)
        t0 = time.perf_counter()
        ck.spmv(cyc.A[-1], xv, yv)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nthr
    ck.set_threads(best_n)
    threads = best_n
    cyc.apply(rhs)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        cyc.apply(rhs)
    t_cyc = (time.perf_counter() - t0) / reps
    return {
        "value": ndof / (t_asm_full + t_cyc),
        "unit": "DOF/s (assembly + one V(2,2) cycle per step)",
        "cores": cores,
        "kind": "port",
        "sample": "the whole element loop, all %d elements with the CSR scatter, OpenMP on %d threads (%.2f s, measured); %d full V(2,2) "
                  "cycles with OpenMP on %d threads (best of several thread counts, %.3f s each)" % (nel, cores, t_asm_full, reps, threads, t_cyc),
        "assembly_s": t_asm_full,
        "assembled_dofs_per_sec": ndof / t_asm_full,
        "vcycle_s": t_cyc,
        "vcycle_threads": threads,
        "vcycles_per_sec": 1.0 / t_cyc,
        "petsc": "PETSc not available -- CPU restatement only",
    }


if __name__ == "__main__":
    main()
