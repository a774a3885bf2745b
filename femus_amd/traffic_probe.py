"""The launches whose HBM traffic `bench.py` reports: the bench problem (3-D Poisson Q2, 8^3 -> 64^3, assembled fine-level operator) and a
few launches of the fine-level fused Jacobi sweep and of the assembly.  bench.py runs this file under `rocprofv3 --pmc <counter>
--kernel-trace` (one counter per pass, as MI355X_MICROARCH.md prescribes) and reads the per-launch counter values of the named kernels.
No checks, no oracle: it only has to launch the same kernels on the same data layout as the bench.   python femus_amd/traffic_probe.py [coarse] [levels] [cycles]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import femus_amd
    from femus_amd.poisson import PoissonMG
    coarse = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ctx = femus_amd.Context(int(os.environ.get("FEMUS_HIP_DEVICE", "0")))
    pb = PoissonMG(ctx, coarse, coarse, coarse, levels, fe="biquadratic", order="seventh", omega=2. / 3., npre=2, npost=2, coarse="galerkin",
                   source_kind=0, params=(1.0,)).init()
    for _ in range(4):
        pb.assemble()
    A = pb.A[-1]
    n = A.m()
    x, y, dinv = ctx.vector(n), ctx.vector(n), ctx.vector(n)
    x.upload(np.random.default_rng(12345).uniform(-1, 1, n))
    A.get_diagonal(dinv)
    d = dinv.to_numpy()
    dinv.upload(1.0 / np.where(d == 0, 1.0, d))
    for _ in range(6):
        y.jacobi_sweep(pb.RES, x, A, dinv, 2. / 3.)
    # the V(2,2) cycle as the bench runs it (hierarchy prepared, cycle replayed from its captured graph): markers 1 / 2 around the cycles let
    # bench.py take the duration of the fine-level sweeps INSIDE the cycle from a kernel trace of this file
    if len(sys.argv) > 3 and sys.argv[3] == "cycles":
        pb.prepare()
        pb.assemble()
        pb.bdc_dev[-1].zero_rows(pb.A[-1], 1.0)
        pb.zero_boundary_residuals()
        for _ in range(3):
            pb.vcycle()
        ctx.marker(1)
        for _ in range(10):
            pb.vcycle()
        ctx.marker(2)
    ctx.sync()
    print("TRAFFIC PROBE DONE", flush=True)


if __name__ == "__main__":
    main()
