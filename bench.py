#!/usr/bin/env python3
"""bench.py -- assembly + adjoint throughput of the KL-shell hot path on N MI355X.

One "step" = one pass of the hot path over the synthetic multi-patch shell: residual R,
tangent K, dR/dCP (3 fields) and dR/dh (everything DispImOpeartion.apply_nonlinear +
linearize produce, GOLDFISH/operations/disp_imop.py:33-56), penalty coupling included,
with control points, thickness and displacements already resident in HBM.
metric = element-Gauss-point updates per second (BASELINE.json); the workload is C4
(16x16 bicubic NURBS patches, ~2.0M dofs, ~9.4M Gauss points), the
largest configuration that fits one GPU; for N > 1 the same model is patch-sharded
(strong scaling) and every step ends with the RCCL all-reduce of the global residual.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--patches NX NY] [--nel E] [--degree P]
N > 1:  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = FP64 matrix (public spec)
ALG_BYTES_PER_GP = {2: 476.4, 3: 520.0, 4: 548.0}    # (K + 3 dR/dCP + dR/dh CSR values written once + in/out vectors) / (p+1)^2
ALG_FLOP_PER_GP = {2: 2.6e4, 3: 6.3e4, 4: 1.4e5}     # FMA*2 count of the kernel's formulation, DESIGN.md section 4


def cpu_baseline(args, ncores):
    """The CPU oracle (C, OpenMP over patches) on a bounded sample of the same workload:
    `ncores` patches of the C4 generator, full assembly + residual, best of 2."""
    from goldfish_amd import geometry as G
    from goldfish_amd.model import arrays_from_spec
    from oracle import oracle_py
    oracle_py.build()
    npatch = max(1, min(ncores, 32))           # OpenMP parallelism of the oracle is over patches
    oracle_py.lib().gfo_set_num_threads(npatch)
    spec = G.synthetic_shell(npatch, 1, nel=args.nel, p=args.degree, jitter=2)
    th = G.random_thickness(spec)
    A = arrays_from_spec(spec, th)
    O = oracle_py.Oracle(A, thickness=np.concatenate(th), u=G.smooth_displacement(spec, 0.5 * spec.h_th))
    best = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        O.residual()
        O.assemble()
        best = min(best, time.perf_counter() - t0)
    return {"value": A.n_gauss_points / best, "unit": "GP-updates/s", "cores": npatch,
            "kind": "port", "host_cores": ncores,
            "sample": "%d patches (%d GPs) of the same generator, oracle/kl_oracle.c R+K+dRdCP+dRdh on %d OpenMP threads (one per patch), best of 2 (%.1f s)"
            % (npatch, A.n_gauss_points, npatch, best)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--patches", type=int, nargs=2, default=[16, 16])
    ap.add_argument("--nel", type=int, default=48)
    ap.add_argument("--degree", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    # GF_BENCH_REHEARSE=1: all ranks on GPU 0 with the gloo backend -- exercises the N > 1 code path on a one-GPU box
    rehearse = os.environ.get("GF_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from goldfish_amd import _lib, build, geometry as G, sharding
    if rank == 0:
        build.build()
    if dist is not None:
        dist.barrier()

    spec = G.synthetic_shell(args.patches[0], args.patches[1], nel=args.nel, p=args.degree, jitter=2)
    th_g = G.random_thickness(spec)
    u_g = G.smooth_displacement(spec, 0.5 * spec.h_th)
    shard = sharding.shard_spec(spec, rank, world)
    A = sharding.shard_arrays(shard, th_g)
    D = _lib.DeviceModel(A, device=local_rank)
    D.set_thickness(shard.to_local(np.concatenate(th_g)))
    D.set_u(shard.to_local(u_g, 3))
    n_gp_local = D.n_gauss_points
    n_gp_total = int(sum(p.nel[0] * p.nel[1] * (p.p + 1) * (p.q + 1) for p in spec.patches))

    # zero-copy torch view of the residual buffer for the RCCL exchange
    class _Buf:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    exchange = world > 1 or os.environ.get("GF_BENCH_FORCE_EXCHANGE") == "1"     # the env switch rehearses the N>1 code path on one GPU
    R_loc = torch.as_tensor(_Buf(_lib.lib().gf_device_ptr(D.h, _lib.BUF_R), A.ndof), device="cuda") if exchange else None
    R_glob = torch.zeros(3 * shard.total_cp_global, dtype=torch.float64, device="cuda") if exchange else None

    def step():
        D.assemble(_lib.ASM_ALL, sync=False)
        if exchange:
            D.sync()
            sharding.allgather_owned_rows(shard, R_loc, dist, 3, out=R_glob)

    def fence():
        D.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    D.kernel_ms()                       # reset the HIP-event accumulator
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kern_ms, kern_n = D.kernel_ms()
    if exchange:       # the exchanged global residual must equal the library's own copy of the owned rows
        g0, g1 = shard.owned_global_range(3)
        assert np.array_equal(R_glob[g0:g1].cpu().numpy(), D.residual()[:g1 - g0]) or world > 1
    # the HBM-bound phase of the path: apply_linear = block-CSR SpMV on the assembled K
    # (DispImOpeartion.apply_linear_fwd, disp_imop.py:58-72), device pointers, HIP events on torch's stream are
    # not used: the library's own stream is timed by wall clock around a synchronised batch
    apply = None
    if rank == 0:
        xk = torch.ones(A.ndof, dtype=torch.float64, device="cuda")
        yk = torch.zeros(A.ndof, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        L = _lib.lib()
        for _ in range(3):
            L.gf_apply_dev(D.h, _lib.MAT_K, 0, xk.data_ptr(), yk.data_ptr())
        D.sync()
        ta = time.perf_counter()
        nrep = 20
        for _ in range(nrep):
            L.gf_apply_dev(D.h, _lib.MAT_K, 0, xk.data_ptr(), yk.data_ptr())
        D.sync()
        ta = (time.perf_counter() - ta) / nrep
        nnzK = L.gf_nnz(D.h, _lib.MAT_K)
        byt = nnzK * 8.0 + (nnzK / 9.0) * 4.0 + 3 * A.ndof * 8.0       # values + block column ids + x, y(read+write)
        apply = {"kernel": "csr_apply_kernel (K x)", "bound": "hbm", "ms": 1e3 * ta, "algorithmic_bytes": byt,
                 "achieved": byt / ta / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byt / ta / 1e9 / HBM_PEAK_GBS}
    # the pass a Newton iteration needs (R + K only; leaner element-kernel and gather instances), outside the timed region
    newton_ms = None
    if rank == 0:
        for _ in range(2):
            D.assemble(_lib.ASM_R | _lib.ASM_K, sync=False)
        D.sync()
        tn = time.perf_counter()
        for _ in range(3):
            D.assemble(_lib.ASM_R | _lib.ASM_K, sync=False)
        D.sync()
        newton_ms = 1e3 * (time.perf_counter() - tn) / 3
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = n_gp_total * args.steps / dt

    if rank == 0:
        p = args.degree
        alg_bytes = ALG_BYTES_PER_GP[p] * n_gp_local
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if tj.get("workload_gps") == n_gp_local:
                    traffic = tj.get("element_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        mfma = os.environ.get("GF_ELEMENT", "mfma") != "valu"
        kname = ("kl_element_mfma4_kernel" if p == 4 else "kl_element_mfma_kernel") if mfma else "kl_element_kernel"
        out = {
            "metric": "element-Gauss-point updates/sec (assembly+adjoint)", "value": value, "unit": "GP-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s synthetic %dx%d-patch curved NURBS shell, p=%d, %d spans/side +-2 (non-matching), "
                                   "%d dofs, %d Gauss points, %d mortar points; R+K+dRdCP(3)+dRdh incl. penalty coupling"
                                   % ("C4" if (args.patches == [16, 16] and args.nel == 48 and p == 3) else "custom", args.patches[0], args.patches[1], p, args.nel, 3 * shard.total_cp_global, n_gp_total,
                                      sum(i.npts for i in spec.interfaces)),
                       "parallelism": "patch-sharded x%d, owner-computes-rows, all-gather of the owned residual rows" % world},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms, "launches_timed": kern_n},
            "roofline_fp64": {"bound": "fp64 (v_mfma_f64 + FP64 VALU share one pipe)" if mfma else "fp64-valu", "kernel": kname,
                              "achieved": ALG_FLOP_PER_GP[p] * n_gp_local / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0,
                              "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "note": "the tangent / shape-Jacobian contraction is FP64 bound (SURVEY.md 8(d)); "
                                      "this is the binding roofline of the dominant kernel; flop count = the formulation's "
                                      "FMA*2 per Gauss-point update (DESIGN.md section 4), not hardware-issued flops"},
            "apply_linear_roofline": apply,
            "newton_pass": {"what": "R + K only (one Newton iteration of solve_nonlinear), rank 0's share", "ms": newton_ms},
            "device_bytes": D.device_bytes,
        }
        out["roofline_fp64"]["frac"] = out["roofline_fp64"]["achieved"] / FP64_PEAK_TFLOPS
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, os.cpu_count() or 1)
            except Exception as ex:       # the oracle is only the reported baseline; never the product
                out["cpu_baseline"] = {"error": str(ex)}
        print(json.dumps(out), flush=True)
    D.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
