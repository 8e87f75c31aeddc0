#!/usr/bin/env python3
"""bench.py -- assembly + adjoint throughput of the KL-shell hot path on N MI355X.

One "step" = one pass of the hot path over the synthetic multi-patch shell: residual R,
tangent K, dR/dCP (3 fields) and dR/dh (everything DispImOpeartion.apply_nonlinear +
linearize produce, GOLDFISH/operations/disp_imop.py:33-56), penalty coupling included,
with control points, thickness and displacements already resident in HBM.
metric = element-Gauss-point updates per second (BASELINE.json names no configuration for it); the default workload is C4
(BASELINE.json configs[3], SURVEY.md 8(d): 16x16 bicubic NURBS patches, ~2.0M dofs, ~9.4M Gauss points) -- the bicubic
configuration the reference's demos correspond to; for N > 1 the same model is patch-sharded (strong scaling) and every step
ends with the RCCL all-gather of the owned residual rows.  C5 (configs[4]: 32x32 quartic patches of the fuselage skin, ~10M dofs,
71.9M Gauss points) is the LARGEST configuration and also fits one MI355X (100 GB): at N = 1 the default run measures it too and
reports it as the `secondary` record of the same JSON line (own roofline / roofline_fp64; skipped, with the reason, when the device
has less than 110 GB free or with --no-secondary).  By hand: `--geometry fuselage --patches 32 32 --nel 53 --degree 4`; one GPU's
share of it: `--geometry fuselage --patches 16 8 --nel 53 --degree 4`.
At N = 1 the default run also adds `device_solver` (not part of `value`): the factorisation and the solves of the K it has just assembled -- the other nine tenths of a Newton
step (`--no-solver` skips it).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--geometry shell|fuselage] [--patches NX NY] [--nel E] [--degree P]
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


def usable_cores():
    """Cores this process can really run on: the affinity mask, capped by the cgroup CPU quota (cpu.max / cfs_quota) of the box."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                n = max(1, min(n, int(round(int(q) / int(per)))))
            break
        except Exception:
            continue
    return n


def cpu_baseline(args, ncores):
    """The CPU oracle (C, OpenMP over patches: the restatement of the reference's algorithm) on a bounded
    sample of the same workload, on this box's host cores: one patch on one thread, and one patch per usable core on all of
    them (`ncores` = the cores this process may run on); each 1 warm-up + best of 5 (3 for the single thread: ~4 s a run)."""
    from goldfish_amd import geometry as G
    from goldfish_amd.model import arrays_from_spec
    from oracle import oracle_py
    oracle_py.build()

    def run(npatch, nthreads, reps):
        oracle_py.lib().gfo_set_num_threads(nthreads)
        spec = make_spec(args, npatch, 1)
        th = G.random_thickness(spec)
        A = arrays_from_spec(spec, th)
        O = oracle_py.Oracle(A, thickness=np.concatenate(th), u=G.smooth_displacement(spec, 0.5 * spec.h_th))
        best = 1e30
        for it in range(reps + 1):                 # first run = warm-up
            t0 = time.perf_counter()
            O.residual()
            O.assemble()
            if it > 0:
                best = min(best, time.perf_counter() - t0)
        return A.n_gauss_points / best, A.n_gauss_points, best

    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    v1, gp1, t1 = run(1, 1, 3)
    nall = max(1, min(ncores, 256))
    # bounded sample: the all-core leg is skipped down to what ~5 s per repetition allows if the cores turn out to be oversubscribed
    vn, gpn, tn = run(nall, nall, 5 if t1 * 6 < 30 else 2) if nall > 1 else (v1, gp1, t1)
    return {"value": vn, "unit": "GP-updates/s", "cores": nall, "kind": "port", "host_cores": os.cpu_count(), "usable_cores": ncores,
            "single_thread": {"value": v1, "cores": 1, "sample": "1 patch (%d GPs), best of 3 after 1 warm-up (%.1f s)" % (gp1, t1)},
            "sample": "%d patches (%d GPs) of the same generator, oracle/kl_oracle.c R+K+dRdCP+dRdh on %d OpenMP threads (one patch per thread), "
                      "best of 5 after 1 warm-up (%.1f s)" % (nall, gpn, nall, tn)}


def make_spec(args, nx, ny):
    """The synthetic generator of the workload: the doubly curved shell of C4 or the cylindrical fuselage skin of C5."""
    from goldfish_amd import geometry as G
    if args.geometry == "fuselage":
        return G.synthetic_fuselage(nx, ny, nel=args.nel, p=args.degree, jitter=2)
    return G.synthetic_shell(nx, ny, nel=args.nel, p=args.degree, jitter=2)


def workload_name(args):
    key = (args.geometry, tuple(args.patches), args.nel, args.degree)
    return {("shell", (16, 16), 48, 3): "C4", ("fuselage", (32, 32), 53, 4): "C5", ("fuselage", (16, 8), 53, 4): "one GPU's share (1/8) of C5"}.get(key, "custom")


def kname_of(D, p):
    """Name of the dominant kernel of the path the handle runs (gf_assembly_path)."""
    return {5: "kl_element_rec4_kernel", 4: "kl_element_rec_kernel", 3: "kl_element_kernel"}.get(D.assembly_path, "kl_element_mfma4_kernel" if p == 4 else "kl_element_mfma_kernel")   # 6 (p = 4 hybrid): the full pass runs the block kernel


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--geometry", choices=["shell", "fuselage"], default="shell")
    ap.add_argument("--patches", type=int, nargs=2, default=[16, 16])
    ap.add_argument("--nel", type=int, default=48)
    ap.add_argument("--degree", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1, default workload: do not add the C5 record")
    ap.add_argument("--no-solver", action="store_true", help="N = 1, default workload: do not add the device_solver record (factorisation / solves of this K)")
    ap.add_argument("--full-pass-only", action="store_true", help="skip the apply_linear and Newton-pass legs (PMC passes: every launch belongs to a full pass)")
    return ap.parse_args(argv)


def git_head():
    """Commit of the tree this runs from: git where there is one, else the record goldfish_amd.build.build() wrote where it last ran WITH git (the driver's box has
    no git; the record travels with the built libraries) -- marked as such."""
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        from goldfish_amd import build
        bi = build.build_info()
        return ("%s%s (goldfish_amd/_build_info.json)" % (bi["head"], "+uncommitted" if bi.get("dirty") else "")) if bi.get("head") else None


def last_commit_of(path):
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "log", "-n", "1", "--format=%h", "--", path], stderr=subprocess.DEVNULL).decode().strip() or None
    except Exception:
        return None


def main():
    args = parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    from goldfish_amd import build
    if rank == 0:
        build.build()                          # hipcc (if anything is stale) runs before this process touches the GPU
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

    out = measure(args, torch, dist, rank, local_rank, world)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, usable_cores())
            except Exception as ex:       # the oracle is only the reported baseline; never the product
                out["cpu_baseline"] = {"error": str(ex)}
        # C5 beside C4 (round-3 verdict 6): the largest configuration also fits one GPU; measured in the same run when there is room
        if world == 1 and workload_name(args) == "C4" and not args.no_secondary and not args.full_pass_only:
            free_b, _tot = torch.cuda.mem_get_info()
            if free_b >= 110e9:
                try:
                    a2 = parse_args(["--geometry", "fuselage", "--patches", "32", "32", "--nel", "53", "--degree", "4", "--steps", "2", "--warmup", "1"])
                    sec = measure(a2, torch, None, 0, local_rank, 1)
                    out["secondary"] = {k: sec[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "roofline_fp64",
                                                            "apply_linear_roofline", "newton_pass", "device_bytes")}
                except Exception as ex:
                    out["secondary"] = {"skipped": "C5 run failed: %s" % ex}
            else:
                out["secondary"] = {"skipped": "C5 needs ~100 GB of device memory; %.0f GB free" % (free_b / 1e9)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def measure(args, torch, dist, rank, local_rank, world):
    """One workload: set-up, warm-up, the timed steps, the apply_linear and Newton-pass legs; returns the record (rank 0) or None."""
    from goldfish_amd import _lib, geometry as G, sharding
    if dist is not None:
        dist.barrier()

    spec = make_spec(args, args.patches[0], args.patches[1])
    th_g = G.random_thickness(spec)
    u_g = G.smooth_displacement(spec, 0.5 * spec.h_th)
    part = sharding.partition_patches(spec, world)          # interface-graph partition balanced by Gauss points
    pq = sharding.partition_quality(spec, part)
    shard = sharding.shard_spec(spec, rank, world, part)
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
    rehearse = os.environ.get("GF_BENCH_REHEARSE") == "1"
    exchange = world > 1 or os.environ.get("GF_BENCH_FORCE_EXCHANGE") == "1"     # the env switch rehearses the N>1 code path on one GPU
    R_loc = torch.as_tensor(_Buf(_lib.lib().gf_device_ptr(D.h, _lib.BUF_R), A.ndof), device="cuda") if exchange else None
    R_glob = torch.zeros(3 * shard.total_cp_global, dtype=torch.float64, device="cuda") if exchange else None

    # the library launches on its own stream: the exchange (torch's stream, RCCL) is ordered behind the assembly by an event,
    # and the next assembly behind the exchange's read of the residual buffer -- no host synchronisation inside a step
    lib_stream = torch.cuda.ExternalStream(D.stream_ptr) if exchange else None

    def step():
        if exchange:
            lib_stream.wait_stream(torch.cuda.current_stream())
        D.assemble(_lib.ASM_ALL, sync=False)
        if exchange:
            torch.cuda.current_stream().wait_stream(lib_stream)
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
    kern_ms_step = kern_ms * kern_n / max(args.steps, 1)      # element-kernel time per step (a step may launch it in chunks)
    if exchange:       # the exchanged global residual must equal the library's own copy of the owned rows
        rows = shard.owned_rows_global(3)
        assert np.array_equal(R_glob[torch.from_numpy(rows).cuda()].cpu().numpy(), D.residual()[:rows.size])
    # the HBM-bound phase of the path: apply_linear = block-CSR SpMV on the assembled K
    # (DispImOpeartion.apply_linear_fwd, disp_imop.py:58-72), device pointers, HIP events on torch's stream are
    # not used: the library's own stream is timed by wall clock around a synchronised batch
    apply = None
    if rank == 0 and not args.full_pass_only:
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
    if rank == 0 and not args.full_pass_only:
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

    out = None
    if rank == 0:
        p = args.degree
        alg_bytes = ALG_BYTES_PER_GP[p] * n_gp_local
        achieved = alg_bytes / (kern_ms_step * 1e-3) / 1e9 if kern_ms_step > 0 else 0.0
        traffic = counter_flop = step_traffic = traffic_source = None
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic*.json"))):     # PMC-derived bytes / flop of the profiled workloads (tools/traffic_from_pmc.py)
            try:
                tj = json.load(open(tf))
                if tj.get("workload_gps") == n_gp_local and str(tj.get("element_kernel", "")).startswith(kname_of(D, args.degree)):
                    traffic = tj.get("element_kernel_bytes_per_launch")
                    counter_flop = tj.get("element_kernel_fp64_flop_issued_per_launch")
                    step_traffic = tj.get("full_pass_bytes_per_step")
                    # stored builder measurement (rocprofv3 PMC passes of an earlier run of this workload), not this run's: say which
                    traffic_source = "%s@%s (from %s)" % (os.path.relpath(tf, ROOT), last_commit_of(tf) or tj.get("generated_at_commit"), tj.get("source", "profiles/*_pmc_*.csv"))
            except Exception:
                pass
        mfma = os.environ.get("GF_ELEMENT", "mfma") != "valu"
        kname = kname_of(D, p)
        out = {
            "metric": "element-Gauss-point updates/sec (assembly+adjoint)", "value": value, "unit": "GP-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s synthetic %dx%d-patch %s, p=%d, %d spans/side +-2 (non-matching), "
                                   "%d dofs, %d Gauss points, %d mortar points; R+K+dRdCP(3)+dRdh incl. penalty coupling"
                                   % (workload_name(args), args.patches[0], args.patches[1], "curved NURBS shell" if args.geometry == "shell" else "cylindrical fuselage skin",
                                      p, args.nel, 3 * shard.total_cp_global, n_gp_total,
                                      sum(i.npts for i in spec.interfaces)),
                       "parallelism": "patch-sharded x%d (interface-graph partition), owner-computes-rows, all-gather of the owned residual rows" % world,
                       "partition": {"gauss_points_per_rank": pq["gauss_points"], "imbalance_max_over_mean": pq["imbalance"],
                                     "cut_interfaces": pq["cut_interfaces"], "ghost_patches_per_rank": pq["ghost_patches"]}},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms_step, "launches_timed": kern_n,
                         "launches_per_step": kern_n / max(args.steps, 1), "step_traffic_all_kernels": step_traffic,
                         "traffic_source": traffic_source, "timing": "HIP events on the library's stream around every launch of the element kernel(s), this run"},
            "roofline_fp64": {"bound": "fp64 (v_mfma_f64 + FP64 VALU share one pipe)" if mfma else "fp64-valu", "kernel": kname,
                              "achieved": ALG_FLOP_PER_GP[p] * n_gp_local / (kern_ms_step * 1e-3) / 1e12 if kern_ms_step > 0 else 0.0,
                              "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "note": "the tangent / shape-Jacobian contraction is FP64 bound (SURVEY.md 8(d)); "
                                      "this is the binding roofline of the dominant kernel; flop count = the formulation's "
                                      "FMA*2 per Gauss-point update (DESIGN.md section 4), not hardware-issued flops",
                              # what the pipe sustains on this GPU: microbenchmarks of an earlier run (tools/ubench_acc, ubench_batch, ubench_f64), NOT measured in this run --
                              # not a second peak, the explanation of where the kernel sits against the data-sheet figure above
                              "pipe_measured": {"source": "profiles/r03_ubench_fp64_mfma.txt (stored microbenchmark output of round 3, commit a6d8311; NOT measured in this run)",
                                                "v_mfma_f64_16x16x4_vgpr_accumulators_tflops": 74.0, "v_mfma_f64_16x16x4_agpr_accumulators_tflops_one_wave_per_simd": 36.0,
                                                "v_mfma_f64_16x16x4_agpr_accumulators_tflops_two_waves_per_simd": 45.0, "v_fma_f64_tflops": 56.0,
                                                "mfma_and_fp64_valu_co_execute": False}},
            "apply_linear_roofline": apply,
            "newton_pass": {"what": "R + K only (one Newton iteration of solve_nonlinear), rank 0's share", "ms": newton_ms},
            "device_bytes": D.device_bytes, "head": git_head(),
        }
        out["roofline_fp64"]["frac"] = out["roofline_fp64"]["achieved"] / FP64_PEAK_TFLOPS
        if counter_flop and kern_ms_step > 0:      # the same fraction with the flop count the SQ counters report for this kernel and workload (profiles/traffic.json)
            out["roofline_fp64"]["counter_flop_per_launch"] = counter_flop
            out["roofline_fp64"]["counter_flop_source"] = traffic_source
            out["roofline_fp64"]["frac_counters"] = counter_flop / (kern_ms_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
        if world == 1 and workload_name(args) == "C4" and not args.full_pass_only and not args.no_solver:
            try:
                out["device_solver"] = solver_record(D, A, np)
            except Exception as ex:
                out["device_solver"] = {"error": str(ex)}
    D.close()
    return out if rank == 0 else None


def solver_record(D, A, np):
    """SURVEY 8(f) N1 beside the headline: the linear solves of a Newton step / an adjoint on the K this run assembled (DispImOpeartion.solve_nonlinear / solve_linear,
    GOLDFISH/operations/disp_imop.py:38-44, 130-142; MUMPS in the reference).  Not part of `value`: its own record, measured live."""
    from goldfish_amd import _lib, _solver
    D.assemble(_lib.ASM_R | _lib.ASM_K); D.sync()
    b = -D.residual()
    X = np.stack([A.cp_hom[f] / A.weights for f in range(3)], 1)
    t = time.perf_counter(); S = _solver.DeviceSolver(D, coords=X); t_first = time.perf_counter() - t
    try:
        info = S.info()
        for _ in range(3):
            S.refactor(); S.solve(b)                      # the library captures its HIP graphs at the fourth call of a sweep
        tf = []
        for _ in range(5):
            t = time.perf_counter(); S.refactor(); tf.append(time.perf_counter() - t)
        t = time.perf_counter(); S.solve(b); t_solve = time.perf_counter() - t
        rr, be = S.rel_residual, S.backward_error
        for _ in range(4):
            S.solve(b, max_refine=0)
        t = time.perf_counter(); S.solve(b, max_refine=0); t_sweep = time.perf_counter() - t
        ts = []
        for _ in range(3):                                   # one Newton step as the library sees it: R + K pass, factorisation, one substitution sweep (no refinement)
            t = time.perf_counter(); D.assemble(_lib.ASM_R | _lib.ASM_K); D.sync(); S.refactor(); S.solve(b, max_refine=0); ts.append(time.perf_counter() - t)
        f = float(np.median(tf))
        return {"what": "K x = -R of this model on the device: nested-dissection multifrontal L D L^T on 64 x 64 FP64-MFMA tiles (goldfish_amd/csrc/gf_solver.hip), K read in place",
                "method": S.method, "dofs": int(A.ndof), "factor_bytes": int(info["device_bytes"]), "factor_flop": float(info["factor_flops"]),
                "ordering_and_first_factorisation_s": t_first, "factorisation_s": f, "factorisation_samples": len(tf), "factorisation_tflops": info["factor_flops"] / f / 1e12,
                "frac_of_fp64_matrix_peak": info["factor_flops"] / f / 1e12 / FP64_PEAK_TFLOPS, "solve_with_refinement_s": t_solve, "relative_residual": rr, "backward_error": be,
                "substitution_sweep_s": t_sweep, "newton_step_s": float(np.median(ts)), "newton_step_what": "R + K assembly pass + factorisation + one substitution sweep, host copies of b and x included",
                "timing": "host wall clock around synchronous calls of the C ABI (host copies of b and x included in the solves), this run"}
    finally:
        S.close()


if __name__ == "__main__":
    main()
