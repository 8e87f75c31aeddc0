"""Measurement (not product): device solver (libgoldfish_solver.so: block-banded L D L^T on the FP64 matrix pipe) against host
SuperLU: factorisation and solve times of a Newton step (K x = -R) and an adjoint solve (K^T lam = g) per model."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, _solver, geometry as G
from goldfish_amd.model import arrays_from_spec
import scipy.sparse.linalg as spla

wing = G.wing_16patch_from_interface_data(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_wing_int_data.npz"), allow_pickle=True))
cases = [("C2 tbeam4", G.tbeam_4patch()), ("C3 wing 16 patches (reference interface data)", wing),
         ("shell 6x6 patches nel=24", G.synthetic_shell(6, 6, nel=24, p=3, jitter=2)), ("C3-size shell 4x4 patches nel=44", G.synthetic_shell(4, 4, nel=44, p=3, jitter=2))]
if os.environ.get("GF_SOLVER_BIG"):
    cases.append(("shell 8x8 patches nel=48 (1/4 of C4)", G.synthetic_shell(8, 8, nel=48, p=3, jitter=2)))
if os.environ.get("GF_SOLVER_C4"):
    cases = [("C4: shell 16x16 patches nel=48", G.synthetic_shell(16, 16, nel=48, p=3, jitter=2))]
host = os.environ.get("GF_SOLVER_HOST", "1") == "1"
for name, spec in cases:
    A = arrays_from_spec(spec)
    D = _lib.DeviceModel(A)
    D.set_thickness(np.full(A.total_cp, spec.h_th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
    D.assemble(_lib.ASM_R | _lib.ASM_K); D.sync()
    b = -D.residual(); g = np.random.default_rng(0).standard_normal(A.ndof)
    X = np.stack([A.cp_hom[f] / A.weights for f in range(3)], 1)
    t = time.perf_counter(); S = _solver.DeviceSolver(D, coords=X, method=os.environ.get("GF_SOLVER_METHOD", "auto")); t_first = time.perf_counter() - t
    info = S.info()
    for _ in range(3): S.refactor(); S.solve(b)           # the library launches the first three sweeps of a kind directly and captures its HIP graph at the fourth
    tfs = []
    for _ in range(int(os.environ.get("GF_SOLVER_REFACTOR_SAMPLES", "7"))):
        t = time.perf_counter(); S.refactor(); tfs.append(time.perf_counter() - t)
    t_f = float(np.median(tfs)); t_fmin = min(tfs)
    tps = []
    for _ in range(5):                                   # the factor storage cleared beforehand (DeviceSolver.prepare: what the Newton loop does under its assembly pass)
        S.prepare()
        import torch; torch.cuda.synchronize()                # the clearing runs on the solver's stream: wait for the device
        t = time.perf_counter(); S.refactor(); tps.append(time.perf_counter() - t)
    t_fp = float(np.median(tps))
    t = time.perf_counter(); x = S.solve(b); t_s = time.perf_counter() - t; rr, be = S.rel_residual, S.backward_error
    t = time.perf_counter(); lam = S.solve(g); t_a = time.perf_counter() - t; ra, bea = S.rel_residual, S.backward_error
    fmt = ("%s [" + S.method + "]: %d dofs, half bandwidth %d, factor storage %.2f GB; device: ordering + first factorisation %.3f s, re-factorisation %.4f s (median of %d; fastest %.4f s; %.1f TFLOP/s; %.4f s with the factor storage cleared beforehand), "
           "Newton solve %.4f s (residual %.1e, backward error %.1e), adjoint solve %.4f s (residual %.1e, backward error %.1e)")
    line = fmt % (name, A.ndof, info["half_bandwidth"], info["device_bytes"] / 1e9, t_first, t_f, len(tfs), t_fmin, info["factor_flops"] / t_f / 1e12, t_fp, t_s, rr, be, t_a, ra, bea)
    # several right-hand sides in one call (gfs_solve_multi: the sweeps next to each other on their own streams in the nested-dissection mode)
    B3 = np.stack([g, b, np.random.default_rng(1).standard_normal(A.ndof)])
    for _ in range(4): S.solve_multi(B3)                                  # first calls: create the extra workspaces / graphs
    t = time.perf_counter(); X3 = S.solve_multi(B3); t_m = time.perf_counter() - t
    line += "; 3 right-hand sides in one call %.4f s (largest residual %.1e, backward error %.1e; max difference to the single solves %.1e)" % (
        t_m, S.rel_residual, S.backward_error, max(np.abs(X3[0] - lam).max() / np.abs(lam).max(), np.abs(X3[1] - x).max() / np.abs(x).max()))
    # the sweeps alone (no refinement): one, two, three and six right-hand sides
    ts = []
    for k in (1, 2, 3, 6):
        Bk = np.stack([g, b, B3[2], g[::-1].copy(), b[::-1].copy(), B3[2][::-1].copy()][:k])
        for _ in range(4): S.solve_multi(Bk, max_refine=0)
        t = time.perf_counter(); S.solve_multi(Bk, max_refine=0); ts.append(time.perf_counter() - t)
    line += "; substitutions only (no refinement, host copies included): 1 / 2 / 3 / 6 right-hand sides %.4f / %.4f / %.4f / %.4f s" % tuple(ts)
    if host and A.ndof < 150000:
        K = D.csr(_lib.MAT_K).tocsc()
        t = time.perf_counter(); lu = spla.splu(K, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True)); t_h = time.perf_counter() - t
        t = time.perf_counter(); xh = lu.solve(b); t_hs = time.perf_counter() - t
        line += "; host SuperLU: factorisation %.2f s, solve %.3f s, nnz(L+U) %d; difference of the solutions %.1e" % (t_h, t_hs, lu.L.nnz + lu.U.nnz, np.abs(x - xh).max() / np.abs(xh).max())
    print(line, flush=True)
    S.close(); D.close()
