"""Diagnostic / measurement (not product): device re-factorisation + solve (libgoldfish_solver.so) against host SuperLU."""
import sys, time, ctypes, faulthandler, numpy as np
faulthandler.dump_traceback_later(1000, exit=True)
sys.path.insert(0, "/root/repo")
log = open("/root/repo/gpurun_out/solver_bench.log", "w")
def say(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
t = time.time(); ctypes.CDLL("/root/repo/goldfish_amd/libgoldfish_solver.so", mode=ctypes.RTLD_GLOBAL); say("load libgoldfish_solver.so (+rocSOLVER, rocSPARSE, rocBLAS) %.1f s" % (time.time() - t))
from goldfish_amd import _lib, _solver, geometry as G
from goldfish_amd.model import arrays_from_spec
import scipy.sparse as sp, scipy.sparse.linalg as spla
for name, spec in (("tbeam2 (342 dofs)", G.tbeam_2patch(6)), ("C2 tbeam4", G.tbeam_4patch()), ("shell 4x4 patches nel=24", G.synthetic_shell(4, 4, nel=24, p=3, jitter=2)),
                   ("shell 6x6 patches nel=24", G.synthetic_shell(6, 6, nel=24, p=3, jitter=2))):
    A = arrays_from_spec(spec)
    D = _lib.DeviceModel(A)
    D.set_thickness(np.full(A.total_cp, spec.h_th)); D.set_u(np.zeros(A.ndof))
    D.assemble(); D.sync()
    rowptr, col = D.pattern(_lib.MAT_K); K = sp.csr_matrix((D.values(_lib.MAT_K), col, rowptr), shape=(A.ndof, A.ndof))
    b = -D.residual()
    t = time.time(); lu = spla.splu(K.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True)); t_host = time.time() - t
    xd = lu.solve(b)
    say("%s: %d dofs, nnz(K) %d; host SuperLU factorisation %.3f s" % (name, A.ndof, K.nnz, t_host))
    t = time.time(); S = _solver.DeviceSolver(D); say("  DeviceSolver (host symbolic + upload + analysis) %.2f s, nnz(L+U) %d, %.1f MB on device" % (time.time() - t, S.nnzT, S.device_bytes / 1e6))
    x = S.solve(b); say("  solve (factors computed on the device): err vs host %.2e" % (np.abs(x - xd).max() / np.abs(xd).max()))
    # new state -> new K values, same pattern: re-factorise on the device
    D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th)); D.assemble(); D.sync()
    K2 = sp.csr_matrix((D.values(_lib.MAT_K), col, rowptr), shape=(A.ndof, A.ndof)); b2 = -D.residual()
    t = time.time(); x2h = spla.splu(K2.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True)).solve(b2); t_host2 = time.time() - t
    t = time.time(); S.refactor(); t_rf = time.time() - t
    t = time.time(); x2 = S.solve(b2); t_sv = time.time() - t
    S.refactor(); t = time.time(); S.refactor(); t_rf2 = time.time() - t
    t = time.time(); S.solve(b2); t_sv2 = time.time() - t
    say("  new K: device refactor %.4f s (again %.4f), solve %.4f s (again %.4f); host factor+solve %.3f s; err vs host %.2e" % (t_rf, t_rf2, t_sv, t_sv2, t_host2, np.abs(x2 - x2h).max() / np.abs(x2h).max()))
    S.close(); D.close()
