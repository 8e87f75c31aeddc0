"""Measurement (not product): where the set-up of the device solver goes at C4 (pattern fetch, control-point graph, nested dissection, boundary maps, gfs_create_nd, first
factorisation incl. the graph capture)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, _solver, _nd, geometry as G
from goldfish_amd.model import arrays_from_spec
n = int(os.environ.get("GF_SIDE", "16"))
spec = G.synthetic_shell(n, n, nel=48, p=3, jitter=2)
A = arrays_from_spec(spec)
D = _lib.DeviceModel(A)
D.set_thickness(np.full(A.total_cp, spec.h_th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
D.assemble(_lib.ASM_R | _lib.ASM_K); D.sync()
X = np.stack([A.cp_hom[f] / A.weights for f in range(3)], 1)
t0 = time.perf_counter(); rowptr, col = D.pattern(_lib.MAT_K); t1 = time.perf_counter()
nb_ptr, nb = _solver.control_point_graph(rowptr, col); t2 = time.perf_counter()
sym = _nd.nested_dissection(nb_ptr, nb, X, leaf=256); t3 = time.perf_counter()
pm = _solver.parent_positions(sym); t4 = time.perf_counter()
print("pattern fetch %.2f s, control-point graph %.2f s, nested dissection %.2f s, boundary maps %.2f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3), flush=True)
t5 = time.perf_counter(); S = _solver.DeviceSolver(D, coords=X, method="nd"); t6 = time.perf_counter()
print("DeviceSolver(...) in all %.2f s (of it the four host steps above: %.2f s)" % (t6 - t5, t4 - t0), flush=True)
t = time.perf_counter(); S.refactor(); print("second factorisation %.3f s" % (time.perf_counter() - t))
