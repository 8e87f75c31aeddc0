"""Measurement: what the first device solve of C4 costs by phase (DeviceSolver constructor)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, _solver, _nd, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
A = arrays_from_spec(spec); D = _lib.DeviceModel(A)
D.set_thickness(np.full(A.total_cp, spec.h_th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th)); D.assemble(3); D.sync()
X = np.stack([A.cp_hom[f] / A.weights for f in range(3)], 1)
t = time.perf_counter(); nb_ptr, nb = D.cp_graph(); t1 = time.perf_counter()
rowptr, col = D.pattern(_lib.MAT_K); n2 = _solver.control_point_graph(rowptr, col); t2 = time.perf_counter()
assert np.array_equal(n2[0], nb_ptr) and np.array_equal(n2[1], nb)
sym, pmap = _nd.nested_dissection_native(nb_ptr, nb, X, leaf=128); t3 = time.perf_counter()
print("gf_cp_graph %.3f s (dof-level pattern + control_point_graph: %.3f s, same lists), native symbolic phase %.3f s (%d fronts)" % (t1 - t, t2 - t1, t3 - t2, sym.nfronts))
del rowptr, col, n2
t = time.perf_counter(); S = _solver.DeviceSolver(D, coords=X); print("DeviceSolver constructor (graph, symbolic phase, handle, first factorisation) %.3f s" % (time.perf_counter() - t))
