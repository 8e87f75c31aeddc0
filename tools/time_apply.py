"""Measurement: one reverse-mode and one forward-mode apply_linear of DispImOpeartion at C4 (K, dR/dCP x 3, dR/dh: gf_apply_many)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
th = G.random_thickness(spec)
A = arrays_from_spec(spec, th)
D = _lib.DeviceModel(A)
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th)); D.assemble(); D.sync()
rng = np.random.default_rng(0)
lam = rng.standard_normal(A.ndof)
for name, which in (("K^T, (dR/dCP_0,1,2)^T, (dR/dh)^T", [0, 1, 2, 3, 4]), ("three Jacobians: K^T, (dR/dCP_0)^T, (dR/dh)^T", [0, 1, 4])):
    ys = [np.zeros(A.ndof if w == 0 else A.total_cp) for w in which]
    D.apply_many(which, [lam], ys, transpose=True)
    t0 = time.perf_counter()
    for _ in range(5): D.apply_many(which, [lam], ys, transpose=True)
    t_many = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        for w, y in zip(which, ys): D.apply(w, lam, y, transpose=True)
    t_sep = (time.perf_counter() - t0) / 5
    print("reverse mode, %s: gf_apply_many %.2f ms, one gf_apply per matrix %.2f ms" % (name, t_many * 1e3, t_sep * 1e3), flush=True)
xs = [rng.standard_normal(A.ndof)] + [rng.standard_normal(A.total_cp) for _ in range(4)]
y = np.zeros(A.ndof)
D.apply_many([0, 1, 2, 3, 4], xs, [y])
t0 = time.perf_counter()
for _ in range(5): D.apply_many([0, 1, 2, 3, 4], xs, [y])
print("forward mode, five products: gf_apply_many %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
