"""Measurement (not product): what gfs_prepare_refactor is worth inside a Newton step at C4 -- one step = assembly pass (R + K) + factorisation + one substitution sweep.
With the preparation the clearing of the factor storage (57 GB of HBM writes) is started BEFORE the assembly pass is launched and runs beside it; without it the
factorisation starts with it.  Result (profiles/r05_prepare_overlap.txt): the step takes the same time -- the fill kernel's waves share the SIMDs with the element kernel's
one wave per SIMD, which is bound by instruction issue, and the pass slows down by what the factorisation gains."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, _solver, geometry as G
from goldfish_amd.model import arrays_from_spec
import torch
n = int(os.environ.get("GF_PATCHES", "16"))
spec = G.synthetic_shell(n, n, nel=48, p=3, jitter=2)
A = arrays_from_spec(spec)
D = _lib.DeviceModel(A)
D.set_thickness(np.full(A.total_cp, spec.h_th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
D.assemble(_lib.ASM_R | _lib.ASM_K); D.sync()
b = -D.residual()
X = np.stack([A.cp_hom[f] / A.weights for f in range(3)], 1)
S = _solver.DeviceSolver(D, coords=X)
for _ in range(4): S.refactor(); S.solve(b, max_refine=0)
def step(prep):
    torch.cuda.synchronize(); t = time.perf_counter()
    if prep: S.prepare()
    D.assemble(_lib.ASM_R | _lib.ASM_K); D.sync(); ta = time.perf_counter() - t
    S.refactor(); tf = time.perf_counter() - t
    x = S.solve(b, max_refine=0); torch.cuda.synchronize()
    return ta, tf - ta, time.perf_counter() - t
for prep in (False, True, False, True):
    r = np.array([step(prep) for _ in range(7)])
    m = np.median(r, 0)
    print("%d dofs, Newton step (R + K pass, factorisation, one sweep), median of 7: %s the preparation: assembly pass %.1f ms, factorisation %.1f ms, whole step %.1f ms"
          % (A.ndof, "with" if prep else "without", 1e3 * m[0], 1e3 * m[1], 1e3 * m[2]), flush=True)
S.close(); D.close()
