"""Diagnostic: transposed products (dR/dCP_f)^T lam and (dR/dh)^T lam at C4, fixed-order gather vs atomics (GF_ATOMIC_T=1)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
import torch
spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
th = G.random_thickness(spec)
A = arrays_from_spec(spec, th)
D = _lib.DeviceModel(A)
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
D.assemble(); D.sync()
x = torch.rand(A.ndof, dtype=torch.float64, device="cuda"); y = torch.zeros(A.total_cp, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
L = _lib.lib()
for which, name in ((_lib.MAT_DRDCP0 + 2, "(dR/dCP_2)^T"), (_lib.MAT_DRDH, "(dR/dh)^T")):
    for _ in range(3): L.gf_apply_dev(D.h, which, 1, x.data_ptr(), y.data_ptr())
    D.sync(); t0 = time.perf_counter()
    for _ in range(20): L.gf_apply_dev(D.h, which, 1, x.data_ptr(), y.data_ptr())
    D.sync(); print("%-14s %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
