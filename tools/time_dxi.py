"""Diagnostic / measurement: gf_penalty_dxi (N3, pen_dxi_kernel) at C4."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
th = G.random_thickness(spec)
A = arrays_from_spec(spec, th)
D = _lib.DeviceModel(A)
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
npts = int(A.if_off[-1])
D.penalty_dxi(npts, 3); D.sync()
t0 = time.perf_counter()
for _ in range(3): B, W = D.penalty_dxi(npts, 3)
print("gf_penalty_dxi: %d mortar vertices, %.1f ms per call incl. the %.2f GB device-to-host copy of the blocks" % (npts, (time.perf_counter() - t0) / 3 * 1e3, B.nbytes / 1e9))
nv = int(A.if_off[1] - A.if_off[0])
D.penalty_dxi(nv, 3, v_first=int(A.if_off[7]))
t0 = time.perf_counter()
for _ in range(10): B, W = D.penalty_dxi(nv, 3, v_first=int(A.if_off[7]))
print("gf_penalty_dxi_range: the %d vertices of one interface, %.3f ms per call (what dRIGAdxi asks for per moving interface)" % (nv, (time.perf_counter() - t0) / 10 * 1e3))
