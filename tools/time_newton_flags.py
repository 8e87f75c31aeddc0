"""Diagnostic: per-kernel cost of the Newton-iteration pass (R + K only) at C4, for rocprofv3 --stats."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
th = G.random_thickness(spec)
D = _lib.DeviceModel(arrays_from_spec(spec, th))
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
for _ in range(5): D.assemble(_lib.ASM_R | _lib.ASM_K)
D.sync()
