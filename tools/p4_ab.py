"""Measurement: p = 4 element kernel and step time of a library variant (GF_LIB) on one GPU's share of C5 (128 patches,
53 spans a side): same box, same lease A/B of library versions (the 54.6 -> 59.4 ms difference between r01_v11 and r01_v15)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_fuselage(16, 8, nel=53, p=4, jitter=2)
th = G.random_thickness(spec)
D = _lib.DeviceModel(arrays_from_spec(spec, th))
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
for _ in range(2): D.assemble()
D.sync(); D.kernel_ms()
t0 = time.perf_counter()
for _ in range(5): D.assemble()
D.sync()
ms = (time.perf_counter() - t0) / 5 * 1e3
k, n = D.kernel_ms()
print("%-22s C5 share (128 patches p=4): %.2f ms per step, element kernel %.2f ms per step (%d launches)" % (os.path.basename(os.environ.get("GF_LIB", "HEAD")), ms, k * n / 5, n), flush=True)
