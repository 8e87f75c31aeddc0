"""Diagnostic (not product): element-kernel time of a library variant (GF_LIB=...) on an 8x8-patch slice of C4."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(8, 8, nel=48, p=int(os.environ.get("GF_P", "3")), jitter=2)
th = G.random_thickness(spec)
D = _lib.DeviceModel(arrays_from_spec(spec, th))
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
import time
for _ in range(3): D.assemble()
D.sync(); ms = []
t0 = time.perf_counter()
for _ in range(10): D.assemble(); ms.append(D.kernel_ms())
D.sync()
print("%-40s element kernel %.3f ms   step %.3f ms" % (os.path.basename(os.environ.get("GF_LIB", "default")), np.mean([m[0] for m in ms]), (time.perf_counter() - t0) * 100))
