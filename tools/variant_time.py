"""Diagnostic (not product): element-kernel time of a library variant (GF_LIB=...) on an 8x8-patch slice of C4:
full pass (R + K + dR/dCP + dR/dh) and Newton pass (R + K)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(8, 8, nel=48, p=int(os.environ.get("GF_P", "3")), jitter=2)
th = G.random_thickness(spec)
D = _lib.DeviceModel(arrays_from_spec(spec, th))
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
out = []
for fl in (_lib.ASM_ALL, _lib.ASM_R | _lib.ASM_K):
    for _ in range(3): D.assemble(fl)
    D.sync(); D.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(10): D.assemble(fl)
    D.sync()
    out.append(((time.perf_counter() - t0) * 100, D.kernel_ms()[0]))
print("%-28s path %d  full: element kernel %.3f ms  step %.3f ms   Newton: element kernel %.3f ms  step %.3f ms"
      % (os.path.basename(os.environ.get("GF_LIB", "default")) + " " + os.environ.get("GF_TAG", ""), D.assembly_path, out[0][1], out[0][0], out[1][1], out[1][0]), flush=True)
