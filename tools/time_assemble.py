import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
th = G.random_thickness(spec)
D = _lib.DeviceModel(arrays_from_spec(spec, th))
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
for _ in range(2): D.assemble()
D.sync(); t0 = time.perf_counter()
for _ in range(5): D.assemble()
D.sync(); print(os.path.basename(os.environ.get("GF_LIB", "default")), "step %.2f ms, element kernel %.2f" % ((time.perf_counter() - t0) / 5 * 1e3, D.kernel_ms()[0]))
