"""Diagnostic: C4 assembly time per flag subset (what a Newton iteration / a linearize call costs)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
th = G.random_thickness(spec)
D = _lib.DeviceModel(arrays_from_spec(spec, th))
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
for name, fl in (("R", _lib.ASM_R), ("R+K (Newton iteration)", _lib.ASM_R | _lib.ASM_K), ("K+dRdCP+dRdh (linearize)", _lib.ASM_K | _lib.ASM_DRDCP | _lib.ASM_DRDH),
                 ("dRdCP+dRdh (linearize after a Newton solve: K is current)", _lib.ASM_DRDCP | _lib.ASM_DRDH), ("dRdCP only", _lib.ASM_DRDCP), ("dRdh only", _lib.ASM_DRDH), ("all", _lib.ASM_ALL)):
    for _ in range(2): D.assemble(fl)
    D.sync(); t0 = time.perf_counter()
    for _ in range(4): D.assemble(fl)
    D.sync(); print("%-28s %.2f ms" % (name, (time.perf_counter() - t0) / 4 * 1e3), flush=True)
