"""Diagnostic: neighbour counts of the control points of C4 (which rows couple across interfaces, how wide)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
D = _lib.DeviceModel(arrays_from_spec(spec, G.random_thickness(spec)))
rowptr, col = D.pattern(_lib.MAT_DRDCP0)
deg = np.diff(rowptr)[::3]
rp_s, _ = D.pattern(_lib.MAT_DRDH)
degs = np.diff(rp_s)[::3]
cpl = deg > degs
print("control points", deg.size, "with coupling", int(cpl.sum()))
h = np.bincount(np.minimum(deg[cpl], 320) // 16)
for k, n in enumerate(h):
    if n: print("deg %3d..%3d: %d" % (16 * k, 16 * k + 15, n))
print("max", deg.max())
