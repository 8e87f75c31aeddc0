"""Diagnostic: symbolic nested dissection of C4's control-point graph with several cut windows (goldfish_amd/_nd.py): flops, tiles, largest front, host time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, _nd, geometry as G
from goldfish_amd.model import arrays_from_spec
n = int(os.environ.get("GF_ND_PATCHES", "16"))
spec = G.synthetic_shell(n, n, nel=48, p=3, jitter=2)
A = arrays_from_spec(spec)
D = _lib.DeviceModel(A)
from goldfish_amd import _solver
rowptr, col = D.pattern(_lib.MAT_K)
nb_ptr, nb = _solver.control_point_graph(rowptr, col)
del rowptr, col
X = np.stack([A.cp_hom[f] / A.weights for f in range(3)], 1)
for cw in (0.0, 0.03, 0.06, 0.1, 0.15):
    t = time.perf_counter(); sym = _nd.nested_dissection(nb_ptr, nb, X, leaf=int(os.environ.get("GF_ND_LEAF", "256")), cut_window=cw); dt = time.perf_counter() - t
    st = sym.stats()
    print("cut_window %.2f: flops %.3e, tiles %d (%.1f GB), largest front %d dofs, fronts %d, host time %.2f s" % (cw, st["flops"], st["tiles"], st["bytes"] / 1e9, st["largest_front_dofs"], st["fronts"], dt), flush=True)
