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
# where the work sits: fronts by size class (the library factors fronts of <= 96 blocks in batched launches per tree height, larger ones per front on streams)
sym = _nd.nested_dissection(nb_ptr, nb, X, leaf=int(os.environ.get("GF_ND_LEAF", "256")))
ne, nbd, be, bb = sym.front_dofs()
bt = be + bb
fl = np.array([2.0 * 64 ** 3 * float(np.sum((t - 1 - np.arange(e)) + (t - 1 - np.arange(e)) * (t - np.arange(e)) / 2.0) + e / 3.0) for e, t in zip(be, bt)])
height = np.zeros(sym.nfronts, int)
for t in range(sym.nfronts):
    if sym.parent[t] >= 0: height[sym.parent[t]] = max(height[sym.parent[t]], height[t] + 1)
for lo, hi in ((0, 24), (24, 48), (48, 96), (96, 160), (160, 1 << 30)):
    m = (bt > lo) & (bt <= hi)
    print("fronts of %d < blocks <= %d: %d fronts, %.2f Tflop, eliminated block columns %d" % (lo, hi, m.sum(), fl[m].sum() / 1e12, be[m].sum()), flush=True)
for hgt in range(height.max() + 1):
    m = height == hgt
    print("height %d: %d fronts, blocks %d..%d, eliminated columns %d..%d, %.2f Tflop" % (hgt, m.sum(), bt[m].min(), bt[m].max(), be[m].min(), be[m].max(), fl[m].sum() / 1e12), flush=True)
