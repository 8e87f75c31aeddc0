"""Analysis (not product, runs without a GPU): the fronts of the nested dissection per tree height for a patch grid of C4's size (16 x 16 patches of 51 x 51 control points, penalty coupling reaching
four rows across an interface) -- how many fronts, how many 64-dof blocks they eliminate and hold, their work and their tiles split into the factor part (L) and the Schur part (S)."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _nd, _dsolver
n, pw = 816, 51          # C4: 16 x 16 patches of 51 control points per side
ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
idx = ii * n + jj
rows, cols = [], []
for di in range(-4, 5):
    for dj in range(-4, 5):
        i2, j2 = ii + di, jj + dj
        ok = (i2 >= 0) & (i2 < n) & (j2 >= 0) & (j2 < n)
        ok &= ((abs(di) <= 3) | ((ii // pw) != (i2 // pw))) & ((abs(dj) <= 3) | ((jj // pw) != (j2 // pw)))
        rows.append(idx[ok]); cols.append((i2 * n + j2)[ok])
rows, cols = np.concatenate(rows), np.concatenate(cols)
o = np.lexsort((cols, rows)); rows, cols = rows[o], cols[o]
nb_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n * n))]).astype(np.int64)
X = np.stack([ii.ravel(), jj.ravel()], 1).astype(float)
sym = _nd.nested_dissection_native(nb_ptr, cols.astype(np.int32), X, leaf=128)[0]
ne, nbd, be, bb = sym.front_dofs()
bt = be + bb
height = np.zeros(sym.nfronts, int)
for t in range(sym.nfronts):
    p = sym.parent[t]
    if p >= 0: height[p] = max(height[p], height[t] + 1)
work = 2.0 * 64**3 * _dsolver.front_work(sym)
tiles = bt * (bt + 1) // 2
print("fronts %d, total %.2f Tflop, %.1f GB tiles" % (sym.nfronts, work.sum()/1e12, tiles.sum()*32768/1e9))
for h in range(height.max()+1):
    m = height == h
    Lt = (be[m] * bt[m] - be[m] * (be[m] - 1) // 2)
    print("height %2d: %5d fronts, elim blocks mean %.1f max %d, total blocks mean %.1f max %d, %.3f Tflop, tiles %.1f GB (L part %.1f GB, S part %.1f GB)" % (
        h, m.sum(), be[m].mean(), be[m].max(), bt[m].mean(), bt[m].max(), work[m].sum()/1e12, tiles[m].sum()*32768/1e9, Lt.sum()*32768/1e9, (tiles[m]-Lt).sum()*32768/1e9))
