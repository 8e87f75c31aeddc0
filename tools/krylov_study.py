"""Measurement: can preconditioned CG replace the direct solver where its factors do not fit (C5: 708 GB)?  K of C4-family models (the bench
generator: thin NURBS shell patches of 1 cm, penalty coefficient 1e3) is assembled on the device and brought to the host; CG (scipy, FP64) runs with the
preconditioners the pieces at hand allow:
  jacobi / bjacobi      the reference's PETSc_ksp_solve default and its 3 x 3 block form (on the device: goldfish_amd/_krylov.py)
  block-Jacobi          exact factors of the diagonal blocks of a partition into patches / patch groups (what non-overlapping subtree factorisations give)
  additive Schwarz      the same blocks grown by L layers of the control-point graph (one layer covers the penalty coupling across an interface and p control
                        points of the shell), exact subdomain factors
  + coarse              a coarse space of the six rigid-body modes of every block (Galerkin coarse matrix, dense)
Prints iterations to |r| / |b| <= 1e-8 (recursive residual) and the true residual reached, per model size."""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import geometry as G, _lib
from goldfish_amd.model import arrays_from_spec


def model(nx, ny, nel, p=3):
    spec = G.synthetic_shell(nx, ny, nel=nel, p=p, jitter=1)
    th = G.random_thickness(spec)
    A = arrays_from_spec(spec, th)
    D = _lib.DeviceModel(A)
    D.set_thickness(np.concatenate(th)); D.set_u(np.zeros(A.ndof)); D.assemble(_lib.ASM_R | _lib.ASM_K)
    K, b = D.csr(_lib.MAT_K).tocsr(), -D.residual()
    cp_off = np.concatenate([[0], np.cumsum([q.ncp for q in spec.patches])])
    X = np.concatenate([(q.control[:, :, :3] / q.control[:, :, 3:4]).transpose(1, 0, 2).reshape(-1, 3) for q in spec.patches])
    w = np.concatenate([q.control[:, :, 3].T.ravel() for q in spec.patches])
    return spec, D, K, b, cp_off, X, w


def pcg(K, b, Minv, tol=1e-8, maxit=2000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; nb = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        Kp = K @ p; a = rz / (p @ Kp); x += a * p; r -= a * Kp
        if np.linalg.norm(r) < tol * nb:
            return x, it
        z = Minv(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, maxit


def study(nx, ny, nel, group, overlap, coarse, p=3, maxit=2000):
    spec, D, K, b, cp_off, X, w = model(nx, ny, nel, p)
    n, ncp = K.shape[0], K.shape[0] // 3
    out = []
    if group == 0:       # point preconditioners: on the device
        from goldfish_amd._krylov import DevicePCG
        for pc in ("jacobi", "bjacobi"):
            S = DevicePCG(D, pc_type=pc)
            t = time.perf_counter(); S.solve(b, rtol=1e-8, max_it=maxit, check_every=50); dt = time.perf_counter() - t
            out.append("%s (device, %.2f ms / iteration): %d iterations, converged %s, true residual %.1e" % (pc, 1e3 * dt / max(S.iterations, 1), S.iterations, S.converged, S.rel_residual))
        D.close()
        print("%dx%d patches, %d spans, n = %d: " % (nx, ny, nel, n) + "; ".join(out), flush=True)
        return
    D.close()
    blk_of_patch = np.array([(s % nx) // group + ((nx + group - 1) // group) * ((s // nx) // group) for s in range(nx * ny)])
    nblk = int(blk_of_patch.max()) + 1
    blk_of_cp = np.repeat(blk_of_patch, np.diff(cp_off))
    Kc_ = K.tocoo()
    Gcp = sp.csr_matrix((np.ones(Kc_.nnz), (Kc_.row // 3, Kc_.col // 3)), shape=(ncp, ncp))
    lus, idxs, tot = [], [], 0
    for bk in range(nblk):
        mask = (blk_of_cp == bk).astype(float)
        for _ in range(overlap):
            mask = ((Gcp @ mask) > 0).astype(float)
        cps = np.flatnonzero(mask); idx = (3 * cps[:, None] + np.arange(3)).ravel(); idxs.append(idx); tot += idx.size
        lus.append(spl.splu(K[idx][:, idx].tocsc()))
    Z = None
    if coarse:
        rows, cols, vals = [], [], []
        for bk in range(nblk):
            cps = np.flatnonzero(blk_of_cp == bk); d = X[cps] - X[cps].mean(0)
            for i in range(3):
                rows.append(3 * cps + i); cols.append(np.full(cps.size, 6 * bk + i)); vals.append(w[cps])
            for j in range(3):
                e = np.zeros(3); e[j] = 1.0; u = np.cross(e, d) * w[cps, None]
                for i in range(3):
                    rows.append(3 * cps + i); cols.append(np.full(cps.size, 6 * bk + 3 + j)); vals.append(u[:, i])
        Z = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 6 * nblk))
        Kc_inv = np.linalg.pinv((Z.T @ (K @ Z)).toarray(), rcond=1e-12)

    def Minv(r):
        z = np.zeros_like(r)
        for idx, lu in zip(idxs, lus):
            z[idx] += lu.solve(r[idx])
        if Z is not None:
            z += Z @ (Kc_inv @ (Z.T @ r))
        return z
    x, it = pcg(K, b, Minv, maxit=maxit)
    print("%dx%d patches, %d spans, n = %d: %d blocks of %dx%d patches, overlap %d layer(s) (subdomain dofs %.2f n), coarse space %s: %d iterations%s, true residual %.1e"
          % (nx, ny, nel, n, nblk, group, group, overlap, tot / n, "6 rigid-body modes per block" if coarse else "none", it, " (limit)" if it == maxit else "",
             np.linalg.norm(b - K @ x) / np.linalg.norm(b)), flush=True)


if __name__ == "__main__":
    for a in [(4, 4, 8, 0, 0, False), (8, 8, 8, 0, 0, False),
              (4, 4, 8, 1, 0, False), (4, 4, 8, 1, 0, True), (4, 4, 8, 1, 1, True), (4, 4, 8, 1, 2, True), (4, 4, 16, 1, 1, True), (4, 4, 16, 1, 2, True),
              (8, 8, 8, 1, 1, True), (8, 8, 8, 1, 2, True), (8, 8, 8, 2, 2, True)]:
        study(*a)
