"""Host-side NURBS patch data model for the KL-shell hot path.

Replaces, for this path only, what the reference gets from tIGAr/igakit objects:
an ``ExtractedSpline`` is consumed by GOLDFISH only through its control net,
knot vectors, weights, ``zeroDofs`` and the IGA<->FE extraction matrices
(GOLDFISH/nonmatching_opt.py:45-127); the extraction matrices vanish here because
assembly happens directly in IGA dofs.

Orderings: scalar fields are flattened u-index fastest,
``flat = i + j*n_u`` (GOLDFISH/utils/bsp_utils.py:14-15); vector dofs are node
major ``3*flat + component``.  Control points are stored homogeneous
(w*x, w*y, w*z, w) like tIGAr's ``cpFuncs``.
"""
import numpy as np


# ---------------------------------------------------------------- 1-D B-spline kernels
def find_span(n, p, U, xi):
    """Index i with U[i] <= xi < U[i+1]; the last non-empty span at the right end."""
    if xi >= U[n]:
        i = n - 1
        while i > p and U[i] >= U[i + 1]:
            i -= 1
        return i
    if xi <= U[p]:
        i = p
        while i < n - 1 and U[i] >= U[i + 1]:
            i += 1
        return i
    return int(np.searchsorted(U, xi, side="right") - 1)


def basis_ders(span, xi, p, U, nders=2):
    """Values and derivatives (rows 0..nders) of the p+1 non-zero basis functions
    (Cox-de Boor with the triangular table of The NURBS Book, A2.3)."""
    ndu = np.zeros((p + 1, p + 1))
    left = np.zeros(p + 1)
    right = np.zeros(p + 1)
    ndu[0, 0] = 1.0
    for j in range(1, p + 1):
        left[j] = xi - U[span + 1 - j]
        right[j] = U[span + j] - xi
        saved = 0.0
        for r in range(j):
            ndu[j, r] = right[r + 1] + left[j - r]
            temp = ndu[r, j - 1] / ndu[j, r]
            ndu[r, j] = saved + right[r + 1] * temp
            saved = left[j - r] * temp
        ndu[j, j] = saved
    ders = np.zeros((nders + 1, p + 1))
    ders[0] = ndu[:, p]
    a = np.zeros((2, p + 1))
    for r in range(p + 1):
        s1, s2 = 0, 1
        a[0, 0] = 1.0
        for k in range(1, min(nders, p) + 1):
            d = 0.0
            rk, pk = r - k, p - k
            if r >= k:
                a[s2, 0] = a[s1, 0] / ndu[pk + 1, rk]
                d = a[s2, 0] * ndu[rk, pk]
            j1 = 1 if rk >= -1 else -rk
            j2 = k - 1 if r - 1 <= pk else p - r
            for j in range(j1, j2 + 1):
                a[s2, j] = (a[s1, j] - a[s1, j - 1]) / ndu[pk + 1, rk + j]
                d += a[s2, j] * ndu[rk + j, pk]
            if r <= pk:
                a[s2, k] = -a[s1, k - 1] / ndu[pk + 1, r]
                d += a[s2, k] * ndu[r, pk]
            ders[k, r] = d
            s1, s2 = s2, s1
    fac = float(p)
    for k in range(1, min(nders, p) + 1):
        ders[k] *= fac
        fac *= p - k
    return ders


def open_uniform_knots(nel, p, a=0.0, b=1.0):
    return np.concatenate([np.full(p, a), np.linspace(a, b, nel + 1), np.full(p, b)])


def greville(knots, p):
    n = len(knots) - p - 1
    return np.array([knots[i + 1:i + p + 1].sum() / p for i in range(n)])


def insert_knots_1d(U, p, Pw, new_knots):
    """Boehm knot insertion along axis 0 of the homogeneous net Pw (n, ..., 4)."""
    U = np.asarray(U, float).copy()
    Pw = np.asarray(Pw, float).copy()
    for x in new_knots:
        n = Pw.shape[0]
        k = find_span(n, p, U, x)
        Q = np.empty((n + 1,) + Pw.shape[1:])
        Q[:k - p + 1] = Pw[:k - p + 1]
        Q[k + 1:] = Pw[k:]
        for i in range(k - p + 1, k + 1):
            al = (x - U[i]) / (U[i + p] - U[i])
            Q[i] = al * Pw[i] + (1 - al) * Pw[i - 1]
        U = np.concatenate([U[:k + 1], [x], U[k + 1:]])
        Pw = Q
    return U, Pw


def elevate_bezier_1d(Pw, p, t):
    """Degree-elevate a single Bezier segment (axis 0) t times."""
    Pw = np.asarray(Pw, float)
    for _ in range(t):
        Q = np.empty((p + 2,) + Pw.shape[1:])
        Q[0] = Pw[0]
        Q[p + 1] = Pw[p]
        for i in range(1, p + 1):
            al = i / (p + 1.0)
            Q[i] = al * Pw[i - 1] + (1 - al) * Pw[i]
        Pw, p = Q, p + 1
    return Pw, p


# ---------------------------------------------------------------- patch
class NURBSPatch:
    """One NURBS surface patch (bivariate), the unit the reference calls a "spline".

    ``control`` is (n_u, n_v, 4) homogeneous (w x, w y, w z, w)."""

    def __init__(self, degree, knots, control):
        self.p, self.q = int(degree[0]), int(degree[1])
        self.knots = [np.asarray(knots[0], float), np.asarray(knots[1], float)]
        self.control = np.asarray(control, float).copy()
        assert self.control.shape[0] == len(self.knots[0]) - self.p - 1
        assert self.control.shape[1] == len(self.knots[1]) - self.q - 1
        self.zero_dofs = set()          # local vector dof ids 3*flat + field (tIGAr zeroDofs)

    # -- sizes / flattening
    @property
    def n_u(self):
        return self.control.shape[0]

    @property
    def n_v(self):
        return self.control.shape[1]

    @property
    def ncp(self):
        return self.n_u * self.n_v

    @property
    def nel(self):
        return (len(np.unique(self.knots[0])) - 1, len(np.unique(self.knots[1])) - 1)

    def flat(self, i, j):
        return i + j * self.n_u

    def cp_hom_flat(self):
        """(ncp, 4) homogeneous control points, u-index fastest."""
        return self.control.transpose(1, 0, 2).reshape(-1, 4)

    def set_cp_hom_flat(self, arr, field):
        self.control[:, :, field] = np.asarray(arr).reshape(self.n_v, self.n_u).T

    # -- boundary conditions (tIGAr: scalarSpline.getSideDofs / addZeroDofs)
    def get_side_dofs(self, direction, side, n_layers=1):
        idx = []
        for layer in range(n_layers):
            if direction == 0:
                i = layer if side == 0 else self.n_u - 1 - layer
                idx += [self.flat(i, j) for j in range(self.n_v)]
            else:
                j = layer if side == 0 else self.n_v - 1 - layer
                idx += [self.flat(i, j) for i in range(self.n_u)]
        return idx

    def add_zero_dofs(self, field, scalar_dofs):
        for a in scalar_dofs:
            self.zero_dofs.add(3 * int(a) + int(field))

    # -- evaluation (host side, setup only)
    def _basis(self, xi, nders):
        su = find_span(self.n_u, self.p, self.knots[0], xi[0])
        sv = find_span(self.n_v, self.q, self.knots[1], xi[1])
        du = basis_ders(su, xi[0], self.p, self.knots[0], nders)
        dv = basis_ders(sv, xi[1], self.q, self.knots[1], nders)
        return su, sv, du, dv

    def eval_hom(self, xi, nders=1):
        """Homogeneous surface derivatives d^{k+l} (wX, w)/du^k dv^l, shape (nders+1, nders+1, 4)."""
        su, sv, du, dv = self._basis(xi, nders)
        net = self.control[su - self.p:su + 1, sv - self.q:sv + 1]
        return np.einsum("ki,lj,ijc->klc", du, dv, net)

    def eval(self, xi):
        Aw = self.eval_hom(xi, 0)[0, 0]
        return Aw[:3] / Aw[3]

    def eval_ders(self, xi):
        """X, X_u, X_v (rational)."""
        A = self.eval_hom(xi, 1)
        w = A[0, 0, 3]
        X = A[0, 0, :3] / w
        Xu = (A[1, 0, :3] - A[1, 0, 3] * X) / w
        Xv = (A[0, 1, :3] - A[0, 1, 3] * X) / w
        return X, Xu, Xv

    def invert(self, point, xi0=(0.5, 0.5), tol=1e-13, max_it=50):
        """Closest-point parametric coordinates (Gauss-Newton, clamped to the domain)."""
        xi = np.array(xi0, float)
        lo = np.array([self.knots[0][0], self.knots[1][0]])
        hi = np.array([self.knots[0][-1], self.knots[1][-1]])
        for _ in range(max_it):
            X, Xu, Xv = self.eval_ders(xi)
            r = X - point
            Jm = np.stack([Xu, Xv], 1)
            dxi = np.linalg.lstsq(Jm, -r, rcond=None)[0]
            xi_new = np.clip(xi + dxi, lo, hi)
            if np.abs(xi_new - xi).max() < tol:
                xi = xi_new
                break
            xi = xi_new
        return xi

    # -- refinement
    def refine(self, direction, new_knots):
        if direction == 0:
            U, Pw = insert_knots_1d(self.knots[0], self.p, self.control, new_knots)
            self.knots[0], self.control = U, Pw
        else:
            U, Pw = insert_knots_1d(self.knots[1], self.q, self.control.transpose(1, 0, 2), new_knots)
            self.knots[1], self.control = U, Pw.transpose(1, 0, 2)
        return self

    def elevate_bezier(self, direction, t):
        """Degree elevation of a patch that is a single Bezier segment in ``direction``."""
        if t <= 0:
            return self
        if direction == 0:
            assert self.n_u == self.p + 1
            Pw, p = elevate_bezier_1d(self.control, self.p, t)
            self.control, self.p = Pw, p
            self.knots[0] = open_uniform_knots(1, p, self.knots[0][0], self.knots[0][-1])
        else:
            assert self.n_v == self.q + 1
            Pw, q = elevate_bezier_1d(self.control.transpose(1, 0, 2), self.q, t)
            self.control, self.q = Pw.transpose(1, 0, 2), q
            self.knots[1] = open_uniform_knots(1, q, self.knots[1][0], self.knots[1][-1])
        return self

    def eval_grid(self, us, vs):
        """Surface points X(us[i], vs[j]) -> (len(us), len(vs), 3) (vectorised, setup only)."""
        def bmat(kn, p, n, xs):
            Bm = np.zeros((len(xs), n))
            for r, x in enumerate(xs):
                sp = find_span(n, p, kn, x)
                Bm[r, sp - p:sp + 1] = basis_ders(sp, x, p, kn, 0)[0]
            return Bm
        Bu, Bv = bmat(self.knots[0], self.p, self.n_u, us), bmat(self.knots[1], self.q, self.n_v, vs)
        Aw = np.einsum("sj,rjc->rsc", Bv, np.tensordot(Bu, self.control, (1, 0)), optimize=True)     # two small products, not one five-index contraction
        return Aw[:, :, :3] / Aw[:, :, 3:4]

    def mean_element_size(self):
        """Average physical element edge length (PENGoLINS spline_mesh_size analogue,
        used only to freeze the penalty parameters, nonmatching_opt.py:122-127)."""
        if getattr(self, "_mes", None) is None:
            X = self.eval_grid(np.unique(self.knots[0]), np.unique(self.knots[1]))
            du = np.linalg.norm(X[1:, :, :] - X[:-1, :, :], axis=2)
            dv = np.linalg.norm(X[:, 1:, :] - X[:, :-1, :], axis=2)
            hu = 0.5 * (du[:, 1:] + du[:, :-1])
            hv = 0.5 * (dv[1:, :] + dv[:-1, :])
            self._mes = float(np.mean(0.5 * (hu + hv)))
        return self._mes

    # -- constructors
    @staticmethod
    def bilinear(pts, nel_u, nel_v, p):
        """Degree-p B-spline patch reproducing the bilinear surface through
        pts = [P00, P10, P01, P11] (igakit ``ruled(line, line)`` + elevate + refine,
        GOLDFISH/tests/test_tbeam.py:5-16).  Exact via linear precision: control
        points sit at the Greville abscissae."""
        P00, P10, P01, P11 = [np.asarray(x, float) for x in pts]
        ku, kv = open_uniform_knots(nel_u, p), open_uniform_knots(nel_v, p)
        gu, gv = greville(ku, p), greville(kv, p)
        ctrl = np.zeros((len(gu), len(gv), 4))
        for i, s in enumerate(gu):
            for j, t in enumerate(gv):
                ctrl[i, j, :3] = (1 - s) * (1 - t) * P00 + s * (1 - t) * P10 + (1 - s) * t * P01 + s * t * P11
                ctrl[i, j, 3] = 1.0
        return NURBSPatch((p, p), (ku, kv), ctrl)

    @staticmethod
    def from_function(func, nel_u, nel_v, p, weights=None):
        """Degree-p patch whose control points are func(greville_u, greville_v) (a smooth
        synthetic surface; not an interpolant)."""
        ku, kv = open_uniform_knots(nel_u, p), open_uniform_knots(nel_v, p)
        gu, gv = greville(ku, p), greville(kv, p)
        ctrl = np.zeros((len(gu), len(gv), 4))
        S, T = np.meshgrid(gu, gv, indexing="ij")
        XYZ = func(S, T)
        w = np.ones_like(S) if weights is None else weights(S, T)
        for k in range(3):
            ctrl[:, :, k] = XYZ[k] * w
        ctrl[:, :, 3] = w
        return NURBSPatch((p, p), (ku, kv), ctrl)

    @staticmethod
    def cylinder_sector(R, ang0, ang1, z0, z1, nel_u, nel_v, p, axis="z"):
        """Exact rational cylindrical sector: u along the arc (angle ang0->ang1, radians,
        less than pi), v along the axis (igakit circle + ruled + elevate + refine,
        GOLDFISH/tests/test_slr.py:6-17)."""
        dth = 0.5 * (ang1 - ang0)
        wm = np.cos(dth)
        mid = 0.5 * (ang0 + ang1)
        arc = np.array([[R * np.cos(ang0), R * np.sin(ang0)],
                        [R * np.cos(mid) / wm, R * np.sin(mid) / wm],
                        [R * np.cos(ang1), R * np.sin(ang1)]])
        warc = np.array([1.0, wm, 1.0])
        ctrl = np.zeros((3, 2, 4))
        for i in range(3):
            for j, z in enumerate((z0, z1)):
                ctrl[i, j] = np.array([arc[i, 0], arc[i, 1], z, 1.0]) * warc[i]
        patch = NURBSPatch((2, 1), (open_uniform_knots(1, 2), open_uniform_knots(1, 1)), ctrl)
        patch.elevate_bezier(0, p - 2).elevate_bezier(1, p - 1)
        patch.refine(0, np.linspace(0, 1, nel_u + 1)[1:-1]).refine(1, np.linspace(0, 1, nel_v + 1)[1:-1])
        return patch
