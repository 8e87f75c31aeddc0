"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libkl_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (the product path under goldfish_amd/ never does)."""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAT_K, MAT_DRDCP0, MAT_DRDCP1, MAT_DRDCP2, MAT_DRDH = range(5)


def build(force=False):
    so = os.path.join(_HERE, "libkl_oracle.so")
    src = os.path.join(_HERE, "kl_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libkl_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libkl_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        L.gfo_create.restype = C.c_void_p
        L.gfo_create.argtypes = [C.c_void_p]
        L.gfo_destroy.argtypes = [C.c_void_p]
        for name in ("gfo_total_cp", "gfo_num_gauss_points", "gfo_num_mortar_points"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        L.gfo_set_cp.argtypes = [C.c_void_p, C.c_int, dp]
        L.gfo_set_thickness.argtypes = [C.c_void_p, dp]
        L.gfo_set_u.argtypes = [C.c_void_p, dp]
        L.gfo_nnz.restype = C.c_int64
        L.gfo_nnz.argtypes = [C.c_void_p, C.c_int]
        L.gfo_pattern.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.gfo_residual.argtypes = [C.c_void_p, dp]
        L.gfo_assemble.argtypes = [C.c_void_p, dp, dp, dp, dp, dp]
        L.gfo_functionals.argtypes = [C.c_void_p, dp] + [dp] * 9 + [C.c_int]
        L.gfo_penalty_point.argtypes = [dp, dp, dp, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp]
        L.gfo_eval_point.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, dp, dp]
        L.gfo_compliance.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, dp, C.c_int]
        L.gfo_stress_forms.argtypes = [C.c_void_p, C.c_int, C.c_double, dp, C.c_double, C.c_int] + [dp] * 7 + [C.c_int]
        L.gfo_shape_regu.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp]
        L.gfo_set_quadrature.argtypes = [C.c_void_p, C.c_int, dp, dp, dp]
        L.gfo_num_threads.restype = C.c_int
        L.gfo_set_num_threads.argtypes = [C.c_int]
        L.gfo_set_strain_mode.argtypes = [C.c_int]
        L.gfo_get_strain_mode.restype = C.c_int
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


class strain_mode:
    """Context manager: how the oracle EVALUATES the strains and the penalty's rotation measures (process-wide switch of kl_oracle.c).  0 (default) = differences
    of the two configurations' metric / curvature coefficients, the arithmetic of the reference's UFL forms; 1 = from the displacement derivatives without
    cancellation, what the HIP kernels evaluate since round 5 (kl_point.hpp: kl_strains, pen_rot_measures).  Same quantities, different round-off."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        self.old = lib().gfo_get_strain_mode()
        lib().gfo_set_strain_mode(self.mode)
        return self

    def __exit__(self, *a):
        lib().gfo_set_strain_mode(self.old)


def two_triangle_rule(quad_deg, scheme="fiat"):
    """(x, y, w) on the unit square of the rule FEniCS applies to one knot span in the reference (SURVEY.md App. A.1,
    [ext-recall]: tIGAr's extraction mesh has 2 triangles per span -- dolfin's default "right" diagonal, (0,0)-(1,1) -- and
    FFC integrates a form of estimated/declared degree ``quad_deg`` with FIAT's default scheme: for degree > 6 the collapsed
    Gauss-Jacobi rule with m = (quad_deg + 2) // 2 points per direction (Gauss-Legendre x Gauss-Jacobi(1, 0), exact to degree
    2m - 1 on the triangle), for degree 6 the symmetric 12-point rule of Dunavant).  ``scheme="collapsed"`` forces the
    collapsed rule for every degree.  quad_deg in the reference: 3p (tests/test_tbeam.py:31), 2p (tests/test_slr.py:37),
    4p (demos_om/thickness_opt/plate/plate_const_th_opt_wint.py:139-144)."""
    from scipy.special import roots_jacobi
    if scheme == "fiat" and quad_deg == 6:
        a1, a2, (b1, b2, b3) = 0.249286745170910, 0.063089014491502, (0.053145049844817, 0.310352451033784, 0.636502499121399)
        pts = [(a1, a1), (1 - 2 * a1, a1), (a1, 1 - 2 * a1), (a2, a2), (1 - 2 * a2, a2), (a2, 1 - 2 * a2),
               (b1, b2), (b2, b1), (b1, b3), (b3, b1), (b2, b3), (b3, b2)]
        wts = [0.116786275726379] * 3 + [0.050844906370207] * 3 + [0.082851075618374] * 6
        X, Wt = np.array(pts), 0.5 * np.array(wts)
    else:
        m = (quad_deg + 2) // 2
        s, ws = roots_jacobi(m, 0.0, 0.0)
        t, wt = roots_jacobi(m, 1.0, 0.0)
        S, T = np.meshgrid(s, t, indexing="ij")
        X = np.stack([0.25 * (1 + S) * (1 - T), 0.5 * (1 + T)], -1).reshape(-1, 2)          # reference triangle (0,0), (1,0), (0,1)
        Wt = (ws[:, None] * wt[None, :] / 8.0).ravel()
    lo = np.stack([X[:, 0] + X[:, 1], X[:, 1]], 1)               # (0,0), (1,0), (1,1): (x, y) -> (x + y, y)
    up = np.stack([X[:, 0], X[:, 0] + X[:, 1]], 1)               # (0,0), (0,1), (1,1) mirrored: (x, y) -> (x, x + y)
    P = np.concatenate([lo, up])
    return np.ascontiguousarray(P[:, 0]), np.ascontiguousarray(P[:, 1]), np.concatenate([Wt, Wt])


class Oracle:
    """CPU restatement of RIGA / dRIGAduIGA / dRIGAdCPIGA / dRIGAdh_th + functionals."""

    def __init__(self, arrays, thickness=None, u=None):
        self.arrays = arrays
        self._desc = arrays.desc()
        self.h = C.c_void_p(lib().gfo_create(C.byref(self._desc)))
        if not self.h:
            raise RuntimeError("gfo_create failed")
        self.total_cp, self.ndof = arrays.total_cp, arrays.ndof
        for f in range(3):
            self.set_cp(f, arrays.cp_hom[f])
        if thickness is not None:
            self.set_thickness(thickness)
        if u is not None:
            self.set_u(u)
        self._pat = {}

    def __del__(self):
        if getattr(self, "h", None):
            lib().gfo_destroy(self.h)
            self.h = None

    def set_cp(self, field, v):
        v = np.ascontiguousarray(v, float)
        assert v.size == self.total_cp
        lib().gfo_set_cp(self.h, field, _dp(v))

    def set_thickness(self, v):
        v = np.ascontiguousarray(v, float)
        assert v.size == self.total_cp
        lib().gfo_set_thickness(self.h, _dp(v))

    def set_u(self, v):
        v = np.ascontiguousarray(v, float)
        assert v.size == self.ndof
        lib().gfo_set_u(self.h, _dp(v))

    def set_quadrature(self, rule=None):
        """rule = (x, y, w) on the unit square for the shell integrals (two_triangle_rule(...)); None: tensor Gauss."""
        if rule is None:
            lib().gfo_set_quadrature(self.h, 0, None, None, None)
        else:
            x, y, w = (np.ascontiguousarray(v, float) for v in rule)
            lib().gfo_set_quadrature(self.h, x.size, _dp(x), _dp(y), _dp(w))

    def pattern(self, which):
        if which not in self._pat:
            nnz = lib().gfo_nnz(self.h, which)
            rowptr = np.zeros(self.ndof + 1, np.int64)
            col = np.zeros(nnz, np.int32)
            lib().gfo_pattern(self.h, which, rowptr.ctypes.data_as(C.POINTER(C.c_int64)),
                              col.ctypes.data_as(C.POINTER(C.c_int32)))
            self._pat[which] = (rowptr, col)
        return self._pat[which]

    def residual(self):
        R = np.zeros(self.ndof)
        lib().gfo_residual(self.h, _dp(R))
        return R

    def assemble(self, K=True, dRdCP=(0, 1, 2), dRdh=True):
        vals = {}
        if K:
            vals[MAT_K] = np.zeros(lib().gfo_nnz(self.h, MAT_K))
        for f in dRdCP:
            vals[MAT_DRDCP0 + f] = np.zeros(lib().gfo_nnz(self.h, MAT_DRDCP0))
        if dRdh:
            vals[MAT_DRDH] = np.zeros(lib().gfo_nnz(self.h, MAT_DRDH))
        lib().gfo_assemble(self.h, *[_dp(vals.get(w)) for w in range(5)])
        return vals

    def csr(self, which, vals):
        rowptr, col = self.pattern(which)
        ncol = self.ndof if which == MAT_K else self.total_cp
        return sp.csr_matrix((vals, col, rowptr), shape=(self.ndof, ncol))

    def functionals(self, apply_bcs=True):
        out = np.zeros(3)
        g = dict(dWdu=np.zeros(self.ndof),
                 dWdcp=[np.zeros(self.total_cp) for _ in range(3)], dWdh=np.zeros(self.total_cp),
                 dVdcp=[np.zeros(self.total_cp) for _ in range(3)], dVdh=np.zeros(self.total_cp))
        lib().gfo_functionals(self.h, _dp(out), _dp(g["dWdu"]), *[_dp(x) for x in g["dWdcp"]], _dp(g["dWdh"]),
                              *[_dp(x) for x in g["dVdcp"]], _dp(g["dVdh"]), int(apply_bcs))
        g.update(Wint=out[0], volume=out[1], Wpen=out[2])
        return g

    def compliance(self, forces, apply_bcs=True):
        f = np.ascontiguousarray(forces, float).ravel()
        out, dCdu = np.zeros(1), np.zeros(self.ndof)
        dCdcp = [np.zeros(self.total_cp) for _ in range(3)]
        lib().gfo_compliance(self.h, _dp(f), _dp(out), _dp(dCdu), *[_dp(x) for x in dCdcp], int(apply_bcs))
        return dict(C=out[0], dCdu=dCdu, dCdcp=dCdcp)

    def stress_forms(self, mode, rho, m_list, sgn=1.0, measure=0, apply_bcs=True, npatch=None):
        """Per-patch aggregation forms of the von Mises stress and their gradients (gfo_stress_forms)."""
        ml = np.ascontiguousarray(m_list, float)
        I, vmax = np.zeros(ml.size), np.zeros(ml.size)
        dIdu, dIdh = np.zeros(self.ndof), np.zeros(self.total_cp)
        dIdcp = [np.zeros(self.total_cp) for _ in range(3)]
        lib().gfo_stress_forms(self.h, int(mode), float(rho), _dp(ml), float(sgn), int(measure), _dp(I), _dp(vmax), _dp(dIdu),
                               *[_dp(x) for x in dIdcp], _dp(dIdh), int(apply_bcs))
        return dict(I=I, vmax=vmax, dIdu=dIdu, dIdcp=dIdcp, dIdh=dIdh)

    def shape_regu(self, field, cp0, coef):
        """Shape regularisation term of the eVTOL demo and its gradient wrt the three coordinate fields (gfo_shape_regu)."""
        cp0, coef = np.ascontiguousarray(cp0, float), np.ascontiguousarray(coef, float)
        val = np.zeros(1)
        dcp = [np.zeros(self.total_cp) for _ in range(3)]
        lib().gfo_shape_regu(self.h, int(field), _dp(cp0), _dp(coef), _dp(val), *[_dp(x) for x in dcp])
        return dict(value=val[0], dcp=dcp)

    def eval_point(self, patch, xi):
        X, U = np.zeros(3), np.zeros(3)
        lib().gfo_eval_point(self.h, patch, float(xi[0]), float(xi[1]), _dp(X), _dp(U))
        return X, U


def penalty_point(y, Y, tau, ad, ar, dt):
    y, Y, tau = (np.ascontiguousarray(a, float) for a in (y, Y, tau))
    en = np.zeros(1)
    g, Hyy, HyY = np.zeros(18), np.zeros((18, 18)), np.zeros((18, 12))
    lib().gfo_penalty_point(_dp(y), _dp(Y), _dp(tau), ad, ar, dt, _dp(en), _dp(g), _dp(Hyy), _dp(HyY))
    return en[0], g, Hyy, HyY
