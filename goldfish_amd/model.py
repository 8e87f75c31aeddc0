"""Flattening of patches + interfaces into the plain-C ``gf_model_desc``
(include/goldfish_model.h) consumed by libgoldfish_hip.so.

Reference counterparts: the constructor bookkeeping of NonMatchingOpt
(GOLDFISH/nonmatching_opt.py:45-90 nest vectors, sizes) and the data PENGoLINS'
``mortar_meshes_setup(mapping_list, mortar_parametric_coords, penalty_coefficient)``
receives (GOLDFISH/nonmatching_opt.py:422-431; the .npz interface files keep
``mapping_list`` and per-side parametric coordinates).
"""
import ctypes as C
import numpy as np


class Interface:
    """One patch intersection: mortar vertices carrying the parametric coordinates
    of the same physical point on both sides (``intersections_para_coords[i][side]``)."""

    def __init__(self, a, b, xi_a, xi_b):
        self.a, self.b = int(a), int(b)
        self.xi_a = np.ascontiguousarray(xi_a, float).reshape(-1, 2)
        self.xi_b = np.ascontiguousarray(xi_b, float).reshape(-1, 2)
        assert self.xi_a.shape == self.xi_b.shape and self.xi_a.shape[0] >= 2
        n = self.xi_a.shape[0]
        # mortar parameter t in [0,1], uniform vertices (mortar_nel = n-1 elements);
        # vertex quadrature == trapezoid weights (nonmatching_opt.py:26-29 docstring)
        self.wt = np.full(n, 1.0 / (n - 1))
        self.wt[0] *= 0.5
        self.wt[-1] *= 0.5
        # tau = d(xi_A)/dt by second-order differences of the vertex coordinates
        self.tau = np.gradient(self.xi_a, 1.0 / (n - 1), axis=0, edge_order=2 if n > 2 else 1)

    @property
    def npts(self):
        return self.xi_a.shape[0]

    @staticmethod
    def from_endpoints(a, b, ends_a, ends_b, mortar_nel):
        """Straight parametric segments on both sides (the reference's
        ``mortar_mesh_locations`` of GOLDFISH/tests/test_slr.py:116-129)."""
        t = np.linspace(0, 1, mortar_nel + 1)[:, None]
        ea, eb = np.asarray(ends_a, float), np.asarray(ends_b, float)
        return Interface(a, b, ea[0] + t * (ea[1] - ea[0]), eb[0] + t * (eb[1] - eb[0]))


def penalty_parameters(patches, thickness, E, nu, interface, penalty_coefficient):
    """alpha_d, alpha_r of Herrema et al. 2019, frozen at setup:
    alpha_d = alpha * min_s(E h/(1-nu^2)) / h_e, alpha_r = alpha * min_s(E h^3/(12(1-nu^2))) / h_e,
    h_e = mean of the two patches' average element sizes (penalty_method="minimum",
    GOLDFISH/nonmatching_opt.py:424)."""
    mem, ben, he = [], [], []
    for s in (interface.a, interface.b):
        h = float(np.mean(thickness[s]))
        mem.append(E[s] * h / (1 - nu[s] ** 2))
        ben.append(E[s] * h ** 3 / (12 * (1 - nu[s] ** 2)))
        he.append(patches[s].mean_element_size())
    h_e = 0.5 * (he[0] + he[1])
    return penalty_coefficient * min(mem) / h_e, penalty_coefficient * min(ben) / h_e


class gf_model_desc(C.Structure):
    _fields_ = [
        ("n_patches", C.c_int32),
        ("degree", C.POINTER(C.c_int32)), ("ncp", C.POINTER(C.c_int32)),
        ("knot_off", C.POINTER(C.c_int64)), ("knots", C.POINTER(C.c_double)),
        ("cp_off", C.POINTER(C.c_int64)), ("weights", C.POINTER(C.c_double)),
        ("young", C.POINTER(C.c_double)), ("poisson", C.POINTER(C.c_double)),
        ("body_force", C.POINTER(C.c_double)),
        ("n_zero_dofs", C.c_int64), ("zero_dofs", C.POINTER(C.c_int64)),
        ("n_point_loads", C.c_int64), ("pl_dof", C.POINTER(C.c_int64)), ("pl_val", C.POINTER(C.c_double)),
        ("n_interfaces", C.c_int32),
        ("if_patch", C.POINTER(C.c_int32)), ("if_off", C.POINTER(C.c_int64)),
        ("if_xi", C.POINTER(C.c_double)), ("if_tau", C.POINTER(C.c_double)),
        ("if_wt", C.POINTER(C.c_double)), ("if_alpha", C.POINTER(C.c_double)),
        ("n_owned_patches", C.c_int32),
        ("load_proj", C.POINTER(C.c_double)),
        ("pressure", C.POINTER(C.c_double)),
        ("edge_traction", C.POINTER(C.c_double)),
    ]


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class ModelArrays:
    """Owns the NumPy buffers behind a ``gf_model_desc`` (keeps them alive)."""

    def __init__(self, patches, E, nu, body_force=None, interfaces=(), alphas=(),
                 point_loads=(), n_owned=0, load_proj=None, pressure=None, edge_traction=None):
        n = len(patches)
        self.n_patches = n
        self.n_owned = int(n_owned) if n_owned else n
        self.degree = np.array([[p.p, p.q] for p in patches], np.int32).ravel()
        self.ncp = np.array([[p.n_u, p.n_v] for p in patches], np.int32).ravel()
        kn, koff = [], [0]
        for p in patches:
            for d in (0, 1):
                kn.append(p.knots[d])
                koff.append(koff[-1] + len(p.knots[d]))
        self.knots = np.ascontiguousarray(np.concatenate(kn), float)
        self.knot_off = np.array(koff, np.int64)
        self.cp_off = np.concatenate([[0], np.cumsum([p.ncp for p in patches])]).astype(np.int64)
        self.total_cp = int(self.cp_off[-1])
        self.ndof = 3 * self.total_cp
        hom = np.concatenate([p.cp_hom_flat() for p in patches], 0)
        self.weights = np.ascontiguousarray(hom[:, 3])
        self.cp_hom = [np.ascontiguousarray(hom[:, f]) for f in range(3)]
        self.young = np.ascontiguousarray(np.broadcast_to(np.asarray(E, float), (n,)))
        self.poisson = np.ascontiguousarray(np.broadcast_to(np.asarray(nu, float), (n,)))
        bf = np.zeros((n, 3)) if body_force is None else np.asarray(body_force, float).reshape(n, 3)
        self.body_force = np.ascontiguousarray(bf).ravel()
        lp = np.zeros((n, 3)) if load_proj is None else np.asarray(load_proj, float).reshape(n, 3)
        self.load_proj = np.ascontiguousarray(lp).ravel()
        # follower pressure per patch; dead edge tractions [(patch, direction, side, (fx, fy, fz)), ...] -> [n][2 * direction + side][3]
        self.pressure = np.zeros(n) if pressure is None else np.ascontiguousarray(np.broadcast_to(np.asarray(pressure, float), (n,)))
        et = np.zeros((n, 4, 3))
        for s_, d_, side_, f_ in (edge_traction or ()):
            et[int(s_), 2 * int(d_) + int(side_)] += np.asarray(f_, float)
        self.edge_traction = np.ascontiguousarray(et).ravel()
        self.symmetric_K = not np.any(self.pressure != 0.0)              # the load stiffness of a follower pressure is not symmetric
        zd = []
        for s, p in enumerate(patches):
            zd += [3 * int(self.cp_off[s]) + d for d in sorted(p.zero_dofs)]
        self.zero_dofs = np.array(zd, np.int64)
        self.pl_dof = np.array([d for d, _ in point_loads], np.int64)
        self.pl_val = np.array([v for _, v in point_loads], float)
        self.n_interfaces = len(interfaces)
        self.if_patch = np.array([[i.a, i.b] for i in interfaces], np.int32).ravel()
        self.if_off = np.concatenate([[0], np.cumsum([i.npts for i in interfaces])]).astype(np.int64)
        if interfaces:
            self.if_xi = np.ascontiguousarray(np.concatenate([np.hstack([i.xi_a, i.xi_b]) for i in interfaces], 0)).ravel()
            self.if_tau = np.ascontiguousarray(np.concatenate([i.tau for i in interfaces], 0)).ravel()
            self.if_wt = np.ascontiguousarray(np.concatenate([i.wt for i in interfaces]))
            self.if_alpha = np.ascontiguousarray(np.asarray(alphas, float).reshape(-1, 2)).ravel()
        else:
            self.if_xi = np.zeros(0)
            self.if_tau = np.zeros(0)
            self.if_wt = np.zeros(0)
            self.if_alpha = np.zeros(0)
        self.n_gauss_points = int(sum(p.nel[0] * p.nel[1] * (p.p + 1) * (p.q + 1) for p in patches[:self.n_owned]))
        self.n_mortar_points = int(self.if_off[-1])

    def desc(self):
        d = gf_model_desc()
        d.n_patches = self.n_patches
        d.degree, d.ncp = _ptr(self.degree, C.c_int32), _ptr(self.ncp, C.c_int32)
        d.knot_off, d.knots = _ptr(self.knot_off, C.c_int64), _ptr(self.knots, C.c_double)
        d.cp_off, d.weights = _ptr(self.cp_off, C.c_int64), _ptr(self.weights, C.c_double)
        d.young, d.poisson = _ptr(self.young, C.c_double), _ptr(self.poisson, C.c_double)
        d.body_force = _ptr(self.body_force, C.c_double)
        d.n_zero_dofs, d.zero_dofs = len(self.zero_dofs), _ptr(self.zero_dofs, C.c_int64)
        d.n_point_loads = len(self.pl_dof)
        d.pl_dof, d.pl_val = _ptr(self.pl_dof, C.c_int64), _ptr(self.pl_val, C.c_double)
        d.n_interfaces = self.n_interfaces
        d.if_patch, d.if_off = _ptr(self.if_patch, C.c_int32), _ptr(self.if_off, C.c_int64)
        d.if_xi, d.if_tau = _ptr(self.if_xi, C.c_double), _ptr(self.if_tau, C.c_double)
        d.if_wt, d.if_alpha = _ptr(self.if_wt, C.c_double), _ptr(self.if_alpha, C.c_double)
        d.n_owned_patches = self.n_owned
        d.load_proj = _ptr(self.load_proj, C.c_double)
        d.pressure = _ptr(self.pressure, C.c_double)
        d.edge_traction = _ptr(self.edge_traction, C.c_double)
        return d


def point_load_entries(patches, cp_off, point_loads):
    """Expand (patch, xi, field, value) point loads into (global dof, value) pairs:
    dolfin ``PointSource(V.sub(field), Point(xi), value)`` applied to the assembled
    residual (GOLDFISH/nonmatching_opt.py:735-738) acts on the homogeneous test
    function, i.e. with the non-rational basis N_a(xi)."""
    from .splines import basis_ders, find_span
    out = []
    for s, xi, field, val in point_loads:
        P = patches[s]
        su = find_span(P.n_u, P.p, P.knots[0], xi[0])
        sv = find_span(P.n_v, P.q, P.knots[1], xi[1])
        du = basis_ders(su, xi[0], P.p, P.knots[0], 0)[0]
        dv = basis_ders(sv, xi[1], P.q, P.knots[1], 0)[0]
        for jv in range(P.q + 1):
            for ju in range(P.p + 1):
                N = du[ju] * dv[jv]
                if N != 0.0:
                    a = int(cp_off[s]) + P.flat(su - P.p + ju, sv - P.q + jv)
                    out.append((3 * a + field, val * N))
    return out


def arrays_from_spec(spec, thickness=None):
    """ProblemSpec -> ModelArrays (+ per-patch thickness list used to freeze the penalty)."""
    n = len(spec.patches)
    if thickness is None:
        thickness = [np.full(p.ncp, spec.h_th) for p in spec.patches]
    E = np.broadcast_to(np.asarray(spec.E, float), (n,))
    nu = np.broadcast_to(np.asarray(spec.nu, float), (n,))
    alphas = [penalty_parameters(spec.patches, thickness, E, nu, itf, spec.penalty_coefficient)
              for itf in spec.interfaces]
    cp_off = np.concatenate([[0], np.cumsum([p.ncp for p in spec.patches])])
    pls = point_load_entries(spec.patches, cp_off, spec.point_loads)
    return ModelArrays(spec.patches, E, nu, spec.body_force, spec.interfaces, alphas, pls, load_proj=getattr(spec, "load_proj", None),
                       pressure=getattr(spec, "pressure", None), edge_traction=getattr(spec, "edge_traction", None))
