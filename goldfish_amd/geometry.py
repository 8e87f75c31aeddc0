"""Geometry / problem generators reproducing the reference's fixtures as *inputs*
 without igakit / pythonOCC, plus the synthetic
multi-patch shells of BASELINE.json's configs C2-C5.

Every generator returns a ``ProblemSpec`` (patches, interfaces, material, loads)
that NonMatchingOpt consumes.  Deterministic: seeded numpy.random.default_rng(20241008).
"""
from dataclasses import dataclass, field as dc_field

import numpy as np

from .model import Interface
from .splines import NURBSPatch

SEED = 20241008


@dataclass
class ProblemSpec:
    patches: list
    interfaces: list
    E: float
    nu: float
    h_th: float
    body_force: list                       # per patch (3,)
    point_loads: list = dc_field(default_factory=list)   # (patch, xi, field, value)
    penalty_coefficient: float = 1.0e3
    name: str = ""
    load_proj: list = None                 # per patch (3,): non-zero = load per unit projected area (gf_model_desc.load_proj)
    pressure: list = None                  # per patch: follower pressure (gf_model_desc.pressure)
    edge_traction: list = None             # [(patch, direction, side, (fx, fy, fz)), ...]: dead force per unit length of the edge xi_direction = side


def tbeam_2patch(num_el=10, p=3, load=(0.0, 0.0, 1.0), tip_load=-10.0):
    """Two-patch T-beam of GOLDFISH/tests/test_dRdt.py:45-119 (same as test_tbeam.py):
    flange [-1,1]x[0,20] at z=0 with (num_el/2 x num_el) elements, web x=0,
    z in [0,-2] with (floor((num_el+1)/2) x (num_el+1)) elements; side 0 of
    parametric direction 1 fixed in all three fields (1 layer); E=1e7, nu=0,
    h=0.1; one interface along (0,0,0)-(0,20,0), mortar_nel = 2*(num_el+1)."""
    L, w, h = 20.0, 2.0, 2.0
    ne0, ne1 = num_el, num_el + 1
    pts0 = [[-w / 2, 0, 0], [w / 2, 0, 0], [-w / 2, L, 0], [w / 2, L, 0]]
    pts1 = [[0, 0, 0], [0, 0, -h], [0, L, 0], [0, L, -h]]
    s0 = NURBSPatch.bilinear(pts0, ne0 // 2, ne0, p)
    s1 = NURBSPatch.bilinear(pts1, ne1 // 2, ne1, p)
    for s in (s0, s1):
        for f in range(3):
            s.add_zero_dofs(f, s.get_side_dofs(1, 0, 1))
    itf = Interface.from_endpoints(0, 1, [[0.5, 0.0], [0.5, 1.0]], [[0.0, 0.0], [0.0, 1.0]], 2 * ne1)
    pls = [(0, (1.0, 1.0), 2, tip_load)] if tip_load else []
    return ProblemSpec([s0, s1], [itf], 1.0e7, 0.0, 0.1, [list(load)] * 2, pls, 1.0e3, "tbeam_2patch")


def tbeam_4patch(nels=((23, 29), (24, 29), (23, 30), (25, 31)), p=3, tip_load=-10.0):
    """C2 of BASELINE.json: the T-beam of tests/test_tbeam.py with the
    flange split in two strips along its width and the web split in two along its
    depth; non-matching element counts, about 10k dofs."""
    L, w, h = 20.0, 2.0, 2.0
    quads = [
        [[-w / 2, 0, 0], [0, 0, 0], [-w / 2, L, 0], [0, L, 0]],          # flange left
        [[0, 0, 0], [w / 2, 0, 0], [0, L, 0], [w / 2, L, 0]],            # flange right
        [[0, 0, 0], [0, 0, -h / 2], [0, L, 0], [0, L, -h / 2]],          # web upper
        [[0, 0, -h / 2], [0, 0, -h], [0, L, -h / 2], [0, L, -h]],        # web lower
    ]
    patches = [NURBSPatch.bilinear(q, ne[0], ne[1], p) for q, ne in zip(quads, nels)]
    for s in patches:
        for f in range(3):
            s.add_zero_dofs(f, s.get_side_dofs(1, 0, 1))
    E0, E1 = [[1.0, 0.0], [1.0, 1.0]], [[0.0, 0.0], [0.0, 1.0]]
    pairs = [(0, 1, E0, E1), (2, 3, E0, E1), (0, 2, E0, E1), (1, 2, E1, E1)]
    itfs = [Interface.from_endpoints(a, b, ea, eb, 2 * max(nels[a][1], nels[b][1])) for a, b, ea, eb in pairs]
    pls = [(1, (1.0, 1.0), 2, tip_load)]
    return ProblemSpec(patches, itfs, 1.0e7, 0.0, 0.1, [[0, 0, 0]] * 4, pls, 1.0e3, "tbeam_4patch")


def scordelis_lo_single(num_el=16, p=3):
    """Single-patch Scordelis-Lo roof (geometry/material/load of
    GOLDFISH/tests/test_slr.py:42-49; known answer 0.3006 at the free-edge midpoint)."""
    R, L = 25.0, 50.0
    s = NURBSPatch.cylinder_sector(R, np.radians(50), np.radians(130), 0.0, L, num_el, num_el, p)
    # roof axis is z (igakit circle in the x-y plane); rigid diaphragms at z=0, L fix x and y
    for f in (0, 1):
        s.add_zero_dofs(f, s.get_side_dofs(1, 0) + s.get_side_dofs(1, 1))
    s.add_zero_dofs(2, [0])
    return ProblemSpec([s], [], 4.32e8, 0.0, 0.25, [[0.0, -90.0, 0.0]], [], 1.0e3, "slr_single")


def scordelis_lo_9patch(num_el=6, p=3, nels=None):
    """Nine non-matching NURBS patches of GOLDFISH/tests/test_slr.py:53-129."""
    R, L = 25.0, 50.0
    angles = [50, 80, 100, 130]
    zl = [0, L / 4, 3 * L / 4, L]
    if nels is None:
        nels = [num_el, num_el - 2, num_el - 1, num_el + 2, num_el + 1, num_el + 3, num_el - 1, num_el, num_el - 2]
    bcs = [[1, 0]] * 3 + [[0, 0]] * 3 + [[0, 1]] * 3
    patches = []
    for i in range(9):
        a0, a1 = angles[i % 3], angles[i % 3 + 1]
        z0, z1 = zl[i // 3], zl[i // 3 + 1]
        s = NURBSPatch.cylinder_sector(R, np.radians(a0), np.radians(a1), z0, z1, nels[i], nels[i], p)
        for f in (0, 1):
            for side in (0, 1):
                if bcs[i][side]:
                    s.add_zero_dofs(f, s.get_side_dofs(1, side))
        if i == 0:
            s.add_zero_dofs(2, [0])
        patches.append(s)
    mapping = [[0, 1], [1, 2], [3, 4], [4, 5], [6, 7], [7, 8], [0, 3], [3, 6], [1, 4], [4, 7], [2, 5], [5, 8]]
    hloc = ([[0.0, 1.0], [1.0, 1.0]], [[0.0, 0.0], [1.0, 0.0]])
    vloc = ([[1.0, 0.0], [1.0, 1.0]], [[0.0, 0.0], [0.0, 1.0]])
    itfs = []
    for j, (a, b) in enumerate(mapping):
        loc = vloc if j < 6 else hloc
        itfs.append(Interface.from_endpoints(a, b, loc[0], loc[1], 3 * (nels[a] + nels[b])))
    return ProblemSpec(patches, itfs, 4.32e8, 0.0, 0.25, [[0.0, -90.0, 0.0]] * 9, [], 1.0e3, "slr_9patch")


def pressurised_tube(nels=((5, 2), (6, 3), (4, 2), (7, 3)), p=3, radius=1.0, length=0.5, E=1.0e12, h_th=0.01, pressure=1.0, shape=None):
    """Tube under internal follower pressure -- the load case of the reference's demos_om/shape_opt/tube/tube_shape_opt_wint.py
    (E = 1e12, nu = 0, h = 0.01, ``pressure`` acting along sqrt(det a / det A) a2, :258-262, 303-324): the closed ring as four
    non-matching NURBS sectors of 90 degrees coupled by the penalty method, u_z = 0 (plane strain with nu = 0), the three
    in-plane rigid-body modes removed by u_y = 0 on the generators at 0 and 180 degrees and u_x = 0 on the generator at 90
    degrees (compatible with every doubly symmetric solution).  With u along the arc and v along the axis the normal
    x_,1 x x_,2 points outwards: ``pressure`` > 0 is an internal pressure.  Known answer for the circle (uniform expansion
    u_r; the curvature change B - b of the KL shell is u_r in covariant components, hence the h^2 term):
        linearised about u = 0:  u_r = p r^2 / (E h (1 + h^2 / (12 r^2)) - p r),
        geometrically exact membrane part:  E h eps = p r,  r' = r sqrt(1 + 2 p r / (E h)).
    ``shape(theta)``: optional radius factor for a non-circular start (shape optimisation example)."""
    patches = []
    for k, ne in enumerate(nels):
        a0, a1 = np.radians(90.0 * k), np.radians(90.0 * (k + 1))
        s = NURBSPatch.cylinder_sector(radius, a0, a1, 0.0, length, ne[0], ne[1], p)
        if shape is not None:                                 # scale every control point radially (homogeneous coordinates keep the weight)
            c = s.control
            f = shape(np.arctan2(c[:, :, 1], c[:, :, 0]))
            c[:, :, 0] *= f
            c[:, :, 1] *= f
        for a in range(s.ncp):
            s.add_zero_dofs(2, [a])
        patches.append(s)
    patches[0].add_zero_dofs(1, patches[0].get_side_dofs(0, 0, 1))      # generator at 0 degrees stays on y = 0
    patches[2].add_zero_dofs(1, patches[2].get_side_dofs(0, 0, 1))      # generator at 180 degrees stays on y = 0
    patches[1].add_zero_dofs(0, patches[1].get_side_dofs(0, 0, 1))      # generator at 90 degrees stays on x = 0
    itfs = [Interface.from_endpoints(k, (k + 1) % 4, [[1.0, 0.0], [1.0, 1.0]], [[0.0, 0.0], [0.0, 1.0]], 3 * (nels[k][1] + nels[(k + 1) % 4][1])) for k in range(4)]
    return ProblemSpec(patches, itfs, E, 0.0, h_th, [[0.0, 0.0, 0.0]] * 4, [], 1.0e3, "pressurised_tube", pressure=[pressure] * 4)


def edge_traction_point_loads(patches, s, direction, side, force, ngauss=None):
    """Dead edge traction ``inner(force, rationalize(v)) * ds`` on the edge ``xi_direction = side`` of patch ``s``
    (force per unit physical length) as consistent nodal forces: Gauss points along the edge, each a point load
    (patch, xi, field, value) of ProblemSpec.point_loads with value = force_i |dX/dt| w_gp / W(xi) (the 1/W turns the
    non-rational test function of a point load into the rational one).  Valid while the geometry is fixed; the device path
    carries the same load as ``ProblemSpec.edge_traction`` (gf_model_desc.edge_traction), including its dR/dCP term -- this
    function is the independent statement the tests compare it with."""
    P = patches[s]
    t_dir = 1 - direction                                   # the parameter that runs along the edge
    kn = np.unique(P.knots[t_dir])
    deg = (P.p, P.q)[t_dir]
    ng = ngauss or deg + 1
    gx, gw = np.polynomial.legendre.leggauss(ng)
    fixed = P.knots[direction][0] if side == 0 else P.knots[direction][-1]
    out = []
    for a, b in zip(kn[:-1], kn[1:]):
        for x, w in zip(gx, gw):
            t = 0.5 * (a + b) + 0.5 * (b - a) * x
            xi = [0.0, 0.0]
            xi[direction], xi[t_dir] = fixed, t
            X, Xu, Xv = P.eval_ders(xi)
            jac = np.linalg.norm(Xv if t_dir == 1 else Xu)
            W = P.eval_hom(xi, 0)[0, 0, 3]
            for i in range(3):
                if force[i] != 0.0:
                    out.append((s, tuple(xi), i, float(force[i]) * jac * 0.5 * (b - a) * w / W))
    return out


def plate_6patch(p=3):
    """C1: the six-strip unit plate of demos_csdl_alpha/thickness_opt/geometry/plate_geometry.igs
    (control nets 7 x {11,12,13,12,11,10}: 4 x {8,9,10,9,8,7} cubic elements, strips of
    width 1/6) with material/BC/load data of
    demos_om/thickness_opt/plate/plate_const_th_opt_wint.py:126-137,164-169,235-250.
    The interface parametric coordinates are those of plate_int_data.npz
    (vertical strip edges, mortar_nels = [17,19,19,17,16])."""
    nv = [8, 9, 10, 9, 8, 7]
    patches = []
    for k in range(6):
        x0, x1 = k / 6.0, (k + 1) / 6.0
        patches.append(NURBSPatch.bilinear([[x0, 0, 0], [x1, 0, 0], [x0, 1, 0], [x1, 1, 0]], 4, nv[k], p))
    s0 = patches[0]
    s0.add_zero_dofs(0, s0.get_side_dofs(0, 0, 1))
    for f in (1, 2):
        s0.add_zero_dofs(f, s0.get_side_dofs(0, 0, 2))
    mn = [17, 19, 19, 17, 16]
    itfs = [Interface.from_endpoints(k, k + 1, [[1.0, 0.0], [1.0, 1.0]], [[0.0, 0.0], [0.0, 1.0]], mn[k]) for k in range(5)]
    # the reference loads the xi_0 = 1 edge of the last patch with -100 per unit length (inner(f1 * bdry1, v) * ds,
    # plate_const_th_opt_wint.py:235-250): gf_model_desc.edge_traction (the edge length measure makes it shape dependent: it
    # enters dR/dCP on the device; edge_traction_point_loads is the same load as fixed nodal forces, kept as the cross-check)
    bf = [[0.0, 0.0, 0.0]] * 6
    return ProblemSpec(patches, itfs, 68e9, 0.35, 1.0e-2, bf, [], 1.0e3, "plate_6patch", edge_traction=[(5, 0, 1, (0.0, 0.0, -100.0))])


def wing_16patch_from_interface_data(int_data, nel=10, p=3, seed=SEED):
    """C3 of BASELINE.json: 16 bicubic patches coupled through the interface graph and the
    parametric intersection curves of the reference's ``wing_int_data.npz`` (name2 = mapping_list, name4 =
    per-side parametric coordinates of the mortar vertices, 62 interfaces incl. surface-edge T junctions).
    The true wing geometry needs pythonOCC + the IGES file (out of scope), so the patches are synthetic
    smooth sheets: the penalty then couples the reference's parametric curves on non-coincident surfaces,
    which is physically meaningless but exercises exactly the general-curve code paths (interior curves,
    many interfaces per control point) with real data.  Used as a parity-test case."""
    rng = np.random.default_rng(seed + 3)
    mapping = np.asarray(int_data["name2"], int)
    npatch = int(mapping.max()) + 1
    patches = []
    for s in range(npatch):
        ne = (int(nel + rng.integers(-2, 3)), int(nel + rng.integers(-2, 3)))
        x0, y0, amp, ph = 1.1 * (s % 4), 1.1 * (s // 4), rng.uniform(0.02, 0.08), rng.uniform(0, 2 * np.pi)

        def surf(S, T, x0=x0, y0=y0, amp=amp, ph=ph):
            return x0 + S, y0 + T, amp * np.sin(np.pi * S + ph) * np.cos(np.pi * T)
        patches.append(NURBSPatch.from_function(surf, ne[0], ne[1], p))
    for f in range(3):
        patches[0].add_zero_dofs(f, patches[0].get_side_dofs(0, 0, 2))
    itfs = [Interface(int(a), int(b), np.asarray(int_data["name4"][i][0], float), np.asarray(int_data["name4"][i][1], float))
            for i, (a, b) in enumerate(mapping)]
    return ProblemSpec(patches, itfs, 68e9, 0.35, 2.0e-3, [[0.0, 0.0, -50.0]] * npatch, [], 1.0e3, "wing_16patch_refdata")


def _grid_interfaces(nx, ny, nels, mult=3):
    """Edge-edge interfaces of an nx x ny patch grid, patch index = ix + iy*nx;
    mortar_nel = mult*(nel_a + nel_b) (tests/test_slr.py:124-125)."""
    itfs = []
    for iy in range(ny):
        for ix in range(nx):
            a = ix + iy * nx
            if ix + 1 < nx:
                b = a + 1
                itfs.append(Interface.from_endpoints(a, b, [[1.0, 0.0], [1.0, 1.0]], [[0.0, 0.0], [0.0, 1.0]],
                                                     mult * (nels[a][1] + nels[b][1])))
            if iy + 1 < ny:
                b = a + nx
                itfs.append(Interface.from_endpoints(a, b, [[0.0, 1.0], [1.0, 1.0]], [[0.0, 0.0], [1.0, 0.0]],
                                                     mult * (nels[a][0] + nels[b][0])))
    return itfs


def synthetic_shell(nx=16, ny=16, nel=48, p=3, jitter=2, rational=True, seed=SEED, mortar_mult=3):
    """C4 of BASELINE.json: nx x ny grid of degree-p patches on the doubly
    curved shell z = 0.1 sin(pi x) sin(pi y) over [0,nx]x[0,ny], nel +- jitter spans per
    side (non-matching), weights in [0.9,1.1] on every other patch (true NURBS),
    E=68e9, nu=0.35.  Patch (ix,iy) covers [ix,ix+1]x[iy,iy+1]."""
    rng = np.random.default_rng(seed)
    patches, nels = [], []
    for iy in range(ny):
        for ix in range(nx):
            ne = (int(nel + rng.integers(-jitter, jitter + 1)), int(nel + rng.integers(-jitter, jitter + 1)))
            nels.append(ne)

            def surf(S, T, ix=ix, iy=iy):
                X, Y = ix + S, iy + T
                return X, Y, 0.1 * np.sin(np.pi * X) * np.sin(np.pi * Y)

            wf = None
            if rational and (ix + iy) % 2 == 1:
                ph = rng.uniform(0, 2 * np.pi, 2)

                def wf(S, T, ph=ph):
                    return 1.0 + 0.1 * np.sin(2 * np.pi * S + ph[0]) * np.cos(2 * np.pi * T + ph[1])
            patches.append(NURBSPatch.from_function(surf, ne[0], ne[1], p, wf))
    # clamp the x = 0 edge (all fields, 2 layers) so the stiffness matrix is regular
    for iy in range(ny):
        s = patches[iy * nx]
        for f in range(3):
            s.add_zero_dofs(f, s.get_side_dofs(0, 0, 2))
    itfs = _grid_interfaces(nx, ny, nels, mortar_mult)
    return ProblemSpec(patches, itfs, 68e9, 0.35, 1.0e-2, [[0.0, 0.0, -1.0e3]] * (nx * ny), [], 1.0e3,
                       "synthetic_shell_%dx%d_p%d" % (nx, ny, p))


def synthetic_fuselage(nx=32, ny=32, nel=53, p=4, jitter=2, seed=SEED, mortar_mult=3, R=2.0, length=16.0):
    """C5 of BASELINE.json: nx (circumferential, over 270 degrees so the skin stays open)
    x ny (axial) degree-p patches on a cylinder; B-spline approximations of the sectors
    (control points on the cylinder at the Greville abscissae)."""
    rng = np.random.default_rng(seed + 5)
    patches, nels = [], []
    span = 1.5 * np.pi
    for iy in range(ny):
        for ix in range(nx):
            ne = (int(nel + rng.integers(-jitter, jitter + 1)), int(nel + rng.integers(-jitter, jitter + 1)))
            nels.append(ne)

            def surf(S, T, ix=ix, iy=iy):
                th = span * (ix + S) / nx
                return R * np.cos(th), R * np.sin(th), length * (iy + T) / ny
            patches.append(NURBSPatch.from_function(surf, ne[0], ne[1], p))
    for ix in range(nx):
        s = patches[ix]
        for f in range(3):
            s.add_zero_dofs(f, s.get_side_dofs(1, 0, 2))
    itfs = _grid_interfaces(nx, ny, nels, mortar_mult)
    return ProblemSpec(patches, itfs, 68e9, 0.35, 1.0e-2, [[0.0, -1.0e3, 0.0]] * (nx * ny), [], 1.0e3,
                       "synthetic_fuselage_%dx%d_p%d" % (nx, ny, p))


def with_double_knots(spec, every=2):
    """Inserts every ``every``-th interior knot of every patch a second time (both directions): C^(p-2) lines inside the patches -- still
    C1 for p = 3 --, i.e. consecutive elements whose control-point windows are two rows / columns apart.  The surfaces, and with them the
    interface data, are unchanged."""
    for patch in spec.patches:
        for d in (0, 1):
            U = np.asarray(patch.knots[d]); p = patch.p if d == 0 else patch.q
            interior = np.unique(U[p + 1:-p - 1])
            patch.refine(d, [float(x) for x in interior[::every]])
    return spec


def random_thickness(spec, seed=SEED, lo=0.8, hi=1.2):
    """Per-control-point thickness h ~ U(lo,hi)*h_th."""
    rng = np.random.default_rng(seed + 1)
    return [spec.h_th * rng.uniform(lo, hi, s.ncp) for s in spec.patches]


def smooth_displacement(spec, amplitude, seed=SEED):
    """Smooth pseudo-random displacement field (amplitude ~ 0.5 h) so that the geometric
    tangent terms are exercised."""
    rng = np.random.default_rng(seed + 2)
    k = rng.uniform(0.5, 1.5, (3, 3))
    ph = rng.uniform(0, 2 * np.pi, (3, 3))
    out = []
    for s in spec.patches:
        hom = s.cp_hom_flat()
        X = hom[:, :3] / hom[:, 3:4]
        U = np.stack([amplitude * np.sin(k[c, 0] * X[:, 0] + ph[c, 0]) * np.cos(k[c, 1] * X[:, 1] + ph[c, 1])
                      * np.cos(k[c, 2] * X[:, 2] + ph[c, 2]) for c in range(3)], 1)
        out.append((U * hom[:, 3:4]).ravel())        # IGA dof = w_a * (NURBS coefficient)
    return np.concatenate(out)
