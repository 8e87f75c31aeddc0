"""Free-form-deformation (FFD) block utilities -- SURVEY.md 8(f) row N2, host side.

Reference: GOLDFISH/utils/ffd_utils.py:35-124 (``CP_FFD_matrix``, ``create_3D_block``) and the
flat ordering of FFD control points ``i + j*l + k*l*m`` (GOLDFISH/nonmatching_opt_ffd.py:6-7,
156-157).  igakit / tIGAr are replaced by the B-spline kernels of goldfish_amd.splines; the FFD
block has the identity geometric mapping (control points at the Greville abscissae scaled to the
limits), exactly what the reference assumes (``ffd_utils.py:37-39``)."""
import numpy as np
from scipy.sparse import coo_matrix

from ..splines import basis_ders, find_span, greville, open_uniform_knots


class FFDBlock:
    """Trivariate B-spline block: ``knots`` (3 arrays), ``control`` (l, m, n, 4), ``degree`` (3)."""

    def __init__(self, knots, control, degree):
        self.knots, self.control, self.degree = [np.asarray(k, float) for k in knots], np.asarray(control, float), list(degree)

    @property
    def shape(self):
        return self.control.shape[0:3]


def create_3D_block(num_els, p, CP_lims):
    """ffd_utils.py:69-124: FFD block with ``num_els`` elements and degree ``p`` per direction whose
    control points span ``CP_lims`` ([[x0,x1],[y0,y1],[z0,z1]]); degenerate ranges are thickened."""
    p_list = list(p) if isinstance(p, (list, tuple)) else [p] * 3
    lims = [list(map(float, l)) for l in CP_lims]
    ranges = [l[1] - l[0] for l in lims]
    for i in range(3):
        if abs(ranges[i]) < 1e-6:
            ranges[i] = np.sort(ranges)[1] * 0.1
            lims[i] = [lims[i][0] - 0.5 * ranges[i], lims[i][1] + 0.5 * ranges[i]]
    knots = [open_uniform_knots(num_els[i], p_list[i], lims[i][0], lims[i][1]) for i in range(3)]
    grev = [greville(knots[i], p_list[i]) for i in range(3)]
    ctrl = np.ones((len(grev[0]), len(grev[1]), len(grev[2]), 4))
    ctrl[..., 0], ctrl[..., 1], ctrl[..., 2] = np.meshgrid(*grev, indexing="ij")
    return FFDBlock(knots, ctrl, p_list)


def CP_FFD_matrix(CP_S, p_V, knots_V, coo=True):
    """ffd_utils.py:35-67: linear operator with ``FFD_mat @ CP_V[:, f] = CP_S[:, f]``: row a holds the
    trivariate basis of the FFD block evaluated at the physical surface control point a."""
    CP_S = np.asarray(CP_S, float)
    p_V = list(p_V) if isinstance(p_V, (list, tuple)) else [p_V] * 3
    n = [len(knots_V[d]) - p_V[d] - 1 for d in range(3)]
    rows, cols, vals = [], [], []
    for a in range(CP_S.shape[0]):
        sp, N = [], []
        for d in range(3):
            x = min(max(CP_S[a, d], knots_V[d][0]), knots_V[d][-1])
            s = find_span(n[d], p_V[d], knots_V[d], x)
            sp.append(s)
            N.append(basis_ders(s, x, p_V[d], knots_V[d], 0)[0])
        for k in range(p_V[2] + 1):
            for j in range(p_V[1] + 1):
                for i in range(p_V[0] + 1):
                    v = N[0][i] * N[1][j] * N[2][k]
                    if v != 0.0:
                        rows.append(a)
                        cols.append((sp[0] - p_V[0] + i) + (sp[1] - p_V[1] + j) * n[0] + (sp[2] - p_V[2] + k) * n[0] * n[1])
                        vals.append(v)
    M = coo_matrix((vals, (rows, cols)), shape=(CP_S.shape[0], n[0] * n[1] * n[2]))
    return M if coo else M.toarray()


def scale_knots(knots, CP):
    """ffd_utils.py:10-33: knot vectors on [0, 1] stretched to the limits of the control points, one per coordinate."""
    nf = len(knots)
    flat = np.asarray(CP, float).reshape(-1, np.asarray(CP).shape[-1])[:, :nf]
    return [np.asarray(knots[f], float) * (flat[:, f].max() - flat[:, f].min()) + flat[:, f].min() for f in range(nf)]


def rationalized_control(nurbs):
    """ffd_utils.py:126-151: physical control points of an object with homogeneous ``control`` (..., 4)."""
    c = np.asarray(nurbs.control, float)
    return c[..., 0:3] / c[..., -1:]


def refine_knot(knot, ref_level=1):
    """ffd_utils.py:154-161: every knot interval (empty ones included, as there) gets its midpoint, ``ref_level`` times."""
    k = np.asarray(knot, float).copy()
    for _ in range(ref_level):
        k = np.sort(np.concatenate([k, 0.5 * (k[:-1] + k[1:])]))
    return k


def update_FFD_block(FFD_block, new_cps, opt_field):
    """ffd_utils.py:348-358: block with the coordinates ``opt_field`` of its control points replaced by ``new_cps``
    (flat order i + j*l + k*l*m)."""
    ctrl = np.asarray(FFD_block.control, float).copy()
    flat = ctrl[..., 0:3].transpose(2, 1, 0, 3).reshape(-1, 3).copy()
    for i, field in enumerate(opt_field):
        flat[:, field] = np.asarray(new_cps[i], float)
    ctrl[..., 0:3] = flat.reshape(ctrl[..., 0:3].transpose(2, 1, 0, 3).shape).transpose(2, 1, 0, 3)
    return FFDBlock(FFD_block.knots, ctrl, FFD_block.degree)
