"""Design -> analysis control nets of the moving-intersection shape optimisation (reference: GOLDFISH/utils/bsp_utils.py:516-1230,
class and method names of the reference): the design variables are the control points of a COARSE, low-order B-spline net per optimised patch; constant
linear operators take them to the analysis net -- alignment (design dofs -> coarse net), order elevation, knot refinement -- and further constant
operators give the linear constraints on the design (pinned edges, monotone nets, distances between patches).

The reference builds the elevation and refinement operators by knot insertion / Bezier degree elevation (bsp_utils.py:89-345, 573-619).  Here both are ONE
statement: a spline space S_c (degree p_c, knots U_c) contained in a spline space S_f is embedded by the matrix T with N^c_j = sum_i T_ij N^f_i; collocating
at the Greville abscissae of S_f (where the collocation matrix of S_f is non-singular: Schoenberg-Whitney) gives T = B_f^{-1} B_c exactly, up to round-off.
Control points are flattened u-fastest (i + j * l), as everywhere in this package."""
import numpy as np
import scipy.sparse as sp

from ..splines import basis_ders, find_span


def ij2dof(l, i, j):
    return i + j * l


def spline_degree(knots, first_knot=None):
    """Degree of an open knot vector = multiplicity of its first knot - 1 (bsp_utils.py:39-52)."""
    knots = np.asarray(knots, float)
    k0 = knots[0] if first_knot is None else first_knot
    return int(np.sum(np.abs(knots - k0) < 1e-14) - 1)


def normalize_knots_vector(knots):
    knots = np.asarray(knots, float)
    return (knots - knots[0]) / (knots[-1] - knots[0])


def _collocation(p, knots, pts):
    knots = np.asarray(knots, float)
    n = knots.size - p - 1
    B = np.zeros((len(pts), n))
    for k, x in enumerate(pts):
        s = find_span(n, p, knots, float(x))
        B[k, s - p:s + 1] = basis_ders(s, float(x), p, knots, 0)[0]
    return B


def curve_embedding_matrix(p_c, knots_c, p_f, knots_f):
    """T (n_f x n_c): control points of a curve of the coarse space -> control points of the SAME curve in the fine space (p_f >= p_c, every knot of the
    coarse vector present in the fine one with multiplicity raised by at least p_f - p_c)."""
    knots_c, knots_f = np.asarray(knots_c, float), np.asarray(knots_f, float)
    n_f = knots_f.size - p_f - 1
    grev = np.array([knots_f[i + 1:i + p_f + 1].mean() for i in range(n_f)]) if p_f > 0 else 0.5 * (knots_f[:-1] + knots_f[1:])
    T = np.linalg.solve(_collocation(p_f, knots_f, grev), _collocation(p_c, knots_c, grev))
    T[np.abs(T) < 1e-14] = 0.0
    return T


def _maybe_coo(A, coo):
    return sp.coo_matrix(A) if coo else np.asarray(A)


def surface_order_elevation_operator(p_input, knots_input, p_output, knots_output, coo=True):
    """bsp_utils.py:573-619: (l_out m_out) x (l_in m_in) operator of the tensor-product surface, degrees [p_u, p_v], knots [U, V]."""
    Tu = curve_embedding_matrix(p_input[0], normalize_knots_vector(knots_input[0]), p_output[0], normalize_knots_vector(knots_output[0]))
    Tv = curve_embedding_matrix(p_input[1], normalize_knots_vector(knots_input[1]), p_output[1], normalize_knots_vector(knots_output[1]))
    return _maybe_coo(np.kron(Tv, Tu), coo)


def surface_knot_refine_operator(init_knots, ref_knots, coo=True):
    """bsp_utils.py:516-554: insert the knots ref_knots = [in u, in v] into the surface with knot vectors init_knots (degrees from the end multiplicities)."""
    T = []
    for d in range(2):
        U = np.asarray(init_knots[d], float)
        p = spline_degree(U)
        Uf = np.sort(np.concatenate([U, np.asarray(ref_knots[d], float).ravel()]))
        T.append(curve_embedding_matrix(p, U, p, Uf))
    return _maybe_coo(np.kron(T[1], T[0]), coo)


def surface_cp_align_operator(cp_shape, align_dir, coo=True):
    """bsp_utils.py:647-670: control points take one value along ``align_dir`` (0: along u, the design dofs are the first column i = 0; 1: along v, design
    dofs = first row; [0, 1]: one value for the whole net).  Returns (operator (l m x n_free), free dofs of the full net)."""
    l, m = int(cp_shape[0]), int(cp_shape[1])
    a = np.arange(l * m)
    if align_dir == 0:
        A, free = sp.coo_matrix((np.ones(l * m), (a, a // l)), shape=(l * m, m)), list(range(0, l * m, l))
    elif align_dir == 1:
        A, free = sp.coo_matrix((np.ones(l * m), (a, a % l)), shape=(l * m, l)), list(range(l))
    elif list(np.atleast_1d(align_dir)) == [0, 1]:
        A, free = sp.coo_matrix(np.ones((l * m, 1))), [0]
    else:
        raise ValueError("Undefined align_dir: %r" % (align_dir,))
    return (A if coo else A.toarray()), free


def surface_cp_regu_operator(cp_shape, regu_dir, rev_dir=False, coo=True):
    """bsp_utils.py:706-739: differences of neighbouring control points along ``regu_dir`` (rows ordered i-major, as the reference)."""
    l, m = int(cp_shape[0]), int(cp_shape[1])
    c = -1.0 if rev_dir else 1.0
    rows, cols, vals = [], [], []
    if regu_dir == 0:
        for i in range(l - 1):
            for j in range(m):
                r = i * m + j
                rows += [r, r]; cols += [ij2dof(l, i, j), ij2dof(l, i + 1, j)]; vals += [-c, c]
        shape = ((l - 1) * m, l * m)
    elif regu_dir == 1:
        for i in range(l):
            for j in range(m - 1):
                r = i * (m - 1) + j
                rows += [r, r]; cols += [ij2dof(l, i, j), ij2dof(l, i, j + 1)]; vals += [-c, c]
        shape = (l * (m - 1), l * m)
    else:
        raise ValueError("Undefined regu_dir: %r" % (regu_dir,))
    A = sp.coo_matrix((vals, (rows, cols)), shape=shape)
    return A if coo else A.toarray()


def _edge_dofs(l, m, direction, sides):
    out = []
    for side in sides:
        if direction == 0:
            out += list(range(side * (l - 1), l * m, l))
        elif direction == 1:
            out += list(range(side * l * (m - 1), side * l * (m - 1) + l))
        else:
            raise ValueError("Undefined pin direction: %r" % (direction,))
    return out


class CPSurfDesign2Analysis(object):
    """bsp_utils.py:758-1230.  ``preprocessor``: a goldfish_amd.cpiga2xi.IntersectionData (its ``patches`` are the analysis surfaces)."""

    def __init__(self, preprocessor, opt_field, shopt_surf_inds, shopt_surf_inds_explicit=None):
        self.preprocessor = preprocessor
        self.opt_field = list(opt_field)
        self.shopt_surf_inds = [list(s) for s in shopt_surf_inds]
        self.shopt_surf_inds_explicit = self.shopt_surf_inds if shopt_surf_inds_explicit is None else shopt_surf_inds_explicit
        P = preprocessor.patches
        self.analysis_cp_shapes_all = [(p.n_u, p.n_v) for p in P]
        self.analysis_knots_all = [[np.asarray(p.knots[0], float), np.asarray(p.knots[1], float)] for p in P]
        self.analysis_degree_all = [[p.p, p.q] for p in P]
        self.analysis_cp_all = [p.cp_hom_flat()[:, :3] / p.cp_hom_flat()[:, 3:4] for p in P]
        nf = len(self.opt_field)
        self.analysis_cp_shapes = [[self.analysis_cp_shapes_all[s] for s in self.shopt_surf_inds[f]] for f in range(nf)]
        self.analysis_knots = [[self.analysis_knots_all[s] for s in self.shopt_surf_inds[f]] for f in range(nf)]
        self.analysis_degree = [[self.analysis_degree_all[s] for s in self.shopt_surf_inds[f]] for f in range(nf)]
        self.init_analysis_cp = [np.concatenate([self.analysis_cp_all[s][:, field] for s in self.shopt_surf_inds[f]]) for f, field in enumerate(self.opt_field)]

    # ---- design nets
    def set_init_knots_by_field(self, p_list, knots_list):
        nf = len(self.opt_field)
        self.design_degree = [[list(p) for p in p_list[f]] for f in range(nf)]
        self.design_knots = [[[np.asarray(k[0], float), np.asarray(k[1], float)] for k in knots_list[f]] for f in range(nf)]
        self.cp_coarse_shapes = [[[len(k[0]) - p[0] - 1, len(k[1]) - p[1] - 1] for p, k in zip(self.design_degree[f], self.design_knots[f])] for f in range(nf)]
        self.cp_coarse_sizes = [[s[0] * s[1] for s in self.cp_coarse_shapes[f]] for f in range(nf)]
        self.align_dir_list = [None] * nf
        self.cp_coarse_free_dofs_decate = [[] for _ in range(nf)]
        self.cp_coarse_align_deriv_sub_list = [[] for _ in range(nf)]
        for f in range(nf):
            off = 0
            for n in self.cp_coarse_sizes[f]:
                self.cp_coarse_align_deriv_sub_list[f].append(sp.identity(n, format="coo"))
                self.cp_coarse_free_dofs_decate[f].append(np.arange(n) + off)
                off += n
        self.cp_coarse_free_dofs = [np.concatenate(d) for d in self.cp_coarse_free_dofs_decate]
        self.cp_coarse_align_deriv_list = [sp.block_diag(b, format="coo") for b in self.cp_coarse_align_deriv_sub_list]
        self.order_ele_operator_list, self.knot_refine_operator_list = [None] * nf, [None] * nf
        self.cp_coarse_pin_field, self.cp_coarse_regu_field, self.cp_coarse_dist_field = [], [], []
        self.cp_coarse_dist_deriv_list, self.cp_coarse_regu_deriv_list, self.cp_coarse_pin_deriv_list = [None] * nf, [None] * nf, [None] * nf
        self.cp_coarse_pin_dofs, self.cp_coarse_pin_vals = [[] for _ in range(nf)], [[] for _ in range(nf)]

    def set_init_knots(self, surf_inds, p_list, knots_list):
        """Per-surface lists (bsp_utils.py:884-931): every field takes the entries of its own patches."""
        surf_inds = list(surf_inds)
        self.set_init_knots_by_field([[p_list[surf_inds.index(s)] for s in inds] for inds in self.shopt_surf_inds],
                                     [[knots_list[surf_inds.index(s)] for s in inds] for inds in self.shopt_surf_inds])

    def set_order_elevation_by_field(self, p_list, knots_list):
        nf = len(self.opt_field)
        self.order_ele_degree = [[list(p) for p in p_list[f]] for f in range(nf)]
        self.order_ele_knots = [[[np.asarray(k[0], float), np.asarray(k[1], float)] for k in knots_list[f]] for f in range(nf)]
        for f in range(nf):
            ops = [surface_order_elevation_operator(self.design_degree[f][i], self.design_knots[f][i], self.order_ele_degree[f][i], self.order_ele_knots[f][i])
                   for i in range(len(self.shopt_surf_inds[f]))]
            self.order_ele_operator_list[f] = sp.block_diag(ops, format="coo")
        return self.order_ele_operator_list

    def set_order_elevation(self, surf_inds, p_list, knots_list):
        surf_inds = list(surf_inds)
        return self.set_order_elevation_by_field([[p_list[surf_inds.index(s)] for s in inds] for inds in self.shopt_surf_inds],
                                                 [[knots_list[surf_inds.index(s)] for s in inds] for inds in self.shopt_surf_inds])

    def set_knot_refinement(self):
        """The knots of the analysis surfaces that the elevated design net lacks (bsp_utils.py:955-978); knot vectors compared on [0, 1]."""
        nf = len(self.opt_field)
        self.ref_knots = [[] for _ in range(nf)]
        for f in range(nf):
            ops = []
            for i in range(len(self.shopt_surf_inds[f])):
                ref = []
                for d in range(2):
                    have = list(normalize_knots_vector(self.order_ele_knots[f][i][d]))
                    add = []
                    for k in normalize_knots_vector(self.analysis_knots[f][i][d]):
                        hit = [j for j, h in enumerate(have) if abs(h - k) < 1e-12]
                        if hit:
                            have.pop(hit[0])
                        else:
                            add.append(k)
                    ref.append(np.array(add))
                self.ref_knots[f].append(ref)
                ops.append(surface_knot_refine_operator([normalize_knots_vector(k) for k in self.order_ele_knots[f][i]], ref))
            self.knot_refine_operator_list[f] = sp.block_diag(ops, format="coo")
        return self.knot_refine_operator_list

    def get_init_cp_coarse(self):
        """Least-squares coarse net of the initial analysis control points (bsp_utils.py:1042-1053)."""
        self.init_cp_coarse, self.init_cp_design = [], []
        for f in range(len(self.opt_field)):
            A = (self.knot_refine_operator_list[f].tocsr() @ self.order_ele_operator_list[f].tocsr()).toarray()
            x = np.linalg.lstsq(A, self.init_analysis_cp[f], rcond=None)[0]
            self.init_cp_coarse.append(x)
            self.init_cp_design.append(x.copy())
        return self.init_cp_coarse

    # ---- constraints built into / put onto the design
    def set_cp_align(self, field, align_dir_list):
        f = self.opt_field.index(field)
        self.align_dir_list[f] = list(align_dir_list)
        off = 0
        for i, ad in enumerate(align_dir_list):
            n = self.cp_coarse_sizes[f][i]
            if ad is not None:
                A, free = surface_cp_align_operator(self.cp_coarse_shapes[f][i], ad)
                self.cp_coarse_align_deriv_sub_list[f][i] = A
            else:
                free = list(range(n))
            self.cp_coarse_free_dofs_decate[f][i] = np.asarray(free) + off
            off += n
        self.cp_coarse_free_dofs[f] = np.concatenate(self.cp_coarse_free_dofs_decate[f])
        self.init_cp_design[f] = self.init_cp_coarse[f][self.cp_coarse_free_dofs[f]]
        self.cp_coarse_align_deriv_list[f] = sp.block_diag(self.cp_coarse_align_deriv_sub_list[f], format="coo")
        return self.cp_coarse_align_deriv_list[f]

    def set_cp_pin(self, field, pin_dir0_list, pin_side0_list, pin_dir1_list=None, pin_side1_list=None, pin_dofs=None, pin_vals=None):
        if field not in self.cp_coarse_pin_field:
            self.cp_coarse_pin_field.append(field)
        f = self.opt_field.index(field)
        if pin_dofs is not None:
            cand = list(pin_dofs)
        else:
            cand = []
            for dirs, sides in ((pin_dir0_list, pin_side0_list), (pin_dir1_list, pin_side1_list)):
                if dirs is None:
                    continue
                off = 0
                for i, d in enumerate(dirs):
                    l, m = self.cp_coarse_shapes[f][i]
                    if d is not None:
                        cand += [x + off for x in _edge_dofs(l, m, d, [s for s in (0, 1) if s in sides[i]])]
                    off += self.cp_coarse_sizes[f][i]
            cand = list(np.unique(cand))
        free = self.cp_coarse_free_dofs[f]
        self.cp_coarse_pin_dofs[f] = sorted(set(self.cp_coarse_pin_dofs[f]) | {int(d) for d in cand if d in free})
        self.cp_coarse_pin_vals[f] = list(pin_vals) if pin_vals is not None else list(self.init_cp_coarse[f][self.cp_coarse_pin_dofs[f]])
        pos = {int(d): c for c, d in enumerate(free)}
        cols = [pos[d] for d in self.cp_coarse_pin_dofs[f]]
        self.cp_coarse_pin_deriv_list[f] = sp.coo_matrix((np.ones(len(cols)), (np.arange(len(cols)), cols)), shape=(len(cols), len(free)))
        return self.cp_coarse_pin_deriv_list[f]

    def set_cp_regu(self, field, regu_dir_list, rev_dir=False):
        if field not in self.cp_coarse_regu_field:
            self.cp_coarse_regu_field.append(field)
        f = self.opt_field.index(field)
        rows = []
        for i, rd in enumerate(regu_dir_list):
            if rd is None:
                continue
            shape = list(self.cp_coarse_shapes[f][i])
            if self.align_dir_list[f] is not None and self.align_dir_list[f][i] in (0, 1):
                shape[self.align_dir_list[f][i]] = 1                  # an aligned net has one design point along that direction
            A = surface_cp_regu_operator(shape, rd, rev_dir=rev_dir)
            blocks = [sp.coo_matrix((A.shape[0], len(self.cp_coarse_free_dofs_decate[f][j]))) for j in range(len(self.shopt_surf_inds[f]))]
            blocks[i] = A
            rows.append(blocks)
        self.cp_coarse_regu_deriv_list[f] = sp.bmat(rows, format="coo")
        return self.cp_coarse_regu_deriv_list[f]

    def set_cp_dist(self, field, surf_inds, rev_dir=False):
        if field not in self.cp_coarse_dist_field:
            self.cp_coarse_dist_field.append(field)
        f = self.opt_field.index(field)
        c = -1.0 if rev_dir else 1.0
        free, dec = self.cp_coarse_free_dofs[f], self.cp_coarse_free_dofs_decate[f]
        pos = {int(d): k for k, d in enumerate(free)}
        mats = []
        for a, b in zip(surf_inds[:-1], surf_inds[1:]):
            ia, ib = self.shopt_surf_inds[f].index(a), self.shopt_surf_inds[f].index(b)
            assert len(dec[ia]) == len(dec[ib])
            n = len(dec[ia])
            r = np.arange(n)
            mats.append(sp.coo_matrix((np.concatenate([-c * np.ones(n), c * np.ones(n)]),
                                       (np.concatenate([r, r]), [pos[int(d)] for d in dec[ia]] + [pos[int(d)] for d in dec[ib]])), shape=(n, len(free))))
        self.cp_coarse_dist_deriv_list[f] = sp.vstack(mats, format="coo")
        return self.cp_coarse_dist_deriv_list[f]
