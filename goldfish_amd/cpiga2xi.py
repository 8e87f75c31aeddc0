"""CPIGA2Xi -- implicit relation between the control points of intersecting patches and the parametric
coordinates of the mortar vertices of their intersection curves ("moving intersections";
reference: GOLDFISH/cpiga2xi.py:18-790, same attribute and method names).

For every differentiated intersection with n mortar vertices the 4n unknowns
    xi_flat_sub = [xiA_0 (2), ..., xiA_{n-1} (2), xiB_0 (2), ..., xiB_{n-1} (2)]
satisfy 4n equations (residual_sub, cpiga2xi.py:401-492):
    3n      F_A(xiA_i) - F_B(xiB_i) = 0                      the two pre-images are the same physical point
    n - 2   |F(x_{i+1}) - F(x_i)|^2 - |F(x_i) - F(x_{i-1})|^2 = 0   equal physical spacing on one side
    2       one parametric coordinate of each end vertex is pinned to the patch edge it started on.
The reference evaluates F and dF/dxi through pythonOCC surfaces (D0/D1, :346-375) and tIGAr B-spline bases;
here they come from goldfish_amd.splines.NURBSPatch (rational; identical for the reference's B-spline surfaces).
Host-side NumPy: the systems have 4n unknowns per intersection.

``preprocessor`` is an ``IntersectionData`` (below): the fields of PENGoLINS' OCCPreprocessing that the
reference reads (BSpline_surfs_data, mapping_list, intersections_para_coords, mortar_nels, diff_int_inds,
intersections_type, diff_int_edge_cons).
"""
from dataclasses import dataclass, field as dc_field

import numpy as np
from scipy.optimize import fsolve
from scipy.sparse import bmat, coo_matrix


@dataclass
class IntersectionData:
    """What CPIGA2Xi reads from the reference's ``preprocessor`` (cpiga2xi.py:27-62)."""
    patches: list                              # NURBSPatch per surface            (BSpline_surfs_data)
    mapping_list: list                         # [s_ind0, s_ind1] per intersection (mapping_list)
    intersections_para_coords: list            # [xiA (n,2), xiB (n,2)] per intersection
    mortar_nels: list = None                   # n - 1 per intersection
    diff_int_inds: list = None                 # intersections to differentiate (default: all)
    intersections_type: list = None            # ['surf-surf'] | ['surf-edge', 's-d.v'] | ['edge-surf', 's-d.v']
    diff_int_edge_cons: list = dc_field(default_factory=list)   # 'side-dir.val' per differentiated intersection

    def __post_init__(self):
        self.intersections_para_coords = [[np.asarray(c[0], float).reshape(-1, 2), np.asarray(c[1], float).reshape(-1, 2)]
                                          for c in self.intersections_para_coords]
        self.num_intersections_all = len(self.mapping_list)
        if self.mortar_nels is None:
            self.mortar_nels = [c[0].shape[0] - 1 for c in self.intersections_para_coords]
        if self.diff_int_inds is None:
            self.diff_int_inds = list(range(self.num_intersections_all))
        if self.intersections_type is None:
            self.intersections_type = [["surf-surf"] for _ in range(self.num_intersections_all)]


    # ---- the reference's intersection cache (PENGoLINS preprocessor.save_intersections_data / load_intersections_data,
    # demos_om/thickness_opt/plate/plate_const_th_opt_wint.py:187-193): an .npz with the keys name1..name6 =
    # num_intersections, mapping_list (n x 2), physical coordinates per intersection, parametric coordinates per intersection
    # (2 x (npts x 2)), a length per intersection, mortar_nels
    def physical_coords(self):
        return [np.array([self.patches[a].eval(x) for x in c[0]]) for (a, b), c in zip(self.mapping_list, self.intersections_para_coords)]

    def save_intersections_data(self, filename):
        phys = self.physical_coords()
        para = np.empty(len(self.mapping_list), dtype=object)
        ph = np.empty(len(self.mapping_list), dtype=object)
        for i, c in enumerate(self.intersections_para_coords):
            para[i] = [np.asarray(c[0]), np.asarray(c[1])]
            ph[i] = phys[i]
        lengths = np.array([np.linalg.norm(np.diff(x, axis=0), axis=1).sum() for x in phys])
        np.savez(filename, name1=self.num_intersections_all, name2=np.asarray(self.mapping_list, dtype=int), name3=ph, name4=para,
                 name5=lengths, name6=np.asarray(self.mortar_nels, dtype=int))

    @classmethod
    def load_intersections_data(cls, filename, patches, **kw):
        d = np.load(filename, allow_pickle=True)
        mapping = [[int(a), int(b)] for a, b in d["name2"]]
        para = [[np.asarray(d["name4"][i][0], float), np.asarray(d["name4"][i][1], float)] for i in range(len(mapping))]
        out = cls(patches=patches, mapping_list=mapping, intersections_para_coords=para, mortar_nels=[int(x) for x in d["name6"]], **kw)
        if int(d["name1"]) != out.num_intersections_all:
            raise ValueError("load_intersections_data: name1 does not match the number of interfaces in name2")
        return out


class CPIGA2Xi(object):

    def __init__(self, preprocessor, opt_surf_inds, opt_field, num_edge_pts=None):
        self.preprocessor = preprocessor
        self.num_field, self.para_dim, self.num_end_pts, self.num_sides = 3, 2, 2, 2
        self.num_intersections = preprocessor.num_intersections_all
        self.mortar_nels = preprocessor.mortar_nels
        self.mortar_pts = [nel + 1 for nel in self.mortar_nels]
        self.mapping_list = preprocessor.mapping_list
        self.implicit_edge = False
        self.diff_int_inds = list(preprocessor.diff_int_inds)
        self.diff_int_types = [preprocessor.intersections_type[i] for i in self.diff_int_inds]
        self.opt_field, self.opt_surf_inds = opt_field, opt_surf_inds

        surf = []
        for ind in self.diff_int_inds:                                # cpiga2xi.py:64-71
            for s in self.mapping_list[ind]:
                if s not in surf:
                    surf.append(int(s))
        self.int_surf_inds = list(np.sort(surf))
        self.num_int_surfs = len(self.int_surf_inds)
        self.surfs = [preprocessor.patches[i] for i in self.int_surf_inds]

        # control points of the surfaces of interest (homogeneous coefficients, u-index fastest; :93-118)
        self.cp_shapes = [(P.n_u, P.n_v) for P in self.surfs]
        self.cp_sizes = [P.ncp for P in self.surfs]
        self.cp_size_global = int(np.sum(self.cp_sizes))
        self.cp_flat_inds = [int(np.sum(self.cp_sizes[:i])) for i in range(self.num_int_surfs)] + [self.cp_size_global]
        self.cps_flat = [P.cp_hom_flat()[:, :3].copy() for P in self.surfs]
        self.cp_flat_global = np.concatenate(self.cps_flat, axis=0)
        self._w = [P.cp_hom_flat()[:, 3].copy() for P in self.surfs]

        # parametric coordinates (:124-149)
        self.diff_int_num_pts = [self.mortar_nels[i] + 1 for i in self.diff_int_inds]
        self.xi_size_global = int(np.sum(self.diff_int_num_pts) * self.num_sides * self.para_dim)
        self.xis, self.xis_flat, self.xi_sizes = [], [], []
        for int_ind, g in enumerate(self.diff_int_inds):
            self.xis.append([preprocessor.intersections_para_coords[g][side] for side in range(2)])
            self.xis_flat.append(np.concatenate([self.xis[int_ind][0].ravel(), self.xis[int_ind][1].ravel()]))
            self.xi_sizes.append(self.xis_flat[int_ind].size)
        self.xi_flat_global = np.concatenate(self.xis_flat)
        self.xi_flat_inds = [int(np.sum(self.xi_sizes[:i])) for i in range(len(self.diff_int_inds))] + [self.xi_size_global]

        # pinned end coordinates (:151-207): of the candidate coordinates of an end vertex the one closest to 0 or 1
        self.end_xi_ind = np.zeros((len(self.diff_int_inds), self.num_end_pts), dtype="int32")
        self.end_xi_val = np.zeros((len(self.diff_int_inds), self.num_end_pts))
        for int_ind, g in enumerate(self.diff_int_inds):
            num_pts = self.diff_int_num_pts[int_ind]
            init = preprocessor.intersections_para_coords[g]
            para_dir_list = self._candidate_dirs(self.diff_int_types[int_ind])
            for end_ind in (0, 1):
                err0, err1, inds = [], [], []
                for side in (0, 1):
                    for para_dir in para_dir_list[side]:
                        c = init[side][-end_ind][para_dir]
                        err0.append(c)
                        err1.append(1 - c)
                        inds.append(side * 2 * num_pts + end_ind * 2 * (num_pts - 1) + para_dir)
                if np.min(err0) < np.min(err1):
                    self.end_xi_ind[int_ind, end_ind] = inds[int(np.argmin(err0))]
                    self.end_xi_val[int_ind, end_ind] = 0.
                else:
                    self.end_xi_ind[int_ind, end_ind] = inds[int(np.argmin(err1))]
                    self.end_xi_val[int_ind, end_ind] = 1.
        self.get_surf_avg_normal_dir()
        self.num_edge_pts = num_edge_pts
        self.get_diff_intersections_edge_cons_info(num_edge_pts=num_edge_pts)

    @staticmethod
    def _edge_dir(int_type):
        s = int_type[1]
        return int(s[s.index('.') - 1])

    def _candidate_dirs(self, int_type):
        if int_type[0] == 'surf-edge':
            return [[0, 1], [1] if self._edge_dir(int_type) == 0 else [0]]
        if int_type[0] == 'edge-surf':
            return [[1] if self._edge_dir(int_type) == 0 else [0], [0, 1]]
        return [[0, 1], [0, 1]]

    def _deriv_side(self, int_ind):
        return 1 if self.diff_int_types[int_ind][0] == 'edge-surf' else 0

    # ---- setup helpers -----------------------------------------------------------------------
    def get_surf_avg_normal_dir(self):
        """Dominant global direction of each surface's average normal on a 17 x 17 grid (:210-233)."""
        pts = np.linspace(0, 1, 17)
        self.int_surf_avg_normal_list, self.int_surf_avg_normal_dir = [], []
        for k, P in enumerate(self.surfs):
            lo, hi = np.array([P.knots[0][0], P.knots[1][0]]), np.array([P.knots[0][-1], P.knots[1][-1]])
            normals = []
            for a in pts:
                for b in pts:
                    _, v = self.dFdxi(k, lo + (hi - lo) * np.array([a, b]))
                    n = np.cross(v[:, 0], v[:, 1])
                    normals.append(n / np.linalg.norm(n))
            avg = np.average(normals, axis=0)
            self.int_surf_avg_normal_list.append(avg)
            self.int_surf_avg_normal_dir.append(int(np.argmax(np.abs(avg))))

    def get_diff_intersections_edge_cons_info(self, num_edge_pts):
        """Dofs / values of the coordinates that stay on a patch edge (:236-304)."""
        self.int_edge_cons_dofs_list_full, self.int_edge_cons_vals_list_full, self.int_edge_cons_local_dofs_list_full = [], [], []
        self.int_edge_cons_dofs_list, self.int_edge_cons_vals_list, self.int_edge_cons_local_dofs_list = [], [], []
        if num_edge_pts is not None and not isinstance(num_edge_pts, list):
            num_edge_pts = [num_edge_pts] * len(self.diff_int_inds)
        for i, g in enumerate(self.diff_int_inds):
            int_type = self.preprocessor.intersections_type[g]
            if int_type[0] not in ('surf-edge', 'edge-surf'):
                continue
            ind = self.preprocessor.diff_int_edge_cons[i]
            side, para_dir, edge_val = int(ind[ind.index('-') - 1]), int(ind[ind.index('-') + 1]), int(ind[ind.index('.') + 1])
            half = self.xi_sizes[i] // 2
            start_local = side * half + (1 if para_dir == 1 else 0)
            end_local = (side + 1) * half
            local = np.arange(start_local, end_local, self.para_dim)
            dofs = (local + self.xi_flat_inds[i]).astype('int32')
            vals = np.ones(dofs.size) * edge_val
            self.int_edge_cons_dofs_list_full.append(dofs)
            self.int_edge_cons_vals_list_full.append(vals)
            self.int_edge_cons_local_dofs_list_full.append(local)
            n_keep = dofs.size if num_edge_pts is None else num_edge_pts[i]
            keep = np.linspace(0, dofs.size - 1, n_keep, dtype='int32')
            self.int_edge_cons_dofs_list.append(dofs[keep])
            self.int_edge_cons_vals_list.append(vals[keep])
            self.int_edge_cons_local_dofs_list.append(local[keep])
        if self.int_edge_cons_dofs_list:
            self.int_edge_cons_dofs = np.concatenate(self.int_edge_cons_dofs_list).astype('int32')
            self.int_edge_cons_vals = np.concatenate(self.int_edge_cons_vals_list)
        else:
            self.int_edge_cons_dofs, self.int_edge_cons_vals = [], []
        cons = set(int(d) for d in self.int_edge_cons_dofs)
        self.int_xi_free_dofs = [i for i in range(self.xi_size_global) if i not in cons]
        return self.int_edge_cons_dofs, self.int_edge_cons_vals

    def local_int_surf_inds(self, int_ind):
        """:387-399."""
        s0, s1 = self.mapping_list[self.diff_int_inds[int_ind]]
        return self.int_surf_inds.index(s0), self.int_surf_inds.index(s1)

    # ---- control points ----------------------------------------------------------------------
    def update_CPs(self, cp_flat_single_field, field):
        """New values of coordinate ``field`` for the optimised surfaces, in opt_surf_inds order (:316-333)."""
        opt = self.opt_surf_inds[self.opt_field.index(field)]
        cp_flat_single_field = np.asarray(cp_flat_single_field, float).ravel()
        off_int = off_opt = 0
        for i, s in enumerate(self.int_surf_inds):
            if s in opt:
                self.cp_flat_global[off_int:off_int + self.cp_sizes[i], field] = \
                    cp_flat_single_field[off_opt:off_opt + self.cp_sizes[i]]
                off_opt += self.cp_sizes[i]
            off_int += self.cp_sizes[i]
        for i in range(self.num_int_surfs):
            self.cps_flat[i] = self.cp_flat_global[self.cp_flat_inds[i]:self.cp_flat_inds[i + 1], :]

    # ---- surface evaluation --------------------------------------------------------------------
    def _nodes_evals(self, k, xi, nders):
        """Support control-point ids and rational basis values / first derivatives at xi."""
        P = self.surfs[k]
        su, sv, du, dv = P._basis(xi, nders)
        iu, iv = np.arange(su - P.p, su + 1), np.arange(sv - P.q, sv + 1)
        nodes = (iu[:, None] + iv[None, :] * P.n_u).ravel()
        w = self._w[k][nodes]
        N = (du[0][:, None] * dv[0][None, :]).ravel()
        W = N @ w
        R = N / W
        if nders == 0:
            return nodes, R, None
        Nu, Nv = (du[1][:, None] * dv[0][None, :]).ravel(), (du[0][:, None] * dv[1][None, :]).ravel()
        Ru, Rv = (Nu - R * (Nu @ w)) / W, (Nv - R * (Nv @ w)) / W
        return nodes, R, np.stack([Ru, Rv], 1)

    def F(self, int_surf_ind, xi, cp_flat_sub=None):
        """:361-373."""
        cp = self.cps_flat[int_surf_ind] if cp_flat_sub is None else cp_flat_sub
        nodes, R, _ = self._nodes_evals(int_surf_ind, xi, 0)
        return cp[nodes].T @ R

    def dFdxi(self, int_surf_ind, xi):
        """Position and its 3 x 2 parametric Jacobian (:375-385)."""
        nodes, R, dR = self._nodes_evals(int_surf_ind, xi, 1)
        cp = self.cps_flat[int_surf_ind][nodes]
        return cp.T @ R, cp.T @ dR

    def dFdCP(self, int_surf_ind, xi, field):
        """d F / d (coordinate ``field`` of every control point): 3 x ncp, one non-zero row (:377-385)."""
        deriv = np.zeros((self.num_field, self.cp_sizes[int_surf_ind]))
        nodes, R, _ = self._nodes_evals(int_surf_ind, xi, 0)
        deriv[field, nodes] = R
        return deriv

    # ---- residual ----------------------------------------------------------------------------
    def residual_sub(self, int_ind, xi_flat_sub):
        """:401-492."""
        num_pts = self.diff_int_num_pts[int_ind]
        xi = np.asarray(xi_flat_sub, float).reshape(-1, self.para_dim)
        res = np.zeros(xi.size)
        k0, k1 = self.local_int_surf_inds(int_ind)
        int_type = self.diff_int_types[int_ind]
        normal_dir = None
        if self.implicit_edge and int_type[0] in ('surf-edge', 'edge-surf'):
            normal_dir = self.int_surf_avg_normal_dir[k0 if int_type[0] == 'surf-edge' else k1]
            cons_dof, cons_val = self.int_edge_cons_local_dofs_list[int_ind], self.int_edge_cons_vals_list[int_ind]
        for i in range(num_pts):
            res[3 * i:3 * i + 3] = self.F(k0, xi[i]) - self.F(k1, xi[i + num_pts])
            if normal_dir is not None:
                res[3 * i + normal_dir] = xi_flat_sub[cons_dof[i]] - cons_val[i]
        side = self._deriv_side(int_ind)
        k, off = (k0, 0) if side == 0 else (k1, num_pts)
        pts = [self.F(k, xi[off + i]) for i in range(num_pts)]
        for i in range(1, num_pts - 1):
            d1, d2 = pts[i] - pts[i - 1], pts[i + 1] - pts[i]
            res[i + 3 * num_pts - 1] = d2 @ d2 - d1 @ d1
        res[-2] = xi_flat_sub[self.end_xi_ind[int_ind, 0]] - self.end_xi_val[int_ind, 0]
        res[-1] = xi_flat_sub[self.end_xi_ind[int_ind, 1]] - self.end_xi_val[int_ind, 1]
        return res

    def residual(self, xi_flat):
        """:494-501."""
        xi_flat = np.asarray(xi_flat, float)
        return np.concatenate([self.residual_sub(i, xi_flat[self.xi_flat_inds[i]:self.xi_flat_inds[i + 1]])
                               for i in range(len(self.diff_int_inds))])

    def solve_xi(self, xi_flat_init, rtol=1e-5, max_iter=200):
        """Root of the coupled system (scipy fsolve with the analytic Jacobian, :503-566)."""
        x0 = np.asarray(xi_flat_init, float)
        if np.abs(self.residual(x0)).max() < 1e-13:                       # already a root (MINPACK would only warn about "no progress")
            return x0.copy()
        xi, info, ier, msg = fsolve(self.residual, x0=x0, fprime=lambda x: np.asarray(self.dRdxi(x, coo=False)), full_output=True, xtol=1e-14)
        if np.abs(self.residual(xi)).max() > 1e-8:
            raise RuntimeError("CPIGA2Xi.solve_xi: no intersection found (%s)" % msg)
        return xi

    # ---- derivatives -------------------------------------------------------------------------
    def dRdxi_sub(self, int_ind, xi_flat_sub, coo=True):
        """:569-660."""
        num_pts = self.diff_int_num_pts[int_ind]
        k0, k1 = self.local_int_surf_inds(int_ind)
        int_type = self.diff_int_types[int_ind]
        xi = np.asarray(xi_flat_sub, float).reshape(-1, self.para_dim)
        D = np.zeros((xi.size, xi.size))
        lc = 2 * num_pts
        normal_dir = None
        if self.implicit_edge and int_type[0] in ('surf-edge', 'edge-surf'):
            normal_dir = self.int_surf_avg_normal_dir[k0 if int_type[0] == 'surf-edge' else k1]
            cons_dof = self.int_edge_cons_local_dofs_list[int_ind]
        FA = [self.dFdxi(k0, xi[i]) for i in range(num_pts)]
        FB = [self.dFdxi(k1, xi[i + num_pts]) for i in range(num_pts)]
        for i in range(num_pts):
            D[3 * i:3 * i + 3, 2 * i:2 * i + 2] = FA[i][1]
            D[3 * i:3 * i + 3, lc + 2 * i:lc + 2 * i + 2] = -FB[i][1]
            if normal_dir is not None:
                D[3 * i + normal_dir] = 0.
                D[3 * i + normal_dir, cons_dof[i]] = 1.
        side = self._deriv_side(int_ind)
        Fs, off = (FA, 0) if side == 0 else (FB, lc)
        ur = 3 * num_pts
        for i in range(1, num_pts - 1):
            (Fl, dFl), (Fi, dFi), (Fr, dFr) = Fs[i - 1], Fs[i], Fs[i + 1]
            D[i + ur - 1, off + 2 * (i - 1):off + 2 * i] = 2 * (Fi - Fl) @ dFl
            D[i + ur - 1, off + 2 * i:off + 2 * (i + 1)] = -2 * (Fr - Fl) @ dFi
            D[i + ur - 1, off + 2 * (i + 1):off + 2 * (i + 2)] = 2 * (Fr - Fi) @ dFr
        D[-2, self.end_xi_ind[int_ind, 0]] = 1.
        D[-1, self.end_xi_ind[int_ind, 1]] = 1.
        return coo_matrix(D) if coo else D

    def dRdxi(self, xi_flat, coo=False):
        """Block-diagonal over the intersections (:662-674)."""
        xi_flat = np.asarray(xi_flat, float)
        n = len(self.diff_int_inds)
        blocks = [[None] * n for _ in range(n)]
        for i in range(n):
            blocks[i][i] = self.dRdxi_sub(i, xi_flat[self.xi_flat_inds[i]:self.xi_flat_inds[i + 1]], coo=True)
        full = bmat(blocks, format='coo')
        return full if coo else full.toarray()

    def dRdCP_sub(self, int_ind, xi_flat_sub, field, coo=True):
        """[dR/dCP_A, dR/dCP_B] for coordinate ``field`` (:676-743)."""
        xi = np.asarray(xi_flat_sub, float).reshape(-1, self.para_dim)
        num_pts = self.diff_int_num_pts[int_ind]
        ks = self.local_int_surf_inds(int_ind)
        int_type = self.diff_int_types[int_ind]
        normal_dir = None
        if self.implicit_edge and int_type[0] in ('surf-edge', 'edge-surf'):
            normal_dir = self.int_surf_avg_normal_dir[ks[0] if int_type[0] == 'surf-edge' else ks[1]]
        D = [np.zeros((xi.size, self.cp_sizes[ks[0]])), np.zeros((xi.size, self.cp_sizes[ks[1]]))]
        for i in range(num_pts):
            for side, sign in ((0, 1.), (1, -1.)):
                D[side][3 * i:3 * i + 3, :] = sign * self.dFdCP(ks[side], xi[i + num_pts * side], field)
                if normal_dir is not None:
                    D[side][3 * i + normal_dir] = 0.
        side = self._deriv_side(int_ind)
        k, off = ks[side], side * num_pts
        Fp = [self.F(k, xi[off + i]) for i in range(num_pts)]
        dFp = [self.dFdCP(k, xi[off + i], field) for i in range(num_pts)]
        for i in range(1, num_pts - 1):
            D[side][3 * num_pts + i - 1, :] = 2 * ((Fp[i + 1] - Fp[i]) @ (dFp[i + 1] - dFp[i]) - (Fp[i] - Fp[i - 1]) @ (dFp[i] - dFp[i - 1]))
        return [coo_matrix(m) for m in D] if coo else D

    def dRdCP(self, xi_flat, field, coo=True):
        """Columns: the optimised surfaces of ``field`` in opt_surf_inds order (:745-790)."""
        xi_flat = np.asarray(xi_flat, float)
        opt = list(self.opt_surf_inds[self.opt_field.index(field)])
        n = len(self.diff_int_inds)
        blocks = [[None] * len(opt) for _ in range(n)]
        for i in range(n):
            sub = self.dRdCP_sub(i, xi_flat[self.xi_flat_inds[i]:self.xi_flat_inds[i + 1]], field, coo=True)
            k0, k1 = self.local_int_surf_inds(i)
            for k, mat in ((k0, sub[0]), (k1, sub[1])):
                s = self.int_surf_inds[k]
                if s in opt:
                    blocks[i][opt.index(s)] = mat
            if all(b is None for b in blocks[i]):
                blocks[i][0] = coo_matrix((self.xi_sizes[i], self.preprocessor.patches[opt[0]].ncp))
        for j, s in enumerate(opt):                                   # an optimised surface without intersection: empty column block
            if all(blocks[i][j] is None for i in range(n)):
                blocks[0][j] = coo_matrix((self.xi_sizes[0], self.preprocessor.patches[s].ncp))
        full = bmat(blocks, format='coo')
        return full if coo else full.toarray()
