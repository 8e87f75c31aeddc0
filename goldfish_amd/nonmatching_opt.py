"""NonMatchingOpt -- the problem surface of GOLDFISH/nonmatching_opt.py on top of the HIP path.

Same class / method names and the same state-update, boundary-condition and ordering
conventions as the reference (file:line cited per method), with three deliberate
differences forced by the missing FEniCS stack:
  * ``splines`` are goldfish_amd.splines.NURBSPatch objects (not tIGAr ExtractedSplines);
  * UFL residual forms are replaced by a declarative load spec: ``set_residuals`` takes one
    :class:`SVKResidual` per patch (body force per unit reference area), because only
    ``SVK_residual(...) - inner(f, v)*dx`` is ever used (tests/*.py, demos_om/**);
  * matrices come back as ``scipy.sparse.csr_matrix`` and vectors as ``numpy.ndarray``
    (the reference returns petsc4py nest Mat/Vec).
All arithmetic happens in libgoldfish_hip.so; there is no CPU fallback.
"""
from dataclasses import dataclass

import os

import numpy as np
import scipy.sparse as sp

from . import _lib
from .model import Interface, ModelArrays, penalty_parameters, point_load_entries


@dataclass
class SVKResidual:
    """Declarative stand-in for ``SVK_residual(spline, u, v, E, nu, h, dWext)`` with
    ``dWext = inner(f, rationalize(v)) * dx`` (GOLDFISH/tests/test_dRdt.py:100-110) plus the two other source terms the reference's
    demos use (follower ``pressure``, ``edge_tractions``; both shape dependent: they enter dR/dCP on the device, the pressure also K).  ``projected`` = a direction d makes
    the load act per unit projected area, ``f cos(beta)`` with ``cos(beta) = d . A2`` -- the source term of
    demos_om/shape_opt/arch/arch_shape_opt_wint.py:294-301 (``force = -load * inner(e_z, A2) e_z``)."""
    body_force: tuple = (0.0, 0.0, 0.0)
    projected: tuple = (0.0, 0.0, 0.0)
    pressure: float = 0.0          # follower pressure: dWext = p sqrt(det a / det A) a2 . z dA (demos_om/shape_opt/tube/tube_shape_opt_wint.py:303-324)
    edge_tractions: tuple = ()     # ((direction, side, (fx, fy, fz)), ...): dead force per unit length on the edge xi_direction = side,
                                   # dWext = f . z ds (demos_om/thickness_opt/plate/plate_const_th_opt_wint.py:235-250)


@dataclass
class PointSource:
    """``PointSource(spline.V.sub(field), Point(xi), value)`` (GOLDFISH/tests/test_tbeam.py:113-119)."""
    xi: tuple
    field: int
    value: float


class NonMatchingOpt:
    """Base class to set up optimisation problems of non-matching shell structures
    (reference: GOLDFISH/nonmatching_opt.py:7, constructor :12-127)."""

    nsd = 3

    def __init__(self, splines, E, h_th, nu, int_V_family='CG', int_V_degree=1,
                 int_dx_metadata=None, contact=None, comm=None, device=0):
        if contact is not None:
            raise NotImplementedError("shell contact (ShNAPr) is outside the device path (DESIGN.md, out of scope)")
        self.splines = list(splines)
        self.num_splines = len(self.splines)
        # ``comm`` (nonmatching_opt.py:12-15, 35-37: the MPI communicator every nest vector lives on): None or a one-rank group = one process, one GPU.
        # torch.distributed itself (the default group) or one of its process groups: the patches are sharded over the ranks (sharding.ShardedDeviceModel: each
        # rank assembles the rows of its own patches on its own GPU; vectors in and out are replicated, as the reference's comm.allgather leaves them,
        # utils/opt_utils.py:41-54).  ``device``: the GPU of this process (under torchrun: LOCAL_RANK).
        self.comm = comm
        self._dist = self._group = None
        self.rank, self.world = 0, 1
        if comm is not None:
            import torch.distributed as dist
            group = None if comm is dist else comm
            if not dist.is_initialized():
                raise RuntimeError("NonMatchingOpt(comm=...): torch.distributed is not initialised (init_process_group first)")
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            if self.world > 1:
                self._dist, self._group = dist, group
        self.device = device
        n = self.num_splines
        self.E = [float(e) for e in np.broadcast_to(np.asarray(E, float), (n,))]
        self.nu = [float(v) for v in np.broadcast_to(np.asarray(nu, float), (n,))]
        if np.isscalar(h_th):
            h_th = [h_th] * n
        assert len(h_th) == n
        self.h_th = [np.full(s.ncp, float(h)) if np.isscalar(h) else np.asarray(h, float).copy()
                     for s, h in zip(self.splines, h_th)]
        self.opt_shape = False
        self.opt_field = []
        self.opt_thickness = False
        self.var_thickness = False
        self.use_aero_pressure = False
        # nest-vector bookkeeping (nonmatching_opt.py:45-65)
        self.vec_scalar_iga_dof_list = [s.ncp for s in self.splines]
        self.vec_iga_dof_list = [3 * s.ncp for s in self.splines]
        self.vec_scalar_iga_dof = int(sum(self.vec_scalar_iga_dof_list))
        self.vec_iga_dof = int(sum(self.vec_iga_dof_list))
        self.cp_off = np.concatenate([[0], np.cumsum(self.vec_scalar_iga_dof_list)]).astype(np.int64)
        self.u_iga = np.zeros(self.vec_iga_dof)
        self.cp_iga = [np.concatenate([s.cp_hom_flat()[:, f] for s in self.splines]) for f in range(3)]
        self.init_cp_iga = None
        self.residuals = [SVKResidual() for _ in range(n)]
        self.point_sources = None
        self.point_source_inds = None
        self.mapping_list = []
        self.interfaces = []
        self.num_intersections = 0
        self.mortar_nels = None
        self.penalty_coefficient = 1000.0
        self._dev = None
        self._touch()

    # ------------------------------------------------------------------ setup (PENGoLINS surface)
    def create_mortar_meshes(self, mortar_nels):
        self.mortar_nels = list(mortar_nels)
        self.num_intersections = len(self.mortar_nels)

    def mortar_meshes_setup(self, mapping_list, mortar_parametric_coords, penalty_coefficient=1000,
                            transfer_mat_deriv=1, penalty_method="minimum"):
        """nonmatching_opt.py:422-431.  ``mortar_parametric_coords[i][side]`` is either the
        (npts, 2) vertex coordinates (the .npz interface files) or the two end points of a
        straight parametric segment, discretised with ``mortar_nels[i]`` elements
        (GOLDFISH/tests/test_slr.py:116-129)."""
        if penalty_method != "minimum":
            raise ValueError("only penalty_method='minimum' is implemented")
        self.mapping_list = [list(m) for m in mapping_list]
        self.penalty_coefficient = float(penalty_coefficient)
        self.interfaces = []
        for i, (a, b) in enumerate(self.mapping_list):
            ca, cb = np.asarray(mortar_parametric_coords[i][0], float), np.asarray(mortar_parametric_coords[i][1], float)
            if ca.shape == (2, 2) and self.mortar_nels is not None and self.mortar_nels[i] + 1 != 2:
                self.interfaces.append(Interface.from_endpoints(a, b, ca, cb, self.mortar_nels[i]))
            else:
                self.interfaces.append(Interface(a, b, ca, cb))
        self.num_intersections = len(self.interfaces)
        self._drop_device()

    def set_residuals(self, residuals, residuals_deriv=None):
        """nonmatching_opt.py:433-452 (the derivative forms are built inside the kernels)."""
        assert len(residuals) == self.num_splines
        self.residuals = list(residuals)
        self._drop_device()

    def set_point_sources(self, point_sources=[], point_source_inds=[]):
        self.point_sources = list(point_sources)
        self.point_source_inds = list(point_source_inds)
        self._drop_device()

    # ------------------------------------------------------------------ design variables
    def set_shopt_surf_inds(self, opt_field, shopt_surf_inds):
        """nonmatching_opt.py:148-196."""
        assert len(opt_field) == len(shopt_surf_inds)
        self.opt_shape = True
        self.opt_field = list(opt_field)
        self.shopt_surf_inds = [list(s) for s in shopt_surf_inds]
        self.shopt_num_desvars = [int(sum(self.vec_scalar_iga_dof_list[s] for s in inds)) for inds in self.shopt_surf_inds]
        self.cpdes_iga_dofs_full_list = []
        self._shopt_cols = []
        for inds in self.shopt_surf_inds:
            off, lst, cols = 0, [], []
            for s in inds:
                lst.append(list(range(off, off + self.vec_scalar_iga_dof_list[s])))
                off += self.vec_scalar_iga_dof_list[s]
                cols.append(np.arange(self.cp_off[s], self.cp_off[s + 1]))
            self.cpdes_iga_dofs_full_list.append(lst)
            self._shopt_cols.append(np.concatenate(cols))
        self.cpdes_iga_dofs_full = [np.concatenate(l) for l in self.cpdes_iga_dofs_full_list]
        self.cpdes_iga_dofs = [[list(sub) for sub in lst] for lst in self.cpdes_iga_dofs_full_list]
        self.cp_shapes = [(s.n_u, s.n_v) for s in self.splines]
        self.shopt_pin_dofs = [[] for _ in self.opt_field]

    def set_geom_preprocessor(self, preprocessor):
        """nonmatching_opt.py:129-141: ``preprocessor`` is a ``cpiga2xi.IntersectionData`` here."""
        self.preprocessor = preprocessor
        self.cp_shapes = [(P.n_u, P.n_v) for P in preprocessor.patches]

    def solve_init_CPIGA(self):
        """nonmatching_opt.py:216-229 L2-projects the FE control functions to IGA dofs; the IGA control points are the
        primary data here."""
        return self.get_init_CPIGA()

    def set_shopt_align_CP(self, align_surf_inds=[], align_dir=[]):
        """nonmatching_opt.py:232-301 (shape optimisation directly on surface control points): the control points of the
        listed patches take one value along parametric direction ``align_dir`` (0: rows of the net move together, the design
        dofs are the first column; 1: columns move together, the design dofs are the first row).  Returns
        ``shopt_dcpaligndcpsurf`` (full control points of the optimised patches x design dofs).  The reference stores the
        block of an aligned patch at its position in ``align_surf_inds``; here at its position among the optimised patches."""
        assert len(align_surf_inds) == len(self.opt_field) and len(align_dir) == len(self.opt_field)
        self.align_surf_inds, self.align_dir = align_surf_inds, align_dir
        sizes = self.vec_scalar_iga_dof_list
        self.align_cp_deriv_list = [[sp.identity(sizes[s], format="csr") for s in inds] for inds in self.shopt_surf_inds]
        self.cpdes_iga_dofs = [[list(sub) for sub in lst] for lst in self.cpdes_iga_dofs_full_list]
        self.shopt_num_desvars = [int(sum(sizes[s] for s in inds)) for inds in self.shopt_surf_inds]
        for fi, field in enumerate(self.opt_field):
            if self.align_surf_inds[fi] is None:
                continue
            for k, s in enumerate(self.align_surf_inds[fi]):
                if s not in self.shopt_surf_inds[fi]:
                    raise ValueError(f"Aligned surface {s} is not optimized.")
                ncol, nrow = self.cp_shapes[s]
                pos = self.shopt_surf_inds[fi].index(s)
                full = np.asarray(self.cpdes_iga_dofs_full_list[fi][pos])
                a = np.arange(sizes[s])
                if self.align_dir[fi][k] == 0:
                    D = sp.coo_matrix((np.ones(sizes[s]), (a, a // ncol)), shape=(sizes[s], nrow))
                    self.cpdes_iga_dofs[fi][pos] = list(full[0:sizes[s]:ncol])
                elif self.align_dir[fi][k] == 1:
                    D = sp.coo_matrix((np.ones(sizes[s]), (a, a % ncol)), shape=(sizes[s], ncol))
                    self.cpdes_iga_dofs[fi][pos] = list(full[0:ncol])
                else:
                    raise ValueError("Undefined direction: {}".format(self.align_dir[fi][k]))
                self.shopt_num_desvars[fi] += D.shape[1] - sizes[s]
                self.align_cp_deriv_list[fi][pos] = D.tocsr()
        self.cpdes_iga_dofs = [np.concatenate(l).astype(int) for l in self.cpdes_iga_dofs]
        init = self.get_init_CPIGA()
        self.shopt_dcpaligndcpsurf = [sp.block_diag(blocks, format="coo") for blocks in self.align_cp_deriv_list]
        self.init_cp_iga_design = [init[fi][self.cpdes_iga_dofs[fi]] for fi in range(len(self.opt_field))]
        return self.shopt_dcpaligndcpsurf

    def set_shopt_pin_CP(self, pin_surf_inds=[], pin_dir=[], pin_side=[], pin_dofs=None, pin_vals=None):
        """nonmatching_opt.py:303-364: design control points on the edge ``pin_side`` of direction ``pin_dir`` of the listed
        patches keep their initial values (linear equality constraint ``shopt_dcppindcpsurf``)."""
        # design dofs per field: flat arrays after set_shopt_align_CP, per-patch lists before
        des = [np.asarray(d if isinstance(d, np.ndarray) else np.concatenate(d)).astype(int).ravel() for d in self.cpdes_iga_dofs]
        if pin_dofs is None:
            assert len(pin_surf_inds) == len(self.opt_field) and len(pin_dir) == len(self.opt_field) and len(pin_side) == len(self.opt_field)
            self.pin_surf_inds, self.pin_dir, self.pin_side = pin_surf_inds, pin_dir, pin_side
            sizes = self.vec_scalar_iga_dof_list
            for fi, field in enumerate(self.opt_field):
                for k, s in enumerate(self.pin_surf_inds[fi]):
                    if s not in self.shopt_surf_inds[fi]:
                        raise ValueError(f"Pinned surface {s} is not optimized.")
                    pos = self.shopt_surf_inds[fi].index(s)
                    ncol, nrow = self.cp_shapes[s]
                    full = self.cpdes_iga_dofs_full_list[fi][pos]
                    d, side = self.pin_dir[fi][k], self.pin_side[fi][k]
                    if d == 0:
                        local = range(0, sizes[s], ncol) if side == 0 else range(ncol - 1, sizes[s], ncol)
                    elif d == 1:
                        local = range(0, ncol) if side == 0 else range(sizes[s] - ncol, sizes[s])
                    else:
                        raise ValueError("Undefined direction: {}".format(d))
                    keep = set(int(x) for x in des[fi])
                    self.shopt_pin_dofs[fi] += [full[i] for i in local if full[i] in keep]
        else:
            self.shopt_pin_dofs = pin_dofs
        init = self.get_init_CPIGA()
        self.shopt_pin_vals = [init[fi][self.shopt_pin_dofs[fi]] for fi in range(len(self.opt_field))] if pin_vals is None else pin_vals
        self.shopt_dcppindcpsurf = []
        for fi in range(len(self.opt_field)):
            pos = {int(d): c for c, d in enumerate(des[fi])}
            cols = [pos[int(d)] for d in self.shopt_pin_dofs[fi]]
            self.shopt_dcppindcpsurf.append(sp.coo_matrix((np.ones(len(cols)), (np.arange(len(cols)), cols)), shape=(len(cols), des[fi].size)))
        return self.shopt_dcppindcpsurf

    # FE <-> IGA: the reference keeps every field twice (FE dofs of the extraction mesh and IGA dofs); here the IGA
    # control variables are the only representation, so the FE-side entry points are the IGA ones
    def vec_IGA2FE(self, v_iga, v_fe=None, s_ind=None):
        """nonmatching_opt.py:433-441."""
        return np.asarray(v_iga, float)

    def vec_scalar_IGA2FE(self, v_iga, v_fe=None, s_ind=None):
        """nonmatching_opt.py:443-451."""
        return np.asarray(v_iga, float)

    def update_CPFE(self, cp_array_fe, field):
        """nonmatching_opt.py:486-493."""
        self.update_CPIGA(cp_array_fe, field)

    def update_h_th_FE(self, h_th_array):
        """nonmatching_opt.py:508-514."""
        self.update_h_th_IGA(h_th_array)

    def set_init_CPIGA(self, cp_iga):
        self.init_cp_iga = cp_iga

    def get_init_CPIGA(self):
        """Initial homogeneous control points of the optimised patches, per opt field
        (replaces solve_init_CPIGA's L2 projection, nonmatching_opt.py:216-229: the IGA
        control points are the primary data here)."""
        if self.init_cp_iga is None:
            self.init_cp_iga = [self.cp_iga[f][self._shopt_cols[i]].copy() for i, f in enumerate(self.opt_field)]
        return self.init_cp_iga

    def set_thickness_opt(self, var_thickness=False):
        """nonmatching_opt.py:372-398.  var_thickness: one value per control point (B-spline
        thickness field); otherwise one value per patch."""
        self.opt_thickness = True
        self.var_thickness = bool(var_thickness)
        if self.var_thickness:
            self.h_th_sizes = list(self.vec_scalar_iga_dof_list)
            self.init_h_th_iga = np.concatenate(self.h_th)
        else:
            self.h_th_sizes = [1] * self.num_splines
            self.h_th_dof = self.num_splines
            self.init_h_th_list = [np.array([float(np.mean(h))]) for h in self.h_th]
            self.init_h_th = np.concatenate(self.init_h_th_list)

    # ------------------------------------------------------------------ attributes the reference's components read
    # (SURVEY.md 8(b)).  "FE" and "IGA" dofs coincide here (assembly is done directly in IGA dofs), so the FE-named
    # attributes are views of the same data.
    @property
    def vec_scalar_fe_dof(self):
        return self.vec_scalar_iga_dof

    @property
    def init_h_th_fe(self):
        return np.concatenate(self.h_th)

    @property
    def h_th_fe_list(self):
        return [h.copy() for h in self.h_th]

    @property
    def cpdes_iga_nest(self):
        """Current homogeneous control-point coordinates of the optimised patches, one array per opt field
        (nested PETSc vectors in the reference, nonmatching_opt.py:158-185)."""
        return [self.cp_iga[f][self._shopt_cols[i]].copy() for i, f in enumerate(self.opt_field)]

    cpdes_fe_nest = cpdes_iga_nest

    @property
    def shopt_cpsurf_fe_hom_list(self):
        """(n, 4) homogeneous control points of the optimised patches (nonmatching_opt_ffd.py:60-141)."""
        cols = self._shopt_cols[0]
        if any(not np.array_equal(c, cols) for c in self._shopt_cols[1:]):
            cols = np.unique(np.concatenate(self._shopt_cols))          # fields with different patch sets: the block is sized over their union
        w = np.concatenate([s.cp_hom_flat()[:, 3] for s in self.splines])[cols]
        return np.stack([self.cp_iga[f][cols] for f in range(3)] + [w], 1)

    @property
    def cpsurf_des_lims(self):
        """Bounding box of the optimised patches' physical control points (used to size the FFD block)."""
        H = self.shopt_cpsurf_fe_hom_list
        return [[float((H[:, f] / H[:, 3]).min()), float((H[:, f] / H[:, 3]).max())] for f in range(3)]

    # ------------------------------------------------------------------ device model
    def _arrays(self):
        alphas = [penalty_parameters(self.splines, self.h_th, self.E, self.nu, itf, self.penalty_coefficient)
                  for itf in self.interfaces]
        pls = []
        if self.point_sources:
            pls = point_load_entries(self.splines, self.cp_off,
                                     [(s, ps.xi, ps.field, ps.value) for ps, s in zip(self.point_sources, self.point_source_inds)])
        bf = [list(r.body_force) for r in self.residuals]
        lp = [list(getattr(r, "projected", (0.0, 0.0, 0.0))) for r in self.residuals]
        pr = [float(getattr(r, "pressure", 0.0)) for r in self.residuals]
        et = [(s, d, side, tuple(f)) for s, r in enumerate(self.residuals) for (d, side, f) in getattr(r, "edge_tractions", ())]
        return ModelArrays(self.splines, self.E, self.nu, bf, self.interfaces, alphas, pls, load_proj=lp, pressure=pr, edge_traction=et)

    def _spec(self):
        """The model as a geometry.ProblemSpec (what the sharding code partitions)."""
        from .geometry import ProblemSpec
        pls = [(s, ps.xi, ps.field, ps.value) for ps, s in zip(self.point_sources or [], self.point_source_inds or [])]
        ets = [(s, d, side, tuple(f)) for s, r in enumerate(self.residuals) for (d, side, f) in getattr(r, "edge_tractions", ())]
        return ProblemSpec(self.splines, list(self.interfaces), list(self.E), list(self.nu), None, [list(r.body_force) for r in self.residuals], pls,
                           self.penalty_coefficient, "NonMatchingOpt", [list(getattr(r, "projected", (0.0, 0.0, 0.0))) for r in self.residuals],
                           pressure=[float(getattr(r, "pressure", 0.0)) for r in self.residuals], edge_traction=ets or None)

    @property
    def sharded(self):
        return self._dist is not None

    @property
    def dev(self):
        if self._dev is None:
            self._arrays_cache = self._arrays()              # the global model's tables (Dirichlet dofs, symmetry of K, interface offsets), on every rank
            if self.sharded:
                from .sharding import ShardedDeviceModel
                self._dev = ShardedDeviceModel(self._spec(), self._dist, self.rank, self.world, device=self.device, thickness_global=self.h_th, group=self._group)
            else:
                self._dev = _lib.DeviceModel(self._arrays_cache, device=self.device)
            for f in range(3):
                self._dev.set_cp(f, self.cp_iga[f])
            self._dev.set_thickness(np.concatenate(self.h_th))
            self._dev.set_u(self.u_iga)
            self.zero_dofs = self._arrays_cache.zero_dofs
            self._touch()
        return self._dev

    # ------------------------------------------------------------------ state updates
    def update_uIGA(self, u_array_iga):
        """nonmatching_opt.py:474-484."""
        u = np.asarray(u_array_iga, float).ravel()
        if u.size != self.vec_iga_dof:
            raise ValueError("update_uIGA: expected %d values, got %d" % (self.vec_iga_dof, u.size))
        if self._dev is not None and np.array_equal(u, self.u_iga):
            return                                  # unchanged input: what was assembled / evaluated for this state stays current
        self.u_iga = u.copy()
        self.dev.set_u(self.u_iga)
        self._touch()

    def update_CPIGA(self, cp_array_iga, field):
        """nonmatching_opt.py:495-506: homogeneous coordinate ``field`` of the patches in
        ``shopt_surf_inds[opt_field.index(field)]``."""
        field_ind = self.opt_field.index(field)
        cols = self._shopt_cols[field_ind]
        v = np.asarray(cp_array_iga, float).ravel()
        if v.size != cols.size:
            raise ValueError("update_CPIGA: expected %d values, got %d" % (cols.size, v.size))
        if self._dev is not None and np.array_equal(self.cp_iga[field][cols], v):
            return
        self.cp_iga[field][cols] = v
        self.dev.set_cp(field, self.cp_iga[field])
        self._touch()

    def update_h_th_IGA(self, h_th_iga_array):
        """nonmatching_opt.py:516-525 (variable thickness, one value per control point)."""
        v = np.asarray(h_th_iga_array, float).ravel()
        if v.size != self.vec_scalar_iga_dof:
            raise ValueError("update_h_th_IGA: expected %d values, got %d" % (self.vec_scalar_iga_dof, v.size))
        if self._dev is not None and np.array_equal(np.concatenate(self.h_th), v):
            return
        self.h_th = [v[self.cp_off[s]:self.cp_off[s + 1]].copy() for s in range(self.num_splines)]
        self.dev.set_thickness(v)
        self._touch()

    def update_h_th(self, h_th_array):
        """nonmatching_opt.py:527-531 (constant thickness per patch)."""
        v = np.asarray(h_th_array, float).ravel()
        if v.size != self.num_splines:
            raise ValueError("update_h_th: expected %d values, got %d" % (self.num_splines, v.size))
        new = np.concatenate([np.full(s.ncp, v[i]) for i, s in enumerate(self.splines)])
        if self._dev is not None and np.array_equal(np.concatenate(self.h_th), new):
            return
        self.h_th = [np.full(s.ncp, v[i]) for i, s in enumerate(self.splines)]
        self.dev.set_thickness(new)
        self._touch()

    # ------------------------------------------------------------------ residual and Jacobians
    def _touch(self):
        """A state input (u, CP, thickness, intersections) changed: what was assembled / evaluated is stale."""
        self._state_version = getattr(self, "_state_version", 0) + 1

    def _assemble(self, flags):
        """Device assembly of the outputs in ``flags`` that are not current for the present state: the reference re-assembles
        on every call (RIGA, dRIGAduIGA, dRIGAdCPIGA, ... each assemble their forms); an OpenMDAO iteration asks for the same
        state several times (apply_nonlinear after solve_nonlinear, linearize after the last Newton step)."""
        dev, sv = self.dev, getattr(self, "_state_version", 0)
        if getattr(self, "_asm_state", None) != (sv, id(dev)):
            self._asm_state, self._asm_flags = (sv, id(dev)), 0
        need = flags & ~self._asm_flags
        if need:
            dev.assemble(need)
            self._asm_flags |= need
            if need & _lib.ASM_K:
                self._k_version = getattr(self, "_k_version", 0) + 1

    def _cached(self, key, fn):
        """One device evaluation of a functional (value + all gradient fields) per state: the operations ask for the value and
        for each gradient separately (IntEnergyExOperation.Wint / dWintduIGA / dWintdCPIGA / dWintdh_th, ...)."""
        tag = (getattr(self, "_state_version", 0), id(self.dev))
        if getattr(self, "_fun_state", None) != tag:
            self._fun_state, self._fun_cache = tag, {}
        if key not in self._fun_cache:
            self._fun_cache[key] = fn()
        return self._fun_cache[key]

    def functionals(self, apply_bcs=True):
        return self._cached(("functionals", bool(apply_bcs)), lambda: self.dev.functionals(apply_bcs=apply_bcs))

    def compliance(self, forces, apply_bcs=True):
        f = np.ascontiguousarray(forces, float)
        return self._cached(("compliance", f.tobytes(), bool(apply_bcs)), lambda: self.dev.compliance(f, apply_bcs=apply_bcs))

    def stress_forms(self, mode, rho, m_list, surf, measure, apply_bcs=True, gradients=True):
        ml = np.ascontiguousarray(m_list, float)
        key = ("stress", int(mode), float(rho), ml.tobytes(), int(surf), int(measure), bool(apply_bcs))
        if not gradients:                                     # a values-only request is served by a cached full evaluation
            full = self._cached_peek(key + (True,))
            if full is not None:
                return full
        return self._cached(key + (bool(gradients),), lambda: self.dev.stress_forms(mode, rho, ml, surf, measure, apply_bcs=apply_bcs, gradients=gradients))

    def _cached_peek(self, key):
        tag = (getattr(self, "_state_version", 0), id(self.dev))
        return self._fun_cache.get(key) if getattr(self, "_fun_state", None) == tag else None

    # ------------------------------------------------------------------ direct solves with K
    linear_solver = os.environ.get("GF_LINEAR_SOLVER", "device")   # "device": block-banded L D L^T on the GPU (default); "host": scipy SuperLU on a copy of K

    linear_solve_rtol = 1e-12     # normwise backward error |b - K x| / (|K|_F |x| + |b|) above which a device solve is rejected (unpivoted L D L^T on an
                                  # indefinite / near-singular tangent; a general-mode refinement that stalled): a stable solve sits at 1e-19 ... 1e-21 here
                                  # (Frobenius norm in the denominator), so 1e-12 is still generous, while 1e-10 admitted relative residuals near O(1) at
                                  # cond(K) ~ 1e10; |b - K x| / |b| itself has a floor of eps cond(K) for any solver
    linear_solve_rtol_small_pivot = 1e-14   # the bar when the factorisation met a pivot below 1e-14 of the largest (gfs_info: small_pivot)

    def _drop_device(self):
        """The device model is stale (coupling, loads or intersections changed): close it together with everything that
        borrows its buffers (the factorisation reads K in place) or was computed for it."""
        ds = getattr(self, "_dsolver", None)
        if ds is not None:
            ds.close()
        if self._dev is not None:
            self._dev.close()
        self._dev = self._dsolver = self._hlu = None
        self._dsolver_permanent_failure = None
        self._touch()

    def _host_solve(self, rhs, ver, transpose=False):
        from scipy.sparse.linalg import splu
        if getattr(self, "_hlu", None) is None or self._hlu_version != ver:
            self._hlu, self._hlu_version = splu(self.dev.csr(_lib.MAT_K).tocsc()), ver
        if np.ndim(rhs) == 2:                           # rows = right-hand sides
            return np.ascontiguousarray(self._hlu.solve(np.ascontiguousarray(np.asarray(rhs, float).T), trans="T" if transpose else "N").T)
        return self._hlu.solve(rhs, trans="T" if transpose else "N")

    @property
    def symmetric_K(self):
        """False when a follower pressure contributes its load stiffness (then K^T differs from K: the device solver factors the
        symmetric part and refines against K or K^T itself, _solver.DeviceSolver(general=True))."""
        self.dev
        return bool(getattr(getattr(self, "_arrays_cache", None), "symmetric_K", True))

    def solve_K(self, rhs, transpose=False, refine=None, stale_factors=False):
        """x = K^{-1} rhs, or K^{-T} rhs with ``transpose`` (the same thing unless a follower pressure makes K non-symmetric), with the
        tangent currently assembled on the device.
        ``linear_solver == "device"`` (default): bandwidth-reducing ordering once on the host, then every call after a new
        assembly is a block-banded L D L^T factorisation + substitutions + iterative refinement on the GPU
        (goldfish_amd/_solver.py, csrc/gf_solver.hip); K's values are read in place from the library's buffer.  The
        factorisation does not pivot across tiles: every solve is checked (normwise backward error after refinement, finite
        values) and one that fails -- or a band that does not fit the device, or a zero pivot -- falls back, with a warning,
        to the host path for this K.  A non-symmetric K (follower pressure) is solved on the device too: factors of its symmetric
        part, refinement against K / K^T; the same backward-error check decides whether that converged.
        ``"host"``: scipy SuperLU on a host copy of K (what MUMPS does in the reference; kept as the cross-check of the tests).
        ``rhs`` may hold several right-hand sides as rows (shape (k, ndof)): they are solved in one call (DeviceSolver.solve_multi).
        ``refine``: refinement sweeps of a device solve (default: the solver's; the Newton loop passes 0 -- its corrections do not need the last decade of the linear
        residual and every sweep reads the factors twice; adjoint solves keep the full refinement).
        Replaces GOLDFISH/utils/opt_utils.py:156-209 (MUMPS on a copy of K per call)."""
        rhs = np.asarray(rhs, float)
        ver = getattr(self, "_k_version", 0)
        if (self.linear_solver == "device" and getattr(self, "_dsolver_failed_version", None) != ver
                and getattr(self, "_dsolver_permanent_failure", None) is None):
            from . import _solver
            import warnings
            why, x, permanent = None, None, False
            try:
                if getattr(self, "_dsolver", None) is None or self._dsolver.D is not self.dev:
                    w = np.concatenate([sp_.cp_hom_flat()[:, 3] for sp_ in self.splines])
                    # a follower pressure's K is not symmetric: the device solver then factors its symmetric part and refines against K / K^T
                    coords = np.stack([self.cp_iga[f] / w for f in range(3)], 1)
                    if self._use_distributed_solver():
                        # sharded problem, stage 2 (goldfish_amd/_dsolver.py): every rank factors the subtrees dealt to it, the top of the elimination tree is replicated
                        from . import _dsolver
                        self._dsolver = _dsolver.DistributedSolver(self.dev, self._dist, self._group, coords=coords)
                    else:
                        self._dsolver = _solver.DeviceSolver(self.dev, coords=coords, general=not self.symmetric_K)
                    self._dsolver_version = ver
                elif self._dsolver_version != ver and not stale_factors:
                    self._dsolver.refactor()
                    self._dsolver_version = ver
                if rhs.ndim == 2:                           # several right-hand sides (adjoints of several functionals): one call, sweeps next to each other
                    x = self._dsolver.solve_multi(rhs, transpose=transpose and not self.symmetric_K)
                else:
                    x = self._dsolver.solve(rhs, transpose=transpose and not self.symmetric_K, max_refine=None if (refine is None or not self.symmetric_K) else refine)
                rr, be = self._dsolver.rel_residual, self._dsolver.backward_error
                self.linear_solve_relative_residual, self.linear_solve_backward_error = rr, be
                small = bool(getattr(self._dsolver, "small_pivot", False))
                tol = min(self.linear_solve_rtol, self.linear_solve_rtol_small_pivot) if small else self.linear_solve_rtol
                # a chord step (factors of an earlier tangent, on purpose) is not an exact solve: its quality is the Newton loop's to judge
                if not (np.all(np.isfinite(x)) and (be <= tol or (stale_factors and self._dsolver_version != ver))):
                    why = "backward error %.3e > %.1e after refinement (relative residual %.3e%s)" % (be, tol, rr, "; the factorisation met a small pivot" if small else "")
            except RuntimeError as e:
                why = str(e)
                ds, self._dsolver = getattr(self, "_dsolver", None), None
                if ds is not None:
                    ds.close()
                # a failure that another assembly cannot cure (the factors do not fit the device, the block pattern is not symmetric) is latched until
                # the device model is dropped: otherwise every Newton iteration rebuilds the solver (pattern download, host nested dissection: seconds
                # at C4) only to fail again
                permanent = any(k in why for k in ("device memory", "out of memory", "does not fit", "not symmetric", "hipMalloc"))
            if getattr(self, "_dist", None) is not None:
                # every rank takes the SAME way out: a failure that only one rank saw (its own GPU ran out of memory under the replicated factorisation)
                # must not send that rank alone into the host solve, whose K gather is a collective the others would never join
                n_fail, n_perm = (int(v) for v in self.dev._allreduce(np.array([float(why is not None), float(permanent)])))
                if n_fail and why is None:
                    why = "the device solve failed on another rank"
                    ds, self._dsolver = getattr(self, "_dsolver", None), None
                    if ds is not None:
                        ds.close()
                permanent = n_perm > 0
            if why is None:
                return x
            if permanent:
                self._dsolver_permanent_failure = why
                self._refuse_host_solve_of_a_huge_model(why)
                warnings.warn("solve_K: the device solver cannot be used for this model (%s); the host sparse LU is used until the model changes" % why, RuntimeWarning)
                return self._host_solve(rhs, ver, transpose and not self.symmetric_K)
            warnings.warn("solve_K: device L D L^T rejected for this tangent (%s); falling back to the host sparse LU" % why, RuntimeWarning)
            self._dsolver_failed_version = ver
        return self._host_solve(rhs, ver, transpose and not self.symmetric_K)

    #: above this many dofs the host sparse LU is not a fallback (C5: 10 M dofs -- SuperLU would need hundreds of GB and hours): solve_K raises instead, naming what does work
    host_solve_max_dofs = int(os.environ.get("GF_HOST_SOLVE_MAX_DOFS", "3000000"))

    def _refuse_host_solve_of_a_huge_model(self, why):
        if int(getattr(self, "vec_iga_dof", 0)) > self.host_solve_max_dofs:
            raise RuntimeError("solve_K: the device factorisation of this model is not possible on one GPU (%s) and a host sparse LU of %d dofs is not a path either "
                               "(host_solve_max_dofs = %d).  Models of this size (C5: 708 GB of factors) need the distributed factorisation: "
                               "NonMatchingOpt(..., comm=torch.distributed) on a multi-GPU node (goldfish_amd/_dsolver.py; C5 fits 8 x 288 GB); preconditioned CG was "
                               "measured and does not converge on these tangents (goldfish_amd/_krylov.py, profiles/r05_krylov_study.txt)." % (why, self.vec_iga_dof, self.host_solve_max_dofs))

    #: direct solves of a sharded problem: "distributed" (stage 2: subtrees of the elimination tree per rank, replicated top; symmetric K, models large enough
    #: for the nested-dissection mode), "replicated" (stage 1: every rank factors the gathered K), "auto": distributed when it applies
    sharded_solver = os.environ.get("GF_SHARDED_SOLVER", "auto")

    #: Newton loop: reuse the factorisation for chord steps while they contract (False: the reference's iteration, a factorisation per step -- the default; True;
    #: "auto": models of at least ``newton_reuse_min_dofs`` dofs on the device solver, where a factorisation costs many times a substitution: C4 252 against 34 ms)
    newton_reuse_factors = False
    newton_reuse_min_dofs = 100000

    def _newton_reuses_factors(self):
        if self.newton_reuse_factors == "auto":
            return self.linear_solver == "device" and self.vec_iga_dof >= self.newton_reuse_min_dofs and getattr(self, "_dsolver_permanent_failure", None) is None
        return bool(self.newton_reuse_factors)

    def _use_distributed_solver(self):
        from . import _solver
        if getattr(self, "_dist", None) is None or self.sharded_solver == "replicated" or not self.symmetric_K or getattr(self, "world", 2) < 2:
            return False
        return self.sharded_solver == "distributed" or self.vec_iga_dof // 3 >= _solver.ND_MIN_CP

    def RIGA(self):
        """Non-matching residual in IGA dofs, Dirichlet rows zeroed (nonmatching_opt.py:941-948)."""
        self._assemble(_lib.ASM_R)
        return self.dev.residual()

    def dRIGAduIGA(self):
        """nonmatching_opt.py:950-959: rows and columns of Dirichlet dofs zeroed, unit diagonal."""
        self._assemble(_lib.ASM_K)
        return self.dev.csr(_lib.MAT_K)

    def dRIGAdCPIGA(self, field):
        """nonmatching_opt.py:992-1004: ndof x (control points of shopt_surf_inds[field_ind]);
        Dirichlet rows zeroed, diagonal 0, no column treatment."""
        field_ind = self.opt_field.index(field)
        self._assemble(_lib.ASM_DRDCP)
        return self.dev.csr(_lib.MAT_DRDCP0 + field)[:, self._shopt_cols[field_ind]].tocsr()

    def dRIGAdh_th(self):
        """nonmatching_opt.py:1006-1015: shell terms only (penalty parameters frozen), no
        Dirichlet treatment.  Constant thickness: columns summed per patch."""
        self._assemble(_lib.ASM_DRDH)
        M = self.dev.csr(_lib.MAT_DRDH)
        if self.var_thickness:
            return M
        return (M @ self._patch_indicator()).tocsr()

    def _patch_indicator(self):
        rows = np.arange(self.vec_scalar_iga_dof)
        cols = np.repeat(np.arange(self.num_splines), self.vec_scalar_iga_dof_list)
        return sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(self.vec_scalar_iga_dof, self.num_splines))

    def dRIGAdCPIGA_FD(self, CP, field, h=1e-8):
        """Finite-difference check of dRIGAdCPIGA (nonmatching_opt.py:975-990)."""
        CP = np.asarray(CP, float)
        self.update_CPIGA(CP, field)
        R0 = self.RIGA()
        J = np.zeros((R0.size, CP.size))
        for k in range(CP.size):
            pert = CP.copy()
            pert[k] += h
            self.update_CPIGA(pert, field)
            J[:, k] = (self.RIGA() - R0) / h
        self.update_CPIGA(CP, field)
        return J

    # ------------------------------------------------------------------ moving intersections
    def create_diff_intersections(self, num_edge_pts=None, preprocessor=None):
        """nonmatching_opt.py:533-556: the intersections listed in ``preprocessor.diff_int_inds`` (default: all) become
        functions of the control points through ``CPIGA2Xi``; ``preprocessor`` is a ``cpiga2xi.IntersectionData``
        (default: built from the patches and the interfaces of ``mortar_meshes_setup``)."""
        from .cpiga2xi import CPIGA2Xi, IntersectionData
        if preprocessor is None:
            preprocessor = IntersectionData(patches=self.splines, mapping_list=self.mapping_list,
                                            intersections_para_coords=[[itf.xi_a, itf.xi_b] for itf in self.interfaces])
        self.preprocessor = preprocessor
        self.cpiga2xi = CPIGA2Xi(preprocessor, self.shopt_surf_inds, self.opt_field, num_edge_pts)
        self.diff_int_inds = list(preprocessor.diff_int_inds)
        self.update_xi(self.cpiga2xi.xi_flat_global)
        self.xi_size = self.cpiga2xi.xi_size_global

    def update_xi(self, xi_flat):
        """nonmatching_opt.py:560-565: store the parametric coordinates; ``update_transfer_matrices`` applies them."""
        xi = np.asarray(xi_flat, float).ravel()
        if xi.size != self.cpiga2xi.xi_size_global:
            raise ValueError("update_xi: expected %d values, got %d" % (self.cpiga2xi.xi_size_global, xi.size))
        self.xi_flat = xi.copy()

    def update_transfer_matrices(self):
        """nonmatching_opt.py:567-600 rebuilds the mortar transfer matrices at the new coordinates; here the mortar-vertex
        tables (support windows, basis values, curve tangents, coupling pattern) live in the device model, which is
        re-created; control points, thickness and displacement are pushed again by ``dev``."""
        c2x = self.cpiga2xi
        patched = self._dev is not None
        for i, g in enumerate(self.diff_int_inds):
            n = c2x.diff_int_num_pts[i]
            sub = self.xi_flat[c2x.xi_flat_inds[i]:c2x.xi_flat_inds[i + 1]]
            a, b = self.mapping_list[g]
            new = Interface(a, b, sub[:2 * n].reshape(-1, 2), sub[2 * n:].reshape(-1, 2))
            unchanged = np.array_equal(new.xi_a, self.interfaces[g].xi_a) and np.array_equal(new.xi_b, self.interfaces[g].xi_b)
            self.interfaces[g] = new
            # while every mortar vertex stays in its knot spans the coupling pattern and all index tables of the device model stay valid: only the vertex
            # tables of the moved interface are re-evaluated (gf_update_interface), the direct solver keeps its symbolic phase; a vertex that crosses a knot
            # line forces a new model (2.7 s at C4 size)
            if patched and not unchanged:
                patched = bool(self._dev.update_interface(g, new))
        if patched:
            self._touch()                                      # same model, new coupling values: everything assembled is stale
        else:
            self._drop_device()                                # factorisations belong to the old coupling pattern

    def dRIGAdxi(self):
        """d RIGA / d xi_flat (nonmatching_opt.py:1042-1088), ndof x xi_size, Dirichlet rows zeroed: the device returns the
        per-vertex blocks (gf_penalty_dxi -> pen_dxi_kernel); here they are scattered to dofs / coordinates, and the
        tangent blocks are chained with d(tau)/d(xi_A) of the vertex stencil (model.Interface: second-order differences)."""
        c2x, dev = self.cpiga2xi, self.dev
        A = self._arrays_cache
        p = self.splines[0].p
        P1, nb = p + 1, (p + 1) ** 2
        rows, cols, vals = [], [], []
        al = np.arange(nb)
        for i, g in enumerate(self.diff_int_inds):
            n = c2x.diff_int_num_pts[i]
            base = c2x.xi_flat_inds[i]
            # the vertices of this moving interface only: (n, 6, 2, nb, 3), (n, 2, 2); sharded: evaluated by the owner of side A, replicated (ShardedDeviceModel.penalty_dxi_if)
            B, W = dev.penalty_dxi_if(g, p) if self.sharded else dev.penalty_dxi(n, p, v_first=int(A.if_off[g]))
            dof = np.zeros((n, 2, nb, 3), dtype=np.int64)
            for sd, s in enumerate(self.mapping_list[g]):
                cp = self.cp_off[s] + (W[:, sd, 0][:, None] + al % P1) + (W[:, sd, 1][:, None] + al // P1) * self.splines[s].n_u
                dof[:, sd] = 3 * cp[:, :, None] + np.arange(3)
            dof = dof.reshape(n, -1)
            k = np.arange(n)
            for d in range(4):                                                        # the vertex itself
                col = base + (d // 2) * 2 * n + 2 * k + d % 2
                rows.append(dof.ravel()); cols.append(np.repeat(col, dof.shape[1])); vals.append(B[:, d].reshape(n, -1).ravel())
            Gm = np.gradient(np.eye(n), 1.0 / (n - 1), axis=0, edge_order=2 if n > 2 else 1)   # tau = Gm @ xi_A
            for d in range(2):                                                        # neighbours through the tangent
                T = B[:, 4 + d].reshape(n, -1)
                kk, kp = np.nonzero(Gm)
                rows.append(dof[kk].ravel()); cols.append(np.repeat(base + 2 * kp + d, dof.shape[1]))
                vals.append((T[kk] * Gm[kk, kp][:, None]).ravel())
        if rows:
            J = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.vec_iga_dof, c2x.xi_size_global)).tocsr()
        else:
            J = sp.csr_matrix((self.vec_iga_dof, c2x.xi_size_global))
        keep = np.ones(self.vec_iga_dof)
        keep[self.zero_dofs] = 0.0
        return sp.diags(keep) @ J

    def dRIGAdxi_rev(self, lam):
        """(dR/dxi)^T lam without forming dR/dxi: the per-vertex blocks are contracted with lam on the device (gf_penalty_dxi_rev: no 0.64 GB host copy at
        C4) and only 6 doubles per mortar vertex come back, chained here with d(tau)/d(xi_A).  Works on a sharded problem (the ranks' owned rows add up).
        Reference: DispMintImOpeartion.apply_linear_rev, operations/disp_mi_imop.py:75-104."""
        c2x, dev = self.cpiga2xi, self.dev
        lam = np.ascontiguousarray(lam, float)
        out = np.zeros(c2x.xi_size_global)
        for i, g in enumerate(self.diff_int_inds):
            n, base = c2x.diff_int_num_pts[i], c2x.xi_flat_inds[i]
            T = dev.penalty_dxi_rev_if(g, lam)                                   # (n, 6)
            k = np.arange(n)
            for d in range(4):
                out[base + (d // 2) * 2 * n + 2 * k + d % 2] += T[:, d]
            Gm = np.gradient(np.eye(n), 1.0 / (n - 1), axis=0, edge_order=2 if n > 2 else 1)   # tau = Gm @ xi_A
            for d in range(2):
                out[base + 2 * k + d] += Gm.T @ T[:, 4 + d]
        return out

    def dRIGAdxi_FD(self, xi_flat, h=1e-8):
        """Forward-difference check of dRIGAdxi (nonmatching_opt.py:1018-1040): one model rebuild per column."""
        xi0 = np.asarray(xi_flat, float).copy()
        self.update_xi(xi0)
        self.update_transfer_matrices()
        R0 = self.RIGA()
        J = np.zeros((R0.size, xi0.size))
        for k in range(xi0.size):
            x = xi0.copy()
            x[k] += h
            self.update_xi(x)
            self.update_transfer_matrices()
            J[:, k] = (self.RIGA() - R0) / h
        self.update_xi(xi0)
        self.update_transfer_matrices()
        return J

    # ------------------------------------------------------------------ solves (host sparse direct: "next" row N1)
    def solve_linear_nonmatching_problem(self, iga_dofs=True):
        """One Newton step from the current state (PENGoLINS solve_linear_nonmatching_problem;
        reference call sites GOLDFISH/tests/test_dRdt.py:121).  The factorisation runs on the
        host (scipy SuperLU, as MUMPS does in the reference) or, with ``linear_solver = "device"``, on the GPU (solve_K)."""
        self._assemble(_lib.ASM_R | _lib.ASM_K)
        du = self.solve_K(-self.dev.residual())
        self.update_uIGA(self.u_iga + du)
        return self.u_iga.copy()

    newton_step_rtol = 1e-9          # |du| / |u| below which a residual stuck at its round-off floor counts as a converged state
    newton_raise_unconverged = False  # True: an unconverged Newton solve raises instead of warning

    newton_load_steps = 1            # > 1: the dead loads are applied in that many equal increments (each a Newton solve started from the previous one's state)

    def solve_nonlinear_nonmatching_problem(self, solver="direct", ref_error=None, rtol=1e-3, max_it=30,
                                            zero_mortar_funcs=True, iga_dofs=True, POINT_SOURCE=True, load_steps=None):
        """Newton iteration on R(u) = 0 (PENGoLINS; used by DispImOpeartion.solve_nonlinear,
        GOLDFISH/operations/disp_imop.py:38-44: max_it=30, rtol=1e-3 relative to the first residual, start from zero when
        zero_mortar_funcs).  Plain Newton as in the reference, with two safeguards it does not have:

        * backtracking: after the first step (the linear solution, whose residual legitimately exceeds |R_0| for a
          geometrically nonlinear shell) a step whose residual exceeds the largest of the last three (non-monotone rule) is halved, at most four times;
        * honesty about the end of the iteration: ``newton_converged`` is True when |R| / ref < rtol and ONLY then (the reference's
          criterion, disp_imop.py:38-44).  The iteration also ends when the Newton correction has become negligible
          (|du| <= newton_step_rtol |u|: ``newton_converged_by_step``) or when the residual stagnates (``newton_stagnated``) -- the residual
          of a thin, stiffly coupled shell has an evaluation floor that a tight rtol cannot pass --, but neither is reported as convergence:
          every solve that ends with |R| / ref >= rtol warns (RuntimeWarning) or raises (``newton_raise_unconverged``);
          ``newton_history`` keeps (|R| / ref, |du| / |u|, step length) per iteration.

        ``load_steps`` (default ``newton_load_steps`` = 1: the reference's single solve): the dead loads are applied in that many equal increments,
        R_s(u) = R(u) - (1 - s) R(0) for s = 1/n ... 1 (R(0) = -F_ext: the internal and penalty forces vanish at u = 0 and a dead load does not depend
        on u, so no kernel knows about s), each increment a Newton solve from the previous state with the tolerance measured against the FULL load's
        first residual.  A geometrically nonlinear shell whose full load is beyond Newton's reach from u = 0 converges this way.  Not for follower
        pressures (their load depends on u); starts from u = 0."""
        n_steps = int(self.newton_load_steps if load_steps is None else load_steps)
        self._newton_load_offset = None
        if n_steps > 1:
            if any(getattr(r, "pressure", 0.0) != 0.0 for r in getattr(self, "residuals", []) or []):
                raise NotImplementedError("solve_nonlinear_nonmatching_problem: load_steps > 1 with a follower pressure (its load depends on u)")
            self.update_uIGA(np.zeros(self.vec_iga_dof))
            self._assemble(_lib.ASM_R)
            R_full0 = self.dev.residual().copy()                       # R(0) = -F_ext
            ref_full = ref_error if ref_error is not None else (float(np.linalg.norm(R_full0)) or 1.0)
            history, iters = [], 0
            try:
                for k in range(1, n_steps + 1):
                    self._newton_load_offset = (1.0 - k / n_steps) * R_full0 if k < n_steps else None
                    # intermediate increments only need to stay on the path: the reference's default tolerance, never tighter than the caller's
                    self._newton_solve(ref_full, max(rtol, 1e-3) if k < n_steps else rtol, max_it)
                    history += self.newton_history
                    iters += self.newton_iterations
                    if not self.newton_converged and k < n_steps and not np.all(np.isfinite(self.u_iga)):
                        break
            finally:
                self._newton_load_offset = None
            self.newton_history, self.newton_iterations, self.newton_load_steps_done = history, iters, k
            return None, self.u_iga.copy()
        if zero_mortar_funcs:
            self.update_uIGA(np.zeros(self.vec_iga_dof))
        self._newton_solve(ref_error, rtol, max_it)
        return None, self.u_iga.copy()

    def _newton_residual(self):
        """R of the current state as the Newton loop sees it: the assembled residual, minus the part of the dead loads a load increment holds back."""
        R = self.dev.residual()
        off = getattr(self, "_newton_load_offset", None)
        return R if off is None else R - off

    def _newton_solve(self, ref_error, rtol, max_it):
        """The Newton iteration of solve_nonlinear_nonmatching_problem from the current state (one load increment)."""
        self._assemble(_lib.ASM_R | _lib.ASM_K)
        R = self._newton_residual()
        nrm = float(np.linalg.norm(R))
        if ref_error is None:
            ref_error = nrm if nrm > 0 else 1.0
        hist, self.newton_history = [nrm], []
        converged, by_step, stagnated, it = nrm / ref_error < rtol, False, False, 0
        # CHORD steps (large models, device solver): once a Newton step has contracted the residual well, the next correction reuses the factors at hand
        # (substitutions only: C4 34 ms against 252 ms of factorisation + 6 ms of tangent assembly) for as long as every such step keeps contracting; a chord step
        # that the acceptance rule below rejects is thrown away and repeated as a Newton step with fresh factors.  The converged state is the same; the
        # reference's iteration (a factorisation per step) is ``newton_reuse_factors = False``.
        reuse = self._newton_reuses_factors()
        chord, self.newton_chord_steps = False, 0
        while not converged and it < max_it:
            u0 = self.u_iga.copy()
            if not chord:
                self._assemble(_lib.ASM_K)            # tangent of the current state (already there unless the last steps were chord steps)
            du = self.solve_K(-R, refine=0, stale_factors=True) if chord else self.solve_K(-R, refine=0)
            ndu = float(np.linalg.norm(du))
            lam = 1.0
            while True:
                self.update_uIGA(u0 + lam * du)
                # the full step is usually accepted: R and K in one pass; a shortened trial needs |R| only (R-only pass: a third of the R + K pass at C4),
                # the tangent of the state that is finally accepted follows below (_assemble launches only what is not current)
                self._assemble(_lib.ASM_R | _lib.ASM_K if (lam == 1.0 and not reuse) else _lib.ASM_R)
                Rn = self._newton_residual()
                nn = float(np.linalg.norm(Rn))
                if chord and not (np.isfinite(nn) and nn <= 0.5 * hist[-1]):
                    break                             # a chord step must contract by itself: no backtracking on stale factors
                # jitter at the evaluation floor is not a failed step: the iteration has contracted well below the FIRST residual (an overshooting first
                # step alone -- hist = 1, 50, ... -- is no contraction: ADVICE r03), this residual is within 10x of the best one, and the step itself is
                # negligible against the state
                contracted = min(hist) < 0.1 * hist[0]
                near_floor = contracted and nn < 10.0 * min(hist) and lam * ndu <= 1e-4 * max(float(np.linalg.norm(self.u_iga)), 1e-300)
                # NON-MONOTONE acceptance (Grippo-Lampariello-Lucidi): the full step stands unless its residual exceeds the largest of the last three.  The
                # residual norm of a thin shell is a poor merit function -- the sliding-web T-beam goes 1, 208, 0.056, ~0.5, 1e-4, ... under plain Newton (the
                # reference's iteration), and a monotone rule cuts its third step sixteen-fold and creeps (round 4; round 3 let that step through only by the
                # overshoot bug the advisor flagged) --, while a diverging iteration (arctan from 3) still exceeds its recent history and is shortened.
                # The residual of an OVERSHOOTING first step (hist[1] > hist[0]: unconditionally accepted, the T-beam's is 208 |R_0|) counts only as the PREVIOUS
                # residual (the step right after it has to come down from there), not as a member of the window of the step after that -- which otherwise
                # accepts anything up to the overshoot at full length
                window = [h for k, h in enumerate(hist) if k >= len(hist) - 3 and not (k == 1 and len(hist) > 2 and hist[1] > hist[0])]
                if (np.isfinite(nn) and (it == 0 or nn <= max(window) or near_floor)) or lam <= 1.0 / 16.0:
                    break
                lam *= 0.5
            if chord and not (np.isfinite(nn) and nn <= 0.5 * hist[-1]):
                self.update_uIGA(u0)                  # rejected chord step: back to the state before it, Newton step with fresh factors next (R is still that state's)
                chord = False
                continue
            was_chord = chord
            if chord:
                self.newton_chord_steps += 1
            rel_step = ndu / max(float(np.linalg.norm(self.u_iga)), 1e-300)    # the full correction: a shortened step says nothing
            # the next step may be a chord step when this one contracted the residual five-fold AND moved the state by less than 5 % (Newton's fast phase: the
            # tangent hardly changes any more; the overshooting first steps of a shell also "contract" by orders of magnitude, but they move the state)
            chord = reuse and np.isfinite(nn) and lam == 1.0 and nn <= 0.2 * hist[-1] and rel_step <= 0.05
            R, nrm = Rn, nn
            hist.append(nrm)
            it += 1
            self.newton_history.append((nrm / ref_error, rel_step, lam))
            if not np.isfinite(nrm):
                break
            if nrm / ref_error < rtol:
                converged = True
            elif rel_step <= self.newton_step_rtol and nrm < hist[0] and not was_chord:
                by_step = True                        # the state no longer moves, the residual is at its evaluation floor ABOVE rtol: the end of the iteration, not
                break                                 # convergence (reported as such below; a chord step converges linearly: its size is no measure of the remaining error)
            elif (len(hist) >= 5 and min(hist[:-3]) < 0.1 * hist[0] and max(hist[-3:]) < hist[0]
                  and min(hist[-3:]) > 0.5 * min(hist[:-3])):
                stagnated = True                      # after a real contraction (below a tenth of the first residual), three iterations that did not halve the best residual: the evaluation floor
                break
        self.newton_relative_residual = nrm / ref_error
        self.newton_converged, self.newton_converged_by_step, self.newton_stagnated, self.newton_iterations = converged, by_step, stagnated, it
        if not converged:
            msg = ("solve_nonlinear_nonmatching_problem: not converged after %d iterations%s: relative residual %.3e >= rtol %.1e, last "
                   "relative Newton correction %.3e" % (it, " (residual stagnates: round-off floor of its evaluation)" if stagnated else
                                                       (" (the Newton correction is negligible, the residual sits at its evaluation floor)" if by_step else ""),
                                                       nrm / ref_error, rtol, self.newton_history[-1][1] if self.newton_history else float("nan")))
            if self.newton_raise_unconverged:
                raise RuntimeError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning)

    # ------------------------------------------------------------------ convenience
    @classmethod
    def from_spec(cls, spec, thickness=None, device=0, klass=None, comm=None):
        """Build the problem from a goldfish_amd.geometry.ProblemSpec."""
        klass = klass or cls
        h = thickness if thickness is not None else spec.h_th
        pb = klass(spec.patches, spec.E, h, spec.nu, comm=comm, device=device)
        if spec.interfaces:
            pb.create_mortar_meshes([i.npts - 1 for i in spec.interfaces])
            pb.mortar_meshes_setup([[i.a, i.b] for i in spec.interfaces], [[i.xi_a, i.xi_b] for i in spec.interfaces],
                                   spec.penalty_coefficient)
        lp = spec.load_proj if getattr(spec, "load_proj", None) is not None else [(0.0, 0.0, 0.0)] * len(spec.patches)
        pr = spec.pressure if getattr(spec, "pressure", None) is not None else [0.0] * len(spec.patches)
        ets = [[] for _ in spec.patches]
        for (s, d, side, f) in (getattr(spec, "edge_traction", None) or ()):
            ets[s].append((d, side, tuple(f)))
        pb.set_residuals([SVKResidual(tuple(f), tuple(d), float(p), tuple(e)) for f, d, p, e in zip(spec.body_force, lp, pr, ets)])
        if spec.point_loads:
            pb.set_point_sources([PointSource(xi, f, v) for (_, xi, f, v) in spec.point_loads],
                                 [s for (s, _, _, _) in spec.point_loads])
        return pb


class NonMatchingOptFFD(NonMatchingOpt):
    """Problem class of GOLDFISH/nonmatching_opt_ffd.py:9 (same constructor signature, :14-17).
    The FFD-block parametrisation and its linear constraint maps are host-side constant
    sparse operators marked "next" (SURVEY.md 8(f) N2); the shape setters that the hot path
    needs are provided so that fixtures written against NonMatchingOptFFD run unchanged."""

    def set_shopt_surf_inds_FFD(self, opt_field, shopt_surf_inds):
        """nonmatching_opt_ffd.py:60-72: ``shopt_surf_inds`` is ONE list of patch indices shared by every opt field (a list
        per field, as ``set_shopt_surf_inds`` takes it, is accepted too)."""
        inds = list(shopt_surf_inds)
        if len(inds) == 0 or not isinstance(inds[0], (list, tuple, np.ndarray)):
            inds = [list(inds)] * len(opt_field)
        self.set_shopt_surf_inds(opt_field, inds)
        self.shopt_multiffd = False

    def set_shopt_FFD(self, shopt_knotsffd, shopt_cpffd):
        """nonmatching_opt_ffd.py:143-182.  ``shopt_cpffd`` is (l, m, n, >=3) in the igakit order
        convention; the FFD block has the identity geometric mapping.  Returns the constant sparse map
        ``shopt_dcpsurf_fedcpffd`` (FFD control points -> homogeneous surface control points of the
        optimised patches; row a is scaled by the weight w_a because cpFuncs are homogeneous)."""
        from .utils.ffd_utils import CP_FFD_matrix
        self.shopt_knotsffd = [np.asarray(k, float) for k in shopt_knotsffd]
        self.shopt_cpffd = np.asarray(shopt_cpffd, float)
        self.shopt_cpffd_flat = self.shopt_cpffd[..., 0:3].transpose(2, 1, 0, 3).reshape(-1, 3)
        self.shopt_ffd_degree = [int(np.sum(k == k[0]) - 1) for k in self.shopt_knotsffd]
        self.shopt_cpffd_shape = self.shopt_cpffd.shape[0:3]
        self.shopt_cpffd_size = int(np.prod(self.shopt_cpffd_shape))
        self.shopt_num_desvars = [self.shopt_cpffd_size for _ in self.opt_field]
        self.shopt_cpffd_design_dof = [list(range(self.shopt_cpffd_size)) for _ in self.opt_field]
        self.shopt_cpffd_design_dof_full = [list(d) for d in self.shopt_cpffd_design_dof]
        # nonmatching_opt_ffd.py:60-72 shares ONE patch list between the fields, so the reference has one map; a list per field (accepted by
        # set_shopt_surf_inds_FFD here) gives one map per field: the FFD block is the same, the rows are the control points of that field's patches
        w_all = np.concatenate([s.cp_hom_flat()[:, 3] for s in self.splines])
        maps = []
        for cols in self._shopt_cols:
            same = next((m for c, m in zip(self._shopt_cols, maps) if np.array_equal(c, cols)), None)
            if same is not None:
                maps.append(same)
                continue
            w = w_all[cols]
            X = np.stack([self.cp_iga[f][cols] / w for f in range(3)], 1)           # physical control points
            maps.append(sp.diags(w).dot(CP_FFD_matrix(X, self.shopt_ffd_degree, self.shopt_knotsffd).tocsr()).tocoo())
        self.shopt_dcpsurf_fedcpffd_list = maps                                      # one entry per opt field (the same object when the patch sets agree)
        self.shopt_dcpsurf_fedcpffd = maps[0]
        self.shopt_ffd_shared_patches = all(m is maps[0] for m in maps)
        self.shopt_init_cpffd_full = [self.shopt_cpffd_flat[:, f].copy() for f in self.opt_field]
        self.shopt_cpffd_pin_dof = [[] for _ in self.opt_field]
        self.shopt_align_dir = [None for _ in self.opt_field]
        return self.shopt_dcpsurf_fedcpffd if self.shopt_ffd_shared_patches else self.shopt_dcpsurf_fedcpffd_list

    # ------------------------------------------------------------------ FFD lattice helpers
    @staticmethod
    def _lattice(shape):
        """(i, j, k) of every FFD control point in dof order i + j*l + k*l*m (ijk2dof of the reference)."""
        l, m, nn = (int(s) for s in shape)
        k, j, i = np.meshgrid(np.arange(nn), np.arange(m), np.arange(l), indexing="ij")
        return np.stack([i.ravel(), j.ravel(), k.ravel()], 1)

    # ------------------------------------------------------------------ shape FFD: linear constraint maps
    def dCPaligndCPFFD(self, field, align_dir, cpffd_shape, side=0):
        """nonmatching_opt_ffd.py:1034-1083.  Control points of the block take the same value along the directions in
        ``align_dir``: the design dofs are the points of the face/edge ``side`` and the returned map replicates them
        (full FFD dofs x design dofs, 0/1 entries)."""
        align_dir = sorted(int(d) for d in align_dir)
        if field in align_dir:
            raise ValueError("Illegal CPFFD align direction %s for opt field %d" % (align_dir, field))
        if not align_dir or any(d not in (0, 1, 2) for d in align_dir) or len(set(align_dir)) != len(align_dir) or len(align_dir) > 2:
            raise ValueError("Undefined CPFFD ailgn direction %s" % (align_dir,))
        ijk = self._lattice(cpffd_shape)
        l, m, _ = (int(s) for s in cpffd_shape)
        rep = ijk.copy()
        for d in align_dir:
            rep[:, d] = side * (int(cpffd_shape[d]) - 1)
        rep_dof = rep[:, 0] + rep[:, 1] * l + rep[:, 2] * l * m
        free_dof = np.unique(rep_dof)                                    # ascending = the reference's loop order
        cols = np.searchsorted(free_dof, rep_dof)
        deriv = sp.coo_matrix((np.ones(rep_dof.size), (np.arange(rep_dof.size), cols)), shape=(rep_dof.size, free_dof.size))
        return [int(d) for d in free_dof], deriv

    def set_shopt_align_CPFFD(self, align_dir=None):
        """nonmatching_opt_ffd.py:691-724 (linear equality constraint built into the parametrisation)."""
        self.shopt_align_dir = [None for _ in self.opt_field] if align_dir is None else list(align_dir)
        assert len(self.shopt_align_dir) == len(self.opt_field)
        size = self.shopt_cpffd_size
        self.shopt_dcpaligndcpffd = []
        for field_ind, field in enumerate(self.opt_field):
            sub = self.shopt_align_dir[field_ind]
            if sub is not None:
                free_dof, deriv = self.dCPaligndCPFFD(field, sub, self.shopt_cpffd_shape)
            else:
                free_dof, deriv = list(range(size)), sp.identity(size, format="coo")
            self.shopt_cpffd_design_dof[field_ind] = free_dof
            self.shopt_dcpaligndcpffd.append(deriv)
        self.shopt_init_cpffd_design = [self.shopt_init_cpffd_full[i][self.shopt_cpffd_design_dof[i]] for i in range(len(self.opt_field))]
        return self.shopt_dcpaligndcpffd

    def CPpinDoFs(self, pin_dir0, pin_side0, pin_dir1, pin_side1, cpffd_shape):
        """nonmatching_opt_ffd.py:1120-1196: dofs of the faces ``pin_dir0``/``pin_side0`` (a surface), restricted to the
        edges ``pin_dir1``/``pin_side1`` when given (a line)."""
        if pin_dir0 not in (0, 1, 2):
            raise ValueError("Unsupported pin_dir0 {}".format(pin_dir0))
        if pin_dir1 is not None and (pin_dir1 not in (0, 1, 2) or pin_dir1 == pin_dir0):
            raise ValueError("Unsupported pin_dir1 {}".format(pin_dir1))
        ijk = self._lattice(cpffd_shape)
        l, m, _ = (int(s) for s in cpffd_shape)
        dof = ijk[:, 0] + ijk[:, 1] * l + ijk[:, 2] * l * m
        out = []
        for side0 in pin_side0:
            on0 = ijk[:, pin_dir0] == int(side0 * (int(cpffd_shape[pin_dir0]) - 1))
            if pin_dir1 is None:
                out.append(dof[on0])
            else:
                for side1 in pin_side1:
                    out.append(dof[on0 & (ijk[:, pin_dir1] == int(side1 * (int(cpffd_shape[pin_dir1]) - 1)))])
        return np.concatenate(out) if out else np.zeros(0, int)

    def dCPpindCPFFD(self, cpffd_des_dof, cpffd_pin_dof):
        """nonmatching_opt_ffd.py:1198-1204: selection of the pinned design dofs."""
        pos = {d: c for c, d in enumerate(cpffd_des_dof)}
        cols = [pos[d] for d in cpffd_pin_dof]
        return sp.coo_matrix((np.ones(len(cols)), (np.arange(len(cols)), cols)), shape=(len(cols), len(cpffd_des_dof)))

    def set_shopt_pin_CPFFD(self, pin_dir0, pin_side0, pin_dir1=None, pin_side1=None):
        """nonmatching_opt_ffd.py:758-815 (linear equality constraint: pinned design dofs keep their initial values)."""
        assert len(pin_dir0) == len(self.opt_field) and len(pin_side0) == len(self.opt_field)
        if not hasattr(self, "shopt_dcpaligndcpffd"):
            self.set_shopt_align_CPFFD(None)
        for field_ind, field in enumerate(self.opt_field):
            if pin_dir0[field_ind] is None:
                continue
            d1 = None if pin_dir1 is None else pin_dir1[field_ind]
            s1 = None if pin_dir1 is None else pin_side1[field_ind]
            cand = self.CPpinDoFs(pin_dir0[field_ind], pin_side0[field_ind], d1, s1, self.shopt_cpffd_shape)
            design = set(self.shopt_cpffd_design_dof[field_ind])
            self.shopt_cpffd_pin_dof[field_ind] += [int(d) for d in cand if int(d) in design]
        self.shopt_cpffd_pin_dof = [sorted(set(p)) for p in self.shopt_cpffd_pin_dof]
        self.shopt_pin_vals = [None for _ in self.opt_field]
        self.shopt_dcppindcpffd = [None for _ in self.opt_field]
        for field_ind, field in enumerate(self.opt_field):
            pins = self.shopt_cpffd_pin_dof[field_ind]
            if len(pins) > 0:
                self.shopt_dcppindcpffd[field_ind] = self.dCPpindCPFFD(self.shopt_cpffd_design_dof[field_ind], pins)
                self.shopt_pin_vals[field_ind] = self.shopt_cpffd_flat[:, field][pins]
        self.pin_field = [f for i, f in enumerate(self.opt_field) if self.shopt_dcppindcpffd[i] is not None]
        return self.shopt_dcppindcpffd

    def dCPregudCPFFD(self, field, l, m, n, cpffd_design_dof):
        """nonmatching_opt_ffd.py:1206-1244: differences of neighbouring design control points along the optimised
        coordinate (keeps the block from folding; linear inequality constraint).  Rows in the reference's loop order."""
        shape = (int(l), int(m), int(n))
        if field not in (0, 1, 2):
            raise ValueError("Unsupported field {}".format(field))
        idx = [np.arange(s - 1 if d == field else s) for d, s in enumerate(shape)]
        gi, gj, gk = np.meshgrid(*idx, indexing="ij")                   # i outermost, k innermost
        lo = gi.ravel() + gj.ravel() * shape[0] + gk.ravel() * shape[0] * shape[1]
        step = (1, shape[0], shape[0] * shape[1])[field]
        nrow = lo.size
        rows = np.concatenate([np.arange(nrow), np.arange(nrow)])
        cols = np.concatenate([lo, lo + step])
        vals = np.concatenate([-np.ones(nrow), np.ones(nrow)])
        return sp.coo_matrix((vals, (rows, cols)), shape=(nrow, len(cpffd_design_dof)))

    def set_shopt_regu_CPFFD(self):
        """nonmatching_opt_ffd.py:870-883."""
        if not hasattr(self, "shopt_dcpaligndcpffd"):
            self.set_shopt_align_CPFFD(None)
        self.shopt_dcpregudcpffd = []
        for field_ind, field in enumerate(self.opt_field):
            l, m, nn = self.shopt_cpffd_shape
            align = self.shopt_align_dir[field_ind]
            if align is not None:
                l, m, nn = (1 if 0 in align else l), (1 if 1 in align else m), (1 if 2 in align else nn)
            self.shopt_dcpregudcpffd.append(self.dCPregudCPFFD(field, l, m, nn, self.shopt_cpffd_design_dof[field_ind]))
        return self.shopt_dcpregudcpffd

    # ------------------------------------------------------------------ shape optimisation with several FFD blocks
    def set_shopt_surf_inds_multiFFD(self, opt_field_mffd, shopt_surf_ind_list_mffd):
        """nonmatching_opt_ffd.py:184-310: block ``k`` drives the coordinates ``opt_field_mffd[k]`` of the patches
        ``shopt_surf_ind_list_mffd[k]``.  ``opt_field`` becomes the sorted union of the fields and
        ``shopt_surf_inds[field]`` the sorted union of the patches of the blocks that drive that field."""
        assert len(opt_field_mffd) == len(shopt_surf_ind_list_mffd)
        self.opt_field_mffd = [list(f) for f in opt_field_mffd]
        self.shopt_surf_ind_list_mffd = [list(s) for s in shopt_surf_ind_list_mffd]
        self.shopt_num_ffd = len(self.shopt_surf_ind_list_mffd)
        opt_field = sorted({f for fl in self.opt_field_mffd for f in fl})
        surf_inds = [sorted({s for k in range(self.shopt_num_ffd) if f in self.opt_field_mffd[k] for s in self.shopt_surf_ind_list_mffd[k]}) for f in opt_field]
        for f in opt_field:                          # a patch may be driven by one block only per field
            drv = [s for k in range(self.shopt_num_ffd) if f in self.opt_field_mffd[k] for s in self.shopt_surf_ind_list_mffd[k]]
            if len(drv) != len(set(drv)):
                raise ValueError("set_shopt_surf_inds_multiFFD: field %d of a patch is driven by more than one FFD block" % f)
        self.set_shopt_surf_inds(opt_field, surf_inds)
        self.shopt_multiffd = True
        self.opt_field_ffdinds = [[k for k in range(self.shopt_num_ffd) if f in self.opt_field_mffd[k]] for f in self.opt_field]
        w = np.concatenate([s.cp_hom_flat()[:, 3] for s in self.splines])
        self._mffd_cols = [np.concatenate([np.arange(self.cp_off[s], self.cp_off[s + 1]) for s in blk]) for blk in self.shopt_surf_ind_list_mffd]
        self._mffd_w = [w[c] for c in self._mffd_cols]
        self._mffd_X = [np.stack([self.cp_iga[f][c] / w[c] for f in range(3)], 1) for c in self._mffd_cols]
        self.shopt_cpsurf_lims_mffd = [[[float(X[:, f].min()), float(X[:, f].max())] for f in range(3)] for X in self._mffd_X]
        self.init_cp_iga = None
        self.get_init_CPIGA()

    def set_shopt_multiFFD(self, shopt_knots_mffd, shopt_cp_mffd):
        """nonmatching_opt_ffd.py:312-390.  Returns, per opt field, the constant sparse map from the concatenated
        control points of the blocks driving that field to the homogeneous surface control points of
        ``shopt_surf_inds[field]`` (rows in that patch order)."""
        from .utils.ffd_utils import CP_FFD_matrix
        assert len(shopt_knots_mffd) == self.shopt_num_ffd and len(shopt_cp_mffd) == self.shopt_num_ffd
        self.shopt_knots_mffd = [[np.asarray(k, float) for k in kn] for kn in shopt_knots_mffd]
        self.shopt_cp_mffd = [np.asarray(c, float) for c in shopt_cp_mffd]
        self.shopt_cp_mffd_flat_decate = [c[..., 0:3].transpose(2, 1, 0, 3).reshape(-1, 3) for c in self.shopt_cp_mffd]
        self.shopt_cp_mffd_flat = np.concatenate(self.shopt_cp_mffd_flat_decate, axis=0)
        self.shopt_cp_mffd_degree = [[int(np.sum(k == k[0]) - 1) for k in kn] for kn in self.shopt_knots_mffd]
        self.shopt_cp_mffd_shape = [c.shape[0:3] for c in self.shopt_cp_mffd]
        self.shopt_cp_mffd_size = [int(np.prod(s)) for s in self.shopt_cp_mffd_shape]
        self.shopt_cp_mffd_design_size = int(np.sum(self.shopt_cp_mffd_size))
        self.shopt_dcpsurf_fedcp_mffd_list = [sp.diags(self._mffd_w[k]).dot(CP_FFD_matrix(self._mffd_X[k], self.shopt_cp_mffd_degree[k],
                                                                                        self.shopt_knots_mffd[k]).tocsr())
                                              for k in range(self.shopt_num_ffd)]
        self.shopt_dcpsurf_fedcp_mffd = []
        for field_ind, field in enumerate(self.opt_field):
            blocks = self.opt_field_ffdinds[field_ind]
            D = sp.block_diag([self.shopt_dcpsurf_fedcp_mffd_list[k] for k in blocks], format="csr")
            # rows of D follow (block, patch of the block); reorder them to the sorted patch order of shopt_surf_inds[field]
            src = np.concatenate([self._mffd_cols[k] for k in blocks])
            order = np.argsort(src, kind="stable")
            assert np.array_equal(src[order], self._shopt_cols[field_ind])
            self.shopt_dcpsurf_fedcp_mffd.append(D[order].tocoo())
        self.shopt_num_desvars = [int(sum(self.shopt_cp_mffd_size[k] for k in self.opt_field_ffdinds[fi])) for fi in range(len(self.opt_field))]
        self.shopt_cp_mffd_design_dof = []
        for fi in range(len(self.opt_field)):
            off, lst = 0, []
            for k in self.opt_field_ffdinds[fi]:
                lst.append(list(range(off, off + self.shopt_cp_mffd_size[k])))
                off += self.shopt_cp_mffd_size[k]
            self.shopt_cp_mffd_design_dof.append(lst)
        self.shopt_cp_mffd_design_dof_full = [[d for sub in lst for d in sub] for lst in self.shopt_cp_mffd_design_dof]
        self.shopt_cp_mffd_design_dof_full_decate = [[list(sub) for sub in lst] for lst in self.shopt_cp_mffd_design_dof]
        self._mffd_block_off = [[sub[0] for sub in lst] for lst in self.shopt_cp_mffd_design_dof]
        self.shopt_init_cp_mffd_full = [self.get_init_CP_multiFFD(f) for f in self.opt_field]
        self.shopt_init_cp_mffd_design = [v.copy() for v in self.shopt_init_cp_mffd_full]
        self.shopt_align_dir_mffd = [None for _ in range(self.shopt_num_ffd)]
        self.shopt_dcpaligndcp_mffd_list = [[sp.identity(self.shopt_cp_mffd_size[k], format="coo") for k in self.opt_field_ffdinds[fi]]
                                            for fi in range(len(self.opt_field))]
        self.shopt_dcpaligndcp_mffd = [sp.block_diag(l, format="coo") for l in self.shopt_dcpaligndcp_mffd_list]
        self.shopt_cp_mffd_pin_dof = [[] for _ in self.opt_field]
        self.shopt_pin_vals = [None for _ in self.opt_field]
        self.shopt_dcppindcp_mffd = [None for _ in self.opt_field]
        self.pin_field = []
        return self.shopt_dcpsurf_fedcp_mffd

    def get_init_CP_multiFFD(self, field):
        """nonmatching_opt_ffd.py:422-429."""
        fi = self.opt_field.index(field)
        return np.concatenate([self.shopt_cp_mffd_flat_decate[k][:, field] for k in self.opt_field_ffdinds[fi]])

    def _mffd_local(self, ffd_ind, field):
        fi = self.opt_field.index(field)
        return fi, self.opt_field_ffdinds[fi].index(ffd_ind)

    def set_shopt_align_CP_multiFFD(self, ffd_ind, align_dir):
        """nonmatching_opt_ffd.py:726-756."""
        assert len(align_dir) == len(self.opt_field_mffd[ffd_ind])
        self.shopt_align_dir_mffd[ffd_ind] = list(align_dir)
        for k, field in enumerate(self.opt_field_mffd[ffd_ind]):
            fi, bi = self._mffd_local(ffd_ind, field)
            if align_dir[k] is not None:
                free_dof, deriv = self.dCPaligndCPFFD(field, align_dir[k], self.shopt_cp_mffd_shape[ffd_ind])
                self.shopt_dcpaligndcp_mffd_list[fi][bi] = deriv
                self.shopt_cp_mffd_design_dof[fi][bi] = [d + self._mffd_block_off[fi][bi] for d in free_dof]
        self.shopt_init_cp_mffd_design = [self.shopt_init_cp_mffd_full[fi][[d for sub in self.shopt_cp_mffd_design_dof[fi] for d in sub]]
                                          for fi in range(len(self.opt_field))]
        self.shopt_dcpaligndcp_mffd = [sp.block_diag(l, format="coo") for l in self.shopt_dcpaligndcp_mffd_list]
        return self.shopt_dcpaligndcp_mffd

    def set_shopt_pin_CP_multiFFD(self, ffd_ind, pin_dir0, pin_side0, pin_dir1=None, pin_side1=None):
        """nonmatching_opt_ffd.py:817-868."""
        assert len(pin_dir0) == len(self.opt_field_mffd[ffd_ind]) and len(pin_side0) == len(self.opt_field_mffd[ffd_ind])
        for k, field in enumerate(self.opt_field_mffd[ffd_ind]):
            fi, bi = self._mffd_local(ffd_ind, field)
            if pin_dir0[k] is None:
                continue
            d1 = None if pin_dir1 is None else pin_dir1[k]
            s1 = None if pin_dir1 is None else pin_side1[k]
            cand = self.CPpinDoFs(pin_dir0[k], pin_side0[k], d1, s1, self.shopt_cp_mffd_shape[ffd_ind]) + self._mffd_block_off[fi][bi]
            design = set(self.shopt_cp_mffd_design_dof[fi][bi])
            self.shopt_cp_mffd_pin_dof[fi] += [int(d) for d in cand if int(d) in design]
        for k, field in enumerate(self.opt_field_mffd[ffd_ind]):
            fi = self.opt_field.index(field)
            pins = self.shopt_cp_mffd_pin_dof[fi]
            if len(pins) > 0:
                design = [d for sub in self.shopt_cp_mffd_design_dof[fi] for d in sub]
                self.shopt_dcppindcp_mffd[fi] = self.dCPpindCPFFD(design, pins)
                self.shopt_pin_vals[fi] = self.shopt_init_cp_mffd_full[fi][pins]
        self.pin_field = [f for fi, f in enumerate(self.opt_field) if self.shopt_dcppindcp_mffd[fi] is not None]
        return self.shopt_dcppindcp_mffd

    def set_shopt_regu_CP_multiFFD(self):
        """nonmatching_opt_ffd.py:885-913."""
        lists = [[None for _ in self.opt_field_ffdinds[fi]] for fi in range(len(self.opt_field))]
        for ffd_ind in range(self.shopt_num_ffd):
            for k, field in enumerate(self.opt_field_mffd[ffd_ind]):
                fi, bi = self._mffd_local(ffd_ind, field)
                l, m, nn = self.shopt_cp_mffd_shape[ffd_ind]
                align = None if self.shopt_align_dir_mffd[ffd_ind] is None else self.shopt_align_dir_mffd[ffd_ind][k]
                if align is not None:
                    l, m, nn = (1 if 0 in align else l), (1 if 1 in align else m), (1 if 2 in align else nn)
                lists[fi][bi] = self.dCPregudCPFFD(field, l, m, nn, self.shopt_cp_mffd_design_dof[fi][bi])
        self.shopt_dcpregudcp_mffd_list = lists
        self.shopt_dcpregudcp_mffd = [sp.block_diag(l, format="coo") for l in lists]
        return self.shopt_dcpregudcp_mffd

    # ------------------------------------------------------------------ thickness FFD
    def set_thopt_surf_inds_FFD(self, thopt_surf_inds):
        """nonmatching_opt_ffd.py:434-464: patches whose thickness field is driven by one FFD block."""
        self.thopt_multiffd = False
        self.thopt_surf_inds = list(thopt_surf_inds)
        self._thopt_cols = np.concatenate([np.arange(self.cp_off[s], self.cp_off[s + 1]) for s in self.thopt_surf_inds])
        w = np.concatenate([s.cp_hom_flat()[:, 3] for s in self.splines])[self._thopt_cols]
        self.thopt_cpsurf_des = np.stack([self.cp_iga[f][self._thopt_cols] / w for f in range(3)], 1)
        self.thopt_cpsurf_des_lims = [[float(self.thopt_cpsurf_des[:, f].min()), float(self.thopt_cpsurf_des[:, f].max())] for f in range(3)]

    def set_thopt_FFD(self, thopt_knotsffd, thopt_cpffd):
        """nonmatching_opt_ffd.py:497-521: the thickness at a control point is the trivariate B-spline of the block's
        thickness coefficients evaluated at the control point's physical position (variable-thickness path)."""
        from .utils.ffd_utils import CP_FFD_matrix
        if not getattr(self, "var_thickness", False):
            self.set_thickness_opt(var_thickness=True)
        self.thopt_knotsffd = [np.asarray(k, float) for k in thopt_knotsffd]
        self.thopt_cpffd = np.asarray(thopt_cpffd, float)
        self.thopt_cpffd_flat = self.thopt_cpffd[..., 0:3].transpose(2, 1, 0, 3).reshape(-1, 3)
        self.thopt_ffd_degree = int(np.sum(self.thopt_knotsffd[0] == self.thopt_knotsffd[0][0]) - 1)
        self.thopt_cpffd_shape = self.thopt_cpffd.shape[0:3]
        self.thopt_cpffd_size = int(np.prod(self.thopt_cpffd_shape))
        self.thopt_cpffd_design_size = self.thopt_cpffd_size
        self.thopt_dcpsurf_fedcpffd = CP_FFD_matrix(self.thopt_cpsurf_des, [self.thopt_ffd_degree] * 3, self.thopt_knotsffd).tocoo()
        self.init_h_th_ffd = None
        return self.thopt_dcpsurf_fedcpffd

    def get_init_h_th_FFD(self):
        """nonmatching_opt_ffd.py:523-532: least-squares block coefficients reproducing the initial thickness."""
        if self.init_h_th_ffd is None:
            A = self.thopt_dcpsurf_fedcpffd.toarray()
            h0 = np.concatenate(self.h_th)[self._thopt_cols]
            self.init_h_th_ffd = np.linalg.lstsq(A, h0, rcond=None)[0]
        return self.init_h_th_ffd

    def dCPaligndCPFFD_thopt(self, align_dir, cp_align_size, cpffd_size, cpffd_shape):
        """nonmatching_opt_ffd.py:1085-1118: coefficient(first layer) - coefficient(layer i) = 0 along each direction."""
        ijk = self._lattice(cpffd_shape)
        l, m, _ = (int(s) for s in cpffd_shape)
        rows, cols, vals, r0 = [], [], [], 0
        for d in align_dir:
            order = {0: (2, 1, 0), 1: (2, 0, 1), 2: (1, 0, 2)}[int(d)]     # loop nest of the reference, outer -> inner
            pts = ijk[ijk[:, d] > 0]
            pts = pts[np.lexsort((pts[:, order[2]], pts[:, order[1]], pts[:, order[0]]))]
            first = pts.copy(); first[:, d] = 0
            nr = len(pts)
            rows += [np.arange(r0, r0 + nr)] * 2
            cols += [first[:, 0] + first[:, 1] * l + first[:, 2] * l * m, pts[:, 0] + pts[:, 1] * l + pts[:, 2] * l * m]
            vals += [np.ones(nr), -np.ones(nr)]
            r0 += nr
        assert r0 == cp_align_size
        return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(cp_align_size, cpffd_size))

    def set_thopt_align_CPFFD(self, align_dir):
        """nonmatching_opt_ffd.py:915-941."""
        self.thopt_align_dir = list(align_dir) if isinstance(align_dir, (list, tuple)) else [align_dir]
        shape = [int(s) for s in self.thopt_cpffd_shape]
        self.thopt_cp_align_size = int(sum(np.prod([s - 1 if i == d else s for i, s in enumerate(shape)]) for d in self.thopt_align_dir))
        self.thopt_dcpaligndcpffd = self.dCPaligndCPFFD_thopt(self.thopt_align_dir, self.thopt_cp_align_size, self.thopt_cpffd_size, self.thopt_cpffd_shape)
        return self.thopt_dcpaligndcpffd

    def set_thopt_regu_CPFFD(self, regu_dir, regu_side, regu_align=None):
        """nonmatching_opt_ffd.py:943-997 with opt_field = [2] as there: differences of neighbouring block coefficients
        along direction 2, optionally only on the face ``regu_dir``/``regu_side`` or on the first layer of an aligned
        direction ``regu_align``."""
        self.thopt_regu_dir, self.thopt_regu_side, self.thopt_regu_align = regu_dir, regu_side, regu_align
        shape = [int(s) for s in self.thopt_cpffd_shape]
        field = 2
        ijk = self._lattice(shape)
        keep = ijk[:, field] < shape[field] - 1
        if regu_dir[0] is not None:
            keep &= ijk[:, regu_dir[0]] == (0 if regu_side[0] == 0 else shape[regu_dir[0]] - 1)
        elif regu_align is not None and regu_align[0] is not None:
            if regu_align[0] == field:
                raise ValueError("Optimization filed cannot equal to align direction")
            keep &= ijk[:, regu_align[0]] == 0
        pts = ijk[keep]
        pts = pts[np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))]        # k outermost as in the reference's face loops (row order is immaterial)
        l, m = shape[0], shape[1]
        lo = pts[:, 0] + pts[:, 1] * l + pts[:, 2] * l * m
        nr = lo.size
        deriv = sp.coo_matrix((np.concatenate([-np.ones(nr), np.ones(nr)]), (np.concatenate([np.arange(nr)] * 2), np.concatenate([lo, lo + l * m]))),
                              shape=(nr, self.thopt_cpffd_size))
        self.thopt_cpregu_sizes = [nr]
        self.thopt_dcpregudcpffd_list = [deriv]
        return self.thopt_dcpregudcpffd_list

    # ------------------------------------------------------------------ thickness with several FFD blocks
    def set_thopt_multiFFD_surf_inds(self, thopt_multiffd_surf_ind_list):
        """nonmatching_opt_ffd.py:534-598: block k drives the thickness field of the patches in
        ``thopt_multiffd_surf_ind_list[k]``; the remaining patches keep one constant thickness each."""
        self.thopt_multiffd = True
        self.thopt_multiffd_surf_ind_list = [list(s) for s in thopt_multiffd_surf_ind_list]
        self.num_thopt_ffd = len(self.thopt_multiffd_surf_ind_list)
        self.thopt_ffd_shell_inds = [s for blk in self.thopt_multiffd_surf_ind_list for s in blk]
        if len(set(self.thopt_ffd_shell_inds)) != len(self.thopt_ffd_shell_inds):
            raise ValueError("set_thopt_multiFFD_surf_inds: a patch is driven by more than one thickness FFD block")
        self.thopt_nonffd_shell_inds = [s for s in range(self.num_splines) if s not in self.thopt_ffd_shell_inds]
        self.num_thopt_nonffd_shells = len(self.thopt_nonffd_shell_inds)
        w = np.concatenate([s.cp_hom_flat()[:, 3] for s in self.splines])
        self._thm_cols = [np.concatenate([np.arange(self.cp_off[s], self.cp_off[s + 1]) for s in blk]) for blk in self.thopt_multiffd_surf_ind_list]
        self._thm_X = [np.stack([self.cp_iga[f][c] / w[c] for f in range(3)], 1) for c in self._thm_cols]
        self.thopt_cpsurf_lims_multiffd = [[[float(X[:, f].min()), float(X[:, f].max())] for f in range(3)] for X in self._thm_X]

    def set_thopt_multiFFD(self, thopt_knotsffd_list, thopt_cpffd_list):
        """nonmatching_opt_ffd.py:621-664.  Returns the map [block coefficients of all blocks | constant thicknesses of the
        remaining patches] -> thickness at every control point (all patches, patch order)."""
        from .utils.ffd_utils import CP_FFD_matrix
        if not getattr(self, "var_thickness", False):
            self.set_thickness_opt(var_thickness=True)
        self.thopt_knotsffd_list = [[np.asarray(k, float) for k in kn] for kn in thopt_knotsffd_list]
        self.thopt_cpffd_list = [np.asarray(c, float) for c in thopt_cpffd_list]
        self.thopt_cpffd_flat_list = [c[..., 0:3].transpose(2, 1, 0, 3).reshape(-1, 3) for c in self.thopt_cpffd_list]
        self.thopt_cpffd_degree_list = [int(np.sum(kn[0] == kn[0][0]) - 1) for kn in self.thopt_knotsffd_list]
        self.thopt_cpffd_shape_list = [c.shape[0:3] for c in self.thopt_cpffd_list]
        self.thopt_cpffd_size_list = [int(np.prod(s)) for s in self.thopt_cpffd_shape_list]
        self.thopt_dcpsurf_fedcpffd_list = [CP_FFD_matrix(self._thm_X[k], [self.thopt_cpffd_degree_list[k]] * 3, self.thopt_knotsffd_list[k]).tocsr()
                                            for k in range(self.num_thopt_ffd)]
        blocks = list(self.thopt_dcpsurf_fedcpffd_list)
        src = list(self._thm_cols)
        for s in self.thopt_nonffd_shell_inds:
            blocks.append(sp.csr_matrix(np.ones((self.vec_scalar_iga_dof_list[s], 1))))
            src.append(np.arange(self.cp_off[s], self.cp_off[s + 1]))
        self.thopt_cpffd_design_size = int(sum(self.thopt_cpffd_size_list)) + self.num_thopt_nonffd_shells
        D = sp.block_diag(blocks, format="csr")
        order = np.argsort(np.concatenate(src), kind="stable")               # rows back to patch order (h_th_FE_reorder, :600-619)
        self.thopt_dcpsurf_fedcpmultiffd = D[order].tocoo()
        self.init_h_th_multiffd = None
        return self.thopt_dcpsurf_fedcpmultiffd

    def get_init_h_th_multiFFD(self):
        """nonmatching_opt_ffd.py:666-685."""
        if self.init_h_th_multiffd is None:
            h = np.concatenate(self.h_th)
            parts = [np.linalg.lstsq(self.thopt_dcpsurf_fedcpffd_list[k].toarray(), h[self._thm_cols[k]], rcond=None)[0] for k in range(self.num_thopt_ffd)]
            parts += [np.array([float(np.mean(self.h_th[s]))]) for s in self.thopt_nonffd_shell_inds]
            self.init_h_th_multiffd = np.concatenate(parts)
        return self.init_h_th_multiffd

    def set_thopt_align_CP_multiFFD(self, align_dir_list):
        """nonmatching_opt_ffd.py:999-1032: per-block alignment rows, zero columns for the constant thicknesses."""
        assert len(align_dir_list) == self.num_thopt_ffd
        self.thopt_align_dir_list = list(align_dir_list)
        mats = []
        for k, ad in enumerate(self.thopt_align_dir_list):
            ad = list(ad) if isinstance(ad, (list, tuple)) else [ad]
            shape = [int(s) for s in self.thopt_cpffd_shape_list[k]]
            size = int(sum(np.prod([s - 1 if i == d else s for i, s in enumerate(shape)]) for d in ad))
            mats.append(self.dCPaligndCPFFD_thopt(ad, size, self.thopt_cpffd_size_list[k], shape))
        self.thopt_dcpaligndcpffd_list = mats
        A = sp.block_diag(mats, format="coo")
        if self.num_thopt_nonffd_shells > 0:
            A = sp.hstack([A, sp.coo_matrix((A.shape[0], self.num_thopt_nonffd_shells))], format="coo")
        self.thopt_dcpaligndcpmultiffd = A
        return self.thopt_dcpaligndcpmultiffd

    @property
    def cpsurf_lims(self):
        """Bounding box of the optimised surfaces' physical control points (reference attribute used to
        size the FFD block, e.g. om_comps/ffd_comps/cpffd2surf_comp.py:71-75)."""
        w = np.concatenate([s.cp_hom_flat()[:, 3] for s in self.splines])
        return [[float((self.cp_iga[f] / w).min()), float((self.cp_iga[f] / w).max())] for f in range(3)]
