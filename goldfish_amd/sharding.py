"""Patch sharding for one-process-per-GPU runs.

Shell terms are patch-independent (block-diagonal K, dR/dCP, dR/dh:
GOLDFISH/nonmatching_opt.py:815-823, 933-937); coupling enters only through the
interfaces (``mapping_list[i] = [a, b]``, :745-752, 789-801).  Each rank owns a
contiguous range of patches and assembles exactly the rows of those patches
("owner computes rows"): for interfaces cut by the partition the neighbour patch
is carried as a *ghost* (geometry + state only, no rows), so no matrix entries ever
cross xGMI.  What does cross it is (a) the residual / forward products (row slices ->
one all-reduce of a zero-padded global vector) and (b) reverse-mode products whose
columns belong to remote patches (all-reduce of the global-length result).
The reference's counterpart is ``comm.allgather`` of every vector to every rank
(GOLDFISH/utils/opt_utils.py:41-54).
"""
from dataclasses import dataclass

import numpy as np

from .geometry import ProblemSpec
from .model import Interface, arrays_from_spec


def partition_patches(spec, world):
    """Contiguous patch ranges balanced by Gauss-point count."""
    w = np.array([p.nel[0] * p.nel[1] * (p.p + 1) * (p.q + 1) for p in spec.patches], float)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    bounds = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        k = int(np.argmin(np.abs(cum - target)))
        k = max(k, bounds[-1] + 1)
        k = min(k, len(spec.patches) - (world - r))
        bounds.append(k)
    bounds.append(len(spec.patches))
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


@dataclass
class Shard:
    rank: int
    world: int
    spec: ProblemSpec            # local: owned patches first, then ghosts
    n_owned: int
    order: list                  # local patch index -> global patch index
    cp_off_global: np.ndarray    # global control-point offsets (all patches)
    cp_off_local: np.ndarray
    patch_ranges: list = None    # (first, end) owned patch range of every rank

    @property
    def total_cp_global(self):
        return int(self.cp_off_global[-1])

    def to_local(self, vec, width=1):
        """Slice a global patch-major vector (width values per control point) to local order."""
        return np.concatenate([vec[width * self.cp_off_global[g]:width * self.cp_off_global[g + 1]] for g in self.order])

    def owned_global_range(self, width=1):
        g0, g1 = self.order[0], self.order[self.n_owned - 1] + 1
        return width * int(self.cp_off_global[g0]), width * int(self.cp_off_global[g1])

    def owned_local_size(self, width=1):
        return width * int(self.cp_off_local[self.n_owned])

    def local_cols_to_global(self):
        """Global control-point id of every local control point (owned + ghost)."""
        return np.concatenate([np.arange(self.cp_off_global[g], self.cp_off_global[g + 1]) for g in self.order])


def shard_spec(spec, rank, world):
    start, end = partition_patches(spec, world)[rank]
    own = list(range(start, end))
    ghosts = set()
    for itf in spec.interfaces:
        if start <= itf.a < end and not (start <= itf.b < end):
            ghosts.add(itf.b)
        if start <= itf.b < end and not (start <= itf.a < end):
            ghosts.add(itf.a)
    order = own + sorted(ghosts)
    g2l = {g: l for l, g in enumerate(order)}
    itfs = []
    for itf in spec.interfaces:
        if (start <= itf.a < end) or (start <= itf.b < end):
            loc = Interface(g2l[itf.a], g2l[itf.b], itf.xi_a, itf.xi_b)   # keeps the (A, B) orientation
            itfs.append(loc)
    pls = [(g2l[s], xi, f, v) for (s, xi, f, v) in spec.point_loads if start <= s < end]
    local = ProblemSpec([spec.patches[g] for g in order], itfs, spec.E, spec.nu, spec.h_th,
                        [spec.body_force[g] for g in order], pls, spec.penalty_coefficient,
                        "%s[rank %d/%d]" % (spec.name, rank, world),
                        None if getattr(spec, "load_proj", None) is None else [spec.load_proj[g] for g in order])
    cpg = np.concatenate([[0], np.cumsum([p.ncp for p in spec.patches])]).astype(np.int64)
    cpl = np.concatenate([[0], np.cumsum([p.ncp for p in local.patches])]).astype(np.int64)
    return Shard(rank, world, local, len(own), order, cpg, cpl, partition_patches(spec, world))


def shard_arrays(shard, thickness_global=None):
    """ModelArrays of the local model (owned + ghost patches); thickness_global is the
    per-patch list used to freeze the penalty parameters (identical on every rank)."""
    th = None
    if thickness_global is not None:
        th = [thickness_global[g] for g in shard.order]
    A = arrays_from_spec(shard.spec, th)
    A.n_owned = shard.n_owned
    A.n_gauss_points = int(sum(p.nel[0] * p.nel[1] * (p.p + 1) * (p.q + 1) for p in shard.spec.patches[:shard.n_owned]))
    return A


def allreduce_owned_rows(shard, local_rows, dist, width=3, out=None):
    """Place this rank's owned rows into a zero-padded global vector and sum over ranks
    (torch.distributed; backend 'nccl' == RCCL over xGMI on the GPU box, 'gloo' in CPU tests).
    ``local_rows`` is a torch tensor holding at least the owned rows first."""
    import torch
    n = width * shard.total_cp_global
    if out is None:
        out = torch.zeros(n, dtype=torch.float64, device=local_rows.device)
    else:
        out.zero_()
    g0, g1 = shard.owned_global_range(width)
    out[g0:g1] = local_rows[:g1 - g0]
    if shard.world > 1:
        dist.all_reduce(out)
    return out


def allgather_owned_rows(shard, local_rows, dist, width=3, out=None):
    """Same result as allreduce_owned_rows with half the traffic: the owned row slices are disjoint and contiguous in the
    global vector, so they are exchanged by ONE all-gather of slices padded to the largest one ((N-1)/N of the vector per
    rank over xGMI instead of 2(N-1)/N for the ring all-reduce of a zero-padded vector) and copied into place."""
    import torch
    n = width * shard.total_cp_global
    if out is None:
        out = torch.zeros(n, dtype=torch.float64, device=local_rows.device)
    rng = [(width * int(shard.cp_off_global[a]), width * int(shard.cp_off_global[b])) for (a, b) in shard.patch_ranges]
    g0, g1 = rng[shard.rank]
    if shard.world == 1:
        out[g0:g1] = local_rows[:g1 - g0]
        return out
    mx = max(b - a for a, b in rng)
    cache = shard.__dict__.setdefault("_exchange_buffers", {})            # staging buffers are allocated once per (width, device)
    key = (width, str(local_rows.device))
    if key not in cache:
        cache[key] = (torch.zeros(mx, dtype=torch.float64, device=local_rows.device),
                      torch.empty(shard.world * mx, dtype=torch.float64, device=local_rows.device))
    send, recv_buf = cache[key]
    send[:g1 - g0] = local_rows[:g1 - g0]                                  # the padding behind it stays zero
    if send.is_cuda:
        # local_rows is usually a view of the library's residual buffer, which the next assembly (on the library's own
        # stream) overwrites: wait for this one copy, not for the collective, so that the exchange overlaps the next step
        ev = torch.cuda.Event()
        ev.record()
        ev.synchronize()
    recv = recv_buf
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(recv, send)
    else:                                                       # gloo (tests, rehearsal): host tensors
        parts = [torch.empty(mx, dtype=torch.float64) for _ in range(shard.world)]
        dist.all_gather(parts, send.cpu())
        recv = torch.cat(parts).to(local_rows.device)
    for r, (a, b) in enumerate(rng):
        out[a:b] = recv[r * mx:r * mx + (b - a)]
    return out


class ShardedDeviceModel:
    """One rank's view of a patch-sharded model with *global-vector* semantics:
    state setters take the replicated global arrays (what OpenMDAO hands every rank, like the reference's
    allgathered vectors, GOLDFISH/utils/opt_utils.py:41-54), results come back as replicated global arrays.

    residual / forward products : owned row slices -> all-reduce of the zero-padded global vector
    reverse products            : local columns (owned + ghost) scattered to global ids -> all-reduce
    functionals                 : owned elements / interfaces owned through side A -> all-reduce of scalars,
                                  gradients like reverse products
    ``dist`` is torch.distributed (backend nccl == RCCL over xGMI on a multi-GPU node, gloo in tests)."""

    def __init__(self, spec, dist, rank, world, device=0, thickness_global=None):
        from . import _lib
        self._lib, self.dist, self.rank, self.world = _lib, dist, rank, world
        self.shard = shard_spec(spec, rank, world)
        self.A = shard_arrays(self.shard, thickness_global)
        self.D = _lib.DeviceModel(self.A, device=device)
        self.cols_g = self.shard.local_cols_to_global()
        self.total_cp, self.ndof = self.shard.total_cp_global, 3 * self.shard.total_cp_global
        self.n_owned_cp = int(self.shard.cp_off_local[self.shard.n_owned])
        self.g0 = int(self.shard.cp_off_global[self.shard.order[0]])

    def close(self):
        self.D.close()

    # -- replicated inputs
    def set_cp(self, field, v):
        self.D.set_cp(field, self.shard.to_local(np.asarray(v, float)))

    def set_thickness(self, v):
        self.D.set_thickness(self.shard.to_local(np.asarray(v, float)))

    def set_u(self, v):
        self.D.set_u(self.shard.to_local(np.asarray(v, float), 3))

    def assemble(self, flags=15):
        self.D.assemble(flags)

    # -- exchange
    def _allreduce(self, arr):
        import torch
        if self.world == 1:
            return arr
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def _rows_to_global(self, local_rows, width):
        out = np.zeros(width * self.total_cp)
        n = width * self.n_owned_cp
        out[width * self.g0:width * self.g0 + n] = local_rows[:n]
        return self._allreduce(out)

    def _cols_to_global(self, local_cols):
        out = np.zeros(self.total_cp)
        np.add.at(out, self.cols_g, local_cols)
        return self._allreduce(out)

    def residual(self):
        return self._rows_to_global(self.D.residual(), 3)

    def apply(self, which, x, transpose=False):
        """Returns A x (or A^T x) for the replicated global x as a replicated global vector."""
        _lib = self._lib
        x = np.asarray(x, float)
        if not transpose:
            xl = self.shard.to_local(x, 3 if which == _lib.MAT_K else 1)
            y = np.zeros(self.A.ndof)
            self.D.apply(which, xl, y)
            return self._rows_to_global(y, 3)
        xl = self.shard.to_local(x, 3)
        xl[3 * self.n_owned_cp:] = 0.0                     # ghost rows are not assembled here
        if which == _lib.MAT_K:
            y = np.zeros(self.A.ndof)
            self.D.apply(which, xl, y, transpose=True)      # K^T: local columns are dofs (owned + ghost)
            out = np.zeros(self.ndof)
            for c in range(3):
                np.add.at(out, 3 * self.cols_g + c, y[c::3])
            return self._allreduce(out)
        y = np.zeros(self.A.total_cp)
        self.D.apply(which, xl, y, transpose=True)
        return self._cols_to_global(y)

    def functionals(self, apply_bcs=True):
        F = self.D.functionals(apply_bcs=apply_bcs)
        sc = self._allreduce(np.array([F["Wint"], F["volume"], F["Wpen"]]))
        out = dict(Wint=sc[0], volume=sc[1], Wpen=sc[2])
        n = self.n_owned_cp

        def own(v, width=1):                                # gradients are assembled for owned control points only
            w = np.array(v, float)
            w[width * n:] = 0.0
            return w
        out["dWdu"] = self._rows_to_global(own(F["dWdu"], 3), 3)
        out["dWdh"] = self._rows_to_global(own(F["dWdh"]), 1)
        out["dVdh"] = self._rows_to_global(own(F["dVdh"]), 1)
        out["dWdcp"] = [self._rows_to_global(own(F["dWdcp"][f]), 1) for f in range(3)]
        out["dVdcp"] = [self._rows_to_global(own(F["dVdcp"][f]), 1) for f in range(3)]
        return out

    def stress_forms(self, mode, rho, m_list, surf=1, measure=0, apply_bcs=True, gradients=True):
        """Global per-patch von Mises aggregation forms (gf_stress_forms): every rank evaluates its owned patches,
        the per-patch values and the owned gradient rows are summed over ranks."""
        order = np.asarray(self.shard.order)
        ml = np.asarray(m_list, float)
        F = self.D.stress_forms(mode, rho, ml[order], surf, measure, apply_bcs=apply_bcs, gradients=gradients)
        no, npg = self.shard.n_owned, ml.size
        I, vmax = np.zeros(npg), np.zeros(npg)
        I[order[:no]], vmax[order[:no]] = F["I"][:no], F["vmax"][:no]
        sc = self._allreduce(np.concatenate([I, vmax]))
        out = dict(I=sc[:npg], vmax=sc[npg:])
        if gradients:
            n = self.n_owned_cp

            def own(v, width=1):
                w = np.array(v, float)
                w[width * n:] = 0.0
                return w
            out["dIdu"] = self._rows_to_global(own(F["dIdu"], 3), 3)
            out["dIdh"] = self._rows_to_global(own(F["dIdh"]), 1)
            out["dIdcp"] = np.stack([self._rows_to_global(own(F["dIdcp"][f]), 1) for f in range(3)])
        return out

    def shape_regu(self, field, cp0_global, coef_global):
        """Global shape regularisation term (gf_shape_regu): owned elements per rank, value and owned gradient rows summed."""
        order = np.asarray(self.shard.order)
        F = self.D.shape_regu(field, self.shard.to_local(np.asarray(cp0_global, float)), np.asarray(coef_global, float)[order])
        n = self.n_owned_cp
        out = dict(value=float(self._allreduce(np.array([F["value"]]))[0]))
        dcp = []
        for f in range(3):
            w = np.array(F["dcp"][f], float)
            w[n:] = 0.0
            dcp.append(self._rows_to_global(w, 1))
        out["dcp"] = np.stack(dcp)
        return out
