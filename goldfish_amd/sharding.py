"""Patch sharding for one-process-per-GPU runs.


Shell terms are patch-independent (block-diagonal K, dR/dCP, dR/dh:
GOLDFISH/nonmatching_opt.py:815-823, 933-937); coupling enters only through the
interfaces (``mapping_list[i] = [a, b]``, :745-752, 789-801).  Each rank owns the
patches a partition of the interface graph assigns to it (balanced by Gauss points,
few cut interfaces: partition_patches) and assembles exactly the rows of those patches
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


def _patch_weights(spec):
    return np.array([p.nel[0] * p.nel[1] * (p.p + 1) * (p.q + 1) for p in spec.patches], float)


def partition_patches(spec, world):
    """Owner rank of every patch: partition of the interface graph balanced by Gauss points.

    Recursive bisection: a group of patches is split in two along the direction in which it is longest (patch centroids, the
    geometric embedding of the interface graph), at the weighted position that gives the two halves Gauss-point counts in
    proportion to the ranks they receive; then every cut is refined by moving patches across it while that lowers the number of cut
    interfaces (weighted by mortar vertices) without unbalancing the halves by more than one patch.  For the 16 x 16 C4 grid at 8 ranks
    this gives the 4 x 2 arrangement of 4 x 8-patch blocks with 64 of the 480 interfaces cut.  Returns an int array (n_patches)."""
    npatch = len(spec.patches)
    w = _patch_weights(spec)
    cen = np.array([(p.control[:, :, :3] / p.control[:, :, 3:4]).reshape(-1, 3).mean(0) for p in spec.patches])
    adj = [dict() for _ in range(npatch)]
    for itf in spec.interfaces:
        wt = float(getattr(itf, "npts", 1) or 1)
        adj[itf.a][itf.b] = adj[itf.a].get(itf.b, 0.0) + wt
        adj[itf.b][itf.a] = adj[itf.b].get(itf.a, 0.0) + wt
    part = np.zeros(npatch, dtype=np.int64)

    def split(ids, r0, nr):
        if nr == 1 or len(ids) <= 1:
            part[ids] = r0
            return
        nl = nr // 2
        ids = np.asarray(ids)
        ext = cen[ids].max(0) - cen[ids].min(0)
        order = ids[np.lexsort((ids, cen[ids, int(np.argmax(ext))]))]       # along the longest direction (ties: patch index)
        cw = np.cumsum(w[order])
        k = int(np.searchsorted(cw, cw[-1] * nl / nr - 1e-9)) + 1
        k = min(max(k, nl), len(order) - (nr - nl))                         # every rank keeps at least one patch
        side = {int(g): (0 if pos < k else 1) for pos, g in enumerate(order)}
        wl, target = cw[k - 1], cw[-1] * nl / nr
        slack = w[order].max()
        for _ in range(4 * len(order)):                                     # refinement: single moves that reduce the cut
            best, gain_best = None, 0.0
            for g in order:
                g = int(g)
                sd = side[g]
                gain = sum(wt * (1 if side[nb] != sd else -1) for nb, wt in adj[g].items() if nb in side)
                wl_new = wl - w[g] if sd == 0 else wl + w[g]
                if gain > gain_best + 1e-12 and abs(wl_new - target) <= max(abs(wl - target), slack) and \
                        sum(1 for x in side.values() if x == sd) > (nl if sd == 0 else nr - nl):
                    best, gain_best = g, gain
            if best is None:
                break
            wl = wl - w[best] if side[best] == 0 else wl + w[best]
            side[best] ^= 1
        left = [g for g in ids if side[int(g)] == 0]
        right = [g for g in ids if side[int(g)] == 1]
        split(left, r0, nl)
        split(right, r0 + nl, nr - nl)

    split(list(range(npatch)), 0, world)
    return part


def partition_quality(spec, part):
    """Diagnostics of a partition: Gauss points per rank, imbalance (max / mean), cut interfaces, ghost patches per rank."""
    part = np.asarray(part)
    world = int(part.max()) + 1
    w = _patch_weights(spec)
    gp = np.array([w[part == r].sum() for r in range(world)])
    cut = sum(1 for itf in spec.interfaces if part[itf.a] != part[itf.b])
    ghosts = [set() for _ in range(world)]
    for itf in spec.interfaces:
        if part[itf.a] != part[itf.b]:
            ghosts[part[itf.a]].add(itf.b)
            ghosts[part[itf.b]].add(itf.a)
    owned = np.array([(part == r).sum() for r in range(world)])
    return {"gauss_points": gp.astype(np.int64).tolist(), "imbalance": float(gp.max() / gp.mean()), "cut_interfaces": int(cut),
            "ghost_patches": [len(g) for g in ghosts], "owned_patches": owned.tolist(),
            "ghost_fraction_max": float(max(len(g) / max(o, 1) for g, o in zip(ghosts, owned))),
            "ghost_fraction_mean": float(np.mean([len(g) / max(o, 1) for g, o in zip(ghosts, owned)]))}


@dataclass
class Shard:
    rank: int
    world: int
    spec: ProblemSpec            # local: owned patches first, then ghosts
    n_owned: int
    order: list                  # local patch index -> global patch index
    cp_off_global: np.ndarray    # global control-point offsets (all patches)
    cp_off_local: np.ndarray
    owned_by_rank: list = None   # owned global patch ids of every rank (ascending)
    if_global: list = None       # global interface id of every local interface (local order)

    @property
    def total_cp_global(self):
        return int(self.cp_off_global[-1])

    def to_local(self, vec, width=1):
        """Slice a global patch-major vector (width values per control point) to local order."""
        return np.concatenate([vec[width * self.cp_off_global[g]:width * self.cp_off_global[g + 1]] for g in self.order])

    def owned_rows_global(self, width=1, rank=None):
        """Global row ids (width rows per control point) of the rows owned by ``rank`` (default: this rank), in its local order."""
        own = self.owned_by_rank[self.rank if rank is None else rank]
        return np.concatenate([np.arange(width * self.cp_off_global[g], width * self.cp_off_global[g + 1]) for g in own])

    def owned_local_size(self, width=1):
        return width * int(self.cp_off_local[self.n_owned])

    def local_cols_to_global(self):
        """Global control-point id of every local control point (owned + ghost)."""
        return np.concatenate([np.arange(self.cp_off_global[g], self.cp_off_global[g + 1]) for g in self.order])


def shard_spec(spec, rank, world, part=None):
    part = partition_patches(spec, world) if part is None else np.asarray(part)
    own = [int(g) for g in np.flatnonzero(part == rank)]
    mine = set(own)
    ghosts = set()
    for itf in spec.interfaces:
        if itf.a in mine and itf.b not in mine:
            ghosts.add(itf.b)
        if itf.b in mine and itf.a not in mine:
            ghosts.add(itf.a)
    order = own + sorted(ghosts)
    g2l = {g: l for l, g in enumerate(order)}
    itfs, if_global = [], []
    for gi, itf in enumerate(spec.interfaces):
        if itf.a in mine or itf.b in mine:
            loc = Interface(g2l[itf.a], g2l[itf.b], itf.xi_a, itf.xi_b)   # keeps the (A, B) orientation
            itfs.append(loc)
            if_global.append(gi)
    pls = [(g2l[s], xi, f, v) for (s, xi, f, v) in spec.point_loads if s in mine]
    def per_patch(v):                # a per-patch list follows the local patch order; a scalar is shared
        return [np.asarray(v).ravel()[g] for g in order] if np.ndim(v) > 0 and np.size(v) == len(spec.patches) else v
    local = ProblemSpec([spec.patches[g] for g in order], itfs, per_patch(spec.E), per_patch(spec.nu), spec.h_th,
                        [spec.body_force[g] for g in order], pls, spec.penalty_coefficient,
                        "%s[rank %d/%d]" % (spec.name, rank, world),
                        None if getattr(spec, "load_proj", None) is None else [spec.load_proj[g] for g in order],
                        pressure=None if getattr(spec, "pressure", None) is None else [spec.pressure[g] for g in order],
                        edge_traction=None if getattr(spec, "edge_traction", None) is None else
                        [(g2l[s], d, side, f) for (s, d, side, f) in spec.edge_traction if s in g2l])
    cpg = np.concatenate([[0], np.cumsum([p.ncp for p in spec.patches])]).astype(np.int64)
    cpl = np.concatenate([[0], np.cumsum([p.ncp for p in local.patches])]).astype(np.int64)
    return Shard(rank, world, local, len(own), order, cpg, cpl, [[int(g) for g in np.flatnonzero(part == r)] for r in range(world)], if_global)


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


def allreduce_owned_rows(shard, local_rows, dist, width=3, out=None, group=None):
    """Place this rank's owned rows into a zero-padded global vector and sum over ranks
    (torch.distributed; backend 'nccl' == RCCL over xGMI on the GPU box, 'gloo' in CPU tests).
    ``local_rows`` is a torch tensor holding at least the owned rows first."""
    import torch
    n = width * shard.total_cp_global
    if out is None:
        out = torch.zeros(n, dtype=torch.float64, device=local_rows.device)
    else:
        out.zero_()
    idx = _rows_index(shard, width, local_rows.device)[shard.rank]
    out[idx] = local_rows[:idx.numel()]
    if shard.world > 1:
        dist.all_reduce(out, group=group)
    return out


def _rows_index(shard, width, device):
    """Per rank: torch index tensor of the global rows it owns, in its local order (cached per (width, device))."""
    import torch
    cache = shard.__dict__.setdefault("_rows_index", {})
    key = (width, str(device))
    if key not in cache:
        cache[key] = [torch.from_numpy(shard.owned_rows_global(width, r)).to(device) for r in range(shard.world)]
    return cache[key]


def allgather_owned_rows(shard, local_rows, dist, width=3, out=None, group=None):
    """Same result as allreduce_owned_rows with half the traffic: the owned row slices are disjoint and contiguous in the
    global vector, so they are exchanged by ONE all-gather of slices padded to the largest one ((N-1)/N of the vector per
    rank over xGMI instead of 2(N-1)/N for the ring all-reduce of a zero-padded vector) and copied into place."""
    import torch
    n = width * shard.total_cp_global
    if out is None:
        out = torch.zeros(n, dtype=torch.float64, device=local_rows.device)
    idx = _rows_index(shard, width, local_rows.device)
    nown = idx[shard.rank].numel()
    if shard.world == 1:
        out[idx[0]] = local_rows[:nown]
        return out
    mx = max(i.numel() for i in idx)
    cache = shard.__dict__.setdefault("_exchange_buffers", {})            # staging buffers are allocated once per (width, device)
    key = (width, str(local_rows.device))
    if key not in cache:
        cache[key] = (torch.zeros(mx, dtype=torch.float64, device=local_rows.device),
                      torch.empty(shard.world * mx, dtype=torch.float64, device=local_rows.device))
    send, recv_buf = cache[key]
    send[:nown] = local_rows[:nown]                                        # the padding behind it stays zero
    recv = recv_buf
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(recv, send, group=group)
    else:                                                       # gloo (tests, rehearsal): host tensors
        parts = [torch.empty(mx, dtype=torch.float64) for _ in range(shard.world)]
        dist.all_gather(parts, send.cpu(), group=group)
        recv = torch.cat(parts).to(local_rows.device)
    # the owned patches of a rank need not be contiguous in the global numbering: one gather + one scatter for all ranks (two launches instead of two per rank: at
    # 8 ranks a step of the bench is ~3 ms, a dozen small launches are a few per cent of it)
    pk = ("place", width, str(local_rows.device))
    if pk not in cache:
        cache[pk] = (torch.cat(idx), torch.cat([r * mx + torch.arange(idx[r].numel(), device=local_rows.device) for r in range(shard.world)]))
    dst, src = cache[pk]
    out[dst] = recv[src]
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

    def __init__(self, spec, dist, rank, world, device=0, thickness_global=None, group=None):
        from . import _lib
        self._lib, self.dist, self.rank, self.world, self.group = _lib, dist, rank, world, group
        self.shard = shard_spec(spec, rank, world)
        self.A = shard_arrays(self.shard, thickness_global)
        self.D = _lib.DeviceModel(self.A, device=device)
        self.device = int(device)
        self.arrays = self.A
        self.cols_g = self.shard.local_cols_to_global()
        self.total_cp, self.ndof = self.shard.total_cp_global, 3 * self.shard.total_cp_global
        self.n_owned_cp = int(self.shard.cp_off_local[self.shard.n_owned])
        self._own_rows = {w: self.shard.owned_rows_global(w) for w in (1, 3)}
        self._gpat = {}
        self._kglob = None
        self._n_if_pts = [int(i.npts) for i in spec.interfaces]
        self._if_side_a = [int(i.a) for i in spec.interfaces]
        # Device-resident exchange (round 5): with a real DeviceModel the products, the residual and their exchange stay on the GPU -- the replicated input is
        # copied in ONCE, sliced to the local numbering by a device gather, multiplied by gf_apply_dev, the owned rows travel by ONE all-gather of device buffers
        # (allgather_owned_rows: RCCL sees device pointers; gloo, in the tests, stages inside the collective wrapper only) and the replicated result is copied
        # out ONCE.  The library's stream and torch's are ordered by events.  (The CPU stand-in of tests/test_distributed_cpu.py has no device: host path.)
        self._tdev = None
        if hasattr(self.D, "h") and hasattr(self.D, "stream_ptr"):
            import torch
            if torch.cuda.is_available():
                self._tdev = torch.device("cuda", self.device)
                self._lib_stream = torch.cuda.ExternalStream(self.D.stream_ptr, device=self._tdev)
                cg = torch.from_numpy(self.cols_g).to(self._tdev)
                self._loc_cp_g = cg                                                     # global control point of every local one (owned + ghost)
                self._loc_dof_g = (3 * cg[:, None] + torch.arange(3, device=self._tdev)).reshape(-1)

    def close(self):
        if getattr(self, "D", None) is not None:
            self.D.close()
        self._kglob = None

    def sync(self):
        self.D.sync()

    # ------------------------------------------------------------------ the DeviceModel surface with global-vector semantics (NonMatchingOpt(comm=...))
    def _allgather_concat(self, arr):
        """Concatenation over the ranks (rank order) of a 1-D array whose length differs per rank: lengths first, then one all-gather of padded arrays."""
        import torch
        arr = np.ascontiguousarray(arr)
        if self.world == 1:
            return arr
        cuda = self.dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", self.device) if cuda else torch.device("cpu")
        n = torch.tensor([arr.size], dtype=torch.int64, device=dev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        self.dist.all_gather(sizes, n, group=self.group)
        sizes = [int(x.item()) for x in sizes]
        self._last_concat_sizes = sizes
        mx = max(sizes)
        send = torch.zeros(mx, dtype=torch.from_numpy(arr[:0]).dtype, device=dev)
        send[:arr.size] = torch.from_numpy(arr).to(dev)
        parts = [torch.empty_like(send) for _ in range(self.world)]
        self.dist.all_gather(parts, send, group=self.group)
        return np.concatenate([p_[:k].cpu().numpy() for p_, k in zip(parts, sizes)])

    def _global_pattern(self, which):
        """(indptr, indices, perm, nnz_owned) of the GLOBAL matrix ``which``: the owned rows of every rank with their columns in global numbering, in CSR order
        with sorted columns -- the layout gf_pattern gives for the unsharded model; perm maps a CSR position to its place in the concatenation of the ranks'
        owned local values.  Collective, built once per pattern (K | dR/dCP | dR/dh)."""
        import scipy.sparse as sp
        _lib = self._lib
        kind = 0 if which == _lib.MAT_K else (2 if which == _lib.MAT_DRDH else 1)
        if kind not in self._gpat:
            rp, col = self.D.pattern(which)
            rp, col = np.asarray(rp, np.int64), np.asarray(col, np.int64)
            nrow = 3 * self.n_owned_cp
            nnz = int(rp[nrow])
            rows_l = np.repeat(np.arange(nrow), np.diff(rp[:nrow + 1]))
            bw = 3 if kind == 0 else 1
            grow = 3 * self.cols_g[rows_l // 3] + rows_l % 3
            gcol = bw * self.cols_g[col[:nnz] // bw] + col[:nnz] % bw
            R, Cc = self._allgather_concat(grow.astype(np.int64)), self._allgather_concat(gcol.astype(np.int64))
            if kind == 0:
                self._k_sizes = list(getattr(self, "_last_concat_sizes", [nnz]))      # owned K values per rank (the layout of the value all-gather)
            ncol = self.ndof if kind == 0 else self.total_cp
            Gm = sp.coo_matrix((np.arange(1, R.size + 1, dtype=np.float64), (R, Cc)), shape=(self.ndof, ncol)).tocsr()
            Gm.sort_indices()
            if Gm.nnz != R.size:
                raise RuntimeError("sharded pattern: a matrix entry is owned by two ranks")
            self._gpat[kind] = (Gm.indptr.astype(np.int64), Gm.indices.astype(np.int32), (Gm.data - 1.0).astype(np.int64), nnz)
        return self._gpat[kind]

    def pattern(self, which):
        ip, ix, _, _ = self._global_pattern(which)
        return ip, ix

    def cp_graph_global(self):
        """(nb_ptr, nb) of the GLOBAL control-point graph of K (lists ascending, the control point itself included) from the ranks' own lists: the owned rows of every
        rank in global numbers, one all-gather of two index arrays -- a ninth of the dof-level ``pattern(MAT_K)`` (11 s at C4), which the distributed solver's symbolic
        phase used to be built from.  Collective, built once."""
        import scipy.sparse as sp
        if getattr(self, "_cpg", None) is None:
            lptr, lnb = self.D.cp_graph()
            n_own = self.n_owned_cp
            rows_l = np.repeat(np.arange(n_own, dtype=np.int64), np.diff(lptr[:n_own + 1]))
            grow, gcol = self.cols_g[rows_l], self.cols_g[np.asarray(lnb[:lptr[n_own]], np.int64)]
            R, Cc = self._allgather_concat(grow.astype(np.int64)), self._allgather_concat(gcol.astype(np.int64))
            Gm = sp.csr_matrix((np.ones(R.size, np.int8), (R, Cc)), shape=(self.total_cp, self.total_cp))
            Gm.sort_indices()
            if Gm.nnz != R.size:
                raise RuntimeError("sharded control-point graph: a row is owned by two ranks")
            self._cpg = (Gm.indptr.astype(np.int64), Gm.indices.astype(np.int32))
        return self._cpg

    def values(self, which):
        """Values of the global matrix in the order of ``pattern`` (all ranks' owned rows: one all-gather of the owned values)."""
        _, _, perm, nnz = self._global_pattern(which)
        return self._allgather_concat(self.D.values(which)[:nnz])[perm]

    def csr(self, which):
        import scipy.sparse as sp
        ip, ix, perm, nnz = self._global_pattern(which)
        ncol = self.ndof if which == self._lib.MAT_K else self.total_cp
        return sp.csr_matrix((self._allgather_concat(self.D.values(which)[:nnz])[perm], ix, ip), shape=(self.ndof, ncol))

    # -- device-resident pieces
    def _dev_view(self, which_buf, n):
        import torch

        class _Buf:
            def __init__(self, p, k):
                self.__cuda_array_interface__ = {"shape": (k,), "typestr": "<f8", "data": (int(p), False), "version": 2}
        return torch.as_tensor(_Buf(self._lib.lib().gf_device_ptr(self.D.h, which_buf), n), device=self._tdev)

    def _to_dev(self, v):
        import torch
        return torch.from_numpy(np.ascontiguousarray(v, float)).to(self._tdev)

    def _apply_dev(self, which, transpose, x_dev, y_dev):
        """y_dev += A x_dev on the library's stream, ordered behind torch's work on x_dev / y_dev and in front of torch's next use of y_dev."""
        import ctypes as C
        import torch
        self._lib_stream.wait_stream(torch.cuda.current_stream(self._tdev))
        if self._lib.lib().gf_apply_dev(self.D.h, int(which), int(bool(transpose)), C.c_void_p(x_dev.data_ptr()), C.c_void_p(y_dev.data_ptr())):
            raise RuntimeError(self._lib.lib().gf_last_error().decode())
        torch.cuda.current_stream(self._tdev).wait_stream(self._lib_stream)

    def _allreduce_dev(self, t):
        """Sum over the ranks of a device tensor: in place over RCCL, through the host under gloo (tests)."""
        if self.world == 1:
            return t
        if self.dist.get_backend(self.group) == "nccl":
            self.dist.all_reduce(t, group=self.group)
            return t
        h = t.cpu()
        self.dist.all_reduce(h, group=self.group)
        return h.to(t.device)

    def apply_fwd_dev(self, which, xg):
        """A x for a replicated global device tensor ``xg``: the replicated global result as a device tensor (local product of the owned rows, ONE all-gather; nothing
        touches the host under RCCL).  What the distributed solver's refinement uses."""
        import torch
        _lib = self._lib
        with torch.cuda.device(self._tdev):
            y = torch.zeros(self.A.ndof, dtype=torch.float64, device=self._tdev)
            self._apply_dev(which, False, xg[self._loc_dof_g if which == _lib.MAT_K else self._loc_cp_g].contiguous(), y)
            return allgather_owned_rows(self.shard, y, self.dist, 3, group=self.group).clone()

    def apply_many(self, which, xs, ys, transpose=False):
        """DeviceModel.apply_many with replicated global vectors: transpose=False: ys[0] += sum_m A_m xs[m]; transpose=True: ys[m] += A_m^T xs[0].
        ONE collective per call (DispImOpeartion.apply_linear_fwd / _rev, disp_imop.py:58-128): forward, the local products are summed on the device and the
        owned rows all-gathered once; reverse, the m results share one all-reduce."""
        if self._tdev is None:
            for m, w in enumerate(which):
                if transpose:
                    ys[m][:] += self.apply(w, xs[0], transpose=True)
                else:
                    ys[0][:] += self.apply(w, xs[m])
            return ys
        import torch
        _lib = self._lib
        with torch.cuda.device(self._tdev):
            if not transpose:
                y = torch.zeros(self.A.ndof, dtype=torch.float64, device=self._tdev)
                for m, w in enumerate(which):
                    xg = self._to_dev(xs[m])
                    self._apply_dev(w, False, xg[self._loc_dof_g if w == _lib.MAT_K else self._loc_cp_g].contiguous(), y)
                ys[0][:] += allgather_owned_rows(self.shard, y, self.dist, 3, group=self.group).cpu().numpy()
                return ys
            xl = self._to_dev(xs[0])[self._loc_dof_g].contiguous()
            xl[3 * self.n_owned_cp:] = 0.0                                  # ghost rows are not assembled here
            sizes = [self.ndof if w == _lib.MAT_K else self.total_cp for w in which]
            out = torch.zeros(int(sum(sizes)), dtype=torch.float64, device=self._tdev)
            off = 0
            for w, n in zip(which, sizes):
                y = torch.zeros(self.A.ndof if w == _lib.MAT_K else self.A.total_cp, dtype=torch.float64, device=self._tdev)
                self._apply_dev(w, True, xl, y)
                out[off:off + n].index_add_(0, self._loc_dof_g if w == _lib.MAT_K else self._loc_cp_g, y)      # local columns (owned + ghost) -> global ids
                off += n
            out = self._allreduce_dev(out).cpu().numpy()
            off = 0
            for m, n in enumerate(sizes):
                ys[m][:] += out[off:off + n]
                off += n
            return ys

    def compliance(self, forces, apply_bcs=True):
        """Global compliance functional (gf_compliance): owned patches per rank, value and owned gradient rows summed over the ranks."""
        order = np.asarray(self.shard.order)
        f = np.asarray(forces, float).reshape(-1, 3)[order]
        F = self.D.compliance(f, apply_bcs=apply_bcs)
        n = self.n_owned_cp
        du, dcp = np.array(F["dCdu"], float), np.array(F["dCdcp"], float)
        res = self._allreduce_packed([np.array([F["C"]]), self._own_rows_global(du, 3)] + [self._own_rows_global(dcp[k], 1) for k in range(3)])      # one all-reduce
        return dict(C=float(res[0][0]), dCdu=res[1], dCdcp=np.stack(res[2:5]))

    def update_interface(self, g, itf):
        """DeviceModel.update_interface for GLOBAL interface g: patched on the ranks that hold it; the model is re-created everywhere when ANY rank reports a vertex
        that crossed a knot line (one all-reduce of a flag)."""
        ok = 1.0
        if g in self.shard.if_global:
            loc = self.shard.spec.interfaces[self.shard.if_global.index(g)]
            from .model import Interface
            ok = 1.0 if self.D.update_interface(self.shard.if_global.index(g), Interface(loc.a, loc.b, itf.xi_a, itf.xi_b)) else 0.0
        return bool(self._allreduce(np.array([1.0 - ok]))[0] == 0.0)

    def penalty_dxi_if(self, g, degree):
        """(blocks (n, 6, 2, nb, 3), windows (n, 2, 2)) of the mortar vertices of GLOBAL interface g, replicated: the rank that owns the interface's side A evaluates
        them on its device (both patches are there, one possibly as a ghost: the per-vertex blocks need geometry and state only) and everybody receives them -- one
        all-reduce to which the other ranks contribute zeros (58 k doubles for a 100-vertex bicubic interface).  The matrix form of dR/dxi on a sharded problem
        (GOLDFISH/nonmatching_opt.py:1042-1341; the reference marks this path "doesn't work in parallel")."""
        n, nb = int(self._n_if_pts[g]), (int(degree) + 1) ** 2
        a_global = self._if_side_a[g]
        owner = next(r for r, own in enumerate(self.shard.owned_by_rank) if a_global in own)
        B, W = np.zeros((n, 6, 2, nb, 3)), np.zeros((n, 2, 2))
        if owner == self.rank:
            itf_l = self.shard.if_global.index(g)
            Bl, Wl = self.D.penalty_dxi(n, degree, v_first=int(self.A.if_off[itf_l]))
            B, W = np.asarray(Bl, float), np.asarray(Wl, float)
        out = self._allreduce_packed([B, W]) if self.world > 1 else [B, W]
        return out[0], np.rint(out[1]).astype(np.int32)

    def penalty_dxi_rev_if(self, g, lam):
        """(n, 6) reverse-mode product of the dR/d(xi, tau) blocks of the mortar vertices of GLOBAL interface g with the replicated lam (moving intersections on
        shards): every rank that holds the interface contracts the rows of its OWN patches on its device (gf_penalty_dxi_rev masks the ghost
        rows), the ranks' results add up -- one all-reduce of 6 doubles per mortar vertex."""
        itf_l = self.shard.if_global.index(g) if g in self.shard.if_global else None
        n = int(self._n_if_pts[g])
        out = np.zeros((n, 6))
        if itf_l is not None:
            off = self.A.if_off
            out = self.D.penalty_dxi_rev(n, self.shard.to_local(np.asarray(lam, float), 3), v_first=int(off[itf_l]))
        return self._allreduce(out.ravel()).reshape(n, 6)

    # -- replicated global K on this rank's device: what the direct solver factors (stage 1 of the sharded solve: every rank gathers the owned value rows
    #    of all ranks and factors the same matrix -- correct and redundant; the assembly is sharded, the factorisation is not)
    def k_values_ptr(self):
        import torch
        if self._kglob is None:                               # the buffer the solver borrows; filled by refresh_k_values before every factorisation
            self._kglob = torch.zeros(self._global_pattern(self._lib.MAT_K)[1].size, dtype=torch.float64, device=torch.device("cuda", self.device))
        return int(self._kglob.data_ptr())

    def refresh_k_values(self):
        """The replicated K of the direct solvers (stage 1 / the distributed factorisation's value source): ONE all-gather of the owned value rows as device buffers
        and one device gather into the global CSR order.  (Round 4 went device -> host -> list all-gather -> host permutation of 3.5e8 doubles -> device: 3.2 s at C4
        with two ranks; under gloo the collective itself is still staged through the host, the permutation is not.)"""
        import torch
        self.k_values_ptr()
        if self._tdev is None or self.world == 1:
            self._kglob.copy_(torch.from_numpy(self.values(self._lib.MAT_K)))
            torch.cuda.synchronize(self.device)
            return
        _, _, perm, nnz = self._global_pattern(self._lib.MAT_K)
        sizes = self._k_sizes
        mx = max(sizes)
        with torch.cuda.device(self._tdev):
            if getattr(self, "_k_place", None) is None:                    # CSR position -> place in the padded all-gather buffer [rank][mx]
                cum = np.concatenate([[0], np.cumsum(sizes)])
                r = np.searchsorted(cum, perm, side="right") - 1
                place = r * mx + (perm - cum[r])
                self._k_place = torch.from_numpy(place.astype(np.int32 if self.world * mx < 2 ** 31 else np.int64)).to(self._tdev)
                self._k_recv = torch.empty(self.world * mx, dtype=torch.float64, device=self._tdev)
                self._k_send = torch.zeros(mx, dtype=torch.float64, device=self._tdev)
            torch.cuda.current_stream(self._tdev).wait_stream(self._lib_stream)
            self._k_send[:nnz] = self._dev_view(self._lib.BUF_VAL_K, nnz)
            if self.dist.get_backend(self.group) == "nccl":
                self.dist.all_gather_into_tensor(self._k_recv, self._k_send, group=self.group)
            else:
                parts = [torch.empty(mx, dtype=torch.float64) for _ in range(self.world)]
                self.dist.all_gather(parts, self._k_send.cpu(), group=self.group)
                self._k_recv.copy_(torch.cat(parts))
            torch.index_select(self._k_recv, 0, self._k_place, out=self._kglob)
            torch.cuda.synchronize(self.device)

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
        if self.dist.get_backend(self.group) == "nccl":
            t = t.to(torch.device("cuda", self.device))
        self.dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()

    def _rows_to_global(self, local_rows, width):
        out = np.zeros(width * self.total_cp)
        n = width * self.n_owned_cp
        out[self._own_rows[width]] = local_rows[:n]
        return self._allreduce(out)

    def _cols_to_global(self, local_cols):
        out = np.zeros(self.total_cp)
        np.add.at(out, self.cols_g, local_cols)
        return self._allreduce(out)

    def residual(self):
        if self._tdev is None:
            return self._rows_to_global(self.D.residual(), 3)
        import torch
        with torch.cuda.device(self._tdev):
            torch.cuda.current_stream(self._tdev).wait_stream(self._lib_stream)         # the assembly that wrote the residual runs on the library's stream
            R = allgather_owned_rows(self.shard, self._dev_view(self._lib.BUF_R, self.A.ndof), self.dist, 3, group=self.group)
            return R.cpu().numpy()

    def apply(self, which, x, y=None, transpose=False):
        """A x (or A^T x) for the replicated global x as a replicated global vector; with ``y`` (DeviceModel.apply's signature) it is added to y in place."""
        if y is not None and not isinstance(y, (bool, np.bool_)):
            y[:] += self.apply(which, x, transpose=transpose)
            return y
        _lib = self._lib
        x = np.asarray(x, float)
        if self._tdev is not None:                              # device-resident: one copy in, one collective, one copy out
            out = [np.zeros(self.ndof if (which == _lib.MAT_K or not transpose) else self.total_cp)]
            self.apply_many([which], [x], out, transpose=transpose)
            return out[0]
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

    def _allreduce_packed(self, parts):
        """ONE all-reduce for several arrays (scalars, per-patch terms, gradient rows placed at their global ids): returns the summed arrays in the same shapes."""
        flat = np.concatenate([np.ravel(p_) for p_ in parts])
        flat = self._allreduce(flat)
        out, off = [], 0
        for p_ in parts:
            n = int(np.size(p_))
            out.append(flat[off:off + n].reshape(np.shape(p_)))
            off += n
        return out

    def _own_rows_global(self, v, width):
        """The owned rows of a local gradient field at their global ids, zeros elsewhere (the ranks' contributions add up)."""
        out = np.zeros(width * self.total_cp)
        n = width * self.n_owned_cp
        out[self._own_rows[width]] = np.asarray(v, float)[:n]              # gradients are assembled for owned control points only
        return out

    def functionals(self, apply_bcs=True):
        """Global W_int, volume, penalty energy and their gradient fields: owned elements per rank; the scalars, the per-patch terms and the eleven gradient rows
        per control point travel in ONE all-reduce (round 4: nine)."""
        F = self.D.functionals(apply_bcs=apply_bcs)
        parts = [np.array([F["Wint"], F["volume"], F["Wpen"]])]
        has_pp = "volume_patch" in F
        if has_pp:                                           # per-patch terms (VolumeExOperation(vol_surf_inds = subset)): the owner reports its patches
            order, no, npg = np.asarray(self.shard.order), self.shard.n_owned, len(self.shard.cp_off_global) - 1
            pp = np.zeros((2, npg))
            pp[0, order[:no]], pp[1, order[:no]] = F["Wint_patch"][:no], F["volume_patch"][:no]
            parts.append(pp)
        parts += [self._own_rows_global(F["dWdu"], 3), self._own_rows_global(F["dWdh"], 1), self._own_rows_global(F["dVdh"], 1)]
        parts += [self._own_rows_global(F["dWdcp"][f], 1) for f in range(3)] + [self._own_rows_global(F["dVdcp"][f], 1) for f in range(3)]
        res = self._allreduce_packed(parts)
        sc = res.pop(0)
        out = dict(Wint=sc[0], volume=sc[1], Wpen=sc[2])
        if has_pp:
            pp = res.pop(0)
            out["Wint_patch"], out["volume_patch"] = pp[0], pp[1]
        out["dWdu"], out["dWdh"], out["dVdh"] = res[0], res[1], res[2]
        out["dWdcp"], out["dVdcp"] = [res[3], res[4], res[5]], [res[6], res[7], res[8]]
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
        parts = [np.concatenate([I, vmax])]
        if gradients:
            parts += [self._own_rows_global(F["dIdu"], 3), self._own_rows_global(F["dIdh"], 1)] + [self._own_rows_global(F["dIdcp"][f], 1) for f in range(3)]
        res = self._allreduce_packed(parts)                                  # one all-reduce for the per-patch values and the owned gradient rows
        out = dict(I=res[0][:npg], vmax=res[0][npg:])
        if gradients:
            out["dIdu"], out["dIdh"], out["dIdcp"] = res[1], res[2], np.stack(res[3:6])
        return out

    def shape_regu(self, field, cp0_global, coef_global):
        """Global shape regularisation term (gf_shape_regu): owned elements per rank, value and owned gradient rows summed."""
        order = np.asarray(self.shard.order)
        F = self.D.shape_regu(field, self.shard.to_local(np.asarray(cp0_global, float)), np.asarray(coef_global, float)[order])
        res = self._allreduce_packed([np.array([F["value"]])] + [self._own_rows_global(F["dcp"][f], 1) for f in range(3)])
        return dict(value=float(res[0][0]), dcp=np.stack(res[1:4]))
