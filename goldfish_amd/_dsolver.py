"""Direct solves with K over several GPUs (SURVEY.md 8(e) + 8(f) N1, stage 2): one process per GPU, an elimination tree that follows the patch partition.

Replaces, on a sharded problem, the replicated factorisation of stage 1 (every rank factors all of K: goldfish_amd/_solver.py with the gathered values) -- the
reference's counterpart is MUMPS on ``comm`` (GOLDFISH/utils/opt_utils.py:156-209), which replicates neither the matrix nor the root fronts.  The symbolic phase
(partition_tree below; deterministic, the same on every rank): the control points on rank boundaries S (the lower-rank end of every edge between two ranks' control
points) are eliminated along the hierarchy of the rank bisection -- one SEPARATOR front per node --, everything else by its owner's own nested dissection.  Then

  1. every rank factors ITS subtrees (handle A: gfs_create_nd_partial in the rank's LOCAL numbering, on the K that rank assembled; a block whose later control point is
     a ghost row is read transposed from the owned row: gfs_set_row_mask; the subtree roots keep their Schur complements),
  2. the rows of S (a few per cent of K) are gathered once per factorisation -- the only K values that travel --,
  3. the separator fronts are factored level by level from the deepest, every front by ONE rank (the lowest rank below it; its own partial handle: the front above STUB
     fronts that carry its children's Schur complements), the fronts of a level side by side; after every level one all-gather of the new Schur complements,

and a solve is: forward sweep of the own subtrees, then of the separator fronts level by level upwards (the boundary contributions, 3 doubles per boundary control
point, all-gathered per level); backward sweep down the levels (the x of a level summed over the ranks into the replicated x of S), then of the own subtrees; one
all-reduce of the pieces of x.  Iterative refinement runs against K itself with the sharded product (device-resident).  Every outcome -- handle creation, every
level of the factorisation, every substitution -- is agreed on by all ranks before the next collective (_raise_together).

Distributed: the subtree factorisations and sweeps, the separator fronts (one rank each: the critical path of the top is one front per level), K.  Replicated: the
rows of S, the x of S, the refinement residuals.  split_tree (round 4's scheme: subtrees of the free dissection dealt by work, K and the top replicated) is kept for
tools/dist_solver_model.py, which prices both.

K must be symmetric (the general mode of the single-GPU solver is not distributed).  ``dist`` is torch.distributed ('nccl' == RCCL on a multi-GPU node; 'gloo' in
the tests, where the buffers travel through the host)."""
import ctypes as C

import numpy as np

from . import _nd
from ._solver import lib, control_point_graph, parent_positions

_i64p = C.POINTER(C.c_int64)


def _bind():
    L = lib()
    if getattr(L, "_gf_partial_bound", False):
        return L
    i32p, vp = C.POINTER(C.c_int32), C.c_void_p
    L.gfs_create_nd_partial.argtypes = [C.c_int, C.c_int64, _i64p, i32p, vp, C.c_int64] + [_i64p] * 8 + [C.POINTER(vp)]
    L.gfs_schur_doubles.argtypes = [vp, C.c_int64]
    L.gfs_schur_doubles.restype = C.c_int64
    for name in ("gfs_export_schur", "gfs_set_schur_source", "gfs_get_fbnd", "gfs_set_fbnd"):
        getattr(L, name).argtypes = [vp, C.c_int64, vp]
    for name in ("gfs_get_fbnd_packed", "gfs_set_fbnd_packed"):
        getattr(L, name).argtypes = [vp, C.c_int64, _i64p, vp]
    L.gfs_set_row_mask.argtypes = [vp, vp]
    L.gfs_x_ptr.argtypes = [vp]
    L.gfs_x_ptr.restype = vp
    L.gfs_forward_dev.argtypes = [vp, vp]
    L.gfs_backward_dev.argtypes = [vp]
    L._gf_partial_bound = True
    return L


def front_work(sym, nb=64):
    """Tile products of the partial factorisation of every front (the measure the subtrees are balanced by)."""
    _, _, be, bb = sym.front_dofs(nb)
    bt = be + bb
    w = np.zeros(sym.nfronts)
    for t in range(sym.nfronts):
        r = bt[t] - 1 - np.arange(be[t])
        w[t] = float(np.sum(1.0 + r + r * (r + 1) / 2.0))
    return w


def split_tree(sym, world):
    """(owner, roots): owner[t] = rank that factors front t, -1 for the replicated top; roots = the subtree roots in post-order.  The tree is opened from the root
    (always the subtree with the most work next); after every opening with at least ``world`` subtrees these are dealt to the ranks largest first, and the split with
    the smallest (largest rank's share + replicated top) wins: opening further balances the ranks but moves work into the top that every rank repeats."""
    nf = sym.nfronts
    work = front_work(sym)
    kids = [[] for _ in range(nf)]
    for t in range(nf):
        if sym.parent[t] >= 0:
            kids[sym.parent[t]].append(t)
    sub = work.copy()
    for t in range(nf):                                  # post-order: children first
        if sym.parent[t] >= 0:
            sub[sym.parent[t]] += sub[t]

    def deal(cand):
        load, who = np.zeros(world), {}
        for t in sorted(cand, key=lambda q: (-sub[q], q)):
            r = int(np.argmin(load))
            load[r] += sub[t]
            who[t] = r
        return load.max(), who

    cand = [t for t in range(nf) if sym.parent[t] < 0]
    top, top_work, best = [], 0.0, None
    while True:
        if len(cand) >= world:
            cost, who = deal(cand)
            if best is None or cost + top_work < best[0]:
                best = (cost + top_work, list(top), dict(who))
        open_ = [t for t in cand if kids[t]]
        if not open_ or len(cand) >= 8 * world:
            break
        t = max(open_, key=lambda q: (sub[q], -q))
        cand.remove(t)
        top.append(t)
        top_work += work[t]
        cand.extend(kids[t])
    if best is None:                                     # fewer subtrees than ranks even when fully opened: some ranks only hold the top
        best = (0.0, list(top), deal(cand)[1])
    _, top, who = best
    owner = np.full(nf, -2, np.int64)
    owner[top] = -1
    for t, r in who.items():                             # the whole subtree of t: post-order puts it in the contiguous range that ends at t
        lo, stack = t, [t]
        while stack:
            q = stack.pop()
            lo = min(lo, q)
            stack.extend(kids[q])
        owner[lo:t + 1] = r
    assert (owner >= -1).all()
    return owner, sorted(who)


def partial_symbolic(sym, keep, stub_roots=(), later_cp=None):
    """The arguments of gfs_create_nd_partial for the fronts ``keep`` (ascending = post-order) of the global tree ``sym`` plus stubs for ``stub_roots`` (fronts of other
    handles whose Schur complement comes in from outside): (fronts, Symbolic with LOCAL front numbers and elimination order, pmap)."""
    ncp = sym.order.size
    keep = np.asarray(keep, np.int64)
    stub_roots = np.asarray(stub_roots, np.int64)
    # local front numbering: post-order of the global tree restricted to keep + stubs (a stub precedes its parent)
    fronts = np.sort(np.concatenate([keep, stub_roots]))
    is_stub = np.isin(fronts, stub_roots)
    local = {int(t): i for i, t in enumerate(fronts)}
    elim_parts, bnd_parts, parent = [], [], np.full(fronts.size, -1, np.int64)
    for i, t in enumerate(fronts):
        elim_parts.append(np.zeros(0, np.int64) if is_stub[i] else sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]])
        bnd_parts.append(sym.bnd[sym.bnd_off[t]:sym.bnd_off[t + 1]])
        parent[i] = local.get(int(sym.parent[t]), -1)
    elim = np.concatenate(elim_parts) if elim_parts else np.zeros(0, np.int64)
    elim_off = np.concatenate([[0], np.cumsum([e.size for e in elim_parts])]).astype(np.int64)
    bnd = np.concatenate(bnd_parts) if bnd_parts else np.zeros(0, np.int64)
    bnd_off = np.concatenate([[0], np.cumsum([b.size for b in bnd_parts])]).astype(np.int64)
    front_of = np.full(ncp, -1, np.int64)
    front_of[elim] = np.repeat(np.arange(fronts.size), np.diff(elim_off))
    # local elimination order: own control points by their place in elim (a monotone restriction of the global order).  Control points of other handles keep their
    # global order, shifted BEHIND the own ones when this handle's fronts can have them on their boundaries (``later_cp``: the control points of the top of the
    # tree, for a handle of subtrees -- an ancestor's separator is eliminated after everything below it) and in front of them (negative) otherwise: an entry of K
    # belongs to the front of the control point eliminated first, and gfs_refactor skips the entries whose first control point is not this handle's.
    # The boundary lists are sorted by the global order; the local order keeps that order among the control points that can appear in them.
    order = sym.order - ncp - 1                                  # < 0
    if later_cp is not None:
        order[later_cp] = elim.size + sym.order[later_cp]
    order[elim] = np.arange(elim.size)
    sub = _nd.Symbolic(elim, elim_off, bnd, bnd_off, parent, order, front_of)
    return fronts, sub, parent_positions(sub)


class _Part:
    """One partial handle of libgoldfish_solver (gfs_create_nd_partial) over partial_symbolic(...), created in its OWN numbering of the control points:
    ``to_own`` maps a global control point to that numbering (-1: not there), ``own_ids`` the other way; nb_ptr / nb: the control-point graph in that numbering;
    dK: the K values in the layout of that graph; ``row_ok``: rows that hold values (None: all)."""

    def __init__(self, L, sym, keep, stub_roots, later_cp, own_ids, nb_ptr, nb, dK, device, row_ok=None):
        fronts, sub, pmap = partial_symbolic(sym, keep, stub_roots, later_cp)
        self.fronts, self.local = fronts, {int(t): i for i, t in enumerate(fronts)}
        self.nbnd = np.diff(sub.bnd_off)
        own_ids = np.asarray(own_ids, np.int64)
        to_own = np.full(sym.order.size, -1, np.int64)
        to_own[own_ids] = np.arange(own_ids.size)
        elim, bnd = to_own[sub.elim], to_own[sub.bnd]
        if (elim < 0).any() or (bnd < 0).any():
            raise RuntimeError("DistributedSolver: a control point of this handle's fronts is not in its numbering (the partition's ghosts do not cover a boundary)")
        order, front_of = sub.order[own_ids], sub.front_of[own_ids]
        self.nb_ptr, self.nb = np.ascontiguousarray(nb_ptr, np.int64), np.ascontiguousarray(nb, np.int32)
        self._keep = [np.ascontiguousarray(a, np.int64) for a in (elim, sub.elim_off, bnd, sub.bnd_off, sub.parent, order, front_of, pmap)]
        self.ncp = int(own_ids.size)
        h = C.c_void_p()
        i64 = lambda a: a.ctypes.data_as(_i64p)
        rc = L.gfs_create_nd_partial(int(device), self.ncp, self.nb_ptr.ctypes.data_as(_i64p), self.nb.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(dK), fronts.size,
                                     *[i64(a) for a in self._keep], C.byref(h))
        if rc:
            raise RuntimeError(L.gfs_last_error().decode())
        self.h = h
        if row_ok is not None:
            self._mask = np.ascontiguousarray(row_ok, np.uint8)
            if L.gfs_set_row_mask(h, self._mask.ctypes.data_as(C.c_void_p)):
                raise RuntimeError(L.gfs_last_error().decode())


class DistributedSolver:
    """K x = b over the ranks of ``dist`` (see the module docstring).  ``dev_model``: the sharded device model (goldfish_amd/sharding.py: ShardedDeviceModel): its
    ``pattern`` gives the GLOBAL control-point graph (symbolic phase only), its local device model (``dev_model.D``) the K this rank assembled, which the rank's
    handle reads in place."""

    method = "nd-distributed"

    def __init__(self, dev_model, dist, group=None, coords=None, leaf=128, max_refine=3):
        import torch
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.D, self.max_refine = dev_model, max_refine
        self.device = int(dev_model.device)
        self.cuda = dist.get_backend(group) == "nccl"
        L = self.L = _bind()
        import os, time
        _t = [time.perf_counter()]
        _lap = lambda: (_t.append(time.perf_counter()), _t[-1] - _t[-2])[1]
        if hasattr(dev_model, "cp_graph_global"):              # the control-point-level graph from the ranks' own lists (a ninth of the dof-level pattern)
            self.nb_ptr, self.nb = dev_model.cp_graph_global()
        else:
            rowptr, col = dev_model.pattern(0)
            self.nb_ptr, self.nb = control_point_graph(rowptr, col)
            del rowptr, col
        self.setup_timings = {"global_pattern": _lap()}
        ncp = self.ncp = self.nb_ptr.size - 1
        self.n = 3 * ncp
        if coords is None:
            raise ValueError("DistributedSolver needs the control points' coordinates")
        # owner rank of every control point (the patch partition), then the elimination tree that follows it: deterministic, the same on every rank
        sh = dev_model.shard
        owner_cp = np.empty(ncp, np.int64)
        for r, own in enumerate(sh.owned_by_rank):
            for g in own:
                owner_cp[sh.cp_off_global[g]:sh.cp_off_global[g + 1]] = r
        self.owner_cp = owner_cp
        sym, self.owner, self.roots = partition_tree(self.nb_ptr, self.nb, coords, owner_cp, self.world, leaf=leaf)
        self.sym = sym
        self.setup_timings["partition_tree"] = _lap()
        self.A = self.B = None
        self.rel_residual = self.backward_error = None
        self.rel_residuals = None
        self.small_pivot = False
        # Everything from here to the first collective of refactor() can fail on ONE rank only (a rank's fronts are checked against that rank's free memory): the
        # outcome is agreed on before anybody enters a collective, and every rank raises the same kind of error (a rank that fell back to the host
        # solve alone met the others inside an all-gather)
        err = None
        try:
            self._create(dev_model, sym)
        except (RuntimeError, MemoryError) as e:
            err = str(e) or type(e).__name__
        self._raise_together(err, "DistributedSolver")
        self.setup_timings["handles"] = _lap()
        self.refactor()
        self.setup_timings["first_factorisation"] = _lap()
        if os.environ.get("GF_DSOLVER_TIMING") == "1" and self.rank == 0:
            print("DistributedSolver set-up (s): " + ", ".join("%s %.2f" % kv for kv in self.setup_timings.items()), flush=True)

    _PERMANENT = ("device memory", "out of memory", "does not fit", "not symmetric", "hipMalloc", "OutOfMemory")

    def _raise_together(self, err, where):
        """Collective: every rank passes its local error text (or None); if any rank failed, ALL raise a RuntimeError whose text says whether the failure is permanent
        (memory: the text contains "out of memory", which NonMatchingOpt.solve_K latches on) -- the same decision on every rank."""
        code = 0 if err is None else (2 if any(k in err for k in self._PERMANENT) else 1)
        worst = self._agree(code)
        if worst == 0:
            return
        self.close()
        mine = (": " + err) if err is not None else ""
        if worst == 2:
            raise RuntimeError("%s: out of memory on at least one rank (rank %d%s)" % (where, self.rank, mine))
        raise RuntimeError("%s: failed on at least one rank (rank %d%s)" % (where, self.rank, mine))

    def _create(self, dev_model, sym):
        torch, L = self.torch, self.L
        ncp = self.ncp
        dev = torch.device("cuda", self.device)
        mine = np.flatnonzero(self.owner == self.rank)
        self.my_roots = [t for t in self.roots if self.owner[t] == self.rank]
        top_f = np.flatnonzero(self.owner == -1)
        top_cp = np.concatenate([sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]] for t in top_f]) if top_f.size else np.zeros(0, np.int64)
        # ---- handle A: this rank's subtrees in its LOCAL numbering, on the K it assembled (no gather)
        loc = dev_model.D                                        # the rank's own DeviceModel
        cols_g = np.asarray(dev_model.cols_g, np.int64)          # local control point -> global
        if mine.size:
            lptr, lnb = loc.cp_graph()
            row_ok = np.arange(cols_g.size) < int(dev_model.n_owned_cp)
            self.A = _Part(L, sym, mine, (), top_cp, cols_g, lptr, lnb, loc.k_values_ptr(), self.device, row_ok=row_ok)
        # ---- the separator fronts: ONE RANK EACH (tree-parallel top, round 5).  The fronts of one level of the rank hierarchy are independent, so they are factored side by
        #      side by different ranks (front t by the lowest rank below it) instead of all of them by every rank: the critical path of the top is one front per level
        #      (C4, 8 ranks: 1.84 of 4.07 Tflop; tools/dist_solver_model.py).  Every such front is its own partial handle (the front + stubs for its children) in the
        #      numbering of S = the separators' control points (ascending global ids), on the gathered rows of S (a few per cent of K, replicated).
        S = np.sort(top_cp)
        self.S = S
        s_index = np.full(ncp, -1, np.int64)
        s_index[S] = np.arange(S.size)
        sptr, snb = _induced(self.nb_ptr, np.asarray(self.nb, np.int64), S, ncp) if S.size else (np.zeros(1, np.int64), np.zeros(0, np.int32))
        self._valB = torch.zeros(max(1, 9 * int(sptr[-1])), dtype=torch.float64, device=dev)
        self._build_top_value_exchange(dev_model, S, s_index, sptr, snb, cols_g, dev)
        nf = sym.nfronts
        kids = [[] for _ in range(nf)]
        for t in range(nf):
            if sym.parent[t] >= 0:
                kids[sym.parent[t]].append(t)
        depth = np.zeros(nf, np.int64)
        for t in range(nf - 1, -1, -1):
            if sym.parent[t] >= 0:
                depth[t] = depth[sym.parent[t]] + 1
        low = np.full(nf, self.world, np.int64)                     # lowest rank with a subtree below every front
        for t in range(nf):
            if self.owner[t] >= 0:
                low[t] = self.owner[t]
            for c in kids[t]:
                low[t] = min(low[t], low[c])
        maxd = int(depth[top_f].max()) if top_f.size else -1
        # stages of the exchange: 0 = the ranks' subtree roots, s >= 1 = the separator fronts at depth maxd + 1 - s (deepest first, the root last)
        self.stage_of = {int(t): 0 for t in self.roots}
        self.prod_of = {int(t): int(self.owner[t]) for t in self.roots}
        for t in top_f:
            self.stage_of[int(t)] = maxd - int(depth[t]) + 1
            self.prod_of[int(t)] = int(low[t]) if low[t] < self.world else 0
        self.nstage = maxd + 2 if top_f.size else 1
        _, _, be_, bb_ = sym.front_dofs()
        self.schur_len = {t: int(bb_[t] * (bb_[t] + 1) // 2) * 4096 for t in self.stage_of}
        self.fb_len = {t: 3 * int(sym.bnd_off[t + 1] - sym.bnd_off[t]) for t in self.stage_of}
        self.stage_fronts = [[[t for t in sorted(self.stage_of) if self.stage_of[t] == st and self.prod_of[t] == r] for r in range(self.world)] for st in range(self.nstage)]
        self.schur_pad = [max(1, max(sum(self.schur_len[t] for t in ts) for ts in per)) for per in self.stage_fronts]
        self.fb_pad = [max(1, max(sum(self.fb_len[t] for t in ts) for ts in per)) for per in self.stage_fronts]
        needs_up = [any(sym.parent[t] >= 0 for per in [self.stage_fronts[st]] for ts in per for t in ts) for st in range(self.nstage)]
        self.stage_exchanges = needs_up                              # a stage whose fronts have no parents (the root) sends nothing up
        self.schur_buf = [torch.zeros(self.world * self.schur_pad[st] if needs_up[st] else 1, dtype=torch.float64, device=dev) for st in range(self.nstage)]
        self.fb_buf = [torch.zeros(self.world * self.fb_pad[st] if needs_up[st] else 1, dtype=torch.float64, device=dev) for st in range(self.nstage)]
        self.schur_off, self.fb_off = {}, {}                         # front -> offset (doubles) of its slot in its stage's buffer
        for st in range(self.nstage):
            for r, ts in enumerate(self.stage_fronts[st]):
                o1, o2 = r * self.schur_pad[st], r * self.fb_pad[st]
                for t in ts:
                    self.schur_off[t], self.fb_off[t] = o1, o2
                    o1 += self.schur_len[t]; o2 += self.fb_len[t]
        # this rank's separator fronts: handle = the front + stubs for its children, whose Schur complements it reads from the stage buffers
        self.T = {}                                                  # front -> _Part
        self.top_kids = {int(t): list(kids[t]) for t in top_f}
        self.top_elim_s = {}
        for t in top_f:
            t = int(t)
            if self.prod_of[t] != self.rank:
                continue
            anc, q = [], sym.parent[t]
            while q >= 0:
                anc.append(q); q = sym.parent[q]
            later = np.concatenate([sym.elim[sym.elim_off[a]:sym.elim_off[a + 1]] for a in anc]) if anc else None
            part = _Part(L, sym, [t], kids[t], later, S, sptr, snb, int(self._valB.data_ptr()), self.device)
            for c in kids[t]:
                stc = self.stage_of[c]
                if L.gfs_set_schur_source(part.h, part.local[c], C.c_void_p(self.schur_buf[stc].data_ptr() + 8 * self.schur_off[c])):
                    raise RuntimeError(L.gfs_last_error().decode())
            part.x = self._view(L.gfs_x_ptr(part.h), 3 * part.ncp)
            self.T[t] = part
            self.top_elim_s[t] = torch.from_numpy((3 * s_index[sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]]][:, None] + np.arange(3)).ravel()).to(dev)
        self.B = None
        self.my_top = sorted(self.T, key=lambda t: self.stage_of[t])
        self._rootsA = np.ascontiguousarray([self.A.local[t] for t in self.my_roots], np.int64) if self.A else np.zeros(0, np.int64)

        # ---- index sets between the replicated global vectors and the handles' numberings
        def dofs(cps):
            return (3 * np.asarray(cps, np.int64)[:, None] + np.arange(3)).ravel()
        t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int64)).to(dev)
        self.loc_dof_g = t64(dofs(cols_g))                       # global dof of every local dof (right-hand side of A)
        own_cp = np.concatenate([sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]] for t in mine]) if mine.size else np.zeros(0, np.int64)
        g2l = np.full(ncp, -1, np.int64)
        g2l[cols_g] = np.arange(cols_g.size)
        self.own_dofs_g, self.own_dofs_l = t64(dofs(own_cp)), t64(dofs(g2l[own_cp]))
        self.S_dofs_g = t64(dofs(S))
        here = np.flatnonzero(s_index[cols_g] >= 0)              # local control points that lie in S: the top's x at the boundaries of the own subtrees
        self.S_here_l, self.S_here_s = t64(dofs(here)), t64(dofs(s_index[cols_g[here]]))
        self.xA = self._view(L.gfs_x_ptr(self.A.h), 3 * self.A.ncp) if self.A else None
        self.b_dev = torch.zeros(self.n, dtype=torch.float64, device=dev)
        # |K|_F from the owned rows of all ranks (the backward error's norm)
        nnz_owned = 9 * int(loc.cp_graph()[0][int(dev_model.n_owned_cp)])
        self._k_owned = self._view(loc.k_values_ptr(), nnz_owned) if nnz_owned else None

    def _build_top_value_exchange(self, dev_model, S, s_index, sptr, snb, cols_g, dev):
        """The rows of S (K restricted to the separators' control points) for the replicated top handle: every rank packs the blocks of the S rows it owns in the
        canonical order (S ascending; per row its S neighbours ascending; per block i, j), ONE all-gather of the packed values, one device gather into the block-CSR
        layout of B.  Source indices into this rank's own K and the placement of every rank's packet are built here once."""
        torch = self.torch
        deg = np.diff(sptr)
        own_of_S = self.owner_cp[S]
        rows_S = np.repeat(np.arange(S.size), deg)                       # row of every block of the S graph
        k_in_row = np.arange(snb.size) - np.repeat(sptr[:-1], deg)
        # destination of (block, i, j) in B's value layout: 9 ptr[a] + i 3 deg(a) + 3 k + j
        dst_blk = 9 * sptr[rows_S] + 3 * k_in_row
        ij_i, ij_j = np.repeat(np.arange(3), 3), np.tile(np.arange(3), 3)
        dst = (dst_blk[:, None] + ij_i[None, :] * 3 * deg[rows_S][:, None] + ij_j[None, :])         # [block][9]
        counts = [int(9 * (own_of_S[rows_S] == r).sum()) for r in range(self.world)]
        self._top_pad = max(1, max(counts))
        place = np.zeros(max(1, 9 * snb.size), np.int64)
        for r in range(self.world):
            sel = np.flatnonzero(own_of_S[rows_S] == r)
            place[dst[sel].ravel()] = r * self._top_pad + np.arange(9 * sel.size)
        self._top_place = torch.from_numpy(place).to(dev)
        # source: this rank's blocks in ITS K (local numbering: the neighbour lists are in local ids, another order than the global one)
        mine = np.flatnonzero(own_of_S[rows_S] == self.rank)
        self._top_send = torch.zeros(self._top_pad, dtype=torch.float64, device=dev)
        self._top_recv = torch.empty(self.world * self._top_pad, dtype=torch.float64, device=dev)
        if mine.size:
            lptr, lnb = dev_model.D.cp_graph()
            g2l = np.full(self.ncp, -1, np.int64)
            g2l[cols_g] = np.arange(cols_g.size)
            a_l, b_g = g2l[S[rows_S[mine]]], S[snb[mine]]
            # slot of neighbour b in the local list of a: search the (row, global neighbour) key in the sorted keys of the local lists of the needed rows
            rows_need = np.unique(a_l)
            ldeg = np.diff(lptr)[rows_need]
            lrow = np.repeat(rows_need, ldeg)
            lpos = np.concatenate([np.arange(lptr[a], lptr[a + 1]) for a in rows_need]) if rows_need.size < 64 else (np.repeat(lptr[rows_need], ldeg) + np.arange(ldeg.sum()) - np.repeat(np.cumsum(ldeg) - ldeg, ldeg))
            lkey = lrow * np.int64(self.ncp) + cols_g[lnb[lpos]]
            o = np.argsort(lkey, kind="stable")
            want = a_l * np.int64(self.ncp) + b_g
            at = np.searchsorted(lkey[o], want)
            if (at >= o.size).any() or (lkey[o][np.minimum(at, o.size - 1)] != want).any():
                raise RuntimeError("DistributedSolver: a block of the separator rows is missing from this rank's K")
            slot = lpos[o][at] - lptr[a_l]                               # k of b in a's local list
            ld = np.diff(lptr)[a_l]
            src = (9 * lptr[a_l] + 3 * slot)[:, None] + ij_i[None, :] * 3 * ld[:, None] + ij_j[None, :]
            self._top_src = torch.from_numpy(src.ravel()).to(dev)
        else:
            self._top_src = None

    def _gather_top_values(self):
        """Collective: the rows of S from their owners into B's value buffer (device buffers; the all-gather is the only exchange of K values there is)."""
        torch = self.torch
        if self._top_src is not None:
            kloc = self._view(self.D.D.k_values_ptr(), 9 * int(self.D.D.cp_graph()[0][-1])) if not hasattr(self, "_k_all") else self._k_all
            self._k_all = kloc
            self._top_send[:self._top_src.numel()] = kloc[self._top_src]
        if self.world > 1:
            self._allgather_into(self._top_recv, self._top_send)
            torch.index_select(self._top_recv, 0, self._top_place, out=self._valB)
        else:
            torch.index_select(self._top_send, 0, self._top_place, out=self._valB)

    def _allgather_into(self, recv, send):
        if self.cuda:
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
        else:                                               # gloo: through the host
            parts = [self.torch.empty(send.numel(), dtype=self.torch.float64) for _ in range(self.world)]
            self.dist.all_gather(parts, send.cpu(), group=self.group)
            recv.copy_(self.torch.cat(parts))

    # -- helpers
    def _view(self, ptr, n):
        class _Buf:
            def __init__(self, p, k):
                self.__cuda_array_interface__ = {"shape": (k,), "typestr": "<f8", "data": (int(p), False), "version": 2}
        return self.torch.as_tensor(_Buf(ptr, n), device=self.torch.device("cuda", self.device))

    def _check(self, rc):
        if rc:
            raise RuntimeError(self.L.gfs_last_error().decode())

    def _allgather(self, buf, pad):
        """In-place all-gather of the ``pad``-sized slice of every rank inside ``buf`` ([world][pad])."""
        mine = buf[self.rank * pad:(self.rank + 1) * pad]
        if self.cuda:
            self.dist.all_gather_into_tensor(buf, mine.clone(), group=self.group)
        else:                                               # gloo: through the host
            parts = [self.torch.empty(pad, dtype=self.torch.float64) for _ in range(self.world)]
            self.dist.all_gather(parts, mine.cpu(), group=self.group)
            buf.copy_(self.torch.cat(parts).to(buf.device))

    def _agree(self, code):
        """Largest ``code`` over the ranks (collective)."""
        t = self.torch.tensor([float(code)], dtype=self.torch.float64, device=self.torch.device("cuda", self.device) if self.cuda else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _any(self, flag):
        """Logical OR of ``flag`` over the ranks (collective)."""
        return self._agree(1 if flag else 0) > 0

    def _parts(self):
        return [p for p in [getattr(self, "A", None)] + list(getattr(self, "T", {}).values()) if p is not None]

    def close(self):
        for p in self._parts():
            if getattr(p, "h", None):
                lib().gfs_destroy(p.h)
                p.h = None

    __del__ = close

    # -- numeric phase
    def _stage_allgather(self, bufs, pads, st):
        if self.stage_exchanges[st] and self.world > 1:
            self._allgather(bufs[st], pads[st])

    def refactor(self):
        """Collective: own subtrees, then the separator fronts level by level (every front by one rank, the Schur complements of a level all-gathered before the next).
        A failure on any rank is agreed on before the next collective and raised on every rank."""
        import time
        L, torch = self.L, self.torch
        st_ = torch.cuda.current_stream(self.b_dev.device)
        self.D.sync()
        t0 = time.perf_counter()
        self._gather_top_values()                              # the rows of the rank separators: the only K values that travel
        st_.synchronize()
        t1 = time.perf_counter()
        err = None
        if self.A is not None:
            try:
                self._check(L.gfs_refactor(self.A.h))
                for t in self.my_roots:
                    self._check(L.gfs_export_schur(self.A.h, self.A.local[t], C.c_void_p(self.schur_buf[0].data_ptr() + 8 * self.schur_off[t])))
            except RuntimeError as e:                          # e.g. a zero pivot in one rank's subtree: every rank must leave the collective phase the same way
                err = str(e)
        self._raise_together(err, "DistributedSolver.refactor (subtrees)")
        t2 = time.perf_counter()
        t_ag = t_top = 0.0
        ta = time.perf_counter()
        self._stage_allgather(self.schur_buf, self.schur_pad, 0)
        st_.synchronize()
        t_ag += time.perf_counter() - ta
        for stg in range(1, self.nstage):
            ta = time.perf_counter()
            err = None
            try:
                for t in self.my_top:
                    if self.stage_of[t] != stg:
                        continue
                    P = self.T[t]
                    self._check(L.gfs_refactor(P.h))
                    if self.sym.parent[t] >= 0:
                        self._check(L.gfs_export_schur(P.h, P.local[t], C.c_void_p(self.schur_buf[stg].data_ptr() + 8 * self.schur_off[t])))
            except RuntimeError as e:
                err = str(e)
            self._raise_together(err, "DistributedSolver.refactor (separator fronts)")
            t_top += time.perf_counter() - ta
            ta = time.perf_counter()
            self._stage_allgather(self.schur_buf, self.schur_pad, stg)
            st_.synchronize()
            t_ag += time.perf_counter() - ta
        #: seconds of the last refactor() on this rank: separator rows of K, own subtrees (+ packing), Schur all-gathers, separator fronts (incl. waiting for their levels)
        self.timings = {"k_values": t1 - t0, "own_subtrees": t2 - t1, "schur_allgather": t_ag, "top": t_top}
        v = (C.c_double * 8)()
        small = False
        for p in self._parts():
            L.gfs_info(p.h, v)
            small = small or bool(v[5])
        self.small_pivot = self._any(small)                    # the same flag on every rank: solve_K's acceptance bar depends on it
        ss = torch.zeros(1, dtype=torch.float64, device=self.b_dev.device)
        if self._k_owned is not None:
            ss += (self._k_owned * self._k_owned).sum()
        if self.world > 1:
            if self.cuda:
                self.dist.all_reduce(ss, group=self.group)
            else:
                h_ = ss.cpu(); self.dist.all_reduce(h_, group=self.group); ss = h_
        self.norm_K = float(ss.sqrt())

    def _substitute(self, b_dev):
        """x = (L D L^T)^-1 b for one right-hand side on the device (torch tensor, replicated); returns a new device tensor (replicated).  Forward: the own subtrees, then
        the separator fronts level by level upwards (boundary contributions all-gathered per level); backward: the levels downwards (the x of a level summed over the
        ranks into the replicated x of S), then the own subtrees.  The library calls return when the device is done with them; torch's stream is drained only where the
        library consumes what torch wrote.  A library error on one rank does not leave the others waiting in a collective: the rank keeps taking part, the failure rides
        in an extra entry of the all-reduced x, and every rank raises."""
        L, torch = self.L, self.torch
        st = torch.cuda.current_stream(b_dev.device)
        err = None
        bS = b_dev[self.S_dofs_g].contiguous()
        try:
            if self.A is not None:
                bl = b_dev[self.loc_dof_g].contiguous()            # the right-hand side in this rank's local numbering
                st.synchronize()
                self._check(L.gfs_forward_dev(self.A.h, C.c_void_p(bl.data_ptr())))
                if self._rootsA.size:
                    t0 = self.my_roots[0]
                    self._check(L.gfs_get_fbnd_packed(self.A.h, self._rootsA.size, self._rootsA.ctypes.data_as(_i64p), C.c_void_p(self.fb_buf[0].data_ptr() + 8 * self.fb_off[t0])))
        except RuntimeError as e:
            err = str(e)
        self._stage_allgather(self.fb_buf, self.fb_pad, 0)
        for stg in range(1, self.nstage):
            try:
                st.synchronize()
                for t in self.my_top:
                    if self.stage_of[t] != stg or err is not None:
                        continue
                    P = self.T[t]
                    for c in self.top_kids[t]:
                        self._check(L.gfs_set_fbnd(P.h, P.local[c], C.c_void_p(self.fb_buf[self.stage_of[c]].data_ptr() + 8 * self.fb_off[c])))
                    self._check(L.gfs_forward_dev(P.h, C.c_void_p(bS.data_ptr())))
                    if self.sym.parent[t] >= 0:
                        self._check(L.gfs_get_fbnd(P.h, P.local[t], C.c_void_p(self.fb_buf[stg].data_ptr() + 8 * self.fb_off[t])))
            except RuntimeError as e:
                err = err or str(e)
            self._stage_allgather(self.fb_buf, self.fb_pad, stg)
        # backward: the separator levels from the root down; xS = the replicated x of S
        xS = torch.zeros(3 * self.S.size, dtype=torch.float64, device=b_dev.device)
        for stg in range(self.nstage - 1, 0, -1):
            delta = torch.zeros_like(xS)
            try:
                for t in self.my_top:
                    if self.stage_of[t] != stg or err is not None:
                        continue
                    P = self.T[t]
                    P.x.copy_(xS)                                       # x of the ancestors' control points at this front's boundary
                    st.synchronize()
                    self._check(L.gfs_backward_dev(P.h))
                    delta[self.top_elim_s[t]] = P.x[self.top_elim_s[t]]
            except RuntimeError as e:
                err = err or str(e)
            if self.world > 1:
                if self.cuda:
                    self.dist.all_reduce(delta, group=self.group)
                else:
                    dc = delta.cpu(); self.dist.all_reduce(dc, group=self.group); delta = dc.to(xS.device)
            xS += delta
        x = torch.zeros(self.n + 1, dtype=torch.float64, device=b_dev.device)
        try:
            if self.A is not None and err is None:
                self.xA[self.S_here_l] = xS[self.S_here_s]              # the separators' x at the boundaries of the own subtrees
                st.synchronize()
                self._check(L.gfs_backward_dev(self.A.h))
                x[self.own_dofs_g] = self.xA[self.own_dofs_l]
            if self.rank == 0:
                x[self.S_dofs_g] = xS
        except RuntimeError as e:
            err = err or str(e)
        x[self.n] = 0.0 if err is None else 1.0
        if self.cuda:
            self.dist.all_reduce(x, group=self.group)
        else:
            xc = x.cpu()
            self.dist.all_reduce(xc, group=self.group)
            x = xc.to(b_dev.device)
        if float(x[self.n]) > 0.0:
            raise RuntimeError("DistributedSolver: a substitution failed on at least one rank (rank %d%s)" % (self.rank, ": " + err if err else ""))
        return x[:self.n]

    def _residual_dev(self, b, x):
        """b - K x with the model's global K (collective on a sharded model); device tensors when the model exchanges on the device, else through the host."""
        if getattr(self.D, "_tdev", None) is not None and hasattr(self.D, "apply_fwd_dev"):
            return b - self.D.apply_fwd_dev(0, x)
        return b - self.torch.from_numpy(self.D.apply(0, x.cpu().numpy())).to(b.device)

    def solve(self, b, transpose=False, max_refine=None):
        """x = K^-1 b; refinement against K itself while a step halves the residual (gfs_solve's rule), ``max_refine`` steps at most.  b comes in and x goes out as host
        arrays (the problem surface's replicated vectors); in between everything stays on the device."""
        torch = self.torch
        if transpose:
            raise NotImplementedError("DistributedSolver: K is symmetric here (general mode is single-GPU)")
        b = np.ascontiguousarray(b, float)
        if b.size != self.n:
            raise ValueError("DistributedSolver.solve: expected %d values, got %d" % (self.n, b.size))
        steps = self.max_refine if max_refine is None else int(max_refine)
        dev = self.b_dev.device
        with torch.cuda.device(dev):
            bd = torch.from_numpy(b).to(dev)
            x = self._substitute(bd)
            nb_ = float(torch.linalg.vector_norm(bd))
            best, x_prev = None, None
            for it in range(steps + 1):
                r = self._residual_dev(bd, x)
                nr = float(torch.linalg.vector_norm(r))
                if best is not None and not nr < 0.5 * best:
                    if nr >= best:
                        x = x_prev
                    else:
                        best = nr
                    break
                best = nr
                if it == steps or nr == 0.0:
                    break
                x_prev = x
                x = x + self._substitute(r.contiguous())
            self.rel_residual = best / nb_ if nb_ > 0 else best
            den = self.norm_K * float(torch.linalg.vector_norm(x)) + nb_
            self.backward_error = best / den if den > 0 else best
            return x.cpu().numpy()

    def solve_multi(self, B, transpose=False, max_refine=None):
        B = np.atleast_2d(np.asarray(B, float))
        X, rr, be = np.empty_like(B), [], 0.0
        for k in range(B.shape[0]):
            X[k] = self.solve(B[k], transpose=transpose, max_refine=max_refine)
            rr.append(self.rel_residual)
            be = max(be, self.backward_error)
        self.rel_residuals = np.array(rr)
        self.rel_residual, self.backward_error = float(max(rr)), be
        return X

    def info(self):
        v = (C.c_double * 8)()
        out = {"device_bytes": 0, "factor_flops": 0.0}
        for p in self._parts():
            self.L.gfs_info(p.h, v)
            out["device_bytes"] += int(v[3])
            out["factor_flops"] += float(v[4])
        out["small_pivot"], out["backward_error"], out["norm_K"] = self.small_pivot, self.backward_error, self.norm_K
        return out


# ---------------------------------------------------------------------------------------------------------------------------------------------------------------
# Round 5: an elimination tree that FOLLOWS THE PATCH PARTITION (VERDICT r04 missing 2 / next 2b).  split_tree above deals the subtrees of the global nested
# dissection to the ranks by work -- whoever factors a subtree needs the K rows of its control points, hence the replicated K (2.8 GB all-gathered per
# factorisation at C4).  MUMPS on ``comm`` (GOLDFISH/utils/opt_utils.py:156-209) does not replicate the matrix.  Here the leaves of the tree are the ranks' own
# control points: with S = the lower-rank end of every edge of the control-point graph that joins control points of two ranks (a vertex separator: no edge joins
# owned(r) \ S and owned(r') \ S), rank r dissects owned(r) \ S by itself, and S is eliminated along the hierarchy of the rank bisection
# (sharding.partition_patches splits the ranks [r0, r0 + n) into [r0, r0 + n // 2) and the rest): a control point of S belongs to the shallowest node of that
# hierarchy at which one of its edges crosses -- one separator front per node, exactly the top of a nested dissection whose first cuts are the partition's.
def rank_tree_nodes(world):
    """Internal nodes of the rank bisection as (r0, n, depth), parents before children."""
    out, stack = [], [(0, world, 0)]
    while stack:
        r0, n, d = stack.pop(0)
        if n <= 1:
            continue
        out.append((r0, n, d))
        stack += [(r0, n // 2, d + 1), (r0 + n // 2, n - n // 2, d + 1)]
    return out


def _lca_node(world, ra, rb):
    """Index (in rank_tree_nodes) of the node at which ranks ra != rb part."""
    nodes = rank_tree_nodes(world)
    index = {(r0, n): i for i, (r0, n, _) in enumerate(nodes)}
    r0, n = 0, world
    while True:
        nl = n // 2
        la, lb = ra < r0 + nl, rb < r0 + nl
        if la != lb:
            return index[(r0, n)]
        r0, n = (r0, nl) if la else (r0 + nl, n - nl)


def _induced(nb_ptr, nb, ids, ncp):
    """Subgraph induced by the control points ``ids`` (ascending): (ptr, nb) in local numbers; the lists keep the control point itself."""
    loc = np.full(ncp, -1, np.int64)
    loc[ids] = np.arange(ids.size)
    deg = np.diff(nb_ptr)[ids]
    rows = np.repeat(np.arange(ids.size), deg)
    cols = loc[np.concatenate([nb[nb_ptr[a]:nb_ptr[a + 1]] for a in ids])] if ids.size < 64 else loc[nb[np.repeat(nb_ptr[ids], deg) + (np.arange(deg.sum()) - np.repeat(np.cumsum(deg) - deg, deg))]]
    keep = cols >= 0
    ptr = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=ids.size))]).astype(np.int64)
    return ptr, cols[keep].astype(np.int32)


def partition_tree(nb_ptr, nb, coords, owner_cp, world, leaf=128, native=True):
    """(Symbolic, owner, roots) of the partition-following elimination tree: ``owner[t]`` = the rank whose control points front t eliminates (-1: a separator front
    of the rank hierarchy, replicated), ``roots`` = the fronts of rank subtrees whose parent is a separator front (or none).  Boundaries and parents come from one
    pass over the fronts in elimination order (bnd(t) = later neighbours of t's own control points + the boundaries of its children, parent(t) = the front of the
    first boundary control point), then the fronts are renumbered in post-order (what the numeric phase expects: a subtree is a contiguous range ending at its root)."""
    nb_ptr, nb = np.asarray(nb_ptr, np.int64), np.asarray(nb, np.int64)
    ncp = nb_ptr.size - 1
    owner_cp = np.asarray(owner_cp, np.int64)
    X = np.asarray(coords, float).reshape(ncp, -1)
    rows = np.repeat(np.arange(ncp, dtype=np.int64), np.diff(nb_ptr))
    cut = owner_cp[rows] < owner_cp[nb]                      # the lower-rank end of an edge between two ranks goes into S ...
    node_of = np.full(ncp, -1, np.int64)                     # ... at the shallowest node of the rank hierarchy one of its edges crosses
    if cut.any():
        pair = owner_cp[rows[cut]] * world + owner_cp[nb[cut]]
        nodes = rank_tree_nodes(world)
        depth_of = np.array([d for (_, _, d) in nodes])
        up, inv = np.unique(pair, return_inverse=True)
        lca = np.array([_lca_node(world, int(p // world), int(p % world)) for p in up])[inv]
        # shallowest node per control point: sort by (cp, depth) and keep the first
        o = np.lexsort((depth_of[lca], rows[cut]))
        cp_s, lca_s = rows[cut][o], lca[o]
        first = np.concatenate([[True], cp_s[1:] != cp_s[:-1]])
        node_of[cp_s[first]] = lca_s[first]
    in_S = node_of >= 0
    # ---- the fronts in a first elimination order: every rank's own nested dissection, then the separator nodes deepest first
    elim_parts, owner_parts = [], []
    for r in range(world):
        ids = np.flatnonzero((owner_cp == r) & ~in_S)
        if ids.size == 0:
            continue
        ptr, nbs = _induced(nb_ptr, nb, ids, ncp)
        sym_r = (_nd.nested_dissection_native(ptr, nbs, X[ids], leaf=leaf)[0] if native else _nd.nested_dissection(ptr, nbs, X[ids], leaf=leaf))
        for t in range(sym_r.nfronts):
            elim_parts.append(ids[sym_r.elim[sym_r.elim_off[t]:sym_r.elim_off[t + 1]]])
            owner_parts.append(r)
    if in_S.any():
        nodes = rank_tree_nodes(world)
        for i in sorted(range(len(nodes)), key=lambda i: (-nodes[i][2], i)):        # deepest first
            ids = np.flatnonzero(node_of == i)
            if ids.size:
                elim_parts.append(ids[np.argsort(X[ids, 0], kind="stable")])
                owner_parts.append(-1)
    nf = len(elim_parts)
    elim = np.concatenate(elim_parts).astype(np.int64)
    assert elim.size == ncp and np.unique(elim).size == ncp
    elim_off = np.concatenate([[0], np.cumsum([e.size for e in elim_parts])]).astype(np.int64)
    order = np.empty(ncp, np.int64); order[elim] = np.arange(ncp)
    front_of = np.repeat(np.arange(nf), np.diff(elim_off))[order]
    # later neighbours of every front's own control points (outside the front), sorted by elimination position
    later = (order[nb] > order[rows]) & (front_of[nb] != front_of[rows])
    key = np.unique(front_of[rows[later]] * np.int64(ncp) + order[nb[later]])
    ef, eb = key // ncp, elim[key % ncp]
    own_off = np.concatenate([[0], np.cumsum(np.bincount(ef, minlength=nf))]).astype(np.int64)
    parent = np.full(nf, -1, np.int64)
    kids = [[] for _ in range(nf)]
    bnds = [None] * nf
    for t in range(nf):
        cand = eb[own_off[t]:own_off[t + 1]]
        if kids[t]:
            cand = np.unique(np.concatenate([cand] + [bnds[c] for c in kids[t]]))
            cand = cand[front_of[cand] != t]
            cand = cand[np.argsort(order[cand], kind="stable")]
        bnds[t] = cand
        if cand.size:
            parent[t] = front_of[cand[0]]
            kids[parent[t]].append(t)
    # ---- post-order renumbering
    post, stack = [], [(t, False) for t in reversed([t for t in range(nf) if parent[t] < 0])]
    while stack:
        t, seen = stack.pop()
        if seen:
            post.append(t)
        else:
            stack.append((t, True))
            stack += [(c, False) for c in reversed(kids[t])]
    new = np.empty(nf, np.int64); new[post] = np.arange(nf)
    elim2 = np.concatenate([elim_parts[t] for t in post]).astype(np.int64)
    elim_off2 = np.concatenate([[0], np.cumsum([elim_parts[t].size for t in post])]).astype(np.int64)
    order2 = np.empty(ncp, np.int64); order2[elim2] = np.arange(ncp)
    front_of2 = new[front_of]
    b2 = [bnds[t][np.argsort(order2[bnds[t]], kind="stable")] for t in post]
    bnd_off2 = np.concatenate([[0], np.cumsum([b.size for b in b2])]).astype(np.int64)
    parent2 = np.array([new[parent[t]] if parent[t] >= 0 else -1 for t in post], np.int64)
    sym = _nd.Symbolic(elim2, elim_off2, np.concatenate(b2).astype(np.int64) if nf else np.zeros(0, np.int64), bnd_off2, parent2, order2, front_of2)
    owner = np.array([owner_parts[t] for t in post], np.int64)
    roots = [t for t in range(nf) if owner[t] >= 0 and (parent2[t] < 0 or owner[parent2[t]] < 0)]
    return sym, owner, roots
