"""Direct solves with K over several GPUs (SURVEY.md 8(e) + 8(f) N1, stage 2): one process per GPU, the nested-dissection elimination tree cut below its top.

Replaces, on a sharded problem, the replicated factorisation of stage 1 (every rank factors all of K: goldfish_amd/_solver.py with the gathered values) -- the
reference's counterpart is MUMPS on ``comm`` (GOLDFISH/utils/opt_utils.py:156-209).  The symbolic phase (goldfish_amd/_nd.py) is deterministic, so every rank
computes the same tree.  The subtrees hanging below the top ``depth`` levels are dealt to the ranks (largest first, by factorisation work); every rank

  1. factors its own subtrees (handle A: gfs_create_nd_partial over the sub-forest; the root fronts keep their Schur complements),
  2. packs the Schur complements of its subtree roots (gfs_export_schur) -- one all-gather of padded buffers,
  3. factors the top of the tree (handle B: the top fronts above STUB fronts that stand for the subtrees and carry the gathered Schur complements) -- the same
     arithmetic on every rank: the top factors are replicated, nothing else is exchanged,

and a solve is: forward sweep of the own subtrees (A), all-gather of the root fronts' boundary contributions (3 doubles per boundary control point), forward and
backward sweep of the top (B, replicated), backward sweep of the own subtrees with the top's x at their boundaries, all-reduce of the pieces of x.  Iterative
refinement runs against K itself (the replicated values: the same ``residual`` on every rank), like gfs_solve.

What is distributed: the subtree factorisations and sweeps (C4 on 8 ranks: 7.2 of the 9.3 Tflop, 1/8 each).  What is replicated: the top fronts (2.1 Tflop), K's
values (stage 1's all-gather of the owned value rows), the refinement residuals.  Factor memory per rank: own subtrees + top + the stubs' Schur complements.

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
    """One partial handle of libgoldfish_solver (gfs_create_nd_partial) over partial_symbolic(...)."""

    def __init__(self, L, sym, nb_ptr, nb, dK, device, keep, stub_roots=(), later_cp=None):
        fronts, sub, pmap = partial_symbolic(sym, keep, stub_roots, later_cp)
        self.fronts, self.local, self.sym = fronts, {int(t): i for i, t in enumerate(fronts)}, sub
        self.nbnd = np.diff(sub.bnd_off)
        self._keep = [np.ascontiguousarray(a, np.int64) for a in (sub.elim, sub.elim_off, sub.bnd, sub.bnd_off, sub.parent, sub.order, sub.front_of, pmap)]
        h = C.c_void_p()
        i64 = lambda a: a.ctypes.data_as(_i64p)
        rc = L.gfs_create_nd_partial(int(device), sym.order.size, nb_ptr.ctypes.data_as(_i64p), nb.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(dK), fronts.size,
                                     *[i64(a) for a in self._keep], C.byref(h))
        if rc:
            raise RuntimeError(L.gfs_last_error().decode())
        self.h = h


class DistributedSolver:
    """K x = b over the ranks of ``dist`` (see the module docstring).  ``dev_model``: the sharded device model (goldfish_amd/sharding.py: ShardedDeviceModel) -- its
    ``pattern`` / ``k_values_ptr`` / ``refresh_k_values`` give the GLOBAL K with replicated values on this rank's GPU, as the stage-1 DeviceSolver uses them."""

    method = "nd-distributed"

    def __init__(self, dev_model, dist, group=None, coords=None, leaf=128, max_refine=3):
        import torch
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.D, self.max_refine = dev_model, max_refine
        self.device = int(dev_model.device)
        self.cuda = dist.get_backend(group) == "nccl"
        L = self.L = _bind()
        rowptr, col = dev_model.pattern(0)
        self.nb_ptr, self.nb = control_point_graph(rowptr, col)
        del rowptr, col
        ncp = self.ncp = self.nb_ptr.size - 1
        self.n = 3 * ncp
        if coords is None:
            raise ValueError("DistributedSolver needs the control points' coordinates")
        sym = self.sym = _nd.nested_dissection_native(self.nb_ptr, self.nb, coords, leaf=leaf)[0]      # deterministic (also across thread counts): the same tree on every rank
        self.owner, self.roots = split_tree(sym, self.world)
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
        self.refactor()

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
        dK = dev_model.k_values_ptr()
        mine = np.flatnonzero(self.owner == self.rank)
        self.my_roots = [t for t in self.roots if self.owner[t] == self.rank]
        top_f = np.flatnonzero(self.owner == -1)
        top_cp = np.concatenate([sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]] for t in top_f]) if top_f.size else np.zeros(0, np.int64)
        self.A = _Part(L, sym, self.nb_ptr, self.nb, dK, self.device, mine, later_cp=top_cp) if mine.size else None
        self.B = _Part(L, sym, self.nb_ptr, self.nb, dK, self.device, top_f, self.roots)
        dev = torch.device("cuda", self.device)
        # exchange buffers: Schur complements (doubles per subtree root, padded to the largest per rank) and boundary contributions
        self.schur_len = {t: int(L.gfs_schur_doubles(self.B.h, self.B.local[t])) for t in self.roots}
        self.fb_len = {t: 3 * int(self.B.nbnd[self.B.local[t]]) for t in self.roots}
        per_rank = [[t for t in self.roots if self.owner[t] == r] for r in range(self.world)]
        self.per_rank = per_rank
        self.schur_pad = max(1, max(sum(self.schur_len[t] for t in ts) for ts in per_rank))
        self.fb_pad = max(1, max(sum(self.fb_len[t] for t in ts) for ts in per_rank))
        self.schur_all = torch.zeros(self.world * self.schur_pad, dtype=torch.float64, device=dev)      # [rank][its roots one after the other]: B's stubs read from here
        self.fb_all = torch.zeros(self.world * self.fb_pad, dtype=torch.float64, device=dev)
        for r, ts in enumerate(per_rank):
            off = r * self.schur_pad
            for t in ts:
                if L.gfs_set_schur_source(self.B.h, self.B.local[t], C.c_void_p(self.schur_all.data_ptr() + 8 * off)):
                    raise RuntimeError(L.gfs_last_error().decode())
                off += self.schur_len[t]
        # the roots of every rank in B's / A's local front numbers (the packed boundary copies of a substitution)
        self._rootsB = [np.ascontiguousarray([self.B.local[t] for t in ts], np.int64) for ts in per_rank]
        self._rootsA = np.ascontiguousarray([self.A.local[t] for t in self.my_roots], np.int64) if self.A else np.zeros(0, np.int64)
        # index sets of the exchange of x: the dofs this rank's subtrees eliminate, the dofs of the top
        def dofs(cps):
            return (3 * np.asarray(cps, np.int64)[:, None] + np.arange(3)).ravel()
        own_cp = np.concatenate([sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]] for t in mine]) if mine.size else np.zeros(0, np.int64)
        self.own_dofs = torch.from_numpy(dofs(own_cp)).to(dev)
        self.top_dofs = torch.from_numpy(dofs(top_cp)).to(dev)
        self.xA = self._view(L.gfs_x_ptr(self.A.h), self.n) if self.A else None
        self.xB = self._view(L.gfs_x_ptr(self.B.h), self.n)
        self.b_dev = torch.zeros(self.n, dtype=torch.float64, device=dev)

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

    def close(self):
        for part in ("A", "B"):
            p = getattr(self, part, None)
            if p is not None and getattr(p, "h", None):
                lib().gfs_destroy(p.h)
                p.h = None

    __del__ = close

    # -- numeric phase
    def refactor(self):
        """Collective: own subtrees, all-gather of their Schur complements, the replicated top.  A failure on any rank is agreed on before the next collective
        and raised on every rank."""
        import time
        L, torch = self.L, self.torch
        self.D.sync()
        t0 = time.perf_counter()
        if hasattr(self.D, "refresh_k_values"):
            self.D.refresh_k_values()
        t1 = time.perf_counter()
        err = None
        if self.A is not None:
            try:
                self._check(L.gfs_refactor(self.A.h))
                off = self.rank * self.schur_pad
                for t in self.my_roots:
                    self._check(L.gfs_export_schur(self.A.h, self.A.local[t], C.c_void_p(self.schur_all.data_ptr() + 8 * off)))
                    off += self.schur_len[t]
            except RuntimeError as e:                          # e.g. a zero pivot in one rank's subtree: every rank must leave the collective phase the same way
                err = str(e)
        self._raise_together(err, "DistributedSolver.refactor (subtrees)")
        t2 = time.perf_counter()
        self._allgather(self.schur_all, self.schur_pad)
        torch.cuda.current_stream(self.b_dev.device).synchronize()
        t3 = time.perf_counter()
        err = None
        try:                                                   # the same arithmetic on every rank -- but a rank may still run out of memory alone
            self._check(L.gfs_refactor(self.B.h))
        except RuntimeError as e:
            err = str(e)
        self._raise_together(err, "DistributedSolver.refactor (top)")
        t4 = time.perf_counter()
        #: seconds of the last refactor() on this rank: K value gather, own subtrees (+ packing), Schur all-gather, replicated top
        self.timings = {"k_values": t1 - t0, "own_subtrees": t2 - t1, "schur_allgather": t3 - t2, "top": t4 - t3}
        v = (C.c_double * 8)()
        small = False
        for p in (self.A, self.B):
            if p is not None:
                L.gfs_info(p.h, v)
                small = small or bool(v[5])
        self.small_pivot = self._any(small)                    # the same flag on every rank: solve_K's acceptance bar depends on it
        L.gfs_info(self.B.h, v)
        self.norm_K = float(v[7])

    def _substitute(self, b_dev):
        """x = (L D L^T)^-1 b for one right-hand side on the device (torch tensor, replicated); returns a new device tensor (replicated).  The library calls return when
        the device is done with them; torch's stream is drained only where the library consumes what torch wrote (no device-wide synchronisations).  A library error on
        one rank does not leave the others waiting in a collective: the rank keeps taking part, the failure rides in an extra entry of the all-reduced x, and every
        rank raises."""
        L, torch = self.L, self.torch
        st = torch.cuda.current_stream(b_dev.device)
        err = None
        st.synchronize()                                        # b_dev may have been written by torch
        try:
            if self.A is not None:
                self._check(L.gfs_forward_dev(self.A.h, C.c_void_p(b_dev.data_ptr())))
                self._check(L.gfs_get_fbnd_packed(self.A.h, self._rootsA.size, self._rootsA.ctypes.data_as(_i64p), C.c_void_p(self.fb_all.data_ptr() + 8 * self.rank * self.fb_pad)))
        except RuntimeError as e:
            err = str(e)
        self._allgather(self.fb_all, self.fb_pad)
        x = torch.zeros(self.n + 1, dtype=torch.float64, device=b_dev.device)
        try:
            st.synchronize()
            for r, rb in enumerate(self._rootsB):
                if rb.size:
                    self._check(L.gfs_set_fbnd_packed(self.B.h, rb.size, rb.ctypes.data_as(_i64p), C.c_void_p(self.fb_all.data_ptr() + 8 * r * self.fb_pad)))
            self._check(L.gfs_forward_dev(self.B.h, C.c_void_p(b_dev.data_ptr())))
            self._check(L.gfs_backward_dev(self.B.h))
            if self.A is not None and err is None:
                self.xA[self.top_dofs] = self.xB[self.top_dofs]          # the top's x at the boundaries of the own subtrees
                st.synchronize()
                self._check(L.gfs_backward_dev(self.A.h))
                x[self.own_dofs] = self.xA[self.own_dofs]
            if self.rank == 0:
                x[self.top_dofs] = self.xB[self.top_dofs]
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
        for p in (self.A, self.B):
            if p is not None:
                self.L.gfs_info(p.h, v)
                out["device_bytes"] += int(v[3])
                out["factor_flops"] += float(v[4])
        out["small_pivot"], out["backward_error"], out["norm_K"] = self.small_pivot, self.backward_error, self.norm_K
        return out
