"""Preconditioned conjugate gradients on the device for K x = b -- the counterpart of the reference's Krylov helper ``PETSc_ksp_solve(A, x, b, ksp_type='cg',
pc_type='jacobi', max_it=10000, rtol=1e-15)`` (GOLDFISH/utils/opt_utils.py:104-131; the reference defines it next to the MUMPS solves and never calls it).

K stays where the assembly left it: the products run through ``gf_apply_dev`` (csr_apply_kernel: 0.66 - 0.74 of HBM peak), the vectors are torch tensors on the
same GPU, ordered against the library's stream by events (no host synchronisation inside the iteration); the dot products are torch reductions of fixed length
(one reduction tree per length: run-to-run reproducible).  Preconditioners: ``"jacobi"`` (the reference's default: the diagonal of K), ``"bjacobi"`` (the 3 x 3
diagonal block of every control point, inverted once), ``"none"``.

What to expect (measured, profiles/r05_krylov_study.txt): the tangent of a penalty-coupled thin shell has cond(K) ~ 1e12 ... 1e15 (alpha_d = 1e3 E h / h_e
against a bending stiffness of E h^3 / L^4), and neither these point preconditioners nor one- or two-level Schwarz preconditioners built from the direct solver's
subtree factorisations bring CG below a few hundred iterations even on 16-patch models (the study lists iteration counts); the direct solver
(goldfish_amd/_solver.py, _dsolver.py) is the solve path of this framework, this module is the reference's utility on the device with an honest status."""
import ctypes as C

import numpy as np


class DevicePCG:
    """CG on K of a goldfish_amd._lib.DeviceModel (single GPU; K symmetric: no follower pressure).  After ``solve``: ``iterations``, ``converged``,
    ``rel_residual`` (|b - K x| / |b| recomputed from x, not the recursion's), ``history`` (recursive relative residual per iteration)."""

    def __init__(self, dev_model, pc_type="jacobi"):
        import torch
        from . import _lib
        self.torch, self._lib, self.D = torch, _lib, dev_model
        self.n = int(dev_model.ndof)
        self.dev = torch.device("cuda", int(dev_model.device))
        self.lib_stream = torch.cuda.ExternalStream(dev_model.stream_ptr, device=self.dev)
        self.pc_type = pc_type
        self._pc = None
        self.iterations, self.converged, self.rel_residual, self.history = 0, False, None, []

    # K's values as a torch view of the library's buffer (no copy)
    def _k_values(self):
        torch = self.torch
        nnz = int(self._lib.lib().gf_nnz(self.D.h, self._lib.MAT_K))

        class _Buf:
            def __init__(self, p, k):
                self.__cuda_array_interface__ = {"shape": (k,), "typestr": "<f8", "data": (int(p), False), "version": 2}
        return torch.as_tensor(_Buf(self.D.k_values_ptr(), nnz), device=self.dev)

    def refresh(self):
        """(Re)build the preconditioner from the K currently assembled (call after a new assembly)."""
        torch = self.torch
        self._pc = None
        if self.pc_type == "none":
            return
        nb_ptr, nb = self.D.cp_graph()
        ncp = nb_ptr.size - 1
        deg = np.diff(nb_ptr)
        rows = np.repeat(np.arange(ncp), deg)
        pos = np.flatnonzero(nb == rows)                         # every control point lists itself
        if pos.size != ncp:
            raise RuntimeError("DevicePCG: K's pattern has no diagonal block for %d control points" % (ncp - pos.size))
        k = pos - nb_ptr[:-1]                                    # position of a in its own neighbour list
        # value of ((a, i), (a, j)) at 9 nb_ptr[a] + i 3 deg(a) + 3 k + j
        base = 9 * nb_ptr[:-1] + 3 * k
        idx = (base[:, None, None] + (3 * deg)[:, None, None] * np.arange(3)[None, :, None] + np.arange(3)[None, None, :]).reshape(-1)
        self.D.sync()
        blocks = self._k_values()[torch.from_numpy(idx).to(self.dev)].reshape(ncp, 3, 3)
        if self.pc_type == "jacobi":
            d = torch.diagonal(blocks, dim1=1, dim2=2).reshape(-1)
            if not bool((d > 0).all()):
                raise RuntimeError("DevicePCG: K has a non-positive diagonal entry (assemble K first)")
            self._pc = ("diag", 1.0 / d)
        elif self.pc_type == "bjacobi":
            self._pc = ("block", torch.linalg.inv(blocks))
        else:
            raise ValueError("DevicePCG: pc_type must be 'jacobi', 'bjacobi' or 'none'")

    def _apply_K(self, x, y):
        torch = self.torch
        y.zero_()
        self.lib_stream.wait_stream(torch.cuda.current_stream(self.dev))
        if self._lib.lib().gf_apply_dev(self.D.h, self._lib.MAT_K, 0, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())):
            raise RuntimeError(self._lib.lib().gf_last_error().decode())
        torch.cuda.current_stream(self.dev).wait_stream(self.lib_stream)
        return y

    def _apply_pc(self, r):
        if self._pc is None:
            return r.clone()
        kind, P = self._pc
        if kind == "diag":
            return P * r
        return self.torch.bmm(P, r.reshape(-1, 3, 1)).reshape(-1)

    def solve(self, b, x0=None, rtol=1e-15, max_it=10000, check_every=50):
        """x with |b - K x| <= rtol |b| (recursive residual; the true one is reported), at most ``max_it`` iterations.  ``b``: host array or torch tensor on the
        model's GPU; the result has the same kind.  The iteration also ends when the recursive residual has not improved on its best value for 20 % of ``max_it``
        iterations (stagnation: ``converged`` stays False)."""
        torch = self.torch
        host = not isinstance(b, torch.Tensor)
        with torch.cuda.device(self.dev):
            bt = torch.as_tensor(np.ascontiguousarray(b, float)).to(self.dev) if host else b
            if bt.numel() != self.n:
                raise ValueError("DevicePCG.solve: expected %d values, got %d" % (self.n, bt.numel()))
            if self._pc is None and self.pc_type != "none":
                self.refresh()
            x = torch.zeros_like(bt) if x0 is None else (torch.as_tensor(np.ascontiguousarray(x0, float)).to(self.dev) if not isinstance(x0, torch.Tensor) else x0.clone())
            Kp = torch.empty_like(bt)
            r = bt - self._apply_K(x, Kp) if x0 is not None else bt.clone()
            nb = float(torch.linalg.vector_norm(bt))
            self.history, self.iterations, self.converged = [], 0, False
            if nb == 0.0:
                self.rel_residual, self.converged = 0.0, True
                return x.cpu().numpy() if host else x
            z = self._apply_pc(r)
            p = z.clone()
            rz = torch.dot(r, z)
            best, best_it = float("inf"), 0
            for it in range(1, int(max_it) + 1):
                self._apply_K(p, Kp)
                alpha = rz / torch.dot(p, Kp)
                x.add_(p * alpha)
                r.sub_(Kp * alpha)
                z = self._apply_pc(r)
                rz_new = torch.dot(r, z)
                p.mul_(rz_new / rz).add_(z)
                rz = rz_new
                self.iterations = it
                if it % check_every == 0 or it == max_it or it < 10:     # the norm is a host read: not in every iteration
                    rr = float(torch.linalg.vector_norm(r)) / nb
                    self.history.append((it, rr))
                    if not np.isfinite(rr):
                        break
                    if rr <= rtol:
                        self.converged = True
                        break
                    if rr < 0.999 * best:
                        best, best_it = rr, it
                    elif it - best_it >= max(2 * check_every, int(0.2 * max_it)):
                        break
            self.rel_residual = float(torch.linalg.vector_norm(bt - self._apply_K(x, Kp))) / nb
            self.converged = bool(self.converged and np.isfinite(self.rel_residual))
            return x.cpu().numpy() if host else x
