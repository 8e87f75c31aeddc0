"""ctypes binding of libgoldfish_hip.so (include/goldfish_hip.h).  There is no CPU
fallback: a missing library or a missing GPU raises."""
import ctypes as C
import os
import weakref
from collections.abc import Mapping

import numpy as np

from .model import gf_model_desc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GF_LIB", os.path.join(_HERE, "libgoldfish_hip.so"))   # GF_LIB: A/B builds while tuning
_LIB = None

ASM_R, ASM_K, ASM_DRDCP, ASM_DRDH, ASM_ALL = 1, 2, 4, 8, 15
MAT_K, MAT_DRDCP0, MAT_DRDCP1, MAT_DRDCP2, MAT_DRDH = range(5)
BUF_CP, BUF_U, BUF_H, BUF_R, BUF_VAL_K, BUF_VAL_C0, BUF_VAL_C1, BUF_VAL_C2, BUF_VAL_H = range(9)

EXPORTS = ["gf_device_count", "gf_last_error", "gf_create", "gf_destroy", "gf_total_cp", "gf_num_dofs",
           "gf_num_elements", "gf_num_gauss_points", "gf_num_mortar_points", "gf_device_bytes", "gf_set_cp",
           "gf_set_thickness", "gf_set_u", "gf_nnz", "gf_pattern", "gf_cp_graph_size", "gf_cp_graph", "gf_assemble", "gf_sync", "gf_get_residual",
           "gf_get_values", "gf_apply", "gf_functionals", "gf_compliance", "gf_stress_forms", "gf_penalty_dxi", "gf_shape_regu", "gf_device_ptr", "gf_apply_dev", "gf_kernel_ms", "gf_assembly_path", "gf_stream", "gf_apply_many", "gf_get_functional_gradient", "gf_penalty_dxi_range", "gf_functionals_per_patch", "gf_penalty_dxi_rev", "gf_update_interface"]


def one_hip_runtime():
    """PyTorch-ROCm bundles its own libamdhip64.so.  A process that loads the system runtime first (through libgoldfish_hip.so / libgoldfish_solver.so) and touches
    torch.cuda later runs TWO HIP runtimes, and torch then reports "No HIP GPUs are available" (measured on the GPU box, round 5: /proc/self/maps shows
    /opt/rocm/lib/libamdhip64.so.7 next to torch/lib/libamdhip64.so).  With torch imported first the loader binds the libraries to torch's copy (same SONAME) and there
    is one runtime -- what the sharded exchange, the distributed solver and the device CG (all torch tensors on the library's buffers) need.  Importing torch does not
    initialise the GPU.  GF_NO_TORCH_PRELOAD=1 skips this (processes that never use torch)."""
    if os.environ.get("GF_NO_TORCH_PRELOAD") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgoldfish_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                               "goldfish_amd has no CPU fallback")
        one_hip_runtime()
        L = C.CDLL(LIB_PATH)
        dp, vp, i64, ci = C.POINTER(C.c_double), C.c_void_p, C.c_int64, C.c_int
        i32p = C.POINTER(C.c_int32)
        sig = {   # name: (argtypes, restype); restype None keeps ctypes' default int
            "gf_device_count": (None, ci), "gf_last_error": (None, C.c_char_p), "gf_create": ([C.POINTER(gf_model_desc), ci, C.POINTER(vp)], None),
            "gf_destroy": ([vp], "void"), "gf_set_cp": ([vp, ci, dp, i64], None), "gf_set_thickness": ([vp, dp, i64], None), "gf_set_u": ([vp, dp, i64], None),
            "gf_nnz": ([vp, ci], i64), "gf_pattern": ([vp, ci, C.POINTER(C.c_int64), i32p], None), "gf_cp_graph_size": ([vp], i64), "gf_cp_graph": ([vp, C.POINTER(C.c_int64), i32p], None), "gf_assemble": ([vp, ci], None), "gf_sync": ([vp], None),
            "gf_get_residual": ([vp, dp, i64], None), "gf_get_values": ([vp, ci, dp, i64], None), "gf_apply": ([vp, ci, ci, dp, i64, dp, i64], None),
            "gf_functionals": ([vp, dp, dp, dp, dp, dp, dp, ci], None), "gf_compliance": ([vp, dp, i64, dp, dp, dp, ci], None),
            "gf_shape_regu": ([vp, ci, dp, i64, dp, i64, dp, dp], None), "gf_penalty_dxi": ([vp, dp, i64, i32p, i64], None),
            "gf_penalty_dxi_range": ([vp, i64, i64, dp, i64, i32p, i64], None),
            "gf_stress_forms": ([vp, ci, C.c_double, dp, i64, ci, ci, dp, dp, dp, dp, dp, ci], None), "gf_device_ptr": ([vp, ci], vp),
            "gf_apply_dev": ([vp, ci, ci, vp, vp], None), "gf_kernel_ms": ([vp, C.POINTER(ci)], C.c_double), "gf_assembly_path": ([vp], None),
            "gf_get_functional_gradient": ([vp, ci, dp, i64], None), "gf_apply_many": ([vp, ci, ci, C.POINTER(ci), C.POINTER(dp), C.POINTER(dp)], None),
            "gf_stream": ([vp], vp), "gf_functionals_per_patch": ([vp, dp, dp, i64], None),
            "gf_penalty_dxi_rev": ([vp, i64, i64, dp, i64, dp, i64], None), "gf_update_interface": ([vp, ci, dp, dp, dp, i64], None)}
        for name in ("gf_total_cp", "gf_num_dofs", "gf_num_elements", "gf_num_gauss_points", "gf_num_mortar_points", "gf_device_bytes"):
            sig[name] = ([vp], i64)
        for name, (argtypes, restype) in sig.items():
            if not hasattr(L, name):
                continue                  # an older library loaded through GF_LIB for an A/B run binds only the entry points it has
            fn = getattr(L, name)
            if argtypes is not None:
                fn.argtypes = argtypes
            if restype is not None:
                fn.restype = None if restype == "void" else restype
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _check(rc, exc=RuntimeError):
    if rc != 0:
        raise exc(lib().gf_last_error().decode())


class _LazyFields(Mapping):
    """Results of a functional evaluation; the gradient fields are fetched from the device on first access (a read-only
    mapping: iteration, len, keys / values / items and dict(...) see every field, fetched or not)."""

    def __init__(self, dev, fields, values=()):
        self._dev, self._pending, self._data = dev, dict(fields), dict(values)

    def _fetch(self, key):
        field, shape = self._pending[key]
        arr = np.zeros(shape)
        _check(lib().gf_get_functional_gradient(self._dev.h, field, _dp(arr), arr.size))
        self._data[key] = arr
        del self._pending[key]                   # only after a successful fetch

    def __getitem__(self, key):
        if key in self._pending:
            self._fetch(key)
        return self._data[key]

    def __setitem__(self, key, value):
        self._pending.pop(key, None)
        self._data[key] = value

    def __iter__(self):
        yield from self._data
        yield from self._pending

    def __len__(self):
        return len(self._data) + len(self._pending)

    def __contains__(self, key):
        return key in self._data or key in self._pending

    def materialize(self):
        for key in list(self._pending):
            self._fetch(key)
        return self._data

    def copy(self):
        return dict(self.materialize())


class DeviceModel:
    """Owns one gf_handle: device-resident state, static CSR patterns and results."""

    def __init__(self, arrays, device=0):
        self.arrays = arrays
        self._desc = arrays.desc()
        h = C.c_void_p()
        _check(lib().gf_create(C.byref(self._desc), int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self.total_cp, self.ndof = arrays.total_cp, arrays.ndof
        self._pat = {}
        for f in range(3):
            self.set_cp(f, arrays.cp_hom[f])

    def close(self):
        if getattr(self, "h", None):
            self._materialize_lazy()              # results handed out stay valid after the handle is gone
            lib().gf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _vec(self, v, n, what):
        v = np.ascontiguousarray(v, dtype=np.float64).ravel()
        if v.size != n:
            raise ValueError("%s: array of length %d, expected %d" % (what, v.size, n))
        return v

    def set_cp(self, field, v):
        v = self._vec(v, self.total_cp, "set_cp")
        _check(lib().gf_set_cp(self.h, int(field), _dp(v), v.size), ValueError)

    def set_thickness(self, v):
        v = self._vec(v, self.total_cp, "set_thickness")
        _check(lib().gf_set_thickness(self.h, _dp(v), v.size), ValueError)

    def set_u(self, v):
        v = self._vec(v, self.ndof, "set_u")
        _check(lib().gf_set_u(self.h, _dp(v), v.size), ValueError)

    def pattern(self, which):
        if which not in self._pat:
            nnz = lib().gf_nnz(self.h, which)
            rowptr = np.zeros(self.ndof + 1, np.int64)
            col = np.zeros(nnz, np.int32)
            _check(lib().gf_pattern(self.h, which, rowptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                    col.ctypes.data_as(C.POINTER(C.c_int32))))
            self._pat[which] = (rowptr, col)
        return self._pat[which]

    def cp_graph(self):
        """(nb_ptr, nb): control-point-level pattern of K (gf_cp_graph) -- what the device solver's symbolic phase takes."""
        n = lib().gf_cp_graph_size(self.h)
        nb_ptr, nb = np.zeros(self.total_cp + 1, np.int64), np.zeros(n, np.int32)
        _check(lib().gf_cp_graph(self.h, nb_ptr.ctypes.data_as(C.POINTER(C.c_int64)), nb.ctypes.data_as(C.POINTER(C.c_int32))))
        return nb_ptr, nb

    def assemble(self, flags=ASM_ALL, sync=True):
        _check(lib().gf_assemble(self.h, int(flags)))
        if sync:
            _check(lib().gf_sync(self.h))

    def sync(self):
        _check(lib().gf_sync(self.h))

    def residual(self):
        R = np.zeros(self.ndof)
        _check(lib().gf_get_residual(self.h, _dp(R), R.size))
        return R

    def values(self, which):
        v = np.zeros(lib().gf_nnz(self.h, which))
        _check(lib().gf_get_values(self.h, which, _dp(v), v.size))
        return v

    def csr(self, which):
        import scipy.sparse as sp
        rowptr, col = self.pattern(which)
        ncol = self.ndof if which == MAT_K else self.total_cp
        return sp.csr_matrix((self.values(which), col, rowptr), shape=(self.ndof, ncol))

    def apply(self, which, x, y, transpose=False):
        """y += A x  (or A^T x), in place on the NumPy array y."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        if not (isinstance(y, np.ndarray) and y.dtype == np.float64 and y.flags.c_contiguous):
            raise ValueError("apply: y must be a contiguous float64 ndarray (updated in place)")
        _check(lib().gf_apply(self.h, which, int(bool(transpose)), _dp(x), x.size, _dp(y), y.size), ValueError)
        return y

    def apply_many(self, which, xs, ys, transpose=False):
        """Several products with one copy in / out per vector (gf_apply_many): transpose=False: ys[0] += sum_m A_m xs[m];
        transpose=True: ys[m] += A_m^T xs[0].  ys are contiguous float64 arrays updated in place."""
        which = [int(w) for w in which]
        xs = [np.ascontiguousarray(x, dtype=np.float64) for x in xs]
        for y in ys:
            if not (isinstance(y, np.ndarray) and y.dtype == np.float64 and y.flags.c_contiguous):
                raise ValueError("apply_many: ys must be contiguous float64 ndarrays (updated in place)")
        nrow = self.ndof
        for m, w in enumerate(which):
            ncol = self.ndof if w == MAT_K else self.total_cp
            x, y = (xs[0], ys[m]) if transpose else (xs[m], ys[0])
            if x.size != (nrow if transpose else ncol) or y.size != (ncol if transpose else nrow):
                raise ValueError("apply_many: vector lengths do not match the matrix shape")
        wa = (C.c_int * len(which))(*which)
        xa = (C.POINTER(C.c_double) * len(xs))(*[_dp(x) for x in xs])
        ya = (C.POINTER(C.c_double) * len(ys))(*[_dp(y) for y in ys])
        _check(lib().gf_apply_many(self.h, int(bool(transpose)), len(which), wa, xa, ya), ValueError)
        return ys

    def functionals(self, apply_bcs=True):
        """W_int, volume, penalty energy and their gradient fields (gf_functionals).  The values come back at once; a gradient
        field is copied from the device when it is first read (each is ndof or total_cp doubles over PCIe: a caller that wants
        dW/du only does not pay for the other four), or all remaining ones before the gradient buffer is reused."""
        self._materialize_lazy()
        out = np.zeros(3)
        null = C.POINTER(C.c_double)()
        _check(lib().gf_functionals(self.h, _dp(out), null, null, null, null, null, int(apply_bcs)))
        npatch, extra = int(self.arrays.n_patches), {}
        if hasattr(lib(), "gf_functionals_per_patch"):
            wp, vp_ = np.zeros(npatch), np.zeros(npatch)
            _check(lib().gf_functionals_per_patch(self.h, _dp(wp), _dp(vp_), npatch))
            extra = dict(Wint_patch=wp, volume_patch=vp_)
        g = _LazyFields(self, {"dWdu": (0, (self.ndof,)), "dWdcp": (1, (3, self.total_cp)), "dWdh": (2, (self.total_cp,)),
                               "dVdcp": (3, (3, self.total_cp)), "dVdh": (4, (self.total_cp,))},
                        dict(Wint=out[0], volume=out[1], Wpen=out[2], **extra))
        self._lazy_fun = weakref.ref(g)           # only a result somebody still holds has to be completed before the buffer is reused
        return g

    def _materialize_lazy(self):
        ref = getattr(self, "_lazy_fun", None)
        g = ref() if ref is not None else None
        if g is not None:
            g.materialize()
        self._lazy_fun = None

    def compliance(self, forces, apply_bcs=True):
        f = np.ascontiguousarray(forces, dtype=np.float64).ravel()
        out, dCdu, dCdcp = np.zeros(1), np.zeros(self.ndof), np.zeros((3, self.total_cp))
        self._materialize_lazy(); _check(lib().gf_compliance(self.h, _dp(f), f.size, _dp(out), _dp(dCdu), _dp(dCdcp), int(apply_bcs)), ValueError)
        return dict(C=out[0], dCdu=dCdu, dCdcp=dCdcp)

    def stress_forms(self, mode, rho, m_list, surf=1, measure=0, apply_bcs=True, gradients=True):
        """Per-patch von Mises aggregation forms and their gradients (gf_stress_forms)."""
        ml = np.ascontiguousarray(m_list, dtype=np.float64).ravel()
        I, vmax = np.zeros(ml.size), np.zeros(ml.size)
        g = dict(dIdu=np.zeros(self.ndof), dIdcp=np.zeros((3, self.total_cp)), dIdh=np.zeros(self.total_cp)) if gradients else \
            dict(dIdu=None, dIdcp=None, dIdh=None)
        self._materialize_lazy(); _check(lib().gf_stress_forms(self.h, int(mode), float(rho), _dp(ml), ml.size, int(surf), int(measure), _dp(I), _dp(vmax),
                                     _dp(g["dIdu"]), _dp(g["dIdcp"]), _dp(g["dIdh"]), int(apply_bcs)), ValueError)
        g.update(I=I, vmax=vmax)
        return g

    def shape_regu(self, field, cp0, coef):
        """Shape regularisation term and its gradient wrt the three homogeneous coordinate fields (gf_shape_regu)."""
        cp0 = np.ascontiguousarray(cp0, dtype=np.float64).ravel()
        coef = np.ascontiguousarray(coef, dtype=np.float64).ravel()
        val, dcp = np.zeros(1), np.zeros((3, self.total_cp))
        self._materialize_lazy(); _check(lib().gf_shape_regu(self.h, int(field), _dp(cp0), cp0.size, _dp(coef), coef.size, _dp(val), _dp(dcp)), ValueError)
        return dict(value=val[0], dcp=dcp)

    def penalty_dxi(self, npts, degree, v_first=0):
        """Per-vertex blocks of d(penalty residual)/d(xi, tau) and the support windows of the mortar vertices
        v_first .. v_first + npts - 1 (gf_penalty_dxi_range; the whole model with the defaults of the callers)."""
        nb = (degree + 1) ** 2
        blocks = np.zeros((npts, 6, 2, nb, 3))
        win = np.zeros((npts, 2, 2), dtype=np.int32)
        _check(lib().gf_penalty_dxi_range(self.h, int(v_first), int(npts), _dp(blocks), blocks.size, win.ctypes.data_as(C.POINTER(C.c_int32)), win.size), ValueError)
        return blocks, win

    def penalty_dxi_rev(self, npts, lam, v_first=0):
        """(npts, 6): the reverse-mode product of the per-vertex dR/d(xi, tau) blocks with lam, formed on the device (gf_penalty_dxi_rev)."""
        lam = np.ascontiguousarray(lam, dtype=np.float64).ravel()
        out = np.zeros((int(npts), 6))
        _check(lib().gf_penalty_dxi_rev(self.h, int(v_first), int(npts), _dp(lam), lam.size, _dp(out), out.size), ValueError)
        return out

    def update_interface(self, g, itf):
        """New parametric coordinates of interface ``g`` (a model.Interface): True when the vertex tables were patched in place (no vertex left its knot spans),
        False -- nothing changed -- when the model has to be re-created (gf_update_interface)."""
        xi = np.ascontiguousarray(np.hstack([itf.xi_a, itf.xi_b]), dtype=np.float64).ravel()
        tau, wt = np.ascontiguousarray(itf.tau, dtype=np.float64).ravel(), np.ascontiguousarray(itf.wt, dtype=np.float64).ravel()
        rc = lib().gf_update_interface(self.h, int(g), _dp(xi), _dp(tau), _dp(wt), wt.size)
        if rc == 2:
            return False
        _check(rc, ValueError)
        off = int(self.arrays.if_off[g])                      # keep the host copy of the model in step (a later re-creation starts from it)
        self.arrays.if_xi[4 * off:4 * off + xi.size] = xi
        self.arrays.if_tau[2 * off:2 * off + tau.size] = tau
        self.arrays.if_wt[off:off + wt.size] = wt
        return True

    def penalty_dxi_rev_if(self, g, lam):
        """penalty_dxi_rev for the mortar vertices of interface ``g`` of the model."""
        off = self.arrays.if_off
        return self.penalty_dxi_rev(int(off[g + 1] - off[g]), lam, v_first=int(off[g]))

    def k_values_ptr(self):
        """Device pointer of K's values (layout of pattern(MAT_K)): what goldfish_amd._solver.DeviceSolver factors in place."""
        return lib().gf_device_ptr(self.h, BUF_VAL_K)

    def kernel_ms(self):
        n = C.c_int(0)
        ms = lib().gf_kernel_ms(self.h, C.byref(n))
        return ms, n.value

    @property
    def stream_ptr(self):
        """hipStream_t of the handle (torch.cuda.ExternalStream(D.stream_ptr) orders torch work against the library's without a host sync)."""
        return lib().gf_stream(self.h)

    @property
    def assembly_path(self):
        """4 walking MFMA kernel + row records + record gather (default for p = 2, 3), 0 MFMA element-block kernel + gather (p = 4; GF_ASSEMBLY=block), 3 VALU element kernel + gather (GF_ELEMENT=valu)."""
        return lib().gf_assembly_path(self.h)

    @property
    def n_gauss_points(self):
        return lib().gf_num_gauss_points(self.h)

    @property
    def n_elements(self):
        return lib().gf_num_elements(self.h)

    @property
    def n_mortar_points(self):
        return lib().gf_num_mortar_points(self.h)

    @property
    def device_bytes(self):
        return lib().gf_device_bytes(self.h)
