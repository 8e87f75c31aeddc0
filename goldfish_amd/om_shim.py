"""Minimal stand-in for the part of ``openmdao.api`` the GOLDFISH components use
(OpenMDAO is not installed in the build image).  The components
in goldfish_amd/om_comps import the real ``openmdao.api`` when it is available and
this shim otherwise; the component code is identical in both cases.

Implemented: options.declare / [] access, add_input / add_output / declare_partials,
Problem(model=<single component>).setup / run_model / [] access, and a directional
``check_partials`` (analytic J.v by the component's own linearize / apply_linear or
compute_partials vs central finite differences)."""
import numpy as np


class _Options(dict):
    def declare(self, name, default=None, **kwargs):
        self.setdefault(name, default)


class _Vec(dict):
    """name -> ndarray with OpenMDAO-like assignment semantics (values copied into place)."""

    def __setitem__(self, k, v):
        if k in self:
            dict.__getitem__(self, k)[...] = np.asarray(v, float).reshape(dict.__getitem__(self, k).shape)
        else:
            dict.__setitem__(self, k, np.array(v, float))


class _Component:
    def __init__(self, **kwargs):
        self.options = _Options()
        self._in, self._out, self._partials = {}, {}, []
        self.initialize()
        for k, v in kwargs.items():
            self.options[k] = v

    def initialize(self):
        pass

    def setup(self):
        pass

    def add_input(self, name, val=1.0, shape=None, **kw):
        shape = (shape,) if isinstance(shape, (int, np.integer)) else shape
        arr = np.array(val, float)
        self._in[name] = np.broadcast_to(arr, shape).copy() if shape is not None else np.atleast_1d(arr).copy()

    def add_output(self, name, val=1.0, shape=None, **kw):
        shape = (shape,) if isinstance(shape, (int, np.integer)) else shape
        arr = np.array(val, float)
        self._out[name] = np.broadcast_to(arr, shape).copy() if shape is not None else np.atleast_1d(arr).copy()

    def declare_partials(self, of, wrt, **kw):
        self._partials.append((of, wrt, kw))


class _Partials(dict):
    """Sub-Jacobians as OpenMDAO keeps them: a pair declared with rows / cols holds its nnz values in that COO order (assigning
    anything of another size raises, as OpenMDAO does: 'wrong size, expected nnz'), a dense pair holds (n_of, n_wrt) values;
    declared ``val`` seeds the entry, so a component with constant declared partials needs no compute_partials."""

    def __init__(self, comp, inputs, outputs):
        super().__init__()
        self._decl, self._shape = {}, {}
        for of, wrt, kw in comp._partials:
            ofs = list(outputs) if of == "*" else [of]
            wrts = (list(inputs) + list(outputs)) if wrt == "*" else [wrt]
            for o in ofs:
                for w in wrts:
                    self._decl[(o, w)] = kw
                    self._shape[(o, w)] = (outputs[o].size, (inputs[w] if w in inputs else outputs[w]).size)
                    if kw.get("val") is not None:
                        self[(o, w)] = kw["val"]

    def __setitem__(self, key, val):
        kw = self._decl.get(key)
        val = np.asarray(val, float)
        if kw is not None and kw.get("rows") is not None:
            nnz = len(kw["rows"])
            if val.size != nnz and val.size != 1:
                raise ValueError("partials%s: sub-jacobian declared with rows/cols has wrong size %d, expected nnz = %d" % (key, val.size, nnz))
        elif key in self._shape and val.size not in (1, self._shape[key][0] * self._shape[key][1]):
            raise ValueError("partials%s: wrong size %d for a dense (%d, %d) sub-jacobian" % ((key, val.size) + self._shape[key]))
        dict.__setitem__(self, key, val)

    def dense(self, of, wrt):
        kw, val = self._decl.get((of, wrt)), self[(of, wrt)]
        shape = self._shape.get((of, wrt))
        if kw is not None and kw.get("rows") is not None:
            J = np.zeros(shape)
            np.add.at(J, (np.asarray(kw["rows"]), np.asarray(kw["cols"])), np.broadcast_to(val.ravel(), (len(kw["rows"]),)))
            return J
        return np.broadcast_to(val, shape) if (shape is not None and val.size == 1) else (val.reshape(shape) if shape is not None else val)


class ImplicitComponent(_Component):
    pass


class ExplicitComponent(_Component):
    pass


class Problem:
    def __init__(self, model=None):
        self.model = model

    def setup(self):
        m = self.model
        m.setup()
        self.inputs = _Vec({k: v.copy() for k, v in m._in.items()})
        self.outputs = _Vec({k: v.copy() for k, v in m._out.items()})
        self.residuals = _Vec({k: np.zeros_like(v) for k, v in m._out.items()})

    def __getitem__(self, name):
        return self.inputs[name] if name in self.inputs else self.outputs[name]

    def __setitem__(self, name, val):
        (self.inputs if name in self.inputs else self.outputs)[name] = val

    def run_model(self):
        m = self.model
        if isinstance(m, ImplicitComponent):
            m.solve_nonlinear(self.inputs, self.outputs)
        else:
            m.compute(self.inputs, self.outputs)

    def check_partials(self, step=1e-6, seed=0, compact_print=True, free_mask=None):
        """Relative error of analytic vs FD directional derivatives, per (of, wrt).
        ``free_mask`` (implicit components): boolean mask of non-Dirichlet state dofs; the
        reference's Dirichlet conventions (unit diagonal in K, untreated rows in dR/dh,
        GOLDFISH/nonmatching_opt.py:660-724, 1006-1015) differ from d(residual) on those rows."""
        m, rng = self.model, np.random.default_rng(seed)
        res = {}
        if isinstance(m, ImplicitComponent):
            of = list(self.outputs)[0]
            m.linearize(self.inputs, self.outputs, None)
            for wrt in list(self.inputs) + [of]:
                tgt = self.inputs if wrt in self.inputs else self.outputs
                base = tgt[wrt].copy()
                v = rng.standard_normal(base.shape)
                if free_mask is not None and wrt == of:
                    v = v * free_mask
                d_in = _Vec({k: np.zeros_like(x) for k, x in self.inputs.items()})
                d_out = _Vec({k: np.zeros_like(x) for k, x in self.outputs.items()})
                d_res = _Vec({k: np.zeros_like(x) for k, x in self.outputs.items()})
                (d_in if wrt in self.inputs else d_out)[wrt] = v
                m.apply_linear(self.inputs, self.outputs, d_in, d_out, d_res, 'fwd')
                an = d_res[of].copy()
                r = []
                for sgn in (1, -1):
                    tgt[wrt] = base + sgn * step * v
                    m.apply_nonlinear(self.inputs, self.outputs, self.residuals)
                    r.append(self.residuals[of].copy())
                tgt[wrt] = base
                fd = (r[0] - r[1]) / (2 * step)
                if free_mask is not None:
                    fd, an = fd[free_mask], an[free_mask]
                res[(of, wrt)] = np.abs(fd - an).max() / max(np.abs(fd).max(), 1e-300)
        else:
            partials = _Partials(m, self.inputs, self.outputs)
            m.compute(self.inputs, self.outputs)
            if hasattr(m, "compute_partials"):
                m.compute_partials(self.inputs, partials)
            for (of, wrt) in list(partials):
                Jm = partials.dense(of, wrt)
                base = self.inputs[wrt].copy()
                v = rng.standard_normal(base.shape)
                if free_mask is not None and base.size == np.size(free_mask):      # gradients with Dirichlet rows zeroed (apply_bcs=True)
                    v = v * np.reshape(free_mask, base.shape)
                an = np.asarray(Jm).reshape(self.outputs[of].size, base.size) @ v.ravel()
                f = []
                for sgn in (1, -1):
                    self.inputs[wrt] = base + sgn * step * v
                    m.compute(self.inputs, self.outputs)
                    f.append(self.outputs[of].copy().ravel())
                self.inputs[wrt] = base
                fd = (f[0] - f[1]) / (2 * step)
                res[(of, wrt)] = np.abs(fd - an).max() / max(np.abs(fd).max(), 1e-300)
            m.compute(self.inputs, self.outputs)
        if compact_print:
            for k, e in res.items():
                print("check_partials %s wrt %s: rel err %.3e" % (k[0], k[1], e))
        return res
