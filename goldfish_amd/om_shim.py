"""Minimal stand-in for the part of ``openmdao.api`` the GOLDFISH components use
(OpenMDAO is not installed in the build image).  The components
in goldfish_amd/om_comps import the real ``openmdao.api`` when it is available and
this shim otherwise; the component code is identical in both cases.

Implemented: options.declare / [] access, add_input / add_output / declare_partials, IndepVarComp, Group (add_subsystem,
connect, add_design_var / add_constraint / add_objective), Problem(model=<component or group>).setup / run_model / [] /
get_val / set_val / compute_totals (reverse mode: explicit components through their sub-Jacobians, implicit ones through
solve_linear + apply_linear in 'rev' mode), and a directional ``check_partials`` (analytic J.v by the component's own linearize /
apply_linear or compute_partials vs central finite differences).

The shim is deliberately as strict as OpenMDAO where a component could otherwise pass here and fail there: a sub-Jacobian that was
not declared cannot be set (KeyError), a value of the wrong size cannot be assigned to a variable or a sub-Jacobian (ValueError),
``connect`` refuses unknown variables and mismatched shapes, ``apply_linear`` must ACCUMULATE (check_partials calls it on pre-filled
vectors, in both modes, and checks <J v, w> = <v, J^T w>), and ``solve_linear`` must invert what ``apply_linear`` applies to the state."""
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
        if kw is None:
            raise KeyError("partials%s: the sub-jacobian of this variable pair was not declared (declare_partials)" % (key,))
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


class IndepVarComp(ExplicitComponent):
    """Source of independent variables (``add_output`` before setup, as in the demos' ``inputs_comp``)."""

    def compute(self, inputs, outputs):
        pass


class Group:
    """``om.Group``: subsystems in execution order, explicit connections ``comp.var -> comp.var`` (no promotion: the demos connect by
    absolute names, demos_om/thickness_opt/plate/plate_const_th_opt_wint.py:89-110), design variables / responses recorded."""

    def __init__(self, **kwargs):
        self.options = _Options()
        self._subs, self._conns, self.design_vars, self.constraints, self.objectives = [], [], {}, {}, {}
        self.initialize()
        for k, v in kwargs.items():
            self.options[k] = v

    def initialize(self):
        pass

    def setup(self):
        pass

    def add_subsystem(self, name, subsys, **kw):
        if any(n == name for n, _ in self._subs):
            raise ValueError("add_subsystem: a subsystem named '%s' exists already" % name)
        self._subs.append((name, subsys))
        return subsys

    def connect(self, src, tgt):
        self._conns.append((src, tgt))

    def add_design_var(self, name, **kw):
        self.design_vars[name] = kw

    def add_constraint(self, name, **kw):
        self.constraints[name] = kw

    def add_objective(self, name, **kw):
        self.objectives[name] = kw


class Problem:
    def __init__(self, model=None):
        self.model = model

    def setup(self):
        m = self.model
        m.setup()
        if isinstance(m, Group):
            self._setup_group(m)
            return
        self.inputs = _Vec({k: v.copy() for k, v in m._in.items()})
        self.outputs = _Vec({k: v.copy() for k, v in m._out.items()})
        self.residuals = _Vec({k: np.zeros_like(v) for k, v in m._out.items()})

    # ---- groups ------------------------------------------------------------------------------------------------------
    def _setup_group(self, g):
        self._comps = {}
        for name, c in g._subs:
            c.setup()
            self._comps[name] = dict(comp=c, inputs=_Vec({k: v.copy() for k, v in c._in.items()}), outputs=_Vec({k: v.copy() for k, v in c._out.items()}),
                                     residuals=_Vec({k: np.zeros_like(v) for k, v in c._out.items()}))
        self._order = [n for n, _ in g._subs]
        self._src = {}                                        # (comp, input) -> (comp, output)
        for src, tgt in g._conns:
            (sc, sv), (tc, tv) = src.split(".", 1), tgt.split(".", 1)
            if sc not in self._comps or sv not in self._comps[sc]["outputs"]:
                raise NameError("connect: output '%s' does not exist" % src)
            if tc not in self._comps or tv not in self._comps[tc]["inputs"]:
                raise NameError("connect: input '%s' does not exist" % tgt)
            if self._comps[sc]["outputs"][sv].shape != self._comps[tc]["inputs"][tv].shape:
                raise ValueError("connect: shapes of '%s' %s and '%s' %s differ" % (src, self._comps[sc]["outputs"][sv].shape, tgt, self._comps[tc]["inputs"][tv].shape))
            if self._order.index(sc) >= self._order.index(tc):
                raise ValueError("connect: '%s' -> '%s' runs against the execution order (feedback needs a solver; the demos have none)" % (src, tgt))
            if (tc, tv) in self._src:
                raise ValueError("connect: input '%s' is already connected" % tgt)
            self._src[(tc, tv)] = (sc, sv)

    def _find(self, name):
        c, v = name.split(".", 1)
        d = self._comps[c]
        return d["inputs"] if v in d["inputs"] else d["outputs"], v

    def get_val(self, name):
        if not isinstance(self.model, Group):
            return self[name]
        vec, v = self._find(name)
        return vec[v]

    def set_val(self, name, val):
        if not isinstance(self.model, Group):
            self[name] = val
            return
        vec, v = self._find(name)
        vec[v] = val

    def _transfer(self, name):
        d = self._comps[name]
        for k in d["inputs"]:
            if (name, k) in self._src:
                sc, sv = self._src[(name, k)]
                d["inputs"][k] = self._comps[sc]["outputs"][sv]

    def _run_group(self):
        for name in self._order:
            d = self._comps[name]
            self._transfer(name)
            c = d["comp"]
            if isinstance(c, ImplicitComponent):
                c.solve_nonlinear(d["inputs"], d["outputs"])
            else:
                c.compute(d["inputs"], d["outputs"])

    def compute_totals(self, of, wrt):
        """Reverse-mode total derivatives d(of) / d(wrt) of the current point (``run_model`` first): {(of, wrt): array (n_of, n_wrt)}.
        Explicit components contribute J^T seeds through their (declared / computed) sub-Jacobians; an implicit component first solves
        its adjoint (``solve_linear`` 'rev': d_residuals = (dR/dy)^-T d_outputs), then ``apply_linear`` 'rev' accumulates (dR/dx)^T of it."""
        if not isinstance(self.model, Group):
            raise TypeError("compute_totals needs a Group model")
        jacs = {}
        for name in self._order:                              # linearise every component at the current point
            d = self._comps[name]
            c = d["comp"]
            if isinstance(c, ImplicitComponent):
                c.linearize(d["inputs"], d["outputs"], None)
            elif not isinstance(c, IndepVarComp):
                P = _Partials(c, d["inputs"], d["outputs"])
                if hasattr(c, "compute_partials"):
                    c.compute_partials(d["inputs"], P)
                jacs[name] = P
        out = {}
        for o in of:
            oc, ov = o.split(".", 1)
            n_of = self._comps[oc]["outputs"][ov].size
            rows = {w: np.zeros((n_of, self._find(w)[0][w.split(".", 1)[1]].size)) for w in wrt}
            for r in range(n_of):
                bar = {(n, k): np.zeros(v.size) for n in self._order for k, v in self._comps[n]["outputs"].items()}
                bar[(oc, ov)][r] = 1.0
                for name in reversed(self._order):
                    d = self._comps[name]
                    c = d["comp"]
                    ybar = {k: bar[(name, k)] for k in d["outputs"]}
                    if not any(np.any(v) for v in ybar.values()) or isinstance(c, IndepVarComp):
                        continue
                    xbar = {k: np.zeros(v.size) for k, v in d["inputs"].items()}
                    if isinstance(c, ImplicitComponent):
                        d_out = _Vec({k: ybar[k].reshape(d["outputs"][k].shape).copy() for k in d["outputs"]})
                        d_res = _Vec({k: np.zeros_like(v) for k, v in d["outputs"].items()})
                        c.solve_linear(d_out, d_res, 'rev')
                        d_in = _Vec({k: np.zeros_like(v) for k, v in d["inputs"].items()})
                        d_o2 = _Vec({k: np.zeros_like(v) for k, v in d["outputs"].items()})
                        c.apply_linear(d["inputs"], d["outputs"], d_in, d_o2, d_res, 'rev')
                        for k in xbar:
                            xbar[k] = -np.asarray(d_in[k]).ravel()         # R(x, y) = 0: dy/dx = -(dR/dy)^-1 dR/dx
                    else:
                        P = jacs[name]
                        for (po, pw) in list(P):
                            if pw in xbar and po in ybar:
                                xbar[pw] += P.dense(po, pw).reshape(d["outputs"][po].size, d["inputs"][pw].size).T @ ybar[po]
                    for k, v in xbar.items():
                        if (name, k) in self._src:
                            bar[self._src[(name, k)]] += v
                for w in wrt:
                    wc, wv = w.split(".", 1)
                    rows[w][r] = bar[(wc, wv)]
            for w in wrt:
                out[(o, w)] = rows[w]
        return out

    # ---- single components ---------------------------------------------------------------------------------------------
    def __getitem__(self, name):
        if isinstance(self.model, Group):
            return self.get_val(name)
        return self.inputs[name] if name in self.inputs else self.outputs[name]

    def __setitem__(self, name, val):
        if isinstance(self.model, Group):
            self.set_val(name, val)
            return
        (self.inputs if name in self.inputs else self.outputs)[name] = val

    def run_model(self):
        m = self.model
        if isinstance(m, Group):
            self._run_group()
        elif isinstance(m, ImplicitComponent):
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
                pre = rng.standard_normal(d_res[of].shape)
                d_res[of] = pre                               # apply_linear must ACCUMULATE into d_residuals (OpenMDAO hands it a shared vector)
                m.apply_linear(self.inputs, self.outputs, d_in, d_out, d_res, 'fwd')
                an = d_res[of].copy() - pre
                # reverse mode on pre-filled vectors: <J v, w> = <v, J^T w>
                w = rng.standard_normal(d_res[of].shape)
                r_in = _Vec({k: rng.standard_normal(x.shape) for k, x in self.inputs.items()})
                r_out = _Vec({k: rng.standard_normal(x.shape) for k, x in self.outputs.items()})
                pre_t = (r_in if wrt in self.inputs else r_out)[wrt].copy()
                m.apply_linear(self.inputs, self.outputs, r_in, r_out, _Vec({of: w.copy()}), 'rev')
                jt_w = (r_in if wrt in self.inputs else r_out)[wrt] - pre_t
                lhs, rhs = float(np.vdot(an, w)), float(np.vdot(v, jt_w))
                if abs(lhs - rhs) > 1e-9 * max(abs(lhs), abs(rhs), 1e-300):
                    raise AssertionError("apply_linear: 'rev' is not the transpose of 'fwd' for (%s, %s) (or it does not accumulate): %.12e vs %.12e" % (of, wrt, lhs, rhs))
                if wrt == of and hasattr(m, "solve_linear"):  # solve_linear inverts what apply_linear applies to the state
                    s_out = _Vec({of: np.zeros_like(an)})
                    m.solve_linear(s_out, _Vec({of: an.copy()}), 'fwd')
                    if np.abs(s_out[of] - v).max() > 1e-6 * max(np.abs(v).max(), 1e-300):
                        raise AssertionError("solve_linear 'fwd' does not invert apply_linear on the state (%.3e)" % (np.abs(s_out[of] - v).max() / np.abs(v).max()))
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
