"""MaxvMStressExOperation -- aggregated maximum von Mises stress of the non-matching structure and its
partials (reference: GOLDFISH/operations/max_vmstress_exop.py:3-440; same constructor arguments, attribute
and method names).

Per patch s the reference assembles the form  I_s = int g(sigma_vM) dA  (KS_symexp :167, pnorm_symexp :170,
induced_power :173), turns it into a local maximum (continuous_*_function :188-219) and aggregates the local
maxima with the discrete version of the same function (discrete_*_function :233-265).  Here one device pass
(gf_stress_forms -> kl_pointfun_kernel<P, 1>) returns all I_s, the largest Gauss-point stress of every patch and the
gradient fields of the forms; the chain rule through the two aggregation levels (:330-440) is host arithmetic
on n_patches numbers.

sigma_vM is PENGoLINS' ``ShellStressSVK(...).vonMisesStress(xi2)`` (:38-47); PENGoLINS is not vendored, see
DESIGN.md "N4" for the restated definition (``stress_measure``: "cauchy" (default) or "pk2").
Differences from the reference, both documented there: ``m`` / ``compute_max_vM`` use the largest Gauss-point
value of a patch where the reference L2-projects onto linears first (:147-165); ``linearize_stress`` is not
available.
"""
import numpy as np

_SURF = {"top": 1, "bottom": -1, "middle": 0}
_MEASURE = {"cauchy": 0, "pk2": 1}


class MaxvMStressExOperation(object):

    def __init__(self, nonmatching_opt, rho=1., alpha=None, m=None, surf="top", method="pnorm",
                 linearize_stress=False, stress_measure="cauchy"):
        if surf not in _SURF:
            raise ValueError("Unknown surface type:", surf)                       # max_vmstress_exop.py:35-36
        if method not in ("KS", "pnorm", "induced power"):
            raise ValueError("Unsupported max stress method " + method)           # :184
        if linearize_stress:
            raise NotImplementedError("linearize_stress=True (linearised strains in ShellStressSVK) is not available")
        self.nonmatching_opt = nonmatching_opt
        self.num_splines = nonmatching_opt.num_splines
        self.splines = nonmatching_opt.splines
        self.opt_field = nonmatching_opt.opt_field
        self.opt_shape = nonmatching_opt.opt_shape
        self.opt_thickness = nonmatching_opt.opt_thickness
        self.surf, self.rho, self.method = surf, float(rho), method
        self.linearize_stress = linearize_stress
        self.stress_measure = stress_measure
        self._measure = _MEASURE[stress_measure]
        if alpha is not None:
            self.alpha = alpha
        else:
            self.compute_alpha()
        if m is not None:
            self.given_m = True
            self.m = m
            self.m_list = [m for _ in range(self.num_splines)]
        else:
            self.given_m = False
            self.compute_m()

    # ---- device pass -------------------------------------------------------------------------
    def _forms(self, exponent=None, gradients=False, apply_bcs=True):
        mode = 0 if self.method == "KS" else 1
        rho = self.rho if exponent is None else exponent
        return self.nonmatching_opt.stress_forms(mode, rho, self.m_list, _SURF[self.surf], self._measure,
                                                     apply_bcs=apply_bcs, gradients=gradients)

    def _gp_max(self):
        return self.nonmatching_opt.stress_forms(1, 1.0, np.ones(self.num_splines), _SURF[self.surf], self._measure,
                                                     gradients=False)["vmax"]

    # ---- setup -------------------------------------------------------------------------------
    def compute_alpha(self):
        """Smallest patch-average parametric cell area (max_vmstress_exop.py:139-145)."""
        cell = []
        for P in self.splines:
            ku, kv = np.unique(P.knots[0]), np.unique(P.knots[1])
            cell.append((ku[-1] - ku[0]) * (kv[-1] - kv[0]) / ((ku.size - 1) * (kv.size - 1)))
        self.alpha = float(np.min(cell))
        return self.alpha

    def compute_m(self):
        """Per-patch and global normalisation stresses (:147-155)."""
        vm = self._gp_max()
        self.m_list = [float(max(v, 1e-300)) for v in vm]
        self.m = float(np.max(self.m_list))
        return self.m_list, self.m

    def compute_max_vM(self):
        """:157-165."""
        self.max_vM_sub_proj = [float(v) for v in self._gp_max()]
        self.max_global = float(np.max(self.max_vM_sub_proj))
        return self.max_vM_sub_proj, self.max_global

    # ---- local (per patch) aggregation of a form value ----------------------------------------
    @staticmethod
    def _clip(val):
        if val == 0:
            return 1e-15
        if val == np.inf:
            return 1e15
        return val

    def continuous_KS_function(self, KS_val, ind):
        """:188-194."""
        return self.m_list[ind] + 1 / self.rho * np.log(1 / self.alpha * self._clip(KS_val))

    def continuous_pnorm_function(self, pnorm_val, ind):
        """:196-202."""
        return self.m_list[ind] * (1 / self.alpha * self._clip(pnorm_val)) ** (1 / self.rho)

    def continuous_induced_power_function(self, ip_val, ind):
        """:204-217."""
        return self.m_list[ind] * self._clip(ip_val[0]) / self._clip(ip_val[1])

    def continuous_max_vM_stress(self, form_val, ind):
        """:219-231; ``form_val`` is the assembled form (a pair for 'induced power')."""
        if self.method == "KS":
            return self.continuous_KS_function(form_val, ind)
        if self.method == "pnorm":
            return self.continuous_pnorm_function(form_val, ind)
        return self.continuous_induced_power_function(form_val, ind)

    # ---- global aggregation of the local maxima ------------------------------------------------
    def discrete_KS_function(self, stress_list):
        """:233-238."""
        s = np.sum(np.exp(self.rho * (np.asarray(stress_list) - self.m)))
        return self.m + 1 / self.rho * np.log(1 / self.alpha * s)

    def discrete_pnorm_function(self, stress_list):
        """:240-245."""
        s = np.sum((np.asarray(stress_list) / self.m) ** self.rho)
        return self.m * (1 / self.alpha * s) ** (1 / self.rho)

    def discrete_induced_power_function(self, stress_list):
        """:247-251."""
        x = np.asarray(stress_list) / self.m
        return self.m * np.sum(x ** (self.rho + 1)) / np.sum(x ** self.rho)

    def discrete_max_vM_stress(self, stress_list):
        """:253-265."""
        if self.method == "KS":
            return self.discrete_KS_function(stress_list)
        if self.method == "pnorm":
            return self.discrete_pnorm_function(stress_list)
        return self.discrete_induced_power_function(stress_list)

    def _form_values(self, gradients=False, apply_bcs=True):
        """Assembled forms of every patch (and their gradient fields)."""
        if self.method == "induced power":
            num = self._forms(self.rho + 1, gradients, apply_bcs)
            den = self._forms(self.rho, gradients, apply_bcs)
            return [(num["I"][s], den["I"][s]) for s in range(self.num_splines)], (num, den)
        f = self._forms(None, gradients, apply_bcs)
        return list(f["I"]), (f,)

    def max_vM_sub(self):
        vals, _ = self._form_values()
        return [self.continuous_max_vM_stress(vals[s], s) for s in range(self.num_splines)]

    def max_vM_stress_global(self):
        """:267-280."""
        if not self.given_m:
            self.compute_m()
        self.compute_max_vM()
        return float(self.discrete_max_vM_stress(self.max_vM_sub()))

    # ---- chain-rule factors (:282-328) ------------------------------------------------------------
    def dglobal_KSdlocal_KS(self, stress_list, ind):
        e = np.exp(self.rho * (np.asarray(stress_list) - self.m))
        return e[ind] / np.sum(e)

    def dlocal_KSdKS_form(self, form_vals, ind):
        return 1. / (self.rho * form_vals[ind])

    def dglobal_pnormdlocal_prnom(self, stress_list, ind):
        x = np.asarray(stress_list) / self.m
        return 1. / self.alpha * (1 / self.alpha * np.sum(x ** self.rho)) ** (1 / self.rho - 1) * x[ind] ** (self.rho - 1)

    def dlocal_pnormdpnorm_form(self, form_vals, ind):
        return self.m_list[ind] * (1 / self.alpha) ** (1 / self.rho) * (1 / self.rho) * form_vals[ind] ** (1 / self.rho - 1)

    def dglobal_induced_powerdlocal_induced_power(self, stress_list, ind):
        x = np.asarray(stress_list) / self.m
        num, den = np.sum(x ** (self.rho + 1)), np.sum(x ** self.rho)
        dnum = 1 / self.m * (self.rho + 1) * x[ind] ** self.rho
        dden = 1 / self.m * self.rho * x[ind] ** (self.rho - 1)
        return self.m * (dnum * den - num * dden) / den ** 2

    def _patch_factors(self, vals):
        """d(global max)/d(form of patch s) for single-form methods; for 'induced power' the pair of factors
        multiplying the gradients of the numerator and denominator forms (:344-352)."""
        sub = [self.continuous_max_vM_stress(vals[s], s) for s in range(self.num_splines)]
        out = []
        for s in range(self.num_splines):
            if self.method == "KS":
                out.append((self.dglobal_KSdlocal_KS(sub, s) * self.dlocal_KSdKS_form(vals, s),))
            elif self.method == "pnorm":
                out.append((self.dglobal_pnormdlocal_prnom(sub, s) * self.dlocal_pnormdpnorm_form(vals, s),))
            else:
                a = self.dglobal_induced_powerdlocal_induced_power(sub, s) * self.m_list[s]
                n, d = vals[s]
                out.append((a / d, -a * n / d ** 2))
        return out

    def _global_gradient(self, key, apply_bcs=True):
        nm = self.nonmatching_opt
        vals, fields = self._form_values(gradients=True, apply_bcs=apply_bcs)
        fac = self._patch_factors(vals)
        ncp = np.diff(nm.cp_off)
        total = None
        for k, f in enumerate(fields):
            w = np.repeat([fac[s][k] for s in range(self.num_splines)], ncp)        # one factor per control point
            g = f[key]
            if key == "dIdu":
                g = g * np.repeat(w, 3)
            else:
                g = g * w
            total = g if total is None else total + g
        return total

    def dmax_vMduIGA_global(self, array=True, apply_bcs=True):
        """:330-365."""
        return self._global_gradient("dIdu", apply_bcs)

    def dmax_vMdCPIGA_global(self, field, array=True):
        """:367-401."""
        nm = self.nonmatching_opt
        return self._global_gradient("dIdcp")[field][nm._shopt_cols[self.opt_field.index(field)]]

    def dmax_vMdh_th_global(self, array=True):
        """:403-440."""
        nm = self.nonmatching_opt
        g = self._global_gradient("dIdh")
        return g if nm.var_thickness else np.add.reduceat(g, nm.cp_off[:-1])
