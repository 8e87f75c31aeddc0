"""IntEnergyExOperation -- internal (strain) energy of the non-matching structure and its
partials (reference: GOLDFISH/operations/int_energy_exop.py:3-107).  One device pass
(gf_functionals) produces the value and every gradient."""
import numpy as np


class ShapeRegu(object):
    """One patch's ``wint_regu`` term (int_energy_exop.py:15-18, 31-32 adds an arbitrary UFL form per patch to the energy form; the one the
    reference's demos build is the shape regularisation of demos_om/shape_opt/eVTOL/int_energy_regu_exop.py:30-38):

        coef * int |grad_s(P_field - P_field^0)|^2 dA

    with the surface gradient on the current geometry, P_field the homogeneous coordinate ``field`` of the control net and P^0 its value when
    the operation is created (or ``cp0``, an array over the patch's control points).  Device: gf_shape_regu -> kl_pointfun_kernel<P, 2>."""

    def __init__(self, coef, field=2, cp0=None):
        self.coef, self.field = float(coef), int(field)
        self.cp0 = None if cp0 is None else np.asarray(cp0, float).ravel()
        if self.field not in (0, 1, 2):
            raise ValueError("ShapeRegu: field must be 0, 1 or 2")


def regu_fields(nm, terms, what):
    """Per-patch list of None / ShapeRegu (the ``wint_regu`` of IntEnergyExOperation, the ``c_regu`` of ComplianceExOperation) -> {field: (per-patch coefficients,
    initial coordinate field)}: what one gf_shape_regu evaluation per regularised field takes."""
    terms = [None] * nm.num_splines if terms is None else list(terms)
    if len(terms) != nm.num_splines:
        raise ValueError("%s: one entry (None or a ShapeRegu) per patch" % what)
    fields = {}
    for s, r in enumerate(terms):
        if r is None:
            continue
        if not isinstance(r, ShapeRegu):
            raise TypeError("%s[%d]: the reference adds a UFL form here; the device path evaluates goldfish_amd.operations."
                            "int_energy_exop.ShapeRegu terms (the regularisation of the reference's eVTOL demo), got %r" % (what, s, type(r)))
        coef, cp0 = fields.setdefault(r.field, (np.zeros(nm.num_splines), nm.cp_iga[r.field].copy()))
        coef[s] = r.coef
        if r.cp0 is not None:
            if r.cp0.size != nm.vec_scalar_iga_dof_list[s]:
                raise ValueError("%s[%d].cp0: expected %d values" % (what, s, nm.vec_scalar_iga_dof_list[s]))
            cp0[nm.cp_off[s]:nm.cp_off[s + 1]] = r.cp0
    return terms, fields


def regu_values(nm, fields, tag):
    """[(value, dcp (3, total_cp))] of the regularisation terms, one device evaluation per regularised coordinate field and state."""
    return [nm._cached((tag, f, coef.tobytes(), cp0.tobytes()), lambda f=f, coef=coef, cp0=cp0: nm.dev.shape_regu(f, cp0, coef))
            for f, (coef, cp0) in sorted(fields.items())]


class IntEnergyExOperation(object):

    def __init__(self, nonmatching_opt, wint_regu=None):
        self.nonmatching_opt = nm = nonmatching_opt
        self.num_splines = nonmatching_opt.num_splines
        self.splines = nonmatching_opt.splines
        self.opt_shape = nonmatching_opt.opt_shape
        self.opt_thickness = nonmatching_opt.opt_thickness
        self.wint_regu, self._regu_fields = regu_fields(nm, wint_regu, "wint_regu")
        if self.opt_shape:
            self.opt_field = nonmatching_opt.opt_field
            self.shopt_surf_inds = nonmatching_opt.shopt_surf_inds

    def _f(self, apply_bcs=True):
        return self.nonmatching_opt.functionals(apply_bcs=apply_bcs)

    def _regu(self):
        """[(value, dcp (3, total_cp))] of the regularisation terms, one device evaluation per regularised coordinate field and state."""
        return regu_values(self.nonmatching_opt, self._regu_fields, "wint_regu")

    def Wint(self):
        """int_energy_exop.py:55-59."""
        return float(self._f()["Wint"]) + float(sum(r["value"] for r in self._regu()))

    def dWintduIGA(self, array=True, apply_bcs=True):
        """int_energy_exop.py:61-73 (Dirichlet rows zeroed by FE2IGA(..., apply_bcs)).  The regularisation does not depend on u."""
        return self._f(apply_bcs)["dWdu"]

    def dWintdCPIGA(self, field, array=True):
        """int_energy_exop.py:75-90."""
        nm = self.nonmatching_opt
        cols = nm._shopt_cols[self.opt_field.index(field)]
        g = self._f()["dWdcp"][field][cols]
        for r in self._regu():
            g = g + r["dcp"][field][cols]
        return g

    def dWintdh_th(self, extract=False, array=True):
        """int_energy_exop.py:92-107.  The regularisation does not depend on the thickness."""
        nm = self.nonmatching_opt
        g = self._f()["dWdh"]
        return g if nm.var_thickness else np.add.reduceat(g, nm.cp_off[:-1])
