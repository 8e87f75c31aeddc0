"""IntEnergyReguExOperation -- internal energy plus a shape regularisation term
(reference: demos_om/shape_opt/eVTOL/int_energy_regu_exop.py:3-120, a demo-level operation of the wing shape + thickness
optimisation; same constructor arguments and method names):

    W = W_int + sum_s c_s int |grad_s(P_f - P_f^0)|^2 dA,   c_s = regu_para E_s h0^3 / (12 ha_s (1 - nu_s^2)),

grad_s = surface gradient on the current geometry (tIGAr ``spline.grad``), P_f the homogeneous coordinate ``regu_field``
(2 there) of the control net, P_f^0 its initial value, h0 = 1e-3 as in the demo.  ``ha_s`` is the mean physical element
area of patch s here (the reference uses the element-area field projected onto linears, nonmatching_opt.py:112-127).
Device: gf_shape_regu -> kl_pointfun_kernel<P, 2>."""
import numpy as np

from .int_energy_exop import IntEnergyExOperation


def mean_element_area(P, n=5):
    """Physical area of a patch / number of elements (Gauss quadrature of |X_u x X_v| per element)."""
    ku, kv = np.unique(P.knots[0]), np.unique(P.knots[1])
    gx, gw = np.polynomial.legendre.leggauss(n)
    area = 0.0
    for a0, a1 in zip(ku[:-1], ku[1:]):
        for b0, b1 in zip(kv[:-1], kv[1:]):
            for x, wx in zip(gx, gw):
                for y, wy in zip(gx, gw):
                    _, Xu, Xv = P.eval_ders((0.5 * (a0 + a1) + 0.5 * (a1 - a0) * x, 0.5 * (b0 + b1) + 0.5 * (b1 - b0) * y))
                    area += 0.25 * (a1 - a0) * (b1 - b0) * wx * wy * np.linalg.norm(np.cross(Xu, Xv))
    return area / ((ku.size - 1) * (kv.size - 1))


class IntEnergyReguExOperation(IntEnergyExOperation):

    def __init__(self, nonmatching_opt, regu_para, regu_field=2, init_h_th=1e-3):
        super().__init__(nonmatching_opt)
        nm = nonmatching_opt
        self.regu_para, self.regu_field, self.init_h_th = float(regu_para), int(regu_field), float(init_h_th)
        E, nu = np.broadcast_to(np.asarray(nm.E, float), (nm.num_splines,)), np.broadcast_to(np.asarray(nm.nu, float), (nm.num_splines,))
        self.ha_phy = np.array([mean_element_area(P) for P in nm.splines])
        self.regu_para_full = self.regu_para * E * self.init_h_th ** 3 / (12.0 * self.ha_phy * (1.0 - nu ** 2))
        self.init_cp = nm.cp_iga[self.regu_field].copy()                 # init_cpfuncs_list[s][regu_field]

    def _r(self):
        nm = self.nonmatching_opt
        return nm._cached(("shape_regu", self.regu_field, self.regu_para), lambda: nm.dev.shape_regu(self.regu_field, self.init_cp, self.regu_para_full))

    def Wint(self):
        """int_energy_regu_exop.py:62-66."""
        return super().Wint() + float(self._r()["value"])

    def dWintdCPIGA(self, field, array=True):
        """int_energy_regu_exop.py:80-97."""
        nm = self.nonmatching_opt
        return super().dWintdCPIGA(field) + self._r()["dcp"][field][nm._shopt_cols[self.opt_field.index(field)]]
