"""VolumeExOperation -- material volume int h dA and its partials
(reference: GOLDFISH/operations/volume_exop.py:3-84)."""
import numpy as np


class VolumeExOperation(object):

    def __init__(self, nonmatching_opt, vol_surf_inds=None):
        self.nonmatching_opt = nonmatching_opt
        self.num_splines = nonmatching_opt.num_splines
        self.splines = nonmatching_opt.splines
        self.opt_shape = nonmatching_opt.opt_shape
        self.opt_thickness = nonmatching_opt.opt_thickness
        # volume_exop.py:15-26: the patches outside vol_surf_inds contribute Constant(0) * dx -- to the value and to every partial
        self.vol_surf_inds = list(range(self.num_splines)) if vol_surf_inds is None else [int(s) for s in vol_surf_inds]
        if any(s < 0 or s >= self.num_splines for s in self.vol_surf_inds):
            raise ValueError("vol_surf_inds: patch index out of range")
        self._subset = sorted(set(self.vol_surf_inds)) != list(range(self.num_splines))
        if self._subset:
            m = np.zeros(nonmatching_opt.vec_scalar_iga_dof)
            for s in set(self.vol_surf_inds):
                m[nonmatching_opt.cp_off[s]:nonmatching_opt.cp_off[s + 1]] = 1.0
            self._cp_mask = m                       # the volume of patch s depends on the control points and thicknesses of patch s only
        if self.opt_shape:
            self.opt_field = nonmatching_opt.opt_field
            self.shopt_surf_inds = nonmatching_opt.shopt_surf_inds

    def _f(self):
        return self.nonmatching_opt.functionals()

    def volume(self):
        """volume_exop.py:46-50."""
        f = self._f()
        if self._subset:                             # per-patch volumes of the same device pass (gf_functionals_per_patch), fixed order
            vp = f["volume_patch"]
            return float(sum(vp[s] for s in sorted(set(self.vol_surf_inds))))
        return float(f["volume"])

    def dvoldh_th(self, array=True):
        """volume_exop.py:52-66."""
        nm = self.nonmatching_opt
        g = self._f()["dVdh"]
        if self._subset:
            g = g * self._cp_mask
        return g if nm.var_thickness else np.add.reduceat(g, nm.cp_off[:-1])

    def dvoldCPIGA(self, field, array=True):
        """volume_exop.py:68-84."""
        nm = self.nonmatching_opt
        g = self._f()["dVdcp"][field]
        if self._subset:
            g = g * self._cp_mask
        return g[nm._shopt_cols[self.opt_field.index(field)]]
