"""IntXiEdgeComp -- parametric coordinates that must stay on a patch edge (linear equality constraint of the moving
intersections; reference: GOLDFISH/om_comps/int_xi_edge_comp.py:5-52, option and variable names of the reference)."""
import numpy as np
from scipy.sparse import coo_matrix

from ._design_io import _REQUIRED, LinearMapsComp


class IntXiEdgeComp(LinearMapsComp):
    OPTIONS = (('nonmatching_opt', _REQUIRED), ('input_xi_name', 'int_xi'), ('output_name', 'int_xi_edge'))

    def _build(self):
        nm = self.nonmatching_opt
        self.int_edge_cons_dofs = np.asarray(nm.cpiga2xi.int_edge_cons_dofs, dtype=np.int64)
        self.int_edge_cons_vals = np.asarray(nm.cpiga2xi.int_edge_cons_vals, float)
        self.input_shape, self.output_shape = nm.xi_size, self.int_edge_cons_dofs.size
        self.init_xi = nm.xi_flat.copy()
        self.deriv = self.get_derivative()
        return [(self.input_xi_name, self.output_name, self.deriv, self.init_xi, self.int_edge_cons_vals)]

    def get_derivative(self, coo=True):
        """Selection of the constrained coordinates (int_xi_edge_comp.py:43-50)."""
        n = self.output_shape
        mat = coo_matrix((np.ones(n), (np.arange(n), self.int_edge_cons_dofs)), shape=(n, self.input_shape))
        return mat if coo else mat.toarray()
