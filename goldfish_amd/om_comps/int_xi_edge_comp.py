"""IntXiEdgeComp -- parametric coordinates that must stay on a patch edge (linear equality constraint of the moving
intersections; reference: GOLDFISH/om_comps/int_xi_edge_comp.py:5-52, same option and variable names)."""
import numpy as np
from scipy.sparse import coo_matrix

from . import om


class IntXiEdgeComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('input_xi_name', default='int_xi')
        self.options.declare('output_name', default='int_xi_edge')

    def init_parameters(self):
        self.nonmatching_opt = self.options['nonmatching_opt']
        self.input_xi_name = self.options['input_xi_name']
        self.output_name = self.options['output_name']
        self.int_edge_cons_dofs = np.asarray(self.nonmatching_opt.cpiga2xi.int_edge_cons_dofs, dtype=np.int64)
        self.int_edge_cons_vals = np.asarray(self.nonmatching_opt.cpiga2xi.int_edge_cons_vals, float)
        self.input_shape = self.nonmatching_opt.xi_size
        self.output_shape = self.int_edge_cons_dofs.size
        self.init_xi = self.nonmatching_opt.xi_flat.copy()
        self.deriv = self.get_derivative()

    def setup(self):
        self.add_input(self.input_xi_name, shape=self.input_shape, val=self.init_xi)
        self.add_output(self.output_name, shape=self.output_shape)
        self.declare_partials(self.output_name, self.input_xi_name, val=self.deriv.data, rows=self.deriv.row, cols=self.deriv.col)

    def compute(self, inputs, outputs):
        outputs[self.output_name] = np.asarray(inputs[self.input_xi_name])[self.int_edge_cons_dofs] - self.int_edge_cons_vals

    def compute_partials(self, inputs, partials):
        partials[self.output_name, self.input_xi_name] = self.deriv.toarray()

    def get_derivative(self, coo=True):
        n = self.output_shape
        mat = coo_matrix((np.ones(n), (np.arange(n), self.int_edge_cons_dofs)), shape=(n, self.input_shape))
        return mat if coo else mat.toarray()
