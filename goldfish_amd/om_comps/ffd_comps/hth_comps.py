"""Thickness design-variable maps (constant sparse linear maps; same option and variable names as the reference):
HthFFD2FEComp   GOLDFISH/om_comps/ffd_comps/hthffd2fe_comp.py:4-36    FFD thickness coefficients -> thickness at the control points
HthFFDAlignComp GOLDFISH/om_comps/ffd_comps/hthffd_align_comp.py:4-36 equality constraint rows (aligned layers)
HthFFDReguComp  GOLDFISH/om_comps/ffd_comps/hthffd_regu_comp.py:4-45  inequality constraint rows (neighbour differences)
HthMapComp      GOLDFISH/om_comps/ffd_comps/hth_map_comp.py:6-50      one thickness per patch -> full thickness vector
HthFE2IGAComp   GOLDFISH/om_comps/hthfe2iga_comp.py:5-70               thickness "FE dofs" -> IGA control-point values: the identity here"""
import numpy as np
import scipy.sparse as sp

from .. import om


class _LinearMapComp(om.ExplicitComponent):
    """y = A x with a constant sparse A (subclasses fill in_name, out_name, deriv, init)."""

    def setup(self):
        self.add_input(self.in_name, shape=self.deriv.shape[1], val=self.init)
        self.add_output(self.out_name, shape=self.deriv.shape[0])
        self.declare_partials(self.out_name, self.in_name, val=self.deriv.data, rows=self.deriv.row, cols=self.deriv.col)

    def compute(self, inputs, outputs):
        outputs[self.out_name] = self.deriv * inputs[self.in_name]

    # the partials are constant and declared in setup (COO values in rows / cols order): no compute_partials


class HthFFD2FEComp(_LinearMapComp):

    def initialize(self):
        self.options.declare('nonmatching_opt_ffd')
        self.options.declare('input_h_th_ffd_name', default='thickness_FFD')
        self.options.declare('output_h_th_fe_name', default='thickness_FE')

    def init_parameters(self):
        nm = self.nonmatching_opt_ffd = self.options['nonmatching_opt_ffd']
        self.in_name = self.input_h_th_ffd_name = self.options['input_h_th_ffd_name']
        self.out_name = self.output_h_th_fe_name = self.options['output_h_th_fe_name']
        if getattr(nm, 'thopt_multiffd', False):
            self.init = self.init_h_th_ffd = nm.get_init_h_th_multiFFD()
            self.deriv = self.deriv_mat = nm.thopt_dcpsurf_fedcpmultiffd.tocoo()
        else:
            self.init = self.init_h_th_ffd = nm.get_init_h_th_FFD()
            self.deriv = self.deriv_mat = nm.thopt_dcpsurf_fedcpffd.tocoo()


class HthFFDAlignComp(_LinearMapComp):

    def initialize(self):
        self.options.declare('nonmatching_opt_ffd')
        self.options.declare('input_h_th_name', default='thickness_FFD')
        self.options.declare('output_h_th_align_name', default='thickness_FFD_align')

    def init_parameters(self):
        nm = self.nonmatching_opt_ffd = self.options['nonmatching_opt_ffd']
        self.in_name = self.input_h_th_name = self.options['input_h_th_name']
        self.out_name = self.output_h_th_align_name = self.options['output_h_th_align_name']
        if getattr(nm, 'thopt_multiffd', False):
            self.init = self.init_h_th_ffd = nm.get_init_h_th_multiFFD()
            self.deriv = nm.thopt_dcpaligndcpmultiffd.tocoo()
        else:
            self.init = self.init_h_th_ffd = nm.get_init_h_th_FFD()
            self.deriv = nm.thopt_dcpaligndcpffd.tocoo()


class HthFFDReguComp(_LinearMapComp):

    def initialize(self):
        self.options.declare('nonmatching_opt_ffd')
        self.options.declare('input_h_th_name', default='thickness_FFD')
        self.options.declare('output_h_th_regu_name', default='thickness_FFD_regu')

    def init_parameters(self):
        nm = self.nonmatching_opt_ffd = self.options['nonmatching_opt_ffd']
        self.in_name = self.input_h_th_name = self.options['input_h_th_name']
        self.out_name = self.output_h_th_regu_name = self.options['output_h_th_regu_name']
        self.init = self.init_h_th_ffd = nm.get_init_h_th_FFD()
        self.deriv = nm.thopt_dcpregudcpffd_list[0].tocoo()
        self.input_shape, self.output_shape = nm.thopt_cpffd_size, nm.thopt_cpregu_sizes[0]


class HthMapComp(_LinearMapComp):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('order', default=0)
        self.options.declare('input_h_th_name_design', default='thickness')
        self.options.declare('output_h_th_name_full', default='thickness_full')

    def init_parameters(self):
        nm = self.nonmatching_opt = self.options['nonmatching_opt']
        self.order = self.options['order']
        if self.order != 0:
            raise ValueError("Order {:2d} is not supported yet".format(self.order))
        self.in_name = self.input_h_th_name_design = self.options['input_h_th_name_design']
        self.out_name = self.output_h_th_name_full = self.options['output_h_th_name_full']
        self.num_splines = nm.num_splines
        # thickness dofs per patch as set_thickness_opt laid them out (reference: nonmatching_opt.h_th_sizes, hth_map_comp.py:48-56): one
        # per patch for a constant thickness (the "FE dofs" of a patch's thickness function collapse to its single value here, so the map is
        # the identity and its output is what DispStatesComp / IntEnergyComp / VolumeComp take as ``thickness``), one per control point
        # for a variable thickness
        sizes = [int(n) for n in getattr(nm, "h_th_sizes", nm.vec_scalar_iga_dof_list)]
        self.init = self.init_val = np.array([float(np.mean(h)) for h in nm.h_th])
        rows = np.arange(sum(sizes))
        cols = np.repeat(np.arange(self.num_splines), sizes)
        self.deriv = self.deriv_mat = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(rows.size, self.num_splines))

    def get_derivative(self, coo=True):
        return self.deriv if coo else self.deriv.toarray()


class HthFE2IGAComp(_LinearMapComp):
    """In the reference an implicit L2 projection of the FE thickness function onto the IGA control-point values
    (operations/hthfe2iga_imop.py); thickness lives directly on the control points here, so the map is the identity.
    Kept so that the demos' wiring thickness_FE -> thickness_IGA works unchanged."""

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('input_h_th_fe_name', default='thickness_FE')
        self.options.declare('output_h_th_iga_name', default='thickness_IGA')

    def init_parameters(self):
        nm = self.nonmatching_opt = self.options['nonmatching_opt']
        self.in_name = self.input_h_th_fe_name = self.options['input_h_th_fe_name']
        self.out_name = self.output_h_th_iga_name = self.options['output_h_th_iga_name']
        self.input_shape = self.output_shape = nm.vec_scalar_iga_dof
        self.init = nm.init_h_th_fe
        self.deriv = sp.identity(nm.vec_scalar_iga_dof, format="coo")
