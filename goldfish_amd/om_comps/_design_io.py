"""What every component of the path shares: the option table, the design inputs (``CP_IGA<field>`` per optimised
coordinate field, ``thickness``) with their shapes and initial values, pushing them into the problem object, and -- for the
explicit components -- the value / partials plumbing of a scalar functional.  The components themselves only state their
option names and defaults (the reference's, file by file) and which operation methods give the value and the gradients."""
from . import om


class DesignIO:
    """Mixin.  ``OPTIONS``: ((option name, default), ...); every option is copied to an attribute of the same name by
    ``_read_options`` (``init_parameters`` of the reference components does the same line by line)."""
    OPTIONS = ()

    def initialize(self):
        for name, default in self.OPTIONS:
            if default is _REQUIRED:
                self.options.declare(name)
            else:
                self.options.declare(name, default=default)

    def _read_options(self):
        for name, _ in self.OPTIONS:
            setattr(self, name, self.options[name])

    def _init_design(self, thickness=True):
        nm = self.nonmatching_opt
        self.opt_field, self.opt_shape = nm.opt_field, nm.opt_shape
        self.opt_thickness, self.var_thickness = (nm.opt_thickness, nm.var_thickness) if thickness else (False, False)
        if self.opt_shape:
            self.init_cp_iga = nm.get_init_CPIGA()
            self.input_cp_shapes = [len(d) for d in nm.cpdes_iga_dofs_full]     # sized over the optimised patches of each field
            self.input_cp_iga_name_list = [self.input_cp_iga_name_pre + str(f) for f in self.opt_field]
        if self.opt_thickness:
            self.input_h_th_shape = nm.vec_scalar_iga_dof if self.var_thickness else nm.h_th_dof
            self.init_h_th = nm.init_h_th_iga if self.var_thickness else nm.init_h_th

    def _design_names(self):
        names = list(self.input_cp_iga_name_list) if self.opt_shape else []
        return names + ([self.input_h_th_name] if self.opt_thickness else [])

    def _add_design_inputs(self, of):
        if self.opt_shape:
            for name, shape, val in zip(self.input_cp_iga_name_list, self.input_cp_shapes, self.init_cp_iga):
                self.add_input(name, shape=shape, val=val)
                self.declare_partials(of, name)
        if self.opt_thickness:
            self.add_input(self.input_h_th_name, shape=self.input_h_th_shape, val=self.init_h_th)
            self.declare_partials(of, self.input_h_th_name)

    def _push_design(self, inputs):
        nm = self.nonmatching_opt
        if self.opt_shape:
            for name, field in zip(self.input_cp_iga_name_list, self.opt_field):
                nm.update_CPIGA(inputs[name], field)
        if self.opt_thickness:
            (nm.update_h_th_IGA if self.var_thickness else nm.update_h_th)(inputs[self.input_h_th_name])

    def _present(self, names, vec):
        """Arrays of ``vec`` for ``names`` in order (None when none is present): the list the operations take."""
        out = [vec[n] for n in names if n in vec]
        return out or None


class _Required:
    pass


_REQUIRED = _Required()


class FunctionalComp(DesignIO, om.ExplicitComponent):
    """Scalar functional J(u, CP, h): subclasses give OPTIONS, the name of the output option, whether J depends on u, and
    ``_operation()``, ``_value()``, ``_du()``, ``_dcp(field)``, ``_dh()``."""
    OUTPUT_OPTION = None
    USES_U = True
    USES_THICKNESS = True
    APPLY_BCS_IN_PARTIALS = False

    def init_parameters(self, *args, **kwargs):
        self._read_options()
        self._init_design(thickness=self.USES_THICKNESS)
        self._of = getattr(self, self.OUTPUT_OPTION)
        self._operation(*args, **kwargs)
        if self.USES_U:
            self.input_u_shape = self.nonmatching_opt.vec_iga_dof
            self.init_disp_array = self._initial_u()

    def _initial_u(self):
        return self.nonmatching_opt.u_iga.copy()

    def setup(self):
        self.add_output(self._of)
        if self.USES_U:
            self.add_input(self.input_u_name, shape=self.input_u_shape, val=self.init_disp_array)
            self.declare_partials(self._of, self.input_u_name)
        self._add_design_inputs(self._of)

    def update_inputs(self, inputs):
        self._push_design(inputs)
        if self.USES_U:
            self.nonmatching_opt.update_uIGA(inputs[self.input_u_name])

    def compute(self, inputs, outputs):
        self.update_inputs(inputs)
        outputs[self._of] = self._value()

    def compute_partials(self, inputs, partials):
        self.update_inputs(inputs)
        if self.USES_U:
            partials[self._of, self.input_u_name] = self._du()
        if self.opt_shape:
            for name, field in zip(self.input_cp_iga_name_list, self.opt_field):
                partials[self._of, name] = self._dcp(field)
        if self.opt_thickness:
            partials[self._of, self.input_h_th_name] = self._dh()


class LinearMapsComp(DesignIO, om.ExplicitComponent):
    """outputs[k] = A_k inputs[k] - b_k for constant sparse A_k (one map per optimised field, or a single one).
    Subclasses give OPTIONS and ``_build()`` -> [(input name, output name, A, initial input, b or None), ...]."""

    def init_parameters(self):
        self._read_options()
        self._maps = [(i, o, A.tocoo(), x0, b) for (i, o, A, x0, b) in self._build()]

    def setup(self):
        for i, o, A, x0, b in self._maps:
            self.add_input(i, shape=A.shape[1], val=x0)
            self.add_output(o, shape=A.shape[0])
            self.declare_partials(o, i, val=A.data, rows=A.row, cols=A.col)

    def compute(self, inputs, outputs):
        for i, o, A, x0, b in self._maps:
            y = A * inputs[i]
            outputs[o] = y if b is None else y - b

    # no compute_partials: the partials are constant and declared (COO values in rows / cols order) in setup
