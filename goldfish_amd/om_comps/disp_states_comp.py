"""DispStatesComp -- implicit component solving for the displacement states
(reference: GOLDFISH/om_comps/disp_states_comp.py:6-144; option names, defaults, variable names and shapes of the reference)."""
from . import om
from ._design_io import _REQUIRED, DesignIO
from ..operations.disp_imop import DispImOpeartion


class StatesComp(DesignIO, om.ImplicitComponent):
    """R(state; inputs) = 0 through an implicit operation ``self._imop`` (apply_linear_fwd/rev, solve_linear_fwd/rev, ...);
    ``STATE_OPTION`` names the option holding the output's name, ``_input_names()`` lists the inputs in the operation's order."""
    STATE_OPTION = 'output_u_name'

    @property
    def _state(self):
        return getattr(self, self.STATE_OPTION)

    def _state_setup(self):
        self.add_output(self.output_u_name, shape=self.output_shape)
        self.declare_partials(self.output_u_name, self.output_u_name)

    def apply_nonlinear(self, inputs, outputs, residuals):
        self.update_inputs_outpus(inputs, outputs)
        residuals[self.output_u_name] = self._imop.apply_nonlinear()

    def solve_nonlinear(self, inputs, outputs):
        self.update_inputs_outpus(inputs, outputs)
        outputs[self.output_u_name] = self._imop.solve_nonlinear(self.nonlinear_solver_max_it, self.nonlinear_solver_rtol)
        self.func_eval_ind += 1

    def linearize(self, inputs, outputs, partials):
        self.update_inputs_outpus(inputs, outputs)
        self._imop.linearize()
        self.func_eval_major_ind += [self.func_eval_ind - 1]
        self.major_iter_ind += 1

    def _products(self, d_inputs, d_outputs, d_residuals, mode):
        u = self._state
        args = (self._present(self._input_names(), d_inputs), d_outputs[u] if u in d_outputs else None, d_residuals[u] if u in d_residuals else None)
        if mode == 'fwd':
            self._imop.apply_linear_fwd(*args)
        elif mode == 'rev':
            self._imop.apply_linear_rev(*args)

    def solve_linear(self, d_outputs, d_residuals, mode):
        u = self._state
        if mode == 'fwd':
            self._imop.solve_linear_fwd(d_outputs[u], d_residuals[u])
        if mode == 'rev':
            self._imop.solve_linear_rev(d_outputs[u], d_residuals[u])

    def _counters(self, save_files, rtol, max_it):
        self.save_files, self.nonlinear_solver_rtol, self.nonlinear_solver_max_it = save_files, rtol, max_it
        self.major_iter_ind, self.func_eval_ind, self.func_eval_major_ind = 0, 0, []
        self.output_shape = self.nonmatching_opt.vec_iga_dof


class DispStatesComp(StatesComp):
    OPTIONS = (('nonmatching_opt', _REQUIRED), ('input_cp_iga_name_pre', 'CP_IGA'), ('input_h_th_name', 'thickness'),
               ('output_u_name', 'displacements'))

    def init_parameters(self, save_files=False, nonlinear_solver_rtol=1e-3, nonlinear_solver_max_it=30):
        """disp_states_comp.py:14-49 (must be called before setup, like the reference)."""
        self._read_options()
        self._counters(save_files, nonlinear_solver_rtol, nonlinear_solver_max_it)
        self.disp_state_imop = self._imop = DispImOpeartion(self.nonmatching_opt)
        self._init_design()
        if self.opt_shape:       # like the reference, CP_IGA<field> covers the optimised patches of that field (all patches in every demo, :37)
            self.input_cp_shapes = [v.size for v in self.init_cp_iga]

    def setup(self):
        self._state_setup()
        self._add_design_inputs(self.output_u_name)

    def _input_names(self):
        return self._design_names()

    def update_inputs_outpus(self, inputs, outputs):
        """disp_states_comp.py:68-79 (the reference's spelling)."""
        self._push_design(inputs)
        self.nonmatching_opt.update_uIGA(outputs[self.output_u_name])

    def apply_linear(self, inputs, outputs, d_inputs, d_outputs, d_residuals, mode):
        """disp_states_comp.py:107-134.  The reference re-pushes inputs here; the state is already on the device after
        linearize, so that redundant update is skipped."""
        self._products(d_inputs, d_outputs, d_residuals, mode)
