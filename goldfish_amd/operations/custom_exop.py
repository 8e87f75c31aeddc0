"""CustomExOperation -- user-defined functional (GOLDFISH/operations/custom_exop.py:3-41).  The reference takes UFL forms
(one per patch) and assembles them through FFC; without a form compiler the functional is a pair of callables evaluated
on the problem object: ``func(nonmatching_opt) -> float`` and ``func_deriv(nonmatching_opt) -> ndarray``."""
import numpy as np


class CustomExOperation(object):

    def __init__(self, nonmatching_opt, func_symb, func_deriv_symb):
        if not callable(func_symb) or not callable(func_deriv_symb):
            raise TypeError("CustomExOperation: UFL forms cannot be compiled here; pass callables of the problem object")
        self.nonmatching_opt = nonmatching_opt
        self.func_symb, self.func_deriv_symb = func_symb, func_deriv_symb
        self.num_splines = nonmatching_opt.num_splines
        self.opt_field = nonmatching_opt.opt_field

    def func(self):
        return float(self.func_symb(self.nonmatching_opt))

    def func_deriv(self, extract=True, scalar=False, array=True):
        return np.asarray(self.func_deriv_symb(self.nonmatching_opt), float)
