"""HthFE2IGAImOperation -- the thickness counterpart of CPFE2IGAImOperation (GOLDFISH/operations/hthfe2iga_imop.py):
identity projection here, same method names."""
from .cpfe2iga_imop import CPFE2IGAImOperation


class HthFE2IGAImOperation(CPFE2IGAImOperation):

    def __init__(self, nonmatching_opt):
        self.nonmatching_opt = nonmatching_opt
        self.opt_thickness = nonmatching_opt.opt_thickness
        self.var_thickness = nonmatching_opt.var_thickness
