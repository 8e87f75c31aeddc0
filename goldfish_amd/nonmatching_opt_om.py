"""Aggregator with the reference's name (GOLDFISH/nonmatching_opt_om.py): ``from goldfish_amd.nonmatching_opt_om import *``
brings the problem classes, the FFD utilities and every OpenMDAO component, as the demos expect."""
from .utils.ffd_utils import *                       # noqa: F401,F403
from .nonmatching_opt import NonMatchingOpt, NonMatchingOptFFD, PointSource, SVKResidual   # noqa: F401
from .cpiga2xi import CPIGA2Xi, IntersectionData     # noqa: F401
from .om_comps import (ComplianceComp, CPIGA2XiComp, DispMintStatesComp, DispStatesComp, IntEnergyComp, IntXiEdgeComp,   # noqa: F401
                       MaxvMStressComp, VolumeComp, om)
from .om_comps.ffd_comps import (CPFE2IGAComp, CPFFD2SurfComp, CPFFDesign2FullComp, CPFFDPinComp, CPFFDReguComp,          # noqa: F401
                                 HthFE2IGAComp, HthFFD2FEComp, HthFFDAlignComp, HthFFDReguComp, HthMapComp)
