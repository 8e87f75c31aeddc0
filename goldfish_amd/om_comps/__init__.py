"""OpenMDAO components of the hot path (reference: GOLDFISH/om_comps/*, aggregated by
GOLDFISH/nonmatching_opt_om.py).  ``om`` is the real openmdao.api when importable."""
try:                                   # pragma: no cover - depends on the environment
    import openmdao.api as om
    HAVE_OPENMDAO = True
except ImportError:
    from .. import om_shim as om
    HAVE_OPENMDAO = False

from .disp_states_comp import DispStatesComp      # noqa: E402,F401
from .int_energy_comp import IntEnergyComp        # noqa: E402,F401
from .volume_comp import VolumeComp               # noqa: E402,F401
from .compliance_comp import ComplianceComp       # noqa: E402,F401
from .max_vmstress_comp import MaxvMStressComp   # noqa: E402,F401
from .cpiga2xi_comp import CPIGA2XiComp           # noqa: E402,F401
from .disp_states_mi_comp import DispMintStatesComp   # noqa: E402,F401
from .int_xi_edge_comp import IntXiEdgeComp         # noqa: E402,F401
from .int_energy_regu_comp import IntEnergyReguComp   # noqa: E402,F401
