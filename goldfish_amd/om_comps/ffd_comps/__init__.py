from .cpffd2surf_comp import CPFFD2SurfComp     # noqa: F401
from .cpfe2iga_comp import CPFE2IGAComp         # noqa: F401
