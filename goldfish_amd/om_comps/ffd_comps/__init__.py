from .cpffd2surf_comp import CPFFD2SurfComp     # noqa: F401
from .cpfe2iga_comp import CPFE2IGAComp         # noqa: F401
from .cpffd_design2full_comp import CPFFDesign2FullComp     # noqa: F401
from .cpffd_pin_comp import CPFFDPinComp                    # noqa: F401
from .cpffd_regu_comp import CPFFDReguComp                  # noqa: F401
from .hth_comps import HthFE2IGAComp, HthFFD2FEComp, HthFFDAlignComp, HthFFDReguComp, HthMapComp   # noqa: F401
