"""CPSurfPinComp (reference module path GOLDFISH/om_comps/surf_comps/cpsurf_pin_comp.py)."""
from . import CPSurfPinComp      # noqa: F401
