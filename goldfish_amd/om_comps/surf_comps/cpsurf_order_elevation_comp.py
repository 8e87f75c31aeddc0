"""CPSurfOrderElevationComp (reference module path GOLDFISH/om_comps/surf_comps/cpsurf_order_elevation_comp.py)."""
from . import CPSurfOrderElevationComp      # noqa: F401
