"""CPSurfDistanceComp (reference module path GOLDFISH/om_comps/surf_comps/cpsurf_distance_comp.py)."""
from . import CPSurfDistanceComp      # noqa: F401
