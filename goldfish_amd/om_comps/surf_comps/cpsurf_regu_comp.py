"""CPSurfReguComp (reference module path GOLDFISH/om_comps/surf_comps/cpsurf_regu_comp.py)."""
from . import CPSurfReguComp      # noqa: F401
