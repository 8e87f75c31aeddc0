"""CPSurfKnotRefinementComp (reference module path GOLDFISH/om_comps/surf_comps/cpsurf_knot_refienment_comp.py)."""
from . import CPSurfKnotRefinementComp      # noqa: F401
