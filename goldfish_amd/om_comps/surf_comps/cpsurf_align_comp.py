"""CPSurfAlignComp (reference module path GOLDFISH/om_comps/surf_comps/cpsurf_align_comp.py)."""
from . import CPSurfAlignComp      # noqa: F401
