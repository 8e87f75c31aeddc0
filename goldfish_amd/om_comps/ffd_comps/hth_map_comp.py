"""Reference module path GOLDFISH/om_comps/ffd_comps/hth_map_comp.py: the thickness-FFD components share hth_comps.py here."""
from .hth_comps import HthMapComp   # noqa: F401
