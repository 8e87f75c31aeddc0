"""Reference module path GOLDFISH/om_comps/ffd_comps/hthffd_align_comp.py: the thickness-FFD components share hth_comps.py here."""
from .hth_comps import HthFFDAlignComp   # noqa: F401
