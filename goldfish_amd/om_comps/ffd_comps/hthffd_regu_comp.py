"""Reference module path GOLDFISH/om_comps/ffd_comps/hthffd_regu_comp.py: the thickness-FFD components share hth_comps.py here."""
from .hth_comps import HthFFDReguComp   # noqa: F401
