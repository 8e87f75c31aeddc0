"""Reference module path GOLDFISH/om_comps/ffd_comps/hthffd2fe_comp.py: the thickness-FFD components share hth_comps.py here."""
from .hth_comps import HthFFD2FEComp   # noqa: F401
