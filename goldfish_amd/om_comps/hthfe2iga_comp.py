"""Reference module path GOLDFISH/om_comps/hthfe2iga_comp.py."""
from .ffd_comps.hth_comps import HthFE2IGAComp   # noqa: F401
