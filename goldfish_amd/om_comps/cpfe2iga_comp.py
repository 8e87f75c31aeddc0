"""Reference module path GOLDFISH/om_comps/cpfe2iga_comp.py: the component lives in ffd_comps/cpfe2iga_comp.py here."""
from .ffd_comps.cpfe2iga_comp import CPFE2IGAComp   # noqa: F401
