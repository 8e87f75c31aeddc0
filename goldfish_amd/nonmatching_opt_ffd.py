"""Reference module path GOLDFISH/nonmatching_opt_ffd.py (NonMatchingOptFFD lives in nonmatching_opt.py here)."""
from .nonmatching_opt import *          # noqa: F401,F403
from .nonmatching_opt import NonMatchingOpt, NonMatchingOptFFD, PointSource, SVKResidual   # noqa: F401
