"""golemflavor_amd: MI355X-native evaluation engine for GolemFlavor's ensemble log-posterior.

Host code is Python and keeps the reference's call surface (`mcmc.mcmc`, `ln_prob(theta)`,
`Param` / `ParamSet`, enums); the per-walker physics is hand-written HIP for gfx950 behind the
C ABI declared in include/golemflavor_hip.h, loaded with ctypes.  There is no CPU fallback: any
evaluation raises if libgolemhip.so or a GPU is missing.
"""
from . import enums, param  # noqa: F401
from .enums import ParamTag, PriorsCateg, Texture  # noqa: F401
from .param import Param, ParamSet  # noqa: F401

__version__ = "0.1.0"
