from .modules import *  # noqa: F401,F403
from .quantizer import *  # noqa: F401,F403
from .utils import KLLossSoft, KDLossSoftandHard  # noqa: F401
