from .lsq import LsqQuantizer, LsqQuantizerWeight  # noqa: F401
from .statsq import StatsQuantizer  # noqa: F401
