"""MI355X-native LM-CMA population path (sample, trajectory cost, elite select).

The compute lives in ``lib/liblmcma_hip.so`` (hand-written HIP for gfx950 behind
the C ABI of ``include/lmcma_hip.h``); this package only binds it.
"""
from ._lib import LIB_PATH, LmcmaError, load  # noqa: F401
from .planner import CMAChol, LMCMA, SepCMA  # noqa: F401
