"""supersonic_amd -- MI355X-native Filter -> Project/Compute -> Aggregate (+Sort)
column-block pipeline behind Supersonic's Expression / Operation / Cursor API.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of
include/ssgpu.h) and `api.py` (the host-side mirror of supersonic/supersonic.h).
"""
from .api import *  # noqa: F401,F403
from . import _lib  # noqa: F401
