"""MI355X-native failure-aware classification path.

``Backend`` is the drop-in scorer; everything it computes runs in the HIP
library behind include/fav.h.  Importing this package does not load the
library; constructing a Backend does, and raises if it (or a gfx950 GPU) is
missing — there is no CPU fallback.
"""
from .backend import Backend, anomaly_score_from_confidence  # noqa: F401
from .distributed import classify_sharded, shard_range  # noqa: F401
from . import synth, weights  # noqa: F401
