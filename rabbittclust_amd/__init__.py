"""rabbittclust_amd -- MI355X-native MinHash/KSSD sketching + all-pairs Mash distance
(the RabbitTClust hot path) behind a C ABI (include/rtclust.h).  No CPU fallback."""
from . import _lib  # noqa: F401
from ._lib import RtcError  # noqa: F401

__all__ = ["_lib", "RtcError"]
