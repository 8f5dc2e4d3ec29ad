"""hiprag -- Python host side of libhiprag.so (MI355X hybrid-retrieval hot path).  No CPU fallbacks."""
from . import _native
from ._native import METRIC_IP, METRIC_L2, HipRagError, LIB_PATH  # noqa: F401
from .index import HipFlatIndex, merge_topk_device  # noqa: F401
from .ivf import HipIVFIndex  # noqa: F401
from .sparse import HipBM25, PostingsCSR, build_postings, build_postings_from_texts, tokenize  # noqa: F401
from .fusion import hybrid_search, hybrid_search_device, rrf_fuse, rrf_fuse_device, RRF_C  # noqa: F401
from .encoder import EncoderConfig, HipEncoder, random_state  # noqa: F401


def init(n_devices: int = 0) -> None:
    """hiprag_init: check that `n_devices` GPUs are visible (0 = at least one) and create their contexts now."""
    _native.call("hiprag_init", int(n_devices))


def shutdown() -> None:
    """hiprag_shutdown: synchronise every device and drop every library handle still alive.  Python objects that wrap a
    handle must not be used afterwards (their close() then reports an unknown handle)."""
    _native.call("hiprag_shutdown")
    from . import sharded
    sharded._SCAN_STREAMS.clear()      # wrappers of the library's scan and tail streams, destroyed with it
    sharded._TAIL_STREAMS.clear()
