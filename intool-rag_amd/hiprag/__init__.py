"""hiprag -- Python host side of libhiprag.so (MI355X hybrid-retrieval hot path).  No CPU fallbacks."""
from ._native import METRIC_IP, METRIC_L2, HipRagError, LIB_PATH  # noqa: F401
from .index import HipFlatIndex, merge_topk_device  # noqa: F401
from .sparse import HipBM25, PostingsCSR, build_postings, build_postings_from_texts, tokenize  # noqa: F401
from .fusion import rrf_fuse, rrf_fuse_device, RRF_C  # noqa: F401
from .encoder import EncoderConfig, HipEncoder, random_state  # noqa: F401
