"""FAISS flat-file import/export: byte layout, round trip, error cases; GPU: load into HipFlatIndex."""
import struct

import numpy as np
import pytest


def test_byte_layout_and_roundtrip(tmp_path):
    from hiprag.faiss_io import read_faiss_flat, write_faiss_flat
    x = np.arange(12, dtype=np.float32).reshape(3, 4) / 7
    p = str(tmp_path / "a_faiss.index")
    write_faiss_flat(p, x, 1)
    raw = open(p, "rb").read()
    assert raw[:4] == b"IxF2"
    assert struct.unpack_from("<i", raw, 4)[0] == 4 and struct.unpack_from("<q", raw, 8)[0] == 3
    assert struct.unpack_from("<qq", raw, 16) == (1 << 20, 1 << 20)
    assert raw[32] == 1 and struct.unpack_from("<i", raw, 33)[0] == 1
    assert struct.unpack_from("<Q", raw, 37)[0] == 12 and len(raw) == 45 + 48
    m, y = read_faiss_flat(p)
    assert m == 1 and np.array_equal(x, y)
    write_faiss_flat(p, x, 0)
    assert open(p, "rb").read(4) == b"IxFI" and read_faiss_flat(p)[0] == 0


def test_rejects_other_files(tmp_path):
    from hiprag.faiss_io import read_faiss_flat
    p = tmp_path / "bad.index"
    p.write_bytes(b"IwFl" + b"\0" * 60)
    with pytest.raises(ValueError):
        read_faiss_flat(str(p))
    p.write_bytes(b"IxF2" + struct.pack("<iqqqBi", 4, 3, 1 << 20, 1 << 20, 1, 1) + struct.pack("<Q", 11) + b"\0" * 44)
    with pytest.raises(ValueError):
        read_faiss_flat(str(p))
    p.write_bytes(b"IxF2\0\0")
    with pytest.raises(ValueError):
        read_faiss_flat(str(p))


@pytest.mark.gpu
def test_faiss_file_loads_into_hip_index(gpu, tmp_path):
    from hiprag.faiss_io import load_faiss_flat_into_hip, write_faiss_flat
    from oracle import hybrid_oracle as ho
    x = ho.synthetic_vectors(700, 96, seed=71)
    p = str(tmp_path / "doc_faiss.index")
    write_faiss_flat(p, x, 1)
    ix = load_faiss_flat_into_hip(p)
    assert ix.ntotal == 700 and ix.d == 96 and ix.metric == 1
    q = ho.synthetic_queries(3, 96, seed=72)
    s, i = ix.search(q, 5)
    es, ei = ho.flat_search(x, q, 5, ho.METRIC_L2)
    assert np.array_equal(i, ei) and np.allclose(s, es, atol=1e-4)
