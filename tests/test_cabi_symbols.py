"""CPU: libhiprag.so loads and exports every function include/hiprag.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(REPO, "include", "hiprag.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hip[a-z0-9]*_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    names = _declared_functions()
    for must in ("hipidx_create", "hipidx_add", "hipidx_search", "hipidx_search_dev", "hipidx_search_begin_dev",
                 "hipidx_search_finish_dev", "hipidx_save", "hipidx_load", "hipbm25_create", "hipbm25_search",
                 "hiprrf_fuse", "hiprag_merge_topk_dev", "hiprag_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from hiprag import _native as nat
    assert os.path.exists(nat.LIB_PATH), "build the library first: __graft_entry__.build()"
    lib = ctypes.CDLL(nat.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} declared in hiprag.h but not exported"


def test_python_binding_covers_every_declared_symbol():
    from hiprag import _native as nat
    nat.load()
    assert sorted(nat.exported_symbols()) == _declared_functions()


def test_errors_surface_as_exceptions_without_a_gpu_or_with_bad_args():
    import pytest
    from hiprag import _native as nat
    h = ctypes.c_uint64()
    with pytest.raises(nat.HipRagError) as e:
        nat.call("hipidx_create", 0, 0, 0, ctypes.byref(h))        # d = 0 is invalid everywhere
    assert "d must be positive" in str(e.value)
    with pytest.raises(nat.HipRagError):
        nat.call("hipidx_destroy", 123456789)


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under intool-rag_amd/ (the shipped host code) may import or execute it,
    and the C/HIP sources must not include anything from it."""
    import pathlib
    import re
    root = pathlib.Path(__file__).resolve().parents[1] / "intool-rag_amd"
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|^\s*#\s*include.*oracle|import_module\([\"']oracle|dlopen.*oracle", re.M)
    offenders = []
    for f in list(root.rglob("*.py")) + list(root.rglob("*.hip")) + list(root.rglob("*.cpp")) + list(root.rglob("*.h")):
        if "__pycache__" in f.parts or "build" in f.parts:
            continue
        if pat.search(f.read_text(errors="ignore")):
            offenders.append(str(f.relative_to(root)))
    assert offenders == []


def test_missing_library_raises_instead_of_falling_back(tmp_path):
    """Without libhiprag.so the product path raises at the first call (and the overlay reports it as the RuntimeError the
    reference raises for a missing FAISS, faiss_index.py:36-37): there is no CPU code path to fall back to.  Run in a
    fresh interpreter so that this process's already loaded library is not involved."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "os.environ['HIPRAG_LIB'] = %r\n"
        "from hiprag import HipFlatIndex, HipRagError, rrf_fuse\n"
        "for f in (lambda: HipFlatIndex(8, 'ip'), lambda: rrf_fuse(np.zeros((1, 2), np.int64), np.zeros((1, 2), np.int64), 1)):\n"
        "    try:\n"
        "        f(); raise SystemExit('no error without the library')\n"
        "    except HipRagError as e:\n"
        "        assert 'not found' in str(e)\n"
        "import rag.storage.hip_index as hi\n"
        "assert not hi.HAS_HIP\n"
        "try:\n"
        "    hi.create_hip_index([[0.0] * 8]); raise SystemExit('overlay did not raise')\n"
        "except RuntimeError as e:\n"
        "    assert 'libhiprag' in str(e)\n"
        "print('raises ok')\n") % (repo, os.path.join(repo, "intool-rag_amd"), str(tmp_path / "absent" / "libhiprag.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "raises ok" in r.stdout, r.stdout + r.stderr
