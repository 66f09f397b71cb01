"""CPU tests: the C-ABI library loads and exports every symbol include/ise_knn.h
declares, argument errors are reported through the ABI, and the product path
fails loudly (never falls back) when no MI355X is present.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ise_knn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ise_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    from image_search_engine_amd import _native

    syms = declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(_native.lib, s), f"{s} declared in include/ise_knn.h but not exported"
    assert sorted(n for n, _, _ in _native.PROTOTYPES) == syms, "ctypes prototypes out of sync with the header"
    assert _native.lib.ise_version() >= 100


def test_argument_errors_through_abi():
    from image_search_engine_amd import _native as n

    h = ctypes.c_void_p()
    assert n.lib.ise_index_create(ctypes.byref(h), 0, n.METRIC_L2, 0) == n.E_INVALID
    assert b"positive" in n.lib.ise_last_error()
    assert n.lib.ise_index_create(ctypes.byref(h), 8, 7, 0) == n.E_INVALID
    assert n.lib.ise_index_create(None, 8, n.METRIC_L2, 0) == n.E_INVALID
    assert n.lib.ise_index_destroy(None) == 0
    assert n.lib.ise_index_search_host(None, None, 1, 1, None, None) == n.E_INVALID
    assert n.lib.ise_normalize_rows_host(None, 3, 4, 0) == n.E_INVALID
    assert n.lib.ise_merge_keys_device(None, 1, 1, 1, 1, None, None, 0, None) == n.E_INVALID


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import image_search_engine_amd.faiss_compat as faiss
    from image_search_engine_amd.utils import create_search_index

    with pytest.raises(RuntimeError, match="MI355X|HIP device"):
        faiss.IndexFlatL2(16)
    with pytest.raises(RuntimeError):
        faiss.normalize_L2(np.ones((2, 4), np.float32))
    with pytest.raises(RuntimeError):
        create_search_index(np.ones((4, 8), np.float32), "l2")


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the product package may import it."""
    pkg = os.path.join(ROOT, "image-search-engine_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "flat_oracle" not in src and "knn_oracle" not in src, f


def test_comm_entry_points_without_a_gpu():
    """ise_comm_*: the id can be drawn anywhere librccl loads; joining needs a GPU and says so."""
    import torch

    from image_search_engine_amd import _native as n

    buf = ctypes.create_string_buffer(128)
    rc = n.lib.ise_comm_unique_id(buf)
    assert rc in (0, n.E_NODEVICE)          # E_NODEVICE: no librccl on this machine
    assert n.lib.ise_comm_unique_id(None) == n.E_INVALID
    h = ctypes.c_void_p()
    assert n.lib.ise_comm_create(ctypes.byref(h), buf.raw, 0, 0, 0) == n.E_INVALID      # world < 1
    assert n.lib.ise_comm_create(ctypes.byref(h), buf.raw, 2, 2, 0) == n.E_INVALID      # rank out of range
    assert n.lib.ise_comm_create(None, buf.raw, 1, 0, 0) == n.E_INVALID
    assert n.lib.ise_comm_allgather_keys(None, None, None, 4, None) == n.E_INVALID
    assert n.lib.ise_comm_destroy(None) == 0
    if not torch.cuda.is_available() and rc == 0:
        assert n.lib.ise_comm_create(ctypes.byref(h), buf.raw, 1, 0, 0) == n.E_NODEVICE
        assert b"GPU" in n.lib.ise_last_error()


def test_header_is_plain_c():
    """include/ise_knn.h is the drop-in boundary: it must compile as C99 on its own (plain pointers and sizes,
    no C++ or torch types), and as C++ behind its extern "C" guard."""
    import shutil
    import subprocess

    hdr = os.path.join(ROOT, "include", "ise_knn.h")
    gcc, gxx = shutil.which("gcc"), shutil.which("g++")
    if not gcc or not gxx:
        pytest.skip("no host C/C++ compiler")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([gxx, "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", hdr],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
