"""The drop-in boundary: libmarlin_gpu.so loads without a GPU and exports every entry point that
include/marlin_gpu.h declares; the product path refuses to run (no CPU fallback) without a device."""
import ctypes as C
import os
import re

import pytest

from marlin_amd import GpuEngine, GpuError, Index, ffi
from util import ROOT

HEADER = os.path.join(ROOT, "include", "marlin_gpu.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return sorted(set(re.findall(r"\b(mgpu_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("mgpu_shard_create", "mgpu_shard_destroy", "mgpu_ctx_create", "mgpu_ctx_destroy",
                 "mgpu_execute", "mgpu_sqout_free", "mgpu_last_error", "mgpu_posting_op"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(ffi.LIB_DIR, "libmarlin_gpu.so"))
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_no_torch_or_cxx_types_cross_the_boundary():
    src = open(HEADER).read()
    assert 'extern "C"' in src
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for bad in ("torch", "at::", "std::", "hipStream_t", "template"):
        assert bad not in src, bad


def test_struct_sizes_match_the_host_mirror():
    ffi.load()          # raises if a ctypes mirror disagrees with the library


def test_product_path_fails_loudly_without_a_gpu():
    lib = ffi.load()
    if lib.mgpu_device_count() > 0:
        pytest.skip("a GPU is visible here")
    ix = Index({"indexedFields": ["t"], "facetFields": []}, num_shards=1).add_documents([{"t": "hello world"}])
    try:
        with pytest.raises(GpuError):
            GpuEngine(ix)
    finally:
        ix.close()


def test_loading_the_library_puts_torch_first_where_it_is_installed():
    """PyTorch ships ROCm runtime libraries of its own; a process that loads libmarlin_gpu.so (linked against /opt/rocm)
    and imports torch AFTERWARDS aborts when it exits (found in round 3).  ffi.load() therefore imports torch first where
    it is installed: a fresh interpreter must have it in sys.modules once the library is loaded."""
    import importlib.util
    import subprocess
    import sys
    if importlib.util.find_spec("torch") is None:
        pytest.skip("torch is not installed")
    code = "import sys; from marlin_amd import ffi; assert 'torch' not in sys.modules; ffi.load(); assert 'torch' in sys.modules"
    subprocess.run([sys.executable, "-c", code], cwd=ROOT, check=True, timeout=300)
