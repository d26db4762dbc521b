"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares.
No compute calls here (no GPU); creating a table must fail loudly rather than fall back."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "meepo_embedding.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mee_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built):
    from meepoembedding_amd import _lib
    names = _declared()
    assert len(names) >= 20
    L = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/meepo_embedding.h but not exported: {missing}"
    assert sorted(_lib.PROTOTYPES) == names, "python prototypes out of sync with the header"
    assert _lib.lib().mee_abi_version() == _lib.ABI_VERSION == 2


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 on its own (a cgo / JNI / ctypes binding includes nothing else), and
    the header-only C++ layer as C++17."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", os.path.join(inc, "meepo_embedding.h")])
    subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-pragma-once-outside-header", "-I", inc,
                           os.path.join(inc, "meepo_embedding.hpp")])


def test_config_struct_layout(built):
    from meepoembedding_amd import _lib
    assert C.sizeof(_lib.Config) == 64 and C.sizeof(_lib.TableInfo) == 48


def test_no_cpu_fallback(built):
    """Without a GPU, table creation raises; with one, a bad ABI size is rejected."""
    import torch

    from meepoembedding_amd import LookupTable, MeepoError, _lib
    if not torch.cuda.is_available():
        with pytest.raises(MeepoError) as e:
            LookupTable(1024, 16)
        assert e.value.code in (_lib.ERR_NO_DEVICE, _lib.ERR_HIP)
    cfg = _lib.Config(struct_size=12)
    h = C.c_void_p()
    rc = _lib.lib().mee_table_create(C.byref(cfg), C.byref(h))
    assert rc == _lib.ERR_INVALID_ARG and b"struct_size" in _lib.lib().mee_last_error()


def test_product_never_imports_oracle():
    """The shipped package must not reference the test oracle in any way."""
    pkg = os.path.join(ROOT, "meepoembedding_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "meepo_oracle" not in txt, f


def test_test_infrastructure_libraries_load():
    """The shared-memory stand-in for librccl (tests/cabi/fake_rccl.cpp) that the one-GPU multi-rank tests bind must at least load: an
    unresolved symbol in it would silently turn those tests' native-transport legs into fallbacks on the GPU box."""
    import ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "build", "libfake_rccl.so")
    assert os.path.exists(so), "build/libfake_rccl.so missing: run __graft_entry__.build()"
    L = ctypes.CDLL(so)
    for sym in ("ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclAllReduce"):
        assert hasattr(L, sym), sym
