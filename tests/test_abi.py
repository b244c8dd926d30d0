"""The C-ABI library loads and exports every symbol include/kdf.h declares
(no compute calls: runs without a GPU)."""
import os
import re

from conftest import ROOT


def test_header_symbols_exported():
    from kmer_denovo_filter_amd import _native
    lib = _native.load()
    hdr = open(os.path.join(ROOT, "include", "kdf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(kdf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    bound = {name for name, _, _ in _native.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert getattr(lib, name) is not None


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a GPU the engine refuses to construct."""
    import ctypes
    from kmer_denovo_filter_amd import _native
    lib = _native.load()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    h = ctypes.c_void_p()
    rc = lib.kdf_create(0, 31, 1024, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert b"HIP" in lib.kdf_last_error(None) or b"device" in lib.kdf_last_error(None)


def test_product_never_imports_oracle():
    """The shipped package must not import, link or dlopen anything under oracle/."""
    import re
    pkg = os.path.join(ROOT, "kmer_denovo_filter_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|libkdf_oracle|kdf_oracle\.c|kdfo_[a-z_]+\s*\(", re.M)
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert not bad.search(src), f"{fn} uses the oracle"
