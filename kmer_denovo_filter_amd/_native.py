"""ctypes binding of include/kdf.h (libkdf.so).

This is the stub a maintainer of the reference would add.
There is no CPU fallback: if the library is missing or no GPU is visible the
calls raise, loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_int, c_int32, c_int64, c_uint16, c_uint32, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libkdf.so")

KDF_OK = 0
KDF_ERR_INVALID, KDF_ERR_HIP, KDF_ERR_NOMEM, KDF_ERR_TABLE_FULL, KDF_ERR_IO, KDF_ERR_STATE = 1, 2, 3, 4, 5, 6

# every symbol include/kdf.h declares: (name, restype, argtypes)
_P = c_void_p
SYMBOLS = [
    ("kdf_create", c_int, [c_int, c_int, c_uint64, POINTER(_P)]),
    ("kdf_destroy", None, [_P]),
    ("kdf_last_error", c_char_p, [_P]),
    ("kdf_set_stream", c_int, [_P, _P]),
    ("kdf_synchronize", c_int, [_P]),
    ("kdf_clear", c_int, [_P]),
    ("kdf_reserve", c_int, [_P, c_uint64]),
    ("kdf_stats", c_int, [_P, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    ("kdf_flush", c_int, [_P]),
    ("kdf_set_option", c_int, [_P, c_char_p, c_int64]),
    ("kdf_get_stat", c_int, [_P, c_char_p, POINTER(c_int64)]),
    ("kdf_profile", c_int, [_P, c_int]),
    ("kdf_profile_read", c_int, [_P, POINTER(ctypes.c_double), POINTER(c_uint64), POINTER(c_uint64)]),
    ("kdf_profile_stages", c_int, [_P, POINTER(ctypes.c_double), POINTER(c_uint64)]),
    ("kdf_count_reads", c_int, [_P, _P, _P, c_uint64]),
    ("kdf_count_reads_dev", c_int, [_P, _P, _P, c_uint64]),
    ("kdf_add_pairs", c_int, [_P, _P, _P, _P, c_uint64]),
    ("kdf_host_alloc", c_int, [c_uint64, POINTER(c_void_p)]),
    ("kdf_host_free", c_int, [_P]),
    ("kdf_upload_reads_async", c_int, [_P, c_int, _P, _P, c_uint64]),
    ("kdf_count_uploaded", c_int, [_P, c_int, c_int]),
    ("kdf_add_pairs_dev", c_int, [_P, _P, _P, _P, c_uint64]),
    ("kdf_set_counts_dev", c_int, [_P, c_void_p, c_void_p, c_void_p, c_uint64]),
    ("kdf_add_pairs_multi_dev", c_int, [_P, c_uint32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_uint64)]),
    ("kdf_load_filter", c_int, [_P, _P, _P, c_uint64]),
    ("kdf_load_filter_dev", c_int, [_P, _P, _P, c_uint64]),
    ("kdf_reset_counts", c_int, [_P]),
    ("kdf_count_reads_filtered", c_int, [_P, _P, _P, c_uint64]),
    ("kdf_count_reads_filtered_dev", c_int, [_P, _P, _P, c_uint64]),
    ("kdf_query", c_int, [_P, _P, _P, c_uint64, _P]),
    ("kdf_query_dev", c_int, [_P, _P, _P, c_uint64, _P]),
    ("kdf_count_ge", c_int, [_P, c_uint32, POINTER(c_uint64)]),
    ("kdf_export_ge", c_int, [_P, c_uint32, _P, _P, _P, c_uint64, POINTER(c_uint64)]),
    ("kdf_export_ge_dev", c_int, [_P, c_uint32, _P, _P, _P, c_uint64, c_int, POINTER(c_uint64)]),
    ("kdf_export_parts_dev", c_int, [_P, c_uint32, c_uint32, _P, _P, _P, c_uint64, POINTER(c_uint64), POINTER(c_uint64)]),
    ("kdf_export_parts_packed_dev", c_int, [_P, c_uint32, c_uint32, _P, c_uint64, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    ("kdf_scan_reads", c_int, [_P, _P, _P, c_uint64, _P, c_int64, _P, _P]),
    ("kdf_scan_reads_dev", c_int, [_P, _P, _P, c_uint64, _P]),
    ("kdf_stream_words", None, [c_uint64, POINTER(c_uint64), POINTER(c_uint64)]),
    ("kdf_pack_reads", c_int, [_P, _P, c_int64, _P, _P, _P, POINTER(c_uint64)]),
    ("kdf_canonical", c_int, [c_char_p, c_int, POINTER(c_uint64), POINTER(c_uint64)]),
    ("kdf_bam_open", c_int, [c_char_p, c_uint32, c_int, c_int, POINTER(_P)]),
    ("kdf_bam_open_range", c_int, [c_char_p, c_uint32, c_int, c_int, c_int, c_int, POINTER(_P)]),
    ("kdf_fasta_open", c_int, [c_char_p, c_int, POINTER(_P)]),
    ("kdf_reader_next", c_int, [_P, c_uint64, c_int64, _P, _P, _P, POINTER(c_int64), POINTER(c_uint64)]),
    ("kdf_reader_last_meta", c_int, [_P, POINTER(POINTER(c_uint16)), POINTER(POINTER(c_int32)),
                                     POINTER(POINTER(c_int32)), POINTER(c_char_p), POINTER(POINTER(c_int64))]),
    ("kdf_reader_want_aux", c_int, [_P, c_int]),
    ("kdf_reader_last_aux", c_int, [_P, POINTER(POINTER(c_uint32)), POINTER(POINTER(c_int64)),
                                    POINTER(c_char_p), POINTER(POINTER(c_int64))]),
    ("kdf_reader_last_quals", c_int, [_P, POINTER(POINTER(ctypes.c_uint8)), POINTER(POINTER(c_int64)),
                                      POINTER(POINTER(ctypes.c_uint8))]),
    ("kdf_device_memory", c_int, [c_int, POINTER(c_uint64), POINTER(c_uint64)]),
    ("kdf_reader_last_ordinals", c_int, [_P, POINTER(POINTER(c_uint64))]),
    ("kdf_bam_write_subset", c_int, [c_char_p, c_char_p, _P, c_uint64, _P, _P, c_int, c_int, POINTER(c_uint64)]),
    ("kdf_reader_ref_count", c_int, [_P]),
    ("kdf_reader_ref_name", c_char_p, [_P, c_int]),
    ("kdf_reader_close", None, [_P]),
    ("kdf_reader_error", c_char_p, [_P]),
]

_lib = None


class KdfError(RuntimeError):
    """Engine failure; mirrors the reference's RuntimeError("jellyfish ... failed: ...")."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"kdf engine failed ({code}): {msg}")
        self.code = code


def _preload_torch_hip():
    """PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  If
    libkdf.so pulled /opt/rocm's copies first, a later `import torch` would load
    a SECOND HSA runtime into the process and find no device.  Loading torch's
    copies first (same SONAMEs) makes both sides share one runtime, in either
    import order.  torch itself is not imported and the GPU is not touched."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


def load() -> ctypes.CDLL:
    """Load libkdf.so, building it first if the source tree has hipcc and no .so."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip()
    if not os.path.exists(LIB_PATH):
        from .build import build_native
        try:
            build_native()
        except Exception as e:  # noqa: BLE001
            raise ImportError(
                f"libkdf.so is not built ({LIB_PATH}) and building it failed: {e}. "
                "Run `python -m kmer_denovo_filter_amd.build`. There is no CPU fallback."
            ) from e
    lib = ctypes.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, handle=None):
    if rc != KDF_OK:
        lib = load()
        msg = lib.kdf_last_error(handle)
        raise KdfError(rc, msg.decode(errors="replace") if msg else "unknown error")


def check_reader(rc: int, reader=None):
    if rc != KDF_OK:
        lib = load()
        msg = lib.kdf_reader_error(reader)
        raise KdfError(rc, msg.decode(errors="replace") if msg else "unknown error")
