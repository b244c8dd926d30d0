"""Build libkdf.so (HIP kernels + C ABI) in-tree for gfx950.

``python -m kmer_denovo_filter_amd.build`` or ``build_native()``.  hipcc
cross-compiles without a GPU; the resulting ``kmer_denovo_filter_amd/libkdf.so``
is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_INC = os.path.join(os.path.dirname(_PKG), "include")
LIB_PATH = os.path.join(_PKG, "libkdf.so")

_SOURCES = [
    # DPP wave scans for uniform-address atomics with per-lane values: this toolchain's default is an iterative
    # loop over the active lanes (~8 instructions per lane), which made the bucket kernels scalar-issue bound
    ("kdf_engine.hip", ["--offload-arch=gfx950", "-O3", "-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP"]),
    ("kdf_sort.hip", ["--offload-arch=gfx950", "-O3"]),
    ("kdf_host.cpp", ["-O2", "-x", "c++"]),          # host only: no device pass
]
# every header any source includes: a header-only edit must trigger a rebuild
_DEPS = ["kdf_device.h", "kdf_binned.h", "kdf_merge.h", os.path.join(_INC, "kdf.h")]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "hipcc")
    extra = os.environ.get("KDF_EXTRA_FLAGS", "").split()      # experiments: -DSK_C_THREADS=512 ...
    deps = [d if os.path.isabs(d) else os.path.join(_CSRC, d) for d in _DEPS]
    objs = []
    for src, flags in _SOURCES:
        s = os.path.join(_CSRC, src)
        o = os.path.join(_CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if not force and _newer(o, [s] + deps):
            continue
        cmd = [hipcc, *flags, *extra, "-fPIC", "-std=c++17", f"-I{_INC}", f"-I{_CSRC}", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    if force or not _newer(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs, "-lz"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
