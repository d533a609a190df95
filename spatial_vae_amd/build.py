"""Compile the HIP C-ABI library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = os.path.join(_HERE, "libsvae_hip.so")
_SOURCES = tuple(sorted(f for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))))   # every file api.hip may include


def library_path():
    return _LIB


def _stale():
    if not os.path.exists(_LIB):
        return True
    t = os.path.getmtime(_LIB)
    deps = [os.path.join(_CSRC, s) for s in _SOURCES] + [os.path.join(_HERE, "..", "include", "svae.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 csrc/api.hip -> libsvae_hip.so next to this file."""
    if not force and not _stale():
        return _LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           os.path.join(_CSRC, "api.hip"), "-o", _LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=_CSRC)
    os.replace(_LIB + ".tmp", _LIB)
    return _LIB
