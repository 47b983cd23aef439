"""In-tree build of libsa_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m suffixarray_amd.build [--force]
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libsa_hip.so")
SOURCES = ["sa_capi.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "sa_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


def build_cython(force=False, verbose=False):
    """The Cython binding (suffix_array.pyx -> suffixarray_amd/suffix_array.*.so), in-tree."""
    import glob
    pyx = os.path.join(_HERE, "suffix_array.pyx")
    built = glob.glob(os.path.join(_HERE, "suffix_array.*.so"))
    if built and not force and all(os.path.getmtime(b) >= max(os.path.getmtime(pyx), os.path.getmtime(LIB)) for b in built):
        return built[0]
    cmd = [sys.executable, os.path.join(_HERE, "setup_cython.py"), "build_ext", "--inplace"]
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return glob.glob(os.path.join(_HERE, "suffix_array.*.so"))[0]


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv, verbose=True)
    print(LIB)
    print(build_cython(force="--force" in sys.argv, verbose=True))
