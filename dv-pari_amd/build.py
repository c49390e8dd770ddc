"""Builds libdvpari_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdvpari_hip.so")
SOURCES = ["capi.cpp", "cache.cpp", "tree_io.cpp", "ecfft.hip", "msm.hip", "codec.hip", "fr_ops.hip", "prove.hip", "setup.hip"]
HEADERS = ["common.h", "fr.cuh", os.path.join("..", "..", "include", "dvpari.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, s.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-c", "-x", "hip",
               os.path.join(CSRC, s), "-o", obj, "-Wno-pass-failed", "-Wno-int-to-pointer-cast"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
