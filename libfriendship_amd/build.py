"""Builds libfriendship_hip.so (in-tree, so it travels to the GPU box with the snapshot).

hipcc cross-compiles gfx950 without a GPU.  Flags that matter for results:
  -ffp-contract=off                       every f32 op rounds once (parity with the reference)
  -fno-slp-vectorize                      keep scalar f32 VALU ops; v_pk_* f32 brings no rate gain on gfx950
  -mllvm -simplifycfg-sink-common=false   keeps the bank kernel's carry registers out of scratch memory
"""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OUT = os.path.join(PKG_DIR, "libfriendship_hip.so")
SOURCES = ["engine.cpp", "graph.cpp", "match.cpp", "stage.cpp", "jit.cpp", "leafjit.cpp", "stagejit.cpp", "comm_rccl.cpp", "kernels.hip"]
HEADERS = ["graph.hpp", "kernels.hpp", "match.hpp", "stage.hpp", "jit.hpp", "leafshape.hpp", "comm.hpp", "range.hpp", os.path.join("..", "..", "include", "friendship_render.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-slp-vectorize", "-mllvm", "-simplifycfg-sink-common=false", "-Wall", "-Wextra"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lhiprtc", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
