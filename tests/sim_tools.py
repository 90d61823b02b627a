"""Builds and loads the host-logic simulator (TEST INFRASTRUCTURE): the engine's own host sources (csrc/engine.cpp,
graph.cpp, match.cpp, stage.cpp, leafjit.cpp, stagejit.cpp) compiled with g++ against host-memory stand-ins for the
HIP runtime (tests/cpp/sim/sim_hip.cpp) and plain-loop restatements of the kernels (tests/cpp/sim/sim_kernels.cpp).
It exports the same C ABI, so CPU tests -- including the world_size > 1 gloo tests -- drive the REAL planner, input
store, ring/window logic and shard exchange through fr_* calls.  It is never loaded by the package, by `-m gpu` tests
(those run the HIP library), by bench.py or by smoke()."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libfriendship_amd", "csrc")
SIM = os.path.join(ROOT, "tests", "cpp", "sim")
OUT = os.path.join(ROOT, "tests", "cpp", "_build", "libfr_simengine.so")
ENGINE_SOURCES = ["engine.cpp", "graph.cpp", "match.cpp", "stage.cpp", "leafjit.cpp", "stagejit.cpp"]
SIM_SOURCES = ["sim_hip.cpp", "sim_kernels.cpp"]


def build_sim():
    if os.environ.get("FR_SIM_LIB"):          # a pre-built variant (e.g. an AddressSanitizer build), used as is
        return os.environ["FR_SIM_LIB"]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(SIM, f) for f in SIM_SOURCES]
    deps.append(os.path.join(ROOT, "include", "friendship_render.h"))
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-I/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", "-Wall", "-Wno-subobject-linkage", "-Wno-unused-result", "-o", OUT]
    cmd += [os.path.join(CSRC, f) for f in ENGINE_SOURCES] + [os.path.join(SIM, f) for f in SIM_SOURCES]
    subprocess.run(cmd, check=True)
    return OUT


def sim_lib():
    from libfriendship_amd.capi import RendererLib
    return RendererLib(build_sim())
