"""The C++ host mirror (libfriendship_amd/host/friendship.hpp) running the reference's own integration tests
(tests/cpp/render_tests.cpp transcribes tests/render_prim.rs, tests/ext_input.rs, tests/load_effect.rs)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "render_tests.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "render_tests")
HDR = os.path.join(ROOT, "libfriendship_amd", "host", "friendship.hpp")
ABI = os.path.join(ROOT, "include", "friendship_render.h")


def build():
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(p) for p in (SRC, HDR, ABI)):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-o", BIN, SRC, "-ldl"], check=True)
    return BIN


def run(lib):
    env = dict(os.environ, FRIENDSHIP_RENDERER_LIB=lib)
    p = subprocess.run([build()], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "19 passed; 0 failed" in p.stdout, p.stdout


def test_reference_tests_through_cpp_dispatch_on_oracle(oracle_lib):
    run(oracle_lib.path)


@pytest.mark.gpu
def test_reference_tests_through_cpp_dispatch_on_hip(hip_lib):
    run(hip_lib.path)


# ---- engine host logic (mirror, lowering, bank recognition, staged planning) without a GPU ---------------------------
PLAN_SRC = os.path.join(ROOT, "tests", "cpp", "plan_tests.cpp")
PLAN_BIN = os.path.join(ROOT, "tests", "cpp", "_build", "plan_tests")
CSRC = os.path.join(ROOT, "libfriendship_amd", "csrc")


def test_engine_host_logic_against_oracle(oracle_lib):
    """tests/cpp/plan_tests.cpp: the lowered graph and the staged plan (banks, rings, level and fused programs,
    general-tree schedules), executed by small CPU interpreters in the test, equal the oracle bit for bit; the source
    generated for compiled stage programs, built with g++, does too."""
    deps = [PLAN_SRC, ABI] + [os.path.join(CSRC, f) for f in ("graph.cpp", "graph.hpp", "match.cpp", "match.hpp", "stage.cpp", "stage.hpp", "stagejit.cpp", "leafjit.cpp", "range.hpp", "jit.hpp", "kernels.hpp")]
    if not os.path.exists(PLAN_BIN) or os.path.getmtime(PLAN_BIN) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(PLAN_BIN), exist_ok=True)
        subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                        "-Wno-subobject-linkage", "-pthread", "-o", PLAN_BIN, PLAN_SRC, "-ldl"], check=True)
    env = dict(os.environ, FRIENDSHIP_ORACLE_LIB=oracle_lib.path)
    p = subprocess.run([PLAN_BIN], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "16 passed; 0 failed" in p.stdout, p.stdout + p.stderr
