import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_oracle():
    src = os.path.join(ROOT, "oracle", "ref_renderer.cpp")
    if not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    return ORACLE_LIB


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure): same C ABI as the product."""
    from libfriendship_amd.capi import RendererLib
    return RendererLib(_build_oracle())


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  GPU tests must use this and only this for the thing under test."""
    import libfriendship_amd
    return libfriendship_amd.hip_lib()


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def selfcheck():
    """Self-consistency vectors (rendered by the oracle when they were made; NOT reference outputs)."""
    with open(os.path.join(ROOT, "tests", "golden", "selfcheck_vectors.json")) as f:
        return json.load(f)
