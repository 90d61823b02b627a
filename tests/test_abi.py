"""CPU-side checks of the drop-in boundary: the product library loads, exports every symbol the header
declares, and refuses to run without a gfx950 device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "friendship_render.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fr_[a-z_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = header_symbols()
    for need in ("fr_renderer_create", "fr_renderer_destroy", "fr_on_add_node", "fr_on_del_node",
                 "fr_on_add_edge", "fr_on_del_edge", "fr_fill_buffer", "fr_fill_buffer_device"):
        assert need in syms


def test_hip_library_exports_every_header_symbol(hip_lib):
    for s in header_symbols():
        assert hasattr(hip_lib.lib, s), f"libfriendship_hip.so does not export {s}"
    assert hip_lib.backend == "hip-gfx950"


def test_oracle_exports_the_same_abi(oracle_lib):
    for s in header_symbols():
        assert hasattr(oracle_lib.lib, s)
    assert oracle_lib.backend == "cpu-oracle"


def test_product_does_not_link_the_oracle(hip_lib):
    import subprocess
    out = subprocess.run(["ldd", hip_lib.path], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", hip_lib.path], capture_output=True, text=True).stdout
    assert "fro_" not in syms


def test_no_cpu_fallback_without_a_gpu(hip_lib):
    """On a box without a GPU, creating a renderer must fail loudly (FR_ERR_NO_DEVICE), never compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from libfriendship_amd.capi import FR_ERR_NO_DEVICE, RenderError, Renderer
    with pytest.raises(RenderError) as ei:
        Renderer(hip_lib)
    assert ei.value.status == FR_ERR_NO_DEVICE
