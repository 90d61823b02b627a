"""libfriendship_amd -- MI355X-native render engine behind libfriendship's `render::Renderer` surface.

Layout:
  csrc/        HIP kernels (gfx950) + host-side graph lowering + the C ABI (include/friendship_render.h)
  host/        C++ mirror of the reference's routing/dispatch surface (RouteGraph, Effect, Dispatch)
  capi.py      ctypes binding of the C ABI (marshalling only)
  synth.py     synthetic additive-synthesis trees (BASELINE.json configs) as primitive graphs

The compute path is libfriendship_hip.so.  There is no CPU fallback: if the library is missing or no
gfx950 device is present, creating a renderer raises.
"""
import os

from .capi import (Effect, RenderError, Renderer, RendererLib, f32_bits, FR_PRIM, PRIMITIVES)  # noqa: F401

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
HIP_LIB_PATH = os.path.join(PKG_DIR, "libfriendship_hip.so")

_hip_lib = None


def hip_lib():
    """The product library.  Raises FileNotFoundError if it has not been built."""
    global _hip_lib
    if _hip_lib is None:
        # torch's ROCm wheel bundles its own libamdhip64 under the same SONAME as /opt/rocm's.  Whichever
        # is loaded first serves the whole process, and torch cannot see the GPU through the system copy.
        # Device buffers and streams are shared with torch (bench.py, tests), so let torch load first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip_lib = RendererLib(HIP_LIB_PATH)
        if _hip_lib.backend != "hip-gfx950":
            raise RuntimeError(f"{HIP_LIB_PATH} reports backend {_hip_lib.backend!r}, expected 'hip-gfx950'")
    return _hip_lib


def HipRenderer(mode="auto", device=-1):
    """A renderer on the HIP engine (the analogue of `SparkleRenderer::default()` in the reference's tests)."""
    return Renderer(hip_lib(), mode=mode, device=device)
