"""Oracle-only entry points (oracle/ref_renderer.cpp `fro_*`), bound for tests and the bench's CPU leg."""
import ctypes as C

import numpy as np


def eval_samples(oracle_renderer, slots, times):
    """out[i] = oracle get_sample(times[i], slots[i]) against its stored input history (random access)."""
    L = oracle_renderer.L
    L.fro_eval_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.fro_eval_samples.restype = C.c_int32
    slots = np.ascontiguousarray(slots, dtype=np.uint32)
    times = np.ascontiguousarray(times, dtype=np.uint64)
    out = np.zeros(len(slots), dtype=np.float32)
    oracle_renderer._check(L.fro_eval_samples(oracle_renderer.h, slots.ctypes.data, times.ctypes.data, len(slots), out.ctypes.data))
    return out


def set_threads(oracle_renderer, n):
    L = oracle_renderer.L
    L.fro_set_threads.argtypes = [C.c_void_p, C.c_uint32]
    L.fro_set_threads.restype = C.c_int32
    oracle_renderer._check(L.fro_set_threads(oracle_renderer.h, n))
