#!/usr/bin/env python3
"""Latency of the reference-shaped call fr_fill_buffer (host rows in, host buffer out, synchronous) for SHORT blocks --
what a real-time host pays per audio block.  usage: python tools/host_short_probe.py [voices partials [T,T,...]]
(FR_HOST_TRACE=1 with ONE block length prints where the time goes: issue / wait / copy)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import libfriendship_amd
from libfriendship_amd import synth

V, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 4096)
r = libfriendship_amd.HipRenderer()
synth.install(r, synth.additive_tree(V, P))
idx = 0
print(f"{V} x {P}, FR_HOST_MAPPED={os.environ.get('FR_HOST_MAPPED', 'default')}")
for T in ([int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (1, 16, 64, 128, 256, 512, 1024)):
    out = np.zeros((V, T), np.float32)
    rows = [synth.time_ramp(k * T, (k + 1) * T) for k in range(8)]
    ts = []
    for k in range(600):
        t0 = time.perf_counter()
        r.fill_buffer(V, idx, idx + T, [rows[k % 8]], out=out)
        ts.append((time.perf_counter() - t0) * 1e6)
        idx += T
    ts = np.array(ts[100:])
    print(f"T={T:5d}: median {np.median(ts):7.1f} us   p90 {np.percentile(ts, 90):7.1f}   ({T / 48000 * 1e6:8.1f} us of audio at 48 kHz)")
