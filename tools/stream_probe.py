#!/usr/bin/env python3
"""Host-to-host latency of one block through the resident launch (fr_stream_block) beside the same block through
fr_fill_buffer.  usage: python tools/stream_probe.py [voices partials]"""
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
for T in (1, 16, 64):
    rows = [synth.time_ramp(k * T, (k + 1) * T) for k in range(8)]
    out = np.zeros((V, T), np.float32)
    idx = 0
    fb = {}
    for gap_us in (0, 300, 1300):
        a = []
        for k in range(600):
            if gap_us:
                t1 = time.perf_counter()
                while (time.perf_counter() - t1) * 1e6 < gap_us:
                    pass
            t0 = time.perf_counter()
            r.fill_buffer(V, idx, idx + T, [rows[k % 8]], out=out)
            a.append((time.perf_counter() - t0) * 1e6)
            idx += T
        fb[gap_us] = a
    r.stream_begin(V)
    for gap_us in (0, 300, 1300):
        a = fb[gap_us]
        b = []
        for k in range(600):
            if gap_us:
                t1 = time.perf_counter()
                while (time.perf_counter() - t1) * 1e6 < gap_us:
                    pass
            t0 = time.perf_counter()
            r.stream_block(idx, rows[k % 8], out=out)
            b.append((time.perf_counter() - t0) * 1e6)
            idx += T
        print(f"T={T:3d}, {gap_us:4d} us idle between blocks: fr_fill_buffer median {np.median(a[100:]):6.1f} us p99 {np.percentile(a[100:], 99):6.1f} | fr_stream_block median {np.median(b[100:]):6.1f} us p99 {np.percentile(b[100:], 99):6.1f}")
    r.stream_end()
