#!/usr/bin/env python3
"""Short fill_buffer calls at config C: wall time per back-to-back call through the device entry point, kernel time from
HIP events around the launch (fr_set_timing), and the host's own time to issue a call (no GPU wait).
usage: python tools/short_call_probe.py [voices partials]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import libfriendship_amd
from libfriendship_amd import synth

V, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 4096)
tree = synth.additive_tree(V, P)
hip = libfriendship_amd.HipRenderer()
synth.install(hip, tree)
d_time = torch.arange(0, 1 << 16, dtype=torch.float32, device="cuda")
d_out = torch.empty((V, 4800), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
idx = 0
print(f"{V} voices x {P} partials; peak = {V * P / (78.64e12 / 6) * 1e6:.4f} us per frame of VALU work")
for T in (1, 8, 32, 64, 128, 256, 512, 1024, 4800):
    N = 400

    def call(k):
        global idx
        hip.fill_buffer_device(d_out.data_ptr(), V, T, idx, d_time.data_ptr() + 4 * ((k * T) % 4096), [0, T], stream)
        idx += T

    for k in range(30):
        call(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        call(k)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / N * 1e6
    hip.set_timing(True)
    hip.reset_timing()
    for k in range(100):
        call(k)
    torch.cuda.synchronize()
    ms, n = hip.get_timing("all")
    hip.set_timing(False)
    ideal = V * P * T / (78.64e12 / 6) * 1e6
    print(f"T={T:5d}: wall {wall:7.2f} us/call   host issue {t_issue / N * 1e6:6.2f} us/call   kernel (events) {ms / n * 1e3:7.2f} us x {n / 100:.0f} launches   "
          f"VALU-ideal {ideal:6.2f} us   valu_frac(wall) {ideal / wall:.3f}")
