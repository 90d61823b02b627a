#!/usr/bin/env python3
"""Soak test of the short-call kernel's in-launch combine (a cross-workgroup hand-off through HBM): the same frames
rendered in long calls (time-major kernel) and in thousands of short calls of random length (chunked kernel, tickets),
every sample compared.  usage: python tools/short_kernel_soak.py [voices partials total_frames seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import libfriendship_amd
from libfriendship_amd import synth

V, P, N, seed = (int(x) for x in (sys.argv[1:5] + ["64", "4096", "240000", "1"][len(sys.argv) - 1:]))
tree = synth.additive_tree(V, P, seed=seed, detune=True)
t = torch.from_numpy((np.arange(N) % (1 << 20)).astype(np.float32)).cuda()
stream = torch.cuda.current_stream().cuda_stream
a, b = libfriendship_amd.HipRenderer(), libfriendship_amd.HipRenderer()
synth.install(a, tree)
synth.install(b, tree)
whole = torch.empty((V, N), dtype=torch.float32, device="cuda")
for s in range(0, N, 4800):          # reference: long calls, written column block by column block
    n = min(4800, N - s)
    blk = torch.empty((V, n), dtype=torch.float32, device="cuda")
    a.fill_buffer_device(blk.data_ptr(), V, n, s, t.data_ptr() + 4 * s, [0, n], stream)
    whole[:, s:s + n] = blk
torch.cuda.synchronize()
rng = np.random.default_rng(seed)
# uneven load while the hand-offs run (FR_SOAK_NOISE=1): large copies and matrix products on a side stream
noise = os.environ.get("FR_SOAK_NOISE") == "1"
if noise:
    side = torch.cuda.Stream()
    na = torch.randn(2048, 2048, device="cuda")
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
bad = calls = 0
t0 = time.time()
s = 0
pending = []
while s < N:
    n = int(min(N - s, rng.choice([1, 2, 7, 31, 64, 100, 128, 200, 256])))
    blk = torch.empty((V, n), dtype=torch.float32, device="cuda")
    b.fill_buffer_device(blk.data_ptr(), V, n, s, t.data_ptr() + 4 * s, [0, n], stream)
    pending.append((s, n, blk))
    s += n
    calls += 1
    if noise and calls % 3 == 0:
        with torch.cuda.stream(side):
            if calls % 2:
                na = (na @ na).clamp_(-1, 1)
            else:
                big.add_(1.0)
    if len(pending) == 256 or s >= N:
        torch.cuda.synchronize()
        for (s0, n0, bk) in pending:
            ref = whole[:, s0:s0 + n0]
            same = (bk.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(bk) & torch.isnan(ref))
            if not bool(same.all()):
                bad += 1
                if bad < 5:
                    print(f"MISMATCH in call [{s0}, {s0 + n0}): {int((~same).sum())} samples")
        pending = []
print(f"{V} x {P}: {N} frames in {calls} short calls ({time.time() - t0:.1f} s), {bad} calls with a mismatch")
sys.exit(1 if bad else 0)
