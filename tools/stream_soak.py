#!/usr/bin/env python3
"""Soak of the resident launch (fr_stream_block): tens of thousands of blocks of random length 1..64 with ordinary and
hostile input rows, every sample compared with what a second renderer's fr_fill_buffer renders for the same frames (that
path is itself checked against the oracle by the parity tests).  FR_SOAK_NOISE=1: matrix products and 256 MB sweeps run on
a side stream meanwhile.  usage: python tools/stream_soak.py [voices partials [blocks]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import libfriendship_amd
from libfriendship_amd import synth

V, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 4096)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
tree = synth.additive_tree(V, P)
a, b = libfriendship_amd.HipRenderer(), libfriendship_amd.HipRenderer()
synth.install(a, tree)
synth.install(b, tree)
noise = os.environ.get("FR_SOAK_NOISE") == "1"
if noise:
    import torch
    side = torch.cuda.Stream()
    m = torch.randn(4096, 4096, device="cuda")
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
rng = np.random.default_rng(7)
special = np.array([0.0, -0.0, -1.0, 0.5, 1e-42, 16777216.0, 4294967296.0, 1e30, np.inf, -np.inf, np.nan], np.float32)
# The resident launch fills every CU's register file (16 waves x 120 VGPRs): nothing else runs on the device while a stream is
# open, so the expected blocks are rendered first, a chunk at a time, and the stream is opened for the comparison.
idx = bad = done = 0
while done < N:
    blocks = []
    for k in range(done, min(done + 1500, N)):
        T = int(rng.integers(1, 65))
        row = synth.time_ramp(idx % (1 << 24), idx % (1 << 24) + T)
        if k % 7 == 6:
            row = row.copy()
            row[rng.integers(T, size=max(1, T // 5))] = special[rng.integers(len(special), size=max(1, T // 5))]
        blocks.append((idx, row, b.fill_buffer(V, idx, idx + T, [row])))
        idx += T
    a.stream_begin(V)
    for k, (i0, row, exp) in enumerate(blocks):
        if noise and k % 50 == 0:
            with torch.cuda.stream(side):     # (queued behind the resident launch: it runs when the stream closes)
                m2 = m @ m
                big.fill_(1.0)
        got = a.stream_block(i0, row)
        same = np.array_equal(np.where(np.isnan(got), np.uint32(0x7FC00000), got.view(np.uint32)), np.where(np.isnan(exp), np.uint32(0x7FC00000), exp.view(np.uint32)))
        bad += 0 if same else 1
    a.stream_end()
    done += len(blocks)
    if done % 12000 < 1500:
        print(f"{done} blocks, {bad} mismatching", flush=True)
print(f"done: {V} x {P}: {N} blocks, {idx} frames, {bad} mismatching blocks" + (" (with work queued behind the launch)" if noise else ""))
sys.exit(1 if bad else 0)
