#!/usr/bin/env python3
"""Soak of the host entry point's streamed output (kernels store rows into mapped pinned memory and publish a flag per
finished row; the host copies rows out while the launch still runs): every call's buffer against the same frames
rendered through the device entry point.  usage: python tools/host_stream_soak.py [voices partials calls]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import libfriendship_amd
from libfriendship_amd import synth

V, P, CALLS = (int(x) for x in (sys.argv[1:4] + ["64", "4096", "1500"][len(sys.argv) - 1:]))
tree = synth.additive_tree(V, P, seed=3, detune=True)
a, b = libfriendship_amd.HipRenderer(), libfriendship_amd.HipRenderer()
synth.install(a, tree)
synth.install(b, tree)
stream = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(1)
noise = os.environ.get("FR_SOAK_NOISE") == "1"
if noise:
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
idx = bad = 0
t0 = time.time()
for k in range(CALLS):
    T = int(rng.choice([4800, 4800, 2048, 1000, 777]))
    row = ((np.arange(idx, idx + T) % (1 << 22)) + rng.integers(0, 5)).astype(np.float32)
    d_row = torch.from_numpy(row).cuda()
    d_out = torch.empty((V, T), dtype=torch.float32, device="cuda")
    a.fill_buffer_device(d_out.data_ptr(), V, T, idx, d_row.data_ptr(), [0, T], stream)
    if noise and k % 2:
        with torch.cuda.stream(side):
            big.add_(1.0)
    out = np.full((V, T), np.float32(-7.0))       # a fresh (cold, pageable) buffer every call, like Array2::zeros
    b.fill_buffer(V, idx, idx + T, [row], out=out)
    ref = d_out.cpu().numpy()
    if not np.array_equal(out.view(np.uint32), ref.view(np.uint32)):
        bad += 1
        if bad < 5:
            w = np.argwhere(out.view(np.uint32) != ref.view(np.uint32))
            print(f"call {k} (T = {T}): {len(w)} samples differ, first at {tuple(w[0])}: {out[tuple(w[0])]} vs {ref[tuple(w[0])]}")
    idx += T
print(f"{V} x {P}: {CALLS} host calls ({time.time() - t0:.1f} s), {bad} with a mismatch; plan: streamed rows = "
      f"{os.environ.get('FR_HOST_STREAM', '1') != '0'}")
sys.exit(1 if bad else 0)
