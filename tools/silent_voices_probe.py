"""Throughput of the bank kernel on voices whose mix is identically zero (all amplitudes 0): every frame takes the
zero-sign path.  python tools/silent_voices_probe.py [voices partials]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import libfriendship_amd
from libfriendship_amd import synth

V = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = 4800
for label, scale in (("sounding", 1.0), ("silent (amp = 0)", 0.0)):
    p = synth.voice_params(V, P, 0x5EED0002)
    if V > 64:   # keep the fundamentals in the audible range: 55 * 2^((v mod 64) / 12), not 55 * 2^(v / 12)
        base = synth.voice_params(64, P, 0x5EED0002)
        detune = (1.0 + 1e-4 * (np.arange(V) // 64)).astype(np.float32)[:, None]   # distinct voices (identical ones are merged at lowering)
        p = {"w": (np.tile(base["w"], (V // 64 + 1, 1))[:V] * detune).astype(np.float32), "amp": np.tile(base["amp"], (V // 64 + 1, 1))[:V]}
    g = synth.GraphArrays()
    leaves = synth.partial_leaves(g, p["w"], p["amp"] * np.float32(scale)).reshape(V, P)
    g.edge(synth.sum_tree(g, leaves), 0, 0, np.arange(V, dtype=np.uint32))
    tree = g.finish(V)
    r = libfriendship_amd.HipRenderer()
    synth.install(r, tree)
    d_t = torch.from_numpy(synth.time_ramp(0, 64 * T) % (1 << 23)).cuda()
    d_out = torch.empty((V, T), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    def run(k0, n):
        for k in range(k0, k0 + n):
            row = d_t[(k % 64) * T:][:T]
            r.fill_buffer_device(d_out.data_ptr(), V, T, k * T, row.data_ptr(), [0, T], s)
        torch.cuda.synchronize()
    run(0, 10)
    t0 = time.perf_counter(); run(10, 50); dt = (time.perf_counter() - t0) / 50
    print(f"{label:18s} {V} x {P}: {dt * 1e6:8.1f} us per {T}-frame call  {T / dt / 1e6:7.2f} Msamples/s")
    r.close()
