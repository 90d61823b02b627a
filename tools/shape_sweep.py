"""Bank throughput over voice shapes: V voices x P partials with V*P fixed, 4800-frame and 512-frame calls (device entry
point, audible fundamentals for any V).   python tools/shape_sweep.py [total_partials [P1,P2,...]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import libfriendship_amd
from libfriendship_amd import synth

total = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
SHAPES = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (32, 64, 128, 256, 512, 1000, 1024, 2048, 4096, 8192, 16384)
for P in SHAPES:
    V = max(1, total // P)
    base = synth.voice_params(min(V, 64), P, 0x5EED0002)
    reps = V // 64 + 1
    detune = (1.0 + 1e-4 * (np.arange(V) // 64)).astype(np.float32)[:, None]
    w = (np.tile(base["w"], (reps, 1))[:V] * detune).astype(np.float32)
    amp = np.tile(base["amp"], (reps, 1))[:V]
    g = synth.GraphArrays()
    leaves = synth.partial_leaves(g, w, amp).reshape(V, P)
    g.edge(synth.sum_tree(g, leaves), 0, 0, np.arange(V, dtype=np.uint32))
    tree = g.finish(V)
    r = libfriendship_amd.HipRenderer()
    synth.install(r, tree)
    line = f"{V:6d} x {P:6d}:"
    for T in (4800, 512):
        d_t = torch.from_numpy(synth.time_ramp(0, 64 * T) % (1 << 23)).cuda()
        d_out = torch.empty((V, T), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        def run(k0, n, base_idx):
            for k in range(k0, k0 + n):
                row = d_t[(k % 64) * T:][:T]
                r.fill_buffer_device(d_out.data_ptr(), V, T, base_idx + k * T, row.data_ptr(), [0, T], s)
            torch.cuda.synchronize()
        base_idx = 0 if T == 4800 else 10_000_000
        run(0, 10, base_idx)
        t0 = time.perf_counter(); run(10, 40, base_idx); dt = (time.perf_counter() - t0) / 40
        line += f"   T={T}: {dt * 1e6:8.1f} us  {V * P * T / dt / 1e12:5.2f} Tpf/s"
    print(line, flush=True)
    r.close()
