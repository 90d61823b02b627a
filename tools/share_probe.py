#!/usr/bin/env python3
"""One GPU's share of a voice-sharded job: wall time per back-to-back 4800-frame call of V voices x P partials through the
device entry point.  The FR_* shape switches are read once per process, so a sweep starts one process per setting:
    python tools/share_probe.py sweep            # 8 / 16 / 32 voices x 4096, the settings in SWEEP
    python tools/share_probe.py V P [T [triangle]]   # one measurement with the current environment"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SWEEP = {
    8: [{}, {"FR_SHORT_WGS": "1200", "FR_SHORT_NW": "4"}, {"FR_SHORT_WGS": "2400", "FR_SHORT_NW": "4"}, {"FR_BANK_SHORT": "0"},
        {"FR_BANK_SHORT": "0", "FR_BANK_NW": "4"}, {"FR_BANK_SHORT": "0", "FR_BANK_NW": "16"}],
    16: [{}, {"FR_BANK_NW": "4"}, {"FR_BANK_NW": "16"}, {"FR_SHORT_PAIRS": "1300", "FR_SHORT_WGS": "2400", "FR_SHORT_NW": "4"},
         {"FR_SHORT_PAIRS": "1300", "FR_SHORT_WGS": "2400", "FR_SHORT_NW": "8"}],
    32: [{}, {"FR_BANK_NW": "4"}, {"FR_BANK_NW": "16"}],
}


def one(V, P, T):
    import torch
    import libfriendship_amd
    from libfriendship_amd import synth
    hip = libfriendship_amd.HipRenderer()
    if len(sys.argv) > 4 and sys.argv[4] == "triangle":   # a leaf the hand-written kernel does not know: hipRTC-specialised
        g = synth.GraphArrays()
        p = synth.voice_params(V, P, seed=P + 1, detune=True)
        import numpy as np
        g.edge(synth.sum_tree(g, synth.triangle_leaves(g, p["w"], p["amp"]).reshape(V, P)), 0, 0, np.arange(V, dtype=np.uint32))
        synth.install(hip, g.finish(V))
    else:
        synth.install(hip, synth.additive_tree(V, P))
    d_time = torch.arange(0, 1 << 16, dtype=torch.float32, device="cuda")
    d_out = torch.empty((V, T), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    idx = 0
    res = []
    for rep in range(4):
        N = 300
        for k in range(60 if rep == 0 else 0):
            hip.fill_buffer_device(d_out.data_ptr(), V, T, idx, d_time.data_ptr(), [0, T], stream); idx += T
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(N):
            hip.fill_buffer_device(d_out.data_ptr(), V, T, idx, d_time.data_ptr(), [0, T], stream); idx += T
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / N * 1e6)
    ideal = V * P * T / (78.64e12 / 6) * 1e6
    s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
    o2 = [torch.empty((V, T), dtype=torch.float32, device="cuda") for _ in range(2)]
    two = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(300):
            hip.fill_buffer_device(o2[k % 2].data_ptr(), V, T, idx, d_time.data_ptr(), [0, T], s2[k % 2].cuda_stream); idx += T
        torch.cuda.synchronize()
        two.append((time.perf_counter() - t0) / 300 * 1e6)
    print(f"{V:3d} x {P} x {T}: two streams {min(two):7.2f} us;  one stream {min(res):7.2f} us (median {sorted(res)[len(res) // 2]:7.2f})  VALU-ideal {ideal:6.2f}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "sweep":
        for V, settings in SWEEP.items():
            for env in settings:
                print(f"  {env or 'default'}", flush=True)
                subprocess.run([sys.executable, __file__, str(V), "4096"], env=dict(os.environ, **env), check=False)
    else:
        one(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 4800)
