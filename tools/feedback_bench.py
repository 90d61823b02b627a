#!/usr/bin/env python3
"""Cost of feedback through Delay on the device: V additive voices of P partials, each through a comb filter
x = voice + g * Delay(x, d) -- a loop per voice, evaluated by one stage program per voice whose threads stride by d frames.
Steady 4800-frame calls (device entry point), a seek (the loops' state is rebuilt by replay from frame 0), and the same patch
without the loops (feed-forward tap) for comparison.
    python tools/feedback_bench.py [--voices 64 --partials 1024 --delays 1,32,441,2400]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from libfriendship_amd import hip_lib, synth  # noqa: E402
from libfriendship_amd.capi import Renderer  # noqa: E402


def comb_tree(V, P, d, feedback=True):
    g = synth.GraphArrays()
    p = synth.voice_params(V, P, 0x5EED0300, wrap=64)
    voices = synth.sum_tree(g, synth.partial_leaves(g, p["w"], p["amp"]).reshape(V, P))
    x = g.nodes(synth.K_SUM2, V)
    dl = g.nodes(synth.K_DELAY, V)
    m = g.binop(synth.K_MUL, dl, synth.C(np.float32(0.6)), V)
    g.edge(voices, x, 0, 0)
    g.edge(m, x, 0, 1)
    g.edge(x if feedback else voices, dl, 0, 0)      # the loop: the Delay reads x itself (feed-forward: the voice)
    g.const(dl, np.float32(d), 1)
    g.edge(x, 0, 0, np.arange(V, dtype=np.uint32))
    return g.finish(V)


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, default=64)
    ap.add_argument("--partials", type=int, default=1024)
    ap.add_argument("--delays", default="1,32,441,2400")
    ap.add_argument("--frames", type=int, default=4800)
    a = ap.parse_args()
    V, P, T = a.voices, a.partials, a.frames
    s = torch.cuda.current_stream().cuda_stream
    d_out = torch.empty((V, T), dtype=torch.float32, device="cuda")
    for d in [int(x) for x in a.delays.split(",")]:
        for fb in (True, False):
            tree = comb_tree(V, P, d, fb)
            with Renderer(hip_lib()) as r:
                synth.install(r, tree)
                d_t = torch.from_numpy(synth.time_ramp(0, 400 * T)).cuda()

                def call(k):
                    r.fill_buffer_device(d_out.data_ptr(), V, T, k * T, d_t[(k % 400) * T:].data_ptr(), [0, T], s)

                for k in range(10):
                    call(k)
                torch.cuda.synchronize()
                r.set_timing(True)
                r.reset_timing()
                t0 = time.perf_counter()
                for k in range(10, 60):
                    call(k)
                torch.cuda.synchronize()
                step = (time.perf_counter() - t0) / 50
                bank_ms, nb = r.get_timing("bank")
                stage_ms, ns = r.get_timing("stage")
                r.set_timing(False)
                plan = r.plan()
                # a seek: to frame 48000 * 10 (ten seconds in): feedback replays from 0
                t0 = time.perf_counter()
                r.fill_buffer_device(d_out.data_ptr(), V, T, 480000, d_t[:T].data_ptr(), [0, T], s)
                torch.cuda.synchronize()
                seek = time.perf_counter() - t0
                print(f"{V} x {P}, delay {d:5d}, {'FEEDBACK    ' if fb else 'feed-forward'}: step {step * 1e6:8.1f} us (bank {bank_ms / max(nb, 1) * 1e3:6.1f} us, "
                      f"stage {stage_ms / 50 * 1e3:7.1f} us in {ns / 50:.0f} launches per call; fused_stride {plan['fused_stride']}, feedback {plan['feedback']}); "
                      f"call after a seek to 10 s: {seek * 1e3:8.2f} ms", flush=True)


if __name__ == "__main__":
    main()
