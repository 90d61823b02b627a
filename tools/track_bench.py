#!/usr/bin/env python3
"""The HBM-bound variant of the hot path (last sentence): every partial's phase increment and amplitude are
control-rate TRACKS -- input rows, 8 bytes per partial-frame -- read in place from the call's dense device matrix
(fr_set_track_inputs + fr_fill_buffer_device_dense).  Reports, per call length, the bank kernel's time (HIP events inside the
library, fr_set_timing) and the achieved fraction of the HBM roofline on ALGORITHMIC bytes: 8 B x V x P x T of tracks + 4 B x T
of time + 4 B x V x T of output.

    python tools/track_bench.py [--voices 64 --partials 4096 --frames 64,512,1024,4800 --steps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from libfriendship_amd import hip_lib, synth  # noqa: E402
from libfriendship_amd.capi import Renderer  # noqa: E402

HBM_PEAK = 8.0e12       # MI355X_MICROARCH.md: HBM3E spec
HBM_ACHIEVABLE = 6.29e12   # ... and what a streaming kernel reaches


def run(V=64, P=4096, frames=(64, 1024), steps=20, log=print):
    """One renderer, one priming call, then per call length `steps` timed device-resident calls.  Returns the records."""
    import torch
    tree = synth.track_tree(V, P)
    R = tree["n_inputs"]
    out = []
    with Renderer(hip_lib()) as r:
        r.set_track_inputs(tree["first_track"])
        t0 = time.perf_counter()
        synth.install(r, tree)
        # the reference has n_slots * n_times input vectors of the largest call so far: one long call makes all R exist
        prime = -(-R // V) + 1
        d_out = torch.empty((V, prime), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        d_t = torch.arange(0, prime, dtype=torch.float32, device="cuda").reshape(1, prime)   # (the time row must be continuous)
        r.fill_buffer_device_dense(d_out.data_ptr(), V, prime, 0, d_t.data_ptr(), 1, s)
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t0
        idx = prime
        plan = r.plan()
        bank = plan["banks"][0]
        assert bank["tracks"] and bank["jit"], plan
        log(f"track tree {V} x {P}: {R} input rows; install + first call {t_first:.2f} s; leaf ops {bank['leaf_ops']}, params per leaf {bank['leaf_params']}")
        for T in frames:
            # several matrices in rotation, > 1 GB together: a 134 MB matrix read again every step would come from the 256 MB
            # Infinity Cache, not from HBM
            n_mat = int(min(8, max(1, -(-(1 << 30) // (R * T * 4)))))
            mats = []
            for k in range(n_mat):
                gen = torch.Generator(device="cuda").manual_seed(T * 8 + k)
                d_m = torch.empty((R, T), dtype=torch.float32, device="cuda")
                # w in [0, 0.05), amp in [0, 0.05): any values do for the timing; parity is tests/test_hip_parity.py::test_track_voices_against_oracle
                d_m.uniform_(0.0, 0.05, generator=gen)
                mats.append(d_m)
            d_o = torch.empty((V, T), dtype=torch.float32, device="cuda")

            def call(k):
                nonlocal idx
                m = mats[k % n_mat]
                m[0] = torch.arange(idx, idx + T, dtype=torch.float32, device="cuda")
                r.fill_buffer_device_dense(d_o.data_ptr(), V, T, idx, m.data_ptr(), R, s)
                idx += T

            for k in range(3):
                call(k)
            torch.cuda.synchronize()
            r.set_timing(True)
            r.reset_timing()
            t0 = time.perf_counter()
            for k in range(steps):
                call(k)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps
            ms, n = r.get_timing("bank")
            r.set_timing(False)
            kern = ms / max(n, 1) * 1e-3
            nbytes = 8.0 * V * P * T + 4.0 * T + 4.0 * V * T
            rec = {"frames": T, "kernel_us": round(kern * 1e6, 2), "call_us": round(wall * 1e6, 2), "algorithmic_bytes": nbytes,
                   "achieved_GBps": round(nbytes / kern / 1e9, 1), "frac_of_8TBps": round(nbytes / kern / HBM_PEAK, 4),
                   "frac_of_6.29TBps": round(nbytes / kern / HBM_ACHIEVABLE, 4), "msamples_per_s": round(V * T / wall / 1e6, 2),
                   "matrices_in_rotation": n_mat, "nonzero": bool(d_o.abs().max().item() > 0)}
            out.append(rec)
            log(f"  T = {T:5d}: kernel {rec['kernel_us']:9.1f} us ({n} launches timed, {n_mat} matrices in rotation), call {rec['call_us']:9.1f} us host-to-host; "
                f"{nbytes / 1e6:9.1f} MB -> {rec['achieved_GBps']:7.0f} GB/s = {rec['frac_of_6.29TBps']:.3f} of 6.29 TB/s ({rec['frac_of_8TBps']:.3f} of 8.0); "
                f"{rec['msamples_per_s']:.1f} Msamples/s")
            del mats, d_o
            torch.cuda.empty_cache()
    return {"voices": V, "partials": P, "rows": R, "runs": out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, default=64)
    ap.add_argument("--partials", type=int, default=4096)
    ap.add_argument("--frames", default="64,128,512,1024,4800")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    res = run(a.voices, a.partials, [int(x) for x in a.frames.split(",")], a.steps)
    if a.json:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
