#!/usr/bin/env python3
"""Parity stress for the hand-written oscillator bank (the headline kernel and its relatives): the N1 partial with
HOSTILE parameters and inputs.  Random voice counts and partial counts (powers of two -> the balanced template kernels
incl. the short-call and many-small-voices forms; ragged -> the general-tree kernels), phase increments and amplitudes
drawn from ordinary values mixed with zeros of both signs, negatives, huge values, denormals, infinities and NaN,
amplitudes whose -16 * amp is inexact; the time input is a ramp from a random start or, on some calls, arbitrary
values (negative, non-integral, huge, NaN).  Calls of random length incl. 1 frame, a seek, a voice silent by
construction.  Everything is compared bit for bit with the CPU oracle -- this is where the 5-op FMA leaf, the v_fract
shortcut and the zero-sign settlement have to hold.
usage: python tools/stress_bank.py [n_seeds [first_seed]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from kat_replay import same_bits  # noqa: E402
import libfriendship_amd  # noqa: E402
from libfriendship_amd import synth  # noqa: E402
from libfriendship_amd.capi import Renderer, RendererLib  # noqa: E402

W_SPECIAL = [0.0, -0.0, 1.0, 0.5, 0.25, -0.001, -1.0, 1e-30, 1e-42, 1e30, 3e38, float("inf"), float("-inf"), float("nan"), 1.0 / 48000, 2.0 ** -25]
A_SPECIAL = [0.0, -0.0, 1.0, -1.0, 1e-40, 1e-38, 2e37, 3e38, float("inf"), float("-inf"), float("nan"), 0.1, 1.0 / 3]
T_SPECIAL = [0.0, -0.0, -1.0, 0.5, -2.75, 1e-42, 16777216.0, 4294967296.0, 4294967808.0, 1e30, float("inf"), float("-inf"), float("nan")]


def params(rng, n, special, hostile_share, ordinary):
    v = ordinary(n).astype(np.float32)
    m = rng.random(n) < hostile_share
    v[m] = np.array(special, np.float32)[rng.integers(len(special), size=int(m.sum()))]
    return v


def build(seed):
    rng = np.random.default_rng(seed)
    kind = rng.random()
    if kind < 0.45:
        V, P = int(rng.integers(1, 9)), int(2 ** rng.integers(3, 13))          # balanced template; up to 8 x 4096
    elif kind < 0.6:
        V, P = int(rng.integers(8, 400)), int(2 ** rng.integers(3, 8))          # many small voices
    else:
        V, P = int(rng.integers(1, 6)), int(rng.integers(9, 700))               # ragged: general trees
    n = V * P
    hostile = float(rng.choice([0.0, 0.0, 0.02, 0.3]))
    w = params(rng, n, W_SPECIAL, hostile, lambda k: np.abs(rng.normal(size=k)) * float(rng.choice([1e-4, 1e-2, 0.3])))
    amp = params(rng, n, A_SPECIAL, hostile, lambda k: 1.0 / (1 + np.arange(k) % P))
    if rng.random() < 0.3:      # one voice silent by construction (every leaf a zero: the sign of the sum is the test)
        w.reshape(V, P)[0] = np.float32(rng.choice([0.0, 1.0, -0.0]))
        if rng.random() < 0.5:
            amp.reshape(V, P)[0] = np.float32(rng.choice([0.0, -0.0]))
    g = synth.GraphArrays()
    leaves = synth.partial_leaves(g, w, amp)
    roots = synth.sum_tree(g, leaves.reshape(V, P))
    g.edge(roots, 0, 0, np.arange(V, dtype=np.uint32))
    n_out = V
    if rng.random() < 0.2:      # a delayed copy mixed in: the bank fills a ring, a program reads it
        d = g.binop(synth.K_SUM2, roots[0:1], g.binop(synth.K_DELAY, roots[0:1], synth.C(np.float32(rng.integers(1, 200))), 1), 1)
        g.edge(d, 0, 0, n_out)
        n_out += 1
    return g.finish(n_out), V, P, hostile


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = [int(x) for x in os.environ["FR_STRESS_SEEDS"].split(",")] if os.environ.get("FR_STRESS_SEEDS") else None
    if os.environ.get("FR_STRESS_LIB") == "sim":   # the host-logic simulator: checks the tool and the planner, not the kernels
        import sim_tools
        hip = sim_tools.sim_lib()
    else:
        hip = libfriendship_amd.hip_lib()
    oracle = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    bad = calls_done = 0
    kinds = {}
    for i, seed in enumerate(only or range(first, first + n)):
        tree, V, P, hostile = build(seed)
        rng = np.random.default_rng(seed + 10**6)
        # the oracle walks ~11 nodes per partial per frame: keep V * P * frames around 1e6 per call
        budget = max(1, int(1e6 // (V * P)))
        with Renderer(oracle) as ref, Renderer(hip) as eng:
            synth.install(ref, tree)
            synth.install(eng, tree)
            idx = int(rng.choice([0, 0, 1000, 2**24 - 40, 2**31]))
            for k in range(5):
                T = int(min(budget, rng.choice([1, 2, 17, 64, 65, 130, 300, 700])))
                if k == 3:
                    idx += int(rng.integers(1, 10**6))      # a seek
                ramp = synth.time_ramp(idx, idx + T)
                if rng.random() < 0.3:                      # arbitrary input values instead of the ramp
                    ramp = params(rng, T, T_SPECIAL, 0.3, lambda m: rng.normal(size=m) * float(rng.choice([1.0, 1e3, 1e7])))
                exp = ref.fill_buffer(tree["n_outputs"], idx, idx + T, [ramp])
                got = eng.fill_buffer(tree["n_outputs"], idx, idx + T, [ramp])
                calls_done += 1
                if not same_bits(got, exp):
                    bad += 1
                    wh = np.argwhere((got.view(np.uint32) != exp.view(np.uint32)) & ~(np.isnan(got) & np.isnan(exp)))
                    print(f"seed {seed} ({V} x {P}, hostile share {hostile}) call {k} (T={T}, idx={idx}): MISMATCH at {len(wh)} samples, first row "
                          f"{wh[0][0]} frame +{wh[0][1]}: got {got[tuple(wh[0])]!r} expected {exp[tuple(wh[0])]!r}")
                    if only:
                        print("   ", eng.plan())
                    break
                idx += T
            for b in eng.plan()["banks"]:
                key = "general-tree" if b["general_tree"] else ("compiled" if b["jit"] else "template")
                kinds[key] = kinds.get(key, 0) + b["voices"]
        if i % 25 == 24:
            print(f"{i + 1} trees, {calls_done} calls, {bad} problems; voices by kernel family so far: {kinds}", flush=True)
    print(f"done: {n if not only else len(only)} trees, {calls_done} calls, {bad} problems; voices by kernel family: {kinds}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
