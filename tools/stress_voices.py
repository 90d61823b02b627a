#!/usr/bin/env python3
"""Parity stress for the voice (bank) path: seeded random LEAF SHAPES -- a random expression over the time input, a
second input, per-partial constants and shared literals (zeros of both signs, infinities, NaN among them) -- instantiated
for every partial of a few voices and summed per voice by a balanced or ragged Sum2 tree, optionally behind a constant
delay or sharing a root.  The engine recognises such voices, generates a leaf function and compiles it with hipRTC
(csrc/match.cpp, leafjit.cpp, jit.cpp); everything is compared bit for bit with the CPU oracle over contiguous calls and
a seek.  tools/stress_parity.py covers arbitrary graphs; this one covers what only voices exercise.
usage: python tools/stress_voices.py [n_seeds [first_seed]]     (FR_STRESS_LIB=sim: the host-logic simulator, CPU)"""
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

OPS = [synth.K_SUM2, synth.K_MUL, synth.K_DIV, synth.K_MOD, synth.K_MIN]
LITERALS = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, 4.0, -16.0, 3.0, 1e-30, float("inf"), float("-inf"), float("nan"), 1e20]


def random_leaf(rng, g, n, depth):
    """One random expression, instantiated n times (one node array per operator).  Returns handles [n] or a C()/IN()."""
    r = rng.random()
    if depth == 0 or r < 0.25:
        k = rng.random()
        if k < 0.35:
            return synth.IN(0)
        if k < 0.45:
            return synth.IN(1)
        if k < 0.75:   # a per-partial parameter
            scale = float(rng.choice([1e-3, 0.05, 1.0, 7.0]))
            return synth.C((rng.normal(size=n) * scale).astype(np.float32))
        return synth.C(np.float32(LITERALS[rng.integers(len(LITERALS))]))
    a = random_leaf(rng, g, n, depth - 1)
    b = random_leaf(rng, g, n, depth - 1)
    if isinstance(a, tuple) and isinstance(b, tuple) and a[0] == "c" and b[0] == "c" and rng.random() < 0.7:
        b = synth.IN(0)   # (mostly avoid all-constant sub-expressions: they fold away)
    return g.binop(int(OPS[rng.integers(len(OPS))]), a, b, n)


def build(seed):
    rng = np.random.default_rng(seed)
    V = int(rng.integers(1, 4))
    P = int(rng.choice([8, 16, 32, 64, 128, 256, 24, 40, 100]))
    g = synth.GraphArrays()
    n = V * P
    leaf = random_leaf(rng, g, n, int(rng.integers(2, 5)))
    if isinstance(leaf, tuple):   # degenerate draw: make it an expression
        leaf = g.binop(synth.K_MUL, leaf, synth.IN(0), n)
    if rng.random() < 0.7:        # an amplitude per partial on top, as every additive voice has
        leaf = g.binop(synth.K_MUL, synth.C((1.0 / (1 + np.arange(n) % P)).astype(np.float32)), leaf, n)
    roots = synth.sum_tree(g, leaf.reshape(V, P))
    n_out = V
    g.edge(roots, 0, 0, np.arange(V, dtype=np.uint32))
    extra = rng.random()
    if extra < 0.25:      # a delayed copy of voice 0 mixed back in: the bank fills a ring
        d = g.binop(synth.K_SUM2, roots[0:1], g.binop(synth.K_DELAY, roots[0:1], synth.C(np.float32(rng.integers(1, 90))), 1), 1)
        g.edge(d, 0, 0, n_out)
        n_out += 1
    elif extra < 0.4:     # the same root on a second row
        g.edge(roots[0:1], 0, 0, n_out)
        n_out += 1
    return g.finish(n_out), V, P


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = [int(x) for x in os.environ["FR_STRESS_SEEDS"].split(",")] if os.environ.get("FR_STRESS_SEEDS") else None
    if os.environ.get("FR_STRESS_LIB") == "sim":
        import sim_tools
        hip = sim_tools.sim_lib()
    else:
        hip = libfriendship_amd.hip_lib()
    oracle = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    bad = jit_voices = template = general = pulled = 0
    for i, seed in enumerate(only or range(first, first + n)):
        tree, V, P = build(seed)
        rng = np.random.default_rng(seed + 10**6)
        T = int(rng.integers(1, 300))
        start = int(rng.choice([0, 0, 1000, 2**24 - 100]))   # (beyond 2^24 the f32 ramp stops being exact integers)
        calls = [(start, start + T), (start + T, start + 2 * T), (start + 2 * T, start + 3 * T), (start + 50 * T + 7, None)]
        for semantics in ("reference", "sparkle") if seed % 4 == 0 else ("reference",):
            with Renderer(oracle, semantics=semantics) as ref, Renderer(hip, semantics=semantics) as eng:
                synth.install(ref, tree)
                synth.install(eng, tree)
                for k, (s, e) in enumerate(calls):
                    e = e if e is not None else s + T
                    rows = [synth.time_ramp(s, e), (rng.normal(size=e - s) * 2).astype(np.float32)]
                    exp = ref.fill_buffer(tree["n_outputs"], s, e, rows)
                    got = eng.fill_buffer(tree["n_outputs"], s, e, rows)
                    if not same_bits(got, exp):
                        bad += 1
                        w = np.argwhere((got.view(np.uint32) != exp.view(np.uint32)) & ~(np.isnan(got) & np.isnan(exp)))
                        print(f"seed {seed} ({V} x {P}, {semantics}) call {k}: MISMATCH at {len(w)} samples, first row {w[0][0]} frame {s + w[0][1]}: "
                              f"got {got[tuple(w[0])]!r} expected {exp[tuple(w[0])]!r}")
                        if only:
                            print("   ", eng.plan())
                        break
                plan = eng.plan()
                jit_voices += sum(b["voices"] for b in plan["banks"] if b["jit"])
                template += sum(b["voices"] for b in plan["banks"] if not b["jit"] and not b["general_tree"])
                general += sum(b["voices"] for b in plan["banks"] if b["general_tree"])
                pulled += plan["pull_rows"] + plan["stage_programs"]
        if i % 50 == 49:
            print(f"{i + 1} shapes, {bad} problems (voices so far: {jit_voices} compiled leaves, {general} of them general trees, "
                  f"{template} template; {pulled} rows/programs outside banks)", flush=True)
    print(f"done: {n if not only else len(only)} leaf shapes, {bad} problems; voices: {jit_voices} compiled, {general} general-tree, {template} template")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
