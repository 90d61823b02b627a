#!/usr/bin/env python3
"""Parity stress for FEEDBACK through Delay and for control-rate TRACKS (4.8).
feedback: random graphs of the seven primitives in which one or two Delays (constant 1..6 frames) have been re-pointed at
  nodes that depend on them (tests/randgraph.py), contiguous calls of 1..40 frames, a seek forward (the loop's state is rebuilt
  by replay from frame 0), a seek back; modes auto and staged; bit-compared with the CPU oracle's recursion.  The oracle costs
  (paths round the loop) ** (frames / delay) per sample: graphs beyond a budget are passed over.
tracks: voices of 32..512 partials whose leaf is a random expression over the time input, two or three per-partial track rows
  and constants, rendered with fr_set_track_inputs through the dense host call against the oracle (which stores the rows).
usage: python tools/stress_feedback.py [n_seeds [first_seed]]      (FR_STRESS_LIB=sim: the host-logic simulator; tracks need hipRTC)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import randgraph  # noqa: E402
from kat_replay import same_bits  # noqa: E402
import libfriendship_amd  # noqa: E402
from libfriendship_amd import synth  # noqa: E402
from libfriendship_amd.capi import FR_ERR_UNSUPPORTED, RenderError, Renderer, RendererLib  # noqa: E402


def second_loop(steps, rng, n_frames):
    """Re-points one more Delay (if the budget allows): two loops, nested or side by side."""
    made = randgraph.random_feedback_graph.__globals__
    kind, inbound = {}, {}
    for s in steps:
        if s[0] == "node":
            kind[s[1]] = made["PRIMITIVES"][s[2].kind]
        else:
            inbound.setdefault(s[2], {})[s[4]] = s
    cands = []
    for h, k in kind.items():
        amt = inbound.get(h, {}).get(1)
        src = inbound.get(h, {}).get(0)
        if k == "Delay" and amt is not None and amt[1] == 1 and src is not None and src[1] < h and src[1] > 1:
            d = float(np.array([amt[3]], dtype=np.uint32).view(np.float32)[0])
            if 2.0 <= d <= 6.0:
                cands.append(h)
    if not cands:
        return steps
    h = int(cands[rng.integers(len(cands))])
    later = [x for x in kind if x > h and kind[x] != "Delay"]
    if not later:
        return steps
    src = int(later[rng.integers(len(later))])
    out = [s for s in steps if not (s[0] == "edge" and s[2] == h and s[4] == 0)]
    out.append(("edge", src, h, 0, 0))
    return out


def feedback(hip, oracle, seeds):
    bad = done = refused = skipped = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        made = randgraph.random_feedback_graph(seed, n_frames=24, n_nodes=int(rng.integers(6, 18)))
        if made is None:
            skipped += 1
            continue
        steps, n_out, d = made
        two = seed % 3 == 0
        if two:
            steps2 = second_loop(steps, rng, 24)
            # (the budget estimate covers one loop: keep the second only for short runs)
            steps = steps2
        total = 12 if two else 24
        seq, idx = [], 0
        while idx < total:
            n = int(min(total - idx, rng.integers(1, 12)))
            seq.append((idx, n))
            idx += n
        seq += [(total + int(rng.integers(1, 6)), 3), (int(rng.integers(0, 6)), 4)]
        for mode in ("auto", "staged"):
            with Renderer(hip, mode=mode) as r, Renderer(oracle) as ref:
                try:
                    randgraph.install_steps(r, steps)
                    randgraph.install_steps(ref, steps)
                    rr = np.random.default_rng(seed + 1)
                    for idx, n in seq:
                        rows = [rr.normal(size=n).astype(np.float32) * 2, rr.integers(-2, 5, size=n).astype(np.float32)]
                        try:
                            got = r.fill_buffer(n_out, idx, idx + n, rows)
                        except RenderError as e:
                            if e.status in (FR_ERR_UNSUPPORTED, 6):   # (6: the second re-pointing made a loop with no constant Delay >= 1 on it)
                                refused += 1
                                raise StopIteration
                            raise
                        exp = ref.fill_buffer(n_out, idx, idx + n, rows)
                        if not same_bits(got, exp):
                            bad += 1
                            print(f"FEEDBACK MISMATCH seed {seed} mode {mode} call at {idx} (+{n}), delay {d}, two loops {two}", flush=True)
                            raise StopIteration
                    done += 1
                except StopIteration:
                    pass
    print(f"feedback: {done} graph runs bit-exact, {refused} refused (unsupported / no evaluable loop), {skipped} seeds without a candidate, {bad} problems", flush=True)
    return bad


OPS = ["Sum2", "Multiply", "Minimum", "Modulo", "Divide"]


def track_voice(rng, P, first, n_tracks):
    """One voice: P leaves of one random expression shape over time (slot 0), n_tracks track rows per leaf, constants."""
    g = synth.GraphArrays()
    slots = first + n_tracks * np.arange(P, dtype=np.uint32)

    def operand(depth):
        r = rng.random()
        if depth > 2 or r < 0.3:
            c = rng.random()
            if c < 0.35:
                return ("track", int(rng.integers(n_tracks)))
            if c < 0.55:
                return ("time",)
            return ("const", float(np.float32(rng.choice([0.5, 1.0, -1.0, 2.0, 0.25, -0.0, 3.0, float(rng.normal())]))))
        return ("op", OPS[rng.integers(len(OPS))], operand(depth + 1), operand(depth + 1))

    expr = ("op", "Multiply", ("track", 0), ("op", OPS[rng.integers(len(OPS))], operand(1), ("track", n_tracks - 1)))

    def build(e):
        if e[0] != "op":
            return e
        a, b = build(e[2]), build(e[3])
        h = g.nodes(synth.FR_PRIM[e[1]], P)
        for opnd, slot in ((a, 0), (b, 1)):
            if isinstance(opnd, np.ndarray):
                g.edge(opnd, h, 0, slot)
            elif opnd[0] == "track":
                g.edge(0, h, slots + opnd[1], slot)
            elif opnd[0] == "time":
                g.edge(0, h, 0, slot)
            else:
                g.const(h, np.float32(opnd[1]), slot)
        return h

    leaves = build(expr)
    root = synth.sum_tree(g, leaves.reshape(1, P))
    g.edge(root, 0, 0, 0)
    return g.finish(1), first + n_tracks * P


def tracks(hip, oracle, seeds):
    bad = done = unmatched = 0
    for seed in seeds:
        rng = np.random.default_rng(90_000 + seed)
        P = int(rng.choice([32, 64, 128, 512]))
        n_tracks = int(rng.integers(2, 4))
        first = int(rng.integers(1, 4))
        tree, R = track_voice(rng, P, first, n_tracks)
        with Renderer(hip) as r, Renderer(oracle) as ref:
            r.set_track_inputs(first)
            synth.install(r, tree)
            synth.install(ref, tree)
            idx, ok = 0, True
            for n in (int(R + 5), 37, 64, 1):   # (the first call is long enough for every row to exist: reference.rs:59-64)
                m = (rng.normal(size=(R, n)) * 0.7).astype(np.float32)
                m[0] = np.arange(idx, idx + n, dtype=np.float32)
                if seed % 4 == 0:
                    m[rng.integers(1, R), rng.integers(n)] = np.float32(rng.choice([np.inf, -np.inf, np.nan, -0.0, 1e30]))
                try:
                    got = r.fill_buffer_dense(1, idx, idx + n, m)
                except RenderError as e:
                    if e.status == FR_ERR_UNSUPPORTED:   # the shape matcher did not take this leaf (too many parameters, a bare input ...)
                        unmatched += 1
                        ok = False
                        break
                    raise
                exp = ref.fill_buffer_dense(1, idx, idx + n, m)
                if not same_bits(got, exp):
                    bad += 1
                    ok = False
                    print(f"TRACKS MISMATCH seed {seed} (P {P}, {n_tracks} tracks per leaf, first {first}) call at {idx} (+{n})", flush=True)
                    break
                idx += n
            done += ok
    print(f"tracks: {done} voices bit-exact, {unmatched} shapes the matcher did not take, {bad} problems", flush=True)
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sim = os.environ.get("FR_STRESS_LIB") == "sim"
    if sim:
        import sim_tools
        hip = sim_tools.sim_lib()
    else:
        hip = libfriendship_amd.hip_lib()
    oracle = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    bad = feedback(hip, oracle, range(first, first + n))
    if not sim:
        bad += tracks(hip, oracle, range(first, first + max(1, n // 3)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
