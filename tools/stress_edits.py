#!/usr/bin/env python3
"""Parity stress for GRAPH EDITS DURING PLAYBACK on patches with voices (banks), envelopes and delay taps: what
incremental re-lowering, the voice matcher's cache, ring reassignment and parameter re-upload have to get right.
Starts from tests/test_shard_sim.py's random patches (voices of assorted sizes and tree shapes behind gains, a shared
envelope, delay taps; rows wired at random) and, between calls, applies batches of random edits through the watcher calls
(graphwatcher.rs:4-9): a constant changed (amplitude, phase increment, gain, delay length -- ordinary or hostile values),
an inner edge cut (the input reads 0, reference.rs:164-173) and later restored, an output row repointed or disconnected, a
node deleted after its edges, a whole new voice added (note-on) and wired to a row.  After every batch all engine modes
render the next block (random length, sometimes after a seek) and are compared bit for bit with the CPU oracle that
received the same edits.
usage: python tools/stress_edits.py [n_seeds [first_seed]]      (FR_STRESS_LIB=sim: the host-logic simulator, CPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from kat_replay import same_bits  # noqa: E402
import libfriendship_amd  # noqa: E402
from libfriendship_amd import synth  # noqa: E402
from libfriendship_amd.capi import Effect, PRIMITIVES, RenderError, Renderer, RendererLib, f32_bits  # noqa: E402
import test_shard_sim  # noqa: E402

SPECIAL = [0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 1e-30, 1e20, float("inf"), float("nan"), 3.0, 37.0, 1.8446744e19]


class Model:
    """The graph as the tool knows it: what is connected to every (node, slot), so that edits stay well-formed."""

    def __init__(self, tree):
        self.kind = {int(h): int(k) for h, k in zip(tree["handles"], tree["kinds"])}
        self.inbound = {}
        for e in tree["edges"]:
            f, t, fs, ts = (int(x) for x in e)
            self.inbound[(t, ts)] = (f, t, fs, ts)
        self.next = max(self.kind) + 1
        self.cut_edges = []

    def consumers_of(self, h):
        return [e for e in self.inbound.values() if e[0] == h]


def random_edits(rng, m, n_rows, renderers):
    """Applies 1-4 random edits to the model and to every renderer."""
    def add(e):
        m.inbound[(e[1], e[3])] = e
        for r in renderers:
            r.on_add_edge(*e)

    def delete(e):
        m.inbound.pop((e[1], e[3]), None)
        for r in renderers:
            r.on_del_edge(*e)

    log = []
    for _ in range(int(rng.integers(1, 5))):
        kind = rng.random()
        edges = list(m.inbound.values())
        if kind < 0.35:                                   # a constant changes
            consts = [e for e in edges if e[0] == synth.CONST_HANDLE]
            if not consts:
                continue
            e = consts[rng.integers(len(consts))]
            old = np.uint32(e[2]).view(np.float32)
            r = rng.random()
            new = (np.float32(SPECIAL[rng.integers(len(SPECIAL))]) if r < 0.15 else
                   np.float32(old * np.float32(rng.choice([0.5, 2.0, -1.0, 1.0009765625]))) if r < 0.6 else np.float32(rng.normal() * 0.3))
            delete(e)
            add((e[0], e[1], f32_bits(new), e[3]))
            log.append(f"const {e[1]}.{e[3]} {old!r} -> {new!r}")
        elif kind < 0.5:                                  # an inner edge is cut
            inner = [e for e in edges if e[0] > synth.CONST_HANDLE and e[1] != 0]
            if not inner:
                continue
            e = inner[rng.integers(len(inner))]
            delete(e)
            m.cut_edges.append(e)
            log.append(f"cut {e}")
        elif kind < 0.62 and m.cut_edges:                 # ... and restored (if both ends still exist and the slot is free)
            e = m.cut_edges.pop(rng.integers(len(m.cut_edges)))
            if e[0] in m.kind and e[1] in m.kind and (e[1], e[3]) not in m.inbound:
                add(e)
                log.append(f"restore {e}")
        elif kind < 0.75:                                 # an output row repointed or disconnected
            row = int(rng.integers(n_rows))
            cur = m.inbound.get((0, row))
            if cur is not None:
                delete(cur)
            if rng.random() < 0.8:
                srcs = [h for h, k in m.kind.items() if h > synth.CONST_HANDLE and PRIMITIVES[k] == "Sum2"]
                if srcs:
                    add((int(srcs[rng.integers(len(srcs))]), 0, 0, row))
            log.append(f"row {row} repointed")
        elif kind < 0.85:                                 # a node is deleted, after its edges
            cand = [h for h, k in m.kind.items() if h > synth.CONST_HANDLE]
            if len(cand) < 8:
                continue
            h = int(cand[rng.integers(len(cand))])
            for e in [e for e in m.inbound.values() if e[0] == h or e[1] == h]:
                delete(e)
            del m.kind[h]
            for r in renderers:
                r.on_del_node(h)
            log.append(f"node {h} deleted")
        else:                                             # note-on: a new voice on some row
            P = int(rng.choice([8, 16, 32, 48, 64]))
            g = synth.GraphArrays()
            g.next = m.next
            p = synth.voice_params(1, P, int(rng.integers(1, 1 << 30)), bool(rng.integers(2)))
            root = synth.sum_tree(g, synth.partial_leaves(g, p["w"], p["amp"]).reshape(1, P))
            row = int(rng.integers(n_rows))
            cur = m.inbound.get((0, row))
            if cur is not None:
                delete(cur)
            g.edge(root, 0, 0, row)
            extra = g.finish(n_rows)
            extra["handles"], extra["kinds"] = extra["handles"][1:], extra["kinds"][1:]   # (the constant node exists already)
            for r in renderers:
                synth.install(r, extra)
            for h, k in zip(extra["handles"], extra["kinds"]):
                m.kind[int(h)] = int(k)
            for e in extra["edges"]:
                f, t, fs, ts = (int(x) for x in e)
                m.inbound[(t, ts)] = (f, t, fs, ts)
            m.next = g.next
            log.append(f"note-on: {P} partials on row {row}")
    return log


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = [int(x) for x in os.environ["FR_STRESS_SEEDS"].split(",")] if os.environ.get("FR_STRESS_SEEDS") else None
    if os.environ.get("FR_STRESS_LIB") == "sim":
        import sim_tools
        hip = sim_tools.sim_lib()
    else:
        hip = libfriendship_amd.hip_lib()
    oracle = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    bad = batches = 0
    for i, seed in enumerate(only or range(first, first + n)):
        rng = np.random.default_rng(50_000 + seed)
        n_rows = int(rng.integers(3, 7))
        tree = test_shard_sim._random_patch(rng, n_rows)
        m = Model(tree)
        engines = {"auto": Renderer(hip, mode="auto"), "staged": Renderer(hip, mode="staged"), "pull": Renderer(hip, mode="pull")}
        with Renderer(oracle) as ref:
            everyone = [ref] + list(engines.values())
            for r in everyone:
                synth.install(r, tree)
            head = 0
            history = []
            ok = True
            for k in range(14):
                T = int(rng.choice([1, 16, 64, 100, 257]))
                if rng.random() < 0.12:
                    head += int(rng.integers(1, 5000))
                rows = [synth.time_ramp(head, head + T), (rng.normal(size=T) * 2).astype(np.float32)]
                try:
                    exp = ref.fill_buffer(n_rows, head, head + T, rows)
                except RenderError as e:   # (an edit made the graph unrenderable for the oracle: not this tool's subject)
                    print(f"seed {seed}: oracle refused after {history[-1:]}: {e}", flush=True)
                    break
                for name, eng in engines.items():
                    try:
                        got = eng.fill_buffer(n_rows, head, head + T, rows)
                        what = None if same_bits(got, exp) else "MISMATCH"
                    except RenderError as e:
                        what = f"refused ({e})"
                    if what:
                        bad += 1
                        ok = False
                        wh = "" if what != "MISMATCH" else f" rows {sorted(set(np.argwhere(got.view(np.uint32) != exp.view(np.uint32))[:, 0].tolist()))}"
                        print(f"seed {seed} engine {name} call {k} (idx {head}, T {T}): {what}{wh}; last edits: {history[-1:]}", flush=True)
                if not ok:
                    break
                head += T
                history.append(random_edits(rng, m, n_rows, everyone))
                batches += 1
        for e in engines.values():
            e.close()
        if i % 20 == 19:
            print(f"{i + 1} patches, {batches} edit batches, {bad} problems", flush=True)
    print(f"done: {n if not only else len(only)} patches, {batches} edit batches, {bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
