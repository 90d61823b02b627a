"""Edit-during-playback latency: wall time of the fill_buffer call that follows a graph edit vs a steady-state call.

    python tools/edit_latency.py [--voices 64 --partials 4096 --frames 4800]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from libfriendship_amd import hip_lib, synth  # noqa: E402
from libfriendship_amd.capi import Renderer, RendererLib, f32_bits  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, default=64)
    ap.add_argument("--partials", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=4800)
    ap.add_argument("--tree", default="additive", choices=["additive", "effects"])
    ap.add_argument("--async-compile", action="store_true", help="the ABI's default: hipRTC on a worker thread (Python's default is sync)")
    ap.add_argument("--note-ons", type=int, default=3)
    ap.add_argument("--lib", default=None, help="another build of the C ABI (the host-logic simulator of tests/sim_tools.py, with --frames 64: host costs only)")
    a = ap.parse_args()
    V, P, T = a.voices, a.partials, a.frames
    tree = synth.additive_tree(V, P) if a.tree == "additive" else synth.effects_tree(V, P)
    e = tree["edges"]
    amp = tree["params"]["amp"]
    rows = np.nonzero((e[:, 0] == synth.CONST_HANDLE) & (e[:, 3] == 0) & (e[:, 2] == f32_bits(amp[0, 100 % P])))[0]
    with Renderer(RendererLib(a.lib) if a.lib else hip_lib(), sync_compile=not a.async_compile) as r:
        t0 = time.perf_counter()
        synth.install(r, tree)
        t_install = time.perf_counter() - t0
        idx = 0

        def call():
            nonlocal idx
            t = synth.time_ramp(idx, idx + T)
            t0 = time.perf_counter()
            r.fill_buffer(V, idx, idx + T, [t])
            idx += T
            return (time.perf_counter() - t0) * 1e3

        first = call()
        plan = r.plan()
        steady = [call() for _ in range(20)]
        after = []
        for i in range(8):
            row = rows[i % len(rows)]
            to, slot, old = int(e[row, 1]), int(e[row, 3]), int(e[row, 2])
            new = f32_bits(np.float32(0.2 + 0.01 * i))
            t0 = time.perf_counter()
            r.on_del_edge(synth.CONST_HANDLE, to, old, slot)
            r.on_add_edge(synth.CONST_HANDLE, to, new, slot)
            t_edit = (time.perf_counter() - t0) * 1e3
            e[row, 2] = new
            after.append((t_edit, call(), r.plan()))
        # note-on: a whole new voice (P partials, ~12 P nodes and ~22 P edges) arrives between two calls
        note_on = []
        for i in range(a.note_ons):
            g = synth.GraphArrays()
            g.next = int(tree["handles"].max()) + 1 + i * 16 * P
            vp = synth.voice_params(1, P, 0x5EED0100 + i)
            root = synth.sum_tree(g, synth.partial_leaves(g, vp["w"], vp["amp"]).reshape(1, P))
            g.edge(root, 0, 0, V + i)
            extra = g.finish(V + i + 1)
            extra["handles"], extra["kinds"] = extra["handles"][1:], extra["kinds"][1:]   # the constant node exists already
            t0 = time.perf_counter()
            synth.install(r, extra)
            t_msgs = (time.perf_counter() - t0) * 1e3
            tt = synth.time_ramp(idx, idx + T)
            t0 = time.perf_counter()
            r.fill_buffer(V + i + 1, idx, idx + T, [tt])
            idx += T
            note_on.append((t_msgs, (time.perf_counter() - t0) * 1e3, r.plan()))
        print(f"{a.tree} tree {V} x {P}, {T} frames per call (host-buffer API)")
        print(f"  install (mirror build)        {t_install * 1e3:9.1f} ms")
        print(f"  first call (lower+plan+run)   {first:9.1f} ms   lower {plan['lower_ms']:.1f} ms, build {plan['build_ms']:.1f} ms")
        print(f"  steady call                   {np.median(steady):9.3f} ms")
        for t_edit, ms, p in after:
            print(f"  call after an edit            {ms:9.3f} ms   edit msgs {t_edit:.3f} ms, {p['lowering']}, "
                  f"{p['relowered_nodes']} nodes re-lowered, build {p['build_ms']:.2f} ms")
        for t_msgs, ms, p in note_on:
            print(f"  call after a note-on          {ms:9.3f} ms   a {P}-partial voice added: batch messages {t_msgs:.1f} ms, {p['lowering']}, "
                  f"{p['relowered_nodes']} nodes lowered, build {p['build_ms']:.2f} ms (lowering {p['lower_ms']:.1f}, hipRTC so far {p['jit_compile_ms']:.0f} ms / "
                  f"{p['jit_kernels_compiled']} kernels, {len(p['banks'])} bank launches)")


if __name__ == "__main__":
    main()
