#!/usr/bin/env python3
"""Writes tests/golden/selfcheck_vectors.json: SELF-CONSISTENCY vectors ("extra golden data").

Unlike reference_kat.json these are NOT outputs of the reference: they are what this repository's CPU oracle
(oracle/ref_renderer.cpp) rendered when the file was made, for seeded random graphs of all seven primitives with
nested composites, constant and signal-driven delays, short input rows, a seek and graph edits between calls, plus
the first and last 64 frames of configs A (one 440 Hz partial) and B (256 partials x 1 voice).  They pin the oracle
against silent changes and give the HIP engine committed data to reproduce.  Same fixture format as the KAT file.

    python tests/golden/make_selfcheck_vectors.py        # needs oracle/_build/libfr_oracle.so
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import randgraph  # noqa: E402
from libfriendship_amd import synth  # noqa: E402
from libfriendship_amd.capi import Renderer, RendererLib  # noqa: E402


def step_json(s):
    if s[0] == "node":
        return {"op": "add_node", "handle": int(s[1]), "effect": s[2].to_json()}
    op = "del_edge" if s[0] == "deledge" else "add_edge"
    return {"op": op, "from": int(s[1]), "to": int(s[2]), "from_slot": int(s[3]), "to_slot": int(s[4])}


def render_step(r, n_slots, start, end, rows, cols=None):
    """rows[0] is always the time ramp of the range: regenerated at replay, not stored."""
    out = r.fill_buffer(n_slots, start, end, rows)
    st = {"op": "render", "range": [int(start), int(end)], "n_slots": int(n_slots), "time_ramp_row0": True,
          "inputs_bits": [[int(x) for x in np.asarray(row, dtype=np.float32).view(np.uint32)] for row in rows[1:]]}
    if cols:
        st["expect_cols"] = cols
        out = np.concatenate([out[:, a:b] for a, b in cols], axis=1)
    st["expect_bits"] = [[int(x) for x in row] for row in np.ascontiguousarray(out).view(np.uint32)]
    return st


def random_case(lib, seed, with_edits):
    rng = np.random.default_rng(424200 + seed)
    steps, n_out = randgraph.random_graph(7000 + seed, n_nodes=int(rng.integers(8, 40)), n_inputs=2, n_outputs=3,
                                          signal_delays=seed % 3 != 0, composites=True)
    T = 40
    out = [step_json(s) for s in steps]
    with Renderer(lib) as r:
        randgraph.install_steps(r, steps)
        calls = [(0, T, T), (T, 2 * T, 13), (2 * T, 3 * T, T), (1000 + 17 * seed, 1000 + 17 * seed + T, T)]
        for k, (a, b, len1) in enumerate(calls):
            rows = [synth.time_ramp(a, b), (rng.normal(size=len1) * 3).astype(np.float32)]
            out.append(render_step(r, n_out, a, b, rows))
            if with_edits and k in (0, 1):
                edits = randgraph.random_edits(rng, steps, int(rng.integers(1, 4)), signal_delays=seed % 3 != 0)
                randgraph.install_steps(r, edits)
                out.extend(step_json(s) for s in edits)
    return {"name": f"random_graph_{seed}" + ("_edited" if with_edits else ""), "ref": "self-consistency (oracle)", "steps": out}


def tree_case(lib, name, voices, partials, frames, cols):
    with Renderer(lib) as r:
        synth.install(r, synth.additive_tree(voices, partials))
        steps = [{"op": "synth_tree", "kind": "additive", "voices": voices, "partials": partials},
                 render_step(r, voices, 0, frames, [synth.time_ramp(0, frames)], cols)]
    return {"name": name, "ref": "self-consistency (oracle)", "steps": steps}


def main():
    lib = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    tests = [random_case(lib, s, with_edits=False) for s in range(8)]
    tests += [random_case(lib, 100 + s, with_edits=True) for s in range(4)]
    tests.append(tree_case(lib, "one_partial_one_second", 1, 1, 48000, [[0, 64], [47936, 48000]]))
    tests.append(tree_case(lib, "config_B_256_partials_x_1_voice", 1, 256, 4800, [[0, 64], [4736, 4800]]))
    path = os.path.join(ROOT, "tests", "golden", "selfcheck_vectors.json")
    with open(path, "w") as f:
        json.dump({"kind": "self-consistency vectors rendered by oracle/ref_renderer.cpp; NOT reference outputs",
                   "tests": tests}, f, separators=(",", ":"))
    print(f"{len(tests)} tests, {os.path.getsize(path)} bytes -> {path}")


if __name__ == "__main__":
    main()
