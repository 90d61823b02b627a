"""Replays a fixture test (tests/golden/*.json) against a renderer through the C ABI."""
import numpy as np

from libfriendship_amd.capi import Renderer


def bits_to_f32(rows):
    return [np.array(r, dtype=np.uint32).view(np.float32) for r in rows]


def same_bits(a, b):
    """Bit-exact comparison; NaNs compare equal to NaNs (payloads are not part of the contract)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    au, bu = a.view(np.uint32), b.view(np.uint32)
    return bool(np.all((au == bu) | (np.isnan(a) & np.isnan(b))))


def replay(rlib, test, mode="auto", make=None):
    """Returns [(step, rendered)] for every render step; raises RenderError on engine errors.
    `make` = a factory of some other object with the renderer's methods (oracle/ref_numpy.py) instead of a C-ABI library."""
    out = []
    with (make() if make else Renderer(rlib, mode=mode)) as r:
        for st in test["steps"]:
            op = st["op"]
            if op == "add_node":
                r.on_add_node(st["handle"], st["effect"])
            elif op == "del_node":
                r.on_del_node(st["handle"])
            elif op == "add_edge":
                r.on_add_edge(st["from"], st["to"], st["from_slot"], st["to_slot"])
            elif op == "del_edge":
                r.on_del_edge(st["from"], st["to"], st["from_slot"], st["to_slot"])
            elif op == "synth_tree":   # one of the repository's seeded synthetic configurations
                from libfriendship_amd import synth
                assert st["kind"] == "additive"
                synth.install(r, synth.additive_tree(st["voices"], st["partials"]))
            elif op == "render":
                rows = bits_to_f32(st["inputs_bits"])
                if st.get("time_ramp_row0"):   # row 0 = the f32 frame ramp of the range (not stored)
                    rows = [np.arange(st["range"][0], st["range"][1], dtype=np.float64).astype(np.float32)] + rows
                got = r.fill_buffer(st["n_slots"], st["range"][0], st["range"][1], rows)
                out.append((st, got))
            else:
                raise ValueError(op)
    return out


def check(rlib, test, mode="auto", make=None):
    for st, got in replay(rlib, test, mode, make):
        if "expect_cols" in st:   # long renders keep only some column ranges of the expected output
            got = np.concatenate([got[:, a:b] for a, b in st["expect_cols"]], axis=1)
        exp = np.array(st["expect_bits"], dtype=np.uint32).view(np.float32).reshape(got.shape)
        assert same_bits(got, exp), f"{test['name']} ({st.get('ref')}): got {got.tolist()} expected {exp.tolist()}"
