#!/usr/bin/env python3
"""Writes tests/golden/reference_kat.json: the reference's own known-answer tests as DATA.

Every entry below is a transcription of the *inputs and expected outputs* of one integration test of
the reference (graph-edit messages, render ranges, input rows, asserted arrays) -- no reference source
text.  Citations are relative to the reference checkout.  Floats are stored as IEEE-754 bit patterns
(u32) so the fixture is exact; `expect` repeats them as decimals for readability only.

The reference does not need to be present to run this script.
"""
import json
import os
import struct

import numpy as np


def bits(x):
    return struct.unpack("<I", struct.pack("<f", float(np.float32(x))))[0]


def node(handle, effect):
    return {"op": "add_node", "handle": handle, "effect": effect}


def edge(frm, to, from_slot, to_slot):
    return {"op": "add_edge", "from": frm, "to": to, "from_slot": from_slot, "to_slot": to_slot}


def cedge(const_handle, to, value, to_slot):
    """Edge out of an F32Constant node: the value travels as from_slot = value.to_bits()."""
    return edge(const_handle, to, bits(value), to_slot)


def render(start, end, n_slots, inputs, expect, ref):
    exp = [[float(np.float32(v)) for v in row] for row in expect]
    return {"op": "render", "range": [start, end], "n_slots": n_slots,
            "inputs_bits": [[bits(v) for v in row] for row in inputs],
            "expect_bits": [[bits(v) for v in row] for row in exp],
            "expect": exp, "ref": ref}


def binop_test(name, effect, a, b, expect, ref_test, ref_assert):
    # tests/render_prim.rs:132-290 all share this shape: op node = 1 -> out0, const 2 -> in0, const 3 -> in1
    return {"name": name, "ref": ref_test, "steps": [
        node(1, effect), edge(1, 0, 0, 0),
        node(2, "F32Constant"), cedge(2, 1, a, 0),
        node(3, "F32Constant"), cedge(3, 1, b, 1),
        render(0, 4, 1, [], [[expect] * 4], ref_assert)]}


f32 = np.float32
tests = [
    {"name": "render_zeros", "ref": "tests/render_prim.rs:69-80", "steps": [
        render(0, 4, 1, [], [[0, 0, 0, 0]], "tests/render_prim.rs:79")]},
    {"name": "render_const", "ref": "tests/render_prim.rs:82-98", "steps": [
        node(1, "F32Constant"), cedge(1, 0, 0.5, 0),
        render(0, 4, 1, [], [[0.5] * 4], "tests/render_prim.rs:97")]},
    {"name": "render_delay", "ref": "tests/render_prim.rs:100-129", "steps": [
        node(1, "Delay"), edge(1, 0, 0, 0),
        node(2, "F32Constant"), cedge(2, 1, 0.5, 0),
        node(3, "F32Constant"), cedge(3, 1, 2.0, 1),
        render(0, 4, 1, [], [[0, 0, 0.5, 0.5]], "tests/render_prim.rs:128")]},
    binop_test("render_mult", "Multiply", 0.5, -3.0, -1.5, "tests/render_prim.rs:131-162", "tests/render_prim.rs:161"),
    binop_test("render_sum2", "Sum2", 0.5, -3.0, -2.5, "tests/render_prim.rs:164-195", "tests/render_prim.rs:194"),
    binop_test("render_div", "Divide", 0.5, -3.0, f32(0.5) / f32(-3.0), "tests/render_prim.rs:197-227", "tests/render_prim.rs:225-226"),
    binop_test("render_mod", "Modulo", -3.5, 2.0, 0.5, "tests/render_prim.rs:229-259", "tests/render_prim.rs:257-258"),
    binop_test("render_min", "Minimum", -3.5, 2.0, -3.5, "tests/render_prim.rs:261-291", "tests/render_prim.rs:289-290"),
    {"name": "ext_render_passthrough", "ref": "tests/ext_input.rs:46-81", "steps": [
        edge(0, 0, 0, 0),
        render(0, 4, 1, [[1, 2, 3, 4]], [[1, 2, 3, 4]], "tests/ext_input.rs:62"),
        render(4, 8, 1, [[0, 1, 2]], [[0, 1, 2, 2]], "tests/ext_input.rs:72"),
        render(0, 4, 1, [], [[0, 0, 0, 0]], "tests/ext_input.rs:80")]},
    {"name": "ext_render_delay", "ref": "tests/ext_input.rs:83-122", "steps": [
        node(1, "Delay"), edge(1, 0, 0, 0), edge(0, 1, 0, 0),
        render(0, 4, 1, [[1, 2, 3, 4]], [[1, 2, 3, 4]], "tests/ext_input.rs:105"),
        node(2, "F32Constant"), cedge(2, 1, 1.0, 1),
        render(4, 8, 1, [[1, 2, 3, 4]], [[4, 1, 2, 3]], "tests/ext_input.rs:121")]},
    {"name": "load_multby2", "ref": "tests/load_effect.rs:42-112", "steps": [
        # The composite "MulBy2" (tests/load_effect.rs:42-65): in0 -> Multiply.0, Const(5.0) -> Multiply.1,
        # Multiply -> out0.  The reference finds it on disk by sha256; the renderer only ever sees the
        # loaded RouteGraph, which is what the fixture carries.
        node(1, {"graph": {
            "nodes": [[1, "Multiply"], [2, "F32Constant"]],
            "edges": [[0, 1, 0, 0], [1, 0, 0, 0], [2, 1, bits(5.0), 1]]}}),
        edge(1, 0, 0, 0),
        node(2, "F32Constant"), cedge(2, 1, 0.5, 0),
        render(0, 4, 1, [], [[2.5] * 4], "tests/load_effect.rs:111")]},
]

doc = {
    "description": "Known-answer tests of Wallacoloo/libfriendship's render path, transcribed as data "
                   "(11 tests, 14 asserted arrays). Node handle 0 = graph I/O. Edge arrays are "
                   "[from, to, from_slot, to_slot]. F32Constant values ride in from_slot as f32 bits.",
    "generator": "tests/golden/make_reference_kat.py",
    "tests": tests,
}

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kat.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    n_arrays = sum(1 for t in tests for s in t["steps"] if s["op"] == "render")
    print(f"wrote {out}: {len(tests)} tests, {n_arrays} asserted arrays")
