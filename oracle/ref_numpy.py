"""TEST INFRASTRUCTURE -- a second, independent CPU restatement of the reference renderer, in numpy.

Only tests/ may import this file.  It exists to triangulate oracle/ref_renderer.cpp (the C++ restatement every parity
test compares against): the two were written separately from the same source text, share no code, and must agree bit
for bit on every graph the tests throw at them (tests/test_oracle_numpy.py).  Like the C++ oracle it is pinned by the
reference's own known-answer tests (tests/golden/reference_kat.json).

Follows /root/reference/src/render/reference.rs:
    fill_buffer            :47-86    input store (seek, the buff.len() quirk, asserts, last-value padding), render loops
    get_sample             :90-96    out-of-range input reads are 0
    GraphWatcher           :115-137  add/del node, add/del edge
    NodeMap::add_edge      :140-153
    get_output / get_maybe_edge_value / get_edge_value   :158-265   the seven primitives and nested effects
Where the reference evaluates one (time, slot) at a time, this file evaluates a whole vector of times per visit (numpy
f32 arithmetic is IEEE exactly-rounded +, *, / and C fmodf, the same operations Rust's f32 uses); the visiting order,
the absence of memoisation and every rule are the reference's.

Unpinned by any reference vector, same choices as the C++ oracle:
    f32::min            (a < b || b != b) ? a : b          [Rust core, 2017]
    NaN as u64          0
`semantics="sparkle"` applies the two documented divergences of the LLVM renderer (sparkle.rs:492-498, :531-534):
    Minimum             select(fcmp ult a, b, a, b)  -> a NaN left operand wins
    Delay               an amount that is not >= 0 (negative or NaN) makes the output 0.0
"""
import sys

import numpy as np

PRIMS = ["Delay", "F32Constant", "Sum2", "Multiply", "Divide", "Modulo", "Minimum"]   # effect.rs:86-112 order
TWO64 = np.float32(18446744073709551616.0)


class RefPanic(AssertionError):
    """Where the reference would panic (assert!/unwrap/expect/index)."""


class _Node:
    def __init__(self, data):
        self.data = data          # a primitive's name, or a _NodeMap (nested effect)
        self.inbound = []         # Option<Edge> per slot


class _NodeMap:
    def __init__(self):
        self.nodes = {}
        self.output_edges = []

    def add_edge(self, edge):                                    # reference.rs:140-153
        frm, to, from_slot, to_slot = edge
        if to == 0:
            inbound = self.output_edges
        else:
            if to not in self.nodes:
                raise RefPanic("add_edge: unknown node (unwrap on None, reference.rs:145)")
            inbound = self.nodes[to].inbound
        while len(inbound) <= to_slot:
            inbound.append(None)
        inbound[to_slot] = edge

    # ---- evaluation: `times` is a vector of u64 sample indices, the result the f32 value at each of them ------------
    def get_output(self, times, slot, get_input, sparkle):       # :158-161
        edge = self.output_edges[slot] if slot < len(self.output_edges) else None
        return self.maybe_edge_value(times, edge, get_input, sparkle)

    def maybe_edge_value(self, times, edge, get_input, sparkle):  # :164-173
        if edge is None:
            return np.zeros(len(times), np.float32)
        return self.edge_value(times, edge, get_input, sparkle)

    def edge_value(self, times, edge, get_input, sparkle):        # :178-265
        frm, _to, from_slot, _to_slot = edge
        if frm == 0:
            return get_input(times, from_slot)
        if frm not in self.nodes:
            raise RefPanic("edge from an unknown node (index panic, reference.rs:186)")
        node = self.nodes[frm]

        def inb(slot):
            return node.inbound[slot] if slot < len(node.inbound) else None

        if isinstance(node.data, _NodeMap):                      # nested effect: its inputs are this node's inbound edges
            return node.data.get_output(times, from_slot,
                                        lambda t2, s2: self.maybe_edge_value(t2, inb(s2), get_input, sparkle), sparkle)
        prim = node.data
        if prim == "F32Constant":                                # the value rides in from_slot (:217-220)
            return np.full(len(times), np.uint32(from_slot).view(np.float32), np.float32)
        if from_slot != 0:
            raise RefPanic("assert!(from_slot == 0)")
        if prim == "Delay":                                      # :197-216
            d = self.maybe_edge_value(times, inb(1), get_input, sparkle)
            out = np.zeros(len(times), np.float32)
            with np.errstate(invalid="ignore"):
                live = ~(d >= TWO64)                             # >= 2^64: "indexing from negative time" -> 0
                if sparkle:
                    live &= d >= np.float32(0)                   # sparkle.rs:531-534: ult 0 (negative or NaN) -> 0.0
                clamp = ~(d >= np.float32(0))                    # d < 0 -> 0 frames; NaN as u64 -> 0
                dd = np.where(clamp | ~live, np.float32(0), d).astype(np.float64)   # exact: f32 -> f64
            hi = dd >= 2.0 ** 63                                 # floor toward zero, all of [0, 2^64)
            frames = np.where(hi, dd - 2.0 ** 63, dd).astype(np.uint64) + np.where(hi, np.uint64(1) << np.uint64(63), np.uint64(0))
            live &= times >= frames                              # checked_sub: before time 0 the value is 0
            if live.any():
                out[live] = self.maybe_edge_value(times[live] - frames[live], inb(0), get_input, sparkle)
            return out
        a = self.maybe_edge_value(times, inb(0), get_input, sparkle)
        b = self.maybe_edge_value(times, inb(1), get_input, sparkle)
        with np.errstate(all="ignore"):
            if prim == "Sum2":
                return a + b
            if prim == "Multiply":
                return a * b
            if prim == "Divide":
                return a / b
            if prim == "Minimum":
                r = np.where((a < b) | (b != b), a, b)
                return np.where(a != a, a, r) if sparkle else r
            if prim == "Modulo":                                 # :249-262
                rem = np.fmod(a, b)
                return np.where(rem < np.float32(0), rem + b, rem)
        raise RefPanic(f"unknown primitive {prim}")


class NumpyRefRenderer:
    """Same surface as libfriendship_amd.capi.Renderer (the calls the tests make)."""

    def __init__(self, semantics="reference"):
        self.nodes = _NodeMap()
        self.inputs = []        # per slot: np.float32 array, or an int = that many zeros (never fed; see fill_buffer)
        self.head = 0
        self.sparkle = semantics == "sparkle"
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    # ---- GraphWatcher (:115-137) ------------------------------------------------------------------------------------
    def _make_node(self, effect):                               # :98-113
        if effect.kind < len(PRIMS):
            return PRIMS[effect.kind]
        sub = _NodeMap()
        for handle, child in effect._nodes:
            sub.nodes[handle] = _Node(self._make_node(child))
        for e in effect._edges:
            sub.add_edge(tuple(int(x) for x in e))
        return sub

    def on_add_node(self, handle, effect):
        if isinstance(effect, (str, dict)):                     # fixture form
            from libfriendship_amd.capi import Effect
            effect = Effect.from_json(effect)
        self.nodes.nodes[handle] = _Node(self._make_node(effect))

    def on_del_node(self, handle):
        self.nodes.nodes.pop(handle, None)

    def on_add_nodes(self, handles, effects):                   # the C ABI's batch forms: loops over the calls above
        many = isinstance(effects, (list, tuple))
        for i, h in enumerate(handles):
            self.on_add_node(int(h), effects[i] if many else effects)

    def on_add_edges(self, edges):
        for e in np.asarray(edges).reshape(-1, 4):
            self.on_add_edge(*(int(x) for x in e))

    def on_add_edge(self, frm, to, from_slot, to_slot):
        self.nodes.add_edge((frm, to, from_slot, to_slot))

    def on_del_edge(self, frm, to, from_slot, to_slot):
        if to == 0:
            inbound = self.nodes.output_edges
        else:
            if to not in self.nodes.nodes:
                raise RefPanic("Attempt to delete edge, but it was never created!")
            inbound = self.nodes.nodes[to].inbound
        if to_slot < len(inbound):
            inbound[to_slot] = None

    # ---- Renderer::fill_buffer (:47-86) -----------------------------------------------------------------------------
    def fill_buffer(self, n_slots, start, end, inputs=()):
        idx, n_times = start, end - start
        if idx != self.head:                                     # seek: forget history (:52-58)
            self.inputs = [idx] * len(self.inputs)
        # `buff.len()` is n_slots * n_times, not a slot count (:60-65); the extra vectors are only ever zeros, kept as a length
        while len(self.inputs) < n_slots * n_times:
            self.inputs.append(idx)
        for row, slot in zip(inputs, range(len(self.inputs))):   # zip: rows beyond the stored vectors are dropped
            cur = self.inputs[slot]
            cur = np.zeros(cur, np.float32) if isinstance(cur, int) else cur
            if len(cur) != idx:
                raise RefPanic("assert_eq!(vec_dest.len(), idx)")
            cur = np.concatenate([cur, np.asarray(row, np.float32).ravel()])
            if len(cur) > idx + n_times:
                raise RefPanic("cannot send inputs ahead of outputs")
            pad = cur[-1] if len(cur) else np.float32(0)
            self.inputs[slot] = np.concatenate([cur, np.full(idx + n_times - len(cur), pad, np.float32)])
        out = np.zeros((n_slots, n_times), np.float32)
        times = np.arange(idx, idx + n_times, dtype=np.uint64)
        for slot in range(n_slots):
            out[slot] = self.nodes.get_output(times, slot, self._get_input, self.sparkle)
        self.head = idx + n_times
        return out

    def _get_input(self, times, slot):                           # :91-95
        out = np.zeros(len(times), np.float32)
        if slot < len(self.inputs) and not isinstance(self.inputs[slot], int):
            v = self.inputs[slot]
            ok = times < np.uint64(len(v))
            out[ok] = v[times[ok].astype(np.int64)]
        return out
