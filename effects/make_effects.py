#!/usr/bin/env python3
"""Writes the north-star building blocks (SURVEY.md 8a, N1 / N5 / N6 and the triangle partial) as effect-definition files in
the reference's own format: the JSON serde derives for `EffectDesc` (src/routing/effect.rs:44-48,59-74; handles as
{"node_handle": n}, 0 = the effect's own inputs / outputs, src/routing/nullable_int.rs:88-102; primitives named by
`primitive:///X` URLs, effect.rs:357-377).  A libfriendship host -- the reference's Dispatch, or the C++ mirror in
libfriendship_amd/host/ -- finds them by sha256 once this directory is added as a search directory (OscResMan::AddDir,
resman.rs:39-97) and instantiates them as composite nodes; the renderer behind it inlines them.
    python effects/make_effects.py        # rewrites effects/*.fnd and effects/SHA256SUMS (deterministic)
tests/cpp/render_tests.cpp (shipped_effect_files) loads every file through the host's ResMan and checks an instance against
the same nodes placed at top level, bit for bit, on the CPU oracle and on the HIP engine."""
import hashlib
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def prim(name):
    return {"name": name, "sha256": None, "urls": ["primitive:///" + name]}


def by_hash(name, sha):
    return {"name": name, "sha256": list(sha), "urls": []}


class Builder:
    def __init__(self):
        self.nodes, self.edges, self.next = [], [], 1
        self.c = self.node(prim("F32Constant"))      # one constant node serves every constant (the value rides on the edge)

    def node(self, ident):
        h = self.next
        self.next += 1
        self.nodes.append([{"node_handle": h}, ident])
        return h

    def edge(self, frm, to, from_slot, to_slot):
        self.edges.append({"from": {"node_handle": frm}, "to": {"node_handle": to}, "weight": {"from_slot": from_slot, "to_slot": to_slot}})

    def op(self, name, a, b):
        """a, b: ("in", slot) | ("c", value) | ("n", handle[, from_slot])"""
        h = self.node(prim(name))
        for slot, x in enumerate((a, b)):
            if x[0] == "in":
                self.edge(0, h, x[1], slot)
            elif x[0] == "c":
                self.edge(self.c, h, bits(x[1]), slot)
            else:
                self.edge(x[1], h, x[2] if len(x) > 2 else 0, slot)
        return ("n", h)

    def out(self, x, slot=0):
        self.edge(x[1], 0, x[2] if len(x) > 2 else 0, slot)

    def desc(self, name, inputs, outputs):
        io = lambda names: [{"name": n, "channel": i} for i, n in enumerate(names)]
        return {"meta": {"id": {"name": name, "sha256": None, "urls": []}, "inputs": io(inputs), "outputs": io(outputs)},
                "adjlist": {"nodes": self.nodes, "edges": self.edges}}


def partial():
    """N1: Partial(t, w, amp) = amp * parabolic_sine(Modulo(t * w, 1))."""
    b = Builder()
    x = b.op("Multiply", ("in", 0), ("in", 1))
    ph = b.op("Modulo", x, ("c", 1.0))
    u = b.op("Sum2", ph, ("c", -0.5))
    nu = b.op("Multiply", ("c", -1.0), u)
    m = b.op("Minimum", u, nu)
    absu = b.op("Multiply", ("c", -1.0), m)
    n1 = b.op("Multiply", ("c", -1.0), absu)
    q = b.op("Sum2", ("c", 0.5), n1)
    p = b.op("Multiply", ("c", -16.0), u)
    y = b.op("Multiply", p, q)
    b.out(b.op("Multiply", ("in", 2), y))
    return b.desc("Partial", ["time", "increment", "amplitude"], ["partial"])


def triangle():
    """TrianglePartial(t, w, amp) = amp * (1 - 4 * |Modulo(t * w, 1) - 0.5|)."""
    b = Builder()
    ph = b.op("Modulo", b.op("Multiply", ("in", 0), ("in", 1)), ("c", 1.0))
    u = b.op("Sum2", ph, ("c", -0.5))
    au = b.op("Multiply", ("c", -1.0), b.op("Minimum", u, b.op("Multiply", ("c", -1.0), u)))
    tri = b.op("Sum2", ("c", 1.0), b.op("Multiply", ("c", -4.0), au))
    b.out(b.op("Multiply", ("in", 2), tri))
    return b.desc("TrianglePartial", ["time", "increment", "amplitude"], ["partial"])


def adsr():
    """N5: Envelope(t, x) = x * max(0, min(min(t/A, max(S, 1 - (1-S)*(t-A)/D)), S*(T_end - t)/R)) with SURVEY.md 8d's
    A = 480, D = 2400, S = 0.6, R = 4800, T_end = 48000 frames; max(a, b) = -Minimum(-a, -b) (effect.rs:106-111)."""
    A, D, S, R, T_end = 480.0, 2400.0, 0.6, 4800.0, 48000.0
    b = Builder()
    neg = lambda v: b.op("Multiply", ("c", -1.0), v)
    vmax = lambda p, q: neg(b.op("Minimum", neg(p), neg(q)))
    t = ("in", 0)
    a = b.op("Divide", t, ("c", A))
    slope = b.op("Divide", b.op("Multiply", ("c", 1.0 - S), b.op("Sum2", t, ("c", -A))), ("c", D))
    d = vmax(("c", S), b.op("Sum2", ("c", 1.0), neg(slope)))
    r = b.op("Divide", b.op("Multiply", ("c", S), b.op("Sum2", ("c", T_end), neg(t))), ("c", R))
    env = vmax(("c", 0.0), b.op("Minimum", b.op("Minimum", a, d), r))
    b.out(b.op("Multiply", env, ("in", 1)))
    return b.desc("Envelope", ["time", "signal"], ["shaped"])


def tap():
    """N6: Tap(x, gain, frames) = x + gain * Delay(x, frames): one feed-forward tap; a delay chain is K of them in series."""
    b = Builder()
    dl = b.op("Delay", ("in", 0), ("in", 2))
    b.out(b.op("Sum2", ("in", 0), b.op("Multiply", ("in", 1), dl)))
    return b.desc("Tap", ["signal", "gain", "frames"], ["mixed"])


def voice4(partial_sha):
    """A nested effect: Voice4(t, f0) = four Partial instances at harmonics 1..4 of increment f0 (N3: Multiply(f0, C(k+1)))
    with amplitudes 1/(k+1), summed by a balanced Sum2 tree (N2) -- Partial is referenced by the sha256 of partial.fnd."""
    b = Builder()
    leaves = []
    for k in range(4):
        w = b.op("Multiply", ("in", 1), ("c", float(k + 1)))
        h = b.node(by_hash("Partial", partial_sha))
        b.edge(0, h, 0, 0)
        b.edge(w[1], h, 0, 1)
        b.edge(b.c, h, bits(1.0 / (k + 1)), 2)
        leaves.append(("n", h))
    b.out(b.op("Sum2", b.op("Sum2", leaves[0], leaves[1]), b.op("Sum2", leaves[2], leaves[3])))
    return b.desc("Voice4", ["time", "increment"], ["voice"])


def write(name, desc):
    text = json.dumps(desc, separators=(",", ":"))      # the compact form serde_json::to_writer produces, same key order
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        f.write(text)
    return hashlib.sha256(text.encode()).digest()


if __name__ == "__main__":
    sums = {}
    sums["partial.fnd"] = write("partial.fnd", partial())
    sums["triangle.fnd"] = write("triangle.fnd", triangle())
    sums["envelope.fnd"] = write("envelope.fnd", adsr())
    sums["tap.fnd"] = write("tap.fnd", tap())
    sums["voice4.fnd"] = write("voice4.fnd", voice4(sums["partial.fnd"]))
    with open(os.path.join(HERE, "SHA256SUMS"), "w") as f:
        for k in sorted(sums):
            f.write(f"{sums[k].hex()}  {k}\n")
    print("\n".join(f"{v.hex()}  {k}" for k, v in sorted(sums.items())))
