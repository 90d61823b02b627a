"""Seeded random graphs of the seven primitives (+ nested composites) for parity tests."""
import numpy as np

from libfriendship_amd.capi import PRIMITIVES, Effect, f32_bits

OPS = ["Delay", "Sum2", "Multiply", "Divide", "Modulo", "Minimum"]
SPECIAL = [0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 3.0, 1e-30, -3.5, 7.25, 1e20, float("inf"), float("nan"), 1.8446744e19, 4.0e19]


def _const(rng):
    r = rng.random()
    if r < 0.25:
        return float(SPECIAL[rng.integers(len(SPECIAL))])
    if r < 0.6:
        return float(rng.integers(-4, 9))
    return float(np.float32(rng.normal() * 3))


def random_effect(rng, depth=0, max_nodes=6):
    """A random composite effect: 1-2 inputs, 1-2 outputs, a few primitive (or nested) nodes."""
    n = int(rng.integers(1, max_nodes + 1))
    n_in = int(rng.integers(1, 3))
    nodes, edges = [], []
    avail = []  # (handle, from_slot) producers inside the effect
    const_h = 1000
    nodes.append((const_h, Effect.primitive("F32Constant")))
    for i in range(n):
        h = i + 1
        if depth < 1 and rng.random() < 0.2:
            sub, sub_in, sub_out = random_effect(rng, depth + 1, 4)
            nodes.append((h, sub))
            arity, outs = sub_in, sub_out
        else:
            nodes.append((h, Effect.primitive(OPS[rng.integers(len(OPS))])))
            arity, outs = 2, 1
        for slot in range(arity):
            r = rng.random()
            if r < 0.15:
                continue  # unconnected -> 0
            if r < 0.45 or not avail:
                if rng.random() < 0.5:
                    edges.append((0, h, int(rng.integers(n_in)), slot))
                else:
                    edges.append((const_h, h, f32_bits(_const(rng)), slot))
            else:
                src = avail[rng.integers(len(avail))]
                edges.append((src[0], h, src[1], slot))
        for o in range(outs):
            avail.append((h, o))
    n_out = int(rng.integers(1, 3))
    for o in range(n_out):
        src = avail[rng.integers(len(avail))]
        edges.append((src[0], 0, src[1], o))
    return Effect.graph(nodes, edges), n_in, n_out


def random_graph(seed, n_nodes=24, n_inputs=2, n_outputs=3, signal_delays=True, composites=True, max_delay=9):
    """Returns a list of steps [('node', h, effect) | ('edge', f, t, fs, ts)] and n_outputs."""
    rng = np.random.default_rng(seed)
    steps = [("node", 1, Effect.primitive("F32Constant"))]
    avail = []
    for i in range(n_nodes):
        h = i + 2
        if composites and rng.random() < 0.15:
            eff, arity, outs = random_effect(rng)
            kind = None
        else:
            kind = OPS[rng.integers(len(OPS))]
            eff, arity, outs = Effect.primitive(kind), 2, 1
        steps.append(("node", h, eff))
        for slot in range(arity):
            r = rng.random()
            if r < 0.08:
                continue
            force_const = kind == "Delay" and slot == 1 and not (signal_delays and rng.random() < 0.4)
            if force_const:
                d = float(rng.integers(0, max_delay)) if rng.random() < 0.8 else _const(rng)   # (max_delay 9: the seeds' original stream)
                steps.append(("edge", 1, h, f32_bits(d), slot))
            elif r < 0.3 or not avail:
                if rng.random() < 0.6:
                    steps.append(("edge", 0, h, int(rng.integers(n_inputs)), slot))
                else:
                    steps.append(("edge", 1, h, f32_bits(_const(rng)), slot))
            else:
                src = avail[rng.integers(len(avail))]
                steps.append(("edge", src[0], h, src[1], slot))
        for o in range(outs):
            avail.append((h, o))
    for o in range(n_outputs):
        if rng.random() < 0.1:
            continue  # unconnected output -> zeros
        src = avail[rng.integers(max(0, len(avail) - 8), len(avail))]
        steps.append(("edge", src[0], 0, src[1], o))
    return steps, n_outputs


def install_steps(r, steps):
    for s in steps:
        if s[0] == "node":
            r.on_add_node(s[1], s[2])
        elif s[0] == "deledge":
            r.on_del_edge(*s[1:])
        else:
            r.on_add_edge(*s[1:])


def random_edits(rng, steps, n_edits, n_inputs=2, n_outputs=3, signal_delays=False):
    """`n_edits` further steps that edit the graph `steps` built (and earlier edits extended): rewire an inbound edge,
    delete one, add a primitive node reading existing ones, or repoint an output.  Edges only run from lower to higher
    handles, so the graph stays acyclic.  Returns the new steps (also appended to `steps`)."""
    prim = {}          # handle -> kind of primitive nodes (composites are left alone as sources)
    inbound = {}       # (to, to_slot) -> edge step
    for s in steps:
        if s[0] == "node":
            prim[s[1]] = PRIMITIVES[s[2].kind] if s[2].kind < len(PRIMITIVES) else None
        elif s[0] == "edge":
            inbound[(s[2], s[4])] = s
        elif s[0] == "deledge":
            inbound.pop((s[2], s[4]), None)
    handles = sorted(h for h in prim if h != 1)
    is_prim = {h for h in handles if prim[h] in OPS}
    out = []

    def source(before):
        cand = [h for h in is_prim if h < before]
        r = rng.random()
        if r < 0.3 or not cand:
            if rng.random() < 0.5:
                return 0, int(rng.integers(n_inputs))
            return 1, f32_bits(_const(rng))
        return cand[rng.integers(len(cand))], 0

    def connect(h, slot):
        if prim.get(h) == "Delay" and slot == 1 and not (signal_delays and rng.random() < 0.4):
            f, fs = 1, f32_bits(float(rng.integers(0, 9)))
        else:
            f, fs = source(h)
        st = ("edge", f, h, fs, slot)
        inbound[(h, slot)] = st
        out.append(st)

    for _ in range(n_edits):
        r = rng.random()
        if r < 0.45 and handles:
            connect(handles[rng.integers(len(handles))], int(rng.integers(2)))
        elif r < 0.55 and inbound:
            keys = [k for k in inbound if k[0] != 0]
            if keys:
                e = inbound.pop(keys[rng.integers(len(keys))])
                out.append(("deledge",) + e[1:])
        elif r < 0.8:
            h = (max(handles) if handles else 1) + 1
            kind = OPS[rng.integers(len(OPS))]
            out.append(("node", h, Effect.primitive(kind)))
            prim[h] = kind
            handles.append(h)
            is_prim.add(h)
            connect(h, 0)
            connect(h, 1)
            if rng.random() < 0.5:
                out.append(("edge", h, 0, 0, int(rng.integers(n_outputs))))
        else:
            f, fs = source(1 << 30)
            out.append(("edge", f, 0, fs, int(rng.integers(n_outputs))))
    steps.extend(out)
    return out


def random_inputs(rng, n_inputs, n_times, kind="mixed"):
    rows = []
    for i in range(n_inputs):
        L = n_times if rng.random() < 0.7 else int(rng.integers(0, n_times + 1))
        if i == 0 and kind != "noise":
            rows.append(None)  # filled by caller with the time ramp
        else:
            rows.append(np.float32(rng.normal(size=L) * 4).astype(np.float32))
    return rows


def random_feedback_graph(seed, n_frames=16, n_nodes=14, n_inputs=2, n_outputs=3, budget=2e5):
    """A random graph of primitives in which one Delay (constant amount 1..6) has been re-pointed at a node that depends on it:
    a feedback loop as the reference evaluates it (reference.rs:197-216).  The oracle's recursion costs (paths round the loop)
    ** (frames / delay) per sample, so candidates beyond `budget` are passed over.  Returns (steps, n_outputs, delay) or None."""
    rng = np.random.default_rng(seed + 77000)
    steps, n_out = random_graph(seed + 77000, n_nodes=n_nodes, n_inputs=n_inputs, n_outputs=n_outputs, signal_delays=False, composites=False, max_delay=7)
    kind, inbound = {}, {}
    for s in steps:
        if s[0] == "node":
            kind[s[1]] = PRIMITIVES[s[2].kind]
        else:
            inbound.setdefault(s[2], {})[s[4]] = s
    delays = []
    for h, k in kind.items():
        amt = inbound.get(h, {}).get(1)
        if k == "Delay" and amt is not None and amt[1] == 1:
            d = float(np.array([amt[3]], dtype=np.uint32).view(np.float32)[0])
            if 1.0 <= d <= 6.0:
                delays.append((h, int(d)))
    rng.shuffle(delays)

    def paths(frm, to, memo):   # operand paths frm -> ... -> to (edges run from lower to higher handles)
        if frm == to:
            return 1
        if frm < to or frm in (0, 1):
            return 0
        if frm not in memo:
            memo[frm] = sum(paths(e[1], to, memo) for e in inbound.get(frm, {}).values())
        return memo[frm]

    for h, d in delays:
        later = [x for x in kind if x >= h and x != 1]
        rng.shuffle(later)
        for src in later:
            p = paths(src, h, {})
            if p == 0:
                continue
            out = [s for s in steps if not (s[0] == "edge" and s[2] == h and s[4] == 0)]
            out.append(("edge", src, h, 0, 0))
            # the loop must be audible: some output depends on the Delay; every way from an output into the loop multiplies the cost
            entries = sum(paths(s[1], h, {}) for s in out if s[0] == "edge" and s[2] == 0)
            if entries == 0 or float(entries) * float(p) ** -(-n_frames // d) > budget:
                continue
            return out, n_out, d
    return None


def safe_feedback_edits(rng, steps, n_edits, n_inputs=2, n_outputs=3):
    """Edits that keep every loop evaluable by the oracle's recursion: another value for a constant operand (never a Delay's
    amount), another amount in 1..6 for a Delay that has a constant one, an output re-pointed at an existing node, a new node
    over two existing ones (sometimes given an output).  Returns the new steps (also appended to `steps`)."""
    kind, inbound = {}, {}
    for s in steps:
        if s[0] == "node":
            kind[s[1]] = PRIMITIVES[s[2].kind]
        elif s[0] == "edge":
            inbound[(s[2], s[4])] = s
        elif s[0] == "deledge":
            inbound.pop((s[2], s[4]), None)
    out = []
    for _ in range(n_edits):
        r = rng.random()
        consts = [k for k, e in inbound.items() if e[1] == 1 and k[0] != 0 and not (kind.get(k[0]) == "Delay" and k[1] == 1)]
        amounts = [k for k, e in inbound.items() if e[1] == 1 and kind.get(k[0]) == "Delay" and k[1] == 1]
        nodes = [h for h in kind if h != 1]
        if r < 0.4 and consts:
            k = consts[rng.integers(len(consts))]
            old = inbound[k]
            new = ("edge", 1, k[0], f32_bits(float(np.float32(rng.normal() * 2))), k[1])
            out += [("deledge",) + old[1:], new]
            inbound[k] = new
        elif r < 0.55 and amounts:
            k = amounts[rng.integers(len(amounts))]
            old = inbound[k]
            d = float(np.array([old[3]], dtype=np.uint32).view(np.float32)[0])
            if 1.0 <= d <= 6.0:   # (a pass-through Delay stays one: it may be what keeps a zero-delay path out of a loop's count;
                #             a loop's Delay never gets shorter: the oracle's recursion costs paths ** (frames / delay))
                new = ("edge", 1, k[0], f32_bits(float(rng.integers(int(d), 8))), k[1])
                out += [("deledge",) + old[1:], new]
                inbound[k] = new
        elif r < 0.75 and nodes:
            o = int(rng.integers(n_outputs))
            old = inbound.get((0, o))
            new = ("edge", int(nodes[rng.integers(len(nodes))]), 0, 0, o)
            if old is not None:
                out.append(("deledge",) + old[1:])
            out.append(new)
            inbound[(0, o)] = new
        elif nodes:
            h = max(kind) + 1
            k2 = ["Sum2", "Multiply", "Minimum"][rng.integers(3)]
            out.append(("node", h, Effect.primitive(k2)))
            kind[h] = k2
            for slot in range(2):
                if rng.random() < 0.6:
                    e = ("edge", int(nodes[rng.integers(len(nodes))]), h, 0, slot)
                elif rng.random() < 0.5:
                    e = ("edge", 0, h, int(rng.integers(n_inputs)), slot)
                else:
                    e = ("edge", 1, h, f32_bits(float(np.float32(rng.normal()))), slot)
                out.append(e)
                inbound[(h, slot)] = e
            if rng.random() < 0.5:
                o = int(rng.integers(n_outputs))
                old = inbound.get((0, o))
                if old is not None:
                    out.append(("deledge",) + old[1:])
                e = ("edge", h, 0, 0, o)
                out.append(e)
                inbound[(0, o)] = e
    steps.extend(out)
    return out
