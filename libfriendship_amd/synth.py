"""Synthetic additive-synthesis trees (BASELINE.json configs) as graphs of the
reference's seven primitives.

The reference has no oscillator/harmonics/detune/envelope effect: each is a sub-graph
of Delay/F32Constant/Sum2/Multiply/Divide/Modulo/Minimum.  This module builds those graphs as plain
arrays (node handles + primitive kinds, 16-byte edges) that any renderer behind the C ABI accepts --
the HIP engine and, in tests, the CPU oracle receive byte-identical graphs.

Definitions (t = external input slot 0, the f32 frame ramp; C(x) = F32Constant edge carrying x):
  partial  : x = Multiply(t, C(w));  phase = Modulo(x, C(1));  u = Sum2(phase, C(-0.5))
             absu = Multiply(C(-1), Minimum(u, Multiply(C(-1), u)))
             y = Multiply(Multiply(C(-16), u), Sum2(C(0.5), Multiply(C(-1), absu)))     parabolic sine
             leaf = Multiply(C(amp), y)                                                   11 nodes
  voice    : balanced binary Sum2 tree over the voice's leaves (adjacent pairs, level by level)
  harmonics: f = Multiply(C(f0), C(k+1))  [N3];  detune: f' = Multiply(f, C(1+delta))  [N4];  w = Divide(f', C(sr))
             -- as graph NODES with `params_as_nodes=True` (the reference has only the seven primitives, so this is how
             harmonics and detune reach a renderer; the engine folds them at lowering with the same exactly-rounded
             f32 ops), or pre-folded in numpy (same roundings, fewer nodes; the default for small tests)
  envelope : ADSR from Minimum/Sum2/Multiply/Divide of t, Multiply(env, mix)
  delay    : K feed-forward taps y = Sum2(x, Multiply(C(g), Delay(x, C(d)))) in series (N6)
"""
import numpy as np

from .capi import FR_PRIM, Effect, f32_bits

K_DELAY, K_CONST, K_SUM2, K_MUL, K_DIV, K_MOD, K_MIN = (FR_PRIM[n] for n in
                                                         ("Delay", "F32Constant", "Sum2", "Multiply", "Divide", "Modulo", "Minimum"))
CONST_HANDLE = 1  # one F32Constant node serves every constant (the value rides on the edge)


def splitmix64(seed, n):
    """n outputs of SplitMix64 started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


class GraphArrays:
    """Accumulates nodes and edges; handles are assigned sequentially from 2 (1 = the constant node)."""

    def __init__(self):
        self.handles = [np.array([CONST_HANDLE], dtype=np.uint32)]
        self.kinds = [np.array([K_CONST], dtype=np.int32)]
        self.edges = []
        self.next = 2

    def nodes(self, kind, n):
        h = np.arange(self.next, self.next + n, dtype=np.uint32)
        self.next += n
        self.handles.append(h)
        self.kinds.append(np.full(n, kind, dtype=np.int32))
        return h

    def edge(self, frm, to, from_slot, to_slot):
        frm, to, from_slot, to_slot = np.broadcast_arrays(
            np.asarray(frm, dtype=np.uint32), np.asarray(to, dtype=np.uint32),
            np.asarray(from_slot, dtype=np.uint32), np.asarray(to_slot, dtype=np.uint32))
        self.edges.append(np.stack([frm.ravel(), to.ravel(), from_slot.ravel(), to_slot.ravel()], axis=1))

    def const(self, to, value, to_slot):
        """C(value) -> to.to_slot"""
        self.edge(CONST_HANDLE, to, bits(value), to_slot)

    def binop(self, kind, a, b, n=None):
        """New nodes kind(a, b); a/b are handle arrays or ('c', values) constants or ('in', slot)."""
        n = n if n is not None else max(_len(a), _len(b))
        h = self.nodes(kind, n)
        for operand, slot in ((a, 0), (b, 1)):
            if isinstance(operand, tuple) and operand[0] == "c":
                self.const(h, operand[1], slot)
            elif isinstance(operand, tuple) and operand[0] == "in":
                self.edge(0, h, operand[1], slot)
            else:
                self.edge(operand, h, 0, slot)
        return h

    def finish(self, n_outputs):
        return {"handles": np.concatenate(self.handles), "kinds": np.concatenate(self.kinds),
                "edges": np.ascontiguousarray(np.concatenate(self.edges, axis=0), dtype=np.uint32),
                "n_outputs": n_outputs}


def _len(x):
    if isinstance(x, tuple):
        return _len(x[1]) if x[0] == "c" else 1
    return int(np.size(x))


def C(v):
    return ("c", np.asarray(v, dtype=np.float32))


def IN(slot):
    return ("in", slot)


def harmonic_nodes(g, p):
    """N3 + N4 as primitive nodes: for every partial f = Multiply(C(f0_v), C(k+1)), [f = Multiply(f, C(1 + delta))],
    w = Divide(f, C(sr)).  `p` = voice_params(...).  Returns the handles of the w nodes, flattened [V * P]."""
    V, P = p["w"].shape
    n = V * P
    f0 = np.broadcast_to(p["f0"], (V, P)).ravel()
    k1 = np.broadcast_to(p["k1"], (V, P)).ravel()
    f = g.binop(K_MUL, C(f0), C(k1), n)
    if p["delta"] is not None:
        f = g.binop(K_MUL, f, C((np.float32(1.0) + p["delta"]).astype(np.float32).ravel()), n)
    return g.binop(K_DIV, f, C(np.full(n, p["sr"], dtype=np.float32)), n)


def partial_leaves(g, w, amp, time_slot=0, w_nodes=None):
    """The 11-node partial oscillator for every (w, amp); returns leaf handles (same shape, flattened).  With
    `w_nodes` (handles, e.g. from harmonic_nodes) the phase increment is a node's output instead of a constant."""
    amp = np.asarray(amp, dtype=np.float32).ravel()
    n = len(amp)
    if w_nodes is not None:
        x = g.binop(K_MUL, IN(time_slot), np.asarray(w_nodes, dtype=np.uint32).ravel(), n)
    else:
        w = np.asarray(w, dtype=np.float32).ravel()
        x = g.binop(K_MUL, IN(time_slot), C(w), n)
    ph = g.binop(K_MOD, x, C(np.float32(1.0)), n)
    u = g.binop(K_SUM2, ph, C(np.float32(-0.5)), n)
    nu = g.binop(K_MUL, C(np.float32(-1.0)), u, n)
    m = g.binop(K_MIN, u, nu, n)
    ab = g.binop(K_MUL, C(np.float32(-1.0)), m, n)
    n1 = g.binop(K_MUL, C(np.float32(-1.0)), ab, n)
    q = g.binop(K_SUM2, C(np.float32(0.5)), n1, n)
    p = g.binop(K_MUL, C(np.float32(-16.0)), u, n)
    y = g.binop(K_MUL, p, q, n)
    return g.binop(K_MUL, C(amp), y, n)


def track_leaves(g, w_slots, amp_slots, time_slot=0):
    """The same 11-node partial oscillator with its phase increment and its amplitude read from INPUT rows (control-rate
    tracks, one pair of rows per partial: reference.rs:66-74,181-183) instead of constants."""
    w_slots = np.asarray(w_slots, dtype=np.uint32).ravel()
    amp_slots = np.asarray(amp_slots, dtype=np.uint32).ravel()
    n = len(w_slots)
    x = g.nodes(K_MUL, n)
    g.edge(0, x, time_slot, 0)
    g.edge(0, x, w_slots, 1)
    ph = g.binop(K_MOD, x, C(np.float32(1.0)), n)
    u = g.binop(K_SUM2, ph, C(np.float32(-0.5)), n)
    nu = g.binop(K_MUL, C(np.float32(-1.0)), u, n)
    m = g.binop(K_MIN, u, nu, n)
    ab = g.binop(K_MUL, C(np.float32(-1.0)), m, n)
    n1 = g.binop(K_MUL, C(np.float32(-1.0)), ab, n)
    q = g.binop(K_SUM2, C(np.float32(0.5)), n1, n)
    p = g.binop(K_MUL, C(np.float32(-16.0)), u, n)
    y = g.binop(K_MUL, p, q, n)
    leaf = g.nodes(K_MUL, n)
    g.edge(0, leaf, amp_slots, 0)
    g.edge(y, leaf, 0, 1)
    return leaf


def track_tree(n_voices, n_partials, time_slot=0, first_track=1):
    """The HBM-bound variant of configs B/C: every partial's phase increment w and amplitude arrive as input
    rows -- slot first_track + 2 (v P + k) holds w, the next one amp -- so a frame of a partial costs 8 bytes of HBM.
    Output slot per voice; `n_inputs` rows per call (time row + tracks)."""
    g = GraphArrays()
    n = n_voices * n_partials
    w_slots = first_track + 2 * np.arange(n, dtype=np.uint32)
    leaves = track_leaves(g, w_slots, w_slots + 1, time_slot).reshape(n_voices, n_partials)
    roots = sum_tree(g, leaves)
    g.edge(roots, 0, 0, np.arange(n_voices, dtype=np.uint32))
    t = g.finish(n_voices)
    t["first_track"] = first_track
    t["n_inputs"] = first_track + 2 * n
    return t


def track_rows(n_voices, n_partials, start, end, seed=0x5EED0200, first_track=1, time_slot=0, sr=48000.0):
    """The dense input matrix [first_track + 2 V P, end - start] of a call of track_tree: the time ramp, and per partial a
    slowly gliding w (vibrato: +-0.3 % at a few hertz, phase per partial) and a decaying, tremolo'd amplitude."""
    p = voice_params(n_voices, n_partials, seed, sr=sr)
    t = np.arange(start, end, dtype=np.float64)[None, :]
    n = n_voices * n_partials
    ph = uniform01(seed, n).astype(np.float64)[:, None] * 6.283185307179586
    rate = (3.0 + 4.0 * uniform01(seed + 1, n).astype(np.float64))[:, None] / sr
    w = p["w"].astype(np.float64).reshape(n, 1) * (1.0 + 0.003 * np.sin(6.283185307179586 * rate * t + ph))
    amp = p["amp"].astype(np.float64).reshape(n, 1) * (0.75 + 0.25 * np.cos(6.283185307179586 * 2.0 * rate * t + ph)) * np.exp(-t / (4.0 * sr))
    m = np.zeros((first_track + 2 * n, end - start), dtype=np.float32)
    m[time_slot] = time_ramp(start, end)
    m[first_track::2] = w.astype(np.float32)
    m[first_track + 1::2] = amp.astype(np.float32)
    return m


def triangle_leaves(g, w, amp, time_slot=0, am_slot=None):
    """A different partial: triangle wave  amp * (1 - 4*|phase - 0.5|), optionally amplitude-modulated by a second
    input (leaf * in[am_slot]).  Not the hand-matched template: the engine specialises it with hipRTC (csrc/jit.hpp)."""
    w = np.asarray(w, dtype=np.float32).ravel()
    amp = np.asarray(amp, dtype=np.float32).ravel()
    n = len(w)
    f = np.float32
    x = g.binop(K_MUL, IN(time_slot), C(w), n)
    ph = g.binop(K_MOD, x, C(f(1.0)), n)
    u = g.binop(K_SUM2, ph, C(f(-0.5)), n)
    au = g.binop(K_MUL, C(f(-1.0)), g.binop(K_MIN, u, g.binop(K_MUL, C(f(-1.0)), u, n), n), n)
    tri = g.binop(K_SUM2, C(f(1.0)), g.binop(K_MUL, C(f(-4.0)), au, n), n)
    leaf = g.binop(K_MUL, C(amp), tri, n)
    if am_slot is not None:
        leaf = g.binop(K_MUL, leaf, IN(am_slot), n)
    return leaf


def sum_tree(g, leaves):
    """Balanced binary Sum2 tree per row of `leaves` [V, P]: adjacent pairs level by level (an odd
    element is carried up unchanged).  Returns the root handle per row."""
    cur = np.asarray(leaves, dtype=np.uint32)
    while cur.shape[1] > 1:
        npair = cur.shape[1] // 2
        a = cur[:, 0:2 * npair:2]
        b = cur[:, 1:2 * npair:2]
        s = g.binop(K_SUM2, a.ravel(), b.ravel(), a.size).reshape(a.shape)
        cur = np.concatenate([s, cur[:, 2 * npair:]], axis=1) if cur.shape[1] % 2 else s
    return cur[:, 0]


def voice_params(n_voices, n_partials, seed, detune=False, sr=48000.0, wrap=None):
    """SURVEY.md 8d parameters, every step rounded to f32 like a graph of f32 primitives would:
    f0_v = 55*2^(v/12); f = f0_v*(k+1) [*(1+delta)]; w = f/sr; amp = 1/(k+1).
    `wrap` (not part of the survey's definition; shape sweeps only): fundamentals repeat every `wrap` voices, each
    repetition detuned by 1e-4 so that voices stay distinct -- with hundreds of voices the survey's formula puts most
    fundamentals far above any audible or even representable pitch (every t*w beyond 2^23: identically zero mixes)."""
    v = np.arange(n_voices, dtype=np.float64)[:, None]
    k1 = np.arange(1, n_partials + 1, dtype=np.float32)[None, :]
    if wrap:
        f0 = (55.0 * 2.0 ** ((v % wrap) / 12.0) * (1.0 + 1e-4 * (v // wrap))).astype(np.float32)
    else:
        f0 = (55.0 * 2.0 ** (v / 12.0)).astype(np.float32)
    f = (f0 * k1).astype(np.float32)
    delta = None
    if detune:
        u = uniform01(seed, n_voices * n_partials).reshape(n_voices, n_partials)
        delta = ((u - 0.5) * 0.01).astype(np.float32)
        f = (f * (np.float32(1.0) + delta).astype(np.float32)).astype(np.float32)
    w = (f / np.float32(sr)).astype(np.float32)
    amp = (np.float32(1.0) / k1).astype(np.float32) * np.ones((n_voices, 1), dtype=np.float32)
    return {"f0": f0, "k1": k1, "delta": delta, "w": w, "amp": amp.astype(np.float32), "sr": np.float32(sr)}


def additive_tree(n_voices, n_partials, seed=0x5EED0002, detune=False, sr=48000.0, time_slot=0, params_as_nodes=False, voices=None):
    """configs B/C (and the oscillator part of D/E): V voices x P partials, one output slot per voice.
    `voices`: keep only these voices of the V (output slot i = voice voices[i]) -- a sub-tree of the same job, for
    checking a few voices of a tree too big for the CPU oracle."""
    p = voice_params(n_voices, n_partials, seed, detune, sr)
    if voices is not None:
        sel = np.asarray(voices, dtype=np.int64)
        p = dict(p, f0=p["f0"][sel], w=p["w"][sel], amp=p["amp"][sel], delta=None if p["delta"] is None else p["delta"][sel])
        n_voices = len(sel)
    g = GraphArrays()
    wn = harmonic_nodes(g, p) if params_as_nodes else None
    leaves = partial_leaves(g, p["w"], p["amp"], time_slot, wn).reshape(n_voices, n_partials)
    roots = sum_tree(g, leaves)
    g.edge(roots, 0, 0, np.arange(n_voices, dtype=np.uint32))
    t = g.finish(n_voices)
    t["params"] = p
    return t


def adsr_envelope(g, attack=480.0, decay=2400.0, sustain=0.6, release=4800.0, t_end=48000.0, time_slot=0):
    """N5: piecewise-linear ADSR from Minimum/Sum2/Multiply/Divide of t, with
    max(a, b) = -Minimum(-a, -b) (the identity noted at reference src/routing/effect.rs:106-111):
      a = t/A;  d = max(S, 1 - (1-S)*(t-A)/D);  r = S*(T_end - t)/R;  env = max(0, min(min(a, d), r)).
    One node each (about 17 nodes in all); returns the handle of env."""
    f = np.float32
    t = IN(time_slot)
    a = g.binop(K_DIV, t, C(f(attack)), 1)
    ta = g.binop(K_SUM2, t, C(f(-attack)), 1)
    fr = g.binop(K_DIV, ta, C(f(decay)), 1)
    dd = g.binop(K_MUL, C(f(1.0) - f(sustain)), fr, 1)
    d0 = g.binop(K_SUM2, C(f(1.0)), g.binop(K_MUL, C(f(-1.0)), dd, 1), 1)
    d = g.binop(K_MUL, C(f(-1.0)), g.binop(K_MIN, C(f(-sustain)), g.binop(K_MUL, C(f(-1.0)), d0, 1), 1), 1)
    rem = g.binop(K_SUM2, C(f(t_end)), g.binop(K_MUL, C(f(-1.0)), t, 1), 1)
    r = g.binop(K_MUL, C(f(sustain)), g.binop(K_DIV, rem, C(f(release)), 1), 1)
    m2 = g.binop(K_MIN, g.binop(K_MIN, a, d, 1), r, 1)
    return g.binop(K_MUL, C(f(-1.0)), g.binop(K_MIN, C(f(0.0)), g.binop(K_MUL, C(f(-1.0)), m2, 1), 1), 1)


def delay_chain(g, x, taps=4, base_delay=2400.0):
    """N6: `taps` feed-forward taps in series, y = Sum2(x, Multiply(C(g_j), Delay(x, C(d_j)))),
    d_j = base_delay*(j+1) frames, g_j = 0.5^(j+1).  x: handle array; returns the handles after the last tap."""
    x = np.asarray(x, dtype=np.uint32).ravel()
    for j in range(taps):
        dl = g.binop(K_DELAY, x, C(np.float32(base_delay * (j + 1))), len(x))
        x = g.binop(K_SUM2, x, g.binop(K_MUL, C(np.float32(0.5 ** (j + 1))), dl, len(x)), len(x))
    return x


def effects_tree(n_voices, n_partials, seed=0x5EED0003, detune=True, envelope=True, taps=4, base_delay=2400.0,
                 sr=48000.0, time_slot=0, wrap=None, params_as_nodes=False):
    """config D: harmonics + per-partial detune + ADSR envelope + delay chain, one output slot per voice."""
    p = voice_params(n_voices, n_partials, seed, detune, sr, wrap)
    g = GraphArrays()
    wn = harmonic_nodes(g, p) if params_as_nodes else None
    leaves = partial_leaves(g, p["w"], p["amp"], time_slot, wn).reshape(n_voices, n_partials)
    x = sum_tree(g, leaves)
    if envelope:
        env = adsr_envelope(g, time_slot=time_slot)
        x = g.binop(K_MUL, np.broadcast_to(env, x.shape), x, len(x))
    if taps:
        x = delay_chain(g, x, taps, base_delay)
    g.edge(x, 0, 0, np.arange(n_voices, dtype=np.uint32))
    t = g.finish(n_voices)
    t["params"] = p
    return t


def chorus_tree(n_voices, n_partials, seed=0x5EED0004, depth=300.0, base=400.0, rate_hz=0.7, taps=0, base_delay=2400.0,
                sr=48000.0, time_slot=0):
    """Voices through a chorus: y = Sum2(x, Multiply(C(0.5), Delay(x, amount(t)))) with the delay amount a SIGNAL,
    amount = base + depth * lfo(t), lfo = Modulo(t * rate/sr * (v+1), 1) in [0, 1] -- the `Delay` amount evaluated
    at the undelayed t, floor to frames (reference src/render/reference.rs:197-216).  Optionally followed by the
    constant-delay chain of config D."""
    p = voice_params(n_voices, n_partials, seed, False, sr)
    g = GraphArrays()
    f = np.float32
    leaves = partial_leaves(g, p["w"], p["amp"], time_slot).reshape(n_voices, n_partials)
    x = sum_tree(g, leaves)
    rates = (f(rate_hz) / f(sr) * np.arange(1, n_voices + 1, dtype=np.float32)).astype(np.float32)
    lfo = g.binop(K_MOD, g.binop(K_MUL, IN(time_slot), C(rates), n_voices), C(f(1.0)), n_voices)
    amount = g.binop(K_SUM2, C(f(base)), g.binop(K_MUL, C(f(depth)), lfo, n_voices), n_voices)
    wet = g.binop(K_DELAY, x, amount, n_voices)
    x = g.binop(K_SUM2, x, g.binop(K_MUL, C(f(0.5)), wet, n_voices), n_voices)
    if taps:
        x = delay_chain(g, x, taps, base_delay)
    g.edge(x, 0, 0, np.arange(n_voices, dtype=np.uint32))
    t = g.finish(n_voices)
    t["params"] = p
    return t


def partial_effect():
    """The partial oscillator as ONE composite effect `Partial(t, w, amp) -> leaf` (what a `.fnd` effect file would
    hold): inputs 0,1,2 = time, w, amp as signals; per instance the host feeds w and amp through F32Constant edges.
    Lowering inlines the instances and folds the constants, so the engine sees the same graph as the flat form."""
    f = np.float32
    c = 100   # the F32Constant node inside the effect
    nodes = [(c, Effect.primitive("F32Constant")),
             (1, Effect.primitive("Multiply")),   # x = t * w
             (2, Effect.primitive("Modulo")),     # phase = x mod 1
             (3, Effect.primitive("Sum2")),       # u = phase + -0.5
             (4, Effect.primitive("Multiply")),   # nu = -1 * u
             (5, Effect.primitive("Minimum")),    # m = min(u, nu)
             (6, Effect.primitive("Multiply")),   # absu = -1 * m
             (7, Effect.primitive("Multiply")),   # n1 = -1 * absu
             (8, Effect.primitive("Sum2")),       # q = 0.5 + n1
             (9, Effect.primitive("Multiply")),   # p = -16 * u
             (10, Effect.primitive("Multiply")),  # y = p * q
             (11, Effect.primitive("Multiply"))]  # leaf = amp * y
    k = lambda v: f32_bits(f(v))
    edges = [(0, 1, 0, 0), (0, 1, 1, 1),
             (1, 2, 0, 0), (c, 2, k(1.0), 1),
             (2, 3, 0, 0), (c, 3, k(-0.5), 1),
             (c, 4, k(-1.0), 0), (3, 4, 0, 1),
             (3, 5, 0, 0), (4, 5, 0, 1),
             (c, 6, k(-1.0), 0), (5, 6, 0, 1),
             (c, 7, k(-1.0), 0), (6, 7, 0, 1),
             (c, 8, k(0.5), 0), (7, 8, 0, 1),
             (c, 9, k(-16.0), 0), (3, 9, 0, 1),
             (9, 10, 0, 0), (8, 10, 0, 1),
             (0, 11, 2, 0), (10, 11, 0, 1),
             (11, 0, 0, 0)]
    return Effect.graph(nodes, edges)


def install_composite_tree(renderer, n_voices, n_partials, seed=0x5EED0002, detune=False, sr=48000.0):
    """The same V x P additive tree as additive_tree(), built from instances of the composite Partial effect."""
    p = voice_params(n_voices, n_partials, seed, detune, sr)
    eff = partial_effect()
    n = n_voices * n_partials
    g = GraphArrays()
    part = np.arange(g.next, g.next + n, dtype=np.uint32)   # handles of the composite instances
    g.next += n
    g.edge(0, part, 0, 0)                                   # time -> input 0
    g.const(part, p["w"].ravel(), 1)                        # w    -> input 1
    g.const(part, p["amp"].ravel(), 2)                      # amp  -> input 2
    roots = sum_tree(g, part.reshape(n_voices, n_partials))
    g.edge(roots, 0, 0, np.arange(n_voices, dtype=np.uint32))
    tree = g.finish(n_voices)
    renderer.on_add_nodes(part, eff)
    install(renderer, tree)
    return p


_PRIM_EFFECTS = None


def install(renderer, tree):
    """Feeds the graph to a renderer through the batch entry points (nodes grouped by kind, then edges)."""
    global _PRIM_EFFECTS
    if _PRIM_EFFECTS is None:
        from .capi import PRIMITIVES
        _PRIM_EFFECTS = [Effect.primitive(n) for n in PRIMITIVES]
    handles, kinds = tree["handles"], tree["kinds"]
    for k in np.unique(kinds):
        renderer.on_add_nodes(handles[kinds == k], _PRIM_EFFECTS[int(k)])
    renderer.on_add_edges(tree["edges"])
    return renderer


def time_ramp(start, end):
    """The f32 frame ramp fed to the time slot (exact below 2^24 frames)."""
    return np.arange(start, end, dtype=np.float64).astype(np.float32)


def bank_reference_numpy(w, amp, t):
    """numpy restatement of one voice (same op order and association as the graph), for test diagnostics.
    w, amp: [P] f32; t: [T] f32 -> [T] f32.  Not used by any product path."""
    w = np.asarray(w, np.float32)[:, None]
    amp = np.asarray(amp, np.float32)[:, None]
    t = np.asarray(t, np.float32)[None, :]
    x = (t * w).astype(np.float32)
    rem = np.fmod(x, np.float32(1.0)).astype(np.float32)
    ph = np.where(rem < 0, (rem + np.float32(1.0)).astype(np.float32), rem)
    u = (ph + np.float32(-0.5)).astype(np.float32)
    nu = (np.float32(-1.0) * u).astype(np.float32)
    m = np.where((u < nu) | np.isnan(nu), u, nu)
    ab = (np.float32(-1.0) * m).astype(np.float32)
    n1 = (np.float32(-1.0) * ab).astype(np.float32)
    q = (np.float32(0.5) + n1).astype(np.float32)
    pp = (np.float32(-16.0) * u).astype(np.float32)
    y = (pp * q).astype(np.float32)
    cur = (amp * y).astype(np.float32)
    while cur.shape[0] > 1:
        npair = cur.shape[0] // 2
        s = (cur[0:2 * npair:2] + cur[1:2 * npair:2]).astype(np.float32)
        cur = np.concatenate([s, cur[2 * npair:]], axis=0) if cur.shape[0] % 2 else s
    return cur[0]
