"""GPU parity tests proper: the HIP engine (through the C ABI) against the CPU oracle on the same
seeded inputs, bit-exact (NaN == NaN).  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest

import kat_replay
import oracle_tools
import randgraph
from kat_replay import same_bits
from libfriendship_amd import synth
from libfriendship_amd.capi import (FR_ERR_CYCLE, FR_ERR_INPUT_HISTORY, FR_ERR_INPUT_TOO_LONG, FR_ERR_NO_SUCH_NODE, FR_ERR_UNSUPPORTED, Effect,
                                    RenderError, Renderer, f32_bits)

pytestmark = pytest.mark.gpu


def first_diff(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    bad = ~((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))
    idx = np.argwhere(bad)
    if len(idx) == 0:
        return "identical"
    i = tuple(idx[0])
    return f"{bad.sum()} of {bad.size} differ; first at {i}: got {a[i]!r} ({a.view(np.uint32)[i]:#x}) expected {b[i]!r} ({b.view(np.uint32)[i]:#x})"


# ---- the reference's own known-answer tests, through the HIP engine -----------------------------
@pytest.mark.parametrize("mode", ["auto", "pull"])
@pytest.mark.parametrize("i", range(11))
def test_reference_kat_on_hip(hip_lib, kat, i, mode):
    kat_replay.check(hip_lib, kat["tests"][i], mode=mode)


# ---- random graphs of all seven primitives ---------------------------------------------------------
@pytest.mark.parametrize("mode", ["auto", "pull", "staged"])
@pytest.mark.parametrize("i", range(14))
def test_selfcheck_vectors_on_hip(hip_lib, selfcheck, i, mode):
    """The committed self-consistency vectors (tests/golden/selfcheck_vectors.json) through the HIP engine."""
    kat_replay.check(hip_lib, selfcheck["tests"][i], mode=mode)


@pytest.mark.parametrize("mode", ["pull", "auto"])
@pytest.mark.parametrize("seed", range(24))
def test_random_graphs(hip_lib, oracle_lib, seed, mode):
    rng = np.random.default_rng(1000 + seed)
    steps, n_out = randgraph.random_graph(seed, n_nodes=int(rng.integers(4, 40)), n_inputs=2, n_outputs=3)
    T = 96
    with Renderer(hip_lib, mode=mode) as hip, Renderer(oracle_lib) as ref:
        randgraph.install_steps(hip, steps)
        randgraph.install_steps(ref, steps)

        def both(start, end, rows):
            try:
                exp = ref.fill_buffer(n_out, start, end, rows)
            except RenderError as e:
                with pytest.raises(RenderError) as ei:
                    hip.fill_buffer(n_out, start, end, rows)
                assert ei.value.status == e.status
                return False
            got = hip.fill_buffer(n_out, start, end, rows)
            assert same_bits(got, exp), f"seed {seed} [{start},{end}): " + first_diff(got, exp)
            return True

        noise = lambda n: (rng.normal(size=n) * 3).astype(np.float32)
        if not both(0, T, [synth.time_ramp(0, T), noise(T)]):
            return
        both(T, 2 * T, [synth.time_ramp(T, 2 * T), noise(int(rng.integers(0, T)))])   # short row: last-value padding
        both(5000, 5000 + T, [synth.time_ramp(5000, 5000 + T), noise(T)])               # seek
        both(5000 + T, 5000 + T + 7, [synth.time_ramp(5000 + T, 5000 + T + 7)])          # slot 1 not fed


def test_deep_chain_graph(hip_lib, oracle_lib):
    """A 600-deep linear chain (depth stresses the pull stack) with a delay in the middle."""
    N = 600
    with Renderer(hip_lib, mode="pull") as hip, Renderer(oracle_lib) as ref:
        for r in (hip, ref):
            r.on_add_node(1, "F32Constant")
            prev = (0, 0)
            for i in range(N):
                h = i + 2
                r.on_add_node(h, "Delay" if i == N // 2 else ("Sum2" if i % 2 else "Multiply"))
                r.on_add_edge(prev[0], h, prev[1], 0)
                r.on_add_edge(1, h, f32_bits(3.0 if i == N // 2 else (0.001 if i % 2 else 1.0001)), 1)
                prev = (h, 0)
            r.on_add_edge(prev[0], 0, 0, 0)
        t = synth.time_ramp(0, 128)
        assert same_bits(hip.fill_buffer(1, 0, 128, [t]), ref.fill_buffer(1, 0, 128, [t]))


# ---- fused oscillator bank ---------------------------------------------------------------------------
BANK_CASES = [  # V, P, T, idx
    (1, 32, 64, 0), (2, 64, 256, 0), (3, 128, 100, 0), (1, 256, 1000, 48000), (5, 512, 333, 7),
    (2, 1024, 129, 1 << 20), (1, 4096, 64, 0), (64, 32, 4800, 0), (1, 8192, 200, 0), (1, 16384, 70, 123),
    (2, 32768, 65, 0),
]


@pytest.mark.parametrize("V,P,T,idx", BANK_CASES)
def test_bank_bit_exact(hip_lib, oracle_lib, V, P, T, idx):
    tree = synth.additive_tree(V, P, seed=P + V, detune=bool(P % 3))
    t = synth.time_ramp(idx, idx + T)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        got = hip.fill_buffer(V, idx, idx + T, [t])
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and sum(b["voices"] for b in plan["banks"]) == V, plan
        if P * T * V <= 1 << 21:
            exp = ref.fill_buffer(V, idx, idx + T, [t])
            assert same_bits(got, exp), first_diff(got, exp)
        else:   # the oracle is random-access in time (a seek per sampled frame; the graph has no Delay)
            rng = np.random.default_rng(P)
            cols = np.unique(np.concatenate([[0, T - 1], rng.integers(0, T, 6)]))
            for c in cols:
                exp = ref.fill_buffer(V, idx + int(c), idx + int(c) + 1, [t[c:c + 1]])
                assert same_bits(got[:, c:c + 1], exp), f"frame {c}: " + first_diff(got[:, c:c + 1], exp)


def test_bank_matches_pull_interpreter(hip_lib):
    """Fused kernel vs the generic interpreter on the same device (sizes the oracle would take minutes for)."""
    tree = synth.additive_tree(4, 256, seed=3, detune=True)
    t = synth.time_ramp(0, 1536)
    with Renderer(hip_lib, mode="auto") as a, Renderer(hip_lib, mode="pull") as b:
        synth.install(a, tree)
        synth.install(b, tree)
        assert same_bits(a.fill_buffer(4, 0, 1536, [t]), b.fill_buffer(4, 0, 1536, [t]))


def test_bank_unusual_time_inputs(hip_lib, oracle_lib):
    """The time slot is just an input signal: negative, fractional, huge, NaN/inf, short (padded) rows."""
    tree = synth.additive_tree(2, 64, seed=11)
    rng = np.random.default_rng(5)
    rows = [
        (-synth.time_ramp(0, 200)),                                   # negative times: slow (general) path
        (rng.normal(size=200) * 1000).astype(np.float32),             # mixed sign, fractional
        np.array([0, 1, 2, np.nan, np.inf, -np.inf, 1e30, 3e38, 1e-30, -0.0] * 20, dtype=np.float32),
        synth.time_ramp(0, 50),                                       # short row: padded with 49.0
        np.zeros(0, np.float32),                                      # empty row: padded with last value
    ]
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for i, row in enumerate(rows):
            got = hip.fill_buffer(2, i * 200, (i + 1) * 200, [row])
            exp = ref.fill_buffer(2, i * 200, (i + 1) * 200, [row])
            assert same_bits(got, exp), f"row {i}: " + first_diff(got, exp)
        # no time row at all after a seek: time reads as 0
        assert same_bits(hip.fill_buffer(2, 0, 100), ref.fill_buffer(2, 0, 100))
        assert hip.plan()["banks"]


def test_silent_and_degenerate_voices(hip_lib, oracle_lib):
    """Voices whose mix is identically zero -- every amplitude +0 or -0 (a silent voice), or every t*w beyond 2^23
    cycles -- have leaves that are all +-0; the sign of the exact-zero sum follows the graph (-0 iff every leaf is
    -0).  Tiles full of zeros take the all-frames pass of the bank kernel; mixed voices take the per-frame one."""
    V, P, T = 6, 256, 300
    p = synth.voice_params(V, P, seed=1)
    w, amp = p["w"].copy(), p["amp"].copy()
    amp[0, :] = 0.0                      # silent: +0 amplitudes
    amp[1, :] = -0.0                     # silent: -0 amplitudes
    amp[2, ::2] = 0.0                    # half the partials silent: a normal non-zero mix
    w[3, :] = (2.0 ** 24) * np.arange(1, P + 1, dtype=np.float32)   # every phase an integer: all leaves zero
    amp[4, :] = 0.0
    amp[4, 7] = -0.0                     # silent with one -0 amplitude among +0
    g = synth.GraphArrays()
    leaves = synth.partial_leaves(g, w, amp).reshape(V, P)
    g.edge(synth.sum_tree(g, leaves), 0, 0, np.arange(V, dtype=np.uint32))
    tree = g.finish(V)
    rng = np.random.default_rng(2)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        rows = [synth.time_ramp(0, T), -synth.time_ramp(0, T), (rng.normal(size=T) * 100).astype(np.float32), synth.time_ramp(5000, 5000 + T)]
        for i, row in enumerate(rows):
            got, exp = hip.fill_buffer(V, i * T, (i + 1) * T, [row]), ref.fill_buffer(V, i * T, (i + 1) * T, [row])
            assert same_bits(got, exp), f"row {i}: " + first_diff(got, exp)
            assert not got[0].any() and not got[1].any() and got[2].any()
        assert hip.plan()["banks"] and not hip.plan()["pull_rows"]
    # the same for voices that are not balanced trees (schedule kernel): 250 and 1000 partials
    for P2 in (250, 1000):
        p = synth.voice_params(4, P2, seed=3)
        w, amp = p["w"].copy(), p["amp"].copy()
        amp[0, :] = 0.0
        amp[1, :] = -0.0
        amp[2, 1::3] = 0.0
        g = synth.GraphArrays()
        leaves = synth.partial_leaves(g, w, amp).reshape(4, P2)
        g.edge(synth.sum_tree(g, leaves), 0, 0, np.arange(4, dtype=np.uint32))
        tree = g.finish(4)
        with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
            synth.install(hip, tree)
            synth.install(ref, tree)
            for i, row in enumerate([synth.time_ramp(0, 130), -synth.time_ramp(0, 130), synth.time_ramp(7000, 7130)]):
                got, exp = hip.fill_buffer(4, i * 130, (i + 1) * 130, [row]), ref.fill_buffer(4, i * 130, (i + 1) * 130, [row])
                assert same_bits(got, exp), f"P={P2} row {i}: " + first_diff(got, exp)
                assert not got[0].any() and not got[1].any() and got[3].any()
            assert any(b["general_tree"] for b in hip.plan()["banks"])


def test_many_small_voices_kernel(hip_lib, oracle_lib, monkeypatch):
    """Enough small voices (<= 256 partials) take bank_multi_kernel: whole voices per wave, several in a row, no LDS.
    Same bits as the quarter-voice kernel (FR_BANK_MULTI=0) on the full output -- silent voices, hostile times and a
    ring-fed delay included -- and as the oracle on sampled frames."""
    for V, P, T in ((1024, 32, 1100), (600, 128, 700), (260, 256, 1030)):
        p = synth.voice_params(V, P, seed=9)
        w, amp = p["w"].copy(), p["amp"].copy()
        w = (w * (1.0 + 1e-4 * (np.arange(V) // 64))[:, None]).astype(np.float32)
        w[:, :] = np.tile(synth.voice_params(64, P, seed=9)["w"], (V // 64 + 1, 1))[:V] * (1.0 + 1e-4 * (np.arange(V) // 64))[:, None].astype(np.float32)
        amp[5, :] = 0.0                   # silent voices: the all-frames zero-sign pass inside the wave
        amp[6, :] = -0.0
        amp[7, 3] = 0.0
        g = synth.GraphArrays()
        leaves = synth.partial_leaves(g, w.astype(np.float32), amp).reshape(V, P)
        roots = synth.sum_tree(g, leaves)
        g.edge(roots, 0, 0, np.arange(V, dtype=np.uint32))
        d = g.binop(synth.K_SUM2, roots[0:1], g.binop(synth.K_DELAY, roots[1:2], synth.C(np.float32(11.0)), 1), 1)
        g.edge(d, 0, 0, V)                # one more row: voice 0 + voice 1 delayed (voice 1 also fills a ring)
        tree = g.finish(V + 1)
        rng = np.random.default_rng(V)
        rows = [synth.time_ramp(0, T), synth.time_ramp(T, 2 * T),
                np.concatenate([-synth.time_ramp(0, 64), [np.nan, np.inf, -0.0, 1e30], synth.time_ramp(0, T - 68)]).astype(np.float32)]
        outs = {}
        for multi in ("1", "0"):
            monkeypatch.setenv("FR_BANK_MULTI", multi)
            with Renderer(hip_lib) as hip:
                synth.install(hip, tree)
                outs[multi] = [hip.fill_buffer(V + 1, i * T, (i + 1) * T, [row]) for i, row in enumerate(rows)]
        for i in range(len(rows)):
            assert same_bits(outs["1"][i], outs["0"][i]), f"V={V} P={P} row {i}: " + first_diff(outs["1"][i], outs["0"][i])
        with Renderer(oracle_lib) as ref:
            synth.install(ref, tree)
            exp = ref.fill_buffer(V + 1, 0, 70, [rows[0][:70]])          # contiguous from 0: the delayed row too
            assert same_bits(outs["1"][0][:, :70], exp), f"V={V} P={P} first frames: " + first_diff(outs["1"][0][:, :70], exp)
            for c in (T - 1, int(rng.integers(70, T))):                    # later frames by seeking: rows without a Delay only
                exp = ref.fill_buffer(V + 1, c, c + 1, [rows[0][c:c + 1]])
                assert same_bits(outs["1"][0][:V, c:c + 1], exp[:V]), f"V={V} P={P} frame {c}: " + first_diff(outs["1"][0][:V, c:c + 1], exp[:V])


def test_many_small_general_voices_kernel(hip_lib, oracle_lib, monkeypatch):
    """The same for voices that are not balanced trees (24, 100, 300 partials): gbank_multi_kernel vs the one-voice-per-
    workgroup schedule kernel (FR_BANK_MULTI=0) on the full output, and vs the oracle on the first frames."""
    for V, P, T in ((1400, 24, 900), (700, 100, 700), (300, 300, 1100)):
        base = synth.voice_params(64, P, seed=4)
        reps = V // 64 + 1
        w = (np.tile(base["w"], (reps, 1))[:V] * (1.0 + 1e-4 * (np.arange(V) // 64))[:, None]).astype(np.float32)
        amp = np.tile(base["amp"], (reps, 1))[:V].copy()
        amp[3, :] = 0.0
        amp[4, :] = -0.0
        amp[5, 1::2] = 0.0
        g = synth.GraphArrays()
        leaves = synth.partial_leaves(g, w, amp).reshape(V, P)
        g.edge(synth.sum_tree(g, leaves), 0, 0, np.arange(V, dtype=np.uint32))
        tree = g.finish(V)
        rows = [synth.time_ramp(0, T), np.concatenate([-synth.time_ramp(0, 64), [np.nan, np.inf, -0.0, 1e30], synth.time_ramp(0, T - 68)]).astype(np.float32)]
        outs = {}
        for multi in ("1", "0"):
            monkeypatch.setenv("FR_BANK_MULTI", multi)
            with Renderer(hip_lib) as hip:
                synth.install(hip, tree)
                outs[multi] = [hip.fill_buffer(V, i * T, (i + 1) * T, [row]) for i, row in enumerate(rows)]
                assert any(b["general_tree"] for b in hip.plan()["banks"])
        for i in range(len(rows)):
            assert same_bits(outs["1"][i], outs["0"][i]), f"V={V} P={P} row {i}: " + first_diff(outs["1"][i], outs["0"][i])
        with Renderer(oracle_lib) as ref:
            synth.install(ref, tree)
            exp = ref.fill_buffer(V, 0, 66, [rows[0][:66]])
            assert same_bits(outs["1"][0][:, :66], exp), f"V={V} P={P}: " + first_diff(outs["1"][0][:, :66], exp)


def test_bank_negative_frequency_and_mixed_outputs(hip_lib, oracle_lib):
    """A voice with negative w (general fract path), next to outputs that are not banks: the bank delayed by 5
    frames (staged: the bank fills a ring, two small programs read it) and the time input itself."""
    g = synth.GraphArrays()
    w = np.linspace(-0.01, 0.02, 64).astype(np.float32)
    amp = np.linspace(1.0, 0.1, 64).astype(np.float32)
    leaves = synth.partial_leaves(g, w, amp).reshape(1, 64)
    root = synth.sum_tree(g, leaves)
    g.edge(root, 0, 0, 0)                       # out0: a bank
    d = g.binop(synth.K_DELAY, root, synth.C(np.float32(5.0)), 1)
    g.edge(d, 0, 0, 1)                          # out1: the bank delayed by 5 (not a bank: pull)
    g.edge(0, 0, 0, 2)                          # out2: the time input itself
    tree = g.finish(3)
    t = synth.time_ramp(0, 300)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        got, exp = hip.fill_buffer(3, 0, 300, [t]), ref.fill_buffer(3, 0, 300, [t])
        assert same_bits(got, exp), first_diff(got, exp)
        plan = hip.plan()
        assert len(plan["banks"]) == 1 and not plan["banks"][0]["fast_ok"] and plan["banks"][0]["to_ring"], plan
        assert plan["pull_rows"] == 0 and plan["rings"] == 1 and plan["stage_programs"] == 3, plan


def test_chunked_calls_equal_one_call(hip_lib):
    """Size-independent property at BASELINE config C's full shape: rendering 4800 frames as one call,
    as 10 x 480 and as ragged chunks gives identical bits (the evaluator is a pure function of t)."""
    V, P, T = 64, 4096, 4800
    tree = synth.additive_tree(V, P)
    t = synth.time_ramp(0, T)
    with Renderer(hip_lib) as a, Renderer(hip_lib) as b, Renderer(hip_lib) as c:
        for r in (a, b, c):
            synth.install(r, tree)
        whole = a.fill_buffer(V, 0, T, [t])
        parts = np.concatenate([b.fill_buffer(V, s, s + 480, [t[s:s + 480]]) for s in range(0, T, 480)], axis=1)
        cuts = [0, 1, 65, 700, 701, 2049, 4799, 4800]
        ragged = np.concatenate([c.fill_buffer(V, s, e, [t[s:e]]) for s, e in zip(cuts[:-1], cuts[1:])], axis=1)
        assert same_bits(whole, parts) and same_bits(whole, ragged)
        assert a.plan()["banks"][0]["partials"] == P


def _oracle_with_history(ref, tree, rows_by_call):
    """The oracle holding `tree` and the input history of the given calls WITHOUT having rendered anything: the graph goes in
    without its output edges (an unconnected slot is 0.0, reference.rs:164), the calls store their rows, then the output
    edges are added.  The evaluator is a pure function of (graph, history, t) (reference.rs:90-96,178-266), so
    oracle_tools.eval_samples then answers any (slot, frame) of the full graph -- random access instead of minutes of CPU."""
    e = tree["edges"]
    synth.install(ref, dict(tree, edges=e[e[:, 1] != 0]))
    for start, row in rows_by_call:
        assert not ref.fill_buffer(1, start, start + len(row), [row]).any()
    ref.on_add_edges(e[e[:, 1] == 0])


def _sampled_parity(ref, got_by_call, slots, frames, what):
    """got_by_call: [(first frame, [V, T] array)]; compares every (slot, frame) of slots x frames bit for bit."""
    slots, frames = np.asarray(slots, np.uint32), np.asarray(frames, np.uint64)
    exp = oracle_tools.eval_samples(ref, np.repeat(slots, len(frames)), np.tile(frames, len(slots))).reshape(len(slots), len(frames))
    got = np.empty_like(exp)
    for j, f in enumerate(frames):
        start, arr = next((st, a) for st, a in got_by_call if st <= f < st + a.shape[1])
        got[:, j] = arr[slots, int(f) - start]
    assert same_bits(got, exp), what + ": " + first_diff(got, exp)
    return got


def test_config_c_full_size_sampled_against_oracle(hip_lib, oracle_lib):
    """BASELINE config C (4096 partials x 64 voices) at full size: EVERY voice at 64 random frames of two consecutive
    4800-frame calls plus the first / last / call-boundary frames, against the real oracle (random access,
    reference.rs:90-96) on the same 3.1 M-node primitive graph."""
    V, P, T = 64, 4096, 4800
    tree = synth.additive_tree(V, P)
    rows = [(0, synth.time_ramp(0, T)), (T, synth.time_ramp(T, 2 * T))]
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        got = [(st, hip.fill_buffer(V, st, st + T, [row])) for st, row in rows]
        _oracle_with_history(ref, tree, rows)
        rng = np.random.default_rng(0)
        frames = np.unique(np.concatenate([[0, 1, 63, 64, T - 1, T, T + 1, 2 * T - 1], rng.integers(0, 2 * T, 64)]))
        _sampled_parity(ref, got, np.arange(V), frames, "config C")


def test_config_e_full_size_sampled_against_oracle(hip_lib, oracle_lib):
    """BASELINE configs[4] in its named size, 16384 partials x 256 voices (4.2 M partials, a 50 M-node primitive graph),
    on one GPU: 32 voices x 32 frames against the oracle.  Voices are independent (reference.rs:78-82), so the oracle gets
    the sub-tree of just those voices.  With the survey's f0 = 55 * 2^(v/12) every voice above v ~ 150 is identically
    zero (every t*w >= 2^23): those take the kernel's zero-sign path for every frame, so a dozen of them are sampled
    too, signs of zero included."""
    V, P, T = 256, 16384, 4800
    picks = [0, 1, 7, 19, 37, 52, 64, 77, 90, 101, 113, 126, 133, 140, 145, 148, 149, 150, 151, 152, 155, 160, 171, 183, 196, 200, 214, 229, 240, 247, 254, 255]
    tree = synth.additive_tree(V, P)
    sub = synth.additive_tree(V, P, voices=picks)
    t = synth.time_ramp(0, T)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        del tree
        got = hip.fill_buffer(V, 0, T, [t])
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and [(b["voices"], b["partials"]) for b in plan["banks"]] == [(V, P)], plan
        _oracle_with_history(ref, sub, [(0, t)])
        rng = np.random.default_rng(0)
        frames = np.unique(np.concatenate([[0, 1, 63, 64, 65, T - 1], rng.integers(0, T, 27)]))[:32]
        assert len(picks) == 32 and len(frames) >= 30
        _sampled_parity(ref, [(0, got[picks])], np.arange(len(picks)), frames, "config E")
        assert np.abs(got[:100]).max() > 0.1 and (got == 0).any()


@pytest.mark.parametrize("V,P,detune", [(4, 1024, True), (64, 256, False)])
def test_harmonics_and_detune_as_graph_nodes(hip_lib, oracle_lib, V, P, detune):
    """N3 / N4: harmonics f0*(k+1), detune *(1+delta) and /sr arrive as Multiply / Divide nodes over
    constants -- the reference has only the seven primitives.  The engine folds them at lowering with exactly-rounded
    f32 ops, still recognises every voice as a bank, and equals the oracle evaluating the same node graph."""
    T = 96
    tree = synth.additive_tree(V, P, seed=21, detune=detune, params_as_nodes=True)
    t = synth.time_ramp(4800, 4800 + T)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        got = hip.fill_buffer(V, 4800, 4800 + T, [t])
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and plan["stage_programs"] == 0, plan
        assert [(b["voices"], b["partials"]) for b in plan["banks"]] == [(V, P)], plan
        if V * P * T <= 1 << 19:
            exp = ref.fill_buffer(V, 4800, 4800 + T, [t])
            assert same_bits(got, exp), first_diff(got, exp)
        else:
            for c in (0, 17, T - 1):
                exp = ref.fill_buffer(V, 4800 + c, 4800 + c + 1, [t[c:c + 1]])
                assert same_bits(got[:, c:c + 1], exp), f"frame {c}: " + first_diff(got[:, c:c + 1], exp)


def test_config_d_nodes_form_keeps_its_plan(hip_lib):
    """Config D with harmonics + detune as nodes plans exactly like the numpy-folded form and renders the same bits."""
    V, P, T = 16, 256, 600
    a_tree = synth.effects_tree(V, P, params_as_nodes=True, base_delay=100.0)
    b_tree = synth.effects_tree(V, P, base_delay=100.0)
    t = synth.time_ramp(0, T)
    with Renderer(hip_lib) as a, Renderer(hip_lib) as b:
        synth.install(a, a_tree)
        synth.install(b, b_tree)
        assert same_bits(a.fill_buffer(V, 0, T, [t]), b.fill_buffer(V, 0, T, [t]))
        pa, pb = a.plan(), b.plan()
        for key in ("banks", "stage_programs", "rings", "max_lookback", "pull_rows", "fused_programs"):
            assert pa[key] == pb[key], (key, pa[key], pb[key])


def test_shared_root_on_two_rows_is_one_launch_per_call(hip_lib, oracle_lib):
    """a non-bank root on two output rows with no delayed read: the fused steady-state form must be ONE
    launch per contiguous call whatever idx is (the sub-window count used to wrap: idx + 1 launches)."""
    with Renderer(hip_lib) as r, Renderer(oracle_lib) as ref:
        for x in (r, ref):
            x.on_add_node(1, "F32Constant")
            x.on_add_node(2, "Multiply")
            x.on_add_node(3, "Sum2")
            x.on_add_edge(0, 2, 0, 0)
            x.on_add_edge(1, 2, f32_bits(0.5), 1)
            x.on_add_edge(2, 3, 0, 0)
            x.on_add_edge(0, 3, 1, 1)
            x.on_add_edge(3, 0, 0, 0)
            x.on_add_edge(3, 0, 0, 1)
        T = 4800
        rng = np.random.default_rng(0)
        r.set_timing(True)
        for k in range(4):
            rows = [synth.time_ramp(k * T, (k + 1) * T), rng.normal(size=T).astype(np.float32)]
            r.reset_timing()
            got = r.fill_buffer(2, k * T, (k + 1) * T, rows)
            _, n = r.get_timing("stage")
            assert same_bits(got, ref.fill_buffer(2, k * T, (k + 1) * T, rows))
            if k > 0:
                assert n == 1, f"call {k} at idx {k * T}: {n} stage launches"


def test_failed_call_leaves_the_input_store_intact(hip_lib, oracle_lib):
    """ADVICE r1 (medium), on the device: see tests/test_sim_engine.py for the host-logic form."""
    import test_sim_engine
    test_sim_engine.test_failed_call_leaves_the_input_store_intact(hip_lib, oracle_lib)


def test_host_entry_point_two_interleaved_banks(hip_lib, oracle_lib):
    """The host entry point reading its input row through mapped pinned memory, on the device."""
    import test_sim_engine
    test_sim_engine.test_host_entry_point_two_interleaved_banks(hip_lib, oracle_lib)


@pytest.mark.parametrize("V,P,T,block", [(64, 4096, 4800, 64), (64, 4096, 2048, 512), (3, 65536, 640, 64), (16, 512, 1280, 128),
                                         (640, 512, 256, 64), (5, 2048, 700, 100), (1, 1 << 20, 192, 64)])
def test_short_call_kernel_equals_block_render(hip_lib, V, P, T, block):
    """The short-call kernel (voices cut into chunks over workgroups, parameters staged through LDS, chunk sums combined by
    the last workgroup to arrive) against the time-major kernel: the same frames rendered as one long call and as many
    short calls must be the same bits -- every frame of every voice, several passes (the in-launch combine is a
    cross-workgroup hand-off: a stale read would show as a wrong sample somewhere)."""
    tree = synth.additive_tree(V, P, seed=V + P, detune=True)
    t = synth.time_ramp(0, T)
    with Renderer(hip_lib) as a, Renderer(hip_lib) as b:
        synth.install(a, tree)
        synth.install(b, tree)
        whole = a.fill_buffer(V, 0, T, [t])
        for rep in range(6):
            base = rep * T   # contiguous calls, the same time VALUES every pass
            parts = np.concatenate([b.fill_buffer(V, base + s, base + min(s + block, T), [t[s:min(s + block, T)]])
                                    for s in range(0, T, block)], axis=1)
            assert same_bits(whole, parts), f"pass {rep}: " + first_diff(parts, whole)


@pytest.mark.parametrize("V,P,T,block", [(8, 4096, 4800, 1024), (16, 4096, 4800, 640), (5, 2048, 4777, 1000), (3, 16384, 6400, 704),
                                         (32, 4096, 4800, 448), (9, 8192, 2000, 512),
                                         (300, 512, 1000, 192), (70, 512, 4000, 320)])   # (>= 4096 workgroups of small voices: one wave each)
def test_few_voice_launch_equals_block_render(hip_lib, V, P, T, block):
    """A GPU's share of a voice-sharded job -- few voices, a long call: a few hundred to a few thousand (voice, tile) pairs,
    8 x 4096 x 4800 being one GPU's step of config C on 8 GPUs -- takes the short-call kernel with chunks + tickets or the
    time-major kernel (bank_shape).  The same frames rendered in blocks take other shapes of those kernels: every frame of
    every voice must be the same bits, several passes over the same tickets."""
    tree = synth.additive_tree(V, P, seed=V + P, detune=True)
    t = synth.time_ramp(0, T)
    with Renderer(hip_lib) as a, Renderer(hip_lib) as b:
        synth.install(a, tree)
        synth.install(b, tree)
        parts = np.concatenate([b.fill_buffer(V, s, min(s + block, T), [t[s:min(s + block, T)]]) for s in range(0, T, block)], axis=1)
        for rep in range(5):
            whole = a.fill_buffer(V, rep * T, (rep + 1) * T, [t])     # contiguous calls, the same time VALUES every pass
            assert same_bits(whole, parts), f"pass {rep}: " + first_diff(whole, parts)


def test_few_voice_launch_against_oracle(hip_lib, oracle_lib):
    """A GPU's share of config C on 8 GPUs (8 voices x 4096 partials x 4800 frames) against the oracle: sampled frames of
    every voice incl. t = 0 (exact zeros whose sign the chunks settle), a silent voice and one with -0 amplitudes (every
    chunk sum a zero), then hostile time rows (negative, fractional, huge, NaN: the general fract path), and the input
    history the launch appended (row 0 reads it back through a Delay)."""
    V, P, T = 8, 4096, 4800
    p = synth.voice_params(V, P, seed=11, detune=True)
    w, amp = p["w"].copy(), p["amp"].copy()
    amp[2, :] = 0.0
    amp[5, :] = -0.0
    amp[6, 1::2] = 0.0
    g = synth.GraphArrays()
    roots = synth.sum_tree(g, synth.partial_leaves(g, w, amp).reshape(V, P))
    d = g.binop(synth.K_DELAY, synth.IN(0), synth.C(np.float32(3.0)), 1)      # row 0: the time input 3 frames ago (reads the appended history)
    g.edge(d, 0, 0, 0)
    g.edge(roots, 0, 0, np.arange(1, V + 1, dtype=np.uint32))                  # rows 1..V: the voices
    tree = g.finish(V + 1)
    rng = np.random.default_rng(5)
    hostile = (rng.normal(size=T) * 1000).astype(np.float32)
    hostile[::97] = np.nan
    hostile[5::211] = np.float32(2.0 ** 33)
    hostile[7::301] = -0.0
    rows = [synth.time_ramp(0, T), synth.time_ramp(T, 2 * T), hostile, synth.time_ramp(3 * T, 4 * T)]
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for i, row in enumerate(rows):
            got = hip.fill_buffer(V + 1, i * T, (i + 1) * T, [row])
            ref.fill_buffer(1, i * T, (i + 1) * T, [row])                    # (the oracle stores the row and renders only the cheap slot)
            cols = np.unique(np.concatenate([[0, 1, 2, 3, 63, 64, 2047, T - 1], rng.integers(0, T, 24), np.arange(0, T, 97)[:8]]))
            slots = np.repeat(np.arange(V + 1), len(cols)).astype(np.uint32)
            times = np.tile(cols + i * T, V + 1).astype(np.uint64)
            exp = oracle_tools.eval_samples(ref, slots, times).reshape(V + 1, len(cols))
            assert same_bits(got[:, cols], exp), f"row {i}: " + first_diff(got[:, cols], exp)
            assert i == 2 or (not got[3].any() and not got[6].any() and got[7].any())   # (a NaN time makes every voice NaN)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and [(b["voices"], b["partials"]) for b in plan["banks"]] == [(V, P)], plan


def test_short_call_kernel_against_oracle(hip_lib, oracle_lib):
    """Short calls of a chunked voice against the oracle, t = 0 (an exact zero whose sign the chunks must settle), negative
    and fractional times (the general fract path) included."""
    V, P = 4, 2048
    tree = synth.additive_tree(V, P, seed=8, detune=True)
    rng = np.random.default_rng(3)
    rows = [synth.time_ramp(0, 64), synth.time_ramp(64, 100), -synth.time_ramp(0, 50), (rng.normal(size=70) * 100).astype(np.float32),
            np.zeros(33, np.float32)]
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        idx = 0
        for row in rows:
            got = hip.fill_buffer(V, idx, idx + len(row), [row])
            exp = ref.fill_buffer(V, idx, idx + len(row), [row])
            assert same_bits(got, exp), first_diff(got, exp)
            idx += len(row)


def test_streamed_host_output(hip_lib, oracle_lib, monkeypatch):
    """fr_fill_buffer streaming rows out under the running launch (row-completion flags in mapped memory), on the device;
    tools/host_stream_soak.py is the long form."""
    import test_sim_engine
    test_sim_engine.test_streamed_host_output(hip_lib, oracle_lib, monkeypatch)


def test_registered_destination_is_written_directly(hip_lib, oracle_lib):
    """fr_host_register on the device: kernels store straight into the caller's page-locked buffer."""
    import test_sim_engine
    test_sim_engine.test_registered_destination_is_written_directly(hip_lib, oracle_lib)


def test_bounded_input_history(hip_lib, oracle_lib):
    """fr_config.history_frames on the device (the host-logic form is tests/test_sim_engine.py)."""
    import test_sim_engine
    test_sim_engine.test_bounded_input_history(hip_lib, oracle_lib)


def test_bounded_history_with_banks_and_device_rows(hip_lib, oracle_lib):
    """Sliding history + the bank kernel's own history append + device-resident rows, on the device."""
    import test_sim_engine
    test_sim_engine.test_bounded_history_with_banks_and_device_rows(hip_lib, oracle_lib)


def test_ten_million_frames_in_bounded_memory(hip_lib):
    """10^7 frames (3.5 minutes of audio) through the device entry point with history_frames set: device memory in use
    stops growing after the first calls; with the reference's unbounded history it grows by 4 bytes per frame and slot."""
    import torch
    T = 50000

    def run(history_frames):
        with Renderer(hip_lib, history_frames=history_frames) as r:
            r.on_add_node(1, "F32Constant")
            r.on_add_node(2, "Delay")
            r.on_add_edge(0, 2, 0, 0)
            r.on_add_edge(1, 2, f32_bits(100.0), 1)
            r.on_add_edge(2, 0, 0, 0)
            d_row = torch.arange(T, dtype=torch.float32, device="cuda")
            d_out = torch.empty((1, T), dtype=torch.float32, device="cuda")
            stream = torch.cuda.current_stream().cuda_stream
            free_at = {}
            for k in range(200):
                r.fill_buffer_device(d_out.data_ptr(), 1, T, k * T, d_row.data_ptr(), [0, T], stream)
                if k in (10, 199):
                    torch.cuda.synchronize()
                    free_at[k] = torch.cuda.mem_get_info()[0]
            out = d_out.cpu().numpy()
            assert out[0, 100] == 0.0 and out[0, 99] == T - 1 and out[0, 150] == 50.0
            return free_at[10] - free_at[199]

    assert run(48000) < (4 << 20)          # bounded: nothing more after the first calls
    assert run(0) > (30 << 20)             # the reference's behaviour: ~38 MB more for 9.5 M further frames


@pytest.mark.parametrize("mode", ["auto", "pull", "staged"])
def test_sparkle_semantics(hip_lib, oracle_lib, mode):
    """FR_SEMANTICS_SPARKLE (Minimum as select-ult, Delay amount < 0 / NaN -> 0.0) on the device, all three evaluators."""
    import test_sim_engine
    test_sim_engine._sparkle_cases(hip_lib, oracle_lib, mode)


@pytest.mark.parametrize("seed", range(8))
def test_random_graphs_sparkle_semantics(hip_lib, oracle_lib, seed, monkeypatch):
    import test_sim_engine
    if seed % 2:
        monkeypatch.setenv("FR_STAGE_JIT", "force")     # the compiled stage programs carry the semantics too
    test_sim_engine.test_random_graphs_sparkle_semantics(hip_lib, oracle_lib, seed)


def test_rccl_is_loadable_and_hands_out_an_id(hip_lib):
    """fr_comm_unique_id = ncclGetUniqueId through the engine's lazily loaded RCCL (one rank cannot exercise more)."""
    a, b = hip_lib.comm_unique_id(), hip_lib.comm_unique_id()
    assert len(a) == 128 and a != b and any(a)


def test_rccl_exchange_with_itself(hip_lib):
    """fr_comm_selftest: a communicator of one rank, ncclSend + ncclRecv to itself in one group on a stream, through the very
    Transport::sendrecv the partial-block exchange uses -- as much of the RCCL path as one GPU can run (every size class:
    a few floats, a frame block, a 4800-frame x 256-voice slab)."""
    for n in (7, 4800, 4800 * 256):
        hip_lib.comm_selftest(n)


# ---- boundary behaviour on the HIP engine ---------------------------------------------------------------
def test_hip_error_codes(hip_lib):
    with Renderer(hip_lib) as r:
        r.on_add_edge(0, 0, 0, 0)
        with pytest.raises(RenderError) as ei:
            r.fill_buffer(1, 0, 4, [[1, 2, 3, 4, 5]])
        assert ei.value.status == FR_ERR_INPUT_TOO_LONG
        assert r.fill_buffer(1, 0, 4, [[1, 2, 3, 4]]).tolist() == [[1, 2, 3, 4]]   # refused call left state intact
    with Renderer(hip_lib) as r:
        r.on_add_edge(0, 0, 0, 0)
        r.on_add_edge(0, 0, 1, 1)
        r.fill_buffer(2, 0, 4, [[1, 2, 3, 4]])
        with pytest.raises(RenderError) as ei:
            r.fill_buffer(2, 4, 8, [[1, 2, 3, 4], [9, 9, 9, 9]])
        assert ei.value.status == FR_ERR_INPUT_HISTORY
    with Renderer(hip_lib) as r:
        with pytest.raises(RenderError) as ei:
            r.on_add_edge(0, 42, 0, 0)
        assert ei.value.status == FR_ERR_NO_SUCH_NODE
    with Renderer(hip_lib) as r:   # a cycle the reference's RouteGraph would have rejected
        r.on_add_node(1, "Sum2")
        r.on_add_edge(1, 1, 0, 0)
        r.on_add_edge(1, 0, 0, 0)
        with pytest.raises(RenderError) as ei:
            r.fill_buffer(1, 0, 4)
        assert ei.value.status == FR_ERR_CYCLE
    with Renderer(hip_lib) as r:   # garbage slot indices are refused, not allocated (no crash, state intact)
        r.on_add_node(1, "Sum2")
        for to in (1, 0):
            with pytest.raises(RenderError) as ei:
                r.on_add_edge(0, to, 0, 0xFFFFFFF0)
            assert ei.value.status == 10   # FR_ERR_UNSUPPORTED
        r.on_add_edge(0, 1, 0, 0)
        r.on_add_edge(1, 0, 0, 0)
        assert r.fill_buffer(1, 0, 3, [[1, 2, 3]]).tolist() == [[1, 2, 3]]


def test_hip_rows_beyond_storage_are_dropped(hip_lib):
    with Renderer(hip_lib) as r:
        r.on_add_edge(0, 0, 1, 0)
        assert r.fill_buffer(1, 0, 1, [[5.0], [7.0]]).tolist() == [[0.0]]
    with Renderer(hip_lib) as r:
        r.on_add_edge(0, 0, 1, 0)
        assert r.fill_buffer(1, 0, 2, [[5.0, 5.0], [7.0, 8.0]]).tolist() == [[7.0, 8.0]]


def test_device_resident_entry_point(hip_lib, oracle_lib):
    """fr_fill_buffer_device: inputs and outputs already in HBM (what bench.py times)."""
    import torch
    tree = synth.additive_tree(3, 128, seed=2)
    T = 700
    t = synth.time_ramp(0, 2 * T)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        d_t = torch.from_numpy(t).cuda()
        d_out = torch.empty((3, T), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for k in range(2):
            row = d_t[k * T:(k + 1) * T] if k == 0 else d_t[k * T:k * T + 100]   # second call: short row, padded on device
            hip.fill_buffer_device(d_out.data_ptr(), 3, T, k * T, row.data_ptr(), [0, row.numel()], s)
            torch.cuda.synchronize()
            exp = ref.fill_buffer(3, k * T, (k + 1) * T, [row.cpu().numpy()])
            assert same_bits(d_out.cpu().numpy(), exp), first_diff(d_out.cpu().numpy(), exp)


def test_calls_on_different_streams_are_ordered(hip_lib, oracle_lib):
    """Device-entry calls are asynchronous; consecutive calls issued on DIFFERENT streams (and a host-buffer call in
    between, which uses the renderer's own stream) still see each other's input history and delay rings: the engine
    chains them with an event.  No synchronisation by the caller until the end."""
    import torch
    V, T, calls = 4, 1024, 6
    tree = synth.effects_tree(V, 64, taps=3, base_delay=300.0)
    t = synth.time_ramp(0, calls * T)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        d_t = torch.from_numpy(t).cuda()
        streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.empty((V, T), dtype=torch.float32, device="cuda") for _ in range(calls)]
        host_out = {}
        torch.cuda.synchronize()
        for k in range(calls):
            if k == 3:   # a host-buffer call in the middle of the asynchronous ones
                host_out[k] = hip.fill_buffer(V, k * T, (k + 1) * T, [t[k * T:(k + 1) * T]])
                continue
            s = streams[k % 3]
            row = d_t[k * T:(k + 1) * T]
            hip.fill_buffer_device(outs[k].data_ptr(), V, T, k * T, row.data_ptr(), [0, T], s.cuda_stream)
        torch.cuda.synchronize()
        for k in range(calls):
            exp = ref.fill_buffer(V, k * T, (k + 1) * T, [t[k * T:(k + 1) * T]])
            got = host_out[k] if k in host_out else outs[k].cpu().numpy()
            assert same_bits(got, exp), f"call {k}: " + first_diff(got, exp)


def test_independent_calls_overlap_on_two_streams(hip_lib, oracle_lib):
    """A plan without delay state: consecutive device-entry calls on alternating streams are not ordered against each
    other (they overlap on the device); each still renders its own frames exactly, the input history they append stays
    intact (a Delay added afterwards reads it), and a dependent call (a seek; the host entry point) waits for all."""
    import torch
    V, P, T, calls = 8, 512, 2048, 8
    tree = synth.additive_tree(V, P)
    t = synth.time_ramp(0, (calls + 2) * T)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        d_t = torch.from_numpy(t).cuda()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.empty((V, T), dtype=torch.float32, device="cuda") for _ in range(calls)]
        torch.cuda.synchronize()
        for k in range(calls):
            row = d_t[k * T:(k + 1) * T]
            hip.fill_buffer_device(outs[k].data_ptr(), V, T, k * T, row.data_ptr(), [0, T], streams[k % 2].cuda_stream)
        # host entry point right behind them: must wait for both streams, then continue the same timeline
        host = hip.fill_buffer(V, calls * T, (calls + 1) * T, [t[calls * T:(calls + 1) * T]])
        torch.cuda.synchronize()
        for k in range(calls):
            c = [0, 1, T - 1, 777]
            got = outs[k].cpu().numpy()
            for col in c:   # the oracle by random access (no Delay in the graph)
                exp = ref.fill_buffer(V, k * T + col, k * T + col + 1, [t[k * T + col:k * T + col + 1]])
                assert same_bits(got[:, col:col + 1], exp), f"call {k} frame {col}: " + first_diff(got[:, col:col + 1], exp)
        exp = ref.fill_buffer(V, calls * T + 5, calls * T + 6, [t[calls * T + 5:calls * T + 6]])
        assert same_bits(host[:, 5:6], exp)
        # the history the overlapped calls appended: a Delay of the time input by 3000 frames, rendered next
        hip.on_add_node(900001, "Delay")
        hip.on_add_edge(0, 900001, 0, 0)
        hip.on_add_edge(synth.CONST_HANDLE, 900001, f32_bits(3000.0), 1)
        hip.on_add_edge(900001, 0, 0, V)
        a, b = (calls + 1) * T, (calls + 2) * T
        got = hip.fill_buffer(V + 1, a, b, [t[a:b]])
        assert same_bits(got[V], t[a - 3000:b - 3000]), first_diff(got[V], t[a - 3000:b - 3000])


def test_ring_banks_append_history_in_steady_state(hip_lib, oracle_lib):
    """Effects chain through the device entry point: once the delay rings are in steady state the bank launch reads
    the caller's time row directly and appends it to the input history itself (no separate copy).  The history must be
    complete afterwards: an edit adds a Delay of the time input reaching back over all those calls."""
    import torch
    V, T, calls = 3, 1024, 5
    tree = synth.effects_tree(V, 64, taps=2, base_delay=300.0)
    t = synth.time_ramp(0, (calls + 1) * T)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        d_t = torch.from_numpy(t).cuda()
        d_out = torch.empty((V, T), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for k in range(calls):
            row = d_t[k * T:(k + 1) * T]
            hip.fill_buffer_device(d_out.data_ptr(), V, T, k * T, row.data_ptr(), [0, T], s)
            torch.cuda.synchronize()
            exp = ref.fill_buffer(V, k * T, (k + 1) * T, [t[k * T:(k + 1) * T]])
            assert same_bits(d_out.cpu().numpy(), exp), f"call {k}: " + first_diff(d_out.cpu().numpy(), exp)
        for r in (hip, ref):
            r.on_add_node(900001, "Delay")
            r.on_add_edge(0, 900001, 0, 0)
            r.on_add_edge(synth.CONST_HANDLE, 900001, f32_bits(float(calls * T - 7)), 1)
            r.on_add_edge(900001, 0, 0, V)
        a, b = calls * T, (calls + 1) * T
        d_out2 = torch.empty((V + 1, T), dtype=torch.float32, device="cuda")
        hip.fill_buffer_device(d_out2.data_ptr(), V + 1, T, a, d_t[a:b].data_ptr(), [0, T], s)
        torch.cuda.synchronize()
        exp = ref.fill_buffer(V + 1, a, b, [t[a:b]])
        assert same_bits(d_out2.cpu().numpy(), exp), first_diff(d_out2.cpu().numpy(), exp)


def test_device_calls_keep_input_history(hip_lib, oracle_lib):
    """With device-resident full rows the bank kernel itself appends the time row to the slot's history;
    a Delay on the same input must still see earlier calls' samples (tests/ext_input.rs:108-121 semantics)."""
    import torch
    g = synth.GraphArrays()
    p = synth.voice_params(2, 64, seed=9)
    leaves = synth.partial_leaves(g, p["w"], p["amp"]).reshape(2, 64)
    roots = synth.sum_tree(g, leaves)
    g.edge(roots, 0, 0, np.arange(2, dtype=np.uint32))
    d = g.binop(synth.K_DELAY, synth.IN(0), synth.C(np.float32(300.0)), 1)   # out2 = time input delayed 300 frames
    g.edge(d, 0, 0, 2)
    tree = g.finish(3)
    T = 256
    t = (synth.time_ramp(0, 4 * T) * np.float32(0.5)).astype(np.float32)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        d_t = torch.from_numpy(t).cuda()
        d_out = torch.empty((3, T), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for k in range(4):
            row = d_t[k * T:(k + 1) * T]
            hip.fill_buffer_device(d_out.data_ptr(), 3, T, k * T, row.data_ptr(), [0, T], s)
            torch.cuda.synchronize()
            exp = ref.fill_buffer(3, k * T, (k + 1) * T, [t[k * T:(k + 1) * T]])
            assert same_bits(d_out.cpu().numpy(), exp), f"call {k}: " + first_diff(d_out.cpu().numpy(), exp)
        assert hip.plan()["banks"] and hip.plan()["pull_rows"] == 0 and hip.plan()["stage_programs"] == 1


# ---- staged evaluator: envelope + delay chains (N5, N6) ------------------------------------------------------------
def _effects_sequence(hip, ref, V, T, calls, seek_to=None):
    """Contiguous calls, then a seek, on both renderers; returns nothing, asserts bit equality."""
    for k in range(calls):
        t = synth.time_ramp(k * T, (k + 1) * T)
        got, exp = hip.fill_buffer(V, k * T, (k + 1) * T, [t]), ref.fill_buffer(V, k * T, (k + 1) * T, [t])
        assert same_bits(got, exp), f"call {k}: " + first_diff(got, exp)
    if seek_to is not None:
        t = synth.time_ramp(seek_to, seek_to + T)
        got, exp = hip.fill_buffer(V, seek_to, seek_to + T, [t]), ref.fill_buffer(V, seek_to, seek_to + T, [t])
        assert same_bits(got, exp), "after seek: " + first_diff(got, exp)


@pytest.mark.parametrize("V,P,taps,delay,T", [(3, 64, 3, 50.0, 128), (2, 32, 4, 7.0, 33), (1, 128, 2, 300.0, 100), (4, 32, 1, 1.0, 64)])
def test_effects_chain_staged(hip_lib, oracle_lib, V, P, taps, delay, T):
    """config D's shape at sizes the oracle can render in full: bank -> envelope -> K delay taps."""
    tree = synth.effects_tree(V, P, taps=taps, base_delay=delay)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        _effects_sequence(hip, ref, V, T, calls=5, seek_to=10 * T + 3)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and plan["stage_programs"] >= V * (taps + 1) and plan["rings"] >= V * taps, plan
        assert plan["banks"] and plan["banks"][0]["to_ring"] is False or plan["rings"] > 0


def test_staged_matches_pull_on_larger_chain(hip_lib):
    """Same graph through the staged evaluator and the pull interpreter, on device (the oracle would take minutes)."""
    tree = synth.effects_tree(8, 256, taps=4, base_delay=100.0)
    T = 700
    with Renderer(hip_lib, mode="auto") as a, Renderer(hip_lib, mode="pull") as b:
        synth.install(a, tree)
        synth.install(b, tree)
        for k in range(3):
            t = synth.time_ramp(k * T, (k + 1) * T)
            x, y = a.fill_buffer(8, k * T, (k + 1) * T, [t]), b.fill_buffer(8, k * T, (k + 1) * T, [t])
            assert same_bits(x, y), f"call {k}: " + first_diff(x, y)
        assert a.plan()["stage_programs"] > 0 and b.plan()["stage_programs"] == 0


def _random_staged_case(hip_lib, oracle_lib, seed, expect_jit=None):
    rng = np.random.default_rng(5000 + seed)
    steps, n_out = randgraph.random_graph(100 + seed, n_nodes=int(rng.integers(4, 40)), n_inputs=2, n_outputs=3,
                                          signal_delays=False)
    T = 80
    with Renderer(hip_lib, mode="staged") as hip, Renderer(oracle_lib) as ref:
        randgraph.install_steps(hip, steps)
        randgraph.install_steps(ref, steps)
        noise = lambda n: (rng.normal(size=n) * 3).astype(np.float32)
        for k, (start, rows) in enumerate([(0, [synth.time_ramp(0, T), noise(T)]),
                                           (T, [synth.time_ramp(T, 2 * T), noise(17)]),
                                           (2 * T, [synth.time_ramp(2 * T, 3 * T), noise(T)]),
                                           (9000, [synth.time_ramp(9000, 9000 + T), noise(T)])]):
            try:
                exp = ref.fill_buffer(n_out, start, start + T, rows)
            except RenderError as e:
                with pytest.raises(RenderError) as ei:
                    hip.fill_buffer(n_out, start, start + T, rows)
                assert ei.value.status == e.status
                return
            got = hip.fill_buffer(n_out, start, start + T, rows)
            assert same_bits(got, exp), f"seed {seed} call {k}: " + first_diff(got, exp)
        plan = hip.plan()
        if expect_jit is not None and plan["stage_programs"] > 0:
            assert plan["stage_jit"] is expect_jit, plan


@pytest.mark.parametrize("seed", range(24))
def test_random_graphs_staged_mode(hip_lib, oracle_lib, seed):
    """Random graphs with constant delays only, forced through the staged evaluator (no fused banks).  Every program
    has its own skeleton here, so they are interpreted (stage_kernel)."""
    _random_staged_case(hip_lib, oracle_lib, seed, expect_jit=False)


@pytest.mark.parametrize("seed", range(0, 24, 2))
def test_random_graphs_compiled_stage_programs(hip_lib, oracle_lib, seed, monkeypatch):
    """The same graphs with FR_STAGE_JIT=force: every program goes through source generation + hipRTC (jit_stage)."""
    monkeypatch.setenv("FR_STAGE_JIT", "force")
    _random_staged_case(hip_lib, oracle_lib, seed, expect_jit=True)


@pytest.mark.parametrize("V,P,taps,delay,T", [(16, 32, 3, 50.0, 128), (24, 16, 4, 7.0, 33)])
def test_effects_chain_compiled_stage_programs(hip_lib, oracle_lib, V, P, taps, delay, T, monkeypatch):
    """Many voices through the same effects chain: the programs share skeletons, so the default plan compiles them
    (level form on the first call / after the seek, fused form in steady state).  FR_STAGE_JIT=0 gives the same bits."""
    tree = synth.effects_tree(V, P, taps=taps, base_delay=delay)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        _effects_sequence(hip, ref, V, T, calls=5, seek_to=10 * T + 3)
        plan = hip.plan()
        assert plan["stage_jit"] is True and 0 < plan["stage_shapes"] <= 2 * (taps + 2), plan
    monkeypatch.setenv("FR_STAGE_JIT", "0")
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        _effects_sequence(hip, ref, V, T, calls=3)
        assert hip.plan()["stage_jit"] is False


@pytest.mark.parametrize("mode", ["auto", "staged", "pull"])
@pytest.mark.parametrize("seed", range(8))
def test_random_edits_between_calls(hip_lib, oracle_lib, seed, mode):
    """Graph edits during playback (reference dispatch.rs:120-131 -> reference.rs:117-136): after every call a few
    random edits go to both renderers; the engine re-lowers only what the edits reach (plan 'lowering': 'incremental')
    and the next contiguous call must still equal the oracle, delay look-back into pre-edit history included."""
    rng = np.random.default_rng(7000 + seed)
    steps, n_out = randgraph.random_graph(300 + seed, n_nodes=int(rng.integers(6, 30)), n_inputs=2, n_outputs=3,
                                          signal_delays=(mode != "staged"), composites=True)
    T = 48
    with Renderer(hip_lib, mode=mode) as hip, Renderer(oracle_lib) as ref:
        randgraph.install_steps(hip, steps)
        randgraph.install_steps(ref, steps)
        incremental = 0
        for k in range(10):
            rows = [synth.time_ramp(k * T, (k + 1) * T), (rng.normal(size=T) * 3).astype(np.float32)]
            try:
                exp = ref.fill_buffer(n_out, k * T, (k + 1) * T, rows)
            except RenderError as e:
                with pytest.raises(RenderError) as ei:
                    hip.fill_buffer(n_out, k * T, (k + 1) * T, rows)
                assert ei.value.status == e.status
                return
            got = hip.fill_buffer(n_out, k * T, (k + 1) * T, rows)
            assert same_bits(got, exp), f"seed {seed} call {k}: " + first_diff(got, exp)
            plan = hip.plan()
            if k > 0:
                assert plan["lowering"] == "incremental" and plan["plans_built"] == k + 1, plan
                incremental += 1
            edits = randgraph.random_edits(rng, steps, int(rng.integers(1, 4)), signal_delays=(mode != "staged"))
            randgraph.install_steps(hip, edits)
            randgraph.install_steps(ref, edits)
        assert incremental > 0


def test_edit_of_a_large_tree_relowers_only_what_changed(hip_lib, oracle_lib):
    """16 x 4096 partials (720 k mirror nodes): changing one partial's amplitude and another's frequency between two
    calls re-lowers the two leaf-to-root paths (a few dozen nodes), re-matches the two voices and re-uploads the
    parameter table -- milliseconds instead of the from-scratch build -- and renders exactly what a renderer built
    from the edited graph renders."""
    V, P, T = 16, 4096, 256
    tree = synth.additive_tree(V, P)
    CONST = synth.CONST_HANDLE
    e = tree["edges"]
    amp, w = tree["params"]["amp"], tree["params"]["w"]

    def const_edge(value, to_slot, nth):   # the nth edge C(value) -> some node's to_slot
        return np.nonzero((e[:, 0] == CONST) & (e[:, 2] == f32_bits(value)) & (e[:, 3] == to_slot))[0][nth]

    rows = [const_edge(amp[3, 100], 0, 3), const_edge(w[9, 7], 1, 0)]   # amp 1/101 exists once per voice: take voice 3's
    edits = [(int(e[r, 1]), int(e[r, 3]), int(e[r, 2]), f32_bits(new)) for r, new in zip(rows, (np.float32(0.25), np.float32(0.01234)))]
    with Renderer(hip_lib) as hip:
        synth.install(hip, tree)
        hip.fill_buffer(V, 0, T, [synth.time_ramp(0, T)])
        plan0 = hip.plan()
        for to, slot, old, new in edits:
            hip.on_del_edge(CONST, to, old, slot)
            hip.on_add_edge(CONST, to, new, slot)
        got = hip.fill_buffer(V, T, 2 * T, [synth.time_ramp(T, 2 * T)])
        plan1 = hip.plan()
    assert plan0["lowering"] == "full" and plan1["lowering"] == "incremental", (plan0, plan1)
    assert plan1["relowered_nodes"] <= 2 * (11 + 12) and plan1["build_ms"] * 10 < plan0["build_ms"], (plan0, plan1)
    assert plan1["banks"] == plan0["banks"]
    for r, (to, slot, old, new) in zip(rows, edits):
        e[r, 2] = new
    with Renderer(hip_lib) as fresh, Renderer(oracle_lib) as ref:
        synth.install(fresh, tree)
        fresh.fill_buffer(V, 0, T, [synth.time_ramp(0, T)])
        exp = fresh.fill_buffer(V, T, 2 * T, [synth.time_ramp(T, 2 * T)])
        assert same_bits(got, exp), first_diff(got, exp)
        synth.install(ref, tree)
        exp4 = ref.fill_buffer(V, T, T + 4, [synth.time_ramp(T, T + 4)])   # no Delay in this graph: a seek is harmless
        assert same_bits(got[:, :4], exp4), first_diff(got[:, :4], exp4)


@pytest.mark.parametrize("V,P,T,taps", [(4, 64, 128, 0), (3, 32, 100, 2), (6, 32, 64, 0)])
def test_chorus_signal_delay_is_staged(hip_lib, oracle_lib, V, P, T, taps):
    """Delay with a SIGNAL amount (reference.rs:197-216: evaluated at the undelayed t, floored): when interval analysis
    bounds the amount (here base + depth * (x mod 1)) the source is kept in a ring like a constant delay's and the
    program computes the offset per sample -- no pull interpreter, voices stay on the bank kernel."""
    tree = synth.chorus_tree(V, P, depth=30.0, base=11.0, rate_hz=900.0, taps=taps, base_delay=70.0)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        _effects_sequence(hip, ref, V, T, calls=5, seek_to=20 * T + 5)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and plan["banks"] and plan["banks"][0]["to_ring"] is True, plan
        assert 41 + 70 * taps * (taps + 1) // 2 <= plan["max_lookback"] <= 43 + 70 * taps * (taps + 1) // 2, plan
    with Renderer(hip_lib, mode="pull") as pull, Renderer(hip_lib) as hip:   # and against the generic evaluator at a larger size
        big = synth.chorus_tree(8, 256, taps=1)
        synth.install(pull, big)
        synth.install(hip, big)
        for k in range(2):
            t = synth.time_ramp(k * 1500, (k + 1) * 1500)
            a, b = hip.fill_buffer(8, k * 1500, (k + 1) * 1500, [t]), pull.fill_buffer(8, k * 1500, (k + 1) * 1500, [t])
            assert same_bits(a, b), first_diff(a, b)


def test_oversized_expression_is_split_not_pulled(hip_lib, oracle_lib):
    """An output whose expression exceeds one stage program (4096 instructions / 48 registers) is cut into pieces that
    hand values over through rings at the same frame; it used to send every non-bank row to the pull interpreter."""
    g = synth.GraphArrays()
    f = np.float32
    n = 600
    x = g.binop(synth.K_MUL, synth.IN(0), synth.C((0.001 * np.arange(1, n + 1)).astype(f)), n)
    y = g.binop(synth.K_MOD, x, synth.C((1.0 + np.arange(n) % 5).astype(f)), n)
    z = g.binop(synth.K_MIN, y, g.binop(synth.K_DIV, x, synth.C((3.0 + np.arange(n) % 7).astype(f)), n), n)
    terms = g.binop(synth.K_SUM2, z, y, n)
    tree = synth.sum_tree(g, terms.reshape(1, n))[0]
    acc = np.array([tree], dtype=np.uint32)
    rng = np.random.default_rng(5)
    for i in range(2500):
        acc = g.binop(synth.K_SUM2 if i % 3 else synth.K_MIN, acc, terms[rng.integers(n):][:1], 1)
        if i == 1200:
            acc = g.binop(synth.K_SUM2, acc, g.binop(synth.K_DELAY, acc, synth.C(f(9.0)), 1), 1)
    g.edge(acc, 0, 0, 0)
    g.edge(tree, 0, 0, 1)
    tree_d = g.finish(2)
    T = 40
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree_d)
        synth.install(ref, tree_d)
        for k in range(3):
            rows = [synth.time_ramp(k * T, (k + 1) * T)]
            got, exp = hip.fill_buffer(2, k * T, (k + 1) * T, rows), ref.fill_buffer(2, k * T, (k + 1) * T, rows)
            assert same_bits(got, exp), f"call {k}: " + first_diff(got, exp)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and plan["stage_programs"] > 6, plan


def test_graph_edit_rebuilds_delay_state(hip_lib, oracle_lib):
    """Edits between calls apply to ALL times evaluated afterwards, look-back included: the rings
    are rebuilt from the input history with the new graph."""
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        for r in (hip, ref):
            r.on_add_node(1, "F32Constant")
            r.on_add_node(2, "Multiply")    # x = in0 * 2
            r.on_add_node(3, "Delay")       # d = Delay(x, 5)
            r.on_add_node(4, "Sum2")        # out = x + d
            r.on_add_edge(0, 2, 0, 0)
            r.on_add_edge(1, 2, f32_bits(2.0), 1)
            r.on_add_edge(2, 3, 0, 0)
            r.on_add_edge(1, 3, f32_bits(5.0), 1)
            r.on_add_edge(2, 4, 0, 0)
            r.on_add_edge(3, 4, 0, 1)
            r.on_add_edge(4, 0, 0, 0)
        rng = np.random.default_rng(3)
        rows = [rng.normal(size=32).astype(np.float32) for _ in range(4)]
        for k in range(2):
            assert same_bits(hip.fill_buffer(1, 32 * k, 32 * (k + 1), [rows[k]]), ref.fill_buffer(1, 32 * k, 32 * (k + 1), [rows[k]]))
        for r in (hip, ref):   # change the gain: history before the edit must now be seen through gain 3
            r.on_del_edge(1, 2, f32_bits(2.0), 1)
            r.on_add_edge(1, 2, f32_bits(3.0), 1)
        for k in range(2, 4):
            got, exp = hip.fill_buffer(1, 32 * k, 32 * (k + 1), [rows[k]]), ref.fill_buffer(1, 32 * k, 32 * (k + 1), [rows[k]])
            assert same_bits(got, exp), first_diff(got, exp)
        assert hip.plan()["rings"] == 1


def test_config_d_full_size_sampled_against_oracle(hip_lib, oracle_lib):
    """BASELINE config D (1024 partials x 128 voices, detune + ADSR + 4-tap delay chain reaching back 24000 frames) at full
    size, rendered as seven contiguous 4800-frame calls from frame 0 -- the first call rebuilds nothing, the later ones
    run on the rings, the seventh crosses the rings' wrap (capacity 32768) -- and compared on 32 voices x 40 frames: the
    envelope's break points, every tap's first live frame and its neighbours, call boundaries, the wrap, random frames.
    The oracle cannot render it (the pull model costs 2^4 upstream evaluations per sample) but answers single samples from
    the stored input history (reference.rs:197-216: a Delay re-evaluates its source at t - d)."""
    V, P, T, calls = 128, 1024, 4800, 7
    tree = synth.effects_tree(V, P)
    rows = [(k * T, synth.time_ramp(k * T, (k + 1) * T)) for k in range(calls)]
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        got = [(st, hip.fill_buffer(V, st, st + T, [row])) for st, row in rows]
        plan = hip.plan()
        # per voice: the bank's mix (read by the envelope stage) + x0..x3 (each read back by the next tap)
        assert plan["pull_rows"] == 0 and plan["rings"] == V * 5 and plan["max_lookback"] == 24000, plan
        _oracle_with_history(ref, tree, rows)
        rng = np.random.default_rng(4)
        voices = np.unique(np.concatenate([[0, 1, 63, 127], rng.integers(0, V, 40)]))[:32]
        frames = np.unique(np.concatenate([[0, 1, 479, 480, 481, 2399, 2400, 2401, 2879, 2880, 2881, T - 1, T, 7199, 7200, 9600, 11999, 12000, 14400,
                                            23999, 24000, 24001, 28800, 32767, 32768, 32769, calls * T - 1], rng.integers(0, calls * T, 13)]))
        assert len(voices) == 32 and len(frames) >= 36
        g = _sampled_parity(ref, got, voices, frames, "config D")
        assert np.abs(g).max() > 0.01


def test_bank_from_composite_effect_instances(hip_lib, oracle_lib):
    """The voice built from instances of ONE composite `Partial(t, w, amp)` effect (the shape an effect file gives):
    lowering inlines the instances, folds the constant inputs, and the planner still finds the bank."""
    V, P, T = 3, 128, 300
    t = synth.time_ramp(0, T)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref, Renderer(hip_lib) as flat:
        synth.install_composite_tree(hip, V, P, seed=4, detune=True)
        synth.install_composite_tree(ref, V, P, seed=4, detune=True)
        synth.install(flat, synth.additive_tree(V, P, seed=4, detune=True))
        got, exp = hip.fill_buffer(V, 0, T, [t]), ref.fill_buffer(V, 0, T, [t])
        assert same_bits(got, exp), first_diff(got, exp)
        assert same_bits(got, flat.fill_buffer(V, 0, T, [t]))
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and plan["banks"][0]["voices"] == V and plan["banks"][0]["partials"] == P, plan


# ---- voices that are not balanced power-of-two trees -------------------------------------------------------------------
@pytest.mark.parametrize("V,P,T", [(2, 1000, 200), (3, 100, 130), (1, 24, 64), (2, 17, 70), (1, 4097, 65), (4, 255, 129)])
def test_bank_general_partial_counts(hip_lib, oracle_lib, V, P, T):
    """Partial counts that are not powers of two: the adjacent-pairs tree carries odd elements up, the planner cuts
    it into complete sub-trees and a merge schedule (match.hpp), the schedule kernel evaluates it in the tree's order."""
    tree = synth.additive_tree(V, P, seed=P, detune=True)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for k, idx in enumerate([0, T, 5 * T]):
            t = synth.time_ramp(idx, idx + T)
            got, exp = hip.fill_buffer(V, idx, idx + T, [t]), ref.fill_buffer(V, idx, idx + T, [t])
            assert same_bits(got, exp), f"call {k}: " + first_diff(got, exp)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and plan["stage_programs"] == 0, plan
        assert plan["banks"][0]["general_tree"] and plan["banks"][0]["partials"] == P, plan


def test_bank_unbalanced_trees(hip_lib, oracle_lib):
    """A left-leaning chain (((l0+l1)+l2)+...), a chain of balanced blocks, and a balanced voice side by side; the
    chain then feeds a Delay (general voice -> ring -> program)."""
    g = synth.GraphArrays()
    p = synth.voice_params(3, 64, seed=21, detune=True)
    leaves = synth.partial_leaves(g, p["w"], p["amp"]).reshape(3, 64)
    acc = leaves[0, 0:1]
    for k in range(1, 40):                                   # left chain over 40 leaves
        acc = g.binop(synth.K_SUM2, acc, leaves[0, k:k + 1], 1)
    blocks = [synth.sum_tree(g, leaves[1:2, i:i + 8]) for i in range(0, 64, 8)]
    chain2 = blocks[0]
    for b in blocks[1:]:                                     # chain of 8-leaf balanced blocks
        chain2 = g.binop(synth.K_SUM2, chain2, b, 1)
    bal = synth.sum_tree(g, leaves[2:3, :])                  # ordinary balanced voice
    d = g.binop(synth.K_SUM2, acc, g.binop(synth.K_DELAY, acc, synth.C(np.float32(9.0)), 1), 1)
    for row, h in enumerate([acc, chain2, bal, d]):
        g.edge(h, 0, 0, row)
    tree = g.finish(4)
    T = 150
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for idx in (0, T):
            t = synth.time_ramp(idx, idx + T)
            got, exp = hip.fill_buffer(4, idx, idx + T, [t]), ref.fill_buffer(4, idx, idx + T, [t])
            assert same_bits(got, exp), first_diff(got, exp)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and any(b["general_tree"] for b in plan["banks"]) and any(not b["general_tree"] for b in plan["banks"]), plan


def test_fused_stage_mode_equals_level_mode(hip_lib, oracle_lib):
    """Steady-state calls no longer than the shortest ring delay run every delay level in ONE launch (sub-windows of
    that length when the call is longer); first call, seek and edits use the level-by-level form.  Same bits."""
    V, P, T = 3, 64, 100
    tree = synth.effects_tree(V, P, taps=3, base_delay=70.0)   # delays 70, 140, 210: fused in sub-windows of 70 frames
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        _effects_sequence(hip, ref, V, T, calls=6, seek_to=40 * T)
        _effects_sequence(hip, ref, V, 64, calls=3)            # seek back to 0 with a shorter call length
        plan = hip.plan()
        assert plan["fused_programs"] > 0 and plan["fused_max_frames"] == 70, plan


@pytest.mark.parametrize("V,P,T", [(2, 256, 1), (3, 1024, 7), (1, 4096, 32), (5, 512, 16), (2, 8192, 3)])
def test_bank_short_calls(hip_lib, oracle_lib, V, P, T):
    """Calls of <= 32 frames use the lanes-over-partials kernel (wavefront-shuffle tree + LDS + chunk combine)."""
    tree = synth.additive_tree(V, P, seed=P + T, detune=True)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        idx = 0
        for k in range(4):   # includes frame 0 (all phases zero: the zero-sign repair) and a longer call in between
            n = T if k != 2 else 100
            t = synth.time_ramp(idx, idx + n)
            got, exp = hip.fill_buffer(V, idx, idx + n, [t]), ref.fill_buffer(V, idx, idx + n, [t])
            assert same_bits(got, exp), f"call {k}: " + first_diff(got, exp)
            idx += n
        assert hip.plan()["pull_rows"] == 0


def test_short_device_calls_keep_history(hip_lib, oracle_lib):
    """Short device-resident calls must still append the time row to the input history (no deferred append there)."""
    import torch
    g = synth.GraphArrays()
    p = synth.voice_params(1, 256, seed=5)
    root = synth.sum_tree(g, synth.partial_leaves(g, p["w"], p["amp"]).reshape(1, 256))
    g.edge(root, 0, 0, 0)
    d = g.binop(synth.K_DELAY, synth.IN(0), synth.C(np.float32(20.0)), 1)
    g.edge(d, 0, 0, 1)
    tree = g.finish(2)
    T = 16
    t = synth.time_ramp(0, 6 * T)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        d_t = torch.from_numpy(t).cuda()
        d_out = torch.empty((2, T), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for k in range(6):
            row = d_t[k * T:(k + 1) * T]
            hip.fill_buffer_device(d_out.data_ptr(), 2, T, k * T, row.data_ptr(), [0, T], s)
            torch.cuda.synchronize()
            exp = ref.fill_buffer(2, k * T, (k + 1) * T, [t[k * T:(k + 1) * T]])
            assert same_bits(d_out.cpu().numpy(), exp), f"call {k}: " + first_diff(d_out.cpu().numpy(), exp)


# ---- hipRTC-specialised voices: leaves of a shape the hand-written kernel does not know -------------------------------
def _triangle_tree(V, P, am=False, delayed=False):
    g = synth.GraphArrays()
    p = synth.voice_params(V, P, seed=P + 1, detune=True)
    leaves = synth.triangle_leaves(g, p["w"], p["amp"], am_slot=1 if am else None).reshape(V, P)
    roots = synth.sum_tree(g, leaves)
    g.edge(roots, 0, 0, np.arange(V, dtype=np.uint32))
    n_out = V
    if delayed:   # one more output: voice 0 plus itself 37 frames ago (the JIT bank then fills a ring)
        d = g.binop(synth.K_SUM2, roots[0:1], g.binop(synth.K_DELAY, roots[0:1], synth.C(np.float32(37.0)), 1), 1)
        g.edge(d, 0, 0, V)
        n_out += 1
    return g.finish(n_out)


@pytest.mark.parametrize("V,P,T,am,delayed", [(3, 64, 200, False, False), (2, 1024, 130, False, False), (2, 128, 100, True, False),
                                              (2, 32, 150, True, True), (1, 8192, 70, False, False)])
def test_jit_specialised_voices(hip_lib, oracle_lib, V, P, T, am, delayed):
    tree = _triangle_tree(V, P, am, delayed)
    n_out = tree["n_outputs"]
    rng = np.random.default_rng(P)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for k, idx in enumerate([0, T, 2 * T, 50 * T]):   # contiguous calls, then a seek
            rows = [synth.time_ramp(idx, idx + T), (rng.normal(size=T) * 2).astype(np.float32)]
            got, exp = hip.fill_buffer(n_out, idx, idx + T, rows), ref.fill_buffer(n_out, idx, idx + T, rows)
            assert same_bits(got, exp), f"call {k}: " + first_diff(got, exp)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and plan["jit_kernels_compiled"] == 1, plan
        jb = [b for b in plan["banks"] if b["jit"]]   # (voices that feed a ring launch separately from direct ones)
        assert sum(b["voices"] for b in jb) == V and all(b["partials"] == P and b["leaf_params"] == 2 for b in jb), plan


def test_jit_compiles_in_the_background(hip_lib, oracle_lib):
    """The ABI's default: hipRTC runs on a worker thread.  The first calls are served without the specialised kernel
    (plan says jit_pending), never waiting ~0.1 s for the compiler; once it is ready the plan switches over.  Same bits
    all along, delay lines behind the voices included (the switch rebuilds their look-back)."""
    import time
    V, P, T = 3, 64, 96
    tree = _triangle_tree(V, P, False, True)
    n_out = tree["n_outputs"]
    rng = np.random.default_rng(1)
    with Renderer(hip_lib, sync_compile=False) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        seen_pending = seen_jit = False
        t_first = None
        for k in range(400):
            rows = [synth.time_ramp(k * T, (k + 1) * T), (rng.normal(size=T) * 2).astype(np.float32)]
            t0 = time.perf_counter()
            got = hip.fill_buffer(n_out, k * T, (k + 1) * T, rows)
            if k == 0:
                t_first = time.perf_counter() - t0
            exp = ref.fill_buffer(n_out, k * T, (k + 1) * T, rows)
            assert same_bits(got, exp), f"call {k}: " + first_diff(got, exp)
            plan = hip.plan()
            if plan["jit_pending"]:
                seen_pending = True
                assert not any(b["jit"] for b in plan["banks"])
            elif any(b["jit"] for b in plan["banks"]):
                seen_jit = True
                if k > 20:
                    break
            time.sleep(0.002)
        assert seen_pending and seen_jit, plan
        assert plan["jit_kernels_compiled"] >= 1


def test_jit_voices_unusual_time_inputs(hip_lib, oracle_lib):
    """Generated leaves have two kernel bodies: Modulo(x, 1.0) as one v_fract_f32 when the host proved the argument
    finite and >= +0 for inputs in [+0, 2^32] AND the wave's inputs are in that range, the fmod form otherwise.
    Negative, fractional, huge, NaN/inf and -0.0 times must take the second body; same bits either way."""
    rng = np.random.default_rng(7)
    rows = [
        synth.time_ramp(0, 200),                                      # in range: fract body
        (-synth.time_ramp(0, 200)),                                   # negative times
        (rng.normal(size=200) * 1000).astype(np.float32),             # mixed sign inside one wave
        np.array([0, 1, 2, np.nan, np.inf, -np.inf, 1e30, 3e38, 1e-30, -0.0] * 20, dtype=np.float32),
        np.array([4294967296.0, 4294967808.0, 16777216.0, 0.5] * 50, dtype=np.float32),   # at and just above 2^32
    ]
    for am in (False, True):
        tree = _triangle_tree(2, 64, am, False)
        with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
            synth.install(hip, tree)
            synth.install(ref, tree)
            for i, row in enumerate(rows):
                ins = [row, (rng.normal(size=200) * 2).astype(np.float32)]
                got = hip.fill_buffer(2, i * 200, (i + 1) * 200, ins)
                exp = ref.fill_buffer(2, i * 200, (i + 1) * 200, ins)
                assert same_bits(got, exp), f"am={am} row {i}: " + first_diff(got, exp)
            assert any(b["jit"] for b in hip.plan()["banks"])
    # negative frequencies: the host cannot prove the argument non-negative, the fmod body runs for every wave
    g = synth.GraphArrays()
    w = (np.linspace(-0.01, 0.02, 128)).astype(np.float32).reshape(2, 64)
    leaves = synth.triangle_leaves(g, w, np.ones_like(w)).reshape(2, 64)
    g.edge(synth.sum_tree(g, leaves), 0, 0, np.arange(2, dtype=np.uint32))
    tree = g.finish(2)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        t = synth.time_ramp(0, 300)
        assert same_bits(hip.fill_buffer(2, 0, 300, [t]), ref.fill_buffer(2, 0, 300, [t]))
        assert any(b["jit"] for b in hip.plan()["banks"])


def test_template_voices_through_generated_kernel(hip_lib, oracle_lib, monkeypatch):
    """FR_BANK_TEMPLATE=0: the N1 partial itself goes through shape matching + hipRTC instead of the hand-written
    kernel -- the generated leaf with its exact peepholes (x mod 1 as fract in the in-range body, Minimum(u, -u) as
    -|u| where u cannot be -0) must reproduce the oracle on ordinary and on hostile time inputs."""
    monkeypatch.setenv("FR_BANK_TEMPLATE", "0")
    tree = synth.additive_tree(3, 256, detune=True)
    rng = np.random.default_rng(3)
    rows = [
        synth.time_ramp(0, 300),
        (-synth.time_ramp(0, 300)),
        (rng.normal(size=300) * 500).astype(np.float32),
        np.array([0, 0.5, 1, np.nan, np.inf, -np.inf, 1e30, 3e38, 1e-30, -0.0, 4294967296.0, 8e9] * 25, dtype=np.float32),
        synth.time_ramp(100000, 100300),
    ]
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for i, row in enumerate(rows):
            got, exp = hip.fill_buffer(3, i * 300, (i + 1) * 300, [row]), ref.fill_buffer(3, i * 300, (i + 1) * 300, [row])
            assert same_bits(got, exp), f"row {i}: " + first_diff(got, exp)
        plan = hip.plan()
        assert plan["banks"] and all(b["jit"] for b in plan["banks"]), plan


def test_many_small_generated_voices_kernel(hip_lib, oracle_lib, monkeypatch):
    """hipRTC-specialised voices (triangle leaves, one amplitude-modulated) in the many-small-voices form
    (jit_bank_multi) against the one-voice-per-workgroup form (FR_BANK_MULTI=0) and the oracle."""
    for V, P, T, am in ((1100, 32, 1100, False), (600, 64, 800, True)):
        tree = _triangle_tree(V, P, am, False)
        rng = np.random.default_rng(P)
        rows = [[synth.time_ramp(0, T), (rng.normal(size=T) * 2).astype(np.float32)],
                [np.concatenate([-synth.time_ramp(0, 64), [np.nan, np.inf, -0.0, 1e30], synth.time_ramp(0, T - 68)]).astype(np.float32),
                 (rng.normal(size=T) * 2).astype(np.float32)]]
        outs = {}
        for multi in ("1", "0"):
            monkeypatch.setenv("FR_BANK_MULTI", multi)
            with Renderer(hip_lib) as hip:
                synth.install(hip, tree)
                outs[multi] = [hip.fill_buffer(V, i * T, (i + 1) * T, r) for i, r in enumerate(rows)]
                assert any(b["jit"] for b in hip.plan()["banks"])
        for i in range(len(rows)):
            assert same_bits(outs["1"][i], outs["0"][i]), f"V={V} P={P} row {i}: " + first_diff(outs["1"][i], outs["0"][i])
        with Renderer(oracle_lib) as ref:
            synth.install(ref, tree)
            exp = ref.fill_buffer(V, 0, 66, [rows[0][0][:66], rows[0][1][:66]])
            assert same_bits(outs["1"][0][:, :66], exp), f"V={V} P={P}: " + first_diff(outs["1"][0][:, :66], exp)


def test_jit_disabled_gives_the_same_bits(hip_lib, monkeypatch):
    """FR_JIT=0: the same voices run as stage programs / pull instead of a specialised kernel; identical output."""
    tree = _triangle_tree(2, 32, am=True)
    T = 96
    rows = [synth.time_ramp(0, T), np.linspace(-1, 1, T).astype(np.float32)]
    with Renderer(hip_lib) as a:
        synth.install(a, tree)
        x = a.fill_buffer(2, 0, T, rows)
        assert any(b["jit"] for b in a.plan()["banks"])
    monkeypatch.setenv("FR_JIT", "0")
    with Renderer(hip_lib) as b:
        synth.install(b, tree)
        y = b.fill_buffer(2, 0, T, rows)
        assert not any(bk["jit"] for bk in b.plan()["banks"])
    assert same_bits(x, y)


# ---- Minimum beside a literal zero: the sign of the tie ---------------------------------------------------------------
def _min_zero_steps():
    """Six output rows: Minimum of input 0 with +0 / -0 constants on either side, and of two expressions that are zeros
    of either sign (0 * x).  A tie between -0 and +0 goes to the RIGHT operand (`(a < b || b != b) ? a : b`), which
    v_min_f32 does not do: found by tools/stress_parity.py in hipRTC-compiled programs, where the backend had turned the
    select into v_min_f32 because the literal operand cannot be NaN."""
    z, nz = f32_bits(0.0), f32_bits(-0.0)
    steps = [("node", 1, Effect.primitive("F32Constant"))]
    for i, (lhs, rhs) in enumerate([("x", z), ("x", nz), (z, "x"), (nz, "x")]):
        h = 10 + i
        steps.append(("node", h, Effect.primitive("Minimum")))
        for slot, operand in enumerate((lhs, rhs)):
            steps.append(("edge", 0, h, 0, slot) if operand == "x" else ("edge", 1, h, operand, slot))
        steps.append(("edge", h, 0, 0, i))
    # rows 4, 5: Minimum(0 * x, 0) and Minimum(0, 0 * x)
    steps += [("node", 20, Effect.primitive("Multiply")), ("edge", 1, 20, z, 0), ("edge", 0, 20, 0, 1),
              ("node", 21, Effect.primitive("Minimum")), ("edge", 20, 21, 0, 0), ("edge", 1, 21, z, 1), ("edge", 21, 0, 0, 4),
              ("node", 22, Effect.primitive("Minimum")), ("edge", 1, 22, z, 0), ("edge", 20, 22, 0, 1), ("edge", 22, 0, 0, 5)]
    return steps, 6


@pytest.mark.gpu
@pytest.mark.parametrize("semantics", ["reference", "sparkle"])
@pytest.mark.parametrize("mode", ["auto", "staged", "staged+jit", "pull"])
def test_minimum_beside_a_literal_zero_keeps_the_sign_of_the_tie(hip_lib, oracle_lib, monkeypatch, mode, semantics):
    steps, n_out = _min_zero_steps()
    x = np.array([0.0, -0.0, np.nan, 1.0, -1.0, np.inf, -np.inf, 1e-40, -1e-40, 3.5, -0.0, 0.0], np.float32)
    if mode == "staged+jit":
        monkeypatch.setenv("FR_STAGE_JIT", "force")
    with Renderer(hip_lib, mode=mode.split("+")[0], semantics=semantics) as hip, Renderer(oracle_lib, semantics=semantics) as ref:
        randgraph.install_steps(hip, steps)
        randgraph.install_steps(ref, steps)
        got, exp = hip.fill_buffer(n_out, 0, len(x), [x]), ref.fill_buffer(n_out, 0, len(x), [x])
        assert same_bits(got, exp), first_diff(got, exp)
        if mode == "staged+jit":
            assert hip.plan()["stage_jit"], hip.plan()
    # the rule itself, spelled out for the ties: the right operand wins
    assert exp.view(np.uint32)[0, 1] == 0x00000000 and exp.view(np.uint32)[1, 0] == 0x80000000


@pytest.mark.gpu
def test_minimum_beside_a_literal_zero_in_a_compiled_voice(hip_lib, oracle_lib):
    """The same tie inside a hipRTC-specialised bank leaf: a partial  amp * Minimum(phase - 0.5, 0)  (a half-wave: zeros
    of both signs for half of every period), summed per voice."""
    V, P, T = 2, 64, 400
    g = synth.GraphArrays()
    p = synth.voice_params(V, P, seed=77, detune=True)
    w, amp = p["w"].ravel(), p["amp"].ravel()
    n = len(w)
    f = np.float32
    ph = g.binop(synth.K_MOD, g.binop(synth.K_MUL, synth.IN(0), synth.C(w), n), synth.C(f(1.0)), n)
    u = g.binop(synth.K_SUM2, ph, synth.C(f(-0.5)), n)
    half = g.binop(synth.K_MIN, g.binop(synth.K_MUL, synth.C(f(0.0)), u, n), g.binop(synth.K_MIN, u, synth.C(f(0.0)), n), n)
    leaf = g.binop(synth.K_MUL, synth.C(amp), half, n)
    roots = synth.sum_tree(g, leaf.reshape(V, P))
    g.edge(roots, 0, 0, np.arange(V, dtype=np.uint32))
    tree = g.finish(V)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for idx in (0, T, 40 * T):
            rows = [synth.time_ramp(idx, idx + T)]
            got, exp = hip.fill_buffer(V, idx, idx + T, rows), ref.fill_buffer(V, idx, idx + T, rows)
            assert same_bits(got, exp), first_diff(got, exp)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and any(b["jit"] for b in plan["banks"]), plan


# ---- Modulo(x, 1.0) of -0 and of negative integers: fmodf keeps the dividend's sign -----------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "staged", "staged+jit", "pull"])
def test_modulo_by_one_keeps_the_sign_of_a_zero_result(hip_lib, oracle_lib, monkeypatch, mode):
    """`-3 % 1` is -0 and stays -0 (`rem < 0` is false, reference.rs:254-261).  The generated code's shortcut for a literal
    divisor of 1.0, a - trunc(a), gave +0 (found by tools/stress_parity.py, seed 5587)."""
    x = np.array([-0.0, 0.0, -3.0, 3.0, -1.0, -2.5, 2.5, -1e-30, 1e-30, -16777216.0, np.inf, -np.inf, np.nan, -7.0], np.float32)
    steps = [("node", 1, Effect.primitive("F32Constant")),
             ("node", 2, Effect.primitive("Modulo")), ("edge", 0, 2, 0, 0), ("edge", 1, 2, f32_bits(1.0), 1), ("edge", 2, 0, 0, 0),
             # the same through a product, so that the zero's sign reaches a non-zero-free consumer: 1 / (x mod 1)
             ("node", 3, Effect.primitive("Divide")), ("edge", 1, 3, f32_bits(1.0), 0), ("edge", 2, 3, 0, 1), ("edge", 3, 0, 0, 1)]
    if mode == "staged+jit":
        monkeypatch.setenv("FR_STAGE_JIT", "force")
    with Renderer(hip_lib, mode=mode.split("+")[0]) as hip, Renderer(oracle_lib) as ref:
        randgraph.install_steps(hip, steps)
        randgraph.install_steps(ref, steps)
        got, exp = hip.fill_buffer(2, 0, len(x), [x]), ref.fill_buffer(2, 0, len(x), [x])
        assert same_bits(got, exp), first_diff(got, exp)
    assert exp.view(np.uint32)[0, 2] == 0x80000000 and exp[1, 2] == -np.inf    # -3 mod 1 = -0; 1 / -0 = -inf


@pytest.mark.gpu
def test_modulo_by_one_in_a_compiled_voice_with_negative_phases(hip_lib, oracle_lib):
    """A sawtooth partial amp * Modulo(x * w, 1) with quarter-integer rates, driven by an input that runs through negative
    integers and zeros of both signs (the general body of the generated kernel; the v_fract body needs inputs >= +0)."""
    V, P, T = 2, 64, 256
    g = synth.GraphArrays()
    n = V * P
    w = (0.25 * (1 + np.arange(n) % P)).astype(np.float32)
    amp = (1.0 / (1 + np.arange(n) % P)).astype(np.float32)
    ph = g.binop(synth.K_MOD, g.binop(synth.K_MUL, synth.IN(0), synth.C(w), n), synth.C(np.float32(1.0)), n)
    leaf = g.binop(synth.K_MUL, synth.C(amp), ph, n)
    roots = synth.sum_tree(g, leaf.reshape(V, P))
    g.edge(roots, 0, 0, np.arange(V, dtype=np.uint32))
    tree = g.finish(V)
    rng = np.random.default_rng(5)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for k, idx in enumerate((0, T, 2 * T)):
            x = np.round(rng.normal(size=T) * 6).astype(np.float32) * np.float32([1.0, -1.0, 0.25][k])   # integers, incl. -0 (0 * -1)
            got, exp = hip.fill_buffer(V, idx, idx + T, [x]), ref.fill_buffer(V, idx, idx + T, [x])
            assert same_bits(got, exp), first_diff(got, exp)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and any(b["jit"] for b in plan["banks"]), plan


@pytest.mark.gpu
def test_registered_destination_lifecycle(hip_lib, oracle_lib):
    """Renderers that register a small, unaligned destination and are destroyed WITHOUT unregistering it, the memory freed
    and handed out again by the allocator to the next one (what tools/stress_calls.py did when a GPU memory fault was
    reported): destroy drops the registration, every render lands in the buffer it was given."""
    steps = [("node", 1, Effect.primitive("F32Constant")), ("node", 2, Effect.primitive("Sum2")),
             ("edge", 0, 2, 0, 0), ("edge", 1, 2, f32_bits(0.5), 1), ("edge", 2, 0, 0, 0), ("edge", 0, 0, 0, 1), ("edge", 2, 0, 0, 2)]
    with Renderer(oracle_lib) as ref:
        randgraph.install_steps(ref, steps)
        x = np.arange(300, dtype=np.float32)
        exp = ref.fill_buffer(3, 0, 300, [x])
    for i in range(40):
        out = np.zeros((3, 300), np.float32)          # 3600 bytes from malloc: not page-aligned, reused from round to round
        r = Renderer(hip_lib)
        randgraph.install_steps(r, steps)
        r.host_register(out)
        got = r.fill_buffer(3, 0, 300, [x], out=out)
        assert got is out and same_bits(out, exp), i
        if i % 2:
            r.host_unregister(out)
        r.close()
        del out, got


# ---- generated leaves: y + (2^k * v) as one fused multiply-add ---------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("c_pow2,c_add,divisor,swap", [(4.0, -2.0, 1.0, False), (-4.0, 1.0, 1.0, True), (2.0, -0.0, 1.0, False),
                                                       (-16.0, 0.0, 0.5, True), (8.0, -4.0, 0.75, False), (2.0, 1e-30, 3.0, True)])
def test_power_of_two_fma_fold_in_compiled_voices(hip_lib, oracle_lib, c_pow2, c_add, divisor, swap):
    """leaf = amp * (c_add + c_pow2 * Modulo(x * w, divisor)), either operand order, on an input that makes the sum cancel
    to zero, the remainder -0 / +0 / tiny, and the phase NaN or infinite: the fused form rounds the same real number as the
    graph's two operations, zero signs included (leafjit.cpp)."""
    V, P, T = 2, 32, 192
    g = synth.GraphArrays()
    n = V * P
    f = np.float32
    w = (0.25 * (1 + np.arange(n) % P)).astype(f)
    amp = (1.0 / (1 + np.arange(n) % 7)).astype(f)
    ph = g.binop(synth.K_MOD, g.binop(synth.K_MUL, synth.IN(0), synth.C(w), n), synth.C(f(divisor)), n)
    prod = g.binop(synth.K_MUL, ph, synth.C(f(c_pow2)), n) if swap else g.binop(synth.K_MUL, synth.C(f(c_pow2)), ph, n)
    s = g.binop(synth.K_SUM2, prod, synth.C(f(c_add)), n) if swap else g.binop(synth.K_SUM2, synth.C(f(c_add)), prod, n)
    leaf = g.binop(synth.K_MUL, synth.C(amp), s, n)
    g.edge(synth.sum_tree(g, leaf.reshape(V, P)), 0, 0, np.arange(V, dtype=np.uint32))
    tree = g.finish(V)
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, 2.0, -2.0, 0.5, -0.5, 1e-30, -1e-30, 1e30, np.inf, -np.inf, np.nan, 3.0, -7.0, 0.125, 1e-42], f)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for k, idx in enumerate((0, T, 2 * T)):
            x = (np.round(rng.normal(size=T) * 8) / 4).astype(f) if k < 2 else synth.time_ramp(idx, idx + T)   # quarter-integers
            x[rng.integers(T, size=24)] = special[rng.integers(len(special), size=24)]
            got, exp = hip.fill_buffer(V, idx, idx + T, [x]), ref.fill_buffer(V, idx, idx + T, [x])
            assert same_bits(got, exp), first_diff(got, exp)
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and any(b["jit"] for b in plan["banks"]), plan


# ---- compiled kernels kept on disk (FR_JIT_CACHE) -----------------------------------------------------------------------
@pytest.mark.gpu
def test_compiled_kernels_are_reused_from_disk(hip_lib, oracle_lib, tmp_path, monkeypatch):
    """With FR_JIT_CACHE set, a second renderer -- a second process in real life -- loads the code object the first one
    compiled instead of running hipRTC again: same bits, `jit_disk_hits` in the plan, first-call time without the ~0.1 s
    compile.  A file whose recorded source text differs (a stale or colliding entry) is ignored."""
    monkeypatch.setenv("FR_JIT_CACHE", str(tmp_path))
    tree = _triangle_tree(2, 64, True, False)
    T = 128
    rows = [synth.time_ramp(0, T), np.linspace(-1, 1, T).astype(np.float32)]
    with Renderer(oracle_lib) as ref:
        synth.install(ref, tree)
        exp = ref.fill_buffer(2, 0, T, rows)
    for attempt in range(3):   # every renderer has its own kernel cache: the second and third load the file the first wrote
        with Renderer(hip_lib) as hip:
            synth.install(hip, tree)
            assert same_bits(hip.fill_buffer(2, 0, T, rows), exp)
            plan = hip.plan()
            assert plan["jit_kernels_compiled"] == 1 and plan["jit_disk_hits"] == (1 if attempt else 0), plan
    files = sorted(tmp_path.glob("fr_*.jitbin"))
    assert len(files) == 1, files                         # one kernel, written once
    # a second process: ask a child interpreter to render the same thing and report where its kernel came from
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = (
        f"import sys, json, numpy as np; sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})\n"
        "import test_hip_parity as t, libfriendship_amd\n"
        "from libfriendship_amd import synth\n"
        "from libfriendship_amd.capi import Renderer\n"
        "tree = t._triangle_tree(2, 64, True, False)\n"
        "rows = [synth.time_ramp(0, 128), np.linspace(-1, 1, 128).astype(np.float32)]\n"
        "with Renderer(libfriendship_amd.hip_lib()) as hip:\n"
        "    synth.install(hip, tree)\n"
        "    out = hip.fill_buffer(2, 0, 128, rows)\n"
        "    p = hip.plan()\n"
        "print(json.dumps({'hits': p['jit_disk_hits'], 'compiled': p['jit_kernels_compiled'], 'ms': p['jit_compile_ms'], 'sum': float(np.nansum(out))}))\n"
    )
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300, env=dict(os.environ, FR_JIT_CACHE=str(tmp_path)))
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["hits"] == 1 and rep["compiled"] == 1 and rep["ms"] < 50.0, rep      # loaded, not compiled (~110 ms)
    assert rep["sum"] == float(np.nansum(exp))
    # a corrupted entry is a miss: flip a byte of the recorded source text, the child compiles again and rewrites the file
    data = bytearray(files[0].read_bytes())
    data[24 + 40] ^= 0x20
    files[0].write_bytes(bytes(data))
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300, env=dict(os.environ, FR_JIT_CACHE=str(tmp_path)))
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["hits"] == 0 and rep["compiled"] == 1 and rep["sum"] == float(np.nansum(exp)), rep


# ---- block streaming: one resident launch renders 64-frame blocks on a doorbell ------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("V,P", [(4, 512), (3, 128), (1, 4096), (16, 128), (5, 1024), (64, 256)])
def test_block_streaming_renders_what_fill_buffer_renders(hip_lib, oracle_lib, V, P):
    """fr_stream_begin / fr_stream_block: blocks of 1..64 frames through the resident kernel, ordinary and hostile input
    rows, equal the oracle's fill_buffer of the same frames bit for bit; an edit retires the launch (the next block is
    refused until the stream is begun again) and the new graph is what the next stream renders."""
    tree = synth.additive_tree(V, P, params_as_nodes=bool(P % 3))
    rng = np.random.default_rng(V * 1000 + P)
    special = np.array([0.0, -0.0, -1.0, 0.5, 1e-42, 16777216.0, 4294967296.0, 4294967808.0, 1e30, np.inf, -np.inf, np.nan], np.float32)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        idx = 0
        for rnd in range(2):
            hip.stream_begin(V)
            for k in range(40):
                T = int(rng.choice([1, 2, 17, 63, 64, 64, 64]))
                row = synth.time_ramp(idx, idx + T)
                if k % 5 == 4:
                    row = row.copy()
                    row[rng.integers(T, size=max(1, T // 4))] = special[rng.integers(len(special), size=max(1, T // 4))]
                got = hip.stream_block(idx, row)
                exp = ref.fill_buffer(V, idx, idx + T, [row])
                assert same_bits(got, exp), f"round {rnd} block {k} (T={T}): " + first_diff(got, exp)
                idx += T
            # an edit: one amplitude changes; the resident launch is retired by it
            e = tree["edges"]
            rows_c = np.nonzero((e[:, 0] == synth.CONST_HANDLE) & (e[:, 3] == 0) & np.isin(e[:, 2], synth.bits(tree["params"]["amp"][tree["params"]["amp"] < 0.4])))[0]   # (0.5 and 1.0 are also constants of the waveform)
            j = int(rows_c[rng.integers(len(rows_c))])       # (an amplitude: the voices stay template voices)
            new = f32_bits(np.float32(0.25 + 0.01 * rnd))
            for r in (hip, ref):
                r.on_del_edge(synth.CONST_HANDLE, int(e[j, 1]), int(e[j, 2]), 0)
                r.on_add_edge(synth.CONST_HANDLE, int(e[j, 1]), new, 0)
            e[j, 2] = new
            with pytest.raises(RenderError):
                hip.stream_block(idx, synth.time_ramp(idx, idx + 8))
            # ordinary calls still work after a stream (the first one is a seek for the engine: nothing of the stream was stored)
            row = synth.time_ramp(idx, idx + 100)
            assert same_bits(hip.fill_buffer(V, idx, idx + 100, [row]), ref.fill_buffer(V, idx, idx + 100, [row]))
            idx += 100
        hip.stream_begin(V)
        hip.stream_end()


@pytest.mark.gpu
def test_block_streaming_refuses_what_it_cannot_serve(hip_lib):
    with Renderer(hip_lib) as hip:
        synth.install(hip, synth.effects_tree(2, 64))      # delay taps: state between calls
        with pytest.raises(RenderError) as ei:
            hip.stream_begin(2)
        assert ei.value.status == 10                        # FR_ERR_UNSUPPORTED
        with pytest.raises(RenderError):
            hip.stream_block(0, synth.time_ramp(0, 8))      # no stream open
    with Renderer(hip_lib) as hip:
        synth.install(hip, synth.additive_tree(2, 64))       # voices too small for 16 waves
        with pytest.raises(RenderError) as ei:
            hip.stream_begin(2)
        assert ei.value.status == 10


@pytest.mark.gpu
def test_block_streaming_launch_ends_itself_without_a_host(hip_lib, oracle_lib, monkeypatch):
    """Nobody rings for FR_STREAM_IDLE_MS of wall clock (2000 ms by default; 300 here): the resident launch ends itself -- a
    host that died leaves no spinning GPU behind; the next block is refused with FR_ERR_DEVICE, a new stream renders on.
    The launch may have ended between two chunks of a voice, so the arrival counters the short-call kernel shares with it
    are cleared before their next use: a short fill_buffer call right after the failure equals the oracle."""
    import time
    monkeypatch.setenv("FR_STREAM_IDLE_MS", "300")      # (read when the renderer is created)
    V, P = 3, 4096                                      # 3 voices x 64 chunks of 64 partials: chunked, tickets in use
    tree = synth.additive_tree(V, P)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        hip.stream_begin(V)
        row = synth.time_ramp(0, 64)
        assert same_bits(hip.stream_block(0, row), ref.fill_buffer(V, 0, 64, [row]))
        t0 = time.monotonic()
        time.sleep(1.0)
        with pytest.raises(RenderError) as ei:
            hip.stream_block(64, synth.time_ramp(64, 128))
        assert ei.value.status == 7                     # FR_ERR_DEVICE: the launch had ended
        assert time.monotonic() - t0 < 3.0
        for k, T in enumerate((64, 200, 17)):           # short calls through the chunked kernel and its tickets
            row = synth.time_ramp(1000 * k, 1000 * k + T)
            got, exp = hip.fill_buffer(V, 1000 * k, 1000 * k + T, [row]), ref.fill_buffer(V, 1000 * k, 1000 * k + T, [row])
            assert same_bits(got, exp), f"call {k} after the failed stream: " + first_diff(got, exp)
        hip.stream_begin(V)
        row = synth.time_ramp(64, 128)
        assert same_bits(hip.stream_block(64, row), ref.fill_buffer(V, 64, 128, [row]))


@pytest.mark.gpu
def test_block_streaming_short_rows_and_buffer_checks(hip_lib, oracle_lib):
    """A block whose row is shorter than the block, or empty, is padded as fill_buffer pads it (reference.rs:72-73): with its
    own last value, or with the last value of the block it continues; the first block of a stream and one that does not
    continue the previous block pad with 0 (nothing stored, as after a seek).  The Python wrapper refuses an `out` of the
    wrong shape (the C call takes no slot count)."""
    V, P = 4, 512
    tree = synth.additive_tree(V, P)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        hip.stream_begin(V)

        def block(idx, n, row):
            out = np.empty((V, n), np.float32)
            row = np.ascontiguousarray(row, np.float32)
            hip._check(hip.L.fr_stream_block(hip.h, out.ctypes.data, n, idx, row.ctypes.data if len(row) else None, len(row)))
            return out

        # the oracle makes the same calls, begun with a seek (idx 5000 != head 0)
        seq = [(5000, 40, synth.time_ramp(5000, 5040)), (5040, 64, synth.time_ramp(5040, 5050)), (5104, 32, np.zeros(0, np.float32)),
               (5136, 64, synth.time_ramp(5136, 5200)), (9000, 16, np.zeros(0, np.float32)), (9016, 8, synth.time_ramp(9016, 9019))]
        for k, (idx, n, row) in enumerate(seq):
            got, exp = block(idx, n, row), ref.fill_buffer(V, idx, idx + n, [row])
            assert same_bits(got, exp), f"block {k}: " + first_diff(got, exp)
        with pytest.raises(ValueError):
            hip.stream_block(9024, synth.time_ramp(9024, 9032), out=np.empty((V - 1, 8), np.float32))
        with pytest.raises(ValueError):
            hip.stream_block(9024, synth.time_ramp(9024, 9032), out=np.empty((V, 9), np.float32))
        hip.stream_end()


# ---- feedback through Delay (the reference as written: routegraph.rs:218-237 never refuses the edge, reference.rs:197-216
# ---- evaluates the loop by recursion, which ends because every trip round passes a Delay of >= 1 frames) -----------------------
def _fb_nodes(rs, nodes, edges, n_out_edges):
    """nodes: {handle: kind}; edges: (from, to, from_slot_or_const_bits, to_slot) with from == 'c' for the constant node (handle 1)."""
    for r in rs:
        r.on_add_node(1, "F32Constant")
        for h, kind in nodes.items():
            r.on_add_node(h, kind)
        for frm, to, fs, ts in edges + n_out_edges:
            r.on_add_edge(1 if frm == "c" else frm, to, f32_bits(fs) if frm == "c" else fs, ts)


def _fb_calls(hip, ref, n_out, seq, n_in=1, seed=0):
    rng = np.random.default_rng(seed)
    for idx, n in seq:
        rows = [rng.normal(size=n).astype(np.float32) for _ in range(n_in)]
        got, exp = hip.fill_buffer(n_out, idx, idx + n, rows), ref.fill_buffer(n_out, idx, idx + n, rows)
        assert same_bits(got, exp), f"call at {idx} (+{n}): " + first_diff(got, exp)


@pytest.mark.parametrize("d", [1, 3, 64, 100])
def test_feedback_echo(hip_lib, oracle_lib, d):
    """x = in0 + 0.5 * Delay(x, d): contiguous calls, a call longer than the delay, a seek forward (the loop's state is rebuilt
    by replaying the frames from 0), an edit of the gain between calls (all of history is heard through the new gain), a seek
    back.  Bit-exact against the oracle's recursion."""
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        _fb_nodes((hip, ref), {2: "Sum2", 3: "Delay", 4: "Multiply"},
                  [(0, 2, 0, 0), (4, 2, 0, 1), (2, 3, 0, 0), ("c", 3, float(d), 1), (3, 4, 0, 0), ("c", 4, 0.5, 1)], [(2, 0, 0, 0)])
        _fb_calls(hip, ref, 1, [(0, 50), (50, 70), (120, 200), (1000, 64)], seed=d)
        plan = hip.plan()
        assert plan["feedback"] and plan["feedback_loops"] == 1 and plan["fused_stride"] == d and plan["pull_rows"] == 0, plan
        for r in (hip, ref):
            r.on_del_edge(1, 4, f32_bits(0.5), 1)
            r.on_add_edge(1, 4, f32_bits(-0.25), 1)
        _fb_calls(hip, ref, 1, [(1064, 64), (1128, 8), (10, 40)], seed=d + 1)
        if d >= 64:   # a steady call longer than anything the rings were sized for: they are re-allocated, the loop's state replayed
            _fb_calls(hip, ref, 1, [(50, 40000)], seed=d + 2)


def test_feedback_loop_with_rows_inside_and_a_tap_behind(hip_lib, oracle_lib):
    """m = x * g, x = in0 + Delay(m, 5): the Delay's source is not the row's root.  Rows: x (computed inline by m's program:
    copied from its ring after the launch), m, and Delay(x, 2) + in1 (another program reading x's ring: a later level)."""
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        _fb_nodes((hip, ref), {2: "Sum2", 3: "Delay", 4: "Multiply", 5: "Delay", 6: "Sum2"},
                  [(0, 2, 0, 0), (3, 2, 0, 1), (4, 3, 0, 0), ("c", 3, 5.0, 1), (2, 4, 0, 0), ("c", 4, 0.75, 1),
                   (2, 5, 0, 0), ("c", 5, 2.0, 1), (5, 6, 0, 0), (0, 6, 1, 1)],
                  [(2, 0, 0, 0), (4, 0, 0, 1), (6, 0, 0, 2)])
        _fb_calls(hip, ref, 3, [(0, 33), (33, 64), (97, 3), (400, 50), (450, 50)], n_in=2, seed=5)
        plan = hip.plan()
        assert plan["feedback"] and plan["copy_programs"] >= 1 and plan["fused_levels"] == 2, plan


def test_feedback_two_taps_and_nested_loops(hip_lib, oracle_lib):
    """Row 0: x = in0 + 0.5 Delay(x, 6) + 0.25 Delay(x, 9) (threads stride by gcd 3).  Row 1: y = x2 + 0.3 Delay(y, 4) over an
    inner loop x2 = in0 + 0.5 Delay(x2, 2)."""
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        _fb_nodes((hip, ref), {2: "Sum2", 3: "Sum2", 4: "Delay", 5: "Multiply", 6: "Delay", 7: "Multiply",
                               12: "Sum2", 13: "Delay", 14: "Multiply", 15: "Sum2", 16: "Delay", 17: "Multiply"},
                  [(0, 2, 0, 0), (3, 2, 0, 1), (5, 3, 0, 0), (7, 3, 0, 1), (2, 4, 0, 0), ("c", 4, 6.0, 1), (4, 5, 0, 0), ("c", 5, 0.5, 1),
                   (2, 6, 0, 0), ("c", 6, 9.0, 1), (6, 7, 0, 0), ("c", 7, 0.25, 1),
                   (0, 12, 0, 0), (14, 12, 0, 1), (12, 13, 0, 0), ("c", 13, 2.0, 1), (13, 14, 0, 0), ("c", 14, 0.5, 1),
                   (12, 15, 0, 0), (17, 15, 0, 1), (15, 16, 0, 0), ("c", 16, 4.0, 1), (16, 17, 0, 0), ("c", 17, 0.3, 1)],
                  [(2, 0, 0, 0), (15, 0, 0, 1)])
        _fb_calls(hip, ref, 2, [(0, 20), (20, 30), (50, 14), (40, 24)], seed=7)   # (the two-tap recursion branches: short)
        plan = hip.plan()
        assert plan["feedback"] and plan["feedback_loops"] == 4 and plan["fused_stride"] == 1, plan


def test_feedback_loop_through_two_delayed_nodes_is_one_program(hip_lib, oracle_lib):
    """x = in0 + Delay(y, 2), y = 0.9 * Delay(x, 3): neither node uses the other at the same frame, so each would be its own
    program reading the other's ring -- they are merged into one so that a thread computes both, frame by frame."""
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        _fb_nodes((hip, ref), {2: "Sum2", 3: "Delay", 4: "Multiply", 5: "Delay"},
                  [(0, 2, 0, 0), (3, 2, 0, 1), (4, 3, 0, 0), ("c", 3, 2.0, 1), (5, 4, 0, 0), ("c", 4, 0.9, 1), (2, 5, 0, 0), ("c", 5, 3.0, 1)],
                  [(2, 0, 0, 0), (4, 0, 0, 1)])
        _fb_calls(hip, ref, 2, [(0, 40), (40, 100), (300, 30)], seed=9)
        plan = hip.plan()
        assert plan["feedback"] and plan["fused_programs"] == 1 and plan["copy_programs"] == 1, plan


def test_feedback_around_a_bank_voice(hip_lib, oracle_lib):
    """An additive voice (rendered by the bank kernel into a ring) feeding a comb filter: x = voice + 0.6 * Delay(x, 7); the
    replay after a seek re-renders the voice's ring chunk by chunk."""
    g = synth.GraphArrays()
    p = synth.voice_params(1, 64, seed=11)
    voice = synth.sum_tree(g, synth.partial_leaves(g, p["w"], p["amp"]).reshape(1, 64))
    x = g.nodes(synth.K_SUM2, 1)
    d = g.binop(synth.K_DELAY, x, synth.C(np.float32(7.0)), 1)
    m = g.binop(synth.K_MUL, d, synth.C(np.float32(0.6)), 1)
    g.edge(voice, x, 0, 0)
    g.edge(m, x, 0, 1)
    g.edge(x, 0, 0, 0)
    tree = g.finish(1)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        synth.install(hip, tree)
        synth.install(ref, tree)
        for idx, n in [(0, 48), (48, 80), (128, 16), (500, 32)]:
            t = synth.time_ramp(idx, idx + n)
            got, exp = hip.fill_buffer(1, idx, idx + n, [t]), ref.fill_buffer(1, idx, idx + n, [t])
            assert same_bits(got, exp), f"call at {idx}: " + first_diff(got, exp)
        plan = hip.plan()
        assert plan["feedback"] and len(plan["banks"]) == 1 and plan["banks"][0]["to_ring"], plan


def test_feedback_that_cannot_be_evaluated_is_refused(hip_lib):
    """FR_ERR_CYCLE only where the reference's recursion would never end: a cycle with no Delay on it, with a Delay of less
    than one frame, of a signal amount, or entered through a Delay's AMOUNT.  FR_ERR_UNSUPPORTED where it would end but this
    engine has no evaluator for it: FR_MODE_PULL, a bounded input history."""
    def loop(r, amount_edges):
        r.on_add_node(1, "F32Constant")
        r.on_add_node(2, "Sum2")
        r.on_add_node(3, "Delay")
        r.on_add_edge(0, 2, 0, 0)
        r.on_add_edge(3, 2, 0, 1)
        r.on_add_edge(2, 3, 0, 0)
        for e in amount_edges:
            r.on_add_edge(*e)
        r.on_add_edge(2, 0, 0, 0)

    for amount in ([(1, 3, f32_bits(0.5), 1)], [(1, 3, f32_bits(-3.0), 1)], [(1, 3, f32_bits(float("nan")), 1)], [], [(0, 3, 1, 1)]):
        with Renderer(hip_lib) as r:
            loop(r, amount)
            with pytest.raises(RenderError) as ei:
                r.fill_buffer(1, 0, 8, [np.ones(8, np.float32), np.full(8, 2.0, np.float32)])
            assert ei.value.status == FR_ERR_CYCLE, amount
    with Renderer(hip_lib) as r:   # the cycle runs through the Delay's amount, its source is acyclic
        r.on_add_node(1, "F32Constant")
        r.on_add_node(2, "Sum2")
        r.on_add_node(3, "Delay")
        r.on_add_edge(0, 2, 0, 0)
        r.on_add_edge(3, 2, 0, 1)
        r.on_add_edge(0, 3, 0, 0)
        r.on_add_edge(2, 3, 0, 1)
        r.on_add_edge(2, 0, 0, 0)
        with pytest.raises(RenderError) as ei:
            r.fill_buffer(1, 0, 8, [np.ones(8, np.float32)])
        assert ei.value.status == FR_ERR_CYCLE
    for kw in ({"mode": "pull"}, {"history_frames": 100}):
        with Renderer(hip_lib, **kw) as r:
            loop(r, [(1, 3, f32_bits(4.0), 1)])
            with pytest.raises(RenderError) as ei:
                r.fill_buffer(1, 0, 8, [np.ones(8, np.float32)])
            assert ei.value.status == FR_ERR_UNSUPPORTED, kw


@pytest.mark.timeout(600)   # (the oracle's recursion is exponential in what a bad estimate lets through)
@pytest.mark.parametrize("seed0", [0, 40])
def test_random_feedback_graphs(hip_lib, oracle_lib, seed0):
    """Random graphs of the seven primitives with one Delay re-pointed at a node that depends on it (randgraph.py), two
    contiguous calls and an edit-free seek back, against the oracle.  Graphs this engine has no evaluator for (the loop
    reaches a row the pull interpreter must take) are passed over -- and counted."""
    done = unsupported = 0
    for seed in range(seed0, seed0 + 40):
        made = randgraph.random_feedback_graph(seed)
        if made is None:
            continue
        steps, n_out, d = made
        with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
            randgraph.install_steps(hip, steps)
            randgraph.install_steps(ref, steps)
            rng = np.random.default_rng(seed)
            try:
                for idx, n in [(0, 9), (9, 7), (3, 5)]:
                    rows = [rng.normal(size=n).astype(np.float32) * 3, rng.integers(-2, 6, size=n).astype(np.float32)]
                    got = hip.fill_buffer(n_out, idx, idx + n, rows)
                    exp = ref.fill_buffer(n_out, idx, idx + n, rows)
                    assert same_bits(got, exp), f"seed {seed} (delay {d}) call at {idx}: " + first_diff(got, exp)
            except RenderError as e:
                assert e.status == FR_ERR_UNSUPPORTED, (seed, e)
                unsupported += 1
                continue
            assert hip.plan()["feedback"], seed
            done += 1
    assert done >= 15 and unsupported <= done // 2, (done, unsupported)


# ---- control-rate tracks: per-partial w / amp rows read in place from the call's dense input matrix (the HBM-bound variant) --
@pytest.mark.parametrize("V,P", [(3, 64), (2, 256), (5, 32)])
def test_track_voices_against_oracle(hip_lib, oracle_lib, V, P):
    """Voices whose partials take their phase increment and amplitude from input rows (synth.track_tree): the engine reads the
    rows in place (fr_set_track_inputs), the oracle stores them like any input.  Host matrix, device matrix and the CSR call;
    a short first call on a fresh renderer, where the reference has only n_slots * n_times input vectors and drops the rows
    beyond them (reference.rs:59-68) -- those partials are silent in both."""
    import torch
    tree = synth.track_tree(V, P)
    R = tree["n_inputs"]
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        hip.set_track_inputs(tree["first_track"])
        synth.install(hip, tree)
        synth.install(ref, tree)
        idx = 0
        for k, n in enumerate([max(200, R // V + 1), 130, 16, 1, 64]):
            m = synth.track_rows(V, P, idx, idx + n)
            exp = ref.fill_buffer_dense(V, idx, idx + n, m)
            if k % 3 == 0:
                got = hip.fill_buffer_dense(V, idx, idx + n, m)
            elif k % 3 == 1:
                got = hip.fill_buffer(V, idx, idx + n, list(m))
            else:
                d_m = torch.from_numpy(m).cuda()
                d_out = torch.empty((V, n), dtype=torch.float32, device="cuda")
                hip.fill_buffer_device_dense(d_out.data_ptr(), V, n, idx, d_m.data_ptr(), R, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                got = d_out.cpu().numpy()
            assert same_bits(got, exp), f"call {k} ({n} frames): " + first_diff(got, exp)
            assert np.abs(got).max() > 0
            idx += n
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and len(plan["banks"]) == 1 and plan["banks"][0]["jit"] and plan["banks"][0]["tracks"], plan
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:   # the reference's input vectors: V * 16 of them, the rest dropped
        hip.set_track_inputs(tree["first_track"])
        synth.install(hip, tree)
        synth.install(ref, tree)
        m = synth.track_rows(V, P, 0, 16)
        got, exp = hip.fill_buffer_dense(V, 0, 16, m), ref.fill_buffer_dense(V, 0, 16, m)
        assert same_bits(got, exp), first_diff(got, exp)
        assert not got[V - 1].any()    # the last voice's tracks lie beyond slot V * 16


def test_tracks_read_by_anything_else_are_refused(hip_lib):
    """A track is visible only to the leaves of the voices of the call that supplies it: a Delay of one, a program, a template
    voice's time input -> FR_ERR_UNSUPPORTED (the reference would serve them from the stored history)."""
    tree = synth.track_tree(2, 32)
    for extra in ("delay", "program"):
        with Renderer(hip_lib) as hip:
            hip.set_track_inputs(tree["first_track"])
            synth.install(hip, tree)
            h = int(tree["handles"].max()) + 1
            hip.on_add_node(h, "Delay" if extra == "delay" else "Sum2")
            hip.on_add_edge(0, h, 5, 0)                      # input slot 5: a track
            hip.on_add_edge(1, h, f32_bits(3.0), 1)
            hip.on_add_edge(h, 0, 0, 2)
            with pytest.raises(RenderError) as ei:
                hip.fill_buffer_dense(3, 0, 64, synth.track_rows(2, 32, 0, 64))
            assert ei.value.status == FR_ERR_UNSUPPORTED, extra


def test_track_row_shared_by_every_leaf(hip_lib, oracle_lib):
    """Every partial reads its own w track but ONE common amplitude track (a master envelope as a control-rate row): the slot
    number of the shared row is a literal of the generated leaf, not a per-leaf parameter."""
    V, P, first = 2, 64, 3
    g = synth.GraphArrays()
    n = V * P
    w_slots = first + 1 + np.arange(n, dtype=np.uint32)
    leaves = synth.track_leaves(g, w_slots, np.full(n, first, dtype=np.uint32), time_slot=0).reshape(V, P)
    roots = synth.sum_tree(g, leaves)
    g.edge(roots, 0, 0, np.arange(V, dtype=np.uint32))
    tree = g.finish(V)
    R = first + 1 + n
    rng = np.random.default_rng(12)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        hip.set_track_inputs(first)
        synth.install(hip, tree)
        synth.install(ref, tree)
        idx = 0
        for T in (R // V + 8, 50):
            m = np.zeros((R, T), np.float32)
            m[0] = synth.time_ramp(idx, idx + T)
            m[1:first] = rng.normal(size=(first - 1, T)).astype(np.float32)          # stored inputs nobody reads
            m[first] = np.linspace(1.0, 0.2, T, dtype=np.float32)                     # the master envelope
            m[first + 1:] = (rng.random((n, T)) * 0.05).astype(np.float32)
            got, exp = hip.fill_buffer_dense(V, idx, idx + T, m), ref.fill_buffer_dense(V, idx, idx + T, m)
            assert same_bits(got, exp), first_diff(got, exp)
            assert np.abs(got).max() > 0
            idx += T
        bank = hip.plan()["banks"][0]
        assert bank["tracks"] and bank["leaf_params"] == 1, bank


def test_few_voice_calls_pipelined_on_two_streams(hip_lib):
    """A host that renders ahead on alternating streams (a rank of a voice-sharded job: 8 x 4096 x 4800 per call): from the second
    call on the launches are whole (voice, tile) pairs without shared scratch, so consecutive calls overlap on the device; every
    call's buffer must hold the same bits as the same frames rendered by blocking calls on one stream."""
    import torch
    V, P, T, calls = 8, 4096, 4800, 8
    tree = synth.additive_tree(V, P, seed=21, detune=True)
    t = synth.time_ramp(0, calls * T)
    with Renderer(hip_lib) as a, Renderer(hip_lib) as b:
        synth.install(a, tree)
        synth.install(b, tree)
        exp = [b.fill_buffer(V, k * T, (k + 1) * T, [t[k * T:(k + 1) * T]]) for k in range(calls)]
        d_t = torch.from_numpy(t).cuda()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.empty((V, T), dtype=torch.float32, device="cuda") for _ in range(calls)]
        torch.cuda.synchronize()
        for k in range(calls):
            a.fill_buffer_device(outs[k].data_ptr(), V, T, k * T, d_t[k * T:].data_ptr(), [0, T], streams[k % 2].cuda_stream)
        torch.cuda.synchronize()
        for k in range(calls):
            got = outs[k].cpu().numpy()
            assert same_bits(got, exp[k]), f"call {k}: " + first_diff(got, exp[k])
        # back on one stream: the chunked launch again, still the same bits
        for k in range(2):
            a.fill_buffer_device(outs[k].data_ptr(), V, T, (calls + k) * T, d_t[k * T:].data_ptr(), [0, T], streams[0].cuda_stream)
        torch.cuda.synchronize()
        assert same_bits(outs[1].cpu().numpy(), exp[1])


def test_feedback_inside_composite_instances(hip_lib, oracle_lib):
    """A composite effect whose own graph holds the loop (an echo: out = x = in + 0.5 * Delay(x, 2)), instantiated twice in series:
    each instance context gets its own cut (two OP_FBREF leaves), rows on the inner and the outer instance."""
    nodes = [(1000, Effect.primitive("F32Constant")), (1, Effect.primitive("Sum2")), (2, Effect.primitive("Delay")), (3, Effect.primitive("Multiply"))]
    edges = [(0, 1, 0, 0), (3, 1, 0, 1), (1, 2, 0, 0), (1000, 2, f32_bits(2.0), 1), (2, 3, 0, 0), (1000, 3, f32_bits(0.5), 1), (1, 0, 0, 0)]
    echo = Effect.graph(nodes, edges)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        for r in (hip, ref):
            r.on_add_node(5, echo)
            r.on_add_node(6, echo)
            r.on_add_edge(0, 5, 0, 0)
            r.on_add_edge(5, 6, 0, 0)
            r.on_add_edge(6, 0, 0, 0)
            r.on_add_edge(5, 0, 0, 1)
        _fb_calls(hip, ref, 2, [(0, 20), (20, 33), (100, 10), (7, 5)], seed=31)
        plan = hip.plan()
        assert plan["feedback"] and plan["feedback_loops"] == 2, plan


@pytest.mark.timeout(600)   # (the oracle's recursion is exponential in what a bad estimate lets through)
@pytest.mark.parametrize("seed0", [0, 60])
def test_feedback_graphs_edited_during_playback(hip_lib, oracle_lib, seed0):
    """Graph edits between calls on graphs with feedback loops (constants, loop delays, outputs re-pointed, new nodes): the lowering
    is incremental -- an edit anywhere on a loop invalidates the Delay that cuts it, and the loop is cut again -- the loop's state is
    rebuilt by replay with the NEW graph (everything heard so far through the edited graph), and a second engine
    that is told the final graph from scratch renders the same bits."""
    done = incremental = 0
    for seed in range(seed0, seed0 + 60):
        made = randgraph.random_feedback_graph(seed, n_frames=20, budget=2e4)
        if made is None:
            continue
        steps, n_out, _d = made
        steps = list(steps)
        rng = np.random.default_rng(seed + 5)
        with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
            randgraph.install_steps(hip, steps)
            randgraph.install_steps(ref, steps)
            idx = 0
            try:
                for k in range(4):
                    n = int(rng.integers(3, 7))
                    rows = [rng.normal(size=n).astype(np.float32), rng.integers(-2, 5, size=n).astype(np.float32)]
                    got = hip.fill_buffer(n_out, idx, idx + n, rows)
                    exp = ref.fill_buffer(n_out, idx, idx + n, rows)
                    assert same_bits(got, exp), f"seed {seed}, call {k}: " + first_diff(got, exp)
                    if k:
                        incremental += hip.plan()["lowering"] == "incremental"
                    idx += n
                    edits = randgraph.safe_feedback_edits(rng, steps, 2)
                    randgraph.install_steps(hip, edits)
                    randgraph.install_steps(ref, edits)
            except RenderError as e:
                assert e.status == FR_ERR_UNSUPPORTED, (seed, e)
                continue
        done += 1
    assert done >= 15 and incremental >= done, (done, incremental)


def test_track_voices_under_voice_sharding(hip_lib, oracle_lib):
    """Two ranks (two renderers sharing the test GPU) render a job whose voices read per-partial track rows: every rank gets the
    same graph and the same dense input matrix, declares the same track slots and renders its block of voices -- no exchange."""
    import shard_harness
    V, P = 5, 64
    tree = synth.track_tree(V, P)
    R = tree["n_inputs"]
    job = shard_harness.Job(hip_lib, 2, "voices")
    with Renderer(oracle_lib) as ref:
        synth.install(ref, tree)
        for ren in job.ranks:
            ren.set_track_inputs(tree["first_track"])
            synth.install(ren, tree)
        idx = 0
        for n in (R // V + 4, 70):
            m = synth.track_rows(V, P, idx, idx + n)
            exp = ref.fill_buffer_dense(V, idx, idx + n, m)

            def one(_r, ren):
                out = np.full((V, n), np.float32(-12345.0), dtype=np.float32)
                return ren.fill_buffer_dense(V, idx, idx + n, m, out=out)
            got = job.assemble(job.each(one), V)
            assert same_bits(got, exp), first_diff(got, exp)
            idx += n
    assert sum(job.boxes.messages) == 0
    job.close()


def test_track_voices_with_the_abi_default_asynchronous_compile(hip_lib, oracle_lib):
    """fr_config.flags = 0 (the C ABI's default) compiles run-time kernels on a worker thread and renders on the generic evaluators
    meanwhile -- which cannot read tracks: for voices that read tracks the first call waits for the compiler instead of failing."""
    V, P = 2, 128
    tree = synth.track_tree(V, P)
    R = tree["n_inputs"]
    with Renderer(hip_lib, sync_compile=False) as hip, Renderer(oracle_lib) as ref:
        hip.set_track_inputs(tree["first_track"])
        synth.install(hip, tree)
        synth.install(ref, tree)
        n = R // V + 3
        m = synth.track_rows(V, P, 0, n)
        got, exp = hip.fill_buffer_dense(V, 0, n, m), ref.fill_buffer_dense(V, 0, n, m)
        assert same_bits(got, exp), first_diff(got, exp)
        assert hip.plan()["banks"][0]["tracks"]


def test_feedback_while_the_stage_kernel_is_compiled_in_the_background(hip_lib, oracle_lib):
    """Sixteen rows, each an echo loop of the same shape, under the ABI's default asynchronous compile: the first calls run the loops on
    the interpreter, a later one on the compiled stage kernel (a new plan: the loops' state is rebuilt by replay); same bits
    throughout."""
    import time
    with Renderer(hip_lib, sync_compile=False) as hip, Renderer(oracle_lib) as ref:
        for r in (hip, ref):
            r.on_add_node(1, "F32Constant")
            for v in range(16):
                b = 10 * (v + 1)
                r.on_add_node(b, "Multiply")        # in0 * c_v
                r.on_add_node(b + 1, "Sum2")        # x_v
                r.on_add_node(b + 2, "Delay")
                r.on_add_node(b + 3, "Multiply")
                r.on_add_edge(0, b, 0, 0)
                r.on_add_edge(1, b, f32_bits(0.31 + 0.1 * v), 1)
                r.on_add_edge(b, b + 1, 0, 0)
                r.on_add_edge(b + 3, b + 1, 0, 1)
                r.on_add_edge(b + 1, b + 2, 0, 0)
                r.on_add_edge(1, b + 2, f32_bits(3.0), 1)
                r.on_add_edge(b + 2, b + 3, 0, 0)
                r.on_add_edge(1, b + 3, f32_bits(0.5), 1)
                r.on_add_edge(b + 1, 0, 0, v)
        rng = np.random.default_rng(17)
        idx, compiled = 0, False
        for k in range(14):
            n = 30
            rows = [rng.normal(size=n).astype(np.float32)]
            got, exp = hip.fill_buffer(16, idx, idx + n, rows), ref.fill_buffer(16, idx, idx + n, rows)
            assert same_bits(got, exp), f"call {k}: " + first_diff(got, exp)
            compiled = compiled or hip.plan()["stage_jit"]
            idx += n
            time.sleep(0.05)
        assert hip.plan()["feedback"] and compiled, hip.plan()
