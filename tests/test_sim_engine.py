"""The engine's HOST logic on the CPU: the same parity cases the GPU suite runs (tests/test_hip_parity.py), driven
through the C ABI of the host-logic simulator (tests/sim_tools.py: the engine's own host sources + plain-loop kernels)
and compared bit-for-bit with the oracle.  This checks the input store, lowering, planning, ring/window handling and
error behaviour before any kernel runs; the kernels themselves are checked by `-m gpu`."""
import numpy as np
import pytest

import sim_tools
import test_hip_parity as G
from kat_replay import same_bits
from libfriendship_amd import synth
from libfriendship_amd.capi import FR_ERR_CYCLE, FR_ERR_INPUT_HISTORY, RenderError, Renderer, f32_bits


@pytest.fixture(scope="module")
def sim():
    return sim_tools.sim_lib()


@pytest.mark.parametrize("mode", ["auto", "pull"])
@pytest.mark.parametrize("i", range(11))
def test_reference_kat(sim, kat, i, mode):
    G.test_reference_kat_on_hip(sim, kat, i, mode)


@pytest.mark.parametrize("mode", ["auto", "staged"])
@pytest.mark.parametrize("i", range(14))
def test_selfcheck_vectors(sim, selfcheck, i, mode):
    G.test_selfcheck_vectors_on_hip(sim, selfcheck, i, mode)


@pytest.mark.parametrize("mode", ["pull", "auto"])
@pytest.mark.parametrize("seed", range(0, 24, 2))
def test_random_graphs(sim, oracle_lib, seed, mode):
    G.test_random_graphs(sim, oracle_lib, seed, mode)


@pytest.mark.parametrize("seed", range(1, 24, 2))
def test_random_graphs_staged_mode(sim, oracle_lib, seed):
    G.test_random_graphs_staged_mode(sim, oracle_lib, seed)


@pytest.mark.parametrize("V,P,taps,delay,T", [(3, 64, 3, 50.0, 128), (2, 32, 4, 7.0, 33)])
def test_effects_chain_staged(sim, oracle_lib, V, P, taps, delay, T):
    G.test_effects_chain_staged(sim, oracle_lib, V, P, taps, delay, T)


@pytest.mark.parametrize("mode", ["auto", "staged", "pull"])
@pytest.mark.parametrize("seed", range(4))
def test_random_edits_between_calls(sim, oracle_lib, seed, mode):
    G.test_random_edits_between_calls(sim, oracle_lib, seed, mode)


def test_error_codes(sim):
    G.test_hip_error_codes(sim)


def test_graph_edit_rebuilds_delay_state(sim, oracle_lib):
    G.test_graph_edit_rebuilds_delay_state(sim, oracle_lib)


def test_fused_stage_mode_equals_level_mode(sim, oracle_lib):
    G.test_fused_stage_mode_equals_level_mode(sim, oracle_lib)


@pytest.mark.parametrize("V,P,T", [(2, 100, 70), (1, 24, 64), (2, 17, 70)])
def test_bank_general_partial_counts(sim, oracle_lib, V, P, T):
    G.test_bank_general_partial_counts(sim, oracle_lib, V, P, T)


def test_chorus_signal_delay_is_staged(sim, oracle_lib):
    G.test_chorus_signal_delay_is_staged(sim, oracle_lib, 3, 32, 100, 2)


def _launches(sim, cls):
    import ctypes
    sim.lib.fr_sim_launch_count.restype = ctypes.c_uint64
    return sim.lib.fr_sim_launch_count(cls)


def test_shared_root_on_two_rows_is_one_launch_per_call(sim, oracle_lib):
    """A non-bank root wired to two output rows (a master gain sent to L and R), no delayed read anywhere: the fused
    steady-state form has no frame limit.  The sub-window count used to wrap and the engine issued idx + 1 launches per
    contiguous call."""
    STAGE = 2
    with Renderer(sim) as r, Renderer(oracle_lib) as ref:
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
        for k in range(4):
            rows = [synth.time_ramp(k * T, (k + 1) * T), rng.normal(size=T).astype(np.float32)]
            before = _launches(sim, STAGE)
            got = r.fill_buffer(2, k * T, (k + 1) * T, rows)
            n = _launches(sim, STAGE) - before
            assert same_bits(got, ref.fill_buffer(2, k * T, (k + 1) * T, rows))
            plan = r.plan()
            assert plan["fused_programs"] > 0 and plan["fused_max_frames"] >= T, plan
            if k > 0:
                assert n == 1, f"call {k} at idx {k * T}: {n} stage launches"


def test_failed_call_leaves_the_input_store_intact(sim, oracle_lib):
    """A call that fails AFTER validation of its rows (here: the graph has a cycle, found when the plan is built) must not
    commit the rows: once the graph is fixed, the retry at the same idx succeeds and sees the history of the calls
    before it."""
    with Renderer(sim) as r, Renderer(oracle_lib) as ref:
        for x in (r, ref):
            x.on_add_node(1, "F32Constant")
            x.on_add_node(2, "Delay")      # out = in0 delayed by 3
            x.on_add_node(3, "Sum2")
            x.on_add_edge(0, 2, 0, 0)
            x.on_add_edge(1, 2, f32_bits(3.0), 1)
            x.on_add_edge(2, 3, 0, 0)
            x.on_add_edge(0, 3, 0, 1)
            x.on_add_edge(3, 0, 0, 0)
        a = np.arange(8, dtype=np.float32) + 1
        assert same_bits(r.fill_buffer(1, 0, 8, [a]), ref.fill_buffer(1, 0, 8, [a]))
        # break the graph: 3 -> 4 -> 3
        r.on_add_node(4, "Multiply")
        r.on_add_edge(3, 4, 0, 0)
        r.on_add_edge(4, 3, 0, 1)
        b = np.arange(8, dtype=np.float32) + 100
        with pytest.raises(RenderError) as ei:
            r.fill_buffer(1, 8, 16, [b])
        assert ei.value.status == FR_ERR_CYCLE
        with pytest.raises(RenderError):           # still broken: same answer, still nothing committed
            r.fill_buffer(1, 8, 16, [b])
        r.on_add_edge(0, 3, 0, 1)                  # repair
        got = r.fill_buffer(1, 8, 16, [b])         # the retry at the same idx: no FR_ERR_INPUT_HISTORY
        assert same_bits(got, ref.fill_buffer(1, 8, 16, [b]))
        # a refused row (too long) leaves the store intact as well
        with pytest.raises(RenderError):
            r.fill_buffer(1, 16, 20, [np.zeros(9, np.float32)])
        c = np.arange(4, dtype=np.float32)
        assert same_bits(r.fill_buffer(1, 16, 20, [c]), ref.fill_buffer(1, 16, 20, [c]))
        # a row that does not continue its slot's history is refused before anything changes: slot 1 was created by the
        # first call (at idx 0) and never fed, so it still holds 0 samples (reference.rs:60-69)
        # (the reference panics half-way through its rows there; only the engine promises an untouched store)
        with pytest.raises(RenderError) as ei:
            r.fill_buffer(1, 20, 24, [c, c])
        assert ei.value.status == FR_ERR_INPUT_HISTORY
        assert same_bits(r.fill_buffer(1, 20, 24, [c]), ref.fill_buffer(1, 20, 24, [c]))


@pytest.mark.parametrize("detune", [False, True])
def test_harmonics_and_detune_as_graph_nodes(sim, oracle_lib, detune):
    """N3 / N4: harmonics f0*(k+1), detune *(1+delta) and /sr reach the renderer as Multiply / Divide nodes
    over constants.  Lowering folds them with exactly-rounded f32 ops, the voices are still recognised as banks, and the
    result equals both the oracle on the same node graph and the numpy-folded form of the tree."""
    V, P, T = 3, 64, 80
    nodes = synth.additive_tree(V, P, seed=5, detune=detune, params_as_nodes=True)
    folded = synth.additive_tree(V, P, seed=5, detune=detune)
    assert len(nodes["handles"]) == len(folded["handles"]) + V * P * (3 if detune else 2)
    t = synth.time_ramp(100, 100 + T)
    with Renderer(sim) as a, Renderer(sim) as b, Renderer(oracle_lib) as ref:
        synth.install(a, nodes)
        synth.install(b, folded)
        synth.install(ref, nodes)
        got = a.fill_buffer(V, 100, 100 + T, [t])
        plan = a.plan()
        assert plan["pull_rows"] == 0 and [(x["voices"], x["partials"]) for x in plan["banks"]] == [(V, P)], plan
        assert same_bits(got, ref.fill_buffer(V, 100, 100 + T, [t]))
        assert same_bits(got, b.fill_buffer(V, 100, 100 + T, [t]))


def test_host_entry_point_two_interleaved_banks(sim, oracle_lib):
    """fr_fill_buffer on a pure bank plan with two banks (different partial counts) interleaved over the rows: the first
    bank launch reads the staged input row through the mapping and appends it to the history, the second reads it too;
    a short (padded) row in the middle."""
    g = synth.GraphArrays()
    p = synth.voice_params(16, 64, 9, True)
    a = synth.sum_tree(g, synth.partial_leaves(g, p["w"][:8, :32], p["amp"][:8, :32]).reshape(8, 32))
    b = synth.sum_tree(g, synth.partial_leaves(g, p["w"][8:], p["amp"][8:]).reshape(8, 64))
    rows = np.arange(16, dtype=np.uint32)
    g.edge(a, 0, 0, rows[0::2])      # banks interleaved over the rows
    g.edge(b, 0, 0, rows[1::2])
    tree = g.finish(16)
    T = 4100
    with Renderer(sim) as r, Renderer(oracle_lib) as ref:
        synth.install(r, tree)
        synth.install(ref, tree)
        for k, rl in enumerate((T, T - 100, T)):
            t = synth.time_ramp(k * T, k * T + rl)
            got = r.fill_buffer(16, k * T, (k + 1) * T, [t])
            plan = r.plan()
            assert len(plan["banks"]) == 2 and plan["pull_rows"] == 0
            cols = [0, 1, 63, 64, T - 101, T - 100, T - 1]
            for c in cols:   # the oracle evaluates sampled frames (a Delay-free graph: random access by seek)
                tt = t[c:c + 1] if c < rl else t[-1:]
                exp = ref.fill_buffer(16, k * T + c, k * T + c + 1, [tt])
                assert same_bits(got[:, c:c + 1], exp), f"call {k} frame {c}"


def test_bounded_input_history(sim, oracle_lib):
    """fr_config.history_frames: the input history slides in a buffer of fixed size instead of growing for ever.  Within
    the cap (rounded up to what the plan's delays need) results equal the reference's; a Delay added later that reaches
    further back than the cap reads 0.0 there -- what the reference returns for times before a seek point."""
    T = 500
    with Renderer(sim, history_frames=1000) as r, Renderer(sim) as full, Renderer(oracle_lib) as ref:
        for x in (r, full, ref):
            x.on_add_node(1, "F32Constant")
            x.on_add_node(2, "Delay")      # in0 delayed by 1800 frames: more than the cap, so the cap is raised to it
            x.on_add_node(3, "Sum2")
            x.on_add_edge(0, 2, 0, 0)
            x.on_add_edge(1, 2, f32_bits(1800.0), 1)
            x.on_add_edge(2, 3, 0, 0)
            x.on_add_edge(0, 3, 1, 1)
            x.on_add_edge(3, 0, 0, 0)
        rng = np.random.default_rng(4)
        hist = []
        for k in range(40):
            rows = [rng.normal(size=T).astype(np.float32), rng.normal(size=T).astype(np.float32)]
            hist.append(rows[0])
            exp = ref.fill_buffer(1, k * T, (k + 1) * T, rows)
            assert same_bits(r.fill_buffer(1, k * T, (k + 1) * T, rows), exp), f"call {k}"
            assert same_bits(full.fill_buffer(1, k * T, (k + 1) * T, rows), exp)
        plan = r.plan()
        assert plan["history_frames"] == 1000 and plan["input_lookback"] == 1800 and not plan["input_lookback_unbounded"], plan
        # a longer Delay arrives: 6000 frames back.  The unbounded renderers still hold those samples; the capped one kept
        # 1800 (+ slack up to its buffer), so its output is 0 wherever t - 6000 is older than what it holds
        for x in (r, full, ref):
            x.on_add_node(4, "Delay")
            x.on_add_edge(0, 4, 0, 0)
            x.on_add_edge(1, 4, f32_bits(6000.0), 1)
            x.on_add_edge(4, 0, 0, 1)
        k = 40
        rows = [rng.normal(size=T).astype(np.float32), rng.normal(size=T).astype(np.float32)]
        exp = ref.fill_buffer(2, k * T, (k + 1) * T, rows)
        assert same_bits(full.fill_buffer(2, k * T, (k + 1) * T, rows), exp)
        got = r.fill_buffer(2, k * T, (k + 1) * T, rows)
        assert same_bits(got[0], exp[0])                       # the old delay is served as before
        mism = got[1] != exp[1]
        assert mism.any() and not got[1][mism].any()           # where it differs from the reference it is exactly 0.0
        # from now on the cap follows the new plan (6000): after 6000 more frames everything matches again
        for k in range(41, 41 + 14):
            rows = [rng.normal(size=T).astype(np.float32), rng.normal(size=T).astype(np.float32)]
            exp = ref.fill_buffer(2, k * T, (k + 1) * T, rows)
            got = r.fill_buffer(2, k * T, (k + 1) * T, rows)
        assert same_bits(got, exp)
        # a seek backwards: the floor must not hide the new samples
        rows = [rng.normal(size=T).astype(np.float32), rng.normal(size=T).astype(np.float32)]
        assert same_bits(r.fill_buffer(2, 100, 100 + T, rows), ref.fill_buffer(2, 100, 100 + T, rows))
        rows = [rng.normal(size=T).astype(np.float32), rng.normal(size=T).astype(np.float32)]
        assert same_bits(r.fill_buffer(2, 100 + T, 100 + 2 * T, rows), ref.fill_buffer(2, 100 + T, 100 + 2 * T, rows))


def test_ten_million_frames_in_bounded_memory(sim):
    """10^7 frames through a renderer with history_frames set: device allocations stop after the first calls (the history
    slides in place), where the unbounded default keeps 40 MB per fed slot by then."""
    import ctypes
    sim.lib.fr_sim_live_bytes.restype = ctypes.c_uint64
    T = 50000
    with Renderer(sim, history_frames=48000) as r:
        r.on_add_node(1, "F32Constant")
        r.on_add_node(2, "Delay")
        r.on_add_edge(0, 2, 0, 0)
        r.on_add_edge(1, 2, f32_bits(100.0), 1)
        r.on_add_edge(2, 0, 0, 0)
        row = np.arange(T, dtype=np.float32)
        out = np.zeros((1, T), np.float32)
        allocs = ctypes.c_uint64.in_dll(sim.lib, "fr_sim_allocs")
        peak = 0
        for k in range(200):   # 200 x 50 000 = 10^7 frames
            r.fill_buffer(1, k * T, (k + 1) * T, [row], out=out)
            if k == 5:
                settled = allocs.value
            peak = max(peak, sim.lib.fr_sim_live_bytes())
            assert out[0, 100] == row[0] and out[0, 99] == (row[T - 1] if k else 0.0)
        assert allocs.value == settled, "device allocations kept happening"
        assert peak < 4 * (2 * (48000 + T) * 4) + (1 << 20), peak   # history buffer + output + slack; not 40 MB


def _sparkle_cases(lib, oracle_lib, mode):
    """Minimum with a NaN on either side and Delay with negative / NaN / fractional / huge amounts -- constant and as
    signals -- under FR_SEMANTICS_SPARKLE, against the oracle under the same flag; and the two semantics really differ."""
    nan = float("nan")
    rows = [np.array([1.0, nan, -2.0, nan, 5.0, -0.0, 0.0, 3.0], np.float32),          # a
            np.array([nan, 2.0, -3.0, nan, 4.0, 0.0, -0.0, nan], np.float32),           # b
            np.array([-1.0, nan, 0.5, 2.0, -0.0, 3.9, 1e30, 1.0], np.float32)]          # delay amounts
    outs = {}
    for sem in ("sparkle", "reference"):
        with Renderer(lib, mode=mode, semantics=sem) as r, Renderer(oracle_lib, semantics=sem) as ref:
            for x in (r, ref):
                x.on_add_node(1, "F32Constant")
                x.on_add_node(2, "Minimum")       # min(a, b)
                x.on_add_node(3, "Minimum")       # min(b, a)
                x.on_add_node(4, "Delay")         # a delayed by the signal in slot 2
                x.on_add_node(5, "Delay")         # a delayed by the constant -3
                x.on_add_node(6, "Delay")         # (a + 1) delayed by NaN: a computed source
                x.on_add_node(7, "Sum2")
                x.on_add_node(8, "Minimum")       # min(NaN constant, a): folds nothing, a NaN on the LEFT
                x.on_add_edge(0, 2, 0, 0); x.on_add_edge(0, 2, 1, 1)
                x.on_add_edge(0, 3, 1, 0); x.on_add_edge(0, 3, 0, 1)
                x.on_add_edge(0, 4, 0, 0); x.on_add_edge(0, 4, 2, 1)
                x.on_add_edge(0, 5, 0, 0); x.on_add_edge(1, 5, f32_bits(-3.0), 1)
                x.on_add_edge(0, 7, 0, 0); x.on_add_edge(1, 7, f32_bits(1.0), 1)
                x.on_add_edge(7, 6, 0, 0); x.on_add_edge(1, 6, f32_bits(nan), 1)
                x.on_add_edge(1, 8, f32_bits(nan), 0); x.on_add_edge(0, 8, 0, 1)
                for i, h in enumerate((2, 3, 4, 5, 6, 8)):
                    x.on_add_edge(h, 0, 0, i)
            got = r.fill_buffer(6, 0, 8, rows)
            exp = ref.fill_buffer(6, 0, 8, rows)
            assert same_bits(got, exp), f"{sem} / {mode}: " + G.first_diff(got, exp)
            outs[sem] = got
    s, f = outs["sparkle"], outs["reference"]
    assert np.isnan(s[0, 1]) and f[0, 1] == 2.0                    # min(NaN, 2): Sparkle returns its left operand
    assert s[0, 0] == 1.0 and f[0, 0] == 1.0                       # min(1, NaN): both return the left operand
    assert not s[3].any() and same_bits(f[3], rows[0])             # Delay by -3: Sparkle 0.0, RefRenderer no delay
    assert s[2, 0] == 0.0 and f[2, 0] == 1.0                       # Delay by the signal -1 at t = 0
    assert not s[4].any() and same_bits(f[4], rows[0] + 1)         # Delay by NaN
    assert np.isnan(s[5]).all() and same_bits(f[5], rows[0])       # min(NaN, a)


@pytest.mark.parametrize("mode", ["auto", "pull", "staged"])
def test_sparkle_semantics(sim, oracle_lib, mode):
    _sparkle_cases(sim, oracle_lib, mode)


@pytest.mark.parametrize("seed", range(8))
def test_random_graphs_sparkle_semantics(sim, oracle_lib, seed):
    """Random graphs (NaN / negative constants and signal delays included) under FR_SEMANTICS_SPARKLE."""
    import randgraph
    rng = np.random.default_rng(2000 + seed)
    steps, n_out = randgraph.random_graph(700 + seed, n_nodes=int(rng.integers(6, 40)), n_inputs=2, n_outputs=3)
    T = 64
    for mode in ("auto", "pull"):
        with Renderer(sim, mode=mode, semantics="sparkle") as r, Renderer(oracle_lib, semantics="sparkle") as ref:
            randgraph.install_steps(r, steps)
            randgraph.install_steps(ref, steps)
            for start in (0, T):
                rows = [synth.time_ramp(start, start + T), (rng.normal(size=T) * 3).astype(np.float32)]
                try:
                    exp = ref.fill_buffer(n_out, start, start + T, rows)
                except RenderError as e:
                    with pytest.raises(RenderError) as ei:
                        r.fill_buffer(n_out, start, start + T, rows)
                    assert ei.value.status == e.status
                    break
                got = r.fill_buffer(n_out, start, start + T, rows)
                assert same_bits(got, exp), f"seed {seed} {mode}: " + G.first_diff(got, exp)


def _device_buffers(lib, n_out, T):
    """(out buffer, row buffer, their pointers, a reader, a writer) for the device entry point: numpy on the simulator (its
    'device' memory is host memory), torch on the GPU."""
    if lib.path.endswith("libfr_simengine.so") or "simasan" in lib.path:
        out = np.zeros((n_out, T), np.float32)
        row = np.zeros(T, np.float32)
        return out.ctypes.data, row.ctypes.data, (lambda: out.copy()), (lambda x: row.__setitem__(slice(None), x)), 0
    import torch
    d_out = torch.zeros((n_out, T), dtype=torch.float32, device="cuda")
    d_row = torch.zeros(T, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def read():
        torch.cuda.synchronize()
        return d_out.cpu().numpy()
    return d_out.data_ptr(), d_row.data_ptr(), read, (lambda x: d_row.copy_(torch.from_numpy(x))), stream


def test_bounded_history_with_banks_and_device_rows(sim, oracle_lib):
    """Bounded history where the bank kernel appends the caller's device-resident row to the history itself, while the
    buffer slides: a voice bank plus a row that reads the time input 1500 frames back, 45 calls of 3000 frames through the
    device entry point with history_frames = 2000 (the buffer slides about every 20 calls).  The delayed row is compared
    with the oracle (which keeps everything) in full, the voices with the numpy restatement of the bank."""
    V, P, T = 2, 32, 3000
    g = synth.GraphArrays()
    p = synth.voice_params(V, P, 3, True)
    roots = synth.sum_tree(g, synth.partial_leaves(g, p["w"], p["amp"]).reshape(V, P))
    g.edge(roots, 0, 0, np.arange(V, dtype=np.uint32))
    dl = g.binop(synth.K_DELAY, synth.IN(0), synth.C(np.float32(1500.0)), 1)
    g.edge(dl, 0, 0, V)
    tree = g.finish(V + 1)
    # the oracle gets only the delayed row (its cost is per output sample of every row it renders)
    g2 = synth.GraphArrays()
    dl2 = g2.binop(synth.K_DELAY, synth.IN(0), synth.C(np.float32(1500.0)), 1)
    g2.edge(dl2, 0, 0, 0)
    delay_only = g2.finish(1)
    with Renderer(sim, history_frames=2000) as r, Renderer(oracle_lib) as ref:
        synth.install(r, tree)
        synth.install(ref, delay_only)
        out_ptr, row_ptr, read, write, stream = _device_buffers(sim, V + 1, T)
        for k in range(45):
            t = (np.arange(k * T, (k + 1) * T) % 7919).astype(np.float32)      # a sawtooth: every call's row differs
            write(t)
            r.fill_buffer_device(out_ptr, V + 1, T, k * T, row_ptr, [0, T], stream)
            got = read()
            exp_delay = ref.fill_buffer(1, k * T, (k + 1) * T, [t])[0]
            assert same_bits(got[V], exp_delay), f"call {k}: " + G.first_diff(got[V], exp_delay)
            for v in range(V):
                assert same_bits(got[v, :64], synth.bank_reference_numpy(p["w"][v], p["amp"][v], t[:64])), f"call {k} voice {v}"
        plan = r.plan()
        assert plan["history_frames"] == 2000 and plan["input_lookback"] == 1500 and len(plan["banks"]) == 1, plan


def test_registered_destination_is_written_directly(sim, oracle_lib):
    """fr_host_register: a page-locked destination is filled by the kernels themselves (no D2H copy); same bits, and a
    buffer that merely overlaps a registered range, or an unregistered one, takes the ordinary path."""
    V, P, T = 4, 64, 300
    tree = synth.effects_tree(V, P, taps=2, base_delay=50.0)     # banks -> rings -> programs: every kind of writer
    big = np.zeros((2 * V, T), np.float32)
    with Renderer(sim) as r, Renderer(oracle_lib) as ref:
        synth.install(r, tree)
        synth.install(ref, tree)
        r.host_register(big)
        for k in range(3):
            t = synth.time_ramp(k * T, (k + 1) * T)
            out = big[:V] if k != 1 else np.zeros((V, T), np.float32)   # call 1 through an unregistered buffer
            got = r.fill_buffer(V, k * T, (k + 1) * T, [t], out=out)
            assert same_bits(got, ref.fill_buffer(V, k * T, (k + 1) * T, [t])), f"call {k}"
        assert not big[V:].any()                                        # nothing beyond the rows asked for
        r.host_unregister(big)
        t = synth.time_ramp(3 * T, 4 * T)
        assert same_bits(r.fill_buffer(V, 3 * T, 4 * T, [t], out=big[:V]), ref.fill_buffer(V, 3 * T, 4 * T, [t]))


def test_streamed_host_output(sim, oracle_lib, monkeypatch):
    """The host entry point's streamed output (rows copied to the caller's buffer as their completion flags arrive, while
    the launch still computes the others) against the staged-copy path (FR_HOST_STREAM=0) and sampled oracle frames; fresh
    destination buffers, varying call lengths, a short (padded) row, a seek."""
    V, P = 12, 512
    tree = synth.additive_tree(V, P, seed=4, detune=True)
    streamed = Renderer(sim)
    monkeypatch.setenv("FR_HOST_STREAM", "0")
    staged = Renderer(sim)
    monkeypatch.delenv("FR_HOST_STREAM")
    with streamed, staged, Renderer(oracle_lib) as ref:
        for x in (streamed, staged, ref):
            synth.install(x, tree)
        idx = 0
        for k, (T, rl) in enumerate([(4800, 4800), (4800, 4700), (1500, 1500), (6000, 6000)]):
            if k == 3:
                idx = 10**6      # a seek
            t = synth.time_ramp(idx, idx + rl)
            a = streamed.fill_buffer(V, idx, idx + T, [t], out=np.full((V, T), np.float32(-3.0)))
            b = staged.fill_buffer(V, idx, idx + T, [t], out=np.full((V, T), np.float32(-5.0)))
            assert same_bits(a, b), f"call {k}: " + G.first_diff(a, b)
            for c in (0, 63, 64, rl - 1, T - 1):
                tt = t[c:c + 1] if c < rl else t[-1:]
                exp = ref.fill_buffer(V, idx + c, idx + c + 1, [tt])
                assert same_bits(a[:, c:c + 1], exp), f"call {k} frame {c}"
            idx += T


def test_block_streaming_is_refused_where_no_launch_can_stay_resident(sim):
    """fr_stream_begin on the host-logic simulator (no device, no resident launches): a clean error, the renderer stays usable."""
    with Renderer(sim) as r:
        synth.install(r, synth.additive_tree(2, 256))
        with pytest.raises(RenderError):
            r.stream_begin(2)
        with pytest.raises(RenderError):
            r.stream_block(0, synth.time_ramp(0, 8))
        assert r.fill_buffer(2, 0, 8, [synth.time_ramp(0, 8)]).shape == (2, 8)


# ---- feedback through Delay: the same cases as on the device (test_hip_parity.py), on the host-logic simulator, whose stage
# ---- launches run threads and programs in the LEAST favourable order (sim_kernels.cpp launch_stage) ------------------------
@pytest.mark.parametrize("d", [1, 3, 64, 100])
def test_feedback_echo(sim, oracle_lib, d):
    G.test_feedback_echo(sim, oracle_lib, d)


def test_feedback_loop_with_rows_inside_and_a_tap_behind(sim, oracle_lib):
    G.test_feedback_loop_with_rows_inside_and_a_tap_behind(sim, oracle_lib)


def test_feedback_two_taps_and_nested_loops(sim, oracle_lib):
    G.test_feedback_two_taps_and_nested_loops(sim, oracle_lib)


def test_feedback_loop_through_two_delayed_nodes_is_one_program(sim, oracle_lib):
    G.test_feedback_loop_through_two_delayed_nodes_is_one_program(sim, oracle_lib)


def test_feedback_around_a_bank_voice(sim, oracle_lib):
    G.test_feedback_around_a_bank_voice(sim, oracle_lib)


def test_feedback_that_cannot_be_evaluated_is_refused(sim):
    G.test_feedback_that_cannot_be_evaluated_is_refused(sim)


@pytest.mark.timeout(600)   # (the oracle's recursion is exponential in what a bad estimate lets through)
@pytest.mark.parametrize("seed0", [0, 40])
def test_random_feedback_graphs(sim, oracle_lib, seed0):
    G.test_random_feedback_graphs(sim, oracle_lib, seed0)


def test_feedback_inside_composite_instances(sim, oracle_lib):
    G.test_feedback_inside_composite_instances(sim, oracle_lib)


@pytest.mark.timeout(600)   # (the oracle's recursion is exponential in what a bad estimate lets through)
@pytest.mark.parametrize("seed0", [0, 60])
def test_feedback_graphs_edited_during_playback(sim, oracle_lib, seed0):
    G.test_feedback_graphs_edited_during_playback(sim, oracle_lib, seed0)
