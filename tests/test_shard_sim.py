"""Sharding is a property of the engine (fr_set_shard): every rank gets the SAME graph through the ordinary ABI.  These
CPU tests run all ranks of a job on the host-logic simulator (the engine's real planner, input store, ring logic and
exchange schedule; tests/sim_tools.py), one thread per rank, and compare the assembled rows bit-for-bit with an
unsharded oracle render.  The gloo multi-process form is tests/test_distributed.py."""
import numpy as np
import pytest

import randgraph
import shard_harness
import sim_tools
from kat_replay import same_bits
from libfriendship_amd import synth
from libfriendship_amd.capi import RenderError, Renderer
from test_hip_parity import first_diff


@pytest.fixture(scope="module")
def sim():
    return sim_tools.sim_lib()


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("seed", range(6))
def test_voices_mode_random_graphs(sim, oracle_lib, seed, world):
    """Voices mode on arbitrary graphs: every rank renders exactly its rows -- composites, signal delays, pull rows, a
    short row, a seek -- and nothing else; no exchange happens."""
    rng = np.random.default_rng(4000 + seed)
    n_out = 5
    steps, _ = randgraph.random_graph(500 + seed, n_nodes=int(rng.integers(6, 36)), n_inputs=2, n_outputs=n_out)
    job = shard_harness.Job(sim, world, "voices")
    T = 40
    with Renderer(oracle_lib) as ref:
        randgraph.install_steps(ref, steps)
        for ren in job.ranks:
            randgraph.install_steps(ren, steps)
        noise = lambda n: (rng.normal(size=n) * 3).astype(np.float32)
        calls = [(0, T, [synth.time_ramp(0, T), noise(T)]), (T, 2 * T, [synth.time_ramp(T, 2 * T), noise(T // 3)]),
                 (900, 900 + T, [synth.time_ramp(900, 900 + T), noise(T)])]
        for start, end, rows in calls:
            try:
                exp = ref.fill_buffer(n_out, start, end, rows)
            except RenderError:
                break   # (a graph the reference panics on: error parity is covered by the unsharded tests)
            got = job.assemble(job.fill(n_out, start, end, rows), n_out)
            assert same_bits(got, exp), first_diff(got, exp)
    assert sum(job.boxes.messages) == 0
    job.close()


@pytest.mark.parametrize("world,V,P", [(2, 3, 64), (4, 5, 128), (8, 3, 256), (4, 2, 64)])
def test_partials_mode_additive_tree(sim, oracle_lib, world, V, P):
    """Partial-block sharding of the additive tree: every voice is cut at the top log2(world) levels, each rank renders
    its block of partials of EVERY voice, one recursive-halving exchange leaves each owner with its voices."""
    tree = synth.additive_tree(V, P, seed=77, detune=True)
    job = shard_harness.Job(sim, world, "partials")
    with Renderer(oracle_lib) as ref:
        synth.install(ref, tree)
        for ren in job.ranks:
            synth.install(ren, tree)
        for start, T in ((1000, 48), (1048, 17)):
            rows = [synth.time_ramp(start, start + T)]
            exp = ref.fill_buffer(V, start, start + T, rows)
            got = job.assemble(job.fill(V, start, start + T, rows), V)
            assert same_bits(got, exp), first_diff(got, exp)
    splittable = P // world >= 32
    for r, ren in enumerate(job.ranks):
        plan = ren.plan()
        assert plan["shard"] == {"rank": r, "world": world, "mode": 2, "split_voices": V if splittable else 0,
                                 "transport": "host-callback"}, plan
        if splittable:   # this rank's bank launch covers ONE block of every voice
            assert [(b["voices"], b["partials"], b["to_exchange"]) for b in plan["banks"]] == [(V, P // world, True)], plan
    if splittable:
        # recursive halving: at most log2(world) sends per rank and call (a range can be empty when a rank's partner
        # side owns no voice), and what a rank sends in all is less than one copy of the partial mixes
        k = world.bit_length() - 1
        assert all(0 < m <= 2 * k for m in job.boxes.messages), job.boxes.messages
        assert all(b <= V * (48 + 17) * 4 for b in job.boxes.bytes_sent), job.boxes.bytes_sent
    job.close()


@pytest.mark.parametrize("world", [2, 4])
def test_partials_mode_effects_tree_with_seek_and_edit(sim, oracle_lib, world):
    """Config D's shape (detune + ADSR envelope + delay chain behind every voice): the voices are split over the ranks,
    the exchange delivers each voice to its owner's delay ring, the envelope / delay programs run on the owner only.
    Contiguous calls (delay lines live across calls), a seek, and a graph edit between calls."""
    V, P = 5, 64 * world
    tree = synth.effects_tree(V, P, taps=3, base_delay=40.0)
    job = shard_harness.Job(sim, world, "partials")
    with Renderer(oracle_lib) as ref:
        synth.install(ref, tree)
        for ren in job.ranks:
            synth.install(ren, tree)
        T = 64

        def check(start):
            rows = [synth.time_ramp(start, start + T)]
            exp = ref.fill_buffer(V, start, start + T, rows)
            got = job.assemble(job.fill(V, start, start + T, rows), V)
            assert same_bits(got, exp), f"[{start}, {start + T}): " + first_diff(got, exp)

        for k in range(4):
            check(k * T)
        plan = job.ranks[0].plan()
        assert plan["shard"]["split_voices"] == V and plan["rings"] > 0 and plan["pull_rows"] == 0, plan
        lo, hi = job.ranks[0].shard_rows(V)
        check(5000)                      # seek: every rank rebuilds the look-back window, exchange over the window
        check(5000 + T)
        # edit: rewire the last voice's output to the previous voice's chain (both renderers, every rank)
        e = tree["edges"]
        last = e[(e[:, 1] == 0) & (e[:, 3] == V - 1)][0]
        prev = e[(e[:, 1] == 0) & (e[:, 3] == V - 2)][0]
        for ren in job.ranks + [ref]:
            ren.on_del_edge(*[int(x) for x in last])
            ren.on_add_edge(int(prev[0]), 0, 0, V - 1)
        check(5000 + 2 * T)
        check(5000 + 3 * T)
    job.close()


@pytest.mark.parametrize("world", [2, 4])
def test_time_tiled_exchange(sim, oracle_lib, world, monkeypatch):
    """The exchange of FR_SHARD_PARTIALS cut into time tiles (tile i's exchange on a second stream under tile
    i + 1's bank kernels; FR_EXCHANGE_MIN_TILE lowered so that these small calls are tiled): the additive tree, whose
    finished voices go to output rows, and the effects tree, whose voices go to delay rings read by programs on the owner --
    contiguous calls, a seek (the look-back window is exchanged tile by tile too), ragged tile ends, a graph edit.  Same
    bits as the oracle; same bytes on the wire as the serial exchange (FR_SHARD_SERIAL_EXCHANGE), more messages."""
    monkeypatch.setenv("FR_EXCHANGE_MIN_TILE", "64")      # (read when a renderer is created)
    V, P = 5, 64 * world
    for tree, calls in ((synth.additive_tree(V, P, seed=3, detune=True), [(0, 300), (300, 556), (556, 620), (9000, 9200)]),
                        (synth.effects_tree(V, P, taps=3, base_delay=40.0), [(0, 256), (256, 556), (556, 700), (7000, 7330), (7330, 7600)])):
        tiled = shard_harness.Job(sim, world, "partials")
        serial = shard_harness.Job(sim, world, "partials", serial_exchange=True)
        with Renderer(oracle_lib) as ref:
            for ren in tiled.ranks + serial.ranks + [ref]:
                synth.install(ren, tree)
            for k, (a, b) in enumerate(calls):
                if k == 3 and "params" in tree:      # an edit between calls: one voice's output moves to the previous voice's
                    e = tree["edges"]
                    last = e[(e[:, 1] == 0) & (e[:, 3] == V - 1)][0]
                    prev = e[(e[:, 1] == 0) & (e[:, 3] == V - 2)][0]
                    for ren in tiled.ranks + serial.ranks + [ref]:
                        ren.on_del_edge(*[int(x) for x in last])
                        ren.on_add_edge(int(prev[0]), 0, 0, V - 1)
                rows = [synth.time_ramp(a, b)]
                exp = ref.fill_buffer(V, a, b, rows)
                got = tiled.assemble(tiled.fill(V, a, b, rows), V)
                assert same_bits(got, exp), f"tiled, call {k}: " + first_diff(got, exp)
                got = serial.assemble(serial.fill(V, a, b, rows), V)
                assert same_bits(got, exp), f"serial, call {k}: " + first_diff(got, exp)
        st, ss = tiled.ranks[0].plan()["exchange_stats"], serial.ranks[0].plan()["exchange_stats"]
        assert st["calls"] == ss["calls"] == len(calls) and ss["tiles"] == ss["calls"] and st["tiles"] > 2 * st["calls"], (st, ss)
        assert st["bytes_sent"] == ss["bytes_sent"] and sum(tiled.boxes.bytes_sent) == sum(serial.boxes.bytes_sent)
        assert sum(tiled.boxes.messages) > sum(serial.boxes.messages)
        tiled.close()
        serial.close()


def test_partials_mode_mixed_graph(sim, oracle_lib):
    """Voices that cannot be split stay whole on their owner: a voice too small for the world size, a voice shared by
    rows of two ranks, a non-bank row; split and unsplit voices coexist in one plan."""
    world = 4
    g = synth.GraphArrays()
    p = synth.voice_params(4, 256, 5, True)
    big = synth.sum_tree(g, synth.partial_leaves(g, p["w"][:2], p["amp"][:2]).reshape(2, 256))          # 2 splittable voices
    small = synth.sum_tree(g, synth.partial_leaves(g, p["w"][2, :64], p["amp"][2, :64]).reshape(1, 64))   # 64 / 4 < 32: whole
    shared = synth.sum_tree(g, synth.partial_leaves(g, p["w"][3, :128], p["amp"][3, :128]).reshape(1, 128))
    gain = g.binop(synth.K_MUL, shared, synth.C(np.float32(0.5)), 1)
    g.edge(big[0], 0, 0, 0)
    g.edge(small[0], 0, 0, 1)
    g.edge(shared[0], 0, 0, 2)       # row 2 (rank 1 of 4 at 6 rows: blocks 2,2,1,1) ...
    g.edge(gain[0], 0, 0, 4)         # ... and, through a gain, row 4 (rank 2): needed by two ranks
    g.edge(big[1], 0, 0, 5)
    g.edge(0, 0, 1, 3)               # row 3: input slot 1 passed through
    tree = g.finish(6)
    job = shard_harness.Job(sim, world, "partials")
    with Renderer(oracle_lib) as ref:
        synth.install(ref, tree)
        for ren in job.ranks:
            synth.install(ren, tree)
        rng = np.random.default_rng(1)
        for start in (0, 50):
            rows = [synth.time_ramp(start, start + 50), rng.normal(size=50).astype(np.float32)]
            exp = ref.fill_buffer(6, start, start + 50, rows)
            got = job.assemble(job.fill(6, start, start + 50, rows), 6)
            assert same_bits(got, exp), first_diff(got, exp)
    assert job.ranks[0].plan()["shard"]["split_voices"] == 2
    job.close()


def test_gather_to_rank0(sim, oracle_lib):
    tree = synth.additive_tree(6, 128, seed=3)
    job = shard_harness.Job(sim, 4, "partials", gather=True)
    with Renderer(oracle_lib) as ref:
        synth.install(ref, tree)
        for ren in job.ranks:
            synth.install(ren, tree)
        rows = [synth.time_ramp(0, 33)]
        exp = ref.fill_buffer(6, 0, 33, rows)
        bufs = job.fill(6, 0, 33, rows)
        assert same_bits(bufs[0], exp), first_diff(bufs[0], exp)   # rank 0 holds every row
    job.close()


def test_shard_argument_checks(sim):
    with Renderer(sim) as r:
        for bad in [dict(rank=2, world=2), dict(rank=0, world=3, mode="partials"), dict(rank=0, world=65)]:
            with pytest.raises(RenderError):
                r.set_shard(bad["rank"], bad["world"], bad.get("mode", "voices"))
        r.set_shard(0, 1, "voices")            # world 1 = unsharded
        assert r.shard_rows(7) == (0, 7)
        r.set_shard(1, 3, "voices")
        assert r.shard_rows(7) == (3, 5) and r.shard_rows(2) == (1, 2) and r.shard_rows(0) == (0, 0)


def test_partials_without_a_transport_fails_loudly(sim):
    from libfriendship_amd.capi import FR_ERR_COMM
    tree = synth.additive_tree(2, 128, seed=3)
    with Renderer(sim) as r:
        synth.install(r, tree)
        r.set_shard(0, 2, "partials")
        with pytest.raises(RenderError) as ei:
            r.fill_buffer(2, 0, 16, [synth.time_ramp(0, 16)])
        assert ei.value.status == FR_ERR_COMM


def _random_patch(rng, n_rows):
    """A random 'patch': voices of assorted sizes and tree shapes (balanced, odd counts, some tiny), each optionally through a
    gain / an envelope shared between voices / a delay tap or two, wired to the rows at random -- some voices to two rows,
    some rows to a plain input or a constant, some left unconnected."""
    g = synth.GraphArrays()
    f = np.float32
    shared_env = synth.adsr_envelope(g, attack=30.0, decay=80.0, sustain=0.5, release=200.0, t_end=900.0)
    heads = []
    for v in range(int(rng.integers(2, 6))):
        P = int(rng.choice([16, 24, 32, 64, 100, 128, 256, 512]))
        p = synth.voice_params(1, P, int(rng.integers(1, 1 << 30)), bool(rng.integers(2)))
        x = synth.sum_tree(g, synth.partial_leaves(g, p["w"], p["amp"]).reshape(1, P))
        kind = int(rng.integers(5))
        if kind == 1:
            x = g.binop(synth.K_MUL, x, synth.C(f(rng.normal())), 1)
        elif kind == 2:
            x = g.binop(synth.K_MUL, np.broadcast_to(shared_env, x.shape), x, 1)
        elif kind >= 3:
            x = synth.delay_chain(g, x, taps=kind - 2, base_delay=float(rng.integers(5, 60)))
        heads.append(x)
    for row in range(n_rows):
        r = rng.random()
        if r < 0.65:
            g.edge(heads[int(rng.integers(len(heads)))][0], 0, 0, row)
        elif r < 0.8:
            g.edge(0, 0, int(rng.integers(2)), row)                                 # an input, passed through
        elif r < 0.9:
            g.edge(synth.CONST_HANDLE, 0, int(synth.bits(f(rng.normal()))), row)     # a constant
    return g.finish(n_rows)


@pytest.mark.parametrize("seed", range(10))
def test_random_patches_sharded_every_way(sim, oracle_lib, seed):
    """Random patches (see _random_patch) under voices and partial-block sharding at world 2, 4 and 8: contiguous calls, a
    seek, every rank's rows bit-equal to the unsharded oracle, nobody writing rows it does not own."""
    rng = np.random.default_rng(9000 + seed)
    n_rows = int(rng.integers(3, 9))
    tree = _random_patch(rng, n_rows)
    T = 48
    calls = [(0, T), (T, 2 * T), (2 * T, 3 * T), (700, 700 + T), (700 + T, 700 + 2 * T)]
    rows_for = {c: [synth.time_ramp(c[0], c[1]), (rng.normal(size=T) * 2).astype(np.float32)] for c in calls}
    with Renderer(oracle_lib) as ref:
        synth.install(ref, tree)
        expect = {c: ref.fill_buffer(n_rows, c[0], c[1], rows_for[c]) for c in calls}
    for mode, world in (("voices", 2), ("voices", 3), ("partials", 2), ("partials", 4), ("partials", 8)):
        job = shard_harness.Job(sim, world, mode)
        for ren in job.ranks:
            synth.install(ren, tree)
        for c in calls:
            got = job.assemble(job.fill(n_rows, c[0], c[1], rows_for[c]), n_rows)
            assert same_bits(got, expect[c]), f"seed {seed} {mode} x{world} call {c}: " + first_diff(got, expect[c])
        job.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_voices_mode_with_feedback_loops(sim, oracle_lib, world):
    """Voice sharding of graphs with feedback through Delay: every rank lowers and plans only its rows -- a loop is cut, planned
    and replayed on the rank that owns the row it feeds -- and partial-block sharding refuses them (FR_ERR_UNSUPPORTED)."""
    done = 0
    for seed in range(40):
        made = randgraph.random_feedback_graph(seed, n_outputs=5)
        if made is None:
            continue
        steps, n_out, _d = made
        job = shard_harness.Job(sim, world, "voices")
        rng = np.random.default_rng(seed)
        with Renderer(oracle_lib) as ref:
            randgraph.install_steps(ref, steps)
            for ren in job.ranks:
                randgraph.install_steps(ren, steps)
            for start, n in [(0, 9), (9, 6), (30, 4)]:
                rows = [rng.normal(size=n).astype(np.float32), rng.integers(-2, 5, size=n).astype(np.float32)]
                exp = ref.fill_buffer(n_out, start, start + n, rows)
                got = job.assemble(job.fill(n_out, start, start + n, rows), n_out)
                assert same_bits(got, exp), f"seed {seed}: " + first_diff(got, exp)
        job.close()
        done += 1
    assert done >= 10
    made = next(m for m in (randgraph.random_feedback_graph(s, n_outputs=5) for s in range(40)) if m is not None)
    job = shard_harness.Job(sim, 2, "partials")
    for ren in job.ranks:
        randgraph.install_steps(ren, made[0])
    with pytest.raises(RenderError) as ei:
        job.fill(made[1], 0, 8, [np.ones(8, np.float32), np.ones(8, np.float32)])
    assert ei.value.status == 10
    job.close()
