"""Triangulates the CPU oracle: oracle/ref_numpy.py is a second restatement of the reference renderer (numpy, written
separately from the same source text, no shared code).  It must reproduce the reference's own known-answer vectors and
agree bit for bit with the C++ oracle on seeded random graphs -- all seven primitives, nested composite effects,
constant and signal-driven delays, short rows, seeks, edits between calls -- under both semantics."""
import os
import sys

import numpy as np
import pytest

import kat_replay
import randgraph
from kat_replay import same_bits
from libfriendship_amd import synth
from libfriendship_amd.capi import RenderError, Renderer

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from ref_numpy import NumpyRefRenderer, RefPanic  # noqa: E402


@pytest.mark.parametrize("i", range(11))
def test_numpy_restatement_matches_reference_kat(kat, i):
    kat_replay.check(None, kat["tests"][i], make=NumpyRefRenderer)


@pytest.mark.parametrize("i", range(14))
def test_numpy_restatement_reproduces_the_selfcheck_vectors(selfcheck, i):
    """The committed self-consistency vectors were rendered by the C++ oracle; the numpy restatement gives the same bits
    (the two big synthetic trees are skipped: minutes in numpy, and covered by test_oracle_bank_matches_numpy)."""
    t = selfcheck["tests"][i]
    if any(s["op"] == "synth_tree" and s["voices"] * s["partials"] > 64 for s in t["steps"]):
        pytest.skip("large synthetic tree")
    kat_replay.check(None, t, make=NumpyRefRenderer)


@pytest.mark.parametrize("semantics", ["reference", "sparkle"])
@pytest.mark.parametrize("block", range(8))
def test_two_restatements_agree_on_random_graphs(oracle_lib, semantics, block):
    n_checked = 0
    for seed in range(block * 40, block * 40 + 40):
        rng = np.random.default_rng(seed)
        steps, n_out = randgraph.random_graph(20_000 + seed, n_nodes=int(rng.integers(3, 70)), n_inputs=2, n_outputs=3,
                                              signal_delays=bool(seed % 3))
        T = int(rng.integers(1, 120))
        calls = [(0, T), (T, 2 * T), (2 * T, 3 * T), (int(rng.integers(4 * T, 10**5)), None)]
        with Renderer(oracle_lib, semantics=semantics) as ref, NumpyRefRenderer(semantics) as npr:
            randgraph.install_steps(ref, steps)
            randgraph.install_steps(npr, steps)
            for k, (s, e) in enumerate(calls):
                e = e if e is not None else s + T
                n_t = e - s
                rows = [synth.time_ramp(s, e)[: n_t if k != 1 else int(rng.integers(0, n_t + 1))],
                        (rng.normal(size=n_t) * 3).astype(np.float32)]
                exp = ref.fill_buffer(n_out, s, e, rows)
                got = npr.fill_buffer(n_out, s, e, rows)
                assert same_bits(got, exp), f"seed {seed} call {k}: numpy restatement and C++ oracle differ"
                n_checked += 1
                if seed % 2 and k in (0, 2):   # edits between calls
                    for st in randgraph.random_edits(rng, steps, 4, signal_delays=bool(seed % 3)):
                        for r in (ref, npr):
                            randgraph.install_steps(r, [st])
    assert n_checked == 160


def test_both_restatements_refuse_the_same_calls(oracle_lib):
    """reference.rs:69 and :71 are asserts: a row arriving for a slot whose history is not `idx` long, and a row longer
    than the call."""
    for bad in ("long_row", "history"):
        with Renderer(oracle_lib) as ref, NumpyRefRenderer() as npr:
            for r in (ref, npr):
                r.on_add_edge(0, 0, 0, 0)
                r.fill_buffer(1, 0, 4, [np.arange(4, dtype=np.float32)])
            args = (1, 4, 8, [np.zeros(5, np.float32)]) if bad == "long_row" else None
            if bad == "history":   # slot 1 was never fed: its stored row is 0 long... until a call creates it at idx 0
                for r in (ref, npr):
                    r.fill_buffer(1, 4, 8, [np.zeros(4, np.float32)])
                args = (1, 8, 12, [np.zeros(4, np.float32), np.zeros(4, np.float32)])
            with pytest.raises(RenderError):
                ref.fill_buffer(*args)
            with pytest.raises(RefPanic):
                npr.fill_buffer(*args)


def test_delay_amounts_at_the_edges_agree(oracle_lib):
    """Delay amounts around 0, 2^63 and 2^64, NaN, infinities, negative zero: same bits from both restatements."""
    amounts = [0.0, -0.0, 0.99, 1.0, 3.7, -1.0, -1e-30, 9.223372e18, 9.2233725e18, 1.8446743e19, 1.8446744e19, 3.0e19,
               float("inf"), float("-inf"), float("nan")]
    for semantics in ("reference", "sparkle"):
        with Renderer(oracle_lib, semantics=semantics) as ref, NumpyRefRenderer(semantics) as npr:
            for r in (ref, npr):
                r.on_add_node(1, "F32Constant")
                r.on_add_node(2, "Delay")
                r.on_add_edge(0, 2, 0, 0)      # source: input 0
                r.on_add_edge(0, 2, 1, 1)      # amount: input 1 (a signal)
                r.on_add_edge(2, 0, 0, 0)
            n = len(amounts)
            rows = [np.arange(1, n + 1, dtype=np.float32), np.array(amounts, np.float32)]
            assert same_bits(npr.fill_buffer(1, 0, n, rows), ref.fill_buffer(1, 0, n, rows))


@pytest.mark.parametrize("params_as_nodes", [False, True])
def test_two_restatements_agree_on_the_synthetic_trees(oracle_lib, params_as_nodes):
    """The benchmark's own graphs at a size numpy walks in seconds: the additive tree (N1-N4) and the effects tree
    (N5 envelope, N6 delay chain), across a call boundary so the delay taps read the previous call's frames."""
    for tree in (synth.additive_tree(3, 32, params_as_nodes=params_as_nodes), synth.effects_tree(2, 16, params_as_nodes=params_as_nodes)):
        with Renderer(oracle_lib) as ref, NumpyRefRenderer() as npr:
            synth.install(ref, tree)
            synth.install(npr, tree)
            n = int(tree["n_outputs"])
            for s, e in ((0, 300), (300, 5000), (5000, 5100)):
                rows = [synth.time_ramp(s, e)]
                assert same_bits(npr.fill_buffer(n, s, e, rows), ref.fill_buffer(n, s, e, rows))
