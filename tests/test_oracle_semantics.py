"""Oracle behaviour beyond the reference's own fixtures: the synthetic oscillator graphs against an
independent numpy restatement, and the input-store rules of reference.rs:47-75."""
import numpy as np
import pytest

from libfriendship_amd import synth
from libfriendship_amd.capi import (FR_ERR_INPUT_HISTORY, FR_ERR_INPUT_TOO_LONG, FR_ERR_NO_SUCH_NODE,
                                    RenderError, Renderer, f32_bits)
from kat_replay import same_bits


@pytest.mark.parametrize("P,V", [(1, 1), (2, 2), (7, 1), (64, 3), (256, 1)])
def test_oracle_bank_matches_numpy(oracle_lib, P, V):
    tree = synth.additive_tree(V, P, seed=7, detune=(P == 64))
    t = synth.time_ramp(1000, 1100)
    with Renderer(oracle_lib) as r:
        synth.install(r, tree)
        out = r.fill_buffer(V, 1000, 1100, [t])
    for v in range(V):
        ref = synth.bank_reference_numpy(tree["params"]["w"][v], tree["params"]["amp"][v], t)
        assert same_bits(out[v], ref)


def test_config_a_single_440hz_partial(oracle_lib):
    """BASELINE.json configs[0]: one 440 Hz partial, 1 voice, 48 kHz, 1 s, through the CPU path."""
    g = synth.GraphArrays()
    w = np.float32(np.float32(440.0) / np.float32(48000.0))
    leaf = synth.partial_leaves(g, [w], [np.float32(1.0)])
    g.edge(leaf, 0, 0, 0)
    tree = g.finish(1)
    t = synth.time_ramp(0, 48000)
    with Renderer(oracle_lib) as r:
        synth.install(r, tree)
        out = r.fill_buffer(1, 0, 48000, [t])[0]
    ref = synth.bank_reference_numpy([w], [1.0], t)
    assert same_bits(out, ref)
    # it is a sine-shaped wave at 440 Hz: compare with sin to the parabola's known 6% bound
    s = np.sin(2 * np.pi * 440.0 * np.arange(48000) / 48000.0)
    assert np.max(np.abs(out - s)) < 0.06
    assert abs(out.max() - 1.0) < 1e-3 and abs(out.min() + 1.0) < 1e-3


def test_input_row_too_long_is_refused(oracle_lib):
    with Renderer(oracle_lib) as r:
        r.on_add_edge(0, 0, 0, 0)
        with pytest.raises(RenderError) as ei:
            r.fill_buffer(1, 0, 4, [[1, 2, 3, 4, 5]])
        assert ei.value.status == FR_ERR_INPUT_TOO_LONG


def test_input_history_mismatch_is_refused(oracle_lib):
    with Renderer(oracle_lib) as r:
        r.on_add_edge(0, 0, 0, 0)
        r.on_add_edge(0, 0, 1, 1)
        r.fill_buffer(2, 0, 4, [[1, 2, 3, 4]])          # slot 1 gets no row: stays at length 0
        with pytest.raises(RenderError) as ei:           # a row for slot 1 now trips assert_eq!(len, idx)
            r.fill_buffer(2, 4, 8, [[1, 2, 3, 4], [9, 9, 9, 9]])
        assert ei.value.status == FR_ERR_INPUT_HISTORY


def test_edge_into_unknown_node_is_refused(oracle_lib):
    with Renderer(oracle_lib) as r:
        with pytest.raises(RenderError) as ei:
            r.on_add_edge(0, 42, 0, 0)
        assert ei.value.status == FR_ERR_NO_SUCH_NODE


def test_rows_beyond_storage_are_dropped(oracle_lib):
    """reference.rs:60-68: storage grows to n_slots*n_times vectors; rows past that are silently dropped."""
    with Renderer(oracle_lib) as r:
        r.on_add_edge(0, 0, 1, 0)   # out0 <- input slot 1
        out = r.fill_buffer(1, 0, 1, [[5.0], [7.0]])   # only 1 vector exists: row 1 is dropped
        assert out.tolist() == [[0.0]]
    with Renderer(oracle_lib) as r:
        r.on_add_edge(0, 0, 1, 0)
        out = r.fill_buffer(1, 0, 2, [[5.0, 5.0], [7.0, 8.0]])   # 2 vectors: row 1 is stored
        assert out.tolist() == [[7.0, 8.0]]


def test_delay_edge_cases(oracle_lib):
    """Unpinned Delay behaviour, fixed by the oracle's documented choices."""
    cases = [(-2.0, [1, 2, 3, 4]),            # negative amount clamps to 0 (reference.rs:206-207)
             (float("nan"), [1, 2, 3, 4]),   # NaN as u64 == 0
             (1.9, [0, 1, 2, 3]),             # flooring conversion
             (1.8446744073709552e19, [0, 0, 0, 0]),   # >= 2^64 -> 0
             (1e10, [0, 0, 0, 0])]            # t - d underflows -> 0
    for d, exp in cases:
        with Renderer(oracle_lib) as r:
            r.on_add_node(1, "Delay")
            r.on_add_node(2, "F32Constant")
            r.on_add_edge(1, 0, 0, 0)
            r.on_add_edge(0, 1, 0, 0)
            r.on_add_edge(2, 1, f32_bits(d), 1)
            out = r.fill_buffer(1, 0, 4, [[1, 2, 3, 4]])
            assert out.tolist() == [[float(x) for x in exp]], d


def test_minimum_ties_and_nan(oracle_lib):
    """Rust >= 1.20 f32::min: (a < b || b.is_nan()) ? a : b -- NaN loses, ties return the right operand."""
    def run(a, b):
        with Renderer(oracle_lib) as r:
            r.on_add_node(1, "Minimum")
            r.on_add_node(2, "F32Constant")
            r.on_add_edge(1, 0, 0, 0)
            r.on_add_edge(2, 1, f32_bits(a), 0)
            r.on_add_edge(2, 1, f32_bits(b), 1)
            return r.fill_buffer(1, 0, 1)[0, 0]
    assert run(float("nan"), 2.0) == 2.0
    assert run(2.0, float("nan")) == 2.0
    assert np.signbit(run(0.0, -0.0)) and not np.signbit(run(-0.0, 0.0))


def test_modulo_edge_cases(oracle_lib):
    def run(a, b):
        with Renderer(oracle_lib) as r:
            r.on_add_node(1, "Modulo")
            r.on_add_node(2, "F32Constant")
            r.on_add_edge(1, 0, 0, 0)
            r.on_add_edge(2, 1, f32_bits(a), 0)
            r.on_add_edge(2, 1, f32_bits(b), 1)
            return r.fill_buffer(1, 0, 1)[0, 0]
    assert run(-3.5, 2.0) == 0.5
    assert run(-1e-10, 1.0) == 1.0             # rem + divisor rounds to the divisor itself
    assert np.isnan(run(1.0, 0.0))
    assert run(-2.0, 1.0) == 0.0 and np.signbit(run(-2.0, 1.0))   # -0.0 passes through
    assert run(3.5, -2.0) == 1.5


def test_graph_edit_between_calls_and_seek(oracle_lib):
    with Renderer(oracle_lib) as r:
        r.on_add_node(1, "Sum2")
        r.on_add_node(2, "F32Constant")
        r.on_add_edge(1, 0, 0, 0)
        r.on_add_edge(0, 1, 0, 0)
        r.on_add_edge(2, 1, f32_bits(10.0), 1)
        assert r.fill_buffer(1, 0, 3, [[1, 2, 3]]).tolist() == [[11, 12, 13]]
        r.on_del_edge(2, 1, f32_bits(10.0), 1)
        assert r.fill_buffer(1, 3, 6, [[4, 5, 6]]).tolist() == [[4, 5, 6]]
        r.on_del_node(1)   # output edge now dangles onto a missing node: the reference panics (reference.rs:186)
        with pytest.raises(RenderError):
            r.fill_buffer(1, 6, 9, [[7, 8, 9]])
