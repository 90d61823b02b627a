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
