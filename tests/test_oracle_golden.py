"""Pins the CPU oracle against every known-answer test the reference ships for the render path
(SURVEY.md 8c): 11 tests / 14 asserted arrays, exact f32 equality."""
import pytest

import kat_replay


def test_fixture_is_complete(kat):
    names = [t["name"] for t in kat["tests"]]
    assert len(names) == 11
    assert sum(1 for t in kat["tests"] for s in t["steps"] if s["op"] == "render") == 14


@pytest.mark.parametrize("i", range(11))
def test_oracle_matches_reference_kat(oracle_lib, kat, i):
    kat_replay.check(oracle_lib, kat["tests"][i])


N_SELFCHECK = 14


def test_selfcheck_fixture_is_complete(selfcheck):
    assert len(selfcheck["tests"]) == N_SELFCHECK and "NOT reference outputs" in selfcheck["kind"]
    assert sum(1 for t in selfcheck["tests"] for s in t["steps"] if s["op"] == "render") == 12 * 4 + 2


@pytest.mark.parametrize("i", range(N_SELFCHECK))
def test_oracle_reproduces_its_committed_vectors(oracle_lib, selfcheck, i):
    """tests/golden/selfcheck_vectors.json (SURVEY.md 8c 'extra golden data'): random graphs with composites, signal
    delays, short rows, a seek and edits between calls; a one-second single partial; config B.  Guards the oracle
    against silent semantic changes -- the reference's own vectors (above) pin only 1x4 outputs."""
    kat_replay.check(oracle_lib, selfcheck["tests"][i])
