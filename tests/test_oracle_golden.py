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
