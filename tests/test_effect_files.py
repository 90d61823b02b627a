"""The effect-definition files the repository ships (effects/*.fnd, the reference's EffectDesc JSON): up to date with their
generator, and rendered as composite nodes -- on the CPU by both restatements of the reference, on the GPU by the HIP engine
against the oracle.  (Loading them through ResMan by sha256 is the C++ host's job: tests/cpp/render_tests.cpp.)"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from kat_replay import same_bits
from libfriendship_amd import synth
from libfriendship_amd.capi import Effect, Renderer, f32_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "effects")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from ref_numpy import NumpyRefRenderer  # noqa: E402

# file, what feeds each input: ("in", external slot) or ("c", constant)
CASES = [("partial.fnd", [("in", 0), ("c", 0.0123), ("c", 0.7)]), ("triangle.fnd", [("in", 0), ("c", 0.031), ("c", 0.4)]),
         ("envelope.fnd", [("in", 0), ("in", 1)]), ("tap.fnd", [("in", 1), ("c", 0.5), ("c", 3.0)]),
         ("voice4.fnd", [("in", 0), ("c", 0.004)])]


def load(name, by_sha):
    """EffectDesc JSON -> capi.Effect (nested definitions resolved by sha256 among the shipped files)."""
    with open(os.path.join(DIR, name), "rb") as f:
        desc = json.loads(f.read())
    nodes = []
    for handle, ident in desc["adjlist"]["nodes"]:
        if ident["sha256"] is None:
            assert ident["urls"][0].startswith("primitive:///")
            nodes.append((handle["node_handle"], Effect.primitive(ident["urls"][0][len("primitive:///"):])))
        else:
            nodes.append((handle["node_handle"], load(by_sha[bytes(ident["sha256"])], by_sha)[0]))
    edges = [(e["from"]["node_handle"], e["to"]["node_handle"], e["weight"]["from_slot"], e["weight"]["to_slot"]) for e in desc["adjlist"]["edges"]]
    return Effect.graph(nodes, edges), desc


def shipped():
    by_sha = {}
    for fn in sorted(os.listdir(DIR)):
        if fn.endswith(".fnd"):
            with open(os.path.join(DIR, fn), "rb") as f:
                by_sha[hashlib.sha256(f.read()).digest()] = fn
    return by_sha


def test_files_match_their_generator_and_checksums(tmp_path):
    by_sha = shipped()
    with open(os.path.join(DIR, "SHA256SUMS")) as f:
        listed = {line.split()[1]: bytes.fromhex(line.split()[0]) for line in f}
    assert {v: k for k, v in by_sha.items()} == listed and len(listed) == 5
    before = {fn: open(os.path.join(DIR, fn), "rb").read() for fn in listed}
    subprocess.run([sys.executable, os.path.join(DIR, "make_effects.py")], check=True, capture_output=True)   # rewrites in place
    assert before == {fn: open(os.path.join(DIR, fn), "rb").read() for fn in listed}, "effects/*.fnd are stale: run effects/make_effects.py"


def install_instance(r, effect, sources):
    r.on_add_node(1, Effect.primitive("F32Constant"))
    r.on_add_node(2, effect)
    for k, (kind, v) in enumerate(sources):
        if kind == "in":
            r.on_add_edge(0, 2, v, k)
        else:
            r.on_add_edge(1, 2, f32_bits(v), k)
    r.on_add_edge(2, 0, 0, 0)


def rows_for(s, e):
    t = np.arange(s, e, dtype=np.uint64)
    return [synth.time_ramp(s, e), (((t * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(2001)).astype(np.float32) / np.float32(1000.0) - np.float32(1.0)]


@pytest.mark.parametrize("name,sources", CASES)
def test_both_restatements_render_the_shipped_effects_alike(oracle_lib, name, sources):
    effect, desc = load(name, shipped())
    assert len(desc["meta"]["inputs"]) == len(sources)
    with Renderer(oracle_lib) as ref, NumpyRefRenderer() as npr:
        install_instance(ref, effect, sources)
        install_instance(npr, effect, sources)
        for s, e in ((0, 64), (64, 700)):
            a, b = ref.fill_buffer(1, s, e, rows_for(s, e)), npr.fill_buffer(1, s, e, rows_for(s, e))
            assert same_bits(a, b) and np.any(a != 0)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "staged", "pull"])
@pytest.mark.parametrize("name,sources", CASES)
def test_shipped_effects_on_the_device(hip_lib, oracle_lib, name, sources, mode):
    effect, _ = load(name, shipped())
    with Renderer(hip_lib, mode=mode) as hip, Renderer(oracle_lib) as ref:
        install_instance(hip, effect, sources)
        install_instance(ref, effect, sources)
        for s, e in ((0, 64), (64, 700), (5000, 5300)):
            assert same_bits(hip.fill_buffer(1, s, e, rows_for(s, e)), ref.fill_buffer(1, s, e, rows_for(s, e)))


@pytest.mark.gpu
def test_a_bank_of_shipped_partials_is_recognised(hip_lib, oracle_lib):
    """64 instances of effects/partial.fnd per voice under a Sum2 tree: lowering inlines the composites, the matcher sees the
    template voice, the bank kernel renders it -- same bits as the oracle walking the nested effects."""
    effect, _ = load("partial.fnd", shipped())
    V, P, T = 2, 64, 300
    p = synth.voice_params(V, P, seed=5, detune=True)
    g = synth.GraphArrays()
    part = np.arange(g.next, g.next + V * P, dtype=np.uint32)
    g.next += V * P
    roots = synth.sum_tree(g, part.reshape(V, P))
    g.edge(roots, 0, 0, np.arange(V, dtype=np.uint32))
    tree = g.finish(V)
    with Renderer(hip_lib) as hip, Renderer(oracle_lib) as ref:
        for r in (hip, ref):
            r.on_add_nodes(part, effect)
            r.on_add_edges(np.stack([np.zeros(V * P), part, np.zeros(V * P), np.zeros(V * P)], axis=1).astype(np.uint32))        # time -> input 0
            synth.install(r, tree)
            r.on_add_edges(np.stack([np.ones(V * P), part, synth.bits(p["w"].ravel()), np.ones(V * P)], axis=1).astype(np.uint32))      # w
            r.on_add_edges(np.stack([np.ones(V * P), part, synth.bits(p["amp"].ravel()), np.full(V * P, 2)], axis=1).astype(np.uint32))  # amp
        for s in (0, T):
            assert same_bits(hip.fill_buffer(V, s, s + T, [synth.time_ramp(s, s + T)]), ref.fill_buffer(V, s, s + T, [synth.time_ramp(s, s + T)]))
        plan = hip.plan()
        assert plan["pull_rows"] == 0 and [b["voices"] for b in plan["banks"] if not b["jit"]] == [V], plan
