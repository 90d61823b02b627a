#!/usr/bin/env python3
"""First-call cost of a big patch: install the tree, one short fill_buffer, the plan's lower_ms / build_ms.
usage: python tools/lower_probe.py [voices partials]   (FR_LOWER_THREADS=n: threads of the from-scratch lowering)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libfriendship_amd
from libfriendship_amd import synth

V, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 4096)
t0 = time.perf_counter()
tree = synth.additive_tree(V, P, params_as_nodes=True)
t1 = time.perf_counter()
hip = libfriendship_amd.HipRenderer()
synth.install(hip, tree)
t2 = time.perf_counter()
hip.fill_buffer(V, 0, 64, [synth.time_ramp(0, 64)])
t3 = time.perf_counter()
p = hip.plan()
print(f"{V} x {P}: threads {os.environ.get('FR_LOWER_THREADS', 'default')}: tree {t1 - t0:.2f} s, install {t2 - t1:.2f} s, first call {t3 - t2:.3f} s "
      f"(lower_ms {p['lower_ms']:.1f}, build_ms {p['build_ms']:.1f}, {p['relowered_nodes']} mirror nodes -> {p['lowered_nodes']} lowered)")
