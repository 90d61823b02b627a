#!/usr/bin/env python3
"""fr_fill_buffer (host buffers, synchronous) at config C under the FR_HOST_MAPPED A/B modes; one subprocess per mode.
usage: python tools/host_api_modes.py [modes...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
import libfriendship_amd
from libfriendship_amd import synth
V, P, T = 64, 4096, 4800
r = libfriendship_amd.HipRenderer()
synth.install(r, synth.additive_tree(V, P))
out = np.zeros((V, T), np.float32)
rows = [synth.time_ramp(k * T, (k + 1) * T) for k in range(260)]
ts = []
for k in range(260):
    t0 = time.perf_counter()
    r.fill_buffer(V, k * T, (k + 1) * T, [rows[k]], out=out)
    ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts[60:])
print("mapped %%s: median %%.1f us  p10 %%.1f  p90 %%.1f  -> %%.1f Msamples/s" %% (os.environ.get("FR_HOST_MAPPED", "default"), np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), T / np.median(ts)))
''' % ROOT
for m in (sys.argv[1:] or ["0", "1", "2", "3"]):
    env = dict(os.environ, FR_HOST_MAPPED=m, FR_HOST_TRACE="1")
    p = subprocess.run([sys.executable, "-c", CHILD + "\nr.close()\n"], env=env, capture_output=True, text=True)
    print(p.stdout.strip() or p.stderr[-500:])
    print("   ", "\n    ".join(l for l in p.stderr.splitlines() if "phases" in l))
