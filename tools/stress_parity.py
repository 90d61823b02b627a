#!/usr/bin/env python3
"""One-off parity stress on an MI355X: many seeded random graphs (all 7 primitives, nested composites, constant and
signal-driven delays) through every engine mode against the CPU oracle, with contiguous calls, short rows, seeks.
Not part of the default test suite (minutes, not seconds).   usage: python tools/stress_parity.py [n_seeds [first_seed]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import randgraph  # noqa: E402
from kat_replay import same_bits  # noqa: E402
import libfriendship_amd  # noqa: E402
from libfriendship_amd import synth  # noqa: E402
from libfriendship_amd.capi import RenderError, Renderer, RendererLib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if os.environ.get("FR_STRESS_LIB") == "sim":     # the host-logic simulator (CPU): planner / input store / rings, no kernels
        import sim_tools
        hip = sim_tools.sim_lib()
    else:
        hip = libfriendship_amd.hip_lib()
    oracle = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    bad = 0
    only = [int(x) for x in os.environ["FR_STRESS_SEEDS"].split(",")] if os.environ.get("FR_STRESS_SEEDS") else None
    for seed in (only or range(first, first + n)):
        rng = np.random.default_rng(seed)
        long_form = bool(os.environ.get("FR_STRESS_LONG"))   # calls of up to 3000 frames over delays of up to 2500: ring
        steps, n_out = randgraph.random_graph(10_000 + seed, n_nodes=int(rng.integers(3, 40 if long_form else 70)), n_inputs=2, n_outputs=3,
                                              signal_delays=bool(seed % 3), max_delay=2500 if long_form else 9)   # wrap, windows
        T = int(rng.integers(1, 3000 if long_form else 200))
        calls = [(0, T), (T, 2 * T), (2 * T, 3 * T), (int(rng.integers(4 * T, 10**6)), None)]
        edit_after = {0, 2} if seed % 2 else set()          # odd seeds: graph edits between calls (incremental re-lowering)
        with Renderer(oracle) as ref, Renderer(oracle, semantics="sparkle") as ref_s:
            randgraph.install_steps(ref, steps)
            randgraph.install_steps(ref_s, steps)
            # FR_STRESS_ASYNC=1: the ABI's default -- hipRTC on a worker thread.  The first calls run on the interpreters, the
            # tool then waits out the compile and the following calls find the plan switched over (rings rebuilt): same bits.
            sync = not os.environ.get("FR_STRESS_ASYNC")
            modes = {m: Renderer(hip, mode=m, sync_compile=sync) for m in ("auto", "staged", "pull")}
            sparkle = {m: Renderer(hip, mode=m, semantics="sparkle") for m in ("auto", "pull")}   # FR_SEMANTICS_SPARKLE vs its oracle
            for r in sparkle.values():
                randgraph.install_steps(r, steps)
            os.environ["FR_STAGE_JIT"] = "force"          # read at renderer creation: every stage program through hipRTC
            modes["staged+jit"] = Renderer(hip, mode="staged", sync_compile=sync)
            del os.environ["FR_STAGE_JIT"]
            for r in modes.values():
                randgraph.install_steps(r, steps)
            for k, (s, e) in enumerate(calls):
                e = e if e is not None else s + T
                n_t = e - s
                if not sync and k in (1, 3):
                    time.sleep(0.14)
                rows = [synth.time_ramp(s, e)[: n_t if k != 1 else int(rng.integers(0, n_t + 1))],
                        (rng.normal(size=n_t) * 3).astype(np.float32)]
                try:
                    exp = ref.fill_buffer(n_out, s, e, rows)
                except RenderError as err:
                    for m, r in modes.items():
                        try:
                            r.fill_buffer(n_out, s, e, rows)
                            print(f"seed {seed} mode {m}: oracle raised {err.status}, engine did not")
                            bad += 1
                        except RenderError as e2:
                            if e2.status != err.status:
                                print(f"seed {seed} mode {m}: status {e2.status} != {err.status}")
                                bad += 1
                    break
                for m, r in modes.items():
                    got = r.fill_buffer(n_out, s, e, rows)
                    if not same_bits(got, exp):
                        print(f"seed {seed} mode {m} call {k}: MISMATCH")
                        bad += 1
                        if only:   # details for a named seed
                            w = np.argwhere((got.view(np.uint32) != exp.view(np.uint32)) & ~(np.isnan(got) & np.isnan(exp)))
                            for r_, c_ in w[:6]:
                                print(f"    row {r_} frame {s + c_}: got {got[r_, c_]!r} ({got.view(np.uint32)[r_, c_]:#x}) expected {exp[r_, c_]!r} ({exp.view(np.uint32)[r_, c_]:#x})")
                            print(f"    {len(w)} samples differ; T={T}; plan {r.plan()}")
                if sparkle:
                    try:
                        exp_s = ref_s.fill_buffer(n_out, s, e, rows)
                        for m, r in sparkle.items():
                            if not same_bits(r.fill_buffer(n_out, s, e, rows), exp_s):
                                print(f"seed {seed} sparkle mode {m} call {k}: MISMATCH")
                                bad += 1
                    except RenderError:
                        for r in sparkle.values():
                            r.close()
                        sparkle = {}
                if k in edit_after:
                    edits = randgraph.random_edits(rng, steps, int(rng.integers(1, 5)), signal_delays=bool(seed % 3))
                    for r in list(modes.values()) + list(sparkle.values()) + [ref, ref_s]:
                        randgraph.install_steps(r, edits)
            for r in list(modes.values()) + list(sparkle.values()):
                r.close()
        if seed % 50 == 49:
            print(f"{seed + 1} graphs, {bad} problems", flush=True)
    print(f"done: {n} graphs x (4 modes + 2 under FR_SEMANTICS_SPARKLE), {bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
