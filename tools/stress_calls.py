#!/usr/bin/env python3
"""Parity stress for the INPUT STORE and call sequencing (reference.rs:47-75): long random call sequences on random
graphs -- call lengths 1..300, forward and backward seeks, rows full / short / missing / too long / arriving for a slot
whose history is out of step (both sides must refuse the call; the next call is then a seek), graph edits in between -- through
every engine mode, with and without a history bound (fr_config.history_frames >= the plan's look-back must change
nothing), through the host entry point with and without a registered destination.  Bit-compared with the CPU oracle.
usage: python tools/stress_calls.py [n_seeds [first_seed]]      (FR_STRESS_LIB=sim: the host-logic simulator, CPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import randgraph  # noqa: E402
from kat_replay import same_bits  # noqa: E402
import libfriendship_amd  # noqa: E402
from libfriendship_amd.capi import RenderError, Renderer, RendererLib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = [int(x) for x in os.environ["FR_STRESS_SEEDS"].split(",")] if os.environ.get("FR_STRESS_SEEDS") else None
    if os.environ.get("FR_STRESS_LIB") == "sim":
        import sim_tools
        hip = sim_tools.sim_lib()
    else:
        hip = libfriendship_amd.hip_lib()
    oracle = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    bad = calls_done = refused = 0
    for i, seed in enumerate(only or range(first, first + n)):
        rng = np.random.default_rng(seed)
        n_in, n_out = 3, 3
        steps, _ = randgraph.random_graph(30_000 + seed, n_nodes=int(rng.integers(3, 30)), n_inputs=n_in, n_outputs=n_out,
                                          signal_delays=bool(seed % 2))
        engines = {"auto": Renderer(hip, mode="auto"), "staged": Renderer(hip, mode="staged"), "pull": Renderer(hip, mode="pull")}
        if not os.environ.get("FR_CALLS_NO_BOUNDED"):
            engines["auto, bounded history"] = Renderer(hip, mode="auto", history_frames=1 << 14)
        reg_out = np.zeros((n_out, 300), np.float32)
        registered = not os.environ.get("FR_CALLS_NO_REG")
        if registered:
            engines["auto"].host_register(reg_out)
        if os.environ.get("FR_CALLS_VERBOSE"):
            print(f"seed {seed}", flush=True)
        with Renderer(oracle) as ref:
            for r in [ref] + list(engines.values()):
                randgraph.install_steps(r, steps)
            base = int(os.environ.get("FR_CALLS_BASE", "0"))   # e.g. 2**40: sample indices far beyond 32 bits (and beyond f32's integers)
            head = base
            ok = True
            after_refusal = False
            for k in range(40):
                T = int(rng.choice([1, 2, 7, 64, 65, 100, 300]))
                if after_refusal or rng.random() < 0.15:      # a seek, either direction
                    head = int(rng.choice([h for h in (base + int(rng.integers(0, head - base + 3000)), head + 1 + T) if h != head]))
                    after_refusal = False
                kind = rng.random()
                rows = []
                for slot in range(int(rng.integers(0, n_in + 1))):          # trailing rows may be missing altogether
                    ln = T if rng.random() < 0.7 else int(rng.integers(0, T + 1))
                    rows.append((rng.normal(size=ln) * 4).astype(np.float32) if slot else
                                np.arange(head, head + ln, dtype=np.float64).astype(np.float32))
                if kind < 0.05 and rows:                      # a row longer than the call (reference.rs:71)
                    rows[-1] = np.zeros(T + int(rng.integers(1, 5)), np.float32)
                try:
                    exp = ref.fill_buffer(n_out, head, head + T, rows)
                    err = None
                except RenderError as e:
                    exp, err = None, e.status
                    refused += 1
                for name, eng in engines.items():
                    try:
                        if name == "auto" and k % 3 == 0:     # the same call into a destination the host registered once
                            out = reg_out[:, :T] if T == 300 else None
                            got = eng.fill_buffer(n_out, head, head + T, rows, out=out)
                        else:
                            got = eng.fill_buffer(n_out, head, head + T, rows)
                        st = None
                    except RenderError as e:
                        got, st = None, e.status
                    calls_done += 1
                    if name.endswith("bounded history") and st is None:
                        pl = eng.plan()   # a look-back the bound does not cover is documented to read zeros: not comparable
                        if pl["input_lookback_unbounded"] or pl["input_lookback"] + 300 > (1 << 14):
                            continue
                    if st != err or (err is None and not same_bits(got, exp)):
                        bad += 1
                        ok = False
                        what = f"status {st} vs oracle {err}" if st != err else "MISMATCH"
                        print(f"seed {seed} engine '{name}' call {k} (idx {head}, T {T}, rows {[len(r) for r in rows]}): {what}", flush=True)
                if not ok:
                    break
                if err is None:
                    head += T
                else:
                    # Where the reference panics it has already stored the rows before the offending one (reference.rs:66-74
                    # works row by row) and there is no "afterwards" to compare with; the oracle reproduces that half-done
                    # state, the engine refuses the call as a whole.  A seek makes every stored row `idx` zeros on both
                    # sides (reference.rs:52-58), so the call after a refusal is one.
                    after_refusal = True
                if rng.random() < 0.1:
                    edits = randgraph.random_edits(rng, steps, int(rng.integers(1, 4)), n_inputs=n_in, n_outputs=n_out, signal_delays=bool(seed % 2))
                    for r in [ref] + list(engines.values()):
                        randgraph.install_steps(r, edits)
        if registered:
            engines["auto"].host_unregister(reg_out)
        for e in engines.values():
            e.close()
        if i % 25 == 24:
            print(f"{i + 1} graphs, {calls_done} engine calls ({refused} refused by the oracle), {bad} problems", flush=True)
    print(f"done: {n if not only else len(only)} graphs, {calls_done} engine calls ({refused} refused by the oracle), {bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
