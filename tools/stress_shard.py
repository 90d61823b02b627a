#!/usr/bin/env python3
"""Parity stress for the sharded engine: tests/test_shard_sim.py's random patches (voices of assorted sizes and tree shapes
behind gains, a shared envelope, delay taps; rows wired at random) rendered by `world` renderers in one process -- a thread
per rank, the exchange through the fr_comm host callback -- under voice sharding (2, 3 ranks) and partial-block sharding
(2, 4, 8 ranks), contiguous calls and a seek, every rank's rows bit-equal to the unsharded oracle.
usage: python tools/stress_shard.py [n_seeds [first_seed]]    (on the GPU: all ranks share the one device;
                                                              FR_STRESS_LIB=sim: the host-logic simulator, CPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import libfriendship_amd  # noqa: E402
from libfriendship_amd.capi import RendererLib  # noqa: E402
import test_shard_sim  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    if os.environ.get("FR_STRESS_LIB") == "sim":
        import sim_tools
        lib = sim_tools.sim_lib()
    else:
        lib = libfriendship_amd.hip_lib()
    oracle = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    bad = 0
    for i, seed in enumerate(range(first, first + n)):
        try:
            test_shard_sim.test_random_patches_sharded_every_way(lib, oracle, seed)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: {str(e)[:400]}", flush=True)
        if i % 20 == 19:
            print(f"{i + 1} patches x 5 shardings, {bad} problems", flush=True)
    print(f"done: {n} patches x 5 shardings, {bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
