"""One rank of a sharded render (launched by tests/test_distributed.py, world size 2 or 4, gloo).

Usage: dist_worker.py <rank> <world> <port> <mode> <lib: oracle|hip> <V> <P> <T> <outdir>
Each rank renders its shard through the C ABI, the ranks exchange over torch.distributed, and every rank checks the
assembled result bit-for-bit against an unsharded oracle render of the full tree."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    mode, libname = sys.argv[4], sys.argv[5]
    V, P, T = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
    outdir = sys.argv[9]
    import torch
    import torch.distributed as dist
    from libfriendship_amd import shard, synth
    from libfriendship_amd.capi import Renderer, RendererLib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    if libname == "hip":
        import libfriendship_amd
        lib = libfriendship_amd.hip_lib()
    else:
        lib = oracle

    tree, info = shard.additive_tree_shard(V, P, rank, world, mode, seed=77, detune=True)
    if mode == "time":
        start, end = shard.time_stripe(rank, world, T, base=1000)
    else:
        start, end = 1000, 1000 + T
    with Renderer(lib) as r:
        synth.install(r, tree)
        local = r.fill_buffer(tree["n_outputs"], start, end, [synth.time_ramp(start, end)])

    # unsharded reference, on the CPU oracle
    full_tree = synth.additive_tree(V, P, seed=77, detune=True)
    n_frames = T * world if mode == "time" else T
    with Renderer(oracle) as ref:
        synth.install(ref, full_tree)
        expect = ref.fill_buffer(V, 1000, 1000 + n_frames, [synth.time_ramp(1000, 1000 + n_frames)])

    if mode == "partials":
        mixes = shard.all_gather_mixes(torch.from_numpy(local), world)
        got = shard.combine_partial_mixes(mixes).numpy()
    elif mode == "voices":
        parts = [None] * world
        dist.all_gather_object(parts, local)
        got = np.concatenate(parts, axis=0)
    else:
        parts = [torch.empty((V, T), dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(local))
        got = np.concatenate([p.numpy() for p in parts], axis=1)
    ok = got.shape == expect.shape and np.array_equal(got.view(np.uint32), expect.view(np.uint32))
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else f"MISMATCH shape {got.shape} vs {expect.shape}")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
