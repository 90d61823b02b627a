"""Multi-GPU sharding of the render path, done at graph level on the host.

The reference evaluator is a pure function value(edge, t) of (graph, input history, t)
(reference src/render/reference.rs:178-266), and output slots are independent (reference.rs:78-82).
So a job splits three ways, none of which needs engine support beyond the ordinary C ABI:

  time      rank r renders its own contiguous stripe of frames of the whole tree.  No exchange.
  voices    rank r gets the sub-graph of its output slots (voices v with v % N == r ... here: contiguous
            blocks).  No exchange; rows are concatenated by the caller if it wants one buffer.
  partials  rank r gets, for EVERY voice, the sub-tree holding partials [r*P/N, (r+1)*P/N) -- a complete
            sub-tree of the balanced Sum2 tree when N is a power of two -- and renders its partial mix
            [V, T].  One exchange step: all-gather of the N partial mixes (RCCL over xGMI on GPUs, gloo in
            the CPU tests), then the top log2(N) tree levels are summed pairwise IN THE GRAPH'S OWN ORDER
            ((s0+s1)+(s2+s3))+... with plain f32 adds.  That keeps the result bit-identical to the
            unsharded render; a ring all-reduce would not (its association differs).

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  This module only builds graphs
and orders the exchange; the compute is whatever renderer the caller drives through the C ABI.
"""
import numpy as np

from . import synth


def is_pow2(n):
    return n >= 1 and (n & (n - 1)) == 0


def time_stripe(rank, world, frames_per_rank, base=0):
    """[start, end) of rank's stripe."""
    s = base + rank * frames_per_rank
    return s, s + frames_per_rank


def voice_block(rank, world, n_voices):
    """Contiguous block of voices owned by `rank` (sizes differ by at most one)."""
    q, r = divmod(n_voices, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def additive_tree_shard(n_voices, n_partials, rank, world, mode, seed=0x5EED0002, detune=False, sr=48000.0):
    """The sub-graph rank `rank` renders for a V x P additive tree under `mode`.

    Returns (tree, info).  mode 'time': the whole tree.  mode 'voices': only this rank's voices (output
    slot i = voice lo+i).  mode 'partials': every voice, restricted to this rank's block of partials; output
    slot v carries that block's sub-tree sum (a partial mix, to be combined with combine_partial_mixes)."""
    p = synth.voice_params(n_voices, n_partials, seed, detune, sr)
    if mode == "time":
        v_lo, v_hi, k_lo, k_hi = 0, n_voices, 0, n_partials
    elif mode == "voices":
        v_lo, v_hi = voice_block(rank, world, n_voices)
        k_lo, k_hi = 0, n_partials
    elif mode == "partials":
        if not (is_pow2(world) and is_pow2(n_partials) and n_partials >= world):
            raise ValueError("partial-block sharding needs power-of-two world size and partial count")
        v_lo, v_hi = 0, n_voices
        blk = n_partials // world
        k_lo, k_hi = rank * blk, (rank + 1) * blk
    else:
        raise ValueError(mode)
    g = synth.GraphArrays()
    nv, nk = v_hi - v_lo, k_hi - k_lo
    if nv and nk:
        leaves = synth.partial_leaves(g, p["w"][v_lo:v_hi, k_lo:k_hi], p["amp"][v_lo:v_hi, k_lo:k_hi]).reshape(nv, nk)
        roots = synth.sum_tree(g, leaves)
        g.edge(roots, 0, 0, np.arange(nv, dtype=np.uint32))
    tree = g.finish(nv)
    tree["params"] = p
    return tree, {"voices": (v_lo, v_hi), "partials": (k_lo, k_hi), "mode": mode}


def combine_partial_mixes(mixes):
    """Top levels of the voices' Sum2 trees over the ranks' partial mixes, pairwise in tree order.

    `mixes`: sequence of N same-shaped f32 arrays or torch tensors, index = rank.  Uses only elementwise
    f32 `+` (exactly rounded in numpy and torch alike), so the result has the bits of the unsharded tree."""
    level = list(mixes)
    if not is_pow2(len(level)):
        raise ValueError("need a power-of-two number of partial mixes")
    while len(level) > 1:
        level = [level[i] + level[i + 1] for i in range(0, len(level), 2)]
    return level[0]


def all_gather_mixes(local_mix, world):
    """All-gather the ranks' [V, T] partial mixes (torch tensor on the backend's device)."""
    import torch
    import torch.distributed as dist
    out = [torch.empty_like(local_mix) for _ in range(world)]
    dist.all_gather(out, local_mix.contiguous())
    return out
