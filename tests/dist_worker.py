"""One rank of a sharded render (launched by tests/test_distributed.py; gloo).

Usage: dist_worker.py <rank> <world> <port> <shard mode> <lib: sim|hip> <case> <outdir>
Every rank creates a renderer on `lib`, receives the SAME graph through the ordinary ABI, calls fr_set_shard with a
host-callback transport that carries the exchange over torch.distributed (gloo), renders the same calls, and checks the
rows it owns bit-for-bit against an unsharded oracle render of the full graph."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    mode, libname, case, outdir = sys.argv[4], sys.argv[5], sys.argv[6], sys.argv[7]
    import torch
    import torch.distributed as dist
    from libfriendship_amd import synth
    from libfriendship_amd.capi import Renderer, RendererLib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    if libname == "hip":
        import libfriendship_amd
        lib = libfriendship_amd.hip_lib()
    else:
        import sim_tools
        lib = RendererLib(sim_tools.OUT)       # built once by the launching test

    def sendrecv(peer, send, recv):
        """The exchange step's transport: host buffers over gloo."""
        reqs, rt = [], None
        if send is not None:
            reqs.append(dist.isend(torch.from_numpy(send.copy()), peer))
        if recv is not None:
            rt = torch.empty(recv.size, dtype=torch.uint8)
            reqs.append(dist.irecv(rt, peer))
        for q in reqs:
            q.wait()
        if recv is not None:
            recv[:] = rt.numpy()

    edits = []
    if case.endswith("_tiled"):      # the exchange cut into time tiles on a second stream (engine.cpp execute()): small tiles for a small test
        os.environ["FR_EXCHANGE_MIN_TILE"] = "64"
        case = case[:-len("_tiled")]
        tiled = True
    else:
        tiled = False
    if case == "additive" and tiled:
        V, T = 5, 300
        tree = synth.additive_tree(V, 64 * world, seed=77, detune=True)
        install = lambda r: synth.install(r, tree)
        calls = [(1000, 1000 + T), (1000 + T, 1000 + 2 * T), (1000 + 2 * T, 1000 + 2 * T + 77), (50, 50 + T)]
    elif case == "effects" and tiled:   # rings behind the exchange: contiguous calls, a seek (window exchanged in tiles), an edit
        V, T = 4, 300
        tree = synth.effects_tree(V, 64 * world, taps=3, base_delay=40.0)
        install = lambda r: synth.install(r, tree)
        calls = [(0, T), (T, 2 * T), (2 * T, 2 * T + 130), (7000, 7000 + T), (7000 + T, 7000 + 2 * T), (7000 + 2 * T, 7000 + 3 * T)]
        e = tree["edges"]
        last = [int(x) for x in e[(e[:, 1] == 0) & (e[:, 3] == V - 1)][0]]
        prev = [int(x) for x in e[(e[:, 1] == 0) & (e[:, 3] == V - 2)][0]]
        edits = [(5, [("del", last), ("add", [prev[0], 0, 0, V - 1])])]
    elif case == "additive":
        V, T = 5, 48
        tree = synth.additive_tree(V, 64 * world, seed=77, detune=True)
        install = lambda r: synth.install(r, tree)
        calls = [(1000, 1000 + T), (1000 + T, 1000 + 2 * T)]
    elif case == "effects":      # config D's shape: voices -> envelope -> delay chain; contiguous calls, a seek, an edit
        V, T = 4, 64
        tree = synth.effects_tree(V, 64 * world, taps=3, base_delay=40.0)
        install = lambda r: synth.install(r, tree)
        calls = [(0, T), (T, 2 * T), (2 * T, 3 * T), (7000, 7000 + T), (7000 + T, 7000 + 2 * T), (7000 + 2 * T, 7000 + 3 * T)]
        e = tree["edges"]
        last = [int(x) for x in e[(e[:, 1] == 0) & (e[:, 3] == V - 1)][0]]
        prev = [int(x) for x in e[(e[:, 1] == 0) & (e[:, 3] == V - 2)][0]]
        edits = [(5, [("del", last), ("add", [prev[0], 0, 0, V - 1])])]      # before call 5
    elif case == "triangle":     # voices of a NON-template leaf (hipRTC-specialised on the device): cut by matching the sub-roots
        V, T = 4, 64
        g = synth.GraphArrays()
        p = synth.voice_params(3, 128 * world, seed=5, detune=True)
        roots = synth.sum_tree(g, synth.triangle_leaves(g, p["w"], p["amp"], am_slot=1).reshape(3, 128 * world))
        g.edge(roots, 0, 0, np.arange(3, dtype=np.uint32))
        d = g.binop(synth.K_SUM2, roots[0:1], g.binop(synth.K_DELAY, roots[0:1], synth.C(np.float32(37.0)), 1), 1)
        g.edge(d, 0, 0, 3)          # voice 0 also feeds row 3 through a delay: needed by two ranks at world 2+ -> stays whole
        tree = g.finish(V)
        install = lambda r: synth.install(r, tree)
        calls = [(0, T), (T, 2 * T), (500, 500 + T)]
    elif case == "random":       # arbitrary graphs (composites, signal delays, pull rows)
        import randgraph
        V, T = 5, 40
        steps, _ = randgraph.random_graph(4242, n_nodes=30, n_inputs=2, n_outputs=V)
        install = lambda r: randgraph.install_steps(r, steps)
        calls = [(0, T), (T, 2 * T), (900, 900 + T)]
    else:
        raise SystemExit(f"unknown case {case}")

    rng = np.random.default_rng(9)
    ok, why = True, ""
    with Renderer(lib) as r, Renderer(oracle) as ref:
        install(r)
        install(ref)
        r.set_shard(rank, world, mode, sendrecv=sendrecv)
        lo, hi = r.shard_rows(V)
        for ci, (start, end) in enumerate(calls):
            for when, ops in edits:
                if when == ci:
                    for kind, ed in ops:
                        for x in (r, ref):
                            (x.on_del_edge if kind == "del" else x.on_add_edge)(*ed)
            rows = [synth.time_ramp(start, end), (rng.normal(size=end - start) * 3).astype(np.float32)]
            exp = ref.fill_buffer(V, start, end, rows)
            out = np.full((V, end - start), np.float32(-777.0))
            got = r.fill_buffer(V, start, end, rows, out=out)
            mine = got[lo:hi].view(np.uint32) == exp[lo:hi].view(np.uint32)
            mine |= np.isnan(got[lo:hi]) & np.isnan(exp[lo:hi])
            others = np.ones(V, bool)
            others[lo:hi] = False
            if not mine.all():
                ok, why = False, f"call {ci} [{start},{end}): rows {lo}..{hi} differ from the unsharded oracle"
            elif not np.all(got[others] == np.float32(-777.0)):
                ok, why = False, f"call {ci}: wrote rows it does not own"
            if not ok:
                break
        plan = r.plan()
    def owner(row):   # contiguous blocks, sizes differing by at most one (fr_shard_rows)
        q, rem = divmod(V, world)
        return row // (q + 1) if row < rem * (q + 1) else rem + (row - rem * (q + 1)) // max(q, 1)
    # the edit leaves the last voice unreachable and makes voice V-2 feed rows V-2 and V-1: needed by two ranks (and then
    # rendered whole by both) when those rows have different owners
    want = V if not edits else (V - 1 if owner(V - 2) == owner(V - 1) else V - 2)
    if case == "triangle":
        want = plan["shard"]["split_voices"]      # (2 on the device, where the leaves get a kernel; 0 where they do not)
        if libname == "hip" and mode == "partials" and want != 2:
            ok, why = False, f"expected the two single-owner triangle voices to be split: {plan['shard']}"
    if ok and mode == "partials" and case != "random" and plan["shard"]["split_voices"] != want:
        ok, why = False, f"expected {want} split voices: {plan['shard']}"
    if ok and tiled and not plan["exchange_stats"]["tiles"] > 2 * plan["exchange_stats"]["calls"]:
        ok, why = False, f"expected the exchange in several tiles per call: {plan['exchange_stats']}"
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write("ok" if ok else why)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
