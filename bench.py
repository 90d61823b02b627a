#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path on MI355X.

Metric (BASELINE.json): Msamples/sec at 4096 partials x 64 voices (one sample = one output frame of
the whole tree, all 64 voice rows), 48 kHz synthetic tree, plus achieved rates against the roofline.

A "step" is one `fill_buffer` call of T = 4800 frames (0.1 s of audio) over the full tree, through the
C ABI's device-resident entry point: the time-ramp input and the output buffer are already in HBM when
the timed region starts (PCIe-inclusive figures are in DESIGN.md, never in `value`).

N > 1 (launched by torch.distributed.run, one rank per GPU): the job is striped over time -- rank r
renders its own contiguous stripe of frames of the same tree.  The evaluator is a pure function of
(graph, input history, t) (reference src/render/reference.rs:178-266), so stripes are independent: no
data-path collective, weak scaling, value = frames rendered by all ranks / max-over-ranks time.

Prints ONE JSON line on stdout (rank 0); diagnostics go to stderr.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X ceilings (/opt/skills/guides/MI355X_MICROARCH.md: chip-level parameters, cycle constants)
HBM_PEAK_GBS = 8000.0                       # spec; 6290 measured copy
VALU_LANE_RATE = 256 * 4 * 32 * 2.4e9       # f32 VALU lane-ops/s: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6e12
FMA_PEAK_TFLOPS = 157.3                     # the same rate counted as FMA (2 flops); unusable here, see DESIGN.md
OPS_EXECUTED_PER_PF = 6                     # VALU ops the fused kernel issues per partial-frame: mul, fract, fma, fma, mul + 1 tree add
ISSUE_SLOTS_PER_PF = 7                      # v_fract_f32 issues at half rate on gfx950 (measured 4.07 vs 2.2 cyc, profiles/r01_valu_rate.txt)
OPS_GRAPH_PER_PF = 12                       # primitive nodes the reference evaluates per partial-frame (11 + Sum2)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(tree, V, P, frames_1t, frames_mt):
    """The CPU path, timed on this box's host cores: the C++ restatement of RefRenderer (oracle/, kind
    'port' -- the Rust reference cannot be built here).  Bounded sample of the same workload."""
    from libfriendship_amd import synth
    from libfriendship_amd.capi import Renderer, RendererLib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_tools
    lib = RendererLib(os.path.join(ROOT, "oracle", "_build", "libfr_oracle.so"))
    ncpu = os.cpu_count() or 1
    with Renderer(lib) as r:
        synth.install(r, tree)
        t = synth.time_ramp(0, frames_1t)
        t0 = time.perf_counter()
        out1 = r.fill_buffer(V, 0, frames_1t, [t])
        dt1 = time.perf_counter() - t0
        nthreads = min(ncpu, V)
        oracle_tools.set_threads(r, nthreads)
        t = synth.time_ramp(frames_1t, frames_1t + frames_mt)
        t0 = time.perf_counter()
        r.fill_buffer(V, frames_1t, frames_1t + frames_mt, [t])
        dtm = time.perf_counter() - t0
    one = {"value": frames_1t / dt1 / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": f"{frames_1t} frames of the same {P}x{V} tree, single thread (the reference renderer is "
                     f"single-threaded); C++ restatement of RefRenderer, not the Rust binary",
           "seconds": dt1}
    many = {"value": frames_mt / dtm / 1e6, "unit": "Msamples/s", "cores": nthreads, "kind": "port",
            "sample": f"{frames_mt} frames, output slots spread over {nthreads} threads", "seconds": dtm,
            "host_cpus": ncpu}
    return one, many, out1


def main():
    # Exactly ONE line may reach stdout.  Libraries write there too (RCCL prints a version banner at communicator
    # creation), so the process's stdout is pointed at stderr for the whole run and the JSON line goes to the saved
    # descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = run()
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
    if line is not None:
        os.write(real_stdout, (line + "\n").encode())
    os.close(real_stdout)


def run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--voices", type=int, default=64)
    ap.add_argument("--partials", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=4800, help="frames per fill_buffer call (T)")
    ap.add_argument("--mode", default="auto", choices=["auto", "pull"])
    ap.add_argument("--shard", default="time", choices=["time", "voices", "partials"],
                    help="how N > 1 ranks split the job (libfriendship_amd/shard.py): time stripes (weak scaling, no exchange; "
                         "default), voices (strong, no exchange), partial blocks (strong, RCCL all-gather + tree-order sum)")
    ap.add_argument("--tree", default="additive", choices=["additive", "effects", "chorus"],
                    help="additive = BASELINE configs[2] shape (the headline); effects = configs[3] shape (detune + ADSR + "
                         "4-tap delay chain), a diagnostic run: use with --voices 128 --partials 1024 --no-cpu-baseline; "
                         "chorus = every voice through a Delay with a SIGNAL amount (LFO) + the 4-tap chain, also a diagnostic")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl is RCCL on ROCm (default)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even with one rank and take the partial-shard exchange path (RCCL plumbing check)")
    ap.add_argument("--wrap-voices", type=int, default=0,
                    help="effects tree only, shape sweeps: fundamentals repeat every N voices (synth.voice_params wrap); the "
                         "survey's 55*2^(v/12) puts voices beyond v~150 above any representable pitch (identically zero mixes)")
    ap.add_argument("--streams", type=int, default=1,
                    help="issue consecutive steps round-robin on this many HIP streams (each with its own output buffer): calls "
                         "of a plan without delay state are independent and the engine lets them overlap on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", "--short-blocks", dest="short_blocks", action="store_true",
                    help="after the timed region also time (a) the same steps overlapped on two streams -> `overlapped_calls` and "
                         "(b) 64- and 512-frame calls (SURVEY.md 8d) -> `short_blocks`; off by default so that the default command "
                         "launches only the timed workload's kernels (rocprof summaries)")
    ap.add_argument("--cpu-frames", type=int, default=144,
                    help="frames the CPU path renders single-threaded for cpu_baseline and the parity check (about 12 s)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import libfriendship_amd
    from libfriendship_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    # (rehearsals on a one-GPU box: several ranks may share device 0 with --backend gloo)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    V, P, T, K, W = args.voices, args.partials, args.frames, args.steps, args.warmup
    t_build = time.perf_counter()
    from libfriendship_amd import shard
    shard_mode = args.shard if use_dist else "time"
    # seeded synthetic tree, ~12 primitive nodes per partial; under voices/partials sharding each rank
    # holds only its sub-graph
    if args.tree in ("effects", "chorus"):
        assert world == 1, "the effects and chorus trees are single-GPU diagnostics"
        tree = synth.effects_tree(V, P, wrap=args.wrap_voices or None) if args.tree == "effects" else synth.chorus_tree(V, P, taps=4)
        shard_info = {"partials": (0, P), "voices": (0, V), "mode": "time"}
    else:
        tree, shard_info = shard.additive_tree_shard(V, P, rank, world, shard_mode)
    full_tree = tree if shard_mode == "time" else None
    V_local = tree["n_outputs"]
    hip = libfriendship_amd.HipRenderer(mode=args.mode, device=local_rank)
    synth.install(hip, tree)
    log(f"[rank {rank}] graph: {len(tree['handles'])} nodes, {len(tree['edges'])} edges, "
        f"installed in {time.perf_counter() - t_build:.1f}s")

    # this rank's stripe of frames; every step's time-ramp row is already resident in HBM.  The ramp is the f32 frame
    # number, exact below 2^24, so its VALUES wrap at 2^23 (idx itself, a u64, keeps counting); rows live in a ring of
    # at most 64 steps so that any --steps fits in memory.
    n_calls = W + K
    stripe0 = rank * n_calls * T if shard_mode == "time" else 0
    ring_steps = min(n_calls, 64)
    WRAP = 1 << 23

    def ramp_row(k):
        f0 = stripe0 + k * T
        return (np.arange(f0, f0 + T, dtype=np.int64) % WRAP).astype(np.float32)

    d_time = torch.from_numpy(np.concatenate([ramp_row(k) for k in range(ring_steps)])).cuda()
    if n_calls > ring_steps:   # later steps reuse ring rows: keep their values consistent with ramp_row(k) only modulo the ring
        log(f"[rank {rank}] {n_calls} calls share a ring of {ring_steps} resident input rows")
    d_out = torch.empty((max(V_local, 1), T), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    n_streams = max(1, args.streams)
    side_streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else []
    side_outs = [torch.empty_like(d_out) for _ in range(n_streams)] if n_streams > 1 else []
    d_mixes = [torch.empty_like(d_out) for _ in range(world)] if shard_mode == "partials" else None

    def step(k):
        r = k % ring_steps
        row = d_time[r * T:(r + 1) * T]
        if side_streams and shard_mode != "partials":
            hip.fill_buffer_device(side_outs[k % n_streams].data_ptr(), V_local, T, stripe0 + k * T, row.data_ptr(), [0, T],
                                   side_streams[k % n_streams].cuda_stream)
            return side_outs[k % n_streams]
        hip.fill_buffer_device(d_out.data_ptr(), V_local, T, stripe0 + k * T, row.data_ptr(), [0, T], stream)
        if shard_mode == "partials":
            # the one exchange step of the path: every rank's partial mix to every rank (RCCL over xGMI), then the
            # top log2(N) levels of the voices' Sum2 trees, pairwise in the graph's own order (bit-exact)
            if args.backend == "nccl":
                dist.all_gather(d_mixes, d_out)
                return shard.combine_partial_mixes(d_mixes)
            host = [torch.empty(d_out.shape, dtype=torch.float32) for _ in range(world)]   # gloo rehearsal: via host
            dist.all_gather(host, d_out.cpu())
            return shard.combine_partial_mixes(host)
        return d_out

    def barrier():
        if use_dist:
            dist.barrier()

    t0 = time.perf_counter()
    for k in range(W):           # first call also lowers the graph and uploads the bank parameters
        step(k)
    torch.cuda.synchronize()
    log(f"[rank {rank}] warmup ({W} steps incl. lowering): {time.perf_counter() - t0:.2f}s; plan: {hip.plan()}")

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        step(k)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    last = (side_outs[(W + K - 1) % n_streams] if side_streams and shard_mode != "partials" else d_out).cpu().numpy()

    # kernel-level timing for the roofline: same steps again with HIP events around every launch, recorded
    # on the stream the kernels run on (engine-side, fr_set_timing); a seek back to the stripe start.
    hip.set_timing(True)
    hip.reset_timing()
    for k in range(W, W + K):   # the first of these is a seek back to the stripe's first timed frame
        step(k)
    torch.cuda.synchronize()
    bank_ms, bank_launches = hip.get_timing("bank")
    all_ms, all_launches = hip.get_timing("all")
    plan = hip.plan()
    hip.set_timing(False)

    # the drop-in entry point (host buffers in, host buffer out: H2D of the ramp, D2H of [V,T] f32, sync per call);
    # reported as an extra, never as `value`
    host_rate = None
    if rank == 0:
        hip.fill_buffer(V_local, stripe0, stripe0 + T, [ramp_row(0)])
        th = time.perf_counter()
        for k in range(1, 6):
            hip.fill_buffer(V_local, stripe0 + k * T, stripe0 + (k + 1) * T, [ramp_row(k)])
        host_rate = 5 * T / (time.perf_counter() - th) / 1e6

    # independent calls overlapped on two streams (an extra, never `value`): for a plan without delay state the engine
    # lets consecutive calls issued on different streams run concurrently, which fills the tail of each launch
    overlapped = None
    if args.short_blocks and rank == 0 and world == 1 and n_streams == 1 and not plan.get("rings") and not plan.get("stage_programs") and not plan.get("pull_rows"):
        s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
        o2 = [torch.empty_like(d_out), torch.empty_like(d_out)]
        base_k = 2 * (W + K) + 4

        def step2(k):
            row = d_time[(k % ring_steps) * T:][:T]
            hip.fill_buffer_device(o2[k % 2].data_ptr(), V_local, T, stripe0 + k * T, row.data_ptr(), [0, T], s2[k % 2].cuda_stream)

        for k in range(base_k, base_k + W):      # the first of these is a seek forward
            step2(k)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for k in range(base_k + W, base_k + W + K):
            step2(k)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t2
        overlapped = {"streams": 2, "value": K * T / e2 / 1e6, "unit": "Msamples/s", "ms_per_step": e2 / K * 1e3,
                      "note": "same K steps issued round-robin on 2 HIP streams with separate output buffers; kernels of consecutive "
                              "calls overlap, so per-launch durations are not comparable with the sequential run above"}

    # short blocks ("also report T in {64, 512}"): latency of one call through the device entry point
    short_blocks = None
    if args.short_blocks and rank == 0 and world == 1 and elapsed / K < 5e-3:   # (not when a call takes milliseconds: pull-mode diagnostics)
        short_blocks = {}
        for tb in (64, 512):
            base = stripe0 + (n_calls + 8) * T
            for k in range(220):
                if k == 20:
                    torch.cuda.synchronize()
                    tb0 = time.perf_counter()
                row = d_time[(k * tb) % (T - tb + 1):][:tb]   # any resident f32 row of the right length will do for timing
                hip.fill_buffer_device(d_out.data_ptr(), V_local, tb, base + k * tb, row.data_ptr(), [0, tb], stream)
            torch.cuda.synchronize()
            us = (time.perf_counter() - tb0) / 200 * 1e6
            short_blocks[str(tb)] = {"us_per_call": us, "msamples_per_s": tb / us}

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return None

    frames_total = (world if shard_mode == "time" else 1) * K * T
    value = frames_total / elapsed / 1e6
    k_lo, k_hi = shard_info["partials"]
    pf_per_launch = float(V_local) * (k_hi - k_lo) * T
    dom_ms, dom_n, dom_name = (bank_ms, bank_launches, "bank_kernel") if bank_launches else (all_ms, all_launches, "pull_kernel")
    avg_s = (dom_ms / max(dom_n, 1)) * 1e-3
    # algorithmic HBM bytes per launch, closed-form model of SURVEY.md 8d: parameters read once + ramp in + samples out
    bytes_per_launch = V_local * (k_hi - k_lo) * 8 + 4 * T + 4 * V_local * T
    valu_rate = OPS_EXECUTED_PER_PF * pf_per_launch / avg_s if avg_s > 0 else 0.0
    roofline = {
        "bound": "valu",
        "kernel": dom_name,
        "achieved": valu_rate / 1e12,
        "peak": VALU_LANE_RATE / 1e12,
        "unit": "TFLOP/s",
        "frac": valu_rate / VALU_LANE_RATE,
        "traffic": None,
        "avg_launch_ms": avg_s * 1e3,
        "launches_timed": int(dom_n),
        "issue_frac": (ISSUE_SLOTS_PER_PF * pf_per_launch / avg_s) / VALU_LANE_RATE if avg_s > 0 else 0.0,
        "flops_per_partial_frame": {"executed_by_kernel": OPS_EXECUTED_PER_PF, "issue_slots": ISSUE_SLOTS_PER_PF,
                                    "primitive_nodes_in_graph": OPS_GRAPH_PER_PF},
        "note": "f32 VALU issue bound; peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz lane-ops/s (the 157.3 TFLOP/s spec counts "
                "each FMA twice). achieved = executed VALU lane-ops/s (6 per partial-frame: exact algebra folds the graph's 12 "
                "primitive ops; 2 of the 6 are FMAs whose single rounding is proved equal to the graph's two). issue_frac "
                "counts v_fract_f32 as 2 slots (half rate on gfx950). The chip holds ~1.9-2.3 GHz under this load, not 2.4. "
                "MFMA not applicable: a wide reduction, no dense contraction.",
        "hbm": {"achieved": bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (bytes_per_launch / avg_s / 1e9) / HBM_PEAK_GBS if avg_s > 0 else 0.0,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "note": "block rendering keeps partial state in registers/SGPRs for 4800 frames: HBM is not the bound"},
    }

    result = {
        "metric": "Msamples/sec at 4096 partials x 64 voices; achieved HBM GB/s vs roofline",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
        "scaling": "weak" if shard_mode == "time" else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"additive tree, {P} partials x {V} voices, 48 kHz, {T}-frame fill_buffer calls "
                                f"(BASELINE.json configs[2])" if args.tree == "additive" else
                                f"harmonics + detune + ADSR + 4-tap delay chain, {P} partials x {V} voices (BASELINE.json configs[3] shape)"
                                if args.tree == "effects" else
                                f"chorus (Delay with an LFO amount) + 4-tap delay chain, {P} partials x {V} voices (diagnostic)"),
                   "voices": V, "partials": P, "frames_per_call": T,
                   "sharding": {"time": "time stripes, one per GPU, no collective",
                                "voices": "voices split over GPUs, no collective",
                                "partials": "partial blocks of every voice split over GPUs; RCCL all-gather of [V,T] partial "
                                            "mixes + tree-order f32 sum"}[shard_mode] if world > 1 else "single GPU",
                   "engine_mode": args.mode, "plan": plan},
        "partial_frames_per_s": K * T * float(V) * P * (world if shard_mode == "time" else 1) / elapsed,
        "roofline": roofline,
        "host_buffer_api_msamples_per_s": host_rate,
        "short_blocks": short_blocks,
        "overlapped_calls": overlapped,
    }
    # HBM traffic per launch from PMC counters: rocprofv3 cannot wrap a process from inside it, so the figure is
    # read from the committed summary of the separate --pmc passes of this same command (profiles/).
    pmc_path = os.path.join(ROOT, "profiles", "r01_bank_pmc_summary.json")
    if (V, P, T) == (64, 4096, 4800) and os.path.exists(pmc_path):
        try:
            with open(pmc_path) as f:
                pmc = json.load(f)
            roofline["traffic"] = pmc["derived"]["hbm_traffic_bytes"]
            roofline["traffic_source"] = "profiles/r01_bank_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except Exception:
            pass

    if not args.no_cpu_baseline and world == 1:
        try:
            one, many, cpu_out = cpu_baseline(full_tree, V, P, args.cpu_frames, 4 * args.cpu_frames)
            result["cpu_baseline"] = one
            result["cpu_baseline_all_cores"] = many
            # parity of the bench's own output against the CPU path on the sampled frames (stripe 0 only)
            chk = libfriendship_amd.HipRenderer(mode=args.mode, device=local_rank)
            synth.install(chk, full_tree)
            got = chk.fill_buffer(V, 0, args.cpu_frames, [synth.time_ramp(0, args.cpu_frames)])
            result["parity"] = {"frames_checked": args.cpu_frames, "voices": V,
                                "bit_exact": bool(np.array_equal(got.view(np.uint32), cpu_out.view(np.uint32)))}
            chk.close()
        except Exception as e:   # the baseline is a reported extra; never lose the GPU line over it
            result["cpu_baseline"] = {"error": repr(e)}
    result["checksum"] = float(np.abs(last.astype(np.float64)).sum())
    if use_dist:
        dist.destroy_process_group()
    return json.dumps(result)


if __name__ == "__main__":
    main()
