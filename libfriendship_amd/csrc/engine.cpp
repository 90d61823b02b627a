// engine.cpp -- the HIP render engine behind include/friendship_render.h.
//
// State contract (reference src/render/reference.rs:21-29): mirrored graph + per-slot input history +
// `head`.  Everything else held here (lowered graph, device tables, bank parameters) is a cache of
// that state, rebuilt when the mirror's version or the number of rendered slots changes.
//
// There is no CPU evaluation path in this file: if no gfx950 device is usable, creating a renderer
// fails with FR_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>

#include "comm.hpp"
#include "graph.hpp"
#include "jit.hpp"
#include "kernels.hpp"
#include "match.hpp"
#include "stage.hpp"

namespace fr {

#define HIP_CHECK(expr)                                                                               \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            throw Error(_e == hipErrorOutOfMemory ? FR_ERR_OUT_OF_MEMORY : FR_ERR_DEVICE,             \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                           \
    } while (0)

// Device allocation owned by the engine.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    void ensure(size_t n) {   // contents are NOT preserved
        if (n <= bytes) return;
        release();
        size_t want = std::max(n, (size_t)256);
        HIP_CHECK(hipMalloc(&p, want));
        bytes = want;
    }
    template <class T> T *as() const { return (T *)p; }
};

// Page-locked host staging memory (DMA without the runtime's own bounce copies; async copies stay async).
struct PinnedBuf {
    void *p = nullptr;
    void *dev = nullptr;     // the same memory as the GPU addresses it (mapped: kernels read and write it over PCIe)
    size_t bytes = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    void ensure(size_t n) {
        if (n <= bytes) return;
        if (p) (void)hipHostFree(p);
        p = dev = nullptr;
        bytes = 0;
        size_t want = std::max(n, (size_t)4096);
        HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocMapped));
        HIP_CHECK(hipHostGetDevicePointer(&dev, p, 0));
        bytes = want;
    }
    template <class T> T *as() const { return (T *)p; }
    template <class T> T *as_dev() const { return (T *)dev; }
};

// One external input slot's history on the device (reference.rs:25 `inputs[slot]`).
struct InSlot {
    bool fed = false;        // ever received a row (otherwise implicit zeros)
    uint64_t base = 0;       // zero prefix (seek / late creation), data[i] is time base+i
    uint64_t len = 0;        // logical length (== base + stored samples)
    DevBuf buf;
    uint64_t cap = 0;        // capacity in floats
};

struct BankStage {
    BankLaunch grp;          // rows/params kept on the host for the plan description
    DevBuf d_params, d_rows, d_groups, d_group_off;
    std::shared_ptr<JitKernel> jit;   // grp.jit: the hipRTC specialisation
};

struct Plan {
    bool valid = false;
    uint64_t version = 0;
    uint64_t shard_epoch = 0;            // fr_set_shard generation the plan was made for
    bool jit_pending = false;            // planned without a kernel that hipRTC is still compiling in the background:
    uint64_t jit_epoch = 0;              // stale as soon as the cache's epoch moves on
    uint32_t n_slots = 0;
    uint32_t max_depth = 0;              // of the lowered graph (pull stack sizing)
    std::vector<BankStage> banks;
    StagedPlan sp;                       // programs / levels / rings (banks moved into `banks`)
    DevBuf d_instrs, d_progs;
    std::shared_ptr<JitKernel> stage_jit;   // compiled form of the programs (null: interpreted by stage_kernel)
    DevBuf d_jprogs, d_ptab;
    uint32_t stage_shapes = 0;
    bool stage_valid = false;            // rings hold [stage_end - lmax, stage_end) of the current graph + history
    uint64_t stage_end = 0;
    std::vector<uint32_t> pull_rows;     // output rows evaluated by the pull interpreter
    DevBuf d_split_dst;                  // partial-block sharding: destination (row, or ring | 1 << 31) per exchange workspace row
    DevBuf d_nodes, d_roots;             // pull: nodes (input slots remapped dense), roots per pull row
    std::vector<uint32_t> input_slots;   // dense input index -> external slot
    std::string json;
};

struct TimerClass {
    double ms = 0;
    uint64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace fr

using namespace fr;

struct fr_renderer {
    int device = 0;
    int mode = FR_MODE_AUTO;
    int semantics = FR_SEMANTICS_REFERENCE;
    uint64_t history_frames = 0;         // fr_config: 0 = keep everything since the last seek (the reference)
    hipStream_t stream = nullptr;
    Mirror mirror;
    // input history bookkeeping, same rules as reference.rs:47-75 (see oracle/ref_renderer.cpp)
    std::vector<InSlot> slots;
    uint64_t n_vecs = 0;
    struct Seg { uint64_t first, last, len; };
    std::vector<Seg> segs;
    uint64_t head = 0;
    Plan plan;
    DevBuf d_out, d_in_table, d_stack_node, d_stack_time, d_stack_val, d_bank_ws, d_tickets;
    // host-buffer entry point: input rows go up through pinned staging (one region per row, no sync between rows); the
    // finished frames come down through h_out_stage, which the kernels write DIRECTLY (mapped pinned memory: the stores
    // travel over PCIe while the launch is still computing), then one wait and one CPU copy into the caller's buffer
    PinnedBuf h_in_stage, h_out_stage;
    // Tracks (fr_set_track_inputs): input slots >= track_from are control-rate rows -- per-partial frequency / amplitude
    // envelopes -- that are NOT stored: the leaves of shape-matched voices read them from the call's own dense input matrix
    // (reference.rs:66-74: `inputs` is an Array2, one row per slot), 8 bytes per partial-frame straight from HBM.
    uint32_t track_from = 0xFFFFFFFFu;
    uint32_t dense_total_rows = 0;          // set around a dense call: rows of the caller's matrix
    const float *call_tracks = nullptr;     // row of slot 0 if the matrix started there (never dereferenced below track_from)
    uint64_t call_track_stride = 0, call_track_rows = 0;
    DevBuf d_tracks_stage;                  // host-buffer calls: the rows' copy in HBM
    // shape-matched voices rendered in pieces (few voices x short call): the pieces' sums and the identity row list
    DevBuf d_chunk_ws, d_chunk_rows;
    uint32_t d_chunk_rows_n = 0;
    bool jit_chunks = true;                 // FR_JIT_CHUNKS=0: one workgroup per (voice, tile) always (A/B)
    uint32_t stage_block_env = 0;           // FR_STAGE_BLOCK: iterations per block of compiled strided programs (A/B; 0 = the rule in build_plan)
    uint64_t jit_chunk_target = 0;          // FR_JIT_CHUNK_TARGET: workgroups below which a voice is cut further (0: 1024, tracks 16384)
    // FR_HOST_MAPPED (A/B): bit 0 = kernels write the output through the mapping, bit 1 = the bank kernel reads the
    // input row through the mapping; 0 = the staged copies of round 1 (H2D row, D2H of the whole buffer)
    bool host_out_mapped = false, host_rows_mapped = true;
    size_t host_small_bytes = 96u << 10; // FR_HOST_SMALL_KB: results up to this size leave through mapped pinned memory
    bool host_direct = true;             // FR_HOST_DIRECT=0: registered destinations are filled by a D2H copy like any other
    // Streamed output of the host entry point (plans whose rows all come straight from the time-major bank kernel): the
    // kernels store into mapped pinned memory and publish a flag per finished row; the host copies rows into the
    // caller's pageable buffer WHILE the launch is still computing the others.  FR_HOST_STREAM=0 turns it off.
    bool host_stream = true;
    PinnedBuf h_row_flags;               // [n_slots] u32: the call's sequence number once the row is complete
    DevBuf d_row_done;                   // per-voice tile counters of the launch (zero between launches)
    uint32_t host_seq = 0;
    std::vector<uint32_t> stream_pending;
    struct FlagOut { uint32_t *host_flags = nullptr; uint32_t *row_done = nullptr; uint32_t value = 0; } flag_out;
    // every output row is a voice of a bank launch that can publish its completion (kernels.hpp bank_publishes_rows)?
    bool can_stream_rows(uint32_t n_slots, uint64_t n_times) const {
        if (!host_stream || sharded() || !plan_current(n_slots) || plan.banks.empty() || bank_leaf_variant != 1) return false;
        if (!plan.sp.progs.empty() || plan.sp.uses_rings() || !plan.pull_rows.empty() || !plan.sp.split.empty()) return false;
        size_t voices = 0;
        for (const BankStage &bs : plan.banks) {
            if (bs.grp.jit || bs.grp.general || bs.grp.to_ring || bs.grp.to_ws) return false;
            BankArgs a{};
            a.log2_p = bs.grp.log2_p;
            a.n_voices = (uint32_t)bs.grp.rows.size();
            bank_shape(a.log2_p, a.n_voices, n_times, a.chunk_log2, a.frames_per_lane, a.waves_per_group, a.small_call, a.voices_per_wave);
            if (a.voices_per_wave && !allow_multi) { a.voices_per_wave = 0; a.frames_per_lane = 1; }
            a.leaf_variant = 1;
            a.host_flags = (uint32_t *)1;   // (asking "would it")
            if (!bank_publishes_rows(a)) return false;
            voices += bs.grp.rows.size();
        }
        return voices == n_slots;   // (each row is written by exactly one voice: a row fed by nothing would be a program)
    }
    std::vector<std::pair<char *, size_t>> registered;   // fr_host_register: page-locked, device-visible host ranges
    bool host_trace = false;             // FR_HOST_TRACE=1: phase times of fr_fill_buffer on stderr at destroy
    double trace_us[3] = {0, 0, 0};
    uint64_t trace_n = 0;
    // Device-entry calls return before their work is done; a following call on ANOTHER stream (or the host entry
    // point, which uses the renderer's own stream) must still see this one's history, rings and plan uploads.
    hipEvent_t ev_last = nullptr;
    hipStream_t last_stream = nullptr;   // stream of the last asynchronous call while its work may still be running
    bool last_pending = false;
    bool last_independent = false;       // the last asynchronous call left nothing behind that a later call reads or reuses
    bool used_scratch = false;           // this call used a buffer shared between calls (the chunk workspace)
    bool host_pipelines = false;         // this call came on another stream than the previous, still pending one, and is independent of it
    bool overlapped_streams = false;     // independent calls were let loose on more than one stream since the last ordering point
    // Calls of a plan without delay lines, programs or pull rows touch only their own input rows and output buffer (and
    // append their own, disjoint part of the input history): on different streams they may overlap on the device.
    bool plan_is_stateless(uint32_t n_slots) const {
        return plan_current(n_slots) && !plan.sp.uses_rings() && plan.sp.progs.empty() && plan.pull_rows.empty() && plan.sp.split.empty();
    }
    void order_after_previous(hipStream_t st, bool this_independent = false) {
        if (!last_pending) return;
        if (overlapped_streams && !this_independent) {    // several streams may hold unfinished calls: wait for all of them
            HIP_CHECK(hipDeviceSynchronize());
            overlapped_streams = false;
            last_pending = false;
            return;
        }
        if (last_stream == st) return;                    // the usual case, one stream: nothing to do, nothing queued
        if (last_independent && this_independent) {       // independent work on another stream: let it overlap
            overlapped_streams = true;
            return;
        }
        // (an event recorded after EVERY call would put a barrier packet between consecutive kernels: ~2 us per call)
        if (!ev_last) HIP_CHECK(hipEventCreateWithFlags(&ev_last, hipEventDisableTiming));
        if (hipEventRecord(ev_last, last_stream) == hipSuccess) {
            HIP_CHECK(hipStreamWaitEvent(st, ev_last, 0));
        } else {                                          // the caller destroyed that stream: wait for whatever is left
            (void)hipGetLastError();
            HIP_CHECK(hipDeviceSynchronize());
        }
        last_pending = false;
    }
    void remember_async(hipStream_t st, bool independent) {
        last_independent = independent;
        last_stream = st;
        last_pending = true;
    }
    // ---- sharding (friendship_render.h fr_shard): this renderer is rank `shard.rank` of `shard.world`, one per GPU ----
    ShardSpec shard;
    uint32_t shard_flags = 0;
    uint64_t shard_epoch = 0;
    std::unique_ptr<Transport> rccl;     // the engine's own communicator (device to device over xGMI), or
    fr_comm host_comm{};                 // the host's callback (ranges staged through pinned memory)
    bool has_host_comm = false;
    DevBuf d_ws, d_xrecv;                // exchange workspace [tile][split voices][tile frames], receive buffer of the same shape
    // Time-tiled exchange: the window is cut into tiles; tile i's exchange runs on `xstream` while the bank kernels
    // of tile i + 1 run on the call's stream.  FR_EXCHANGE_TILES (most tiles per call, default 4; 1 = serial, as does the
    // FR_SHARD_SERIAL_EXCHANGE flag), FR_EXCHANGE_MIN_TILE (fewest frames worth a tile, default 1024).
    hipStream_t xstream = nullptr;
    std::vector<hipEvent_t> x_events;    // [tile] banks of the tile done (call's stream) ... and x_events.back(): exchange done (xstream)
    uint32_t x_max_tiles = 4, x_min_tile = 1024;
    bool x_tiles_explicit = false;       // either knob came from the environment: tile whatever the transport
    uint64_t exchange_bytes = 0;         // sent by this rank since the renderer was made (fr_plan_json)
    uint64_t exchange_calls = 0, exchange_tiles = 0;
    PinnedBuf h_xsend, h_xrecv;
    bool sharded() const { return shard.world > 1 && shard.mode != FR_SHARD_NONE; }
    void my_rows(uint32_t n_slots, uint32_t &lo, uint32_t &hi) const {
        lo = 0;
        hi = n_slots;
        if (sharded()) shard_row_range(shard.rank, shard.world, n_slots, lo, hi);
    }
    bool plan_current(uint32_t n_slots) const {
        return plan.valid && plan.version == mirror.version && plan.n_slots == n_slots && plan.shard_epoch == shard_epoch &&
               !(plan.jit_pending && plan.jit_epoch != jit_cache.epoch());
    }

    // Pairwise exchange with `peer` on stream st (device pointers, counts in floats).
    void xfer(uint32_t peer, const float *d_send, size_t n_send, float *d_recv, size_t n_recv, hipStream_t st) {
        if (!n_send && !n_recv) return;
        if (rccl) { rccl->sendrecv(peer, d_send, n_send, d_recv, n_recv, st); return; }
        if (!has_host_comm) throw Error(FR_ERR_COMM, "the sharded plan needs an exchange but fr_set_shard was given no transport");
        h_xsend.ensure(n_send * sizeof(float));
        h_xrecv.ensure(n_recv * sizeof(float));
        if (n_send) HIP_CHECK(hipMemcpyAsync(h_xsend.p, d_send, n_send * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));   // the send range is on the host; an earlier upload from h_xrecv is done
        const int32_t rc = host_comm.sendrecv(host_comm.ctx, peer, h_xsend.p, n_send * sizeof(float), h_xrecv.p, n_recv * sizeof(float));
        if (rc != 0) throw Error(FR_ERR_COMM, "transport callback failed with code " + std::to_string(rc) + " exchanging with rank " + std::to_string(peer));
        if (n_recv) HIP_CHECK(hipMemcpyAsync(d_recv, h_xrecv.p, n_recv * sizeof(float), hipMemcpyHostToDevice, st));
    }

    // The one exchange step of the path (FR_SHARD_PARTIALS): recursive halving over the ranks.  The workspace holds, for
    // every split voice, this rank's sub-tree sum over the window; rows are ordered by the owner's bits, lowest bit
    // first.  Step j pairs rank with rank ^ 2^j: each keeps the rows whose owner agrees with it in bit j, sends the
    // rest, and adds what it receives -- left operand from the rank whose bit j is 0: that is the voice's Sum2 node j
    // levels above the sub-tree roots, evaluated with the graph's own operands in the graph's own order.  After
    // log2(world) steps each rank holds the finished voices it owns; the last add stores them where the unsharded
    // plan would (ring or output row).
    // One time tile of the window: frames [x0 + off, x0 + off + len) of every split voice.  The workspaces are tile-major
    // ([tile][split voice][tile frame]), so a tile's rows are contiguous ranges for the transport and tiles in flight on the
    // exchange stream never share a byte with the tile the bank kernels are writing.
    void run_exchange(float *d_dst, uint64_t n_times, uint64_t idx, uint64_t x0, uint64_t off, uint64_t len, hipStream_t st) {
        const std::vector<SplitVoice> &sv = plan.sp.split;
        uint32_t k = 0;
        while ((1u << k) < shard.world) ++k;
        size_t lo = 0, hi = sv.size();
        float *ws = d_ws.as<float>() + sv.size() * off;
        float *rbuf = d_xrecv.as<float>() + sv.size() * off;
        for (uint32_t j = 0; j < k; ++j) {
            const uint32_t bit = 1u << j, peer = shard.rank ^ bit;
            size_t mid = lo;
            while (mid < hi && !(sv[mid].owner & bit)) ++mid;
            const bool upper = (shard.rank & bit) != 0;
            const size_t keep_lo = upper ? mid : lo, keep_hi = upper ? hi : mid;
            const size_t send_lo = upper ? lo : mid, send_hi = upper ? mid : hi;
            xfer(peer, ws + send_lo * len, (send_hi - send_lo) * len, rbuf, (keep_hi - keep_lo) * len, st);
            exchange_bytes += (send_hi - send_lo) * len * sizeof(float);
            ShardCombineArgs c{};
            float *mine = ws + keep_lo * len;
            c.lo = upper ? rbuf : mine;
            c.hi = upper ? mine : rbuf;
            c.n_rows = (uint32_t)(keep_hi - keep_lo);
            c.len = len;
            if (j + 1 < k) {
                c.dst_ws = mine;
            } else {
                // the finished voices go where the unsharded plan puts them: ring frame x0 + off + t, or -- frames of the call
                // only, the look-back part of a window is not output -- column x0 + off + t - idx of the output row
                const uint64_t skip = idx - x0;                      // window frames before the call's first frame
                c.dst = plan.d_split_dst.as<uint32_t>() + keep_lo;
                c.out = d_dst + (off > skip ? off - skip : 0);
                c.out_stride = n_times;
                c.out_skip = skip > off ? skip - off : 0;
                c.rings = d_rings.as<float>();
                c.ring_mask = ring_cap ? ring_cap - 1 : 0;
                c.ring_t0 = x0 + off;
            }
            Scope sc(this, &t_stage, st);
            HIP_CHECK(launch_shard_combine(c, st));
            sc.done();
            lo = keep_lo;
            hi = keep_hi;
        }
        for (size_t i = lo; i < hi; ++i)
            if (sv[i].owner != shard.rank) throw Error(FR_ERR_COMM, "internal: exchange order does not match voice ownership");
    }

    // FR_SHARD_GATHER: every rank's rows to rank 0's buffer.
    void gather_rows(float *d_dst, uint32_t n_slots, uint64_t n_times, hipStream_t st) {
        if (!sharded() || !(shard_flags & FR_SHARD_GATHER)) return;
        if (shard.rank == 0) {
            for (uint32_t p = 1; p < shard.world; ++p) {
                uint32_t lo, hi;
                shard_row_range(p, shard.world, n_slots, lo, hi);
                xfer(p, nullptr, 0, d_dst + (size_t)lo * n_times, (size_t)(hi - lo) * n_times, st);
            }
        } else {
            uint32_t lo, hi;
            my_rows(n_slots, lo, hi);
            xfer(0, d_dst + (size_t)lo * n_times, (size_t)(hi - lo) * n_times, nullptr, 0, st);
        }
    }

    bool timing = false;
    // A/B switches (environment, read at create; defaults are the measured best):
    uint32_t bank_leaf_variant = 1;      // FR_BANK_LEAF=0: product-form leaves (kernels.hpp BankArgs::leaf_variant)
    // Block streaming (fr_stream_*): one resident launch renders 64-frame blocks on a doorbell (kernels.hpp BankStreamCtl)
    bool streaming = false;
    double stream_trace_us[2] = {0, 0};
    uint64_t stream_trace_n = 0;
    uint32_t stream_seq = 0, stream_slots = 0;
    uint64_t stream_head = 0;            // first frame after the last streamed block
    float stream_last = 0.0f;            // that block's last (padded) input value: what a short row continuing it is padded with
    bool stream_have_last = false;
    uint32_t stream_idle_ms = BANK_STREAM_IDLE_MS;   // FR_STREAM_IDLE_MS (tests shorten it)
    int device_cus = 0;
    // The in-launch combine's arrival counters (d_tickets) and the row-completion counters (d_row_done) are "all zero
    // between launches": every launch that uses them leaves them so.  A launch that ended abnormally may not have: whatever
    // can leave them dirty sets this, and the next user clears them before its launch.
    bool counters_dirty = false;
    void clean_counters(hipStream_t st) {
        if (!counters_dirty) return;
        if (d_tickets.p) HIP_CHECK(hipMemsetAsync(d_tickets.p, 0, d_tickets.bytes, st));
        if (d_row_done.p) HIP_CHECK(hipMemsetAsync(d_row_done.p, 0, d_row_done.bytes, st));
        counters_dirty = false;
    }
    PinnedBuf h_stream_ctl, h_stream_out;
    DevBuf d_stream_dev;
    // `clean`: the launch was answering when the stop was rung (every block it took was finished).  Otherwise some chunks of
    // a voice may have taken their ticket and others never run: the counters are cleared before anyone uses them again.
    void end_stream(bool clean = true) {
        if (!streaming) return;
        BankStreamCtl *ctl = (BankStreamCtl *)h_stream_ctl.p;
        for (int i = 0; i < 64; ++i) __atomic_store_n(&ctl->row[i], (unsigned long long)BANK_STREAM_STOP << 32, __ATOMIC_RELEASE);
        if (hipStreamSynchronize(stream) != hipSuccess) { (void)hipGetLastError(); clean = false; }   // the kernel sees the stop within a poll, or ends itself after its bound
        streaming = false;
        stream_have_last = false;
        if (!clean) counters_dirty = true;
        head = UINT64_MAX;                       // the streamed frames were not stored: whatever comes next is a seek
    }
    bool allow_jit = true;               // FR_JIT=0: no hipRTC specialisation (those voices run as programs / pull)
    bool allow_template = true;          // FR_BANK_TEMPLATE=0: template voices go through the JIT path literally
    bool allow_multi = true;             // FR_BANK_MULTI=0: never the whole-voices-per-wave kernel for small voices
    bool fused_strided_ok = true;        // FR_STAGE_STRIDED=0: a long steady call of the fused form as one launch per sub-window (A/B)
    int stage_jit_mode = 1;              // FR_STAGE_JIT=0: programs always interpreted; 1: compiled when >= 4 programs share
                                         // a skeleton on average; 2 ("force"): compiled whenever they fit one kernel
    JitCache jit_cache;
    bool jit_async_configured = true;    // fr_config: hipRTC on the worker thread unless FR_CONFIG_SYNC_COMPILE
    Lowering lowering;                   // lowered graph, kept up to date across edits
    std::unique_ptr<BankMatcher> matcher;   // voice recognition memo over lowering's graph (same generation)
    uint64_t matcher_gen = 0;
    uint64_t plans_built = 0, plans_incremental = 0;
    TimerClass t_bank, t_pull, t_stage;
    DevBuf d_rings, d_in_table_stage;
    uint64_t ring_cap = 0;               // floats per ring (power of two)
    std::vector<hipEvent_t> event_pool;
    std::string last_error;
    std::string jit_error;
    std::string plan_json_cache;

    ~fr_renderer() {
        (void)hipSetDevice(device);
        end_stream();
        for (TimerClass *tc : {&t_bank, &t_pull, &t_stage})
            for (auto &pr : tc->pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
        for (Retired &g : graveyard) (void)hipEventDestroy(g.ev);
        if (ev_last) (void)hipEventDestroy(ev_last);
        for (hipEvent_t e : x_events) (void)hipEventDestroy(e);
        if (xstream) (void)hipStreamDestroy(xstream);
        // ranges the host registered and never unregistered: the page-lock and the device mapping must not outlive the
        // renderer that made them (the host may free that memory next; a later registration of the same addresses by
        // another renderer would otherwise meet a stale one)
        if (!registered.empty()) (void)hipStreamSynchronize(stream);
        for (auto &rg : registered) { if (hipHostUnregister(rg.first) != hipSuccess) (void)hipGetLastError(); }
        if (host_trace && stream_trace_n)
            std::fprintf(stderr, "fr_stream_block over %llu blocks: ring + wait %.1f us, copy out %.1f us\n", (unsigned long long)stream_trace_n,
                         stream_trace_us[0] / stream_trace_n, stream_trace_us[1] / stream_trace_n);
        if (host_trace && trace_n)
            std::fprintf(stderr, "fr_fill_buffer phases over %llu calls (mapped out %d, mapped in %d): issue %.1f us, %s %.1f us, %s %.1f us\n",
                         (unsigned long long)trace_n, (int)host_out_mapped, (int)host_rows_mapped, trace_us[0] / trace_n,
                         host_out_mapped ? "wait" : "D2H issue", trace_us[1] / trace_n, host_out_mapped ? "CPU copy" : "wait", trace_us[2] / trace_n);
        if (stream) (void)hipStreamDestroy(stream);
    }

    uint64_t implicit_len(uint64_t slot) const {
        for (const Seg &s : segs) if (slot >= s.first && slot < s.last) return s.len;
        return 0;
    }

    // ---- timing ------------------------------------------------------------------------------
    hipEvent_t get_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    struct Scope {
        fr_renderer *r; TimerClass *tc; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
        Scope(fr_renderer *rr, TimerClass *t, hipStream_t st) : r(rr), tc(t), s(st) {
            if (r->timing) { a = r->get_event(); b = r->get_event(); HIP_CHECK(hipEventRecord(a, s)); }
        }
        void done() {
            if (a) { HIP_CHECK(hipEventRecord(b, s)); tc->pending.emplace_back(a, b); a = nullptr; }
        }
    };
    void resolve(TimerClass &tc) {
        for (auto &pr : tc.pending) {
            HIP_CHECK(hipEventSynchronize(pr.second));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
            tc.ms += ms;
            tc.launches += 1;
            event_pool.push_back(pr.first);
            event_pool.push_back(pr.second);
        }
        tc.pending.clear();
    }

    // ---- input store (reference.rs:47-75) -------------------------------------------------------
    // Frames of input history kept per slot (0 = everything since the last seek: the reference, reference.rs:25).  With
    // fr_config.history_frames set, at least what the current plan's constant delays and proven bounds can reach.
    uint64_t keep_frames() const {
        if (!history_frames) return 0;
        return std::max<uint64_t>(history_frames, plan.valid ? plan.sp.input_lookback : 0);
    }
    // Buffers replaced by bigger ones are freed once the stream has passed the copy out of them -- never by waiting.
    uint64_t call_idx = 0;               // first frame of the call being served
    uint64_t history_floor = 0;          // bounded history: input samples before this frame read as 0.0 (monotone)
    struct Retired { DevBuf buf; hipEvent_t ev; };
    std::vector<Retired> graveyard;
    void bury(DevBuf &&b, hipStream_t st) {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, st) != hipSuccess) {
            (void)hipGetLastError();
            HIP_CHECK(hipStreamSynchronize(st));   // (cannot track it: wait once)
            if (ev) (void)hipEventDestroy(ev);
            return;                                // `b` is freed here
        }
        graveyard.push_back(Retired{std::move(b), ev});
    }
    void reap() {
        for (size_t i = 0; i < graveyard.size();) {
            if (hipEventQuery(graveyard[i].ev) == hipSuccess) {
                (void)hipEventDestroy(graveyard[i].ev);
                graveyard[i] = std::move(graveyard.back());
                graveyard.pop_back();
            } else {
                (void)hipGetLastError();
                ++i;
            }
        }
    }
    // Room for n_times more frames in slot s.  Bounded history slides in place (the newest `keep` frames move to the
    // front of a buffer sized 2 * (keep + call) -- one small device copy every `keep` frames, no allocation); unbounded
    // history doubles into a new buffer with an asynchronous copy, the old buffer retired without a stream wait.
    void make_room(InSlot &s, uint64_t n_times, hipStream_t st) {
        uint64_t stored = s.len - s.base;
        if (stored + n_times <= s.cap) return;
        if (last_pending) HIP_CHECK(hipDeviceSynchronize());   // a call on another stream may still be appending to this buffer
        const uint64_t keep = keep_frames();
        if (keep && stored > keep) {
            const uint64_t drop = stored - keep;
            if (drop >= keep && keep + n_times <= s.cap) {         // source and destination do not overlap
                HIP_CHECK(hipMemcpyAsync(s.buf.p, s.buf.as<float>() + drop, keep * sizeof(float), hipMemcpyDeviceToDevice, st));
                s.base += drop;
                return;
            }
        }
        const uint64_t kept = keep ? std::min(stored, keep) : stored;
        const uint64_t cap = keep ? std::max<uint64_t>(2 * (keep + n_times), 1u << 16)
                                  : std::max<uint64_t>(stored + n_times, std::max<uint64_t>(s.cap * 2, 1u << 20));
        DevBuf nb;
        nb.ensure(cap * sizeof(float));
        if (kept) HIP_CHECK(hipMemcpyAsync(nb.p, s.buf.as<float>() + (stored - kept), kept * sizeof(float), hipMemcpyDeviceToDevice, st));
        s.base += stored - kept;
        if (s.buf.p) bury(std::move(s.buf), st);
        s.buf = std::move(nb);
        s.cap = cap;
    }

    // A full-length device-resident row feeding a bank's time slot is not copied here: the bank kernel reads
    // the caller's row directly and appends it to the history itself (BankArgs::hist_dst).
    struct Deferred { uint32_t slot; const float *src; float *dst; };
    std::vector<Deferred> deferred;
    // Will execute() render exactly [idx, idx + n_times) for the staged part (steady state), or rebuild a look-back
    // window first?  Same conditions as execute() uses (ring capacity, contiguity with what the rings hold).
    bool steady_call(uint64_t idx, uint64_t n_times) const {
        const StagedPlan &sp = plan.sp;
        if (!sp.uses_rings()) return true;
        uint64_t need = sp.lmax + n_times, cap = 1024;
        while (cap < need) cap <<= 1;
        if (cap > ring_cap || (size_t)sp.n_rings * ring_cap * sizeof(float) > d_rings.bytes) return false;
        return plan.stage_valid && plan.stage_end == idx;
    }
    bool bank_time_slot(uint32_t n_slots, uint64_t n_times, uint32_t slot, uint64_t idx) const {
        if (!plan_current(n_slots)) return false;
        // (bank_small_kernel, used for the shortest calls of shapes the short-call kernel does not take, does not append)
        for (const BankStage &bs : plan.banks) {
            if (bs.grp.jit || bs.grp.general || bs.grp.input_slot != slot) continue;
            uint32_t c, f, w, small, vpw;
            bank_shape(bs.grp.log2_p, (uint32_t)bs.grp.rows.size(), n_times, c, f, w, small, vpw);
            if (small == 1) return false;
        }
        // a window with look-back reads the stored history, so the row must be there first; in steady state the bank
        // launch (always ahead of the programs on the stream) reads the caller's row and appends it like any other
        if ((plan.sp.uses_rings() || !plan.sp.progs.empty()) && !steady_call(idx, n_times)) return false;
        bool any = false;
        for (const BankStage &bs : plan.banks) {
            if (bs.grp.jit) {
                for (uint32_t sl : bs.grp.shape.input_slots) if (sl == slot) return false;
                continue;
            }
            if (bs.grp.input_slot != slot) continue;
            if (bs.grp.general || bs.grp.jit) return false;   // only the balanced hand-written kernel appends history
            any = true;
        }
        return any;
    }

    // What a failed call must put back: the input store is committed before the kernels run, and execute() can still
    // fail (device errors, out of memory).  Without a seek the slots' lengths, the implicit segments and the vec count
    // go back to what they were; after a seek the old samples may already be overwritten, so the store stays in the
    // state the seek itself produces (everything zero before idx) -- the retry at the same idx seeks again anyway.
    struct StoreSnapshot {
        struct S { bool fed; uint64_t base, len; };
        std::vector<S> slots;
        std::vector<Seg> segs;
        uint64_t n_vecs = 0;
        bool seeked = false;
        uint64_t idx = 0;
    };
    StoreSnapshot snapshot_store(uint64_t idx) const {
        StoreSnapshot sn;
        sn.slots.reserve(slots.size());
        for (const InSlot &s : slots) sn.slots.push_back({s.fed, s.base, s.len});
        sn.segs = segs;
        sn.n_vecs = n_vecs;
        sn.seeked = idx != head;
        sn.idx = idx;
        return sn;
    }
    void rollback_store(const StoreSnapshot &sn) {
        deferred.clear();
        plan.stage_valid = false;   // rings may hold part of the failed call's window
        counters_dirty = true;      // a launch of the failed call may have stopped half-way through its tickets / row counts
        if (sn.seeked) {
            for (InSlot &s : slots) { s.base = sn.idx; s.len = sn.idx; }
            return;
        }
        for (size_t i = 0; i < slots.size(); ++i) {
            if (i < sn.slots.size()) {
                // (a bounded history may have slid forward meanwhile: the buffer then starts at the later base)
                slots[i].fed = sn.slots[i].fed;
                slots[i].base = std::min(std::max(slots[i].base, sn.slots[i].base), sn.slots[i].len);
                slots[i].len = sn.slots[i].len;
            }
            else { slots[i].fed = false; slots[i].base = slots[i].len = 0; }
        }
        segs = sn.segs;
        n_vecs = sn.n_vecs;
    }

    // `device_rows`: in_data is a device pointer (fr_fill_buffer_device).
    // `dense`: the rows are an [n_rows][n_times] matrix (fr_fill_buffer_dense); `offs` then covers the stored rows only.
    void store_inputs(uint32_t n_slots, uint64_t n_times, uint64_t idx, const float *in_data,
                      const uint64_t *offs, uint32_t n_rows, bool device_rows, hipStream_t st, bool dense = false) {
        deferred.clear();
        reap();
        call_idx = idx;
        call_tracks = nullptr;
        call_track_stride = call_track_rows = 0;
        if (dense_total_rows) { dense = true; n_rows = dense_total_rows; }   // (fr_fill_buffer_dense: `offs` covers min(rows, track_from) rows)
        {   // rows beyond the reference's input vectors -- n_slots * n_times of the largest call so far -- are dropped (reference.rs:59-68)
            const uint64_t vecs_after = std::max<uint64_t>(n_vecs, (uint64_t)n_slots * n_times);
            if (track_from != 0xFFFFFFFFu && n_rows > vecs_after) n_rows = (uint32_t)vecs_after;
        }
        if (n_rows > track_from && n_times > 0) {   // rows of track slots: not stored, read in place by this call's voices
            if (!dense)
                for (uint32_t r = track_from; r < n_rows; ++r)
                    if (offs[r + 1] - offs[r] != n_times)
                        throw Error(FR_ERR_UNSUPPORTED, "track input row " + std::to_string(r) + " must hold exactly the " + std::to_string(n_times) + " frames rendered");
            const float *first = in_data + (dense ? (uint64_t)track_from * n_times : offs[track_from]);
            const uint64_t rows_t = n_rows - track_from;
            if (!device_rows) {
                d_tracks_stage.ensure(rows_t * n_times * sizeof(float));
                HIP_CHECK(hipMemcpyAsync(d_tracks_stage.p, first, rows_t * n_times * sizeof(float), hipMemcpyHostToDevice, st));
                first = d_tracks_stage.as<float>();
            }
            call_tracks = reinterpret_cast<const float *>(reinterpret_cast<uintptr_t>(first) - (uintptr_t)track_from * n_times * sizeof(float));
            call_track_stride = n_times;
            call_track_rows = rows_t;
            n_rows = track_from;
        } else if (dense_total_rows) {
            n_rows = std::min(n_rows, track_from);
        }
        const bool seek = idx != head;   // forget history, act as if inputs were 0 before idx (renderer.rs:12-15)
        // validate everything before mutating so a refused call leaves the history intact: the lengths the rows must
        // continue are those AFTER the seek (idx for every vec) and after the vec count grew (reference.rs:52-71)
        {
            const uint64_t want = (uint64_t)n_slots * n_times;
            const uint64_t vecs_after = std::max(n_vecs, want);
            const uint32_t rows = (uint32_t)std::min<uint64_t>(n_rows, vecs_after);
            for (uint32_t r = 0; r < rows; ++r) {
                uint64_t cur = idx;   // after a seek, and for vecs created by this call
                if (!seek && r < n_vecs) cur = (r < slots.size() && slots[r].fed) ? slots[r].len : implicit_len(r);
                if (cur != idx)
                    throw Error(FR_ERR_INPUT_HISTORY, "input slot " + std::to_string(r) + " holds " + std::to_string(cur) +
                                                          " samples, expected idx=" + std::to_string(idx));
                if (offs[r + 1] < offs[r] || offs[r + 1] - offs[r] > n_times)
                    throw Error(FR_ERR_INPUT_TOO_LONG, "input row " + std::to_string(r) + " longer than the range rendered");
            }
        }
        if (seek) {
            for (InSlot &s : slots) { s.base = idx; s.len = idx; }
            segs.clear();
            if (n_vecs) segs.push_back({0, n_vecs, idx});
            history_floor = 0;   // (everything before idx is zero now anyway)
        }
        {   // bounded history: what this call may see, and no later call may see more (deterministic whatever slack the
            // buffers happen to hold when a longer Delay arrives)
            const uint64_t keep = keep_frames();
            if (keep && idx > keep) history_floor = std::max(history_floor, idx - keep);
        }
        uint64_t want = (uint64_t)n_slots * n_times;   // `buff.len()`, reference.rs:60 (element count, a quirk)
        if (n_vecs < want) { segs.push_back({n_vecs, want, idx}); n_vecs = want; }
        uint32_t rows = (uint32_t)std::min<uint64_t>(n_rows, n_vecs);   // zip stops at the shorter (:68)
        if (rows > slots.size()) slots.resize(rows);
        // (every host-buffer call ends with a stream synchronisation, so the staging buffer is idle here)
        if (!device_rows && rows) h_in_stage.ensure((size_t)rows * n_times * sizeof(float));
        for (uint32_t r = 0; r < rows; ++r) {
            InSlot &s = slots[r];
            if (!s.fed) { s.fed = true; s.base = implicit_len(r); s.len = s.base; }
            uint64_t rl = offs[r + 1] - offs[r];
            uint64_t stored = s.len - s.base;
            make_room(s, n_times, st);
            stored = s.len - s.base;
            float *dst = s.buf.as<float>() + stored;
            if (device_rows && rl == n_times && rl > 0 && bank_time_slot(n_slots, n_times, r, idx)) {
                deferred.push_back(Deferred{r, in_data + offs[r], dst});
            } else if (device_rows) {
                if (rl) HIP_CHECK(hipMemcpyAsync(dst, in_data + offs[r], rl * sizeof(float), hipMemcpyDeviceToDevice, st));
                if (rl < n_times) {
                    // pad with the last value now stored, or 0 (reference.rs:72-73)
                    const float *last = (stored + rl) ? dst + rl - 1 : nullptr;
                    HIP_CHECK(launch_pad(dst + rl, n_times - rl, last, st));
                }
            } else {
                // host rows: row + padding assembled in this row's region of the pinned staging buffer, one H2D copy,
                // no sync (the region is not touched again before the call's final synchronisation)
                float pad = 0.0f;
                if (rl) pad = in_data[offs[r] + rl - 1];
                else if (stored) {
                    HIP_CHECK(hipMemcpyAsync(&pad, dst - 1, sizeof(float), hipMemcpyDeviceToHost, st));
                    HIP_CHECK(hipStreamSynchronize(st));
                }
                float *stage = h_in_stage.as<float>() + (size_t)r * n_times;
                if (rl) std::memcpy(stage, in_data + offs[r], rl * sizeof(float));
                std::fill(stage + rl, stage + n_times, pad);
                if (host_rows_mapped && n_times > 0 && bank_time_slot(n_slots, n_times, r, idx)) {
                    // no copy at all: the bank kernel reads the row through the mapping and appends it to the history in
                    // HBM itself (as it does for device-resident rows)
                    deferred.push_back(Deferred{r, h_in_stage.as_dev<float>() + (size_t)r * n_times, dst});
                } else {
                    HIP_CHECK(hipMemcpyAsync(dst, stage, n_times * sizeof(float), hipMemcpyHostToDevice, st));
                }
            }
            s.len += n_times;
        }
    }

    DevInput dev_input(uint32_t slot) const {
        DevInput d{nullptr, 0, 0};
        if (slot < slots.size() && slots[slot].fed && slot < n_vecs) {
            const InSlot &s = slots[slot];
            d.data = s.buf.as<float>();
            d.base = s.base;
            d.len = s.len;
            // bounded history: exactly `keep` frames before the call's first frame are visible, whatever slack the
            // buffer still holds (so that results do not depend on when the buffer last slid)
            if (history_floor > d.base) {
                const uint64_t floor = std::min(history_floor, d.len);
                d.data += floor - d.base;
                d.base = floor;
            }
        }
        return d;
    }

    // ---- planning ---------------------------------------------------------------------------------
    void build_plan(uint32_t n_slots, hipStream_t st) {
        Plan p;
        p.version = mirror.version;
        p.shard_epoch = shard_epoch;
        p.n_slots = n_slots;
        const ShardSpec *shard_spec = sharded() ? &shard : nullptr;
        const auto t_begin = std::chrono::steady_clock::now();
        // a rank of a voice-sharded job lowers only the rows it owns (1/world of the first-call cost; graph errors on another
        // rank's rows are that rank's to report); partial-block sharding analyses the whole graph on every rank
        uint32_t lower_lo = 0, lower_hi = UINT32_MAX;
        if (sharded() && shard.mode == FR_SHARD_VOICES) my_rows(n_slots, lower_lo, lower_hi);
        // (partial-block sharding: every rank must arrive at the same node ids -- the exchange is ordered by them)
        const FlatGraph &fg = lowering.update(mirror, n_slots, lower_lo, lower_hi, sharded() && shard.mode == FR_SHARD_PARTIALS);
        const double lower_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        p.max_depth = fg.max_depth;
        bool use_jit = allow_jit && mode == FR_MODE_AUTO;
        if (!matcher || matcher_gen != lowering.generation()) {
            matcher.reset(new BankMatcher(fg, 20, use_jit, allow_template, track_from));
            matcher_gen = lowering.generation();
        }
        p.sp = plan_stages(fg, mode == FR_MODE_AUTO, mode != FR_MODE_PULL, 20, use_jit, allow_template, matcher.get(), shard_spec, track_from);
        std::vector<std::shared_ptr<JitKernel>> jits(p.sp.banks.size());
        p.jit_epoch = jit_cache.epoch();   // (read first: a compile finishing from here on makes this plan stale)
        if (use_jit) {
            bool without = false;
            try {
                for (size_t i = 0; i < p.sp.banks.size(); ++i)
                    if (p.sp.banks[i].jit) {
                        // (voices that read tracks have no other evaluator to render with meanwhile: wait for the compiler)
                        const bool wait = p.sp.banks[i].tracks && jit_async_configured;
                        if (wait) jit_cache.set_async(false);
                        struct Restore { JitCache &c; bool on; ~Restore() { if (on) c.set_async(true); } } restore{jit_cache, wait};
                        jits[i] = jit_cache.get(p.sp.banks[i].shape, p.sp.banks[i].varying, p.sp.banks[i].literal_bits, p.sp.banks[i].alias);
                        if (!jits[i]) p.jit_pending = without = true;   // being compiled on the worker thread: do not wait
                    }
            } catch (const Error &e) {   // hipRTC unavailable or the generated source did not compile: plan without it
                jit_error = e.what();
                without = true;
                // ... unless the plan must be the SAME on every rank (partial-block sharding: the list of split voices is the
                // exchange's schedule): a rank that re-planned on its own would send ranges its peers do not expect, and the
                // job would hang in the transport.  There the failure is the call's.  (FR_ERR_UNSUPPORTED = no run-time compiler in
                // this build at all -- the same on every rank of it, so planning without is still planning alike.)
                if (sharded() && shard.mode == FR_SHARD_PARTIALS && e.code != FR_ERR_UNSUPPORTED)
                    throw Error(FR_ERR_DEVICE, std::string("a kernel of the sharded plan could not be compiled on this rank (every rank must plan alike): ") + e.what());
            }
            if (without) {
                p.sp = plan_stages(fg, true, true, 20, false, true, nullptr, shard_spec, track_from);
                jits.assign(p.sp.banks.size(), nullptr);
            }
        }
        size_t bank_i = 0;
        for (BankLaunch &bg : p.sp.banks) {
            BankStage bs;
            bs.jit = jits[bank_i++];
            bs.grp = std::move(bg);
            bs.d_params.ensure(bs.grp.params.size() * sizeof(float));
            bs.d_rows.ensure(bs.grp.rows.size() * sizeof(uint32_t));
            HIP_CHECK(hipMemcpyAsync(bs.d_params.p, bs.grp.params.data(), bs.grp.params.size() * sizeof(float), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(bs.d_rows.p, bs.grp.rows.data(), bs.grp.rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            if (bs.grp.general) {
                bs.d_groups.ensure(bs.grp.groups.size() * sizeof(uint32_t));
                bs.d_group_off.ensure(bs.grp.group_off.size() * sizeof(uint32_t));
                HIP_CHECK(hipMemcpyAsync(bs.d_groups.p, bs.grp.groups.data(), bs.grp.groups.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                HIP_CHECK(hipMemcpyAsync(bs.d_group_off.p, bs.grp.group_off.data(), bs.grp.group_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            }
            p.banks.push_back(std::move(bs));
        }
        p.sp.banks.clear();
        if (!p.sp.split.empty()) {
            std::vector<uint32_t> dst(p.sp.split.size());
            for (size_t i = 0; i < dst.size(); ++i) dst[i] = p.sp.split[i].dst | (p.sp.split[i].to_ring ? 0x80000000u : 0u);
            p.d_split_dst.ensure(dst.size() * sizeof(uint32_t));
            HIP_CHECK(hipMemcpyAsync(p.d_split_dst.p, dst.data(), dst.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st));   // `dst` goes out of scope
        }
        if (!p.sp.progs.empty()) {
            p.d_instrs.ensure(p.sp.instrs.size() * sizeof(StageInstr));
            p.d_progs.ensure(p.sp.progs.size() * sizeof(StageProg));
            HIP_CHECK(hipMemcpyAsync(p.d_instrs.p, p.sp.instrs.data(), p.sp.instrs.size() * sizeof(StageInstr), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(p.d_progs.p, p.sp.progs.data(), p.sp.progs.size() * sizeof(StageProg), hipMemcpyHostToDevice, st));
            StageJitPlan sj;
            if (allow_jit && stage_jit_mode != 0 && plan_stage_jit(p.sp.progs, p.sp.instrs, 32, stage_jit_mode == 2, sj, mirror.sparkle,
                                                                              stage_block_env ? stage_block_env : (p.sp.feedback ? 16u : 1u),
                                                                              p.sp.feedback && p.sp.fused_carry_only)) {
                try {
                    p.stage_jit = jit_cache.get_source(sj.source, "jit_stage");
                    if (!p.stage_jit) {   // still compiling: the interpreter serves the calls until the plan is rebuilt
                        p.jit_pending = true;
                        throw Error(FR_OK, "");
                    }
                    p.stage_shapes = sj.n_shapes;
                    p.d_jprogs.ensure(sj.progs.size() * sizeof(JitStageProg));
                    p.d_ptab.ensure(std::max<size_t>(sj.ptab.size(), 1) * sizeof(uint32_t));
                    HIP_CHECK(hipMemcpyAsync(p.d_jprogs.p, sj.progs.data(), sj.progs.size() * sizeof(JitStageProg), hipMemcpyHostToDevice, st));
                    if (!sj.ptab.empty())
                        HIP_CHECK(hipMemcpyAsync(p.d_ptab.p, sj.ptab.data(), sj.ptab.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    HIP_CHECK(hipStreamSynchronize(st));   // `sj` goes out of scope
                } catch (const Error &e) {   // keep the interpreter
                    if (e.code != FR_OK) jit_error = e.what();
                    p.stage_jit = nullptr;
                }
            }
        }
        p.pull_rows = p.sp.pull_rows;
        if (!p.pull_rows.empty()) {
            // dense input table: OP_INPUT.a becomes an index into input_slots
            std::vector<DevNode> dn(fg.nodes.size());
            std::unordered_map<uint32_t, uint32_t> dense;
            for (size_t i = 0; i < dn.size(); ++i) {
                const FlatNode &n = fg.nodes[i];
                dn[i] = DevNode{n.op, n.a, n.b, n.depth};
                if (n.op == OP_INPUT) {
                    auto it = dense.emplace(n.a, (uint32_t)p.input_slots.size());
                    if (it.second) p.input_slots.push_back(n.a);
                    dn[i].a = it.first->second;
                }
            }
            std::vector<uint32_t> roots(n_slots);
            for (uint32_t r = 0; r < n_slots; ++r) roots[r] = fg.outputs[r];
            p.d_nodes.ensure(dn.size() * sizeof(DevNode));
            p.d_roots.ensure(roots.size() * sizeof(uint32_t));
            HIP_CHECK(hipMemcpyAsync(p.d_nodes.p, dn.data(), dn.size() * sizeof(DevNode), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(p.d_roots.p, roots.data(), roots.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        }
        HIP_CHECK(hipStreamSynchronize(st));   // host vectors above go out of scope
        // description
        std::ostringstream js;
        js << "{\"backend\":\"hip-gfx950\",\"mode\":" << mode << ",\"n_slots\":" << n_slots
           << ",\"lowered_nodes\":" << fg.nodes.size() << ",\"max_depth\":" << fg.max_depth
           << ",\"lowering\":\"" << (lowering.last_was_full() ? "full" : "incremental") << "\",\"relowered_nodes\":" << lowering.last_relowered()
           << ",\"lower_ms\":" << lower_ms << ",\"plans_built\":" << plans_built + 1 << ",\"banks\":[";
        for (size_t i = 0; i < p.banks.size(); ++i) {
            const BankLaunch &g = p.banks[i].grp;
            js << (i ? "," : "") << "{\"voices\":" << g.rows.size() << ",\"partials\":" << (g.general ? g.max_leaves : (1u << g.log2_p))
               << ",\"general_tree\":" << (g.general ? "true" : "false") << ",\"jit\":" << (g.jit ? "true" : "false")
               << ",\"leaf_ops\":" << (g.jit ? g.shape.ops.size() : 0) << ",\"leaf_params\":" << (g.jit ? g.k : 2)
               << ",\"input_slot\":" << g.input_slot << ",\"fast_ok\":" << (g.fast_ok ? "true" : "false")
               << ",\"to_ring\":" << (g.to_ring ? "true" : "false") << ",\"to_exchange\":" << (g.to_ws ? "true" : "false")
               << ",\"tracks\":" << (g.tracks ? "true" : "false")
               << ",\"param_bytes\":" << g.params.size() * sizeof(float) << "}";
        }
        js << "],\"stage_programs\":" << (p.sp.progs.size() - p.sp.fused_count - p.sp.post_count) << ",\"stage_instrs\":" << p.sp.instrs.size()
           << ",\"fused_programs\":" << p.sp.fused_count << ",\"fused_max_frames\":" << p.sp.fused_max_frames << ",\"fused_stride\":" << p.sp.fused_stride
           << ",\"feedback\":" << (p.sp.feedback ? "true" : "false") << ",\"feedback_loops\":" << p.sp.feedback_loops
           << ",\"fused_levels\":" << (p.sp.fused_level_first.empty() ? 0 : p.sp.fused_level_first.size() - 1) << ",\"copy_programs\":" << p.sp.post_count
           << ",\"stage_levels\":" << (p.sp.level_first.empty() ? 0 : p.sp.level_first.size() - 1)
           << ",\"stage_jit\":" << (p.stage_jit ? "true" : "false") << ",\"stage_shapes\":" << p.stage_shapes
           << ",\"rings\":" << p.sp.n_rings << ",\"max_lookback\":" << p.sp.lmax
           << ",\"input_lookback\":" << p.sp.input_lookback << ",\"input_lookback_unbounded\":" << (p.sp.input_lookback_unbounded ? "true" : "false")
           << ",\"history_frames\":" << history_frames
           << ",\"jit_pending\":" << (p.jit_pending ? "true" : "false") << ",\"jit_kernels_compiled\":" << jit_cache.compiled() << ",\"jit_compile_ms\":" << jit_cache.compile_ms() << ",\"jit_disk_hits\":" << jit_cache.disk_hits()
           << ",\"pull_rows\":" << p.pull_rows.size()
           << ",\"shard\":{\"rank\":" << shard.rank << ",\"world\":" << shard.world << ",\"mode\":" << (sharded() ? shard.mode : 0)
           << ",\"split_voices\":" << p.sp.split.size() << ",\"transport\":\"" << (rccl ? "rccl" : has_host_comm ? "host-callback" : "none") << "\"}"
           << ",\"exchange\":{\"max_tiles\":" << ((shard_flags & FR_SHARD_SERIAL_EXCHANGE) ? 1u : x_max_tiles) << ",\"min_tile_frames\":" << x_min_tile
           << "}"
           << ",\"build_ms\":" << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count() << "}";
        ++plans_built;
        p.json = js.str();
        p.valid = true;
        plan = std::move(p);
    }

    void ensure_plan(uint32_t n_slots, hipStream_t st) {
        if (!plan_current(n_slots)) build_plan(n_slots, st);
    }

    // ---- execution --------------------------------------------------------------------------------
    void execute(float *d_dst, uint32_t n_slots, uint64_t n_times, uint64_t idx, hipStream_t st) {
        ensure_plan(n_slots, st);
        if (n_slots == 0 || n_times == 0) return;
        const StagedPlan &sp = plan.sp;

        // Window of the staged part.  Contiguous with what the rings already hold: just this call's frames.
        // Otherwise (first call, seek, graph edit, larger call): rebuild the look-back from the input history.
        uint64_t w0 = idx;
        // Feedback plans (stage.hpp StagedPlan::feedback): no window bounds a loop's look-back, so rings that are not current
        // are brought up to date by replaying every frame from 0 in chunks -- the ring-bound banks and the fused programs over
        // [c0, c0 + len), nothing written to the output -- before the call's own frames run in steady-state form.
        constexpr uint64_t FB_CHUNK = 16384, FB_MAX_REPLAY = 1ull << 28;
        if (sp.feedback && history_frames != 0)
            throw Error(FR_ERR_UNSUPPORTED, "feedback through Delay needs the full input history (fr_config.history_frames = 0)");
        if (sp.uses_rings()) {
            // (a feedback plan's rings always have room for a replay chunk: growing them later would lose the loop's state)
            uint64_t need = sp.lmax + std::max<uint64_t>(n_times, sp.feedback ? FB_CHUNK : 0);
            uint64_t cap = 1024;
            while (cap < need) cap <<= 1;
            if (cap > ring_cap) {
                d_rings.ensure((size_t)sp.n_rings * cap * sizeof(float));
                ring_cap = cap;
                plan.stage_valid = false;
            } else if ((size_t)sp.n_rings * ring_cap * sizeof(float) > d_rings.bytes) {
                d_rings.ensure((size_t)sp.n_rings * ring_cap * sizeof(float));
                plan.stage_valid = false;
            }
            if (!(plan.stage_valid && plan.stage_end == idx)) w0 = idx > sp.lmax ? idx - sp.lmax : 0;
            if (sp.feedback) w0 = idx;   // (the replay below has brought the rings to idx by the time this window runs)
        }
        // (decided AFTER the rings may have been re-allocated above: a longer call than any before loses what they held)
        const bool fb_replay = sp.feedback && !(plan.stage_valid && plan.stage_end == idx) && idx != 0;
        if (fb_replay && idx > FB_MAX_REPLAY)
            throw Error(FR_ERR_UNSUPPORTED, "a feedback loop's state at frame " + std::to_string(idx) + " would take replaying more than 2^28 frames");
        const uint64_t w_len = idx + n_times - w0;
        // Split voices (partial-block sharding): every rank renders its sub-trees over the SAME window -- the look-back
        // window when any split voice feeds a ring (lmax, ring capacity and validity are the same on every rank: same
        // graph, same calls), else just this call's frames.
        bool x_ring = false;
        for (const SplitVoice &v : sp.split) x_ring = x_ring || v.to_ring;
        const uint64_t x0 = x_ring ? w0 : idx, xlen = x_ring ? w_len : n_times;
        if (!sp.split.empty()) {
            used_scratch = true;
            d_ws.ensure(sp.split.size() * xlen * sizeof(float));
            d_xrecv.ensure(sp.split.size() * xlen * sizeof(float));
        }

        // Tiles of the exchange window (partial-block sharding): whole 64-frame kernel tiles, at most x_max_tiles of them, none
        // shorter than x_min_tile.  One tile = the serial form of round 2.
        std::vector<std::pair<uint64_t, uint64_t>> xt;   // (offset in the window, frames)
        if (!sp.split.empty()) {
            // (the host-callback transport pays a host round trip and a stream synchronisation per message -- 4 tiles over gloo
            //  measured 0.63 ms per call against 0.28 serial, profiles/r03_exchange_rehearsal.txt: it stays serial unless asked;
            //  RCCL sends are enqueued like kernels, there tiling hides them)
            const bool serial = (shard_flags & FR_SHARD_SERIAL_EXCHANGE) || (!rccl && !x_tiles_explicit);
            uint64_t nt = serial ? 1 : std::min<uint64_t>(x_max_tiles, xlen / std::max<uint32_t>(x_min_tile, 64u));
            nt = std::max<uint64_t>(nt, 1);
            const uint64_t tl = (((xlen + nt - 1) / nt) + 63) / 64 * 64;
            for (uint64_t off = 0; off < xlen; off += tl) xt.push_back({off, std::min(tl, xlen - off)});
        }
        const bool x_window_is_call = x0 == idx && xlen == n_times;
        std::unordered_set<uint32_t> tile_slots, tile_appended;   // input slots whose deferred row the tiles append / this tile has appended
        // One bank launch over the window [b0, b0 + blen).  `tile_off` >= 0: a tile of the exchange window (b0 = x0 + tile_off),
        // written to the tile-major workspace; such a launch appends ITS part of a deferred input row.
        auto launch_bank_window = [&](BankStage &bs, uint64_t b0, uint64_t blen, int64_t tile_off) {
            const bool ring = bs.grp.to_ring, ws = bs.grp.to_ws;
            BankArgs a{};
            a.params = bs.d_params.as<float2>();
            // time-slot history for window [b0, b0 + blen): zero before the stored history (seek), zero beyond it
            DevInput di = dev_input(bs.grp.input_slot);
            if (di.data && di.len > di.base) {
                uint64_t start = std::max(b0, di.base);
                a.time_skip = std::min(start - b0, blen);
                a.time = di.data + (start - di.base);
                a.time_valid = di.len > start ? di.len - start : 0;
            }
            if (tile_off < 0 && b0 == idx && blen == n_times)   // (direct output, or a ring in steady state)
                for (Deferred &d : deferred)
                    if (d.slot == bs.grp.input_slot) {   // read the caller's row; the first bank on this slot appends it
                        a.time = d.src;
                        a.time_skip = 0;
                        a.time_valid = n_times;
                        a.hist_dst = d.dst;
                        d.dst = nullptr;
                    }
            if (tile_off >= 0 && x_window_is_call)   // a tile of the call itself: its part of the caller's row, appended by the first bank on the slot
                for (Deferred &d : deferred)
                    if (d.slot == bs.grp.input_slot) {
                        a.time = d.src + tile_off;
                        a.time_skip = 0;
                        a.time_valid = blen;
                        a.hist_dst = (d.dst && !tile_appended.count(d.slot)) ? d.dst + tile_off : nullptr;
                        tile_slots.insert(d.slot);
                        tile_appended.insert(d.slot);
                    }
            a.rows = bs.d_rows.as<uint32_t>();
            if (ring) {
                a.out = d_rings.as<float>();
                a.out_stride = ring_cap;
                a.ring_mask = ring_cap - 1;
                a.ring_t0 = b0;
            } else if (ws) {
                a.out = d_ws.as<float>() + sp.split.size() * (uint64_t)(tile_off > 0 ? tile_off : 0);   // tile-major workspace
                a.out_stride = blen;
            } else {
                a.out = d_dst;
                a.out_stride = n_times;
            }
            a.n_voices = (uint32_t)bs.grp.rows.size();
            a.log2_p = bs.grp.log2_p;
            a.n_times = blen;
            a.fast_ok = bs.grp.fast_ok ? 1u : 0u;
            if (bs.grp.jit) {
                JitBankArgs j{};
                j.params = bs.d_params.as<float>();
                for (size_t i = 0; i < bs.grp.shape.input_slots.size(); ++i) {   // every input row over the same window
                    DevInput dj = dev_input(bs.grp.shape.input_slots[i]);
                    if (dj.data && dj.len > dj.base) {
                        uint64_t start = std::max(b0, dj.base);
                        j.in_skip[i] = std::min(start - b0, blen);
                        j.in[i] = dj.data + (start - dj.base);
                        j.in_valid[i] = dj.len > start ? dj.len - start : 0;
                    }
                }
                j.out = a.out;
                j.rows = a.rows;
                j.out_stride = a.out_stride;
                j.ring_mask = a.ring_mask;
                j.ring_t0 = a.ring_t0;
                j.n_times = blen;
                j.n_voices = a.n_voices;
                j.log2_p = a.log2_p;
                j.tiles = (uint32_t)((blen + 63) / 64);
                j.nblocks = j.tiles * j.n_voices;
                if (allow_multi && a.log2_p <= 8 && bs.jit->fn_multi) {   // many small voices: whole voices per wave
                    uint32_t vpw = std::max(2u, 256u >> a.log2_p);
                    auto nb = [&](uint32_t per_wave) { return ((a.n_voices + 4ull * per_wave - 1) / (4ull * per_wave)) * j.tiles; };
                    while (vpw > 1 && nb(vpw) < 2048) vpw >>= 1;
                    if (nb(vpw) >= 1024) { j.voices_per_wave = vpw; j.nblocks = (uint32_t)nb(vpw); }
                }
                j.fract_ok = bs.grp.fast_ok ? 1u : 0u;
                // Few voices, short call: one workgroup per (voice, 64-frame tile) leaves most of the chip idle (64 voices x 64
                // frames = 64 workgroups).  Render every voice as 2^c consecutive pieces of its leaves instead -- to the kernel
                // 2^c times as many voices of 2^-c the size, rows of a workspace -- and add the pieces up in the tree's order.
                uint32_t pieces_log2 = 0;
                if (jit_chunks && !ring && !ws && !j.voices_per_wave && a.n_voices <= 1024u) {
                    // (voices that stream tracks from HBM want many small workgroups -- 64 x 4096 x 1024 frames: 0.84 of the achievable
                    //  bandwidth with 1024 workgroups, 0.93 with 16 384; profiles/r03_tracks.txt -- the arithmetic-bound ones only a full chip)
                    const uint64_t target = jit_chunk_target ? jit_chunk_target : (bs.grp.tracks ? 16384u : 1024u);
                    while (pieces_log2 < 6 && a.log2_p - pieces_log2 > 5 && ((uint64_t)j.nblocks << pieces_log2) < target) ++pieces_log2;
                    // (T = 64: pieces of 256 partials beat 128 and 64 -- 34.5 / 37.3 / 35.3 us at 64 x 4096)
                    if (bs.grp.tracks && a.n_times <= 128 && a.log2_p >= 8 && a.log2_p - pieces_log2 < 8) pieces_log2 = a.log2_p - 8;
                }
                if (pieces_log2) {
                    used_scratch = true;   // (the pieces' workspace is shared between calls: no overlap with the next one on another stream)
                    const uint32_t pv = a.n_voices << pieces_log2;
                    d_chunk_ws.ensure((size_t)pv * blen * sizeof(float));
                    if (d_chunk_rows_n < pv) {
                        std::vector<uint32_t> seq(std::max<uint32_t>(pv, 4096));
                        for (uint32_t i = 0; i < seq.size(); ++i) seq[i] = i;
                        d_chunk_rows.ensure(seq.size() * sizeof(uint32_t));
                        HIP_CHECK(hipMemcpyAsync(d_chunk_rows.p, seq.data(), seq.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                        HIP_CHECK(hipStreamSynchronize(st));   // (`seq` is a stack object; once per growth)
                        d_chunk_rows_n = (uint32_t)seq.size();
                    }
                    j.out = d_chunk_ws.as<float>();
                    j.rows = d_chunk_rows.as<uint32_t>();
                    j.out_stride = blen;
                    j.n_voices = pv;
                    j.log2_p = a.log2_p - pieces_log2;
                    j.nblocks = j.tiles * j.n_voices;
                }
                if (bs.grp.tracks && call_tracks) {
                    if (b0 != idx || blen != n_times) throw Error(FR_ERR_UNSUPPORTED, "internal: a voice that reads tracks rendered over another window than the call's");
                    j.tracks = call_tracks;
                    j.track_stride = call_track_stride;
                    j.track_limit = (uint32_t)std::min<uint64_t>((uint64_t)track_from + call_track_rows, 0xFFFFFFFFull);
                }
                Scope sc(this, &t_bank, st);
                HIP_CHECK(launch_jit_bank(*bs.jit, j, st));
                if (pieces_log2) {
                    ChunkCombineArgs c{};
                    c.ws = d_chunk_ws.as<float>();
                    c.out = a.out;
                    c.rows = a.rows;
                    c.out_stride = a.out_stride;
                    c.n_times = blen;
                    c.n_voices = a.n_voices;
                    c.log2_c = pieces_log2;
                    HIP_CHECK(launch_chunk_combine(c, st));
                }
                sc.done();
                return;
            }
            if (bs.grp.general) {
                a.groups = bs.d_groups.as<uint32_t>();
                a.group_off = bs.d_group_off.as<uint32_t>();
                a.hist_dst = nullptr;   // (the schedule kernel does not append history)
                // many small voices: whole voices per wave (gbank_multi_kernel), like bank_multi_kernel for balanced ones
                a.voices_per_wave = 0;
                if (allow_multi && bs.grp.max_leaves <= 512) {
                    const uint64_t tiles = (blen + 63) / 64;
                    uint32_t vpw = std::max<uint32_t>(1u, std::min<uint32_t>(8u, 256u / std::max<uint32_t>(bs.grp.max_leaves, 1u)));
                    auto nb = [&](uint32_t per_wave) { return ((a.n_voices + 4ull * per_wave - 1) / (4ull * per_wave)) * tiles; };
                    while (vpw > 1 && nb(vpw) < 2048) vpw >>= 1;
                    if (nb(vpw) >= 1024) a.voices_per_wave = vpw;
                }
                Scope sc(this, &t_bank, st);
                HIP_CHECK(launch_gbank(a, st));
                sc.done();
                return;
            }
            // (a host that renders ahead on alternating streams gets launches that can overlap: a GPU's share of a voice-sharded
            //  job, 8 x 4096 x 4800, takes 16.3 us per call that way against 20.9 with chunks + tickets on one stream -- the tail of
            //  one call's few latency-bound waves fills with the next call's first; profiles/r03_fewvoices.txt)
            bank_shape(a.log2_p, a.n_voices, blen, a.chunk_log2, a.frames_per_lane, a.waves_per_group, a.small_call, a.voices_per_wave, host_pipelines);
            if (a.voices_per_wave && !allow_multi) {   // A/B: the quarter-voice-per-wave kernel, one frame per lane
                a.voices_per_wave = 0;
                a.frames_per_lane = 1;
            }
            if (a.small_call == 1 && a.hist_dst) throw Error(FR_ERR_DEVICE, "internal: deferred history append on a short call");
            a.leaf_variant = bank_leaf_variant;
            // small voices, many workgroups: ONE wave per (voice, tile) -- no LDS combine, no barrier (256 x 512 x 4800: 71.6 -> 66.4 us;
            // at 1024 partials and above 4 waves are as fast or faster: profiles/r03_bank_waves.txt)
            if (a.log2_p <= 9 && a.log2_p >= 3 && a.chunk_log2 == a.log2_p && !a.small_call && !a.voices_per_wave && a.waves_per_group == 4 &&
                a.leaf_variant == 1 && !flag_out.host_flags && a.frames_per_lane == 1 && bank_blocks(a) >= 4096)
                a.waves_per_group = 1;
            if (flag_out.host_flags) {
                a.host_flags = flag_out.host_flags;
                a.row_done = flag_out.row_done;
                a.flag_value = flag_out.value;
                if (!bank_publishes_rows(a)) throw Error(FR_ERR_DEVICE, "internal: a bank launch cannot publish row flags");
            }
            if (a.chunk_log2 != a.log2_p) {
                used_scratch = true;
                d_bank_ws.ensure(((size_t)a.n_voices << (a.log2_p - a.chunk_log2)) * blen * sizeof(float));
                a.ws = d_bank_ws.as<float>();
                if (a.small_call == 2) {   // arrival counters of the in-launch combine: zero between launches (the kernel resets them)
                    const size_t need = (size_t)a.n_voices * ((blen + 63) / 64) * BANK_TICKET_STRIDE * sizeof(uint32_t);
                    clean_counters(st);
                    if (need > d_tickets.bytes) {
                        d_tickets.ensure(need * 2);
                        HIP_CHECK(hipMemsetAsync(d_tickets.p, 0, d_tickets.bytes, st));
                    }
                    a.tickets = d_tickets.as<uint32_t>();
                }
            }
            Scope sc(this, &t_bank, st);
            HIP_CHECK(launch_bank(a, st));
            sc.done();
        };
        // Stage programs: the input table of a launch and the launch itself (a range of programs over a window of frames).
        std::vector<DevInput> tab(sp.input_slots.size());
        auto fill_tab = [&] {
            for (size_t i = 0; i < tab.size(); ++i) tab[i] = dev_input(sp.input_slots[i]);
            if (tab.size() > STAGE_INLINE_INPUTS) {   // rare: more input slots than fit in the kernel arguments
                d_in_table_stage.ensure(tab.size() * sizeof(DevInput));
                HIP_CHECK(hipMemcpyAsync(d_in_table_stage.p, tab.data(), tab.size() * sizeof(DevInput), hipMemcpyHostToDevice, st));
                HIP_CHECK(hipStreamSynchronize(st));   // `tab` is a stack object
            }
        };
        uint64_t launch_stride = 0;
        auto launch_range = [&](uint32_t first, uint32_t count, uint64_t s0, uint64_t slen) {
            for (uint32_t off = 0; off < count && plan.stage_jit; off += 65535u) {   // grid.y limit
                JitStageArgs a{};
                a.ptab = plan.d_ptab.as<uint32_t>();
                a.progs = plan.d_jprogs.as<JitStageProg>() + first + off;
                a.rings = d_rings.as<float>();
                a.ring_mask = ring_cap ? ring_cap - 1 : 0;
                a.inputs = reinterpret_cast<const JitInput *>(d_in_table_stage.as<DevInput>());
                a.n_inputs = (uint32_t)tab.size();
                for (size_t i = 0; i < tab.size() && i < STAGE_INLINE_INPUTS; ++i) a.inline_inputs[i] = JitInput{tab[i].data, tab[i].base, tab[i].len};
                a.out = d_dst;
                a.n_times = n_times;
                a.idx = idx;
                a.w0 = s0;
                a.w_len = slen;
                a.stride = launch_stride;
                a.carry_only = (launch_stride && sp.feedback && sp.fused_carry_only) ? 1u : 0u;
                Scope sc(this, &t_stage, st);
                HIP_CHECK(launch_jit_stage(*plan.stage_jit, a, std::min<uint32_t>(count - off, 65535u), st));
                sc.done();
            }
            for (uint32_t off = 0; off < count && !plan.stage_jit; off += 65535u) {
                StageArgs a{};
                a.instrs = plan.d_instrs.as<StageInstr>();
                a.progs = plan.d_progs.as<StageProg>() + first + off;
                a.n_progs = std::min<uint32_t>(count - off, 65535u);
                a.rings = d_rings.as<float>();
                a.ring_mask = ring_cap ? ring_cap - 1 : 0;
                a.inputs = d_in_table_stage.as<DevInput>();
                a.n_inputs = (uint32_t)tab.size();
                for (size_t i = 0; i < tab.size() && i < STAGE_INLINE_INPUTS; ++i) a.inline_inputs[i] = tab[i];
                a.out = d_dst;
                a.n_times = n_times;
                a.idx = idx;
                a.w0 = s0;
                a.w_len = slen;
                a.stride = launch_stride;
                a.carry_only = (launch_stride && sp.feedback && sp.fused_carry_only) ? 1u : 0u;
                a.use_carry = (launch_stride && sp.feedback) ? 1u : 0u;
                a.sparkle = mirror.sparkle ? 1u : 0u;
                Scope sc(this, &t_stage, st);
                HIP_CHECK(launch_stage(a, st));
                sc.done();
            }
        };
        auto launch_fused_levels = [&](uint64_t s0, uint64_t slen) {   // a feedback plan's fused form: a strided launch per level
            launch_stride = sp.fused_stride;
            for (size_t l = 0; l + 1 < sp.fused_level_first.size(); ++l)
                launch_range(sp.fused_first + sp.fused_level_first[l], sp.fused_level_first[l + 1] - sp.fused_level_first[l], s0, slen);
            launch_stride = 0;
        };
        if (fb_replay) {
            fill_tab();
            for (uint64_t c0 = 0; c0 < idx; c0 += FB_CHUNK) {
                const uint64_t len = std::min<uint64_t>(FB_CHUNK, idx - c0);
                for (BankStage &bs : plan.banks)
                    if (bs.grp.to_ring) launch_bank_window(bs, c0, len, -1);
                launch_fused_levels(c0, len);
            }
        }
        // The exchange window first, tile by tile, every tile's bank kernels on the call's stream; then the tiles' exchanges on
        // the exchange stream, each behind its tile's event: tile i's exchange runs under the bank kernels of tiles i + 1 ...
        // (also with a transport that blocks the host: the kernels are all enqueued before the first exchange starts).
        const bool pipelined = !sp.split.empty() && xt.size() > 1;
        if (pipelined) {
            if (!xstream) HIP_CHECK(hipStreamCreateWithFlags(&xstream, hipStreamNonBlocking));
            while (x_events.size() < xt.size() + 1) {
                hipEvent_t e;
                HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                x_events.push_back(e);
            }
        }
        if (!sp.split.empty()) {
            ++exchange_calls;
            exchange_tiles += xt.size();
        }
        for (size_t ti = 0; ti < xt.size(); ++ti) {
            tile_appended.clear();       // (each tile appends its own part of a deferred row, once)
            for (BankStage &bs : plan.banks)
                if (bs.grp.to_ws) launch_bank_window(bs, x0 + xt[ti].first, xt[ti].second, (int64_t)xt[ti].first);
            if (pipelined) HIP_CHECK(hipEventRecord(x_events[ti], st));
        }
        for (size_t ti = 0; ti < xt.size(); ++ti) {
            if (pipelined) HIP_CHECK(hipStreamWaitEvent(xstream, x_events[ti], 0));
            run_exchange(d_dst, n_times, idx, x0, xt[ti].first, xt[ti].second, pipelined ? xstream : st);
        }
        if (!xt.empty() && x_window_is_call)
            for (Deferred &d : deferred)
                if (tile_slots.count(d.slot)) d.dst = nullptr;               // appended tile by tile
        // everything else (voices that stay whole on this rank, unsharded plans) -- overlapping the exchange's tail
        for (BankStage &bs : plan.banks) {
            if (bs.grp.to_ws) continue;
            const bool ring = bs.grp.to_ring;
            launch_bank_window(bs, ring ? w0 : idx, ring ? w_len : n_times, -1);
        }
        if (pipelined) {   // the call's stream goes on only behind the last tile's exchange
            HIP_CHECK(hipEventRecord(x_events[xt.size()], xstream));
            HIP_CHECK(hipStreamWaitEvent(st, x_events[xt.size()], 0));
        }

        for (const Deferred &d : deferred)
            if (d.dst) throw Error(FR_ERR_DEVICE, "internal: an input row deferred to the bank launch was not appended");
        if (!sp.progs.empty()) {
            fill_tab();
            // Steady state: every delayed ring read of the fused form reaches at least fused_max_frames back, so the call
            // is cut into sub-windows of that length, one fused launch each, when that takes fewer launches than levels.
            const size_t n_levels = sp.level_first.size() - 1;
            const uint64_t fused_step = std::max<uint64_t>(sp.fused_max_frames, 1);
            const uint64_t n_sub = sp.fused_count ? (n_times - 1) / fused_step + 1 : 0;   // (n_times > 0 here; no overflow)
            const bool fused = sp.fused_count != 0 && w0 == idx && plan.stage_valid && n_sub < n_levels;
            // ... or ONE launch whose threads stride through the sub-windows themselves, when every delayed read of a program ring
            // reaches back a multiple of fused_stride frames into a ring its own program stores (the delay chains of an effects
            // patch: 2400, 4800, 7200 ...): a thread then reads only what it stored itself.  Worth it for a handful of strides
            // (each one is a dependent round trip to memory inside the launch; a launch boundary costs ~5 us at this size).
            const uint64_t strided_sub = sp.fused_stride ? (n_times - 1) / sp.fused_stride + 1 : 0;
            const bool strided = sp.fused_count != 0 && w0 == idx && plan.stage_valid && sp.fused_stride >= 256 && strided_sub >= 2 &&
                                 strided_sub <= 8 && fused_strided_ok;
            if (sp.feedback) {
                launch_fused_levels(idx, n_times);
                launch_range(sp.post_first, sp.post_count, idx, n_times);
            } else if (strided) {
                launch_stride = sp.fused_stride;
                launch_range(sp.fused_first, sp.fused_count, idx, n_times);
                launch_stride = 0;
            } else if (fused) {
                for (uint64_t done = 0; done < n_times;) {   // by frames still to do: no sum that could wrap
                    const uint64_t len = std::min<uint64_t>(fused_step, n_times - done);
                    launch_range(sp.fused_first, sp.fused_count, idx + done, len);
                    done += len;
                }
            } else {
                for (size_t l = 0; l < n_levels; ++l)
                    launch_range(sp.level_first[l], sp.level_first[l + 1] - sp.level_first[l], w0, w_len);
            }
        }
        if (sp.uses_rings()) {
            plan.stage_valid = true;
            plan.stage_end = idx + n_times;
        }
        if (!plan.pull_rows.empty()) run_pull(d_dst, n_slots, n_times, idx, st);
    }

    void run_pull(float *d_dst, uint32_t n_slots, uint64_t n_times, uint64_t idx, hipStream_t st) {
        // input table for this call
        std::vector<DevInput> tab(plan.input_slots.size());
        for (size_t i = 0; i < tab.size(); ++i) tab[i] = dev_input(plan.input_slots[i]);
        d_in_table.ensure(std::max<size_t>(tab.size(), 1) * sizeof(DevInput));
        if (!tab.empty()) {
            HIP_CHECK(hipMemcpyAsync(d_in_table.p, tab.data(), tab.size() * sizeof(DevInput), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st));
        }
        const uint64_t depth = std::max<uint32_t>(plan.max_depth, 1);
        // pull rows are processed as contiguous runs of rows; stack workspace bounded to ~1 GiB
        const uint64_t budget = 1ull << 30;
        uint64_t chunk = std::max<uint64_t>(budget / (depth * 16), 256);
        size_t i = 0;
        while (i < plan.pull_rows.size()) {
            size_t j = i + 1;
            while (j < plan.pull_rows.size() && plan.pull_rows[j] == plan.pull_rows[j - 1] + 1) ++j;
            uint64_t first = (uint64_t)plan.pull_rows[i] * n_times;
            uint64_t total = (uint64_t)(j - i) * n_times;
            for (uint64_t off = 0; off < total; off += chunk) {
                uint64_t cnt = std::min(chunk, total - off);
                d_stack_node.ensure(depth * cnt * sizeof(uint32_t));
                d_stack_time.ensure(depth * cnt * sizeof(uint64_t));
                d_stack_val.ensure(depth * cnt * sizeof(float));
                PullArgs a{};
                a.nodes = plan.d_nodes.as<DevNode>();
                a.outputs = plan.d_roots.as<uint32_t>();
                a.inputs = d_in_table.as<DevInput>();
                a.n_inputs = (uint32_t)tab.size();
                a.out = d_dst;
                a.n_slots = n_slots;
                a.n_times = n_times;
                a.idx = idx;
                a.first = first + off;
                a.count = cnt;
                a.sparkle = mirror.sparkle ? 1u : 0u;
                a.st_node = d_stack_node.as<uint32_t>();
                a.st_time = d_stack_time.as<uint64_t>();
                a.st_val = d_stack_val.as<float>();
                Scope sc(this, &t_pull, st);
                HIP_CHECK(launch_pull(a, st));
                sc.done();
            }
            i = j;
        }
    }
};

namespace {

template <class F>
fr_status guarded(fr_renderer *r, F &&f, bool keeps_stream = false) {
    if (!r) return FR_ERR_INVALID_ARG;
    try {
        if (r->streaming && !keeps_stream) r->end_stream();   // any other call first retires the resident launch
        f();
        r->last_error.clear();
        return FR_OK;
    } catch (const Error &e) {
        r->last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        r->last_error = "host out of memory";
        return FR_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        r->last_error = e.what();
        return FR_ERR_INVALID_ARG;
    }
}

void check_fill_args(const void *out, uint32_t n_slots, uint64_t n_times, const float *in_data,
                     const uint64_t *offs, uint32_t n_rows) {
    if (!out && n_slots != 0 && n_times != 0) throw Error(FR_ERR_INVALID_ARG, "null output buffer");
    if (n_rows && !offs) throw Error(FR_ERR_INVALID_ARG, "null row offsets");
    if (n_rows && offs[n_rows] > offs[0] && !in_data) throw Error(FR_ERR_INVALID_ARG, "null input data");
    if (n_slots && n_times > (1ull << 40) / n_slots) throw Error(FR_ERR_INVALID_ARG, "render range too large");
}

}  // namespace

extern "C" {

fr_status fr_renderer_create(const fr_config *cfg, fr_renderer **out) {
    if (!out) return FR_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg && cfg->abi_version != FR_ABI_VERSION) return FR_ERR_INVALID_ARG;
    int mode = cfg ? cfg->mode : FR_MODE_AUTO;
    if (mode < FR_MODE_AUTO || mode > FR_MODE_STAGED) return FR_ERR_INVALID_ARG;
    if (cfg && ((cfg->flags & ~FR_CONFIG_SYNC_COMPILE) != 0 || cfg->reserved != 0)) return FR_ERR_INVALID_ARG;
    if (cfg && cfg->semantics != FR_SEMANTICS_REFERENCE && cfg->semantics != FR_SEMANTICS_SPARKLE) return FR_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FR_ERR_NO_DEVICE;
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) return FR_ERR_NO_DEVICE;
    }
    if (dev >= ndev) return FR_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return FR_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FR_ERR_NO_DEVICE;   // code objects are gfx950 only
    if (hipSetDevice(dev) != hipSuccess) return FR_ERR_NO_DEVICE;
    fr_renderer *r = new (std::nothrow) fr_renderer();
    if (!r) return FR_ERR_OUT_OF_MEMORY;
    // The first C++ exception a process throws makes the unwinder walk every loaded object's unwind tables (under the
    // loader's lock: other throwing threads queue behind it) -- 82 ms measured inside a fill_buffer call with the HIP
    // runtime and hipRTC loaded.  The engine uses exceptions on rare paths of a call (a lowering thread handing its sub-tree
    // to the sequential pass, a program a compiled kernel cannot take); pay for the walk here instead.
    try { throw Error(FR_OK, ""); } catch (const Error &) {}
    r->device = dev;
    r->mode = mode;
    r->jit_async_configured = !(cfg && (cfg->flags & FR_CONFIG_SYNC_COMPILE));
    r->jit_cache.set_async(r->jit_async_configured);
    r->jit_cache.set_sparkle(cfg && cfg->semantics == FR_SEMANTICS_SPARKLE);
    r->mirror.sparkle = cfg && cfg->semantics == FR_SEMANTICS_SPARKLE;
    r->semantics = cfg ? cfg->semantics : FR_SEMANTICS_REFERENCE;
    r->history_frames = cfg ? cfg->history_frames : 0;
    if (const char *lv = std::getenv("FR_BANK_LEAF")) r->bank_leaf_variant = (lv[0] == '1') ? 1u : 0u;
    if (const char *jv = std::getenv("FR_JIT")) r->allow_jit = jv[0] != '0';
    if (const char *tv = std::getenv("FR_BANK_TEMPLATE")) r->allow_template = tv[0] != '0';
    if (const char *mv = std::getenv("FR_BANK_MULTI")) r->allow_multi = mv[0] != '0';
    if (const char *tv2 = std::getenv("FR_HOST_TRACE")) r->host_trace = tv2[0] == '1';
    if (const char *dv = std::getenv("FR_HOST_DIRECT")) r->host_direct = dv[0] != '0';
    if (const char *kv = std::getenv("FR_HOST_SMALL_KB")) r->host_small_bytes = (size_t)std::max(0, std::atoi(kv)) << 10;
    if (const char *sv2 = std::getenv("FR_HOST_STREAM")) r->host_stream = sv2[0] != '0';
    if (const char *hv = std::getenv("FR_HOST_MAPPED")) {
        const int m = std::atoi(hv);
        r->host_out_mapped = (m & 1) != 0;
        r->host_rows_mapped = (m & 2) != 0;
    }
    if (const char *sv = std::getenv("FR_STAGE_JIT")) r->stage_jit_mode = sv[0] == '0' ? 0 : (sv[0] == '1' ? 1 : 2);
    if (const char *fv = std::getenv("FR_STAGE_STRIDED")) r->fused_strided_ok = fv[0] != '0';
    if (const char *cv = std::getenv("FR_JIT_CHUNKS")) r->jit_chunks = cv[0] != '0';
    if (const char *bv = std::getenv("FR_STAGE_BLOCK")) r->stage_block_env = (uint32_t)std::max(0, std::atoi(bv));
    if (const char *cv = std::getenv("FR_JIT_CHUNK_TARGET")) r->jit_chunk_target = (uint64_t)std::max(1, std::atoi(cv));
    if (const char *xv = std::getenv("FR_EXCHANGE_TILES")) { r->x_max_tiles = (uint32_t)std::min(64, std::max(1, std::atoi(xv))); r->x_tiles_explicit = true; }
    if (const char *xv = std::getenv("FR_EXCHANGE_MIN_TILE")) { r->x_min_tile = (uint32_t)std::min(1 << 20, std::max(64, std::atoi(xv))); r->x_tiles_explicit = true; }
    if (const char *iv = std::getenv("FR_STREAM_IDLE_MS")) r->stream_idle_ms = (uint32_t)std::min(60000, std::max(1, std::atoi(iv)));
    r->device_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) {
        delete r;
        return FR_ERR_DEVICE;
    }
    *out = r;
    return FR_OK;
}

void fr_renderer_destroy(fr_renderer *r) { delete r; }

fr_status fr_on_add_node(fr_renderer *r, uint32_t handle, const fr_effect *effect) {
    return guarded(r, [&] { r->mirror.add_node(handle, effect); });
}
fr_status fr_on_del_node(fr_renderer *r, uint32_t handle) {
    return guarded(r, [&] { r->mirror.del_node(handle); });
}
fr_status fr_on_add_edge(fr_renderer *r, const fr_edge *edge) {
    return guarded(r, [&] {
        if (!edge) throw Error(FR_ERR_INVALID_ARG, "null edge");
        r->mirror.add_edge(*edge);
    });
}
fr_status fr_on_del_edge(fr_renderer *r, const fr_edge *edge) {
    return guarded(r, [&] {
        if (!edge) throw Error(FR_ERR_INVALID_ARG, "null edge");
        r->mirror.del_edge(*edge);
    });
}
fr_status fr_on_add_nodes(fr_renderer *r, const uint32_t *handles, const fr_effect *const *effects, size_t n) {
    return guarded(r, [&] {
        if (n && (!handles || !effects)) throw Error(FR_ERR_INVALID_ARG, "null array");
        r->mirror.nodes.reserve(r->mirror.nodes.size() + n);
        for (size_t i = 0; i < n; ++i) r->mirror.add_node(handles[i], effects[i]);
    });
}
fr_status fr_on_add_edges(fr_renderer *r, const fr_edge *edges, size_t n) {
    return guarded(r, [&] {
        if (n && !edges) throw Error(FR_ERR_INVALID_ARG, "null array");
        for (size_t i = 0; i < n; ++i) r->mirror.add_edge(edges[i]);
    });
}

fr_status fr_fill_buffer(fr_renderer *r, float *out, uint32_t n_slots, uint64_t n_times, uint64_t idx,
                         const float *in_data, const uint64_t *in_row_offsets, uint32_t n_in_rows) {
    return guarded(r, [&] {
        check_fill_args(out, n_slots, n_times, in_data, in_row_offsets, n_in_rows);
        HIP_CHECK(hipSetDevice(r->device));
        hipStream_t st = r->stream;
        r->order_after_previous(st);
        r->ensure_plan(n_slots, st);           // graph errors surface before the input store is touched
        const auto snap = r->snapshot_store(idx);
        try {
        r->store_inputs(n_slots, n_times, idx, in_data, in_row_offsets, n_in_rows, false, st);
        size_t bytes = (size_t)n_slots * n_times * sizeof(float);
        // (Rendering a long call as 2-4 sub-calls so that chunk c's D2H overlaps chunk c+1's kernels was tried: every
        //  extra sub-call costs ~35 us of launch/sync overhead and smaller, less efficient launches -- 197 us became
        //  231 / 242 / 277 us at 2 / 3 / 4 chunks for config C; profiles/r01_host_path.txt.)
        // sharded: only the rows this rank owns come back (rank 0 under FR_SHARD_GATHER: every row)
        uint32_t row_lo, row_hi;
        r->my_rows(n_slots, row_lo, row_hi);
        const bool gather = r->sharded() && (r->shard_flags & FR_SHARD_GATHER);
        // (Pipelining the call -- row groups chained over a copy stream with events, or two halves on two streams of
        //  different priority -- was measured and is slower: every cross-stream hand-off costs tens of microseconds on this
        //  stack, profiles/r02_host_path.txt.  What helps is not moving bytes twice.)
        using clk = std::chrono::steady_clock;
        const auto t_a = clk::now();
        // A destination the host has page-locked (fr_host_register) is written by the kernels themselves: no copy of the
        // 1.2 MB at all (the launch runs ~19 us longer for its stores crossing PCIe, against ~40 us of D2H).
        float *direct = nullptr;
        if (!gather && bytes && r->host_direct)
            for (const auto &rg : r->registered)
                if ((char *)out >= rg.first && (char *)out + bytes <= rg.first + rg.second) {
                    void *dp = nullptr;
                    if (hipHostGetDevicePointer(&dp, out, 0) == hipSuccess) direct = (float *)dp;
                    else (void)hipGetLastError();
                    break;
                }
        if (direct) {
            r->execute(direct, n_slots, n_times, idx, st);
            HIP_CHECK(hipStreamSynchronize(st));   // synchronous contract: dispatch.rs:150-151
        } else if (bytes >= (256u << 10) && r->can_stream_rows(n_slots, n_times)) {   // (below that one D2H is as quick)
            // Streamed: rows are copied to the caller's buffer as their flags arrive, under the rest of the launch.
            r->h_out_stage.ensure(bytes);
            if ((size_t)n_slots * sizeof(uint32_t) > r->h_row_flags.bytes) {
                r->h_row_flags.ensure((size_t)n_slots * 2 * sizeof(uint32_t));
                std::memset(r->h_row_flags.p, 0, r->h_row_flags.bytes);
                r->host_seq = 0;
            }
            r->clean_counters(st);
            if ((size_t)n_slots * sizeof(uint32_t) > r->d_row_done.bytes) {
                r->d_row_done.ensure((size_t)n_slots * 2 * sizeof(uint32_t));
                HIP_CHECK(hipMemsetAsync(r->d_row_done.p, 0, r->d_row_done.bytes, st));
            }
            if (++r->host_seq == 0) {   // (wrapped: flags restart from a clean slate)
                std::memset(r->h_row_flags.p, 0, r->h_row_flags.bytes);
                r->host_seq = 1;
            }
            const uint32_t seq = r->host_seq;
            r->flag_out.host_flags = r->h_row_flags.as_dev<uint32_t>();
            r->flag_out.row_done = r->d_row_done.as<uint32_t>();
            r->flag_out.value = seq;
            struct Clear { fr_renderer *r; ~Clear() { r->flag_out = fr_renderer::FlagOut{}; } } clear{r};
            r->execute(r->h_out_stage.as_dev<float>(), n_slots, n_times, idx, st);
            const uint32_t *flags = r->h_row_flags.as<uint32_t>();
            const float *stage = r->h_out_stage.as<float>();
            r->stream_pending.resize(n_slots);
            for (uint32_t i = 0; i < n_slots; ++i) r->stream_pending[i] = i;
            size_t left = n_slots;
            uint64_t idle = 0;
            while (left) {
                bool progress = false;
                for (size_t i = 0; i < left;) {
                    const uint32_t row = r->stream_pending[i];
                    if (__atomic_load_n(flags + row, __ATOMIC_ACQUIRE) == seq) {
                        std::memcpy(out + (size_t)row * n_times, stage + (size_t)row * n_times, n_times * sizeof(float));
                        r->stream_pending[i] = r->stream_pending[--left];
                        progress = true;
                    } else {
                        ++i;
                    }
                }
                if (!progress && (++idle & 0xFFFFu) == 0 && hipStreamQuery(st) != hipErrorNotReady) {
                    // the launch is over (or failed) and a flag never came: report rather than spin for ever
                    HIP_CHECK(hipStreamSynchronize(st));
                    for (size_t i = 0; i < left; ++i)
                        if (__atomic_load_n(flags + r->stream_pending[i], __ATOMIC_ACQUIRE) != seq)
                            throw Error(FR_ERR_DEVICE, "internal: a rendered row was never published to the host");
                }
            }
            HIP_CHECK(hipStreamSynchronize(st));   // (everything is done; this only retires the launch)
        } else if ((r->host_out_mapped || bytes <= r->host_small_bytes) && !gather) {
            // small results (a real-time block: 64 frames x 64 voices = 16 KB): the kernels store straight into mapped pinned
            // memory and the host copies the few KB itself -- a D2H copy costs ~12 us of launch latency more than it moves
            // (64-frame call at config C: 35 -> 23 us, tools/host_short_probe.py); large results pay for it in cold CPU reads
            // kernels store finished frames straight into mapped pinned memory; one wait, one copy to the caller's buffer
            r->h_out_stage.ensure(bytes);
            r->execute(r->h_out_stage.as_dev<float>(), n_slots, n_times, idx, st);
            const auto t_b = clk::now();
            HIP_CHECK(hipStreamSynchronize(st));   // synchronous contract: dispatch.rs:150-151
            const auto t_c = clk::now();
            const size_t off = (size_t)row_lo * n_times, cnt = (size_t)(row_hi - row_lo) * n_times;
            if (cnt) std::memcpy(out + off, r->h_out_stage.as<float>() + off, cnt * sizeof(float));
            if (r->host_trace) {
                r->trace_us[0] += std::chrono::duration<double, std::micro>(t_b - t_a).count();
                r->trace_us[1] += std::chrono::duration<double, std::micro>(t_c - t_b).count();
                r->trace_us[2] += std::chrono::duration<double, std::micro>(clk::now() - t_c).count();
                ++r->trace_n;
            }
        } else {
            r->d_out.ensure(bytes);
            r->execute(r->d_out.as<float>(), n_slots, n_times, idx, st);
            if (gather) {
                r->gather_rows(r->d_out.as<float>(), n_slots, n_times, st);
                if (r->shard.rank == 0) { row_lo = 0; row_hi = n_slots; }
            }
            const size_t off = (size_t)row_lo * n_times, cnt = (size_t)(row_hi - row_lo) * n_times;
            const auto t_b = clk::now();
            if (cnt) HIP_CHECK(hipMemcpyAsync(out + off, r->d_out.as<float>() + off, cnt * sizeof(float), hipMemcpyDeviceToHost, st));
            const auto t_c = clk::now();
            HIP_CHECK(hipStreamSynchronize(st));
            if (r->host_trace) {
                r->trace_us[0] += std::chrono::duration<double, std::micro>(t_b - t_a).count();
                r->trace_us[1] += std::chrono::duration<double, std::micro>(t_c - t_b).count();
                r->trace_us[2] += std::chrono::duration<double, std::micro>(clk::now() - t_c).count();
                ++r->trace_n;
            }
        }
        } catch (...) {
            r->rollback_store(snap);
            throw;
        }
        r->last_pending = false;               // everything issued so far, on any stream, is complete (order_after_previous)
        r->head = idx + n_times;               // reference.rs:84
    });
}

fr_status fr_fill_buffer_device(fr_renderer *r, float *d_out, uint32_t n_slots, uint64_t n_times, uint64_t idx,
                                const float *d_in_data, const uint64_t *in_row_offsets, uint32_t n_in_rows,
                                void *stream) {
    return guarded(r, [&] {
        check_fill_args(d_out, n_slots, n_times, d_in_data, in_row_offsets, n_in_rows);
        HIP_CHECK(hipSetDevice(r->device));
        hipStream_t st = (hipStream_t)stream;
        // Independent of the previous call (may overlap with it on another stream): a plan without state, no seek, no new
        // plan, and every row full length -- padding a short row reads the slot's last stored sample, which the previous
        // call may still be writing.
        bool independent = r->plan_is_stateless(n_slots) && idx == r->head;
        for (uint32_t i = 0; independent && i < n_in_rows; ++i) independent = in_row_offsets[i + 1] - in_row_offsets[i] == n_times;
        r->host_pipelines = independent && r->last_pending && r->last_stream != st;
        struct Reset { fr_renderer *r; ~Reset() { r->host_pipelines = false; } } reset{r};
        r->order_after_previous(st, independent);
        r->used_scratch = false;
        r->ensure_plan(n_slots, st);
        const auto snap = r->snapshot_store(idx);
        try {
            r->store_inputs(n_slots, n_times, idx, d_in_data, in_row_offsets, n_in_rows, true, st);
            r->execute(d_out, n_slots, n_times, idx, st);
            r->gather_rows(d_out, n_slots, n_times, st);
        } catch (...) {
            r->rollback_store(snap);
            throw;
        }
        r->remember_async(st, independent && !r->used_scratch);
        r->head = idx + n_times;
    });
}

fr_status fr_set_track_inputs(fr_renderer *r, uint32_t first_slot) {
    return guarded(r, [&] {
        if (r->track_from == first_slot) return;
        r->track_from = first_slot;
        r->plan.valid = false;      // voices are matched differently
        r->matcher.reset();
    });
}

namespace {
fr_status fill_dense(fr_renderer *r, float *out, uint32_t n_slots, uint64_t n_times, uint64_t idx, const float *in, uint32_t n_in_rows,
                     bool device, void *stream) {
    if (!r) return FR_ERR_INVALID_ARG;
    if (n_in_rows && n_times && !in) { r->last_error = "null input matrix"; return FR_ERR_INVALID_ARG; }
    const uint32_t stored = std::min(n_in_rows, r->track_from);
    std::vector<uint64_t> offs((size_t)stored + 1);
    for (uint32_t i = 0; i <= stored; ++i) offs[i] = (uint64_t)i * n_times;
    r->dense_total_rows = n_in_rows;
    const fr_status st = device ? fr_fill_buffer_device(r, out, n_slots, n_times, idx, in, offs.data(), stored, stream)
                                : fr_fill_buffer(r, out, n_slots, n_times, idx, in, offs.data(), stored);
    r->dense_total_rows = 0;
    return st;
}
}   // namespace

fr_status fr_fill_buffer_dense(fr_renderer *r, float *out, uint32_t n_slots, uint64_t n_times, uint64_t idx, const float *in, uint32_t n_in_rows) {
    return fill_dense(r, out, n_slots, n_times, idx, in, n_in_rows, false, nullptr);
}
fr_status fr_fill_buffer_device_dense(fr_renderer *r, float *d_out, uint32_t n_slots, uint64_t n_times, uint64_t idx, const float *d_in,
                                      uint32_t n_in_rows, void *stream) {
    return fill_dense(r, d_out, n_slots, n_times, idx, d_in, n_in_rows, true, stream);
}

fr_status fr_host_register(fr_renderer *r, void *p, size_t bytes) {
    return guarded(r, [&] {
        if (!p || !bytes) throw Error(FR_ERR_INVALID_ARG, "empty range");
        HIP_CHECK(hipSetDevice(r->device));
        HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterMapped));
        r->registered.push_back({(char *)p, bytes});
    });
}

fr_status fr_host_unregister(fr_renderer *r, void *p) {
    return guarded(r, [&] {
        HIP_CHECK(hipSetDevice(r->device));
        for (size_t i = 0; i < r->registered.size(); ++i)
            if (r->registered[i].first == (char *)p) { r->registered.erase(r->registered.begin() + (ptrdiff_t)i); break; }
        HIP_CHECK(hipHostUnregister(p));
    });
}

fr_status fr_comm_unique_id(uint8_t id[FR_COMM_ID_BYTES]) {
    if (!id) return FR_ERR_INVALID_ARG;
    try {
        rccl_unique_id(id);
        return FR_OK;
    } catch (const Error &e) {
        return e.code;
    }
}

// ---- block streaming ------------------------------------------------------------------------------------------------------
fr_status fr_stream_begin(fr_renderer *r, uint32_t n_slots) {
    return guarded(r, [&] {
        HIP_CHECK(hipSetDevice(r->device));
        if (n_slots == 0) throw Error(FR_ERR_INVALID_ARG, "no output slots");
        if (r->sharded()) throw Error(FR_ERR_UNSUPPORTED, "block streaming of a sharded renderer");
        r->order_after_previous(r->stream);
        r->ensure_plan(n_slots, r->stream);
        const StagedPlan &sp = r->plan.sp;
        // what one resident launch can serve: every row straight from one balanced template voice, nothing stored between calls
        if (!r->plan_is_stateless(n_slots) || r->plan.banks.size() != 1)
            throw Error(FR_ERR_UNSUPPORTED, "block streaming needs a plan that is one voice bank (this one: " + std::to_string(r->plan.banks.size()) + " bank launches, " +
                                                std::to_string(sp.progs.size()) + " programs, " + std::to_string(r->plan.pull_rows.size()) + " pull rows" +
                                                (sp.uses_rings() ? ", rings" : "") + (r->plan_current(n_slots) ? "" : ", plan not current") + ")");
        const BankStage &bs = r->plan.banks[0];
        if (bs.grp.general || bs.grp.jit || bs.grp.to_ring || bs.grp.to_ws || bs.grp.rows.size() != n_slots || !sp.pull_rows.empty())
            throw Error(FR_ERR_UNSUPPORTED, "block streaming needs balanced template voices, one per output row");
        if (r->bank_leaf_variant != 1) throw Error(FR_ERR_UNSUPPORTED, "block streaming with FR_BANK_LEAF=0");
        if (bs.grp.input_slot != 0) throw Error(FR_ERR_UNSUPPORTED, "block streaming feeds input slot 0; these voices read another slot");
        if (bs.grp.log2_p < 7) throw Error(FR_ERR_UNSUPPORTED, "block streaming needs voices of at least 128 partials (16 waves x one group of 8)");
        BankArgs a{};
        a.params = bs.d_params.as<float2>();
        a.rows = bs.d_rows.as<uint32_t>();
        a.n_voices = n_slots;
        a.log2_p = bs.grp.log2_p;
        a.n_times = 64;
        a.fast_ok = bs.grp.fast_ok ? 1u : 0u;
        a.leaf_variant = 1;
        a.small_call = 2;
        a.waves_per_group = 16;
        a.frames_per_lane = 1;
        // Every workgroup of the launch must be resident at once (a workgroup that never starts never counts its voice in), one
        // per CU: as many as the device has CUs (a partitioned gfx950 has fewer than 256), and no more than the kernel's limit.
        const uint64_t max_wgs = std::min<uint64_t>(BANK_STREAM_WGS, (uint64_t)std::max(r->device_cus, 1));
        uint32_t c = a.log2_p;                    // chunks of >= 128 partials (a wave needs a group of 8) until the CUs are used
        while (c > 7 && ((uint64_t)n_slots << (a.log2_p - c + 1)) <= max_wgs && a.log2_p - c < 8) --c;
        a.chunk_log2 = c;
        if (((uint64_t)n_slots << (a.log2_p - c)) > max_wgs)
            throw Error(FR_ERR_UNSUPPORTED, "block streaming serves at most one voice per CU (" + std::to_string(max_wgs) + " here)");
        if (c != a.log2_p) {
            r->d_bank_ws.ensure(((size_t)n_slots << (a.log2_p - c)) * 64 * sizeof(float));
            a.ws = r->d_bank_ws.as<float>();
            const size_t need = (size_t)n_slots * BANK_TICKET_STRIDE * sizeof(uint32_t);
            r->clean_counters(r->stream);
            if (need > r->d_tickets.bytes) {
                r->d_tickets.ensure(need * 2);
                HIP_CHECK(hipMemsetAsync(r->d_tickets.p, 0, r->d_tickets.bytes, r->stream));
            }
            a.tickets = r->d_tickets.as<uint32_t>();
        }
        r->h_stream_ctl.ensure(sizeof(BankStreamCtl));
        r->h_stream_out.ensure((size_t)n_slots * 64 * sizeof(float));
        r->d_stream_dev.ensure(sizeof(BankStreamDev));
        std::memset(r->h_stream_ctl.p, 0, sizeof(BankStreamCtl));
        HIP_CHECK(hipMemsetAsync(r->d_stream_dev.p, 0, sizeof(BankStreamDev), r->stream));
        a.out = r->h_stream_out.as_dev<float>();
        a.out_stride = 64;
        HIP_CHECK(launch_bank_stream(a, r->h_stream_ctl.as_dev<BankStreamCtl>(), r->d_stream_dev.as<BankStreamDev>(), r->stream_idle_ms, r->stream));
        r->streaming = true;
        r->stream_seq = 0;
        r->stream_slots = n_slots;
        r->stream_have_last = false;
        r->last_pending = false;
    });
}

fr_status fr_stream_block(fr_renderer *r, float *out, uint64_t n_times, uint64_t idx, const float *row, uint64_t row_len) {
    return guarded(r, [&] {
        if (!r->streaming) throw Error(FR_ERR_INVALID_ARG, "no stream is open (fr_stream_begin; any other call on the renderer closes it)");
        if (!out || n_times == 0 || n_times > 64 || row_len > n_times || (row_len && !row)) throw Error(FR_ERR_INVALID_ARG, "a streamed block is 1..64 frames");
        // (a plan served here reads nothing but this block's row: `idx` enters only through the padding rule)
        BankStreamCtl *ctl = (BankStreamCtl *)r->h_stream_ctl.p;
        const auto t_in = std::chrono::steady_clock::now();
        // the row, padded like a short row of fill_buffer (reference.rs:72-73) with the slot's last stored value: its own last
        // value, or -- an empty row -- the last value of the block it continues (idx == where that block ended); the first
        // block of a stream and a block that does not continue the previous one are what a seek leaves: nothing stored, pad 0.
        // Every word is tagged with the block's number and length: the words are the doorbell (kernels.hpp BankStreamCtl)
        const float pad = row_len ? row[row_len - 1] : ((r->stream_have_last && idx == r->stream_head) ? r->stream_last : 0.0f);
        r->stream_seq = (r->stream_seq + 1u) & 0xFFFFFFu;
        if (r->stream_seq == 0 || r->stream_seq == 0xFFFFFFu) r->stream_seq = 1;
        const uint32_t seq = r->stream_seq << 8 | (uint32_t)n_times;
        for (uint64_t i = 0; i < 64; ++i) {
            const float v = i < row_len ? row[i] : (i < n_times ? pad : 0.0f);
            uint32_t bits;
            std::memcpy(&bits, &v, 4);
            __atomic_store_n(&ctl->row[i], (unsigned long long)seq << 32 | bits, __ATOMIC_RELAXED);
        }
        uint64_t spins = 0;
        const auto t_ring = std::chrono::steady_clock::now();
        while (__atomic_load_n(&ctl->done, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 0xFFFFFu) != 0) continue;
            if (hipStreamQuery(r->stream) != hipErrorNotReady) {   // the launch is gone (its own bound, or a fault)
                (void)hipStreamSynchronize(r->stream);
                r->streaming = false;
                r->stream_have_last = false;
                r->counters_dirty = true;                          // (it may have ended between two chunks of a voice)
                r->head = UINT64_MAX;
                throw Error(FR_ERR_DEVICE, "the resident launch ended before the block was rendered");
            }
            // a resident launch answers in tens of microseconds; a quarter of a second without an answer means it is not all
            // resident (something else holds CUs) or the device is in trouble: give the audio thread back
            if (std::chrono::steady_clock::now() - t_ring > std::chrono::milliseconds(250)) {
                r->end_stream(false);
                throw Error(FR_ERR_DEVICE, "the resident launch did not answer within 250 ms");
            }
        }
        const auto t_done = std::chrono::steady_clock::now();
        const float *res = r->h_stream_out.as<float>();
        for (uint32_t v = 0; v < r->stream_slots; ++v) std::memcpy(out + (size_t)v * n_times, res + (size_t)v * 64, n_times * sizeof(float));
        r->stream_last = n_times <= row_len ? row[n_times - 1] : pad;
        r->stream_head = idx + n_times;
        r->stream_have_last = true;
        if (r->host_trace) {   // FR_HOST_TRACE=1: inside the call, without the caller's wrapper
            r->stream_trace_us[0] += std::chrono::duration<double, std::micro>(t_done - t_in).count();
            r->stream_trace_us[1] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_done).count();
            ++r->stream_trace_n;
        }
    }, true);
}

fr_status fr_stream_end(fr_renderer *r) {
    return guarded(r, [&] { HIP_CHECK(hipSetDevice(r->device)); });   // (guarded() itself retires the launch)
}

fr_status fr_comm_selftest(int32_t device, uint64_t n_floats) {
    if (n_floats == 0 || n_floats > (1ull << 26)) return FR_ERR_INVALID_ARG;   // (256 MB each way is plenty for a check)
    float *d_send = nullptr, *d_recv = nullptr;
    hipStream_t st = nullptr;
    fr_status rc = FR_OK;
    try {
        if (device >= 0) HIP_CHECK(hipSetDevice(device));
        uint8_t id[FR_COMM_ID_BYTES];
        rccl_unique_id(id);
        std::unique_ptr<Transport> t = make_rccl_transport(id, 0, 1);
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIP_CHECK(hipMalloc((void **)&d_send, n_floats * sizeof(float)));
        HIP_CHECK(hipMalloc((void **)&d_recv, n_floats * sizeof(float)));
        std::vector<float> h(n_floats), back(n_floats, -1.0f);
        for (uint64_t i = 0; i < n_floats; ++i) h[i] = (float)(i % 8191) * 0.25f - 3.0f;
        HIP_CHECK(hipMemcpyAsync(d_send, h.data(), n_floats * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemsetAsync(d_recv, 0xFF, n_floats * sizeof(float), st));
        t->sendrecv(0, d_send, n_floats, d_recv, n_floats, st);
        HIP_CHECK(hipMemcpyAsync(back.data(), d_recv, n_floats * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (std::memcmp(h.data(), back.data(), n_floats * sizeof(float)) != 0) rc = FR_ERR_COMM;
    } catch (const Error &e) {
        rc = e.code;
    } catch (...) {
        rc = FR_ERR_DEVICE;
    }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    return rc;
}

fr_status fr_set_shard(fr_renderer *r, const fr_shard *sh) {
    return guarded(r, [&] {
        HIP_CHECK(hipSetDevice(r->device));
        HIP_CHECK(hipDeviceSynchronize());   // nothing of the previous arrangement is still in flight
        r->last_pending = false;
        ShardSpec spec;
        uint32_t flags = 0;
        std::unique_ptr<Transport> transport;
        fr_comm comm{};
        bool has_comm = false;
        if (sh && sh->world > 1 && sh->mode != FR_SHARD_NONE) {
            if (sh->mode != FR_SHARD_VOICES && sh->mode != FR_SHARD_PARTIALS) throw Error(FR_ERR_INVALID_ARG, "unknown shard mode");
            if (sh->world > 64 || sh->rank >= sh->world) throw Error(FR_ERR_INVALID_ARG, "shard rank/world out of range (world <= 64)");
            if (sh->mode == FR_SHARD_PARTIALS && (sh->world & (sh->world - 1)) != 0)
                throw Error(FR_ERR_INVALID_ARG, "partial-block sharding needs a power-of-two world size");
            if (sh->flags & ~(FR_SHARD_GATHER | FR_SHARD_SERIAL_EXCHANGE)) throw Error(FR_ERR_INVALID_ARG, "unknown shard flags");
            spec.rank = sh->rank;
            spec.world = sh->world;
            spec.mode = sh->mode;
            flags = sh->flags;
            if (sh->comm) {
                if (!sh->comm->sendrecv) throw Error(FR_ERR_INVALID_ARG, "fr_comm without a sendrecv function");
                comm = *sh->comm;
                has_comm = true;
            }
            if (sh->rccl_id) transport = make_rccl_transport(sh->rccl_id, sh->rank, sh->world);   // collective
        }
        // Partial-block sharding needs every rank to arrive at the SAME plan (the same list of split voices) on the same
        // call; a kernel that finishes compiling at different moments on different ranks would break that, so compile in
        // the call from here on.
        // (and back to the configured behaviour when the renderer leaves that mode)
        r->jit_cache.set_async(spec.mode == FR_SHARD_PARTIALS ? false : r->jit_async_configured);
        r->shard = spec;
        r->shard_flags = flags;
        r->rccl = std::move(transport);
        r->host_comm = comm;
        r->has_host_comm = has_comm;
        ++r->shard_epoch;                    // the plan depends on all of it
    });
}

fr_status fr_shard_rows(const fr_renderer *r, uint32_t n_slots, uint32_t *lo, uint32_t *hi) {
    if (!r || !lo || !hi) return FR_ERR_INVALID_ARG;
    r->my_rows(n_slots, *lo, *hi);
    return FR_OK;
}

const char *fr_last_error(const fr_renderer *r) { return r ? r->last_error.c_str() : "null renderer"; }

const char *fr_status_string(fr_status s) {
    switch (s) {
    case FR_OK: return "ok";
    case FR_ERR_INVALID_ARG: return "invalid argument";
    case FR_ERR_INPUT_TOO_LONG: return "input row extends past the rendered range";
    case FR_ERR_INPUT_HISTORY: return "input row does not continue the slot's stored history";
    case FR_ERR_NO_SUCH_NODE: return "no such node";
    case FR_ERR_BAD_SLOT: return "primitive read through a non-zero output slot";
    case FR_ERR_CYCLE: return "dependency cycle";
    case FR_ERR_DEVICE: return "device error";
    case FR_ERR_NO_DEVICE: return "no usable gfx950 device";
    case FR_ERR_OUT_OF_MEMORY: return "out of memory";
    case FR_ERR_UNSUPPORTED: return "unsupported";
    case FR_ERR_COMM: return "communication error";
    default: return "unknown status";
    }
}

const char *fr_backend_name(void) { return "hip-gfx950"; }
uint32_t fr_abi_version(void) { return FR_ABI_VERSION; }

const char *fr_plan_json(fr_renderer *r) {
    if (!r) return "{}";
    r->plan_json_cache = r->plan.valid ? r->plan.json : "{}";
    if (r->plan.valid && r->plan_json_cache.size() > 1) {   // live counters of the exchange step (partial-block sharding)
        r->plan_json_cache.pop_back();
        r->plan_json_cache += ",\"exchange_stats\":{\"calls\":" + std::to_string(r->exchange_calls) + ",\"tiles\":" + std::to_string(r->exchange_tiles) +
                              ",\"bytes_sent\":" + std::to_string(r->exchange_bytes) + "}}";
    }
    return r->plan_json_cache.c_str();
}

fr_status fr_set_timing(fr_renderer *r, int32_t enabled) {
    if (!r) return FR_ERR_INVALID_ARG;
    r->timing = enabled != 0;
    return FR_OK;
}

fr_status fr_get_timing(fr_renderer *r, const char *kernel_class, double *ms, uint64_t *launches) {
    return guarded(r, [&] {
        if (!kernel_class) throw Error(FR_ERR_INVALID_ARG, "null kernel class");
        HIP_CHECK(hipSetDevice(r->device));
        r->resolve(r->t_bank);
        r->resolve(r->t_pull);
        r->resolve(r->t_stage);
        std::string k(kernel_class);
        double m = 0;
        uint64_t n = 0;
        if (k == "bank" || k == "all") { m += r->t_bank.ms; n += r->t_bank.launches; }
        if (k == "pull" || k == "all") { m += r->t_pull.ms; n += r->t_pull.launches; }
        if (k == "stage" || k == "all") { m += r->t_stage.ms; n += r->t_stage.launches; }
        if (k != "bank" && k != "pull" && k != "all" && k != "stage") throw Error(FR_ERR_INVALID_ARG, "unknown kernel class " + k);
        if (ms) *ms = m;
        if (launches) *launches = n;
    });
}

fr_status fr_reset_timing(fr_renderer *r) {
    return guarded(r, [&] {
        HIP_CHECK(hipSetDevice(r->device));
        r->resolve(r->t_bank);
        r->resolve(r->t_pull);
        r->resolve(r->t_stage);
        r->t_bank.ms = r->t_pull.ms = r->t_stage.ms = 0;
        r->t_bank.launches = r->t_pull.launches = r->t_stage.launches = 0;
    });
}

}  // extern "C"
