/*
 * friendship_render.h -- C-ABI boundary of the MI355X-native render engine.
 *
 * This header is the drop-in boundary for libfriendship's render hot path.  Each entry point
 * replaces one method of the reference's `render::Renderer` / `routing::GraphWatcher` traits
 * (all citations relative to the reference checkout):
 *
 *   fr_renderer_create / _destroy   <- `Default::default()` / drop of a renderer
 *                                      (src/render/reference.rs:20, src/dispatch.rs:99-106)
 *   fr_on_add_node                  <- GraphWatcher::on_add_node (src/routing/graphwatcher.rs:5,
 *                                      impl src/render/reference.rs:117-120, make_node :98-113)
 *   fr_on_del_node                  <- GraphWatcher::on_del_node (graphwatcher.rs:6, reference.rs:121-123)
 *   fr_on_add_edge                  <- GraphWatcher::on_add_edge (graphwatcher.rs:7, reference.rs:124-126,141-153)
 *   fr_on_del_edge                  <- GraphWatcher::on_del_edge (graphwatcher.rs:8, reference.rs:127-136)
 *   fr_fill_buffer                  <- Renderer::fill_buffer (src/render/renderer.rs:16,
 *                                      impl src/render/reference.rs:47-85), called from
 *                                      src/dispatch.rs:150 only
 *
 * Plain pointers and sizes only: no C++/torch/Rust types cross this line.  A Rust host binds these
 * with an `extern "C"` block; the C++ host mirror in libfriendship_amd/host/
 * and the Python test harness bind the same symbols.
 *
 * Two shared libraries export this exact symbol set:
 *   libfriendship_amd/libfriendship_hip.so  -- the product: HIP/gfx950 engine.  No CPU fallback.
 *   oracle/_build/libfr_oracle.so           -- TEST INFRASTRUCTURE ONLY: CPU restatement of RefRenderer.
 *
 * Threading: one host thread per handle; calls on one handle are serialised by the caller
 * (the reference's Dispatch is !Send, src/routing/routegraph.rs:27).  fr_fill_buffer is synchronous:
 * when it returns, `out` is complete (src/dispatch.rs:150-151 hands the buffer on immediately).
 *
 * Errors: the reference's trait methods return () and panic on contract violations
 * (reference.rs:69,71,131,145,186,199...).  Here every call returns an fr_status; a conforming shim
 * turns non-zero into panic!.  Nothing in this library aborts the process.
 */
#ifndef FRIENDSHIP_RENDER_H
#define FRIENDSHIP_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_ABI_VERSION 2u

/* ---- status codes ------------------------------------------------------------------------- */
typedef int32_t fr_status;
enum {
    FR_OK = 0,
    FR_ERR_INVALID_ARG = 1,        /* null pointer, bad enum value, malformed effect description   */
    FR_ERR_INPUT_TOO_LONG = 2,     /* input row extends past idx+n_times (reference.rs:71 assert)  */
    FR_ERR_INPUT_HISTORY = 3,      /* row for a slot whose stored length != idx (reference.rs:69)  */
    FR_ERR_NO_SUCH_NODE = 4,       /* edge to/from a node the renderer was never told about
                                      (reference.rs:131,145,186 unwrap/expect/index panics)        */
    FR_ERR_BAD_SLOT = 5,           /* non-constant primitive read through from_slot != 0
                                      (reference.rs:199,223,230,237,244,251 asserts)               */
    FR_ERR_CYCLE = 6,              /* dependency cycle reachable from a rendered output on which no
                                      Delay of a constant >= 1 frames lies: the reference's recursion
                                      (reference.rs:197-216) would never end.  A cycle closed through
                                      such a Delay -- feedback -- IS rendered, as the reference as
                                      written renders it (routegraph.rs:218-237 never refuses the
                                      edge); FR_ERR_UNSUPPORTED where this engine has no evaluator for
                                      it: FR_MODE_PULL, history_frames != 0, FR_SHARD_PARTIALS, a loop
                                      reached by a row left to the pull interpreter, a seek of more
                                      than 2^28 frames (a loop's state is rebuilt by replay from 0)  */
    FR_ERR_DEVICE = 7,             /* HIP runtime failure; see fr_last_error                       */
    FR_ERR_NO_DEVICE = 8,          /* no gfx950 device / HIP code object unusable                  */
    FR_ERR_OUT_OF_MEMORY = 9,
    FR_ERR_UNSUPPORTED = 10,       /* well-formed request the engine cannot evaluate (documented)  */
    FR_ERR_COMM = 11               /* multi-GPU exchange failed                                    */
};

/* ---- graph data --------------------------------------------------------------------------- */

/* routing::Edge (src/routing/routegraph.rs:20-24,38-44): 16 bytes.  Node handle 0 is the null
 * handle = the graph's own inputs (as `from`) / outputs (as `to`), per NullableInt
 * (src/routing/nullable_int.rs:65-72). */
typedef struct fr_edge {
    uint32_t from;
    uint32_t to;
    uint32_t from_slot;
    uint32_t to_slot;
} fr_edge;

/* routing::effect::PrimitiveEffect in declaration order (src/routing/effect.rs:86-112). */
enum {
    FR_PRIM_DELAY = 0,
    FR_PRIM_F32CONSTANT = 1,
    FR_PRIM_SUM2 = 2,
    FR_PRIM_MULTIPLY = 3,
    FR_PRIM_DIVIDE = 4,
    FR_PRIM_MODULO = 5,
    FR_PRIM_MINIMUM = 6,
    FR_EFFECT_GRAPH = 7            /* EffectData::RouteGraph (effect.rs:79-82): a composite effect */
};

/* What on_add_node receives in place of `&Rc<Effect>`: the part of `Effect` the renderer reads
 * (EffectData, effect.rs:79-82).  For kind < 7 the arrays are ignored.  For FR_EFFECT_GRAPH the
 * arrays are the effect's RouteGraph in AdjList shape (src/routing/adjlist.rs:11-15): nodes as
 * (handle, effect) pairs, edges as 16-byte fr_edge.  The renderer copies everything it needs
 * before returning (RefRenderer deep-copies too, reference.rs:98-113); pointers are borrowed for
 * the duration of the call only. */
typedef struct fr_effect {
    int32_t kind;
    uint32_t n_nodes;
    const uint32_t *node_handles;                /* [n_nodes], none may be 0                      */
    const struct fr_effect *const *node_effects; /* [n_nodes]                                     */
    uint32_t n_edges;
    const fr_edge *edges;                        /* [n_edges]                                     */
} fr_effect;

/* ---- renderer ----------------------------------------------------------------------------- */
typedef struct fr_renderer fr_renderer;

/* Execution strategy.  AUTO picks fused kernels where the lowered graph matches a known
 * structure and the materialised evaluator elsewhere; the other values force one strategy and exist
 * for parity tests and A/B measurements. */
enum {
    FR_MODE_AUTO = 0,
    FR_MODE_PULL = 1,              /* per-(slot,t) pull interpreter: the reference's recursion    */
    FR_MODE_STAGED = 2             /* materialised stage pipeline, no fused oscillator banks      */
};

/* Which of the reference's two renderers to reproduce where they disagree (both unpinned by any reference
 * test):
 *   FR_SEMANTICS_REFERENCE  RefRenderer (src/render/reference.rs:197-248): Delay amount < 0 or NaN clamps to 0
 *                           frames; Minimum is Rust's f32::min (NaN loses).  The named oracle; the default.
 *   FR_SEMANTICS_SPARKLE    SparkleRenderer (src/render/sparkle.rs:492-498,531-542): Delay amount < 0 or NaN
 *                           outputs 0.0; Minimum is select(fcmp ult a, b): a NaN operand returns `a`.
 * A host that replaces `SparkleRenderer::default()` (tests/render_prim.rs:30) and wants its corner cases picks
 * the second; everything the reference's tests pin is identical under both. */
enum {
    FR_SEMANTICS_REFERENCE = 0,
    FR_SEMANTICS_SPARKLE = 1
};

/* fr_config.flags.  Kernels specialised at run time (hipRTC, ~0.1 s per distinct shape) are compiled on a worker
 * thread by default: fill_buffer never waits for a compiler, calls are served by the generic evaluators meanwhile and
 * the plan switches over when the kernel is ready (same bits either way).  FR_CONFIG_SYNC_COMPILE compiles inside the
 * call that first needs the kernel instead: deterministic plans for tests, benchmarks and offline rendering. */
#define FR_CONFIG_SYNC_COMPILE 1u

typedef struct fr_config {
    uint32_t abi_version;          /* FR_ABI_VERSION                                              */
    int32_t device;                /* HIP device ordinal; -1 = current device                     */
    int32_t mode;                  /* FR_MODE_*                                                   */
    uint32_t flags;                /* FR_CONFIG_*                                                 */
    int32_t semantics;             /* FR_SEMANTICS_*                                              */
    uint32_t reserved;             /* 0                                                           */
    uint64_t history_frames;       /* Input history kept per slot, in frames; 0 = everything since the last
                                      seek, which is the reference's behaviour (reference.rs:25,70-73).  With a
                                      cap, a call rendering [idx, idx + n) sees input samples older than
                                      idx - history_frames as 0.0 -- what the reference returns for times before
                                      a seek point -- so a later-added Delay reaching further back than the cap
                                      sees silence there.  The cap is raised to what the current plan's constant
                                      delays and proven delay bounds need, so a graph never loses samples it can
                                      be shown to read; memory per fed slot is then 2 * (cap + call length).     */
} fr_config;

/* cfg may be NULL (device -1, FR_MODE_AUTO). */
fr_status fr_renderer_create(const fr_config *cfg, fr_renderer **out);
void fr_renderer_destroy(fr_renderer *r);

fr_status fr_on_add_node(fr_renderer *r, uint32_t handle, const fr_effect *effect);
fr_status fr_on_del_node(fr_renderer *r, uint32_t handle);
fr_status fr_on_add_edge(fr_renderer *r, const fr_edge *edge);
fr_status fr_on_del_edge(fr_renderer *r, const fr_edge *edge);

/* Batch forms of the two calls above (same semantics as calling them in array order; stop at the
 * first failure).  Not part of the reference surface: they exist so a host can hand over a
 * multi-million-node tree without one FFI crossing per element. */
fr_status fr_on_add_nodes(fr_renderer *r, const uint32_t *handles,
                          const fr_effect *const *effects, size_t n);
fr_status fr_on_add_edges(fr_renderer *r, const fr_edge *edges, size_t n);

/* ---- multi-GPU: one process (and one renderer) per GPU --------------------------------------------
 * Output slots are independent in the reference (reference.rs:78-82), so a job shards over `world`
 * renderers with no change to the graph API: EVERY rank receives the SAME graph edits and makes the
 * SAME fill_buffer calls (same n_slots, n_times, idx; same input rows).  Rank r owns the contiguous
 * block of output rows fr_shard_rows() reports and renders exactly those; rows it does not own are left
 * untouched in `out` (the host allocates zeros, src/dispatch.rs:149).
 *   FR_SHARD_VOICES    every rank evaluates only what its own rows need.  No exchange.
 *   FR_SHARD_PARTIALS  additionally, an oscillator bank needed by one rank only is cut at the top
 *                      log2(world) levels of its Sum2 tree: rank r renders sub-tree r of EVERY such bank
 *                      (a partial mix), and one exchange step -- recursive halving, log2(world) pairwise
 *                      sends, each followed by the tree's own f32 add in the tree's own association --
 *                      leaves the owner with the bank's value, bit-identical to the unsharded render.
 *                      Effects behind the bank (envelopes, delay lines) then run on the owner.
 *                      world must be a power of two.  (Ranks must plan identically on the same call, so in this
 *                      mode run-time kernels are compiled inside the call, as with FR_CONFIG_SYNC_COMPILE.)
 * FR_SHARD_GATHER (flag): after rendering, rank 0's `out` also receives every other rank's rows.
 * Transport of the exchange: the engine's own RCCL communicator over xGMI (pass the same
 * fr_comm_unique_id() bytes on every rank), or a host-staged callback (`comm`; MPI, gloo, a test
 * harness): the engine copies the ranges through pinned host memory around it.  Neither is needed in
 * FR_SHARD_VOICES without FR_SHARD_GATHER. */
enum {
    FR_SHARD_NONE = 0,
    FR_SHARD_VOICES = 1,
    FR_SHARD_PARTIALS = 2
};
#define FR_SHARD_GATHER 1u
/* Diagnostics: the exchange of FR_SHARD_PARTIALS as one step after the bank kernels, on the call's stream.  By default the
 * window is cut into time tiles and tile i's exchange runs on a second stream under tile i+1's bank kernels (same bits). */
#define FR_SHARD_SERIAL_EXCHANGE 2u
#define FR_COMM_ID_BYTES 128

typedef struct fr_comm {
    void *ctx;
    /* Blocking pairwise exchange of HOST buffers with rank `peer`: send send_bytes from `send`, receive
     * recv_bytes into `recv` (either may be 0).  The peer makes the matching call.  Returns 0 on success. */
    int32_t (*sendrecv)(void *ctx, uint32_t peer, const void *send, size_t send_bytes, void *recv,
                        size_t recv_bytes);
} fr_comm;

typedef struct fr_shard {
    uint32_t rank;
    uint32_t world;                /* <= 64 */
    int32_t mode;                  /* FR_SHARD_*                                                  */
    uint32_t flags;                /* FR_SHARD_GATHER | FR_SHARD_SERIAL_EXCHANGE                  */
    const uint8_t *rccl_id;        /* FR_COMM_ID_BYTES bytes from fr_comm_unique_id, or NULL      */
    const fr_comm *comm;           /* host-staged transport, or NULL; copied                      */
} fr_shard;

/* ncclGetUniqueId through the engine, so that a host needs no RCCL binding of its own: call on one rank,
 * hand the bytes to the others by any means, pass them to fr_set_shard on every rank. */
fr_status fr_comm_unique_id(uint8_t id[FR_COMM_ID_BYTES]);
/* Diagnostic: one exchange of n_floats through the RCCL transport on `device` (-1 = current) with this rank as its own
 * peer (a communicator of one rank; ncclSend + ncclRecv to self inside one group, on a stream), received data compared with
 * what was sent.  Exercises the binding -- library lookup, symbols, argument order, datatype, stream -- on a machine with a
 * single GPU, where the exchange of a sharded job can never run.  FR_OK, or FR_ERR_COMM / FR_ERR_DEVICE. */
fr_status fr_comm_selftest(int32_t device, uint64_t n_floats);
/* Collective when rccl_id is given (ncclCommInitRank): every rank calls it.  NULL or world <= 1 unshards. */
fr_status fr_set_shard(fr_renderer *r, const fr_shard *shard);
/* The block of output rows [*lo, *hi) this renderer owns when n_slots rows are rendered. */
fr_status fr_shard_rows(const fr_renderer *r, uint32_t n_slots, uint32_t *lo, uint32_t *hi);

/* Renderer::fill_buffer.  `out` is the row-major [n_slots, n_times] f32 buffer (Array2, allocated
 * by the caller, src/dispatch.rs:149); every element is overwritten.  `idx` is the absolute sample
 * index of column 0; idx != previous idx+n_times is a seek (renderer.rs:12-15).  Inputs are the
 * Jagged2<f32> in CSR form: row r (feeds input slot r) = in_data[in_row_offsets[r] ..
 * in_row_offsets[r+1]); in_row_offsets has n_in_rows+1 entries.  n_in_rows == 0 allows NULLs.
 * A call that returns an error -- where the reference panics (reference.rs:69,71) -- leaves the renderer as it was before
 * the call: no row of it is stored, `out` is unspecified, the next call may follow as if this one had not been made.  (The
 * reference has no "afterwards" there; it stores row by row, so a caught panic would find the rows before the offending one
 * stored.  The CPU oracle library reproduces that half-done state; make the call after a refused one a seek when comparing.) */
fr_status fr_fill_buffer(fr_renderer *r, float *out, uint32_t n_slots, uint64_t n_times,
                         uint64_t idx, const float *in_data, const uint64_t *in_row_offsets,
                         uint32_t n_in_rows);

/* Same contract with every buffer already resident in device memory (out, in_data device pointers;
 * in_row_offsets stays a host pointer -- it is control data).  Work is enqueued on `stream`
 * (a hipStream_t; NULL = the default stream) and NOT synchronised: the caller synchronises before
 * reading d_out and keeps d_in_data alive until then.  Calls on one handle are still issued one
 * after another by the host; when consecutive calls use different streams (or a host-buffer call
 * follows) the library itself orders their device work with an event, so the renderer's state --
 * input history, delay lines -- is always that of the previous call.
 * The CPU oracle library returns FR_ERR_UNSUPPORTED. */
fr_status fr_fill_buffer_device(fr_renderer *r, float *d_out, uint32_t n_slots, uint64_t n_times,
                                uint64_t idx, const float *d_in_data,
                                const uint64_t *in_row_offsets, uint32_t n_in_rows, void *stream);

/* Block streaming, for a host that renders short blocks back to back in real time.  Through fr_fill_buffer every block pays
 * for a kernel launch -- ~12 us from enqueue to first wave on this stack, 22 us host to host for a 64-frame block whose
 * arithmetic takes 1.3 us.  fr_stream_begin launches ONE kernel that stays resident; fr_stream_block then hands it a block
 * of 1..64 frames through a doorbell in mapped memory and returns when the rows have arrived in `out`, which must hold
 * [n_slots of fr_stream_begin, n_times] floats, row-major.
 * Served: plans that are one bank of balanced template voices, one voice per output row, at most one workgroup per compute
 * unit of the device (256 on a whole MI355X; FR_ERR_UNSUPPORTED otherwise -- render such graphs with fr_fill_buffer).
 * `row` is the block's input row for slot 0; such plans read no other input and no history.  A row shorter than n_times is
 * padded as fill_buffer pads it (reference.rs:72-73) with the slot's last stored value: the row's own last value, or -- an
 * empty row -- the last value of the previous block when `idx` continues it (idx == previous idx + previous n_times); the
 * first block of a stream, and a block that does not continue the previous one, find nothing stored (as after a seek) and
 * pad with 0.  Results are the bits fr_fill_buffer renders for the same sequence of calls begun with a seek.
 * ANY other call on the renderer (an edit, fr_fill_buffer, fr_stream_end) first retires the resident launch; the frames
 * streamed in between were not stored, so the call after that is a seek (reference.rs:52-58).
 * Bounds: the kernel ends itself when no block has arrived for FR_STREAM_IDLE_MS = 2000 ms of wall clock (the environment
 * variable of that name overrides it at fr_renderer_create; fr_stream_block then returns FR_ERR_DEVICE: begin again), and
 * fr_stream_block gives up with FR_ERR_DEVICE when a block is not answered within 250 ms (the launch is not fully
 * resident: something else holds compute units).
 * The resident launch keeps one compute unit per (voice, chunk) workgroup -- a whole MI355X for 64 voices of 4096
 * partials, 40 of its 256 units for 5 voices of 1024 -- and nothing else runs on THOSE units until the stream is closed
 * or ends itself; other kernels, of this process or any other, run on the units it leaves. */
fr_status fr_stream_begin(fr_renderer *r, uint32_t n_slots);
fr_status fr_stream_block(fr_renderer *r, float *out, uint64_t n_times, uint64_t idx, const float *row, uint64_t row_len);
fr_status fr_stream_end(fr_renderer *r);

/* ---- control-rate tracks ---------------------------------------------------------------------------------------------
 * Renderer::fill_buffer takes its inputs as an Array2 -- one row per input slot, one column per frame rendered
 * (reference.rs:66-74) -- and every row is appended to that slot's history (:68-74) because a Delay may read it later.
 * Per-partial frequency / amplitude envelopes are such rows too: 2 x 4096 x 64 of them at BASELINE config C, 8 bytes per
 * partial-frame, which no history can afford to keep and nothing ever reads back.  fr_set_track_inputs(first_slot) declares
 * the input slots >= first_slot to be TRACKS: their rows must span exactly the call's frames, are NOT stored, and are read
 * in place -- from the caller's device matrix, or from one H2D copy of it -- by the leaves of the voices of that call.  Only
 * leaves of shape-matched voices can read a track, and not under a Delay: anything else that reads one makes
 * fill_buffer return FR_ERR_UNSUPPORTED, as does a voice whose track slots the call did not supply.  Results are those of
 * the reference given the same rows.  UINT32_MAX (the default): no tracks.  This is the one workload of the hot path that
 * is HBM-bound (bench.py `tracks`).
 * The _dense calls take the reference's own input shape, `in` = [n_in_rows][n_times] row-major (host / device memory), and
 * cost O(1) host work per track row where the CSR form costs a length check each. */
fr_status fr_set_track_inputs(fr_renderer *r, uint32_t first_slot);
fr_status fr_fill_buffer_dense(fr_renderer *r, float *out, uint32_t n_slots, uint64_t n_times, uint64_t idx,
                               const float *in, uint32_t n_in_rows);
fr_status fr_fill_buffer_device_dense(fr_renderer *r, float *d_out, uint32_t n_slots, uint64_t n_times, uint64_t idx,
                                      const float *d_in, uint32_t n_in_rows, void *stream);

/* Optional: page-locks [p, p + bytes) and maps it for the device (hipHostRegister).  An `out` buffer of fr_fill_buffer
 * that lies inside a registered range is then written by the kernels themselves -- no device-to-host copy of the
 * samples at all (config C: 166 -> 142 us per call) -- for a host that REUSES its sample buffer between calls (the
 * reference allocates a fresh Array2 per call, src/dispatch.rs:149; such a host has nothing to register).  The range
 * must stay allocated until fr_host_unregister (fr_renderer_destroy unregisters what is left); results are the same bits
 * either way. */
fr_status fr_host_register(fr_renderer *r, void *p, size_t bytes);
fr_status fr_host_unregister(fr_renderer *r, void *p);

/* ---- introspection (host diagnostics, tests, bench) ---------------------------------------- */

/* Text of the most recent failure on this handle ("" if none).  Valid until the next call. */
const char *fr_last_error(const fr_renderer *r);
const char *fr_status_string(fr_status s);

/* "hip-gfx950" for the product library, "cpu-oracle" for the test oracle. */
const char *fr_backend_name(void);
uint32_t fr_abi_version(void);

/* Description of the execution plan built by the last fill_buffer, as a JSON object in a
 * NUL-terminated string owned by the handle (valid until the next call on it): stages, kernels,
 * fused bank shapes, algorithmic bytes.  The oracle returns "{}". */
const char *fr_plan_json(fr_renderer *r);

/* Kernel time (milliseconds, HIP events on the engine's stream) accumulated by fill_buffer calls
 * since the last reset, for the kernel class named (`"bank"`, `"stage"`, `"pull"`, `"all"`);
 * *launches receives the launch count.  Enabled by fr_set_timing(r, 1); off by default. */
fr_status fr_set_timing(fr_renderer *r, int32_t enabled);
fr_status fr_get_timing(fr_renderer *r, const char *kernel_class, double *ms, uint64_t *launches);
fr_status fr_reset_timing(fr_renderer *r);

#ifdef __cplusplus
}
#endif
#endif /* FRIENDSHIP_RENDER_H */
