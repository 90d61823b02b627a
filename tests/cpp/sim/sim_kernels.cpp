// sim_kernels.cpp -- TEST INFRASTRUCTURE ONLY.  What each gfx950 kernel of csrc/kernels.hip computes, restated as plain
// host loops behind the same launch interface (csrc/kernels.hpp), for the host-logic simulator (see sim_hip.cpp).
// The loops follow the kernels' DEFINITIONS (every add is the tree's own add, leaves in the graph's product form), not
// their mappings; the kernels themselves are checked against the oracle on the GPU (`pytest -m gpu`).
#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../../libfriendship_amd/csrc/comm.hpp"
#include "../../../libfriendship_amd/csrc/graph.hpp"
#include "../../../libfriendship_amd/csrc/jit.hpp"
#include "../../../libfriendship_amd/csrc/kernels.hpp"

extern "C" {
// launch counters by kernel class, readable from tests (bank, gbank, stage, pull, pad, combine)
std::atomic<uint64_t> fr_sim_launches[6];
uint64_t fr_sim_launch_count(int cls) { return cls >= 0 && cls < 6 ? fr_sim_launches[cls].load() : 0; }
void fr_sim_reset_launch_counts(void) { for (auto &c : fr_sim_launches) c = 0; }
}

namespace fr {
namespace {

enum { C_BANK = 0, C_GBANK = 1, C_STAGE = 2, C_PULL = 3, C_PAD = 4, C_COMBINE = 5 };

float prim_mod(float a, float b) {
    float rem = std::fmod(a, b);
    return rem < 0.0f ? rem + b : rem;
}
float prim_min(float a, float b, bool sparkle) { return (sparkle && a != a) ? a : ((a < b || b != b) ? a : b); }
float prim_binop(uint32_t op, float a, float b, bool sparkle) {
    switch (op) {
    case OP_SUM2: return a + b;
    case OP_MUL: return a * b;
    case OP_DIV: return a / b;
    case OP_MOD: return prim_mod(a, b);
    default: return prim_min(a, b, sparkle);
    }
}
bool delay_frames(float d, uint64_t &frames, bool sparkle) {
    if (d >= 18446744073709551616.0f) return false;
    if (sparkle && !(d >= 0.0f)) return false;
    frames = (d < 0.0f || d != d) ? 0ull : (uint64_t)d;
    return true;
}
float read_input(const DevInput &s, uint64_t t) { return (t < s.base || t >= s.len) ? 0.0f : s.data[t - s.base]; }

// the partial in the graph's own arithmetic (kernels.hip bank_leaf<false, true>)
float leaf(float t, float w, float A4) {
    float x = t * w;
    float r = x - std::trunc(x);
    r = r < 0.0f ? r + 1.0f : r;
    float u = r - 0.5f;
    float q = 0.5f - std::fabs(u);
    float z = u * q;
    return (4.0f * A4) * z;
}
float tree(std::vector<float> &cur) {
    size_t n = cur.size();
    while (n > 1) {
        for (size_t i = 0; i < n / 2; ++i) cur[i] = cur[2 * i] + cur[2 * i + 1];
        n /= 2;
    }
    return cur[0];
}
float bank_time(const BankArgs &a, uint64_t ti) { return (ti >= a.time_skip && ti - a.time_skip < a.time_valid) ? a.time[ti - a.time_skip] : 0.0f; }
uint64_t out_index(const BankArgs &a, uint64_t ti) { return a.ring_mask ? ((a.ring_t0 + ti) & a.ring_mask) : ti; }

float eval_node(const PullArgs &a, uint32_t id, uint64_t t) {
    const DevNode &n = a.nodes[id];
    switch (n.op) {
    case OP_CONST: { float f; std::memcpy(&f, &n.a, 4); return f; }
    case OP_INPUT: return n.a < a.n_inputs ? read_input(a.inputs[n.a], t) : 0.0f;
    case OP_DELAY: {
        uint64_t fr_;
        if (!delay_frames(eval_node(a, n.b, t), fr_, a.sparkle != 0) || fr_ > t) return 0.0f;
        return eval_node(a, n.a, t - fr_);
    }
    default: { float x = eval_node(a, n.a, t); return prim_binop(n.op, x, eval_node(a, n.b, t), a.sparkle != 0); }
    }
}

}  // namespace

hipError_t launch_pull(const PullArgs &a, hipStream_t) {
    ++fr_sim_launches[C_PULL];
    for (uint64_t e = 0; e < a.count; ++e) {
        const uint64_t lin = a.first + e;
        const uint32_t slot = (uint32_t)(lin / a.n_times);
        a.out[lin] = eval_node(a, a.outputs[slot], a.idx + (lin - (uint64_t)slot * a.n_times));
    }
    return hipSuccess;
}

void bank_shape(uint32_t log2_p, uint32_t, uint64_t, uint32_t &chunk_log2, uint32_t &frames_per_lane, uint32_t &waves_per_group,
                uint32_t &small_call, uint32_t &voices_per_wave, bool) {
    chunk_log2 = log2_p;   // (one chunk: the simulator never needs the combine workspace)
    frames_per_lane = 1;
    waves_per_group = 4;
    small_call = 0;
    voices_per_wave = 0;
}
uint64_t bank_blocks(const BankArgs &a) { return ((a.n_times + 63) / 64) * a.n_voices; }
bool bank_publishes_rows(const BankArgs &a) { return a.host_flags && !a.small_call && !a.voices_per_wave && a.leaf_variant == 1 && a.chunk_log2 == a.log2_p; }

hipError_t launch_bank_stream(const BankArgs &, BankStreamCtl *, BankStreamDev *, uint32_t, hipStream_t) { return hipErrorNotSupported; }   // (no resident launches on the simulator)
hipError_t launch_bank(const BankArgs &a, hipStream_t) {
    ++fr_sim_launches[C_BANK];
    const size_t P = (size_t)1 << a.log2_p;
    std::vector<float> cur(P);
    if (a.hist_dst) {
        if (a.time_skip != 0) return hipErrorInvalidValue;
        for (uint64_t ti = 0; ti < a.time_valid && ti < a.n_times; ++ti) a.hist_dst[ti] = a.time[ti];
    }
    for (uint32_t v = 0; v < a.n_voices; ++v)
        for (uint64_t ti = 0; ti < a.n_times; ++ti) {
            const float t = bank_time(a, ti);
            for (size_t k = 0; k < P; ++k) cur[k] = leaf(t, a.params[v * P + k].x, a.params[v * P + k].y);
            a.out[(size_t)a.rows[v] * a.out_stride + out_index(a, ti)] = tree(cur);
        }
    if (a.host_flags)   // row-completion flags of the host entry point's streamed output
        for (uint32_t v = 0; v < a.n_voices; ++v) __atomic_store_n(a.host_flags + a.rows[v], a.flag_value, __ATOMIC_RELEASE);
    return hipSuccess;
}

hipError_t launch_gbank(const BankArgs &a, hipStream_t) {
    ++fr_sim_launches[C_GBANK];
    if (!a.groups || !a.group_off) return hipErrorInvalidValue;
    for (uint32_t v = 0; v < a.n_voices; ++v)
        for (uint64_t ti = 0; ti < a.n_times; ++ti) {
            const float t = bank_time(a, ti);
            std::vector<float> st;
            size_t pair0 = (size_t)a.group_off[2 * v + 1] * 8;
            for (uint32_t g = a.group_off[2 * v]; g < a.group_off[2 * v + 2]; ++g) {
                const uint32_t j = a.groups[g] & 15u;
                uint32_t m = a.groups[g] >> 4;
                std::vector<float> cur((size_t)1 << j);
                for (size_t k = 0; k < cur.size(); ++k) cur[k] = leaf(t, a.params[pair0 + k].x, a.params[pair0 + k].y);
                pair0 += cur.size() < 8 ? 8 : cur.size();
                float val = tree(cur);
                for (; m; --m) { val = st.back() + val; st.pop_back(); }
                st.push_back(val);
            }
            a.out[(size_t)a.rows[v] * a.out_stride + out_index(a, ti)] = st.empty() ? 0.0f : st[0];
        }
    return hipSuccess;
}

hipError_t launch_stage(const StageArgs &a, hipStream_t) {
    ++fr_sim_launches[C_STAGE];
    auto input = [&](uint32_t slot, uint64_t t) {
        if (slot >= a.n_inputs) return 0.0f;
        return read_input(a.n_inputs <= STAGE_INLINE_INPUTS ? a.inline_inputs[slot] : a.inputs[slot], t);
    };
    auto ring = [&](uint32_t buf, uint64_t t) -> float & { return a.rings[(size_t)buf * (a.ring_mask + 1) + (t & a.ring_mask)]; };
    // Threads of a launch run in no particular order: the simulator takes the order LEAST favourable to a plan that silently
    // relies on one -- programs last to first, threads last to first; only the frames of one strided thread are in order
    // (kernels.hip stage_kernel) -- so a plan that needs the strided form, or an order between programs, fails here too.
    const uint64_t span = a.stride ? a.stride : a.w_len;
    for (uint32_t pi = a.n_progs; pi-- > 0;) {
        const StageProg &pg = a.progs[pi];
        for (uint64_t wi0 = std::min<uint64_t>(span, a.w_len); wi0-- > 0;)
        for (uint64_t wi = wi0; wi < a.w_len; wi += span) {
            const uint64_t t = a.w0 + wi;
            float tmp[STAGE_REGS] = {0};
            for (uint32_t i = 0; i < pg.n_instr; ++i) {
                const StageInstr &in = a.instrs[pg.first_instr + i];
                float v = 0.0f;
                switch (in.op) {
                case S_CONST: std::memcpy(&v, &in.imm, 4); break;
                case S_INPUT: v = input(in.imm, t); break;
                case S_READ: v = t >= in.d_lo ? ring(in.buf, t - in.d_lo) : 0.0f; break;
                case S_READ_INPUT: v = t >= in.d_lo ? input(in.imm, t - in.d_lo) : 0.0f; break;
                case S_STEP: if (t >= in.d_lo) std::memcpy(&v, &in.imm, 4); break;
                case S_STORE: ring(in.buf, t) = tmp[in.a]; continue;
                case S_READ_DYN: case S_READ_INPUT_DYN: case S_STEP_DYN: {
                    uint64_t fr_;
                    if (delay_frames(tmp[in.a], fr_, a.sparkle != 0) && t >= fr_) {
                        if (in.op == S_READ_DYN) v = ring(in.buf, t - fr_);
                        else if (in.op == S_READ_INPUT_DYN) v = input(in.imm, t - fr_);
                        else std::memcpy(&v, &in.imm, 4);
                    }
                    break;
                }
                case S_SUM2: v = tmp[in.a] + tmp[in.b]; break;
                case S_MUL: v = tmp[in.a] * tmp[in.b]; break;
                case S_DIV: v = tmp[in.a] / tmp[in.b]; break;
                case S_MOD: v = prim_mod(tmp[in.a], tmp[in.b]); break;
                default: v = prim_min(tmp[in.a], tmp[in.b], a.sparkle != 0); break;
                }
                tmp[in.dst] = v;
            }
            const float r = tmp[pg.result_reg];
            if (pg.dst_ring != 0xFFFFFFFFu) ring(pg.dst_ring, t) = r;
            if (pg.out_row >= 0 && t >= a.idx) a.out[(size_t)pg.out_row * a.n_times + (t - a.idx)] = r;
        }
    }
    return hipSuccess;
}

hipError_t launch_chunk_combine(const ChunkCombineArgs &a, hipStream_t) {
    std::vector<float> v((size_t)1 << a.log2_c);
    for (uint32_t voice = 0; voice < a.n_voices; ++voice)
        for (uint64_t t = 0; t < a.n_times; ++t) {
            for (size_t i = 0; i < v.size(); ++i) v[i] = a.ws[(((size_t)voice << a.log2_c) | i) * a.n_times + t];
            for (size_t n = v.size(); n > 1; n >>= 1)
                for (size_t i = 0; i < n / 2; ++i) v[i] = v[2 * i] + v[2 * i + 1];
            a.out[(size_t)a.rows[voice] * a.out_stride + t] = v[0];
        }
    return hipSuccess;
}

hipError_t launch_pad(float *dst, uint64_t n, const float *src_last, hipStream_t) {
    ++fr_sim_launches[C_PAD];
    const float v = src_last ? *src_last : 0.0f;
    for (uint64_t i = 0; i < n; ++i) dst[i] = v;
    return hipSuccess;
}

hipError_t launch_shard_combine(const ShardCombineArgs &a, hipStream_t) {
    ++fr_sim_launches[C_COMBINE];
    for (uint32_t row = 0; row < a.n_rows; ++row)
        for (uint64_t t = 0; t < a.len; ++t) {
            const size_t e = (size_t)row * a.len + t;
            const float v = a.lo[e] + a.hi[e];
            if (a.dst_ws) { a.dst_ws[e] = v; continue; }
            const uint32_t d = a.dst[row];
            if (d & 0x80000000u) a.rings[(size_t)(d & 0x7FFFFFFFu) * (a.ring_mask + 1) + ((a.ring_t0 + t) & a.ring_mask)] = v;
            else if (t >= a.out_skip) a.out[(size_t)d * a.out_stride + (t - a.out_skip)] = v;
        }
    return hipSuccess;
}

// ---- no hipRTC in the simulator: plans fall back to the hand-written kernels' forms / interpreted programs --------------
JitKernel::~JitKernel() {}
struct JitCache::Impl {};
JitCache::JitCache() : impl_(nullptr) {}
JitCache::~JitCache() {}
uint64_t JitCache::epoch() const { return 0; }
size_t JitCache::compiled() const { return 0; }
double JitCache::compile_ms() const { return 0; }
size_t JitCache::disk_hits() const { return 0; }
std::shared_ptr<JitKernel> JitCache::get(const LeafShape &, const std::vector<bool> &, const std::vector<uint32_t> &, const std::vector<uint32_t> &) {
    throw Error(FR_ERR_UNSUPPORTED, "jit: not available in the host-logic simulator");
}
std::shared_ptr<JitKernel> JitCache::get_source(const std::string &, const char *) {
    throw Error(FR_ERR_UNSUPPORTED, "jit: not available in the host-logic simulator");
}
hipError_t launch_jit_bank(const JitKernel &, const JitBankArgs &, hipStream_t) { return hipErrorInvalidValue; }
hipError_t launch_jit_stage(const JitKernel &, const JitStageArgs &, uint32_t, hipStream_t) { return hipErrorInvalidValue; }

// ---- no RCCL either: the exchange goes through the host callback (fr_comm) ------------------------------------------------
void rccl_unique_id(uint8_t *) { throw Error(FR_ERR_COMM, "the simulator has no RCCL transport; pass an fr_comm callback"); }
std::unique_ptr<Transport> make_rccl_transport(const uint8_t *, uint32_t, uint32_t) {
    throw Error(FR_ERR_COMM, "the simulator has no RCCL transport; pass an fr_comm callback");
}

}  // namespace fr
