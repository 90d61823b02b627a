// jit.cpp -- code generation + hipRTC compilation of shape-specialised bank kernels (see jit.hpp).
#include "jit.hpp"

#include <hip/hiprtc.h>

#include <hip/hip_version.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <cstdio>
#include <functional>
#include <map>
#include <sstream>

#include "graph.hpp"

namespace fr {

JitKernel::~JitKernel() {
    if (module) (void)hipModuleUnload(module);
}

// The skeleton is the hand-written bank kernel's structure (kernels.hip) with one 64-frame tile per wave, 4 waves
// per workgroup and one workgroup per (voice, tile); only leaf() and K differ between specialisations.
static const char *kSkeleton = R"JIT(
typedef float __attribute__((address_space(4))) const *cptr;

LEAF_FUNCTION

// one wave's share of a voice: Pw leaves in groups of 8, group sums merged by the binary-counter carry chain
template <bool FAST>
__device__ __forceinline__ float wave_sum(cptr p, const float *x TRACK_PARAMS, unsigned ngroups, unsigned levels) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0, s8 = 0;
#if NT > 0
    // per-leaf tracks: the group's 8 * NT row values are requested one group AHEAD (the slot numbers come through the scalar
    // cache, the values through one coalesced 256-byte load per wave and row), so a wave always has a group's loads in flight
    // under the arithmetic of the one before -- the kernel is HBM-bound (8 bytes per partial-frame)
    float cn[8 * K], tvn[8 * NT];
#pragma unroll
    for (int i = 0; i < 8 * K; ++i) cn[i] = p[i];
    TRACK_LOADS(tvn, cn)
#endif
    for (unsigned g = 0; g < ngroups; ++g) {
        float c[8 * K];
#if NT > 0
        float tv[8 * NT];
#pragma unroll
        for (int i = 0; i < 8 * K; ++i) c[i] = cn[i];
#pragma unroll
        for (int i = 0; i < 8 * NT; ++i) tv[i] = tvn[i];
        if (g + 1 < ngroups) {
#pragma unroll
            for (int i = 0; i < 8 * K; ++i) cn[i] = p[(size_t)(g + 1) * (8 * K) + i];
            TRACK_LOADS(tvn, cn)
        }
#else
#pragma unroll
        for (int i = 0; i < 8 * K; ++i) c[i] = p[(size_t)g * (8 * K) + i];
#endif
        float l0 = LEAF_CALL(0), l1 = LEAF_CALL(1), l2 = LEAF_CALL(2), l3 = LEAF_CALL(3);
        float l4 = LEAF_CALL(4), l5 = LEAF_CALL(5), l6 = LEAF_CALL(6), l7 = LEAF_CALL(7);
        float v = ((l0 + l1) + (l2 + l3)) + ((l4 + l5) + (l6 + l7));
        // binary-counter carry chain over the tree levels above the 8-leaf group (g < 2^levels stops it)
        do {
            if (!(g & 1u)) { s0 = v; break; } v = s0 + v;
            if (!(g & 2u)) { s1 = v; break; } v = s1 + v;
            if (!(g & 4u)) { s2 = v; break; } v = s2 + v;
            if (!(g & 8u)) { s3 = v; break; } v = s3 + v;
            if (!(g & 16u)) { s4 = v; break; } v = s4 + v;
            if (!(g & 32u)) { s5 = v; break; } v = s5 + v;
            if (!(g & 64u)) { s6 = v; break; } v = s6 + v;
            if (!(g & 128u)) { s7 = v; break; } v = s7 + v;
            s8 = v;
        } while (0);
    }
    float r = s8;
    r = levels == 7u ? s7 : r; r = levels == 6u ? s6 : r; r = levels == 5u ? s5 : r; r = levels == 4u ? s4 : r;
    r = levels == 3u ? s3 : r; r = levels == 2u ? s2 : r; r = levels == 1u ? s1 : r; r = levels == 0u ? s0 : r;
    return r;
}

extern "C" __global__ void __launch_bounds__(256) jit_bank(JitBankArgs a) {
    unsigned b = blockIdx.x;
    unsigned lid = (a.nblocks % 8u == 0u) ? (b % 8u) * (a.nblocks / 8u) + b / 8u : b;   // XCD-contiguous work ranges
    const unsigned voice = lid / a.tiles, tile = lid - voice * a.tiles;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long ti = (unsigned long long)tile * 64u + lane;
    const unsigned long long tt = ti < a.n_times ? ti : a.n_times - 1;   // (tracks: the lanes past the call's end re-read its last frame)
    (void)tt;
    float x[NIN > 0 ? NIN : 1];
#pragma unroll
    for (int i = 0; i < NIN; ++i)
        x[i] = (ti >= a.in_skip[i] && ti - a.in_skip[i] < a.in_valid[i]) ? a.in[i][ti - a.in_skip[i]] : jit_opaque(0.0f);

    const unsigned P = 1u << a.log2_p, Pw = P >> 2, ngroups = Pw >> 3, levels = a.log2_p - 5u;
    cptr p = (cptr)(a.params + ((size_t)voice * P + (size_t)wave * Pw) * K);
    float r;
#if HAS_MOD1
    // the body with Modulo(x, 1.0) as v_fract_f32 is exact when the host proved it for inputs in [+0, 2^32]
    // (a.fract_ok) and this wave's inputs are in that range
    bool in_range = a.fract_ok != 0u;
#pragma unroll
    for (int i = 0; i < NIN; ++i)
        if ((FRACT_INPUTS >> i) & 1u) in_range = in_range && __builtin_bit_cast(unsigned, x[i]) <= 0x4F800000u;
    if (__builtin_amdgcn_ballot_w64(!in_range) == 0ull) r = wave_sum<true>(p, x TRACK_ARGS, ngroups, levels);
    else
#endif
        r = wave_sum<false>(p, x TRACK_ARGS, ngroups, levels);

    __shared__ float sm[4][64];
    sm[wave][lane] = r;
    __syncthreads();
    if (wave == 0 && ti < a.n_times) {
        float t = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
        unsigned long long o = a.ring_mask ? ((a.ring_t0 + ti) & a.ring_mask) : ti;
        a.out[(size_t)a.rows[voice] * a.out_stride + o] = t;
    }
}

// Many small voices: a wave sums WHOLE voices, voices_per_wave in a row, for one tile (see bank_multi_kernel in kernels.hip)
extern "C" __global__ void __launch_bounds__(256) jit_bank_multi(JitBankArgs a) {
    unsigned b = blockIdx.x;
    unsigned lid = (a.nblocks % 8u == 0u) ? (b % 8u) * (a.nblocks / 8u) + b / 8u : b;
    const unsigned vb = lid / a.tiles, tile = lid - vb * a.tiles;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long ti = (unsigned long long)tile * 64u + lane;
    const unsigned long long tt = ti < a.n_times ? ti : a.n_times - 1;   // (tracks: the lanes past the call's end re-read its last frame)
    (void)tt;
    float x[NIN > 0 ? NIN : 1];
#pragma unroll
    for (int i = 0; i < NIN; ++i)
        x[i] = (ti >= a.in_skip[i] && ti - a.in_skip[i] < a.in_valid[i]) ? a.in[i][ti - a.in_skip[i]] : jit_opaque(0.0f);
    bool fast = false;
#if HAS_MOD1
    bool in_range = a.fract_ok != 0u;
#pragma unroll
    for (int i = 0; i < NIN; ++i)
        if ((FRACT_INPUTS >> i) & 1u) in_range = in_range && __builtin_bit_cast(unsigned, x[i]) <= 0x4F800000u;
    fast = __builtin_amdgcn_ballot_w64(!in_range) == 0ull;
#endif
    const unsigned P = 1u << a.log2_p, ngroups = P >> 3, levels = a.log2_p - 3u;
    const unsigned v0 = (vb * 4u + wave) * a.voices_per_wave;
    const unsigned long long o = a.ring_mask ? ((a.ring_t0 + ti) & a.ring_mask) : ti;
    for (unsigned j = 0; j < a.voices_per_wave; ++j) {
        const unsigned voice = v0 + j;
        if (voice >= a.n_voices) break;
        cptr p = (cptr)(a.params + (size_t)voice * P * K);
        const float r = fast ? wave_sum<true>(p, x TRACK_ARGS, ngroups, levels) : wave_sum<false>(p, x TRACK_ARGS, ngroups, levels);
        if (ti < a.n_times) a.out[(size_t)a.rows[voice] * a.out_stride + o] = r;
    }
}
)JIT";

std::string JitCache::generate_source(const LeafShape &shape, const std::vector<bool> &varying, const std::vector<uint32_t> &literal_bits,
                                      const std::vector<uint32_t> &alias, bool sparkle) {
    LeafSource ls = generate_leaf_source(shape, varying, literal_bits, alias, sparkle);
    std::ostringstream call;
    call << "leaf<FAST>(x" << (ls.tracks ? ", trk, tstride, tlimit, tt" : "");
    auto track_index = [&](uint32_t i) { for (size_t q = 0; q < ls.track_params.size(); ++q) if (ls.track_params[q] == i) return (int)q; return -1; };
    for (uint32_t i = 0; i < ls.k; ++i) {
        const int q = track_index(i);
        if (q >= 0) call << ", tv[(j) * NT + " << q << "]";
        else call << ", c[(j) * K + " << i << "]";
    }
    call << ")";
    std::ostringstream loads;   // TRACK_LOADS(dst, src): the 8 leaves' track values of a group whose parameters are in src
    for (int j = 0; j < 8; ++j)
        for (size_t q = 0; q < ls.track_params.size(); ++q)
            loads << " dst[" << j << " * NT + " << q << "] = jit_track(trk, tstride, tlimit, tt, src[" << j << " * K + " << ls.track_params[q] << "]);";
    std::ostringstream src;
    src << "#pragma clang fp contract(off)\n";
    src << FR_STR(FR_JIT_ARGS_TEXT) << "\n";
    src << "#define K " << ls.k << "\n#define NIN " << shape.input_slots.size() << "\n#define HAS_MOD1 " << (ls.has_mod1 ? 1 : 0)
        << "\n#define FRACT_INPUTS " << ls.fract_inputs << "u\n";
    src << "#define LEAF_CALL(j) " << call.str() << "\n";
    src << "#define NT " << ls.track_params.size() << "\n#define TRACK_LOADS(dst, src)" << loads.str() << "\n";
    src << (ls.tracks ? "#define TRACK_PARAMS , const float *trk, unsigned long long tstride, unsigned tlimit, unsigned long long tt\n#define TRACK_ARGS , a.tracks, a.track_stride, a.track_limit, tt\n"
                      : "#define TRACK_PARAMS\n#define TRACK_ARGS\n");
    std::string body = kSkeleton;
    const std::string tag = "LEAF_FUNCTION";
    body.replace(body.find(tag), tag.size(), ls.text);
    src << body;
    return src.str();
}

// One cache entry: the source's code object once hipRTC is through with it (worker thread), then the loaded module
// (made on the calling thread, which has the device current).
struct JitCache::Impl {
    struct Entry {
        enum State { COMPILING, CODE_READY, LOADED, FAILED } state = COMPILING;
        std::string fn_name, error;
        std::vector<char> code;
        std::shared_ptr<JitKernel> kernel;
        double ms = 0;
    };
    std::mutex mu;
    std::map<std::string, std::unique_ptr<Entry>> cache;
    std::atomic<uint64_t> epoch{0};
    size_t compiled = 0;
    size_t disk_hits = 0;   // of them, code objects that came from FR_JIT_CACHE instead of the compiler
    double compile_ms = 0;
    // ONE worker thread for the whole cache, fed through a queue: a burst of N new voice shapes compiles one after the
    // other beside the audio thread, not N hipRTC compilers at once (each takes a core for ~0.1 s).
    std::thread worker;
    std::deque<std::function<void()>> jobs;
    std::condition_variable cv;
    bool stop = false;
    void enqueue(std::function<void()> job) {
        {
            std::lock_guard<std::mutex> g(mu);
            jobs.push_back(std::move(job));
            if (!worker.joinable())
                worker = std::thread([this] {
                    for (;;) {
                        std::function<void()> j;
                        {
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [this] { return stop || !jobs.empty(); });
                            if (stop) return;            // (entries still COMPILING are never looked at again: the cache is going away)
                            j = std::move(jobs.front());
                            jobs.pop_front();
                        }
                        j();
                    }
                });
        }
        cv.notify_one();
    }
};

JitCache::JitCache() : impl_(new Impl) {}
JitCache::~JitCache() {
    {
        std::lock_guard<std::mutex> g(impl_->mu);
        impl_->stop = true;
    }
    impl_->cv.notify_all();
    if (impl_->worker.joinable()) impl_->worker.join();   // (waits for the compile in progress, not for the queue)
    delete impl_;
}
uint64_t JitCache::epoch() const { return impl_->epoch.load(); }
size_t JitCache::compiled() const { std::lock_guard<std::mutex> g(impl_->mu); return impl_->compiled; }
double JitCache::compile_ms() const { std::lock_guard<std::mutex> g(impl_->mu); return impl_->compile_ms; }
size_t JitCache::disk_hits() const { std::lock_guard<std::mutex> g(impl_->mu); return impl_->disk_hits; }

std::shared_ptr<JitKernel> JitCache::get(const LeafShape &shape, const std::vector<bool> &varying, const std::vector<uint32_t> &literal_bits,
                                         const std::vector<uint32_t> &alias) {
    std::shared_ptr<JitKernel> jk = get_source(generate_source(shape, varying, literal_bits, alias, sparkle_), "jit_bank");
    if (jk && !jk->k)
        for (size_t c = 0; c < varying.size(); ++c) jk->k += (varying[c] && alias[c] == c) ? 1 : 0;
    return jk;
}

// ---- code objects kept on disk (FR_JIT_CACHE=<directory>; off when unset) --------------------------------------------
// A host that renders the same patches day after day compiles each distinct kernel once per toolchain, not once per
// process (~0.1 s each).  One file per kernel, named by a hash of everything the code object depends on -- architecture,
// the toolchain's identity (hipRTC version, the HIP version and build this library was compiled against, the runtime
// and driver versions it runs on: results depend on backend code generation down to the sign of a zero),
// this engine's generator tag, options, source text -- and carrying that text itself: a file is used only if its text
// equals the request's (a hash collision or a stale file is a miss, never a wrong kernel) AND the checksum stored with
// the code bytes matches them (a truncated or bit-rotten file is a miss).  The checksum is no defence against someone
// who can write the directory -- they can write matching checksums too: FR_JIT_CACHE must name a directory only its
// owner can write.  Written to a temporary name and renamed, so a reader never sees half a file; any
// I/O failure just means compiling as if there were no cache.
static const char *kJitOptions[] = {"-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                                    "-fno-slp-vectorize", "-mllvm", "-simplifycfg-sink-common=false"};
#ifndef HIP_VERSION_GITHASH
#define HIP_VERSION_GITHASH "?"
#endif
static uint64_t fnv1a(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *c = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}
static std::string disk_key_text(const std::string &src, const std::string &arch) {
    int major = 0, minor = 0, rt = 0, drv = 0;
    (void)hiprtcVersion(&major, &minor);
    if (hipRuntimeGetVersion(&rt) != hipSuccess) { (void)hipGetLastError(); rt = 0; }
    if (hipDriverGetVersion(&drv) != hipSuccess) { (void)hipGetLastError(); drv = 0; }
    // "fr-jit-12": bumped whenever this engine's code generators change what they print for the same request
    std::string key = "fr-jit-12|" + arch + "|hiprtc " + std::to_string(major) + "." + std::to_string(minor) + "|hip " +
                      std::to_string(HIP_VERSION_MAJOR) + "." + std::to_string(HIP_VERSION_MINOR) + "." + std::to_string(HIP_VERSION_PATCH) + " " +
                      HIP_VERSION_GITHASH + "|runtime " + std::to_string(rt) + "|driver " + std::to_string(drv) + "|";
    for (const char *o : kJitOptions) { key += o; key += ' '; }
    key += "|\n";
    key += src;
    return key;
}
static std::string disk_path(const std::string &key_text) {
    const char *dir = std::getenv("FR_JIT_CACHE");
    if (!dir || !dir[0]) return "";
    const uint64_t h = fnv1a(key_text.data(), key_text.size());
    char name[40];
    std::snprintf(name, sizeof name, "/fr_%016llx.jitbin", (unsigned long long)h);
    return std::string(dir) + name;
}
static bool disk_load(const std::string &path, const std::string &key_text, std::vector<char> &code) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    uint64_t hdr[4] = {0, 0, 0, 0};   // magic, key length, code length, FNV-1a of the code bytes
    if (std::fread(hdr, sizeof hdr, 1, f) == 1 && hdr[0] == 0x324E49424A5246ull && hdr[1] == key_text.size() && hdr[2] > 0 && hdr[2] < (1ull << 30)) {
        std::string k(hdr[1], '\0');
        code.resize(hdr[2]);
        ok = std::fread(&k[0], 1, k.size(), f) == k.size() && k == key_text && std::fread(code.data(), 1, code.size(), f) == code.size() &&
             fnv1a(code.data(), code.size()) == hdr[3];
    }
    std::fclose(f);
    if (!ok) code.clear();
    return ok;
}
static void disk_store(const std::string &path, const std::string &key_text, const std::vector<char> &code) {
    const std::string tmp = path + ".tmp" + std::to_string((unsigned long long)std::hash<std::thread::id>{}(std::this_thread::get_id()));
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;
    const uint64_t hdr[4] = {0x324E49424A5246ull, key_text.size(), code.size(), fnv1a(code.data(), code.size())};
    const bool ok = std::fwrite(hdr, sizeof hdr, 1, f) == 1 && std::fwrite(key_text.data(), 1, key_text.size(), f) == key_text.size() &&
                    std::fwrite(code.data(), 1, code.size(), f) == code.size();
    if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
}

// hipRTC only: source text -> code object.  No HIP runtime state is touched, so it may run on any thread.
static void compile_source(const std::string &src, const std::string &arch, std::vector<char> &code) {
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "fr_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        throw Error(FR_ERR_DEVICE, "jit: hiprtcCreateProgram failed");
    std::vector<const char *> opts{arch.c_str()};
    for (const char *o : kJitOptions) opts.push_back(o);
    hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        throw Error(FR_ERR_DEVICE, "jit: compilation failed: " + log.substr(0, 2000));
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
}

std::shared_ptr<JitKernel> JitCache::get_source(const std::string &src, const char *fn_name) {
    std::unique_lock<std::mutex> lock(impl_->mu);
    auto it = impl_->cache.find(src);
    if (it == impl_->cache.end()) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            throw Error(FR_ERR_DEVICE, "jit: cannot query the device");
        const std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
        it = impl_->cache.emplace(src, std::unique_ptr<Impl::Entry>(new Impl::Entry)).first;
        Impl::Entry *e = it->second.get();
        e->fn_name = fn_name;
        if (const char *dir = std::getenv("FR_JIT_DUMP")) {   // debugging aid: keep every generated source
            const std::string path = std::string(dir) + "/" + fn_name + "_" + std::to_string(std::hash<std::string>{}(src)) + ".hip";
            if (FILE *f = std::fopen(path.c_str(), "w")) { std::fwrite(src.data(), 1, src.size(), f); std::fclose(f); }
        }
        Impl *impl = impl_;
        auto job = [impl, e, src, arch] {   // (`src` by value: the map key may outlive nothing else here)
            auto t0 = std::chrono::steady_clock::now();
            std::vector<char> code;
            std::string error;
            bool from_disk = false;
            try {
                const std::string key_text = disk_key_text(src, arch), path = disk_path(key_text);
                from_disk = !path.empty() && disk_load(path, key_text, code);
                if (!from_disk) {
                    compile_source(src, arch, code);
                    if (!path.empty()) disk_store(path, key_text, code);
                }
            } catch (const std::exception &ex) {
                error = ex.what();
            }
            std::lock_guard<std::mutex> g(impl->mu);
            if (from_disk) ++impl->disk_hits;
            e->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            e->code = std::move(code);
            e->error = std::move(error);
            e->state = e->error.empty() ? Impl::Entry::CODE_READY : Impl::Entry::FAILED;
            impl->epoch.fetch_add(1);
        };
        if (async_) {
            lock.unlock();
            impl_->enqueue(job);
            return nullptr;
        }
        lock.unlock();
        job();
        lock.lock();
    }
    Impl::Entry *e = it->second.get();
    if (e->state == Impl::Entry::COMPILING) return nullptr;
    if (e->state == Impl::Entry::FAILED) throw Error(FR_ERR_DEVICE, e->error);
    if (e->state == Impl::Entry::CODE_READY) {   // load on this thread: it has the device current
        auto jk = std::make_shared<JitKernel>();
        const bool ok = hipModuleLoadData(&jk->module, e->code.data()) == hipSuccess &&
                        hipModuleGetFunction(&jk->fn, jk->module, e->fn_name.c_str()) == hipSuccess;
        if (!ok) {
            e->state = Impl::Entry::FAILED;
            e->error = "jit: loading the compiled code object failed";
            throw Error(FR_ERR_DEVICE, e->error);
        }
        if (e->fn_name == "jit_bank" && hipModuleGetFunction(&jk->fn_multi, jk->module, "jit_bank_multi") != hipSuccess) jk->fn_multi = nullptr;
        e->code.clear();
        e->code.shrink_to_fit();
        e->kernel = jk;
        e->state = Impl::Entry::LOADED;
        impl_->compile_ms += e->ms;
        ++impl_->compiled;
    }
    return e->kernel;
}

hipError_t launch_jit_bank(const JitKernel &k, const JitBankArgs &a, hipStream_t s) {
    if (a.nblocks == 0) return hipSuccess;
    JitBankArgs copy = a;
    void *args[] = {&copy};
    return hipModuleLaunchKernel(a.voices_per_wave && k.fn_multi ? k.fn_multi : k.fn, a.nblocks, 1, 1, 256, 1, 1, 0, s, args, nullptr);
}

hipError_t launch_jit_stage(const JitKernel &k, const JitStageArgs &a, uint32_t n_progs, hipStream_t s) {
    if (n_progs == 0 || a.w_len == 0) return hipSuccess;
    JitStageArgs copy = a;
    void *args[] = {&copy};
    const unsigned long long span = a.stride ? std::min(a.stride, a.w_len) : a.w_len;   // (stride: one launch walks the window in strides)
    return hipModuleLaunchKernel(k.fn, (uint32_t)((span + 255) / 256), n_progs, 1, 256, 1, 1, 0, s, args, nullptr);
}

}  // namespace fr
