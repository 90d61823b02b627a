// bank_bench.hip -- A/B harness for the fused bank kernel: all variants in ONE process on ONE device,
// interleaved rounds, median/min reported (cdna_hip_programming.md 5.4 rule 24), plus the shader clock the
// chip actually holds under this kernel (s_memtime vs s_memrealtime) so that cycles/instruction can be
// computed instead of guessed.  Not part of the product; includes the kernels' translation unit directly.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
//              -mllvm -simplifycfg-sink-common=false -o tools/bank_bench tools/bank_bench.hip
#include "../libfriendship_amd/csrc/kernels.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// Shader clock under a pure-VALU load: every wave runs `iters` x 8 independent v_fma_f32 and stamps
// s_memtime (shader cycles) and s_memrealtime (100 MHz) around the loop.
__global__ void __launch_bounds__(256) clock_kernel(unsigned long long *stamps, float *sink, int iters, float a, float b, int mode) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {
        for (int i = 0; i < iters; ++i)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (mode == 1) {
        for (int i = 0; i < iters; ++i)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (mode == 2) {
        for (int i = 0; i < iters; ++i)
            asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n"
                         "v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else {
        for (int i = 0; i < iters; ++i)
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv) {
    int V = 64, log2p = 12, T = 4800, rounds = 15;
    if (argc > 1) V = std::atoi(argv[1]);
    if (argc > 2) log2p = std::atoi(argv[2]);
    if (argc > 3) T = std::atoi(argv[3]);
    if (argc > 4) rounds = std::atoi(argv[4]);
    const int P = 1 << log2p;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("device: %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);

    // ---- clock + raw VALU rate ----
    {
        int blocks = 256 * 8, iters = 100000;
        unsigned long long *d_st; float *d_sink;
        CK(hipMalloc(&d_st, (size_t)blocks * 4 * 2 * sizeof(unsigned long long)));
        CK(hipMalloc(&d_sink, (size_t)blocks * 256 * sizeof(float)));
        const char *names[4] = {"v_fma_f32 v,v,v", "v_mul/v_add VOP2 v,v", "v_fract_f32", "v_mul_f32 v,s,v"};
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(clock_kernel, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, iters / 10, 0.999f, 1e-3f, mode);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(clock_kernel, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, iters, 0.999f, 1e-3f, mode);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> st((size_t)blocks * 8);
            CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> clk, cpi;
            for (size_t w = 0; w < (size_t)blocks * 4; ++w) {
                clk.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 100e6);
                cpi.push_back((double)st[2 * w] / ((double)iters * 8));
            }
            double rate = (double)blocks * 256 * iters * 8 / (ms * 1e-3);
            std::printf("clock[%-22s]: %.3f GHz median shader clock; %.2f cycles/wave-instr as seen by one of 8 waves on its SIMD "
                        "=> %.2f cyc/instr/SIMD; chip %.2f T lane-ops/s (%.1f lanes/clk/SIMD at that clock)\n",
                        names[mode], median(clk) / 1e9, median(cpi), median(cpi) / 8.0, rate / 1e12, rate / 1024.0 / median(clk));
        }
        CK(hipFree(d_st)); CK(hipFree(d_sink));
    }

    // ---- bank kernel variants ----
    std::vector<float> params((size_t)V * P * 2);
    for (int v = 0; v < V; ++v)
        for (int k = 0; k < P; ++k) {
            float f0 = 55.0f * std::pow(2.0f, (v % 64) / 12.0f) * (1.0f + 1e-4f * (float)(v / 64));   // audible fundamentals for any V
            params[((size_t)v * P + k) * 2] = f0 * (k + 1) / 48000.0f;
            params[((size_t)v * P + k) * 2 + 1] = -4.0f / (k + 1);
        }
    std::vector<float> time(T);
    for (int i = 0; i < T; ++i) time[i] = (float)(i + 48000);
    std::vector<uint32_t> rows(V);
    for (int v = 0; v < V; ++v) rows[v] = v;
    float *d_params, *d_time, *d_out, *d_ws; uint32_t *d_rows;
    CK(hipMalloc(&d_params, params.size() * 4)); CK(hipMalloc(&d_time, T * 4)); CK(hipMalloc(&d_out, (size_t)V * T * 4));
    CK(hipMalloc(&d_ws, (size_t)V * T * 4 * 64)); CK(hipMalloc(&d_rows, V * 4));
    CK(hipMemcpy(d_params, params.data(), params.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_time, time.data(), T * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rows, rows.data(), V * 4, hipMemcpyHostToDevice));

    struct Var { int f, leaf, chunk; const char *name; int nw = 4; int small = 0; int vpw = 0; };
    std::vector<Var> vars;
    int full = std::min(log2p, 13);
    for (int f : {1, 2, 4})
        for (int leaf : {0, 1}) vars.push_back({f, leaf, full, ""});
    vars.push_back({2, 2, full, ""});
    if (T <= 32 && log2p >= 8) vars.push_back({1, 1, 8, "small", 4, 1});
    if (log2p <= 8) for (int vpw : {2, 4, 8, 16}) for (int f : {1, 2}) vars.push_back({f, 1, log2p, "multi", 4, 0, vpw});
    vars.push_back({1, 1, full, "", 8});
    vars.push_back({1, 0, full, "", 8});
    vars.push_back({2, 1, full, "", 8});
    if (log2p <= 12) { vars.push_back({1, 1, full, "", 2}); vars.push_back({2, 1, full, "", 2}); }
    if (log2p <= 11) { vars.push_back({1, 1, full, "", 1}); vars.push_back({2, 1, full, "", 1}); }
    for (int c = full - 1; c >= 9 && c >= full - 3; --c) { vars.push_back({2, 1, c, ""}); vars.push_back({1, 1, c, ""}); }
    std::vector<std::vector<double>> times(vars.size());
    std::vector<std::vector<float>> outs(vars.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds + 1; ++r) {
        for (size_t i = 0; i < vars.size(); ++i) {
            fr::BankArgs a{};
            a.params = (const float2 *)d_params; a.time = d_time; a.time_valid = T; a.out = d_out; a.rows = d_rows;
            a.n_voices = V; a.log2_p = log2p; a.n_times = T; a.fast_ok = 1;
            a.chunk_log2 = vars[i].chunk; a.frames_per_lane = vars[i].f; a.waves_per_group = vars[i].nw; a.out_stride = T; a.small_call = vars[i].small; if (vars[i].small) a.chunk_log2 = 8; a.leaf_variant = vars[i].leaf; a.ws = d_ws; a.voices_per_wave = vars[i].vpw;
            CK(hipEventRecord(e0));
            if (fr::launch_bank(a, 0) != hipSuccess) { (void)hipGetLastError(); if (r == 0) outs[i].assign((size_t)V * T, -1.0f); if (r > 0) times[i].push_back(1e9); continue; }   // shape not supported by this variant
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) times[i].push_back(ms);
            if (r == 0) { outs[i].resize((size_t)V * T); CK(hipMemcpy(outs[i].data(), d_out, outs[i].size() * 4, hipMemcpyDeviceToHost)); }
        }
    }
    double pf = (double)V * P * T;
    std::printf("bank kernel, V=%d P=%d T=%d, %d interleaved rounds (ms: median / min; Msamples/s at median; same bits as variant 0?)\n", V, P, T, rounds);
    for (size_t i = 0; i < vars.size(); ++i) {
        double med = median(times[i]), mn = *std::min_element(times[i].begin(), times[i].end());
        bool same = std::memcmp(outs[i].data(), outs[0].data(), outs[0].size() * 4) == 0;
        if (vars[i].vpw) std::printf("  whole voices per wave, %2d in a row ", vars[i].vpw);
        std::printf("  %sNW=%d F=%d leaf=%d chunk=2^%d : %.4f / %.4f ms   %.2f Msamples/s   %.2f Tpf/s   same=%d\n", vars[i].small ? "lanes-over-partials " : "", vars[i].nw, vars[i].f, vars[i].leaf,
                    vars[i].chunk, med, mn, T / (med * 1e-3) / 1e6, pf / (med * 1e-3) / 1e12, (int)same);
    }
    return 0;
}
