// fewvoice_diag.hip -- where a few-voice launch spends its time: per-wave timestamps (s_memrealtime) and placement (HW_ID)
// from a DIAGNOSTIC build of the kernels (FR_DIAG_STAMPS; the product has no stamps).  Prints, per variant: when waves
// start and end relative to the first start, how long a wave's compute takes, and when each SIMD of the chip goes idle.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false \
//              -o tools/_build/fewvoice_diag tools/fewvoice_diag.hip
#define FR_DIAG_STAMPS 1
#include "../libfriendship_amd/csrc/kernels.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static double pct(std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[std::min(v.size() - 1, (size_t)(p * (v.size() - 1) + 0.5))]; }

int main(int argc, char **argv) {
    const int V = argc > 1 ? std::atoi(argv[1]) : 8, log2p = argc > 2 ? std::atoi(argv[2]) : 12, T = argc > 3 ? std::atoi(argv[3]) : 4800;
    const int P = 1 << log2p;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    std::vector<float> params((size_t)V * P * 2), time(T);
    for (int v = 0; v < V; ++v)
        for (int k = 0; k < P; ++k) {
            params[((size_t)v * P + k) * 2] = 55.0f * std::pow(2.0f, (v % 64) / 12.0f) * (k + 1) / 48000.0f;
            params[((size_t)v * P + k) * 2 + 1] = -4.0f / (k + 1);
        }
    for (int i = 0; i < T; ++i) time[i] = (float)(i + 4800);
    std::vector<uint32_t> rows(V);
    for (int v = 0; v < V; ++v) rows[v] = v;
    float *d_params, *d_time, *d_out, *d_ws; uint32_t *d_rows, *d_tickets; unsigned long long *d_diag;
    const size_t max_wgs = (size_t)V * ((T + 63) / 64) * 8 + 4096;
    CK(hipMalloc(&d_params, params.size() * 4)); CK(hipMalloc(&d_time, T * 4)); CK(hipMalloc(&d_out, (size_t)V * T * 4));
    CK(hipMalloc(&d_ws, (std::max<size_t>((size_t)V * T * 8, (size_t)cus * 4096) + (size_t)V * ((T + 63) / 64) * 16 * 64) * 4)); CK(hipMalloc(&d_rows, V * 4));
    CK(hipMalloc(&d_tickets, max_wgs * fr::BANK_TICKET_STRIDE * 4)); CK(hipMalloc(&d_diag, max_wgs * 16 * 8 * 8));
    CK(hipMemcpy(d_params, params.data(), params.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_time, time.data(), T * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rows, rows.data(), V * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_tickets, 0, max_wgs * fr::BANK_TICKET_STRIDE * 4));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(fr::g_diag), &d_diag, sizeof(d_diag)));
    struct Var { std::string name; uint32_t small, chunk, nw, mult; };
    std::vector<Var> vars = {{"time-major 8 waves", 0, (uint32_t)log2p, 8, 1}, {"short 2 chunks x 8 waves", 2, (uint32_t)log2p - 1, 8, 1},
                             {"short 4 chunks x 4 waves", 2, (uint32_t)log2p - 2, 4, 1}};
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (const Var &v : vars) {
        fr::BankArgs a{};
        a.params = (const float2 *)d_params; a.time = d_time; a.time_valid = T; a.out = d_out; a.rows = d_rows;
        a.n_voices = V; a.log2_p = log2p; a.n_times = T; a.fast_ok = 1; a.out_stride = T;
        a.chunk_log2 = v.chunk; a.frames_per_lane = 1; a.waves_per_group = v.nw; a.small_call = v.small; a.leaf_variant = 1;
        a.ws = d_ws; a.tickets = d_tickets;
        for (int k = 0; k < 30; ++k) {
            if (k == 29) CK(hipMemsetAsync(d_diag, 0, max_wgs * 16 * 8 * 8, st));
            if (fr::launch_bank(a, st) != hipSuccess) { std::printf("%s: launch failed\n", v.name.c_str()); break; }
        }
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> d(max_wgs * 16 * 8);
        CK(hipMemcpy(d.data(), d_diag, d.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t_min = ~0ull, t_max = 0;
        size_t waves = 0;
        for (size_t w = 0; w < max_wgs * 16; ++w)
            if (d[w * 8]) { t_min = std::min(t_min, d[w * 8]); t_max = std::max(t_max, std::max(d[w * 8 + 2], d[w * 8 + 1])); ++waves; }
        std::vector<double> starts, ends, comp;
        std::map<unsigned, double> simd_end, simd_start, simd_busy;   // key: xcc, se, sh, cu, simd
        for (size_t w = 0; w < max_wgs * 16; ++w) {
            if (!d[w * 8]) continue;
            const double s0 = (d[w * 8] - t_min) * 0.01, c1 = (d[w * 8 + 1] - t_min) * 0.01, e2 = (std::max(d[w * 8 + 2], d[w * 8 + 1]) - t_min) * 0.01;
            starts.push_back(s0); ends.push_back(e2); comp.push_back(c1 - s0);
            const unsigned hw = (unsigned)(d[w * 8 + 3] & 0xFFFFFFFFu), xcc = (unsigned)((d[w * 8 + 3] >> 32) & 15u);
            
            const unsigned key = (xcc << 16) | (((hw >> 13) & 7u) << 12) | (((hw >> 12) & 1u) << 11) | (((hw >> 8) & 15u) << 4) | ((hw >> 4) & 3u);
            simd_end[key] = std::max(simd_end[key], e2);
            if (!simd_start.count(key)) simd_start[key] = s0; else simd_start[key] = std::min(simd_start[key], s0);
            simd_busy[key] += c1 - s0;
        }
        std::vector<double> clk;
        std::map<unsigned, int> simd_waves;
        for (size_t w = 0; w < max_wgs * 16; ++w) {
            if (!d[w * 8]) continue;
            const double rt = (double)(std::max(d[w * 8 + 2], d[w * 8 + 1]) - d[w * 8]) * 10e-9;       // seconds (100 MHz)
            const double cy = (double)(std::max(d[w * 8 + 6], d[w * 8 + 5]) - d[w * 8 + 4]);
            if (rt > 0) clk.push_back(cy / rt / 1e9);
            const unsigned hw = (unsigned)(d[w * 8 + 3] & 0xFFFFFFFFu), xcc = (unsigned)((d[w * 8 + 3] >> 32) & 15u);
            ++simd_waves[(xcc << 16) | (((hw >> 13) & 7u) << 12) | (((hw >> 12) & 1u) << 11) | (((hw >> 8) & 15u) << 4) | ((hw >> 4) & 3u)];
        }
        std::vector<double> wps;
        for (auto &kv : simd_waves) wps.push_back(kv.second);
        std::vector<double> se, ss, sb;
        for (auto &kv : simd_end) { se.push_back(kv.second); ss.push_back(simd_start[kv.first]); sb.push_back(simd_busy[kv.first]); }
        std::printf("\n== %s: %d x %d x %d; %zu waves stamped, span %.2f us (first start -> last end)\n", v.name.c_str(), V, P, T, waves, (t_max - t_min) * 0.01);
        std::printf("   wave starts  us: p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f\n", pct(starts, 0), pct(starts, .1), pct(starts, .5), pct(starts, .9), pct(starts, 1));
        std::printf("   wave ends    us: p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f\n", pct(ends, 0), pct(ends, .1), pct(ends, .5), pct(ends, .9), pct(ends, 1));
        std::printf("   start->compute done per wave us: p0 %.2f p50 %.2f p100 %.2f;\n", pct(comp, 0), pct(comp, .5), pct(comp, 1));
        std::printf("   %zu SIMDs seen; a SIMD's first wave starts: p0 %.2f p50 %.2f p100 %.2f; its last wave ends: p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f\n", se.size(),
                    pct(ss, 0), pct(ss, .5), pct(ss, 1), pct(se, 0), pct(se, .1), pct(se, .5), pct(se, .9), pct(se, 1));
        std::printf("   shader clock seen by the waves (s_memtime / s_memrealtime) GHz: p0 %.2f p50 %.2f p100 %.2f; waves stamped per SIMD: p0 %.0f p50 %.0f p100 %.0f\n",
                    pct(clk, 0), pct(clk, .5), pct(clk, 1), pct(wps, 0), pct(wps, .5), pct(wps, 1));
        std::printf("   sum over a SIMD's waves of (start -> done) us: p0 %.1f p50 %.1f p100 %.1f\n", pct(sb, 0), pct(sb, .5), pct(sb, 1));
    }
    return 0;
}
