// fewvoice_bench.hip -- a GPU's share of a voice-sharded job (few voices, long calls): the time-major kernel and the short-call
// kernel with chunks + ticket combine at several shapes, in ONE process, interleaved rounds.  (Round 3 also ran two new
// kernels through it -- equal static shares per CU, and parameters stationary in SGPRs with tiles streaming -- both slower;
// they live in the history at commit f986746, the numbers in profiles/r03_fewvoices.txt.)
// Per variant: kernel time from HIP events around single launches (median / min), wall time per launch of a back-to-back
// train on one stream (what consecutive fill_buffer calls pay: includes the launch boundary), and whether the bits equal
// the product-form reference launch.  Not part of the product; includes the kernels' translation unit directly.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
//              -mllvm -simplifycfg-sink-common=false -o tools/_build/fewvoice_bench tools/fewvoice_bench.hip
// Usage: fewvoice_bench [rounds]      (shapes are listed in main)
#include "../libfriendship_amd/csrc/kernels.hip"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

struct Var { std::string name; uint32_t small, chunk, nw, leaf; };

int main(int argc, char **argv) {
    int rounds = argc > 1 ? std::atoi(argv[1]) : 15;
    const bool silent = argc > 2 && std::strcmp(argv[2], "silent") == 0;   // voice 3 with zero amplitudes: every sum of it an exact zero
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    std::printf("device: %s, %u CUs\n", prop.gcnArchName, cus);
    struct Shape { int V, log2p, T; };
    const Shape shapes[] = {{8, 12, 4800}, {16, 12, 4800}, {32, 12, 4800}, {64, 12, 3072}, {4, 12, 4800}, {8, 12, 2400}, {8, 13, 4800},
                            {32, 14, 4800}, {8, 11, 4800}, {64, 12, 1024}, {64, 12, 512}, {5, 12, 4800}, {7, 12, 4777}, {128, 10, 1600}};
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (const Shape &sh : shapes) {
        const int V = sh.V, log2p = sh.log2p, T = sh.T, P = 1 << log2p;
        std::vector<float> params((size_t)V * P * 2);
        for (int v = 0; v < V; ++v)
            for (int k = 0; k < P; ++k) {
                float f0 = 55.0f * std::pow(2.0f, (v % 64) / 12.0f);
                params[((size_t)v * P + k) * 2] = f0 * (k + 1) / 48000.0f;
                params[((size_t)v * P + k) * 2 + 1] = (silent && v == 3 && V > 4) ? 0.0f : -4.0f / (k + 1);   // voice 3 silent: every unit sum a zero
            }
        std::vector<float> time(T);
        for (int i = 0; i < T; ++i) time[i] = (float)i;                                            // (t = 0: exact zeros with signs)
        std::vector<uint32_t> rows(V);
        for (int v = 0; v < V; ++v) rows[v] = (uint32_t)(V - 1 - v);                               // (rows are a permutation)
        float *d_params, *d_time, *d_out, *d_ws, *d_hist; uint32_t *d_rows, *d_tickets;
        const size_t ws_floats = std::max<size_t>((size_t)V * T * 8, (size_t)cus * 2048) + (size_t)V * ((T + 63) / 64) * 16 * 64;
        const size_t n_tickets = std::max<size_t>((size_t)V * ((T + 63) / 64), 2 * cus) * fr::BANK_TICKET_STRIDE;
        CK(hipMalloc(&d_params, params.size() * 4)); CK(hipMalloc(&d_time, T * 4)); CK(hipMalloc(&d_out, (size_t)V * T * 4));
        CK(hipMalloc(&d_ws, ws_floats * 4)); CK(hipMalloc(&d_rows, V * 4)); CK(hipMalloc(&d_tickets, n_tickets * 4)); CK(hipMalloc(&d_hist, T * 4));
        CK(hipMemcpy(d_params, params.data(), params.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_time, time.data(), T * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_rows, rows.data(), V * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_tickets, 0, n_tickets * 4));

        std::vector<Var> vars;
        vars.push_back({"time-major 8 waves, product-form leaves (reference bits)", 0, (uint32_t)log2p, 8, 0});
        vars.push_back({"time-major 8 waves", 0, (uint32_t)log2p, 8, 1});
        vars.push_back({"time-major 4 waves", 0, (uint32_t)log2p, 4, 1});
        if (log2p >= 10) vars.push_back({"short-call kernel, 2 chunks x 8 waves", 2, (uint32_t)log2p - 1, 8, 1});
        if (log2p >= 11) vars.push_back({"short-call kernel, 4 chunks x 8 waves", 2, (uint32_t)log2p - 2, 8, 1});
        if (log2p >= 11) vars.push_back({"short-call kernel, 4 chunks x 4 waves", 2, (uint32_t)log2p - 2, 4, 1});
        auto make = [&](const Var &v) {
            fr::BankArgs a{};
            a.params = (const float2 *)d_params; a.time = d_time; a.time_valid = T; a.out = d_out; a.rows = d_rows;
            a.n_voices = V; a.log2_p = log2p; a.n_times = T; a.fast_ok = 1; a.out_stride = T;
            a.chunk_log2 = v.chunk; a.frames_per_lane = 1; a.waves_per_group = v.nw; a.small_call = v.small; a.leaf_variant = v.leaf;
            a.ws = d_ws; a.tickets = d_tickets; a.hist_dst = d_hist;
            return a;
        };
        std::vector<std::vector<double>> kt(vars.size());
        std::vector<double> wall(vars.size(), 0.0);
        std::vector<int> same(vars.size(), -1), hist_ok(vars.size(), -1);
        std::vector<float> ref;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r <= rounds; ++r)
            for (size_t i = 0; i < vars.size(); ++i) {
                fr::BankArgs a = make(vars[i]);
                if (r == 0) { CK(hipMemsetAsync(d_out, 0xFF, (size_t)V * T * 4, st)); CK(hipMemsetAsync(d_hist, 0xFF, T * 4, st)); }
                CK(hipEventRecord(e0, st));
                if (fr::launch_bank(a, st) != hipSuccess) { (void)hipGetLastError(); same[i] = -2; continue; }
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) kt[i].push_back(ms * 1e3);
                if (r == 0) {
                    std::vector<float> out((size_t)V * T), hist(T);
                    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hist.data(), d_hist, T * 4, hipMemcpyDeviceToHost));
                    if (i == 0) ref = out;
                    same[i] = std::memcmp(out.data(), ref.data(), out.size() * 4) == 0;
                    hist_ok[i] = std::memcmp(hist.data(), time.data(), T * 4) == 0;
                }
            }
        // back-to-back trains
        const int N = 300;
        for (size_t i = 0; i < vars.size(); ++i) {
            if (same[i] == -2) continue;
            fr::BankArgs a = make(vars[i]);
            for (int k = 0; k < 20; ++k) (void)fr::launch_bank(a, st);
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < N; ++k) (void)fr::launch_bank(a, st);
            CK(hipStreamSynchronize(st));
            wall[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        }
        // after the trains: still the same bits (tickets left clean by every launch)?
        const double ideal = (double)V * P * T * 6 / 78.6432e12 * 1e6;
        std::printf("\n== %d voices x %d partials x %d frames: %llu pairs on %u CUs; VALU work at the nominal lane-op peak %.2f us\n", V, P, T,
                    (unsigned long long)((T + 63) / 64) * V, cus, ideal);
        for (size_t i = 0; i < vars.size(); ++i) {
            if (same[i] == -2) { std::printf("  %-58s  (shape not served)\n", vars[i].name.c_str()); continue; }
            fr::BankArgs a = make(vars[i]);
            CK(hipMemsetAsync(d_out, 0xFF, (size_t)V * T * 4, st));
            (void)fr::launch_bank(a, st);
            CK(hipStreamSynchronize(st));
            std::vector<float> out((size_t)V * T);
            CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
            const int again = std::memcmp(out.data(), ref.data(), out.size() * 4) == 0;
            std::printf("  %-58s  kernel %7.2f / %7.2f us (median / min)   train %7.2f us per launch   frac(train) %.3f   bits %d/%d  hist %d\n", vars[i].name.c_str(),
                        median(kt[i]), *std::min_element(kt[i].begin(), kt[i].end()), wall[i], ideal / wall[i], same[i], again, hist_ok[i]);
        }
        CK(hipFree(d_params)); CK(hipFree(d_time)); CK(hipFree(d_out)); CK(hipFree(d_ws)); CK(hipFree(d_rows)); CK(hipFree(d_tickets)); CK(hipFree(d_hist));
    }
    return 0;
}
