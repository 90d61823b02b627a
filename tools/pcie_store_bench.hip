// pcie_store_bench.hip -- how fast can a kernel store its results straight into pinned host memory?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/pcie_store_bench tools/pcie_store_bench.hip
// Kernels write N floats (default 307200 = config C's [64, 4800] output) from registers to (a) device memory,
// (b) mapped pinned host memory, coherent (fine-grained), (c) mapped pinned host memory, non-coherent (coarse-grained:
// cached in the GPU's L2, written back when the kernel ends); with 4-byte and 16-byte stores per lane.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void store4(float *dst, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v + (float)i;
}
__global__ void store16(float4 *dst, size_t n4, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) dst[i] = make_float4(v + i, v, v, v);
}
// a compute-ish kernel that stores at the end of ~100 us of VALU work per block (like the bank kernel's tiles)
__global__ void busy_store4(float *dst, size_t n, float v, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float x = v + (float)threadIdx.x;
    for (int k = 0; k < iters; ++k) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
    if (i < n) dst[i] = x;
}

// the bank kernel's pattern exactly: 4-wave blocks, only wave 0 stores its 64 results (256 B) when the block's work is done
__global__ void busy_store_wave0(float *dst, size_t n, float v, int iters) {
    float x = v + (float)threadIdx.x;
    for (int k = 0; k < iters; ++k) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
    size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && i < n) dst[i] = x;
}

int main(int argc, char **argv) {
    size_t n = argc > 1 ? (size_t)std::atoll(argv[1]) : 307200;
    float *dev, *hc, *hn;
    CK(hipMalloc(&dev, n * 4));
    CK(hipHostMalloc((void **)&hc, n * 4, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void **)&hn, n * 4, hipHostMallocMapped | hipHostMallocNonCoherent));
    float *dhc, *dhn;
    CK(hipHostGetDevicePointer((void **)&dhc, hc, 0));
    CK(hipHostGetDevicePointer((void **)&dhn, hn, 0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct Target { const char *name; float *p; float *host; } targets[] = {{"device HBM", dev, nullptr}, {"host coherent", dhc, hc}, {"host non-coherent", dhn, hn}};
    std::vector<float> pageable(n);
    for (auto &t : targets) {
        for (int width : {4, 16}) {
            auto launch = [&](float v) {
                if (width == 4) hipLaunchKernelGGL(store4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t.p, n, v);
                else hipLaunchKernelGGL(store16, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (float4 *)t.p, n / 4, v);
            };
            for (int i = 0; i < 5; ++i) launch(1.0f);
            CK(hipStreamSynchronize(st));
            const int reps = 50;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) { launch((float)i); CK(hipStreamSynchronize(st)); }
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            bool ok = !t.host || t.host[width == 4 ? 5 : 4] == (float)(reps - 1) + (width == 4 ? 5.0f : 1.0f);
            std::printf("%-18s %2d B/lane: launch + sync %7.1f us  (%.1f GB/s)%s\n", t.name, width, us, n * 4 / us / 1e3, ok ? "" : "  STALE/WRONG on the host");
        }
    }
    // stores at the end of real work: 4800 blocks x 256 threads, only the first wave of a block stores (like the bank kernel)
    for (auto &t : targets) {
        const unsigned blocks = 4800;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(busy_store4, dim3(blocks), dim3(256), 0, st, t.p, n, 1.0f, 6000);
        CK(hipStreamSynchronize(st));
        const int reps = 30;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(busy_store4, dim3(blocks), dim3(256), 0, st, t.p, n, (float)i, 6000); CK(hipStreamSynchronize(st)); }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("busy kernel (4800 blocks) storing 4 B/lane to %-18s: %7.1f us per launch + sync\n", t.name, us);
    }
    for (auto &t : targets) {
        const unsigned blocks = 4800;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(busy_store_wave0, dim3(blocks), dim3(256), 0, st, t.p, n, 1.0f, 6000);
        CK(hipStreamSynchronize(st));
        const int reps = 30;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(busy_store_wave0, dim3(blocks), dim3(256), 0, st, t.p, n, (float)i, 6000); CK(hipStreamSynchronize(st)); }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("busy kernel, wave 0 of each of 4800 blocks stores 256 B to %-18s: %7.1f us per launch + sync\n", t.name, us);
    }
    // D2H alternatives for the same bytes
    {
        const int reps = 50;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) { CK(hipMemcpyAsync(pageable.data(), dev, n * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("hipMemcpyAsync D2H into pageable: %7.1f us\n", us);
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) { CK(hipMemcpyAsync(hc, dev, n * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("hipMemcpyAsync D2H into pinned:   %7.1f us\n", us);
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) std::memcpy(pageable.data(), hn, n * 4);
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("CPU memcpy pinned -> pageable:    %7.1f us\n", us);
    }
    return 0;
}
