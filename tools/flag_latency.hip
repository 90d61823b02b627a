// flag_latency.hip -- how long after a launch does the host KNOW the kernel is done?
//   (a) hipLaunchKernel + hipStreamSynchronize            (the runtime's completion signal)
//   (b) the kernel's last act is a system-scope store of a sequence number into mapped pinned memory; the host spins on it
//   (c) like (b), the number written by a second, one-thread kernel queued behind the first
// hipcc --offload-arch=gfx950 -O2 -o tools/_build/flag_latency tools/flag_latency.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(float *buf, int n, uint32_t *flag, uint32_t seq, uint32_t *done, uint32_t nblocks) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = buf[i] * 1.0001f + 1.0f;
    if (flag) {   // last workgroup to finish publishes
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            uint32_t t = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t == nblocks - 1) {
                __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
__global__ void signal(uint32_t *flag, uint32_t seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 256 * 256;
    float *buf;
    CK(hipMalloc(&buf, n * sizeof(float)));
    CK(hipMemset(buf, 0, n * sizeof(float)));
    uint32_t *hflag, *dflag, *done;
    CK(hipHostMalloc(&hflag, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&dflag, hflag, 0));
    CK(hipMalloc(&done, 4));
    CK(hipMemset(done, 0, 4));
    *hflag = 0;
    using clk = std::chrono::steady_clock;
    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    uint32_t seq = 0;
    for (int gap_us : {0, 50, 1000}) {   // idle time between calls (a real-time host sleeps between blocks)
        std::vector<double> a, b, c;
        for (int it = 0; it < 400; ++it) {
            auto idle = [&] { auto t = clk::now(); while (std::chrono::duration<double, std::micro>(clk::now() - t).count() < gap_us) {} };
            idle();
            auto t0 = clk::now();
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, st, buf, n, (uint32_t *)nullptr, 0u, done, 256u);
            CK(hipStreamSynchronize(st));
            a.push_back(std::chrono::duration<double, std::micro>(clk::now() - t0).count());
            idle();
            ++seq;
            t0 = clk::now();
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, st, buf, n, dflag, seq, done, 256u);
            while (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) != seq) {}
            b.push_back(std::chrono::duration<double, std::micro>(clk::now() - t0).count());
            CK(hipStreamSynchronize(st));
            idle();
            ++seq;
            t0 = clk::now();
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, st, buf, n, (uint32_t *)nullptr, 0u, done, 256u);
            hipLaunchKernelGGL(signal, dim3(1), dim3(1), 0, st, dflag, seq);
            while (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) != seq) {}
            c.push_back(std::chrono::duration<double, std::micro>(clk::now() - t0).count());
            CK(hipStreamSynchronize(st));
        }
        std::printf("idle %4d us between calls: launch + hipStreamSynchronize %6.2f us | flag written by the kernel's last workgroup %6.2f us | by a second kernel %6.2f us\n",
                    gap_us, med(a), med(b), med(c));
    }
    return 0;
}
