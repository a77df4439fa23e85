// copybw.hip -- developer yardstick (not part of the product): what a plain float4 streaming kernel achieves on
// this MI355X at the warp kernels' byte counts.  VERDICT r1 asked for this instead of torch.copy_ as the bar.
//   copy : read N bytes + write N bytes (grid-stride, 16 B/lane)       -> "same-bytes copy" of warp forward
//   read : read 2N bytes, reduce (no stores)                            -> shape of warp backward (dOut + U)
//   fill : write N bytes
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/copybw.hip -o tools/ubench/copybw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a_, float4* __restrict__ b_, size_t n) {
    const f4* a = reinterpret_cast<const f4*>(a_); f4* b = reinterpret_cast<f4*>(b_);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_read2(const float4* __restrict__ a, const float4* __restrict__ b, float* out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = a[i], w = b[i];
        s += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_fill(float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = make_float4(1, 2, 3, 4);
}

int main() {
    const size_t sizes[] = {117964800, 235929600, 471859200};      // B=64/128 240x320x3 f32, B=128 480x640x3 f32
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* out; CK(hipMalloc(&out, 4));
    for (size_t bytes : sizes) {
        float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
        const size_t n = bytes / 16;
        for (int grid : {2048, 4096, 8192, (int)((n + 255) / 256)}) {
            for (int which = 0; which < 4; ++which) {
                auto run = [&]() {
                    if (which == 0) hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, 0, a, b, n);
                    if (which == 1) hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, a, b, n);
                    if (which == 2) hipLaunchKernelGGL(k_read2, dim3(grid), dim3(256), 0, 0, a, b, out, n);
                    if (which == 3) hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, b, n);
                };
                for (int i = 0; i < 3; ++i) run();
                CK(hipDeviceSynchronize());
                const int iters = 30;
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) run();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / iters;
                const double moved = which == 3 ? (double)bytes : 2.0 * bytes;
                const char* nm[] = {"copy", "copy_nt", "read2", "fill"};
                printf("{\"kernel\": \"%s\", \"MB_each\": %.2f, \"grid\": %d, \"us\": %.2f, \"TBs\": %.3f, \"frac_of_8TBs\": %.3f}\n",
                       nm[which], bytes / 1e6, grid, us, moved / us / 1e6, moved / us / 1e6 / 8.0);
            }
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
