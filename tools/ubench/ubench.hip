// ubench.hip -- developer micro-benchmarks for gfx950 (not part of the product):
//   (1) VALU issue cost (cycles per wave64 instruction per SIMD) of the ops the warp kernels use
//   (2) L1 (TA/TCP) cost of per-lane 4/8/12/16-byte loads/stores on L1-resident data
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench.hip -o tools/ubench/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNR = 16;

#define VALU_KERNEL(NAME, DECL, ASM, SINK)                                              \
__global__ __launch_bounds__(256) void NAME(float* out, float seed) {                      \
    DECL;                                                                                  \
    for (int it = 0; it < ITERS; ++it) {                                                   \
        _Pragma("unroll") for (int u = 0; u < UNR / 4; ++u) { ASM; }                       \
    }                                                                                      \
    out[blockIdx.x * 256 + threadIdx.x] = SINK;                                            \
}

// four independent chains a,b,c,d so that dependent-issue latency is hidden by 8 waves/SIMD anyway
#define F4 float a = seed + threadIdx.x, b = a + 1.f, c = a + 2.f, d = a + 3.f, x = seed * 0.5f, y = seed * 0.25f
#define I4 unsigned a = (unsigned)seed + threadIdx.x, b = a + 1, c = a + 2, d = a + 3, x = (unsigned)seed + 7, y = x + 3
#define OP4F(op) asm volatile(op " %0, %4, %5, %0\n" op " %1, %4, %5, %1\n" op " %2, %4, %5, %2\n" op " %3, %4, %5, %3" \
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y))
#define OP4F2(op) asm volatile(op " %0, %4, %0\n" op " %1, %4, %1\n" op " %2, %4, %2\n" op " %3, %4, %3" \
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x))
#define OP4F1(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3" \
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

VALU_KERNEL(k_fma, F4, OP4F("v_fma_f32"), a + b + c + d)
VALU_KERNEL(k_add, F4, OP4F2("v_add_f32"), a + b + c + d)
VALU_KERNEL(k_mul, F4, OP4F2("v_mul_f32"), a + b + c + d)
VALU_KERNEL(k_med3, F4, OP4F("v_med3_f32"), a + b + c + d)
VALU_KERNEL(k_max, F4, OP4F2("v_max_f32"), a + b + c + d)
VALU_KERNEL(k_min, F4, OP4F2("v_min_f32"), a + b + c + d)
VALU_KERNEL(k_sub, F4, OP4F2("v_sub_f32"), a + b + c + d)
VALU_KERNEL(k_trunc, F4, OP4F1("v_trunc_f32"), a + b + c + d)
VALU_KERNEL(k_fract, F4, OP4F1("v_fract_f32"), a + b + c + d)
VALU_KERNEL(k_mov, F4, OP4F1("v_mov_b32"), a + b + c + d)
VALU_KERNEL(k_and, I4, OP4F2("v_and_b32"), (float)(a + b + c + d))
VALU_KERNEL(k_lshl, I4, OP4F2("v_lshlrev_b32"), (float)(a + b + c + d))
VALU_KERNEL(k_min_i32, I4, OP4F2("v_min_i32"), (float)(a + b + c + d))
VALU_KERNEL(k_med3_i32, I4, OP4F("v_med3_i32"), (float)(a + b + c + d))
VALU_KERNEL(k_add3, I4, OP4F("v_add3_u32"), (float)(a + b + c + d))
VALU_KERNEL(k_cvt_u32, F4, OP4F1("v_cvt_u32_f32"), a + b + c + d)
VALU_KERNEL(k_cndmask_s, F4, asm volatile("v_cndmask_b32 %0, %4, %0, s[20:21]\nv_cndmask_b32 %1, %4, %1, s[20:21]\nv_cndmask_b32 %2, %4, %2, s[20:21]\nv_cndmask_b32 %3, %4, %3, s[20:21]" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x) : "s20", "s21"), a + b + c + d)
VALU_KERNEL(k_cmp_cnd, F4, asm volatile("v_cmp_lt_f32 vcc, %0, %4\nv_cndmask_b32 %0, %4, %0, vcc\nv_cmp_lt_f32 vcc, %1, %4\nv_cndmask_b32 %1, %4, %1, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x) : "vcc"), a + b + c + d)
VALU_KERNEL(k_rcp, F4, OP4F1("v_rcp_f32"), a + b + c + d)
VALU_KERNEL(k_floor, F4, OP4F1("v_floor_f32"), a + b + c + d)
VALU_KERNEL(k_cvt_i32, F4, OP4F1("v_cvt_i32_f32"), a + b + c + d)
VALU_KERNEL(k_cvt_f32, F4, OP4F1("v_cvt_f32_i32"), a + b + c + d)
VALU_KERNEL(k_mul_lo, I4, OP4F2("v_mul_lo_u32"), (float)(a + b + c + d))
VALU_KERNEL(k_mul_u24, I4, OP4F2("v_mul_u32_u24"), (float)(a + b + c + d))
VALU_KERNEL(k_mad_u24, I4, OP4F("v_mad_u32_u24"), (float)(a + b + c + d))
VALU_KERNEL(k_lshl_add, I4, OP4F("v_lshl_add_u32"), (float)(a + b + c + d))
VALU_KERNEL(k_add_u32, I4, OP4F2("v_add_u32"), (float)(a + b + c + d))
VALU_KERNEL(k_div_fixup, F4, OP4F("v_div_fixup_f32"), a + b + c + d)
VALU_KERNEL(k_cndmask, F4, asm volatile("v_cndmask_b32 %0, %4, %0, vcc\nv_cndmask_b32 %1, %4, %1, vcc\nv_cndmask_b32 %2, %4, %2, vcc\nv_cndmask_b32 %3, %4, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x) : "vcc"), a + b + c + d)
VALU_KERNEL(k_cmp, F4, asm volatile("v_cmp_lt_f32 vcc, %0, %4\nv_cmp_lt_f32 vcc, %1, %4\nv_cmp_lt_f32 vcc, %2, %4\nv_cmp_lt_f32 vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x) : "vcc"), a + b + c + d)
VALU_KERNEL(k_div_scale, F4, asm volatile("v_div_scale_f32 %0, vcc, %4, %4, %0\nv_div_scale_f32 %1, vcc, %4, %4, %1\nv_div_scale_f32 %2, vcc, %4, %4, %2\nv_div_scale_f32 %3, vcc, %4, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x) : "vcc"), a + b + c + d)
VALU_KERNEL(k_div_fmas, F4, asm volatile("v_div_fmas_f32 %0, %4, %5, %0\nv_div_fmas_f32 %1, %4, %5, %1\nv_div_fmas_f32 %2, %4, %5, %2\nv_div_fmas_f32 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y) : "vcc"), a + b + c + d)

typedef float f2 __attribute__((ext_vector_type(2)));
#define P4 f2 a = {seed + threadIdx.x, seed}, b = a + 1.f, c = a + 2.f, d = a + 3.f, x = {seed * 0.5f, seed}, y = {seed * 0.25f, seed}
VALU_KERNEL(k_pk_fma, P4, OP4F("v_pk_fma_f32"), a.x + b.y + c.x + d.y)
VALU_KERNEL(k_pk_mul, P4, OP4F2("v_pk_mul_f32"), a.x + b.y + c.x + d.y)
VALU_KERNEL(k_pk_add, P4, OP4F2("v_pk_add_f32"), a.x + b.y + c.x + d.y)

typedef unsigned long long u64;
#define L4 u64 a = (u64)seed + threadIdx.x, b = a + 1, c = a + 2, d = a + 3; unsigned x = (unsigned)seed + 7, y = x + 3
VALU_KERNEL(k_mad_u64, L4, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\nv_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y) : "vcc"), (float)(a + b + c + d))
VALU_KERNEL(k_lshl_add_u64, L4, asm volatile("v_lshl_add_u64 %0, %0, 2, %0\nv_lshl_add_u64 %1, %1, 2, %1\nv_lshl_add_u64 %2, %2, 2, %2\nv_lshl_add_u64 %3, %3, 2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)), (float)(a + b + c + d))

// ---- memory: every wave re-reads its own small window (L1 resident) with per-lane loads of W bytes ----
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int MITERS = 512;

template <int WORDS, int SHIFT_BYTES>
__global__ __launch_bounds__(256) void k_load(const unsigned* __restrict__ buf, unsigned* out, int window_bytes) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 0x7fffffff, 0x00020000);
    unsigned base = (unsigned)(wave & 1023) * (unsigned)window_bytes + SHIFT_BYTES;
    unsigned acc = 0;
    for (int it = 0; it < MITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned off = base + (unsigned)u * 64 * WORDS * 4 + lane * WORDS * 4;
            if constexpr (WORDS == 1) acc += __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
            else if constexpr (WORDS == 2) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); acc += v.x ^ v.y; }
            else if constexpr (WORDS == 3) { u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 0); acc += v.x ^ v.y ^ v.z; }
            else { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
        asm volatile("" : "+v"(acc), "+v"(base));      // opaque: the loads cannot be hoisted out of the loop
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WORDS>
__global__ __launch_bounds__(256) void k_store(unsigned* __restrict__ buf, int window_bytes) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 0x7fffffff, 0x00020000);
    unsigned base = (unsigned)wave * (unsigned)window_bytes;
    for (int it = 0; it < MITERS / 4; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned off = base + (unsigned)u * 64 * WORDS * 4 + lane * WORDS * 4;
            unsigned v = it + u;
            if constexpr (WORDS == 1) __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, 0);
            else if constexpr (WORDS == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2{v, v}, r, off, 0, 0);
            else if constexpr (WORDS == 3) __builtin_amdgcn_raw_buffer_store_b96(u32x3{v, v, v}, r, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(u32x4{v, v, v, v}, r, off, 0, 0);
        }
    }
}

// LDS read cost: per-lane reads of WORDS dwords at lane stride STRIDE_WORDS
template <int WORDS, int STRIDE_WORDS>
__global__ __launch_bounds__(256) void k_lds(unsigned* out, int shift) {
    __shared__ unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    const unsigned* p = lds + wave * 1024 + lane * STRIDE_WORDS + shift;
    for (int it = 0; it < MITERS * 4; ++it) {
        if constexpr (WORDS == 1) { unsigned v; asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p)); acc += v; }
        else if constexpr (WORDS == 3) { u32x3 v; asm volatile("ds_read_b96 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p)); acc += v.x ^ v.z; }
        else if constexpr (WORDS == 2) { u32x2 v; asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p)); acc += v.x ^ v.y; }
        else if constexpr (WORDS == 12) { u32x2 v; asm volatile("ds_read2_b32 %0, %1 offset1:1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p)); acc += v.x ^ v.y; }
        else if constexpr (WORDS == 13) { u32x2 v; unsigned w; asm volatile("ds_read2_b32 %0, %2 offset1:1\nds_read_b32 %1, %2 offset:8\ns_waitcnt lgkmcnt(0)" : "=v"(v), "=v"(w) : "v"((unsigned)(size_t)p)); acc += v.x ^ v.y ^ w; }
        else if constexpr (WORDS == 33) { unsigned a, b, c; asm volatile("ds_read_b32 %0, %3\nds_read_b32 %1, %3 offset:4\nds_read_b32 %2, %3 offset:8\ns_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c) : "v"((unsigned)(size_t)p)); acc += a ^ b ^ c; }
        else { u32x4 v; asm volatile("ds_read_b128 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p)); acc += v.x ^ v.w; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
    printf("device %s CUs %d clock %.3f GHz\n", prop.name, cus, ghz);
    const int blocks = cus * 8;                      // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
    float* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN_VALU(K) { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, out, 1.0f); CK(hipDeviceSynchronize());       \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, out, 1.0f); CK(hipEventRecord(e1)); \
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                      \
        double cyc = ms * 1e-3 * ghz * 1e9; double instr_per_simd = 8.0 * ITERS * UNR;                                    \
        printf("%-16s %8.3f ms  %6.2f cycles / wave64 instr / SIMD\n", #K, ms, cyc / instr_per_simd); }
    RUN_VALU(k_fma) RUN_VALU(k_add) RUN_VALU(k_mul) RUN_VALU(k_pk_fma) RUN_VALU(k_pk_mul) RUN_VALU(k_pk_add)
    RUN_VALU(k_max) RUN_VALU(k_min) RUN_VALU(k_sub) RUN_VALU(k_trunc) RUN_VALU(k_fract) RUN_VALU(k_mov) RUN_VALU(k_and) RUN_VALU(k_lshl)
    RUN_VALU(k_min_i32) RUN_VALU(k_med3_i32) RUN_VALU(k_add3) RUN_VALU(k_cvt_u32) RUN_VALU(k_cndmask_s) RUN_VALU(k_cmp_cnd)
    RUN_VALU(k_med3) RUN_VALU(k_rcp) RUN_VALU(k_floor) RUN_VALU(k_cvt_i32) RUN_VALU(k_cvt_f32) RUN_VALU(k_cndmask) RUN_VALU(k_cmp)
    RUN_VALU(k_mul_lo) RUN_VALU(k_mul_u24) RUN_VALU(k_mad_u24) RUN_VALU(k_lshl_add) RUN_VALU(k_add_u32)
    RUN_VALU(k_mad_u64) RUN_VALU(k_lshl_add_u64) RUN_VALU(k_div_scale) RUN_VALU(k_div_fmas) RUN_VALU(k_div_fixup)

    unsigned* buf; CK(hipMalloc(&buf, 64u << 20)); CK(hipMemset(buf, 1, 64u << 20));
#define RUN_MEM(K, WORDS, LABEL) { const int win = 4 * 64 * WORDS * 4 + 256;                                              \
        hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, buf, (unsigned*)out, win); CK(hipDeviceSynchronize());    \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, buf, (unsigned*)out, win);        \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));              \
        double cyc = ms * 1e-3 * ghz * 1e9; double instr_per_cu = 32.0 * MITERS * 4;                                    \
        printf("%-28s %8.3f ms  %6.2f cycles / wave-load / CU   %6.1f B/clk/CU\n", LABEL, ms, cyc / instr_per_cu,       \
               64.0 * WORDS * 4 / (cyc / instr_per_cu)); }
    RUN_MEM((k_load<1, 0>), 1, "load b32 aligned")   RUN_MEM((k_load<2, 0>), 2, "load b64 aligned")
    RUN_MEM((k_load<3, 0>), 3, "load b96 (12B stride)") RUN_MEM((k_load<4, 0>), 4, "load b128 aligned")
    RUN_MEM((k_load<3, 36>), 3, "load b96 shifted 36B") RUN_MEM((k_load<4, 4>), 4, "load b128 misaligned 4B")
    RUN_MEM((k_load<4, 16>), 4, "load b128 shifted 16B")
#define RUN_ST(K, WORDS, LABEL) { const int win = 4 * 64 * WORDS * 4;                                                    \
        hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, buf, win); CK(hipDeviceSynchronize());                    \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, buf, win);                        \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));              \
        double cyc = ms * 1e-3 * ghz * 1e9; double instr_per_cu = 32.0 * MITERS;                                        \
        printf("%-28s %8.3f ms  %6.2f cycles / wave-store / CU  %6.1f B/clk/CU\n", LABEL, ms, cyc / instr_per_cu,       \
               64.0 * WORDS * 4 / (cyc / instr_per_cu)); }
    RUN_ST((k_store<1>), 1, "store b32") RUN_ST((k_store<3>), 3, "store b96 (12B stride)") RUN_ST((k_store<4>), 4, "store b128")
#define RUN_LDS(K, WORDS, LABEL, SHIFT) { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, (unsigned*)out, SHIFT); CK(hipDeviceSynchronize()); \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, (unsigned*)out, SHIFT);           \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));              \
        double cyc = ms * 1e-3 * ghz * 1e9; double instr_per_cu = 32.0 * MITERS * 4;                                    \
        printf("%-28s %8.3f ms  %6.2f cycles / wave-read / CU   %6.1f B/clk/CU\n", LABEL, ms, cyc / instr_per_cu,       \
               64.0 * WORDS * 4 / (cyc / instr_per_cu)); }
    RUN_LDS((k_lds<1, 1>), 1, "lds b32 stride 1", 0) RUN_LDS((k_lds<1, 3>), 1, "lds b32 stride 3", 0)
    RUN_LDS((k_lds<3, 3>), 3, "lds b96 stride 3", 0) RUN_LDS((k_lds<3, 3>), 3, "lds b96 stride 3 shift 1", 1)
    RUN_LDS((k_lds<4, 4>), 4, "lds b128 stride 4", 0)
    RUN_LDS((k_lds<4, 3>), 4, "lds b128 stride 3 (4B align)", 0) RUN_LDS((k_lds<4, 3>), 4, "lds b128 stride 3 shift 1", 1)
    RUN_LDS((k_lds<2, 2>), 2, "lds b64 stride 2", 0) RUN_LDS((k_lds<2, 3>), 2, "lds b64 stride 3 (4B align)", 0)
    RUN_LDS((k_lds<12, 3>), 2, "lds read2_b32 stride 3", 0) RUN_LDS((k_lds<13, 3>), 3, "lds read2+read stride 3", 0)
    RUN_LDS((k_lds<33, 3>), 3, "lds 3x b32 stride 3", 0)
    return 0;
}
