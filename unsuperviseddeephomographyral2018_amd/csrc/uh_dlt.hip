// uh_dlt.hip -- Tensor-DLT: 8 corner deltas -> 3x3 homography, forward and backward (gfx950).
//
// One 64-lane wavefront owns one 8x8 system: lane l holds A[l/8][l%8].  The factorisation is the
// unblocked partial-pivot LU that tf.matrix_solve runs on CPU (Eigen PartialPivLU; pivot = first
// max |a_ik|, true division, a -= l*u) -- /root/reference/code/homography_model.py:242 -- with the
// row broadcasts done by v_readlane / ds_bpermute instead of memory.  No LDS, no barriers; 4 systems
// per 256-thread block.  Latency-bound by construction (0.4 kB per pair): the point is one launch
// instead of ~40 TF ops, and bit-reproducible results.
#include <mutex>

#include "uh_device.h"
#include "uh_host.h"

namespace uh {

// Solve the 8x8 system held one element per lane.  `a` = A[r][c] (r = lane>>3, c = lane&7),
// `b` = rhs[r] replicated over the 8 lanes of row r.  On return x[0..7] is wave-uniform.
// Operation order == oracle/hotpath_numpy.py:lu_solve_partial_pivot (one rounding per op).
template <typename T>
__device__ __forceinline__ void wave_lu_solve(T a, T b, T x[8]) {
    const int lane = threadIdx.x & 63;
    const int r = lane >> 3, c = lane & 7;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // pivot search down column k (rows >= k); strict '>' keeps the first maximum
        int p = k;
        T best = fabs(__shfl(a, k * 8 + k, UH_WAVE));
#pragma unroll
        for (int i = k + 1; i < 8; ++i) {
            T v = fabs(__shfl(a, i * 8 + k, UH_WAVE));
            if (v > best) { best = v; p = i; }
        }
        // swap rows k <-> p (whole rows, L part included, and the rhs)
        int src_r = (r == k) ? p : ((r == p) ? k : r);
        a = __shfl(a, src_r * 8 + c, UH_WAVE);
        b = __shfl(b, src_r * 8, UH_WAVE);
        T piv = __shfl(a, k * 8 + k, UH_WAVE);
        T lik = __shfl(a, r * 8 + k, UH_WAVE) / piv;      // l[r][k]  (meaningful for r > k)
        T ukc = __shfl(a, k * 8 + c, UH_WAVE);            // u[k][c]
        T bk  = __shfl(b, k * 8, UH_WAVE);
        if (r > k) {
            if (c == k) a = lik;
            else if (c > k) a = a - lik * ukc;
            b = b - lik * bk;                              // forward substitution, j ascending
        }
    }
    // back substitution, column-oriented: b[i] -= u[i][j]*x[j] for j descending, divide last
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        T xj = __shfl(b, j * 8, UH_WAVE) / __shfl(a, j * 8 + j, UH_WAVE);
        x[j] = xj;
        T uij = __shfl(a, r * 8 + j, UH_WAVE);
        if (r < j) b = b - uij * xj;
    }
}

// A[r][c] and rhs[r] of the DLT system for this lane (homography_model.py:223-238, Aux_M* selectors)
//   row 2i   : [0,0,0,-x,-y,-1,  y'x,  y'y] . h = -y'
//   row 2i+1 : [x,y,1, 0, 0, 0, -x'x, -x'y] . h =  x'
template <typename T>
__device__ __forceinline__ void dlt_entry(const float* __restrict__ pts1, const float* __restrict__ h4p,
                                          int r, int c, T& a, T& b) {
    const int i = r >> 1;
    const float x = pts1[2 * i], y = pts1[2 * i + 1];
    // p2 = pts1 + h4p is an f32 add in the reference (homography_model.py:176) even on the f64 path
    const float xp = h4p[2 * i] + x, yp = h4p[2 * i + 1] + y;
    T v = 0;
    if ((r & 1) == 0) {
        if (c == 3) v = -(T)x; else if (c == 4) v = -(T)y; else if (c == 5) v = -1;
        else if (c == 6) v = (T)yp * (T)x; else if (c == 7) v = (T)yp * (T)y;
        b = -(T)yp;
    } else {
        if (c == 0) v = (T)x; else if (c == 1) v = (T)y; else if (c == 2) v = 1;
        else if (c == 6) v = (T)xp * -(T)x; else if (c == 7) v = (T)xp * -(T)y;
        b = (T)xp;
    }
    a = v;
}

struct Mat3 { float a[9]; };

// out = (L @ X) @ R, k-sequential, one rounding per op (oracle: _matmul3)
__device__ __forceinline__ void sandwich3(const Mat3& L, const float X[9], const Mat3& R, float out[9]) {
    float tmp[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            tmp[i * 3 + j] = (L.a[i * 3] * X[j] + L.a[i * 3 + 1] * X[3 + j]) + L.a[i * 3 + 2] * X[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            out[i * 3 + j] = (tmp[i * 3] * R.a[j] + tmp[i * 3 + 1] * R.a[3 + j]) + tmp[i * 3 + 2] * R.a[6 + j];
}

template <typename T>
__global__ __launch_bounds__(256) void dlt_forward_kernel(const float* __restrict__ pts1,
                                                          const float* __restrict__ h4p,
                                                          float* __restrict__ H, float* __restrict__ theta,
                                                          Mat3 M, Mat3 Minv, int B) {
    const int sys = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sys >= B) return;                                   // whole wave exits together
    const int lane = threadIdx.x & 63;
    T a, b, x[8];
    dlt_entry<T>(pts1 + (size_t)sys * 8, h4p + (size_t)sys * 8, lane >> 3, lane & 7, a, b);
    wave_lu_solve<T>(a, b, x);
    float Hm[9];
#pragma unroll
    for (int j = 0; j < 8; ++j) Hm[j] = (float)x[j];
    Hm[8] = 1.0f;                                           // homography_model.py:247-250
    float th[9];
    if (theta) sandwich3(Minv, Hm, M, th);                  // homography_model.py:254
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 9; ++j) H[(size_t)sys * 9 + j] = Hm[j];
        if (theta) {
#pragma unroll
            for (int j = 0; j < 9; ++j) theta[(size_t)sys * 9 + j] = th[j];
        }
    }
}

// pairs whose gradient UH_DLT_ZERO_NONFINITE_GRAD zeroed since the last reset, on this device (uh_dlt_zeroed_pairs)
__device__ unsigned long long g_dlt_zeroed_pairs = 0ull;
__device__ unsigned long long g_dlt_zeroed_taken = 0ull;
// read (and clear) the counter in ONE atomic: an increment can never fall between a read and a separate write of zero
__global__ void dlt_take_zeroed_kernel(int reset) {
    g_dlt_zeroed_taken = reset ? atomicExch(&g_dlt_zeroed_pairs, 0ull) : atomicAdd(&g_dlt_zeroed_pairs, 0ull);
}

// Backward: g_b = A^-T g_h (tf MatrixSolveGrad: matrix_solve(A, grad, adjoint=True), i.e. an LU of
// A^T), g_A = -g_b h^T; only columns 6,7 of A and the rhs depend on p2 = pts1 + h4p, so
//   d/dx'_i =  g_b[2i+1] * (h6 x_i + h7 y_i + 1),   d/dy'_i = -g_b[2i] * (h6 x_i + h7 y_i + 1).
template <typename T>
__global__ __launch_bounds__(256) void dlt_backward_kernel(const float* __restrict__ pts1,
                                                           const float* __restrict__ h4p,
                                                           const float* __restrict__ H,
                                                           const float* __restrict__ dH,
                                                           const float* __restrict__ dtheta,
                                                           Mat3 MT, Mat3 MinvT,
                                                           float* __restrict__ dh4p, int B, int zero_nonfinite) {
    const int sys = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sys >= B) return;
    const int lane = threadIdx.x & 63;
    const int r = lane >> 3, c = lane & 7;
    const float* p1 = pts1 + (size_t)sys * 8;
    float g[9];
    if (dtheta) {                                           // dH = Minv^T dtheta M^T
        float dt[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) dt[j] = dtheta[(size_t)sys * 9 + j];
        sandwich3(MinvT, dt, MT, g);
    } else {
#pragma unroll
        for (int j = 0; j < 9; ++j) g[j] = dH[(size_t)sys * 9 + j];
    }
    // transpose of A: this lane holds A^T[r][c] = A[c][r]; rhs = g_h[r]
    T a, dummy, x[8];
    dlt_entry<T>(p1, h4p + (size_t)sys * 8, c, r, a, dummy);
    T b = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (r == j) b = (T)g[j];
    wave_lu_solve<T>(a, b, x);                              // x = g_b, wave-uniform
    if (lane < 8) {
        const int i = lane >> 1;
        const T h6 = (T)H[(size_t)sys * 9 + 6], h7 = (T)H[(size_t)sys * 9 + 7];
        const T s = (h6 * (T)p1[2 * i] + h7 * (T)p1[2 * i + 1]) + (T)1;
        T gb_odd = 0, gb_even = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i == j) { gb_even = x[2 * j]; gb_odd = x[2 * j + 1]; }
        float v = (float)((lane & 1) ? -gb_even * s : gb_odd * s);
        if (zero_nonfinite) {                               // UH_DLT_ZERO_NONFINITE_GRAD: all 8 or nothing
            const bool bad = !(fabsf(v) <= 3.402823466e38f);          // NaN or Inf
            if (__ballot(bad) != 0ull) {                              // (only lanes 0..7 are active here)
                v = 0.f;
                if (lane == 0) atomicAdd(&g_dlt_zeroed_pairs, 1ull);  // rare: once in millions of pairs
            }
        }
        dh4p[(size_t)sys * 8 + lane] = v;
    }
}

}  // namespace uh

// ---- C ABI ------------------------------------------------------------------------------------
using namespace uh;

static Mat3 load_mat3(const float* host, bool transpose) {
    Mat3 m;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m.a[i * 3 + j] = host ? (transpose ? host[j * 3 + i] : host[i * 3 + j]) : 0.f;
    return m;
}

extern "C" int uh_dlt_forward(const float* pts1, const float* h4p, float* H, float* theta,
                              const float* M_host, const float* Minv_host, int B, unsigned flags,
                              uh_stream_t stream) {
    if (!pts1 || !h4p || !H) return UH_E_NULL;
    if (theta && (!M_host || !Minv_host)) return UH_E_NULL;
    if (B <= 0) return UH_E_SHAPE;
    Mat3 M = load_mat3(M_host, false), Minv = load_mat3(Minv_host, false);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((B + 3) / 4), block(256);
    ProfScope prof(UH_K_DLT_FWD, s);
    if (flags & UH_DLT_SOLVE_F64)
        hipLaunchKernelGGL(dlt_forward_kernel<double>, grid, block, 0, s, pts1, h4p, H, theta, M, Minv, B);
    else
        hipLaunchKernelGGL(dlt_forward_kernel<float>, grid, block, 0, s, pts1, h4p, H, theta, M, Minv, B);
    return (int)hipGetLastError();
}

extern "C" int uh_dlt_backward(const float* pts1, const float* h4p, const float* H, const float* dH,
                               const float* dtheta, const float* M_host, const float* Minv_host,
                               float* dh4p, int B, unsigned flags, uh_stream_t stream) {
    if (!pts1 || !h4p || !H || !dh4p) return UH_E_NULL;
    if ((dH == nullptr) == (dtheta == nullptr)) return UH_E_NULL;
    if (dtheta && (!M_host || !Minv_host)) return UH_E_NULL;
    if (B <= 0) return UH_E_SHAPE;
    Mat3 MT = load_mat3(M_host, true), MinvT = load_mat3(Minv_host, true);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((B + 3) / 4), block(256);
    const int zn = (flags & UH_DLT_ZERO_NONFINITE_GRAD) ? 1 : 0;
    ProfScope prof(UH_K_DLT_BWD, s);
    if (flags & UH_DLT_SOLVE_F64)
        hipLaunchKernelGGL(dlt_backward_kernel<double>, grid, block, 0, s, pts1, h4p, H, dH, dtheta, MT, MinvT, dh4p, B, zn);
    else
        hipLaunchKernelGGL(dlt_backward_kernel<float>, grid, block, 0, s, pts1, h4p, H, dH, dtheta, MT, MinvT, dh4p, B, zn);
    return (int)hipGetLastError();
}

extern "C" int uh_dlt_zeroed_pairs(unsigned long long* count, int reset, uh_stream_t stream) {
    // Synchronous on `stream` ONLY (rounds 2-4 synchronised the whole device and launched on the NULL stream, which also
    // stalled the dataloader's upload stream at every log line): one atomicExch kernel ordered after the work already
    // enqueued on `stream` takes (and clears) the counter, an async copy brings it to the host, the stream is waited for.
    // Pairs zeroed by launches on OTHER streams that have not executed yet are counted by the next call: nothing is lost,
    // nothing is counted twice.  A capturing stream is refused (a synchronising call would invalidate the capture).
    // The take lands in ONE device word (g_dlt_zeroed_taken) before it is copied out: two host threads on two streams could
    // otherwise interleave take A, take B, copy A, copy B and lose A's count -> the take / copy / wait sequence is serialised.
    if (!count) return UH_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    // (the NULL stream cannot be captured itself, but it may not be synchronised while a blocking capture is open elsewhere:
    //  HIP reports that as hipErrorStreamCaptureImplicit)
    const hipError_t q = hipStreamIsCapturing(s, &cs);
    if (q == hipErrorStreamCaptureImplicit) { (void)hipGetLastError(); return UH_E_CAPTURING; }
    if (q != hipSuccess) return (int)q;
    if (cs != hipStreamCaptureStatusNone) return UH_E_CAPTURING;
    static std::mutex take_mutex;
    std::lock_guard<std::mutex> lock(take_mutex);
    hipLaunchKernelGGL(uh::dlt_take_zeroed_kernel, dim3(1), dim3(1), 0, s, reset ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyFromSymbolAsync(count, HIP_SYMBOL(uh::g_dlt_zeroed_taken), sizeof(*count), 0, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return (int)e;
    return (int)hipStreamSynchronize(s);
}
