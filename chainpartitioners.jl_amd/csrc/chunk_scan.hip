// chunk_scan.hip -- pack_stripe(A, DynamicTotalChunker(ConstrainedCost(f, VertexCount(), w_max))) as a parallel
// (min,+) scan (/root/reference/src/DynamicChunker.jl:20-56 computes the same table one row after the other):
//
//     cst[j'] = min_{max(1, j'-w) <= j < j'} cst[j] + f(j, j')          ties -> smallest j (strict < scanning j upwards)
//
// With a width window w the recurrence is linear over the (min,+) semiring on the state (cst[t], .., cst[t-w+1]):
// s_t = M_t (x) s_{t-1}, M_t a companion matrix whose first row holds f(t-d, t), d = 1..w.  Matrix products are
// associative, so the n steps split into B blocks:
//   1. k_cs_block : column c of a block's product = the same recurrence started from the unit state e_c -- w
//                   independent scalar chains per block, one lane each, window in registers; row 0 of every partial
//                   product (R[t][c]) and the block product P_b are stored;
//   2. k_cs_scan  : one wave folds the B block products into the state in front of every block (B is a few thousand);
//   3. k_cs_apply : cst[t] = min_c R[t][c] + S_b[c], one lane per row;
//   4. k_cs_argmin: spl[t] = the reference's arg min over the <= w candidates of the finished cost row, one lane per row.
// (min,+) over Int64 is exact, so cst -- and with it every tie -- equals the sequential sweep's; Float64 models use
// this path only when all parameters are integer-valued (sums exact below 2^53), otherwise the one-wave literal kernel.
#include "csr.hpp"
#include "model.hpp"
#include "dp.hpp"

namespace cpk {

template <typename TC> struct CsInf;
template <> struct CsInf<int64_t> { __host__ __device__ static int64_t v() { return (int64_t)1 << 61; } };
template <> struct CsInf<double> { __host__ __device__ static double v() { return __builtin_huge_val(); } };

__device__ __forceinline__ int64_t cs_add(int64_t a, int64_t b)
{
    const int64_t BIG = CsInf<int64_t>::v();
    return (a >= BIG || b >= BIG) ? BIG : cadd(a, b);
}
__device__ __forceinline__ double cs_add(double a, double b) { return a + b; }

__device__ __forceinline__ int64_t cs_bcast(int64_t v, int src)
{
    int lo = __shfl((int)(v & 0xffffffffll), src), hi = __shfl((int)(v >> 32), src);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ double cs_bcast(double v, int src) { return __longlong_as_double((long long)cs_bcast((int64_t)__double_as_longlong(v), src)); }

// F[t * (wmax+1) + d] = f(t - d, t) for 1 <= d <= min(wmax, t-1) (k_window_table); rows t = 2 .. n+1 are the steps
template <typename TC, int WD>
__global__ void __launch_bounds__(256) k_cs_block(int64_t n, int wmax, int64_t L, int64_t B, const TC *__restrict__ F,
                                                  TC *__restrict__ R, TC *__restrict__ P)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t b = gid / WD;
    int c = (int)(gid % WD);
    if (b >= B) return;
    const TC INF = CsInf<TC>::v();
    int64_t t0 = 2 + b * L, t1 = t0 + L;
    if (t1 > n + 2) t1 = n + 2;
    TC win[WD];                                   // win[i] = cst[t-1-i] of this chain
#pragma unroll
    for (int i = 0; i < WD; i++) win[i] = (i == c) ? (TC)0 : INF;
    for (int64_t t = t0; t < t1; t++) {
        const TC *Fr = F + t * (int64_t)(wmax + 1);
        TC v = INF;
#pragma unroll
        for (int d = 1; d <= WD; d++) {
            if (d <= wmax && d <= t - 1) {
                TC x = cs_add(win[d - 1], Fr[d]);
                v = x < v ? x : v;
            }
        }
#pragma unroll
        for (int i = WD - 1; i >= 1; i--) win[i] = win[i - 1];
        win[0] = v;
        R[t * WD + c] = v;
    }
#pragma unroll
    for (int i = 0; i < WD; i++) P[(b * WD + i) * WD + c] = win[i];
}

template <typename TC, int WD>
__global__ void __launch_bounds__(64) k_cs_scan(int64_t B, const TC *__restrict__ P, TC *__restrict__ S)
{
    int lane = threadIdx.x;
    const TC INF = CsInf<TC>::v();
    TC s = (lane == 0) ? (TC)0 : INF;             // state in front of step t = 2: cst[1] = 0, nothing before it
    int row_lane = lane < WD ? lane : 0;
    for (int64_t b = 0; b < B; b++) {
        if (lane < WD) S[b * WD + lane] = s;
        TC row[WD];
#pragma unroll
        for (int c = 0; c < WD; c++) row[c] = P[(b * WD + row_lane) * WD + c];
        TC ns = INF;
#pragma unroll
        for (int c = 0; c < WD; c++) {
            TC sc = cs_bcast(s, c);
            TC x = cs_add(sc, row[c]);
            ns = x < ns ? x : ns;
        }
        s = lane < WD ? ns : INF;
    }
}

template <typename TC, int WD>
__global__ void __launch_bounds__(256) k_cs_apply(int64_t n, int64_t L, const TC *__restrict__ R, const TC *__restrict__ S, TC *__restrict__ cst1)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;       // 1-based row j'
    if (t > n + 1) return;
    if (t == 1) { cst1[0] = (TC)0; return; }
    const TC INF = CsInf<TC>::v();
    int64_t b = (t - 2) / L;
    TC v = INF;
#pragma unroll
    for (int c = 0; c < WD; c++) {
        TC x = cs_add(S[b * WD + c], R[t * WD + c]);
        v = x < v ? x : v;
    }
    cst1[t - 1] = v;
}

// the reference's inner loop on the finished cost row (DynamicChunker.jl:41-49): strict < while j runs upwards
template <typename TC>
__global__ void __launch_bounds__(256) k_cs_argmin(int64_t n, int wmax, const TC *__restrict__ F, const TC *__restrict__ cst1, int64_t *__restrict__ spl1)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 2;
    if (t > n + 1) return;
    const TC *Fr = F + t * (int64_t)(wmax + 1);
    int64_t j0 = t - wmax > 1 ? t - wmax : 1;
    TC best_c = cadd(cst1[j0 - 1], Fr[t - j0]);
    int64_t best_j = j0;
    for (int64_t j = j0 + 1; j <= t - 1; j++) {
        TC c = cadd(cst1[j - 1], Fr[t - j]);
        if (c < best_c) { best_c = c; best_j = j; }
    }
    spl1[t - 1] = best_j;
}

template <typename TC, int WD>
static void cs_run(hipStream_t s, int64_t n, int wmax, const TC *F, TC *cst1, int64_t *spl1)
{
    int64_t L = cdiv(n, (int64_t)4096);
    if (L < 64) L = 64;
    int64_t B = cdiv(n, L);
    DBuf<TC> R((size_t)(n + 2) * WD), P((size_t)B * WD * WD), S((size_t)B * WD);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_block<TC, WD>), dim3((unsigned)cdiv(B * WD, 256)), dim3(256), 0, s, n, wmax, L, B, F, R.p, P.p);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_scan<TC, WD>), dim3(1), dim3(64), 0, s, B, P.p, S.p);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_apply<TC, WD>), dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, n, L, R.p, S.p, cst1);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cs_argmin<TC>), dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, n, wmax, F, cst1, spl1);
    CP_HIP(hipGetLastError());
    CP_HIP(hipStreamSynchronize(s));              // R / P / S are released on return
}

// cst1 / spl1 are the 1-based arrays of the sequential kernel (index j' - 1); returns false when the window is too wide
template <typename TC>
bool pack_dynamic_scan(hipStream_t s, int64_t n, int64_t wmax, const TC *F, TC *cst1, int64_t *spl1)
{
    if (n < 1 || wmax < 1 || wmax > 16) return false;
    if (wmax <= 4) cs_run<TC, 4>(s, n, (int)wmax, F, cst1, spl1);
    else if (wmax <= 8) cs_run<TC, 8>(s, n, (int)wmax, F, cst1, spl1);
    else cs_run<TC, 16>(s, n, (int)wmax, F, cst1, spl1);
    return true;
}

template bool pack_dynamic_scan<int64_t>(hipStream_t, int64_t, int64_t, const int64_t *, int64_t *, int64_t *);
template bool pack_dynamic_scan<double>(hipStream_t, int64_t, int64_t, const double *, double *, int64_t *);

}  // namespace cpk
