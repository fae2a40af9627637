// lazy.hip -- partition_stripe(A, K, LazyBisectCostBottleneckSplitter(f::AbstractConnectivityModel, eps))
// (/root/reference/src/LazyBisectCostBottleneckSplitter.jl:140-258) as ONE kernel launch.
//
// The reference bisects on the cost c; a probe is one forward scan over the columns that closes a part at the
// first column whose cost exceeds c.  Its cached array cch[q] (previous column holding the row of nonzero q,
// :153-170) is exactly the link array `prev` that ensure_links builds, so probe_init and probe read the same data
// here; they differ only in the reference's control flow at k == K (:171, :180 vs :208-211), which is kept.
//
// One workgroup of 1024 lanes streams the link array in 16 Ki-entry chunks: 16 entries per lane, flags
// (prev[q] < part start) packed into a 16-bit mask, one block-wide scan of the per-lane counts; the count of a
// column is the prefix at its last entry; the first exceeding column of the chunk is a block-wide minimum.  After
// a split the stream restarts behind the split column with the new threshold.  All control state is block-uniform.
#include "csr.hpp"
#include "model.hpp"

namespace cpk {

__device__ __forceinline__ bool lz_le(int64_t v, double c)
{
    if (c != c) return false;
    if (c >= 9223372036854775808.0) return true;
    if (c < -9223372036854775808.0) return false;
    return v <= (int64_t)floor(c);
}
__device__ __forceinline__ bool lz_le(double v, double c) { return v <= c; }

constexpr int LZ_T = 1024;             // lanes
constexpr int LZ_E = 16;               // link entries per lane and chunk
constexpr int LZ_CH = LZ_T * LZ_E;

struct LazyShared {
    int32_t tbase[LZ_T];               // exclusive prefix of the per-lane flag counts
    uint16_t tmask[LZ_T];              // flags of the lane's 16 entries
    int32_t wsum[LZ_T / 64];
    int32_t red[LZ_T / 64];
    int32_t total;
};

// #flagged entries in [qa, qa + i)
__device__ __forceinline__ int32_t lz_prefix(const LazyShared &S, int32_t i)
{
    if (i >= LZ_CH) return S.total;
    int t = i >> 4, r = i & 15;
    return S.tbase[t] + __popc((uint32_t)S.tmask[t] & ((1u << r) - 1u));
}

template <typename TC>
__global__ void __launch_bounds__(LZ_T) k_lazy_bisect(DevModel<TC> M, int64_t n, int64_t N, int64_t K, const int32_t *__restrict__ pos,
                                                      const int32_t *__restrict__ prev, double c_lo, double c_hi, double eps,
                                                      int64_t *__restrict__ spl, int64_t *__restrict__ spl_hi,
                                                      int64_t *__restrict__ nprobes)
{
    __shared__ LazyShared S;
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // spl / spl_hi live in global memory; every store is made by lane 0 only and read back after a barrier
    if (tid == 0) {
        for (int64_t k = 0; k <= K; k++) { spl[k] = 0; spl_hi[k] = n + 1; }
        spl[0] = 1; spl_hi[0] = 1;                                     // :146-150
    }
    for (int64_t k = 1; k <= K; k++) {                                 // :233-235  c_lo = max(c_lo, f(0, 0, 0, k))
        double v = (double)dm_apply(M, dm_alpha(M, k), (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0);
        c_lo = c_lo < v ? v : c_lo;
    }
    int64_t probes = 0;
    bool first = true, stuck = false;
    while (c_lo * (1 + eps) < c_hi) {                                  // :237-247, :249-257
        double c = (c_lo + c_hi) / 2;
        probes++;
        bool res = true;
        int64_t k = 1;
        int32_t j0 = 0;                                                // 0-based first column of the open part
        int32_t col = 0;                                               // next column to close
        int32_t qs = 0;                                                // next link entry to read
        int32_t cnt0 = 0;                                              // nets of [j0, col) plus flagged entries in [pos[col], qs)
        if (tid == 0) spl[0] = 1;
        while (col < n) {
            // ---- one chunk of link entries [qs, qe), loaded as aligned 16-byte pieces
            int32_t qa = qs & ~3;
            int32_t qe = (int32_t)((int64_t)qa + LZ_CH < N ? (int64_t)qa + LZ_CH : N);
            uint32_t mask = 0;
            {
                int32_t b = qa + tid * LZ_E;
#pragma unroll
                for (int v4 = 0; v4 < LZ_E / 4; v4++) {
                    int32_t x = b + 4 * v4;
                    if (x < qe) {                                      // arrays are padded by 8 entries
                        int4 v = *reinterpret_cast<const int4 *>(prev + x);
                        if (x >= qs && x < qe && v.x < j0) mask |= 1u << (4 * v4);
                        if (x + 1 >= qs && x + 1 < qe && v.y < j0) mask |= 1u << (4 * v4 + 1);
                        if (x + 2 >= qs && x + 2 < qe && v.z < j0) mask |= 1u << (4 * v4 + 2);
                        if (x + 3 >= qs && x + 3 < qe && v.w < j0) mask |= 1u << (4 * v4 + 3);
                    }
                }
            }
            int32_t mine = __popc(mask), incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { int32_t pv = __shfl_up(incl, o); if (lane >= o) incl += pv; }
            if (lane == 63) S.wsum[wave] = incl;
            __syncthreads();
            int32_t wbase = 0;
            for (int w = 0; w < wave; w++) wbase += S.wsum[w];
            S.tbase[tid] = wbase + incl - mine;
            S.tmask[tid] = (uint16_t)mask;
            if (tid == LZ_T - 1) S.total = wbase + incl;
            __syncthreads();
            // ---- close the columns that end inside the chunk, 1024 at a time
            bool restarted = false;
            bool checks = !first || k < K;                             // probe_init stops checking once k == K (:171)
            while (col < n) {
                int32_t c_me = col + tid;
                int32_t e = (c_me < n) ? pos[c_me + 1] : INT32_MAX;
                bool complete = c_me < n && e <= qe;
                bool exceed = false;
                if (complete && checks) {
                    int64_t nn = (int64_t)cnt0 + lz_prefix(S, e - qa);
                    TC v = dm_apply(M, dm_alpha(M, k), (int64_t)(c_me - j0 + 1), (int64_t)(e - pos[j0]), nn, (int64_t)0);
                    exceed = !lz_le(v, c);
                }
                // first exceeding column and number of complete columns of this batch
                unsigned long long em = __ballot(exceed), cm = __ballot(complete);
                if (lane == 0) { S.red[wave] = em ? (wave * 64 + __ffsll((long long)em) - 1) : INT32_MAX; S.wsum[wave] = __popcll(cm); }
                __syncthreads();
                int32_t fx = INT32_MAX, ncomp = 0;
                for (int w = 0; w < LZ_T / 64; w++) { fx = S.red[w] < fx ? S.red[w] : fx; ncomp += S.wsum[w]; }
                __syncthreads();
                if (fx != INT32_MAX) {
                    // ---- split in front of column cx (:208-219): the column opens the next part on its own
                    int32_t cx = col + fx;
                    int32_t deg = pos[cx + 1] - pos[cx];
                    bool fail = false;
                    while (true) {
                        if (!first && k == K) { fail = true; break; }  // :209-211
                        if (tid == 0) spl[k] = (int64_t)cx + 1;
                        j0 = cx;
                        k += 1;
                        bool again = (!first || k < K) &&
                                     !lz_le(dm_apply(M, dm_alpha(M, k), (int64_t)1, (int64_t)deg, (int64_t)deg, (int64_t)0), c);
                        if (!again) break;
                    }
                    if (fail) { res = false; col = (int32_t)n; restarted = true; break; }
                    col = cx + 1;
                    qs = pos[cx + 1];
                    cnt0 = deg;
                    restarted = true;
                    break;
                }
                col += ncomp;
                if (ncomp < LZ_T) break;                               // the next column ends beyond the chunk
            }
            if (!restarted) {
                cnt0 += S.total;                                       // everything flagged in [qs, qe) belongs to [j0, col]
                qs = qe;
            }
            __syncthreads();
        }
        if (res) {
            if (first) {                                               // :180  res = k < K || f(...) <= c
                int64_t nv = n - j0, np = (n > 0 ? (int64_t)pos[n] - pos[j0] : 0);
                res = k < K || lz_le(dm_apply(M, dm_alpha(M, K), nv, np, (int64_t)cnt0, (int64_t)0), c);
            }
            if (tid == 0) for (int64_t t = k; t <= K; t++) spl[t] = n + 1;    // :181-184 / :221-224
        }
        __syncthreads();
        // no bound moved: the reference would repeat this probe forever (non-positive bounds); block-uniform exit
        if ((res ? c_hi : c_lo) == c || probes > 4096) { stuck = true; break; }
        if (res) {
            c_hi = c;
            for (int64_t t = tid; t <= K; t += LZ_T) spl_hi[t] = spl[t];
        } else {
            c_lo = c;
        }
        first = false;
        __syncthreads();
    }
    if (tid == 0) *nprobes = stuck ? -1 : probes;
}

template <typename TC>
int32_t run_lazy(cp_csr_s *A, int64_t K, const cp_model_t *mdl, double c_lo, double c_hi, double eps, int64_t *spl_out)
{
    hipStream_t s = A->stream;
    HostModel<TC> HM;
    build_dev_model<TC>(mdl, HM, s);
    ensure_links(A);
    DBuf<int64_t> buf((size_t)(2 * (K + 1) + 1));
    int64_t *d_spl = buf.p, *d_hi = buf.p + (K + 1), *d_np = buf.p + 2 * (K + 1);
    {
        ProfScope ps(PROF_BISECT, s, 0.0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lazy_bisect<TC>), dim3(1), dim3(LZ_T), 0, s, HM.d, A->n, A->N, K, A->pos32.p, A->prev.p,
                           c_lo, c_hi, eps, d_spl, d_hi, d_np);
    }
    CP_HIP(hipGetLastError());
    int64_t np_host = 0;
    CP_HIP(hipMemcpyAsync(spl_out, d_hi, sizeof(int64_t) * (size_t)(K + 1), hipMemcpyDeviceToHost, s));
    CP_HIP(hipMemcpyAsync(&np_host, d_np, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    CP_REQUIRE(np_host >= 0, CP_EINVAL, "cost bisection cannot terminate on these bounds (the reference loops forever: non-positive costs)");
    return CP_OK;
}

}  // namespace cpk

using namespace cpk;

extern "C" int32_t cp_partition_lazy_bisect_cost(cp_csr_t A, int64_t K, const cp_model_t *model, double eps, int64_t *spl_out)
{
    try {
        CP_REQUIRE(A && model && spl_out && K >= 1, CP_EINVAL, "bad argument");
        // only AbstractConnectivityModel reaches the specialised method; other models hit the generic one whose g() asserts
        // false for them (LazyBisectCostBottleneckSplitter.jl:486-501)
        CP_REQUIRE(model->kind == CP_MODEL_CONNECTIVITY || model->kind == CP_MODEL_COLBLOCK, CP_EINVAL,
                   "LazyBisectCost: the reference asserts on models that are not connectivity models");
        CP_HIP(hipSetDevice(A->device));
        int64_t li, hi; double lf, hf;
        int32_t rc = cp_bound_stripe(A, K, model, &li, &hi, &lf, &hf);        // :231
        if (rc != CP_OK) return rc;
        CP_REQUIRE(A->N < ((int64_t)1 << 31) - LZ_CH, CP_EUNSUPPORTED, "LazyBisectCost needs nnz < 2^31");
        if (model->dtype == CP_I64) return run_lazy<int64_t>(A, K, model, lf, hf, eps, spl_out);
        return run_lazy<double>(A, K, model, lf, hf, eps, spl_out);
    } CP_CATCH_ALL
}
