// capi.hip -- extern "C" entry points of libchainpart.so (include/chainpart.h) and the DP drivers.
#include "csr.hpp"
#include "model.hpp"
#include "dp.hpp"
#include <cmath>
#include <memory>

using namespace cpk;

namespace cpk { struct DpBase; }
struct cp_dp_s { int32_t dtype; cpk::DpBase *impl; cp_csr_s *A; };

namespace cpk {

// ------------------------------------------------------------------ small kernels used by the drivers
// per-column count of entries whose previous occurrence lies before `thr` (thr = 0: first occurrences)
__global__ void k_col_count_prev_lt(const int64_t *__restrict__ pos, const int32_t *__restrict__ prev, int32_t thr,
                                    int32_t *__restrict__ out, int64_t n)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    int32_t k = 0;
    for (int64_t q = pos[c]; q < pos[c + 1]; q++) k += (prev[q] < thr);
    out[c] = k;
}

// layer 1: cst[r] = f(1, j', 1) with nets(0, r) = #first occurrences in columns [0, r)   (DynamicSplitter.jl:26-31)
template <typename TC>
__global__ void k_layer1(int64_t n, const int64_t *__restrict__ pos, const int64_t *__restrict__ firsts_before,
                         const int64_t *__restrict__ lpos, DevModel<TC> M, TC alpha, TC *__restrict__ cst, int32_t *__restrict__ ptr)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    int64_t nn = firsts_before ? firsts_before[r] : 0;
    int64_t nl = (M.kind == CP_MODEL_HYPEREDGE_CUT) ? lpos[r] : 0;     // rows whose last column < r
    cst[r] = dm_apply(M, alpha, r, pos[r], nn, nl);
    ptr[r] = 0;
}

// counts for (p, r) ranges: nets = #{q in cols [p,r) : prev[q] < p}, selfnets = #{rows with first in [p,r) and last < r}.
// gridDim.y blocks share one query (grid-stride over its entries) and combine with one atomic per block.
__global__ void __launch_bounds__(256) k_range_counts(int64_t nq, const int64_t *__restrict__ P, const int64_t *__restrict__ Rr,
                                                      const int64_t *__restrict__ pos, const int32_t *__restrict__ prev,
                                                      const int64_t *__restrict__ fpos, const int32_t *__restrict__ flast,
                                                      unsigned long long *__restrict__ nets, unsigned long long *__restrict__ selfnets)
{
    int64_t i = blockIdx.x;
    if (i >= nq) return;
    __shared__ int64_t sh[256];
    int64_t p = P[i], r = Rr[i];
    int64_t stride = (int64_t)gridDim.y * 256, start = (int64_t)blockIdx.y * 256 + threadIdx.x;
    int64_t c = 0;
    if (r > p) {
        int64_t q0 = pos[p], q1 = pos[r];
        int32_t thr = (int32_t)p;
        for (int64_t q = q0 + start; q < q1; q += stride) c += (prev[q] < thr);
    }
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < (unsigned)o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && sh[0]) atomicAdd(&nets[i], (unsigned long long)sh[0]);
    __syncthreads();
    if (selfnets) {
        c = 0;
        if (r > p) {
            int64_t s0 = fpos[p], s1 = fpos[r];
            int32_t thr = (int32_t)r;
            for (int64_t s = s0 + start; s < s1; s += stride) c += (flast[s] < thr);
        }
        sh[threadIdx.x] = c;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < (unsigned)o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0 && sh[0]) atomicAdd(&selfnets[i], (unsigned long long)sh[0]);
    }
}

template <typename TC>
__global__ void k_apply_model(int64_t nq, const int64_t *__restrict__ P, const int64_t *__restrict__ Rr, const int64_t *__restrict__ Kk,
                              const int64_t *__restrict__ pos, const int64_t *__restrict__ nets, const int64_t *__restrict__ selfnets,
                              DevModel<TC> M, TC *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int64_t p = P[i], r = Rr[i];
    TC alpha = dm_alpha(M, Kk ? Kk[i] : (int64_t)0);
    out[i] = dm_apply(M, alpha, r - p, pos[r] - pos[p], nets ? nets[i] : (int64_t)0, selfnets ? selfnets[i] : (int64_t)0);
}

// ------------------------------------------------------------------ helpers
template <typename F>
static int32_t guarded(F &&f)
{
    try { return f(); }
    catch (const HipFail &e) { return e.code; }
    catch (const std::bad_alloc &) { set_error("host allocation failed"); return CP_EHIP; }
}

static bool model_known(const cp_model_t *m)
{
    if (!m || m->kind < CP_MODEL_FEASIBLE || m->kind > CP_MODEL_SECONDARY) return false;
    if (m->kind == CP_MODEL_POWER_WORK) return m->dtype == CP_F64;
    return m->dtype == CP_I64 || m->dtype == CP_F64;
}


// is the O(n log^2 n) total-cost scheme exact for this model?  Needs W[p]+f(p,r) inverse-Monge:
// modular terms (alpha, vertices, pins) are free; the net count is submodular, so beta_net >= 0;
// hyperedge cost = d*b_cut + l*(b_self - b_cut) needs b_cut >= 0 and b_self <= b_cut.
// Float64 models qualify only when every parameter is integer-valued (then all sums are exact).
static bool fast_total_ok(const cp_model_t *m, int64_t n, int64_t N, int64_t K)
{
    if (!model_exact_on(m, n, N, K)) return false;      // (Float64: integer-valued parameters AND every reachable total below 2^53)
    auto P = [&](int i) { return m->dtype == CP_I64 ? (double)m->p_i64[i] : m->p_f64[i]; };
    if (m->kind == CP_MODEL_WORK) return true;
    if (m->kind == CP_MODEL_CONNECTIVITY) return P(CP_P_NET) >= 0;
    if (m->kind == CP_MODEL_HYPEREDGE_CUT) return P(CP_P_CUT_NET) >= 0 && P(CP_P_SELF_NET) <= P(CP_P_CUT_NET);
    return false;
}

// is the valley search of dp_bottleneck.hip exact for this model?  Needs a cost that grows with its part: every beta >= 0
// (hyperedge cut: cost = d*b_cut + l*(b_self - b_cut) with d, l growing, so b_cut >= 0 and b_self >= b_cut).  alpha, alpha[k]
// are free.  Element type: Work / Connectivity costs are sums of terms that are EACH monotone in the part (counts times a
// non-negative beta) and IEEE addition is monotone, so non-integral Float64 parameters keep the valley.  The hyperedge cost is
// evaluated as fl(l*b_self) + fl((d-l)*b_cut) (HyperedgeCutCosts.jl:21) and its (d - l) term is NOT monotone in the part:
// with non-integral betas the rounded value can rise by an ulp while the part shrinks and the valley breaks (35 of 360 layers
// differed from the literal sweep for (0,0,0,.1,.1), (0,0,0,.7,.1), (.3,.1,0,.3,.3)).  Those go to the general sweep; with
// integer-valued parameters and totals below 2^53 every product and sum is exact and the real-number argument holds.
static bool fast_bottleneck_ok(const cp_model_t *m, int64_t n, int64_t N, int64_t K)
{
    auto P = [&](int i) { return m->dtype == CP_I64 ? (double)m->p_i64[i] : m->p_f64[i]; };
    if (!(P(CP_P_VERTEX) >= 0 && P(CP_P_PIN) >= 0)) return false;
    if (m->kind == CP_MODEL_WORK) return true;
    if (m->kind == CP_MODEL_CONNECTIVITY) return P(CP_P_NET) >= 0;
    if (m->kind == CP_MODEL_HYPEREDGE_CUT)
        return P(CP_P_CUT_NET) >= 0 && P(CP_P_SELF_NET) >= P(CP_P_CUT_NET) && (m->dtype == CP_I64 || model_exact_on(m, n, N, K));
    return false;
}

template <typename TC> static TC host_alpha(const cp_model_t *m, int64_t k)
{
    if (m->alpha_k && k >= 1 && k <= m->n_alpha_k) return ((const TC *)m->alpha_k)[k - 1];
    return model_param<TC>(m, CP_P_ALPHA);
}

// diff[0] |= "the two cost rows differ somewhere" (bitwise comparison: the tables are compared, not the values' meaning)
template <typename TC>
__global__ void __launch_bounds__(256) k_rows_differ(const TC *__restrict__ a, const TC *__restrict__ b, int64_t n1, int32_t *__restrict__ diff)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool d = false;
    if (i < n1) {
        static_assert(sizeof(TC) == 8, "cost rows are 8-byte elements");
        d = reinterpret_cast<const unsigned long long *>(a)[i] != reinterpret_cast<const unsigned long long *>(b)[i];
    }
    if (__ballot(d) && (threadIdx.x & 63) == 0) atomicOr(diff, 1);
}

// ------------------------------------------------------------------ the K-part DP driver (unconstrained)
template <typename TC>
static int32_t run_dynamic(cp_csr_s *A, int64_t K, int32_t combine, int32_t order, const cp_model_t *mdl,
                           int64_t *spl_out, int64_t *ptr_tab, TC *cst_tab)
{
    hipStream_t s = A->stream;
    int64_t n = A->n;
    bool need_self = mdl->kind == CP_MODEL_HYPEREDGE_CUT;
    bool fast = combine == CP_COMBINE_SUM && fast_total_ok(mdl, A->n, A->N, K) && !g_opt_force_brute;
    const bool fast_bn = combine == CP_COMBINE_MAX && fast_bottleneck_ok(mdl, A->n, A->N, K) && !g_opt_force_brute;
    if (!fast && !fast_bn)
        CP_REQUIRE(n <= g_opt_brute_max_n, CP_EUNSUPPORTED,
                   "model/objective outside the O(n log^2 n) class and n too large for the O(n^2) device sweep");
    ensure_links(A);
    if (need_self) ensure_self(A);
    HostModel<TC> HM;
    build_dev_model<TC>(mdl, HM, s);
    size_t n1 = (size_t)n + 1;
    DBuf<TC> cstA(n1), cstB(n1);
    DBuf<int32_t> ptr((size_t)K * n1);
    DBuf<int32_t> cnt0((size_t)(n > 0 ? n : 1));
    DBuf<int64_t> firsts(n1), scratch;
    // layer 1
    bool has_nets = mdl->kind == CP_MODEL_CONNECTIVITY || mdl->kind == CP_MODEL_HYPEREDGE_CUT || mdl->kind == CP_MODEL_COLBLOCK;
    if (has_nets) {
        if (n > 0) hipLaunchKernelGGL(k_col_count_prev_lt, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, A->pos.p, A->prev.p, 0, cnt0.p, n);
        exclusive_scan_i32(cnt0.p, firsts.p, n, scratch, s);
    }
    // splitter order passes the part index (per-part alpha[k]); the chunker loop order calls f(j,j') (DynamicSplitter.jl:64)
    auto alpha_of = [&](int64_t k) { return order == CP_ORDER_SPLITTER ? host_alpha<TC>(mdl, k) : model_param<TC>(mdl, CP_P_ALPHA); };
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_layer1<TC>), dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, n, A->pos.p,
                       has_nets ? firsts.p : nullptr, need_self ? A->lpos.p : nullptr, HM.d, alpha_of(1), cstA.p, ptr.p);
    CP_HIP(hipGetLastError());
    auto dump_layer = [&](int64_t k, const TC *cst_dev, bool last_only) {
        if (!ptr_tab) return;
        std::vector<TC> hc(n1);
        std::vector<int32_t> hp(n1);
        CP_HIP(hipMemcpyAsync(hc.data(), cst_dev, sizeof(TC) * n1, hipMemcpyDeviceToHost, s));
        CP_HIP(hipMemcpyAsync(hp.data(), ptr.p + (size_t)(k - 1) * n1, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
        for (size_t r = 0; r < n1; r++) {
            bool keep = !last_only || r == (size_t)n;
            ptr_tab[(size_t)(k - 1) * n1 + r] = keep ? (int64_t)hp[r] + 1 : 0;               // zeros(Ti, n+1, K)
            cst_tab[(size_t)(k - 1) * n1 + r] = keep ? hc[r] : CostTraits<TC>::typemax();     // fill(typemax, ...)
        }
    };
    dump_layer(1, cstA.p, false);
    void *work = fast ? dp_total_work_get<TC>(A) : nullptr;       // (kept in the handle between calls)
    TC *prevc = cstA.p, *curc = cstB.p;
    // cp_set_option("fixed_point", 1) -- OFF by default: a layer is a function of the previous layer's cost row alone (the model
    // does not depend on k unless per-part alphas are given), so once a full layer reproduces its input row bit for bit every
    // later layer repeats it: its argmin row is copied instead of recomputed.  Exact, but it turns K layers into two for
    // costs where empty parts are free (alpha = 0): a property of the input, kept out of the default so that timings mean
    // "K layers computed".
    const bool fp_ok = g_opt_fixed_point && !(order == CP_ORDER_SPLITTER && mdl->alpha_k && mdl->n_alpha_k > 0);
    DBuf<int32_t> diff(1);
    for (int64_t k = 2; k <= K; k++) {
        int32_t *pk = ptr.p + (size_t)(k - 1) * n1;
        bool last = (k == K);
        if (fast) dp_total_layer<TC>(A, HM.d, alpha_of(k), prevc, curc, pk, work, last ? n : 0, n);   // layer K: row n+1 only (:34)
        else if (fast_bn) dp_bottleneck_layer<TC>(A, HM.d, alpha_of(k), prevc, curc, pk, last ? n : 0, n);
        else dp_brute_layer<TC>(A, HM.d, alpha_of(k), combine, prevc, curc, pk, last ? n : 0, n);   // layer K: row n+1 only (:34)
        dump_layer(k, curc, last);
        if (fp_ok && !last) {
            int32_t hd = 1;
            CP_HIP(hipMemsetAsync(diff.p, 0, sizeof(int32_t), s));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rows_differ<TC>), dim3((unsigned)cdiv((int64_t)n1, 256)), dim3(256), 0, s, prevc, curc, (int64_t)n1, diff.p);
            CP_HIP(hipMemcpyAsync(&hd, diff.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            CP_HIP(hipStreamSynchronize(s));
            if (!hd) {                               // fixed point: layers k+1 .. K repeat layer k
                for (int64_t k2 = k + 1; k2 <= K; k2++) {
                    CP_HIP(hipMemcpyAsync(ptr.p + (size_t)(k2 - 1) * n1, pk, sizeof(int32_t) * n1, hipMemcpyDeviceToDevice, s));
                    dump_layer(k2, curc, k2 == K);
                }
                break;
            }
        }
        std::swap(prevc, curc);
    }
    // unravel_splits (DynamicSplitter.jl:89-99): K dependent single-element reads of ptr
    std::vector<int64_t> spl((size_t)K + 1);
    spl[K] = n;
    for (int64_t k = K; k >= 1; k--) {
        int32_t v = 0;
        CP_HIP(hipMemcpyAsync(&v, ptr.p + (size_t)(k - 1) * n1 + (size_t)spl[k], sizeof(int32_t), hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
        spl[k - 1] = v;
    }
    for (int64_t k = 0; k <= K; k++) spl_out[k] = spl[k] + 1;
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    return CP_OK;
}

// ------------------------------------------------------------------ the K-part DP under a width constraint
// partition_stripe(A, K, DynamicTotal{Splitter,Chunker}(ConstrainedCost(f, VertexCount(), w_max)))   DynamicSplitter.jl:206-314.
// Layer k lives on the rows j' in [j'_lo[k], j'_hi[k]] (column_constraints :144-172; for the width weight: closed forms), its
// candidates are j in [max(j'_lo[k-1], j' - w_max), min(j', j'_hi[k-1])] (:233-246), ties -> largest j.  The previous layer's
// window enters through its cost row -- a value no real total reaches outside the window -- and the width through the windowed
// geometry of dp_total_layer; the rows are restricted to the layer's window (the row-tile mechanism of the multi-GPU path).
// The chunker loop order (:260-314) fills the same cells with the same recurrence (part_constraints :174-204 describes the
// same windows column by column) and calls the cost without the part index.
template <typename TC> struct BigCost;
template <> struct BigCost<int64_t> { static __host__ __device__ int64_t v() { return (int64_t)1 << 61; } };
template <> struct BigCost<double> { static __host__ __device__ double v() { return 1152921504606846976.0; } };      // 2^60

// W[p] = cst[p] inside [lo, hi] (0-based rows), a huge value outside
template <typename TC>
__global__ void __launch_bounds__(256) k_mask_row(int64_t n1, int64_t lo, int64_t hi, const TC *__restrict__ cst, TC *__restrict__ W)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n1) W[p] = (p >= lo && p <= hi) ? cst[p] : BigCost<TC>::v();
}

static void width_windows(int64_t n, int64_t K, int64_t w, std::vector<int64_t> &lo, std::vector<int64_t> &hi)
{
    lo.assign((size_t)K + 1, 0); hi.assign((size_t)K + 1, 0);           // 1-based k, 1-based j'
    int64_t jp = n + 1;
    for (int64_t k = K; k >= 1; k--) { lo[(size_t)k] = jp; jp = std::max<int64_t>(1, jp > w ? jp - w : 1); }
    int64_t j = 1;
    for (int64_t k = 1; k <= K; k++) { hi[(size_t)k] = (w >= n + 1 - j) ? n + 1 : j + w; j = hi[(size_t)k]; }
}

// combine = CP_COMBINE_MAX (DynamicBottleneck*(ConstrainedCost(...))): the same windows; the valley search of dp_bottleneck.hip takes
// the layer's candidate limits directly (no masked row: the crossing is searched inside [max(lo[k-1], j' - w), min(j', hi[k-1])],
// where the previous layer's costs are finite and still grow with the prefix -- dropping the last column of a feasible prefix
// keeps every width <= w).
// Any monotone weight w(j, j') = alpha + b_v (j' - j) + b_p (pos[j'] - pos[j]) with b_v, b_p >= 0 (AffineWorkModel: pins per part,
// work per part): the part [j, j') fits iff j >= j0(j'), the first column whose part up to j' fits -- non-decreasing in j'.  The
// reference finds it by advancing j0 while w(j0, j', k) > w_max (DynamicSplitter.jl:235-237); with a monotone weight that is this
// array, found by bisection in the weight's own arithmetic (WorkCosts.jl:17).  j0[r] (0-based row r = j' - 1, 0-based column),
// r + 1 when not even the empty part fits.
template <typename TW>
__global__ void __launch_bounds__(256) k_weight_j0(int64_t n, const int64_t *__restrict__ pos, TW alpha, TW bv, TW bp, TW wmax, int32_t *__restrict__ j0)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    const int64_t pr = pos[r];
    auto fits = [&](int64_t p) { return cadd(cadd(alpha, cmulc(r - p, bv)), cmulc(pr - pos[p], bp)) <= wmax; };
    int64_t lo = 0, hi = r + 1;                            // first p in [0, r] that fits; r + 1: none
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (fits(mid)) hi = mid; else lo = mid + 1; }
    j0[r] = (int32_t)lo;
}

// column_constraints (DynamicSplitter.jl:144-172) from that array: j'_lo walks back from n + 1 (:150-158, the first step taken
// unconditionally), j'_hi forward from 1 (:161-169).  1-based j', as the reference's vectors.
static void weight_windows(int64_t n, int64_t K, const std::vector<int32_t> &j0, std::vector<int64_t> &lo, std::vector<int64_t> &hi)
{
    lo.assign((size_t)K + 1, 0); hi.assign((size_t)K + 1, 0);
    int64_t jp = n + 1;
    for (int64_t k = K; k >= 1; k--) { lo[(size_t)k] = jp; jp = std::min<int64_t>(jp, (int64_t)j0[(size_t)jp - 1] + 1); }
    int64_t j = 1;
    for (int64_t k = 1; k <= K; k++) {
        // the largest j' >= j with j0(j') <= j: j0 is non-decreasing in j'
        const int64_t fit = (int64_t)(std::upper_bound(j0.begin(), j0.end(), (int32_t)(j - 1)) - j0.begin());     // #rows with j0 <= j - 1 (0-based) = the largest such j' (1-based)
        hi[(size_t)k] = std::max<int64_t>(j, fit);
        j = hi[(size_t)k];
    }
}

template <typename TC>
static int32_t run_dynamic_windowed(cp_csr_s *A, int64_t K, int32_t order, const cp_model_t *mdl, int64_t wmax,
                                    int64_t *spl_out, int64_t *ptr_tab, TC *cst_tab, int64_t *win_lo, int64_t *win_hi,
                                    int32_t combine = CP_COMBINE_SUM, const cp_model_t *weight = nullptr, int64_t wmax_i64 = 0, double wmax_f64 = 0)
{
    hipStream_t s = A->stream;
    const int64_t n = A->n;
    const size_t n1 = (size_t)n + 1;
    std::vector<int64_t> lo, hi;
    DBuf<int32_t> j0;
    if (weight) {                                                        // a general monotone weight (bottleneck only): its j0 array
        CP_REQUIRE(combine == CP_COMBINE_MAX && weight->kind == CP_MODEL_WORK && !weight->alpha_k, CP_EINTERNAL, "general weights: bottleneck DP only");
        j0.alloc(n1);
        const unsigned gw = (unsigned)cdiv((int64_t)n1, 256);
        if (weight->dtype == CP_I64)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_weight_j0<int64_t>), dim3(gw), dim3(256), 0, s, n, A->pos.p, weight->p_i64[CP_P_ALPHA],
                               weight->p_i64[CP_P_VERTEX], weight->p_i64[CP_P_PIN], wmax_i64, j0.p);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_weight_j0<double>), dim3(gw), dim3(256), 0, s, n, A->pos.p, weight->p_f64[CP_P_ALPHA],
                               weight->p_f64[CP_P_VERTEX], weight->p_f64[CP_P_PIN], wmax_f64, j0.p);
        CP_HIP(hipGetLastError());
        std::vector<int32_t> hj(n1);
        CP_HIP(hipMemcpyAsync(hj.data(), j0.p, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
        weight_windows(n, K, hj, lo, hi);
    } else {
        width_windows(n, K, wmax, lo, hi);
    }
    if (win_lo) for (int64_t k = 1; k <= K; k++) { win_lo[k - 1] = lo[(size_t)k]; win_hi[k - 1] = hi[(size_t)k]; }
    if (ptr_tab) for (size_t i = 0; i < (size_t)K * n1; i++) { ptr_tab[i] = 0; cst_tab[i] = CostTraits<TC>::typemax(); }
    if (hi[(size_t)K] < n + 1) {                                         // infeasible (:217-222): a degenerate partition, no exception
        for (int64_t k = 0; k < K; k++) spl_out[k] = 1;
        spl_out[K] = n + 1;
        return CP_INFEASIBLE;
    }
    const int64_t w = weight ? 0 : std::min<int64_t>(wmax, std::max<int64_t>(n, 1));     // (wider than the matrix: every window is [0, r])
    const bool need_self = mdl->kind == CP_MODEL_HYPEREDGE_CUT;
    ensure_links(A);
    if (need_self) ensure_self(A);
    HostModel<TC> HM;
    build_dev_model<TC>(mdl, HM, s);
    DBuf<TC> cst(n1), Wm(n1);
    DBuf<int32_t> ptr((size_t)K * n1);
    DBuf<int32_t> cnt0((size_t)(n > 0 ? n : 1));
    DBuf<int64_t> firsts(n1), scratch;
    const bool has_nets = mdl->kind == CP_MODEL_CONNECTIVITY || mdl->kind == CP_MODEL_HYPEREDGE_CUT;
    if (has_nets) {
        if (n > 0) hipLaunchKernelGGL(k_col_count_prev_lt, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, A->pos.p, A->prev.p, 0, cnt0.p, n);
        exclusive_scan_i32(cnt0.p, firsts.p, n, scratch, s);
    }
    auto alpha_of = [&](int64_t k) { return order == CP_ORDER_SPLITTER ? host_alpha<TC>(mdl, k) : model_param<TC>(mdl, CP_P_ALPHA); };
    const unsigned g1 = (unsigned)cdiv((int64_t)n1, 256);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_layer1<TC>), dim3(g1), dim3(256), 0, s, n, A->pos.p, has_nets ? firsts.p : nullptr,
                       need_self ? A->lpos.p : nullptr, HM.d, alpha_of(1), cst.p, ptr.p);
    CP_HIP(hipGetLastError());
    auto dump_layer = [&](int64_t k) {
        if (!ptr_tab) return;
        std::vector<TC> hc(n1);
        std::vector<int32_t> hp(n1);
        CP_HIP(hipMemcpyAsync(hc.data(), cst.p, sizeof(TC) * n1, hipMemcpyDeviceToHost, s));
        CP_HIP(hipMemcpyAsync(hp.data(), ptr.p + (size_t)(k - 1) * n1, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
        for (int64_t r = lo[(size_t)k] - 1; r <= hi[(size_t)k] - 1; r++) {
            ptr_tab[(size_t)(k - 1) * n1 + (size_t)r] = (int64_t)hp[(size_t)r] + 1;
            cst_tab[(size_t)(k - 1) * n1 + (size_t)r] = hc[(size_t)r];
        }
    };
    dump_layer(1);
    void *work = combine == CP_COMBINE_SUM ? dp_total_work_get<TC>(A) : nullptr;
    for (int64_t k = 2; k <= K; k++) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mask_row<TC>), dim3(g1), dim3(256), 0, s, (int64_t)n1, lo[(size_t)k - 1] - 1, hi[(size_t)k - 1] - 1, cst.p, Wm.p);
        if (combine == CP_COMBINE_SUM)
            dp_total_layer<TC>(A, HM.d, alpha_of(k), Wm.p, cst.p, ptr.p + (size_t)(k - 1) * n1, work, lo[(size_t)k] - 1, hi[(size_t)k] - 1, w);
        else
            dp_bottleneck_layer<TC>(A, HM.d, alpha_of(k), Wm.p, cst.p, ptr.p + (size_t)(k - 1) * n1, lo[(size_t)k] - 1, hi[(size_t)k] - 1,
                                    w, lo[(size_t)k - 1] - 1, hi[(size_t)k - 1] - 1, weight ? j0.p : nullptr);
        dump_layer(k);
    }
    // unravel_splits (DynamicSplitter.jl:89-99); every visited cell lies in its layer's window
    std::vector<int64_t> spl((size_t)K + 1);
    spl[(size_t)K] = n;
    for (int64_t k = K; k >= 1; k--) {
        int32_t v = 0;
        CP_HIP(hipMemcpyAsync(&v, ptr.p + (size_t)(k - 1) * n1 + (size_t)spl[(size_t)k], sizeof(int32_t), hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
        spl[(size_t)k - 1] = v;
    }
    for (int64_t k = 0; k <= K; k++) spl_out[k] = spl[(size_t)k] + 1;
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    return CP_OK;
}

// the scalable path takes: total cost, a model of the inverse-Monge class, the width weight, w_max >= 1
static bool windowed_ok(cp_csr_s *A, int64_t K, int32_t combine, const cp_model_t *model, const cp_model_t *weight, int64_t wmax)
{
    // bottleneck: the searched-crossings walk carries the candidate limits (Int64 costs; dp_bottleneck.hip)
    if (combine == CP_COMBINE_MAX)
        return weight && weight->kind == CP_MODEL_VERTEX_COUNT && wmax >= 1 && !g_opt_force_brute && model->dtype == CP_I64 &&
               g_opt_bn_wave >= 2 && (model->kind == CP_MODEL_WORK || model->kind == CP_MODEL_CONNECTIVITY || model->kind == CP_MODEL_HYPEREDGE_CUT) &&
               fast_bottleneck_ok(model, A->n, A->N, K);
    return combine == CP_COMBINE_SUM && weight && weight->kind == CP_MODEL_VERTEX_COUNT && wmax >= 1 && !g_opt_force_brute &&
           (model->kind == CP_MODEL_WORK || model->kind == CP_MODEL_CONNECTIVITY || model->kind == CP_MODEL_HYPEREDGE_CUT) &&
           fast_total_ok(model, A->n, A->N, K);
}

// A weight that is a function of the WIDTH only -- VertexCount(), or AffineWorkModel(alpha, c, 0) with c > 0 (the reference's own
// tests constrain with AffineWorkModel(0, 1, 0), test/test_Partitioners.jl:178-183,256-261) -- bounds the parts by a number of
// columns: the largest nv with w(nv) = alpha + nv c <= w_max, evaluated in the weight's own arithmetic and order (WorkCosts.jl:17;
// the pin term is nv * 0 = 0).  -> that width (0: only empty parts fit, -1: not even those), or -2: not a width weight.
static int64_t width_of_weight(const cp_model_t *w, int64_t n, int64_t wmax_i64, double wmax_f64)
{
    if (!w) return -2;
    if (w->kind == CP_MODEL_VERTEX_COUNT) return wmax_i64;
    if (w->kind != CP_MODEL_WORK || w->alpha_k) return -2;
    auto fits_i = [&](int64_t nv) { return cadd(cadd(w->p_i64[CP_P_ALPHA], cmulc(nv, w->p_i64[CP_P_VERTEX])), cmulc((int64_t)0, w->p_i64[CP_P_PIN])) <= wmax_i64; };
    auto fits_f = [&](int64_t nv) { return cadd(cadd(w->p_f64[CP_P_ALPHA], cmulc(nv, w->p_f64[CP_P_VERTEX])), cmulc((int64_t)0, w->p_f64[CP_P_PIN])) <= wmax_f64; };
    const bool is_i = w->dtype == CP_I64;
    if (is_i ? !(w->p_i64[CP_P_PIN] == 0 && w->p_i64[CP_P_VERTEX] > 0) : !(w->p_f64[CP_P_PIN] == 0.0 && w->p_f64[CP_P_VERTEX] > 0.0)) return -2;
    auto fits = [&](int64_t nv) { return is_i ? fits_i(nv) : fits_f(nv); };
    if (!fits(0)) return -1;
    int64_t lo = 0, hi = n + 1;                       // fits(lo); the weight grows with nv: the largest nv <= n + 1 that fits
    if (fits(hi)) return hi;
    while (hi - lo > 1) { const int64_t mid = lo + ((hi - lo) >> 1); if (fits(mid)) lo = mid; else hi = mid; }
    return lo;
}

// AffineWorkModel(alpha, b_v, b_p) with b_v, b_p >= 0: grows with its part (k_weight_j0)
static bool monotone_work_weight(const cp_model_t *w)
{
    if (!w || w->kind != CP_MODEL_WORK || w->alpha_k) return false;
    return w->dtype == CP_I64 ? (w->p_i64[CP_P_VERTEX] >= 0 && w->p_i64[CP_P_PIN] >= 0) : (w->p_f64[CP_P_VERTEX] >= 0 && w->p_f64[CP_P_PIN] >= 0);
}

// ------------------------------------------------------------------ row-tiled DP (one rank = one tile of rows per layer)
// cp_dp_*: the same layers as run_dynamic, but a rank computes only rows [row_lo, row_hi) of every layer and the caller
// completes the layer's cost vector with a collective (RCCL all_gather over xGMI) before the next layer.
struct DpBase { virtual ~DpBase() {} };
template <typename TC>
struct DpRun : DpBase {
    cp_csr_s *A = nullptr;
    int64_t K = 0, rlo = 0, rhi = 0;          // 0-based inclusive row window
    int64_t wwin = 0;                          // > 0: layers k >= 2 take their candidates from the width window max(0, r - wwin) <= p <= r (cp_dp_set_window)
    std::vector<int64_t> lay_lo, lay_hi;       // the row tile every layer was computed with (cp_dp_set_rows moves it between layers)
    int32_t combine = 0, order = 0;
    cp_model_t mdl{};
    std::vector<TC> alpha_k_host;
    HostModel<TC> HM;
    bool fast = false, fast_bn = false, need_self = false;
    void *work = nullptr;
    DBuf<int32_t> ptr;                         // K x (n+1); only the tile rows of layers >= 2 are meaningful
    ~DpRun() {}                                  // (the layer scratch belongs to the handle)
    TC alpha_of(int64_t k) const
    {
        if (order == CP_ORDER_SPLITTER && !alpha_k_host.empty() && k >= 1 && k <= (int64_t)alpha_k_host.size()) return alpha_k_host[(size_t)k - 1];
        return model_param<TC>(&mdl, CP_P_ALPHA);
    }
};

template <typename TC>
static int32_t dp_begin(cp_csr_s *A, int64_t K, int32_t combine, int32_t order, const cp_model_t *model, int64_t row_lo, int64_t row_hi,
                        DpBase **out)
{
    std::unique_ptr<DpRun<TC>> D(new DpRun<TC>());
    int64_t n = A->n;
    D->A = A; D->K = K; D->combine = combine; D->order = order; D->mdl = *model;
    D->rlo = row_lo - 1; D->rhi = row_hi - 2;
    D->lay_lo.assign((size_t)K + 1, 0); D->lay_hi.assign((size_t)K + 1, -1);      // (a layer this rank never computes owns no row)
    if (model->alpha_k && model->n_alpha_k > 0) {
        D->alpha_k_host.assign((const TC *)model->alpha_k, (const TC *)model->alpha_k + model->n_alpha_k);
        D->mdl.alpha_k = D->alpha_k_host.data();
    }
    D->need_self = model->kind == CP_MODEL_HYPEREDGE_CUT;
    D->fast = combine == CP_COMBINE_SUM && fast_total_ok(model, A->n, A->N, K) && !g_opt_force_brute;
    D->fast_bn = combine == CP_COMBINE_MAX && fast_bottleneck_ok(model, A->n, A->N, K) && !g_opt_force_brute;
    if (!D->fast && !D->fast_bn)
        CP_REQUIRE(n <= g_opt_brute_max_n, CP_EUNSUPPORTED,
                   "model/objective outside the O(n log^2 n) class and n too large for the O(n^2) device sweep");
    ensure_links(A);
    if (D->need_self) ensure_self(A);
    build_dev_model<TC>(&D->mdl, D->HM, A->stream);
    D->ptr.alloc((size_t)K * (size_t)(n + 1));
    CP_HIP(hipMemsetAsync(D->ptr.p, 0, D->ptr.bytes(), A->stream));
    if (D->fast) D->work = dp_total_work_get<TC>(A);
    CP_HIP(hipStreamSynchronize(A->stream));
    *out = D.release();
    return CP_OK;
}

template <typename TC>
static int32_t dp_layer(DpRun<TC> *D, int64_t k, const TC *prev, TC *cur)
{
    cp_csr_s *A = D->A;
    hipStream_t s = A->stream;
    int64_t n = A->n;
    size_t n1 = (size_t)n + 1;
    int32_t *pk = D->ptr.p + (size_t)(k - 1) * n1;
    if (k == 1) {                                  // every rank computes the whole first layer: a column scan, no exchange needed
        const cp_model_t *mdl = &D->mdl;
        bool has_nets = mdl->kind == CP_MODEL_CONNECTIVITY || mdl->kind == CP_MODEL_HYPEREDGE_CUT || mdl->kind == CP_MODEL_COLBLOCK;
        DBuf<int32_t> cnt0((size_t)(n > 0 ? n : 1));
        DBuf<int64_t> firsts(n1), scratch;
        if (has_nets) {
            if (n > 0) hipLaunchKernelGGL(k_col_count_prev_lt, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, A->pos.p, A->prev.p, 0, cnt0.p, n);
            exclusive_scan_i32(cnt0.p, firsts.p, n, scratch, s);
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_layer1<TC>), dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, n, A->pos.p,
                           has_nets ? firsts.p : nullptr, D->need_self ? A->lpos.p : nullptr, D->HM.d, D->alpha_of(1), cur, pk);
        CP_HIP(hipGetLastError());
        CP_HIP(hipStreamSynchronize(s));
        return CP_OK;
    }
    CP_REQUIRE(prev && cur && k >= 2 && k <= D->K, CP_EINVAL, "bad layer");
    int64_t rlo = D->rlo < 0 ? 0 : D->rlo, rhi = D->rhi > n ? n : D->rhi;
    D->lay_lo[(size_t)k] = rlo; D->lay_hi[(size_t)k] = rhi;
    if (rhi >= rlo) {
        CP_REQUIRE(D->wwin == 0 || D->fast, CP_EUNSUPPORTED, "the width window needs the O(n log^2 n) path");
        if (D->fast) dp_total_layer<TC>(A, D->HM.d, D->alpha_of(k), prev, cur, pk, D->work, rlo, rhi, D->wwin);
        else if (D->fast_bn) dp_bottleneck_layer<TC>(A, D->HM.d, D->alpha_of(k), prev, cur, pk, rlo, rhi);
        else dp_brute_layer<TC>(A, D->HM.d, D->alpha_of(k), D->combine, prev, cur, pk, rlo, rhi);
    }
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    return CP_OK;
}

// ------------------------------------------------------------------ ocl(j, j', k) batches
template <typename TC>
static int32_t run_oracle_eval(cp_csr_s *A, const cp_model_t *mdl, int64_t nq, const int64_t *j, const int64_t *jp,
                               const int64_t *k, TC *out)
{
    hipStream_t s = A->stream;
    if (nq <= 0) return CP_OK;
    bool has_nets = mdl->kind == CP_MODEL_CONNECTIVITY || mdl->kind == CP_MODEL_HYPEREDGE_CUT || mdl->kind == CP_MODEL_COLBLOCK;
    bool need_self = mdl->kind == CP_MODEL_HYPEREDGE_CUT;
    if (has_nets) ensure_links(A);
    if (need_self) ensure_self(A);
    std::vector<int64_t> hp((size_t)nq), hr((size_t)nq);
    for (int64_t i = 0; i < nq; i++) {
        CP_REQUIRE(j[i] >= 1 && jp[i] >= j[i] && jp[i] <= A->n + 1, CP_EINVAL, "oracle query needs 1 <= j <= j' <= n+1");
        hp[i] = j[i] - 1; hr[i] = jp[i] - 1;
    }
    DBuf<int64_t> dP((size_t)nq), dR((size_t)nq), dK, dN, dS;
    DBuf<TC> dO((size_t)nq);
    CP_HIP(hipMemcpyAsync(dP.p, hp.data(), sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
    CP_HIP(hipMemcpyAsync(dR.p, hr.data(), sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
    if (k) { dK.alloc((size_t)nq); CP_HIP(hipMemcpyAsync(dK.p, k, sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s)); }
    if (has_nets) {
        dN.alloc((size_t)nq);
        CP_HIP(hipMemsetAsync(dN.p, 0, dN.bytes(), s));
        if (need_self) { dS.alloc((size_t)nq); CP_HIP(hipMemsetAsync(dS.p, 0, dS.bytes(), s)); }
        int64_t per = nq > 0 ? A->N / nq : 0;
        unsigned gy = (unsigned)(per / 65536 + 1);                   // ~64k entries per block
        if (gy > 2048) gy = 2048;
        ProfScope ps(PROF_QUERY, s, 0.0);
        hipLaunchKernelGGL(k_range_counts, dim3((unsigned)nq, gy), dim3(256), 0, s, nq, dP.p, dR.p, A->pos.p, A->prev.p,
                           need_self ? A->fpos.p : nullptr, need_self ? A->flast.p : nullptr, (unsigned long long *)dN.p,
                           need_self ? (unsigned long long *)dS.p : nullptr);
    }
    HostModel<TC> HM;
    build_dev_model<TC>(mdl, HM, s);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply_model<TC>), dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, s, nq, dP.p, dR.p,
                       k ? dK.p : nullptr, A->pos.p, has_nets ? dN.p : nullptr, need_self ? dS.p : nullptr, HM.d, dO.p);
    CP_HIP(hipGetLastError());
    CP_HIP(hipMemcpyAsync(out, dO.p, sizeof(TC) * (size_t)nq, hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    return CP_OK;
}

static int64_t fld_i64(int64_t a, int64_t b)
{
    int64_t q = a / b, r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) q -= 1;
    return q;
}

}  // namespace cpk

// =================================================================== extern "C"
extern "C" {

const char *cp_last_error(void) { return g_last_error.c_str(); }
int32_t cp_version(void) { return 100; }

int32_t cp_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

static int32_t csr_create_impl(int64_t m, int64_t n, int64_t N, const int64_t *colptr, const int64_t *rowval,
                               int32_t device, bool on_device, cp_csr_t *out)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(out && colptr && (rowval || N == 0), CP_EINVAL, "null argument");
        CP_REQUIRE(m >= 0 && n >= 0 && N >= 0, CP_EINVAL, "negative dimension");
        CP_REQUIRE(n < (int64_t)1 << 30 && m < (int64_t)1 << 30 && N < ((int64_t)1 << 31) - 65536, CP_EINVAL,     // 32-bit entry positions; kernels look ahead by up to 16 Ki entries
                   "dimensions exceed the 32-bit link-array layout");
        CP_REQUIRE(cp_device_count() > 0, CP_EHIP, "no HIP device visible: libchainpart has no CPU fallback");
        CP_HIP(hipSetDevice(device));
        std::unique_ptr<cp_csr_s> A(new cp_csr_s());
        A->device = device; A->m = m; A->n = n; A->N = N;
        CP_HIP(hipStreamCreate(&A->stream));
        A->own_stream = true;
        csr_upload(A.get(), colptr, rowval, on_device);
        *out = A.release();
        return CP_OK;
    });
}

int32_t cp_csr_create(int64_t m, int64_t n, int64_t N, const int64_t *colptr, const int64_t *rowval, int32_t device, cp_csr_t *out)
{
    if (colptr && n >= 0 && (colptr[0] != 1 || colptr[n] != N + 1)) { set_error("colptr must be 1-based with colptr[n+1] == nnz+1"); return CP_EINVAL; }
    return csr_create_impl(m, n, N, colptr, rowval, device, false, out);
}

int32_t cp_csr_create_device(int64_t m, int64_t n, int64_t N, const int64_t *colptr_device, const int64_t *rowval_device,
                             int32_t device, cp_csr_t *out)
{
    return csr_create_impl(m, n, N, colptr_device, rowval_device, device, true, out);
}

int32_t cp_csr_destroy(cp_csr_t A)
{
    if (!A) return CP_OK;
    (void)hipSetDevice(A->device);
    if (A->stream) (void)hipStreamSynchronize(A->stream);
    delete A;                                        // (the destructor frees the DP scratch and the handle's own stream)
    return CP_OK;
}

int32_t cp_adjoint(cp_csr_t A, cp_csr_t *out)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && out, CP_EINVAL, "null argument");
        CP_HIP(hipSetDevice(A->device));
        std::unique_ptr<cp_csr_s> T(new cp_csr_s());
        T->device = A->device;
        CP_HIP(hipStreamCreate(&T->stream));
        T->own_stream = true;
        csr_adjoint(A, T.get());
        *out = T.release();
        return CP_OK;
    });
}

int32_t cp_csr_download(cp_csr_t A, int64_t *dims_out, int64_t *colptr_out, int64_t *rowval_out)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A, CP_EINVAL, "null argument");
        if (dims_out) { dims_out[0] = A->m; dims_out[1] = A->n; dims_out[2] = A->N; }
        if (!colptr_out) return CP_OK;
        CP_REQUIRE(rowval_out || A->N == 0, CP_EINVAL, "rowval_out is null");
        CP_HIP(hipSetDevice(A->device));
        csr_download(A, colptr_out, rowval_out);
        return CP_OK;
    });
}

int32_t cp_csr_reset_cache(cp_csr_t A)
{
    if (!A) return CP_EINVAL;
    return guarded([&]() -> int32_t { CP_HIP(hipStreamSynchronize(A->stream)); drop_cache(A); return CP_OK; });
}

int32_t cp_set_stream(cp_csr_t A, void *hip_stream)
{
    if (!A) return CP_EINVAL;
    if (A->own_stream && A->stream) { (void)hipStreamSynchronize(A->stream); (void)hipStreamDestroy(A->stream); }
    A->stream = (hipStream_t)hip_stream;
    A->own_stream = false;
    return CP_OK;
}

int32_t cp_get_stat(const char *name, int64_t *out)
{
    if (!name || !out) return CP_EINVAL;
    if (!strcmp(name, "spec_redo")) { *out = g_spec_redo; return CP_OK; }
    if (!strcmp(name, "poison_hits")) { *out = g_poison_hits; return CP_OK; }
    return CP_EINVAL;
}

int32_t cp_reset_stream(cp_csr_t A)
{
    if (!A) return CP_EINVAL;
    return guarded([&]() -> int32_t {
        if (A->stream || !A->own_stream) (void)hipStreamSynchronize(A->stream);      // (a borrowed stream may be the null stream)
        if (A->own_stream) return CP_OK;
        CP_HIP(hipSetDevice(A->device));
        hipStream_t s = nullptr;
        CP_HIP(hipStreamCreate(&s));
        A->stream = s; A->own_stream = true;
        return CP_OK;
    });
}

int32_t cp_set_option(const char *name, int64_t value)
{
    if (!name) return CP_EINVAL;
    if (!strcmp(name, "force_brute")) { g_opt_force_brute = value; return CP_OK; }
    if (!strcmp(name, "brute_max_n")) { g_opt_brute_max_n = value; return CP_OK; }
    if (!strcmp(name, "dbg")) { g_opt_dbg = value; return CP_OK; }
    if (!strcmp(name, "short_t")) { g_opt_short_t = value; return CP_OK; }
    if (!strcmp(name, "short_e")) { g_opt_short_e = value; return CP_OK; }
    if (!strcmp(name, "rpass_ch")) { int64_t v = 16; while (v < value && v < 4096) v <<= 1; g_opt_rpass_ch = v; return CP_OK; }
    if (!strcmp(name, "prof_only")) { g_prof_only = (int)value; return CP_OK; }
    if (!strcmp(name, "gap_tau")) { g_opt_gap_tau = value > 20 ? 20 : value; return CP_OK; }
    if (!strcmp(name, "gap_min")) { g_opt_gap_min = value < 8 ? 8 : value; return CP_OK; }
    if (!strcmp(name, "gap_nr")) { g_opt_gap_nr = value >= 2 ? 2 : 1; return CP_OK; }
    if (!strcmp(name, "pool")) { g_opt_pool = value ? 1 : 0; if (!value) dev_pool_trim(); return CP_OK; }      // (1: keep freed device blocks >= 1 MB for reuse; 0: return them)
    if (!strcmp(name, "ra_cache")) { g_opt_ra_cache = value; return CP_OK; }
    if (!strcmp(name, "leaf")) { g_opt_leaf = value; return CP_OK; }
    if (!strcmp(name, "poison")) { g_opt_poison = value; if (value) { g_poison_hits = 0; g_spec_redo = 0; } return CP_OK; }
    if (!strcmp(name, "block_tables")) { g_opt_block_tables = value; return CP_OK; }
    if (!strcmp(name, "rpass_small_tau")) { g_opt_rpass_small_tau = value; return CP_OK; }
    if (!strcmp(name, "force_max")) { g_opt_force_max = value < 0 ? 0 : value; return CP_OK; }
    if (!strcmp(name, "setup_bs")) { int64_t v = 64; while (v < value && v < 1024) v <<= 1; g_opt_setup_bs = v; return CP_OK; }
    if (!strcmp(name, "rpass_cap")) { g_opt_rpass_cap = value < 1 ? 1 : value; return CP_OK; }
    if (!strcmp(name, "fixed_point")) { g_opt_fixed_point = value; return CP_OK; }
    if (!strcmp(name, "nospec")) { g_opt_nospec = value; return CP_OK; }
    if (!strcmp(name, "own_min")) { g_opt_own_min = value < 64 ? 64 : value; return CP_OK; }
    if (!strcmp(name, "bn_chunk")) { g_opt_bn_chunk = value < 1 ? 1 : value; return CP_OK; }
    if (!strcmp(name, "bn_wave")) { g_opt_bn_wave = value; return CP_OK; }
    if (!strcmp(name, "bn_slack")) { g_opt_bn_slack = value < 0 ? 0 : value; return CP_OK; }
    if (!strcmp(name, "bn_run")) { g_opt_bn_run = value < 2 ? 2 : value; return CP_OK; }
    set_error("unknown option");
    return CP_EINVAL;
}

int32_t cp_prof_enable(int32_t on) { g_prof_on = on != 0; return CP_OK; }
int32_t cp_prof_reset(void)
{
    for (auto &p : g_prof) { p.launches = 0; p.ms = 0; p.alg_bytes = 0; }
    return CP_OK;
}
int32_t cp_prof_get(int32_t slot, const char **name, int64_t *launches, double *total_ms, double *alg_bytes)
{
    if (slot >= 0 && slot < PROF_NSLOTS) {
        if (name) *name = g_prof[slot].name;
        if (launches) *launches = g_prof[slot].launches;
        if (total_ms) *total_ms = g_prof[slot].ms;
        if (alg_bytes) *alg_bytes = g_prof[slot].alg_bytes;
    }
    return PROF_NSLOTS;
}

int32_t cp_partition_equi(int64_t n, int64_t K, int64_t *spl_out)
{
    if (K < 1 || n < 0 || !spl_out) return CP_EINVAL;
    for (int64_t k = 0; k <= K; k++) spl_out[k] = k * (n / K) + ((n % K) < k ? (n % K) : k) + 1;   // EquiPartitioner.jl:7
    return CP_OK;
}

int32_t cp_pack_equi(int64_t n, int64_t w, int64_t *spl_out, int64_t *K_out)
{
    if (w < 1 || n < 0 || !spl_out || !K_out) return CP_EINVAL;
    int64_t K = 0;
    for (int64_t j = 1; j <= n; j += w) spl_out[K++] = j;                                           // EquiPartitioner.jl:20
    spl_out[K] = n + 1;
    *K_out = K;
    return CP_OK;
}

int32_t cp_partition_dynamic(cp_csr_t A, int64_t K, int32_t combine, int32_t order, const cp_model_t *model,
                             const cp_rowpart_t *Pi, const cp_model_t *weight, int64_t wmax_i64, double wmax_f64, int64_t *spl_out)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && spl_out && model_known(model) && K >= 1, CP_EINVAL, "bad argument");
        CP_REQUIRE(combine == CP_COMBINE_SUM || combine == CP_COMBINE_MAX, CP_EINVAL, "bad combine");
        CP_HIP(hipSetDevice(A->device));
        bool constrained = weight && weight->kind != CP_MODEL_FEASIBLE;
        if (model->kind == CP_MODEL_PRIMARY || model->kind == CP_MODEL_SECONDARY) {
            CP_REQUIRE(!constrained, CP_EUNSUPPORTED, "ConstrainedCost over a plaid connectivity model has no device path");
            if (model->dtype == CP_I64) return run_plaid_dynamic<int64_t>(A, K, combine, order, model, Pi, spl_out);
            return run_plaid_dynamic<double>(A, K, combine, order, model, Pi, spl_out);
        }
        if (constrained) {
            CP_REQUIRE(weight->kind == CP_MODEL_VERTEX_COUNT || (weight->kind == CP_MODEL_WORK && !weight->alpha_k), CP_EINVAL,
                       "weight must be VertexCount or an AffineWorkModel");
            // width weights (VertexCount, AffineWorkModel(alpha, c, 0)): the equivalent number of columns
            const int64_t wv = width_of_weight(weight, A->n, wmax_i64, wmax_f64);
            cp_model_t vc{}; vc.kind = CP_MODEL_VERTEX_COUNT; vc.dtype = CP_I64;
            if (wv >= 1 && windowed_ok(A, K, combine, model, &vc, wv)) {      // O(K n log^2 n): the windowed geometry of dp_total.hip
                if (model->dtype == CP_I64) return run_dynamic_windowed<int64_t>(A, K, order, model, wv, spl_out, nullptr, nullptr, nullptr, nullptr, combine);
                return run_dynamic_windowed<double>(A, K, order, model, wv, spl_out, nullptr, nullptr, nullptr, nullptr, combine);
            }
            // bottleneck under any monotone work weight (pins, vertices + pins): the valley search with the weight's j0 array
            if (combine == CP_COMBINE_MAX && wv == -2 && monotone_work_weight(weight) && model->dtype == CP_I64 && windowed_ok(A, K, combine, model, &vc, 1))
                return run_dynamic_windowed<int64_t>(A, K, order, model, 0, spl_out, nullptr, nullptr, nullptr, nullptr, combine, weight, wmax_i64, wmax_f64);
            if (model->dtype == CP_I64) return run_dyn_constrained<int64_t>(A, K, combine, order, model, Pi, weight, wmax_i64, wmax_f64, spl_out);
            return run_dyn_constrained<double>(A, K, combine, order, model, Pi, weight, wmax_i64, wmax_f64, spl_out);
        }
        CP_REQUIRE(model->kind == CP_MODEL_WORK || model->kind == CP_MODEL_CONNECTIVITY || model->kind == CP_MODEL_HYPEREDGE_CUT ||
                       model->kind == CP_MODEL_COLBLOCK || model->kind == CP_MODEL_POWER_WORK,
                   CP_EUNSUPPORTED, "model kind has no device DP path yet");
        if (model->dtype == CP_I64) return run_dynamic<int64_t>(A, K, combine, order, model, spl_out, nullptr, nullptr);
        return run_dynamic<double>(A, K, combine, order, model, spl_out, nullptr, nullptr);
    });
}

int32_t cp_dynamic_tables(cp_csr_t A, int64_t K, int32_t combine, const cp_model_t *model, const cp_rowpart_t *Pi,
                          int64_t *ptr_out, int64_t *cst_i64, double *cst_f64)
{
    (void)Pi;
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && ptr_out && model_known(model) && K >= 1, CP_EINVAL, "bad argument");
        CP_REQUIRE(combine == CP_COMBINE_SUM || combine == CP_COMBINE_MAX, CP_EINVAL, "bad combine");
        CP_REQUIRE(model->kind == CP_MODEL_WORK || model->kind == CP_MODEL_CONNECTIVITY || model->kind == CP_MODEL_HYPEREDGE_CUT ||
                       model->kind == CP_MODEL_COLBLOCK || model->kind == CP_MODEL_POWER_WORK,
                   CP_EUNSUPPORTED, "model kind has no device DP path");
        CP_HIP(hipSetDevice(A->device));
        std::vector<int64_t> spl((size_t)K + 1);
        if (model->dtype == CP_I64) return run_dynamic<int64_t>(A, K, combine, CP_ORDER_SPLITTER, model, spl.data(), ptr_out, cst_i64);
        return run_dynamic<double>(A, K, combine, CP_ORDER_SPLITTER, model, spl.data(), ptr_out, cst_f64);
    });
}

int32_t cp_dynamic_tables_constrained_combine(cp_csr_t A, int64_t K, int32_t combine, const cp_model_t *model, const cp_model_t *weight,
                                              int64_t wmax_i64, double wmax_f64, int64_t *win_lo, int64_t *win_hi, int64_t *ptr_out,
                                              int64_t *cst_i64, double *cst_f64)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && ptr_out && win_lo && win_hi && model_known(model) && K >= 1, CP_EINVAL, "bad argument");
        CP_REQUIRE(combine == CP_COMBINE_SUM || combine == CP_COMBINE_MAX, CP_EINVAL, "bad combine");
        CP_REQUIRE(!weight || weight->kind == CP_MODEL_VERTEX_COUNT || (weight->kind == CP_MODEL_WORK && !weight->alpha_k), CP_EINVAL,
                   "weight must be VertexCount or an AffineWorkModel");
        cp_model_t vc{}; vc.kind = CP_MODEL_VERTEX_COUNT; vc.dtype = CP_I64;
        const int64_t wv = weight ? width_of_weight(weight, A->n, wmax_i64, wmax_f64) : wmax_i64;      // (null: the width weight)
        const bool general = wv == -2 && combine == CP_COMBINE_MAX && monotone_work_weight(weight) && model->dtype == CP_I64;
        CP_REQUIRE(windowed_ok(A, K, combine, model, &vc, general ? 1 : wv), CP_EUNSUPPORTED, "outside the windowed scalable path");
        CP_HIP(hipSetDevice(A->device));
        std::vector<int64_t> spl((size_t)K + 1);
        if (general) return run_dynamic_windowed<int64_t>(A, K, CP_ORDER_SPLITTER, model, 0, spl.data(), ptr_out, cst_i64, win_lo, win_hi, combine, weight, wmax_i64, wmax_f64);
        if (model->dtype == CP_I64) return run_dynamic_windowed<int64_t>(A, K, CP_ORDER_SPLITTER, model, wv, spl.data(), ptr_out, cst_i64, win_lo, win_hi, combine);
        return run_dynamic_windowed<double>(A, K, CP_ORDER_SPLITTER, model, wv, spl.data(), ptr_out, cst_f64, win_lo, win_hi, combine);
    });
}

int32_t cp_dynamic_tables_constrained(cp_csr_t A, int64_t K, const cp_model_t *model, int64_t wmax, int64_t *win_lo, int64_t *win_hi,
                                      int64_t *ptr_out, int64_t *cst_i64, double *cst_f64)
{
    return cp_dynamic_tables_constrained_combine(A, K, CP_COMBINE_SUM, model, nullptr, wmax, (double)wmax, win_lo, win_hi, ptr_out, cst_i64, cst_f64);
}

int32_t cp_oracle_eval(cp_csr_t A, const cp_model_t *model, const cp_rowpart_t *Pi, int32_t hint, int64_t nq,
                       const int64_t *j, const int64_t *jp, const int64_t *k, int64_t *out_i64, double *out_f64)
{
    (void)hint;
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && model_known(model) && (nq == 0 || (j && jp)), CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        if (model->kind == CP_MODEL_PRIMARY || model->kind == CP_MODEL_SECONDARY) {
            if (model->dtype == CP_I64) return run_plaid_eval<int64_t>(A, model, Pi, nq, j, jp, k, out_i64);
            return run_plaid_eval<double>(A, model, Pi, nq, j, jp, k, out_f64);
        }
        if (model->kind == CP_MODEL_BLOCK) {          // stateful step oracle: evaluated in query order by one wave (seq.hip)
            if (model->dtype == CP_I64) return run_seq_eval<int64_t>(A, model, Pi, nq, j, jp, k, out_i64);
            return run_seq_eval<double>(A, model, Pi, nq, j, jp, k, out_f64);
        }
        if (model->dtype == CP_I64) return run_oracle_eval<int64_t>(A, model, nq, j, jp, k, out_i64);
        return run_oracle_eval<double>(A, model, nq, j, jp, k, out_f64);
    });
}

int32_t cp_oracle_step(cp_csr_t A, const cp_model_t *model, const cp_rowpart_t *Pi, int64_t nq, const int32_t *move_j, const int64_t *j,
                       const int32_t *move_jp, const int64_t *jp, const int64_t *k, int64_t *out_i64, double *out_f64)
{
    if (nq > 0 && (!move_j || !move_jp || !j || !jp)) { set_error("null argument"); return CP_EINVAL; }
    for (int64_t t = 0; t < nq; t++) {
        for (int side = 0; side < 2; side++) {
            const int32_t mv = side ? move_jp[t] : move_j[t];
            const int64_t *x = side ? jp : j;
            if (mv < CP_MOVE_SAME || mv > CP_MOVE_JUMP) { set_error("Step: unknown move code"); return CP_EINVAL; }
            if (t == 0 || mv == CP_MOVE_JUMP) continue;
            const int64_t want = mv == CP_MOVE_SAME ? x[t - 1] : mv == CP_MOVE_NEXT ? x[t - 1] + 1 : x[t - 1] - 1;
            if (x[t] != want) { set_error("Step: a Same / Next / Prev move does not match the previous position"); return CP_EINVAL; }
        }
    }
    return cp_oracle_eval(A, model, Pi, CP_HINT_STEP, nq, j, jp, k, out_i64, out_f64);
}

int32_t cp_objective(cp_csr_t A, int64_t K, const int64_t *spl, const cp_model_t *model, const cp_rowpart_t *Pi,
                     int32_t combine, int64_t *out_i64, double *out_f64)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && spl && model_known(model) && K >= 1, CP_EINVAL, "bad argument");
        if (model->kind == CP_MODEL_SECONDARY) {
            // SecondaryConnectivityCosts.jl:103-108: the primary objective of the adjoint with the two partitions swapped
            CP_REQUIRE(Pi && Pi->spl && Pi->K == K, CP_EINVAL, "the secondary objective needs a SplitPartition Pi with K parts");
            cp_csr_t T = nullptr;
            int32_t rc = cp_adjoint(A, &T);
            if (rc != CP_OK) return rc;
            cp_model_t pm = *model; pm.kind = CP_MODEL_PRIMARY;
            cp_rowpart_t rp; rp.K = K; rp.asg = nullptr; rp.spl = spl;
            rc = cp_objective(T, K, Pi->spl, &pm, &rp, combine, out_i64, out_f64);
            cp_csr_destroy(T);
            return rc;
        }
        std::vector<int64_t> j((size_t)K), jp((size_t)K), kk((size_t)K);
        for (int64_t k = 0; k < K; k++) { j[k] = spl[k]; jp[k] = spl[k + 1]; kk[k] = k + 1; }
        if (model->dtype == CP_I64) {
            std::vector<int64_t> v((size_t)K);
            int32_t rc = cp_oracle_eval(A, model, Pi, CP_HINT_STEP, K, j.data(), jp.data(), kk.data(), v.data(), nullptr);
            if (rc != CP_OK) return rc;
            int64_t acc = combine == CP_COMBINE_SUM ? 0 : INT64_MIN;                  // objective_identity Costs.jl:23-24
            for (int64_t k = 0; k < K; k++) acc = combine == CP_COMBINE_SUM ? cadd(acc, v[k]) : (acc > v[k] ? acc : v[k]);
            *out_i64 = acc;
        } else {
            std::vector<double> v((size_t)K);
            int32_t rc = cp_oracle_eval(A, model, Pi, CP_HINT_STEP, K, j.data(), jp.data(), kk.data(), nullptr, v.data());
            if (rc != CP_OK) return rc;
            double acc = combine == CP_COMBINE_SUM ? 0.0 : -INFINITY;
            for (int64_t k = 0; k < K; k++) acc = combine == CP_COMBINE_SUM ? acc + v[k] : (acc > v[k] ? acc : v[k]);
            *out_f64 = acc;
        }
        return CP_OK;
    });
}

int32_t cp_bound_stripe(cp_csr_t A, int64_t K, const cp_model_t *model, int64_t *lo_i64, int64_t *hi_i64, double *lo_f64, double *hi_f64)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && model_known(model) && K >= 1, CP_EINVAL, "bad argument");
        int64_t n = A->n, N = A->N;
        if (model->kind == CP_MODEL_WORK) {                                            // WorkCosts.jl:39-51
            if (model->dtype == CP_I64) {
                int64_t a = model->p_i64[0], bv = model->p_i64[1], bp = model->p_i64[2];
                *lo_i64 = a + fld_i64(bv * n + bp * N, K);
                if (bv >= 0 && bp >= 0) *hi_i64 = a + bv * n + bp * N;
                else if (bv <= 0 && bp <= 0) *hi_i64 = a;
                else { set_error("bound_stripe: mixed-sign work model"); return CP_EINVAL; }
                *lo_f64 = (double)*lo_i64; *hi_f64 = (double)*hi_i64;
            } else {
                double a = model->p_f64[0], bv = model->p_f64[1], bp = model->p_f64[2];
                *lo_f64 = a + std::floor((bv * (double)n + bp * (double)N) / (double)K);
                if (bv >= 0 && bp >= 0) *hi_f64 = a + (double)n * bv + (double)N * bp;
                else if (bv <= 0 && bp <= 0) *hi_f64 = a;
                else { set_error("bound_stripe: mixed-sign work model"); return CP_EINVAL; }
            }
            return CP_OK;
        }
        if (model->kind == CP_MODEL_PRIMARY) {                                         // PrimaryConnectivityCosts.jl:43-51
            bool neg = model->dtype == CP_I64 ? (model->p_i64[1] < 0 || model->p_i64[2] < 0 || model->p_i64[3] < 0 || model->p_i64[4] < 0)
                                              : (model->p_f64[1] < 0 || model->p_f64[2] < 0 || model->p_f64[3] < 0 || model->p_f64[4] < 0);
            CP_REQUIRE(!neg, CP_EINVAL, "bound_stripe asserts beta >= 0");
            cp_model_t c = *model; c.kind = CP_MODEL_CONNECTIVITY; c.alpha_k = nullptr; c.n_alpha_k = 0;
            int64_t di, dh; double df, dg;
            if (model->dtype == CP_I64) { c.p_i64[CP_P_NET] = std::max(model->p_i64[3], model->p_i64[4]); c.p_i64[4] = 0; }
            else { c.p_f64[CP_P_NET] = std::max(model->p_f64[3], model->p_f64[4]); c.p_f64[4] = 0; }
            int32_t rc = cp_bound_stripe(A, K, &c, &di, hi_i64, &df, hi_f64);
            if (rc != CP_OK) return rc;
            if (model->dtype == CP_I64) c.p_i64[CP_P_NET] = std::min(model->p_i64[3], model->p_i64[4]);
            else c.p_f64[CP_P_NET] = std::min(model->p_f64[3], model->p_f64[4]);
            return cp_bound_stripe(A, K, &c, lo_i64, &dh, lo_f64, &dg);
        }
        if (model->kind == CP_MODEL_CONNECTIVITY && model->alpha_k && model->n_alpha_k > 0) {
            // per-part alpha = the reference tests' FunkyConnectivityModel; its bound_stripe (test_Partitioners.jl:36-41) is
            // (minimum, maximum) of (minimum(alpha), maximum(alpha), maximum_k ocl(1, n+1, k))
            CP_REQUIRE(model->n_alpha_k >= K, CP_EINVAL, "bound_stripe: fewer per-part alphas than parts");
            std::vector<int64_t> one((size_t)K, 1), np1((size_t)K, n + 1), ks((size_t)K);
            for (int64_t k = 0; k < K; k++) ks[(size_t)k] = k + 1;
            if (model->dtype == CP_I64) {
                std::vector<int64_t> v((size_t)K);
                int32_t rc = cp_oracle_eval(A, model, nullptr, CP_HINT_STEP, K, one.data(), np1.data(), ks.data(), v.data(), nullptr);
                if (rc != CP_OK) return rc;
                const int64_t *al = (const int64_t *)model->alpha_k;
                int64_t amin = al[0], amax = al[0], fmax = v[0];
                for (int64_t k = 1; k < K; k++) { amin = std::min(amin, al[k]); amax = std::max(amax, al[k]); fmax = std::max(fmax, v[(size_t)k]); }
                *lo_i64 = std::min(amin, std::min(amax, fmax)); *hi_i64 = std::max(amin, std::max(amax, fmax));
                *lo_f64 = (double)*lo_i64; *hi_f64 = (double)*hi_i64;
            } else {
                std::vector<double> v((size_t)K);
                int32_t rc = cp_oracle_eval(A, model, nullptr, CP_HINT_STEP, K, one.data(), np1.data(), ks.data(), nullptr, v.data());
                if (rc != CP_OK) return rc;
                const double *al = (const double *)model->alpha_k;
                double amin = al[0], amax = al[0], fmax = v[0];
                for (int64_t k = 1; k < K; k++) { amin = std::min(amin, al[k]); amax = std::max(amax, al[k]); fmax = std::max(fmax, v[(size_t)k]); }
                *lo_f64 = std::min(amin, std::min(amax, fmax)); *hi_f64 = std::max(amin, std::max(amax, fmax));
            }
            return CP_OK;
        }
        if (model->kind == CP_MODEL_CONNECTIVITY) {                                    // ConnectivityCosts.jl:25-35
            int64_t one = 1, np1 = n + 1;
            if (model->dtype == CP_I64) {
                CP_REQUIRE(model->p_i64[1] >= 0 && model->p_i64[2] >= 0 && model->p_i64[3] >= 0, CP_EINVAL, "bound_stripe asserts beta >= 0");
                int64_t chi = 0;
                int32_t rc = cp_oracle_eval(A, model, nullptr, CP_HINT_STEP, 1, &one, &np1, nullptr, &chi, nullptr);
                if (rc != CP_OK) return rc;
                *hi_i64 = chi; *lo_i64 = model->p_i64[0] + fld_i64(chi - model->p_i64[0], K);
                *lo_f64 = (double)*lo_i64; *hi_f64 = (double)*hi_i64;
            } else {
                CP_REQUIRE(model->p_f64[1] >= 0 && model->p_f64[2] >= 0 && model->p_f64[3] >= 0, CP_EINVAL, "bound_stripe asserts beta >= 0");
                double chi = 0;
                int32_t rc = cp_oracle_eval(A, model, nullptr, CP_HINT_STEP, 1, &one, &np1, nullptr, nullptr, &chi);
                if (rc != CP_OK) return rc;
                *hi_f64 = chi; *lo_f64 = model->p_f64[0] + std::floor((chi - model->p_f64[0]) / (double)K);
            }
            return CP_OK;
        }
        set_error("bound_stripe has no method for this model (the reference raises MethodError)");
        return CP_EUNSUPPORTED;
    });
}

// bound_stripe(A, K, Pi, mdl): Costs.jl:17-19 drops Pi for every model but the secondary one
// (SecondaryConnectivityCosts.jl:21-31 == :42-61: per part of Pi, c_lo = max work cost, c_hi = max work cost + nets * b_remote)
int32_t cp_bound_stripe_pi(cp_csr_t A, int64_t K, const cp_rowpart_t *Pi, const cp_model_t *model, int64_t *lo_i64, int64_t *hi_i64,
                           double *lo_f64, double *hi_f64)
{
    if (!model || model->kind != CP_MODEL_SECONDARY) return cp_bound_stripe(A, K, model, lo_i64, hi_i64, lo_f64, hi_f64);
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && model_known(model) && K >= 1 && Pi && Pi->spl, CP_EINVAL, "bad argument");
        bool neg = model->dtype == CP_I64 ? (model->p_i64[1] < 0 || model->p_i64[2] < 0 || model->p_i64[3] < 0 || model->p_i64[4] < 0)
                                          : (model->p_f64[1] < 0 || model->p_f64[2] < 0 || model->p_f64[3] < 0 || model->p_f64[4] < 0);
        CP_REQUIRE(!neg, CP_EINVAL, "bound_stripe asserts beta >= 0");
        // the two bounds are the secondary cost with all nets local resp. all nets remote: evaluate the oracle on the empty and the
        // full column range of every part with (b_local, b_remote) = (0, 0) resp. (b_remote, b_remote)
        int64_t Kp = Pi->K;
        std::vector<int64_t> one((size_t)Kp, 1), ks((size_t)Kp);
        for (int64_t k = 0; k < Kp; k++) ks[(size_t)k] = k + 1;
        cp_model_t lo_m = *model, hi_m = *model;
        if (model->dtype == CP_I64) { lo_m.p_i64[3] = 0; lo_m.p_i64[4] = 0; hi_m.p_i64[3] = model->p_i64[4]; }
        else { lo_m.p_f64[3] = 0; lo_m.p_f64[4] = 0; hi_m.p_f64[3] = model->p_f64[4]; }
        lo_m.alpha_k = nullptr; lo_m.n_alpha_k = 0; hi_m.alpha_k = nullptr; hi_m.n_alpha_k = 0;
        if (model->dtype == CP_I64) {
            std::vector<int64_t> a((size_t)Kp), b((size_t)Kp);
            int32_t rc = cp_oracle_eval(A, &lo_m, Pi, CP_HINT_STEP, Kp, one.data(), one.data(), ks.data(), a.data(), nullptr);
            if (rc != CP_OK) return rc;
            rc = cp_oracle_eval(A, &hi_m, Pi, CP_HINT_STEP, Kp, one.data(), one.data(), ks.data(), b.data(), nullptr);
            if (rc != CP_OK) return rc;
            int64_t clo = 0, chi = 0;                                  // "c_lo = 0; c_hi = 0" (:44-45)
            for (int64_t k = 0; k < Kp; k++) { clo = std::max(clo, a[(size_t)k]); chi = std::max(chi, b[(size_t)k]); }
            *lo_i64 = clo; *hi_i64 = chi; *lo_f64 = (double)clo; *hi_f64 = (double)chi;
        } else {
            std::vector<double> a((size_t)Kp), b((size_t)Kp);
            int32_t rc = cp_oracle_eval(A, &lo_m, Pi, CP_HINT_STEP, Kp, one.data(), one.data(), ks.data(), nullptr, a.data());
            if (rc != CP_OK) return rc;
            rc = cp_oracle_eval(A, &hi_m, Pi, CP_HINT_STEP, Kp, one.data(), one.data(), ks.data(), nullptr, b.data());
            if (rc != CP_OK) return rc;
            double clo = 0, chi = 0;
            for (int64_t k = 0; k < Kp; k++) { clo = std::max(clo, a[(size_t)k]); chi = std::max(chi, b[(size_t)k]); }
            *lo_f64 = clo; *hi_f64 = chi;
        }
        return CP_OK;
    });
}

// ---- entry points whose device kernels land in later files; until then they refuse loudly ----
#define CP_TODO(msg) do { set_error(msg); return CP_EUNSUPPORTED; } while (0)

int32_t cp_link_array(cp_csr_t A, int64_t *out)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && out, CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        ensure_links(A);
        std::vector<int32_t> h((size_t)(A->N > 0 ? A->N : 1));
        CP_HIP(hipMemcpyAsync(h.data(), A->prev.p, sizeof(int32_t) * (size_t)A->N, hipMemcpyDeviceToHost, A->stream));
        CP_HIP(hipStreamSynchronize(A->stream));
        for (int64_t q = 0; q < A->N; q++) out[q] = (A->n + 1) - ((int64_t)h[q] + 1);   // idx'[q] = (n+1) - hst[i]
        return CP_OK;
    });
}

// ---- row-tiled DP across ranks (multi-GPU): see include/chainpart.h
int32_t cp_dp_begin(cp_csr_t A, int64_t K, int32_t combine, int32_t order, const cp_model_t *model, int64_t row_lo, int64_t row_hi,
                    cp_dp_t *out)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(A && out && model_known(model) && K >= 1, CP_EINVAL, "bad argument");
        CP_REQUIRE(row_lo >= 1 && row_hi >= row_lo && row_hi <= A->n + 2, CP_EINVAL, "row tile must satisfy 1 <= row_lo <= row_hi <= n+2");
        CP_REQUIRE(model->kind == CP_MODEL_WORK || model->kind == CP_MODEL_CONNECTIVITY || model->kind == CP_MODEL_HYPEREDGE_CUT ||
                       model->kind == CP_MODEL_COLBLOCK, CP_EUNSUPPORTED, "model kind has no device DP path");
        CP_HIP(hipSetDevice(A->device));
        DpBase *impl = nullptr;
        int32_t rc = model->dtype == CP_I64 ? dp_begin<int64_t>(A, K, combine, order, model, row_lo, row_hi, &impl)
                                            : dp_begin<double>(A, K, combine, order, model, row_lo, row_hi, &impl);
        if (rc != CP_OK) return rc;
        cp_dp_s *h = new cp_dp_s();
        h->dtype = model->dtype; h->impl = impl; h->A = A;
        *out = h;
        return CP_OK;
    });
}

int32_t cp_dp_layer(cp_dp_t dp, int64_t k, const void *cst_prev_device, void *cst_cur_device)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(dp && cst_cur_device && k >= 1, CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(dp->A->device));
        if (dp->dtype == CP_I64) return dp_layer<int64_t>(static_cast<DpRun<int64_t> *>(dp->impl), k, (const int64_t *)cst_prev_device, (int64_t *)cst_cur_device);
        return dp_layer<double>(static_cast<DpRun<double> *>(dp->impl), k, (const double *)cst_prev_device, (double *)cst_cur_device);
    });
}

int32_t cp_dp_ptr_at(cp_dp_t dp, int64_t k, int64_t jp, int64_t *out)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(dp && out && k >= 1 && jp >= 1 && jp <= dp->A->n + 1, CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(dp->A->device));
        int64_t rlo, rhi; DBuf<int32_t> *ptr; int64_t K;
        if (dp->dtype == CP_I64) { auto *D = static_cast<DpRun<int64_t> *>(dp->impl); ptr = &D->ptr; K = D->K; CP_REQUIRE(k <= K, CP_EINVAL, "bad layer");
                                   rlo = (size_t)k < D->lay_lo.size() ? D->lay_lo[(size_t)k] : D->rlo; rhi = (size_t)k < D->lay_hi.size() ? D->lay_hi[(size_t)k] : D->rhi; }
        else { auto *D = static_cast<DpRun<double> *>(dp->impl); ptr = &D->ptr; K = D->K; CP_REQUIRE(k <= K, CP_EINVAL, "bad layer");
               rlo = (size_t)k < D->lay_lo.size() ? D->lay_lo[(size_t)k] : D->rlo; rhi = (size_t)k < D->lay_hi.size() ? D->lay_hi[(size_t)k] : D->rhi; }
        *out = 0;
        if (k == 1) { *out = 1; return CP_OK; }                 // ptr[:, 1] == 1 on every rank
        int64_t r = jp - 1;
        if (r < rlo || r > rhi) return CP_OK;                   // another rank owns this row
        int32_t v = 0;
        CP_HIP(hipMemcpyAsync(&v, ptr->p + (size_t)(k - 1) * (size_t)(dp->A->n + 1) + (size_t)r, sizeof(int32_t), hipMemcpyDeviceToHost, dp->A->stream));
        CP_HIP(hipStreamSynchronize(dp->A->stream));
        *out = (int64_t)v + 1;
        return CP_OK;
    });
}

int32_t cp_dp_set_rows(cp_dp_t dp, int64_t row_lo, int64_t row_hi)
{
    if (!dp || row_lo < 1 || row_hi < row_lo || row_hi > dp->A->n + 2) return CP_EINVAL;
    if (dp->dtype == CP_I64) { auto *D = static_cast<DpRun<int64_t> *>(dp->impl); D->rlo = row_lo - 1; D->rhi = row_hi - 2; }
    else { auto *D = static_cast<DpRun<double> *>(dp->impl); D->rlo = row_lo - 1; D->rhi = row_hi - 2; }
    return CP_OK;
}

int32_t cp_dp_set_window(cp_dp_t dp, int64_t wmax)
{
    if (!dp || wmax < 0) return CP_EINVAL;
    if (dp->dtype == CP_I64) static_cast<DpRun<int64_t> *>(dp->impl)->wwin = wmax;
    else static_cast<DpRun<double> *>(dp->impl)->wwin = wmax;
    return CP_OK;
}

int32_t cp_dp_ptr_row(cp_dp_t dp, int64_t k, int64_t *out)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(dp && out && k >= 1, CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(dp->A->device));
        int64_t rlo, rhi; DBuf<int32_t> *ptr; int64_t K;
        if (dp->dtype == CP_I64) { auto *D = static_cast<DpRun<int64_t> *>(dp->impl); ptr = &D->ptr; K = D->K; CP_REQUIRE(k <= K, CP_EINVAL, "bad layer");
                                   rlo = (size_t)k < D->lay_lo.size() ? D->lay_lo[(size_t)k] : D->rlo; rhi = (size_t)k < D->lay_hi.size() ? D->lay_hi[(size_t)k] : D->rhi; }
        else { auto *D = static_cast<DpRun<double> *>(dp->impl); ptr = &D->ptr; K = D->K; CP_REQUIRE(k <= K, CP_EINVAL, "bad layer");
               rlo = (size_t)k < D->lay_lo.size() ? D->lay_lo[(size_t)k] : D->rlo; rhi = (size_t)k < D->lay_hi.size() ? D->lay_hi[(size_t)k] : D->rhi; }
        const int64_t n = dp->A->n;
        if (k == 1) { for (int64_t r = 0; r <= n; r++) out[r] = 1; return CP_OK; }
        std::vector<int32_t> h((size_t)n + 1);
        CP_HIP(hipMemcpyAsync(h.data(), ptr->p + (size_t)(k - 1) * (size_t)(n + 1), sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToHost, dp->A->stream));
        CP_HIP(hipStreamSynchronize(dp->A->stream));
        for (int64_t r = 0; r <= n; r++) out[r] = (r < rlo || r > rhi) ? 0 : (int64_t)h[(size_t)r] + 1;
        return CP_OK;
    });
}

int32_t cp_dp_block_tables(cp_dp_t dp, int32_t *nplanes_out, int64_t *opt_out, int64_t *nets_out, int64_t *selfnets_out)
{
    return guarded([&]() -> int32_t {
        CP_REQUIRE(dp && opt_out && nets_out, CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(dp->A->device));
        int nb = 0;
        if (dp->dtype == CP_I64) {
            auto *D = static_cast<DpRun<int64_t> *>(dp->impl);
            CP_REQUIRE(D->fast && D->work, CP_EUNSUPPORTED, "block tables exist on the O(n log^2 n) path only");
            nb = dp_total_block_tables<int64_t>(dp->A, D->work, opt_out, nets_out, selfnets_out);
        } else {
            auto *D = static_cast<DpRun<double> *>(dp->impl);
            CP_REQUIRE(D->fast && D->work, CP_EUNSUPPORTED, "block tables exist on the O(n log^2 n) path only");
            nb = dp_total_block_tables<double>(dp->A, D->work, opt_out, nets_out, selfnets_out);
        }
        if (nplanes_out) *nplanes_out = nb;
        return CP_OK;
    });
}

int32_t cp_dp_destroy(cp_dp_t dp)
{
    if (!dp) return CP_OK;
    (void)hipSetDevice(dp->A->device);
    delete dp->impl;
    delete dp;
    return CP_OK;
}

}  // extern "C"
