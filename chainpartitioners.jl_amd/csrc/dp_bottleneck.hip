// dp_bottleneck.hip -- one layer of the bottleneck (g = max) K-part DP in O(n) column steps, exact to the reference's literal
// O(n^2) sweep including ties (largest j wins):
//
//     cst[j',k] = min_{1<=j<=j'} max(cst[j,k-1], f(j,j',k))        /root/reference/src/DynamicSplitter.jl:7,33-46
//
// For a cost that grows with its part (every beta >= 0; bound_stripe asserts the same, ConnectivityCosts.jl:25-27) the previous
// layer W[p] = cst[p+1,k-1] is non-decreasing in p and f(p, r) is non-increasing in p, so h(p) = max(W[p], f(p, r)) is a valley:
// with the CROSSING c(r) = min{p <= r : W[p] >= f(p, r)} (r + 1 if there is none),
//     h(p) = f(p, r) below c (non-increasing),   h(p) = W[p] from c on (non-decreasing),
// the minimum is min(f(c-1, r), W[c]) and the LARGEST minimiser -- the reference scans j upwards with <= -- is
//     W[c] <= f(c-1, r):  the right end of W's run of equal values through c, capped at r;      otherwise  c - 1.
// (SURVEY.md section 7 sketches this search; executable check against the recurrence: tests/test_oracle_bottleneck.py.)
// The crossing is non-decreasing in r, so a chunk of consecutive rows is one two-pointer walk: moving r adds a column on the
// right of the part [c, r), moving c removes one on its left, both are one pass over the column's link entries
//     nets(c, r+1) = nets(c, r) + #{q in col r : prev[q] < c},      nets(c+1, r) = nets(c, r) - #{q in col c : next[q] >= r}
// (self nets of the hyperedge-cut cost: rows bucketed by last / first column).  2 N link entries per layer in total.  The walk
// of a chunk starts at its first row's crossing, found by a binary search over p with random-access counts from the wavelet
// counter (the reference's NetCount / SelfNetCount query, SparseColorArrays.jl:121-125, 225-229).
// Floating point: every term of the Work / Connectivity models is monotone in its count and IEEE addition is monotone, so the
// valley holds for their non-integral Float64 parameters as well; max() is exact.  NOT so for the hyperedge-cut cost, whose
// (d - l) * b_cut term shrinks while the part grows: its rounded sum can rise by an ulp against the real-number order, and
// fast_bottleneck_ok (capi.hip) admits it only with integer-valued parameters (exact arithmetic) -- the rest runs dp_brute.hip.
#include "csr.hpp"
#include "model.hpp"
#include "dp.hpp"
#include "wavelet.hpp"

namespace cpk {

template <typename TC>
struct BnCtx {
    DevModel<TC> M; TC alpha;
    int64_t n; int32_t hyp;
    const int64_t *pos, *lpos;
    const int32_t *pos32, *prev, *next, *fpos32, *flast, *lpos32, *lfirst;
    WaveletDev net, self;
    const TC *W;
};

template <typename TC>
__device__ __forceinline__ TC bn_cost(const BnCtx<TC> &C, int64_t p, int64_t r, int64_t nn, int64_t nl)
{
    return dm_apply(C.M, C.alpha, r - p, (int64_t)(C.pos32[r] - C.pos32[p]), nn, nl);
}

// runend[p] = last index of the run of equal values of W through p (W is non-decreasing: runs are contiguous)
//   pass 1: per block of 1024 rows, local answer or "continues beyond the block" (-1); pass 2: blocks resolved right to left by
//   one thread (n / 1024 steps); pass 3: fill
template <typename TC>
__global__ void __launch_bounds__(1024) k_bn_run1(int64_t n1, const TC *__restrict__ W, int32_t *__restrict__ runend, int32_t *__restrict__ blk_first_end)
{
    __shared__ int32_t s_end[1024];
    const int64_t base = (int64_t)blockIdx.x * 1024, p = base + threadIdx.x;
    // e[p] = p if the run ends at p (p is the last row or W[p+1] differs), else "unknown"
    int32_t e = INT32_MAX;
    if (p < n1) e = (p == n1 - 1 || W[p + 1] != W[p]) ? (int32_t)p : INT32_MAX;
    s_end[threadIdx.x] = e;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                    // suffix minimum inside the block
        int32_t v = threadIdx.x + o < 1024 ? s_end[threadIdx.x + o] : INT32_MAX;
        __syncthreads();
        if (v < s_end[threadIdx.x]) s_end[threadIdx.x] = v;
        __syncthreads();
    }
    if (p < n1) runend[p] = s_end[threadIdx.x];            // INT32_MAX: the run continues into the next block
    if (threadIdx.x == 0) blk_first_end[blockIdx.x] = s_end[0];
}
__global__ void __launch_bounds__(1024) k_bn_run2(int64_t nblk, int32_t *__restrict__ blk_first_end)
{
    // blk_first_end[b] = end of the run that starts (or continues) at the first row of block b: a suffix minimum over the blocks
    // (INT32_MAX = "continues"), one workgroup: every thread owns a contiguous share, suffix minima of the shares' minima, fill
    __shared__ int32_t sh[1024];
    const int64_t per = (nblk + 1023) / 1024, lo = (int64_t)threadIdx.x * per, hi = lo + per < nblk ? lo + per : nblk;
    int32_t m = INT32_MAX;
    for (int64_t b = hi - 1; b >= lo; b--) if (blk_first_end[b] < m) m = blk_first_end[b];
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int32_t v = threadIdx.x + o < 1024 ? sh[threadIdx.x + o] : INT32_MAX;
        __syncthreads();
        if (v < sh[threadIdx.x]) sh[threadIdx.x] = v;
        __syncthreads();
    }
    int32_t run = threadIdx.x + 1 < 1024 ? sh[threadIdx.x + 1] : INT32_MAX;      // minimum over the shares to the right
    for (int64_t b = hi - 1; b >= lo; b--) { if (blk_first_end[b] == INT32_MAX) blk_first_end[b] = run; else run = blk_first_end[b]; }
}
__global__ void __launch_bounds__(1024) k_bn_run3(int64_t n1, int64_t nblk, int32_t *__restrict__ runend, const int32_t *__restrict__ blk_first_end)
{
    const int64_t p = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (p < n1 && runend[p] == INT32_MAX) runend[p] = blockIdx.x + 1 < nblk ? blk_first_end[blockIdx.x + 1] : (int32_t)(n1 - 1);
}

// counts of the part [p, r) by random access
template <typename TC>
__device__ __forceinline__ void bn_counts(const BnCtx<TC> &C, int64_t p, int64_t r, int64_t &nn, int64_t &nl)
{
    nn = 0; nl = 0;
    if (p >= r) return;
    if (C.M.kind != CP_MODEL_WORK) nn = (C.pos[r] - C.pos[p]) - wt_count_le(C.net, C.n - p, C.pos[r]);
    if (C.hyp) nl = wt_count_le(C.self, C.n - p, C.lpos[r]);
}

// first row of every chunk: its crossing by binary search (the predicate W[p] >= f(p, r) is monotone in p).  The crossing is
// non-decreasing in the row, so the search runs coarse to fine: every 64th chunk start over the whole range first (stride 64,
// phase 0), then the others between the crossings of their coarse neighbours (phase 1) -- a handful of probes instead of log2(n).
template <typename TC>
__device__ __forceinline__ bool bn_pred(const BnCtx<TC> &C, int64_t p, int64_t r)      // W[p] >= f(p, r)   (p <= r)
{
    int64_t nn, nl;
    bn_counts(C, p, r, nn, nl);
    return C.W[p] >= bn_cost(C, p, r, nn, nl);
}

// hint (phase 0 only, may be null): the crossings of the same rows in the previous layer.  Layers converge, so the search gallops
// away from the hint -- two probes where the crossing did not move, 2 log2(distance) where it did -- instead of bisecting the
// whole range; any hint is safe (it only chooses where the search starts).
template <typename TC>
__global__ void __launch_bounds__(256) k_bn_starts(BnCtx<TC> C, int64_t rlo, int64_t rhi, int64_t CH, int64_t nchunk, int stride, int coarse,
                                                   int32_t *__restrict__ c0, int32_t *__restrict__ nn0, int32_t *__restrict__ nl0,
                                                   const int32_t *__restrict__ hint, const int32_t *__restrict__ hint2)
{
    // chunks t = 0, stride, 2 stride, ...; coarse != 0: those that are multiples of `coarse` are known already and bracket the rest
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (t >= nchunk || (coarse && t % coarse == 0)) return;
    const int64_t r = rlo + t * CH;
    int64_t lo = 0, hi = r + 1;                             // the answer lies in [lo, hi]; pred is true at hi (r + 1: "no crossing")
    if (coarse) {
        const int64_t tl = t - t % coarse, tr = tl + coarse;
        lo = c0[tl];                                        // c(r) >= c(left coarse row)
        if (tr < nchunk && (int64_t)c0[tr] < hi) hi = c0[tr];      // c(r) <= c(right coarse row)
        if (lo > hi) lo = hi;
    } else if (hint) {
        int64_t g = hint[t];
        if (hint2) g += g - (int64_t)hint2[t];                // (the crossing of the layer before as well: continue its move)
        if (g < 0) g = 0;
        if (g > r) g = r;
        if (bn_pred(C, g, r)) {                             // the crossing is at or left of g: gallop left
            hi = g;
            int64_t d = 1;
            while (hi - d >= 0 && bn_pred(C, hi - d, r)) { hi -= d; d <<= 1; }
            lo = hi - d >= 0 ? hi - d + 1 : 0;
        } else {                                            // right of g
            int64_t cur = g, d = 1;
            while (cur + d <= r && !bn_pred(C, cur + d, r)) { cur += d; d <<= 1; }
            lo = cur + 1;
            if (cur + d <= r) hi = cur + d;
        }
    }
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;                 // mid <= r
        if (bn_pred(C, mid, r)) hi = mid; else lo = mid + 1;
    }
    int64_t nn = 0, nl = 0;
    if (lo <= r) bn_counts(C, lo, r, nn, nl);
    c0[t] = (int32_t)lo; nn0[t] = (int32_t)nn; nl0[t] = (int32_t)nl;
}

// a lane's pass over one column: eight independent loads in flight (a loop of single loads pays one memory latency per entry --
// ~30 per row -- and that chain, not bandwidth or divergence, was the whole cost of the walk)
template <bool GE>
__device__ __forceinline__ int32_t bn_scan(const int32_t *__restrict__ arr, int32_t q0, int32_t q1, int32_t thr)
{
    int32_t c = 0;
    for (int32_t q = q0; q < q1; q += 8) {
        int32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = q + k < q1 ? arr[q + k] : (GE ? INT32_MIN : INT32_MAX);
#pragma unroll
        for (int k = 0; k < 8; k++) c += GE ? (v[k] >= thr) : (v[k] < thr);
    }
    return c;
}

// one lane per chunk: the two-pointer walk
// (Tried: the walk out of LDS -- the 64 chunks of a wave read two contiguous pieces of the link arrays, two of colptr and one of W,
//  fetched coalesced with everything in flight and kept as saturated 16-bit offsets, 38 KB per wave: 3.9 ms per layer against 2.1 ms.
//  The walks are chains of ~2600 dependent LDS instructions per wave, and at one wave per SIMD (LDS) nothing hides their latency;
//  eight waves per SIMD on global memory do better.)
// (Tried: the walk as a per-lane state machine -- one loop per lane consuming entries of whatever scan the lane is in, so that a
//  wave does not wait for its longest column at every scan: 1.5x slower; the cost was the serial loads, not the divergence.)
template <typename TC>
__global__ void __launch_bounds__(256) k_bn_walk(BnCtx<TC> C, int64_t rlo, int64_t rhi, int64_t CH, int64_t nchunk,
                                                 const int32_t *__restrict__ c0, const int32_t *__restrict__ nn0, const int32_t *__restrict__ nl0,
                                                 const int32_t *__restrict__ runend, TC *__restrict__ cst, int32_t *__restrict__ ptr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunk) return;
    const int32_t r0 = (int32_t)(rlo + t * CH);
    int32_t r1 = (int32_t)(r0 + CH - 1);
    if (r1 > rhi) r1 = (int32_t)rhi;
    int32_t c = c0[t], nn = nn0[t], nl = nl0[t];
    const bool nets = C.M.kind != CP_MODEL_WORK;
    for (int32_t r = r0; r <= r1; r++) {
        if (r > r0) {
            // column r - 1 joins the part [c, r - 1) on the right (an empty part stays empty when c == r)
            if (c <= r - 1) {
                if (nets) nn += bn_scan<false>(C.prev, C.pos32[r - 1], C.pos32[r], c);
                if (C.hyp) nl += bn_scan<true>(C.lfirst, C.lpos32[r - 1], C.lpos32[r], c);
            }
            // the crossing only moves right
            while (c <= r) {
                if (!(C.W[c] < bn_cost(C, c, r, nn, nl))) break;
                if (c < r) {                                // column c leaves on the left
                    if (nets) nn -= bn_scan<true>(C.next, C.pos32[c], C.pos32[c + 1], r);
                    if (C.hyp) nl -= bn_scan<false>(C.flast, C.fpos32[c], C.fpos32[c + 1], r);
                }
                c++;
            }
            if (c > r) { nn = 0; nl = 0; }
        }
        // f(c - 1, r): column c - 1 joins on the left
        bool have_fm = c >= 1;
        TC fm = (TC)0;
        if (have_fm) {
            int32_t a = 0, al = 0;
            if (c - 1 < r) {
                a = nn; al = nl;
                if (nets) a += bn_scan<true>(C.next, C.pos32[c - 1], C.pos32[c], r);
                if (C.hyp) al += bn_scan<false>(C.flast, C.fpos32[c - 1], C.fpos32[c], r);
            }
            fm = bn_cost(C, c - 1, r, a, al);
        }
        TC v; int32_t p;
        if (c <= r && (!have_fm || C.W[c] <= fm)) { v = C.W[c]; p = runend[c] < r ? runend[c] : r; }
        else { v = fm; p = c - 1; }
        cst[r] = v; ptr[r] = p;
    }
}

struct BnWork {
    bool have_net = false, have_self = false;
    WaveletHost net, self;
    DBuf<int32_t> runend, blk, c0, nn0, nl0, hint, hint2;
    int64_t hint_rlo = -1, hint_nchunk = -1, hint_ch = -1;      // the tiling the hints belong to (-1: none)
    int hint_layers = 0;                                        // how many consecutive layers of that tiling the hints cover (hint: last, hint2: the one before)
};

static BnWork *bn_work_get(cp_csr_s *A)
{
    if (!A->bn_work) {
        A->bn_work = new BnWork();
        A->bn_work_free_fn = [](void *w) { delete reinterpret_cast<BnWork *>(w); };
        A->bn_work_reset_fn = [](void *w) { auto *B = reinterpret_cast<BnWork *>(w); B->have_net = false; B->have_self = false; B->hint_nchunk = -1; B->hint_layers = 0; };   // (cp_csr_reset_cache)
    }
    return reinterpret_cast<BnWork *>(A->bn_work);
}

int64_t g_opt_bn_chunk = 8;       // (config 3 matrix, K = 64: 8 rows 0.50 s, 32 rows 0.58 s, 128 rows 0.97 s, 256 rows 1.32 s per partition)

template <typename TC>
void dp_bottleneck_layer(cp_csr_s *A, const DevModel<TC> &M, TC alpha, const TC *W, TC *cst_out, int32_t *ptr_out,
                         int64_t rlo, int64_t rhi)
{
    hipStream_t s = A->stream;
    const int64_t n = A->n, n1 = n + 1;
    if (rhi < rlo) return;
    BnWork *B = bn_work_get(A);
    const bool hyp = M.kind == CP_MODEL_HYPEREDGE_CUT;
    ensure_links(A);
    if (hyp) ensure_self(A);
    if (M.kind != CP_MODEL_WORK && !B->have_net) { ProfScope ps(PROF_WAVELET, s, 0.0); ensure_net_counter(A, B->net); B->have_net = true; }
    if (hyp && !B->have_self) { ProfScope ps(PROF_WAVELET, s, 0.0); ensure_selfnet_counter(A, B->self); B->have_self = true; }
    BnCtx<TC> C;
    C.M = M; C.alpha = alpha; C.n = n; C.hyp = hyp ? 1 : 0;
    C.pos = A->pos.p; C.lpos = hyp ? A->lpos.p : nullptr;
    C.pos32 = A->pos32.p; C.prev = A->prev.p; C.next = A->next.p;
    C.fpos32 = hyp ? A->fpos32.p : nullptr; C.flast = hyp ? A->flast.p : nullptr;
    C.lpos32 = hyp ? A->lpos32.p : nullptr; C.lfirst = hyp ? A->lfirst.p : nullptr;
    C.net = B->net.d; C.self = B->self.d; C.W = W;
    const int64_t nblk = cdiv(n1, 1024);
    B->runend.ensure((size_t)n1); B->blk.ensure((size_t)nblk + 1);
    const int64_t CH = std::max<int64_t>(1, g_opt_bn_chunk), nchunk = cdiv(rhi - rlo + 1, CH);
    B->c0.ensure((size_t)nchunk); B->nn0.ensure((size_t)nchunk); B->nl0.ensure((size_t)nchunk);
    ProfScope ps(PROF_BRUTE, s, 8.0 * (double)A->N + 24.0 * (double)(rhi - rlo + 1));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_run1<TC>), dim3((unsigned)nblk), dim3(1024), 0, s, n1, W, B->runend.p, B->blk.p);
    hipLaunchKernelGGL(k_bn_run2, dim3(1), dim3(1024), 0, s, nblk, B->blk.p);
    hipLaunchKernelGGL(k_bn_run3, dim3((unsigned)nblk), dim3(1024), 0, s, n1, nblk, B->runend.p, B->blk.p);
    const bool hinted = B->hint_nchunk == nchunk && B->hint_rlo == rlo && B->hint_ch == CH && nchunk > 1;
    if (!hinted) B->hint_layers = 0;
    B->hint.ensure((size_t)nchunk); B->hint2.ensure((size_t)nchunk);
    // every 64th chunk from the previous layer's hint, the rest bracketed by those.  Tried, config 3 matrix, K = 64 (302 ms as is):
    //  - a level of every 8th chunk in between: 6 probes of the counter for most chunks instead of 9, but a third chain of
    //    dependent probes: 318 ms;
    //  - the coarse starts by one WAVE per row, a 64-ary search over the whole range (4 rounds instead of ~25 dependent probes):
    //    64 lanes x 24 levels x 5 loads of different cache lines per round keep the L1 busy for longer than the chain took
    //    (1.58 ms per layer against 0.9 ms);
    //  - the fine starts as a 4-ary search, three interleaved descents per step: 1.0 ms against 0.9 ms -- 1.25 M lanes are bound by
    //    the number of line requests, not by the length of the chain.
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_starts<TC>), dim3((unsigned)cdiv(cdiv(nchunk, 64), 256)), dim3(256), 0, s, C, rlo, rhi, CH, nchunk, 64, 0, B->c0.p, B->nn0.p,
                       B->nl0.p, hinted ? B->hint.p : (const int32_t *)nullptr, hinted && B->hint_layers >= 2 && !(g_opt_dbg & 4194304) ? B->hint2.p : (const int32_t *)nullptr);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_starts<TC>), dim3((unsigned)cdiv(nchunk, 256)), dim3(256), 0, s, C, rlo, rhi, CH, nchunk, 1, 64, B->c0.p, B->nn0.p, B->nl0.p,
                       (const int32_t *)nullptr, (const int32_t *)nullptr);
    std::swap(B->hint.p, B->hint2.p); std::swap(B->hint.n, B->hint2.n);      // (hint2 <- the previous layer's crossings)
    CP_HIP(hipMemcpyAsync(B->hint.p, B->c0.p, sizeof(int32_t) * (size_t)nchunk, hipMemcpyDeviceToDevice, s));
    B->hint_layers++;
    B->hint_nchunk = nchunk; B->hint_rlo = rlo; B->hint_ch = CH;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_walk<TC>), dim3((unsigned)cdiv(nchunk, 256)), dim3(256), 0, s, C, rlo, rhi, CH, nchunk, B->c0.p, B->nn0.p, B->nl0.p,
                       B->runend.p, cst_out, ptr_out);
    CP_HIP(hipGetLastError());
}

template void dp_bottleneck_layer<int64_t>(cp_csr_s *, const DevModel<int64_t> &, int64_t, const int64_t *, int64_t *, int32_t *, int64_t, int64_t);
template void dp_bottleneck_layer<double>(cp_csr_s *, const DevModel<double> &, double, const double *, double *, int32_t *, int64_t, int64_t);

}  // namespace cpk
