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

__device__ __forceinline__ int32_t rdl_bn(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int64_t rdl64_bn(int64_t v, int src)
{
    return ((int64_t)__builtin_amdgcn_readlane((int)(v >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(v & 0xffffffffll), src);
}
__device__ __forceinline__ double rdl64_bn(double v, int src) { return __longlong_as_double((long long)rdl64_bn((int64_t)__double_as_longlong(v), src)); }

__device__ __forceinline__ int64_t shfl64_bn(int64_t v, int src)
{
    int lo = __shfl((int)(v & 0xffffffffll), src), hi = __shfl((int)(v >> 32), src);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

template <typename TC>
struct BnCtx {
    DevModel<TC> M; TC alpha;
    int64_t n; int32_t hyp;
    int64_t N, NF;                       // lengths of the link arrays (next / prev) and of the row buckets (flast / lfirst)
    // candidate limits of a weight-constrained layer (DynamicSplitter.jl:233-246; 0-based): row r takes max(p_lo0, j0(r)) <= p <=
    // min(r, p_hi0) with j0(r) = r - wwin (width weights) or j0[r] (any monotone weight: the first column whose part up to r fits);
    // unconstrained: wwin = 0, j0 = null (no lower cut), p_lo0 = 0, p_hi0 = n
    int64_t wwin, p_lo0, p_hi0;
    const int32_t *j0;
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

template <typename TC> __device__ __forceinline__ int64_t bn_plo(const BnCtx<TC> &C, int64_t r)
{
    int64_t lo = C.p_lo0;
    if (C.j0) { const int64_t j = C.j0[r]; if (j > lo) lo = j; }
    else if (C.wwin > 0 && r - C.wwin > lo) lo = r - C.wwin;
    return lo;
}
template <typename TC> __device__ __forceinline__ int64_t bn_phi(const BnCtx<TC> &C, int64_t r) { return r < C.p_hi0 ? r : C.p_hi0; }

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
                                                   const int32_t *__restrict__ hint, const int32_t *__restrict__ hint2, int32_t slack, int64_t rmin)
{
    // slack >= 0 (the wave-per-run walks): ANY column at or left of the crossing will do as a start -- the walk searches forward
    // from it -- so the gallop starts with steps of `slack` columns and the bisection stops at a bracket of `slack` columns: a
    // third of the probes of an exact search.  slack < 0: the exact crossing.
    // chunks t = 0, stride, 2 stride, ...; coarse != 0: those that are multiples of `coarse` are known already and bracket the rest
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (t >= nchunk || (coarse && t % coarse == 0)) return;
    const int64_t r = rlo + t * CH > rmin ? rlo + t * CH : rmin;      // (rmin > rlo: a tiling aligned to multiples of CH whose first chunk starts inside it)
    const int64_t lo0 = bn_plo(C, r), phi = bn_phi(C, r);   // the row's candidates (the whole range [0, r] without a window)
    int64_t lo = lo0, hi = phi + 1;                         // the answer lies in [lo, hi]; pred counts as true at hi (phi + 1: "no crossing")
    if (lo > hi) lo = hi;
    if (coarse) {
        const int64_t tl = t - t % coarse, tr = tl + coarse;
        if ((int64_t)c0[tl] > lo) lo = c0[tl];              // c(r) >= c(left coarse row)
        if (tr < nchunk && (int64_t)c0[tr] < hi) hi = c0[tr];      // c(r) <= c(right coarse row)
        if (lo > hi) lo = hi;
    } else if (hint) {
        int64_t g = hint[t];
        if (hint2) {
            // the crossing of the layer before as well: continue its move.  The last part of a prefix shrinks like 1 / k with the number
            // of parts k (s_k = r - c_k(r) ~ r / k): rho = s_k / s_(k-1) = (k - 1) / k gives s_(k+1) = s_k k / (k + 1) = s_k / (2 - rho)
            // without knowing k -- a linear continuation is off by thousands of columns in the first dozen layers (a dozen probes of 24
            // dependent levels each per start), this one by a few hundred.  Only a hint: any value is safe.
            const int64_t h2 = hint2[t], s1 = r - g, s2 = r - h2;
            if (s1 > 0 && s2 >= s1) {
                const double rho = (double)s1 / (double)s2;
                g = r - (int64_t)((double)s1 / (2.0 - rho));
            } else g += g - h2;
        }
        if (g < lo0) g = lo0;
        if (g > phi) g = phi;
        const int64_t d0 = slack > 0 ? slack : 1;          // first gallop step
        if (g >= lo0 && bn_pred(C, g, r)) {                 // the crossing is at or left of g: gallop left
            hi = g;
            int64_t d = d0;
            while (hi - d >= lo0 && bn_pred(C, hi - d, r)) { hi -= d; d <<= 1; }
            lo = hi - d >= lo0 ? hi - d + 1 : lo0;
        } else if (g >= lo0) {                              // right of g
            int64_t cur = g, d = d0;
            while (cur + d <= phi && !bn_pred(C, cur + d, r)) { cur += d; d <<= 1; }
            lo = cur + 1;
            if (cur + d <= phi) hi = cur + d;
        }
    }
    // slack >= 0: the bracket [lo, hi] of the crossing is narrowed to `slack` columns only -- lo, a column at or left of the
    // crossing, is all the wave walks need
    const int64_t tol = slack >= 0 ? slack : 0;
    while (hi - lo > tol) {
        const int64_t mid = (lo + hi) >> 1;                 // mid <= phi
        if (bn_pred(C, mid, r)) hi = mid; else lo = mid + 1;
    }
    // the wave walks take a column at or left of the crossing WITH the counts of its part: "no crossing" (lo == phi + 1) becomes the
    // last candidate itself (the lane-per-chunk walk keeps the exact convention, slack < 0)
    if (slack >= 0 && lo > phi) lo = phi;
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


// ------------------------------------------------------------------ the walk, wave per run of rows (lane = row)
// The two-pointer walk above gives every LANE eight rows: three dependent column scans per row, every one a gather of its own
// cache lines (one address per lane and cycle through the texture path), 0.03 of the bandwidth the 2 N link entries would need.
// Here a WAVE owns a run of rows and walks them 64 at a time, lane i <-> row a + i of the sub-run anchored at row a:
//   * every row of the sub-run has its crossing at or right of the anchor's crossing cs (the crossing is monotone), so all 64
//     rows start from the SAME column:  nets(cs, r) = nets(cs, a) + #{q in cols [a, r) : prev[q] < cs}  -- the right parts of all
//     rows from ONE coalesced pass over the sub-run's own link entries (a compare into a lane mask per 64 entries, every row-lane
//     counts the mask bits in front of its column's end);
//   * then the rows advance in LOCKSTEP over the columns x = cs, cs + 1, ..: a row still looking for its crossing tests
//     W[x] < f(x, r), and if so drops column x from its part: nets -= #{q in col x : next[q] >= r}.  Column x is the same for all
//     rows: its entries come from an LDS ring filled by coalesced loads running ahead, and the count is the same for every row
//     except for the entries whose next column falls inside the sub-run ("special": counted by the rows up to it) -- one scalar
//     popcount plus a short loop over the specials.  Rows drop out as they find their crossing (from the lowest row upwards);
//     the loop ends with the last row's crossing: about 64 steps for 64 rows, each a dozen vector instructions;
//   * f(c - 1, r), needed for the value, is the cost the row tested one step before it stopped (or, for a row that never
//     moved, one extra count over column cs - 1 at the start).
// The next sub-run is anchored at this one's last row (its crossing and counts are in lane 63): CH rows per wave need ONE start
// from the binary search over the wavelet counter instead of CH / 8.
constexpr int BN_RING = 512;      // link entries per LDS ring (power of two), refilled 256 at a time

template <typename TC, bool HYP>
__global__ void __launch_bounds__(256) k_bn_walk64(BnCtx<TC> C, int64_t rlo, int64_t rhi, int64_t CH, int64_t nchunk,
                                                   const int32_t *__restrict__ c0, const int32_t *__restrict__ nn0, const int32_t *__restrict__ nl0,
                                                   const int32_t *__restrict__ runend, TC *__restrict__ cst, int32_t *__restrict__ ptr, int32_t *__restrict__ hint_out)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t t = (int64_t)blockIdx.x * 4 + wv;
    if (t >= nchunk) return;                                             // (wave-uniform)
    __shared__ int32_t s_ring_all[4][HYP ? 2 : 1][BN_RING];
    int32_t(*s_ring)[BN_RING] = s_ring_all[wv];
    const bool nets = C.M.kind != CP_MODEL_WORK;
    const int32_t n = (int32_t)C.n;
    const int32_t r_begin = (int32_t)(rlo + t * CH);
    int32_t r_end = (int32_t)(r_begin + CH - 1);
    if (r_end > rhi) r_end = (int32_t)rhi;
    int32_t cs = c0[t], nn_a = nn0[t], nl_a = nl0[t];
    for (int32_t a = r_begin;;) {
        if (cs > a) { cs = a; nn_a = 0; nl_a = 0; }                      // no crossing at the anchor: start from the empty part [a, a)
        const int32_t r = a + lane;
        const bool valid = r <= r_end;
        const int32_t rq = valid ? r : r_end;
        const int32_t posr = C.pos32[rq];
        const int32_t lposr = HYP ? C.lpos32[rq] : 0;
        const int32_t alast = a + 63 < r_end ? a + 63 : r_end;           // last row of the sub-run
        // ---- right parts of all rows: columns [a, r) joining a part that starts at cs
        int32_t nn = nn_a, nl = nl_a;
        if (nets) {
            const int32_t q0 = __builtin_amdgcn_readfirstlane(posr), q1 = C.pos32[alast];
            for (int32_t base0 = q0; base0 < q1; base0 += 256) {
                int32_t v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int32_t q = base0 + 64 * k + lane; v[k] = q < q1 ? C.prev[q] : INT32_MAX; }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int32_t w = posr - (base0 + 64 * k);
                    const uint64_t lm = w <= 0 ? 0ull : (w >= 64 ? ~0ull : ((1ull << w) - 1ull));
                    nn += (int32_t)__popcll(__ballot(v[k] < cs) & lm);
                }
            }
        }
        if (HYP) {
            const int32_t q0 = __builtin_amdgcn_readfirstlane(lposr), q1 = C.lpos32[alast];
            for (int32_t base = q0; base < q1; base += 64) {
                const int32_t q = base + lane;
                const int32_t v = q < q1 ? C.lfirst[q] : INT32_MIN;
                const int32_t w = lposr - base;
                const uint64_t lm = w <= 0 ? 0ull : (w >= 64 ? ~0ull : ((1ull << w) - 1ull));
                nl += (int32_t)__popcll(__ballot(v >= cs) & lm);
            }
        }
        // ---- lockstep over the columns x = cs - 1 (value f(cs - 1, r) only), cs, cs + 1, ...
        // Int64 costs: f(x, r) = [alpha + r b_v + pos[r] b_p + net terms] - [x b_v + pos[x] b_p] -- the first bracket lives in the row's
        // lane and changes only when the row drops a column, the second is one scalar per step: the test W[x] < f(x, r) is ONE vector
        // compare (integer arithmetic: exact, the comparison is the reference's).  Float64 costs are evaluated term by term in
        // the reference's order (non-integral parameters round differently under any regrouping).
        constexpr bool EXACT = CostTraits<TC>::is_int;
        // (the model's coefficients once, outside the step loop: dm_apply's dispatch on the model kind would be re-done per step)
        const TC kV = C.M.p[CP_P_VERTEX], kP = C.M.p[CP_P_PIN];
        const TC kN = C.M.kind == CP_MODEL_CONNECTIVITY ? C.M.p[CP_P_NET] : (C.M.kind == CP_MODEL_HYPEREDGE_CUT ? C.M.p[CP_P_CUT_NET] : (TC)0);
        const TC kL = C.M.kind == CP_MODEL_HYPEREDGE_CUT ? csub(C.M.p[CP_P_SELF_NET], C.M.p[CP_P_CUT_NET]) : (TC)0;
        int32_t c = cs;
        bool active = valid, have_fm = false;
        TC fm = (TC)0;
        TC rhs = EXACT ? dm_apply(C.M, C.alpha, (int64_t)r, (int64_t)posr, (int64_t)nn, (int64_t)nl) : (TC)0;
        // rings of link entries: [.., whi) of `next` (and of the rows bucketed by first column) are in LDS
        const int32_t xs = cs >= 1 ? cs - 1 : 0;
        int32_t whi = C.pos32[xs], whi2 = HYP ? C.fpos32[xs] : 0;
        int32_t xb = xs - 64;                                            // window of W / pos over the columns xb .. xb + 63 (lane i <-> xb + i)
        TC Ww = (TC)0; int32_t pw = 0, pw1 = 0, fw = 0, fw1 = 0;
        for (int32_t x = xs;; x++) {
            if (x >= xb + 64) {
                xb = x;
                const int32_t xi = xb + lane <= n ? xb + lane : n, xi1 = xi + 1 <= n ? xi + 1 : n;
                Ww = C.W[xi]; pw = C.pos32[xi]; pw1 = C.pos32[xi1];
                if (HYP) { fw = C.fpos32[xi]; fw1 = C.fpos32[xi1]; }
            }
            const int xl = x - xb;
            const int32_t posx = rdl_bn(pw, xl), posx1 = rdl_bn(pw1, xl);
            const TC Wx = rdl64_bn(Ww, xl);
            const TC colterm = EXACT ? cadd(cmulc((int64_t)x, kV), cmulc((int64_t)posx, kP)) : (TC)0;      // (scalar)
            const bool pre = x < cs;                                     // the extra column cs - 1: counts only, nobody moves
            // cost of the part [x, r) for every row still walking
            bool adv = false;
            if (!pre) {
                TC tv;
                if (EXACT) { adv = active && (cadd(Wx, colterm) < rhs); tv = csub(rhs, colterm); }
                else { tv = dm_apply(C.M, C.alpha, (int64_t)(r - x), (int64_t)(posr - posx), (int64_t)nn, (int64_t)nl); adv = active && (Wx < tv); }
                if (active && !adv) { c = x; active = false; }           // the crossing
                if (adv) { fm = tv; have_fm = true; }
            }
            const uint64_t need = __ballot(pre ? valid : (adv && x < r));
            if (!pre && !__ballot(adv)) break;                           // every row has its crossing
            if (need) {
                // #{q in col x : next[q] >= r} per row: the same for all rows except the entries that end inside the sub-run
                int32_t cnt = 0, cnt2 = 0;
                if (nets) {
                    for (int32_t q = posx; q < posx1; q += 64) {
                        const int32_t pe = q + 64 < posx1 ? q + 64 : posx1;
                        if (whi < q) whi = q;                            // (columns nobody stepped over)
                        while (whi < pe) {                               // refill: 256 entries, nothing unread is overwritten (whi - q < 64)
#pragma unroll
                            for (int k = 0; k < 4; k++) { const int32_t i = whi + 64 * k + lane; s_ring[0][i & (BN_RING - 1)] = i < C.N ? C.next[i] : 0; }
                            whi += 256;
                            __threadfence_block();
                        }
                        const int32_t e = s_ring[0][(q + lane) & (BN_RING - 1)];
                        const bool in = q + lane < pe;
                        cnt += (int32_t)__popcll(__ballot(in && e >= a + 63));
                        uint64_t ms = __ballot(in && e >= a && e < a + 63);
                        while (ms) {
                            const int u = __ffsll((unsigned long long)ms) - 1;
                            ms &= ms - 1;
                            cnt += (rdl_bn(e, u) >= r);
                        }
                    }
                }
                if (HYP) {
                    const int32_t fx = rdl_bn(fw, xl), fx1 = rdl_bn(fw1, xl);
                    for (int32_t q = fx; q < fx1; q += 64) {
                        const int32_t pe = q + 64 < fx1 ? q + 64 : fx1;
                        if (whi2 < q) whi2 = q;
                        while (whi2 < pe) {
#pragma unroll
                            for (int k = 0; k < 4; k++) { const int32_t i = whi2 + 64 * k + lane; s_ring[HYP ? 1 : 0][i & (BN_RING - 1)] = i < C.NF ? C.flast[i] : 0; }
                            whi2 += 256;
                            __threadfence_block();
                        }
                        const int32_t e = s_ring[HYP ? 1 : 0][(q + lane) & (BN_RING - 1)];
                        const bool in = q + lane < pe;
                        cnt2 += (int32_t)__popcll(__ballot(in && e < a));             // rows ending before every row of the sub-run
                        uint64_t ms = __ballot(in && e >= a && e < a + 63);
                        while (ms) {
                            const int u = __ffsll((unsigned long long)ms) - 1;
                            ms &= ms - 1;
                            cnt2 += (rdl_bn(e, u) < r);
                        }
                    }
                }
                if (pre) {
                    if (valid) {
                        if (EXACT) fm = csub(cadd(rhs, cadd(cmulc((int64_t)cnt, kN), cmulc((int64_t)cnt2, kL))), colterm);
                        else fm = dm_apply(C.M, C.alpha, (int64_t)(r - x), (int64_t)(posr - posx), (int64_t)(nn + cnt), (int64_t)(nl + cnt2));
                        have_fm = true;
                    }
                } else if (adv && x < r) {
                    nn -= cnt; nl -= cnt2;
                    if (EXACT) rhs = csub(rhs, cadd(cmulc((int64_t)cnt, kN), cmulc((int64_t)cnt2, kL)));
                }
            }
            if (adv) {
                c = x + 1;
                if (x + 1 > r) { active = false; nn = 0; nl = 0; }       // no crossing: c = r + 1, the empty part
            }
        }
        // ---- values: min(f(c - 1, r), W[c]); the largest minimiser
        if (valid) {
            TC v; int32_t p;
            const TC Wc = c <= r ? C.W[c] : (TC)0;
            if (c <= r && (!have_fm || Wc <= fm)) { const int32_t re = runend[c]; v = Wc; p = re < r ? re : r; }
            else { v = fm; p = c - 1; }
            cst[r] = v; ptr[r] = p;
        }
        if (a == r_begin && lane == 0) hint_out[t] = c;                  // the next layer's starts gallop from this run's true first crossing
        if (alast >= r_end) break;
        // the next sub-run is anchored at this one's last row
        a = alast;
        cs = rdl_bn(c, 63); nn_a = rdl_bn(nn, 63); nl_a = rdl_bn(nl, 63);
    }
}

// ------------------------------------------------------------------ the walk, vectorised (Int64 costs)
// k_bn_walk64 walks its 64 rows in lockstep over the columns: one step per column, and every step is mostly SCALAR work (loop
// control, lane masks, readlanes) -- the scalar unit is shared by the four SIMDs of a CU, and 70 scalar instructions per row made
// it the bottleneck (profiles/r03_pmc_bn_walk.txt).  For Int64 costs the crossing can be SEARCHED instead:
//   f(x, r) = [alpha + r b_v + pos[r] b_p] - [x b_v + pos[x] b_p] + k_N nets(x, r) + k_L selfnets(x, r)       (k_N, k_L: the net terms)
//   nets(x_j, r) = nets(x_0, r) - P(j) - S(j, r)    over a window of 64 columns x_j = xw + j (lane j):
//       P(j) = #{entries of the columns x_0 .. x_(j-1) whose next column lies right of the whole sub-run}  (the same for all rows:
//              per-column counts from coalesced passes over the window's entries, then a wave prefix sum),
//       S(j, r) = the few "special" entries ending INSIDE the sub-run that row r counts (a short list in LDS).
//   W[x_j] >= f(x_j, r)  <=>  G(j) + k_N S(j, r) + k_L S_L(j, r) >= T(r),   G(j) = W[x_j] + x_j b_v + pos[x_j] b_p + k_N P(j) + k_L P_L(j)
// with G non-decreasing in j (W is, the betas are >= 0: fast_bottleneck_ok) -- every row finds its crossing by a 7-step binary
// search over the lanes (ds_bpermute gathers of G), all 64 rows at once; rows whose crossing lies beyond the window go on to the
// next 64 columns.  A few hundred vector instructions and hardly any scalar ones per 64 rows instead of ~6000.  Exactness: integer
// arithmetic -- the regrouped comparison is the reference's comparison (no rounding; magnitudes far below 2^63).
constexpr int BN_SPEC = 256;      // specials of a window kept in LDS per list; more: the rows count them from the entries themselves

template <bool HYP>
__global__ void __launch_bounds__(256) k_bn_walk_vec(BnCtx<int64_t> C, int64_t rlo, int64_t rhi, int64_t CH, int64_t nchunk,
                                                     const int32_t *__restrict__ c0, const int32_t *__restrict__ nn0, const int32_t *__restrict__ nl0,
                                                     const int32_t *__restrict__ runend, int64_t *__restrict__ cst, int32_t *__restrict__ ptr, int32_t *__restrict__ hint_out,
                                                     int64_t rmin)
{
    typedef int64_t TC;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t t = (int64_t)blockIdx.x * 4 + wv;
    if (t >= nchunk) return;                                             // (wave-uniform)
    __shared__ int32_t s_spec_all[4][HYP ? 2 : 1][BN_SPEC];
    int32_t(*s_spec)[BN_SPEC] = s_spec_all[wv];
    const bool nets = C.M.kind != CP_MODEL_WORK;
    const int32_t n = (int32_t)C.n;
    const TC kV = C.M.p[CP_P_VERTEX], kP = C.M.p[CP_P_PIN];
    const TC kN = C.M.kind == CP_MODEL_CONNECTIVITY ? C.M.p[CP_P_NET] : (C.M.kind == CP_MODEL_HYPEREDGE_CUT ? C.M.p[CP_P_CUT_NET] : (TC)0);
    const TC kL = C.M.kind == CP_MODEL_HYPEREDGE_CUT ? csub(C.M.p[CP_P_SELF_NET], C.M.p[CP_P_CUT_NET]) : (TC)0;
    const int32_t r_begin = (int32_t)(rlo + t * CH > rmin ? rlo + t * CH : rmin);
    int32_t r_end = (int32_t)(rlo + (t + 1) * CH - 1);
    if (r_end > rhi) r_end = (int32_t)rhi;
    int32_t cs = c0[t], nn_a = nn0[t], nl_a = nl0[t];
    for (int32_t a = r_begin;;) {
        if (cs > a) { cs = a; nn_a = 0; nl_a = 0; }                      // no crossing at the anchor: start from the empty part [a, a)
        const int32_t r = a + lane;
        const bool valid = r <= r_end;
        const int32_t rq = valid ? r : r_end;
        const int32_t posr = C.pos32[rq];
        const int32_t lposr = HYP ? C.lpos32[rq] : 0;
        const int32_t alast = a + 63 < r_end ? a + 63 : r_end;
        // ---- right parts of all rows: nn = nets(cs, r), nl = selfnets(cs, r)
        int32_t nn = nn_a, nl = nl_a;
        if (nets) {
            const int32_t q0 = __builtin_amdgcn_readfirstlane(posr), q1 = C.pos32[alast];
            for (int32_t base0 = q0; base0 < q1; base0 += 256) {
                int32_t v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int32_t q = base0 + 64 * k + lane; v[k] = q < q1 ? C.prev[q] : INT32_MAX; }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int32_t w = posr - (base0 + 64 * k);
                    const uint64_t lm = w <= 0 ? 0ull : (w >= 64 ? ~0ull : ((1ull << w) - 1ull));
                    nn += (int32_t)__popcll(__ballot(v[k] < cs) & lm);
                }
            }
        }
        if (HYP) {
            const int32_t q0 = __builtin_amdgcn_readfirstlane(lposr), q1 = C.lpos32[alast];
            for (int32_t base = q0; base < q1; base += 64) {
                const int32_t q = base + lane;
                const int32_t v = q < q1 ? C.lfirst[q] : INT32_MIN;
                const int32_t w = lposr - base;
                const uint64_t lm = w <= 0 ? 0ull : (w >= 64 ? ~0ull : ((1ull << w) - 1ull));
                nl += (int32_t)__popcll(__ballot(v >= cs) & lm);
            }
        }
        const TC rowpart = cadd(cadd(C.alpha, cmulc((int64_t)r, kV)), cmulc((int64_t)posr, kP));
        const int32_t plo_r = (int32_t)bn_plo(C, (int64_t)rq), phi_r = (int32_t)bn_phi(C, (int64_t)rq);     // this row's candidates [plo_r, phi_r]
        // ---- windows of 64 columns from cs - 1 on
        int32_t c = cs, c_carry = cs;
        bool open = valid, have_fm = false, have_prev = false;
        TC fm = (TC)0, fprev = (TC)0;
        int jb = cs >= 1 ? 1 : 0;                                        // lane of the column the counts nn / nl refer to
        for (int32_t xw = cs - jb;; xw += 64) {
            const int32_t xi = xw + lane <= n ? xw + lane : n, xi1 = xi + 1 <= n ? xi + 1 : n;
            const TC Wj = C.W[xi];
            const int32_t pj = C.pos32[xi], pj1 = C.pos32[xi1];
            int32_t fj = 0, fj1 = 0;
            if (HYP) { fj = C.fpos32[xi]; fj1 = C.fpos32[xi1]; }
            // per column: entries every row of the sub-run counts; the specials go to the lists
            int32_t nb = 0, nb2 = 0, ns = 0, ns2 = 0;
            const int32_t e_lo = rdl_bn(pj, 0), e_hi = rdl_bn(pj1, 63);
            if (nets) {
                for (int32_t base0 = e_lo; base0 < e_hi; base0 += 256) {
                    int32_t e[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { const int32_t q = base0 + 64 * k + lane; e[k] = q < e_hi ? C.next[q] : INT32_MIN; }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int32_t base = base0 + 64 * k;
                        if (base < e_hi) {                               // (wave-uniform)
                            const uint64_t mb = __ballot(e[k] >= a + 63);
                            const int32_t w0 = pj - base, w1 = pj1 - base;
                            const uint64_t m0 = w0 <= 0 ? 0ull : (w0 >= 64 ? ~0ull : ((1ull << w0) - 1ull));
                            const uint64_t m1 = w1 <= 0 ? 0ull : (w1 >= 64 ? ~0ull : ((1ull << w1) - 1ull));
                            nb += (int32_t)__popcll(mb & m1 & ~m0);
                            uint64_t ms = __ballot(e[k] >= a && e[k] < a + 63);
                            while (ms) {
                                const int u = __ffsll((unsigned long long)ms) - 1;
                                ms &= ms - 1;
                                const int32_t je = (int32_t)__popcll(__ballot(pj1 <= base + u));      // the window lane of the entry's column
                                if (lane == 0 && ns < BN_SPEC) s_spec[0][ns] = (je << 8) | (rdl_bn(e[k], u) - a);
                                ns++;
                            }
                        }
                    }
                }
            }
            if (HYP) {
                const int32_t g_lo = rdl_bn(fj, 0), g_hi = rdl_bn(fj1, 63);
                for (int32_t base = g_lo; base < g_hi; base += 64) {
                    const int32_t q = base + lane;
                    const int32_t e = q < g_hi ? C.flast[q] : INT32_MAX;
                    const uint64_t mb = __ballot(e < a);
                    const int32_t w0 = fj - base, w1 = fj1 - base;
                    const uint64_t m0 = w0 <= 0 ? 0ull : (w0 >= 64 ? ~0ull : ((1ull << w0) - 1ull));
                    const uint64_t m1 = w1 <= 0 ? 0ull : (w1 >= 64 ? ~0ull : ((1ull << w1) - 1ull));
                    nb2 += (int32_t)__popcll(mb & m1 & ~m0);
                    uint64_t ms = __ballot(e >= a && e < a + 63);
                    while (ms) {
                        const int u = __ffsll((unsigned long long)ms) - 1;
                        ms &= ms - 1;
                        const int32_t je = (int32_t)__popcll(__ballot(fj1 <= base + u));
                        if (lane == 0 && ns2 < BN_SPEC) s_spec[HYP ? 1 : 0][ns2] = (je << 8) | (rdl_bn(e, u) - a);
                        ns2++;
                    }
                }
            }
            __threadfence_block();
            const bool listed = ns <= BN_SPEC && ns2 <= BN_SPEC;
            // exclusive prefix sums over the lanes: P(j), P_L(j); totals in lane 63's inclusive value
            int32_t Pn = nb, Pl = nb2;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int32_t u1 = __shfl_up(Pn, o), u2 = HYP ? __shfl_up(Pl, o) : 0;
                if (lane >= o) { Pn += u1; Pl += u2; }
            }
            const int32_t totn = rdl_bn(Pn, 63), totl = HYP ? rdl_bn(Pl, 63) : 0;
            Pn -= nb; Pl -= nb2;
            const TC colterm = cadd(cmulc((int64_t)xi, kV), cmulc((int64_t)pj, kP));
            const TC G = cadd(cadd(Wj, colterm), cadd(cmulc((int64_t)Pn, kN), cmulc((int64_t)Pl, kL)));
            // the specials row r counts among the columns left of lane j: nets list: next >= r; self list: last < r
            auto spec_counts = [&](int j, int32_t &sn, int32_t &sl) {
                sn = 0; sl = 0;
                if (listed) {
                    for (int i = 0; i < ns; i++) { const int32_t w = s_spec[0][i]; sn += ((w >> 8) < j) && ((w & 255) >= lane); }
                    if (HYP) for (int i = 0; i < ns2; i++) { const int32_t w = s_spec[HYP ? 1 : 0][i]; sl += ((w >> 8) < j) && ((w & 255) < lane); }
                } else {
                    // too many for the lists: straight from the entries of the columns x_0 .. x_(j-1)
                    if (nets) {
                        const int32_t qe = __shfl(pj, j < 64 ? j : 63) ;
                        const int32_t qend = j < 64 ? qe : e_hi;
                        for (int32_t base = e_lo; base < e_hi; base += 64) {
                            const int32_t x = base + lane < e_hi ? C.next[base + lane] : INT32_MIN;
                            const int m = e_hi - base < 64 ? e_hi - base : 64;
                            for (int u = 0; u < m; u++) { const int32_t v = rdl_bn(x, u); sn += (base + u < qend) && v >= a && v < a + 63 && v >= r; }
                        }
                    }
                    if (HYP) {
                        const int32_t g_lo = rdl_bn(fj, 0), g_hi = rdl_bn(fj1, 63);
                        const int32_t qe = __shfl(fj, j < 64 ? j : 63);
                        const int32_t qend = j < 64 ? qe : g_hi;
                        for (int32_t base = g_lo; base < g_hi; base += 64) {
                            const int32_t x = base + lane < g_hi ? C.flast[base + lane] : INT32_MAX;
                            const int m = g_hi - base < 64 ? g_hi - base : 64;
                            for (int u = 0; u < m; u++) { const int32_t v = rdl_bn(x, u); sl += (base + u < qend) && v >= a && v < a + 63 && v < r; }
                        }
                    }
                }
            };
            int32_t snb, slb;
            spec_counts(jb, snb, slb);
            // T(r): the right-hand side of the regrouped test
            const int32_t Pnb = __shfl(Pn, jb), Plb = HYP ? __shfl(Pl, jb) : 0;
            const TC T = cadd(rowpart, cadd(cmulc((int64_t)(nn + Pnb + snb), kN), cmulc((int64_t)(nl + Plb + slb), kL)));
            int jmax = phi_r - xw;                                       // largest lane whose column is a candidate of the row
            if (jmax > 63) jmax = 63;
            const int jmin = plo_r > xw ? plo_r - xw : 0;                // ... and the smallest (width-constrained layers)
            // lower bound over [jmin, jmax + 1): the first lane with G(j) + k_N S(j, r) + k_L S_L(j, r) >= T
            int lo = jmin, hi = jmax + 1;
            if (hi < lo) hi = lo;
            const bool anyspec = ns > 0 || ns2 > 0;
#pragma unroll
            for (int it = 0; it < 7; it++) {
                const int mid = (lo + hi) >> 1;
                const TC g = shfl64_bn(G, mid & 63);
                TC lhs = g;
                if (anyspec) { int32_t sn, sl; spec_counts(mid, sn, sl); lhs = cadd(lhs, cadd(cmulc((int64_t)sn, kN), cmulc((int64_t)sl, kL))); }
                const bool pred = lhs >= T;
                if (lo < hi) { if (pred) hi = mid; else lo = mid + 1; }
            }
            const int js = lo;                                           // crossing lane, or jmax + 1: none in this window
            // (every shuffle is done by ALL lanes, outside the divergent bookkeeping below: a lane that reads a register of an
            //  inactive lane through ds_bpermute gets zero)
            const int jp = js >= 1 ? js - 1 : 0;
            const int jm = jmax >= 0 ? (jmax & 63) : 0;
            int32_t snp = 0, slp = 0, sns = 0, sls = 0, sn63 = 0, sl63 = 0, sn64 = 0, sl64 = 0, snm = 0, slm = 0;
            if (anyspec) { spec_counts(jp, snp, slp); spec_counts(js, sns, sls); spec_counts(63, sn63, sl63); spec_counts(64, sn64, sl64); spec_counts(jm, snm, slm); }
            const TC gp = shfl64_bn(G, jp & 63), wp = shfl64_bn(Wj, jp & 63);
            const int32_t Pns = __shfl(Pn, js & 63), Pls = HYP ? __shfl(Pl, js & 63) : 0;
            const int32_t Pnm = __shfl(Pn, jm), Plm = HYP ? __shfl(Pl, jm) : 0;
            const TC g63 = rdl64_bn(G, 63), w63 = rdl64_bn(Wj, 63);
            if (open) {
                const bool found = js <= jmax;
                const bool none = !found && jmax < 63;                   // the row ends inside the window: c = r + 1
                if (found || none) {
                    c = found ? xw + js : phi_r + 1;
                    // f(c - 1, r): the cost at lane js - 1 (the previous window's last lane when js == 0) -- if c - 1 is a candidate
                    if (js >= 1) { fm = csub(csub(T, csub(gp, wp)), cadd(cmulc((int64_t)snp, kN), cmulc((int64_t)slp, kL))); have_fm = js - 1 >= jmin; }
                    else { fm = fprev; have_fm = have_prev && xw - 1 >= plo_r; }
                    // column and counts the next sub-run starts from: the crossing, or -- none -- the row's last candidate phi
                    if (found) { c_carry = c; nn = nn - (Pns - Pnb) - (sns - snb); nl = nl - (Pls - Plb) - (sls - slb); }
                    else { c_carry = phi_r; nn = nn - (Pnm - Pnb) - (snm - snb); nl = nl - (Plm - Plb) - (slm - slb); }
                    open = false;
                } else {
                    // beyond the window: the cost at its last lane, the counts at the next window's first column
                    fprev = csub(csub(T, csub(g63, w63)), cadd(cmulc((int64_t)sn63, kN), cmulc((int64_t)sl63, kL)));
                    have_prev = true;
                    nn = nn - (totn - Pnb) - (sn64 - snb);
                    nl = nl - (totl - Plb) - (sl64 - slb);
                }
            }
            jb = 0;
            if (!__ballot(open)) break;
        }
        // ---- values: min(f(c - 1, r), W[c]); the largest minimiser
        if (valid) {
            TC v; int32_t p;
            const TC Wc = c <= phi_r ? C.W[c] : (TC)0;
            if (plo_r > phi_r) { v = CostTraits<TC>::typemax(); p = plo_r; }      // no candidate at all (the reference reads an out-of-window cell: typemax)
            else if (c <= phi_r && (!have_fm || Wc <= fm)) { const int32_t re = runend[c]; v = Wc; p = re < phi_r ? re : phi_r; }
            else { v = fm; p = c - 1; }
            cst[r] = v; ptr[r] = p;
        }
        if (a == r_begin && lane == 0) hint_out[t] = c;                  // the next layer's starts gallop from this run's true first crossing
        if (alast >= r_end) break;
        a = alast;
        cs = rdl_bn(c_carry, 63); nn_a = rdl_bn(nn, 63); nl_a = rdl_bn(nl, 63);
    }
}

struct BnWork {
    bool have_net = false, have_self = false;
    WaveletHost net, self;
    DBuf<int32_t> runend, blk, c0, nn0, nl0, hint, hint2;
    int64_t hint_rlo = -1, hint_nchunk = -1, hint_ch = -1;      // the tiling the hints belong to (-1: none)
    int hint_limited = -1;                                      // ... 1: a weight-constrained layer (hints per absolute chunk), 0: a plain one
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

int64_t g_opt_bn_chunk = 8;       // lane-per-chunk walk (bn_wave 0): rows per lane (config 3 matrix, K = 64: 8 rows 0.50 s, 32 rows 0.58 s, 128 rows 0.97 s, 256 rows 1.32 s per partition)
int64_t g_opt_bn_wave = 2;        // 0: lane per chunk (k_bn_walk); 1: wave per run, lockstep (k_bn_walk64); 2: wave per run, searched crossings for Int64 costs (k_bn_walk_vec)
int64_t g_opt_bn_slack = 64;      // hinted starts of the wave walks: columns left of the predicted crossing
int64_t g_opt_bn_run = 253;       // ... rows per wave (1 + 63 m: m sub-runs of 64 rows sharing their end rows)

template <typename TC>
void dp_bottleneck_layer(cp_csr_s *A, const DevModel<TC> &M, TC alpha, const TC *W, TC *cst_out, int32_t *ptr_out,
                         int64_t rlo, int64_t rhi, int64_t wwin, int64_t p_lo0, int64_t p_hi0, const int32_t *j0)
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
    C.N = A->N; C.NF = hyp ? A->nrows_nonempty : 0;
    const bool limited = wwin > 0 || j0 || p_lo0 > 0 || (p_hi0 >= 0 && p_hi0 < n);
    C.j0 = j0;
    C.wwin = wwin > 0 ? wwin : 0; C.p_lo0 = p_lo0 > 0 ? p_lo0 : 0; C.p_hi0 = (p_hi0 >= 0 && p_hi0 < n) ? p_hi0 : n;
    CP_REQUIRE(!limited || (g_opt_bn_wave >= 2 && CostTraits<TC>::is_int), CP_EINTERNAL, "candidate limits need the searched-crossings walk (Int64 costs)");
    C.pos = A->pos.p; C.lpos = hyp ? A->lpos.p : nullptr;
    C.pos32 = A->pos32.p; C.prev = A->prev.p; C.next = A->next.p;
    C.fpos32 = hyp ? A->fpos32.p : nullptr; C.flast = hyp ? A->flast.p : nullptr;
    C.lpos32 = hyp ? A->lpos32.p : nullptr; C.lfirst = hyp ? A->lfirst.p : nullptr;
    C.net = B->net.d; C.self = B->self.d; C.W = W;
    const int64_t nblk = cdiv(n1, 1024);
    B->runend.ensure((size_t)n1); B->blk.ensure((size_t)nblk + 1);
    const bool wave = g_opt_bn_wave != 0;
    const int64_t CH = wave ? std::max<int64_t>(2, g_opt_bn_run) : std::max<int64_t>(1, g_opt_bn_chunk);
    // A weight-constrained layer's row window moves from layer to layer: its chunks are aligned to multiples of CH (the first one
    // starts inside its chunk, at rmin) and the hints are kept per ABSOLUTE chunk, so a row's crossing in the previous layer still
    // guides its start (the windows of consecutive layers overlap almost entirely).  Any hint is safe: it only chooses where the
    // search starts.
    const int64_t rmin = rlo;
    if (limited) rlo = (rlo / CH) * CH;
    const int64_t tbase = limited ? rlo / CH : 0;
    const int64_t nchunk = cdiv(rhi - rlo + 1, CH);
    B->c0.ensure((size_t)nchunk); B->nn0.ensure((size_t)nchunk); B->nl0.ensure((size_t)nchunk);
    ProfScope ps(PROF_BRUTE, s, 8.0 * (double)A->N + 24.0 * (double)(rhi - rlo + 1));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_run1<TC>), dim3((unsigned)nblk), dim3(1024), 0, s, n1, W, B->runend.p, B->blk.p);
    hipLaunchKernelGGL(k_bn_run2, dim3(1), dim3(1024), 0, s, nblk, B->blk.p);
    hipLaunchKernelGGL(k_bn_run3, dim3((unsigned)nblk), dim3(1024), 0, s, n1, nblk, B->runend.p, B->blk.p);
    const bool hinted = B->hint_ch == CH && nchunk > 1 && B->hint_limited == (limited ? 1 : 0) &&
                        (limited ? B->hint_layers >= 1 : (B->hint_nchunk == nchunk && B->hint_rlo == rlo));
    if (!hinted) B->hint_layers = 0;
    {
        const size_t need = (size_t)std::max<int64_t>(nchunk, n1 / CH + 2);
        if (B->hint.n < need || B->hint2.n < need) {
            B->hint.ensure(need); B->hint2.ensure(need);
            CP_HIP(hipMemsetAsync(B->hint.p, 0, B->hint.bytes(), s)); CP_HIP(hipMemsetAsync(B->hint2.p, 0, B->hint2.bytes(), s));
        }
    }
    // every 64th chunk from the previous layer's hint, the rest bracketed by those.  Tried, config 3 matrix, K = 64 (302 ms as is):
    //  - a level of every 8th chunk in between: 6 probes of the counter for most chunks instead of 9, but a third chain of
    //    dependent probes: 318 ms;
    //  - the coarse starts by one WAVE per row, a 64-ary search over the whole range (4 rounds instead of ~25 dependent probes):
    //    64 lanes x 24 levels x 5 loads of different cache lines per round keep the L1 busy for longer than the chain took
    //    (1.58 ms per layer against 0.9 ms);
    //  - the fine starts as a 4-ary search, three interleaved descents per step: 1.0 ms against 0.9 ms -- 1.25 M lanes are bound by
    //    the number of line requests, not by the length of the chain.
    const int32_t *h1 = hinted ? B->hint.p + tbase : (const int32_t *)nullptr;
    const int32_t *h2 = hinted && B->hint_layers >= 2 && !(g_opt_dbg & 4194304) ? B->hint2.p + tbase : (const int32_t *)nullptr;
    if (hinted && wave && !(g_opt_dbg & 33554432)) {
        // the wave-per-run walks need one start per few hundred rows, and any column left of the crossing will do: the crossing of
        // the same row in the previous layer (continued by that layer's move) minus a slack, one confirming probe
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_starts<TC>), dim3((unsigned)cdiv(nchunk, 256)), dim3(256), 0, s, C, rlo, rhi, CH, nchunk, 1, 0, B->c0.p, B->nn0.p, B->nl0.p, h1, h2,
                           (int32_t)g_opt_bn_slack, rmin);
    } else {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_starts<TC>), dim3((unsigned)cdiv(cdiv(nchunk, 64), 256)), dim3(256), 0, s, C, rlo, rhi, CH, nchunk, 64, 0, B->c0.p, B->nn0.p,
                       B->nl0.p, h1, h2, -1, rmin);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_starts<TC>), dim3((unsigned)cdiv(nchunk, 256)), dim3(256), 0, s, C, rlo, rhi, CH, nchunk, 1, 64, B->c0.p, B->nn0.p, B->nl0.p,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, -1, rmin);
    }
    std::swap(B->hint.p, B->hint2.p); std::swap(B->hint.n, B->hint2.n);      // (hint2 <- the previous layer's crossings)
    if (!wave) CP_HIP(hipMemcpyAsync(B->hint.p, B->c0.p, sizeof(int32_t) * (size_t)nchunk, hipMemcpyDeviceToDevice, s));      // (the wave walks store their first crossings themselves)
    B->hint_layers++;
    B->hint_nchunk = nchunk; B->hint_rlo = rlo; B->hint_ch = CH; B->hint_limited = limited ? 1 : 0;
    if (wave && g_opt_bn_wave >= 2 && CostTraits<TC>::is_int) {
        // Int64 costs: the crossings by binary search over windows of 64 columns
        const BnCtx<int64_t> &Ci = reinterpret_cast<const BnCtx<int64_t> &>(C);
        int64_t *co = reinterpret_cast<int64_t *>(cst_out);
        if (hyp) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_walk_vec<true>), dim3((unsigned)cdiv(nchunk, 4)), dim3(256), 0, s, Ci, rlo, rhi, CH, nchunk, B->c0.p, B->nn0.p, B->nl0.p,
                                    B->runend.p, co, ptr_out, B->hint.p + tbase, rmin);
        else     hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_walk_vec<false>), dim3((unsigned)cdiv(nchunk, 4)), dim3(256), 0, s, Ci, rlo, rhi, CH, nchunk, B->c0.p, B->nn0.p, B->nl0.p,
                                    B->runend.p, co, ptr_out, B->hint.p + tbase, rmin);
    } else if (wave && hyp)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_walk64<TC, true>), dim3((unsigned)cdiv(nchunk, 4)), dim3(256), 0, s, C, rlo, rhi, CH, nchunk, B->c0.p, B->nn0.p, B->nl0.p,
                           B->runend.p, cst_out, ptr_out, B->hint.p);
    else if (wave)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_walk64<TC, false>), dim3((unsigned)cdiv(nchunk, 4)), dim3(256), 0, s, C, rlo, rhi, CH, nchunk, B->c0.p, B->nn0.p, B->nl0.p,
                           B->runend.p, cst_out, ptr_out, B->hint.p);
    else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bn_walk<TC>), dim3((unsigned)cdiv(nchunk, 256)), dim3(256), 0, s, C, rlo, rhi, CH, nchunk, B->c0.p, B->nn0.p, B->nl0.p,
                       B->runend.p, cst_out, ptr_out);
    CP_HIP(hipGetLastError());
}

template void dp_bottleneck_layer<int64_t>(cp_csr_s *, const DevModel<int64_t> &, int64_t, const int64_t *, int64_t *, int32_t *, int64_t, int64_t, int64_t, int64_t, int64_t, const int32_t *);
template void dp_bottleneck_layer<double>(cp_csr_s *, const DevModel<double> &, double, const double *, double *, int32_t *, int64_t, int64_t, int64_t, int64_t, int64_t, const int32_t *);

}  // namespace cpk
