// dp_total.hip -- one layer of the total-cost (g = +) K-part DP in O(n log^2 n) streamed column steps,
// exact to the reference's literal O(n^2) sweep including ties (largest j wins):
//
//     cst[j',k] = min_{1<=j<=j'} cst[j,k-1] + f(j,j',k)        /root/reference/src/DynamicSplitter.jl:33-46
//
// Scheme (executable spec: tests/dc_model.py; derivation: DESIGN.md section 4).  0-based boundary
// positions p = j-1, r = j'-1.
//  * candidates [0,r) of row r split, Fenwick style, into one block per set bit b of r:
//    [r_b - 2^b, r_b), r_b = r with the bits below b cleared.  All rows sharing (b, r_b) form a
//    full rectangle rows (r_b, r_b+2^b) x cols [r_b-2^b, r_b), on which the cost matrix
//    W[p] + f(p,r) is inverse-Monge for the eligible models (coverage counts are submodular), so
//    the RIGHTMOST row argmin is non-increasing in r: a monotone divide and conquer applies.
//  * rows are processed in rounds by tau = ctz(r) (high to low) after a first round for the rows
//    with r == r_b; a row's candidate range is bounded by the argmins of its two tree neighbours.
//  * nets(p,r) along the staircase path anchor -> (B,r) -> (a,r) is a prefix sum of per-column
//    steps: adding column c on the right of a part starting at B adds #{q in c : prev[q] < B};
//    adding column p on the left of a part ending before r adds #{q in p : next[q] >= r}.
//    All steps of all rows of a round are flattened into one array, counted in a streaming pass over
//    the link arrays (k_expand), segment-scanned (tile scan + carry), evaluated and arg-min reduced
//    (k_eval, k_fix).
#include "csr.hpp"
#include "model.hpp"
#include "dp.hpp"

namespace cpk {

constexpr int TILE_T = 256;
constexpr int TILE_I = 4;
constexpr int TILE = TILE_T * TILE_I;

struct RoundDesc {
    int32_t isA, tau, nbits, _pad;
    int64_t n, ntask;
    int64_t tbase[36];
};

__device__ __forceinline__ void decode_task(const RoundDesc &R, int64_t t, int64_t &r, int &b)
{
    if (R.isA) { r = t + 1; b = __ffsll((long long)r) - 1; return; }
    int bb = R.tau + 1;
    while (t >= R.tbase[bb + 1]) bb++;
    int64_t l = t - R.tbase[bb];
    int sh = bb - R.tau - 1;
    int64_t base = l >> sh, v = l & (((int64_t)1 << sh) - 1);
    r = (base << (bb + 1)) | ((int64_t)1 << bb) | (((v << 1) | 1) << R.tau);
    b = bb;
}

// ------------------------------------------------------------------ task setup
__global__ void __launch_bounds__(256) k_setup(RoundDesc R, const int32_t *__restrict__ opt, const int32_t *__restrict__ nnopt,
                                               int32_t *__restrict__ tB, int32_t *__restrict__ tS0, int32_t *__restrict__ tr,
                                               uint8_t *__restrict__ tb, int32_t *__restrict__ len)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R.ntask) return;
    int64_t r; int b;
    decode_task(R, t, r, b);
    int64_t n1 = R.n + 1;
    int64_t B, a, S0, nR;
    if (R.isA) {
        B = r; a = r - ((int64_t)1 << b); S0 = 0; nR = 0;
    } else {
        int64_t rb = (r >> b) << b;
        int64_t rL = r - ((int64_t)1 << R.tau), rR = r + ((int64_t)1 << R.tau);
        B = opt[(int64_t)b * n1 + rL];
        S0 = nnopt[(int64_t)b * n1 + rL];
        a = ((rR - rb) < ((int64_t)1 << b) && rR <= R.n) ? (int64_t)opt[(int64_t)b * n1 + rR] : rb - ((int64_t)1 << b);
        nR = (int64_t)1 << R.tau;
        if (a > B) a = B;          // cannot happen for an inverse-Monge cost; keeps every task well-formed
    }
    tB[t] = (int32_t)B; tS0[t] = (int32_t)S0; tr[t] = (int32_t)r; tb[t] = (uint8_t)b;
    len[t] = (int32_t)(nR + (B - a));
}

// ------------------------------------------------------------------ tile helpers
struct TileTasks {
    int64_t t0;          // first task overlapping the tile
    int32_t cnt;         // number of task offsets loaded (tasks t0 .. t0+cnt-1 start at s_off[0..cnt-1])
};

// loads the offsets of the tasks overlapping [tile_start, tile_start+TILE) into s_off (absolute int64)
__device__ __forceinline__ void load_tile_tasks(const int64_t *__restrict__ offs, int64_t ntask, int64_t tile_start,
                                                int64_t *s_off, int64_t *s_t0, int32_t *s_cnt)
{
    if (threadIdx.x == 0) {
        int64_t lo = 0, hi = ntask;         // last t with offs[t] <= tile_start  (offs[0] = 0)
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if (offs[mid] <= tile_start) lo = mid; else hi = mid;
        }
        *s_t0 = lo;
        int64_t c = ntask - lo;             // every task has len >= 1, so at most TILE tasks start inside the tile
        *s_cnt = (int32_t)(c > TILE + 1 ? TILE + 1 : c);
    }
    __syncthreads();
    int64_t t0 = *s_t0;
    int32_t cnt = *s_cnt;
    for (int i = threadIdx.x; i < cnt; i += TILE_T) s_off[i] = offs[t0 + i];
    __syncthreads();
}

// local task index: last i in [0,cnt) with s_off[i] <= e
__device__ __forceinline__ int find_local(const int64_t *s_off, int32_t cnt, int64_t e)
{
    int lo = 0, hi = cnt;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (s_off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------ expand: per-step column counts + tile-local segmented scan
// loc[e] = sum of the step counts of e's task from max(task start, tile start) through e.
__global__ void __launch_bounds__(TILE_T) k_expand(RoundDesc R, int64_t T, const int64_t *__restrict__ offs,
                                                   const int32_t *__restrict__ tB, const int32_t *__restrict__ tr,
                                                   const int64_t *__restrict__ pos, const int32_t *__restrict__ prev,
                                                   const int32_t *__restrict__ next,
                                                   int32_t *__restrict__ loc, int32_t *__restrict__ tileF, int32_t *__restrict__ tileS)
{
    __shared__ int64_t s_off[TILE + 2];
    __shared__ int64_t s_t0;
    __shared__ int32_t s_cnt;
    __shared__ int32_t s_F[TILE_T], s_S[TILE_T];
    int64_t tile_start = (int64_t)blockIdx.x * TILE;
    load_tile_tasks(offs, R.ntask, tile_start, s_off, &s_t0, &s_cnt);
    int64_t t0 = s_t0;
    int32_t cnt = s_cnt;
    int64_t nR = R.isA ? 0 : ((int64_t)1 << R.tau);

    int32_t d[TILE_I];
    bool h[TILE_I];
    int64_t e0 = tile_start + (int64_t)threadIdx.x * TILE_I;
#pragma unroll
    for (int k = 0; k < TILE_I; k++) {
        int64_t e = e0 + k;
        d[k] = 0; h[k] = true;
        if (e < T) {
            int li = find_local(s_off, cnt, e);
            int64_t t = t0 + li;
            int64_t i = e - s_off[li];
            h[k] = (i == 0);
            int64_t B = tB[t], r = tr[t];
            int32_t c = 0;
            if (i < nR) {                       // right step: column (r - 2^tau + i) joins a part that starts at B
                int64_t col = r - nR + i;
                int64_t q0 = pos[col], q1 = pos[col + 1];
                int32_t thr = (int32_t)B;
                for (int64_t q = q0; q < q1; q++) c += (prev[q] < thr);
            } else {                            // left step: column p joins a part that ends before r
                int64_t p = B - 1 - (i - nR);
                int64_t q0 = pos[p], q1 = pos[p + 1];
                int32_t thr = (int32_t)r;
                for (int64_t q = q0; q < q1; q++) c += (next[q] >= thr);
            }
            d[k] = c;
        }
    }
    // thread-local segmented inclusive scan
    int32_t x[TILE_I];
    int32_t run = 0;
    bool anyh = false;
    int firsth = TILE_I;
#pragma unroll
    for (int k = 0; k < TILE_I; k++) {
        if (h[k]) { run = 0; if (!anyh) firsth = k; anyh = true; }
        run += d[k];
        x[k] = run;
    }
    s_F[threadIdx.x] = anyh; s_S[threadIdx.x] = run;
    __syncthreads();
    // block-level inclusive segmented scan of (F,S): (F1,S1)+(F2,S2) = (F1|F2, F2 ? S2 : S1+S2)
    for (int o = 1; o < TILE_T; o <<= 1) {
        int32_t f = 0, sv = 0;
        bool take = threadIdx.x >= (unsigned)o;
        if (take) { f = s_F[threadIdx.x - o]; sv = s_S[threadIdx.x - o]; }
        __syncthreads();
        if (take) {
            int32_t mf = s_F[threadIdx.x], ms = s_S[threadIdx.x];
            s_S[threadIdx.x] = mf ? ms : sv + ms;
            s_F[threadIdx.x] = mf | f;
        }
        __syncthreads();
    }
    int32_t carry = threadIdx.x > 0 ? s_S[threadIdx.x - 1] : 0;
#pragma unroll
    for (int k = 0; k < TILE_I; k++) {
        int64_t e = e0 + k;
        if (e < T) loc[e] = x[k] + (k < firsth ? carry : 0);
    }
    if (threadIdx.x == TILE_T - 1) { tileF[blockIdx.x] = s_F[TILE_T - 1]; tileS[blockIdx.x] = s_S[TILE_T - 1]; }
}

// ------------------------------------------------------------------ carry across tiles (single block)
// carry[i] = sum of the counts of tile i's first segment that lie in earlier tiles (0 if tile i starts with a head;
// a stale non-zero value there is harmless: k_eval applies the carry only to a segment that started earlier).
__global__ void __launch_bounds__(1024) k_carry(const int32_t *__restrict__ tileF, const int32_t *__restrict__ tileS,
                                                int32_t *__restrict__ carry, int64_t ntile)
{
    __shared__ int32_t s_F[1024], s_S[1024];
    int64_t chunk = (ntile + 1023) / 1024;
    int64_t lo = (int64_t)threadIdx.x * chunk, hi = lo + chunk < ntile ? lo + chunk : ntile;
    int32_t f = 0, sv = 0;
    for (int64_t i = lo; i < hi; i++) {
        if (tileF[i]) { f = 1; sv = tileS[i]; } else sv += tileS[i];
    }
    s_F[threadIdx.x] = f; s_S[threadIdx.x] = sv;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int32_t pf = 0, psv = 0;
        bool take = threadIdx.x >= (unsigned)o;
        if (take) { pf = s_F[threadIdx.x - o]; psv = s_S[threadIdx.x - o]; }
        __syncthreads();
        if (take) {
            int32_t mf = s_F[threadIdx.x], ms = s_S[threadIdx.x];
            s_S[threadIdx.x] = mf ? ms : psv + ms;
            s_F[threadIdx.x] = mf | pf;
        }
        __syncthreads();
    }
    int32_t run = threadIdx.x > 0 ? s_S[threadIdx.x - 1] : 0;     // inclusive value of everything before lo
    for (int64_t i = lo; i < hi; i++) {
        carry[i] = run;
        if (tileF[i]) run = tileS[i]; else run += tileS[i];
    }
}

// ------------------------------------------------------------------ evaluate candidates + segmented arg-min
template <typename TC>
struct Best { TC v; int32_t p; int32_t nn; };

template <typename TC>
__device__ __forceinline__ Best<TC> better(const Best<TC> &a, const Best<TC> &b)   // a is earlier (larger p): wins ties
{
    if (a.p < 0) return b;
    if (b.p < 0) return a;
    return (b.v < a.v) ? b : a;
}

template <typename TC>
__global__ void __launch_bounds__(TILE_T) k_eval(RoundDesc R, int64_t T, const int64_t *__restrict__ offs,
                                                 const int32_t *__restrict__ tB, const int32_t *__restrict__ tS0,
                                                 const int32_t *__restrict__ tr, const uint8_t *__restrict__ tb,
                                                 const int64_t *__restrict__ pos, const int32_t *__restrict__ loc,
                                                 const int32_t *__restrict__ carry, const TC *__restrict__ W,
                                                 DevModel<TC> M, TC alpha,
                                                 int32_t *__restrict__ opt, int32_t *__restrict__ nnopt,
                                                 Best<TC> *__restrict__ partL, Best<TC> *__restrict__ partR,
                                                 int64_t *__restrict__ taskR)
{
    __shared__ int64_t s_off[TILE + 2];
    __shared__ int64_t s_t0;
    __shared__ int32_t s_cnt;
    __shared__ int32_t s_F[TILE_T];
    __shared__ Best<TC> s_B[TILE_T];
    int64_t tile_start = (int64_t)blockIdx.x * TILE;
    load_tile_tasks(offs, R.ntask, tile_start, s_off, &s_t0, &s_cnt);
    int64_t t0 = s_t0;
    int32_t cnt = s_cnt;
    int64_t nR = R.isA ? 0 : ((int64_t)1 << R.tau);
    int32_t cin = carry[blockIdx.x];
    int64_t n1 = R.n + 1;

    Best<TC> x[TILE_I];
    bool h[TILE_I];
    int64_t tsk[TILE_I];
    int64_t e0 = tile_start + (int64_t)threadIdx.x * TILE_I;
#pragma unroll
    for (int k = 0; k < TILE_I; k++) {
        int64_t e = e0 + k;
        x[k].p = -1; x[k].nn = 0; x[k].v = (TC)0; h[k] = true; tsk[k] = -1;
        if (e < T) {
            int li = find_local(s_off, cnt, e);
            int64_t t = t0 + li;
            tsk[k] = t;
            int64_t toff = s_off[li];
            int64_t i = e - toff;
            h[k] = (i == 0);
            int64_t B = tB[t], r = tr[t];
            int64_t nn = (int64_t)tS0[t] + loc[e] + (toff < tile_start ? cin : 0);
            int64_t p = -1;
            if (i >= nR) p = B - 1 - (i - nR);
            else if (i == nR - 1) p = B;
            if (p >= 0) {
                TC f = dm_apply(M, alpha, r - p, pos[r] - pos[p], nn, (int64_t)0);
                x[k].v = cadd(W[p], f);
                x[k].p = (int32_t)p;
                x[k].nn = (int32_t)nn;
            }
        }
    }
    // thread-local segmented inclusive "best so far"
    Best<TC> run; run.p = -1; run.nn = 0; run.v = (TC)0;
    bool anyh = false;
    int firsth = TILE_I;
#pragma unroll
    for (int k = 0; k < TILE_I; k++) {
        if (h[k]) { run.p = -1; if (!anyh) firsth = k; anyh = true; }
        run = better(run, x[k]);
        x[k] = run;
    }
    s_F[threadIdx.x] = anyh; s_B[threadIdx.x] = run;
    __syncthreads();
    for (int o = 1; o < TILE_T; o <<= 1) {
        int32_t f = 0; Best<TC> pb; pb.p = -1; pb.nn = 0; pb.v = (TC)0;
        bool take = threadIdx.x >= (unsigned)o;
        if (take) { f = s_F[threadIdx.x - o]; pb = s_B[threadIdx.x - o]; }
        __syncthreads();
        if (take) {
            int32_t mf = s_F[threadIdx.x];
            Best<TC> mb = s_B[threadIdx.x];
            s_B[threadIdx.x] = mf ? mb : better(pb, mb);
            s_F[threadIdx.x] = mf | f;
        }
        __syncthreads();
    }
    Best<TC> cb; cb.p = -1; cb.nn = 0; cb.v = (TC)0;
    if (threadIdx.x > 0) cb = s_B[threadIdx.x - 1];
    int64_t tile_last = tile_start + TILE - 1;
    if (tile_last > T - 1) tile_last = T - 1;
#pragma unroll
    for (int k = 0; k < TILE_I; k++) {
        int64_t e = e0 + k;
        if (e >= T) continue;
        Best<TC> res = (k < firsth) ? better(cb, x[k]) : x[k];
        int64_t t = tsk[k];
        int li = (int)(t - t0);
        int64_t seg_start = s_off[li];
        int64_t seg_last = ((li + 1 < cnt) ? s_off[li + 1] : offs[t + 1]) - 1;
        bool head_in_tile = seg_start >= tile_start;
        if (e == seg_last) {
            if (head_in_tile) {
                int64_t r = tr[t]; int b = tb[t];
                opt[(int64_t)b * n1 + r] = res.p;
                nnopt[(int64_t)b * n1 + r] = res.nn;
            } else {
                partL[blockIdx.x] = res;
            }
        } else if (e == tile_last) {
            if (head_in_tile) { partR[blockIdx.x] = res; taskR[blockIdx.x] = t; }
            else partL[blockIdx.x] = res;
        }
    }
}

// a task whose steps span several tiles: combine the head tile's partial with the partials of the tiles it covers
template <typename TC>
__global__ void __launch_bounds__(256) k_fix(int64_t ntile, const int64_t *__restrict__ offs, const int64_t *__restrict__ taskR,
                                             const Best<TC> *__restrict__ partL, const Best<TC> *__restrict__ partR,
                                             const int32_t *__restrict__ tr, const uint8_t *__restrict__ tb,
                                             int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int64_t n1)
{
    int64_t tile = blockIdx.x;
    int64_t t = taskR[tile];
    if (t < 0) return;
    __shared__ Best<TC> s_B[256];
    __shared__ int64_t s_K[256];
    int64_t end_tile = (offs[t + 1] - 1) / TILE;
    Best<TC> acc; acc.p = -1; acc.nn = 0; acc.v = (TC)0;
    int64_t acck = INT64_MAX;
    for (int64_t k = tile + 1 + threadIdx.x; k <= end_tile; k += 256) {
        Best<TC> c = partL[k];          // tiles are visited in increasing order by each thread: earlier wins ties
        if (acc.p < 0 || (c.p >= 0 && c.v < acc.v)) { if (c.p >= 0) { acc = c; acck = k; } }
    }
    s_B[threadIdx.x] = acc; s_K[threadIdx.x] = acck;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < (unsigned)o) {
            Best<TC> a = s_B[threadIdx.x], b = s_B[threadIdx.x + o];
            int64_t ka = s_K[threadIdx.x], kb = s_K[threadIdx.x + o];
            bool takeb = (a.p < 0) ? (b.p >= 0) : (b.p >= 0 && (b.v < a.v || (b.v == a.v && kb < ka)));
            if (takeb) { s_B[threadIdx.x] = b; s_K[threadIdx.x] = kb; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        Best<TC> res = better(partR[tile], s_B[0]);
        int64_t r = tr[t]; int b = tb[t];
        opt[(int64_t)b * n1 + r] = res.p;
        nnopt[(int64_t)b * n1 + r] = res.nn;
    }
}

// ------------------------------------------------------------------ combine the per-bit winners of every row
template <typename TC>
__global__ void __launch_bounds__(256) k_combine(int64_t n, int nbits, const int64_t *__restrict__ pos,
                                                 const int32_t *__restrict__ opt, const int32_t *__restrict__ nnopt,
                                                 const TC *__restrict__ W, DevModel<TC> M, TC alpha,
                                                 TC *__restrict__ cst, int32_t *__restrict__ ptr)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    int64_t n1 = n + 1;
    TC bv = cadd(W[r], dm_apply(M, alpha, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0));   // j = j' (empty part)
    int64_t bp = r;
    for (int b = 0; b < nbits; b++) {
        if (!((r >> b) & 1)) continue;
        int64_t p = opt[(int64_t)b * n1 + r];
        int64_t nn = nnopt[(int64_t)b * n1 + r];
        TC v = cadd(W[p], dm_apply(M, alpha, r - p, pos[r] - pos[p], nn, (int64_t)0));
        if (v < bv) { bv = v; bp = p; }            // lower bits hold larger p: strict < keeps the largest p on ties
    }
    cst[r] = bv;
    ptr[r] = (int32_t)bp;
}

// ------------------------------------------------------------------ host driver for one layer
template <typename TC>
struct LayerWork {
    int64_t n = -1; int nbits = 0;
    DBuf<int32_t> opt, nnopt, tB, tS0, tr, len, loc, tileF, tileS, carry;
    DBuf<uint8_t> tb;
    DBuf<int64_t> offs, scratch, taskR;
    DBuf<Best<TC>> partL, partR;
    int64_t max_tasks = 0;
};

static void make_round(RoundDesc &R, bool isA, int tau, int nbits, int64_t n)
{
    memset(&R, 0, sizeof(R));
    R.isA = isA; R.tau = tau; R.nbits = nbits; R.n = n;
    if (isA) { R.ntask = n; return; }
    int64_t acc = 0;
    for (int b = 0; b <= tau; b++) R.tbase[b] = 0;
    for (int b = tau + 1; b < nbits; b++) {
        R.tbase[b] = acc;
        int sh = b - tau - 1;
        int64_t V = (int64_t)1 << sh;
        int64_t nfull = (n + ((int64_t)1 << tau)) >> (b + 1);
        int64_t cnt = nfull * V;
        int64_t x = n - (nfull << (b + 1)) - ((int64_t)1 << b);
        if (x >= ((int64_t)1 << tau)) cnt += ((x >> tau) + 1) >> 1;
        acc += cnt;
    }
    for (int b = nbits; b < 36; b++) R.tbase[b] = acc;
    R.ntask = acc;
}

template <typename TC>
void dp_total_layer(cp_csr_s *A, const DevModel<TC> &M, TC alpha, const TC *W, TC *cst_out, int32_t *ptr_out, void *work_)
{
    auto &Wk = *reinterpret_cast<LayerWork<TC> *>(work_);
    hipStream_t s = A->stream;
    int64_t n = A->n;
    int nbits = 1;
    while (((int64_t)1 << nbits) <= n) nbits++;
    if (Wk.n != n) {
        Wk.n = n; Wk.nbits = nbits;
        Wk.opt.alloc((size_t)nbits * (size_t)(n + 1));
        Wk.nnopt.alloc((size_t)nbits * (size_t)(n + 1));
        int64_t mx = n;
        for (int tau = 0; tau < nbits; tau++) { RoundDesc R; make_round(R, false, tau, nbits, n); if (R.ntask > mx) mx = R.ntask; }
        Wk.max_tasks = mx > 0 ? mx : 1;
        size_t mt = (size_t)Wk.max_tasks;
        Wk.tB.alloc(mt); Wk.tS0.alloc(mt); Wk.tr.alloc(mt); Wk.len.alloc(mt); Wk.tb.alloc(mt); Wk.offs.alloc(mt + 1);
    }
    for (int rd = 0; rd <= nbits; rd++) {
        RoundDesc R;
        if (rd == 0) make_round(R, true, 0, nbits, n);
        else make_round(R, false, nbits - rd, nbits, n);
        if (R.ntask <= 0) continue;
        {
            ProfScope ps(PROF_SETUP, s, 17.0 * (double)R.ntask);
            hipLaunchKernelGGL(k_setup, dim3((unsigned)cdiv(R.ntask, 256)), dim3(256), 0, s, R, Wk.opt.p, Wk.nnopt.p,
                               Wk.tB.p, Wk.tS0.p, Wk.tr.p, Wk.tb.p, Wk.len.p);
        }
        {
            ProfScope ps(PROF_SCAN, s, 12.0 * (double)R.ntask);
            exclusive_scan_i32(Wk.len.p, Wk.offs.p, R.ntask, Wk.scratch, s);
        }
        int64_t T = 0;
        CP_HIP(hipMemcpyAsync(&T, Wk.offs.p + R.ntask, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
        if (T <= 0) continue;
        int64_t ntile = cdiv(T, TILE);
        Wk.loc.ensure((size_t)T);
        if (Wk.tileF.n < (size_t)ntile) {
            Wk.tileF.alloc((size_t)ntile); Wk.tileS.alloc((size_t)ntile); Wk.carry.alloc((size_t)ntile);
            Wk.partL.alloc((size_t)ntile); Wk.partR.alloc((size_t)ntile); Wk.taskR.alloc((size_t)ntile);
        }
        {
            // algorithmic bytes of the step-count pass: one link entry (4 B) per nonzero of every stepped
            // column is not known on the host; account per step: colptr pair (16 B) + count out (4 B) here,
            // the link traffic is added by the bench from the average column degree.
            ProfScope ps(PROF_EXPAND, s, (double)T);
            hipLaunchKernelGGL(k_expand, dim3((unsigned)ntile), dim3(TILE_T), 0, s, R, T, Wk.offs.p, Wk.tB.p, Wk.tr.p,
                               A->pos.p, A->prev.p, A->next.p, Wk.loc.p, Wk.tileF.p, Wk.tileS.p);
        }
        {
            ProfScope ps(PROF_CARRY, s, 12.0 * (double)ntile);
            hipLaunchKernelGGL(k_carry, dim3(1), dim3(1024), 0, s, Wk.tileF.p, Wk.tileS.p, Wk.carry.p, ntile);
        }
        CP_HIP(hipMemsetAsync(Wk.taskR.p, 0xFF, sizeof(int64_t) * (size_t)ntile, s));
        {
            ProfScope ps(PROF_EVAL, s, (double)T);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_eval<TC>), dim3((unsigned)ntile), dim3(TILE_T), 0, s, R, T, Wk.offs.p, Wk.tB.p,
                               Wk.tS0.p, Wk.tr.p, Wk.tb.p, A->pos.p, Wk.loc.p, Wk.carry.p, W, M, alpha, Wk.opt.p, Wk.nnopt.p,
                               Wk.partL.p, Wk.partR.p, Wk.taskR.p);
        }
        {
            ProfScope ps(PROF_FIX, s, 8.0 * (double)ntile);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fix<TC>), dim3((unsigned)ntile), dim3(256), 0, s, ntile, Wk.offs.p, Wk.taskR.p,
                               Wk.partL.p, Wk.partR.p, Wk.tr.p, Wk.tb.p, Wk.opt.p, Wk.nnopt.p, n + 1);
        }
        CP_HIP(hipGetLastError());
    }
    {
        ProfScope ps(PROF_COMBINE, s, 24.0 * (double)(n + 1));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_combine<TC>), dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, n, nbits, A->pos.p,
                           Wk.opt.p, Wk.nnopt.p, W, M, alpha, cst_out, ptr_out);
    }
    CP_HIP(hipGetLastError());
}

template <typename TC> void *dp_total_work_new() { return new LayerWork<TC>(); }
template <typename TC> void dp_total_work_free(void *w) { delete reinterpret_cast<LayerWork<TC> *>(w); }

template void dp_total_layer<int64_t>(cp_csr_s *, const DevModel<int64_t> &, int64_t, const int64_t *, int64_t *, int32_t *, void *);
template void dp_total_layer<double>(cp_csr_s *, const DevModel<double> &, double, const double *, double *, int32_t *, void *);
template void *dp_total_work_new<int64_t>();
template void *dp_total_work_new<double>();
template void dp_total_work_free<int64_t>(void *);
template void dp_total_work_free<double>(void *);

}  // namespace cpk
