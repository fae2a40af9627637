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
//    with r == r_b; a row's candidate range [a, B] is bounded by the argmins of its two tree neighbours
//    r - 2^tau (gives B and the anchor count) and r + 2^tau (gives a).
//  * nets(p,r) along the staircase path anchor (B, r-2^tau) -> (B,r) -> (a,r):
//      right part: columns [r-2^tau, r) join a part starting at B: + #{q : prev[q] < B}.  Only the TOTAL is
//        needed, and the same columns serve every set bit b of r (different thresholds B_b): one row-major
//        reduction pass per round (k_rpass_*), N/2 link entries per round.
//      left part: columns p = B-1 .. a join a part ending before r: + #{q in p : next[q] >= r}, one
//        candidate per column.  Per round (k_setup_short -> scan -> k_tile_t0 -> k_lpass -> k_span_short / k_open / k_fix):
//          - tasks with a handful of candidates are finished by one lane in k_setup_short;
//          - the left steps of the other tasks are flattened (1 + B - a elements per task) and cut into 256-step tiles;
//          - a tile that lies inside ONE task (4 of 5 tiles) is one contiguous run of the link array: k_lpass streams
//            it with suffix counts and evaluates it with tile-local counts (the costs are affine in the counts);
//          - the other tiles are counted cooperatively (ballot/popcount per column), segment-scanned with wave
//            shuffles, evaluated and arg-min reduced per task;
//          - tasks spanning tiles are merged by k_span_short (one lane), or k_open + k_fix (work lists).
//  * plane arrays are stored level-major (prow): the rows of one round are contiguous in every plane.
//  * width-windowed layers (candidates max(0, r - w) <= p <= r: the ConstrainedCost DP, DynamicSplitter.jl:206-258) run the same
//    kernels on another tiling of the candidates: every row has a task in every plane b <= floor(log2 w) -- standard, common
//    and mirrored blocks, see struct Geo below and DESIGN.md section 4b.
#include "csr.hpp"
#include "model.hpp"
#include "dp.hpp"
#include <algorithm>

namespace cpk {

// The O(n log^2 n) scheme only admits the affine Work / Connectivity / HyperedgeCut models (fast_total_ok): its kernels evaluate
// costs through this three-way form, so the block-cost tables and pow() of the general dm_apply stay out of their registers.
template <typename TC>
__device__ __forceinline__ TC dm_apply_affine(const DevModel<TC> &m, TC alpha, int64_t nv, int64_t np, int64_t nn, int64_t nl)
{
    TC v = cadd(cadd(alpha, cmulc(nv, m.p[CP_P_VERTEX])), cmulc(np, m.p[CP_P_PIN]));       // (left to right, as the reference sums)
    if (m.kind == CP_MODEL_CONNECTIVITY) return cadd(v, cmulc(nn, m.p[CP_P_NET]));
    if (m.kind == CP_MODEL_HYPEREDGE_CUT) return cadd(cadd(v, cmulc(nl, m.p[CP_P_SELF_NET])), cmulc(nn - nl, m.p[CP_P_CUT_NET]));
    return v;
}
#define dm_apply dm_apply_affine

constexpr int LT = 256;          // steps per tile (one wave owns one tile)
constexpr int NBMAX = 31;        // bit planes (n < 2^30)

// Plane arrays (opt / nnopt / nlopt / cr / crl) hold row r of a bit plane at slot prow(r): rows are grouped by their level
// ctz(r) -- the rows one round reads and writes are then CONTIGUOUS in every plane (in row-major order they sit 2^(tau+1)
// entries apart and every round drags whole planes through HBM).  #rows in [1, n] with ctz < t is n - (n >> t).
__device__ __forceinline__ int64_t prow(int64_t r, int64_t n)
{
    int t = __ffsll((long long)r) - 1;
    return (n - (n >> t)) + (r >> (t + 1));
}
#define PR(x) prow((x), n1 - 1)

// Width-windowed layers (the ConstrainedCost DP, DynamicSplitter.jl:206-258 with a VertexCount weight; executable spec:
// tests/dc_model.py layer_windowed): candidates of row r are max(0, r - w) <= p <= r.  With s = floor(log2 w), S = 2^s, every row
// has a task in EVERY plane b <= s:
//   b < s, bit b of r set   : the standard Fenwick block [r_b - 2^b, r_b)
//   b == s                  : the common block [rho - w + S - 1, rho), rho = r with the bits below s cleared
//   b < s, bit b of r clear : the mirrored block [rho + 2^b - 1 - w, rho + 2^(b+1) - 1 - w), rho = r with the bits <= b cleared
// (clamped at column 0).  The rows sharing a block are always the aligned block of 2^b rows holding r -- full rectangles again.
struct Geo { int32_t win, s; int64_t w; };

// candidates [cs, ce) of row r in plane b; false: the row has no task in this plane (or the block is empty)
__host__ __device__ __forceinline__ bool geo_block(const Geo &G, int64_t r, int b, int64_t &cs, int64_t &ce)
{
    const int64_t lo = (r >> b) << b;
    if (!G.win) { cs = lo - ((int64_t)1 << b); ce = lo; return (r >> b) & 1; }
    if (b > G.s) return false;
    if (b == G.s) { cs = lo - G.w + ((int64_t)1 << b) - 1; ce = lo; }
    else if ((r >> b) & 1) { cs = lo - ((int64_t)1 << b); ce = lo; }
    else { cs = lo + ((int64_t)1 << b) - 1 - G.w; ce = cs + ((int64_t)1 << b); }
    if (cs < 0) cs = 0;
    if (ce < 0) ce = 0;
    return ce > cs;
}

struct RoundDesc {
    int32_t isA, tau, nbits, nextra;
    Geo G;                      // G.win: nbits = s + 1 planes, every row takes part in each of them
    int64_t aoff[32];           // windowed round A: anchors of the mirrored head tasks of plane b start at aoff[b] (index r >> (b+1))
    int64_t n, ntask;
    int64_t tbase[36];          // tau rounds: tasks of bit plane b occupy [tbase[b], tbase[b+1])
    int64_t tskip[36];          // ... and start at linear index tskip[b] of that plane (row-tiled runs skip rows left of the tile)
    int64_t a_r0, a_nmain;      // round A: rows a_r0 .. a_r0 + a_nmain - 1, then the `extra` rows (ancestors outside the tile)
    int64_t extra[64];
    // round A also takes the LAST row n in every bit plane above its lowest one (tasks after the extra rows): no row lies to
    // its right, so nothing bounds the rows next to it from the left; with opt[b][n] known first, they start there instead of
    // at the block start -- which every round would otherwise re-scan (n is not a power of two: ~n steps per round)
    int32_t nlast, last_b[31];
    int32_t skip_std, skip_mir; // windowed round A: the standard / mirrored heads (bit b of the row set / clear, b < s) come from the cached counts
    int32_t fin_stamp;          // a cell of the `fin` plane is set iff it holds this layer's stamp (1 .. 255: the plane is cleared every 255 layers, not every layer)
};

// Per-round counters, on the device (one record per round, kept for the whole layer).  Kernels read their loop bounds from
// here, so the host can enqueue a whole layer without waiting for a count to come back (dp_total_layer).
struct RoundCounts {
    int32_t n_open, n_fix;            // work lists of k_span_short (tiles of long spans)
    int32_t nlong, nown;              // flattened tasks / tasks with tiles of their own (k_setup_short)
    unsigned long long own_steps;     // steps of the latter
    int64_t T, NT;                    // flattened steps / own tiles (scan totals)
    int32_t ntile, err;               // cdiv(T, LT); 1: a buffer sized from the prediction is too small (the layer is redone)
    int32_t n_wide, _pad;             // own-tiled tasks of more than FIX_SERIAL tiles (k_fix_own_lane -> k_fix_own)
    int32_t n_glong, n_gslots;        // gap tasks of more than GAPSEG tiles and their segment slots (k_gap_finish -> k_gap_seg)
    int32_t n_gslow, n_sslow;         // work items of k_gap_finish / k_gap_seg that met a tile with more than SMAX specials (redone by the SLOW variants)
};

__device__ __forceinline__ void decode_task(const RoundDesc &R, int64_t t, int64_t &r, int &b)
{
    if (R.G.win) {
        // plane-major.  Round A: plane b <= s holds the rows that are multiples of 2^b (the heads of its rectangles);
        // round tau: every plane b in (tau, s] holds all rows with ctz == tau
        int bb = R.isA ? 0 : R.tau + 1;
        while (t >= R.tbase[bb + 1]) bb++;
        const int64_t l = t - R.tbase[bb] + R.tskip[bb];
        r = R.isA ? ((l + 1) << bb) : (((l << 1) | 1) << R.tau);
        b = bb;
        return;
    }
    if (R.isA) {
        if (t >= R.a_nmain + R.nextra) { r = R.n; b = R.last_b[t - R.a_nmain - R.nextra]; return; }
        r = t < R.a_nmain ? R.a_r0 + t : R.extra[t - R.a_nmain]; b = __ffsll((long long)r) - 1; return;
    }
    int bb = R.tau + 1;
    while (t >= R.tbase[bb + 1]) bb++;
    int64_t l = t - R.tbase[bb] + R.tskip[bb];
    int sh = bb - R.tau - 1;
    int64_t base = l >> sh, v = l & (((int64_t)1 << sh) - 1);
    r = (base << (bb + 1)) | ((int64_t)1 << bb) | (((v << 1) | 1) << R.tau);
    b = bb;
}

// ------------------------------------------------------------------ right part: row-major threshold counts
// cr[b][r] = #{q in columns [r-2^tau, r) : prev[q] < opt[b][r-2^tau]} for every set bit b > tau of r.
// Small ranges: one lane per row (tau <= 3).
// `ge` selects the comparison: 0: link < threshold (nets: prev[q] < B); 1: link >= threshold (self nets: first >= B
// over the rows bucketed by their LAST column).
// The 64 rows of a wave are consecutive, so they share every bit above tau + 6: planes whose bit is clear in the whole
// wave are skipped with a wave-uniform test (about half of them).
// (Tried: G = 2 .. 16 lanes per row, the group reading 16 G contiguous entries per step (a load instruction within 8 cache lines
//  instead of 64) and adding up by a butterfly: 238 / 262 / 288 / 325 us at tau = 4 .. 1 against 161 / 149 / 167 / 217 -- the
//  per-wave prelude (thresholds of two dozen planes) is then paid per 64 / G rows.)
// (Tried: deciding all planes but one by the position of the link value among the Fenwick blocks, with lane-private LDS
//  histograms -- 4x fewer VALU instructions but 300 B of LDS per row leave 2 waves per SIMD: slower.)
template <bool ge, int NB>
__global__ void __launch_bounds__(256) k_rpass_small(int tau, int nbits, int64_t n, int64_t u0, int64_t nrows, const int64_t *__restrict__ pos,
                                                     const int32_t *__restrict__ prev, const int32_t *__restrict__ opt,
                                                     int32_t *__restrict__ cr, int allp, int cap)
{
    // allp (windowed layers): every row takes part in every plane tau < b < nbits (an empty block holds the threshold -1 or a
    // stale value: its count is never read)
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = u < nrows;
    u += u0;
    int64_t r = ((u << 1) | 1) << tau;
    if (r > n) live = false;
    if (!live) r = 0;                                          // no bits: takes part in the ballots only
    int64_t n1 = n + 1, rL = r - ((int64_t)1 << tau);
    int32_t thr[NB], cnt[NB];
    uint32_t used = 0;                                         // planes with a set bit somewhere in the wave (uniform)
    int64_t prL = live ? PR(rL) : 0;
    // all threshold loads are issued back to back (a lane without bit b reads a valid dummy slot): one memory latency, not 23
#pragma unroll
    for (int b = 0; b < NB; b++) {
        cnt[b] = 0;
        thr[b] = ge ? INT32_MAX : INT32_MIN;
        if (b <= tau || b >= nbits) continue;                  // wave-uniform
        bool on = live && (allp || ((r >> b) & 1));
        if (!__ballot(on)) continue;                           // wave-uniform
        used |= 1u << b;
        int32_t v = opt[(int64_t)b * n1 + (on ? prL : 0)];
        if (on) thr[b] = v;
    }
    // Degrees vary a lot (the bench matrix: mean 10, one column in a hundred above 60) and a wave steps as often as its longest
    // lane: every lane takes the first `cap` entries of its row on its own, sixteen per step, and what a row has beyond them
    // is counted by the whole wave, 64 entries per coalesced load with the row's thresholds broadcast (the k_rpass_wave scheme).
    int64_t q0 = 0;
    int32_t len = 0;
    if (live) { q0 = pos[rL]; len = (int32_t)(pos[r] - q0); }
    const int32_t NEVER = ge ? INT32_MIN : INT32_MAX;          // a link value that is never counted
    const int32_t mlen = len < cap ? len : cap;
    for (int32_t o = 0; o < mlen; o += 16) {                   // sixteen entries in flight per lane
        int32_t v[16];
        // four 16-byte loads (dword-aligned: the global path takes unaligned vectors; the link arrays carry 16 entries of
        // slack) instead of sixteen 4-byte ones: every lane reads another cache line, and the L1 takes a line per lane and
        // instruction whatever its width
        struct __attribute__((packed, aligned(4))) V4 { int32_t a, b, c, d; };
        const V4 *pv = reinterpret_cast<const V4 *>(prev + q0 + o);
        const int32_t rem = mlen - o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            V4 t = {NEVER, NEVER, NEVER, NEVER};
            if (4 * j < rem) t = pv[j];
            v[4 * j] = t.a; v[4 * j + 1] = t.b; v[4 * j + 2] = t.c; v[4 * j + 3] = t.d;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = j < rem ? v[j] : NEVER;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (!((used >> b) & 1)) continue;                  // wave-uniform
            int32_t t = thr[b];
            int32_t c = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) c += ge ? (v[j] >= t) : (v[j] < t);
            cnt[b] += c;
        }
    }
    {
        const int lane = threadIdx.x & 63;
        uint32_t mypl = 0;                                     // the planes this lane's row takes part in
        if (live)
            for (int b = tau + 1; b < nbits; b++) if (allp || ((r >> b) & 1)) mypl |= 1u << b;
        uint64_t rest = __ballot(len > cap);
        const int32_t q0lo = (int32_t)(uint32_t)q0, q0hi = (int32_t)(q0 >> 32);
        while (rest) {                                         // wave-uniform
            const int src = __ffsll((unsigned long long)rest) - 1;
            rest &= rest - 1;
            const int64_t a = (((int64_t)__builtin_amdgcn_readlane(q0hi, src) << 32) | (uint32_t)__builtin_amdgcn_readlane(q0lo, src)) + cap;
            const int32_t ln = __builtin_amdgcn_readlane(len, src) - cap;
            const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int32_t)mypl, src);
            for (int32_t o = 0; o < ln; o += 64) {
                const int32_t v = o + lane < ln ? prev[a + o + lane] : NEVER;
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if (!((pl >> b) & 1)) continue;            // scalar branch
                    const int32_t t = __builtin_amdgcn_readlane(thr[b], src);
                    const int32_t c = (int32_t)__popcll(__ballot(ge ? (v >= t) : (v < t)));
                    if (lane == src) cnt[b] += c;
                }
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int b = 0; b < NB; b++)
        if (b > tau && b < nbits && (allp || ((r >> b) & 1))) cr[(int64_t)b * n1 + PR(r)] = cnt[b];
}

// Larger ranges: one wave per (row, chunk of CH columns); coalesced 64-entry loads.  The row -- hence every threshold -- is
// wave-uniform: thresholds live in SGPRs, a plane's count of 64 entries is ONE vector compare into a lane mask plus a scalar
// popcount and add (s_bcnt1 / s_add on the scalar unit), and the wave total needs no cross-lane reduction at the end.
// (Round 1 kept per-lane counters: a compare and an add per entry and plane on the vector unit, 6 shuffles per plane to finish.)
template <bool ge, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_rpass_wave(int tau, int nbits, int64_t n, int64_t u0, int64_t nrows, int chunks_per_row, int ch_cols,
                                                         const int64_t *__restrict__ pos, const int32_t *__restrict__ prev,
                                                         const int32_t *__restrict__ opt, int32_t *__restrict__ cr, int allp)
{
    // WPB > 1 (the host gives such a block WPB chunks of ONE row: chunks_per_row % WPB == 0): the waves add their counts in
    // LDS and one of them updates the row.  Thousands of chunks of a row of a high round each sending an atomic to the same
    // dozen addresses cost 35 ns apiece, one after the other: 70 us of an 85 us pass.
    int64_t w = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int64_t u = w / chunks_per_row;
    int ck = (int)(w - u * chunks_per_row);
    if (u >= nrows) return;                                    // (block-uniform when WPB > 1)
    u += u0;
    int64_t r = ((u << 1) | 1) << tau;
    if (r > n) return;
    int64_t n1 = n + 1, rL = r - ((int64_t)1 << tau);
    int64_t c0 = rL + (int64_t)ck * ch_cols, c1 = c0 + ch_cols;
    if (c1 > r) c1 = r;
    int64_t prL = PR(rL);
    // planes of this row (wave-uniform).  Lane b holds plane b: it fetches the plane's threshold (ONE vector memory instruction
    // for the row -- a uniform-address load per plane cost a dozen), keeps its count and stores it at the end.  The counting
    // loop walks the set bits of `act`: v_readlane hands the threshold to the scalar side, a plane's count of 64 entries is one
    // vector compare into a lane mask and a scalar popcount.  (Thresholds and counts in scalar register ARRAYS, one slot per
    // plane, spilled 40-110 SGPRs into vector lanes and paid a skipped branch for every plane the row does not have.)
    uint32_t act = 0;
    for (int b = tau + 1; b < nbits; b++) if (allp || ((r >> b) & 1)) act |= 1u << b;
    act = __builtin_amdgcn_readfirstlane(act);
    const bool mine = lane < 32 && ((act >> lane) & 1);
    const int32_t tv = mine ? opt[(int64_t)lane * n1 + prL] : 0;
    int32_t acc = 0;
    const int64_t q0 = pos[c0], q1 = pos[c1];
    const int32_t NEVER = ge ? INT32_MIN : INT32_MAX;          // a link value that is never counted
    // 512 entries per step as two 16-byte loads per lane, the next step's issued before this one's are counted: 4 KB in flight per
    // wave at 39 VGPRs.  (With scalar-register arrays for thresholds and counts, four 16-byte loads per lane and 1024 entries per
    // step needed 101 VGPRs -- occupancy 4, 18 % slower.)
    int32_t a[8], nx[8];
    struct __attribute__((packed, aligned(4))) V4 { int32_t x, y, z, w; };      // (16-byte loads, dword-aligned: a quarter of the memory instructions)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int64_t q = q0 + 4 * lane + 256 * k;
        V4 t = {NEVER, NEVER, NEVER, NEVER};
        if (q < q1) t = *reinterpret_cast<const V4 *>(prev + q);
        a[4 * k] = t.x; a[4 * k + 1] = q + 1 < q1 ? t.y : NEVER; a[4 * k + 2] = q + 2 < q1 ? t.z : NEVER; a[4 * k + 3] = q + 3 < q1 ? t.w : NEVER;
    }
    for (int64_t qb = q0; qb < q1; qb += 512) {                // (a wave-uniform trip count keeps the counters on the scalar unit)
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int64_t q = qb + 512 + 4 * lane + 256 * k;
            V4 t = {NEVER, NEVER, NEVER, NEVER};
            if (q < q1) t = *reinterpret_cast<const V4 *>(prev + q);
            nx[4 * k] = t.x; nx[4 * k + 1] = q + 1 < q1 ? t.y : NEVER; nx[4 * k + 2] = q + 2 < q1 ? t.z : NEVER; nx[4 * k + 3] = q + 3 < q1 ? t.w : NEVER;
        }
        for (uint32_t rem = act; rem; rem &= rem - 1) {
            const int b = __ffs(rem) - 1;
            const int32_t t = __builtin_amdgcn_readlane(tv, b);
            int32_t c = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) c += (int32_t)__popcll(__ballot(ge ? (a[k] >= t) : (a[k] < t)));
            acc += lane == b ? c : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = nx[k];
    }
    int32_t outv = acc;
    if (WPB > 1) {
        __shared__ int32_t sacc[WPB > 1 ? WPB : 1][32];
        const int wv = threadIdx.x >> 6;
        if (lane < 32) sacc[wv][lane] = mine ? outv : 0;
        __syncthreads();
        if (wv != 0) return;
        if (mine) {
            int32_t t = 0;
#pragma unroll
            for (int k = 0; k < WPB; k++) t += sacc[k][lane];
            outv = t;
        }
    }
    if (mine) {
        if (chunks_per_row == WPB) cr[(int64_t)lane * n1 + PR(r)] = outv;
        else atomicAdd(&cr[(int64_t)lane * n1 + PR(r)], outv);
    }
}

// ------------------------------------------------------------------ the last row n in the planes above its lowest bit
// anchors of its round-A tasks: out[b] = #{q in columns [r_b, n) : prev[q] < r_b}, out[32 + b] = #{rows with last column in
// [r_b, n) and first column >= r_b} (hyperedge costs), r_b = n with the bits below b cleared.  Entry q lies in a column >= r_b
// iff q >= pos[r_b].  Then the rows are marked final: the round of ctz(n) skips them.
__global__ void __launch_bounds__(256) k_last_row_counts(RoundDesc R, const int32_t *__restrict__ pos, const int32_t *__restrict__ prev,
                                                         const int32_t *__restrict__ lpos, const int32_t *__restrict__ lfirst, int32_t *__restrict__ out,
                                                         uint8_t *__restrict__ fin)
{
    __shared__ int32_t s_acc[64];
    if (threadIdx.x < 64) s_acc[threadIdx.x] = 0;
    __syncthreads();
    const int64_t n = R.n, n1 = n + 1;
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)R.nlast) fin[(int64_t)R.last_b[threadIdx.x] * n1 + prow(n, n)] = (uint8_t)R.fin_stamp;
    for (int pass = 0; pass < (lpos ? 2 : 1); pass++) {
        const int32_t *cp = pass ? lpos : pos, *arr = pass ? lfirst : prev;
        for (int i = 0; i < R.nlast; i++) {                 // (the ranges are nested; together at most 2 N entries)
            const int64_t rb = (n >> R.last_b[i]) << R.last_b[i];
            int32_t c = 0;
            for (int64_t q = (int64_t)cp[rb] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x, q1 = cp[n]; q < q1; q += (int64_t)gridDim.x * blockDim.x) {
                int32_t v = arr[q];
                c += pass ? (v >= rb) : (v < rb);
            }
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_acc[pass * 32 + i], c);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && s_acc[threadIdx.x]) {
        int pass = threadIdx.x >> 5, i = threadIdx.x & 31;
        if (i < R.nlast) atomicAdd(&out[pass * 32 + R.last_b[i]], s_acc[threadIdx.x]);
    }
}

// ------------------------------------------------------------------ task setup
// element 0 of a task is the candidate p = B (its count = anchor + right part); elements i >= 1 are the
// left steps p = B - i.  Round A: B = r is virtual (not a candidate), anchor 0.
__global__ void __launch_bounds__(256) k_setup(RoundDesc R, const int32_t *__restrict__ opt, const int32_t *__restrict__ nnopt,
                                               int32_t *__restrict__ cr, int zero_cr_only,
                                               int4 *__restrict__ tdesc, const int32_t *__restrict__ pos32,
                                               uint8_t *__restrict__ tb, int32_t *__restrict__ len,
                                               const int32_t *__restrict__ nlopt, int32_t *__restrict__ crl, int32_t *__restrict__ tS0l)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R.ntask) return;
    int64_t r; int b;
    decode_task(R, t, r, b);
    int64_t n1 = R.n + 1;
    if (zero_cr_only) { cr[(int64_t)b * n1 + PR(r)] = 0; if (crl) crl[(int64_t)b * n1 + PR(r)] = 0; return; }
    int64_t B, a, S0, S0l = 0;
    if (R.isA) {
        B = r; a = r - ((int64_t)1 << b); S0 = 0;           // (zero_cr_only is the only use of this kernel in round A)
    } else {
        int64_t rb = (r >> b) << b;
        int64_t rL = r - ((int64_t)1 << R.tau), rR = r + ((int64_t)1 << R.tau);
        B = opt[(int64_t)b * n1 + PR(rL)];
        S0 = (int64_t)nnopt[(int64_t)b * n1 + PR(rL)] + cr[(int64_t)b * n1 + PR(r)];
        if (tS0l) S0l = (int64_t)nlopt[(int64_t)b * n1 + PR(rL)] + crl[(int64_t)b * n1 + PR(r)];
        a = (rR - rb) < ((int64_t)1 << b) ? (int64_t)opt[(int64_t)b * n1 + PR(rR <= R.n ? rR : R.n)] : rb - ((int64_t)1 << b);
        if (a > B) a = B;          // cannot happen for an inverse-Monge cost; keeps every task well-formed
    }
    tdesc[t] = make_int4((int32_t)B, (int32_t)S0, (int32_t)r, pos32[r]);      // one 16-byte record per task
    tb[t] = (uint8_t)b;
    len[t] = (int32_t)(1 + (B - a));
    if (tS0l) tS0l[t] = (int32_t)S0l;
}

// ------------------------------------------------------------------ wave-level segmented scans (64 lanes)
template <typename TC, bool HYP> struct Best;
template <typename TC> struct Best<TC, false> { TC v; int32_t p; int32_t nn; };                         // 16 bytes
template <typename TC> struct Best<TC, true> { TC v; int32_t p; int32_t nn; int32_t nl; int32_t _pad; };
template <typename TC> __device__ __forceinline__ void best_clear(Best<TC, false> &b) { b.v = (TC)0; b.p = -1; b.nn = 0; }
template <typename TC> __device__ __forceinline__ void best_clear(Best<TC, true> &b) { b.v = (TC)0; b.p = -1; b.nn = 0; b.nl = 0; b._pad = 0; }
template <typename TC> __device__ __forceinline__ int32_t best_nl(const Best<TC, false> &) { return 0; }
template <typename TC> __device__ __forceinline__ int32_t best_nl(const Best<TC, true> &b) { return b.nl; }
template <typename TC> __device__ __forceinline__ void best_set_nl(Best<TC, false> &, int32_t) {}
template <typename TC> __device__ __forceinline__ void best_set_nl(Best<TC, true> &b, int32_t v) { b.nl = v; }

// (member-wise selects on values: returning `cond ? b : a` through references makes the compiler select between the two
//  ADDRESSES -- one of them often a global-memory record -- and keeps the local operand in private memory: 24-96 B of scratch
//  per lane and 0.5 GB of HBM writes per launch of the streaming kernel in round 1's profile)
template <typename TC>
__device__ __forceinline__ Best<TC, false> best_sel(bool tb, const Best<TC, false> a, const Best<TC, false> b)
{
    Best<TC, false> r; r.v = tb ? b.v : a.v; r.p = tb ? b.p : a.p; r.nn = tb ? b.nn : a.nn; return r;
}
template <typename TC>
__device__ __forceinline__ Best<TC, true> best_sel(bool tb, const Best<TC, true> a, const Best<TC, true> b)
{
    Best<TC, true> r; r.v = tb ? b.v : a.v; r.p = tb ? b.p : a.p; r.nn = tb ? b.nn : a.nn; r.nl = tb ? b.nl : a.nl; r._pad = 0; return r;
}
template <typename TC, bool HYP>
__device__ __forceinline__ Best<TC, HYP> better(const Best<TC, HYP> a, const Best<TC, HYP> b)   // a is earlier (larger p): wins ties
{
    const bool tb = a.p < 0 || (b.p >= 0 && b.v < a.v);
    return best_sel<TC>(tb, a, b);
}

__device__ __forceinline__ int64_t shfl_up64(int64_t v, int o)
{
    int lo = __shfl_up((int)(v & 0xffffffffll), o), hi = __shfl_up((int)(v >> 32), o);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ double shfl_up64(double v, int o) { return __longlong_as_double((long long)shfl_up64((int64_t)__double_as_longlong(v), o)); }
__device__ __forceinline__ int64_t shfl64(int64_t v, int src)
{
    int lo = __shfl((int)(v & 0xffffffffll), src), hi = __shfl((int)(v >> 32), src);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ double shfl64(double v, int src) { return __longlong_as_double((long long)shfl64((int64_t)__double_as_longlong(v), src)); }

// inclusive segmented sum: v = sum from the last head at or before this lane (or lane 0), f = "a head lies in [0, lane]"
__device__ __forceinline__ void wave_segsum(int32_t &v, int &f, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int32_t pv = __shfl_up(v, o);
        int pf = __shfl_up(f, o);
        if (lane >= o) { if (!f) v += pv; f |= pf; }
    }
}

template <typename TC, bool HYP>
__device__ __forceinline__ void wave_segmin(Best<TC, HYP> &x, int &f, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Best<TC, HYP> p;
        best_clear(p);
        p.v = shfl_up64(x.v, o);
        p.p = __shfl_up(x.p, o);
        p.nn = __shfl_up(x.nn, o);
        if (HYP) best_set_nl(p, __shfl_up(best_nl(x), o));
        int pf = __shfl_up(f, o);
        if (lane >= o) { if (!f) x = better(p, x); f |= pf; }
    }
}

// ------------------------------------------------------------------ task setup, second form: short tasks finish here
// One lane per task.  A task with at most `short_t` candidates and at most `short_e` link entries to step over is finished
// by its lane on the spot (at low tau more than half of all tasks have ONE candidate: B == a, nothing to compare); the
// rest are appended -- in order within a block -- to the list of long tasks that the flattened kernels below process.
// (cp_set_option("short_t" / "short_e"): tunables, defaults chosen on config 3.)
template <typename TC, bool HYP>
__global__ void __launch_bounds__(1024) k_setup_short(RoundDesc R, int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int32_t *__restrict__ nlopt,
                                                      const int32_t *__restrict__ cr, const int32_t *__restrict__ crl,
                                                      const int32_t *__restrict__ pos32, const int32_t *__restrict__ next,
                                                      const int32_t *__restrict__ fpos32, const int32_t *__restrict__ flast,
                                                      const TC *__restrict__ W, DevModel<TC> M, TC alpha,
                                                      int4 *__restrict__ tdesc, uint8_t *__restrict__ tb, int32_t *__restrict__ len,
                                                      int32_t *__restrict__ tS0l, int32_t *__restrict__ nlong, int32_t SHORT_T, int32_t SHORT_E,
                                                      int4 *__restrict__ o_tdesc, uint8_t *__restrict__ o_tb, int32_t *__restrict__ o_rlen,
                                                      int32_t *__restrict__ o_ntl, int32_t *__restrict__ o_tS0l, int32_t *__restrict__ n_own,
                                                      unsigned long long *__restrict__ own_steps, int32_t OWN_MIN, int32_t o_cap,
                                                      int32_t *__restrict__ err, const uint8_t *__restrict__ fin, const int32_t *__restrict__ last_s0,
                                                      int32_t *__restrict__ dbg_ntriv, const int32_t *__restrict__ anch, const int32_t *__restrict__ anch2,
                                                      int32_t *__restrict__ pz)
{
    // pz (cp_set_option("poison", 1)): the planes were filled with an out-of-range column before the layer; a task whose bounds
    // come out of range has read a cell nobody wrote -- counted, and clamped so that the task stays well-formed
    int lane = threadIdx.x & 63;
    bool live;
    bool is_long = false, is_own = false, is_short = false;
    int64_t r = 0, B = 0, a = 0, S0 = 0, S0l = 0; int b = 0;
    uint8_t finb = 0;
    if (R.isA) {
        int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        live = t < R.ntask;
        if (live) decode_task(R, t, r, b);
    } else {
        // tau rounds: blockIdx.y is the bit plane (no per-thread search for it), blockIdx.x * 1024 + thread the task inside the plane
        b = R.tau + 1 + (int)blockIdx.y;
        int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        live = l < R.tbase[b + 1] - R.tbase[b];
        if (live) {
            l += R.tskip[b];
            if (R.G.win) r = ((l << 1) | 1) << R.tau;            // windowed layers: every row with ctz == tau, in every plane
            else {
            int sh = b - R.tau - 1;
            int64_t base = l >> sh, v = l & (((int64_t)1 << sh) - 1);
            r = (base << (b + 1)) | ((int64_t)1 << b) | (((v << 1) | 1) << R.tau);
            }
            // (rows of a tau round have ctz == tau: their plane slot needs no bit search)
            // finished by a gap pass of an earlier round?  (the flag is looked at AFTER the gathers below are on their way: one memory latency less)
            if (fin) finb = fin[(int64_t)b * (R.n + 1) + (R.n - (R.n >> R.tau)) + (r >> (R.tau + 1))] == (uint8_t)R.fin_stamp;
        }
    }
    if (live && R.G.win) {
        // windowed geometry: the block of (r, b); round A: the head row of a rectangle scans the whole block below the virtual
        // candidate B = block end, anchor = nets(block end, r) (0 for the standard and common blocks, which end at r; cached per
        // partition for the mirrored ones); round tau: bounds from the tree neighbours inside the row block of 2^b rows
        const int64_t n1 = R.n + 1;
        int64_t cs, ce;
        if (!geo_block(R.G, r, b, cs, ce)) live = false;
        else if (R.isA && R.skip_std && b < R.G.s && ((r >> b) & 1)) live = false;      // (a standard head: done by k_ra_cols)
        else if (R.isA && R.skip_mir && b < R.G.s && !((r >> b) & 1)) live = false;     // (a mirrored head: done by k_ra_layer on the mirrored table)
        else if (R.isA) {
            B = ce; a = cs;
            if (ce != r) { S0 = anch[R.aoff[b] + (r >> (b + 1))]; if (HYP) S0l = anch2[R.aoff[b] + (r >> (b + 1))]; }
        } else {
            const int64_t rL = r - ((int64_t)1 << R.tau), rR = r + ((int64_t)1 << R.tau), rect_hi = ((r >> b) + 1) << b;
            B = opt[(int64_t)b * n1 + PR(rL)];
            S0 = (int64_t)nnopt[(int64_t)b * n1 + PR(rL)] + cr[(int64_t)b * n1 + PR(r)];
            if (HYP) S0l = (int64_t)nlopt[(int64_t)b * n1 + PR(rL)] + crl[(int64_t)b * n1 + PR(r)];
            a = (rR < rect_hi && rR <= R.n) ? (int64_t)opt[(int64_t)b * n1 + PR(rR)] : cs;
            if (a > B) a = B;
        }
    }
    if (live && !R.G.win) {
        int64_t n1 = R.n + 1;
        if (R.isA) {
            // B = r_b is virtual (not a candidate).  Rows r == r_b: nothing lies right of B, anchor 0; the last row n in a
            // plane above its lowest bit: the anchor is the count of the columns [r_b, n) (k_last_row_counts)
            B = (r >> b) << b; a = B - ((int64_t)1 << b);
            if (B != r) { S0 = last_s0[b]; if (HYP) S0l = last_s0[32 + b]; }
        } else {
            int64_t rb = (r >> b) << b;
            int64_t rL = r - ((int64_t)1 << R.tau), rR = r + ((int64_t)1 << R.tau);
            B = opt[(int64_t)b * n1 + PR(rL)];
            S0 = (int64_t)nnopt[(int64_t)b * n1 + PR(rL)] + cr[(int64_t)b * n1 + PR(r)];
            if (HYP) S0l = (int64_t)nlopt[(int64_t)b * n1 + PR(rL)] + crl[(int64_t)b * n1 + PR(r)];
            // left end of the range: the winner of the right neighbour -- or, where that lies beyond the matrix, of the last
            // row n (same rectangle; known since round A); the block start for the last row of a rectangle
            a = (rR - rb) < ((int64_t)1 << b) ? (int64_t)opt[(int64_t)b * n1 + PR(rR <= R.n ? rR : R.n)] : rb - ((int64_t)1 << b);
            if (a > B) a = B;          // cannot happen for an inverse-Monge cost; keeps every task well-formed
        }
    }
    if (finb) live = false;
    if (live && pz && !R.isA && ((uint64_t)B > (uint64_t)R.n || (uint64_t)a > (uint64_t)R.n)) {
        atomicAdd(pz, 1);
        B = B < 0 ? 0 : (B > R.n ? R.n : B); a = a < 0 ? 0 : (a > B ? B : a);
    }
    if (live) {
        int64_t L = 1 + (B - a);
        if (dbg_ntriv && L > 1 && !R.isA) atomicAdd(dbg_ntriv, 1);
        is_short = L <= SHORT_T && (pos32[B] - pos32[a]) <= SHORT_E && (!HYP || (fpos32[B] - fpos32[a]) <= SHORT_E);
        if (is_short) {
        } else if (o_tdesc && L >= OWN_MIN) {
            is_own = true;                           // long enough for tiles of its own (k_lpass_own): every tile is uniform
        } else {
            is_long = true;
        }
    }
    // append the remaining tasks to their list -- the flattened one, or the one of tasks with tiles of their own: ONE atomic per
    // list and 1024-lane block (same-address atomics serialise in L2), order kept inside the block
    __shared__ int32_t s_wcnt[2][16];
    __shared__ unsigned long long s_wsteps[16];
    __shared__ int32_t s_base[2];
    unsigned long long ml = __ballot(is_long), mo = __ballot(is_own);
    int wv = threadIdx.x >> 6;
    int64_t Lmine = is_own ? 1 + (B - a) : 0;
    unsigned long long lsum = 0;
    if (mo) {                                                        // wave-uniform
        lsum = (unsigned long long)Lmine;
        for (int o = 32; o > 0; o >>= 1) lsum += ((unsigned long long)(uint32_t)__shfl_down((int)(lsum >> 32), o) << 32) | (uint32_t)__shfl_down((int)(lsum & 0xffffffffull), o);
    }
    if (lane == 0) { s_wcnt[0][wv] = (int32_t)__popcll(ml); s_wcnt[1][wv] = (int32_t)__popcll(mo); s_wsteps[wv] = lsum; }
    __syncthreads();
    int32_t base0 = 0, base1 = 0;
    if (threadIdx.x == 0) {
        int32_t t0 = 0, t1 = 0; unsigned long long st = 0;
        for (int w = 0, nw = (int)(blockDim.x >> 6); w < nw; w++) {
            int32_t c0 = s_wcnt[0][w], c1 = s_wcnt[1][w];
            s_wcnt[0][w] = t0; s_wcnt[1][w] = t1; t0 += c0; t1 += c1; st += s_wsteps[w];
        }
        // (the two list positions come back from L2 while this lane, like all the others, finishes its short task below: the
        //  block used to sit at a barrier for the length of these round trips)
        base0 = t0 ? atomicAdd(nlong, t0) : 0;
        base1 = t1 ? atomicAdd(n_own, t1) : 0;
        if (st) atomicAdd(own_steps, st);
    }
    // the short tasks are finished by their lanes -- between the two barriers, while the list positions are on their way
    if (is_short) {
        const int64_t n1 = R.n + 1, L = 1 + (B - a);
        int64_t rw = (int64_t)b * n1 + PR(r);
        if (L == 1 && !R.isA) {                  // one candidate: nothing to compare
            opt[rw] = (int32_t)B; nnopt[rw] = (int32_t)S0; if (HYP) nlopt[rw] = (int32_t)S0l;
        } else {
            // The walk is a chain of dependent loads (the kernel spends 2/3 of its wave cycles waiting on memory): the column pointer and
            // the previous-layer cost of the NEXT candidate are requested before the current column is stepped over, and a column's
            // entries go eight at a time -- one memory latency per candidate instead of four or five.
            const int32_t rr = (int32_t)r, posr = pos32[r];
            int64_t nn = S0, nl = S0l;
            Best<TC, HYP> best; best_clear(best);
            int32_t pp = pos32[B], pq = 0, fp = HYP ? fpos32[B] : 0, fq = 0;      // colptr of the candidate (pp, fp) and of the column after it (pq, fq)
            TC wp = W[B];
            for (int64_t i = 0; i < L; i++) {    // decreasing p: an earlier candidate wins ties
                const int64_t p = B - i;
                int32_t pn = 0, fn = 0; TC wn = (TC)0;
                if (i + 1 < L) { pn = pos32[p - 1]; wn = W[p - 1]; if (HYP) fn = fpos32[p - 1]; }
                if (i > 0) {                     // step over column p: its entries are [pp, pq)
                    for (int32_t q = pp; q < pq; q += 8) {
                        int32_t v[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) v[k] = q + k < pq ? next[q + k] : -1;
#pragma unroll
                        for (int k = 0; k < 8; k++) nn += (v[k] >= rr);
                    }
                    if (HYP) for (int32_t q = fp; q < fq; q++) nl += (flast[q] < rr);
                }
                if (!(i == 0 && R.isA)) {        // (round A: p = r is not a candidate)
                    TC fv = dm_apply(M, alpha, r - p, (int64_t)(posr - pp), nn, nl);
                    Best<TC, HYP> c; best_clear(c); c.v = cadd(wp, fv); c.p = (int32_t)p; c.nn = (int32_t)nn; best_set_nl(c, (int32_t)nl);
                    best = better(best, c);
                }
                pq = pp; pp = pn; wp = wn; fq = fp; fp = fn;
            }
            opt[rw] = best.p; nnopt[rw] = best.nn; if (HYP) nlopt[rw] = best_nl(best);
        }
    }
    if (threadIdx.x == 0) { s_base[0] = base0; s_base[1] = base1; }
    __syncthreads();
    if (is_long) {
        int32_t idx = s_base[0] + s_wcnt[0][wv] + __popcll(ml & ((1ull << lane) - 1ull));
        tdesc[idx] = make_int4((int32_t)B, (int32_t)S0, (int32_t)r, pos32[r]);
        tb[idx] = (uint8_t)b;
        len[idx] = (int32_t)(1 + (B - a));
        if (HYP) tS0l[idx] = (int32_t)S0l;
    }
    if (is_own) {
        int32_t idx = s_base[1] + s_wcnt[1][wv] + __popcll(mo & ((1ull << lane) - 1ull));
        if (idx >= o_cap) { *err = 1; return; }          // (an own task is never short) cannot happen (LayerWork sizes the list for the worst case); never write outside
        o_tdesc[idx] = make_int4((int32_t)B, (int32_t)S0, (int32_t)r, pos32[r]);
        o_tb[idx] = (uint8_t)b;
        o_rlen[idx] = (int32_t)Lmine;
        o_ntl[idx] = (int32_t)((Lmine + LT - 1) / LT);
        if (HYP) o_tS0l[idx] = (int32_t)S0l;
    }
}

// ------------------------------------------------------------------ tile task table (one wave = one tile)
// first task overlapping each tile: last t with offs[t] <= tile*LT (one thread per tile; offs[0] = 0)
__global__ void __launch_bounds__(256) k_tile_t0(const RoundCounts *__restrict__ rc, const int64_t *__restrict__ offs,
                                                 const int4 *__restrict__ tdesc, int64_t *__restrict__ tile_t0, int4 *__restrict__ tile_rec, int no_interior,
                                                 int64_t *__restrict__ taskR)
{
    const int64_t ntask = rc->nlong, ntile = rc->ntile, T = rc->T;
    for (int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tile < ntile; tile += (int64_t)gridDim.x * blockDim.x) {
    taskR[tile] = -1;                                  // "no task continues beyond this tile" until k_lpass says otherwise
    int64_t tile_start = tile * LT;
    int64_t lo = 0, hi = ntask;
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (offs[mid] <= tile_start) lo = mid; else hi = mid;
    }
    tile_t0[tile] = lo;
    // interior tile: every step belongs to ONE task whose head lies in an earlier tile.  Record {first column, row, pos[row], 1}.
    int64_t toff = offs[lo], nx = offs[lo + 1], tend = tile_start + LT < T ? tile_start + LT : T;
    int4 rec = make_int4(0, 0, 0, 0);
    if (toff < tile_start && nx >= tend && !no_interior) {
        int4 td = tdesc[lo];
        rec = make_int4(td.x - (int32_t)(tile_start - toff), td.z, td.w, 1);
    }
    tile_rec[tile] = rec;
    }
}

// loads the offsets (relative to the tile start, 32-bit) of the tasks overlapping the tile and a bitmap of the
// task heads inside it
__device__ __forceinline__ void load_tile_tasks(const int64_t *__restrict__ offs, int64_t ntask, int64_t t0, int64_t tile_start,
                                                bool active, int32_t *s_off, unsigned long long *s_hd, int lane, int &cnt)
{
    int64_t c = ntask - t0;                  // every task has len >= 1: at most LT tasks start inside the tile
    cnt = active ? (int)(c > LT + 1 ? LT + 1 : c) : 0;
    if (lane < LT / 64) s_hd[lane] = 0ull;
    for (int i = lane; i < cnt; i += 64) {
        int64_t rel = offs[t0 + i] - tile_start;
        if (rel < -(int64_t)0x3fffffff) rel = -(int64_t)0x3fffffff;
        if (rel > (int64_t)0x3fffffff) rel = (int64_t)0x3fffffff;
        s_off[i] = (int32_t)rel;
        if (rel >= 0 && rel < LT) atomicOr(&s_hd[rel >> 6], 1ull << (rel & 63));
    }
}

// ------------------------------------------------------------------ cooperative column counts
// Every lane owns one column-like range [s, en) of `arr` and a threshold; lanes whose ranges are adjacent and
// descending form a run whose entries [q_lo, q_hi) are read as aligned 16-byte-per-lane loads (1 KiB per wave
// instruction, two in flight); position x + 4*lane' + j sits in component j of lane'.  Returns, per lane,
// #{entries of its range with  v >= thr (GE)  or  v < thr (!GE)}.  One ballot pass per distinct threshold
// (= task) touching a 256-entry block.
template <bool GE>
__device__ __forceinline__ int32_t coop_count(const int32_t *__restrict__ arr, int32_t s, int32_t en, int32_t thr, bool valid, int lane)
{
    int32_t prev_s = __shfl_up(s, 1);
    int prev_valid = __shfl_up((int)valid, 1);
    bool cont = valid && lane > 0 && prev_valid && (en == prev_s);
    unsigned long long heads = __ballot(valid && !cont);
    unsigned long long vm = __ballot(valid);
    int32_t d = 0;
    const int32_t FILL = GE ? INT32_MIN : INT32_MAX;          // never counted
    while (heads) {
        int h0 = __ffsll((long long)heads) - 1;
        heads &= heads - 1;
        int h1 = heads ? (__ffsll((long long)heads) - 1) : 64;
        unsigned long long inrun = vm & (h1 == 64 ? ~0ull : ((1ull << h1) - 1)) & ~((1ull << h0) - 1);
        int hl = 63 - __clzll((long long)inrun);          // last valid lane of the run
        int32_t q_hi = __shfl(en, h0), q_lo = __shfl(s, hl);
        bool mine = lane >= h0 && lane <= hl && en > s;
        for (int32_t x = q_lo & ~3; x < q_hi; x += 512) {
            int4 v0 = make_int4(FILL, FILL, FILL, FILL), v1 = v0;
            int32_t b0 = x + 4 * lane, b1 = b0 + 256;
            if (b0 < q_hi) v0 = *reinterpret_cast<const int4 *>(arr + b0);        // arrays are padded by 8 entries
            if (b1 < q_hi) v1 = *reinterpret_cast<const int4 *>(arr + b1);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                int32_t xc = x + c * 256;
                if (xc >= q_hi) break;                    // wave-uniform
                int4 v = c ? v1 : v0;
                // (entries outside the run need no masking: a lane only counts inside its own range, which lies in the run)
                bool ov = mine && s < xc + 256 && en > xc;
                // this lane's range covers positions [a0, a1) of the block: lanes [lo_j, hi_j) of component j
                int a0 = ov ? ((s > xc ? s : xc) - xc) : 0, a1 = ov ? ((en < xc + 256 ? en : xc + 256) - xc) : 0;
                unsigned long long rem = __ballot(ov);
                while (rem) {                             // one pass per distinct threshold touching the block
                    int l = __ffsll((long long)rem) - 1;
                    int32_t tu = __shfl(thr, l);
                    const bool f0 = GE ? v.x >= tu : v.x < tu, f1 = GE ? v.y >= tu : v.y < tu, f2 = GE ? v.z >= tu : v.z < tu, f3 = GE ? v.w >= tu : v.w < tu;
                    const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2), m3 = __ballot(f3);
                    // F(x) = flagged entries at block positions < x = (flags in the lanes below lane x/4) + (flags of that lane below
                    // component x%4): the lane-below counts come from v_mbcnt chains, packed with the lane's own flags, and a lane
                    // takes  F(a1) - F(a0)  for its range [a0, a1) with two cross-lane reads
                    uint32_t P0 = 0;
                    P0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, P0));
                    P0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, P0));
                    P0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, P0));
                    P0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m3, P0));
                    const int32_t pack = (int32_t)((P0 << 3) | (uint32_t)f0 | ((uint32_t)f1 << 1) | ((uint32_t)f2 << 2));
                    const int32_t tot = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);         // (uniform)
                    const int32_t g0 = __shfl(pack, (a0 >> 2) & 63), g1 = __shfl(pack, (a1 >> 2) & 63);      // (every lane takes part)
                    bool same = ov && thr == tu;
                    if (same) {
                        int32_t F0 = (g0 >> 3) + __popc((uint32_t)g0 & ((1u << (a0 & 3)) - 1u));
                        int32_t F1 = a1 >= 256 ? tot : (g1 >> 3) + __popc((uint32_t)g1 & ((1u << (a1 & 3)) - 1u));
                        d += F1 - F0;
                    }
                    rem &= ~__ballot(same);
                }
            }
        }
    }
    return d;
}

// Interior tiles of a long task: the tile's steps e = 0 .. tl are adjacent descending columns p_first - e of ONE task, so
// the step order is the (reversed) entry order of one contiguous run [Q_lo, Q_hi) and the inclusive step prefix is a
// suffix count:  x(e) = #{q in [s(e), Q_hi) : flagged(q)},  s(e) = first entry of the step's column.  No per-column
// counts, no wave scan, no carry.  The run is streamed once in 256-entry blocks (the next 512 entries are in flight while
// a block is counted); per block: four ballots, a per-lane popcount prefix, and one gather per step whose column starts
// inside the block.  Lane l owns the steps l, l+64, l+128, l+192.
// DET (gap passes, see k_gap_finish): the entries of the run with a value strictly between sp_lo and sp_hi ("specials") are
// appended to the wave's list by the lanes that hold them (LDS counter s_cnt): s_es[i] = the entry's position (the caller turns it
// into the first step whose candidate has the entry's column on its right), s_v[i] = the value, tagged with `kind` in bit 31.
// Only the first SMAX specials are stored; s_cnt keeps counting.
// (Tried: a step's count taken once, in the block its column starts in, as TOTAL - (flagged entries below the start), the groups
//  that can start in a block picked on the scalar side -- 30 instead of 70 vector instructions per block, but 65 VGPRs: 254 ms
//  against 232; forced back to 64 VGPRs: 241 ms.)
// (Tried: the link values in 3 bytes (n < 2^24), a 12-byte load and six vector instructions to unpack four entries -- 19 % fewer
//  bytes per step, and k_lpass_own 9 % SLOWER (254 vs 232 ms per partition): at 0.8 of the HBM roofline the kernel has no vector
//  issue slots to spare.)
constexpr int SMAX = 31;
template <bool GE, bool DET = false>
__device__ __forceinline__ void interior_stream(const int32_t *__restrict__ arr, const int32_t *__restrict__ cpos, int32_t p_first, int32_t tl,
                                                int32_t thr, int lane, int32_t acc[4], int32_t sk[4], int head = 0, int32_t sp_lo = 0, int32_t sp_hi = 0,
                                                int32_t *s_es = nullptr, int32_t *s_v = nullptr, int32_t *s_cnt = nullptr, int kind = 0)
{
    const int32_t FILL = GE ? INT32_MIN : INT32_MAX;      // never flagged
#pragma unroll
    for (int k = 0; k < 4; k++) { int32_t e = lane + 64 * k; sk[k] = e <= tl ? cpos[p_first - e] : INT32_MAX; acc[k] = 0; }
    // head != 0: step 0 is the candidate p_first itself (no column stepped over): the run ends in front of that column
    int32_t Q_hi = cpos[p_first + 1 - head], Q_lo = cpos[p_first - tl];       // wave-uniform
    int32_t x = Q_lo & ~3;
    int4 c0 = make_int4(FILL, FILL, FILL, FILL), c1 = c0;
    if (x + 4 * lane < Q_hi) c0 = *reinterpret_cast<const int4 *>(arr + x + 4 * lane);          // arrays are padded by 8 entries
    if (x + 256 + 4 * lane < Q_hi) c1 = *reinterpret_cast<const int4 *>(arr + x + 256 + 4 * lane);
    for (; x < Q_hi; x += 512) {
        int4 n0 = make_int4(FILL, FILL, FILL, FILL), n1 = n0;
        if (x + 512 + 4 * lane < Q_hi) n0 = *reinterpret_cast<const int4 *>(arr + x + 512 + 4 * lane);
        if (x + 768 + 4 * lane < Q_hi) n1 = *reinterpret_cast<const int4 *>(arr + x + 768 + 4 * lane);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            int32_t xc = x + c * 256;
            if (xc >= Q_hi) break;                        // wave-uniform
            int4 v = c ? c1 : c0;
            int32_t pb = xc + 4 * lane;
            // entries at or above Q_hi belong to columns above the tile: never flagged (those below Q_lo lie below every s)
            // the lane's four flags, and the four wave masks
            // (a block that lies inside the run -- all but the first and the last one -- needs none of the position guards: wave-uniform)
            const bool inner = xc >= Q_lo && xc + 256 <= Q_hi;
            bool f0 = GE ? v.x >= thr : v.x < thr, f1 = GE ? v.y >= thr : v.y < thr, f2 = GE ? v.z >= thr : v.z < thr, f3 = GE ? v.w >= thr : v.w < thr;
            if (!inner) { f0 = f0 && pb < Q_hi; f1 = f1 && pb + 1 < Q_hi; f2 = f2 && pb + 2 < Q_hi; f3 = f3 && pb + 3 < Q_hi; }
            const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2), m3 = __ballot(f3);
            if (DET) {
                const uint32_t span = (uint32_t)(sp_hi - sp_lo - 1);           // v in (sp_lo, sp_hi)  <=>  (uint)(v - sp_lo - 1) < span
                bool s0 = (uint32_t)(v.x - sp_lo - 1) < span, s1 = (uint32_t)(v.y - sp_lo - 1) < span, s2 = (uint32_t)(v.z - sp_lo - 1) < span,
                     s3 = (uint32_t)(v.w - sp_lo - 1) < span;
                if (!inner) {
                    s0 = s0 && pb >= Q_lo && pb < Q_hi; s1 = s1 && pb + 1 >= Q_lo && pb + 1 < Q_hi;
                    s2 = s2 && pb + 2 >= Q_lo && pb + 2 < Q_hi; s3 = s3 && pb + 3 >= Q_lo && pb + 3 < Q_hi;
                }
                if (s0 || s1 || s2 || s3) {                           // rare; only the lanes holding a special work here
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (j == 0 ? s0 : j == 1 ? s1 : j == 2 ? s2 : s3) {
                            int32_t val = j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
                            int slot = atomicAdd(s_cnt, 1);
                            if (slot <= SMAX - 1) {
                                s_es[slot] = pb + j;                  // (the position; its column is looked up after the stream: no load stalls the loop)
                                s_v[slot] = (int32_t)(((uint32_t)val & 0x7fffffffu) | ((uint32_t)kind << 31));
                            }
                        }
                    }
                }
            }
            int32_t tot = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);                       // uniform
            // flagged entries in the lanes below this one (v_mbcnt chains), packed with the lane's own first three flags: one
            // cross-lane read then gives a step everything it needs about the lane its column starts in
            uint32_t P0 = 0;
            P0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, P0));
            P0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, P0));
            P0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, P0));
            P0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m3, P0));
            const int32_t pack = (int32_t)((P0 << 3) | (uint32_t)f0 | ((uint32_t)f1 << 1) | ((uint32_t)f2 << 2));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int32_t idx = sk[k] - xc;                 // block position of the step's first entry (huge for invalid steps)
                bool inside = idx > 0 && idx < 256;
                if (__ballot(inside)) {                   // wave-uniform: some column of this group starts inside the block
                    int32_t cidx = inside ? idx : 0;
                    int src = cidx >> 2, comp = cidx & 3;
                    const int32_t g = __shfl(pack, src);
                    int32_t Fb = (g >> 3) + __popc((uint32_t)g & ((1u << comp) - 1u));        // flagged entries of the block below position idx
                    acc[k] += idx <= 0 ? tot : (inside ? tot - Fb : 0);
                } else {
                    acc[k] += idx <= 0 ? tot : 0;
                }
            }
        }
        c0 = n0; c1 = n1;
    }
}

// ------------------------------------------------------------------ long tasks with tiles of their own
// A task with >= LT steps is cut into tiles counted from ITS OWN head (the last one partial): every tile lies in one task,
// so all of them -- head and tail included -- take the uniform path: one contiguous run of the link array, suffix counts,
// tile-local evaluation.  k_own_map: tile -> (task, tile index inside the task); k_lpass_own: one wave per tile;
// k_fix_own: one wave per task merges its tiles (adding the counts made before each tile).
__global__ void __launch_bounds__(256) k_own_map(const RoundCounts *__restrict__ rc, const int64_t *__restrict__ toffs, const int4 *__restrict__ tdesc,
                                                 const int32_t *__restrict__ rlen, int4 *__restrict__ rec, int32_t *__restrict__ tile_task,
                                                 const uint8_t *__restrict__ tb, int32_t *__restrict__ gap_hi, int tau, int64_t n)
{
    const int64_t ntask = rc->nown, ntile = rc->NT;
    for (int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tile < ntile; tile += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = ntask;                       // last task with toffs[task] <= tile
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (toffs[mid] <= tile) lo = mid; else hi = mid;
    }
    int32_t kt = (int32_t)(tile - toffs[lo]);
    int4 td = tdesc[lo];
    int32_t rest = rlen[lo] - kt * LT;                // steps of the task from this tile on
    int32_t tl = (rest < LT ? rest : LT) - 1;
    // {column of step 0, row, pos[row], tl | tile of a gap task | head}
    const bool isgap = gap_hi != nullptr;
    rec[tile] = make_int4(td.x - kt * LT, td.z, td.w, (tl << 2) | (isgap ? 2 : 0) | (kt == 0 ? 1 : 0));
    tile_task[tile] = (int32_t)lo;
    if (isgap) {                                       // gap pass: rows (r - 2^tau, hi) of the rectangle are finished together
        int64_t r = td.z, b = tb[lo];
        int64_t hi = r + ((int64_t)1 << tau), re = (((r >> b) + 1) << b);
        if (hi > re) hi = re;
        if (hi > n + 1) hi = n + 1;
        gap_hi[tile] = (int32_t)hi;
    }
    }
}

template <typename TC, bool HYP, bool GAP>
__global__ void __launch_bounds__(256) k_lpass_own(int isA, const RoundCounts *__restrict__ rc, const int32_t *__restrict__ a_pos, const int32_t *__restrict__ a_next,
                                                   const int32_t *__restrict__ a_fpos, const int32_t *__restrict__ a_flast,
                                                   int32_t *__restrict__ a_tileS, int32_t *__restrict__ a_tileS2, const int4 *__restrict__ a_rec,
                                                   const TC *__restrict__ W, DevModel<TC> M, TC alpha, Best<TC, HYP> *__restrict__ part,
                                                   int tau, const int32_t *__restrict__ gap_hi, uint8_t *__restrict__ spec, int force_spec,
                                                   Best<TC, HYP> *__restrict__ sub, int32_t *__restrict__ spv,
                                                   const int32_t *__restrict__ a_col, const int32_t *__restrict__ a_ffirst)
{
    // GAP (rounds tau <= gap_tau, see k_gap_finish): the tile is evaluated for ALL rows r' of (r - 2^tau, hi) at once: the counts
    // taken here are those of the entries every such row counts (next >= hi - 1; last <= r - 2^tau), the candidates are valued
    // with the task's own row (the rows differ by terms that do not depend on the candidate), and `spec` says whether the tile
    // holds an entry that only some of the rows count.
    // (Tried: tiles visited in column order, one contiguous share of the order per XCD, so that the ~10 passes of the bit planes
    //  over a column meet in L2: 10 % off this kernel, less than the sort of the tile list costs.)
    // (No grid-stride loop here: it costs 14 VGPRs = two waves per SIMD.  The host launches one wave per tile of the buffer's
    //  CAPACITY, which k_round_finish has checked the true count against.)
    int lane = threadIdx.x & 63;
    int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= rc->NT) return;
    int4 rec = a_rec[tile];
    int head = rec.w & 1;
    int32_t tl = rec.w >> 2;
    int32_t acc[4], acc2[4] = {0, 0, 0, 0}, sk[4], sk2[4];
    __shared__ int32_t s_es_all[4][2][SMAX + 1], s_v_all[4][2][SMAX + 1];      // the wave's specials: as found / sorted by step
    __shared__ int32_t s_cnt_all[4];
    int32_t(*s_es)[SMAX + 1] = s_es_all[threadIdx.x >> 6], (*s_v)[SMAX + 1] = s_v_all[threadIdx.x >> 6];
    int ns = 0;
    if (GAP && (rec.w & 2)) {
        int32_t *s_cnt = &s_cnt_all[threadIdx.x >> 6];
        if (lane == 0) *s_cnt = 0;
        __threadfence_block();
        int32_t rL = rec.y - (1 << tau), hi1 = gap_hi[tile] - 1;
        interior_stream<true, true>(a_next, a_pos, rec.x, tl, hi1, lane, acc, sk, head, rL, hi1, s_es[0], s_v[0], s_cnt, 0);
        if (HYP) interior_stream<false, true>(a_flast, a_fpos, rec.x, tl, rL + 1, lane, acc2, sk2, head, rL, hi1, s_es[0], s_v[0], s_cnt, 1);
        __threadfence_block();
        ns = *s_cnt;
        if (force_spec) ns = SMAX + 1;
        if (ns > 0 && ns <= SMAX) {                     // positions -> steps: first step whose candidate has the entry's column on its right
            if (lane < ns) {
                const int32_t q = s_es[0][lane];
                const bool k1 = s_v[0][lane] < 0;       // (bit 31: an entry of the second list)
                s_es[0][lane] = rec.x - (k1 ? a_ffirst[q] : a_col[q]);
            }
            __threadfence_block();
        }
    } else {
        interior_stream<true>(a_next, a_pos, rec.x, tl, rec.y, lane, acc, sk, head);
        if (HYP) interior_stream<false>(a_flast, a_fpos, rec.x, tl, rec.y, lane, acc2, sk2, head);
    }
    int sel = tl >> 6;
    if (lane == (tl & 63)) {
        a_tileS[tile] = sel == 0 ? acc[0] : sel == 1 ? acc[1] : sel == 2 ? acc[2] : acc[3];
        if (HYP) a_tileS2[tile] = sel == 0 ? acc2[0] : sel == 1 ? acc2[1] : sel == 2 ? acc2[2] : acc2[3];
    }
    // tile-local evaluation (the costs are affine in the counts; k_fix_own adds the counts made before the tile)
    Best<TC, HYP> cand[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int32_t e = lane + 64 * k;
        best_clear(cand[k]);
        if (e <= tl && !(head && isA && e == 0)) {     // round A: the head element p = r is not a candidate
            int32_t p = rec.x - e;
            TC fv = dm_apply(M, alpha, (int64_t)(rec.y - p), (int64_t)(rec.z - sk[k]), (int64_t)acc[k], (int64_t)acc2[k]);
            cand[k].v = cadd(W[p], fv); cand[k].p = p; cand[k].nn = acc[k]; best_set_nl(cand[k], acc2[k]);
        }
    }
    if (GAP && ns > SMAX) {                             // too many specials: k_gap_finish walks the tile entry by entry
        if (lane == 0) spec[tile] = 255;
        return;
    }
    if (!GAP || ns == 0) {                              // one winner
        Best<TC, HYP> best; best_clear(best);
#pragma unroll
        for (int k = 0; k < 4; k++) best = better(best, cand[k]);       // increasing e = decreasing p: an earlier candidate wins ties
        for (int o = 32; o > 0; o >>= 1) {              // wave arg-min; ties -> larger p
            int src = (lane + o) & 63;
            Best<TC, HYP> c; best_clear(c); c.v = shfl64(best.v, src); c.p = __shfl(best.p, src); c.nn = __shfl(best.nn, src);
            if (HYP) best_set_nl(c, __shfl(best_nl(best), src));
            bool take = (best.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < best.v || (c.v == best.v && c.p > best.p)));
            if (lane + o < 64 && take) best = c;
        }
        if (lane == 0) { part[tile] = best; if (GAP) spec[tile] = 0; }
        return;
    }
    // GAP with specials: they cut the tile into segments of candidates that see the same specials on their right; every
    // segment keeps a winner of its own (the rows weigh the segments differently): one segmented arg-min scan.
    __threadfence_block();                              // (lane 0 wrote the list)
    if (lane < ns) {                                    // sort by step: rank = number of specials in front
        int32_t es_i = s_es[0][lane]; int rank = 0;
        for (int j = 0; j < ns; j++) { int32_t es_j = s_es[0][j]; rank += (es_j < es_i) || (es_j == es_i && j < lane); }
        s_es[1][rank] = es_i; s_v[1][rank] = s_v[0][lane];
    }
    __threadfence_block();
    int seg[4] = {0, 0, 0, 0}, hd[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < ns; i++) {
        int32_t es_i = s_es[1][i];                      // (uniform)
#pragma unroll
        for (int k = 0; k < 4; k++) { seg[k] += (es_i <= lane + 64 * k); hd[k] |= (es_i == lane + 64 * k); }
    }
    if (lane == 0) hd[0] = 1;
    const int64_t sbase = tile * (SMAX + 1);
    if (lane <= ns) {                                   // segments without a candidate (two specials at one step, a special at step 0)
        int32_t lo = lane == 0 ? 0 : s_es[1][lane - 1], hi = lane == ns ? tl + 1 : s_es[1][lane];
        if (hi <= lo) { Best<TC, HYP> z; best_clear(z); sub[sbase + lane] = z; }
    }
    Best<TC, HYP> bcarry; best_clear(bcarry);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int32_t e = lane + 64 * k;
        Best<TC, HYP> x = cand[k];
        int f = hd[k];
        wave_segmin<TC, HYP>(x, f, lane);
        if (!f) x = better(bcarry, x);
        {
            Best<TC, HYP> nb; best_clear(nb); nb.v = shfl64(x.v, 63); nb.p = __shfl(x.p, 63); nb.nn = __shfl(x.nn, 63);
            if (HYP) best_set_nl(nb, __shfl(best_nl(x), 63));
            bcarry = nb;
        }
        int nh = __shfl_down(hd[k], 1), nh0 = __shfl(hd[k + 1], 0);
        if (lane == 63) nh = nh0;
        if (e <= tl && (e == tl || nh)) sub[sbase + seg[k]] = x;       // the last candidate of a segment holds its winner
    }
    if (lane < ns) spv[sbase + lane] = s_v[1][lane];
    if (lane == 0) spec[tile] = (uint8_t)ns;
}

// ------------------------------------------------------------------ gap passes
// The arg-min staircase of a rectangle jumps: between two neighbouring rows rL < rR of a round the winners B = opt[rL] and
// a = opt[rR] can lie thousands of columns apart, and the divide and conquer re-scans that gap [a, B] once per level below.
// In the rounds tau <= gap_tau a task with such a gap instead finishes EVERY row r' of (rL, hi), hi = min(rR, rectangle end,
// n + 1), in one pass over the gap: each of them has its (rightmost) winner inside [a, B].
//   nets(p, r') = nets(B, rL) + #{q in cols [rL, r') : prev[q] < B} + #{q in cols [p, B) : next[q] >= r'}
// and the last term is the same for all these rows except for the entries with rL < next[q] < hi - 1 ("special": counted by
// some rows only).  k_lpass_own<GAP> streams the gap once with the common threshold and flags the tiles holding specials; here
// one wave per (task, 64 rows) gives every lane a row and walks the task's tiles in candidate order: a plain tile contributes
// its winner (the same for all rows: their values differ by a row-only term plus cum(r') = the specials passed so far that
// the row counts); a flagged tile is re-walked entry by entry with every lane counting for its own row.  Ties: strict <
// in walking order = the larger p wins, as everywhere.  The rows are marked final (fin): later rounds skip them.
template <bool GE>
__device__ __forceinline__ int32_t gap_right_counts(const int32_t *__restrict__ cpos, const int32_t *__restrict__ arr, int32_t rL, int ck, int32_t thr,
                                                    int32_t n, int lane)
{
    // #{entries of columns [rL, r') passing the test}, r' = rL + 1 + 64 ck + lane.  Coalesced: the columns of the earlier row
    // chunks are summed by the whole wave; the wave's own 64 columns are one contiguous run of entries, flagged 64 at a time,
    // and every lane counts the flags in front of the end of its column.
    int32_t cfirst = rL + 64 * ck;
    if (cfirst > n) cfirst = n;
    int32_t c = 0;
    for (int32_t q = cpos[rL] + lane, q1 = cpos[cfirst]; q < q1; q += 64) { int32_t v = arr[q]; c += GE ? (v >= thr) : (v < thr); }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    int32_t jn = cfirst + lane + 1, cl = cfirst + 64;
    if (jn > n) jn = n;
    if (cl > n) cl = n;
    const int32_t myend = cpos[jn], e0 = cpos[cfirst], e1 = cpos[cl];
    int32_t mine = 0;
    for (int32_t base = e0; base < e1; base += 256) {       // four loads in flight
        bool fl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int32_t q = base + 64 * j + lane;
            fl[j] = false;
            if (q < e1) { int32_t v = arr[q]; fl[j] = GE ? (v >= thr) : (v < thr); }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned long long m = __ballot(fl[j]);
            int32_t w = myend - (base + 64 * j);
            if (w > 0) mine += __popcll(w >= 64 ? m : (m & ((1ull << w) - 1ull)));
        }
    }
    return c + mine;
}

template <typename TC, bool HYP>
struct GapCtx {
    const Best<TC, HYP> *part, *sub; const int32_t *spv; const uint8_t *spec;
    const int64_t *tilePS, *tilePS2;
    const int32_t *pos, *next, *fpos, *flast;
    const TC *W;
};

// wave-uniform lane index: one v_readlane instead of a ds_bpermute
__device__ __forceinline__ int32_t rdl(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int64_t rdl64(int64_t v, int src)
{
    return ((int64_t)__builtin_amdgcn_readlane((int)(v >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(v & 0xffffffffll), src);
}
__device__ __forceinline__ double rdl64(double v, int src) { return __longlong_as_double((long long)rdl64((int64_t)__double_as_longlong(v), src)); }

// the tiles [ka, kz) of a task (first tile k0, head B, L steps, row r) walked for NR x 64 rows of the wave (lane l holds the rows
// rr[0 .. NR-1], one per 64-row chunk: the tile records are fetched once for all of them): the winners so far (bv, bp, bl, bl2)
// and the specials the rows count so far (cum, cum2) are carried in and out
// SLOW = false: the entry-by-entry walk of a tile with more than SMAX specials is compiled out (its column-pointer and cost
// registers cost the kernels two waves per SIMD, and they wait on memory three quarters of their time): such an item returns false
// and is redone by the SLOW variant of its kernel.
template <typename TC, bool HYP, int NR, bool SLOW>
__device__ __forceinline__ bool gap_walk(const GapCtx<TC, HYP> &C, const DevModel<TC> &M, TC alpha, int64_t ka, int64_t kz, int64_t k0,
                                         int32_t B, int32_t L, int32_t r, int32_t posr, const int32_t (&rr)[NR], const int32_t (&rcmp)[NR],
                                         const bool (&valid)[NR], int lane,
                                         TC (&bv)[NR], int32_t (&bp)[NR], int32_t (&bl)[NR], int32_t (&bl2)[NR], int32_t (&cum)[NR], int32_t (&cum2)[NR])
{
    const int64_t ps0 = C.tilePS[k0], ps20 = HYP ? C.tilePS2[k0] : 0;
    for (int64_t kb = ka; kb < kz; kb += 64) {
    const int64_t km = kb + lane < kz ? kb + lane : kz - 1;
    const int32_t m_base = (int32_t)(C.tilePS[km] - ps0), m_base2 = HYP ? (int32_t)(C.tilePS2[km] - ps20) : 0;
    const int m_ns = C.spec[km];
    Best<TC, HYP> m_part; best_clear(m_part);
    if (m_ns == 0) m_part = C.part[km];
    const int nb = (int)(kz - kb < 64 ? kz - kb : 64);
    // the segment records of the NEXT tile are fetched while the current tile is walked (the walk is a chain of dependent loads)
    Best<TC, HYP> nx_c; best_clear(nx_c);
    int32_t nx_sv = 0;
    {
        const int ns0 = rdl(m_ns, 0);
        if (ns0 > 0 && ns0 <= SMAX) { if (lane <= ns0) nx_c = C.sub[kb * (SMAX + 1) + lane]; if (lane < ns0) nx_sv = C.spv[kb * (SMAX + 1) + lane]; }
    }
    for (int i = 0; i < nb; i++) {
        const int64_t k = kb + i;
        const int32_t base = rdl(m_base, i), base2 = HYP ? rdl(m_base2, i) : 0;
        const int ns = rdl(m_ns, i);
        const Best<TC, HYP> cur_c = nx_c;
        const int32_t cur_sv = nx_sv;
        if (i + 1 < nb) {
            const int ns1 = rdl(m_ns, i + 1);
            if (ns1 > 0 && ns1 <= SMAX) { if (lane <= ns1) nx_c = C.sub[(k + 1) * (SMAX + 1) + lane]; if (lane < ns1) nx_sv = C.spv[(k + 1) * (SMAX + 1) + lane]; }
        }
        if (ns <= SMAX) {
            // plain tile: one winner; tile with ns specials: ns + 1 segment winners, a special between two segments
            Best<TC, HYP> c = cur_c;
            int32_t sv = cur_sv;
            if (ns == 0) {
                best_clear(c);
                c.v = rdl64(m_part.v, i); c.p = rdl(m_part.p, i); c.nn = rdl(m_part.nn, i);
                if (HYP) best_set_nl(c, rdl(best_nl(m_part), i));
            }
            for (int sg = 0; sg <= ns; sg++) {
                Best<TC, HYP> d = c;
                if (ns) { d.v = rdl64(c.v, sg); d.p = rdl(c.p, sg); d.nn = rdl(c.nn, sg); if (HYP) best_set_nl(d, rdl(best_nl(c), sg)); }
                if (d.p >= 0) {
#pragma unroll
                    for (int u = 0; u < NR; u++) {
                        int32_t lc = d.nn + base + cum[u], lc2 = HYP ? best_nl(d) + base2 + cum2[u] : 0;
                        TC v = cadd(d.v, dm_apply(M, (TC)0, (int64_t)0, (int64_t)0, (int64_t)(base + cum[u]), (int64_t)(base2 + cum2[u])));
                        if (bp[u] < 0 || v < bv[u]) { bv[u] = v; bp[u] = d.p; bl[u] = lc; bl2[u] = lc2; }
                    }
                }
                if (sg < ns) {                          // the candidates from here on have this special on their right
                    const int32_t s1 = rdl(sv, sg), val = s1 & 0x7fffffff;
                    const bool first_list = s1 >= 0;        // (two unconditional adds: an if / else here becomes ONE add through a selected address, i.e. scratch)
#pragma unroll
                    for (int u = 0; u < NR; u++) {
                        cum[u] += (first_list && val >= rcmp[u]);
                        cum2[u] += (!first_list && valid[u] && val < rr[u]);
                    }
                }
            }
            continue;
        }
        // too many specials: every lane counts for its own rows, entry by entry
        if (!SLOW) return false;
        if (SLOW) {
        const int head = k == k0 ? 1 : 0;
        const int32_t pf = B - (int32_t)(k - k0) * LT;
        int32_t tlk = L - (int32_t)(k - k0) * LT; tlk = (tlk < LT ? tlk : LT) - 1;
        // C.pos[pf + 1 - j] and C.W[pf - j] of the tile's columns are fetched lane-strided; the entries come through 64-entry
        // windows (one coalesced load each) and are handed to all lanes one at a time
#define CP_SEL5(a_, i_) ((i_) < 64 ? a_[0] : (i_) < 128 ? a_[1] : (i_) < 192 ? a_[2] : (i_) < 256 ? a_[3] : a_[4])
        int32_t pk[5], pk2[5] = {0, 0, 0, 0, 0}; TC wk[4];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            int32_t e = lane + 64 * j;
            pk[j] = e <= tlk + 1 ? C.pos[pf + 1 - e] : 0;
            if (HYP) pk2[j] = e <= tlk + 1 ? C.fpos[pf + 1 - e] : 0;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { int32_t e = lane + 64 * j; wk[j] = e <= tlk ? C.W[pf - e] : (TC)0; }
        int32_t run[NR], run2[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) { run[u] = 0; run2[u] = 0; }
        int32_t wb = INT32_MAX, reg = 0, wb2 = INT32_MAX, reg2 = 0;
        for (int32_t e = 0; e <= tlk; e++) {
            const int32_t p = pf - e, e1 = e + 1;
            const int32_t en = __shfl(CP_SEL5(pk, e), e & 63), st = __shfl(CP_SEL5(pk, e1), e1 & 63);      // C.pos[p + 1], C.pos[p]
            if (!(head && e == 0)) {                    // step over column p (the head element steps over nothing)
                for (int32_t q = en - 1; q >= st; q--) {
                    if (q < wb || q - wb > 63) { wb = q - 63; reg = wb + lane >= 0 ? C.next[wb + lane] : 0; }
                    const int32_t x = __shfl(reg, q - wb);           // (every lane takes part: no short-circuit around a shuffle)
#pragma unroll
                    for (int u = 0; u < NR; u++) run[u] += (x >= rcmp[u]);
                }
                if (HYP) {
                    const int32_t en2 = __shfl(CP_SEL5(pk2, e), e & 63), st2 = __shfl(CP_SEL5(pk2, e1), e1 & 63);
                    for (int32_t q = en2 - 1; q >= st2; q--) {
                        if (q < wb2 || q - wb2 > 63) { wb2 = q - 63; reg2 = wb2 + lane >= 0 ? C.flast[wb2 + lane] : 0; }
                        const int32_t x = __shfl(reg2, q - wb2);
#pragma unroll
                        for (int u = 0; u < NR; u++) run2[u] += (valid[u] && x < rr[u]);
                    }
                }
            }
            const TC wp = shfl64(e < 64 ? wk[0] : e < 128 ? wk[1] : e < 192 ? wk[2] : wk[3], e & 63);
#pragma unroll
            for (int u = 0; u < NR; u++) {
                int32_t lc = base + cum[u] + run[u], lc2 = base2 + cum2[u] + run2[u];
                TC v = cadd(wp, dm_apply(M, alpha, (int64_t)(r - p), (int64_t)(posr - st), (int64_t)lc, (int64_t)lc2));      // (the task's row: the rows' values differ by a row-only term)
                if (bp[u] < 0 || v < bv[u]) { bv[u] = v; bp[u] = p; bl[u] = lc; bl2[u] = lc2; }
            }
        }
#undef CP_SEL5
        // from here on the rows also count the specials of this tile they passed
#pragma unroll
        for (int u = 0; u < NR; u++) {
            cum[u] += run[u] - (int32_t)(C.tilePS[k + 1] - C.tilePS[k]);
            if (HYP) cum2[u] += run2[u] - (int32_t)(C.tilePS2[k + 1] - C.tilePS2[k]);
        }
        }
    }
    }
    return true;
}

// rows of the wave, right parts and anchors of a gap task: shared by the kernels below
struct GapRows { int32_t rL, hi, rr, rcmp; bool valid, none; };
__device__ __forceinline__ GapRows gap_rows(int tau, int ck, int32_t r, int b, int64_t n, int lane)
{
    GapRows g;
    g.rL = r - (1 << tau);
    int64_t hi64 = (int64_t)r + ((int64_t)1 << tau), re = ((((int64_t)r >> b) + 1) << b);
    if (hi64 > re) hi64 = re;
    if (hi64 > n + 1) hi64 = n + 1;
    g.hi = (int32_t)hi64;
    g.rr = g.rL + 1 + 64 * ck + lane;                       // this lane's row
    g.none = g.rL + 1 + 64 * ck >= g.hi;                    // (wave-uniform: the chunk lies beyond the gap's rows)
    g.valid = g.rr < g.hi;
    g.rcmp = g.valid ? g.rr : INT32_MAX;                    // an invalid lane counts nothing
    return g;
}

constexpr int GAPSEG = 16;      // tiles one wave walks; longer tasks are walked by several waves (k_gap_seg) and merged (k_gap_merge)

template <typename TC, bool HYP>
struct GapSegRec { TC v; int32_t p, l, l2, cum, cum2, _pad; };

// NR: 64-row chunks per wave (a lane holds one row of each): the task's records -- descriptor, tile winners, segment lists -- are
// fetched once for NR x 64 rows.  The kernel waits on memory three quarters of its time at four waves per SIMD (profiles/
// r03_pmc_gap_finish.txt), i.e. it is bound by the number of (task, rows) items times the dependent loads of one item.
template <typename TC, bool HYP, int NR, bool SLOW>
__global__ void __launch_bounds__(256, 4) k_gap_finish(int tau, int nchunk, RoundCounts *__restrict__ rc, int64_t n,
                                                    const int64_t *__restrict__ toffs, GapCtx<TC, HYP> C,
                                                    const int4 *__restrict__ tdesc, const uint8_t *__restrict__ tb, const int32_t *__restrict__ rlen,
                                                    const int32_t *__restrict__ prev, const int32_t *__restrict__ lpos, const int32_t *__restrict__ lfirst,
                                                    DevModel<TC> M, TC alpha,
                                                    int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int32_t *__restrict__ nlopt, uint8_t *__restrict__ fin,
                                                    int2 *__restrict__ glist, int2 *__restrict__ gslot, int fin_stamp, int32_t *__restrict__ slow)
{
    // SLOW: the items the fast variant listed in `slow` (rc->n_gslow of them)
    const int lane = threadIdx.x & 63;
    const int nitem = (nchunk + NR - 1) / NR;
    const int64_t nwork = SLOW ? (int64_t)rc->n_gslow : (int64_t)rc->nown * nitem, n1 = n + 1;
    for (int64_t wi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); wi < nwork; wi += (int64_t)gridDim.x * 4) {
        const int64_t w = SLOW ? (int64_t)slow[wi] : wi;
        const int64_t t = w / nitem;
        const int ck0 = (int)(w - t * nitem) * NR;
        const int64_t k0 = toffs[t], k1 = toffs[t + 1];
        if (k1 - k0 > GAPSEG) {                             // (wave-uniform) a long task: listed for k_gap_seg / k_gap_merge
            if (!SLOW && ck0 == 0 && lane == 0) {
                int nseg = (int)((k1 - k0 + GAPSEG - 1) / GAPSEG);
                int idx = atomicAdd(&rc->n_glong, 1), slot0 = atomicAdd(&rc->n_gslots, nseg);
                glist[idx] = make_int2((int)t, slot0);
                for (int j = 0; j < nseg; j++) gslot[slot0 + j] = make_int2((int)t, j);       // slot -> (task, segment) for k_gap_seg
            }
            continue;
        }
        const int4 td = tdesc[t];                           // {B, anchor + right part of the task's own row, row r, pos[r]}
        const int b = tb[t];
        const int32_t B = td.x, r = td.z, posr = td.w, L = rlen[t];
        GapRows g[NR];
        int32_t rr[NR], rcmp[NR]; bool valid[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) { g[u] = gap_rows(tau, ck0 + u, r, b, n, lane); rr[u] = g[u].rr; rcmp[u] = g[u].rcmp; valid[u] = g[u].valid; }
        if (g[0].none) continue;
        // right parts of the rows and the anchor (counts at (B, rL))
        int32_t Rr[NR], Rr2[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) {
            Rr[u] = 0; Rr2[u] = 0;
            if (u == 0 || !g[u].none) {                     // (wave-uniform)
                Rr[u] = gap_right_counts<false>(C.pos, prev, g[0].rL, ck0 + u, B, (int32_t)n, lane);
                if (HYP) Rr2[u] = gap_right_counts<true>(lpos, lfirst, g[0].rL, ck0 + u, B, (int32_t)n, lane);
            }
        }
        const int64_t rwL = (int64_t)b * n1 + PR((int64_t)g[0].rL);
        const int32_t anchor = nnopt[rwL], anchor2 = HYP ? nlopt[rwL] : 0;
        TC bv[NR]; int32_t bp[NR], bl[NR], bl2[NR], cum[NR], cum2[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) { bv[u] = (TC)0; bp[u] = -1; bl[u] = 0; bl2[u] = 0; cum[u] = 0; cum2[u] = 0; }
        if (!gap_walk<TC, HYP, NR, SLOW>(C, M, alpha, k0, k1, k0, B, L, r, posr, rr, rcmp, valid, lane, bv, bp, bl, bl2, cum, cum2)) {
            if (lane == 0) slow[atomicAdd(&rc->n_gslow, 1)] = (int32_t)w;
            continue;
        }
#pragma unroll
        for (int u = 0; u < NR; u++) {
            if (valid[u]) {
                int64_t rw = (int64_t)b * n1 + PR((int64_t)rr[u]);
                opt[rw] = bp[u]; nnopt[rw] = anchor + Rr[u] + bl[u];
                if (HYP) nlopt[rw] = anchor2 + Rr2[u] + bl2[u];
                fin[rw] = (uint8_t)fin_stamp;
            }
        }
    }
}

// long gap tasks, phase 1: one wave per (segment slot = GAPSEG tiles of a listed task, 64 rows) walks its tiles as if nothing lay
// in front of them
template <typename TC, bool HYP, bool SLOW>
__global__ void __launch_bounds__(256) k_gap_seg(int tau, int nchunk, RoundCounts *__restrict__ rc, int64_t n, const int64_t *__restrict__ toffs,
                                                 GapCtx<TC, HYP> C, const int4 *__restrict__ tdesc, const uint8_t *__restrict__ tb,
                                                 const int32_t *__restrict__ rlen, DevModel<TC> M, TC alpha, const int2 *__restrict__ gslot,
                                                 GapSegRec<TC, HYP> *__restrict__ gseg, int32_t *__restrict__ slow)
{
    const int lane = threadIdx.x & 63;
    const int64_t nwork = SLOW ? (int64_t)rc->n_sslow : (int64_t)rc->n_gslots * nchunk;
    for (int64_t wi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); wi < nwork; wi += (int64_t)gridDim.x * 4) {
        const int64_t w = SLOW ? (int64_t)slow[wi] : wi;
        const int64_t slot = w / nchunk;
        const int ck = (int)(w - slot * nchunk);
        const int2 se = gslot[slot];
        const int64_t t = se.x, k0 = toffs[t], k1 = toffs[t + 1];
        const int sg = se.y;
        const int4 td = tdesc[t];
        const GapRows g = gap_rows(tau, ck, td.z, tb[t], n, lane);
        if (g.none) continue;
        TC bv[1] = {(TC)0}; int32_t bp[1] = {-1}, bl[1] = {0}, bl2[1] = {0}, cum[1] = {0}, cum2[1] = {0};
        const int32_t rr1[1] = {g.rr}, rcmp1[1] = {g.rcmp}; const bool valid1[1] = {g.valid};
        const int64_t ka = k0 + (int64_t)sg * GAPSEG, kz = ka + GAPSEG < k1 ? ka + GAPSEG : k1;
        if (!gap_walk<TC, HYP, 1, SLOW>(C, M, alpha, ka, kz, k0, td.x, rlen[t], td.z, td.w, rr1, rcmp1, valid1, lane, bv, bp, bl, bl2, cum, cum2)) {
            if (lane == 0) slow[atomicAdd(&rc->n_sslow, 1)] = (int32_t)w;
            continue;
        }
        GapSegRec<TC, HYP> rec; rec.v = bv[0]; rec.p = bp[0]; rec.l = bl[0]; rec.l2 = bl2[0]; rec.cum = cum[0]; rec.cum2 = cum2[0]; rec._pad = 0;
        gseg[(slot * nchunk + ck) * 64 + lane] = rec;
    }
}

// phase 2: one wave per (listed task, 64 rows) merges the segment winners in walking order, adding the specials of the
// segments in front
template <typename TC, bool HYP>
__global__ void __launch_bounds__(256) k_gap_merge(int tau, int nchunk, const RoundCounts *__restrict__ rc, int64_t n, const int64_t *__restrict__ toffs,
                                                   const int4 *__restrict__ tdesc, const uint8_t *__restrict__ tb,
                                                   const int32_t *__restrict__ pos, const int32_t *__restrict__ prev, const int32_t *__restrict__ lpos,
                                                   const int32_t *__restrict__ lfirst, DevModel<TC> M, const int2 *__restrict__ glist,
                                                   const GapSegRec<TC, HYP> *__restrict__ gseg,
                                                   int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int32_t *__restrict__ nlopt, uint8_t *__restrict__ fin, int fin_stamp)
{
    const int lane = threadIdx.x & 63;
    const int64_t nwork = (int64_t)rc->n_glong * nchunk, n1 = n + 1;
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nwork; w += (int64_t)gridDim.x * 4) {
        const int i = (int)(w / nchunk), ck = (int)(w - (int64_t)i * nchunk);
        const int2 le = glist[i];
        const int64_t t = le.x, k0 = toffs[t], k1 = toffs[t + 1];
        const int nseg = (int)((k1 - k0 + GAPSEG - 1) / GAPSEG);
        const int4 td = tdesc[t];
        const int b = tb[t];
        const GapRows g = gap_rows(tau, ck, td.z, b, n, lane);
        if (g.none) continue;
        const int32_t Rr = gap_right_counts<false>(pos, prev, g.rL, ck, td.x, (int32_t)n, lane);
        const int32_t Rr2 = HYP ? gap_right_counts<true>(lpos, lfirst, g.rL, ck, td.x, (int32_t)n, lane) : 0;
        const int64_t rwL = (int64_t)b * n1 + PR((int64_t)g.rL);
        const int32_t anchor = nnopt[rwL], anchor2 = HYP ? nlopt[rwL] : 0;
        TC bv = (TC)0; int32_t bp = -1, bl = 0, bl2 = 0, cum = 0, cum2 = 0;
        // (the task over the top rectangle's whole block has thousands of segments: eight records per lane in flight -- the running
        //  counts make the merge sequential, the loads need not be)
        for (int s0 = 0; s0 < nseg; s0 += 8) {
            GapSegRec<TC, HYP> rr[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (s0 + u < nseg) rr[u] = gseg[((int64_t)(le.y + s0 + u) * nchunk + ck) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (s0 + u >= nseg) break;
                const GapSegRec<TC, HYP> rec = rr[u];
                if (rec.p >= 0) {
                    TC v = cadd(rec.v, dm_apply(M, (TC)0, (int64_t)0, (int64_t)0, (int64_t)cum, (int64_t)cum2));
                    if (bp < 0 || v < bv) { bv = v; bp = rec.p; bl = rec.l + cum; bl2 = rec.l2 + cum2; }
                }
                cum += rec.cum; cum2 += rec.cum2;
            }
        }
        if (g.valid) {
            int64_t rw = (int64_t)b * n1 + PR((int64_t)g.rr);
            opt[rw] = bp; nnopt[rw] = anchor + Rr + bl;
            if (HYP) nlopt[rw] = anchor2 + Rr2 + bl2;
            fin[rw] = (uint8_t)fin_stamp;
        }
    }
}

// Merging the tiles of a task: one LANE per task looks at the tile count -- up to FIX_SERIAL: merged by the lane; more: appended
// to the list k_fix_own walks (one wave or one block per task).  (Tried: k_lpass_own writing the winner of a single-tile task
// itself -- the dependent loads at the end of every wave cost it more than the merge saves.)
constexpr int FIX_SERIAL = 16;

template <typename TC, bool HYP>
__device__ __forceinline__ void fix_merge(Best<TC, HYP> &acc, Best<TC, HYP> c, int64_t base, int64_t base2, const DevModel<TC> &M)
{
    if (c.p >= 0) {
        c.v = cadd(c.v, dm_apply(M, (TC)0, (int64_t)0, (int64_t)0, base, base2));
        c.nn = (int32_t)(c.nn + base);
        if (HYP) best_set_nl(c, (int32_t)(best_nl(c) + base2));
    }
    bool take = (acc.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < acc.v || (c.v == acc.v && c.p > acc.p)));
    if (take) acc = c;
}

template <typename TC, bool HYP>
__global__ void __launch_bounds__(256) k_fix_own_lane(RoundCounts *__restrict__ rc, const int64_t *__restrict__ toffs, const Best<TC, HYP> *__restrict__ part,
                                                      const int4 *__restrict__ tdesc, const uint8_t *__restrict__ tb, const int32_t *__restrict__ tS0l,
                                                      const int64_t *__restrict__ tilePS, const int64_t *__restrict__ tilePS2, DevModel<TC> M,
                                                      int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int32_t *__restrict__ nlopt, int64_t n1,
                                                      int32_t *__restrict__ wide_list)
{
    const int64_t ntask = rc->nown;
    int lane = threadIdx.x & 63;
    for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x; t0 < ntask; t0 += (int64_t)gridDim.x * blockDim.x) {      // wave-uniform
        int64_t t = t0 + threadIdx.x;
        int64_t k0 = 0, k1 = 0;
        if (t < ntask) { k0 = toffs[t]; k1 = toffs[t + 1]; }
        int64_t nt = k1 - k0;
        bool wide = nt > FIX_SERIAL;
        unsigned long long mw = __ballot(wide);
        if (mw) {                                                   // wave-aggregated append
            int32_t base = 0;
            if (lane == 0) base = atomicAdd(&rc->n_wide, (int32_t)__popcll(mw));
            base = __shfl(base, 0);
            if (wide) wide_list[base + __popcll(mw & ((1ull << lane) - 1ull))] = (int32_t)t;
        }
        if (nt < 1 || wide) continue;
        int4 td = tdesc[t];
        int64_t S0 = td.y, S0l = HYP ? (int64_t)tS0l[t] : 0;
        Best<TC, HYP> acc; best_clear(acc);
        for (int64_t k = k0; k < k1; k++)                          // increasing k = decreasing p: larger p wins ties
            fix_merge<TC, HYP>(acc, part[k], S0 + (tilePS[k] - tilePS[k0]), HYP ? S0l + (tilePS2[k] - tilePS2[k0]) : 0, M);
        int64_t rw = (int64_t)tb[t] * n1 + PR((int64_t)td.z);
        opt[rw] = acc.p; nnopt[rw] = acc.nn;
        if (HYP) nlopt[rw] = best_nl(acc);
    }
}

template <typename TC, bool HYP, int WPT>
__global__ void __launch_bounds__(256) k_fix_own(const RoundCounts *__restrict__ rc, const int64_t *__restrict__ toffs, const Best<TC, HYP> *__restrict__ part,
                                                 const int4 *__restrict__ tdesc, const uint8_t *__restrict__ tb, const int32_t *__restrict__ tS0l,
                                                 const int64_t *__restrict__ tilePS, const int64_t *__restrict__ tilePS2, DevModel<TC> M,
                                                 int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int32_t *__restrict__ nlopt, int64_t n1,
                                                 const int32_t *__restrict__ wide_list)
{
    // WPT = 1: one wave per task (four tasks per block); WPT = 4: the whole block works on one task (rounds with a few
    // tasks of thousands of tiles each)
    __shared__ Best<TC, HYP> s_part[4];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t nlist = rc->n_wide;
    // (the loop is block-uniform when WPT == 4: the barrier below is reached by the whole block)
    for (int64_t i = WPT == 1 ? (int64_t)blockIdx.x * 4 + wv : (int64_t)blockIdx.x; i < nlist; i += WPT == 1 ? (int64_t)gridDim.x * 4 : (int64_t)gridDim.x) {
    int64_t t = wide_list[i];
    int64_t k0 = toffs[t], k1 = toffs[t + 1];
    int4 td = tdesc[t];
    int64_t S0 = td.y, S0l = HYP ? (int64_t)tS0l[t] : 0;
    Best<TC, HYP> acc; best_clear(acc);
    // (In every round the last row of the top rectangle owns a task over its whole block -- tens of thousands of tiles: eight tiles
    //  per lane and step are in flight; one at a time, that task alone took 30 us of every round.)
    const int64_t ps0 = tilePS[k0], ps20 = HYP ? tilePS2[k0] : 0;
    for (int64_t kb = k0 + (WPT == 1 ? lane : (int64_t)threadIdx.x); kb < k1; kb += 8 * 64 * WPT) {
        Best<TC, HYP> cc[8]; int64_t ps[8], ps2[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t k = kb + (int64_t)u * 64 * WPT;
            best_clear(cc[u]); ps[u] = 0; ps2[u] = 0;
            if (k < k1) { cc[u] = part[k]; ps[u] = tilePS[k]; if (HYP) ps2[u] = tilePS2[k]; }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            Best<TC, HYP> c = cc[u];
            if (c.p >= 0) {
                int64_t base = S0 + (ps[u] - ps0);
                int64_t base2 = HYP ? S0l + (ps2[u] - ps20) : 0;
                c.v = cadd(c.v, dm_apply(M, (TC)0, (int64_t)0, (int64_t)0, base, base2));
                c.nn = (int32_t)(c.nn + base);
                if (HYP) best_set_nl(c, (int32_t)(best_nl(c) + base2));
            }
            bool take = (acc.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < acc.v || (c.v == acc.v && c.p > acc.p)));
            if (take) acc = c;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        int src = (lane + o) & 63;
        Best<TC, HYP> c; best_clear(c); c.v = shfl64(acc.v, src); c.p = __shfl(acc.p, src); c.nn = __shfl(acc.nn, src);
        if (HYP) best_set_nl(c, __shfl(best_nl(acc), src));
        bool take = (acc.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < acc.v || (c.v == acc.v && c.p > acc.p)));
        if (lane + o < 64 && take) acc = c;
    }
    if (WPT > 1) {
        __syncthreads();                                 // (the previous task's partials have been read)
        if (lane == 0) s_part[wv] = acc;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int w = 1; w < 4; w++) {
                Best<TC, HYP> c = s_part[w];
                bool take = (acc.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < acc.v || (c.v == acc.v && c.p > acc.p)));
                if (take) acc = c;
            }
    }
    if (threadIdx.x == (WPT == 1 ? (unsigned)(wv * 64) : 0u)) {
        int b = tb[t];
        int64_t rw = (int64_t)b * n1 + PR((int64_t)td.z);
        opt[rw] = acc.p; nnopt[rw] = acc.nn;
        if (HYP) nlopt[rw] = best_nl(acc);
    }
    }
}

// ------------------------------------------------------------------ left part: stream, scan, evaluate, arg-min
// HYP: hyperedge-cut costs carry a second count (self nets): rows bucketed by their FIRST column (fpos / flast),
// a column p joining on the left adds the rows with first == p and last < r.

template <typename TC, bool HYP>
__global__ void __launch_bounds__(256, 6) k_lpass(RoundDesc R, const RoundCounts *__restrict__ rc, const int64_t *__restrict__ a_offs,
                                                  const int4 *__restrict__ a_tdesc, const int32_t *__restrict__ a_tS0l,
                                                  const uint8_t *__restrict__ a_tb, const int32_t *__restrict__ a_pos,
                                                  const int32_t *__restrict__ a_next, const int32_t *__restrict__ a_fpos,
                                                  const int32_t *__restrict__ a_flast, int32_t *__restrict__ a_opt,
                                                  int32_t *__restrict__ a_nnopt, int32_t *__restrict__ a_nlopt,
                                                  int32_t *__restrict__ a_loc, int32_t *__restrict__ a_loc2,
                                                  int32_t *__restrict__ a_tileS, int32_t *__restrict__ a_tileS2,
                                                  int64_t *__restrict__ a_taskR, const int64_t *__restrict__ a_tile_t0,
                                                  const int4 *__restrict__ a_tile_rec,
                                                  const TC *__restrict__ W, DevModel<TC> M, TC alpha, Best<TC, HYP> *__restrict__ partR,
                                                  Best<TC, HYP> *__restrict__ partL)
{
    __shared__ int32_t s_off_all[4][LT + 2];
    __shared__ unsigned long long s_hd_all[4][LT / 64];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (no grid-stride loop: the host launches one wave per tile of the buffers' capacity, see k_lpass_own)
    const int64_t T = rc->T, ntask = rc->nlong;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    int64_t tile_start = tile * LT;
    bool active = tile_start < T;
    int32_t *s_off = s_off_all[wave];
    unsigned long long *s_hd = s_hd_all[wave];
    // interior tile (k_tile_t0 classified it): four out of five tiles -- the D&C re-scans the wide gaps of the arg-min
    // staircase at every level.  Wave-uniform fast path: nothing to look up, one threshold, nothing to evaluate here
    // (k_open finishes these tiles from `loc`).
    int4 rec = active ? a_tile_rec[tile] : make_int4(0, 0, 0, 0);
    bool interior = rec.w != 0;
    int64_t t0 = 0; int cnt = 0;
    if (!interior) {
        t0 = active ? a_tile_t0[tile] : 0;
        load_tile_tasks(a_offs, ntask, t0, tile_start, active, s_off, s_hd, lane, cnt);
    }
    __syncthreads();
    if (!active) return;
    if (interior) {
        // The admitted costs are affine in the counts with constant coefficients, and all steps of the tile share the count
        // made before the tile (the same task): the tile's arg-min does not depend on it.  Evaluate with the counts since
        // the tile start; k_fix adds the task's base when it merges the tiles.
        int32_t tl = (T - tile_start < LT) ? (int32_t)(T - tile_start) - 1 : LT - 1;
        int32_t acc[4], acc2[4] = {0, 0, 0, 0}, sk[4], sk2[4];
        interior_stream<true>(a_next, a_pos, rec.x, tl, rec.y, lane, acc, sk);
        if (HYP) interior_stream<false>(a_flast, a_fpos, rec.x, tl, rec.y, lane, acc2, sk2);
        int sel = tl >> 6;
        if (lane == (tl & 63)) {
            a_tileS[tile] = sel == 0 ? acc[0] : sel == 1 ? acc[1] : sel == 2 ? acc[2] : acc[3];
            if (HYP) a_tileS2[tile] = sel == 0 ? acc2[0] : sel == 1 ? acc2[1] : sel == 2 ? acc2[2] : acc2[3];
        }
        Best<TC, HYP> best; best_clear(best);
#pragma unroll
        for (int k = 0; k < 4; k++) {                       // increasing e = decreasing p: an earlier candidate wins ties
            int32_t e = lane + 64 * k;
            if (e <= tl) {
                int32_t p = rec.x - e;
                TC fv = dm_apply(M, alpha, (int64_t)(rec.y - p), (int64_t)(rec.z - sk[k]), (int64_t)acc[k], (int64_t)acc2[k]);
                Best<TC, HYP> c; best_clear(c); c.v = cadd(W[p], fv); c.p = p; c.nn = acc[k]; best_set_nl(c, acc2[k]);
                best = better(best, c);
            }
        }
        for (int o = 32; o > 0; o >>= 1) {                  // wave arg-min; ties -> larger p
            int src = (lane + o) & 63;
            Best<TC, HYP> c; best_clear(c); c.v = shfl64(best.v, src); c.p = __shfl(best.p, src); c.nn = __shfl(best.nn, src);
            if (HYP) best_set_nl(c, __shfl(best_nl(best), src));
            bool take = (best.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < best.v || (c.v == best.v && c.p > best.p)));
            if (lane + o < 64 && take) best = c;
        }
        if (lane == 0) partL[tile] = best;
        return;
    }
    // local task index of a step = (#heads at or before it in the tile) - (1 if the tile starts with a head)
    int head_at0 = (s_off[0] == 0) ? 1 : 0;
    int hd_before = 0;                                  // heads in the groups already processed
    int64_t n1 = R.n + 1;
    int32_t tile_last = (T - tile_start < LT) ? (int32_t)(T - tile_start) - 1 : LT - 1;     // relative to the tile start
    // everything below is 32-bit: step indices are relative to the tile, nonzero positions are < 2^31

    constexpr int NG = LT / 64;
    int32_t carryS = 0, carryS2 = 0;                   // segmented-sum state across the groups of the tile
    Best<TC, HYP> bcarry; best_clear(bcarry);
    for (int g = 0; g < NG; g++) {
        int32_t e = g * 64 + lane;
        bool valid = e <= tile_last;
        int32_t t = 0, i = 0, r = 0, B = 0, p = -1, toff = 0, seg_last = -1, s = 0, en = 0, s2 = 0, en2 = 0, posr = 0;
        int32_t s0 = 0, s0l = 0; TC wp = (TC)0;
        unsigned long long hmask = s_hd[g];
        if (valid) {
            int li = hd_before + __popcll(hmask & ((2ull << lane) - 1)) - head_at0;
            t = (int32_t)t0 + li;
            toff = s_off[li];
            if (li + 1 < cnt) seg_last = s_off[li + 1] - 1;
            else { int64_t nx = a_offs[(int64_t)t + 1] - tile_start; seg_last = nx > 0x3fffffff ? 0x3fffffff : (int32_t)nx - 1; }
            i = e - toff;
            int4 td = a_tdesc[t];
            B = td.x; s0 = td.y; r = td.z; posr = td.w;
            if (HYP) s0l = a_tS0l[t];
            p = B - i;
            if (i == 0) { s = en = a_pos[B]; if (HYP) s2 = en2 = a_fpos[B]; }
            else { s = a_pos[p]; en = a_pos[p + 1]; if (HYP) { s2 = a_fpos[p]; en2 = a_fpos[p + 1]; } }
            wp = W[p];
            if (i == 0 && R.isA) p = -1;              // round A: element 0 (p = r) is not a candidate
        }
        hd_before += __popcll(hmask);
        int32_t thr = valid ? r : INT32_MAX;
        int32_t d = coop_count<true>(a_next, s, en, thr, valid, lane);                  // next[q] >= r
        int32_t d2 = 0;
        if (HYP) d2 = coop_count<false>(a_flast, s2, en2, thr, valid, lane);           // last < r
        // ---- segmented prefix of the step counts (restart at every task head)
        int f = valid ? (i == 0) : 1;
        int32_t x = d;
        wave_segsum(x, f, lane);
        if (!f) x += carryS;
        carryS = __shfl(x, 63);
        int32_t x2 = 0;
        if (HYP) {
            int f2 = valid ? (i == 0) : 1;
            x2 = d2;
            wave_segsum(x2, f2, lane);
            if (!f2) x2 += carryS2;
            carryS2 = __shfl(x2, 63);
        }
        // ---- evaluate (only where the task head lies in this tile: the count is final)
        bool head_in_tile = valid && toff >= 0;
        Best<TC, HYP> bx; best_clear(bx);
        if (head_in_tile && p >= 0) {
            int64_t nn = (int64_t)s0 + x, nl = HYP ? (int64_t)s0l + x2 : 0;
            TC fv = dm_apply(M, alpha, (int64_t)(r - p), (int64_t)(posr - s), nn, nl);     // s == pos[p] for every candidate
            bx.v = cadd(wp, fv); bx.p = p; bx.nn = (int32_t)nn; best_set_nl(bx, (int32_t)nl);
        } else if (valid && !head_in_tile) {
            a_loc[tile_start + e] = x;                  // counts since the tile start; finished by k_open
            if (HYP) a_loc2[tile_start + e] = x2;
        }
        int bf = valid ? (i == 0) : 1;
        wave_segmin<TC, HYP>(bx, bf, lane);
        if (!bf) bx = better(bcarry, bx);
        {
            Best<TC, HYP> nb; best_clear(nb); nb.v = shfl64(bx.v, 63); nb.p = __shfl(bx.p, 63); nb.nn = __shfl(bx.nn, 63);
            if (HYP) best_set_nl(nb, __shfl(best_nl(bx), 63));
            bcarry = nb;
        }
        if (head_in_tile) {
            if (e == seg_last) {
                int b = a_tb[t];
                a_opt[(int64_t)b * n1 + PR(r)] = bx.p;
                a_nnopt[(int64_t)b * n1 + PR(r)] = bx.nn;
                if (HYP) a_nlopt[(int64_t)b * n1 + PR(r)] = best_nl(bx);
            } else if (e == tile_last) {
                partR[tile] = bx; a_taskR[tile] = (int64_t)t;
            }
        }
    }
    if (lane == 0) { a_tileS[tile] = carryS; if (HYP) a_tileS2[tile] = carryS2; }   // counts since the last head of the tile
}

// ------------------------------------------------------------------ tasks that span tiles
// A task whose steps continue beyond its head tile is finished in one of two ways:
//  * short spans (the task ends in the next tile, at most SPAN_SHORT steps there): ONE LANE evaluates the remaining
//    steps from the counts k_lpass left in `loc`, merges them with the head tile's partial and writes the winner
//    (k_span_short, one lane per tile -- at low tau nearly every tile boundary cuts a short task);
//  * long spans: every covered tile is reduced by one wave (k_open), then one wave per head tile merges the
//    partials (k_fix).  k_span_short appends those tiles to two work lists; the wave kernels walk the lists.
constexpr int SPAN_SHORT = 32;

template <typename TC, bool HYP>
__global__ void __launch_bounds__(256) k_span_short(RoundDesc R, RoundCounts *__restrict__ rc, const int64_t *__restrict__ a_offs,
                                                    const int4 *__restrict__ a_tdesc, const int32_t *__restrict__ a_tS0l,
                                                    const uint8_t *__restrict__ a_tb, const int32_t *__restrict__ a_pos,
                                                    const int32_t *__restrict__ a_loc, const int32_t *__restrict__ a_loc2,
                                                    const int64_t *__restrict__ a_tile_t0, const int64_t *__restrict__ a_taskR,
                                                    const int32_t *__restrict__ a_tileS, const int32_t *__restrict__ a_tileS2,
                                                    const Best<TC, HYP> *__restrict__ partR, const TC *__restrict__ W, DevModel<TC> M, TC alpha,
                                                    int32_t *__restrict__ a_opt, int32_t *__restrict__ a_nnopt, int32_t *__restrict__ a_nlopt,
                                                    int32_t *__restrict__ open_list, int32_t *__restrict__ fix_list,
                                                    const int4 *__restrict__ a_tile_rec)
{
    int lane = threadIdx.x & 63;
    const int64_t ntile = rc->ntile;
    for (int64_t tbase = (int64_t)blockIdx.x * blockDim.x; tbase < ntile; tbase += (int64_t)gridDim.x * blockDim.x) {      // wave-uniform
    int64_t tile = tbase + threadIdx.x;
    bool in = tile < ntile;
    int64_t n1 = R.n + 1;
    // ---- role 1: this tile is the head tile of a task that continues beyond it
    bool fix_long = false;
    if (in) {
        int64_t t = a_taskR[tile];
        if (t >= 0) {
            int64_t end = a_offs[t + 1] - 1, ntl = (tile + 1) * LT;
            int64_t open_len = end - ntl + 1;
            // (a partial last tile can be both "fully covered" = interior, evaluated by k_lpass, and short: interior wins)
            if (end / LT == tile + 1 && open_len <= SPAN_SHORT && a_tile_rec[tile + 1].w == 0) {
                int4 td = a_tdesc[t];
                int64_t toff = a_offs[t];
                int64_t B = td.x, r = td.z;
                int64_t base = (int64_t)td.y + a_tileS[tile];           // counts up to the end of the head tile
                int64_t base2 = HYP ? (int64_t)a_tS0l[t] + a_tileS2[tile] : 0;
                Best<TC, HYP> best = partR[tile];
                for (int64_t e = ntl; e <= end; e++) {                  // decreasing p: earlier candidates win ties
                    int64_t p = B - (e - toff);
                    int64_t nn = base + a_loc[e], nl = HYP ? base2 + a_loc2[e] : 0;
                    TC fv = dm_apply(M, alpha, r - p, (int64_t)(td.w - a_pos[p]), nn, nl);
                    Best<TC, HYP> c; best_clear(c); c.v = cadd(W[p], fv); c.p = (int32_t)p; c.nn = (int32_t)nn; best_set_nl(c, (int32_t)nl);
                    best = better(best, c);
                }
                int b = a_tb[t];
                a_opt[(int64_t)b * n1 + PR(r)] = best.p;
                a_nnopt[(int64_t)b * n1 + PR(r)] = best.nn;
                if (HYP) a_nlopt[(int64_t)b * n1 + PR(r)] = best_nl(best);
            } else {
                fix_long = true;
            }
        }
    }
    // ---- role 2: this tile starts inside a task of a long span
    bool open_long = false;
    if (in && a_tile_rec[tile].w == 0) {                   // interior tiles were evaluated by k_lpass
        int64_t tile_start = tile * LT;
        int64_t t0 = a_tile_t0[tile];
        int64_t toff0 = a_offs[t0];
        if (toff0 < tile_start) {
            int64_t end0 = a_offs[t0 + 1] - 1, h = toff0 / LT;
            bool is_short = (end0 / LT == h + 1) && (end0 - tile_start + 1 <= SPAN_SHORT);
            open_long = !is_short;
        }
    }
    // wave-aggregated appends
    unsigned long long mo = __ballot(open_long), mf = __ballot(fix_long);
    int32_t bo = 0, bf = 0;
    if (lane == 0) {
        if (mo) bo = atomicAdd(&rc->n_open, (int32_t)__popcll(mo));
        if (mf) bf = atomicAdd(&rc->n_fix, (int32_t)__popcll(mf));
    }
    bo = __shfl(bo, 0); bf = __shfl(bf, 0);
    unsigned long long below = (1ull << lane) - 1ull;
    if (open_long) open_list[bo + __popcll(mo & below)] = (int32_t)tile;
    if (fix_long) fix_list[bf + __popcll(mf & below)] = (int32_t)tile;
    }
}

// open-left part of a tile of a long span (the task started in an earlier tile): one wave per listed tile
template <typename TC, bool HYP>
__global__ void __launch_bounds__(256) k_open(RoundDesc R, const RoundCounts *__restrict__ rc, const int64_t *__restrict__ a_offs,
                                              const int4 *__restrict__ a_tdesc,
                                              const int32_t *__restrict__ a_tS0l,
                                              const int32_t *__restrict__ a_pos, const int32_t *__restrict__ a_loc,
                                              const int32_t *__restrict__ a_loc2, const int64_t *__restrict__ a_tile_t0,
                                              const int64_t *__restrict__ tilePS,
                                              const int64_t *__restrict__ tilePS2, const TC *__restrict__ W, DevModel<TC> M, TC alpha,
                                              Best<TC, HYP> *__restrict__ partL, const int32_t *__restrict__ open_list)
{
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t cnt = rc->n_open, T = rc->T;
    for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < cnt; w += nw) {
        int64_t tile = open_list[w];
        int64_t tile_start = tile * LT;
        int64_t t = a_tile_t0[tile];                        // task of the tile's first step
        int64_t toff = a_offs[t];
        int64_t last = a_offs[t + 1] - 1;
        int64_t tile_last = tile_start + LT - 1;
        if (tile_last > T - 1) tile_last = T - 1;
        if (last > tile_last) last = tile_last;
        int4 td = a_tdesc[t];
        int64_t r = td.z, B = td.x;
        // counts of this task in earlier tiles: its head tile contributes "since the last head", every tile in
        // between is covered entirely by the task: a range sum over the per-tile tails (tilePS = their prefix sums)
        int64_t base = (int64_t)td.y + (tilePS[tile] - tilePS[toff / LT]);
        int64_t base2 = HYP ? (int64_t)a_tS0l[t] + (tilePS2[tile] - tilePS2[toff / LT]) : 0;
        Best<TC, HYP> best; best_clear(best);
        for (int64_t e = tile_start + lane; e <= last; e += 64) {
            int64_t p = B - (e - toff);
            int64_t nn = base + a_loc[e], nl = HYP ? base2 + a_loc2[e] : 0;
            TC fv = dm_apply(M, alpha, r - p, (int64_t)(td.w - a_pos[p]), nn, nl);
            Best<TC, HYP> c; best_clear(c); c.v = cadd(W[p], fv); c.p = (int32_t)p; c.nn = (int32_t)nn; best_set_nl(c, (int32_t)nl);
            best = better(best, c);                         // a lane visits its steps in increasing e (decreasing p)
        }
        // wave arg-min; ties -> larger p
        for (int o = 32; o > 0; o >>= 1) {
            int src = (lane + o) & 63;
            Best<TC, HYP> c; best_clear(c); c.v = shfl64(best.v, src); c.p = __shfl(best.p, src); c.nn = __shfl(best.nn, src);
            if (HYP) best_set_nl(c, __shfl(best_nl(best), src));
            bool take = (best.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < best.v || (c.v == best.v && c.p > best.p)));
            if (lane + o < 64 && take) best = c;
        }
        if (lane == 0) partL[tile] = best;
    }
}

// a task of a long span: combine the head tile's partial with the partials of the tiles it covers
// (one wave per listed head tile)
template <typename TC, bool HYP>
__global__ void __launch_bounds__(256) k_fix(const RoundCounts *__restrict__ rc, const int64_t *__restrict__ offs, const int64_t *__restrict__ taskR,
                                             const Best<TC, HYP> *__restrict__ partL, const Best<TC, HYP> *__restrict__ partR,
                                             const int4 *__restrict__ tdesc, const uint8_t *__restrict__ tb,
                                             int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int32_t *__restrict__ nlopt, int64_t n1,
                                             const int32_t *__restrict__ fix_list,
                                             const int4 *__restrict__ tile_rec, const int64_t *__restrict__ tilePS, const int64_t *__restrict__ tilePS2,
                                             const int32_t *__restrict__ tS0l, DevModel<TC> M)
{
    int lane = threadIdx.x & 63;
    int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t cnt = rc->n_fix;
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < cnt; w += nw) {
        int64_t tile = fix_list[w];
        int64_t t = taskR[tile];
        int64_t end_tile = (offs[t + 1] - 1) / LT;
        int64_t S0 = tdesc[t].y, S0l = HYP ? (int64_t)tS0l[t] : 0;
        Best<TC, HYP> acc; best_clear(acc);
        for (int64_t k = tile + 1 + lane; k <= end_tile; k += 64) {
            Best<TC, HYP> c = partL[k];
            if (tile_rec[k].w != 0 && c.p >= 0) {            // interior tile: counts and value are relative to the tile start
                int64_t base = S0 + (tilePS[k] - tilePS[tile]);
                int64_t base2 = HYP ? S0l + (tilePS2[k] - tilePS2[tile]) : 0;
                c.v = cadd(c.v, dm_apply(M, (TC)0, (int64_t)0, (int64_t)0, base, base2));       // the costs are affine in the counts
                c.nn = (int32_t)(c.nn + base);
                if (HYP) best_set_nl(c, (int32_t)(best_nl(c) + base2));
            }
            bool take = (acc.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < acc.v || (c.v == acc.v && c.p > acc.p)));
            if (take) acc = c;
        }
        for (int o = 32; o > 0; o >>= 1) {
            int src = (lane + o) & 63;
            Best<TC, HYP> c; best_clear(c); c.v = shfl64(acc.v, src); c.p = __shfl(acc.p, src); c.nn = __shfl(acc.nn, src);
            if (HYP) best_set_nl(c, __shfl(best_nl(acc), src));
            bool take = (acc.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < acc.v || (c.v == acc.v && c.p > acc.p)));
            if (lane + o < 64 && take) acc = c;
        }
        if (lane == 0) {
            Best<TC, HYP> res = better(partR[tile], acc);   // the head tile holds the larger p: wins ties
            int64_t r = tdesc[t].z; int b = tb[t];
            opt[(int64_t)b * n1 + PR(r)] = res.p;
            nnopt[(int64_t)b * n1 + PR(r)] = res.nn;
            if (HYP) nlopt[(int64_t)b * n1 + PR(r)] = best_nl(res);
        }
    }
}

// ------------------------------------------------------------------ combine the per-bit winners of every row
// Which row a thread of the combine kernels takes.  lvl = 0: row rlo + i.  lvl = 1 (an experiment kept behind a debug option, see
// run_layer): the row of plane slot i -- the order the planes are stored in (prow); row 0 is thread n.  Returns -1 for a thread
// without a row.
__device__ __forceinline__ int64_t combine_row(int64_t i, int64_t n, int64_t rlo, int64_t rhi, int lvl)
{
    int64_t r;
    if (!lvl) r = rlo + i;
    else if (i == n) r = 0;
    else if (i > n) return -1;
    else {
        int t = 0;
        while (t < 62 && n - (n >> (t + 1)) <= i) t++;             // level of slot i: base(t) <= i < base(t + 1), base(t) = n - (n >> t)
        r = (((i - (n - (n >> t))) << 1) | 1) << t;
    }
    return (r < rlo || r > rhi) ? -1 : r;
}

template <typename TC>
__global__ void __launch_bounds__(256) k_combine(int lvl, int64_t n, int64_t rlo, int64_t rhi, int nbits, const int32_t *__restrict__ pos,
                                                 const int32_t *__restrict__ opt, const int32_t *__restrict__ nnopt,
                                                 const int32_t *__restrict__ nlopt,
                                                 const TC *__restrict__ W, DevModel<TC> M, TC alpha,
                                                 TC *__restrict__ cst, int32_t *__restrict__ ptr, int sh, int32_t *__restrict__ pz)
{
    // sh > 0 (after a leaf pass): the rows (rlo + i) << sh only -- the leaf rows were combined where they were computed
    int64_t r = combine_row((int64_t)blockIdx.x * blockDim.x + threadIdx.x, n, rlo, rhi, lvl);
    if (r < 0) return;
    r <<= sh;
    int64_t n1 = n + 1;
    TC bv = cadd(W[r], dm_apply(M, alpha, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0));   // j = j' (empty part)
    int64_t bp = r;
    for (int b = 0; b < nbits; b++) {
        if (!((r >> b) & 1)) continue;
        int64_t p = opt[(int64_t)b * n1 + PR(r)];
        if (pz && (uint64_t)p > (uint64_t)n) { atomicAdd(pz, 1); p = r; }      // (poison mode: a cell nobody wrote)
        int64_t nn = nnopt[(int64_t)b * n1 + PR(r)];
        int64_t nl = nlopt ? nlopt[(int64_t)b * n1 + PR(r)] : 0;
        TC v = cadd(W[p], dm_apply(M, alpha, r - p, (int64_t)(pos[r] - pos[p]), nn, nl));
        if (v < bv) { bv = v; bp = p; }            // lower bits hold larger p: strict < keeps the largest p on ties
    }
    cst[r] = bv;
    ptr[r] = (int32_t)bp;
}

// windowed layers: the candidates of row r from the right: the diagonal, the standard planes by ascending bit, the common plane,
// the mirrored planes by descending bit -- strict < while moving left keeps the largest p on ties
template <typename TC>
__global__ void __launch_bounds__(256) k_combine_win(int lvl, Geo G, int64_t n, int64_t rlo, int64_t rhi, const int32_t *__restrict__ pos,
                                                     const int32_t *__restrict__ opt, const int32_t *__restrict__ nnopt,
                                                     const int32_t *__restrict__ nlopt,
                                                     const TC *__restrict__ W, DevModel<TC> M, TC alpha,
                                                     TC *__restrict__ cst, int32_t *__restrict__ ptr, int sh, int32_t *__restrict__ pz)
{
    int64_t r = combine_row((int64_t)blockIdx.x * blockDim.x + threadIdx.x, n, rlo, rhi, lvl);
    if (r < 0) return;
    r <<= sh;                                                   // (after a leaf pass: the multiples of 64 only)
    const int64_t n1 = n + 1;
    TC bv = cadd(W[r], dm_apply(M, alpha, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0));   // j = j' (empty part)
    int64_t bp = r;
    if (r >= 1) {
        const int64_t slot = PR(r);
        for (int i = 0; i <= 2 * G.s; i++) {
            const int b = i <= G.s ? i : 2 * G.s - i;                       // 0 .. s, then s-1 .. 0
            if (b < G.s && (((r >> b) & 1) != (i <= G.s))) continue;      // first pass: set bits (standard); second pass: clear bits (mirrored)
            if (b == G.s && i != G.s) continue;
            int64_t cs, ce;
            if (!geo_block(G, r, b, cs, ce)) continue;
            int64_t p = opt[(int64_t)b * n1 + slot];
            if (pz && (uint64_t)p > (uint64_t)n) { atomicAdd(pz, 1); p = r; }
            const int64_t nn = nnopt[(int64_t)b * n1 + slot];
            const int64_t nl = nlopt ? nlopt[(int64_t)b * n1 + slot] : 0;
            const TC v = cadd(W[p], dm_apply(M, alpha, r - p, (int64_t)(pos[r] - pos[p]), nn, nl));
            if (v < bv) { bv = v; bp = p; }
        }
    }
    cst[r] = bv;
    ptr[r] = (int32_t)bp;
}

// ---- anchors of the mirrored head tasks (windowed layers), once per (pattern, w)
// hist[e] = #{q : e(q) = e}, e(q) = min(next[q], col[q] + w, n): then  nets(max(0, r - w), r) = pos[r] - #{q : e(q) < r}   (an entry is counted by
// the window ending before r iff it lies in a column < r, is the LAST occurrence of its row before r, and its column is >= r - w)
// Without atomics (10^8 scattered atomic adds took 12.5 ms; this takes the time of two column passes): the entries with e = x are
//   * next[q] = x <= col[q] + w : one per entry q' of COLUMN x whose previous occurrence is within the window, prev[q'] >= max(0, x - w);
//   * col[q] + w = x < next[q]  : the entries of column x - w that do not come back by column x;
// one lane per column x < n (hist[n] is never read: E[r] sums the values below r <= n), eight loads in flight.
__global__ void __launch_bounds__(256) k_win_hist(int64_t n, int64_t w, const int32_t *__restrict__ pos, const int32_t *__restrict__ prev,
                                                  const int32_t *__restrict__ next, int32_t *__restrict__ hist)
{
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x > n) return;
    int32_t c = 0;
    if (x < n) {
        const int32_t lo = (int32_t)(x > w ? x - w : 0);
        for (int32_t q = pos[x], q1 = pos[x + 1]; q < q1; q += 8) {
            int32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = q + k < q1 ? prev[q + k] : -1;
#pragma unroll
            for (int k = 0; k < 8; k++) c += (v[k] >= lo);
        }
        if (x >= w) {
            const int32_t xx = (int32_t)x;
            for (int32_t q = pos[x - w], q1 = pos[x - w + 1]; q < q1; q += 8) {
                int32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = q + k < q1 ? next[q + k] : INT32_MIN;
#pragma unroll
                for (int k = 0; k < 8; k++) c += (v[k] > xx);
            }
        }
    }
    hist[x] = c;
}
// the same for whole rows (self nets): rows with first >= r - w and last < r  =  #{last < r} - #{max(last, first + w) < r}
__global__ void __launch_bounds__(256) k_win_hist_rows(int64_t m, int64_t n, int64_t w, const int32_t *__restrict__ rfirst, const int32_t *__restrict__ rlast,
                                                       int32_t *__restrict__ hist)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int64_t f = rfirst[i];
    if (f < 0) return;                                      // empty row
    int64_t e = f + w, l = rlast[i];
    if (l > e) e = l;
    if (e > n) e = n;
    atomicAdd(&hist[e], 1);
}
// one wave per mirrored head (plane b < s, rho = idx << (b+1)): entries of the block's columns passing the test
//   tot[aoff[b] + idx] = #{q in the columns of the block : next[q] >= rho}   (GE)   /   #{rows starting in the block with last < rho}
template <bool GE>
__global__ void __launch_bounds__(256) k_win_block_totals(RoundDesc R, int64_t nitems, const int32_t *__restrict__ cpos, const int32_t *__restrict__ arr,
                                                          int32_t *__restrict__ tot)
{
    const int lane = threadIdx.x & 63;
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < nitems; it += (int64_t)gridDim.x * 4) {
        int b = 0;
        while (it >= R.aoff[b + 1]) b++;
        const int64_t idx = it - R.aoff[b], rho = idx << (b + 1);
        int32_t c = 0;
        int64_t cs, ce;
        if (idx >= 1 && rho <= R.n && geo_block(R.G, rho, b, cs, ce)) {
            const int32_t thr = (int32_t)rho;
            for (int32_t q = cpos[cs] + lane, q1 = cpos[ce]; q < q1; q += 64) { const int32_t v = arr[q]; c += GE ? (v >= thr) : (v < thr); }
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) tot[it] = c;
    }
}
// anchor of (b, rho) = count of the whole window [max(0, rho - w), rho) minus the totals of the blocks b' <= b left of the block's end
__global__ void __launch_bounds__(256) k_win_anchors(RoundDesc R, int64_t nitems, const int64_t *__restrict__ base, const int64_t *__restrict__ E,
                                                     const int32_t *__restrict__ tot, int32_t *__restrict__ anch)
{
    int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= nitems) return;
    int b = 0;
    while (it >= R.aoff[b + 1]) b++;
    const int64_t idx = it - R.aoff[b], rho = idx << (b + 1);
    if (idx < 1 || rho > R.n) { anch[it] = 0; return; }
    int64_t v = base[rho] - E[rho];
    for (int bb = 0; bb <= b; bb++) v -= tot[R.aoff[bb] + (rho >> (bb + 1))];
    anch[it] = (int32_t)v;
}

// leaf pass: the part [64 g + 62 - w, 64 g) of every group (E from the histogram of width w - 62)
__global__ void __launch_bounds__(256) k_leaf_anchors(int64_t ng, int64_t n, const int64_t *__restrict__ base, const int64_t *__restrict__ E,
                                                      int32_t *__restrict__ anch)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const int64_t r = g << 6;
    anch[g] = r <= n ? (int32_t)(base[r] - E[r]) : 0;
}

// ------------------------------------------------------------------ round A from cached counts
// Round A (row r against its lowest block [r - 2^b, r), b = ctz(r)) is the one round whose candidate ranges do not depend on
// the layer: nets(p, r) for all its (row, candidate) pairs -- sum_r 2^ctz(r) = n log2(n) / 2 values -- is computed ONCE per
// partition and kept (4 B each); a layer then evaluates round A by streaming counts, W and pos (16 B per candidate instead
// of the candidate's whole column).  Elements are stored level by level (b), row by row, candidate p = r - 1 - i at index i;
// every level is padded to whole tiles of LT elements.
// tiles of level b: [tbase[b], tbase[b+1]); rows of the levels >= 9: [rbase[b], rbase[b+1]).
// mir_w = 0: the unconstrained scheme's round-A rows -- level b holds the rows r = (2u+1) 2^b, candidates r - 1 - i, i < 2^b.
// mir_w = w > 0: the MIRRORED heads of the windowed geometry (geo_block) -- level b < s holds the rows r = (u+1) 2^(b+1),
// candidates ce - 1 - i with ce = r + 2^(b+1) - 1 - w (those below column 0 do not exist); the counts carry the head's anchor
// nets(ce, r), at aoff[b] + (r >> (b+1)) of the window anchors.  tlo / thi: the tiles of each level a windowed layer needs.
struct RATab { int32_t nbits, _pad; int64_t n; int64_t tbase[33]; int64_t rbase[33]; int64_t mir_w; int64_t aoff[33]; int64_t tlo[33], thi[33]; };

__device__ __forceinline__ int64_t ra_row(const RATab &T, int b, int64_t u)
{
    return T.mir_w ? (u + 1) << (b + 1) : ((u << 1) | 1) << b;
}

__device__ __forceinline__ bool ra_decode(const RATab &T, int64_t tile, int j, int &b, int64_t &r, int64_t &p, int64_t &i)
{
    b = 0;
    while (tile >= T.tbase[b + 1]) b++;                     // (uniform per tile)
    int64_t e = (tile - T.tbase[b]) * LT + j;               // element inside the level
    int64_t u = e >> b;
    i = e & (((int64_t)1 << b) - 1);
    r = ra_row(T, b, u);
    p = (T.mir_w ? r + ((int64_t)2 << b) - 1 - T.mir_w : r) - 1 - i;
    return r <= T.n && p >= 0;
}

// d[E] = #{q in column p : next[q] >= r} (d2: rows whose first column is p and whose last column is < r)
__global__ void __launch_bounds__(256) k_ra_colcount(RATab T, const int32_t *__restrict__ pos, const int32_t *__restrict__ next,
                                                     const int32_t *__restrict__ fpos, const int32_t *__restrict__ flast,
                                                     int32_t *__restrict__ d, int32_t *__restrict__ d2)
{
    int b; int64_t r, p, i;
    bool ok = ra_decode(T, blockIdx.x, threadIdx.x, b, r, p, i);
    int64_t E = (int64_t)blockIdx.x * LT + threadIdx.x;
    int32_t c = 0, c2 = 0;
    if (ok) {
        int32_t rr = (int32_t)r;
        for (int32_t q = pos[p], q1 = pos[p + 1]; q < q1; q += 8) {        // eight loads in flight (one at a time: a memory latency per entry)
            int32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = q + k < q1 ? next[q + k] : INT32_MIN;
#pragma unroll
            for (int k = 0; k < 8; k++) c += (v[k] >= rr);
        }
        if (d2) for (int32_t q = fpos[p], q1 = fpos[p + 1]; q < q1; q++) c2 += (flast[q] < rr);
    }
    d[E] = c;
    if (d2) d2[E] = c2;
}

// c[E] = sum of d over the row's elements 0 .. i  (G = exclusive prefix sums of d over all elements)
__global__ void __launch_bounds__(256) k_ra_final(RATab T, const int64_t *__restrict__ G, int32_t *__restrict__ c, const int32_t *__restrict__ anch)
{
    int b; int64_t r, p, i;
    ra_decode(T, blockIdx.x, threadIdx.x, b, r, p, i);
    int64_t E = (int64_t)blockIdx.x * LT + threadIdx.x;
    int32_t v = (int32_t)(G[E + 1] - G[E - i]);
    if (anch && r <= T.n) v += anch[T.aoff[b] + (r >> (b + 1))];            // (mirrored heads: the part [ce, r) the block's columns stand on)
    c[E] = v;
}

template <typename TC, bool HYP>
__device__ __forceinline__ bool ra_takes(const Best<TC, HYP> &x, const Best<TC, HYP> &c)      // c replaces x: smaller value, or equal value and larger p
{
    return (x.p < 0) ? (c.p >= 0) : (c.p >= 0 && (c.v < x.v || (c.v == x.v && c.p > x.p)));
}

// one wave per tile of LT elements; a lane holds four consecutive elements.  Rows of up to LT candidates (b <= 8) lie inside
// one tile and are finished here; longer rows leave one partial per tile for k_ra_merge.
template <typename TC, bool HYP>
__global__ void __launch_bounds__(256) k_ra_layer(RATab T, int64_t ntile, const int32_t *__restrict__ cnt, const int32_t *__restrict__ cnt2,
                                                  const int32_t *__restrict__ pos, const TC *__restrict__ W, DevModel<TC> M, TC alpha,
                                                  int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int32_t *__restrict__ nlopt,
                                                  Best<TC, HYP> *__restrict__ part)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntile) return;
    const int64_t n1 = T.n + 1;
    const int64_t E0 = tile * LT + 4 * lane;
    const int4 c4 = *reinterpret_cast<const int4 *>(cnt + E0);
    int4 d4 = make_int4(0, 0, 0, 0);
    if (HYP) d4 = *reinterpret_cast<const int4 *>(cnt2 + E0);
    // the lane's first element is decoded; from level 2 on the other three are the next candidates of the same row
    int b = 0;
    int64_t r0, p0, i0;
    const bool ok0 = ra_decode(T, tile, 4 * lane, b, r0, p0, i0);
    if (tile < T.tlo[b] || tile >= T.thi[b]) return;        // (uniform) a windowed layer: rows far from its window
    Best<TC, HYP> cand[4];
    int64_t rk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int64_t r = r0, p = p0 - k;
        bool ok = ok0 && p >= 0;                            // (mirrored heads: a block may be cut off at column 0)
        if (b < 2) { int bb; int64_t ii; ok = ra_decode(T, tile, 4 * lane + k, bb, r, p, ii); }      // (uniform branch)
        rk[k] = ok ? r : 0;
        best_clear(cand[k]);
        if (ok) {
            int32_t c = k == 0 ? c4.x : k == 1 ? c4.y : k == 2 ? c4.z : c4.w, c2 = k == 0 ? d4.x : k == 1 ? d4.y : k == 2 ? d4.z : d4.w;
            cand[k].v = cadd(W[p], dm_apply(M, alpha, r - p, (int64_t)(pos[r] - pos[p]), (int64_t)c, (int64_t)c2));
            cand[k].p = (int32_t)p; cand[k].nn = c; best_set_nl(cand[k], c2);
        }
    }
    if (b < 2) {                                            // rows of one or two candidates: finished inside the lane
        const int per = 1 << b;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((k & (per - 1)) != 0 || rk[k] == 0) continue;
            Best<TC, HYP> x = cand[k];
            if (per == 2 && ra_takes(x, cand[k + 1 < 4 ? k + 1 : 3])) x = cand[k + 1 < 4 ? k + 1 : 3];
            int64_t rw = (int64_t)b * n1 + PR(rk[k]);
            opt[rw] = x.p; nnopt[rw] = x.nn;
            if (HYP) nlopt[rw] = best_nl(x);
        }
        return;
    }
    Best<TC, HYP> x = cand[0];
#pragma unroll
    for (int k = 1; k < 4; k++) if (ra_takes(x, cand[k])) x = cand[k];
    const int lpr = b >= 8 ? 64 : (1 << (b - 2));          // lanes per row
    for (int o = 1; o < lpr; o <<= 1) {                     // butterfly inside the row's lanes: every lane ends with the winner
        Best<TC, HYP> c; best_clear(c);
        c.v = shfl64(x.v, lane ^ o); c.p = __shfl(x.p, lane ^ o); c.nn = __shfl(x.nn, lane ^ o);
        if (HYP) best_set_nl(c, __shfl(best_nl(x), lane ^ o));
        if (ra_takes(x, c)) x = c;
    }
    if ((lane & (lpr - 1)) == 0) {
        if (b <= 8) {
            if (rk[0]) {
                int64_t rw = (int64_t)b * n1 + PR(rk[0]);
                opt[rw] = x.p; nnopt[rw] = x.nn;
                if (HYP) nlopt[rw] = best_nl(x);
            }
        } else {
            part[tile - T.tbase[9]] = x;
        }
    }
}

// The same round, COLUMN-major: one wave per 256 consecutive candidates p, all levels.  A candidate p belongs to the round-A row
// of every level b whose bit is clear in p (r = p with the bits below b cleared, plus 2^b), a dozen rows on average: k_ra_layer reads
// its cost W[p] and its column pointer once per row (16 B per (candidate, row)); here they are read once per candidate and
// only the 4-byte cached count is read per row -- 60 B instead of 190 B per candidate and layer.  Lane l holds the candidates
// P0 + 4 l .. + 3; the count elements of a row are stored by descending p (ra_decode), so a lane's four are one aligned vector.
// Rows of up to 256 candidates (b <= 8) lie inside the wave's columns and are finished here (butterfly over the row's lanes);
// longer rows leave one partial per 256 candidates for k_ra_merge, in the slot k_ra_layer would use.
template <typename TC, bool HYP, bool MIR>
__global__ void __launch_bounds__(256) k_ra_cols(RATab T, const int32_t *__restrict__ cnt, const int32_t *__restrict__ cnt2,
                                                 const int32_t *__restrict__ pos, const TC *__restrict__ W, DevModel<TC> M, TC alpha,
                                                 int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int32_t *__restrict__ nlopt,
                                                 Best<TC, HYP> *__restrict__ part, int64_t tile0, int64_t tile1, int bmin)
{
    // levels bmin .. nbits - 1 (the leaf pass computes the levels below LEAF_T itself: their rows are never multiples of 64).
    // tiles [tile0, tile1) of 256 consecutive values of the coordinate z (a windowed layer needs the rows near its window only).
    // MIR = false: z = p, the candidate p belongs to the level-b row of the block [r - 2^b, r) iff bit b of z is CLEAR.
    // MIR = true (table with mir_w = w): z = p + w + 1, p belongs to the mirrored head rho = z with the bits <= b cleared
    // (block [rho + 2^b - 1 - w, rho + 2^(b+1) - 1 - w)) iff bit b of z is SET.  In both maps a row's 2^b candidates are an aligned
    // block of z and its elements run by descending z: element i = (2^b - 1) - (z mod 2^b).
    const int lane = threadIdx.x & 63;
    const int64_t n = T.n, n1 = n + 1, off = MIR ? T.mir_w + 1 : 0;
    const int64_t Z0 = (tile0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * LT;
    if (Z0 - off >= n || Z0 >= tile1 * LT) return;          // (wave-uniform) candidates are the columns 0 .. n - 1
    const int64_t z0 = Z0 + 4 * lane, p0 = z0 - off;
    TC wv[4]; int32_t pv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { int64_t p = p0 + k; p = p < 0 ? 0 : p < n ? p : n; wv[k] = W[p]; pv[k] = pos[p]; }
    for (int b = bmin; b < T.nbits; b++) {
        if (b >= 8 && (bool)((Z0 >> b) & 1) != MIR) continue;       // (wave-uniform) no row of this level over the wave's columns
        // the lane's row of this level and its candidates among z0 .. z0 + 3:
        //   b = 0: two rows of one candidate each; b = 1: one row of two; b >= 2: all four
        const int64_t mask = ((int64_t)1 << b) - 1;
        const bool cov = b < 2 || (bool)((z0 >> b) & 1) == MIR;
        const int64_t zr = (z0 >> (b + 1)) << (b + 1);              // z0 with the bits <= b cleared
        const int64_t r = MIR ? zr : zr + ((int64_t)1 << b);        // (b = 0: the first of the lane's two rows)
        const int64_t u = MIR ? (z0 >> (b + 1)) - 1 : z0 >> (b + 1);
        // the lane's candidates k0, k0 + 1 (b = 1) or 0 .. 3 (b >= 2); element of the LAST one (the smallest index)
        const int k0 = (b == 1 && MIR) ? 2 : 0;
        const int64_t zlast = b == 1 ? z0 + k0 + 1 : z0 + 3;
        const int64_t E = (T.tbase[b] << 8) + ((u << b) | (mask - (zlast & mask)));
        int32_t c[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
        if (b == 0) {
            // single candidates: z0 + k (k = 0, 2; mirrored: 1, 3) -> rows r, r + 2; their elements are neighbours
            const int kk = MIR ? 1 : 0;
            const int64_t E0 = (T.tbase[0] << 8) + u;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int64_t rk = r + 2 * j, pk = p0 + kk + 2 * j;
                if (rk >= 1 && rk <= n && pk >= 0 && (!MIR || u + j >= 0)) {
                    const int64_t rw = PR(rk);
                    opt[rw] = (int32_t)pk; nnopt[rw] = cnt[E0 + j]; if (HYP) nlopt[rw] = cnt2[E0 + j];
                }
            }
            continue;
        }
        const bool rok = cov && r >= 1 && r <= n && u >= 0;
        if (rok) {
            if (b >= 2) {
                const int4 t = *reinterpret_cast<const int4 *>(cnt + E);
                c[3] = t.x; c[2] = t.y; c[1] = t.z; c[0] = t.w;
                if (HYP) { const int4 t2 = *reinterpret_cast<const int4 *>(cnt2 + E); d[3] = t2.x; d[2] = t2.y; d[1] = t2.z; d[0] = t2.w; }
            } else {
                const int2 t = *reinterpret_cast<const int2 *>(cnt + E);
                c[k0 + 1] = t.x; c[k0] = t.y;
                if (HYP) { const int2 t2 = *reinterpret_cast<const int2 *>(cnt2 + E); d[k0 + 1] = t2.x; d[k0] = t2.y; }
            }
        }
        const int32_t posr = pos[rok ? r : 0];
        Best<TC, HYP> x; best_clear(x);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (b == 1 && (k < k0 || k > k0 + 1)) continue;
            if (!rok || p0 + k < 0) continue;               // (mirrored heads: a block may be cut off at column 0)
            Best<TC, HYP> cc; best_clear(cc);
            cc.v = cadd(wv[k], dm_apply(M, alpha, r - (p0 + k), (int64_t)(posr - pv[k]), (int64_t)c[k], (int64_t)d[k]));
            cc.p = (int32_t)(p0 + k); cc.nn = c[k]; best_set_nl(cc, d[k]);
            if (ra_takes(x, cc)) x = cc;
        }
        const int lpr = b <= 2 ? 1 : b >= 8 ? 64 : (1 << (b - 2));          // lanes per row
        for (int o = 1; o < lpr; o <<= 1) {                 // butterfly inside the row's lanes: every lane ends with the winner
            Best<TC, HYP> cc; best_clear(cc);
            cc.v = shfl64(x.v, lane ^ o); cc.p = __shfl(x.p, lane ^ o); cc.nn = __shfl(x.nn, lane ^ o);
            if (HYP) best_set_nl(cc, __shfl(best_nl(x), lane ^ o));
            if (ra_takes(x, cc)) x = cc;
        }
        if ((lane & (lpr - 1)) == 0 && rok) {
            if (b <= 8) {
                if (x.p >= 0) {
                    const int64_t rw = (int64_t)b * n1 + PR(r);
                    opt[rw] = x.p; nnopt[rw] = x.nn;
                    if (HYP) nlopt[rw] = best_nl(x);
                }
            } else {
                // lane 0: the wave's smallest element of the row is that of its last column
                const int64_t el = (u << b) | (mask - ((Z0 + LT - 1) & mask));
                part[T.tbase[b] + (el >> 8) - T.tbase[9]] = x;
            }
        }
    }
}

// rows of more than LT candidates (b >= 9): one wave per row merges the row's tile partials
template <typename TC, bool HYP>
__global__ void __launch_bounds__(256) k_ra_merge(RATab T, int64_t nrow, const Best<TC, HYP> *__restrict__ part,
                                                  int32_t *__restrict__ opt, int32_t *__restrict__ nnopt, int32_t *__restrict__ nlopt)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nrow) return;
    int b = 9;
    while (w >= T.rbase[b + 1]) b++;
    const int64_t u = w - T.rbase[b], r = ra_row(T, b, u), n1 = T.n + 1;
    const int64_t t0 = T.tbase[b] + (u << (b - 8)) - T.tbase[9], cntt = (int64_t)1 << (b - 8);
    Best<TC, HYP> x; best_clear(x);
    // (the top row's block is half the matrix -- 2^15 partials at n = 10^7: eight loads per lane in flight; one at a time, that row
    //  alone made the kernel 160 us long)
    for (int64_t kb = lane; kb < cntt; kb += 512) {
        Best<TC, HYP> cc[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { best_clear(cc[u]); if (kb + 64 * u < cntt) cc[u] = part[t0 + kb + 64 * u]; }
#pragma unroll
        for (int u = 0; u < 8; u++) if (kb + 64 * u < cntt && ra_takes(x, cc[u])) x = cc[u];
    }
    for (int o = 32; o > 0; o >>= 1) {
        Best<TC, HYP> c; best_clear(c);
        c.v = shfl64(x.v, lane ^ o); c.p = __shfl(x.p, lane ^ o); c.nn = __shfl(x.nn, lane ^ o);
        if (HYP) best_set_nl(c, __shfl(best_nl(x), lane ^ o));
        if (ra_takes(x, c)) x = c;
    }
    if (lane == 0 && r <= T.n) {
        int64_t rw = (int64_t)b * n1 + PR(r);
        opt[rw] = x.p; nnopt[rw] = x.nn;
        if (HYP) nlopt[rw] = best_nl(x);
    }
}

// ------------------------------------------------------------------ end of a round's counting phase
// One thread: derives the tile count and checks the scan totals against the capacities of the buffers the host sized from
// its prediction (the previous layer); on overflow the round's work is dropped and the flag makes the host redo the layer.
__global__ void k_round_finish(RoundCounts *__restrict__ rc, int64_t capT, int64_t capNT)
{
    if (rc->T > capT || rc->NT > capNT || rc->err) { rc->err = 1; rc->T = 0; rc->NT = 0; rc->nlong = 0; rc->nown = 0; }
    rc->ntile = (int32_t)((rc->T + LT - 1) / LT);
}

#include "dp_leaf.inc"

// ------------------------------------------------------------------ host driver for one layer
template <typename TC>
struct LayerWork {
    int64_t n = -1; int nbits = 0; bool hyp = false;
    DBuf<int32_t> opt, nnopt, cr, len, loc, tileS;
    DBuf<int4> tdesc;                                    // per task {B, anchor count, row r, pos32[r]}
    DBuf<int32_t> nlopt, crl, tS0l, loc2, tileS2;        // hyperedge-cut: second (self-net) count
    DBuf<uint8_t> tb;
    DBuf<int64_t> offs, scratch, taskR, tilePS, tilePS2, tile_t0;
    ScanWS scanws;                                      // single-launch scans of the rounds
    DBuf<Best<TC, true>> partL, partR;                  // sized for the larger record; reinterpreted per variant
    DBuf<int32_t> open_list, fix_list;                  // tiles of long spans (k_span_short -> k_open / k_fix)
    DBuf<int4> tile_rec;                                // {first column, row, -, interior?} per tile (k_tile_t0)
    // tasks with tiles of their own (k_lpass_own)
    DBuf<int4> o_tdesc, o_rec;
    DBuf<uint8_t> o_tb;
    DBuf<int32_t> o_rlen, o_ntl, o_tS0l, o_task, o_tileS, o_tileS2, o_wide, o_hi;
    DBuf<Best<TC, true>> o_sub;                         // gap passes: segment winners of the tiles with specials, [tile][SMAX + 1]
    DBuf<int32_t> o_spv;                                // ... and the specials between them
    // round A from cached counts
    bool ra_built = false; RATab ra_tab; int64_t ra_ntile = 0, ra_nrow = 0;
    bool mir_built = false; RATab mir_tab; int64_t mir_ntile = 0, mir_nrow = 0, mir_w = 0;      // the mirrored heads of a window width (ra_build_mir)
    DBuf<int32_t> mir_c, mir_c2;
    DBuf<Best<TC, true>> mir_part;
    DBuf<int32_t> ra_c, ra_c2;
    DBuf<int64_t> ra_G;                                 // prefix sums while the counts are built
    DBuf<Best<TC, true>> ra_part;
    DBuf<int2> g_slot;                                  // segment slot -> {task, segment}
    DBuf<int2> g_list;                                  // gap tasks of more than GAPSEG tiles: {task, first segment slot}
    DBuf<char> g_seg;                                   // their segment records (GapSegRec)
    DBuf<int32_t> g_slow, g_sslow;                      // work items of k_gap_finish / k_gap_seg left to the SLOW variants
    DBuf<int32_t> last_s0;                              // anchors of the last row's round-A tasks ([b], [32 + b])
    DBuf<uint8_t> o_spec, fin;                          // gap passes: flagged tiles; rows already final (per plane slot)
    int fin_stamp = 255;                                // ... iff the cell holds the current layer's stamp (run_layer; 255: cleared before the next layer)
    DBuf<int64_t> o_toffs, o_tilePS, o_tilePS2;
    DBuf<Best<TC, true>> o_part;
    int64_t max_tasks = 0;
    bool planes_full = false;                           // the last layer stored every per-block winner (cp_dp_block_tables)
    DBuf<int32_t> pz;                                   // poison mode: cells read that nobody wrote
    // windowed layers: geometry of the current call; anchors of the mirrored head tasks, cached per (pattern, w)
    Geo G{0, 0, 0};
    bool win_built = false; int64_t win_w = -1, win_nitems = 0, win_aoff[33];
    DBuf<int32_t> w_anch, w_anch2, w_tot, w_hist;
    DBuf<int32_t> leaf_anch, leaf_anch2;                // leaf pass of windowed layers: nets / self nets of [64 g + 62 - w, 64 g) per group
    DBuf<int64_t> w_E;
    DBuf<RoundCounts> rc;                               // per-round counters of the current layer (device)
    std::vector<RoundCounts> pred;                      // ... of the previous layer (host): sizes the next one
    bool pred_ok = false; int64_t pred_rlo = 0, pred_rhi = 0; double pred_scale = 1.0; int pred_win = 0; int64_t pred_w = 0;
    // rounds whose flattened stage served a handful of tasks in the last layer that ran it: their medium tasks get tiles of their
    // own from then on and the stage (six dependent launches, ~40 us) is skipped -- see run_layer
    std::vector<uint8_t> force_own;
    int64_t force_rows = 0;               // rows of the layer the flags come from (a one-row last layer says nothing about a full one)
    // (o_rec / loc are the size witnesses of their groups: they are released first and allocated LAST, so a hipMalloc failure in
    //  the middle leaves the witness empty and the next call allocates the whole group again)
    void ensure_own(size_t NT) {                        // per-tile arrays of the own-tiled tasks
        if (o_rec.n >= NT && o_rec.n > 0) return;
        size_t c = NT > 0 ? NT : 1;
        o_rec.release();
        o_task.alloc(c); o_tileS.alloc(c); o_tilePS.alloc(c + 1); o_part.alloc(c); o_hi.alloc(c); o_spec.alloc(c);
        o_sub.alloc(c * (SMAX + 1)); o_spv.alloc(c * (SMAX + 1));
        if (hyp) { o_tileS2.alloc(c); o_tilePS2.alloc(c + 1); }
        o_rec.alloc(c);
    }
    void ensure_flat(size_t T) {                        // per-step and per-tile arrays of the flattened tasks
        if (loc.n >= T && loc.n > 0) return;
        size_t c = T > 0 ? T : 1, nt = (c + LT - 1) / LT;
        loc.release();
        if (hyp) loc2.alloc(c);
        tileS.alloc(nt); tilePS.alloc(nt + 1); tile_t0.alloc(nt); partL.alloc(nt); partR.alloc(nt); taskR.alloc(nt);
        open_list.alloc(nt); fix_list.alloc(nt); tile_rec.alloc(nt);
        if (hyp) { tileS2.alloc(nt); tilePS2.alloc(nt + 1); }
        loc.alloc(c);
    }
};

// number of rows r <= x of the form (base << (b+1)) | (1 << b) | ((2v+1) << tau): the rows with ctz == tau whose bit b is set
static int64_t count_rows(int b, int tau, int64_t x)
{
    if (x < 0) return 0;
    int sh = b - tau - 1;
    int64_t V = (int64_t)1 << sh;
    int64_t nfull = (x + ((int64_t)1 << tau)) >> (b + 1);
    int64_t cnt = nfull * V;
    int64_t y = x - (nfull << (b + 1)) - ((int64_t)1 << b);
    if (y >= ((int64_t)1 << tau)) cnt += ((y >> tau) + 1) >> 1;
    return cnt;
}

// Rows computed by a run restricted to the row tile [rlo, rhi]: every row r with
//     r - 2^ctz(r) <= rhi  and  r + 2^ctz(r) >= rlo
// -- the tile plus the O(log n) tree ancestors its rows take their candidate bounds from (closed under ancestors).
static void make_round(RoundDesc &R, bool isA, int tau, int nbits, int64_t n, int64_t rlo, int64_t rhi, const Geo &G = Geo{0, 0, 0},
                       const int64_t *aoff = nullptr)
{
    memset(&R, 0, sizeof(R));
    R.isA = isA; R.tau = tau; R.nbits = nbits; R.n = n; R.G = G;
    if (G.win) {
        // every row takes part in every plane b <= s.  Round A: plane b holds the heads of its rectangles, the multiples of 2^b
        // (index l <-> row (l + 1) << b); round tau: every plane b in (tau, s] holds the rows with ctz == tau (index l <-> row
        // (2 l + 1) << tau).  A row tile [rlo, rhi] needs the rows within 2^tau of it in round tau (their tree neighbours are then
        // computed by the earlier rounds) and the heads within 2^(b+1) of it in plane b.
        if (aoff) for (int b = 0; b < 32; b++) R.aoff[b] = aoff[b];
        int64_t acc = 0;
        for (int b = 0; b < 36; b++) {
            R.tbase[b] = acc;
            if (b > G.s || (!isA && b <= tau)) continue;
            if (isA) {
                int64_t lo = rlo - ((int64_t)2 << b), hi = rhi + ((int64_t)1 << b);
                if (lo < ((int64_t)1 << b)) lo = (int64_t)1 << b;
                if (hi > n) hi = n;
                int64_t l0 = ((lo + ((int64_t)1 << b) - 1) >> b) - 1, l1 = (hi >> b) - 1;      // first / last head index
                R.tskip[b] = l0;
                acc += l1 >= l0 ? l1 - l0 + 1 : 0;
            } else {
                int64_t rmin = rlo - ((int64_t)1 << tau), rmax = rhi + ((int64_t)1 << tau);
                if (rmax > n) rmax = n;
                int64_t u0 = rmin <= 0 ? 0 : (((rmin - 1) >> tau) + 1) >> 1, u1 = ((rmax >> tau) + 1) >> 1;
                R.tskip[b] = u0;
                acc += u1 > u0 ? u1 - u0 : 0;
            }
        }
        R.ntask = acc;
        return;
    }
    if (isA) {
        int64_t r0 = rlo > 1 ? rlo : 1, r1 = rhi < n ? rhi : n;
        R.a_r0 = r0; R.a_nmain = r1 >= r0 ? r1 - r0 + 1 : 0;
        R.nextra = 0;
        for (int b = 0; b < nbits; b++) {            // rows with ctz == b just outside the tile
            int64_t step = (int64_t)1 << b;
            // left of the tile: r in [rlo - 2^b, rlo) ; right: r in (rhi, rhi + 2^b]
            for (int side = 0; side < 2; side++) {
                int64_t lo = side == 0 ? rlo - step : rhi + 1, hi = side == 0 ? rlo - 1 : rhi + step;
                if (lo < 1) lo = 1;
                if (hi > n) hi = n;
                for (int64_t r = ((lo + step - 1) >> b) << b; r <= hi; r += step)
                    if (r >= 1 && ((r >> b) & 1) && R.nextra < 64) R.extra[R.nextra++] = r;      // ctz(r) == b exactly
            }
        }
        R.nlast = 0;
        if (n >= 1) {
            int lowest = 0;
            while (!((n >> lowest) & 1)) lowest++;
            for (int b = lowest + 1; b < nbits; b++) if ((n >> b) & 1) R.last_b[R.nlast++] = b;
        }
        R.ntask = R.a_nmain + R.nextra + R.nlast;
        return;
    }
    int64_t rmin = rlo - ((int64_t)1 << tau), rmax = rhi + ((int64_t)1 << tau);
    if (rmax > n) rmax = n;
    int64_t acc = 0;
    for (int b = 0; b <= tau; b++) R.tbase[b] = 0;
    for (int b = tau + 1; b < nbits; b++) {
        R.tbase[b] = acc;
        int64_t skip = count_rows(b, tau, rmin - 1);
        int64_t cnt = count_rows(b, tau, rmax) - skip;
        R.tskip[b] = skip;
        acc += cnt > 0 ? cnt : 0;
    }
    for (int b = nbits; b < 36; b++) R.tbase[b] = acc;
    R.ntask = acc;
}

// right part of one round for one counter: rows with ctz == tau stream their 2^tau columns once for all bit planes
static void launch_rpass(hipStream_t s, const RoundDesc &R, int nbits, int64_t n, int64_t rlo, int64_t rhi, const int64_t *cpos,
                         const int32_t *link, int ge, const int32_t *opt, int32_t *cr, double mean_deg)
{
    const int allp = R.G.win;
    int tau = R.tau;
    int64_t rmin = rlo - ((int64_t)1 << tau), rmax = rhi + ((int64_t)1 << tau);
    if (rmax > n) rmax = n;
    if (rmin < 0) rmin = 0;
    // rows (2u+1)<<tau in [rmin, rmax]
    int64_t u0 = rmin <= 0 ? 0 : (((rmin - 1) >> tau) + 1) >> 1;       // #rows < rmin
    int64_t u1 = ((rmax >> tau) + 1) >> 1;                              // #rows <= rmax
    int64_t nrows = u1 - u0;
    if (nrows <= 0) return;
    if (tau <= g_opt_rpass_small_tau) {
        // lane-private share of a row: rpass_cap per cent of the mean number of entries of 2^tau columns, in steps of 16
        int64_t cap = (int64_t)(0.01 * (double)g_opt_rpass_cap * mean_deg * (double)((int64_t)1 << tau));
        cap = std::max<int64_t>(16, std::min<int64_t>((cap + 15) / 16 * 16, 1 << 20));
#define RPS(GE, NB) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rpass_small<GE, NB>), dim3((unsigned)cdiv(nrows, 256)), dim3(256), 0, s, tau, nbits, n, u0, nrows, cpos, link, \
                                       opt, cr, allp, (int)cap)
        // (up to 24 bit planes -- n < 2^24 -- the per-plane registers of the kernels are a quarter fewer: one more wave per SIMD)
        if (nbits <= 24) { if (ge) RPS(true, 24); else RPS(false, 24); }
        else             { if (ge) RPS(true, NBMAX); else RPS(false, NBMAX); }
#undef RPS
    } else {
        int ch_cols = (int)g_opt_rpass_ch;          // columns per wave (cp_set_option("rpass_ch"))
        int64_t cpr = ((int64_t)1 << tau) > ch_cols ? (((int64_t)1 << tau) / ch_cols) : 1;
        if (cpr == 1) ch_cols = 1 << tau;
        int64_t waves = nrows * cpr;
#define RPW(GE, WPB) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rpass_wave<GE, WPB>), dim3((unsigned)cdiv(waves, WPB)), dim3(64 * WPB), 0, s, tau, nbits, n, u0, \
                                        nrows, (int)cpr, ch_cols, cpos, link, opt, cr, allp)
        // the chunks of a row that share a workgroup add up in LDS before the row's counters see them
        if (cpr % 16 == 0)     { if (ge) RPW(true, 16); else RPW(false, 16); }
        else if (cpr % 4 == 0) { if (ge) RPW(true, 4);  else RPW(false, 4); }
        else                   { if (ge) RPW(true, 1);  else RPW(false, 1); }
#undef RPW
    }
}

// builds the cached round-A counts (once per partition: they depend on the pattern only)
template <typename TC>
static void ra_build(cp_csr_s *A, LayerWork<TC> &Wk)
{
    hipStream_t s = A->stream;
    const int64_t n = A->n;
    const bool hyp = Wk.hyp;
    RATab &T = Wk.ra_tab;
    memset(&T, 0, sizeof(T));
    T.nbits = Wk.nbits; T.n = n;
    for (int b = 0; b < 33; b++) { T.tlo[b] = 0; T.thi[b] = INT64_MAX; }
    int64_t tb = 0, rb = 0;
    for (int b = 0; b < 33; b++) {
        T.tbase[b] = tb; T.rbase[b] = rb;
        if (b >= Wk.nbits) continue;
        int64_t nrows = ((n >> b) + 1) >> 1;
        tb += cdiv(nrows << b, LT);
        if (b >= 9) rb += nrows;
    }
    Wk.ra_ntile = tb; Wk.ra_nrow = rb;
    const size_t total = (size_t)tb * LT;
    Wk.ra_c.ensure(total + 8);
    if (hyp) Wk.ra_c2.ensure(total + 8);
    Wk.ra_part.ensure((size_t)std::max<int64_t>(1, tb - T.tbase[9]));
    if (tb <= 0) { Wk.ra_built = true; return; }
    DBuf<int64_t> &G = Wk.ra_G, &scratch = Wk.scratch;        // (kept with the layer scratch: 8 B per element)
    G.ensure(total + 1);
    hipLaunchKernelGGL(k_ra_colcount, dim3((unsigned)tb), dim3(LT), 0, s, T, A->pos32.p, A->next.p, hyp ? A->fpos32.p : (const int32_t *)nullptr,
                       hyp ? A->flast.p : (const int32_t *)nullptr, Wk.ra_c.p, hyp ? Wk.ra_c2.p : (int32_t *)nullptr);
    exclusive_scan_i32(Wk.ra_c.p, G.p, (int64_t)total, scratch, s);
    hipLaunchKernelGGL(k_ra_final, dim3((unsigned)tb), dim3(LT), 0, s, T, G.p, Wk.ra_c.p, (const int32_t *)nullptr);
    if (hyp) {
        exclusive_scan_i32(Wk.ra_c2.p, G.p, (int64_t)total, scratch, s);
        hipLaunchKernelGGL(k_ra_final, dim3((unsigned)tb), dim3(LT), 0, s, T, G.p, Wk.ra_c2.p, (const int32_t *)nullptr);
    }
    CP_HIP(hipGetLastError());
    Wk.ra_built = true;
}

// anchors of the mirrored head tasks of the windowed layers (k_win_*): once per (pattern, w)
template <typename TC>
static void win_build(cp_csr_s *A, LayerWork<TC> &Wk)
{
    hipStream_t s = A->stream;
    const int64_t n = A->n, w = Wk.G.w;
    const int sb = Wk.G.s;
    int64_t acc = 0;
    for (int b = 0; b < 33; b++) { Wk.win_aoff[b] = acc; if (b < sb) acc += (n >> (b + 1)) + 1; }      // heads of plane b: rho = idx << (b+1), idx <= n >> (b+1)
    Wk.win_nitems = acc;
    const size_t ni = (size_t)(acc > 0 ? acc : 1);
    Wk.w_anch.ensure(ni); Wk.w_tot.ensure(ni); Wk.w_hist.ensure((size_t)n + 2); Wk.w_E.ensure((size_t)n + 2);
    if (Wk.hyp) Wk.w_anch2.ensure(ni);
    Wk.win_built = false;
    if (acc > 0) {
        RoundDesc R;
        make_round(R, true, 0, sb + 1, n, 0, n, Wk.G, Wk.win_aoff);
        const unsigned wg = (unsigned)std::min<int64_t>(cdiv(acc, 4), 65536);
        for (int pass = 0; pass < (Wk.hyp ? 2 : 1); pass++) {
            if (pass == 1 || A->N == 0) CP_HIP(hipMemsetAsync(Wk.w_hist.p, 0, sizeof(int32_t) * (size_t)(n + 1), s));
            if (pass == 0) { if (A->N > 0) hipLaunchKernelGGL(k_win_hist, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, n, w, A->pos32.p, A->prev.p, A->next.p, Wk.w_hist.p); }
            else if (A->m > 0) hipLaunchKernelGGL(k_win_hist_rows, dim3((unsigned)cdiv(A->m, 256)), dim3(256), 0, s, A->m, n, w, A->rfirst.p, A->rlast.p, Wk.w_hist.p);
            exclusive_scan_i32(Wk.w_hist.p, Wk.w_E.p, n + 1, Wk.scratch, s);
            if (pass == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_win_block_totals<true>), dim3(wg), dim3(256), 0, s, R, acc, A->pos32.p, A->next.p, Wk.w_tot.p);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_win_block_totals<false>), dim3(wg), dim3(256), 0, s, R, acc, A->fpos32.p, A->flast.p, Wk.w_tot.p);
            hipLaunchKernelGGL(k_win_anchors, dim3((unsigned)cdiv(acc, 256)), dim3(256), 0, s, R, acc, pass == 0 ? A->pos.p : A->lpos.p, Wk.w_E.p, Wk.w_tot.p,
                               pass == 0 ? Wk.w_anch.p : Wk.w_anch2.p);
        }
        CP_HIP(hipGetLastError());
    }
    // leaf pass (dp_leaf.inc): anchors of the inner mirrored blocks, nets(64 g + 62 - w, 64 g) for every group g -- the same
    // histogram with the width w - 62
    Wk.leaf_anch.release(); Wk.leaf_anch2.release();
    if (sb >= LEAF_T && w > 62) {
        const int64_t ng = (n >> LEAF_T) + 1;
        Wk.leaf_anch.alloc((size_t)ng);
        if (Wk.hyp) Wk.leaf_anch2.alloc((size_t)ng);
        for (int pass = 0; pass < (Wk.hyp ? 2 : 1); pass++) {
            if (pass == 1 || A->N == 0) CP_HIP(hipMemsetAsync(Wk.w_hist.p, 0, sizeof(int32_t) * (size_t)(n + 1), s));
            if (pass == 0) { if (A->N > 0) hipLaunchKernelGGL(k_win_hist, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, n, w - 62, A->pos32.p, A->prev.p, A->next.p, Wk.w_hist.p); }
            else if (A->m > 0) hipLaunchKernelGGL(k_win_hist_rows, dim3((unsigned)cdiv(A->m, 256)), dim3(256), 0, s, A->m, n, w - 62, A->rfirst.p, A->rlast.p, Wk.w_hist.p);
            exclusive_scan_i32(Wk.w_hist.p, Wk.w_E.p, n + 1, Wk.scratch, s);
            hipLaunchKernelGGL(k_leaf_anchors, dim3((unsigned)cdiv(ng, 256)), dim3(256), 0, s, ng, n, pass == 0 ? A->pos.p : A->lpos.p, Wk.w_E.p,
                               pass == 0 ? Wk.leaf_anch.p : Wk.leaf_anch2.p);
        }
        CP_HIP(hipGetLastError());
    }
    Wk.win_built = true; Wk.win_w = w;
}

// the gap round's finishing kernels: fast variants first, then the SLOW ones over what they listed
template <typename TC, bool HYP>
static void launch_gap(hipStream_t s, cp_csr_s *A, LayerWork<TC> &Wk, int tau, int nchunk, int gnr, RoundCounts *rc, int64_t n, const TC *W,
                       const DevModel<TC> &M, TC alpha, unsigned gg, unsigned gs_grid, unsigned gm, unsigned gslow_grid)
{
    GapCtx<TC, HYP> C{reinterpret_cast<const Best<TC, HYP> *>(Wk.o_part.p), reinterpret_cast<const Best<TC, HYP> *>(Wk.o_sub.p), Wk.o_spv.p, Wk.o_spec.p,
                      Wk.o_tilePS.p, HYP ? Wk.o_tilePS2.p : nullptr, A->pos32.p, A->next.p, HYP ? A->fpos32.p : nullptr, HYP ? A->flast.p : nullptr, W};
    auto *gs = reinterpret_cast<GapSegRec<TC, HYP> *>(Wk.g_seg.p);
    const int32_t *lp = HYP ? A->lpos32.p : nullptr, *lf = HYP ? A->lfirst.p : nullptr;
    int32_t *nl = HYP ? Wk.nlopt.p : nullptr;
#define GF_ARGS tau, nchunk, rc, n, Wk.o_toffs.p, C, Wk.o_tdesc.p, Wk.o_tb.p, Wk.o_rlen.p, A->prev.p, lp, lf, M, alpha, Wk.opt.p, Wk.nnopt.p, nl, Wk.fin.p, \
                Wk.g_list.p, Wk.g_slot.p, Wk.fin_stamp, Wk.g_slow.p
    if (gnr == 2) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gap_finish<TC, HYP, 2, false>), dim3(gg), dim3(256), 0, s, GF_ARGS);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gap_finish<TC, HYP, 2, true>), dim3(gslow_grid), dim3(256), 0, s, GF_ARGS);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gap_finish<TC, HYP, 1, false>), dim3(gg), dim3(256), 0, s, GF_ARGS);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gap_finish<TC, HYP, 1, true>), dim3(gslow_grid), dim3(256), 0, s, GF_ARGS);
    }
#undef GF_ARGS
#define GS_ARGS tau, nchunk, rc, n, Wk.o_toffs.p, C, Wk.o_tdesc.p, Wk.o_tb.p, Wk.o_rlen.p, M, alpha, Wk.g_slot.p, gs, Wk.g_sslow.p
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gap_seg<TC, HYP, false>), dim3(gs_grid), dim3(256), 0, s, GS_ARGS);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gap_seg<TC, HYP, true>), dim3(gslow_grid), dim3(256), 0, s, GS_ARGS);
#undef GS_ARGS
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gap_merge<TC, HYP>), dim3(gm), dim3(256), 0, s, tau, nchunk, rc, n, Wk.o_toffs.p, Wk.o_tdesc.p, Wk.o_tb.p, A->pos32.p, A->prev.p,
                       lp, lf, M, Wk.g_list.p, gs, Wk.opt.p, Wk.nnopt.p, nl, Wk.fin.p, Wk.fin_stamp);
}

constexpr int64_t LB_MAX = 1 << 20;      // largest scan (elements) done in a single launch

// the same for the mirrored heads of the windowed geometry (once per pattern and window width; after win_build: the counts carry
// the heads' anchors)
template <typename TC>
static void ra_build_mir(cp_csr_s *A, LayerWork<TC> &Wk)
{
    hipStream_t s = A->stream;
    const int64_t n = A->n;
    const bool hyp = Wk.hyp;
    const Geo G = Wk.G;
    RATab &T = Wk.mir_tab;
    memset(&T, 0, sizeof(T));
    T.nbits = G.s; T.n = n; T.mir_w = G.w;
    for (int b = 0; b < 33; b++) { T.tlo[b] = 0; T.thi[b] = INT64_MAX; T.aoff[b] = b < 32 ? Wk.win_aoff[b] : 0; }
    int64_t tb = 0, rb = 0;
    for (int b = 0; b < 33; b++) {
        T.tbase[b] = tb; T.rbase[b] = rb;
        if (b >= G.s) continue;
        const int64_t nrows = n >> (b + 1);                  // rows (u + 1) 2^(b+1) <= n
        tb += cdiv(nrows << b, LT);
        if (b >= 9) rb += nrows;
    }
    Wk.mir_ntile = tb; Wk.mir_nrow = rb;
    const size_t total = (size_t)tb * LT;
    Wk.mir_c.ensure(total + 8);
    if (hyp) Wk.mir_c2.ensure(total + 8);
    Wk.mir_part.ensure((size_t)std::max<int64_t>(1, tb - T.tbase[9]));
    Wk.mir_built = true; Wk.mir_w = G.w;
    if (tb <= 0) return;
    DBuf<int64_t> &Gs = Wk.ra_G, &scratch = Wk.scratch;
    Gs.ensure(total + 1);
    hipLaunchKernelGGL(k_ra_colcount, dim3((unsigned)tb), dim3(LT), 0, s, T, A->pos32.p, A->next.p, hyp ? A->fpos32.p : (const int32_t *)nullptr,
                       hyp ? A->flast.p : (const int32_t *)nullptr, Wk.mir_c.p, hyp ? Wk.mir_c2.p : (int32_t *)nullptr);
    exclusive_scan_i32(Wk.mir_c.p, Gs.p, (int64_t)total, scratch, s);
    hipLaunchKernelGGL(k_ra_final, dim3((unsigned)tb), dim3(LT), 0, s, T, Gs.p, Wk.mir_c.p, (const int32_t *)Wk.w_anch.p);
    if (hyp) {
        exclusive_scan_i32(Wk.mir_c2.p, Gs.p, (int64_t)total, scratch, s);
        hipLaunchKernelGGL(k_ra_final, dim3((unsigned)tb), dim3(LT), 0, s, T, Gs.p, Wk.mir_c2.p, (const int32_t *)Wk.w_anch2.p);
    }
    CP_HIP(hipGetLastError());
}

// Runs the rounds of one layer.  `spec`: the per-round counts of the PREVIOUS layer (Wk.pred) size the grids, the buffers and
// decide which stages are launched; every kernel reads its true loop bounds from the device (RoundCounts) and walks them with
// grid strides, so a wrong prediction costs time, never correctness -- except a stage skipped or a buffer too small, which
// run_layer reports (false) after the layer and the caller redoes the layer with spec = false: one host sync per round brings
// the exact counts back before the dependent launches (the only mode of the first layer).
template <typename TC>
static bool run_layer(cp_csr_s *A, const DevModel<TC> &M, TC alpha, const TC *W, TC *cst_out, int32_t *ptr_out, LayerWork<TC> &Wk,
                      int64_t rlo, int64_t rhi, bool spec, bool allow_force = true)
{
    hipStream_t s = A->stream;
    const int64_t n = A->n;
    const bool hyp = Wk.hyp;
    const Geo G = Wk.G;
    const int nbits = G.win ? G.s + 1 : Wk.nbits;          // windowed layers: planes 0 .. s
    const double avg_deg = n > 0 ? (double)A->N / (double)n : 0.0;
    const double self_deg = (hyp && n > 0) ? (double)A->nrows_nonempty / (double)n : 0.0;
    // bytes a step of the streaming kernels moves: the stepped column's link entries (4 B each; hyperedge costs: plus the rows starting
    // in it), its 32-bit column pointer (4 B), the candidate's previous-layer cost (8 B), and the tile's record + partial
    // (16 B + 16 B per 256 steps).  (Round 1 priced a step at 4 deg + 24 B -- 8-byte column pointers and a per-step descriptor share
    // the kernels do not read -- which put the achieved rate above the box's copy rate.)
    const double step_bytes = 4.0 * (avg_deg + self_deg) + 12.0 + 32.0 / (double)LT;
    const bool own_tiles = !(g_opt_dbg & 64);      // cp_set_option("dbg", 64): keep every long task in the flattened space
    const int NR = nbits + 1;
    CP_HIP(hipMemsetAsync(Wk.rc.p, 0, sizeof(RoundCounts) * (size_t)NR, s));
    const bool gaps = own_tiles && g_opt_gap_tau >= 0;
    // cp_set_option("poison", 1) (tests): every layer starts from planes full of an out-of-range column; the kernels that turn
    // plane cells into addresses count and clamp what they read of it.  The invariant behind the speculative layers -- "whatever a
    // layer that is NOT redone has read was written by that layer" -- then reads: hits > 0 implies the layer is flagged for a redo.
    int32_t *pzp = nullptr;
    if (g_opt_poison) {
        Wk.pz.ensure(2);
        pzp = Wk.pz.p;
        CP_HIP(hipMemsetAsync(Wk.pz.p, 0, sizeof(int32_t) * 2, s));
        CP_HIP(hipMemsetAsync(Wk.opt.p, 0x7F, Wk.opt.bytes(), s));
        CP_HIP(hipMemsetAsync(Wk.nnopt.p, 0x7F, Wk.nnopt.bytes(), s));
        if (hyp) CP_HIP(hipMemsetAsync(Wk.nlopt.p, 0x7F, Wk.nlopt.bytes(), s));
    }
    // the rounds tau < LEAF_T are one pass over groups of 64 rows (dp_leaf.inc); windowed layers: when the window spans a group
    // (s >= 6) and no per-block table is asked for
    const bool leaf = g_opt_leaf && (!G.win || (G.s >= LEAF_T && !g_opt_block_tables && Wk.leaf_anch.p));
    Wk.planes_full = !leaf || g_opt_block_tables;
    // `fin` flags hold the layer's stamp: no 240 MB clear per layer (every attempt of a layer -- a redo included -- takes a new stamp)
    if (++Wk.fin_stamp > 255 || Wk.fin_stamp <= 0) { CP_HIP(hipMemsetAsync(Wk.fin.p, 0, Wk.fin.bytes(), s)); Wk.fin_stamp = 1; }
    std::vector<RoundCounts> used((size_t)NR);          // what the host sized each round with
    memset(used.data(), 0, sizeof(RoundCounts) * (size_t)NR);
    struct Patch { size_t idx; int rd; int kind; };
    std::vector<Patch> patches;                          // profile records whose algorithmic bytes depend on the true counts
    auto note = [&](int rd, int kind, int slot) { if (prof_active(slot)) patches.push_back({g_prof_pending.size() - 1, rd, kind}); };
    auto grow = [](int64_t v) { return v + (v >> 2) + 1024; };     // head room over the prediction
    bool any_forced = false;

    for (int rd = 0; rd <= nbits; rd++) {
        RoundDesc R;
        if (rd == 0) make_round(R, true, 0, nbits, n, rlo, rhi, G, Wk.win_aoff);
        else make_round(R, false, nbits - rd, nbits, n, rlo, rhi, G, Wk.win_aoff);
        R.fin_stamp = Wk.fin_stamp;
        if (leaf && !R.isA && R.tau < LEAF_T) continue;
        if (rd == 0 && g_opt_ra_cache && rlo <= 1 && rhi >= n && n >= 1 && !G.win) {
            // a full layer: round A of the rows' lowest blocks from the cached counts; what is left of round A below is the
            // last row in its upper planes
            if (!Wk.ra_built) { ProfScope ps(PROF_LINKS, s, 0.0); ra_build<TC>(A, Wk); }
            ProfScope ps(PROF_RA, s, 16.0 * (double)Wk.ra_ntile * LT);
            const bool colmajor = !(g_opt_dbg & 262144);       // (dbg 262144: the row-major kernel k_ra_layer)
            const unsigned cgrid = (unsigned)cdiv(cdiv(n, LT), 4);
            const int ra_bmin = (leaf && !Wk.planes_full && !(g_opt_dbg & 33554432)) ? LEAF_T : 0;      // (dbg 33554432: all levels, as without the leaf pass)
            if (colmajor && hyp)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_cols<TC, true, false>), dim3(cgrid), dim3(256), 0, s, Wk.ra_tab, Wk.ra_c.p, Wk.ra_c2.p, A->pos32.p, W, M, alpha,
                                   Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, Wk.ra_part.p, (int64_t)0, cdiv(n, LT), ra_bmin);
            else if (colmajor)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_cols<TC, false, false>), dim3(cgrid), dim3(256), 0, s, Wk.ra_tab, Wk.ra_c.p, (const int32_t *)nullptr, A->pos32.p, W, M, alpha,
                                   Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr, reinterpret_cast<Best<TC, false> *>(Wk.ra_part.p), (int64_t)0, cdiv(n, LT), ra_bmin);
            if (hyp) {
                if (!colmajor)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_layer<TC, true>), dim3((unsigned)cdiv(Wk.ra_ntile, 4)), dim3(256), 0, s, Wk.ra_tab, Wk.ra_ntile, Wk.ra_c.p, Wk.ra_c2.p,
                                   A->pos32.p, W, M, alpha, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, Wk.ra_part.p);
                if (Wk.ra_nrow > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_merge<TC, true>), dim3((unsigned)cdiv(Wk.ra_nrow, 4)), dim3(256), 0, s, Wk.ra_tab, Wk.ra_nrow,
                                                       Wk.ra_part.p, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p);
            } else {
                auto *pa = reinterpret_cast<Best<TC, false> *>(Wk.ra_part.p);
                if (!colmajor)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_layer<TC, false>), dim3((unsigned)cdiv(Wk.ra_ntile, 4)), dim3(256), 0, s, Wk.ra_tab, Wk.ra_ntile, Wk.ra_c.p,
                                   (const int32_t *)nullptr, A->pos32.p, W, M, alpha, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr, pa);
                if (Wk.ra_nrow > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_merge<TC, false>), dim3((unsigned)cdiv(Wk.ra_nrow, 4)), dim3(256), 0, s, Wk.ra_tab, Wk.ra_nrow,
                                                       pa, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr);
            }
            CP_HIP(hipGetLastError());
            R.a_nmain = 0; R.ntask = R.nextra + R.nlast;
        }
        if (rd == 0 && g_opt_ra_cache && G.win && G.s >= 1 && n >= 1 && !(g_opt_dbg & 1048576)) {
            // Windowed layer: the STANDARD heads of round A (rows with ctz == b < s, block [r - 2^b, r)) are the unconstrained scheme's
            // round-A rows of the levels below s -- same blocks, same layer-independent counts: k_ra_cols computes them for all rows
            // from the cache (60 B per candidate) and the generic round keeps the mirrored and common heads only.  The heads the layer
            // reads are those make_round lists -- within 2^(b+1) of the row tile in plane b <= s: their blocks start at rlo - 3 * 2^s or later.
            if (!Wk.ra_built) { ProfScope ps(PROF_LINKS, s, 0.0); ra_build<TC>(A, Wk); }
            RATab T2 = Wk.ra_tab;
            T2.nbits = std::min<int32_t>(G.s, Wk.ra_tab.nbits);
            const int64_t nrow2 = T2.nbits > 9 ? Wk.ra_tab.rbase[T2.nbits] : 0;
            const int64_t c_lo = std::max<int64_t>(0, (rlo > 0 ? rlo : 0) - ((int64_t)4 << G.s)), c_hi = std::min<int64_t>(n, rhi);
            const int64_t tile0 = c_lo / LT, tile1 = cdiv(c_hi, LT);
            ProfScope ps(PROF_RA, s, 60.0 * (double)(tile1 - tile0) * LT);
            const unsigned cgrid = (unsigned)std::max<int64_t>(1, cdiv(tile1 - tile0, 4));
            // (the standard heads below LEAF_T are rows inside a leaf group: the leaf pass computes them; the MIRRORED heads of those
            //  levels include the multiples of 64 -- every row has a task in every plane -- and stay)
            const int ra_bmin2 = (leaf && !Wk.planes_full && !(g_opt_dbg & 33554432)) ? LEAF_T : 0;
            if (hyp) {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_cols<TC, true, false>), dim3(cgrid), dim3(256), 0, s, T2, Wk.ra_c.p, Wk.ra_c2.p, A->pos32.p, W, M, alpha,
                                   Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, Wk.ra_part.p, tile0, tile1, ra_bmin2);
                if (nrow2 > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_merge<TC, true>), dim3((unsigned)cdiv(nrow2, 4)), dim3(256), 0, s, T2, nrow2,
                                                  Wk.ra_part.p, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p);
            } else {
                auto *pa = reinterpret_cast<Best<TC, false> *>(Wk.ra_part.p);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_cols<TC, false, false>), dim3(cgrid), dim3(256), 0, s, T2, Wk.ra_c.p, (const int32_t *)nullptr, A->pos32.p, W, M, alpha,
                                   Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr, pa, tile0, tile1, ra_bmin2);
                if (nrow2 > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_merge<TC, false>), dim3((unsigned)cdiv(nrow2, 4)), dim3(256), 0, s, T2, nrow2,
                                                  pa, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr);
            }
            CP_HIP(hipGetLastError());
            R.skip_std = 1;
            if (!(g_opt_dbg & 2097152)) {
                // the MIRRORED heads the same way, from their own table (row-major kernel; per level the tiles of the rows near the window)
                if (!Wk.mir_built || Wk.mir_w != G.w) { ProfScope ps2(PROF_LINKS, s, 0.0); ra_build_mir<TC>(A, Wk); }
                if (Wk.mir_ntile > 0) {
                    RATab T3 = Wk.mir_tab;
                    const int64_t r_lo = std::max<int64_t>(1, (rlo > 0 ? rlo : 0) - ((int64_t)4 << G.s)), r_hi = std::min<int64_t>(n, rhi);
                    double elems = 0;
                    for (int b = 0; b < T3.nbits; b++) {
                        const int64_t u_lo = std::max<int64_t>(0, (r_lo >> (b + 1)) - 1), u_hi = r_hi >> (b + 1);        // rows u_lo .. u_hi - 1
                        T3.tlo[b] = T3.tbase[b] + ((u_lo << b) / LT);
                        T3.thi[b] = std::min<int64_t>(T3.tbase[b + 1], T3.tbase[b] + cdiv(std::max<int64_t>(u_hi, 0) << b, LT));
                        elems += (double)std::max<int64_t>(0, T3.thi[b] - T3.tlo[b]) * LT;
                    }
                    // column-major like the standard heads (z = p + w + 1: the blocks of the rows r_lo .. r_hi lie in [r_lo, r_hi + 2^s)); the
                    // row-major kernel (dbg 8388608) reads 16 B per (candidate, level) instead of 4
                    const bool mcols = !(g_opt_dbg & 8388608);
                    const int64_t zt0 = r_lo / LT, zt1 = cdiv(std::min<int64_t>(n + G.w + 1, r_hi + ((int64_t)1 << G.s) + 1), LT);
                    ProfScope ps2(PROF_RA, s, mcols ? 80.0 * (double)(zt1 - zt0) * LT : 16.0 * elems);
                    const unsigned mgrid = (unsigned)std::max<int64_t>(1, cdiv(zt1 - zt0, 4));
                    if (hyp) {
                        if (mcols) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_cols<TC, true, true>), dim3(mgrid), dim3(256), 0, s, T3, Wk.mir_c.p, Wk.mir_c2.p, A->pos32.p, W, M, alpha,
                                                      Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, Wk.mir_part.p, zt0, zt1, 0);
                        else
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_layer<TC, true>), dim3((unsigned)cdiv(Wk.mir_ntile, 4)), dim3(256), 0, s, T3, Wk.mir_ntile, Wk.mir_c.p, Wk.mir_c2.p,
                                           A->pos32.p, W, M, alpha, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, Wk.mir_part.p);
                        if (Wk.mir_nrow > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_merge<TC, true>), dim3((unsigned)cdiv(Wk.mir_nrow, 4)), dim3(256), 0, s, T3, Wk.mir_nrow,
                                                                Wk.mir_part.p, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p);
                    } else {
                        auto *pm = reinterpret_cast<Best<TC, false> *>(Wk.mir_part.p);
                        if (mcols) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_cols<TC, false, true>), dim3(mgrid), dim3(256), 0, s, T3, Wk.mir_c.p, (const int32_t *)nullptr, A->pos32.p, W, M,
                                                      alpha, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr, pm, zt0, zt1, 0);
                        else
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_layer<TC, false>), dim3((unsigned)cdiv(Wk.mir_ntile, 4)), dim3(256), 0, s, T3, Wk.mir_ntile, Wk.mir_c.p,
                                           (const int32_t *)nullptr, A->pos32.p, W, M, alpha, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr, pm);
                        if (Wk.mir_nrow > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ra_merge<TC, false>), dim3((unsigned)cdiv(Wk.mir_nrow, 4)), dim3(256), 0, s, T3, Wk.mir_nrow,
                                                                pm, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr);
                    }
                    CP_HIP(hipGetLastError());
                    R.skip_mir = 1;
                }
            }
        }
        if (R.ntask <= 0) continue;
        CP_REQUIRE(R.ntask <= Wk.max_tasks, CP_EINTERNAL, "a DP round has more tasks than the task buffers hold");
        RoundCounts *rc = Wk.rc.p + rd;
        const bool gap = gaps && !R.isA && R.tau <= g_opt_gap_tau;       // long tasks of this round finish all the rows of their gap
        // In the rounds above the gap rounds the flattened stage (tile table, stream, two scans, span / open / fix: six dependent
        // launches, ~40 us) typically serves three or four medium tasks.  Where the last layer that ran it saw at most force_max (1024), every task
        // that is not finished in setup gets tiles of its own instead, and the stage disappears from the round.
        const bool forced = allow_force && own_tiles && !gap && !R.isA && (size_t)rd < Wk.force_own.size() && Wk.force_own[(size_t)rd] &&
                            (rhi - rlo + 1) <= 2 * Wk.force_rows && !(g_opt_dbg & 524288);
        any_forced |= forced;
        if (!R.isA) {
            int64_t cols = (((n >> R.tau) + 1) >> 1) << R.tau;
            ProfScope ps(PROF_RPASS, s, 4.0 * (avg_deg + self_deg) * (double)cols + 8.0 * (double)R.ntask);
            if (((int64_t)1 << R.tau) > g_opt_rpass_ch)   // several chunks per row accumulate with atomics: clear first
                hipLaunchKernelGGL(k_setup, dim3((unsigned)cdiv(R.ntask, 256)), dim3(256), 0, s, R, Wk.opt.p, Wk.nnopt.p, Wk.cr.p, 1,
                                   Wk.tdesc.p, A->pos32.p, Wk.tb.p, Wk.len.p, (const int32_t *)nullptr, hyp ? Wk.crl.p : (int32_t *)nullptr,
                                   (int32_t *)nullptr);
            launch_rpass(s, R, nbits, n, rlo, rhi, A->pos.p, A->prev.p, 0, Wk.opt.p, Wk.cr.p, avg_deg);                   // prev[q] < B
            if (hyp) launch_rpass(s, R, nbits, n, rlo, rhi, A->lpos.p, A->lfirst.p, 1, Wk.opt.p, Wk.crl.p, self_deg);      // rows ending in the column with first >= B
        }
        if (R.isA && R.nlast > 0) {
            CP_HIP(hipMemsetAsync(Wk.last_s0.p, 0, Wk.last_s0.bytes(), s));
            hipLaunchKernelGGL(k_last_row_counts, dim3(1024), dim3(256), 0, s, R, A->pos32.p, A->prev.p, hyp ? A->lpos32.p : (const int32_t *)nullptr,
                               hyp ? A->lfirst.p : (const int32_t *)nullptr, Wk.last_s0.p, Wk.fin.p);
        }
        {
            // per task: four gathers from the plane arrays + the record; short tasks also step over their columns here
            ProfScope ps(PROF_SETUP, s, 29.0 * (double)R.ntask);
#define SS_ARGS R, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, Wk.cr.p, Wk.crl.p, A->pos32.p, A->next.p, hyp ? A->fpos32.p : (const int32_t *)nullptr,   \
                hyp ? A->flast.p : (const int32_t *)nullptr, W, M, alpha, Wk.tdesc.p, Wk.tb.p, Wk.len.p, Wk.tS0l.p, &rc->nlong,               \
                (int32_t)g_opt_short_t, (int32_t)g_opt_short_e, own_tiles ? Wk.o_tdesc.p : (int4 *)nullptr, Wk.o_tb.p, Wk.o_rlen.p, Wk.o_ntl.p,            \
                Wk.o_tS0l.p, &rc->nown, &rc->own_steps, (int32_t)(forced ? 2 : gap ? g_opt_gap_min : g_opt_own_min),                                         \
                (int32_t)std::min<size_t>(Wk.o_ntl.n, (size_t)INT32_MAX), &rc->err, Wk.fin.p, Wk.last_s0.p, (g_opt_dbg & 4096) ? &rc->_pad : (int32_t *)nullptr, Wk.w_anch.p, Wk.w_anch2.p, pzp
            const int sbs = (int)g_opt_setup_bs;          // lanes per block: one list atomic per block, but the block's waves meet at two barriers
            dim3 sgrid((unsigned)cdiv(R.ntask, sbs));
            if (!R.isA) {                                // one grid row per bit plane above tau
                int64_t mx = 1;
                for (int bb = R.tau + 1; bb < nbits; bb++) mx = std::max<int64_t>(mx, R.tbase[bb + 1] - R.tbase[bb]);
                sgrid = dim3((unsigned)cdiv(mx, sbs), (unsigned)std::max(1, nbits - R.tau - 1));
            }
            if (hyp) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_setup_short<TC, true>), sgrid, dim3(sbs), 0, s, SS_ARGS);
            else     hipLaunchKernelGGL(HIP_KERNEL_NAME(k_setup_short<TC, false>), sgrid, dim3(sbs), 0, s, SS_ARGS);
#undef SS_ARGS
        }
        RoundCounts P;                                   // the counts this round is sized with
        if (spec) {
            P = Wk.pred[(size_t)rd];
            if (Wk.pred_scale != 1.0) {              // (windowed layers: the previous layer's tile had another number of rows)
                auto sc = [&](int64_t v) { return (int64_t)((double)v * Wk.pred_scale) + (v > 0 ? 1 : 0); };
                P.nlong = (int32_t)sc(P.nlong); P.nown = (int32_t)sc(P.nown); P.T = sc(P.T); P.NT = sc(P.NT);
                P.own_steps = (unsigned long long)sc((int64_t)P.own_steps);
            }
            if ((g_opt_dbg & 1024) && (rd & 1)) { P.nown = 0; P.NT = 0; P.nlong = 0; P.T = 0; }      // test: a prediction that skips stages with work
            // buffers from the prediction (grown only here; k_round_finish checks the true totals against them)
            Wk.ensure_own((size_t)grow(P.NT));
            Wk.ensure_flat((size_t)grow(P.T));
        }
        {
            // both scans read their element count on the device
            ProfScope ps(PROF_SCAN, s, 12.0 * (double)R.ntask);
            // (single-launch scans take their block index from one counter: fine for a few hundred blocks, not for 27 000)
            if (R.ntask <= LB_MAX) exclusive_scan_i32_lb(Wk.len.p, Wk.offs.p, &rc->nlong, R.ntask, &rc->T, Wk.scanws, s);
            else exclusive_scan_i32_devn(Wk.len.p, Wk.offs.p, &rc->nlong, R.ntask, &rc->T, Wk.scratch, s);
            if (own_tiles)
                {
                    const int64_t nm = std::min<int64_t>(R.ntask, (int64_t)Wk.o_ntl.n);
                    if (nm <= LB_MAX) exclusive_scan_i32_lb(Wk.o_ntl.p, Wk.o_toffs.p, &rc->nown, nm, &rc->NT, Wk.scanws, s);
                    else exclusive_scan_i32_devn(Wk.o_ntl.p, Wk.o_toffs.p, &rc->nown, nm, &rc->NT, Wk.scratch, s);
                }
            const bool tiny = spec && (g_opt_dbg & 2048);       // test: pretend the buffers sized from the prediction are too small
            hipLaunchKernelGGL(k_round_finish, dim3(1), dim3(1), 0, s, rc, spec ? (tiny ? (int64_t)64 : (int64_t)Wk.loc.n) : INT64_MAX,
                               spec ? (tiny ? (int64_t)1 : (int64_t)Wk.o_rec.n) : INT64_MAX);
        }
        if (!spec) {
            CP_HIP(hipMemcpyAsync(&P, rc, sizeof(P), hipMemcpyDeviceToHost, s));
            CP_HIP(hipStreamSynchronize(s));
            Wk.ensure_own((size_t)P.NT);
            Wk.ensure_flat((size_t)P.T);
        }
        used[(size_t)rd] = P;
        if (g_opt_dbg & 8) fprintf(stderr, "round isA=%d tau=%d ntask=%lld %s long=%d own=%d T=%lld NT=%lld nontrivial=%d\n", R.isA, R.tau, (long long)R.ntask,
                                   spec ? "predicted" : "exact", P.nlong, P.nown, (long long)P.T, (long long)P.NT, P._pad);
        if (P.nown > 0 && P.NT > 0) {
            // ---- long tasks with tiles of their own: map, stream + evaluate, merge
            const int64_t gNT = spec ? (int64_t)Wk.o_rec.n : P.NT, gown = spec ? grow(P.nown) : P.nown;      // (capacity >= the true NT, checked)
            if (g_opt_dbg & 128) {                // poison what this block must write before it reads
                int pat = (g_opt_dbg & 256) ? 0x00 : 0x7F;
                CP_HIP(hipMemsetAsync(Wk.o_part.p, pat, Wk.o_part.bytes(), s));
                CP_HIP(hipMemsetAsync(Wk.o_tileS.p, pat, Wk.o_tileS.bytes(), s));
                CP_HIP(hipMemsetAsync(Wk.o_rec.p, pat, Wk.o_rec.bytes(), s));
                if (hyp) CP_HIP(hipMemsetAsync(Wk.o_tileS2.p, pat, Wk.o_tileS2.bytes(), s));
            }
            hipLaunchKernelGGL(k_own_map, dim3((unsigned)std::min<int64_t>(cdiv(gNT, 256), 8192)), dim3(256), 0, s, rc, Wk.o_toffs.p, Wk.o_tdesc.p, Wk.o_rlen.p, Wk.o_rec.p, Wk.o_task.p,
                               Wk.o_tb.p, gap ? Wk.o_hi.p : (int32_t *)nullptr, R.tau, n);
            {
                ProfScope ps(gap ? PROF_GAPSTREAM : PROF_OWN, s, (double)P.own_steps * step_bytes);      // same bytes per step as dp_lpass
#define LO_ARGS R.isA, rc, A->pos32.p, A->next.p, hyp ? A->fpos32.p : (const int32_t *)nullptr, hyp ? A->flast.p : (const int32_t *)nullptr,            \
                Wk.o_tileS.p, Wk.o_tileS2.p, Wk.o_rec.p, W, M, alpha
#define LO_TAIL R.tau, Wk.o_hi.p, Wk.o_spec.p, (int)((g_opt_dbg & 512) != 0)
                Best<TC, false> *sb0 = reinterpret_cast<Best<TC, false> *>(Wk.o_sub.p);
                unsigned og = (unsigned)cdiv(gNT, 4);
                Best<TC, false> *pp0 = reinterpret_cast<Best<TC, false> *>(Wk.o_part.p);
                if (hyp) { if (gap) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpass_own<TC, true, true>), dim3(og), dim3(256), 0, s, LO_ARGS, Wk.o_part.p, LO_TAIL, Wk.o_sub.p, Wk.o_spv.p, A->col.p, A->ffirst.p);
                           else     hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpass_own<TC, true, false>), dim3(og), dim3(256), 0, s, LO_ARGS, Wk.o_part.p, LO_TAIL, Wk.o_sub.p, Wk.o_spv.p, A->col.p, A->ffirst.p); }
                else     { if (gap) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpass_own<TC, false, true>), dim3(og), dim3(256), 0, s, LO_ARGS, pp0, LO_TAIL, sb0, Wk.o_spv.p, A->col.p, (const int32_t *)nullptr);
                           else     hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpass_own<TC, false, false>), dim3(og), dim3(256), 0, s, LO_ARGS, pp0, LO_TAIL, sb0, Wk.o_spv.p, A->col.p, (const int32_t *)nullptr); }
#undef LO_TAIL
#undef LO_ARGS
            }
            note(rd, 0, gap ? PROF_GAPSTREAM : PROF_OWN);
            {
                ProfScope ps(PROF_CARRY, s, 12.0 * (double)P.NT);
                const int32_t *ntp = reinterpret_cast<const int32_t *>(&rc->NT);       // (NT < 2^31: the low word)
                exclusive_scan_i32_lb(Wk.o_tileS.p, Wk.o_tilePS.p, ntp, (int64_t)Wk.o_tileS.n, nullptr, Wk.scanws, s);
                if (hyp) exclusive_scan_i32_lb(Wk.o_tileS2.p, Wk.o_tilePS2.p, ntp, (int64_t)Wk.o_tileS.n, nullptr, Wk.scanws, s);
            }
            if (gap) {
                // every row of the gaps gets its winner: one wave per (task, 64 rows); tasks of more than GAPSEG tiles in two steps
                ProfScope ps(PROF_GAP, s, 24.0 * (double)P.NT);
                const int nchunk = (int)std::max<int64_t>(1, ((int64_t)2 << R.tau) / 64);
                const size_t ncap = Wk.o_rec.n / GAPSEG + 2;
                Wk.g_list.ensure(ncap); Wk.g_slot.ensure(2 * ncap);
                Wk.g_seg.ensure((2 * ncap) * (size_t)nchunk * 64 * sizeof(GapSegRec<TC, true>));
                // items that meet a tile with more than SMAX specials are listed by the fast kernels and redone by the SLOW ones
                Wk.g_slow.ensure((size_t)gown * (size_t)nchunk + 64); Wk.g_sslow.ensure((2 * ncap) * (size_t)nchunk + 64);
                CP_REQUIRE((int64_t)gown * nchunk < INT32_MAX && (int64_t)(2 * ncap) * nchunk < INT32_MAX, CP_EINTERNAL, "gap work index beyond 32 bits");
                const int gnr = (g_opt_gap_nr >= 2 && nchunk >= 2) ? 2 : 1;       // 64-row chunks per wave of k_gap_finish
                unsigned gg = (unsigned)std::min<int64_t>(cdiv(gown * cdiv((int64_t)nchunk, gnr), 4), 16384);
                unsigned gs_grid = (unsigned)std::min<int64_t>(cdiv((int64_t)(2 * ncap) * nchunk, 4), 8192);
                unsigned gm = (unsigned)std::min<int64_t>(cdiv((int64_t)ncap * nchunk, 4), 4096);
                const unsigned gslow_grid = 2048;                               // (grid-stride over the device-side counts n_gslow / n_sslow: usually none)
                if (hyp) launch_gap<TC, true>(s, A, Wk, R.tau, nchunk, gnr, rc, n, W, M, alpha, gg, gs_grid, gm, gslow_grid);
                else launch_gap<TC, false>(s, A, Wk, R.tau, nchunk, gnr, rc, n, W, M, alpha, gg, gs_grid, gm, gslow_grid);
            } else {
                ProfScope ps(PROF_FIX, s, 24.0 * (double)P.NT);
                // one lane per task: single tiles are final already, short tasks are merged on the spot, the rest is listed
                const bool wide = true;                          // one block per listed task (the longest task sets the pace)
                unsigned lgrid = (unsigned)std::min<int64_t>(cdiv(gown, 256), 4096), wgrid = (unsigned)std::min<int64_t>(wide ? gown : cdiv(gown, 4), 8192);
#define FO_ARGS rc, Wk.o_toffs.p
#define FO_TAIL Wk.o_tdesc.p, Wk.o_tb.p
                if (hyp) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fix_own_lane<TC, true>), dim3(lgrid), dim3(256), 0, s, FO_ARGS, Wk.o_part.p, FO_TAIL, Wk.o_tS0l.p,
                                       Wk.o_tilePS.p, Wk.o_tilePS2.p, M, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, n + 1, Wk.o_wide.p);
                    if (wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fix_own<TC, true, 4>), dim3(wgrid), dim3(256), 0, s, FO_ARGS, Wk.o_part.p, FO_TAIL,
                                                 Wk.o_tS0l.p, Wk.o_tilePS.p, Wk.o_tilePS2.p, M, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, n + 1, Wk.o_wide.p);
                    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fix_own<TC, true, 1>), dim3(wgrid), dim3(256), 0, s, FO_ARGS, Wk.o_part.p,
                                            FO_TAIL, Wk.o_tS0l.p, Wk.o_tilePS.p, Wk.o_tilePS2.p, M, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, n + 1, Wk.o_wide.p);
                } else {
                    const Best<TC, false> *pp = reinterpret_cast<const Best<TC, false> *>(Wk.o_part.p);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fix_own_lane<TC, false>), dim3(lgrid), dim3(256), 0, s, FO_ARGS, pp, FO_TAIL, (const int32_t *)nullptr,
                                       Wk.o_tilePS.p, (const int64_t *)nullptr, M, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr, n + 1, Wk.o_wide.p);
                    if (wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fix_own<TC, false, 4>), dim3(wgrid), dim3(256), 0, s, FO_ARGS, pp, FO_TAIL,
                                                 (const int32_t *)nullptr, Wk.o_tilePS.p, (const int64_t *)nullptr, M, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr, n + 1, Wk.o_wide.p);
                    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fix_own<TC, false, 1>), dim3(wgrid), dim3(256), 0, s, FO_ARGS, pp, FO_TAIL,
                                            (const int32_t *)nullptr, Wk.o_tilePS.p, (const int64_t *)nullptr, M, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr, n + 1, Wk.o_wide.p);
                }
#undef FO_ARGS
#undef FO_TAIL
            }
            CP_HIP(hipGetLastError());
        }
        if (P.nlong <= 0 || P.T <= 0) continue;
        // ---- from here on the round consists of the flattened tasks only (their number: rc->nlong)
        const int64_t gtile = spec ? (int64_t)Wk.tileS.n : cdiv(P.T, LT);      // (capacity >= the true tile count, checked)
        if (g_opt_dbg & 128) {                    // poison everything a round must write before it reads
            int pat = (g_opt_dbg & 256) ? 0x00 : 0x7F;
            CP_HIP(hipMemsetAsync(Wk.loc.p, pat, Wk.loc.bytes(), s));
            if (hyp) CP_HIP(hipMemsetAsync(Wk.loc2.p, pat, Wk.loc2.bytes(), s));
            CP_HIP(hipMemsetAsync(Wk.partL.p, pat, Wk.partL.bytes(), s));
            CP_HIP(hipMemsetAsync(Wk.partR.p, pat, Wk.partR.bytes(), s));
            CP_HIP(hipMemsetAsync(Wk.tileS.p, pat, Wk.tileS.bytes(), s));
            if (hyp) CP_HIP(hipMemsetAsync(Wk.tileS2.p, pat, Wk.tileS2.bytes(), s));
        }
        hipLaunchKernelGGL(k_tile_t0, dim3((unsigned)cdiv(gtile, 256)), dim3(256), 0, s, rc, Wk.offs.p, Wk.tdesc.p, Wk.tile_t0.p,
                           Wk.tile_rec.p, (int)((g_opt_dbg & 32) != 0), Wk.taskR.p);
        {
            // algorithmic bytes of one launch: per flattened step the stepped column's link
            // entries (4 B x N/n, plus 4 B x nonempty-rows/n for hyperedge costs), its colptr entry (8 B), the
            // candidate's previous-layer cost (8 B) and the task-descriptor share (offsets/B/anchor/row, amortised 8 B)
            ProfScope ps(PROF_EXPAND, s, (double)P.T * step_bytes);
#define LP_ARGS R, rc, Wk.offs.p, Wk.tdesc.p, Wk.tS0l.p, Wk.tb.p, A->pos32.p, A->next.p, hyp ? A->fpos32.p : (const int32_t *)nullptr,           \
                hyp ? A->flast.p : (const int32_t *)nullptr, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, Wk.loc.p, Wk.loc2.p, Wk.tileS.p, Wk.tileS2.p,     \
                Wk.taskR.p, Wk.tile_t0.p, Wk.tile_rec.p, W, M, alpha
            if (hyp) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpass<TC, true>), dim3((unsigned)cdiv(gtile, 4)), dim3(256), 0, s, LP_ARGS, Wk.partR.p, Wk.partL.p);
            else     hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpass<TC, false>), dim3((unsigned)cdiv(gtile, 4)), dim3(256), 0, s, LP_ARGS,
                                        reinterpret_cast<Best<TC, false> *>(Wk.partR.p), reinterpret_cast<Best<TC, false> *>(Wk.partL.p));
#undef LP_ARGS
        }
        note(rd, 1, PROF_EXPAND);
        {
            ProfScope ps(PROF_CARRY, s, 12.0 * (double)cdiv(P.T, LT));
            exclusive_scan_i32_lb(Wk.tileS.p, Wk.tilePS.p, &rc->ntile, (int64_t)Wk.tileS.n, nullptr, Wk.scanws, s);
            if (hyp) exclusive_scan_i32_lb(Wk.tileS2.p, Wk.tilePS2.p, &rc->ntile, (int64_t)Wk.tileS.n, nullptr, Wk.scanws, s);
        }
        unsigned wgrid = (unsigned)std::min<int64_t>(cdiv(gtile, 4), 8192);     // the wave kernels walk work lists
        {
            ProfScope ps(PROF_EVAL, s, 0.0);
#define SP_ARGS R, rc, Wk.offs.p, Wk.tdesc.p, Wk.tS0l.p, Wk.tb.p, A->pos32.p, Wk.loc.p, Wk.loc2.p, Wk.tile_t0.p, Wk.taskR.p, Wk.tileS.p, Wk.tileS2.p
            if (hyp) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_span_short<TC, true>), dim3((unsigned)cdiv(gtile, 256)), dim3(256), 0, s, SP_ARGS,
                               Wk.partR.p, W, M, alpha, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, Wk.open_list.p, Wk.fix_list.p, Wk.tile_rec.p);
            else     hipLaunchKernelGGL(HIP_KERNEL_NAME(k_span_short<TC, false>), dim3((unsigned)cdiv(gtile, 256)), dim3(256), 0, s, SP_ARGS,
                               reinterpret_cast<const Best<TC, false> *>(Wk.partR.p), W, M, alpha, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr,
                               Wk.open_list.p, Wk.fix_list.p, Wk.tile_rec.p);
#undef SP_ARGS
#define OP_ARGS R, rc, Wk.offs.p, Wk.tdesc.p, Wk.tS0l.p, A->pos32.p, Wk.loc.p, Wk.loc2.p, Wk.tile_t0.p, Wk.tilePS.p
            if (hyp) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_open<TC, true>), dim3(wgrid), dim3(256), 0, s, OP_ARGS, Wk.tilePS2.p, W, M, alpha, Wk.partL.p,
                               Wk.open_list.p);
            else     hipLaunchKernelGGL(HIP_KERNEL_NAME(k_open<TC, false>), dim3(wgrid), dim3(256), 0, s, OP_ARGS, (const int64_t *)nullptr, W, M, alpha,
                               reinterpret_cast<Best<TC, false> *>(Wk.partL.p), Wk.open_list.p);
#undef OP_ARGS
        }
        {
            ProfScope ps(PROF_FIX, s, 8.0 * (double)cdiv(P.T, LT));
            if (hyp) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fix<TC, true>), dim3(wgrid), dim3(256), 0, s, rc, Wk.offs.p, Wk.taskR.p,
                               Wk.partL.p, Wk.partR.p, Wk.tdesc.p, Wk.tb.p, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, n + 1, Wk.fix_list.p,
                               Wk.tile_rec.p, Wk.tilePS.p, Wk.tilePS2.p, Wk.tS0l.p, M);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fix<TC, false>), dim3(wgrid), dim3(256), 0, s, rc, Wk.offs.p, Wk.taskR.p,
                               reinterpret_cast<const Best<TC, false> *>(Wk.partL.p), reinterpret_cast<const Best<TC, false> *>(Wk.partR.p),
                               Wk.tdesc.p, Wk.tb.p, Wk.opt.p, Wk.nnopt.p, (int32_t *)nullptr, n + 1, Wk.fix_list.p,
                               Wk.tile_rec.p, Wk.tilePS.p, (const int64_t *)nullptr, (const int32_t *)nullptr, M);
        }
        CP_HIP(hipGetLastError());
    }
    if (leaf) {
        // every row that is not a multiple of 64: inner planes, outer planes and the combine in one pass (dp_leaf.inc)
        const int64_t c0 = rlo > 1 ? rlo : 1, c1 = rhi < n ? rhi : n;
        if (c1 >= c0) {
            const int64_t g0 = c0 >> LEAF_T, g1 = c1 >> LEAF_T;
            // bytes: the group's link entries twice (prev, next), column pointers, previous-layer costs, the two output rows
            ProfScope ps(PROF_LEAF, s, (8.0 * (avg_deg + self_deg) + 4.0 + 8.0 + 12.0) * (double)((g1 - g0 + 1) << LEAF_T));
            const int need = G.win ? G.s - LEAF_T + 2 : nbits - LEAF_T;           // outer planes a group can have (+ the pseudo-plane)
            CP_REQUIRE(need <= 25, CP_EINTERNAL, "leaf pass: more outer planes than counters");
            // With a gap pass in round 6 every task of that round with gap_min candidates or more has finished its rows (and the
            // ranges of a leaf group in a plane b > 6 lie inside the range of the round-6 task of one of its two bounding rows): a longer
            // range in a leaf group can only come from plane cells a mispredicted speculative layer never wrote (zeros after
            // allocation: a range of millions of candidates per group, seconds of work for a layer that is redone anyway).  Such a
            // group flags the layer (err) and skips the plane.
            const int32_t lcap = (gaps && g_opt_gap_tau >= LEAF_T) ? (int32_t)std::max<int64_t>(g_opt_gap_min, 2) : INT32_MAX;
            LeafArgs<TC, false> L0{n, g0, g1 - g0 + 1, c0, c1, nbits, (int32_t)(g_opt_block_tables != 0), A->pos32.p, A->prev.p, A->next.p, A->col.p,
                                   nullptr, nullptr, nullptr, nullptr, nullptr, Wk.opt.p, Wk.nnopt.p, nullptr, Wk.fin.p, Wk.fin_stamp, W, M, alpha, cst_out, ptr_out,
                                   G, Wk.leaf_anch.p, nullptr, lcap, &Wk.rc.p->err, pzp};
            if (hyp) {
                LeafArgs<TC, true> L1{n, g0, g1 - g0 + 1, c0, c1, nbits, (int32_t)(g_opt_block_tables != 0), A->pos32.p, A->prev.p, A->next.p, A->col.p,
                                      A->fpos32.p, A->flast.p, A->ffirst.p, A->lpos32.p, A->lfirst.p, Wk.opt.p, Wk.nnopt.p, Wk.nlopt.p, Wk.fin.p, Wk.fin_stamp, W, M, alpha,
                                      cst_out, ptr_out, G, Wk.leaf_anch.p, Wk.leaf_anch2.p, lcap, &Wk.rc.p->err, pzp};
                const unsigned lg = (unsigned)cdiv(L1.ngroups, LeafWPB<true>::v);
                if (need <= 18) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_leaf<TC, true, 18>), dim3(lg), dim3(64 * LeafWPB<true>::v), 0, s, L1);
                else             hipLaunchKernelGGL(HIP_KERNEL_NAME(k_leaf<TC, true, 25>), dim3(lg), dim3(64 * LeafWPB<true>::v), 0, s, L1);
            } else {
                const unsigned lg = (unsigned)cdiv(L0.ngroups, LeafWPB<false>::v);
                if (need <= 18) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_leaf<TC, false, 18>), dim3(lg), dim3(64 * LeafWPB<false>::v), 0, s, L0);
                else             hipLaunchKernelGGL(HIP_KERNEL_NAME(k_leaf<TC, false, 25>), dim3(lg), dim3(64 * LeafWPB<false>::v), 0, s, L0);
            }
            CP_HIP(hipGetLastError());
        }
    }
    if (leaf) {
        ProfScope ps(PROF_COMBINE, s, 24.0 * (double)((n >> LEAF_T) + 1));
        const int64_t c0 = rlo > 0 ? rlo : 0, c1 = rhi < n ? rhi : n;
        const int64_t f0 = (c0 + LEAF_G - 1) >> LEAF_T, f1 = c1 >> LEAF_T;       // the multiples of 64 in [c0, c1]
        if (c1 >= c0 && f1 >= f0 && G.win)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_combine_win<TC>), dim3((unsigned)cdiv(f1 - f0 + 1, 256)), dim3(256), 0, s, 0, G, n, f0, f1, A->pos32.p,
                           Wk.opt.p, Wk.nnopt.p, hyp ? Wk.nlopt.p : (const int32_t *)nullptr, W, M, alpha, cst_out, ptr_out, LEAF_T, pzp);
        else if (c1 >= c0 && f1 >= f0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_combine<TC>), dim3((unsigned)cdiv(f1 - f0 + 1, 256)), dim3(256), 0, s, 0, n, f0, f1, nbits, A->pos32.p,
                           Wk.opt.p, Wk.nnopt.p, hyp ? Wk.nlopt.p : (const int32_t *)nullptr, W, M, alpha, cst_out, ptr_out, LEAF_T, pzp);
    } else {
        ProfScope ps(PROF_COMBINE, s, 24.0 * (double)(n + 1));
        int64_t c0 = rlo > 0 ? rlo : 0, c1 = rhi < n ? rhi : n;
        // threads in row order.  (lvl = 1, cp_set_option("dbg", 16384): in plane-slot order, so that the dozen plane reads of a row are
        // contiguous across a wave -- measured 867 us against 650 us at config 3: the 2 x 12 gathers W[p], pos[p] of a row are what
        // counts, and they are neighbours only for neighbouring rows.)
        const int lvl = (g_opt_dbg & 16384) ? (2 * (c1 - c0 + 1) >= n + 1) : 0;
        const int64_t nthr = lvl ? n + 1 : c1 - c0 + 1;
        if (c1 >= c0 && G.win)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_combine_win<TC>), dim3((unsigned)cdiv(nthr, 256)), dim3(256), 0, s, lvl, G, n, c0, c1, A->pos32.p,
                           Wk.opt.p, Wk.nnopt.p, hyp ? Wk.nlopt.p : (const int32_t *)nullptr, W, M, alpha, cst_out, ptr_out, 0, pzp);
        else if (c1 >= c0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_combine<TC>), dim3((unsigned)cdiv(nthr, 256)), dim3(256), 0, s, lvl, n, c0, c1, nbits, A->pos32.p,
                           Wk.opt.p, Wk.nnopt.p, hyp ? Wk.nlopt.p : (const int32_t *)nullptr, W, M, alpha, cst_out, ptr_out, 0, pzp);
    }
    CP_HIP(hipGetLastError());
    // ---- the true counts of every round: the next layer's prediction, this layer's verdict
    std::vector<RoundCounts> got((size_t)NR);
    int32_t pz_hits[2] = {0, 0};
    CP_HIP(hipMemcpyAsync(got.data(), Wk.rc.p, sizeof(RoundCounts) * (size_t)NR, hipMemcpyDeviceToHost, s));
    if (pzp) CP_HIP(hipMemcpyAsync(pz_hits, pzp, sizeof(pz_hits), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    bool ok = true;
    for (int rd = 0; rd < NR; rd++) {
        const RoundCounts &g = got[(size_t)rd], &u = used[(size_t)rd];
        CP_REQUIRE(!(g.err && !spec && !any_forced), CP_EINTERNAL, "DP work list overflow");
        if (g.err) ok = false;                                                                  // a buffer was too small (the caller runs the layer again, plainly)
        if (g.nown > 0 && g.NT > 0 && !(u.nown > 0 && u.NT > 0)) ok = false;                    // a stage was skipped
        if (g.nlong > 0 && g.T > 0 && !(u.nlong > 0 && u.T > 0)) ok = false;
    }
    if (pzp) {
        g_poison_hits += pz_hits[0];
        CP_REQUIRE(!(ok && pz_hits[0] > 0), CP_EINTERNAL, "a DP layer that is not redone read plane cells nobody wrote (poison mode)");
    }
    if (ok) {
        for (auto &pt : patches) {
            const RoundCounts &g = got[(size_t)pt.rd];
            if (pt.idx < g_prof_pending.size())
                g_prof_pending[pt.idx].bytes = (pt.kind == 0 ? (double)g.own_steps : (double)g.T) * step_bytes;
        }
        Wk.pred = got; Wk.pred_ok = true; Wk.pred_rlo = rlo; Wk.pred_rhi = rhi; Wk.pred_win = G.win; Wk.pred_w = G.w;
        if (allow_force && 2 * (rhi - rlo + 1) >= Wk.force_rows) {
            Wk.force_own.resize((size_t)NR, 0);
            Wk.force_rows = rhi - rlo + 1;
            for (int rd = 1; rd < NR; rd++) {
                const RoundCounts &g = got[(size_t)rd];
                if (g.nlong > 0) Wk.force_own[(size_t)rd] = g.nlong <= g_opt_force_max;         // (a forced round reports none: its flag stays)
            }
        }
    }
    return ok;
}

template <typename TC>
void dp_total_layer(cp_csr_s *A, const DevModel<TC> &M, TC alpha, const TC *W, TC *cst_out, int32_t *ptr_out, void *work_,
                    int64_t rlo, int64_t rhi, int64_t wwin)
{
    auto &Wk = *reinterpret_cast<LayerWork<TC> *>(work_);
    int64_t n = A->n;
    bool hyp = M.kind == CP_MODEL_HYPEREDGE_CUT;
    int nbits = 1;
    while (((int64_t)1 << nbits) <= n) nbits++;
    CP_REQUIRE(nbits <= NBMAX, CP_EINVAL, "n exceeds the bit-plane budget");
    if (Wk.n != n || Wk.hyp != hyp) {
        // (the shape is recorded only after every allocation succeeded: a hipMalloc failure leaves n == -1 and the next call starts over)
        Wk.n = -1; Wk.nbits = nbits; Wk.hyp = hyp; Wk.pred_ok = false; Wk.ra_built = false; Wk.win_built = false; Wk.mir_built = false; Wk.force_own.clear(); Wk.force_rows = 0;      // (the window anchors of a hyperedge model carry a second array)
        Wk.o_rec.release(); Wk.loc.release();       // (the per-tile arrays are re-made for the new shape on first use)
        size_t plane = (size_t)nbits * (size_t)(n + 1);
        Wk.opt.alloc(plane); Wk.nnopt.alloc(plane); Wk.cr.alloc(plane);
        if (hyp) { Wk.nlopt.alloc(plane); Wk.crl.alloc(plane); }
        // The winners are zeroed once: a speculatively sized layer may skip a stage that turns out to have work (the layer is then
        // redone), and until the redo later rounds read plane cells nobody wrote.  Whatever they hold must be a valid column
        // index -- a stale winner of an earlier layer is, fresh device memory is not (a k_lpass_own wave once streamed from
        // column 13868 of a 3000-column matrix).
        CP_HIP(hipMemsetAsync(Wk.opt.p, 0, Wk.opt.bytes(), A->stream));
        CP_HIP(hipMemsetAsync(Wk.nnopt.p, 0, Wk.nnopt.bytes(), A->stream));
        CP_HIP(hipMemsetAsync(Wk.cr.p, 0, Wk.cr.bytes(), A->stream));
        if (hyp) { CP_HIP(hipMemsetAsync(Wk.nlopt.p, 0, Wk.nlopt.bytes(), A->stream)); CP_HIP(hipMemsetAsync(Wk.crl.p, 0, Wk.crl.bytes(), A->stream)); }
        int64_t mx = n;
        for (int tau = 0; tau < nbits; tau++) { RoundDesc R; make_round(R, false, tau, nbits, n, 0, n); if (R.ntask > mx) mx = R.ntask; }
        // windowed layers give every row of a round a task in EVERY plane above tau (up to nbits - 1 - tau of them), and round A
        // one per head: sum_b (n >> b) < 2 n
        for (int tau = 0; tau < nbits; tau++) mx = std::max<int64_t>(mx, ((n >> (tau + 1)) + 1) * (int64_t)(nbits - 1 - tau));
        mx = std::max<int64_t>(mx, 2 * n + nbits);
        Wk.max_tasks = mx > 0 ? mx : 1;
        size_t mt = (size_t)Wk.max_tasks;
        Wk.tdesc.alloc(mt); Wk.len.alloc(mt); Wk.tb.alloc(mt); Wk.offs.alloc(mt + 1);
        if (hyp) Wk.tS0l.alloc(mt);
        // tasks with tiles of their own have >= own_min >= 64 candidates; the ranges of one plane and round overlap in their end
        // points only: at most n / 64 + (rectangles) of them per plane -- (n / 32 + 64) per plane is a safe cap (k_setup_short guards it)
        int64_t mmin = std::max<int64_t>(2, std::min(g_opt_own_min, g_opt_gap_tau >= 0 ? g_opt_gap_min : g_opt_own_min));
        size_t mo = (size_t)(2 * n / mmin + 64) * (size_t)nbits + 1024;
        if (mo > mt) mo = mt;
        Wk.o_tdesc.alloc(mo); Wk.o_tb.alloc(mo); Wk.o_rlen.alloc(mo); Wk.o_ntl.alloc(mo); Wk.o_toffs.alloc(mo + 1); Wk.o_wide.alloc(mo);
        if (hyp) Wk.o_tS0l.alloc(mo);
        Wk.rc.alloc((size_t)NBMAX + 2);
        Wk.fin.alloc(plane); Wk.last_s0.alloc(64); Wk.fin_stamp = 255;      // (fresh memory: the first layer clears it)
        Wk.n = n;
    }
    // geometry of this call: wwin > 0: candidates of row r are max(0, r - wwin) <= p <= r
    Geo G{0, 0, 0};
    if (wwin > 0) { G.win = 1; G.w = wwin; G.s = 0; while (((int64_t)2 << G.s) <= wwin) G.s++; if (G.s > nbits - 1) G.s = nbits - 1; }
    // (w >= 2^(nbits-1) > n/2: planes 0 .. nbits-1 with S = 2^(nbits-1) <= w still tile every window -- the common block just
    //  reaches further left than column 0 and is clamped)
    Wk.G = G;
    if (G.win && (!Wk.win_built || Wk.win_w != wwin)) win_build<TC>(A, Wk);
    bool spec = Wk.pred_ok && !g_opt_nospec && Wk.pred_win == G.win && Wk.pred_w == G.w;
    Wk.pred_scale = 1.0;
    if (spec && (Wk.pred_rlo != rlo || Wk.pred_rhi != rhi)) {
        // another row tile than the layer the counts come from: the unconstrained driver keeps its tile, the constrained one moves
        // a window over the rows -- take the prediction per row of the tile
        const double a = (double)(Wk.pred_rhi - Wk.pred_rlo + 1), b = (double)(rhi - rlo + 1);
        // (a much larger tile than the counts come from -- the one-row last layer of an earlier call on this handle -- would
        //  scale a handful of tasks into buffers of any size: such a layer takes its exact counts instead)
        if (!G.win || a < 1 || b < 1 || b > 2.0 * a) spec = false;
        else Wk.pred_scale = b / a;
    }
    if (Wk.pred_win != G.win || Wk.pred_w != G.w || (int)Wk.force_own.size() != (G.win ? G.s + 1 : Wk.nbits) + 1) { Wk.force_own.clear(); Wk.force_rows = 0; }      // (another geometry: other rounds)
    if (run_layer<TC>(A, M, alpha, W, cst_out, ptr_out, Wk, rlo, rhi, spec)) return;
    // the prediction missed (a stage that had been empty, or a buffer too small): the same layer again with exact counts
    g_spec_redo++;
    Wk.force_own.clear(); Wk.force_rows = 0;
    bool ok = run_layer<TC>(A, M, alpha, W, cst_out, ptr_out, Wk, rlo, rhi, false, false);
    CP_REQUIRE(ok, CP_EINTERNAL, "DP layer failed with exact counts");
}

// Table-level window on a finished layer (tests): the per-block winners the combine step merges.  For every bit plane b and
// every row r (0-based) with bit b set, opt_out[b * (n+1) + r] = 1-based j of the RIGHTMOST arg-min of W[p] + f(p, r) over the
// row's Fenwick block [r_b - 2^b, r_b) and nn_out / nl_out the net (self-net) count of that part; 0 where bit b is clear.
// Valid for the rows the last dp_total_layer call computed.
template <typename TC>
int dp_total_block_tables(cp_csr_s *A, void *work_, int64_t *opt_out, int64_t *nn_out, int64_t *nl_out)
{
    auto &Wk = *reinterpret_cast<LayerWork<TC> *>(work_);
    const int64_t n = A->n, n1 = n + 1;
    CP_REQUIRE(Wk.n == n && Wk.opt.p, CP_EINVAL, "no layer has been computed by the O(n log^2 n) scheme on this handle");
    CP_REQUIRE(Wk.planes_full, CP_EINVAL, "the last layer kept no per-block winners for its leaf rows: cp_set_option(\"block_tables\", 1) before the layer");
    const size_t plane = (size_t)Wk.nbits * (size_t)n1;
    std::vector<int32_t> ho(plane), hn(plane), hl;
    CP_HIP(hipMemcpyAsync(ho.data(), Wk.opt.p, plane * sizeof(int32_t), hipMemcpyDeviceToHost, A->stream));
    CP_HIP(hipMemcpyAsync(hn.data(), Wk.nnopt.p, plane * sizeof(int32_t), hipMemcpyDeviceToHost, A->stream));
    if (Wk.hyp && nl_out) { hl.resize(plane); CP_HIP(hipMemcpyAsync(hl.data(), Wk.nlopt.p, plane * sizeof(int32_t), hipMemcpyDeviceToHost, A->stream)); }
    CP_HIP(hipStreamSynchronize(A->stream));
    for (int b = 0; b < Wk.nbits; b++)
        for (int64_t r = 0; r <= n; r++) {
            const size_t o = (size_t)b * (size_t)n1 + (size_t)r;
            opt_out[o] = 0; nn_out[o] = 0; if (nl_out) nl_out[o] = 0;
            if (r == 0 || !((r >> b) & 1)) continue;
            int t = 0; while (!((r >> t) & 1)) t++;
            const size_t slot = (size_t)b * (size_t)n1 + (size_t)((n - (n >> t)) + (r >> (t + 1)));      // prow(r, n)
            opt_out[o] = (int64_t)ho[slot] + 1; nn_out[o] = hn[slot];
            if (nl_out && !hl.empty()) nl_out[o] = hl[slot];
        }
    return Wk.nbits;
}
template int dp_total_block_tables<int64_t>(cp_csr_s *, void *, int64_t *, int64_t *, int64_t *);
template int dp_total_block_tables<double>(cp_csr_s *, void *, int64_t *, int64_t *, int64_t *);

template <typename TC> void *dp_total_work_new() { return new LayerWork<TC>(); }
template <typename TC> static void work_free_fn(void *w) { delete reinterpret_cast<LayerWork<TC> *>(w); }
template <typename TC> static void work_reset_fn(void *w) { auto *W = reinterpret_cast<LayerWork<TC> *>(w); W->ra_built = false; W->pred_ok = false; W->win_built = false; W->mir_built = false; W->force_own.clear(); W->force_rows = 0; }
template <typename TC> void *dp_total_work_get(cp_csr_s *A)
{
    const int i = sizeof(TC) == sizeof(double) && ((TC)0.5 != (TC)0) ? 1 : 0;
    if (!A->dp_work[i]) { A->dp_work[i] = new LayerWork<TC>(); A->dp_work_free_fn[i] = work_free_fn<TC>; A->dp_work_reset_fn[i] = work_reset_fn<TC>; }
    return A->dp_work[i];
}
template void *dp_total_work_get<int64_t>(cp_csr_s *);
template void *dp_total_work_get<double>(cp_csr_s *);
template <typename TC> void dp_total_work_free(void *w) { delete reinterpret_cast<LayerWork<TC> *>(w); }

template void dp_total_layer<int64_t>(cp_csr_s *, const DevModel<int64_t> &, int64_t, const int64_t *, int64_t *, int32_t *, void *, int64_t, int64_t, int64_t);
template void dp_total_layer<double>(cp_csr_s *, const DevModel<double> &, double, const double *, double *, int32_t *, void *, int64_t, int64_t, int64_t);
template void *dp_total_work_new<int64_t>();
template void *dp_total_work_new<double>();
template void dp_total_work_free<int64_t>(void *);
template void dp_total_work_free<double>(void *);

}  // namespace cpk
