// seq.hip -- the inherently sequential partitioners of the hot path, run on the device by ONE wave whose
// 64 lanes execute the same scalar program (every lane performs every store, so a lane only ever reads what it
// wrote itself) and fan out only where the reference has an independent inner loop:
//
//   pack_stripe(A, DynamicTotalChunker(f | ConstrainedCost(f,w,w_max)))          DynamicChunker.jl:20-75
//   partition_stripe(A, K, Dynamic*{Splitter,Chunker}(ConstrainedCost(...)))      DynamicSplitter.jl:206-314
//   pack_stripe / partition_stripe with ConvexTotal{Chunker,Splitter}             ConvexTotalChunker.jl:9-265
//
// Oracle values come from the device counters (wavelet rank queries for nets / self-nets, colptr for pins),
// or from the stateful block oracle (BlockCosts.jl:66-142) restated on device arrays.  Tie rules, the
// Extended{T} infinity arithmetic and the WindowConstrainedMatrix read/write semantics follow the reference
// statement for statement.  These kernels are latency-bound by design (one dependent chain); the throughput
// paths are dp_total.hip / bisect.hip.
#include "csr.hpp"
#include "model.hpp"
#include "wavelet.hpp"
#include "dp.hpp"
#include <type_traits>
#include <memory>

namespace cpk {

// ------------------------------------------------------------------ device oracle ocl(j, j', k)
template <typename TC>
struct SeqOracle {
    DevModel<TC> M;
    const int64_t *pos;
    const int32_t *row;
    int64_t n, m;
    int32_t has_net, has_self;
    WaveletDev net, self;
    const int64_t *lpos;
    // BlockComponentCostStepOracle state (BlockCosts.jl:46-64), all on device
    const int32_t *P_asg;      // m   : part of each row (1-based part ids)
    const int64_t *P_spl;      // Kr+1 (1-based values)
    int64_t Kr;
    int64_t *hst;              // Kr
    TC *Delta;                 // R x (n+1), column-major
    TC *d;                     // R
    int64_t *cursor;           // [0] = ocl.j, [1] = ocl.j'
    // optional table of every pair inside a width window: Ftab[j' * (Wc+1) + (j'-j)] = f(j, j') for j'-j <= Wc
    const TC *Ftab; int64_t Wc;
    // block-row components
    int32_t br_const[CP_MAX_R]; TC br_c[CP_MAX_R]; const TC *br_tab[CP_MAX_R]; int64_t br_len[CP_MAX_R], br_lo[CP_MAX_R];
};

template <typename TC>
__device__ TC block_call(const SeqOracle<TC> &O, int64_t j, int64_t jp)
{
    const DevModel<TC> &f = O.M;
    int R = f.R;
    int64_t oj = O.cursor[0], ojp = O.cursor[1];
    TC *d = O.d, *Dl = O.Delta;
#define DL(r, c) Dl[((c) - 1) * R + ((r) - 1)]
    if (jp < ojp) {                                          // :82-87
        oj = 1; ojp = 1;
        for (int r = 0; r < R; r++) d[r] = (TC)0;
        for (int64_t k = 1; k <= O.Kr; k++) O.hst[k - 1] = 1;
    }
    while (ojp < jp) {                                       // :88-118
        int64_t q = O.pos[ojp - 1], qp = O.pos[ojp];          // 0-based nonzero range of column ojp
        for (int r = 1; r <= R; r++) DL(r, ojp + 1) = (TC)0;
        for (int64_t _q = q; _q < qp; _q++) {
            int64_t i = O.row[_q];                            // 0-based row
            int64_t k = O.P_asg[i];
            int64_t j0 = O.hst[k - 1] - 1;
            int64_t u = O.P_spl[k] - O.P_spl[k - 1];
            if (j0 < ojp) {
                for (int r = 1; r <= R; r++) DL(r, j0 + 1) = cadd(DL(r, j0 + 1), (TC)0 - dm_comp(O.br_const[r - 1], O.br_c[r - 1], O.br_tab[r - 1], O.br_len[r - 1], O.br_lo[r - 1], u));
                for (int r = 1; r <= R; r++) DL(r, ojp + 1) = cadd(DL(r, ojp + 1), dm_comp(O.br_const[r - 1], O.br_c[r - 1], O.br_tab[r - 1], O.br_len[r - 1], O.br_lo[r - 1], u));
            }
            if (j0 < oj)
                for (int r = 1; r <= R; r++) d[r - 1] = cadd(d[r - 1], dm_comp(O.br_const[r - 1], O.br_c[r - 1], O.br_tab[r - 1], O.br_len[r - 1], O.br_lo[r - 1], u));
            O.hst[k - 1] = ojp + 1;
        }
        ojp += 1;
    }
    while (j < oj) { oj -= 1; for (int r = 1; r <= R; r++) d[r - 1] = cadd(d[r - 1], DL(r, oj + 1)); }
    while (j > oj) { for (int r = 1; r <= R; r++) d[r - 1] = cadd(d[r - 1], (TC)0 - DL(r, oj + 1)); oj += 1; }
#undef DL
    int64_t w = jp - j;
    TC c = dm_comp(f.ac_const, f.ac_c, f.ac_tab, f.ac_len, f.ac_lo, w);
    for (int r = 1; r <= R; r++) c = cadd(c, cmulv(d[r - 1], dm_comp(f.bc_const[r - 1], f.bc_c[r - 1], f.bc_tab[r - 1], f.bc_len[r - 1], f.bc_lo[r - 1], w)));
    O.cursor[0] = oj; O.cursor[1] = ojp;
    return c;
}

template <typename TC>
__device__ TC ocl(const SeqOracle<TC> &O, int64_t j, int64_t jp, int64_t k)
{
    if (O.M.kind == CP_MODEL_BLOCK) return block_call(O, j, jp);
    if (O.Ftab && jp >= j && jp - j <= O.Wc) return O.Ftab[jp * (O.Wc + 1) + (jp - j)];     // (j > j' happens: see cp_component_t)
    int64_t p = j - 1, r = jp - 1;
    int64_t np = O.pos[r] - O.pos[p];
    int64_t nn = 0, nl = 0;
    if (O.has_net) nn = np - wt_count_le(O.net, O.n - p, O.pos[r]);
    if (O.has_self) nl = wt_count_le(O.self, O.n - p, O.lpos[r]);
    return dm_apply(O.M, dm_alpha(O.M, k), r - p, np, nn, nl);
}

// every pair (j, j') with j' - j <= Wc, one thread per j': walks the candidate j downwards with the same link-array
// left steps as the DP (nets += #{q in column : next[q] >= r}; self nets += #{rows first == column, last < r})
template <typename TC>
__global__ void __launch_bounds__(256) k_window_table(DevModel<TC> M, int64_t n, int64_t Wc, const int64_t *__restrict__ pos,
                                                      const int32_t *__restrict__ next, const int64_t *__restrict__ fpos,
                                                      const int32_t *__restrict__ flast, TC *__restrict__ F)
{
    int64_t jp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;       // 1-based j'
    if (jp > n + 1) return;
    int64_t r = jp - 1;
    TC alpha = M.p[CP_P_ALPHA];
    bool nets = M.kind == CP_MODEL_CONNECTIVITY || M.kind == CP_MODEL_COLBLOCK || M.kind == CP_MODEL_HYPEREDGE_CUT;
    bool self = M.kind == CP_MODEL_HYPEREDGE_CUT;
    int64_t nn = 0, nl = 0;
    TC *row = F + jp * (Wc + 1);
    row[0] = dm_apply(M, alpha, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0);
    int32_t rr = (int32_t)r;
    for (int64_t d = 1; d <= Wc; d++) {
        int64_t p = r - d;
        if (p < 0) break;
        if (nets) for (int64_t q = pos[p]; q < pos[p + 1]; q++) nn += (next[q] >= rr);
        if (self) for (int64_t q = fpos[p]; q < fpos[p + 1]; q++) nl += (flast[q] < rr);
        row[d] = dm_apply(M, alpha, d, pos[r] - pos[p], nn, nl);
    }
}

// ------------------------------------------------------------------ weights of ConstrainedCost
struct SeqWeight {
    int32_t kind;          // FEASIBLE / VERTEX_COUNT / WORK
    int32_t dtype;
    int64_t pi[3]; double pf[3];
    int64_t wmax_i; double wmax_f;
    const int64_t *pos;
};
__device__ __forceinline__ bool wgt_gt(const SeqWeight &W, int64_t j, int64_t jp)
{
    if (W.kind == CP_MODEL_FEASIBLE) return false;
    if (W.kind == CP_MODEL_VERTEX_COUNT) return (jp - j) > W.wmax_i;
    int64_t nv = jp - j, np = W.pos[jp - 1] - W.pos[j - 1];
    if (W.dtype == CP_I64) return (W.pi[0] + nv * W.pi[1] + np * W.pi[2]) > W.wmax_i;
    return (W.pf[0] + (double)nv * W.pf[1] + (double)np * W.pf[2]) > W.wmax_f;
}
__device__ __forceinline__ bool wgt_le(const SeqWeight &W, int64_t j, int64_t jp) { return !wgt_gt(W, j, jp); }

#define A1(a, k) ((a)[(k) - 1])

// ------------------------------------------------------------------ pack_stripe, DynamicTotalChunker (DynamicChunker.jl:20-56)
// candidates j = j0 .. j'-1 of one row are evaluated by the lanes in parallel (stateless models); strict <
// while scanning j upwards == the smallest j among the minima.
template <typename TC>
__global__ void __launch_bounds__(64) k_pack_dynamic(SeqOracle<TC> O, SeqWeight W, TC *__restrict__ cst, int64_t *__restrict__ spl,
                                                     int32_t *__restrict__ status)
{
    int lane = threadIdx.x;
    int64_t n = O.n;
    bool serial = O.M.kind == CP_MODEL_BLOCK;             // the block oracle is stateful: literal serial sweep
    A1(cst, 1) = (TC)0;
    int64_t j0 = 1;
    for (int64_t jp = 2; jp <= n + 1; jp++) {
        while (wgt_gt(W, j0, jp)) j0 += 1;
        if (!(j0 < jp)) { *status = CP_EINVAL; return; }   // @assert j0 < j'
        TC best_c; int64_t best_j;
        if (serial) {
            best_c = cadd(A1(cst, j0), ocl(O, j0, jp, 0)); best_j = j0;
            for (int64_t j = j0 + 1; j <= jp - 1; j++) {
                TC c = cadd(A1(cst, j), ocl(O, j, jp, 0));
                if (c < best_c) { best_c = c; best_j = j; }
            }
        } else {
            bool have = false; best_c = (TC)0; best_j = 0;
            for (int64_t j = j0 + lane; j <= jp - 1; j += 64) {
                TC c = cadd(A1(cst, j), ocl(O, j, jp, 0));
                if (!have || c < best_c) { best_c = c; best_j = j; have = true; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                int src = (lane + o) & 63;
                int oh = __shfl((int)have, src);
                int64_t oj;
                TC oc;
                {
                    int lo = __shfl((int)(best_j & 0xffffffffll), src), hi = __shfl((int)(best_j >> 32), src);
                    oj = ((int64_t)hi << 32) | (uint32_t)lo;
                    union { TC t; int2 i; } u; u.t = best_c;
                    int2 r2; r2.x = __shfl(u.i.x, src); r2.y = __shfl(u.i.y, src);
                    union { TC t; int2 i; } v; v.i = r2; oc = v.t;
                }
                if (lane + o < 64 && oh && (!have || oc < best_c || (oc == best_c && oj < best_j))) { best_c = oc; best_j = oj; have = true; }
            }
            // broadcast lane 0's result
            {
                int lo = __shfl((int)(best_j & 0xffffffffll), 0), hi = __shfl((int)(best_j >> 32), 0);
                best_j = ((int64_t)hi << 32) | (uint32_t)lo;
                union { TC t; int2 i; } u; u.t = best_c;
                int2 r2; r2.x = __shfl(u.i.x, 0); r2.y = __shfl(u.i.y, 0);
                union { TC t; int2 i; } v; v.i = r2; best_c = v.t;
            }
        }
        A1(cst, jp) = best_c;
        A1(spl, jp) = best_j;
    }
}

// ------------------------------------------------------------------ Extended{T} (Costs.jl:79-103)
template <typename TC> struct Ext { int32_t inf; TC x; };
template <typename TC> __device__ __forceinline__ Ext<TC> ext_of(TC x) { Ext<TC> e; e.inf = 0; e.x = x; return e; }
template <typename TC> __device__ __forceinline__ Ext<TC> ext_inf() { Ext<TC> e; e.inf = 1; e.x = (TC)0; return e; }
template <typename TC> __device__ __forceinline__ Ext<TC> ext_add(Ext<TC> a, Ext<TC> b) { Ext<TC> e; e.inf = a.inf | b.inf; e.x = cadd(a.x, b.x); return e; }
template <typename TC> __device__ __forceinline__ bool ext_lt(Ext<TC> a, Ext<TC> b) { return (!a.inf && b.inf) || ((!a.inf && !b.inf) && (a.x < b.x)); }
template <typename TC> __device__ __forceinline__ bool ext_eq(Ext<TC> a, Ext<TC> b) { return (a.inf && b.inf) || ((!a.inf && !b.inf) && (a.x == b.x)); }
template <typename TC> __device__ __forceinline__ bool ext_le(Ext<TC> a, Ext<TC> b) { return ext_lt(a, b) || ext_eq(a, b); }

// the f' closure handed to chunk_convex! (ConvexTotalChunker.jl:19,46,199,246)
template <typename TC>
struct Cvx {
    int mode;                  // 0: base[j] + extend(f(j,j',k))   3: sigma re-indexing of `inner`
    int64_t k;
    const Ext<TC> *base; int64_t blo, bhi;       // base[j] valid for j in [blo,bhi] (1-based index into base)
    const Cvx<TC> *inner; const int64_t *sig_j, *sig_jp; int64_t I;
    const SeqWeight *w;        // non-null: f is a ConstrainedCostOracle (Costs.jl:140-146): infinity where w(j,j') > w_max
};

template <typename TC>
__device__ Ext<TC> cvx_eval(const SeqOracle<TC> &O, const Cvx<TC> &F, int64_t j, int64_t jp)
{
    if (F.mode == 3) { j = A1(F.sig_j, F.I - j); jp = A1(F.sig_jp, F.I - jp); const Cvx<TC> &G = *F.inner;
        Ext<TC> b = (G.blo <= j && j <= G.bhi) ? G.base[j] : ext_inf<TC>();
        return ext_add(b, ext_of(ocl(O, j, jp, G.k))); }
    Ext<TC> b = (F.blo <= j && j <= F.bhi) ? F.base[j] : ext_inf<TC>();
    if (F.w && wgt_gt(*F.w, j, jp)) return ext_add(b, ext_inf<TC>());
    return ext_add(b, ext_of(ocl(O, j, jp, F.k)));
}

// destination view with WindowConstrainedMatrix semantics (reads outside give infinity, writes are dropped)
template <typename TC>
struct CView { Ext<TC> *cst; int64_t *ptr; int64_t lo, hi; };
template <typename TC> __device__ __forceinline__ Ext<TC> cv_get(const CView<TC> &V, int64_t jp) { return (V.lo <= jp && jp <= V.hi) ? V.cst[jp] : ext_inf<TC>(); }
template <typename TC> __device__ __forceinline__ void cv_set(const CView<TC> &V, int64_t jp, Ext<TC> c, int64_t p) { if (V.lo <= jp && jp <= V.hi) { V.cst[jp] = c; V.ptr[jp] = p; } }

// chunk_convex!(cst, ptr, f, j0, j'1, ftr)  ConvexTotalChunker.jl:57-112 ; ftr = stack of (j, h) pairs
template <typename TC>
__device__ void chunk_convex(const SeqOracle<TC> &O, const CView<TC> &V, const Cvx<TC> &F, int64_t j0, int64_t jp1, int64_t *ftr)
{
    int64_t top = 0;
#define PUSH(a, b) do { ftr[2 * top] = (a); ftr[2 * top + 1] = (b); top++; } while (0)
    PUSH(j0, jp1 + 1);
    for (int64_t jp = j0 + 1; jp <= jp1; jp++) {
        int64_t j = ftr[2 * (top - 1)], h = ftr[2 * (top - 1) + 1];
        Ext<TC> c = cvx_eval(O, F, j, jp);
        Ext<TC> c2 = cvx_eval(O, F, jp - 1, jp);
        if (ext_le(c, c2)) {
            if (ext_le(c, cv_get(V, jp))) cv_set(V, jp, c, j);
            if (h == jp + 1) top--;
        } else {
            if (ext_le(c2, cv_get(V, jp))) cv_set(V, jp, c2, jp - 1);
            while (top > 0) {
                j = ftr[2 * (top - 1)]; h = ftr[2 * (top - 1) + 1];
                if (ext_lt(cvx_eval(O, F, jp - 1, h - 1), cvx_eval(O, F, j, h - 1))) top--; else break;
            }
            if (top == 0) {
                PUSH(jp - 1, jp1 + 1);
            } else {
                j = ftr[2 * (top - 1)]; h = ftr[2 * (top - 1) + 1];
                int64_t h_lo = jp + 1, h_hi = h - 1;
                while (h_lo <= h_hi) {
                    h = (int64_t)(((uint64_t)(h_lo + h_hi)) >> 1);
                    if (ext_lt(cvx_eval(O, F, jp - 1, h - 1), cvx_eval(O, F, j, h - 1))) h_lo = h + 1; else h_hi = h - 1;
                }
                h = h_hi;
                if (jp + 1 != h) PUSH(jp - 1, h);
            }
        }
    }
#undef PUSH
}

// chunk_convex_constrained!  ConvexTotalChunker.jl:211-265
template <typename TC>
__device__ int32_t chunk_convex_constrained(const SeqOracle<TC> &O, const CView<TC> &V, const Cvx<TC> &F, const SeqWeight &W,
                                            int64_t J0, int64_t Jp1, int64_t *ftr, int64_t *sig_j, int64_t *sig_jp,
                                            int64_t *sig_ptr, Ext<TC> *sig_cst)
{
    int64_t jp1 = J0 + 1;
    while (jp1 < Jp1 && wgt_le(W, J0, jp1 + 1)) jp1 += 1;
    int64_t j0 = J0;
    for (;;) {
        chunk_convex(O, V, F, j0, jp1, ftr);
        if (jp1 == Jp1) break;
        int64_t jp = jp1, I = 1;
        for (int64_t j = j0 + 1; j <= jp1; j++) {
            if (jp > jp1) { A1(sig_jp, I) = jp; I += 1; A1(sig_j, I) = j; }
            while (jp < Jp1 && wgt_le(W, j, jp + 1)) { jp += 1; A1(sig_jp, I) = jp; I += 1; A1(sig_j, I) = j; }
        }
        I += 1;
        if (I == 2) return CP_EINVAL;          // a single column exceeds w_max (the reference indexes sigma_j'[0])
        for (int64_t i = 2; i <= I - 1; i++) sig_cst[i] = ext_inf<TC>();
        Cvx<TC> G; G.mode = 3; G.k = 0; G.base = nullptr; G.blo = 0; G.bhi = -1; G.inner = &F; G.sig_j = sig_j; G.sig_jp = sig_jp; G.I = I; G.w = nullptr;
        CView<TC> SV; SV.cst = sig_cst; SV.ptr = sig_ptr; SV.lo = 1; SV.hi = I - 1;
        chunk_convex(O, SV, G, (int64_t)1, I - 1, ftr);
        for (int64_t ip = 2; ip <= I - 1; ip++) cv_set(V, A1(sig_jp, I - ip), sig_cst[ip], A1(sig_j, I - sig_ptr[ip]));
        j0 = jp1;
        jp1 = A1(sig_jp, I - 2);
    }
    return CP_OK;
}

// pack_stripe(A, ConvexTotalChunker(..))  :9-24, :141-168 ; all arrays are 1-based at index j' (slot 0 unused)
template <typename TC>
__global__ void __launch_bounds__(64) k_pack_convex(SeqOracle<TC> O, SeqWeight W, int constrained, Ext<TC> *cst, int64_t *ptr,
                                                    int64_t *ftr, int64_t *sig_j, int64_t *sig_jp, int64_t *sig_ptr, Ext<TC> *sig_cst,
                                                    int32_t *status)
{
    int64_t n = O.n;
    for (int64_t t = 0; t <= n + 1; t++) { cst[t] = ext_inf<TC>(); ptr[t] = 0; }
    cst[1] = ext_of((TC)0);
    Cvx<TC> F; F.mode = 0; F.k = 0; F.base = cst; F.blo = 1; F.bhi = n + 1; F.inner = nullptr; F.sig_j = nullptr; F.sig_jp = nullptr; F.I = 0; F.w = nullptr;
    CView<TC> V; V.cst = cst; V.ptr = ptr; V.lo = 1; V.hi = n + 1;
    int32_t rc = CP_OK;
    if (!constrained) chunk_convex(O, V, F, (int64_t)1, n + 1, ftr);
    else if (n >= 1) rc = chunk_convex_constrained(O, V, F, W, (int64_t)1, n + 1, ftr, sig_j, sig_jp, sig_ptr, sig_cst);
    if (threadIdx.x == 0) *status = rc;
}

// ------------------------------------------------------------------ ConvexTotalChunker under a width window, LDS-resident
// pack_stripe(A, ConvexTotalChunker(ConstrainedCost(f, VertexCount(), w)))  ConvexTotalChunker.jl:141-168, :211-265.
// The stack algorithm is a dependent chain and its result on a cost that is not convex (the reference's own benchmark cost,
// ColumnBlockComponentCostModel(3, w -> 1 + w), runbenchmarks.jl:21,33) is defined by nothing but its execution, so it is
// executed -- but not out of HBM.  With the width weight the outer loop of chunk_convex_constrained! advances a window of w
// columns per iteration and only ever touches columns [j0, j'1 + w] (:216-264): their costs (a ring of 64 columns), the rows of
// the window table F[j'][j' - j] they need, the sigma re-indexing arrays and the candidate stack all live in LDS; the table
// rows of the window after next are fetched into registers while the current window runs and enter the ring at its end, so
// no step of the chain waits for HBM.  ptr leaves through plain stores.  Same statements, same order, same tie rules as
// chunk_convex! / chunk_convex_constrained! above (which remain the general path: other weights, w > CW_MAXW).
constexpr int CW_RING = 64;          // columns resident (>= 4 w + 2)
constexpr int CW_MAXW = 15;
constexpr int CW_FW = 2 * CW_MAXW + 3;      // table row: widths 0 .. 2 w + 2

template <typename TC>
struct CwState {
    Ext<TC> cst[CW_RING];
    TC F[CW_RING][CW_FW + 1];
    int32_t sig_j[CW_RING], sig_jp[CW_RING], sig_ptr[CW_RING];
    Ext<TC> sig_cst[CW_RING];
    int32_t ftr[2 * CW_RING];
};

// scalars of the walk (registers); S is the kernel's __shared__ state, always passed by reference into force-inlined code so
// that every access stays a DS instruction
struct CwEnv {
    int32_t n, Wc, lo_res, hi_res;       // rows of F / columns of cst resident: [lo_res, hi_res]
    int32_t mode, I;                     // mode 1: sigma view -- indices are positions of the staircase problem
};

// the rare pair outside the resident table goes through the general oracle; kept out of line (its wavelet loops would sit in
// the registers of the whole kernel)
#ifdef CW_STATS
__device__ unsigned long long g_cw_stats[8];
#define CW_STAT(i) do { if (threadIdx.x == 0) g_cw_stats[i]++; } while (0)
#else
#define CW_STAT(i) do { } while (0)
#endif
template <typename TC>
__device__ __noinline__ TC cw_ocl_general(const SeqOracle<TC> *O, int32_t j, int32_t jp) { CW_STAT(jp < j ? 0 : 1); return ocl(*O, (int64_t)j, (int64_t)jp, (int64_t)0); }

// wave-uniform values read from LDS go to the scalar unit: the walk's index arithmetic and branches then cost SALU slots, not a
// dependent VALU chain of the whole wave
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// f'(j, j') = cst[j] + f(j, j')  (:153-155).  Inside the walk every j lies in the current window and every j' in [j, j'1 + w], all
// resident in the ring (k_pack_convex_win keeps [j0, j'1 + 3 w]); a pair with j > j' or j' - j > Wc -- possible only through
// stale stack entries of a non-convex cost -- takes the general oracle
template <typename TC>
__device__ __forceinline__ Ext<TC> cw_base_eval(CwState<TC> &S, const CwEnv &E, const SeqOracle<TC> *O, int32_t j, int32_t jp)
{
    // both LDS reads are issued unconditionally (the table index clamped into the row), so that the reads of several evaluations
    // of one step go out together; the rare pair outside the table replaces the value afterwards
    const uint32_t d = (uint32_t)(jp - j);
    const bool in_tab = d <= (uint32_t)E.Wc;
    const Ext<TC> b = S.cst[j & (CW_RING - 1)];
    TC fv = S.F[jp & (CW_RING - 1)][in_tab ? d : 0u];
    if (!in_tab) fv = cw_ocl_general<TC>(O, j, jp);
    Ext<TC> r; r.inf = b.inf; r.x = cadd(b.x, fv);
    return r;
}
// the sigma view (:246): position a as a candidate is the real column sigma_j[I - a], position b as a target is sigma_j'[I - b]
template <int MODE> __device__ __forceinline__ int32_t cw_xj(const int32_t *sig_j, const CwEnv &E, int32_t a) { return MODE ? uni(sig_j[E.I - a]) : a; }
template <int MODE> __device__ __forceinline__ int32_t cw_xjp(const int32_t *sig_jp, const CwEnv &E, int32_t b) { return MODE ? uni(sig_jp[E.I - b]) : b; }
template <typename TC, int MODE>
__device__ __forceinline__ Ext<TC> cw_get(CwState<TC> &S, const CwEnv &E, int32_t jp)
{
    if (MODE) return S.sig_cst[jp];
    return S.cst[jp & (CW_RING - 1)];
}
template <typename TC, int MODE>
__device__ __forceinline__ void cw_set(CwState<TC> &S, const CwEnv &E, int64_t *__restrict__ ptr_g, int32_t jp, Ext<TC> c, int32_t p)
{
    if (MODE) { S.sig_cst[jp] = c; S.sig_ptr[jp] = p; return; }
    S.cst[jp & (CW_RING - 1)] = c; ptr_g[jp] = p;
}

// chunk_convex!  ConvexTotalChunker.jl:57-112 on the LDS state (every j' visited lies in (j0, j'1]: inside the view, so the
// WindowConstrainedMatrix range checks of the general path are vacuous here).  The top of the stack is mirrored in scalars
// (tj, th): a step reads the stack only after a pop.
template <typename TC, int MODE>
__device__ __forceinline__ void cw_chunk_convex(CwState<TC> &S, const CwEnv &E, const SeqOracle<TC> *O, int64_t *__restrict__ ptr_g, int32_t j0, int32_t jp1)
{
    // (tj, th) mirror the top of the stack, rtj is the real column of tj: a step translates positions to columns once
    int32_t top = 0, tj, th, rtj;
#define CW_PUSH(a, ra, b) do { S.ftr[2 * top] = (a); S.ftr[2 * top + 1] = (b); top++; tj = (a); th = (b); rtj = (ra); } while (0)
#define CW_POP() do { top--; if (top > 0) { tj = uni(S.ftr[2 * (top - 1)]); th = uni(S.ftr[2 * (top - 1) + 1]); rtj = cw_xj<MODE>(S.sig_j, E, tj); } } while (0)
#define CW_F(rj, b) cw_base_eval<TC>(S, E, O, (rj), cw_xjp<MODE>(S.sig_jp, E, (b)))
    CW_PUSH(j0, cw_xj<MODE>(S.sig_j, E, j0), jp1 + 1);
    for (int32_t jp = j0 + 1; jp <= jp1; jp++) {
        CW_STAT(2);
        const int32_t rjm1 = cw_xj<MODE>(S.sig_j, E, jp - 1), rjp = cw_xjp<MODE>(S.sig_jp, E, jp);
        const Ext<TC> c2 = cw_base_eval<TC>(S, E, O, rjm1, rjp);           // f(j' - 1, j')  (independent of the stack: issued first)
        const Ext<TC> cur = cw_get<TC, MODE>(S, E, jp);
        const int32_t j = tj, h = th;                                       // (j, h) = last(ftr)
        const Ext<TC> c = cw_base_eval<TC>(S, E, O, rtj, rjp);             // f(j, j')
        if (ext_le(c, c2)) {
            if (ext_le(c, cur)) cw_set<TC, MODE>(S, E, ptr_g, jp, c, j);
            if (h == jp + 1) CW_POP();
        } else {
            if (ext_le(c2, cur)) cw_set<TC, MODE>(S, E, ptr_g, jp, c2, jp - 1);
            while (top > 0) {
                CW_STAT(3);
                const int32_t rh = cw_xjp<MODE>(S.sig_jp, E, th - 1);
                if (ext_lt(cw_base_eval<TC>(S, E, O, rjm1, rh), cw_base_eval<TC>(S, E, O, rtj, rh))) CW_POP(); else break;
            }
            if (top == 0) {
                CW_PUSH(jp - 1, rjm1, jp1 + 1);
            } else {
                int32_t hh = th, h_lo = jp + 1, h_hi = th - 1;
                while (h_lo <= h_hi) {
                    hh = (int32_t)(((uint32_t)(h_lo + h_hi)) >> 1);
                    CW_STAT(4);
                    const int32_t rh = cw_xjp<MODE>(S.sig_jp, E, hh - 1);
                    if (ext_lt(cw_base_eval<TC>(S, E, O, rjm1, rh), cw_base_eval<TC>(S, E, O, rtj, rh))) h_lo = hh + 1; else h_hi = hh - 1;
                }
                hh = h_hi;
                if (jp + 1 != hh) CW_PUSH(jp - 1, rjm1, hh);
            }
        }
    }
#undef CW_PUSH
#undef CW_POP
#undef CW_F
}

// where the window-table entry F[j'][d] = f(j' - d, j') comes from: the request's own table, or (batches: B requests on one
// pattern) a SHARED table of net counts that the request's model turns into a cost on the way into LDS
template <typename TC> struct CwTabF {
    const TC *__restrict__ F; int32_t FW;
    __device__ __forceinline__ TC operator()(int32_t row, int32_t d) const { return F[(int64_t)row * FW + d]; }
};
template <typename TC> struct CwTabNets {
    const int32_t *__restrict__ NT; int32_t FW; const int64_t *__restrict__ pos; DevModel<TC> M; TC alpha;
    __device__ __forceinline__ TC operator()(int32_t row, int32_t d) const
    {
        const int32_t p = row - 1 - d;                                   // 0-based first column of the part [j' - d, j')
        if (p < 0) return (TC)0;                                         // (outside the matrix: never evaluated)
        return dm_apply(M, alpha, (int64_t)d, pos[row - 1] - pos[p], (int64_t)NT[(int64_t)row * FW + d], (int64_t)0);
    }
};

template <typename TC, typename TAB>
__device__ __forceinline__ void cw_pack_body(CwState<TC> &S, const TAB Ftab_at, int32_t n, int32_t Wc, int32_t w, int64_t *__restrict__ ptr,
                                             int32_t *__restrict__ status, const SeqOracle<TC> *__restrict__ Odev)
{
    const int lane = threadIdx.x;
    const int32_t Jp1 = n + 1, FWg = Wc + 1;
    // ptr[] starts at zeros(Ti, n + 1) (:151): filled by the whole wave
    for (int64_t t = lane; t <= (int64_t)n + 1; t += 64) ptr[t] = 0;
    CwEnv E; E.n = n; E.Wc = Wc; E.mode = 0; E.I = 0;
    for (int i = lane; i < CW_RING; i += 64) S.cst[i] = ext_inf<TC>();
    __syncthreads();
    // rows 1 .. hi of the table into the ring (row j' = F[j' * (Wc + 1) + d], d = j' - j)
    int32_t hi_res = 1 + 4 * w;
    if (hi_res > Jp1) hi_res = Jp1;
    for (int32_t e = lane; e < hi_res * FWg; e += 64) {
        const int32_t row = 1 + e / FWg, d = e - (row - 1) * FWg;
        S.F[row & (CW_RING - 1)][d] = Ftab_at(row, d);
    }
    S.cst[1] = ext_of((TC)0);                                         // cst[1] = zero (:152)
    __syncthreads();
    E.lo_res = 1; E.hi_res = hi_res;
    // chunk_convex_constrained!(cst, spl, f', w, w_max, 1, n + 1, ...)  (:211-265)
    const int32_t J0 = 1;
    int32_t jp1 = J0 + 1;
    while (jp1 < Jp1 && (jp1 + 1 - J0) <= w) jp1 += 1;               // w(J0, j'1 + 1) <= w_max
    int32_t j0 = J0;
    int32_t rc = CP_OK;
    if (n >= 1) for (;;) {
        // the table rows of the columns that become visible after this iteration: (hi_res, min(n + 1, j'1 + 4 w)] -- into registers now,
        // into the ring at the end
        int32_t nx_hi = jp1 + 4 * w;
        if (nx_hi > Jp1) nx_hi = Jp1;
        const int32_t nx_cnt = nx_hi > E.hi_res ? (nx_hi - E.hi_res) * FWg : 0;
        TC pre[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int32_t e = lane + 64 * k;
            pre[k] = (TC)0;
            if (e < nx_cnt) { const int32_t ro = e / FWg, d = e - ro * FWg; pre[k] = Ftab_at(E.hi_res + 1 + ro, d); }
        }
        cw_chunk_convex<TC, 0>(S, E, Odev, ptr, j0, jp1);
        if (jp1 == Jp1) break;
        int32_t jp = jp1, I = 1;
        for (int32_t j = j0 + 1; j <= jp1; j++) {                     // (:224-238)
            if (jp > jp1) { S.sig_jp[I] = jp; I += 1; S.sig_j[I] = j; }
            while (jp < Jp1 && (jp + 1 - j) <= w) { jp += 1; S.sig_jp[I] = jp; I += 1; S.sig_j[I] = j; }
        }
        I += 1;
        if (I == 2) { rc = CP_EINVAL; break; }
        for (int32_t i = 2; i <= I - 1; i++) S.sig_cst[i] = ext_inf<TC>();
        E.I = I;
        cw_chunk_convex<TC, 1>(S, E, Odev, ptr, 1, I - 1);
        for (int32_t ip = 2; ip <= I - 1; ip++) cw_set<TC, 0>(S, E, ptr, uni(S.sig_jp[I - ip]), S.sig_cst[ip], uni(S.sig_j[I - uni(S.sig_ptr[ip])]));
        j0 = jp1;
        jp1 = S.sig_jp[I - 2];
        // the window moved: columns below the new j0 leave the ring, the prefetched rows enter it
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int32_t e = lane + 64 * k;
            if (e < nx_cnt) { const int32_t ro = e / FWg, d = e - ro * FWg, row = E.hi_res + 1 + ro; S.F[row & (CW_RING - 1)][d] = pre[k]; if (d == 0) S.cst[row & (CW_RING - 1)] = ext_inf<TC>(); }
        }
        __syncthreads();
        if (nx_hi > E.hi_res) E.hi_res = nx_hi;
        E.lo_res = j0;
    }
    if (lane == 0) *status = rc;
}

template <typename TC>
__global__ void __launch_bounds__(64) k_pack_convex_win(SeqOracle<TC> Oarg, int32_t w, int64_t *__restrict__ ptr, int32_t *__restrict__ status,
                                                        const SeqOracle<TC> *__restrict__ Odev)
{
    __shared__ CwState<TC> S;
    const CwTabF<TC> T{Oarg.Ftab, (int32_t)Oarg.Wc + 1};
    cw_pack_body<TC>(S, T, (int32_t)Oarg.n, (int32_t)Oarg.Wc, w, ptr, status, Odev);
}

// B requests (model, w_max) on ONE pattern, one wave (workgroup) each: the stack algorithm is a dependent chain that fills one wave
// however long the matrix is, so a sweep over the cost constants / widths fills the chip.  The requests share the table of net
// counts NT[j'][d] = nets(j' - d, j') (stride FWs, built once); request b applies its own model while the rows enter LDS.
template <typename TC>
struct CwReq { DevModel<TC> M; TC alpha; int32_t w, _pad; int64_t *ptr; };

template <typename TC>
__global__ void __launch_bounds__(64) k_pack_convex_win_batch(const CwReq<TC> *__restrict__ req, const SeqOracle<TC> *__restrict__ Odev, int32_t n,
                                                              const int32_t *__restrict__ NT, int32_t FWs, const int64_t *__restrict__ pos,
                                                              int32_t *__restrict__ status)
{
    __shared__ CwState<TC> S;
    const CwReq<TC> R = req[blockIdx.x];
    const CwTabNets<TC> T{NT, FWs, pos, R.M, R.alpha};
    cw_pack_body<TC>(S, T, n, 2 * R.w + 2, R.w, R.ptr, status + blockIdx.x, Odev + blockIdx.x);
}

// the shared table of a batch: nets(j' - d, j') for d <= Wc, one thread per j' (the walk of k_window_table without a model)
__global__ void __launch_bounds__(256) k_window_nets(int64_t n, int64_t Wc, const int64_t *__restrict__ pos, const int32_t *__restrict__ next,
                                                     int32_t *__restrict__ NT)
{
    const int64_t jp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (jp > n + 1) return;
    const int64_t r = jp - 1;
    int32_t *row = NT + jp * (Wc + 1);
    int32_t nn = 0;
    row[0] = 0;
    for (int64_t d = 1; d <= Wc; d++) {
        const int64_t p = r - d;
        if (p < 0) { row[d] = 0; continue; }
        for (int64_t q = pos[p]; q < pos[p + 1]; q++) nn += (next[q] >= (int32_t)r);
        row[d] = nn;
    }
}

// ------------------------------------------------------------------ ConcaveTotalChunker.jl
// chunk_concave!(cst, ptr, f, j0, j'1, ftr)  :57-114 ; ftr = CircularDeque of (j, h) pairs in dq[2*cap]
template <typename TC>
__device__ int32_t chunk_concave(const SeqOracle<TC> &O, const CView<TC> &V, const Cvx<TC> &F, int64_t j0, int64_t jp1, int64_t *dq, int64_t cap)
{
    int64_t head = 0, len = 0;
#define DQ_SLOT(i) (((head + (i)) % cap) * 2)
#define DQ_PUSH(a, b) do { int64_t s_ = DQ_SLOT(len); dq[s_] = (a); dq[s_ + 1] = (b); len++; } while (0)
    DQ_PUSH(j0, j0 + 1);
    for (int64_t jp = j0 + 1; jp <= jp1; jp++) {
        int64_t s0 = DQ_SLOT(0);
        int64_t j = dq[s0], h = dq[s0 + 1];
        Ext<TC> c = cvx_eval(O, F, j, jp);
        Ext<TC> c2 = cvx_eval(O, F, jp - 1, jp);
        if (ext_le(c2, c)) {
            if (ext_le(c2, cv_get(V, jp))) cv_set(V, jp, c2, jp - 1);
            head = 0; len = 0;
            DQ_PUSH(jp - 1, jp + 1);
        } else {
            if (ext_le(c, cv_get(V, jp))) cv_set(V, jp, c, j);
            for (;;) {                                                // :76-78
                if (len == 0) return CP_EINVAL;                       // the reference would throw on an empty deque
                int64_t sl = DQ_SLOT(len - 1);
                j = dq[sl]; h = dq[sl + 1];
                if (ext_le(cvx_eval(O, F, jp - 1, h), cvx_eval(O, F, j, h))) len--; else break;
            }
            int64_t h_lo = h + 1, h_hi = jp1;                         // :89-99
            while (h_lo <= h_hi) {
                h = (int64_t)(((uint64_t)(h_lo + h_hi)) >> 1);
                if (ext_lt(cvx_eval(O, F, j, h), cvx_eval(O, F, jp - 1, h))) h_lo = h + 1; else h_hi = h - 1;
            }
            h = h_lo;
            if (h != jp1 + 1) DQ_PUSH(jp - 1, h);
            s0 = DQ_SLOT(0);
            j = dq[s0];
            head = (head + 1) % cap; len--;                           // popfirst!
            if (len == 0 || dq[DQ_SLOT(0) + 1] != jp + 1) {           // pushfirst!(ftr, (j, j' + 1))
                head = (head + cap - 1) % cap;
                dq[2 * head] = j; dq[2 * head + 1] = jp + 1; len++;
            }
        }
    }
#undef DQ_SLOT
#undef DQ_PUSH
    return CP_OK;
}

// pack_stripe(A, ConcaveTotalChunker(f | ConstrainedCost))  :9-24
template <typename TC>
__global__ void __launch_bounds__(64) k_pack_concave(SeqOracle<TC> O, SeqWeight W, int constrained, Ext<TC> *cst, int64_t *ptr, int64_t *dq,
                                                     int32_t *status)
{
    int64_t n = O.n;
    for (int64_t t = 0; t <= n + 1; t++) { cst[t] = ext_inf<TC>(); ptr[t] = 0; }
    cst[1] = ext_of((TC)0);
    Cvx<TC> F; F.mode = 0; F.k = 0; F.base = cst; F.blo = 1; F.bhi = n + 1; F.inner = nullptr; F.sig_j = nullptr; F.sig_jp = nullptr; F.I = 0;
    F.w = constrained ? &W : nullptr;
    CView<TC> V; V.cst = cst; V.ptr = ptr; V.lo = 1; V.hi = n + 1;
    int32_t rc = chunk_concave(O, V, F, (int64_t)1, n + 1, dq, n + 1);
    if (threadIdx.x == 0) *status = rc;
}

// partition_stripe(A, K, ConcaveTotalSplitter(..))  :26-55, :140-180.  cst/ptr: K layers of (n+2) slots.
template <typename TC>
__global__ void __launch_bounds__(64) k_partition_concave(SeqOracle<TC> O, SeqWeight W, int constrained, int64_t K, Ext<TC> *cst, int64_t *ptr,
                                                          int64_t *jlo, int64_t *jhi, int64_t *dq, int64_t *spl, int32_t *status);

// column_constraints  DynamicSplitter.jl:144-172
__device__ void column_constraints(int64_t n, int64_t K, const SeqWeight &W, int64_t *jlo, int64_t *jhi)
{
    int64_t jp = n + 1;
    for (int64_t k = K; k >= 1; k--) {
        A1(jlo, k) = jp;
        int64_t j = jp;
        while (j - 1 >= 1 && wgt_le(W, j - 1, jp)) j -= 1;
        jp = j;
    }
    int64_t j = 1;
    for (int64_t k = 1; k <= K; k++) {
        jp = j;
        while (jp + 1 <= n + 1 && wgt_le(W, j, jp + 1)) jp += 1;
        A1(jhi, k) = jp;
        j = jp;
    }
}

// part_constraints  DynamicSplitter.jl:174-204
__device__ void part_constraints(int64_t n, int64_t K, const SeqWeight &W, int64_t *klo, int64_t *khi)
{
    for (int64_t t = 1; t <= n + 1; t++) { A1(klo, t) = 0; A1(khi, t) = 0; }
    int64_t jp = n + 1;
    A1(khi, n + 1) = K;
    for (int64_t k = K; k >= 1; k--) {
        int64_t j = jp;
        while (j - 1 >= 1 && wgt_le(W, j - 1, jp)) { j -= 1; A1(khi, j) = k - 1; }
        jp = j;
    }
    int64_t j = 1;
    A1(klo, 1) = 1;
    for (int64_t k = 1; k <= K; k++) {
        jp = j;
        while (jp + 1 <= n + 1 && wgt_le(W, j, jp + 1)) { jp += 1; A1(klo, jp) = k; }
        j = jp;
    }
}

// partition_stripe(A, K, ConvexTotalSplitter(..))  :26-55, :170-209.  cst/ptr: K layers of (n+2) slots.
template <typename TC>
__global__ void __launch_bounds__(64) k_partition_convex(SeqOracle<TC> O, SeqWeight W, int constrained, int64_t K, Ext<TC> *cst, int64_t *ptr,
                                                         int64_t *jlo, int64_t *jhi, int64_t *ftr, int64_t *sig_j, int64_t *sig_jp,
                                                         int64_t *sig_ptr, Ext<TC> *sig_cst, int64_t *spl, int32_t *status)
{
    int64_t n = O.n, ld = n + 2;
    if (constrained) {
        column_constraints(n, K, W, jlo, jhi);
        if (A1(jhi, K) < n + 1) {
            for (int64_t k = 1; k <= K + 1; k++) A1(spl, k) = 1;
            A1(spl, K + 1) = n + 1;
            if (threadIdx.x == 0) *status = CP_INFEASIBLE;
            return;
        }
    } else {
        for (int64_t k = 1; k <= K; k++) { A1(jlo, k) = 1; A1(jhi, k) = n + 1; }
    }
    for (int64_t t = 0; t < ld * K; t++) { cst[t] = ext_inf<TC>(); ptr[t] = 0; }
    for (int64_t jp = A1(jlo, 1); jp <= A1(jhi, 1); jp++) { cst[jp] = ext_of(ocl(O, 1, jp, 1)); ptr[jp] = 1; }
    int32_t rc = CP_OK;
    for (int64_t k = 2; k <= K && rc == CP_OK; k++) {
        Cvx<TC> F; F.mode = 0; F.k = k; F.base = cst + (k - 2) * ld; F.blo = A1(jlo, k - 1); F.bhi = A1(jhi, k - 1);
        F.inner = nullptr; F.sig_j = nullptr; F.sig_jp = nullptr; F.I = 0; F.w = nullptr;
        CView<TC> V; V.cst = cst + (k - 1) * ld; V.ptr = ptr + (k - 1) * ld; V.lo = A1(jlo, k); V.hi = A1(jhi, k);
        for (int64_t jp = A1(jlo, k); jp <= A1(jhi, k); jp++) cv_set(V, jp, cvx_eval(O, F, jp, jp), jp);
        if (!constrained) chunk_convex(O, V, F, (int64_t)1, n + 1, ftr);
        else rc = chunk_convex_constrained(O, V, F, W, A1(jlo, k - 1), A1(jhi, k), ftr, sig_j, sig_jp, sig_ptr, sig_cst);
    }
    if (rc == CP_OK) {
        A1(spl, K + 1) = n + 1;
        for (int64_t k = K; k >= 1; k--) {
            int64_t jp = A1(spl, k + 1);
            A1(spl, k) = (A1(jlo, k) <= jp && jp <= A1(jhi, k)) ? ptr[(k - 1) * ld + jp] : 0;
        }
    }
    if (threadIdx.x == 0) *status = rc;
}

template <typename TC>
__global__ void __launch_bounds__(64) k_partition_concave(SeqOracle<TC> O, SeqWeight W, int constrained, int64_t K, Ext<TC> *cst, int64_t *ptr,
                                                          int64_t *jlo, int64_t *jhi, int64_t *dq, int64_t *spl, int32_t *status)
{
    int64_t n = O.n, ld = n + 2;
    if (constrained) {
        column_constraints(n, K, W, jlo, jhi);                         // :150
        if (A1(jhi, K) < n + 1) {                                      // :152-157
            for (int64_t k = 1; k <= K + 1; k++) A1(spl, k) = 1;
            A1(spl, K + 1) = n + 1;
            if (threadIdx.x == 0) *status = CP_INFEASIBLE;
            return;
        }
    } else {
        for (int64_t k = 1; k <= K; k++) { A1(jlo, k) = 1; A1(jhi, k) = n + 1; }
    }
    for (int64_t t = 0; t < ld * K; t++) { cst[t] = ext_inf<TC>(); ptr[t] = 0; }
    for (int64_t jp = A1(jlo, 1); jp <= A1(jhi, 1); jp++) { cst[jp] = ext_of(ocl(O, 1, jp, 1)); ptr[jp] = 1; }
    int32_t rc = CP_OK;
    for (int64_t k = 2; k <= K && rc == CP_OK; k++) {
        Cvx<TC> F; F.mode = 0; F.k = k; F.base = cst + (k - 2) * ld; F.blo = A1(jlo, k - 1); F.bhi = A1(jhi, k - 1);
        F.inner = nullptr; F.sig_j = nullptr; F.sig_jp = nullptr; F.I = 0; F.w = nullptr;
        CView<TC> V; V.cst = cst + (k - 1) * ld; V.ptr = ptr + (k - 1) * ld; V.lo = A1(jlo, k); V.hi = A1(jhi, k);
        for (int64_t jp = A1(jlo, k); jp <= A1(jhi, k); jp++) cv_set(V, jp, cvx_eval(O, F, jp, jp), jp);
        if (!constrained) rc = chunk_concave(O, V, F, (int64_t)1, n + 1, dq, n + 1);
        else rc = chunk_concave(O, V, F, A1(jlo, k - 1), A1(jhi, k), dq, n + 1);                       // :176
    }
    if (rc == CP_OK) {
        A1(spl, K + 1) = n + 1;
        for (int64_t k = K; k >= 1; k--) {
            int64_t jp = A1(spl, k + 1);
            A1(spl, k) = (A1(jlo, k) <= jp && jp <= A1(jhi, k)) ? ptr[(k - 1) * ld + jp] : 0;
        }
    }
    if (threadIdx.x == 0) *status = rc;
}

// ------------------------------------------------------------------ constrained K-part DPs (DynamicSplitter.jl:206-314)
// window-constrained tables stored ragged like WindowConstrainedMatrix: val[pos[c] + i - lo[c]]
template <typename TC>
__global__ void __launch_bounds__(64) k_dyn_constrained(SeqOracle<TC> O, SeqWeight W, int64_t K, int32_t g, int32_t order,
                                                        int64_t *lo, int64_t *hi, int64_t *wpos, TC *cv, int64_t *pv, int64_t cap,
                                                        int64_t *spl, int32_t *status)
{
    int64_t n = O.n;
    const TC TMAX = CostTraits<TC>::typemax();
    if (order == CP_ORDER_SPLITTER) {                       // :206-258 ; columns of the window matrix = parts k
        column_constraints(n, K, W, lo, hi);
        if (A1(hi, K) < n + 1) {
            for (int64_t k = 1; k <= K + 1; k++) A1(spl, k) = 1;
            A1(spl, K + 1) = n + 1;
            if (threadIdx.x == 0) *status = CP_INFEASIBLE;
            return;
        }
        A1(wpos, 1) = 1;
        for (int64_t k = 1; k <= K; k++) A1(wpos, k + 1) = A1(wpos, k) + A1(hi, k) - A1(lo, k) + 1;
        if (A1(wpos, K + 1) - 1 > cap) { if (threadIdx.x == 0) *status = CP_EUNSUPPORTED; return; }
#define IN(i, c) (A1(lo, c) <= (i) && (i) <= A1(hi, c))
#define AT(i, c) (A1(wpos, c) + (i) - A1(lo, c))
#define CGET(i, c) (IN(i, c) ? A1(cv, AT(i, c)) : TMAX)
#define CSET(i, c, v) do { if (IN(i, c)) A1(cv, AT(i, c)) = (v); } while (0)
#define PGET(i, c) (IN(i, c) ? A1(pv, AT(i, c)) : (int64_t)0)
#define PSET(i, c, v) do { if (IN(i, c)) A1(pv, AT(i, c)) = (v); } while (0)
        for (int64_t jp = A1(lo, 1); jp <= A1(hi, 1); jp++) { CSET(jp, 1, ocl(O, 1, jp, 1)); PSET(jp, 1, 1); }
        for (int64_t k = 2; k <= K; k++) {
            int64_t j0 = A1(lo, k - 1);
            for (int64_t jp = A1(lo, k); jp <= A1(hi, k); jp++) {
                while (wgt_gt(W, j0, jp)) j0 += 1;
                CSET(jp, k, comb(g, CGET(j0, k - 1), ocl(O, j0, jp, k)));
                PSET(jp, k, j0);
                int64_t jend = jp < A1(hi, k - 1) ? jp : A1(hi, k - 1);
                for (int64_t j = j0 + 1; j <= jend; j++) {
                    TC c2 = comb(g, CGET(j, k - 1), ocl(O, j, jp, k));
                    if (c2 <= CGET(jp, k)) { CSET(jp, k, c2); PSET(jp, k, j); }
                }
            }
        }
        A1(spl, K + 1) = n + 1;
        for (int64_t k = K; k >= 1; k--) A1(spl, k) = PGET(A1(spl, k + 1), k);
    } else {                                                // :260-314 ; columns of the window matrix = positions j'
        part_constraints(n, K, W, lo, hi);
        if (A1(lo, n + 1) == 0) {
            for (int64_t k = 1; k <= K + 1; k++) A1(spl, k) = 1;
            A1(spl, K + 1) = n + 1;
            if (threadIdx.x == 0) *status = CP_INFEASIBLE;
            return;
        }
        A1(wpos, 1) = 1;
        for (int64_t c = 1; c <= n + 1; c++) A1(wpos, c + 1) = A1(wpos, c) + A1(hi, c) - A1(lo, c) + 1;
        if (A1(wpos, n + 2) - 1 > cap) { if (threadIdx.x == 0) *status = CP_EUNSUPPORTED; return; }
        int64_t j0 = 1;
        for (int64_t jp = 1; jp <= n + 1; jp++) {
            while (wgt_gt(W, j0, jp)) j0 += 1;
            if (!(j0 <= jp)) { if (threadIdx.x == 0) *status = CP_EINVAL; return; }
            TC dc = ocl(O, j0, jp, 0);
            if (j0 == 1) { CSET(1, jp, dc); PSET(1, jp, 1); }
            int64_t ka = (A1(lo, j0) + 1 > A1(lo, jp)) ? A1(lo, j0) + 1 : A1(lo, jp);
            int64_t kb = (A1(hi, j0) + 1 < A1(hi, jp)) ? A1(hi, j0) + 1 : A1(hi, jp);
            for (int64_t k = ka; k <= kb; k++) { CSET(k, jp, comb(g, CGET(k - 1, j0), dc)); PSET(k, jp, j0); }
            for (int64_t j = j0 + 1; j <= jp; j++) {
                dc = ocl(O, j, jp, 0);
                ka = (A1(lo, j) + 1 > A1(lo, jp)) ? A1(lo, j) + 1 : A1(lo, jp);
                kb = (A1(hi, j) + 1 < A1(hi, jp)) ? A1(hi, j) + 1 : A1(hi, jp);
                for (int64_t k = ka; k <= kb; k++) {
                    TC c2 = comb(g, CGET(k - 1, j), dc);
                    if (c2 <= CGET(k, jp)) { CSET(k, jp, c2); PSET(k, jp, j); }
                }
            }
        }
        A1(spl, K + 1) = n + 1;
        for (int64_t k = K; k >= 1; k--) A1(spl, k) = PGET(k, A1(spl, k + 1));
#undef IN
#undef AT
#undef CGET
#undef CSET
#undef PGET
#undef PSET
    }
    if (threadIdx.x == 0) *status = CP_OK;
}

// ------------------------------------------------------------------ host side: oracle assembly
template <typename TC>
struct SeqCtx {
    HostModel<TC> HM;
    SeqOracle<TC> O;
    WaveletHost net, self;
    DBuf<int32_t> asg;
    DBuf<int64_t> pspl, hst, cursor;
    DBuf<TC> Delta, dvec, brtab[CP_MAX_R], ftab;
};

template <typename TC>
static void seq_oracle(cp_csr_s *A, const cp_model_t *mdl, const cp_rowpart_t *Pi, SeqCtx<TC> &C)
{
    hipStream_t s = A->stream;
    build_dev_model<TC>(mdl, C.HM, s);
    memset(&C.O, 0, sizeof(C.O));
    C.O.M = C.HM.d; C.O.pos = A->pos.p; C.O.row = A->row.p; C.O.n = A->n; C.O.m = A->m;
    if (mdl->kind == CP_MODEL_CONNECTIVITY || mdl->kind == CP_MODEL_COLBLOCK || mdl->kind == CP_MODEL_HYPEREDGE_CUT) {
        ensure_net_counter(A, C.net); C.O.has_net = 1; C.O.net = C.net.d;
    }
    if (mdl->kind == CP_MODEL_HYPEREDGE_CUT) {
        ensure_selfnet_counter(A, C.self); C.O.has_self = 1; C.O.self = C.self.d; C.O.lpos = A->lpos.p;
    }
    if (mdl->kind == CP_MODEL_BLOCK) {
        CP_REQUIRE(Pi && Pi->spl && Pi->K >= 1, CP_EINVAL, "BlockComponentCostModel needs a SplitPartition of the rows");
        CP_REQUIRE(Pi->spl[0] == 1 && Pi->spl[Pi->K] == A->m + 1, CP_EINVAL, "row partition does not cover 1:m");
        int64_t K = Pi->K, m = A->m;
        std::vector<int32_t> asg((size_t)(m > 0 ? m : 1));
        for (int64_t k = 1; k <= K; k++) for (int64_t i = Pi->spl[k - 1]; i <= Pi->spl[k] - 1; i++) asg[(size_t)i - 1] = (int32_t)k;
        C.asg.alloc(asg.size());
        CP_HIP(hipMemcpyAsync(C.asg.p, asg.data(), sizeof(int32_t) * asg.size(), hipMemcpyHostToDevice, s));
        C.pspl.alloc((size_t)K + 1);
        CP_HIP(hipMemcpyAsync(C.pspl.p, Pi->spl, sizeof(int64_t) * (size_t)(K + 1), hipMemcpyHostToDevice, s));
        std::vector<int64_t> ones((size_t)K, 1);
        C.hst.alloc((size_t)K);
        CP_HIP(hipMemcpyAsync(C.hst.p, ones.data(), sizeof(int64_t) * (size_t)K, hipMemcpyHostToDevice, s));
        int64_t cur[2] = {1, 1};
        C.cursor.alloc(2);
        CP_HIP(hipMemcpyAsync(C.cursor.p, cur, sizeof(cur), hipMemcpyHostToDevice, s));
        int R = mdl->R > 0 ? mdl->R : 1;
        C.Delta.alloc((size_t)R * (size_t)(A->n + 1)); C.dvec.alloc((size_t)R);
        CP_HIP(hipMemsetAsync(C.Delta.p, 0, C.Delta.bytes(), s));
        CP_HIP(hipMemsetAsync(C.dvec.p, 0, C.dvec.bytes(), s));
        C.O.P_asg = C.asg.p; C.O.P_spl = C.pspl.p; C.O.Kr = K; C.O.hst = C.hst.p; C.O.Delta = C.Delta.p; C.O.d = C.dvec.p; C.O.cursor = C.cursor.p;
        for (int r = 0; r < mdl->R && r < CP_MAX_R; r++) {
            const cp_component_t &c = mdl->beta_row[r];
            C.O.br_const[r] = c.is_const; C.O.br_c[r] = comp_const<TC>(c); C.O.br_tab[r] = nullptr; C.O.br_len[r] = 0; C.O.br_lo[r] = c.lo;
            if (!c.is_const && c.table && c.len > 0) {
                C.brtab[r].alloc((size_t)c.len);
                CP_HIP(hipMemcpyAsync(C.brtab[r].p, c.table, sizeof(TC) * (size_t)c.len, hipMemcpyHostToDevice, s));
                C.O.br_tab[r] = C.brtab[r].p; C.O.br_len[r] = c.len;
            }
        }
        CP_HIP(hipStreamSynchronize(s));      // host staging vectors die at scope end
    }
}

// pairs inside a VertexCount window come from a table built in one parallel pass over the link arrays
template <typename TC>
static void seq_window_table(cp_csr_s *A, const cp_model_t *mdl, const cp_model_t *w, int64_t wmax, SeqCtx<TC> &C, int64_t span = 1)
{
    if (!w || w->kind != CP_MODEL_VERTEX_COUNT || mdl->kind == CP_MODEL_BLOCK || mdl->alpha_k || wmax < 1) return;
    // span 2: the convex chunker's staircase sub-problems pair columns up to two windows apart (ConvexTotalChunker.jl:246)
    int64_t n = A->n, Wc = span * wmax + (span > 1 ? 2 : 0);
    if ((double)(n + 2) * (double)(Wc + 1) > 6e8) return;
    hipStream_t s = A->stream;
    ensure_links(A);
    if (mdl->kind == CP_MODEL_HYPEREDGE_CUT) ensure_self(A);
    C.ftab.alloc((size_t)(n + 2) * (size_t)(Wc + 1));
    ProfScope ps(PROF_CHUNK, s, 0.0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_window_table<TC>), dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, C.HM.d, n, Wc, A->pos.p,
                       A->next.p, A->fpos.p, A->flast.p, C.ftab.p);
    CP_HIP(hipGetLastError());
    C.O.Ftab = C.ftab.p; C.O.Wc = Wc;
}

static SeqWeight make_weight(cp_csr_s *A, const cp_model_t *w, int64_t wi, double wf)
{
    SeqWeight W; memset(&W, 0, sizeof(W));
    W.kind = w ? w->kind : CP_MODEL_FEASIBLE;
    W.pos = A->pos.p; W.wmax_i = wi; W.wmax_f = wf;
    if (w) { W.dtype = w->dtype; for (int i = 0; i < 3; i++) { W.pi[i] = w->p_i64[i]; W.pf[i] = w->p_f64[i]; } }
    return W;
}
static bool weight_ok(const cp_model_t *w)
{
    return !w || w->kind == CP_MODEL_FEASIBLE || w->kind == CP_MODEL_VERTEX_COUNT || (w->kind == CP_MODEL_WORK && !w->alpha_k);
}

// unravel_chunks!(spl, n)  DynamicChunker.jl:58-75 (host: K dependent reads of a host copy)
static int64_t unravel_chunks_host(std::vector<int64_t> &spl, int64_t n)
{
    int64_t K = 0, jp = n + 1, len = n + 1;
    while (jp != 1) { int64_t j = spl[(size_t)jp - 1]; spl[(size_t)(len - K) - 1] = jp; K += 1; jp = j; }
    spl[0] = 1;
    for (int64_t k = 1; k <= K; k++) spl[(size_t)k] = spl[(size_t)(len - K + k) - 1];
    return K;
}

template <typename TC>
int32_t run_pack_dynamic(cp_csr_s *A, const cp_model_t *mdl, const cp_rowpart_t *Pi, const cp_model_t *w, int64_t wi, double wf,
                         int64_t *spl_out, int64_t *K_out)
{
    hipStream_t s = A->stream;
    int64_t n = A->n;
    std::unique_ptr<SeqCtx<TC>> C(new SeqCtx<TC>());
    seq_oracle<TC>(A, mdl, Pi, *C);
    seq_window_table<TC>(A, mdl, w, wi, *C);
    SeqWeight W = make_weight(A, w, wi, wf);
    DBuf<TC> cst((size_t)n + 2);
    DBuf<int64_t> spl((size_t)n + 2);
    DBuf<int32_t> st(1);
    CP_HIP(hipMemsetAsync(st.p, 0, sizeof(int32_t), s));
    CP_HIP(hipMemsetAsync(spl.p, 0, spl.bytes(), s));
    {
        ProfScope ps(PROF_CHUNK, s, 0.0);
        // width-windowed costs whose sums are exact: parallel (min,+) scan (chunk_scan.hip); tables of block-component
        // models hold tabulated component values, which the host marshals as exact integers too
        bool scanned = false;
        bool exact = std::is_same<TC, int64_t>::value || (model_exact_on(mdl, n, A->N, n + 1) && mdl->kind != CP_MODEL_COLBLOCK);
        if (W.kind == CP_MODEL_VERTEX_COUNT && C->O.Ftab && C->O.Wc == wi && exact && !g_opt_force_brute)
            scanned = pack_dynamic_scan<TC>(s, n, wi, C->O.Ftab, cst.p, spl.p);
        if (!scanned)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pack_dynamic<TC>), dim3(1), dim3(64), 0, s, C->O, W, cst.p, spl.p, st.p);
    }
    CP_HIP(hipGetLastError());
    int32_t rc = 0;
    std::vector<int64_t> h((size_t)n + 1);
    CP_HIP(hipMemcpyAsync(&rc, st.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipMemcpyAsync(h.data(), spl.p, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    if (rc != CP_OK) { set_error("pack_stripe: a single column exceeds w_max (@assert j0 < j')"); return rc; }
    int64_t K = unravel_chunks_host(h, n);
    for (int64_t k = 0; k <= K; k++) spl_out[k] = h[(size_t)k];
    *K_out = K;
    return CP_OK;
}

template <typename TC>
int32_t run_pack_convex(cp_csr_s *A, const cp_model_t *mdl, const cp_rowpart_t *Pi, const cp_model_t *w, int64_t wi, double wf,
                        int64_t *spl_out, int64_t *K_out)
{
    hipStream_t s = A->stream;
    int64_t n = A->n, cap = 2 * n + 4;
    std::unique_ptr<SeqCtx<TC>> C(new SeqCtx<TC>());
    seq_oracle<TC>(A, mdl, Pi, *C);
    seq_window_table<TC>(A, mdl, w, wi, *C, 2);
    SeqWeight W = make_weight(A, w, wi, wf);
    int constrained = W.kind != CP_MODEL_FEASIBLE;
    DBuf<Ext<TC>> cst((size_t)n + 2), sig_cst((size_t)cap);
    DBuf<int64_t> ptr((size_t)n + 2), ftr((size_t)(2 * cap)), sig_j((size_t)cap), sig_jp((size_t)cap), sig_ptr((size_t)cap);
    DBuf<int32_t> st(1);
    CP_HIP(hipMemsetAsync(st.p, 0, sizeof(int32_t), s));
    CP_HIP(hipMemsetAsync(sig_ptr.p, 0, sig_ptr.bytes(), s));
    {
        ProfScope ps(PROF_CHUNK, s, 0.0);
        // width window with its table resident: the LDS form of the same algorithm (k_pack_convex_win)
        const bool win = W.kind == CP_MODEL_VERTEX_COUNT && C->O.Ftab && wi >= 1 && wi <= CW_MAXW && C->O.Wc == 2 * wi + 2 && !g_opt_force_brute &&
                         mdl->kind != CP_MODEL_BLOCK;
        DBuf<SeqOracle<TC>> odev(1);                      // (the general oracle, for the rare pair outside the resident table)
        if (win) {
            CP_HIP(hipMemcpyAsync(odev.p, &C->O, sizeof(SeqOracle<TC>), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pack_convex_win<TC>), dim3(1), dim3(64), 0, s, C->O, (int32_t)wi, ptr.p, st.p, odev.p);
            CP_HIP(hipStreamSynchronize(s));              // (odev leaves scope)
#ifdef CW_STATS
            unsigned long long st8[8];
            CP_HIP(hipMemcpyFromSymbol(st8, HIP_SYMBOL(g_cw_stats), sizeof(st8)));
            fprintf(stderr, "cw stats: general(j>j')=%llu general(other)=%llu steps=%llu pops=%llu bsearch=%llu\n", st8[0], st8[1], st8[2], st8[3], st8[4]);
#endif
        }
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pack_convex<TC>), dim3(1), dim3(64), 0, s, C->O, W, constrained, cst.p, ptr.p, ftr.p,
                           sig_j.p, sig_jp.p, sig_ptr.p, sig_cst.p, st.p);
    }
    CP_HIP(hipGetLastError());
    int32_t rc = 0;
    std::vector<int64_t> h((size_t)n + 2);
    CP_HIP(hipMemcpyAsync(&rc, st.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipMemcpyAsync(h.data(), ptr.p, sizeof(int64_t) * (size_t)(n + 2), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    if (rc != CP_OK) { set_error("ConvexTotalChunker: a single column exceeds w_max"); return rc; }
    std::vector<int64_t> sp((size_t)n + 1);
    for (int64_t jp = 1; jp <= n + 1; jp++) sp[(size_t)jp - 1] = h[(size_t)jp];
    int64_t K = unravel_chunks_host(sp, n);
    for (int64_t k = 0; k <= K; k++) spl_out[k] = sp[(size_t)k];
    *K_out = K;
    return CP_OK;
}

// pack_stripe(A, ConvexTotalChunker(ConstrainedCost(f_b, VertexCount(), w_b))) for B requests on one pattern: ONE launch, the net
// counter and the window table of net counts built once.  Every request must be one the LDS kernel takes on its own (width weight,
// 1 <= w <= CW_MAXW, a net-count model without per-part alpha); the results are those of B single calls.
template <typename TC>
int32_t run_pack_convex_batch(cp_csr_s *A, int64_t B, const cp_model_t *models, const int64_t *wmax, int64_t ld, int64_t *spl_out, int64_t *K_out)
{
    hipStream_t s = A->stream;
    const int64_t n = A->n;
    int64_t wall = 1;
    for (int64_t b = 0; b < B; b++) wall = std::max(wall, wmax[b]);
    const int64_t Wc = 2 * wall + 2, FWs = Wc + 1;
    ensure_links(A);
    std::vector<std::unique_ptr<SeqCtx<TC>>> C((size_t)B);
    std::vector<SeqOracle<TC>> ho((size_t)B);
    std::vector<CwReq<TC>> hr((size_t)B);
    DBuf<int64_t> ptr((size_t)B * (size_t)(n + 2));
    WaveletHost net;                                          // (the general oracle of the rare pair outside the table: one counter for all)
    ensure_net_counter(A, net);
    for (int64_t b = 0; b < B; b++) {
        C[(size_t)b].reset(new SeqCtx<TC>());
        SeqCtx<TC> &c = *C[(size_t)b];
        build_dev_model<TC>(models + b, c.HM, s);
        memset(&c.O, 0, sizeof(c.O));
        c.O.M = c.HM.d; c.O.pos = A->pos.p; c.O.row = A->row.p; c.O.n = n; c.O.m = A->m;
        c.O.has_net = 1; c.O.net = net.d;
        ho[(size_t)b] = c.O;
        CwReq<TC> &R = hr[(size_t)b];
        memset(&R, 0, sizeof(R));
        R.M = c.HM.d; R.alpha = model_param<TC>(models + b, CP_P_ALPHA); R.w = (int32_t)wmax[b]; R.ptr = ptr.p + (size_t)b * (size_t)(n + 2);
    }
    DBuf<int32_t> NT((size_t)(n + 2) * (size_t)FWs), st((size_t)B);
    DBuf<SeqOracle<TC>> odev((size_t)B);
    DBuf<CwReq<TC>> dreq((size_t)B);
    CP_HIP(hipMemsetAsync(st.p, 0, st.bytes(), s));
    CP_HIP(hipMemcpyAsync(odev.p, ho.data(), sizeof(SeqOracle<TC>) * (size_t)B, hipMemcpyHostToDevice, s));
    CP_HIP(hipMemcpyAsync(dreq.p, hr.data(), sizeof(CwReq<TC>) * (size_t)B, hipMemcpyHostToDevice, s));
    {
        ProfScope ps(PROF_CHUNK, s, 0.0);
        hipLaunchKernelGGL(k_window_nets, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, n, Wc, A->pos.p, A->next.p, NT.p);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pack_convex_win_batch<TC>), dim3((unsigned)B), dim3(64), 0, s, dreq.p, odev.p, (int32_t)n, NT.p, (int32_t)FWs,
                           A->pos.p, st.p);
    }
    CP_HIP(hipGetLastError());
    std::vector<int32_t> hst((size_t)B);
    std::vector<int64_t> h((size_t)n + 2), sp((size_t)n + 1);
    CP_HIP(hipMemcpyAsync(hst.data(), st.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    for (int64_t b = 0; b < B; b++) {
        if (hst[(size_t)b] != CP_OK) { set_error("ConvexTotalChunker: a single column exceeds w_max"); return hst[(size_t)b]; }
        CP_HIP(hipMemcpy(h.data(), ptr.p + (size_t)b * (size_t)(n + 2), sizeof(int64_t) * (size_t)(n + 2), hipMemcpyDeviceToHost));
        for (int64_t jp = 1; jp <= n + 1; jp++) sp[(size_t)jp - 1] = h[(size_t)jp];
        const int64_t K = unravel_chunks_host(sp, n);
        CP_REQUIRE(K + 1 <= ld, CP_EINVAL, "ld is smaller than a request's number of chunks + 1 (n + 1 always suffices)");
        for (int64_t k = 0; k <= K; k++) spl_out[(size_t)b * (size_t)ld + (size_t)k] = sp[(size_t)k];
        K_out[b] = K;
    }
    return CP_OK;
}

template <typename TC>
int32_t run_partition_convex(cp_csr_s *A, int64_t K, const cp_model_t *mdl, const cp_rowpart_t *Pi, const cp_model_t *w, int64_t wi,
                             double wf, int64_t *spl_out)
{
    hipStream_t s = A->stream;
    int64_t n = A->n, cap = 2 * n + 4;
    SeqWeight W = make_weight(A, w, wi, wf);
    int constrained = W.kind != CP_MODEL_FEASIBLE;
    if (!constrained && K == 1) { spl_out[0] = 1; spl_out[1] = n + 1; return CP_OK; }         // :32-34
    CP_REQUIRE((double)K * (double)(n + 2) < 4e8, CP_EUNSUPPORTED, "ConvexTotalSplitter tables exceed the device budget");
    std::unique_ptr<SeqCtx<TC>> C(new SeqCtx<TC>());
    seq_oracle<TC>(A, mdl, Pi, *C);
    DBuf<Ext<TC>> cst((size_t)K * (size_t)(n + 2)), sig_cst((size_t)cap);
    DBuf<int64_t> ptr((size_t)K * (size_t)(n + 2)), jlo((size_t)K), jhi((size_t)K), ftr((size_t)(2 * cap)), sig_j((size_t)cap),
        sig_jp((size_t)cap), sig_ptr((size_t)cap), spl((size_t)K + 1);
    DBuf<int32_t> st(1);
    CP_HIP(hipMemsetAsync(st.p, 0, sizeof(int32_t), s));
    CP_HIP(hipMemsetAsync(sig_ptr.p, 0, sig_ptr.bytes(), s));
    {
        ProfScope ps(PROF_CHUNK, s, 0.0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_partition_convex<TC>), dim3(1), dim3(64), 0, s, C->O, W, constrained, K, cst.p, ptr.p, jlo.p,
                           jhi.p, ftr.p, sig_j.p, sig_jp.p, sig_ptr.p, sig_cst.p, spl.p, st.p);
    }
    CP_HIP(hipGetLastError());
    int32_t rc = 0;
    CP_HIP(hipMemcpyAsync(&rc, st.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipMemcpyAsync(spl_out, spl.p, sizeof(int64_t) * (size_t)(K + 1), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    if (rc == CP_EINVAL) set_error("ConvexTotalSplitter: a single column exceeds w_max");
    return rc;
}

template <typename TC>
int32_t run_pack_concave(cp_csr_s *A, const cp_model_t *mdl, const cp_rowpart_t *Pi, const cp_model_t *w, int64_t wi, double wf,
                         int64_t *spl_out, int64_t *K_out)
{
    hipStream_t s = A->stream;
    int64_t n = A->n;
    std::unique_ptr<SeqCtx<TC>> C(new SeqCtx<TC>());
    seq_oracle<TC>(A, mdl, Pi, *C);
    SeqWeight W = make_weight(A, w, wi, wf);
    int constrained = W.kind != CP_MODEL_FEASIBLE;
    DBuf<Ext<TC>> cst((size_t)n + 2);
    DBuf<int64_t> ptr((size_t)n + 2), dq((size_t)(2 * (n + 2)));
    DBuf<int32_t> st(1);
    CP_HIP(hipMemsetAsync(st.p, 0, sizeof(int32_t), s));
    {
        ProfScope ps(PROF_CHUNK, s, 0.0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pack_concave<TC>), dim3(1), dim3(64), 0, s, C->O, W, constrained, cst.p, ptr.p, dq.p, st.p);
    }
    CP_HIP(hipGetLastError());
    int32_t rc = 0;
    std::vector<int64_t> h((size_t)n + 2);
    CP_HIP(hipMemcpyAsync(&rc, st.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipMemcpyAsync(h.data(), ptr.p, sizeof(int64_t) * (size_t)(n + 2), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    if (rc != CP_OK) { set_error("ConcaveTotalChunker: the candidate deque ran empty (the reference throws)"); return rc; }
    std::vector<int64_t> sp((size_t)n + 1);
    for (int64_t jp = 1; jp <= n + 1; jp++) sp[(size_t)jp - 1] = h[(size_t)jp];
    int64_t K = unravel_chunks_host(sp, n);
    for (int64_t k = 0; k <= K; k++) spl_out[k] = sp[(size_t)k];
    *K_out = K;
    return CP_OK;
}

template <typename TC>
int32_t run_partition_concave(cp_csr_s *A, int64_t K, const cp_model_t *mdl, const cp_rowpart_t *Pi, const cp_model_t *w, int64_t wi,
                              double wf, int64_t *spl_out)
{
    hipStream_t s = A->stream;
    int64_t n = A->n;
    SeqWeight W = make_weight(A, w, wi, wf);
    int constrained = W.kind != CP_MODEL_FEASIBLE;
    if (!constrained && K == 1) { spl_out[0] = 1; spl_out[1] = n + 1; return CP_OK; }         // :32-34
    CP_REQUIRE((double)K * (double)(n + 2) < 4e8, CP_EUNSUPPORTED, "ConcaveTotalSplitter tables exceed the device budget");
    std::unique_ptr<SeqCtx<TC>> C(new SeqCtx<TC>());
    seq_oracle<TC>(A, mdl, Pi, *C);
    DBuf<Ext<TC>> cst((size_t)K * (size_t)(n + 2));
    DBuf<int64_t> ptr((size_t)K * (size_t)(n + 2)), jlo((size_t)K), jhi((size_t)K), dq((size_t)(2 * (n + 2))), spl((size_t)K + 1);
    DBuf<int32_t> st(1);
    CP_HIP(hipMemsetAsync(st.p, 0, sizeof(int32_t), s));
    {
        ProfScope ps(PROF_CHUNK, s, 0.0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_partition_concave<TC>), dim3(1), dim3(64), 0, s, C->O, W, constrained, K, cst.p, ptr.p, jlo.p,
                           jhi.p, dq.p, spl.p, st.p);
    }
    CP_HIP(hipGetLastError());
    int32_t rc = 0;
    CP_HIP(hipMemcpyAsync(&rc, st.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipMemcpyAsync(spl_out, spl.p, sizeof(int64_t) * (size_t)(K + 1), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    if (rc == CP_EINVAL) set_error("ConcaveTotalSplitter: the candidate deque ran empty (the reference throws)");
    return rc;
}

template <typename TC>
int32_t run_dyn_constrained(cp_csr_s *A, int64_t K, int32_t g, int32_t order, const cp_model_t *mdl, const cp_rowpart_t *Pi,
                            const cp_model_t *w, int64_t wi, double wf, int64_t *spl_out)
{
    hipStream_t s = A->stream;
    int64_t n = A->n;
    // DynamicSplitter.jl:279 rebuilds f WITHOUT the row partition in the chunker loop order
    if (order == CP_ORDER_CHUNKER && mdl->kind == CP_MODEL_BLOCK) { set_error("no oracle_stripe(BlockComponentCostModel, A) without a row partition (DynamicSplitter.jl:279)"); return CP_EUNSUPPORTED; }
    std::unique_ptr<SeqCtx<TC>> C(new SeqCtx<TC>());
    seq_oracle<TC>(A, mdl, order == CP_ORDER_CHUNKER ? nullptr : Pi, *C);
    SeqWeight W = make_weight(A, w, wi, wf);
    int64_t ncol = order == CP_ORDER_SPLITTER ? K : n + 1;
    int64_t cap = (int64_t)2e8;
    double est = (double)K * (double)(n + 1);
    if (est < (double)cap) cap = (int64_t)est + 16;
    DBuf<int64_t> lo((size_t)ncol + 1), hi((size_t)ncol + 1), wpos((size_t)ncol + 2), pv((size_t)cap), spl((size_t)K + 1);
    DBuf<TC> cv((size_t)cap);
    DBuf<int32_t> st(1);
    CP_HIP(hipMemsetAsync(st.p, 0, sizeof(int32_t), s));
    CP_HIP(hipMemsetAsync(pv.p, 0, pv.bytes(), s));
    {
        ProfScope ps(PROF_CHUNK, s, 0.0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dyn_constrained<TC>), dim3(1), dim3(64), 0, s, C->O, W, K, g, order, lo.p, hi.p, wpos.p, cv.p,
                           pv.p, cap, spl.p, st.p);
    }
    CP_HIP(hipGetLastError());
    int32_t rc = 0;
    CP_HIP(hipMemcpyAsync(&rc, st.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipMemcpyAsync(spl_out, spl.p, sizeof(int64_t) * (size_t)(K + 1), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    if (rc == CP_EUNSUPPORTED) set_error("constrained DP window tables exceed the device budget");
    return rc;
}

// ocl(j, j', k) for a batch, evaluated in order by the stateful oracle (BlockComponentCostStepOracle)
template <typename TC>
__global__ void __launch_bounds__(64) k_seq_eval(SeqOracle<TC> O, int64_t nq, const int64_t *__restrict__ j, const int64_t *__restrict__ jp,
                                                 const int64_t *__restrict__ k, TC *__restrict__ out)
{
    for (int64_t t = 0; t < nq; t++) out[t] = ocl(O, j[t], jp[t], k ? k[t] : (int64_t)0);
}

template <typename TC>
int32_t run_seq_eval(cp_csr_s *A, const cp_model_t *mdl, const cp_rowpart_t *Pi, int64_t nq, const int64_t *j, const int64_t *jp,
                     const int64_t *k, TC *out)
{
    hipStream_t s = A->stream;
    if (nq <= 0) return CP_OK;
    for (int64_t t = 0; t < nq; t++) CP_REQUIRE(j[t] >= 1 && jp[t] >= j[t] && jp[t] <= A->n + 1, CP_EINVAL, "oracle query needs 1 <= j <= j' <= n+1");
    std::unique_ptr<SeqCtx<TC>> C(new SeqCtx<TC>());
    seq_oracle<TC>(A, mdl, Pi, *C);
    DBuf<int64_t> dj((size_t)nq), djp((size_t)nq), dk;
    DBuf<TC> dout((size_t)nq);
    CP_HIP(hipMemcpyAsync(dj.p, j, sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
    CP_HIP(hipMemcpyAsync(djp.p, jp, sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
    if (k) { dk.alloc((size_t)nq); CP_HIP(hipMemcpyAsync(dk.p, k, sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s)); }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seq_eval<TC>), dim3(1), dim3(64), 0, s, C->O, nq, dj.p, djp.p, k ? dk.p : nullptr, dout.p);
    CP_HIP(hipGetLastError());
    CP_HIP(hipMemcpyAsync(out, dout.p, sizeof(TC) * (size_t)nq, hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    return CP_OK;
}
template int32_t run_seq_eval<int64_t>(cp_csr_s *, const cp_model_t *, const cp_rowpart_t *, int64_t, const int64_t *, const int64_t *, const int64_t *, int64_t *);
template int32_t run_seq_eval<double>(cp_csr_s *, const cp_model_t *, const cp_rowpart_t *, int64_t, const int64_t *, const int64_t *, const int64_t *, double *);

template int32_t run_dyn_constrained<int64_t>(cp_csr_s *, int64_t, int32_t, int32_t, const cp_model_t *, const cp_rowpart_t *, const cp_model_t *, int64_t, double, int64_t *);
template int32_t run_dyn_constrained<double>(cp_csr_s *, int64_t, int32_t, int32_t, const cp_model_t *, const cp_rowpart_t *, const cp_model_t *, int64_t, double, int64_t *);

}  // namespace cpk

using namespace cpk;

static bool seq_model_ok(const cp_model_t *m)
{
    return m && (m->kind == CP_MODEL_WORK || m->kind == CP_MODEL_CONNECTIVITY || m->kind == CP_MODEL_HYPEREDGE_CUT ||
                 m->kind == CP_MODEL_COLBLOCK || m->kind == CP_MODEL_BLOCK || (m->kind == CP_MODEL_POWER_WORK && m->dtype == CP_F64)) &&
           (m->dtype == CP_I64 || m->dtype == CP_F64);
}

extern "C" {

int32_t cp_pack_dynamic(cp_csr_t A, const cp_model_t *model, const cp_rowpart_t *Pi, const cp_model_t *weight, int64_t wmax_i64,
                        double wmax_f64, int64_t *spl_out, int64_t *K_out)
{
    try {
        CP_REQUIRE(A && spl_out && K_out && seq_model_ok(model) && weight_ok(weight), CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        if (model->dtype == CP_I64) return run_pack_dynamic<int64_t>(A, model, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out);
        return run_pack_dynamic<double>(A, model, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out);
    } CP_CATCH_ALL
}

int32_t cp_pack_convex(cp_csr_t A, const cp_model_t *model, const cp_rowpart_t *Pi, const cp_model_t *weight, int64_t wmax_i64,
                       double wmax_f64, int64_t *spl_out, int64_t *K_out)
{
    try {
        CP_REQUIRE(A && spl_out && K_out && seq_model_ok(model) && weight_ok(weight), CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        if (model->dtype == CP_I64) return run_pack_convex<int64_t>(A, model, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out);
        return run_pack_convex<double>(A, model, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out);
    } CP_CATCH_ALL
}

int32_t cp_pack_convex_batch(cp_csr_t A, int64_t B, const cp_model_t *models, const int64_t *wmax, int64_t ld, int64_t *spl_out, int64_t *K_out)
{
    try {
        CP_REQUIRE(A && models && wmax && spl_out && K_out && B >= 1 && B <= 65535 && ld >= 2, CP_EINVAL, "bad argument");
        CP_REQUIRE(A->n >= 1 && (double)(A->n + 2) * (double)(2 * CW_MAXW + 3) < 2e9, CP_EUNSUPPORTED, "pattern too small / too large for the window table");
        for (int64_t b = 0; b < B; b++) {
            CP_REQUIRE(seq_model_ok(models + b) && models[b].dtype == models[0].dtype, CP_EINVAL, "bad model in the batch (one element type per batch)");
            CP_REQUIRE((models[b].kind == CP_MODEL_COLBLOCK || models[b].kind == CP_MODEL_CONNECTIVITY || models[b].kind == CP_MODEL_WORK) && !models[b].alpha_k,
                       CP_EUNSUPPORTED, "a batch takes ColumnBlock / Connectivity / Work models without per-part alpha");
            CP_REQUIRE(wmax[b] >= 1 && wmax[b] <= CW_MAXW, CP_EUNSUPPORTED, "a batch takes width limits 1 .. 15 (the LDS-resident window kernel)");
        }
        CP_HIP(hipSetDevice(A->device));
        if (models[0].dtype == CP_I64) return run_pack_convex_batch<int64_t>(A, B, models, wmax, ld, spl_out, K_out);
        return run_pack_convex_batch<double>(A, B, models, wmax, ld, spl_out, K_out);
    } CP_CATCH_ALL
}

int32_t cp_partition_convex(cp_csr_t A, int64_t K, const cp_model_t *model, const cp_rowpart_t *Pi, const cp_model_t *weight,
                            int64_t wmax_i64, double wmax_f64, int64_t *spl_out)
{
    try {
        CP_REQUIRE(A && spl_out && K >= 1 && seq_model_ok(model) && weight_ok(weight), CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        if (model->dtype == CP_I64) return run_partition_convex<int64_t>(A, K, model, Pi, weight, wmax_i64, wmax_f64, spl_out);
        return run_partition_convex<double>(A, K, model, Pi, weight, wmax_i64, wmax_f64, spl_out);
    } CP_CATCH_ALL
}

int32_t cp_pack_concave(cp_csr_t A, const cp_model_t *model, const cp_rowpart_t *Pi, const cp_model_t *weight, int64_t wmax_i64,
                        double wmax_f64, int64_t *spl_out, int64_t *K_out)
{
    try {
        CP_REQUIRE(A && spl_out && K_out && seq_model_ok(model) && weight_ok(weight), CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        if (model->dtype == CP_I64) return run_pack_concave<int64_t>(A, model, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out);
        return run_pack_concave<double>(A, model, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out);
    } CP_CATCH_ALL
}

int32_t cp_partition_concave(cp_csr_t A, int64_t K, const cp_model_t *model, const cp_rowpart_t *Pi, const cp_model_t *weight,
                             int64_t wmax_i64, double wmax_f64, int64_t *spl_out)
{
    try {
        CP_REQUIRE(A && spl_out && K >= 1 && seq_model_ok(model) && weight_ok(weight), CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        if (model->dtype == CP_I64) return run_partition_concave<int64_t>(A, K, model, Pi, weight, wmax_i64, wmax_f64, spl_out);
        return run_partition_concave<double>(A, K, model, Pi, weight, wmax_i64, wmax_f64, spl_out);
    } CP_CATCH_ALL
}

}  // extern "C"
