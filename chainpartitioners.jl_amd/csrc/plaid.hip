// plaid.hip -- the 2-D (plaid) connectivity costs of SURVEY 8(f)-4 on the device:
//   AffinePrimaryConnectivityModel    /root/reference/src/PrimaryConnectivityCosts.jl:5-78
//   AffineSecondaryConnectivityModel  /root/reference/src/SecondaryConnectivityCosts.jl:5-108
// A row partition Pi splits the nets of a column part into LOCAL ones (row owned by the same part number) and REMOTE
// ones.  Primary: cost of the column range [j, j') as part k.  Secondary: cost of giving the column range [j, j') to part
// k of a SplitPartition Pi of the rows -- the alternating partitioners call it on the adjoint.
//
// The reference builds partwise(A, Pi) plus a second net counter; on the device the same counts come from the link
// array and the owner of each row:
//   local nets of [p, r) for part k  = #{q in columns [p, r) : prev[q] < p  and  owner[row[q]] == k}
//   secondary local count            = #{columns c in [p, r) that hold a row owned by k}  (prefix sums per part)
// Entry points: oracle values for batches (objective / bounds / oracle_stripe) and the K-part DP by the general
// O(n^2) candidate sweep (the costs depend on k through the ownership, so the O(n log^2 n) scheme does not apply).
#include "csr.hpp"
#include "model.hpp"
#include "dp.hpp"
#include <vector>

namespace cpk {

// one wave per query: nets and local nets of the column range [p, r) for part k (primary),
// or the number of columns of [p, r) / of the whole matrix holding a row owned by k (secondary)
__global__ void __launch_bounds__(256) k_plaid_counts(int64_t nq, const int64_t *__restrict__ P, const int64_t *__restrict__ Rr,
                                                      const int64_t *__restrict__ Kk, int secondary, int64_t n,
                                                      const int64_t *__restrict__ pos, const int32_t *__restrict__ row,
                                                      const int32_t *__restrict__ prev, const int32_t *__restrict__ owner,
                                                      int64_t *__restrict__ out_a, int64_t *__restrict__ out_b)
{
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (i >= nq) return;
    int64_t p = P[i], r = Rr[i];
    int32_t k = (int32_t)Kk[i];
    int64_t a = 0, b = 0;
    if (!secondary) {
        int32_t pp = (int32_t)p;
        for (int64_t q = pos[p] + lane; q < pos[r]; q += 64)
            if (prev[q] < pp) { a += 1; b += (owner[row[q]] == k); }           // a: nets, b: local nets
    } else {
        for (int64_t c = lane; c < n; c += 64) {                               // a: columns of part k anywhere, b: inside [p, r)
            bool has = false;
            for (int64_t q = pos[c]; q < pos[c + 1] && !has; q++) has = owner[row[q]] == k;
            a += has; b += has && c >= p && c < r;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += ((int64_t)__shfl_down((int)(a >> 32), o) << 32) | (uint32_t)__shfl_down((int)(a & 0xffffffffll), o);
        b += ((int64_t)__shfl_down((int)(b >> 32), o) << 32) | (uint32_t)__shfl_down((int)(b & 0xffffffffll), o);
    }
    if (lane == 0) { out_a[i] = a; out_b[i] = b; }
}

template <typename TC>
__global__ void k_plaid_apply(int64_t nq, const int64_t *__restrict__ P, const int64_t *__restrict__ Rr, const int64_t *__restrict__ Kk,
                              int secondary, const int64_t *__restrict__ pos, const int64_t *__restrict__ tpos, const int64_t *__restrict__ pspl,
                              const int64_t *__restrict__ ca, const int64_t *__restrict__ cb, DevModel<TC> M, TC *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int64_t p = P[i], r = Rr[i], k = Kk[i];
    TC alpha = dm_alpha(M, k);
    if (!secondary) out[i] = dm_apply(M, alpha, r - p, pos[r] - pos[p], cb[i], ca[i] - cb[i]);      // local, remote
    else {
        int64_t s0 = pspl[k - 1] - 1, s1 = pspl[k] - 1;                          // rows of part k (0-based range)
        out[i] = dm_apply(M, alpha, s1 - s0, tpos[s1] - tpos[s0], cb[i], ca[i] - cb[i]);
    }
}

// device copy of the row partition: owner (1-based part of every row) and the split vector (secondary)
struct PlaidPart {
    DBuf<int32_t> owner;
    DBuf<int64_t> spl;
    std::vector<int64_t> hspl;
    int64_t K = 0;
};

static void upload_part(cp_csr_s *A, const cp_rowpart_t *Pi, bool need_spl, PlaidPart &R)
{
    CP_REQUIRE(Pi && (Pi->asg || Pi->spl) && Pi->K >= 1, CP_EINVAL, "this cost model needs a row partition Pi");
    CP_REQUIRE(!need_spl || Pi->spl, CP_EINVAL, "the secondary connectivity model needs a SplitPartition of the rows");
    hipStream_t s = A->stream;
    int64_t m = A->m, K = Pi->K;
    R.K = K;
    std::vector<int32_t> ho((size_t)(m > 0 ? m : 1), 0);
    if (Pi->asg) for (int64_t i = 0; i < m; i++) { CP_REQUIRE(Pi->asg[i] >= 1 && Pi->asg[i] <= K, CP_EINVAL, "row owner out of range"); ho[(size_t)i] = (int32_t)Pi->asg[i]; }
    else for (int64_t k = 1; k <= K; k++) for (int64_t i = Pi->spl[k - 1]; i <= Pi->spl[k] - 1; i++) { CP_REQUIRE(i >= 1 && i <= m, CP_EINVAL, "split vector out of range"); ho[(size_t)i - 1] = (int32_t)k; }
    R.owner.alloc(ho.size());
    CP_HIP(hipMemcpyAsync(R.owner.p, ho.data(), sizeof(int32_t) * ho.size(), hipMemcpyHostToDevice, s));
    if (Pi->spl) {
        R.hspl.assign(Pi->spl, Pi->spl + K + 1);
        CP_REQUIRE(R.hspl[0] == 1 && R.hspl[(size_t)K] == m + 1, CP_EINVAL, "split vector must run from 1 to m+1");
        R.spl.alloc((size_t)K + 1);
        CP_HIP(hipMemcpyAsync(R.spl.p, R.hspl.data(), sizeof(int64_t) * (size_t)(K + 1), hipMemcpyHostToDevice, s));
    }
    CP_HIP(hipStreamSynchronize(s));
}

// ocl(j, j', k) for a batch (1-based queries)
template <typename TC>
int32_t run_plaid_eval(cp_csr_s *A, const cp_model_t *mdl, const cp_rowpart_t *Pi, int64_t nq, const int64_t *j, const int64_t *jp,
                       const int64_t *k, TC *out)
{
    hipStream_t s = A->stream;
    if (nq <= 0) return CP_OK;
    bool sec = mdl->kind == CP_MODEL_SECONDARY;
    CP_REQUIRE(k, CP_EINVAL, "the plaid cost models take the part number k");
    ensure_links(A);
    PlaidPart R;
    upload_part(A, Pi, sec, R);
    std::vector<int64_t> hp((size_t)nq), hr((size_t)nq);
    for (int64_t i = 0; i < nq; i++) {
        CP_REQUIRE(j[i] >= 1 && jp[i] >= j[i] && jp[i] <= A->n + 1 && k[i] >= 1 && k[i] <= R.K, CP_EINVAL, "oracle query out of range");
        hp[(size_t)i] = j[i] - 1; hr[(size_t)i] = jp[i] - 1;
    }
    DBuf<int64_t> dP((size_t)nq), dR((size_t)nq), dK((size_t)nq), ca((size_t)nq), cb((size_t)nq);
    DBuf<TC> dO((size_t)nq);
    CP_HIP(hipMemcpyAsync(dP.p, hp.data(), sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
    CP_HIP(hipMemcpyAsync(dR.p, hr.data(), sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
    CP_HIP(hipMemcpyAsync(dK.p, k, sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
    HostModel<TC> HM;
    build_dev_model<TC>(mdl, HM, s);
    hipLaunchKernelGGL(k_plaid_counts, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, s, nq, dP.p, dR.p, dK.p, sec ? 1 : 0, A->n, A->pos.p, A->row.p,
                       A->prev.p, R.owner.p, ca.p, cb.p);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_plaid_apply<TC>), dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, s, nq, dP.p, dR.p, dK.p, sec ? 1 : 0,
                       A->pos.p, A->tpos.p, sec ? R.spl.p : (const int64_t *)nullptr, ca.p, cb.p, HM.d, dO.p);
    CP_HIP(hipGetLastError());
    CP_HIP(hipMemcpyAsync(out, dO.p, sizeof(TC) * (size_t)nq, hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    return CP_OK;
}

// ------------------------------------------------------------------ K-part DP, general sweep (DynamicSplitter.jl:15-87)
// 64 rows per wave walk the candidates p downwards together; the link entries of the stepped column are wave-uniform loads.
// Primary: nets and local nets grow with every step.  Secondary: the cost is a function of prefix counts Lk only.
// first != 0: layer 1, the only candidate is p = 0 (cst[j', 1] = f(1, j', 1)).
template <typename TC>
__global__ void __launch_bounds__(256) k_plaid_layer(int64_t n, int64_t r_lo, int64_t r_hi, int secondary, int first, int32_t k,
                                                     const int64_t *__restrict__ pos, const int32_t *__restrict__ row,
                                                     const int32_t *__restrict__ next, const int32_t *__restrict__ owner,
                                                     const int32_t *__restrict__ Lk, int64_t nvk, int64_t wk, int64_t dk,
                                                     DevModel<TC> M, TC alpha, int32_t g, const TC *__restrict__ W,
                                                     TC *__restrict__ cst, int32_t *__restrict__ ptr)
{
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    int64_t r0 = r_lo + wave * 64;
    if (r0 > r_hi) return;
    int64_t r = r0 + lane;
    bool live = r <= r_hi;
    int64_t rtop = r0 + 63 < r_hi ? r0 + 63 : r_hi;
    int64_t nn = 0, nl = 0;
    TC best = (TC)0;
    int64_t bp = -1;
    int32_t rr = (int32_t)r;
    for (int64_t p = rtop; p >= 0; p--) {
        if (!secondary && p < n) {
            bool step = live && (r > p);                       // lanes with r > p take the left step over column p
            for (int64_t q = pos[p]; q < pos[p + 1]; q++) {
                int32_t nx = next[q];                          // wave-uniform addresses
                bool mine = owner[row[q]] == k;
                if (step && nx >= rr) { nn += 1; nl += mine; }
            }
        }
        if (live && r >= p && (!first || p == 0)) {
            TC f;
            if (!secondary) f = dm_apply(M, alpha, r - p, pos[r] - pos[p], nl, nn - nl);
            else { int64_t l = (int64_t)Lk[r] - Lk[p]; f = dm_apply(M, alpha, nvk, wk, l, dk - l); }
            TC v = first ? f : comb(g, W[p], f);
            if (bp < 0 || v < best) { best = v; bp = p; }      // walking p downwards: the largest j wins ties
        }
    }
    if (live) { cst[r] = best; ptr[r] = (int32_t)bp; }
}

// flags of the columns holding a row owned by k, then their exclusive prefix sums Lk[0 .. n]
__global__ void k_sec_flags(int64_t n, int32_t k, const int64_t *__restrict__ pos, const int32_t *__restrict__ row,
                            const int32_t *__restrict__ owner, int32_t *__restrict__ flag)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    bool has = false;
    for (int64_t q = pos[c]; q < pos[c + 1] && !has; q++) has = owner[row[q]] == k;
    flag[c] = has ? 1 : 0;
}

template <typename TC> static TC plaid_alpha(const cp_model_t *m, int64_t k)
{
    if (m->alpha_k && k >= 1 && k <= m->n_alpha_k) return ((const TC *)m->alpha_k)[k - 1];
    return model_param<TC>(m, CP_P_ALPHA);
}

template <typename TC>
int32_t run_plaid_dynamic(cp_csr_s *A, int64_t K, int32_t combine, int32_t order, const cp_model_t *mdl, const cp_rowpart_t *Pi,
                          int64_t *spl_out)
{
    hipStream_t s = A->stream;
    int64_t n = A->n;
    bool sec = mdl->kind == CP_MODEL_SECONDARY;
    CP_REQUIRE(order == CP_ORDER_SPLITTER, CP_EUNSUPPORTED, "the plaid cost models need the part number: splitter loop order only");
    CP_REQUIRE(n <= g_opt_brute_max_n, CP_EUNSUPPORTED, "plaid cost models run the O(n^2) device sweep: n too large");
    ensure_links(A);
    PlaidPart R;
    upload_part(A, Pi, sec, R);
    CP_REQUIRE(!sec || K <= R.K, CP_EINVAL, "the secondary model is evaluated for parts 1..K of Pi");
    HostModel<TC> HM;
    build_dev_model<TC>(mdl, HM, s);
    size_t n1 = (size_t)n + 1;
    DBuf<TC> cstA(n1), cstB(n1);
    DBuf<int32_t> ptr((size_t)K * n1), flag((size_t)(n > 0 ? n : 1)), Lk(n1 + 1);
    DBuf<int64_t> scratch;
    std::vector<int64_t> htpos;
    if (sec) {                                                       // pins of every part of Pi: tpos = row pointer of A
        htpos.resize((size_t)A->m + 1);
        CP_HIP(hipMemcpyAsync(htpos.data(), A->tpos.p, sizeof(int64_t) * (size_t)(A->m + 1), hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
    }
    TC *prevc = cstA.p, *curc = cstB.p;
    for (int64_t k = 1; k <= K; k++) {
        int64_t nvk = 0, wk = 0, dk = 0;
        if (sec) {
            if (n > 0) hipLaunchKernelGGL(k_sec_flags, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, n, (int32_t)k, A->pos.p, A->row.p, R.owner.p, flag.p);
            exclusive_scan_i32_i32(flag.p, Lk.p, n, scratch, s);
            int32_t tot = 0;
            CP_HIP(hipMemcpyAsync(&tot, Lk.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            CP_HIP(hipStreamSynchronize(s));
            dk = tot;
            int64_t s0 = R.hspl[(size_t)k - 1] - 1, s1 = R.hspl[(size_t)k] - 1;
            nvk = s1 - s0; wk = htpos[(size_t)s1] - htpos[(size_t)s0];
        }
        int32_t *pk = ptr.p + (size_t)(k - 1) * n1;
        bool last = (k == K) && K > 1;
        int64_t rlo = last ? n : 0;                                   // layer K: row n+1 only (DynamicSplitter.jl:34)
        int64_t waves = cdiv(n - rlo + 1, (int64_t)64);
        ProfScope ps(PROF_BRUTE, s, 0.0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_plaid_layer<TC>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, s, n, rlo, n, sec ? 1 : 0, k == 1 ? 1 : 0,
                           (int32_t)k, A->pos.p, A->row.p, A->next.p, R.owner.p, Lk.p, nvk, wk, dk, HM.d, plaid_alpha<TC>(mdl, k), combine,
                           prevc, curc, pk);
        CP_HIP(hipGetLastError());
        std::swap(prevc, curc);
    }
    std::vector<int64_t> spl((size_t)K + 1);
    spl[(size_t)K] = n;
    for (int64_t k = K; k >= 1; k--) {                               // unravel_splits (DynamicSplitter.jl:89-99)
        int32_t v = 0;
        CP_HIP(hipMemcpyAsync(&v, ptr.p + (size_t)(k - 1) * n1 + (size_t)spl[(size_t)k], sizeof(int32_t), hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
        spl[(size_t)k - 1] = v;
    }
    for (int64_t k = 0; k <= K; k++) spl_out[k] = spl[(size_t)k] + 1;
    prof_collect();
    return CP_OK;
}

template int32_t run_plaid_eval<int64_t>(cp_csr_s *, const cp_model_t *, const cp_rowpart_t *, int64_t, const int64_t *, const int64_t *, const int64_t *, int64_t *);
template int32_t run_plaid_eval<double>(cp_csr_s *, const cp_model_t *, const cp_rowpart_t *, int64_t, const int64_t *, const int64_t *, const int64_t *, double *);
template int32_t run_plaid_dynamic<int64_t>(cp_csr_s *, int64_t, int32_t, int32_t, const cp_model_t *, const cp_rowpart_t *, int64_t *);
template int32_t run_plaid_dynamic<double>(cp_csr_s *, int64_t, int32_t, int32_t, const cp_model_t *, const cp_rowpart_t *, int64_t *);

}  // namespace cpk
