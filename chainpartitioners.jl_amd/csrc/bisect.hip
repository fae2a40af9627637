// bisect.hip -- partition_stripe(A, K, [Flip]BisectCostBottleneckSplitter(f, eps))
// (/root/reference/src/BisectCostBottleneckSplitter.jl:6-63, flip :70-127) as ONE kernel launch.
//
// The algorithm is a sequential chain by nature: Float64 bisection on the cost c; each probe is K-1 binary
// searches whose starting point is the previous search's result.  One wave runs it:
//  * every lane carries the same scalar state (c_lo, c_hi, the three split vectors in HBM), so the control
//    flow is wave-uniform and bit-identical to the reference's loop (midpoints (c_lo+c_hi)/2, test
//    c_lo*(1+eps) < c_hi, exact Int64-vs-Float64 comparison);
//  * inside one binary search the 63 midpoints of the next SIX levels of the decision tree are evaluated at
//    once, one per lane (for ConnectivityCosts each is an H-level rank query on the wavelet counter), and the
//    wave then replays the six decisions -- the visited midpoints and the result are exactly those of the
//    sequential search for ANY cost function, monotone or not.
#include "csr.hpp"
#include "model.hpp"
#include "wavelet.hpp"

namespace cpk {

// v <= c with v::Int64, c::Float64 compared exactly (Julia semantics); Float64 costs compare natively
__device__ __forceinline__ bool le_f64(int64_t v, double c)
{
    if (c != c) return false;
    if (c >= 9223372036854775808.0) return true;
    if (c < -9223372036854775808.0) return false;
    return v <= (int64_t)floor(c);
}
__device__ __forceinline__ bool le_f64(double v, double c) { return v <= c; }
__device__ __forceinline__ bool le_f64(int64_t v, int64_t c) { return v <= c; }
// v < c with v::Int64, c::Float64, exactly
__device__ __forceinline__ bool lt_f64(int64_t v, double c)
{
    if (c != c) return false;
    if (c >= 9223372036854775808.0) return true;
    if (c <= -9223372036854775808.0) return false;
    return v < (int64_t)ceil(c);
}

template <typename TC>
struct OracleDev {
    DevModel<TC> M;
    const int64_t *pos;
    int64_t n;
    int32_t has_net;
    WaveletDev net;
    // plaid connectivity costs: 0 none, 1 primary, 2 secondary.  partwise(A, Pi) (PartwiseCounts.jl:1-60):
    // pios (K+1) / prm (n') hold 1-based values as the reference does; primary adds the net counter of the partwise matrix
    int32_t plaid;
    const int64_t *pios, *prm;
    WaveletDev lcn; int64_t np; const int64_t *ppos;
    const int64_t *spl, *tpos;            // secondary: the row split (1-based) and the row pointer of A (0-based)
};

// PartwiseCount rank: pios[k] + searchsortedfirst(prm[pios[k] : pios[k+1]-1], j) - 1   (PartwiseCounts.jl:86-88)
template <typename TC>
__device__ __forceinline__ int64_t pw_rank(const OracleDev<TC> &O, int64_t j, int64_t k)
{
    int64_t lo = O.pios[k - 1], hi = O.pios[k];       // 1-based positions into prm
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (O.prm[mid - 1] < j) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <typename TC>
__device__ __forceinline__ TC oracle_eval_dev(const OracleDev<TC> &O, int64_t j, int64_t jp, int64_t k)
{
    int64_t p = j - 1, r = jp - 1;
    if (O.plaid == 2) {                                                   // SecondaryConnectivityCosts.jl:79-86
        int64_t l = pw_rank(O, jp, k) - pw_rank(O, j, k);
        int64_t s0 = O.spl[k - 1] - 1, s1 = O.spl[k] - 1;
        return dm_apply(O.M, dm_alpha(O.M, k), s1 - s0, O.tpos[s1] - O.tpos[s0], l, (O.pios[k] - O.pios[k - 1]) - l);
    }
    int64_t np = O.pos[r] - O.pos[p];
    int64_t nn = 0;
    if (O.has_net) nn = np - wt_count_le(O.net, O.n - p, O.pos[r]);      // SparseColorArrays.jl:121-125
    if (O.plaid == 1) {                                                   // PrimaryConnectivityCosts.jl:67-74
        int64_t pj = pw_rank(O, j, k) - 1, pr = pw_rank(O, jp, k) - 1;    // columns of the partwise matrix, 0-based
        int64_t l = (O.ppos[pr] - O.ppos[pj]) - wt_count_le(O.lcn, O.np - pj, O.ppos[pr]);
        return dm_apply(O.M, dm_alpha(O.M, k), r - p, np, l, nn - l);
    }
    return dm_apply(O.M, dm_alpha(O.M, k), r - p, np, nn, (int64_t)0);
}

template <typename TC, typename CT>
__device__ int64_t search6(const OracleDev<TC> &O, int64_t j, int64_t lo, int64_t hi, int64_t k, CT c, int flip, int lane)
{
    if (lo < j) lo = j;                                   // j'_lo = max(j, j'_lo)  (:17)
    while (lo <= hi) {
        int node = lane;                                  // heap index 1..63; lane 0 idles
        int64_t l = lo, h = hi;
        bool alive = node >= 1;
        if (alive) {
            int depth = 31 - __clz(node);
            for (int dd = depth - 1; dd >= 0; dd--) {
                if (l > h) break;
                int64_t mid = (int64_t)(((uint64_t)(l + h)) >> 1);
                if ((node >> dd) & 1) l = mid + 1; else h = mid - 1;
            }
        }
        bool has = alive && l <= h;
        int64_t mid = (int64_t)(((uint64_t)(l + h)) >> 1);
        bool le = false;
        if (has) le = le_f64(oracle_eval_dev(O, j, mid, k), c);
        unsigned long long lem = __ballot(le), hasm = __ballot(has);
        int nd = 1;
        for (int step = 0; step < 6; step++) {
            if (!((hasm >> nd) & 1)) break;
            int64_t m2 = (int64_t)(((uint64_t)(lo + hi)) >> 1);
            bool isle = (lem >> nd) & 1;
            bool right;
            if (!flip) { if (isle) { lo = m2 + 1; right = true; } else { hi = m2 - 1; right = false; } }
            else       { if (isle) { hi = m2 - 1; right = false; } else { lo = m2 + 1; right = true; } }
            nd = 2 * nd + (right ? 1 : 0);
        }
    }
    return flip ? lo : hi;
}

template <typename TC>
__device__ __forceinline__ void bisect_cost_body(const OracleDev<TC> &O, int64_t K, double c_lo, double c_hi, double eps, int flip,
                                                 int64_t *__restrict__ spl_lo, int64_t *__restrict__ spl_hi, int64_t *__restrict__ spl,
                                                 int64_t *__restrict__ out, int64_t *__restrict__ nprobes)
{
    int lane = threadIdx.x;
    int64_t n = O.n;
    // every lane performs every (wave-uniform) store, so each lane later reads its own writes
    for (int64_t k = 1; k <= K + 1; k++) { spl_lo[k - 1] = 1; spl_hi[k - 1] = n + 1; spl[k - 1] = 0; }
    spl_lo[K] = n + 1;
    spl_hi[0] = 1;
    spl[0] = 1;
    spl[K] = n + 1;
    int64_t probes = 0;
    bool stuck = false;
    while (c_lo * (1 + eps) < c_hi) {                     // :41
        double c = (c_lo + c_hi) / 2;
        probes++;
        spl[0] = 1;
        bool chk = true;
        for (int64_t k = 1; k <= K - 1; k++) {
            int64_t j = spl[k - 1];
            int64_t rr = search6(O, j, spl_lo[k], spl_hi[k], k, c, flip, lane);
            spl[k] = rr;
            if (!flip) {
                if (rr < j) { chk = false; for (int64_t t = k + 1; t <= K; t++) spl[t - 1] = j; break; }
            } else {
                if (rr > n + 1) { chk = false; for (int64_t t = k + 1; t <= K; t++) spl[t - 1] = n + 1; break; }
            }
        }
        bool feas = false;
        if (chk) feas = le_f64(oracle_eval_dev(O, spl[K - 1], spl[K], K), c);
        // no bound moved (or far too many probes): the reference's loop would repeat this probe forever -- it does for
        // non-positive bounds, where c_lo * (1 + eps) < c_hi can never become false.  A kernel must not.
        if ((feas ? c_hi : c_lo) == c || probes > 4096) { stuck = true; break; }
        if (feas) {
            c_hi = c;
            int64_t *dst = flip ? spl_lo : spl_hi;
            for (int64_t k = 0; k <= K; k++) dst[k] = spl[k];
        } else {
            c_lo = c;
            int64_t *dst = flip ? spl_hi : spl_lo;
            for (int64_t k = 0; k <= K; k++) dst[k] = spl[k];
        }
    }
    const int64_t *src = flip ? spl_lo : spl_hi;
    for (int64_t k = lane; k <= K; k += 64) out[k] = src[k];
    if (lane == 0) *nprobes = stuck ? -1 : probes;
}

template <typename TC>
__global__ void __launch_bounds__(64) k_bisect(OracleDev<TC> O, int64_t K, double c_lo, double c_hi, double eps, int flip,
                                               int64_t *__restrict__ spl_lo, int64_t *__restrict__ spl_hi, int64_t *__restrict__ spl,
                                               int64_t *__restrict__ out, int64_t *__restrict__ nprobes)
{
    bisect_cost_body<TC>(O, K, c_lo, c_hi, eps, flip, spl_lo, spl_hi, spl, out, nprobes);
}

// B independent requests (K, model, eps, flip) on ONE pattern, one wave each: the sequential probe chain of a single partition
// leaves 255 of 256 CUs idle -- a sweep over K / eps / model constants fills them, and the counting structure is built once
template <typename TC>
struct BisectReq { DevModel<TC> M; int64_t K; double c_lo, c_hi, eps; int32_t flip, _pad; };

template <typename TC>
__global__ void __launch_bounds__(64) k_bisect_batch(OracleDev<TC> O, const BisectReq<TC> *__restrict__ req, int64_t ld, int64_t *__restrict__ work,
                                                     int64_t *__restrict__ out, int64_t *__restrict__ nprobes)
{
    const BisectReq<TC> R = req[blockIdx.x];
    O.M = R.M;
    int64_t *w = work + (int64_t)blockIdx.x * 3 * ld;
    bisect_cost_body<TC>(O, R.K, R.c_lo, R.c_hi, R.eps, R.flip, w, w + ld, w + 2 * ld, out + (int64_t)blockIdx.x * ld, nprobes + blockIdx.x);
}

// ------------------------------------------------------------------ BisectIndexBottleneckSplitter.jl:5-83, flip :85-166
// c_lo / c_hi start as Float64 (bound_stripe ./ 1, :39) and are later overwritten with cost values of type Tc (:60, :64);
// Int64 costs keep the two representations apart so that every comparison is exact.
template <typename TC> struct IdxBound;
template <> struct IdxBound<double> {
    double v;
    __device__ void init(double d) { v = d; }
    __device__ void set(double c) { v = c; }
    __device__ bool le_c(double c) const { return v <= c; }        // bound <= c
    __device__ bool c_lt(double c) const { return c < v; }         // c < bound
};
template <> struct IdxBound<int64_t> {
    double f; int64_t i; int is_int;
    __device__ void init(double d) { f = d; i = 0; is_int = 0; }
    __device__ void set(int64_t c) { i = c; is_int = 1; }
    __device__ bool le_c(int64_t c) const { return is_int ? i <= c : !lt_f64(c, f); }
    __device__ bool c_lt(int64_t c) const { return is_int ? c < i : lt_f64(c, f); }
};

template <typename TC>
__global__ void __launch_bounds__(64) k_bisect_index(OracleDev<TC> O, int64_t K, double c_lo0, double c_hi0, int flip,
                                                     int64_t *__restrict__ spl_lo, int64_t *__restrict__ spl_hi, int64_t *__restrict__ spl,
                                                     int64_t *__restrict__ out, int64_t *__restrict__ nprobes)
{
    int lane = threadIdx.x;
    int64_t n = O.n;
    for (int64_t k = 1; k <= K + 1; k++) { spl_lo[k - 1] = 1; spl_hi[k - 1] = n + 1; spl[k - 1] = 0; }
    spl_lo[K] = n + 1;
    spl_hi[0] = 1;
    spl[0] = 1;
    spl[K] = n + 1;
    IdxBound<TC> clo, chi;
    clo.init(c_lo0); chi.init(c_hi0);
    int64_t probes = 0;
    for (int64_t k = 1; k <= K; k++) {
        int64_t jhi = spl_hi[k];
        int64_t jlo = spl[k - 1] > spl_lo[k] ? spl[k - 1] : spl_lo[k];
        while (jlo <= jhi) {
            int64_t jp = (int64_t)(((uint64_t)(jlo + jhi)) >> 1);
            TC c = oracle_eval_dev(O, spl[k - 1], jp, k);
            if (clo.le_c(c) && chi.c_lt(c)) {                  // c_lo <= c < c_hi
                probes++;
                bool chk = true;
                spl[k] = jp;
                for (int64_t kk = k + 1; kk <= K - 1; kk++) {
                    int64_t j = spl[kk - 1];
                    int64_t rr = search6(O, j, spl_lo[kk], spl_hi[kk], kk, c, flip, lane);
                    spl[kk] = rr;
                    if (!flip) {
                        if (rr < j) { chk = false; for (int64_t t = kk + 1; t <= K; t++) spl[t - 1] = j; break; }
                    } else {
                        if (rr > n + 1) { chk = false; for (int64_t t = kk + 1; t <= K; t++) spl[t - 1] = n + 1; break; }
                    }
                }
                bool ok = chk && le_f64(oracle_eval_dev(O, spl[K - 1], spl[K], K), c);
                int64_t *dst;
                if (ok) {
                    chi.set(c);
                    if (!flip) { jhi = jp - 1; dst = spl_hi; } else { jlo = jp + 1; dst = spl_lo; }
                } else {
                    clo.set(c);
                    if (!flip) { jlo = jp + 1; dst = spl_lo; } else { jhi = jp - 1; dst = spl_hi; }
                }
                for (int64_t t = 0; t <= K; t++) dst[t] = spl[t];
            } else if (!chi.c_lt(c)) {                         // c >= c_hi
                if (!flip) jhi = jp - 1; else jlo = jp + 1;
            } else {
                if (!flip) jlo = jp + 1; else jhi = jp - 1;
            }
        }
        if (!flip) {
            if (jhi < spl[k - 1]) break;                       // :74
            spl[k] = jhi;
        } else {
            if (jlo > n + 1) break;                            // :157
            spl[k] = jlo;
        }
    }
    const int64_t *src = flip ? spl_lo : spl_hi;
    for (int64_t k = lane; k <= K; k += 64) out[k] = src[k];
    if (lane == 0) *nprobes = probes;
}

struct CsrGuard { cp_csr_t h = nullptr; ~CsrGuard() { if (h) cp_csr_destroy(h); } };

template <typename TC>
int32_t run_bisect(cp_csr_s *A, int64_t K, const cp_model_t *mdl, const cp_rowpart_t *Pi, double c_lo, double c_hi, double eps, int flip,
                   int64_t *spl_out, bool by_index = false)
{
    hipStream_t s = A->stream;
    HostModel<TC> HM;
    build_dev_model<TC>(mdl, HM, s);
    OracleDev<TC> O;
    memset(&O, 0, sizeof(O));
    O.M = HM.d; O.pos = A->pos.p; O.n = A->n; O.has_net = 0;
    WaveletHost net, lcn;
    CsrGuard Ap;
    DBuf<int64_t> d_pios, d_prm, d_spl2;
    if (mdl->kind == CP_MODEL_CONNECTIVITY || mdl->kind == CP_MODEL_PRIMARY) { ensure_net_counter(A, net); O.has_net = 1; O.net = net.d; }
    if (mdl->kind == CP_MODEL_PRIMARY || mdl->kind == CP_MODEL_SECONDARY) {
        bool sec = mdl->kind == CP_MODEL_SECONDARY;
        CP_REQUIRE(Pi && (Pi->asg || Pi->spl) && Pi->K >= 1, CP_EINVAL, "this cost model needs a row partition Pi");
        CP_REQUIRE(!sec || (Pi->spl && K <= Pi->K), CP_EINVAL, "the secondary connectivity model needs a SplitPartition of the rows with >= K parts");
        int64_t m = A->m, N = A->N, Kp = Pi->K;
        std::vector<int64_t> asg((size_t)(m > 0 ? m : 1), 1);
        if (Pi->asg) for (int64_t i = 0; i < m; i++) asg[(size_t)i] = Pi->asg[i];
        else for (int64_t k = 1; k <= Kp; k++) for (int64_t i = Pi->spl[k - 1]; i <= Pi->spl[k] - 1; i++) { CP_REQUIRE(i >= 1 && i <= m, CP_EINVAL, "split vector out of range"); asg[(size_t)i - 1] = k; }
        std::vector<int64_t> pios((size_t)Kp + 1), prm((size_t)(N > 0 ? N : 1)), ppos((size_t)N + 2), pidx((size_t)(N > 0 ? N : 1));
        int64_t npr = 0;
        int32_t rc = cp_partwise(A, Kp, asg.data(), &npr, pios.data(), prm.data(), ppos.data(), pidx.data());
        if (rc != CP_OK) return rc;
        d_pios.alloc((size_t)Kp + 1); d_prm.alloc((size_t)(npr > 0 ? npr : 1));
        CP_HIP(hipMemcpyAsync(d_pios.p, pios.data(), sizeof(int64_t) * (size_t)(Kp + 1), hipMemcpyHostToDevice, s));
        if (npr > 0) CP_HIP(hipMemcpyAsync(d_prm.p, prm.data(), sizeof(int64_t) * (size_t)npr, hipMemcpyHostToDevice, s));
        O.plaid = sec ? 2 : 1; O.pios = d_pios.p; O.prm = d_prm.p;
        if (!sec) {
            rc = cp_csr_create(m, npr, N, ppos.data(), pidx.data(), A->device, &Ap.h);          // the partwise matrix and its net counter
            if (rc != CP_OK) return rc;
            ensure_net_counter(Ap.h, lcn);
            O.lcn = lcn.d; O.np = npr; O.ppos = Ap.h->pos.p;
        } else {
            d_spl2.alloc((size_t)Kp + 1);
            CP_HIP(hipMemcpyAsync(d_spl2.p, Pi->spl, sizeof(int64_t) * (size_t)(Kp + 1), hipMemcpyHostToDevice, s));
            ensure_links(A);
            O.spl = d_spl2.p; O.tpos = A->tpos.p;
        }
        CP_HIP(hipStreamSynchronize(s));              // host staging vectors die at scope end
    }
    DBuf<int64_t> buf((size_t)(4 * (K + 1) + 1));
    int64_t *d_lo = buf.p, *d_hi = buf.p + (K + 1), *d_spl = buf.p + 2 * (K + 1), *d_out = buf.p + 3 * (K + 1), *d_np = buf.p + 4 * (K + 1);
    {
        ProfScope ps(PROF_BISECT, s, 0.0);
        if (by_index) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bisect_index<TC>), dim3(1), dim3(64), 0, s, O, K, c_lo, c_hi, flip, d_lo, d_hi, d_spl, d_out, d_np);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bisect<TC>), dim3(1), dim3(64), 0, s, O, K, c_lo, c_hi, eps, flip, d_lo, d_hi, d_spl, d_out, d_np);
    }
    CP_HIP(hipGetLastError());
    int64_t np_host = 0;
    CP_HIP(hipMemcpyAsync(spl_out, d_out, sizeof(int64_t) * (size_t)(K + 1), hipMemcpyDeviceToHost, s));
    CP_HIP(hipMemcpyAsync(&np_host, d_np, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    CP_REQUIRE(np_host >= 0, CP_EINVAL, "cost bisection cannot terminate on these bounds (the reference loops forever: non-positive costs)");
    return CP_OK;
}

template <typename TC>
int32_t run_bisect_batch(cp_csr_s *A, int64_t B, const int64_t *K, const cp_model_t *models, const double *eps, const int32_t *flip, int64_t ld,
                         int64_t *spl_out)
{
    hipStream_t s = A->stream;
    std::vector<HostModel<TC>> HM((size_t)B);
    std::vector<BisectReq<TC>> req((size_t)B);
    bool any_net = false;
    for (int64_t b = 0; b < B; b++) {
        const cp_model_t *m = models + b;
        build_dev_model<TC>(m, HM[(size_t)b], s);
        int64_t li, hi; double lf, hf;
        int32_t rc = cp_bound_stripe(A, K[b], m, &li, &hi, &lf, &hf);      // (c_lo, c_hi) = bound_stripe(A, K, f) ./ 1  (BisectCostBottleneckSplitter.jl:39)
        if (rc != CP_OK) return rc;
        BisectReq<TC> &R = req[(size_t)b];
        memset(&R, 0, sizeof(R));
        R.M = HM[(size_t)b].d; R.K = K[b]; R.c_lo = lf; R.c_hi = hf; R.eps = eps[b]; R.flip = flip ? flip[b] : 0;
        any_net |= m->kind == CP_MODEL_CONNECTIVITY;
    }
    OracleDev<TC> O;
    memset(&O, 0, sizeof(O));
    O.pos = A->pos.p; O.n = A->n;
    WaveletHost net;
    if (any_net) { ensure_net_counter(A, net); O.has_net = 1; O.net = net.d; }      // built ONCE for the whole batch
    DBuf<BisectReq<TC>> dreq((size_t)B);
    DBuf<int64_t> work((size_t)(3 * B * ld)), out((size_t)(B * ld)), np((size_t)B);
    CP_HIP(hipMemcpyAsync(dreq.p, req.data(), sizeof(BisectReq<TC>) * (size_t)B, hipMemcpyHostToDevice, s));
    {
        ProfScope ps(PROF_BISECT, s, 0.0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bisect_batch<TC>), dim3((unsigned)B), dim3(64), 0, s, O, dreq.p, ld, work.p, out.p, np.p);
    }
    CP_HIP(hipGetLastError());
    std::vector<int64_t> hnp((size_t)B);
    CP_HIP(hipMemcpyAsync(spl_out, out.p, sizeof(int64_t) * (size_t)(B * ld), hipMemcpyDeviceToHost, s));
    CP_HIP(hipMemcpyAsync(hnp.data(), np.p, sizeof(int64_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    prof_collect();
    for (int64_t b = 0; b < B; b++)
        CP_REQUIRE(hnp[(size_t)b] >= 0, CP_EINVAL, "cost bisection cannot terminate on these bounds (the reference loops forever: non-positive costs)");
    return CP_OK;
}

}  // namespace cpk

using namespace cpk;

extern "C" int32_t cp_partition_bisect_cost_batch(cp_csr_t A, int64_t B, const int64_t *K, const cp_model_t *models, const double *eps,
                                                  const int32_t *flip, int64_t ld, int64_t *spl_out)
{
    try {
        CP_REQUIRE(A && K && models && eps && spl_out && B >= 1 && B <= 65535, CP_EINVAL, "bad argument");
        for (int64_t b = 0; b < B; b++) {
            CP_REQUIRE(K[b] >= 1 && K[b] + 1 <= ld, CP_EINVAL, "every request needs 1 <= K and K + 1 <= ld");
            CP_REQUIRE((models[b].kind == CP_MODEL_WORK || models[b].kind == CP_MODEL_CONNECTIVITY) && models[b].dtype == models[0].dtype, CP_EUNSUPPORTED,
                       "a batch takes Work / Connectivity models of one element type");
        }
        CP_HIP(hipSetDevice(A->device));
        if (models[0].dtype == CP_I64) return run_bisect_batch<int64_t>(A, B, K, models, eps, flip, ld, spl_out);
        return run_bisect_batch<double>(A, B, K, models, eps, flip, ld, spl_out);
    } CP_CATCH_ALL
}

extern "C" int32_t cp_partition_bisect_cost_pi(cp_csr_t A, int64_t K, const cp_model_t *model, const cp_rowpart_t *Pi, double eps,
                                               int32_t flip, int64_t *spl_out)
{
    try {
        CP_REQUIRE(A && model && spl_out && K >= 1, CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        int64_t li, hi; double lf, hf;
        int32_t rc = cp_bound_stripe_pi(A, K, Pi, model, &li, &hi, &lf, &hf);    // (c_lo, c_hi) = bound_stripe(A, K, args..., f) ./ 1  (:39)
        if (rc != CP_OK) return rc;
        if (model->dtype == CP_I64) return run_bisect<int64_t>(A, K, model, Pi, lf, hf, eps, flip, spl_out);
        return run_bisect<double>(A, K, model, Pi, lf, hf, eps, flip, spl_out);
    } CP_CATCH_ALL
}

extern "C" int32_t cp_partition_bisect_cost(cp_csr_t A, int64_t K, const cp_model_t *model, double eps, int32_t flip, int64_t *spl_out)
{
    return cp_partition_bisect_cost_pi(A, K, model, nullptr, eps, flip, spl_out);
}

// partition_stripe(A, K, [Flip]BisectIndexBottleneckSplitter(f))  BisectIndexBottleneckSplitter.jl:5-166
extern "C" int32_t cp_partition_bisect_index_pi(cp_csr_t A, int64_t K, const cp_model_t *model, const cp_rowpart_t *Pi, int32_t flip,
                                                int64_t *spl_out)
{
    try {
        CP_REQUIRE(A && model && spl_out && K >= 1, CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        int64_t li, hi; double lf, hf;
        int32_t rc = cp_bound_stripe_pi(A, K, Pi, model, &li, &hi, &lf, &hf);    // (c_lo, c_hi) = bound_stripe(...) ./ 1  (:39)
        if (rc != CP_OK) return rc;
        if (model->dtype == CP_I64) return run_bisect<int64_t>(A, K, model, Pi, lf, hf, 0.0, flip, spl_out, true);
        return run_bisect<double>(A, K, model, Pi, lf, hf, 0.0, flip, spl_out, true);
    } CP_CATCH_ALL
}

extern "C" int32_t cp_partition_bisect_index(cp_csr_t A, int64_t K, const cp_model_t *model, int32_t flip, int64_t *spl_out)
{
    return cp_partition_bisect_index_pi(A, K, model, nullptr, flip, spl_out);
}
