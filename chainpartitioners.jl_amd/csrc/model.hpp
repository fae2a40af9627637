// model.hpp -- device-side evaluation of the cost models, bit-for-bit in the reference's
// evaluation order (left to right, one Int->Float conversion per product, no FMA contraction;
// the translation unit is compiled with -ffp-contract=off):
//   AffineWorkModel           alpha + nv*b_vertex + np*b_pin                       WorkCosts.jl:17
//   AffineConnectivityModel   ... + nn*b_net                                       ConnectivityCosts.jl:20
//   AffineHyperedgeCutModel   ... + nl*b_self_net + (nn-nl)*b_cut_net              HyperedgeCutCosts.jl:21
//   ColumnBlockComponentCostModel  alpha_col(w) + nn*beta_col(w)                   BlockCosts.jl:17
#pragma once
#include "common.hpp"
#include <cmath>
#include <algorithm>

namespace cpk {

template <typename TC> struct CostTraits;
template <> struct CostTraits<int64_t> {
    static constexpr bool is_int = true;
    __host__ __device__ static int64_t typemax() { return INT64_MAX; }
    __host__ __device__ static int64_t typemin() { return INT64_MIN; }
};
template <> struct CostTraits<double> {
    static constexpr bool is_int = false;
    __host__ __device__ static double typemax() { return __builtin_huge_val(); }
    __host__ __device__ static double typemin() { return -__builtin_huge_val(); }
};

// Julia Int64 arithmetic wraps; do integer adds/muls through uint64
__host__ __device__ __forceinline__ int64_t cadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
__host__ __device__ __forceinline__ double cadd(double a, double b) { return a + b; }
__host__ __device__ __forceinline__ int64_t csub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
__host__ __device__ __forceinline__ double csub(double a, double b) { return a - b; }
__host__ __device__ __forceinline__ int64_t cmulc(int64_t cnt, int64_t b) { return (int64_t)((uint64_t)cnt * (uint64_t)b); }
__host__ __device__ __forceinline__ double cmulc(int64_t cnt, double b) { return (double)cnt * b; }
__host__ __device__ __forceinline__ int64_t cmulv(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
__host__ __device__ __forceinline__ double cmulv(double a, double b) { return a * b; }

template <typename TC>
struct DevModel {
    int32_t kind;
    TC p[5];
    const TC *alpha_k;      // device copy or nullptr
    int64_t n_alpha_k;
    // block-column components (COLBLOCK / BLOCK): constants or device tables
    int32_t ac_const, bc_const[CP_MAX_R];
    TC ac_c, bc_c[CP_MAX_R];
    const TC *ac_tab, *bc_tab[CP_MAX_R];
    int64_t ac_len, bc_len[CP_MAX_R];
    int64_t ac_lo, bc_lo[CP_MAX_R];       // width of table entry 0
    int32_t R;
};

template <typename TC>
__host__ __device__ __forceinline__ TC dm_alpha(const DevModel<TC> &m, int64_t k)
{
    if (m.alpha_k && k >= 1 && k <= m.n_alpha_k) return m.alpha_k[k - 1];
    return m.p[CP_P_ALPHA];
}

template <typename TC>
__device__ __forceinline__ TC dm_comp(int32_t is_const, TC c, const TC *tab, int64_t len, int64_t lo, int64_t w)
{
    if (is_const) return c;
    w -= lo;
    if (w < 0) w = 0;
    if (w >= len) w = len - 1;     // the host mirror sizes the tables for every width a method can evaluate
    return tab[w];
}

// alpha + (nv*b_vertex + np*b_pin)^gamma  (test_Partitioners.jl:60,70); Float64 models only
__device__ __forceinline__ double pw_apply(double alpha, const double *p, int64_t nv, int64_t np)
{
    double x = (double)nv * p[CP_P_VERTEX] + (double)np * p[CP_P_PIN];
    double g = p[CP_P_GAMMA];
    return alpha + (g == 2.0 ? x * x : pow(x, g));
}
__device__ __forceinline__ int64_t pw_apply(int64_t, const int64_t *, int64_t, int64_t) { return 0; }

// model applied to counts; alpha is resolved by the caller (per-part alpha[k] or scalar)
template <typename TC>
__device__ __forceinline__ TC dm_apply(const DevModel<TC> &m, TC alpha, int64_t nv, int64_t np, int64_t nn, int64_t nl)
{
    switch (m.kind) {
    case CP_MODEL_WORK:
        return cadd(cadd(alpha, cmulc(nv, m.p[CP_P_VERTEX])), cmulc(np, m.p[CP_P_PIN]));
    case CP_MODEL_CONNECTIVITY:
        return cadd(cadd(cadd(alpha, cmulc(nv, m.p[CP_P_VERTEX])), cmulc(np, m.p[CP_P_PIN])), cmulc(nn, m.p[CP_P_NET]));
    case CP_MODEL_HYPEREDGE_CUT:
        return cadd(cadd(cadd(cadd(alpha, cmulc(nv, m.p[CP_P_VERTEX])), cmulc(np, m.p[CP_P_PIN])),
                         cmulc(nl, m.p[CP_P_SELF_NET])), cmulc(nn - nl, m.p[CP_P_CUT_NET]));
    case CP_MODEL_COLBLOCK:
        return cadd(dm_comp(m.ac_const, m.ac_c, m.ac_tab, m.ac_len, m.ac_lo, nv),
                    cmulc(nn, dm_comp(m.bc_const[0], m.bc_c[0], m.bc_tab[0], m.bc_len[0], m.bc_lo[0], nv)));
    case CP_MODEL_VERTEX_COUNT:
        return (TC)nv;
    case CP_MODEL_POWER_WORK:
        return pw_apply(alpha, m.p, nv, np);
    case CP_MODEL_PRIMARY: case CP_MODEL_SECONDARY:     // nn = local nets, nl = remote nets (PrimaryConnectivityCosts.jl:20)
        return cadd(cadd(cadd(cadd(alpha, cmulc(nv, m.p[CP_P_VERTEX])), cmulc(np, m.p[CP_P_PIN])), cmulc(nn, m.p[CP_P_LOCAL_NET])),
                    cmulc(nl, m.p[CP_P_REMOTE_NET]));
    default:
        return (TC)0;
    }
}

template <typename TC>
__device__ __forceinline__ TC comb(int32_t g, TC a, TC b)
{
    if (g == CP_COMBINE_SUM) return cadd(a, b);
    return a > b ? a : b;
}

// Int64 model, or a Float64 model whose scalar parameters are all integer-valued (then every cost is an exactly
// represented integer and sums are associative below 2^53)
inline bool model_all_integral(const cp_model_t *m)
{
    if (m->dtype == CP_I64) return true;
    if (m->kind == CP_MODEL_POWER_WORK) return false;
    for (int i = 0; i < 5; i++) if (std::floor(m->p_f64[i]) != m->p_f64[i] || std::fabs(m->p_f64[i]) > 9e15) return false;
    if (m->alpha_k) for (int64_t i = 0; i < m->n_alpha_k; i++) { double v = ((const double *)m->alpha_k)[i]; if (std::floor(v) != v) return false; }
    return true;
}

// ... and bounded so: the exact-arithmetic paths (O(n log^2 n) DP, (min,+) chunk scan) reassociate sums and argue about ties, which
// is only the reference's sequential Float64 arithmetic while every cost and every running total stays below 2^53.  Largest
// reachable total of a K-part partition of an n-column, N-nonzero pattern: K*|alpha| + n*|b_vertex| + N*|b_pin| + N*max|b_net-like|.
inline bool model_exact_on(const cp_model_t *m, int64_t n, int64_t N, int64_t K)
{
    if (!model_all_integral(m)) return false;
    if (m->dtype == CP_I64) return true;
    double amax = std::fabs(m->p_f64[CP_P_ALPHA]);
    if (m->alpha_k) for (int64_t i = 0; i < m->n_alpha_k; i++) amax = std::max(amax, std::fabs(((const double *)m->alpha_k)[i]));
    double bnet = std::max(std::fabs(m->p_f64[3]), std::fabs(m->p_f64[4]));
    double bound = amax * (double)(K > 0 ? K : 1) + std::fabs(m->p_f64[CP_P_VERTEX]) * (double)n + std::fabs(m->p_f64[CP_P_PIN]) * (double)N + bnet * (double)N;
    return bound < 9007199254740992.0;       // 2^53
}

// host: build a DevModel from the C-ABI struct, uploading alpha_k / tables
template <typename TC>
struct HostModel {
    DevModel<TC> d;
    DBuf<TC> alpha_k;
    DBuf<TC> tabs[2 + 2 * CP_MAX_R];
};

template <typename TC> inline TC model_param(const cp_model_t *m, int i);
template <> inline int64_t model_param<int64_t>(const cp_model_t *m, int i) { return m->p_i64[i]; }
template <> inline double model_param<double>(const cp_model_t *m, int i) { return m->p_f64[i]; }
template <typename TC> inline TC comp_const(const cp_component_t &c);
template <> inline int64_t comp_const<int64_t>(const cp_component_t &c) { return c.c_i64; }
template <> inline double comp_const<double>(const cp_component_t &c) { return c.c_f64; }

template <typename TC>
void build_dev_model(const cp_model_t *m, HostModel<TC> &H, hipStream_t s)
{
    memset(&H.d, 0, sizeof(H.d));
    H.d.kind = m->kind;
    for (int i = 0; i < 5; i++) H.d.p[i] = model_param<TC>(m, i);
    if (m->alpha_k && m->n_alpha_k > 0) {
        H.alpha_k.alloc((size_t)m->n_alpha_k);
        CP_HIP(hipMemcpyAsync(H.alpha_k.p, m->alpha_k, sizeof(TC) * (size_t)m->n_alpha_k, hipMemcpyHostToDevice, s));
        H.d.alpha_k = H.alpha_k.p;
        H.d.n_alpha_k = m->n_alpha_k;
    }
    H.d.R = m->R;
    auto up = [&](const cp_component_t &c, int slot, int32_t &is_c, TC &cc, const TC *&tab, int64_t &len, int64_t &lo) {
        is_c = c.is_const;
        cc = comp_const<TC>(c);
        tab = nullptr; len = 0; lo = c.lo;
        if (!c.is_const && c.table && c.len > 0) {
            H.tabs[slot].alloc((size_t)c.len);
            CP_HIP(hipMemcpyAsync(H.tabs[slot].p, c.table, sizeof(TC) * (size_t)c.len, hipMemcpyHostToDevice, s));
            tab = H.tabs[slot].p; len = c.len;
        }
    };
    if (m->kind == CP_MODEL_COLBLOCK || m->kind == CP_MODEL_BLOCK) {
        up(m->alpha_col, 0, H.d.ac_const, H.d.ac_c, H.d.ac_tab, H.d.ac_len, H.d.ac_lo);
        for (int r = 0; r < m->R && r < CP_MAX_R; r++)
            up(m->beta_col[r], 1 + r, H.d.bc_const[r], H.d.bc_c[r], H.d.bc_tab[r], H.d.bc_len[r], H.d.bc_lo[r]);
    }
}

}  // namespace cpk
