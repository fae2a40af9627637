/*
 * orc_api.c -- TEST ORACLE (not product code): dtype dispatch for the entry points of
 * orc.h, plus EquiSplitter / EquiChunker (EquiPartitioner.jl:3-22).
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

#define DECL(T, S)                                                                                         \
    int32_t orc_api_oracle_eval##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,            \
                                   const cp_model_t *, const cp_rowpart_t *, int32_t, int64_t,             \
                                   const int64_t *, const int64_t *, const int64_t *, T *);                \
    int32_t orc_api_oracle_step##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,            \
                                   const cp_model_t *, const cp_rowpart_t *, int64_t, const int32_t *,     \
                                   const int64_t *, const int32_t *, const int64_t *, const int64_t *, T *); \
    int32_t orc_api_bound_stripe##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *, int64_t,  \
                                    const cp_model_t *, T *, T *);                                         \
    int32_t orc_api_bound_stripe_pi##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *, int64_t,   \
                                       const cp_rowpart_t *, const cp_model_t *, T *, T *);                    \
    int32_t orc_api_objective##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *, int64_t,     \
                                 const int64_t *, const cp_model_t *, const cp_rowpart_t *, int32_t, T *); \
    int32_t orc_api_partition_dynamic##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,      \
                                         int64_t, int32_t, int32_t, const cp_model_t *,                    \
                                         const cp_rowpart_t *, const cp_model_t *, int64_t, double,        \
                                         int64_t *);                                                       \
    int32_t orc_api_dynamic_tables##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,         \
                                      int64_t, int32_t, const cp_model_t *, const cp_rowpart_t *,          \
                                      int64_t *, T *);                                                     \
    int32_t orc_api_dynamic_tables_constrained##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *, int64_t, int32_t, \
                                                  const cp_model_t *, const cp_rowpart_t *, const cp_model_t *, int64_t, double,    \
                                                  int64_t *, int64_t *, int64_t *, T *);                                            \
    int32_t orc_api_pack_dynamic##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,           \
                                    const cp_model_t *, const cp_rowpart_t *, const cp_model_t *, int64_t, \
                                    double, int64_t *, int64_t *);                                         \
    int32_t orc_api_partition_bisect_cost##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,  \
                                             int64_t, const cp_model_t *, const cp_rowpart_t *, double,    \
                                             int32_t, int64_t *, int64_t *);                                                   \
    int32_t orc_api_pack_convex##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,            \
                                   const cp_model_t *, const cp_rowpart_t *, const cp_model_t *, int64_t,  \
                                   double, int64_t *, int64_t *);                                          \
    int32_t orc_api_partition_convex##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,       \
                                        int64_t, const cp_model_t *, const cp_rowpart_t *,                 \
                                        const cp_model_t *, int64_t, double, int64_t *);               \
    int32_t orc_api_pack_concave##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,           \
                                    const cp_model_t *, const cp_rowpart_t *, const cp_model_t *, int64_t, \
                                    double, int64_t *, int64_t *);                                         \
    int32_t orc_api_partition_concave##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *,      \
                                         int64_t, const cp_model_t *, const cp_rowpart_t *,                \
                                         const cp_model_t *, int64_t, double, int64_t *);                  \
    int32_t orc_api_partition_bisect_index##S(int64_t, int64_t, int64_t, const int64_t *, const int64_t *, \
                                              int64_t, const cp_model_t *, const cp_rowpart_t *, int32_t,  \
                                              int64_t *, int64_t *);                                       \
    int32_t orc_api_partition_lazy_bisect_cost##S(int64_t, int64_t, int64_t, const int64_t *,              \
                                                  const int64_t *, int64_t, const cp_model_t *, double,    \
                                                  int64_t *, int64_t *);
DECL(int64_t, _i64)
DECL(double, _f64)

#define IS_I(m) ((m)->dtype == CP_I64)

int32_t orc_oracle_eval(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                        const cp_model_t *mdl, const cp_rowpart_t *Pi, int32_t hint,
                        int64_t nq, const int64_t *j, const int64_t *jp, const int64_t *k,
                        int64_t *out_i64, double *out_f64)
{
    return IS_I(mdl) ? orc_api_oracle_eval_i64(m, n, N, pos, idx, mdl, Pi, hint, nq, j, jp, k, out_i64)
                     : orc_api_oracle_eval_f64(m, n, N, pos, idx, mdl, Pi, hint, nq, j, jp, k, out_f64);
}

int32_t orc_bound_stripe(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                         int64_t K, const cp_model_t *mdl,
                         int64_t *lo_i64, int64_t *hi_i64, double *lo_f64, double *hi_f64)
{
    if (IS_I(mdl)) {
        int32_t rc = orc_api_bound_stripe_i64(m, n, N, pos, idx, K, mdl, lo_i64, hi_i64);
        if (rc == CP_OK) { *lo_f64 = (double)*lo_i64; *hi_f64 = (double)*hi_i64; }
        return rc;
    }
    return orc_api_bound_stripe_f64(m, n, N, pos, idx, K, mdl, lo_f64, hi_f64);
}

int32_t orc_bound_stripe_pi(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                            int64_t K, const cp_rowpart_t *Pi, const cp_model_t *mdl,
                            int64_t *lo_i64, int64_t *hi_i64, double *lo_f64, double *hi_f64)
{
    if (IS_I(mdl)) {
        int32_t rc = orc_api_bound_stripe_pi_i64(m, n, N, pos, idx, K, Pi, mdl, lo_i64, hi_i64);
        if (rc == CP_OK) { *lo_f64 = (double)*lo_i64; *hi_f64 = (double)*hi_i64; }
        return rc;
    }
    return orc_api_bound_stripe_pi_f64(m, n, N, pos, idx, K, Pi, mdl, lo_f64, hi_f64);
}

int32_t orc_objective(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                      int64_t K, const int64_t *spl, const cp_model_t *mdl, const cp_rowpart_t *Pi,
                      int32_t combine, int64_t *out_i64, double *out_f64)
{
    return IS_I(mdl) ? orc_api_objective_i64(m, n, N, pos, idx, K, spl, mdl, Pi, combine, out_i64)
                     : orc_api_objective_f64(m, n, N, pos, idx, K, spl, mdl, Pi, combine, out_f64);
}

int32_t orc_partition_dynamic(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                              int64_t K, int32_t combine, int32_t order,
                              const cp_model_t *mdl, const cp_rowpart_t *Pi,
                              const cp_model_t *weight, int64_t wmax_i64, double wmax_f64, int64_t *spl_out)
{
    return IS_I(mdl) ? orc_api_partition_dynamic_i64(m, n, N, pos, idx, K, combine, order, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out)
                     : orc_api_partition_dynamic_f64(m, n, N, pos, idx, K, combine, order, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out);
}

int32_t orc_dynamic_tables(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                           int64_t K, int32_t combine, const cp_model_t *mdl, const cp_rowpart_t *Pi,
                           int64_t *ptr_out, int64_t *cst_i64, double *cst_f64)
{
    return IS_I(mdl) ? orc_api_dynamic_tables_i64(m, n, N, pos, idx, K, combine, mdl, Pi, ptr_out, cst_i64)
                     : orc_api_dynamic_tables_f64(m, n, N, pos, idx, K, combine, mdl, Pi, ptr_out, cst_f64);
}

int32_t orc_oracle_step(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                        const cp_model_t *mdl, const cp_rowpart_t *Pi, int64_t nq, const int32_t *move_j, const int64_t *j,
                        const int32_t *move_jp, const int64_t *jp, const int64_t *k, int64_t *out_i64, double *out_f64)
{
    return IS_I(mdl) ? orc_api_oracle_step_i64(m, n, N, pos, idx, mdl, Pi, nq, move_j, j, move_jp, jp, k, out_i64)
                     : orc_api_oracle_step_f64(m, n, N, pos, idx, mdl, Pi, nq, move_j, j, move_jp, jp, k, out_f64);
}

int32_t orc_dynamic_tables_constrained(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                                       int64_t K, int32_t combine, const cp_model_t *mdl, const cp_rowpart_t *Pi,
                                       const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                                       int64_t *win_lo, int64_t *win_hi, int64_t *ptr_out, int64_t *cst_i64, double *cst_f64)
{
    return IS_I(mdl) ? orc_api_dynamic_tables_constrained_i64(m, n, N, pos, idx, K, combine, mdl, Pi, weight, wmax_i64, wmax_f64, win_lo, win_hi, ptr_out, cst_i64)
                     : orc_api_dynamic_tables_constrained_f64(m, n, N, pos, idx, K, combine, mdl, Pi, weight, wmax_i64, wmax_f64, win_lo, win_hi, ptr_out, cst_f64);
}

int32_t orc_pack_dynamic(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                         const cp_model_t *mdl, const cp_rowpart_t *Pi,
                         const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                         int64_t *spl_out, int64_t *K_out)
{
    return IS_I(mdl) ? orc_api_pack_dynamic_i64(m, n, N, pos, idx, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out)
                     : orc_api_pack_dynamic_f64(m, n, N, pos, idx, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out);
}

int32_t orc_partition_bisect_cost(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                                  int64_t K, const cp_model_t *mdl, const cp_rowpart_t *Pi, double eps, int32_t flip,
                                  int64_t *spl_out, int64_t *n_probes_out)
{
    return IS_I(mdl) ? orc_api_partition_bisect_cost_i64(m, n, N, pos, idx, K, mdl, Pi, eps, flip, spl_out, n_probes_out)
                     : orc_api_partition_bisect_cost_f64(m, n, N, pos, idx, K, mdl, Pi, eps, flip, spl_out, n_probes_out);
}

int32_t orc_pack_convex(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                        const cp_model_t *mdl, const cp_rowpart_t *Pi,
                        const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                        int64_t *spl_out, int64_t *K_out)
{
    return IS_I(mdl) ? orc_api_pack_convex_i64(m, n, N, pos, idx, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out)
                     : orc_api_pack_convex_f64(m, n, N, pos, idx, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out);
}

int32_t orc_partition_convex(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                             int64_t K, const cp_model_t *mdl, const cp_rowpart_t *Pi,
                             const cp_model_t *weight, int64_t wmax_i64, double wmax_f64, int64_t *spl_out)
{
    return IS_I(mdl) ? orc_api_partition_convex_i64(m, n, N, pos, idx, K, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out)
                     : orc_api_partition_convex_f64(m, n, N, pos, idx, K, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out);
}

int32_t orc_partition_bisect_index(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                                   int64_t K, const cp_model_t *mdl, const cp_rowpart_t *Pi, int32_t flip, int64_t *spl_out,
                                   int64_t *n_probes_out)
{
    return IS_I(mdl) ? orc_api_partition_bisect_index_i64(m, n, N, pos, idx, K, mdl, Pi, flip, spl_out, n_probes_out)
                     : orc_api_partition_bisect_index_f64(m, n, N, pos, idx, K, mdl, Pi, flip, spl_out, n_probes_out);
}

int32_t orc_partition_lazy_bisect_cost(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                                       int64_t K, const cp_model_t *mdl, double eps, int64_t *spl_out, int64_t *n_probes_out)
{
    return IS_I(mdl) ? orc_api_partition_lazy_bisect_cost_i64(m, n, N, pos, idx, K, mdl, eps, spl_out, n_probes_out)
                     : orc_api_partition_lazy_bisect_cost_f64(m, n, N, pos, idx, K, mdl, eps, spl_out, n_probes_out);
}

int32_t orc_pack_concave(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                         const cp_model_t *mdl, const cp_rowpart_t *Pi,
                         const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                         int64_t *spl_out, int64_t *K_out)
{
    return IS_I(mdl) ? orc_api_pack_concave_i64(m, n, N, pos, idx, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out)
                     : orc_api_pack_concave_f64(m, n, N, pos, idx, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out, K_out);
}

int32_t orc_partition_concave(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                              int64_t K, const cp_model_t *mdl, const cp_rowpart_t *Pi,
                              const cp_model_t *weight, int64_t wmax_i64, double wmax_f64, int64_t *spl_out)
{
    return IS_I(mdl) ? orc_api_partition_concave_i64(m, n, N, pos, idx, K, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out)
                     : orc_api_partition_concave_f64(m, n, N, pos, idx, K, mdl, Pi, weight, wmax_i64, wmax_f64, spl_out);
}

/* EquiPartitioner.jl:7 : spl[k] = (k-1)*fld(n,K) + min(n % K, k-1) + 1, k = 1..K+1 */
void orc_partition_equi(int64_t n, int64_t K, int64_t *spl_out)
{
    for (int64_t k = 1; k <= K + 1; k++) {
        int64_t r = n % K, km1 = k - 1;
        spl_out[k - 1] = km1 * (n / K) + (r < km1 ? r : km1) + 1;
    }
}

/* EquiPartitioner.jl:20 : [1:w:n; n+1], K = cld(n, w) */
int64_t orc_pack_equi(int64_t n, int64_t w, int64_t *spl_out)
{
    int64_t K = 0;
    for (int64_t j = 1; j <= n; j += w) spl_out[K++] = j;
    spl_out[K] = n + 1;
    return K;
}
