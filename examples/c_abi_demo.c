/* c_abi_demo.c -- the drop-in boundary used from plain C: no Python, no torch, only include/chainpart.h and
 * libchainpart.so.  Builds a small banded pattern in Julia's layout (1-based colptr / rowval, Int64), then calls the
 * entry points a ChainPartitioners.jl shim would bind:
 *   partition_stripe(A, K, DynamicTotalSplitter(AffineConnectivityModel(0, 0, 0, 1)))      -> cp_partition_dynamic
 *   total_value(A, Phi, mdl)                                                                 -> cp_objective
 *   partition_stripe(A, K, BisectCostBottleneckSplitter(AffineWorkModel(0, 10, 1), 0.01))  -> cp_partition_bisect_cost
 * Build + run (GPU box):  make -C examples && ./examples/c_abi_demo
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "chainpart.h"

int main(void)
{
    if (cp_device_count() <= 0) { fprintf(stderr, "no HIP device: libchainpart has no CPU fallback\n"); return 2; }
    const int64_t n = 2000, m = 2000, hb = 3, K = 8;
    int64_t *colptr = malloc((size_t)(n + 1) * sizeof(int64_t));
    int64_t *rowval = malloc((size_t)(n * (2 * hb + 1)) * sizeof(int64_t));
    int64_t N = 0;
    for (int64_t j = 1; j <= n; j++) {                       /* column j holds rows j-hb .. j+hb */
        colptr[j - 1] = N + 1;
        for (int64_t i = j - hb; i <= j + hb; i++) if (i >= 1 && i <= m && ((i * 7 + j) % 3 != 0 || i == j)) rowval[N++] = i;
    }
    colptr[n] = N + 1;

    cp_csr_t A = NULL;
    int32_t rc = cp_csr_create(m, n, N, colptr, rowval, 0, &A);
    if (rc != CP_OK) { fprintf(stderr, "cp_csr_create: %d %s\n", rc, cp_last_error()); return 1; }

    cp_model_t net; memset(&net, 0, sizeof(net));
    net.kind = CP_MODEL_CONNECTIVITY; net.dtype = CP_I64; net.p_i64[CP_P_NET] = 1;       /* AffineConnectivityModel(0, 0, 0, 1) */
    int64_t spl[9], total = 0; double unused = 0;
    rc = cp_partition_dynamic(A, K, CP_COMBINE_SUM, CP_ORDER_SPLITTER, &net, NULL, NULL, 0, 0.0, spl);
    if (rc != CP_OK) { fprintf(stderr, "cp_partition_dynamic: %d %s\n", rc, cp_last_error()); return 1; }
    rc = cp_objective(A, K, spl, &net, NULL, CP_COMBINE_SUM, &total, &unused);
    if (rc != CP_OK) { fprintf(stderr, "cp_objective: %d %s\n", rc, cp_last_error()); return 1; }
    printf("DynamicTotalSplitter  spl =");
    for (int k = 0; k <= K; k++) printf(" %lld", (long long)spl[k]);
    printf("   total nets = %lld\n", (long long)total);

    cp_model_t work; memset(&work, 0, sizeof(work));
    work.kind = CP_MODEL_WORK; work.dtype = CP_I64; work.p_i64[CP_P_VERTEX] = 10; work.p_i64[CP_P_PIN] = 1;   /* AffineWorkModel(0, 10, 1) */
    int64_t bott = 0;
    rc = cp_partition_bisect_cost(A, K, &work, 0.01, 0, spl);
    if (rc != CP_OK) { fprintf(stderr, "cp_partition_bisect_cost: %d %s\n", rc, cp_last_error()); return 1; }
    rc = cp_objective(A, K, spl, &work, NULL, CP_COMBINE_MAX, &bott, &unused);
    printf("BisectCost(eps=0.01)  spl =");
    for (int k = 0; k <= K; k++) printf(" %lld", (long long)spl[k]);
    printf("   bottleneck = %lld\n", (long long)bott);

    int ok = spl[0] == 1 && spl[K] == n + 1;
    for (int k = 0; k < K; k++) ok &= spl[k] <= spl[k + 1];
    cp_csr_destroy(A);
    free(colptr); free(rowval);
    printf(ok ? "OK\n" : "BAD SPLIT\n");
    return ok ? 0 : 1;
}
