/*
 * chainpart.h -- C ABI of libchainpart.so, the MI355X (gfx950) engine for the contiguous
 * partitioning hot path of ChainPartitioners.jl.
 *
 * The reference has no FFI layer (it is pure Julia; the boundary is multiple dispatch).
 * These are the entry points a `ccall` shim binds to replace the reference methods named
 * beside each declaration (paths under /root/reference/src).  INTEGRATION.md shows the shim.
 *
 * Conventions
 *   - plain pointers and sizes only; all index VALUES 1-based exactly as Julia stores them;
 *     arrays are caller-owned host memory unless the name says _device;
 *   - every function returns a CP_* status (chainpart_types.h); cp_last_error() gives text;
 *   - calls are synchronous (return after the stream is idle); handles are opaque,
 *     single-owner, not thread-safe, freed by the matching *_destroy;
 *   - there is NO CPU fallback: without a HIP device every compute entry returns CP_EHIP.
 */
#ifndef CHAINPART_H
#define CHAINPART_H

#include "chainpart_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cp_csr_s *cp_csr_t;        /* device-resident sparsity pattern + link arrays */
typedef struct cp_count_s *cp_count_t;    /* device-resident counting structure */

const char *cp_last_error(void);
int32_t cp_version(void);
/* number of visible HIP devices (0 if none); does not initialise a context */
int32_t cp_device_count(void);

/* ---- residency of A.colptr / A.rowval in HBM (SparseMatrixCSC fields; the reference reads
 * them in every oracle constructor, e.g. SparseColorArrays.jl:101-118) ---- */
int32_t cp_csr_create(int64_t m, int64_t n, int64_t N, const int64_t *colptr, const int64_t *rowval,
                      int32_t device, cp_csr_t *out);
/* same, but colptr/rowval already live in device memory (1-based int64, as Julia would upload them) */
int32_t cp_csr_create_device(int64_t m, int64_t n, int64_t N, const int64_t *colptr_device,
                             const int64_t *rowval_device, int32_t device, cp_csr_t *out);
int32_t cp_csr_destroy(cp_csr_t csr);
/* adjointpattern(A) (util.jl:67-95): the transposed pattern (n x m, rows of every column ascending) as a NEW
 * device-resident handle -- the counting sort the reference runs is the (row, column) order the link-array build
 * already produces.  First block of SURVEY 8(f)-4 (the 2-D callers partition A and its adjoint alternately). */
int32_t cp_adjoint(cp_csr_t csr, cp_csr_t *out);
/* copy a handle's pattern back to the host in Julia's layout: colptr int64[n+1], rowval int64[nnz], 1-based;
 * dims_out = {m, n, nnz} (call with NULL arrays to query the sizes first) */
int32_t cp_csr_download(cp_csr_t csr, int64_t *dims_out, int64_t *colptr_out, int64_t *rowval_out);
/* drop cached derived structures (link arrays, counters) so the next call rebuilds them:
 * lets a benchmark time "one partition_stripe call including oracle construction" */
int32_t cp_csr_reset_cache(cp_csr_t csr);

/* ---- counting structures ----
 * dominancecount(hint, A)   SparsePrefixMatrices.jl:438-458   kind CP_COUNT_DOM : C[i,j]
 * netcount(hint, A)         SparseColorArrays.jl:57-58,101-125 kind CP_COUNT_NET : net[j,j']
 * selfnetcount(hint, A)     SparseColorArrays.jl:165-229       kind CP_COUNT_SELFNET
 * One exact device structure (wavelet bit-vectors) serves every hint: query results are
 * integers independent of the structure. */
#define CP_COUNT_DOM     0
#define CP_COUNT_NET     1
#define CP_COUNT_SELFNET 2
int32_t cp_count_build(cp_csr_t csr, int32_t kind, int32_t hint, cp_count_t *out);
int32_t cp_count_query(cp_count_t h, int64_t nq, const int64_t *a, const int64_t *b, int64_t *out);
int32_t cp_count_destroy(cp_count_t h);
/* the NetCount link array idx'[q] = (n+1) - hst[i] (SparseColorArrays.jl:106-113), for tests */
int32_t cp_link_array(cp_csr_t csr, int64_t *out /* N */);

/* ---- weighted dominance (used by no cost oracle of the path, kept for the reference's AbstractMatrix API) ----
 * dominancesum(hint, A)                 SparsePrefixMatrices.jl:1-250   S[i, j] = sum(A[1:i-1, 1:j-1])  (test_SparsePrefixMatrices.jl:15)
 * rookcount!(hint, N, idx) / rooksum!(hint, N, idx, val)  :825-1273   the same over ONE point (idx[j], j) per column
 * val: nnz (rook: N) 8-byte weights in the pattern's entry order -- CP_I64: Int / UInt words, summed with wrap-around exactly as
 * Julia does; CP_F64: Float64 (a prefix-sum structure like the reference's `scn`: equal up to rounding, not bit for bit).
 * One query returns the count (rows < i among the entries of the columns < j) and the weighted sum. */
typedef struct cp_wsum_s *cp_wsum_t;
int32_t cp_domsum_build(cp_csr_t csr, int32_t dtype, const void *val, cp_wsum_t *out);
int32_t cp_rook_build(int64_t N, const int64_t *idx, int32_t dtype, const void *val /* NULL: counts only */, int32_t device, cp_wsum_t *out);
int32_t cp_wsum_query(cp_wsum_t h, int64_t nq, const int64_t *i, const int64_t *j, int64_t *count_out /* may be NULL */,
                      int64_t *sum_i64, double *sum_f64);
int32_t cp_wsum_destroy(cp_wsum_t h);

/* partwise(A, Pi) PartwiseCounts.jl:1-60 */
int32_t cp_partwise(cp_csr_t csr, int64_t K, const int64_t *asg, int64_t *nprime_out,
                    int64_t *pios_out /* K+1 */, int64_t *prm_out /* <= N */,
                    int64_t *pos_out /* <= N+1 */, int64_t *idx_out /* N */);

/* ---- cost oracles ----
 * ocl(j, j', k...) for a batch  (WorkCosts.jl:30-35, ConnectivityCosts.jl:58-64,
 * HyperedgeCutCosts.jl:44-51, BlockCosts.jl:66-142); k may be NULL */
int32_t cp_oracle_eval(cp_csr_t csr, const cp_model_t *model, const cp_rowpart_t *Pi, int32_t hint,
                       int64_t nq, const int64_t *j, const int64_t *jp, const int64_t *k,
                       int64_t *out_i64, double *out_f64);
/* Step(ocl)(move_j(j), move_j'(j'), Same(k)) along a walk of nq calls (Costs.jl:174-195; the specialised methods
 * ConnectivityCosts.jl:66-76, HyperedgeCutCosts.jl:53-64, SparseColorArrays.jl:127-152, 231-256): move codes CP_MOVE_*.
 * A move is the caller's promise about the previous call's position (Same: equal, Next: +1, Prev: -1, Jump: anything); the
 * reference's stepwise structures rely on it, the device counters are random-access and do not -- so the promise is CHECKED
 * (CP_EINVAL on a broken one; the first call may carry any move) and the value is ocl(j, j', k), which is what every Step
 * method of the reference returns (Costs.jl:195). */
int32_t cp_oracle_step(cp_csr_t csr, const cp_model_t *model, const cp_rowpart_t *Pi, int64_t nq,
                       const int32_t *move_j, const int64_t *j, const int32_t *move_jp, const int64_t *jp,
                       const int64_t *k, int64_t *out_i64, double *out_f64);
/* bound_stripe(A, K, mdl)  WorkCosts.jl:37-51, ConnectivityCosts.jl:22-35 */
int32_t cp_bound_stripe(cp_csr_t csr, int64_t K, const cp_model_t *model,
                        int64_t *lo_i64, int64_t *hi_i64, double *lo_f64, double *hi_f64);
/* bound_stripe(A, K, Pi, mdl) (Costs.jl:17-19): Pi only matters to CP_MODEL_SECONDARY (SecondaryConnectivityCosts.jl:21-31) */
int32_t cp_bound_stripe_pi(cp_csr_t csr, int64_t K, const cp_rowpart_t *Pi, const cp_model_t *model,
                           int64_t *lo_i64, int64_t *hi_i64, double *lo_f64, double *hi_f64);
/* total_value / bottleneck_value  Costs.jl:26-66 */
int32_t cp_objective(cp_csr_t csr, int64_t K, const int64_t *spl, const cp_model_t *model,
                     const cp_rowpart_t *Pi, int32_t combine, int64_t *out_i64, double *out_f64);

/* ---- partitioners ---- */
/* partition_stripe(A, K, Dynamic{Total,Bottleneck}{Splitter,Chunker}(f | ConstrainedCost(f,w,w_max)), [Pi])
 * DynamicSplitter.jl:15-50 (order SPLITTER), :52-87 (order CHUNKER), :206-314 (constrained);
 * Reference{Total,Bottleneck}Splitter (ReferenceSplitter.jl:1-13) are the same entry. */
int32_t cp_partition_dynamic(cp_csr_t csr, int64_t K, int32_t combine, int32_t order,
                             const cp_model_t *model, const cp_rowpart_t *Pi,
                             const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                             int64_t *spl_out /* K+1 */);
/* pack_stripe(A, DynamicTotalChunker(f | ConstrainedCost(f,w,w_max)), [Pi])  DynamicChunker.jl:15-75 */
int32_t cp_pack_dynamic(cp_csr_t csr, const cp_model_t *model, const cp_rowpart_t *Pi,
                        const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                        int64_t *spl_out /* n+1 */, int64_t *K_out);
/* partition_stripe(A, K, [Flip]BisectCostBottleneckSplitter(f, eps))  BisectCostBottleneckSplitter.jl:6-127 */
int32_t cp_partition_bisect_cost(cp_csr_t csr, int64_t K, const cp_model_t *model, double eps,
                                 int32_t flip, int64_t *spl_out /* K+1 */);
/* B independent BisectCostBottleneckSplitter partitions of ONE pattern in one launch (BisectCostBottleneckSplitter.jl:6-63 per request
 * b: K[b], models[b] (Work or Connectivity, one element type for the batch), eps[b], flip[b] (NULL: 0)): the probe chain of a single
 * partition is sequential and fills one wave; a sweep over K / eps / model constants fills the chip and shares the counting
 * structure.  Row b of spl_out (ld >= max K + 1 entries per row) holds the K[b] + 1 split indices of request b -- exactly what
 * cp_partition_bisect_cost returns for it. */
int32_t cp_partition_bisect_cost_batch(cp_csr_t csr, int64_t B, const int64_t *K, const cp_model_t *models, const double *eps,
                                       const int32_t *flip, int64_t ld, int64_t *spl_out);
/* the same two with the row partition the plaid cost models need: partition_stripe(A, K, method, Pi)
 * (CP_MODEL_PRIMARY / CP_MODEL_SECONDARY; Pi is ignored by every other model, Costs.jl:5-7) */
int32_t cp_partition_bisect_cost_pi(cp_csr_t csr, int64_t K, const cp_model_t *model, const cp_rowpart_t *Pi, double eps,
                                    int32_t flip, int64_t *spl_out /* K+1 */);
int32_t cp_partition_bisect_index_pi(cp_csr_t csr, int64_t K, const cp_model_t *model, const cp_rowpart_t *Pi, int32_t flip,
                                     int64_t *spl_out /* K+1 */);
/* pack_stripe(A, ConcaveTotalChunker(f | ConstrainedCost(f,w,w_max)), [Pi]) ConcaveTotalChunker.jl:9-24, chunk_concave! :57-114;
 * partition_stripe(A, K, ConcaveTotalSplitter(..), [Pi]) :26-55 and the ConstrainedCost method :140-180.
 * The candidate-deque algorithm is exact for concave (Monge) costs, e.g. CP_MODEL_POWER_WORK with gamma >= 1. */
int32_t cp_pack_concave(cp_csr_t csr, const cp_model_t *model, const cp_rowpart_t *Pi,
                        const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                        int64_t *spl_out /* n+1 */, int64_t *K_out);
int32_t cp_partition_concave(cp_csr_t csr, int64_t K, const cp_model_t *model, const cp_rowpart_t *Pi,
                             const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                             int64_t *spl_out /* K+1 */);
/* partition_stripe(A, K, [Flip]BisectIndexBottleneckSplitter(f))  BisectIndexBottleneckSplitter.jl:5-83, :85-166
 * (exact bottleneck: bisection over the split index of every part in turn) */
int32_t cp_partition_bisect_index(cp_csr_t csr, int64_t K, const cp_model_t *model, int32_t flip,
                                  int64_t *spl_out /* K+1 */);
/* partition_stripe(A, K, LazyBisectCostBottleneckSplitter(f::AbstractConnectivityModel, eps))
 * LazyBisectCostBottleneckSplitter.jl:140-258 (forward-scan probes over the link array).
 * Models that are not connectivity models return CP_EINVAL (the reference's generic method asserts, :486-501). */
int32_t cp_partition_lazy_bisect_cost(cp_csr_t csr, int64_t K, const cp_model_t *model, double eps,
                                      int64_t *spl_out /* K+1 */);
/* pack_stripe(A, ConvexTotalChunker(..), [Pi]) / partition_stripe(A, K, ConvexTotalSplitter(..), [Pi])
 * ConvexTotalChunker.jl:9-265 */
int32_t cp_pack_convex(cp_csr_t csr, const cp_model_t *model, const cp_rowpart_t *Pi,
                       const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                       int64_t *spl_out /* n+1 */, int64_t *K_out);
/* B independent pack_stripe(A, ConvexTotalChunker(ConstrainedCost(models[b], VertexCount(), wmax[b]))) calls on ONE pattern in one
 * launch (ConvexTotalChunker.jl:141-168, 211-265 per request): the stack algorithm is a dependent chain that occupies one wave
 * whatever the size of the matrix; a sweep over the cost constants / width limits occupies one wave per request, and the requests
 * share the net counter and the window table of net counts.  Requests: ColumnBlock / Connectivity / Work models (one element type,
 * no per-part alpha), 1 <= wmax[b] <= 15.  Row b of spl_out (ld entries per row; n + 1 always suffices) holds the K_out[b] + 1 chunk
 * boundaries of request b -- exactly what cp_pack_convex returns for it. */
int32_t cp_pack_convex_batch(cp_csr_t csr, int64_t B, const cp_model_t *models, const int64_t *wmax, int64_t ld, int64_t *spl_out, int64_t *K_out);
int32_t cp_partition_convex(cp_csr_t csr, int64_t K, const cp_model_t *model, const cp_rowpart_t *Pi,
                            const cp_model_t *weight, int64_t wmax_i64, double wmax_f64,
                            int64_t *spl_out /* K+1 */);
/* EquiSplitter / EquiChunker  EquiPartitioner.jl:3-22 (closed forms, host arithmetic) */
int32_t cp_partition_equi(int64_t n, int64_t K, int64_t *spl_out /* K+1 */);
int32_t cp_pack_equi(int64_t n, int64_t w, int64_t *spl_out /* cld(n,w)+1 */, int64_t *K_out);

/* full (n+1) x K tables of the splitter-order DP, column-major like the reference's cst/ptr
 * (DynamicSplitter.jl:23-24), for table-level parity tests.  Layer K holds row n+1 only. */
int32_t cp_dynamic_tables(cp_csr_t csr, int64_t K, int32_t combine, const cp_model_t *model,
                          const cp_rowpart_t *Pi, int64_t *ptr_out, int64_t *cst_i64, double *cst_f64);

/* the same window on the ConstrainedCost splitter with the width weight (DynamicSplitter.jl:206-258,
 * DynamicTotalSplitter(ConstrainedCost(f, VertexCount(), w_max))): win_lo / win_hi receive j'_lo[k] / j'_hi[k] of
 * column_constraints (:144-172), the tables are written densely ((n+1) x K column-major; a cell outside its window holds what
 * the reference's WindowConstrainedMatrix returns for it, 0 / typemax, :127-134).  Infeasible windows: CP_INFEASIBLE.
 * Only the models the O(K n log^2 n) path takes (else CP_EUNSUPPORTED). */
int32_t cp_dynamic_tables_constrained(cp_csr_t csr, int64_t K, const cp_model_t *model, int64_t wmax,
                                      int64_t *win_lo /* K */, int64_t *win_hi /* K */,
                                      int64_t *ptr_out, int64_t *cst_i64, double *cst_f64);
/* ... with the combine op and the weight given (NULL: VertexCount).  CP_COMBINE_MAX is DynamicBottleneckSplitter(ConstrainedCost(f,
 * w, w_max)) (DynamicSplitter.jl:206-258 with g = max) on the valley search with candidate limits: Int64 cost models; width
 * weights (VertexCount, AffineWorkModel(alpha, c, 0)) or any AffineWorkModel weight with b_v, b_p >= 0 (pins per part).
 * CP_COMBINE_SUM takes width weights only.  w_max in the weight's element type, as in cp_partition_dynamic. */
int32_t cp_dynamic_tables_constrained_combine(cp_csr_t csr, int64_t K, int32_t combine, const cp_model_t *model, const cp_model_t *weight,
                                              int64_t wmax_i64, double wmax_f64, int64_t *win_lo /* K */, int64_t *win_hi /* K */,
                                              int64_t *ptr_out, int64_t *cst_i64, double *cst_f64);

/* ---- row-tiled DP: one process per GPU, the layer's cost vector is completed by the caller's collective ----
 * The rows j' of every DP layer (DynamicSplitter.jl:33-46: all cst[j',k] of a layer depend only on layer k-1) are
 * tiled contiguously over the ranks.  Rank g calls cp_dp_begin with its tile [row_lo, row_hi) (1-based, half-open,
 * tiles cover 1..n+1), then for k = 1..K: cp_dp_layer(dp, k, prev, cur) -- layer 1 is computed in full by every rank;
 * for k >= 2 it fills cur[row_lo-1 .. row_hi-2] from the COMPLETE previous layer `prev` -- followed by an all_gather of
 * the tile slices (torch.distributed / RCCL over xGMI) so that `cur` is complete everywhere.  cst buffers are
 * caller-owned device arrays of n+1 cost elements (int64 or double like the model).  unravel_splits
 * (DynamicSplitter.jl:89-99) asks the owner of each row: cp_dp_ptr_at returns ptr[j',k] (1-based) on the owning rank
 * and 0 elsewhere, so a MAX all_reduce of one integer per layer walks the chain. */
typedef struct cp_dp_s *cp_dp_t;
int32_t cp_dp_begin(cp_csr_t csr, int64_t K, int32_t combine, int32_t order, const cp_model_t *model,
                    int64_t row_lo, int64_t row_hi, cp_dp_t *out);
int32_t cp_dp_layer(cp_dp_t dp, int64_t k, const void *cst_prev_device, void *cst_cur_device);
int32_t cp_dp_ptr_at(cp_dp_t dp, int64_t k, int64_t jp, int64_t *out);
/* table-level windows for parity tests: the whole ptr[:, k] row of this rank's tile (0 outside the tile), and -- on the
 * O(n log^2 n) path -- the per-block state the last cp_dp_layer(k >= 2) left behind: for bit plane b < nplanes and row
 * j' = r + 1 with bit b of r set, opt_out[b*(n+1) + r] = the LARGEST j minimising cst_prev[j] + f(j, j', k) over the
 * row's Fenwick block j - 1 in [r_b - 2^b, r_b) (r_b = r with the bits below b cleared), nets_out / selfnets_out the counts
 * of that part; 0 where bit b is clear.  These are the candidates the layer's final combine merges into
 * cst[j', k] / ptr[j', k] (DynamicSplitter.jl:36-43); arrays hold 31 * (n+1) entries (selfnets_out may be NULL). */
int32_t cp_dp_ptr_row(cp_dp_t dp, int64_t k, int64_t *out /* n+1 */);
/* layers k >= 2 computed after this call take their candidates from the width window j' - w_max <= j <= j' (the candidate
 * range of the ConstrainedCost splitter with a VertexCount weight, DynamicSplitter.jl:233-246); 0 restores the full range.
 * The layer windows j'_lo / j'_hi are the caller's business (the row tile, and a huge cost outside the previous window). */
int32_t cp_dp_set_window(cp_dp_t dp, int64_t wmax);
/* moves this rank's row tile [row_lo, row_hi) for the layers computed from now on (the constrained DP tiles every layer's own
 * window j'_lo[k] .. j'_hi[k] over the ranks); cp_dp_ptr_at / cp_dp_ptr_row answer with the tile each layer was computed with */
int32_t cp_dp_set_rows(cp_dp_t dp, int64_t row_lo, int64_t row_hi);
int32_t cp_dp_block_tables(cp_dp_t dp, int32_t *nplanes_out, int64_t *opt_out, int64_t *nets_out, int64_t *selfnets_out);
int32_t cp_dp_destroy(cp_dp_t dp);

/* ---- execution control / measurement ---- */
/* run subsequent launches of this csr on an existing hipStream_t (e.g. torch's current stream) */
int32_t cp_set_stream(cp_csr_t csr, void *hip_stream);
/* back to a stream of the handle's own (waits for the borrowed one first): a handle must not stay bound to a caller's stream
 * that may be destroyed before the handle is (chainpartitioners.jl_amd/distributed.py restores it when the tiled run returns) */
int32_t cp_reset_stream(cp_csr_t csr);
/* diagnostics counters of the library (tests): "spec_redo" -- DP layers enqueued from the previous layer's counts that had to be
 * run again with exact counts; "poison_hits" -- with cp_set_option("poison", 1): plane cells read that the layer had not written
 * (each such layer must be one of the redone ones; the library checks it and fails with CP_EINTERNAL otherwise).  Both are reset
 * when "poison" is switched on. */
int32_t cp_get_stat(const char *name, int64_t *out);
/* Library-wide tunables and test switches; results never depend on them (tests/test_gpu_dynamic.py runs every one against the
 * oracle).  "force_brute" 1: the general O(K n^2) device DP even where the O(K n log^2 n) scheme applies; "brute_max_n": its size
 * limit.  Layer driver of the O(K n log^2 n) scheme: "short_t"/"short_e" (tasks finished during setup),
 * "own_min" (shortest task with tiles of its own), "gap_tau"/"gap_min" (gap passes: rounds and task lengths; -1: none), "gap_nr" (64-row chunks per wave of the gap finish: 1 or 2), "pool" (1, default: freed device blocks of 1 MB and more are kept for reuse by the next call; 0: returned to HIP at once, and the pool is emptied),
 * "ra_cache" (round A from counts cached per partition), "nospec" 1 (one host sync per round instead of sizing a layer from the
 * previous one), "rpass_ch"/"rpass_small_tau"/"rpass_cap" (right-part passes: columns per wave, last
 * lane-per-row round, lane-private share of a row in per cent of the mean), "setup_bs" (lanes per block of the task setup), "force_max" (a round whose flattened
 * stage served at most this many tasks gives them tiles of their own from the next layer on), "leaf" (1: the rounds tau < 6 of a
 * layer are one leaf pass, csrc/dp_leaf.inc; 0: divide-and-conquer rounds down to tau = 0), "block_tables" (1: the leaf pass also
 * stores every per-block winner, needed by cp_dp_block_tables), "poison" (1: test mode of cp_get_stat above), "fixed_point" (1: layers
 * after one that reproduced its input row are copied), "bn_wave" (bottleneck DP walk: 0 lane per chunk of "bn_chunk" rows, 1 wave
 * per run of "bn_run" rows in lockstep, 2 = default: searched crossings for Int64 costs), "bn_slack" (columns of the start bracket), "prof_only" slot (events on one profile slot only), "dbg"
 * (diagnostic bit mask).  Unknown names return CP_EINVAL. */
int32_t cp_set_option(const char *name, int64_t value);
/* built-in per-kernel HIP-event timing of the named hot kernels on the launch stream */
int32_t cp_prof_enable(int32_t on);
int32_t cp_prof_reset(void);
/* slot-wise totals: name, launches, total ms, algorithmic bytes; returns number of slots */
int32_t cp_prof_get(int32_t slot, const char **name, int64_t *launches, double *total_ms, double *alg_bytes);

#ifdef __cplusplus
}
#endif
#endif
