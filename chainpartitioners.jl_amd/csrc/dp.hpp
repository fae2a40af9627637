// dp.hpp -- internal interfaces between the DP translation units.
#pragma once
#include "csr.hpp"
#include "model.hpp"

namespace cpk {

// one layer of the total-cost DP by the O(n log^2 n) scheme (dp_total.hip)
template <typename TC>
void dp_total_layer(cp_csr_s *A, const DevModel<TC> &M, TC alpha, const TC *W, TC *cst_out, int32_t *ptr_out, void *work,
                    int64_t rlo, int64_t rhi,       // computes rows r in [rlo, rhi] (0-based); the full range is [0, n]
                    int64_t wwin = 0);              // > 0: width window -- the candidates of row r are max(0, r - wwin) <= p <= r
template <typename TC> int dp_total_block_tables(cp_csr_s *A, void *work, int64_t *opt_out, int64_t *nn_out, int64_t *nl_out);   // per-block winners of the last layer (tests)
template <typename TC> void *dp_total_work_get(cp_csr_s *A);      // the handle's scratch for cost type TC (created on first use)
template <typename TC> void *dp_total_work_new();
template <typename TC> void dp_total_work_free(void *w);

// one layer by the general O(n^2) sweep, any combine / any model sign pattern (dp_brute.hip).
// rows [r_lo, r_hi] are computed; window (lo/hi per row) optional.
template <typename TC>
void dp_brute_layer(cp_csr_s *A, const DevModel<TC> &M, TC alpha, int32_t combine, const TC *W, TC *cst_out,
                    int32_t *ptr_out, int64_t r_lo, int64_t r_hi);

// one layer of the bottleneck (g = max) DP for costs that grow with their part, by the valley search (dp_bottleneck.hip)
template <typename TC>
void dp_bottleneck_layer(cp_csr_s *A, const DevModel<TC> &M, TC alpha, const TC *W, TC *cst_out, int32_t *ptr_out,
                         int64_t r_lo, int64_t r_hi,
                         // candidate limits of a weight-constrained layer (0-based; defaults: none): row r takes max(p_lo0, j0(r)) <= p <= min(r, p_hi0),
                         // j0(r) = r - wwin, or j0[r] when the device array j0 (n + 1 entries) is given
                         int64_t wwin = 0, int64_t p_lo0 = 0, int64_t p_hi0 = -1, const int32_t *j0 = nullptr);
extern int64_t g_opt_bn_wave, g_opt_bn_run, g_opt_bn_slack;    // wave-per-run walk (default) and its rows per wave
extern int64_t g_opt_bn_chunk;                 // rows per two-pointer walk (one lane each)

// seq.hip
template <typename TC>
int32_t run_dyn_constrained(cp_csr_s *A, int64_t K, int32_t g, int32_t order, const cp_model_t *mdl, const cp_rowpart_t *Pi,
                            const cp_model_t *w, int64_t wi, double wf, int64_t *spl_out);
template <typename TC>
int32_t run_seq_eval(cp_csr_s *A, const cp_model_t *mdl, const cp_rowpart_t *Pi, int64_t nq, const int64_t *j, const int64_t *jp,
                     const int64_t *k, TC *out);

// plaid.hip: primary / secondary connectivity costs (need a row partition)
template <typename TC>
int32_t run_plaid_eval(cp_csr_s *A, const cp_model_t *mdl, const cp_rowpart_t *Pi, int64_t nq, const int64_t *j, const int64_t *jp,
                       const int64_t *k, TC *out);
template <typename TC>
int32_t run_plaid_dynamic(cp_csr_s *A, int64_t K, int32_t combine, int32_t order, const cp_model_t *mdl, const cp_rowpart_t *Pi,
                          int64_t *spl_out);

// chunk_scan.hip: DynamicTotalChunker under a VertexCount window as a (min,+) scan; false = not applicable
template <typename TC>
bool pack_dynamic_scan(hipStream_t s, int64_t n, int64_t wmax, const TC *Ftab, TC *cst1, int64_t *spl1);

extern int64_t g_opt_gap_nr;                      // 64-row chunks per wave of the gap finish (1 or 2)
extern int64_t g_opt_gap_tau, g_opt_gap_min;   // gap passes in the rounds tau <= gap_tau (-1: none) for tasks of >= gap_min candidates
extern int64_t g_opt_poison, g_poison_hits;    // poison mode (tests): see run_layer
extern int64_t g_opt_leaf;                     // 1: the rounds tau < 6 of an unconstrained layer are one leaf pass (dp_leaf.inc)
extern int64_t g_opt_block_tables;             // 1: the leaf pass also stores the per-block winners (cp_dp_block_tables)
extern int64_t g_opt_ra_cache;                 // 1: round A from counts computed once per partition
extern int64_t g_opt_fixed_point;              // 1: layers after a layer that reproduced its input row are copied (run_dynamic)
extern int64_t g_opt_nospec;                   // 1: every layer waits for its exact counts (one host sync per round)
extern int64_t g_spec_redo;                    // layers redone because the prediction missed (diagnostics)
extern int64_t g_opt_rpass_small_tau;           // rounds tau <= this use one lane per row in the right-part pass
extern int64_t g_opt_force_max;                 // see run_layer (force_own)
extern int64_t g_opt_setup_bs;                  // lanes per block of k_setup_short
extern int64_t g_opt_rpass_cap;                 // lane-private entries per row in k_rpass_small, per cent of the mean (200)
extern int64_t g_opt_rpass_ch;                 // columns per wave in k_rpass_wave (power of two >= 16)
extern int64_t g_opt_own_min;                  // tasks with at least this many steps get tiles of their own (>= 64)
extern int64_t g_opt_short_t, g_opt_short_e;   // k_setup_short: tasks with <= short_t candidates and <= short_e link entries finish in setup
extern int64_t g_opt_force_brute;      // cp_set_option("force_brute", 1)
extern int64_t g_opt_brute_max_n;
extern int64_t g_opt_dbg;            // timing experiments only (cp_set_option("dbg", mask)); results are wrong when non-zero      // largest n the O(n^2) path accepts

}  // namespace cpk
