// dp_brute.hip -- general device form of one DP layer: the literal O(n^2) candidate sweep of
// /root/reference/src/DynamicSplitter.jl:33-46 for ANY combine (+ / max) and ANY model sign pattern
// (decreasing "Funky" costs, bottleneck objective, hyperedge cut ...), used where the O(n log^2 n)
// scheme of dp_total.hip does not apply.  Also: per-range count kernels used by the objective /
// oracle-evaluation entry points.
//
// One lane owns one row r (= j'-1); the 64 rows of a wave walk the candidate p downwards TOGETHER, so the
// link entries of column p are wave-uniform (scalar) loads while every lane compares them with its own r:
//   nets(p, r)     = nets(p+1, r) + #{q in col p : next[q] >= r}          (SparsePrefixMatrices.jl:807-821 "Prev i")
//   selfnets(p, r) = selfnets(p+1, r) + #{rows with first == p and last < r}
// Candidates are visited from p = r down to 0 and replace the incumbent only when strictly better, which is
// the reference's "largest j wins ties" (it scans upwards with <=).
#include "csr.hpp"
#include "model.hpp"
#include "dp.hpp"

namespace cpk {

int64_t g_opt_force_brute = 0;
int64_t g_opt_short_t = 8, g_opt_short_e = 64;
int64_t g_opt_own_min = 64;
int64_t g_opt_rpass_ch = 256;
int64_t g_opt_rpass_small_tau = 4;
int64_t g_opt_force_max = 1024;           // a round's flattened stage is replaced by own tiles when it served at most this many tasks
int64_t g_opt_setup_bs = 1024;          // lanes per block of k_setup_short (a multiple of 64, at most 1024)
int64_t g_opt_rpass_cap = 200;          // lane-private entries per row in k_rpass_small, per cent of the mean
int64_t g_opt_nospec = 0, g_spec_redo = 0;
int64_t g_opt_gap_tau = 6, g_opt_gap_min = 64, g_opt_gap_nr = 2;
int64_t g_opt_ra_cache = 1;
int64_t g_opt_leaf = 1, g_opt_block_tables = 0;
int64_t g_opt_poison = 0, g_poison_hits = 0;
int64_t g_opt_fixed_point = 0;
int64_t g_opt_brute_max_n = 200000;
int64_t g_opt_dbg = 0;

template <typename TC>
__global__ void __launch_bounds__(256) k_brute_layer(int64_t n, int64_t r_lo, int64_t r_hi, const int64_t *__restrict__ pos,
                                                     const int32_t *__restrict__ next, const int64_t *__restrict__ fpos,
                                                     const int32_t *__restrict__ flast, DevModel<TC> M, TC alpha,
                                                     int32_t g, const TC *__restrict__ W, TC *__restrict__ cst,
                                                     int32_t *__restrict__ ptr)
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
    bool self = (M.kind == CP_MODEL_HYPEREDGE_CUT);
    int32_t rr = (int32_t)r;
    for (int64_t p = rtop; p >= 0; p--) {
        if (p < n) {
            // lanes with r > p take the left step over column p
            bool step = live && (r > p);
            int64_t q0 = pos[p], q1 = pos[p + 1];
            for (int64_t q = q0; q < q1; q++) {
                int32_t nx = next[q];                       // wave-uniform address
                if (step) nn += (nx >= rr);
            }
            if (self) {
                int64_t s0 = fpos[p], s1 = fpos[p + 1];
                for (int64_t sidx = s0; sidx < s1; sidx++) {
                    int32_t la = flast[sidx];
                    if (step) nl += (la < rr);
                }
            }
        }
        if (live && r >= p) {
            TC f = dm_apply(M, alpha, r - p, pos[r] - pos[p], nn, nl);
            TC v = comb(g, W[p], f);
            if (bp < 0 || v < best) { best = v; bp = p; }
        }
    }
    if (live) { cst[r] = best; ptr[r] = (int32_t)bp; }
}

template <typename TC>
void dp_brute_layer(cp_csr_s *A, const DevModel<TC> &M, TC alpha, int32_t combine, const TC *W, TC *cst_out,
                    int32_t *ptr_out, int64_t r_lo, int64_t r_hi)
{
    hipStream_t s = A->stream;
    int64_t rows = r_hi - r_lo + 1;
    int64_t waves = cdiv(rows, 64);
    int64_t blocks = cdiv(waves, 4);
    ProfScope ps(PROF_BRUTE, s, 0.0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_brute_layer<TC>), dim3((unsigned)blocks), dim3(256), 0, s, A->n, r_lo, r_hi, A->pos.p,
                       A->next.p, A->fpos.p, A->flast.p, M, alpha, combine, W, cst_out, ptr_out);
    CP_HIP(hipGetLastError());
}

template void dp_brute_layer<int64_t>(cp_csr_s *, const DevModel<int64_t> &, int64_t, int32_t, const int64_t *, int64_t *, int32_t *, int64_t, int64_t);
template void dp_brute_layer<double>(cp_csr_s *, const DevModel<double> &, double, int32_t, const double *, double *, int32_t *, int64_t, int64_t);

}  // namespace cpk
