// partwise.hip -- partwise(A, Pi) (/root/reference/src/PartwiseCounts.jl:1-60) on the device: regroup the
// nonzeros by the part k = asg[i] of their row.  The reference does two counting passes with per-part cursors;
// here: one stable radix sort of (k -> nonzero id) keeps the column-major order inside every part, a flag/scan
// pass finds the boundaries of the non-empty (k, j) pairs (= columns of A'), and a scatter writes pos', prm and
// the part offsets pios.
#include "csr.hpp"
#include <rocprim/rocprim.hpp>

namespace cpk {

__global__ void k_pw_keys(const int32_t *__restrict__ row, const int32_t *__restrict__ asg, uint32_t *__restrict__ key,
                          uint32_t *__restrict__ val, int64_t N)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < N) { key[q] = (uint32_t)asg[row[q]]; val[q] = (uint32_t)q; }
}

// flag[s] = 1 where a new (part, column) pair starts in the sorted order
__global__ void k_pw_flags(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sq, const int32_t *__restrict__ col,
                           int32_t *__restrict__ flag, int64_t N)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    flag[s] = (s == 0) || skey[s] != skey[s - 1] || col[sq[s]] != col[sq[s - 1]];
}

__global__ void k_pw_emit(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sq, const int32_t *__restrict__ col,
                          const int32_t *__restrict__ row, const int32_t *__restrict__ flag, const int64_t *__restrict__ rank,
                          int64_t N, int64_t *__restrict__ pos_out, int64_t *__restrict__ prm_out, int64_t *__restrict__ idx_out,
                          int64_t *__restrict__ pios)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    uint32_t q = sq[s];
    idx_out[s] = (int64_t)row[q] + 1;
    if (flag[s]) {
        int64_t jj = rank[s];                       // 0-based column of A'
        pos_out[jj] = s + 1;
        prm_out[jj] = (int64_t)col[q] + 1;
        if (s == 0 || skey[s] != skey[s - 1]) {
            // first column of part k: every part in (previous part, k] starts here
            uint32_t k = skey[s], kprev = (s == 0) ? 0u : skey[s - 1];
            for (uint32_t t = kprev + 1; t <= k; t++) pios[t - 1] = jj + 1;
        }
    }
}

}  // namespace cpk

using namespace cpk;

extern "C" int32_t cp_partwise(cp_csr_t A, int64_t K, const int64_t *asg, int64_t *nprime_out, int64_t *pios_out, int64_t *prm_out,
                               int64_t *pos_out, int64_t *idx_out)
{
    try {
        CP_REQUIRE(A && asg && nprime_out && pios_out && prm_out && pos_out && idx_out && K >= 1, CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        ensure_links(A);                            // col[]
        hipStream_t s = A->stream;
        int64_t N = A->N, m = A->m;
        std::vector<int32_t> h_asg((size_t)(m > 0 ? m : 1));
        for (int64_t i = 0; i < m; i++) { CP_REQUIRE(asg[i] >= 1 && asg[i] <= K, CP_EINVAL, "asg out of 1:K"); h_asg[(size_t)i] = (int32_t)asg[i]; }
        size_t Na = (size_t)(N > 0 ? N : 1);
        DBuf<int32_t> d_asg(h_asg.size()), flag(Na);
        DBuf<uint32_t> kin(Na), kout(Na), vin(Na), vout(Na);
        DBuf<int64_t> rank(Na + 1), scratch, d_pos(Na + 1), d_prm(Na), d_idx(Na), d_pios((size_t)K + 1);
        CP_HIP(hipMemcpyAsync(d_asg.p, h_asg.data(), sizeof(int32_t) * h_asg.size(), hipMemcpyHostToDevice, s));
        int64_t nprime = 0;
        std::vector<int64_t> h_pios((size_t)K + 1, 0);
        if (N > 0) {
            hipLaunchKernelGGL(k_pw_keys, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, A->row.p, d_asg.p, kin.p, vin.p, N);
            unsigned end_bit = 1;
            while (end_bit < 32 && ((uint64_t)1 << end_bit) <= (uint64_t)K) end_bit++;
            size_t tmp_bytes = 0;
            CP_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin.p, kout.p, vin.p, vout.p, (size_t)N, 0u, end_bit, s));
            DBuf<char> tmp(tmp_bytes > 0 ? tmp_bytes : 1);
            CP_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tmp_bytes, kin.p, kout.p, vin.p, vout.p, (size_t)N, 0u, end_bit, s));
            hipLaunchKernelGGL(k_pw_flags, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, kout.p, vout.p, A->col.p, flag.p, N);
            exclusive_scan_i32(flag.p, rank.p, N, scratch, s);
            CP_HIP(hipMemsetAsync(d_pios.p, 0, d_pios.bytes(), s));
            hipLaunchKernelGGL(k_pw_emit, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, kout.p, vout.p, A->col.p, A->row.p, flag.p, rank.p,
                               N, d_pos.p, d_prm.p, d_idx.p, d_pios.p);
            CP_HIP(hipGetLastError());
            CP_HIP(hipMemcpyAsync(&nprime, rank.p + N, sizeof(int64_t), hipMemcpyDeviceToHost, s));
            CP_HIP(hipStreamSynchronize(s));
            CP_HIP(hipMemcpyAsync(pos_out, d_pos.p, sizeof(int64_t) * (size_t)nprime, hipMemcpyDeviceToHost, s));
            CP_HIP(hipMemcpyAsync(prm_out, d_prm.p, sizeof(int64_t) * (size_t)nprime, hipMemcpyDeviceToHost, s));
            CP_HIP(hipMemcpyAsync(idx_out, d_idx.p, sizeof(int64_t) * (size_t)N, hipMemcpyDeviceToHost, s));
            CP_HIP(hipMemcpyAsync(h_pios.data(), d_pios.p, sizeof(int64_t) * (size_t)(K + 1), hipMemcpyDeviceToHost, s));
            CP_HIP(hipStreamSynchronize(s));
        }
        pos_out[nprime] = N + 1;
        // parts with no nonzero (and the tail) start where the next non-empty part starts: fill backwards
        h_pios[(size_t)K] = nprime + 1;
        for (int64_t k = K - 1; k >= 0; k--) if (h_pios[(size_t)k] == 0) h_pios[(size_t)k] = h_pios[(size_t)k + 1];
        for (int64_t k = 0; k <= K; k++) pios_out[k] = h_pios[(size_t)k];
        *nprime_out = nprime;
        return CP_OK;
    } CP_CATCH_ALL
}
