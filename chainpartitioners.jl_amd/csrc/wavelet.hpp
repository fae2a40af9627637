// wavelet.hpp -- exact 2-D dominance counter on the device: the reference's BinaryDominanceCount
// (/root/reference/src/SparsePrefixMatrices.jl:606-689) laid out for HBM.  One structure serves every
// hint (NoHint/RandomHint/SparseHint/StepHint): query results are integers independent of the structure.
//
// A key sequence (one key per point, points ordered by column) is stably bit-partitioned level by level:
// level h (H..1) holds the sequence sorted by (key >> h); bit h-1 of every key is stored in the bit-vector
// byt[h] with a cumulative popcount cnt[h] per 64-bit word, exactly the reference's `byt` / `cnt` arrays
// (0-based here).  qos0[k] = #{keys < k} is the reference's qos[k+1] - 1.
#pragma once
#include "common.hpp"

namespace cpk {

struct WaveletDev {
    int32_t H;              // levels
    int64_t Nk;             // number of keys
    int64_t W;              // words per level = 1 + ceil(Nk / 64)
    const uint64_t *byt;    // [H][W]   (level h at (h-1)*W)
    const int32_t *cnt;     // [H][W+1] cumulative ones before each word
    const int32_t *qos0;    // [2^H + 1]
};

// number of keys <= kmax among the first dq entries (column order)   -- SparsePrefixMatrices.jl:660-689
__device__ __forceinline__ int64_t wt_count_le(const WaveletDev &T, int64_t kmax, int64_t dq)
{
    if (kmax < 0 || dq <= 0) return 0;
    if (kmax >= ((int64_t)1 << T.H) - 1) return dq;
    int64_t i = kmax, s = 0;
    for (int h = T.H; h >= 1; h--) {
        int64_t ip = i & ~(((int64_t)1 << h) - 1);
        int64_t q1 = T.qos0[ip];
        int64_t q2 = q1 + dq;
        int64_t d = (i >> (h - 1)) & 1;
        const uint64_t *bv = T.byt + (int64_t)(h - 1) * T.W;
        const int32_t *cv = T.cnt + (int64_t)(h - 1) * (T.W + 1);
        int64_t Q1 = q1 >> 6, Q2 = q2 >> 6;
        int64_t ones = (int64_t)cv[Q2] - cv[Q1];
        ones += __popcll(bv[Q2] & (((uint64_t)1 << (q2 & 63)) - 1));
        ones -= __popcll(bv[Q1] & (((uint64_t)1 << (q1 & 63)) - 1));
        int64_t zeros = dq - ones;
        s += d ? zeros : 0;
        dq = d ? ones : zeros;
    }
    return s + dq;
}

struct WaveletHost {
    WaveletDev d{};
    DBuf<uint64_t> byt;
    DBuf<int32_t> cnt, qos0;
};

// keys: device int32 array of Nk values in [0, 2^H); destroyed (used as scratch)
// the key histogram from the pattern instead of from the keys (no atomics): kind 1 = net counter (pos = colptr, link = next),
// kind 2 = self-net counter (pos = rows bucketed by first column)
struct WaveletHist { int kind; int64_t n; const int64_t *pos; const int32_t *link; };
void wavelet_build(WaveletHost &WT, DBuf<int32_t> &keys, int64_t Nk, int32_t H, hipStream_t s, int32_t hot_key = -1, const WaveletHist *pre = nullptr);      // hot_key: a key many entries share (counted per wave)

}  // namespace cpk

// counting-structure handle of the C ABI
struct cp_count_s {
    cp_csr_s *A;
    int32_t kind;
    cpk::WaveletHost wt;
};

namespace cpk {
// lazily built per-matrix counters used by the partitioners (net / selfnet), cached on the csr
struct CsrCounters {
    bool have_net = false, have_self = false;
    WaveletHost net, self;
};
void ensure_net_counter(cp_csr_s *A, WaveletHost &out);
void ensure_selfnet_counter(cp_csr_s *A, WaveletHost &out);
void build_dom_counter(cp_csr_s *A, WaveletHost &out);
}  // namespace cpk
