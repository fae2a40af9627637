// wavelet.hip -- device construction of the dominance counter (wavelet.hpp) and the counting-structure
// entry points of the C ABI: dominancecount / netcount / selfnetcount
// (/root/reference/src/SparsePrefixMatrices.jl:606-689, SparseColorArrays.jl:101-125, 177-229).
//
// Construction = H stable 0/1 partitions of the key sequence inside the buckets of equal high bits, the same
// passes the reference runs sequentially (:624-653), each done here as: ballot the bit into 64-bit words,
// scan the word popcounts (this IS the rank directory `cnt`), scatter every key to
//   bucket_start + (zeros before it in the bucket)                       if its bit is 0
//   bucket_start + (zeros in the bucket) + (ones before it in the bucket) if its bit is 1.
#include "csr.hpp"
#include "wavelet.hpp"
#include <memory>
#include <vector>

namespace cpk {

static int32_t cllog2_i(int64_t x)      // util.jl:3-8 : ceil(log2(x)), cllog2(1) = 0
{
    int32_t h = 0;
    while (((int64_t)1 << h) < x) h++;
    return h;
}

// hot: a key that a large share of the entries carries (net keys: n + 1 = "no previous occurrence", one entry in ten on the bench
// matrices) -- same-address atomics serialise in L2 (115 ms of a 270 ms build at N = 10^8), so its count is taken per wave
// (Tried: bucket starts from a rocprim radix sort of the keys instead of this histogram: the build went from 75 to 266 ms at N = 10^8.)
__global__ void __launch_bounds__(256) k_hist_keys(const int32_t *__restrict__ keys, int64_t Nk, int32_t *__restrict__ hist, int32_t hot)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t key = q < Nk ? keys[q] : -1;
    const bool is_hot = key >= 0 && key == hot;
    const unsigned long long m = __ballot(is_hot);
    if (m && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(&hist[hot], (int32_t)__popcll(m));
    if (key >= 0 && !is_hot) atomicAdd(&hist[key], 1);
}

// (eight 64-key words per wave, their loads issued together: with one key per thread the 390 000 blocks of a level at N = 10^8 spent
//  their time being launched -- 0.20 ms per level for a 400 MB read)
constexpr int WT_E = 8;
__global__ void __launch_bounds__(256) k_wt_bits(const int32_t *__restrict__ keys, int64_t Nk, int h, int64_t W,
                                                 uint64_t *__restrict__ byt, int32_t *__restrict__ popc)
{
    const int64_t base = ((int64_t)blockIdx.x * WT_E) * blockDim.x + threadIdx.x;
    int32_t k[WT_E];
#pragma unroll
    for (int e = 0; e < WT_E; e++) { const int64_t q = base + (int64_t)e * blockDim.x; k[e] = q < Nk ? keys[q] : 0; }
#pragma unroll
    for (int e = 0; e < WT_E; e++) {
        const int64_t q = base + (int64_t)e * blockDim.x;
        const int bit = (k[e] >> (h - 1)) & 1;                 // (keys beyond Nk were read as 0)
        const unsigned long long m = __ballot(bit);
        const int64_t w = q >> 6;
        if ((threadIdx.x & 63) == 0 && w < W) { byt[w] = m; popc[w] = __popcll(m); }
    }
}

__device__ __forceinline__ int64_t ones_before(const uint64_t *__restrict__ byt, const int32_t *__restrict__ cnt, int64_t x)
{
    return (int64_t)cnt[x >> 6] + __popcll(byt[x >> 6] & (((uint64_t)1 << (x & 63)) - 1));
}

__global__ void __launch_bounds__(256) k_wt_scatter(const int32_t *__restrict__ keys, int32_t *__restrict__ out, int64_t Nk, int h,
                                                    const uint64_t *__restrict__ byt, const int32_t *__restrict__ cnt,
                                                    const int32_t *__restrict__ qos0)
{
    const int64_t base = ((int64_t)blockIdx.x * WT_E) * blockDim.x + threadIdx.x;
    int32_t k[WT_E];
#pragma unroll
    for (int e = 0; e < WT_E; e++) { const int64_t q = base + (int64_t)e * blockDim.x; k[e] = q < Nk ? keys[q] : 0; }
#pragma unroll
    for (int e = 0; e < WT_E; e++) {
        const int64_t q = base + (int64_t)e * blockDim.x;
        if (q >= Nk) continue;
        const int32_t key = k[e];
        int64_t bucket = (int64_t)key >> h;
        int64_t s = qos0[bucket << h], en = qos0[(bucket + 1) << h];
        int64_t ob_q = ones_before(byt, cnt, q), ob_s = ones_before(byt, cnt, s), ob_e = ones_before(byt, cnt, en);
        int64_t zeros_in_bucket = (en - s) - (ob_e - ob_s);
        int d = (key >> (h - 1)) & 1;
        int64_t np = d ? s + zeros_in_bucket + (ob_q - ob_s) : s + ((q - ob_q) - (s - ob_s));
        out[np] = key;
    }
}

// The key histogram of the net counter without atomics: key n - c counts the entries whose PREVIOUS occurrence lies in column c,
// i.e. the entries OF column c that have a next occurrence (next[q] < n) -- one lane per column over its own entries; key n + 1
// (no previous occurrence) gets what is left of N.  (The atomic histogram k_hist_keys: 17 ms at N = 10^8, a quarter of the build.)
__global__ void __launch_bounds__(256) k_hist_net(const int64_t *__restrict__ pos, const int32_t *__restrict__ next, int64_t n, int32_t *__restrict__ hist,
                                                  unsigned long long *__restrict__ total)
{
    __shared__ int32_t sh[256];
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t cnt = 0;
    if (c < n) {
        for (int64_t q = pos[c], q1 = pos[c + 1]; q < q1; q++) cnt += next[q] < n;
        hist[n - c] = cnt;
    }
    sh[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < (unsigned)o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && sh[0]) atomicAdd(total, (unsigned long long)sh[0]);
}
__global__ void k_hist_net_first(int64_t N, int64_t n, const unsigned long long *__restrict__ total, int32_t *__restrict__ hist)
{
    hist[n + 1] = (int32_t)(N - (int64_t)*total);
}
// ... of the self-net counter: key n - c counts the rows whose first column is c (the buckets fpos)
__global__ void __launch_bounds__(256) k_hist_self(const int64_t *__restrict__ fpos, int64_t n, int32_t *__restrict__ hist)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) hist[n - c] = (int32_t)(fpos[c + 1] - fpos[c]);
}

void wavelet_build(WaveletHost &WT, DBuf<int32_t> &keys, int64_t Nk, int32_t H, hipStream_t s, int32_t hot_key, const WaveletHist *pre)
{
    ProfScope ps(PROF_WAVELET, s, 8.0 * (double)Nk * (double)H);
    int64_t W = 1 + cdiv(Nk, 64);
    int64_t nkeys = ((int64_t)1 << H) + 1;
    int32_t Hd = H > 0 ? H : 1;
    WT.byt.alloc((size_t)(W * Hd));
    WT.cnt.alloc((size_t)((W + 1) * Hd));
    WT.qos0.alloc((size_t)nkeys + 1);
    CP_HIP(hipMemsetAsync(WT.byt.p, 0, WT.byt.bytes(), s));
    CP_HIP(hipMemsetAsync(WT.cnt.p, 0, WT.cnt.bytes(), s));
    DBuf<int32_t> hist((size_t)nkeys), popc((size_t)W), tmp((size_t)(Nk > 0 ? Nk : 1));
    DBuf<int64_t> scratch;
    CP_HIP(hipMemsetAsync(hist.p, 0, hist.bytes(), s));
    DBuf<unsigned long long> total(1);
    if (pre && pre->kind == 1 && Nk > 0) {
        CP_HIP(hipMemsetAsync(total.p, 0, sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_hist_net, dim3((unsigned)cdiv(pre->n, 256)), dim3(256), 0, s, pre->pos, pre->link, pre->n, hist.p, total.p);
        hipLaunchKernelGGL(k_hist_net_first, dim3(1), dim3(1), 0, s, Nk, pre->n, total.p, hist.p);
    } else if (pre && pre->kind == 2 && Nk > 0) {
        hipLaunchKernelGGL(k_hist_self, dim3((unsigned)cdiv(pre->n, 256)), dim3(256), 0, s, pre->pos, pre->n, hist.p);
    } else if (Nk > 0) hipLaunchKernelGGL(k_hist_keys, dim3((unsigned)cdiv(Nk, 256)), dim3(256), 0, s, keys.p, Nk, hist.p, hot_key);
    exclusive_scan_i32_i32(hist.p, WT.qos0.p, nkeys, scratch, s);
    int32_t *cur = keys.p, *oth = tmp.p;
    for (int h = H; h >= 1 && Nk > 0; h--) {
        uint64_t *bv = WT.byt.p + (size_t)(h - 1) * W;
        int32_t *cv = WT.cnt.p + (size_t)(h - 1) * (W + 1);
        CP_HIP(hipMemsetAsync(popc.p, 0, popc.bytes(), s));
        hipLaunchKernelGGL(k_wt_bits, dim3((unsigned)cdiv(Nk, 256 * WT_E)), dim3(256), 0, s, cur, Nk, h, W, bv, popc.p);
        exclusive_scan_i32_i32(popc.p, cv, W, scratch, s);
        if (h > 1) {
            hipLaunchKernelGGL(k_wt_scatter, dim3((unsigned)cdiv(Nk, 256 * WT_E)), dim3(256), 0, s, cur, oth, Nk, h, bv, cv, WT.qos0.p);
            int32_t *t = cur; cur = oth; oth = t;
        }
    }
    CP_HIP(hipGetLastError());
    CP_HIP(hipStreamSynchronize(s));        // scratch buffers die here
    WT.d.H = H; WT.d.Nk = Nk; WT.d.W = W; WT.d.byt = WT.byt.p; WT.d.cnt = WT.cnt.p; WT.d.qos0 = WT.qos0.p;
}

// ------------------------------------------------------------------ weighted dominance: DominanceSum / RookSum / RookCount
// dominancesum(hint, A): S[i, j] = sum(A[1:i-1, 1:j-1]) (SparsePrefixMatrices.jl:1-250; definition test_SparsePrefixMatrices.jl:15);
// rookcount!(hint, N, idx) / rooksum!(hint, N, idx, val): the same over the permutation pattern with ONE point (idx[j], j) per column
// (:825-1273).  The reference carries `wgt` / `scn` prefix sums of nzval through its radix tree; here the wavelet counter carries,
// per level h, Z_h[x] = sum of the weights of the entries before position x of the level-h order whose bit h-1 is 0 (exactly the
// entries a rank step adds when the query's bit is 1), and P_0 = prefix sums in the fully sorted order (the entries whose key
// equals the query's).  Weights are summed as 64-bit words with wrap-around (Julia's Int / UInt arithmetic: exact) or as Float64.
template <typename TW> __device__ __forceinline__ TW w_add(TW a, TW b);
template <> __device__ __forceinline__ uint64_t w_add<uint64_t>(uint64_t a, uint64_t b) { return a + b; }
template <> __device__ __forceinline__ double w_add<double>(double a, double b) { return a + b; }

// exclusive prefix sums of 8-byte weights (masked): three kernels, blocks of 2048
template <typename TW>
__global__ void __launch_bounds__(256) k_wsum_reduce(const TW *__restrict__ w, const int32_t *__restrict__ keys, int bitpos, int64_t n, TW *__restrict__ bsum)
{
    __shared__ TW sh[256];
    int64_t base = (int64_t)blockIdx.x * 2048 + (int64_t)threadIdx.x * 8;
    TW s = (TW)0;
    for (int k = 0; k < 8; k++) if (base + k < n && (bitpos < 0 || !((keys[base + k] >> bitpos) & 1))) s = w_add<TW>(s, w[base + k]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < (unsigned)o) sh[threadIdx.x] = w_add<TW>(sh[threadIdx.x], sh[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}
template <typename TW>
__global__ void k_wsum_blocks(TW *__restrict__ bsum, int64_t nb)
{
    TW run = (TW)0;
    for (int64_t i = 0; i < nb; i++) { TW v = bsum[i]; bsum[i] = run; run = w_add<TW>(run, v); }      // (tests-sized structure: nb = N / 2048)
}
template <typename TW>
__global__ void __launch_bounds__(256) k_wsum_apply(const TW *__restrict__ w, const int32_t *__restrict__ keys, int bitpos, int64_t n, const TW *__restrict__ bsum,
                                                    TW *__restrict__ out)
{
    __shared__ TW sh[256];
    int64_t base = (int64_t)blockIdx.x * 2048 + (int64_t)threadIdx.x * 8;
    TW v[8], s = (TW)0;
    for (int k = 0; k < 8; k++) { v[k] = (base + k < n && (bitpos < 0 || !((keys[base + k] >> bitpos) & 1))) ? w[base + k] : (TW)0; s = w_add<TW>(s, v[k]); }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        TW t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : (TW)0;
        __syncthreads();
        sh[threadIdx.x] = w_add<TW>(sh[threadIdx.x], t);
        __syncthreads();
    }
    TW run = w_add<TW>(bsum[blockIdx.x], threadIdx.x > 0 ? sh[threadIdx.x - 1] : (TW)0);      // exclusive: the lanes before this one
    for (int k = 0; k < 8; k++) if (base + k <= n) { out[base + k] = run; run = w_add<TW>(run, v[k]); }
}
template <typename TW>
static void wsum_scan(const TW *w, const int32_t *keys, int bitpos, int64_t n, TW *out /* n + 1 */, DBuf<TW> &scratch, hipStream_t s)
{
    int64_t nb = cdiv(n + 1, 2048);
    scratch.ensure((size_t)nb + 1);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wsum_reduce<TW>), dim3((unsigned)nb), dim3(256), 0, s, w, keys, bitpos, n, scratch.p);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wsum_blocks<TW>), dim3(1), dim3(1), 0, s, scratch.p, nb);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wsum_apply<TW>), dim3((unsigned)nb), dim3(256), 0, s, w, keys, bitpos, n, scratch.p, out);
}

template <typename TW>
__global__ void __launch_bounds__(256) k_wt_scatter_w(const int32_t *__restrict__ keys, int32_t *__restrict__ out, const TW *__restrict__ w, TW *__restrict__ wout,
                                                      int64_t Nk, int h, const uint64_t *__restrict__ byt, const int32_t *__restrict__ cnt, const int32_t *__restrict__ qos0)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Nk) return;
    int32_t key = keys[q];
    int64_t bucket = (int64_t)key >> h;
    int64_t s = qos0[bucket << h], e = qos0[(bucket + 1) << h];
    int64_t ob_q = ones_before(byt, cnt, q), ob_s = ones_before(byt, cnt, s), ob_e = ones_before(byt, cnt, e);
    int64_t zeros_in_bucket = (e - s) - (ob_e - ob_s);
    int d = (key >> (h - 1)) & 1;
    int64_t np = d ? s + zeros_in_bucket + (ob_q - ob_s) : s + ((q - ob_q) - (s - ob_s));
    out[np] = key; wout[np] = w[q];
}

template <typename TW>
struct WsumDev { WaveletDev T; const TW *Z; const TW *P0; int64_t stride; };      // Z: [H][stride], level h at (h-1)*stride

// (count, sum) of the keys <= kmax among the first dq entries (column order): wt_count_le with the weights riding along
template <typename TW>
__device__ __forceinline__ void wt_sum_le(const WsumDev<TW> &S, int64_t kmax, int64_t dq, int64_t &cnt_out, TW &sum_out)
{
    const WaveletDev &T = S.T;
    cnt_out = 0; sum_out = (TW)0;
    if (kmax < 0 || dq <= 0) return;
    if (kmax > ((int64_t)1 << T.H) - 1) kmax = ((int64_t)1 << T.H) - 1;
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
        if (d) { s += zeros; const TW *Z = S.Z + (int64_t)(h - 1) * S.stride; sum_out = w_add<TW>(sum_out, (TW)(Z[q2] - Z[q1])); }
        dq = d ? ones : zeros;
    }
    const int64_t q0 = T.qos0[i];
    sum_out = w_add<TW>(sum_out, (TW)(S.P0[q0 + dq] - S.P0[q0]));
    cnt_out = s + dq;
}

template <typename TW>
__global__ void k_wsum_query(WsumDev<TW> S, const int64_t *__restrict__ colstart, int64_t nq, const int64_t *__restrict__ a, const int64_t *__restrict__ b,
                             int64_t *__restrict__ cnt, TW *__restrict__ sum)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq) return;
    int64_t c; TW sm;
    wt_sum_le<TW>(S, a[t] - 1, colstart ? colstart[b[t] - 1] : b[t] - 1, c, sm);          // rows < i among the entries of the columns < j
    if (cnt) cnt[t] = c;
    sum[t] = sm;
}

}  // namespace cpk

struct cp_wsum_s {
    int device = 0; hipStream_t stream = nullptr;
    int32_t dtype = 0;
    int64_t Nk = 0, ncols = 0, nrows = 0;
    cpk::WaveletHost wt;
    cpk::DBuf<uint64_t> Z, P0;              // 8-byte words (uint64 or double bit patterns)
    cpk::DBuf<int64_t> colstart;            // ncols + 1 (dominance over a CSR pattern); empty for rooks (column j starts at j - 1)
    int64_t stride = 0;
    ~cp_wsum_s() { if (stream) (void)hipStreamDestroy(stream); }
};

namespace cpk {

template <typename TW>
static void wsum_build(cp_wsum_s *Wd, DBuf<int32_t> &keys, const TW *w_dev, int64_t Nk, int32_t H)
{
    hipStream_t s = Wd->stream;
    WaveletHost &WT = Wd->wt;
    int64_t W = 1 + cdiv(Nk, 64);
    int64_t nkeys = ((int64_t)1 << H) + 1;
    int32_t Hd = H > 0 ? H : 1;
    WT.byt.alloc((size_t)(W * Hd)); WT.cnt.alloc((size_t)((W + 1) * Hd)); WT.qos0.alloc((size_t)nkeys + 1);
    CP_HIP(hipMemsetAsync(WT.byt.p, 0, WT.byt.bytes(), s));
    CP_HIP(hipMemsetAsync(WT.cnt.p, 0, WT.cnt.bytes(), s));
    Wd->stride = Nk + 1;
    Wd->Z.alloc((size_t)Hd * (size_t)(Nk + 1)); Wd->P0.alloc((size_t)Nk + 1);
    CP_HIP(hipMemsetAsync(Wd->Z.p, 0, Wd->Z.bytes(), s));
    CP_HIP(hipMemsetAsync(Wd->P0.p, 0, Wd->P0.bytes(), s));
    DBuf<int32_t> hist((size_t)nkeys), popc((size_t)W), tmp((size_t)(Nk > 0 ? Nk : 1));
    DBuf<TW> wa((size_t)(Nk > 0 ? Nk : 1)), wb((size_t)(Nk > 0 ? Nk : 1)), wscr;
    DBuf<int64_t> scratch;
    CP_HIP(hipMemsetAsync(hist.p, 0, hist.bytes(), s));
    if (Nk > 0) {
        CP_HIP(hipMemcpyAsync(wa.p, w_dev, sizeof(TW) * (size_t)Nk, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_hist_keys, dim3((unsigned)cdiv(Nk, 256)), dim3(256), 0, s, keys.p, Nk, hist.p, (int32_t)-1);
    }
    exclusive_scan_i32_i32(hist.p, WT.qos0.p, nkeys, scratch, s);
    int32_t *cur = keys.p, *oth = tmp.p;
    TW *wc = wa.p, *wo = wb.p;
    for (int h = H; h >= 1 && Nk > 0; h--) {
        uint64_t *bv = WT.byt.p + (size_t)(h - 1) * W;
        int32_t *cv = WT.cnt.p + (size_t)(h - 1) * (W + 1);
        CP_HIP(hipMemsetAsync(popc.p, 0, popc.bytes(), s));
        hipLaunchKernelGGL(k_wt_bits, dim3((unsigned)cdiv(Nk, 256 * WT_E)), dim3(256), 0, s, cur, Nk, h, W, bv, popc.p);
        exclusive_scan_i32_i32(popc.p, cv, W, scratch, s);
        wsum_scan<TW>(wc, cur, h - 1, Nk, reinterpret_cast<TW *>(Wd->Z.p) + (size_t)(h - 1) * (size_t)(Nk + 1), wscr, s);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wt_scatter_w<TW>), dim3((unsigned)cdiv(Nk, 256)), dim3(256), 0, s, cur, oth, wc, wo, Nk, h, bv, cv, WT.qos0.p);
        { int32_t *t = cur; cur = oth; oth = t; TW *u = wc; wc = wo; wo = u; }
    }
    if (Nk > 0) wsum_scan<TW>(wc, cur, -1, Nk, reinterpret_cast<TW *>(Wd->P0.p), wscr, s);      // fully sorted order (H == 0: the original order)
    CP_HIP(hipGetLastError());
    CP_HIP(hipStreamSynchronize(s));
    WT.d.H = H; WT.d.Nk = Nk; WT.d.W = W; WT.d.byt = WT.byt.p; WT.d.cnt = WT.cnt.p; WT.d.qos0 = WT.qos0.p;
}

// ---- key sequences of the three counters
__global__ void k_keys_dom(const int32_t *__restrict__ row, int32_t *__restrict__ keys, int64_t N)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < N) keys[q] = row[q] + 1;                       // the reference keys on the 1-based row index
}
__global__ void k_keys_net(const int32_t *__restrict__ prev, int32_t *__restrict__ keys, int64_t N, int32_t n)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < N) keys[q] = n - prev[q];                      // idx'[q] = (n+1) - hst[i], hst = prev+1  (SparseColorArrays.jl:110)
}
__global__ void k_keys_self(const int32_t *__restrict__ lfirst, int32_t *__restrict__ keys, int64_t Np, int32_t n)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < Np) keys[q] = n - lfirst[q];                   // idx'[q] = (n+1) - first  (SparseColorArrays.jl:215)
}

void build_dom_counter(cp_csr_s *A, WaveletHost &out)
{
    hipStream_t s = A->stream;
    int64_t N = A->N;
    DBuf<int32_t> keys((size_t)(N > 0 ? N : 1));
    if (N > 0) hipLaunchKernelGGL(k_keys_dom, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, A->row.p, keys.p, N);
    wavelet_build(out, keys, N, cllog2_i(A->m + 1), s);
}
void ensure_net_counter(cp_csr_s *A, WaveletHost &out)
{
    ensure_links(A);
    hipStream_t s = A->stream;
    int64_t N = A->N;
    DBuf<int32_t> keys((size_t)(N > 0 ? N : 1));
    if (N > 0) hipLaunchKernelGGL(k_keys_net, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, A->prev.p, keys.p, N, (int32_t)A->n);
    const WaveletHist pre{1, A->n, A->pos.p, A->next.p};                           // (the key histogram straight from the columns: no atomics)
    wavelet_build(out, keys, N, cllog2_i(A->n + 2), s, (int32_t)(A->n + 1), &pre);      // key n + 1: the first occurrence of a row
}
void ensure_selfnet_counter(cp_csr_s *A, WaveletHost &out)
{
    ensure_self(A);
    hipStream_t s = A->stream;
    int64_t Np = A->nrows_nonempty;
    DBuf<int32_t> keys((size_t)(Np > 0 ? Np : 1));
    if (Np > 0) hipLaunchKernelGGL(k_keys_self, dim3((unsigned)cdiv(Np, 256)), dim3(256), 0, s, A->lfirst.p, keys.p, Np, (int32_t)A->n);
    const WaveletHist pre{2, A->n, A->fpos.p, nullptr};
    wavelet_build(out, keys, Np, cllog2_i(A->n + 2), s, -1, &pre);
}

// queries are 1-based like the reference: DOM C[i,j]; NET / SELFNET [j, j']
__global__ void k_count_query(int32_t kind, WaveletDev T, int64_t n, const int64_t *__restrict__ colstart, const int64_t *__restrict__ pos,
                              int64_t nq, const int64_t *__restrict__ a, const int64_t *__restrict__ b, int64_t *__restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq) return;
    if (kind == CP_COUNT_DOM) {
        out[t] = wt_count_le(T, a[t] - 1, colstart[b[t] - 1]);
    } else {
        int64_t p = a[t] - 1, r = b[t] - 1;
        int64_t c = wt_count_le(T, n - p, colstart[r]);
        out[t] = (kind == CP_COUNT_NET) ? (pos[r] - pos[p]) - c : c;
    }
}

}  // namespace cpk

using namespace cpk;

extern "C" {

int32_t cp_count_build(cp_csr_t A, int32_t kind, int32_t hint, cp_count_t *out)
{
    (void)hint;         // every hint is served by the same exact structure
    try {
        CP_REQUIRE(A && out && kind >= CP_COUNT_DOM && kind <= CP_COUNT_SELFNET, CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        cp_count_s *h = new cp_count_s();
        h->A = A; h->kind = kind;
        try {
            if (kind == CP_COUNT_DOM) build_dom_counter(A, h->wt);
            else if (kind == CP_COUNT_NET) ensure_net_counter(A, h->wt);
            else ensure_selfnet_counter(A, h->wt);
        } catch (...) { delete h; throw; }
        prof_collect();
        *out = h;
        return CP_OK;
    } CP_CATCH_ALL
}

int32_t cp_count_query(cp_count_t h, int64_t nq, const int64_t *a, const int64_t *b, int64_t *out)
{
    try {
        CP_REQUIRE(h && (nq == 0 || (a && b && out)), CP_EINVAL, "bad argument");
        if (nq == 0) return CP_OK;
        cp_csr_s *A = h->A;
        CP_HIP(hipSetDevice(A->device));
        hipStream_t s = A->stream;
        for (int64_t t = 0; t < nq; t++) {
            if (h->kind == CP_COUNT_DOM) CP_REQUIRE(a[t] >= 1 && a[t] <= A->m + 1 && b[t] >= 1 && b[t] <= A->n + 1, CP_EINVAL, "C[i,j] needs 1<=i<=m+1, 1<=j<=n+1");
            else CP_REQUIRE(a[t] >= 1 && a[t] <= A->n + 1 && b[t] >= 1 && b[t] <= A->n + 1, CP_EINVAL, "count[j,j'] needs 1<=j,j'<=n+1");
        }
        DBuf<int64_t> da((size_t)nq), db((size_t)nq), dout((size_t)nq);
        CP_HIP(hipMemcpyAsync(da.p, a, sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
        CP_HIP(hipMemcpyAsync(db.p, b, sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
        const int64_t *colstart = (h->kind == CP_COUNT_SELFNET) ? A->lpos.p : A->pos.p;
        {
            ProfScope ps(PROF_QUERY, s, 0.0);
            hipLaunchKernelGGL(k_count_query, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, s, h->kind, h->wt.d, A->n, colstart, A->pos.p,
                               nq, da.p, db.p, dout.p);
        }
        CP_HIP(hipGetLastError());
        CP_HIP(hipMemcpyAsync(out, dout.p, sizeof(int64_t) * (size_t)nq, hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
        prof_collect();
        return CP_OK;
    } CP_CATCH_ALL
}

int32_t cp_count_destroy(cp_count_t h)
{
    if (h) { (void)hipSetDevice(h->A->device); delete h; }
    return CP_OK;
}

// ---- weighted dominance sums / rooks (SparsePrefixMatrices.jl:1-250, 825-1273)
static int32_t wsum_make(int device, int32_t dtype, int64_t nrows, int64_t ncols, int64_t Nk, const int64_t *colstart_dev, const int32_t *keys_src_dev,
                         const int64_t *idx_host, const void *val_host, cp_wsum_t *out)
{
    CP_REQUIRE(dtype == CP_I64 || dtype == CP_F64, CP_EINVAL, "weights are 8-byte integers (wrap-around sums) or Float64");
    CP_HIP(hipSetDevice(device));
    std::unique_ptr<cp_wsum_s> Wd(new cp_wsum_s());
    Wd->device = device; Wd->dtype = dtype; Wd->Nk = Nk; Wd->ncols = ncols; Wd->nrows = nrows;
    CP_HIP(hipStreamCreate(&Wd->stream));
    hipStream_t s = Wd->stream;
    DBuf<int32_t> keys((size_t)(Nk > 0 ? Nk : 1));
    if (keys_src_dev) { if (Nk > 0) hipLaunchKernelGGL(k_keys_dom, dim3((unsigned)cdiv(Nk, 256)), dim3(256), 0, s, keys_src_dev, keys.p, Nk); }
    else {
        std::vector<int32_t> hk((size_t)(Nk > 0 ? Nk : 1));
        std::vector<char> seen((size_t)Nk + 1, 0);
        for (int64_t q = 0; q < Nk; q++) {
            CP_REQUIRE(idx_host[q] >= 1 && idx_host[q] <= Nk && !seen[(size_t)idx_host[q]], CP_EINVAL, "rook: idx must be a permutation of 1..N");
            seen[(size_t)idx_host[q]] = 1; hk[(size_t)q] = (int32_t)idx_host[q];
        }
        CP_HIP(hipMemcpyAsync(keys.p, hk.data(), sizeof(int32_t) * (size_t)Nk, hipMemcpyHostToDevice, s));
        CP_HIP(hipStreamSynchronize(s));
    }
    if (colstart_dev) { Wd->colstart.alloc((size_t)ncols + 1); CP_HIP(hipMemcpyAsync(Wd->colstart.p, colstart_dev, sizeof(int64_t) * (size_t)(ncols + 1), hipMemcpyDeviceToDevice, s)); }
    DBuf<uint64_t> wv((size_t)(Nk > 0 ? Nk : 1));
    if (val_host) { if (Nk > 0) CP_HIP(hipMemcpyAsync(wv.p, val_host, 8 * (size_t)Nk, hipMemcpyHostToDevice, s)); }
    else CP_HIP(hipMemsetAsync(wv.p, 0, wv.bytes(), s));
    int32_t H = cllog2_i(nrows + 1);
    if (dtype == CP_I64) wsum_build<uint64_t>(Wd.get(), keys, wv.p, Nk, H);
    else wsum_build<double>(Wd.get(), keys, reinterpret_cast<const double *>(wv.p), Nk, H);
    *out = Wd.release();
    return CP_OK;
}

int32_t cp_domsum_build(cp_csr_t A, int32_t dtype, const void *val, cp_wsum_t *out)
{
    try {
        CP_REQUIRE(A && out && (val || A->N == 0), CP_EINVAL, "bad argument");
        CP_HIP(hipSetDevice(A->device));
        CP_HIP(hipStreamSynchronize(A->stream));
        return wsum_make(A->device, dtype, A->m, A->n, A->N, A->pos.p, A->row.p, nullptr, val, out);
    } CP_CATCH_ALL
}

int32_t cp_rook_build(int64_t N, const int64_t *idx, int32_t dtype, const void *val, int32_t device, cp_wsum_t *out)
{
    try {
        CP_REQUIRE(out && N >= 0 && (idx || N == 0) && N < ((int64_t)1 << 30), CP_EINVAL, "bad argument");
        CP_REQUIRE(cp_device_count() > 0, CP_EHIP, "no HIP device visible: libchainpart has no CPU fallback");
        return wsum_make(device, dtype, N, N, N, nullptr, nullptr, idx, val, out);
    } CP_CATCH_ALL
}

int32_t cp_wsum_query(cp_wsum_t h, int64_t nq, const int64_t *i, const int64_t *j, int64_t *count_out, int64_t *sum_i64, double *sum_f64)
{
    try {
        CP_REQUIRE(h && (nq == 0 || (i && j)) && (h->dtype == CP_I64 ? sum_i64 != nullptr : sum_f64 != nullptr), CP_EINVAL, "bad argument");
        if (nq == 0) return CP_OK;
        CP_HIP(hipSetDevice(h->device));
        hipStream_t s = h->stream;
        const int64_t nr = h->nrows;
        for (int64_t t = 0; t < nq; t++) CP_REQUIRE(i[t] >= 1 && i[t] <= nr + 1 && j[t] >= 1 && j[t] <= h->ncols + 1, CP_EINVAL, "S[i,j] needs 1<=i<=m+1, 1<=j<=n+1");
        DBuf<int64_t> da((size_t)nq), db((size_t)nq), dc((size_t)nq);
        DBuf<uint64_t> ds((size_t)nq);
        CP_HIP(hipMemcpyAsync(da.p, i, sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
        CP_HIP(hipMemcpyAsync(db.p, j, sizeof(int64_t) * (size_t)nq, hipMemcpyHostToDevice, s));
        const int64_t *cs = h->colstart.n ? h->colstart.p : nullptr;
        if (h->dtype == CP_I64) {
            WsumDev<uint64_t> S{h->wt.d, h->Z.p, h->P0.p, h->stride};
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wsum_query<uint64_t>), dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, s, S, cs, nq, da.p, db.p, dc.p, ds.p);
        } else {
            WsumDev<double> S{h->wt.d, reinterpret_cast<const double *>(h->Z.p), reinterpret_cast<const double *>(h->P0.p), h->stride};
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wsum_query<double>), dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, s, S, cs, nq, da.p, db.p, dc.p, reinterpret_cast<double *>(ds.p));
        }
        CP_HIP(hipGetLastError());
        if (count_out) CP_HIP(hipMemcpyAsync(count_out, dc.p, sizeof(int64_t) * (size_t)nq, hipMemcpyDeviceToHost, s));
        CP_HIP(hipMemcpyAsync(h->dtype == CP_I64 ? (void *)sum_i64 : (void *)sum_f64, ds.p, 8 * (size_t)nq, hipMemcpyDeviceToHost, s));
        CP_HIP(hipStreamSynchronize(s));
        return CP_OK;
    } CP_CATCH_ALL
}

int32_t cp_wsum_destroy(cp_wsum_t h)
{
    if (h) { (void)hipSetDevice(h->device); if (h->stream) (void)hipStreamSynchronize(h->stream); delete h; }
    return CP_OK;
}

}  // extern "C"
