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

namespace cpk {

static int32_t cllog2_i(int64_t x)      // util.jl:3-8 : ceil(log2(x)), cllog2(1) = 0
{
    int32_t h = 0;
    while (((int64_t)1 << h) < x) h++;
    return h;
}

__global__ void k_hist_keys(const int32_t *__restrict__ keys, int64_t Nk, int32_t *__restrict__ hist)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < Nk) atomicAdd(&hist[keys[q]], 1);
}

__global__ void __launch_bounds__(256) k_wt_bits(const int32_t *__restrict__ keys, int64_t Nk, int h, int64_t W,
                                                 uint64_t *__restrict__ byt, int32_t *__restrict__ popc)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int bit = (q < Nk) ? ((keys[q] >> (h - 1)) & 1) : 0;
    unsigned long long m = __ballot(bit);
    int64_t w = q >> 6;
    if ((threadIdx.x & 63) == 0 && w < W) { byt[w] = m; popc[w] = __popcll(m); }
}

__device__ __forceinline__ int64_t ones_before(const uint64_t *__restrict__ byt, const int32_t *__restrict__ cnt, int64_t x)
{
    return (int64_t)cnt[x >> 6] + __popcll(byt[x >> 6] & (((uint64_t)1 << (x & 63)) - 1));
}

__global__ void __launch_bounds__(256) k_wt_scatter(const int32_t *__restrict__ keys, int32_t *__restrict__ out, int64_t Nk, int h,
                                                    const uint64_t *__restrict__ byt, const int32_t *__restrict__ cnt,
                                                    const int32_t *__restrict__ qos0)
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
    out[np] = key;
}

void wavelet_build(WaveletHost &WT, DBuf<int32_t> &keys, int64_t Nk, int32_t H, hipStream_t s)
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
    if (Nk > 0) hipLaunchKernelGGL(k_hist_keys, dim3((unsigned)cdiv(Nk, 256)), dim3(256), 0, s, keys.p, Nk, hist.p);
    exclusive_scan_i32_i32(hist.p, WT.qos0.p, nkeys, scratch, s);
    int32_t *cur = keys.p, *oth = tmp.p;
    for (int h = H; h >= 1 && Nk > 0; h--) {
        uint64_t *bv = WT.byt.p + (size_t)(h - 1) * W;
        int32_t *cv = WT.cnt.p + (size_t)(h - 1) * (W + 1);
        CP_HIP(hipMemsetAsync(popc.p, 0, popc.bytes(), s));
        hipLaunchKernelGGL(k_wt_bits, dim3((unsigned)cdiv(Nk, 256)), dim3(256), 0, s, cur, Nk, h, W, bv, popc.p);
        exclusive_scan_i32_i32(popc.p, cv, W, scratch, s);
        if (h > 1) {
            hipLaunchKernelGGL(k_wt_scatter, dim3((unsigned)cdiv(Nk, 256)), dim3(256), 0, s, cur, oth, Nk, h, bv, cv, WT.qos0.p);
            int32_t *t = cur; cur = oth; oth = t;
        }
    }
    CP_HIP(hipGetLastError());
    CP_HIP(hipStreamSynchronize(s));        // scratch buffers die here
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
    wavelet_build(out, keys, N, cllog2_i(A->n + 2), s);
}
void ensure_selfnet_counter(cp_csr_s *A, WaveletHost &out)
{
    ensure_self(A);
    hipStream_t s = A->stream;
    int64_t Np = A->nrows_nonempty;
    DBuf<int32_t> keys((size_t)(Np > 0 ? Np : 1));
    if (Np > 0) hipLaunchKernelGGL(k_keys_self, dim3((unsigned)cdiv(Np, 256)), dim3(256), 0, s, A->lfirst.p, keys.p, Np, (int32_t)A->n);
    wavelet_build(out, keys, Np, cllog2_i(A->n + 2), s);
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
    } catch (const HipFail &e) { return e.code; }
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
    } catch (const HipFail &e) { return e.code; }
}

int32_t cp_count_destroy(cp_count_t h)
{
    if (h) { (void)hipSetDevice(h->A->device); delete h; }
    return CP_OK;
}

}  // extern "C"
