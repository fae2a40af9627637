// core.hip -- globals, profiling, device-wide scan, CSR residency and link-array construction.
//
// Link arrays replace the reference's sequential `hst[i]` sweep (NetCount constructor,
// /root/reference/src/SparseColorArrays.jl:101-118; SelfNetCount :177-222): prev[q] is the column
// the reference stores as hst[i] when it visits nonzero q (0-based, -1 for "none"), so the
// reference's idx'[q] equals (n+1) - (prev[q] + 1).
#include "csr.hpp"
#include <rocprim/rocprim.hpp>
#include <map>
#include <mutex>

namespace cpk {

thread_local std::string g_last_error;
ProfSlot g_prof[PROF_NSLOTS] = {
    {"dp_lpass", 0, 0, 0}, {"dp_open_segments", 0, 0, 0}, {"dp_task_setup", 0, 0, 0},
    {"scan", 0, 0, 0}, {"dp_tile_carry", 0, 0, 0}, {"dp_span_fix", 0, 0, 0}, {"dp_combine", 0, 0, 0},
    {"link_build", 0, 0, 0}, {"dp_brute", 0, 0, 0}, {"wavelet_build", 0, 0, 0}, {"count_query", 0, 0, 0},
    {"bisect_probe", 0, 0, 0}, {"chunker", 0, 0, 0}, {"dp_rpass", 0, 0, 0}, {"dp_lpass_own", 0, 0, 0}, {"dp_gap_finish", 0, 0, 0}, {"dp_lpass_gap", 0, 0, 0}, {"dp_round_a", 0, 0, 0}, {"dp_leaf", 0, 0, 0}};
bool g_prof_on = false;
int g_prof_only = -1;
std::vector<ProfPending> g_prof_pending;
std::vector<hipEvent_t> g_event_pool;

void prof_collect()
{
    for (auto &p : g_prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_prof[p.slot].launches += 1;
            g_prof[p.slot].ms += ms;
            g_prof[p.slot].alg_bytes += p.bytes;
        }
        g_event_pool.push_back(p.a);
        g_event_pool.push_back(p.b);
    }
    g_prof_pending.clear();
}

// ------------------------------------------------------------------ pooled device allocations (common.hpp: DBuf)
namespace {
// (never destroyed: buffers owned by objects with static lifetime may be released after this file's statics are gone)
std::mutex &g_pool_mu = *new std::mutex;
std::map<std::pair<int, size_t>, std::vector<void *>> &g_pool = *new std::map<std::pair<int, size_t>, std::vector<void *>>;      // (device, bytes) -> free blocks
size_t g_pool_bytes = 0;
constexpr size_t POOL_MIN = (size_t)1 << 20, POOL_CAP = (size_t)48 << 30;
}
int64_t g_opt_pool = 1;

static void pool_trim_locked()
{
    for (auto &kv : g_pool) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != kv.first.first) (void)hipSetDevice(kv.first.first);
        for (void *p : kv.second) (void)hipFree(p);
        if (cur != kv.first.first) (void)hipSetDevice(cur);
    }
    g_pool.clear();
    g_pool_bytes = 0;
}
void dev_pool_trim() { std::lock_guard<std::mutex> lk(g_pool_mu); pool_trim_locked(); }

void *dev_alloc(size_t bytes)
{
    int dev = 0;
    if (bytes >= POOL_MIN && g_opt_pool) {
        CP_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool.find({dev, bytes});
        if (it != g_pool.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            g_pool_bytes -= bytes;
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {                               // out of memory with blocks parked in the pool: give them back and try once more
        (void)hipGetLastError();
        dev_pool_trim();
        CP_HIP(hipMalloc(&p, bytes));
    }
    return p;
}

void dev_free(void *p, size_t bytes)
{
    if (!p) return;
    if (bytes >= POOL_MIN && g_opt_pool) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            if (g_pool_bytes + bytes > POOL_CAP) pool_trim_locked();
            g_pool[{dev, bytes}].push_back(p);
            g_pool_bytes += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

// ------------------------------------------------------------------ exclusive scan int32 -> int64
constexpr int SCAN_T = 256;
constexpr int SCAN_I = 8;
constexpr int SCAN_TILE = SCAN_T * SCAN_I;

// n_dev != nullptr: the element count lives on the device (written by an earlier kernel of the stream); the grid is sized
// for an upper bound and blocks beyond the count contribute zeros
__global__ void __launch_bounds__(SCAN_T) k_scan_reduce(const int32_t *__restrict__ in, int64_t *__restrict__ bsum, int64_t n,
                                                        const int32_t *__restrict__ n_dev)
{
    __shared__ int64_t sh[SCAN_T / 64];
    if (n_dev) n = *n_dev;
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_I;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_I; k++) if (base + k < n) s += in[base + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < SCAN_T / 64; w++) t += sh[w];
        bsum[blockIdx.x] = t;
    }
}

// single block: exclusive scan of bsum[0..nb) in place, total written to bsum[nb]
__global__ void __launch_bounds__(1024) k_scan_blocksums(int64_t *__restrict__ bsum, int64_t nb)
{
    __shared__ int64_t sh[1024];
    int64_t chunk = (nb + 1023) / 1024;
    int64_t lo = (int64_t)threadIdx.x * chunk, hi = lo + chunk < nb ? lo + chunk : nb;
    int64_t s = 0;
    for (int64_t i = lo; i < hi; i++) s += bsum[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int64_t v = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    int64_t run = sh[threadIdx.x] - s;
    for (int64_t i = lo; i < hi; i++) { int64_t v = bsum[i]; bsum[i] = run; run += v; }
    if (threadIdx.x == 1023) bsum[nb] = sh[1023];
}

template <typename OutT>
__global__ void __launch_bounds__(SCAN_T) k_scan_apply(const int32_t *__restrict__ in, OutT *__restrict__ out,
                                                       const int64_t *__restrict__ bsum, int64_t n, int64_t nb,
                                                       const int32_t *__restrict__ n_dev, int64_t *__restrict__ total_out)
{
    __shared__ int64_t sh[SCAN_T];
    if (n_dev) n = *n_dev;
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_I;
    int32_t v[SCAN_I];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_I; k++) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_T; o <<= 1) {
        int64_t t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t run = bsum[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < SCAN_I; k++) if (base + k < n) { out[base + k] = (OutT)run; run += v[k]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[n] = (OutT)bsum[nb]; if (total_out) *total_out = bsum[nb]; }
}

void exclusive_scan_i32(const int32_t *in, int64_t *out, int64_t n, DBuf<int64_t> &scratch, hipStream_t s)
{
    if (n <= 0) { CP_HIP(hipMemsetAsync(out, 0, sizeof(int64_t), s)); return; }
    int64_t nb = cdiv(n, SCAN_TILE);
    scratch.ensure((size_t)nb + 1);
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, scratch.p, n, (const int32_t *)nullptr);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(1024), 0, s, scratch.p, nb);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<int64_t>), dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, out, scratch.p, n, nb,
                       (const int32_t *)nullptr, (int64_t *)nullptr);
    CP_HIP(hipGetLastError());
}

// same with the element count on the device (n_dev <= n_max); the total also goes to *total_out (device)
void exclusive_scan_i32_devn(const int32_t *in, int64_t *out, const int32_t *n_dev, int64_t n_max, int64_t *total_out,
                             DBuf<int64_t> &scratch, hipStream_t s)
{
    if (n_max <= 0) n_max = 1;
    int64_t nb = cdiv(n_max, SCAN_TILE);
    scratch.ensure((size_t)nb + 1);
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, scratch.p, (int64_t)0, n_dev);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(1024), 0, s, scratch.p, nb);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<int64_t>), dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, out, scratch.p, (int64_t)0, nb,
                       n_dev, total_out);
    CP_HIP(hipGetLastError());
}

// ---- single-launch scan: chained blocks, decoupled look-back
__device__ __forceinline__ unsigned long long lb_pack(uint32_t epoch, uint32_t status, int64_t v)
{
    return ((unsigned long long)epoch << 42) | ((unsigned long long)status << 40) | ((unsigned long long)v & 0xFFFFFFFFFFull);
}
__global__ void __launch_bounds__(SCAN_T) k_scan_lb(const int32_t *__restrict__ in, int64_t *__restrict__ out, const int32_t *__restrict__ n_dev,
                                                    int64_t *__restrict__ total_out, unsigned long long *__restrict__ st, uint32_t *__restrict__ ticket,
                                                    uint32_t tbase, uint32_t epoch)
{
    __shared__ int64_t sh[SCAN_T];
    __shared__ uint32_t s_bid;
    __shared__ int64_t s_prefix;
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u) - tbase;
    __syncthreads();
    const int64_t bid = s_bid, n = *n_dev, blast = n / SCAN_TILE;      // the block holding index n writes the total
    if (bid > blast) return;
    int64_t base = bid * SCAN_TILE + (int64_t)threadIdx.x * SCAN_I;
    int32_t v[SCAN_I];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_I; k++) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_T; o <<= 1) {
        int64_t t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int64_t tot = sh[SCAN_T - 1];
        int64_t prefix = 0;
        if (bid > 0) {
            __hip_atomic_store(&st[bid], lb_pack(epoch, 1, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t j = bid - 1;; ) {
                unsigned long long w = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(w >> 42) != epoch || ((w >> 40) & 3ull) == 0ull) { __builtin_amdgcn_s_sleep(1); continue; }      // not there yet
                prefix += (int64_t)(w & 0xFFFFFFFFFFull);
                if (((w >> 40) & 3ull) == 2ull) break;              // an inclusive prefix: done
                j--;
            }
        }
        __hip_atomic_store(&st[bid], lb_pack(epoch, 2, prefix + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_prefix = prefix;
        if (bid == blast) { out[n] = prefix + tot; if (total_out) *total_out = prefix + tot; }
    }
    __syncthreads();
    int64_t run = s_prefix + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < SCAN_I; k++) if (base + k < n) { out[base + k] = run; run += v[k]; }
}

void exclusive_scan_i32_lb(const int32_t *in, int64_t *out, const int32_t *n_dev, int64_t n_max, int64_t *total_out, ScanWS &ws, hipStream_t s)
{
    if (n_max <= 0) n_max = 1;
    const int64_t nb = n_max / SCAN_TILE + 1;                       // (covers the block that holds index n when n == n_max)
    if (ws.st.n < (size_t)nb || !ws.ticket.p) {
        ws.st.alloc((size_t)nb + 1024); ws.ticket.alloc(1);
        CP_HIP(hipMemsetAsync(ws.st.p, 0, ws.st.bytes(), s));
        CP_HIP(hipMemsetAsync(ws.ticket.p, 0, sizeof(uint32_t), s));
        ws.epoch = 0; ws.tbase = 0;
    }
    if (++ws.epoch >= (1u << 22)) { CP_HIP(hipMemsetAsync(ws.st.p, 0, ws.st.bytes(), s)); ws.epoch = 1; }
    hipLaunchKernelGGL(k_scan_lb, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, out, n_dev, total_out, ws.st.p, ws.ticket.p, ws.tbase, ws.epoch);
    ws.tbase += (uint32_t)nb;
    CP_HIP(hipGetLastError());
}

void exclusive_scan_i32_i32(const int32_t *in, int32_t *out, int64_t n, DBuf<int64_t> &scratch, hipStream_t s)
{
    if (n <= 0) { CP_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), s)); return; }
    int64_t nb = cdiv(n, SCAN_TILE);
    scratch.ensure((size_t)nb + 1);
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, scratch.p, n, (const int32_t *)nullptr);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(1024), 0, s, scratch.p, nb);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<int32_t>), dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, out, scratch.p, n, nb,
                       (const int32_t *)nullptr, (int64_t *)nullptr);
    CP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ CSR upload / normalisation
// The same pass validates what every later kernel indexes with: colptr[1] == 1, colptr non-decreasing, colptr[n+1] == nnz + 1,
// 1 <= rowval <= m.  A violation sets *bad (the handle is refused with CP_EINVAL instead of a GPU fault in the link build).
__global__ void k_norm_pos(const int64_t *__restrict__ colptr, int64_t *__restrict__ pos, int64_t n1, int64_t N, int32_t *__restrict__ bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const int64_t c = colptr[i];
    pos[i] = c - 1;
    bool ok = c >= 1 && c <= N + 1;
    if (i == 0) ok = ok && c == 1;
    if (i == n1 - 1) ok = ok && c == N + 1;
    if (i + 1 < n1) ok = ok && colptr[i + 1] >= c;
    if (!ok) *bad = 1;
}
__global__ void k_norm_row(const int64_t *__restrict__ rowval, int32_t *__restrict__ row, int64_t N, int64_t m, int32_t *__restrict__ bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t r = rowval[i];
    if (r < 1 || r > m) { *bad = 1; row[i] = 0; return; }
    row[i] = (int32_t)(r - 1);
}

void csr_upload(cp_csr_s *A, const int64_t *colptr, const int64_t *rowval, bool on_device)
{
    hipStream_t s = A->stream;
    A->pos.alloc((size_t)A->n + 1);
    A->row.alloc((size_t)(A->N > 0 ? A->N : 1));
    DBuf<int64_t> tmp_c, tmp_r;
    DBuf<int32_t> bad(1);
    const int64_t *dc = colptr, *dr = rowval;
    if (!on_device) {
        tmp_c.alloc((size_t)A->n + 1);
        tmp_r.alloc((size_t)(A->N > 0 ? A->N : 1));
        CP_HIP(hipMemcpyAsync(tmp_c.p, colptr, sizeof(int64_t) * (size_t)(A->n + 1), hipMemcpyHostToDevice, s));
        if (A->N > 0) CP_HIP(hipMemcpyAsync(tmp_r.p, rowval, sizeof(int64_t) * (size_t)A->N, hipMemcpyHostToDevice, s));
        dc = tmp_c.p; dr = tmp_r.p;
    }
    CP_HIP(hipMemsetAsync(bad.p, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_norm_pos, dim3((unsigned)cdiv(A->n + 1, 256)), dim3(256), 0, s, dc, A->pos.p, A->n + 1, A->N, bad.p);
    if (A->N > 0)
        hipLaunchKernelGGL(k_norm_row, dim3((unsigned)cdiv(A->N, 256)), dim3(256), 0, s, dr, A->row.p, A->N, A->m, bad.p);
    CP_HIP(hipGetLastError());
    int32_t hb = 0;
    CP_HIP(hipMemcpyAsync(&hb, bad.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    CP_REQUIRE(!hb, CP_EINVAL, "malformed pattern: colptr must start at 1, be non-decreasing and end at nnz+1; rowval must lie in 1..m");
}

// ------------------------------------------------------------------ adjointpattern / download
__global__ void k_adj_rows(const int32_t *__restrict__ tq, const int32_t *__restrict__ col, int32_t *__restrict__ row_t, int64_t N)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < N) row_t[q] = col[tq[q]];             // row-major order of A = column-major order of its adjoint
}
__global__ void k_to_julia(const int64_t *__restrict__ pos, const int32_t *__restrict__ row, int64_t *__restrict__ colptr,
                           int64_t *__restrict__ rowval, int64_t n1, int64_t N)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) colptr[i] = pos[i] + 1;
    if (i < N) rowval[i] = (int64_t)row[i] + 1;
}

void csr_adjoint(cp_csr_s *A, cp_csr_s *T)
{
    ensure_links(A);                               // tpos (row pointer) and tq (nonzeros by (row, column)) come from here
    hipStream_t s = A->stream;
    T->m = A->n; T->n = A->m; T->N = A->N;
    T->pos.alloc((size_t)T->n + 1);
    T->row.alloc((size_t)(T->N > 0 ? T->N : 1));
    CP_HIP(hipMemcpyAsync(T->pos.p, A->tpos.p, sizeof(int64_t) * (size_t)(T->n + 1), hipMemcpyDeviceToDevice, s));
    if (T->N > 0)
        hipLaunchKernelGGL(k_adj_rows, dim3((unsigned)cdiv(T->N, 256)), dim3(256), 0, s, A->tq.p, A->col.p, T->row.p, T->N);
    CP_HIP(hipGetLastError());
    CP_HIP(hipStreamSynchronize(s));
}

void csr_download(cp_csr_s *A, int64_t *colptr, int64_t *rowval)
{
    hipStream_t s = A->stream;
    int64_t n1 = A->n + 1, N = A->N, mx = n1 > N ? n1 : N;
    DBuf<int64_t> dc((size_t)n1), dr((size_t)(N > 0 ? N : 1));
    hipLaunchKernelGGL(k_to_julia, dim3((unsigned)cdiv(mx, 256)), dim3(256), 0, s, A->pos.p, A->row.p, dc.p, dr.p, n1, N);
    CP_HIP(hipGetLastError());
    CP_HIP(hipMemcpyAsync(colptr, dc.p, sizeof(int64_t) * (size_t)n1, hipMemcpyDeviceToHost, s));
    if (N > 0) CP_HIP(hipMemcpyAsync(rowval, dr.p, sizeof(int64_t) * (size_t)N, hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------ link arrays
// col[q]: one wave per 64 columns would idle on skew; a flat binary search per nonzero is regular.
__global__ void k_fill_col(const int64_t *__restrict__ pos, int32_t *__restrict__ col, int64_t n, int64_t N)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N) return;
    int64_t lo = 0, hi = n;            // largest c with pos[c] <= q
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (pos[mid] <= q) lo = mid; else hi = mid;
    }
    col[q] = (int32_t)lo;
}

__global__ void k_narrow(const int64_t *__restrict__ in, int32_t *__restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

__global__ void k_iota(uint32_t *__restrict__ v, int64_t N)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < N) v[q] = (uint32_t)q;
}

// sorted by (row, column): neighbours in the sorted order are the previous / next occurrence of a row
__global__ void k_links(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sq,
                        const int32_t *__restrict__ col, int32_t *__restrict__ prev, int32_t *__restrict__ next,
                        int32_t *__restrict__ rfirst, int32_t *__restrict__ rlast,
                        int64_t N, int32_t n)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    uint32_t r = skey[s], q = sq[s];
    int32_t c = col[q];
    bool first = (s == 0) || skey[s - 1] != r;
    bool last = (s + 1 == N) || skey[s + 1] != r;
    prev[q] = first ? -1 : col[sq[s - 1]];
    next[q] = last ? n : col[sq[s + 1]];
    if (first) rfirst[r] = c;
    if (last) rlast[r] = c;
}

// tpos[r] = first sorted position whose row >= r (row pointer of the transpose), r = 0..m
__global__ void k_tpos(const uint32_t *__restrict__ skey, int64_t *__restrict__ tpos, int64_t m, int64_t N)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > m) return;
    int64_t lo = 0, hi = N;            // first s with skey[s] >= r
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)skey[mid] < r) lo = mid + 1; else hi = mid;
    }
    tpos[r] = lo;
}

void ensure_links(cp_csr_s *A)
{
    if (A->have_links) return;
    hipStream_t s = A->stream;
    int64_t N = A->N, n = A->n, m = A->m;
    size_t Na = (size_t)(N > 0 ? N : 1);
    ProfScope ps(PROF_LINKS, s, 8.0 * (double)N + 8.0 * (double)(n + 1));
    A->col.ensure(Na); A->prev.ensure(Na + 16); A->next.ensure(Na + 16);   // +16: vector loads may over-read the tail
    A->rfirst.ensure((size_t)(m > 0 ? m : 1)); A->rlast.ensure((size_t)(m > 0 ? m : 1));
    A->tpos.ensure((size_t)m + 1); A->tq.ensure(Na);
    A->pos32.ensure((size_t)n + 1);
    hipLaunchKernelGGL(k_narrow, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, A->pos.p, A->pos32.p, n + 1);
    CP_HIP(hipMemsetAsync(A->rfirst.p, 0xFF, sizeof(int32_t) * (size_t)(m > 0 ? m : 1), s));
    CP_HIP(hipMemsetAsync(A->rlast.p, 0xFF, sizeof(int32_t) * (size_t)(m > 0 ? m : 1), s));
    CP_HIP(hipMemsetAsync(A->tpos.p, 0, sizeof(int64_t) * ((size_t)m + 1), s));
    if (N > 0) {
        hipLaunchKernelGGL(k_fill_col, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, A->pos.p, A->col.p, n, N);
        DBuf<uint32_t> kin(Na), kout(Na), vin(Na);
        CP_HIP(hipMemcpyAsync(kin.p, A->row.p, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_iota, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, vin.p, N);
        unsigned end_bit = 1;
        while (end_bit < 32 && ((uint64_t)1 << end_bit) < (uint64_t)(m > 1 ? m : 2)) end_bit++;
        size_t tmp_bytes = 0;
        CP_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin.p, kout.p, vin.p, (uint32_t *)A->tq.p, (size_t)N, 0u, end_bit, s));
        DBuf<char> tmp(tmp_bytes > 0 ? tmp_bytes : 1);
        CP_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tmp_bytes, kin.p, kout.p, vin.p, (uint32_t *)A->tq.p, (size_t)N, 0u, end_bit, s));
        hipLaunchKernelGGL(k_links, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, kout.p, (const uint32_t *)A->tq.p,
                           A->col.p, A->prev.p, A->next.p, A->rfirst.p, A->rlast.p, N, (int32_t)n);
        hipLaunchKernelGGL(k_tpos, dim3((unsigned)cdiv(m + 1, 256)), dim3(256), 0, s, kout.p, A->tpos.p, m, N);
        CP_HIP(hipGetLastError());
        CP_HIP(hipStreamSynchronize(s));   // kin/kout/vin/tmp die here
    }
    A->have_links = true;
}

// ------------------------------------------------------------------ rows bucketed by first / last column
__global__ void k_count_first_last(const int32_t *__restrict__ rfirst, const int32_t *__restrict__ rlast,
                                   int32_t *__restrict__ cf, int32_t *__restrict__ cl, int64_t m)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m || rfirst[r] < 0) return;
    atomicAdd(&cf[rfirst[r]], 1);
    atomicAdd(&cl[rlast[r]], 1);
}
__global__ void k_scatter_first_last(const int32_t *__restrict__ rfirst, const int32_t *__restrict__ rlast,
                                     const int64_t *__restrict__ fpos, const int64_t *__restrict__ lpos,
                                     int32_t *__restrict__ cf, int32_t *__restrict__ cl,
                                     int32_t *__restrict__ flast, int32_t *__restrict__ lfirst, int32_t *__restrict__ ffirst, int64_t m)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m || rfirst[r] < 0) return;
    // order inside a bucket is irrelevant: only counts of entries below/above a threshold are used
    int a = atomicAdd(&cf[rfirst[r]], 1);
    flast[fpos[rfirst[r]] + a] = rlast[r];
    ffirst[fpos[rfirst[r]] + a] = rfirst[r];
    int b = atomicAdd(&cl[rlast[r]], 1);
    lfirst[lpos[rlast[r]] + b] = rfirst[r];
}

void ensure_self(cp_csr_s *A)
{
    ensure_links(A);
    if (A->have_self) return;
    hipStream_t s = A->stream;
    int64_t n = A->n, m = A->m;
    DBuf<int32_t> cf((size_t)n + 1), cl((size_t)n + 1);
    DBuf<int64_t> scratch;
    CP_HIP(hipMemsetAsync(cf.p, 0, cf.bytes(), s));
    CP_HIP(hipMemsetAsync(cl.p, 0, cl.bytes(), s));
    if (m > 0) hipLaunchKernelGGL(k_count_first_last, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, s, A->rfirst.p, A->rlast.p, cf.p, cl.p, m);
    A->fpos.ensure((size_t)n + 1); A->lpos.ensure((size_t)n + 1);
    exclusive_scan_i32(cf.p, A->fpos.p, n, scratch, s);
    exclusive_scan_i32(cl.p, A->lpos.p, n, scratch, s);
    int64_t tot = 0;
    CP_HIP(hipMemcpyAsync(&tot, A->fpos.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    CP_HIP(hipStreamSynchronize(s));
    A->nrows_nonempty = tot;
    A->flast.ensure((size_t)(tot > 0 ? tot : 1) + 16); A->lfirst.ensure((size_t)(tot > 0 ? tot : 1) + 16);   // +16: vector over-read
    A->ffirst.ensure((size_t)(tot > 0 ? tot : 1) + 8);
    CP_HIP(hipMemsetAsync(cf.p, 0, cf.bytes(), s));
    CP_HIP(hipMemsetAsync(cl.p, 0, cl.bytes(), s));
    if (m > 0) hipLaunchKernelGGL(k_scatter_first_last, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, s, A->rfirst.p, A->rlast.p,
                                  A->fpos.p, A->lpos.p, cf.p, cl.p, A->flast.p, A->lfirst.p, A->ffirst.p, m);
    A->fpos32.ensure((size_t)n + 1); A->lpos32.ensure((size_t)n + 1);
    hipLaunchKernelGGL(k_narrow, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, A->fpos.p, A->fpos32.p, n + 1);
    hipLaunchKernelGGL(k_narrow, dim3((unsigned)cdiv(n + 1, 256)), dim3(256), 0, s, A->lpos.p, A->lpos32.p, n + 1);
    CP_HIP(hipGetLastError());
    CP_HIP(hipStreamSynchronize(s));
    A->have_self = true;
}

void drop_cache(cp_csr_s *A)
{
    A->have_links = false; A->have_self = false;
    for (int i = 0; i < 2; i++) if (A->dp_work[i] && A->dp_work_reset_fn[i]) A->dp_work_reset_fn[i](A->dp_work[i]);
    if (A->bn_work && A->bn_work_reset_fn) A->bn_work_reset_fn(A->bn_work);
    // (the arrays themselves stay allocated: the next build refills them -- a multi-GB hipFree + hipMalloc pair per call buys nothing)
}

}  // namespace cpk
