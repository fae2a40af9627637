// csr.hpp -- device-resident sparsity pattern and the link arrays every oracle is built from.
#pragma once
#include "common.hpp"

struct cp_csr_s {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t m = 0, n = 0, N = 0;
    cpk::DBuf<int64_t> pos;     // n+1 : 0-based start of each column
    cpk::DBuf<int32_t> row;     // N   : 0-based row of each nonzero (column-major, rows ascending)
    // ---- derived, built lazily (ensure_links) ----
    bool have_links = false;
    cpk::DBuf<int32_t> col;     // N : column of each nonzero
    cpk::DBuf<int32_t> prev;    // N : previous column holding the same row, -1 if none
    cpk::DBuf<int32_t> next;    // N : next column holding the same row, n if none
    cpk::DBuf<int32_t> rfirst;  // m : first column of each row (-1 if the row is empty)
    cpk::DBuf<int32_t> rlast;   // m : last column
    cpk::DBuf<int32_t> pos32;   // n+1 : pos narrowed to 32 bits (N < 2^31) for the DP kernels
    cpk::DBuf<int64_t> tpos;    // m+1 : start of each row in the row-major order (transpose pointer)
    cpk::DBuf<int32_t> tq;      // N : nonzero ids sorted by (row, column)
    // rows bucketed by first column (self-net left steps) and by last column (right steps)
    bool have_self = false;
    cpk::DBuf<int64_t> fpos;    // n+1
    cpk::DBuf<int32_t> flast;   // #nonempty rows : last column of rows whose first column is c
    cpk::DBuf<int64_t> lpos;    // n+1
    cpk::DBuf<int32_t> fpos32, lpos32;   // 32-bit copies
    cpk::DBuf<int32_t> lfirst;  // #nonempty rows : first column of rows whose last column is c
    cpk::DBuf<int32_t> ffirst;  // #nonempty rows : the bucket (= first column) of every flast entry
    int64_t nrows_nonempty = 0;
    // scratch of the total-cost DP layers ([0]: Int64 costs, [1]: Float64 costs), kept between calls: several GB that would
    // otherwise be allocated and freed by every partition call.  Its pattern-dependent parts are dropped with the cache.
    void *dp_work[2] = {nullptr, nullptr};
    void (*dp_work_free_fn[2])(void *) = {nullptr, nullptr};
    void (*dp_work_reset_fn[2])(void *) = {nullptr, nullptr};
    // state of the bottleneck DP layers (wavelet counters for the chunk starts, per-layer scratch): dp_bottleneck.hip
    void *bn_work = nullptr;
    void (*bn_work_free_fn)(void *) = nullptr;
    void (*bn_work_reset_fn)(void *) = nullptr;
    ~cp_csr_s()
    {
        for (int i = 0; i < 2; i++) if (dp_work[i] && dp_work_free_fn[i]) dp_work_free_fn[i](dp_work[i]);
        if (bn_work && bn_work_free_fn) bn_work_free_fn(bn_work);
        if (own_stream && stream) (void)hipStreamDestroy(stream);      // (also on the error paths of the create entry points)
    }
};

namespace cpk {
void csr_upload(cp_csr_s *A, const int64_t *colptr, const int64_t *rowval, bool on_device);
void ensure_links(cp_csr_s *A);
void ensure_self(cp_csr_s *A);
void drop_cache(cp_csr_s *A);
void csr_adjoint(cp_csr_s *A, cp_csr_s *T);                                 // T: fresh handle (device / stream set by the caller)
void csr_download(cp_csr_s *A, int64_t *colptr, int64_t *rowval);
}  // namespace cpk
