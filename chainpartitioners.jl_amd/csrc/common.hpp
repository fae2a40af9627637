// common.hpp -- shared device/host utilities of libchainpart (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include "../../include/chainpart.h"

namespace cpk {

// ------------------------------------------------------------------ errors
extern thread_local std::string g_last_error;
inline void set_error(const std::string &s) { g_last_error = s; }

struct HipFail { int32_t code; };

#define CP_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            char _b[512];                                                                   \
            snprintf(_b, sizeof(_b), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,          \
                     hipGetErrorString(_e));                                                \
            cpk::set_error(_b);                                                             \
            throw cpk::HipFail{CP_EHIP};                                                    \
        }                                                                                   \
    } while (0)

#define CP_REQUIRE(cond, code, msg)                                                         \
    do {                                                                                    \
        if (!(cond)) { cpk::set_error(msg); throw cpk::HipFail{code}; }                     \
    } while (0)

// closes the try block of an extern "C" entry point: nothing may unwind through the C boundary (HIP / argument failures carry
// their status code; a failed host allocation -- std::vector, new -- becomes CP_EHIP with a message)
#define CP_CATCH_ALL                                                                         \
    catch (const cpk::HipFail &e) { return e.code; }                                         \
    catch (const std::bad_alloc &) { cpk::set_error("host allocation failed"); return CP_EHIP; }

// ------------------------------------------------------------------ device buffers (RAII)
// Allocations of 1 MB and more go through a per-process pool keyed by (device, exact byte size): the builds and drivers allocate the
// same multi-hundred-MB temporaries in every call, and handing them back to the HIP runtime each time makes it stall for ~1.3 s
// once per ~50 GB of such churn (tools/stall_probe.py: every ~20th link build; a 0.1 s bottleneck partition then takes 1 s).
// A block freed while its stream still runs kernels is only ever handed to work enqueued LATER by the next call -- every entry
// point synchronises its stream before it returns -- so reuse is stream-ordered.  cp_set_option("pool", 0) turns the pool off and
// returns what it holds; over 48 GB it empties itself; a failed hipMalloc empties it and retries.
void *dev_alloc(size_t bytes);
void dev_free(void *p, size_t bytes);
void dev_pool_trim();
extern int64_t g_opt_pool;

template <typename T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    DBuf() = default;
    explicit DBuf(size_t count) { alloc(count); }
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    DBuf(DBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DBuf &operator=(DBuf &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DBuf() { release(); }
    void alloc(size_t count) {
        release();
        if (count) p = (T *)dev_alloc(count * sizeof(T));      // (throws with n == 0, p == nullptr: a failed buffer never claims a size)
        n = count;
    }
    void ensure(size_t count) { if (count > n) alloc(count); }
    void release() { if (p) { dev_free(p, n * sizeof(T)); p = nullptr; } n = 0; }
    size_t bytes() const { return n * sizeof(T); }
};

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ per-kernel HIP-event profiling
struct ProfSlot { const char *name; int64_t launches; double ms; double alg_bytes; };
enum { PROF_EXPAND = 0, PROF_EVAL, PROF_SETUP, PROF_SCAN, PROF_CARRY, PROF_FIX, PROF_COMBINE, PROF_LINKS,
       PROF_BRUTE, PROF_WAVELET, PROF_QUERY, PROF_BISECT, PROF_CHUNK, PROF_RPASS, PROF_OWN, PROF_GAP, PROF_GAPSTREAM, PROF_RA, PROF_LEAF, PROF_NSLOTS };
extern ProfSlot g_prof[PROF_NSLOTS];
extern bool g_prof_on;
extern int g_prof_only;              // >= 0: only this slot records events (cp_set_option("prof_only")): 2 events per round instead of ~40
inline bool prof_active(int slot) { return g_prof_on && (g_prof_only < 0 || g_prof_only == slot); }

struct ProfPending { int slot; hipEvent_t a, b; double bytes; };
extern std::vector<ProfPending> g_prof_pending;
extern std::vector<hipEvent_t> g_event_pool;

inline hipEvent_t prof_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e; CP_HIP(hipEventCreate(&e)); return e;
}

// RAII scope: records an event pair around the launches issued inside it
struct ProfScope {
    int slot; hipStream_t s; hipEvent_t a{}, b{}; double bytes; bool on;
    ProfScope(int slot_, hipStream_t s_, double alg_bytes) : slot(slot_), s(s_), bytes(alg_bytes), on(prof_active(slot_)) {
        if (on) { a = prof_event(); b = prof_event(); CP_HIP(hipEventRecord(a, s)); }
    }
    ~ProfScope() {
        if (on) { (void)hipEventRecord(b, s); g_prof_pending.push_back({slot, a, b, bytes}); }
    }
};

void prof_collect();   // resolve pending event pairs (after a stream sync)

// ------------------------------------------------------------------ device-wide exclusive scan (int32 -> int64)
// out has n+1 entries: out[i] = sum_{t<i} in[t], out[n] = total.
void exclusive_scan_i32(const int32_t *in, int64_t *out, int64_t n, DBuf<int64_t> &scratch, hipStream_t s);
void exclusive_scan_i32_devn(const int32_t *in, int64_t *out, const int32_t *n_dev, int64_t n_max, int64_t *total_out,
                             DBuf<int64_t> &scratch, hipStream_t s);
// Single-launch form (decoupled look-back) for the scans inside the DP rounds: dozens of small scans per layer, where three
// launches each are most of the cost.  Sums must stay below 2^40 (they are step / tile counts).  One ScanWS per stream of work.
struct ScanWS {
    DBuf<unsigned long long> st;      // per block: [epoch:22][status:2][value:40]
    DBuf<uint32_t> ticket;            // blocks take their index here (a block only waits for blocks that already run)
    uint32_t epoch = 0, tbase = 0;
};
void exclusive_scan_i32_lb(const int32_t *in, int64_t *out, const int32_t *n_dev, int64_t n_max, int64_t *total_out, ScanWS &ws, hipStream_t s);
void exclusive_scan_i32_i32(const int32_t *in, int32_t *out, int64_t n, DBuf<int64_t> &scratch, hipStream_t s);

// wave64 helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

}  // namespace cpk
