// todo.hip -- entry points declared in include/chainpart.h whose device paths are not written yet.
// They refuse loudly (CP_EUNSUPPORTED + message); nothing here computes on the CPU.
#include "common.hpp"
using namespace cpk;
#define CP_TODO(msg) do { set_error(msg); return CP_EUNSUPPORTED; } while (0)
extern "C" {
int32_t cp_partwise(cp_csr_t, int64_t, const int64_t *, int64_t *, int64_t *, int64_t *, int64_t *, int64_t *) { CP_TODO("partwise: device path pending"); }
int32_t cp_pack_dynamic(cp_csr_t, const cp_model_t *, const cp_rowpart_t *, const cp_model_t *, int64_t, double, int64_t *, int64_t *) { CP_TODO("pack_stripe(DynamicTotalChunker): device path pending"); }
int32_t cp_pack_convex(cp_csr_t, const cp_model_t *, const cp_rowpart_t *, const cp_model_t *, int64_t, double, int64_t *, int64_t *) { CP_TODO("ConvexTotalChunker: device path pending"); }
int32_t cp_partition_convex(cp_csr_t, int64_t, const cp_model_t *, const cp_rowpart_t *, const cp_model_t *, int64_t, double, int64_t *) { CP_TODO("ConvexTotalSplitter: device path pending"); }
}
