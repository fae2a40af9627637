// todo.hip -- entry points declared in include/chainpart.h whose device paths are not written yet.
// They refuse loudly (CP_EUNSUPPORTED + message); nothing here computes on the CPU.
#include "common.hpp"
using namespace cpk;
#define CP_TODO(msg) do { set_error(msg); return CP_EUNSUPPORTED; } while (0)
extern "C" {
int32_t cp_partwise(cp_csr_t, int64_t, const int64_t *, int64_t *, int64_t *, int64_t *, int64_t *, int64_t *) { CP_TODO("partwise: device path pending"); }
}
