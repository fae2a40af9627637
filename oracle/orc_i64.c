/* TEST ORACLE (not product code): Int64 cost instantiation of orc_algos.inc */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
#include "orc.h"
#define TC int64_t
#define SFX(x) x##_i64
#define TC_IS_INT 1
#define TC_TYPEMAX INT64_MAX
#define TC_TYPEMIN INT64_MIN
#include "orc_algos.inc"
