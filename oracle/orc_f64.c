/* TEST ORACLE (not product code): Float64 cost instantiation of orc_algos.inc */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
#include "orc.h"
#define TC double
#define SFX(x) x##_f64
#define TC_IS_INT 0
#define TC_TYPEMAX ((double)INFINITY)
#define TC_TYPEMIN (-(double)INFINITY)
#include "orc_algos.inc"
