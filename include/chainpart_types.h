/*
 * chainpart_types.h -- plain-C value types shared by the product C-ABI
 * (include/chainpart.h) and by the test oracle (oracle/orc.h).
 *
 * Every struct here is a flat, pointer-and-size description of something the
 * reference expresses as a Julia struct.  All index VALUES are 1-based, exactly
 * as Julia stores them (colptr[1] == 1, rowval in 1:m, spl[1] == 1, ...).
 *
 * Reference types mirrored (file:line under /root/reference/src):
 *   AffineWorkModel             WorkCosts.jl:5-17
 *   AffineConnectivityModel     ConnectivityCosts.jl:7-20
 *   AffineHyperedgeCutModel     HyperedgeCutCosts.jl:7-21
 *   ColumnBlockComponentCostModel / BlockComponentCostModel  BlockCosts.jl:1-44
 *   VertexCount (a weight)      SparseColorArrays.jl:1-6
 *   FeasibleCost (no weight)    Costs.jl:153-171
 *   per-part alpha[k] models ("Funky*")  test/test_Partitioners.jl:1-8
 */
#ifndef CHAINPART_TYPES_H
#define CHAINPART_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes returned by every entry point */
#define CP_OK            0
#define CP_EINVAL        1   /* bad argument / violated precondition (Julia: AssertionError) */
#define CP_INFEASIBLE    2   /* width constraint infeasible: degenerate partition [1,..,1,n+1] written
                                (DynamicSplitter.jl:217-222) */
#define CP_EHIP          3   /* HIP / RCCL runtime error */
#define CP_EUNSUPPORTED  4   /* (method, model) pair has no device path; nothing written */
#define CP_EINTERNAL     5   /* an internal invariant failed (a bug in the library): nothing valid written */

/* cost element type Tc (= the model's Tv) */
#define CP_I64 0
#define CP_F64 1

/* model kinds */
#define CP_MODEL_FEASIBLE       0  /* FeasibleCost(): "no constraint" weight */
#define CP_MODEL_WORK           1  /* alpha + nv*b_vertex + np*b_pin */
#define CP_MODEL_CONNECTIVITY   2  /* ... + nets*b_net */
#define CP_MODEL_HYPEREDGE_CUT  3  /* ... + selfnets*b_self_net + cutnets*b_cut_net */
#define CP_MODEL_COLBLOCK       4  /* alpha_col(w) + nets*beta_col(w) */
#define CP_MODEL_BLOCK          5  /* rank-R separable 2-D VBR cost, needs a row partition */
#define CP_MODEL_VERTEX_COUNT   6  /* VertexCount(): j' - j, always Int */
#define CP_MODEL_PRIMARY        8  /* AffinePrimaryConnectivityModel: alpha + nv*b_vertex + np*b_pin + local*b_local_net +
                                      remote*b_remote_net; nets of the part split by the row partition Pi into those owned by
                                      the same part number (local) and the others (PrimaryConnectivityCosts.jl:5-20, :53-78) */
#define CP_MODEL_SECONDARY      9  /* AffineSecondaryConnectivityModel (SecondaryConnectivityCosts.jl:5-20, :63-86): cost of
                                      giving the column range [j, j') to part k of the SplitPartition Pi of the ROWS */
#define CP_MODEL_POWER_WORK     7  /* alpha + (nv*b_vertex + np*b_pin)^gamma, Float64 only, gamma in p_f64[3]: the
                                      ConvexWorkModel (gamma = 0.8) / ConcaveWorkModel (gamma = 2) the reference's
                                      tests define (test/test_Partitioners.jl:54-74); gamma == 2 is evaluated as x*x
                                      (Julia's literal_pow), any other exponent with pow() */

/* parameter slots of p_i64 / p_f64 */
#define CP_P_ALPHA      0
#define CP_P_VERTEX     1
#define CP_P_PIN        2
#define CP_P_NET        3   /* connectivity: b_net ; hyperedge: b_self_net */
#define CP_P_SELF_NET   3
#define CP_P_CUT_NET    4
#define CP_P_GAMMA      3   /* power work model: the exponent */
#define CP_P_LOCAL_NET  3   /* primary / secondary connectivity */
#define CP_P_REMOTE_NET 4

#define CP_MAX_R 4

/* block_component(f, w) (BlockCosts.jl:41-44): a number, or a closure/tuple/array
 * that the host tabulates for w = lo .. lo+len-1 (closures cannot cross a C ABI).  lo < 0 is needed by the
 * Convex/Concave chunkers: on a cost that is not convex their candidate stack goes stale and they evaluate f(j, j')
 * with j > j' (ConvexTotalChunker.jl:76, :99), i.e. the closure at a negative width. */
typedef struct cp_component {
    int32_t is_const;      /* 1: value is c_*; 0: value is table[w - lo] */
    int32_t _pad;
    int64_t c_i64;
    double  c_f64;
    const void *table;     /* int64_t[len] or double[len] according to the model dtype */
    int64_t len;
    int64_t lo;            /* width of table[0] (0 for plain tables) */
} cp_component_t;

typedef struct cp_model {
    int32_t kind;          /* CP_MODEL_* */
    int32_t dtype;         /* CP_I64 / CP_F64 */
    int64_t p_i64[5];      /* used when dtype == CP_I64 */
    double  p_f64[5];      /* used when dtype == CP_F64 */
    const void *alpha_k;   /* optional per-part alpha[k], k = 1..n_alpha_k (element type = dtype); NULL if none */
    int64_t n_alpha_k;
    int32_t R;             /* rank of CP_MODEL_BLOCK (<= CP_MAX_R) */
    int32_t _pad;
    cp_component_t alpha_row, alpha_col;
    cp_component_t beta_row[CP_MAX_R], beta_col[CP_MAX_R];
} cp_model_t;

/* a row partition Pi handed to block models / partwise counts:
 * MapPartition.asg (length m, values 1..K) and SplitPartition.spl (length K+1) */
typedef struct cp_rowpart {
    int64_t K;
    const int64_t *asg;    /* may be NULL when only spl is meaningful */
    const int64_t *spl;    /* may be NULL for a pure MapPartition */
} cp_rowpart_t;

/* objective combiners: DynamicTotal* uses +, DynamicBottleneck* uses max */
#define CP_COMBINE_SUM 0
#define CP_COMBINE_MAX 1

/* loop order of the K-part DP (DynamicSplitter.jl:15-50 vs :52-87) */
#define CP_ORDER_SPLITTER 0
#define CP_ORDER_CHUNKER  1

/* access-pattern hints select the counting structure in the reference
 * (SparsePrefixMatrices.jl:448-458) */
#define CP_HINT_NONE   0   /* BinaryDominanceCount */
#define CP_HINT_RANDOM 1   /* BinaryDominanceCount */
#define CP_HINT_SPARSE 2   /* DominanceCount (radix tree) */
#define CP_HINT_STEP   3   /* SparseStepwiseDominanceCount */

/* moves of the Step protocol: Step(ocl)(Same(j) | Next(j) | Prev(j) | Jump(j), ...)  (Costs.jl:174-195) */
#define CP_MOVE_SAME 0
#define CP_MOVE_NEXT 1
#define CP_MOVE_PREV 2
#define CP_MOVE_JUMP 3

#ifdef __cplusplus
}
#endif
#endif
