/*
 * orc_counts.c -- TEST ORACLE (not product code): literal C restatement of the
 * reference's sparse counting structures.  See orc.h for the parity statement.
 *
 * Follows /root/reference/src:
 *   util.jl:3-15                       fllog2 / cllog2 / fld2
 *   SparsePrefixMatrices.jl:462-604    DominanceCount (radix tree, SparseHint)
 *   SparsePrefixMatrices.jl:606-689    BinaryDominanceCount (NoHint / RandomHint)
 *   SparsePrefixMatrices.jl:693-821    SparseStepwiseDominanceCount (StepHint)
 *   SparseColorArrays.jl:47-152        NetCount
 *   SparseColorArrays.jl:156-256       SelfNetCount
 *   PartwiseCounts.jl:1-60             partwise
 *
 * Julia precedence reminders (<<,>> bind tighter than * and &, which bind tighter
 * than + - |):  "1 << b + 1" = (1<<b)+1 ;  "i >> s & m + 1" = ((i>>s)&m)+1 ;
 * "i & ~(1 << h - 1) + 1" = (i & ~((1<<h)-1)) + 1 ;  "N >> b' + 1" = (N>>b')+1 ;
 * "i' + d << s" = i' + (d<<s).
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

#define A1(a, k) ((a)[(k) - 1])            /* Julia 1-based element access */
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

static inline int64_t fllog2(int64_t x) { return 63 - __builtin_clzll((unsigned long long)x); }
/* cllog2(1) = fllog2(0)+1 where leading_zeros(0)=64 -> 63-64+1 = 0 (util.jl:6-8) */
int64_t orc_cllog2(int64_t x) { return (x - 1) == 0 ? 0 : fllog2(x - 1) + 1; }
static inline int64_t cld(int64_t a, int64_t b) { return (a + b - 1) / b; }  /* a,b > 0 */

struct orc_dom {
    int32_t hint;
    int64_t m, n, N;
    int64_t *pos;           /* n+1 */
    /* binary */
    int64_t H;
    int64_t *qos;           /* m+2 */
    uint64_t *byt;          /* (1+cld(N,64)) x H column-major */
    int64_t *cnt;           /* (1+cld(N,64)) x H */
    int64_t W;              /* leading dimension 1+cld(N,64) */
    /* radix */
    int64_t b, bp;
    int64_t *rbyt;          /* N permuted keys */
    int64_t *rcnt;          /* ((1<<b)+1) x ((N>>bp)+1) x H */
    int64_t d1, d2;         /* dims 1 and 2 of rcnt */
    /* stepwise */
    int64_t si, sj, sc;
    int64_t *idx;           /* N (copy) */
    int64_t *delta;         /* m */
};

/* ---------------- BinaryDominanceCount: SparsePrefixMatrices.jl:610-655 ---------------- */
static void binary_build(orc_dom *D, int64_t *idx)
{
    int64_t m = D->m, N = D->N;
    int64_t H = orc_cllog2(m + 1);
    D->H = H;
    int64_t W = 1 + (N + 63) / 64;          /* 1 + cld(N, nbits(UInt)) */
    D->W = W;
    int64_t *qos = (int64_t *)calloc((size_t)(m + 2), sizeof(int64_t));
    A1(qos, 1) = 1;
    A1(qos, m + 2) = N + 1;
    int64_t Hd = H > 0 ? H : 1;
    int64_t *cnt = (int64_t *)calloc((size_t)(W * Hd), sizeof(int64_t));
    uint64_t *byt = (uint64_t *)calloc((size_t)(W * Hd), sizeof(uint64_t));
    int64_t *idx2 = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    int64_t qend = m + 2; /* "end" of qos */
#define CNT(Q, h) cnt[((h) - 1) * W + (Q) - 1]
#define BYT(Q, h) byt[((h) - 1) * W + (Q) - 1]
    for (int64_t h = H; h >= 1; h--) {
        int64_t _cnt = 0;
        for (int64_t ip = 1; ip <= m + 1; ip += ((int64_t)1 << h)) {
            int64_t bkt_1 = 0, bkt_2 = 0;
            int64_t q_lo = A1(qos, ip);
            int64_t q_hi = A1(qos, MIN(ip + ((int64_t)1 << h), qend)) - 1;
            for (int64_t q = q_lo; q <= q_hi; q++) {
                int64_t i = A1(idx, q);
                int64_t d = (i >> (h - 1)) & 1;
                int64_t Q = ((q - 1) >> 6) + 1;
                BYT(Q, h) |= (uint64_t)d << ((q - 1) & 63);
                _cnt += d;
                CNT(Q + 1, h) = _cnt;
                bkt_2 += 1 - d;
            }
            bkt_1 = A1(qos, ip);
            bkt_2 += bkt_1;
            for (int64_t q = q_lo; q <= q_hi; q++) {
                int64_t i = A1(idx, q);
                int64_t d = (i >> (h - 1)) & 1;
                int64_t qp = d == 0 ? bkt_1 : bkt_2;
                A1(idx2, qp) = i;
                bkt_1 += 1 - d;
                bkt_2 += d;
            }
            A1(qos, MIN(ip + ((int64_t)1 << (h - 1)), qend)) = bkt_1;
        }
        int64_t *t = idx; idx = idx2; idx2 = t;
    }
#undef CNT
#undef BYT
    /* idx / idx2 are the two scratch key buffers (swapped H times); neither is needed
     * by queries */
    D->qos = qos; D->cnt = cnt; D->byt = byt;
    free(idx); free(idx2);
    D->idx = NULL;
}

/* SparsePrefixMatrices.jl:660-689 */
static int64_t binary_query(const orc_dom *D, int64_t i, int64_t j)
{
    int64_t H = D->H, W = D->W;
    const int64_t *qos = D->qos, *cnt = D->cnt;
    const uint64_t *byt = D->byt;
    int64_t dq = A1(D->pos, j) - 1;
    i = i - 1;
    int64_t s = 0;
    for (int64_t h = H; h >= 1; h--) {
        int64_t ip = (i & ~(((int64_t)1 << h) - 1)) + 1;
        int64_t q1 = A1(qos, ip) - 1;
        int64_t q2 = q1 + dq;
        int64_t d = (i >> (h - 1)) & 1;
        int64_t Q1 = (q1 >> 6) + 1;
        int64_t Q2 = (q2 >> 6) + 1;
        int64_t bkt_2 = cnt[(h - 1) * W + Q2 - 1] - cnt[(h - 1) * W + Q1 - 1];
        bkt_2 += __builtin_popcountll(byt[(h - 1) * W + Q2 - 1] & (((uint64_t)1 << (q2 & 63)) - 1));
        bkt_2 -= __builtin_popcountll(byt[(h - 1) * W + Q1 - 1] & (((uint64_t)1 << (q1 & 63)) - 1));
        int64_t bkt_1 = dq - bkt_2;
        s += d == 0 ? 0 : bkt_1;
        dq = d == 0 ? bkt_1 : bkt_2;
    }
    return s + dq;
}

/* ---------------- DominanceCount (radix): SparsePrefixMatrices.jl:462-604 ---------------- */
static void radix_build(orc_dom *D, int64_t *idx, int64_t b, int64_t H, int64_t bp)
{
    int64_t m = D->m, N = D->N;
    if (b <= 0) {
        if (H <= 0) b = cld(orc_cllog2(m + 1), 3);
        else b = cld(orc_cllog2(m + 1), H);
        if (b <= 0) b = 1;                 /* cld(0,3)=0 when m=0; keep the tree well-formed */
    }
    if (H <= 0) H = cld(orc_cllog2(m + 1), b);
    if (H <= 0) H = 1;
    if (bp <= 0) bp = b + orc_cllog2(H);
    D->b = b; D->H = H; D->bp = bp;
    int64_t nb = (int64_t)1 << b;
    int64_t *qos = (int64_t *)calloc((size_t)(m + 2), sizeof(int64_t));
    int64_t qend = m + 2;
    A1(qos, 1) = 1;
    A1(qos, qend) = N + 1;
    int64_t *bkt = (int64_t *)malloc((size_t)(nb + 1) * sizeof(int64_t));
    int64_t d1 = nb + 1, d2 = (N >> bp) + 1;
    D->d1 = d1; D->d2 = d2;
    int64_t *cnt = (int64_t *)calloc((size_t)(d1 * d2 * H), sizeof(int64_t));
    int64_t *byt = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
#define RCNT(d, Q, h) cnt[(((h) - 1) * d2 + ((Q) - 1)) * d1 + ((d) - 1)]
    for (int64_t h = H; h >= 1; h--) {
        int64_t span = (int64_t)1 << (h * b);
        int64_t sh = (h - 1) * b;
        for (int64_t ip = 1; ip <= m + 1; ip += span) {
            memset(bkt, 0, (size_t)(nb + 1) * sizeof(int64_t));
            int64_t q_lo = A1(qos, ip);
            int64_t q_hi = A1(qos, MIN(ip + span, qend)) - 1;
            for (int64_t q = q_lo; q <= q_hi; q++) {
                int64_t i = A1(idx, q);
                int64_t d = ((i >> sh) & (nb - 1)) + 1;
                A1(bkt, d + 1) += 1;
            }
            A1(bkt, 1) = A1(qos, ip);
            for (int64_t d = 1; d <= nb; d++) A1(bkt, d + 1) = A1(bkt, d) + A1(bkt, d + 1);
            for (int64_t q = q_lo; q <= q_hi; q++) {
                int64_t i = A1(idx, q);
                int64_t d = ((i >> sh) & (nb - 1)) + 1;
                int64_t qp = A1(bkt, d);
                int64_t lowmask = ((int64_t)1 << sh) - 1;
                /* literal (:508): HIGH bits (digit h and above) come from the element that sat
                 * at q' BEFORE this level's shuffle, LOW bits from the element moved there.
                 * The final byt therefore carries, for every level h, digit h of the level-h
                 * pre-shuffle sequence at each position -- which is what the query scans. */
                A1(byt, qp) = (A1(idx, qp) & ~lowmask) | (i & lowmask);
                A1(bkt, d) = qp + 1;
            }
            for (int64_t d = 1; d <= nb; d++)
                A1(qos, MIN(ip + (d << sh), qend)) = A1(bkt, d);
        }
        RCNT(1, 1, h) = 0;
        for (int64_t d = 1; d <= nb; d++) RCNT(d + 1, 1, h) = 0;
        memset(bkt, 0, (size_t)(nb + 1) * sizeof(int64_t));
        for (int64_t q = 1; q <= N; q++) {
            int64_t i = A1(idx, q);
            int64_t d = ((i >> sh) & (nb - 1)) + 1;
            A1(bkt, d) += 1;
            if ((q & (((int64_t)1 << bp) - 1)) == 0) {
                int64_t Q = (q >> bp) + 1;
                RCNT(1, Q, h) = 0;
                for (int64_t dd = 1; dd <= nb; dd++) RCNT(dd + 1, Q, h) = A1(bkt, dd) + RCNT(dd, Q, h);
            }
        }
        int64_t *t = idx; idx = byt; byt = t;
    }
#undef RCNT
    /* after the loop the reference does `byt = idx` (:546): the last written array */
    D->rbyt = idx;
    D->rcnt = cnt;
    D->qos = qos;
    free(bkt);
    if (byt != idx) free(byt);
    D->idx = NULL;
}

/* SparsePrefixMatrices.jl:551-604 */
static int64_t radix_query(const orc_dom *D, int64_t i, int64_t j)
{
    int64_t b = D->b, bp = D->bp, H = D->H, d1 = D->d1, d2 = D->d2;
    const int64_t *qos = D->qos, *byt = D->rbyt, *cnt = D->rcnt;
    int64_t nb = (int64_t)1 << b;
#define RCNT(d, Q, h) cnt[(((h) - 1) * d2 + ((Q) - 1)) * d1 + ((d) - 1)]
    int64_t dq = A1(D->pos, j) - 1;
    i = i - 1;
    int64_t s = 0;
    for (int64_t h = H; h >= 2; h--) {
        int64_t ip = (i & ~(((int64_t)1 << (h * b)) - 1)) + 1;
        int64_t q1 = A1(qos, ip) - 1;
        int64_t q2 = q1 + dq;
        int64_t d = ((i >> ((h - 1) * b)) & (nb - 1)) + 1;
        int64_t Q1 = (q1 >> bp) + 1;
        int64_t Q2 = (q2 >> bp) + 1;
        s += RCNT(d, Q2, h) - RCNT(d, Q1, h);
        dq = (RCNT(d + 1, Q2, h) - RCNT(d, Q2, h)) - (RCNT(d + 1, Q1, h) - RCNT(d, Q1, h));
        int64_t msk = (nb - 1) << ((h - 1) * b);
        int64_t cmp = (d - 1) << ((h - 1) * b);
        for (int64_t q = ((Q1 - 1) << bp) + 1; q <= q1; q++) {
            int64_t dp = A1(byt, q) & msk;
            s -= dp < cmp;
            dq -= dp == cmp;
        }
        for (int64_t q = ((Q2 - 1) << bp) + 1; q <= q2; q++) {
            int64_t dp = A1(byt, q) & msk;
            s += dp < cmp;
            dq += dp == cmp;
        }
    }
    int64_t ip = (i & ~(nb - 1)) + 1;
    int64_t q1 = A1(qos, ip) - 1;
    int64_t q2 = q1 + dq;
    int64_t d = (i & (nb - 1)) + 1;
    int64_t Q1 = (q1 >> bp) + 1;
    int64_t Q2 = (q2 >> bp) + 1;
    s += RCNT(d + 1, Q2, 1) - RCNT(d + 1, Q1, 1);
    int64_t msk = nb - 1;
    int64_t cmp = d - 1;
    for (int64_t q = ((Q1 - 1) << bp) + 1; q <= q1; q++) s -= (A1(byt, q) & msk) <= cmp;
    for (int64_t q = ((Q2 - 1) << bp) + 1; q <= q2; q++) s += (A1(byt, q) & msk) <= cmp;
#undef RCNT
    return s;
}

/* ---------------- SparseStepwiseDominanceCount: SparsePrefixMatrices.jl:693-821 ---------------- */
static int64_t stepwise_jump(orc_dom *D, int64_t i, int64_t j)
{
    i -= 1; j -= 1;
    int64_t c = D->sc;
    int64_t *dl = D->delta;
    const int64_t *pos = D->pos, *idx = D->idx;
    int64_t ai = D->si, aj = D->sj;
    /* reset case (:716-720) */
    if ((D->m + A1(pos, j + 1)) < A1(pos, aj + 1) - A1(pos, j + 1)) {
        aj = 0; c = 0;
        memset(dl, 0, (size_t)(D->m > 0 ? D->m : 1) * sizeof(int64_t));
    }
    for (int64_t q = A1(pos, j + 1); q <= A1(pos, aj + 1) - 1; q++) {
        A1(dl, A1(idx, q)) -= 1;
        c -= A1(idx, q) <= ai;
    }
    for (int64_t q = A1(pos, aj + 1); q <= A1(pos, j + 1) - 1; q++) {
        A1(dl, A1(idx, q)) += 1;
        c += A1(idx, q) <= ai;
    }
    for (int64_t q = i + 1; q <= ai; q++) c -= A1(dl, q);
    for (int64_t q = ai + 1; q <= i; q++) c += A1(dl, q);
    D->si = i; D->sj = j; D->sc = c;
    return c;
}

orc_dom *orc_dom_build(int32_t hint, int64_t m, int64_t n, int64_t N,
                       const int64_t *pos, const int64_t *idx, int64_t b, int64_t H, int64_t bp)
{
    orc_dom *D = (orc_dom *)calloc(1, sizeof(orc_dom));
    D->hint = hint; D->m = m; D->n = n; D->N = N;
    /* pos carries n+1 entries (the reference also passes "n+1 columns" with an (n+1)-entry
     * pos for NetCount, SparseColorArrays.jl:116; only pos[j], j <= n+1 is ever read) */
    D->pos = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
    memcpy(D->pos, pos, (size_t)(n + 1) * sizeof(int64_t));
    int64_t *ic = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    if (N > 0) memcpy(ic, idx, (size_t)N * sizeof(int64_t));
    D->idx = ic;
    if (hint == CP_HINT_STEP) {
        D->si = D->sj = 0; D->sc = 0;
        D->delta = (int64_t *)calloc((size_t)(m > 0 ? m : 1), sizeof(int64_t));
    } else if (hint == CP_HINT_SPARSE) {
        radix_build(D, ic, b, H, bp);
    } else {
        binary_build(D, ic);
    }
    return D;
}

int64_t orc_dom_query(orc_dom *D, int64_t i, int64_t j)
{
    if (D->hint == CP_HINT_STEP) return stepwise_jump(D, i, j);
    if (D->hint == CP_HINT_SPARSE) return radix_query(D, i, j);
    return binary_query(D, i, j);
}

int64_t orc_dom_step(orc_dom *D, int32_t mi, int64_t i, int32_t mj, int64_t j)
{
    /* only the stepwise structure specialises Step; the rest de-step (Costs.jl:195) */
    if (D->hint != CP_HINT_STEP) return orc_dom_query(D, i, j);
    int64_t c = D->sc;
    int64_t *dl = D->delta;
    const int64_t *pos = D->pos, *idx = D->idx;
    if (mi == 0 && mj == 0) return c;                       /* :742-747 */
    if (mi == 0 && mj == 1) {                               /* Same i, Next j :749-768 */
        i -= 1; j -= 1;
        for (int64_t q = A1(pos, j); q <= A1(pos, j + 1) - 1; q++) {
            A1(dl, A1(idx, q)) += 1;
            c += A1(idx, q) <= i;
        }
        D->sj = j; D->sc = c;
        return c;
    }
    if (mi == 0 && mj == 2) {                               /* Same i, Prev j :770-789 */
        i -= 1; j -= 1;
        for (int64_t q = A1(pos, j + 1); q <= A1(pos, j + 2) - 1; q++) {
            A1(dl, A1(idx, q)) -= 1;
            c -= A1(idx, q) <= i;
        }
        D->sj = j; D->sc = c;
        return c;
    }
    if (mi == 1 && mj == 0) {                               /* Next i, Same j :791-805 */
        i -= 1; j -= 1;
        c += A1(dl, i);
        D->si = i; D->sc = c;
        return c;
    }
    if (mi == 2 && mj == 0) {                               /* Prev i, Same j :807-821 */
        i -= 1; j -= 1;
        c -= A1(dl, i + 1);
        D->si = i; D->sc = c;
        return c;
    }
    return stepwise_jump(D, i, j);                          /* Jump / mixed: plain call */
}

void orc_dom_free(orc_dom *D)
{
    if (!D) return;
    free(D->pos); free(D->qos); free(D->byt); free(D->cnt);
    free(D->rbyt); free(D->rcnt); free(D->idx); free(D->delta);
    free(D);
}

/* ---------------- NetCount / SelfNetCount ---------------- */
struct orc_net {
    int32_t self;
    int64_t n;
    int64_t *pos;       /* NetCount keeps A's pos (n+1) */
    orc_dom *lnk;
};

void orc_net_link_array(int64_t m, int64_t n, int64_t N, const int64_t *pos,
                        const int64_t *idx, int64_t *out)
{
    (void)N;
    int64_t *hst = (int64_t *)calloc((size_t)(m > 0 ? m : 1), sizeof(int64_t));
    for (int64_t j = 1; j <= n; j++)
        for (int64_t q = A1(pos, j); q <= A1(pos, j + 1) - 1; q++) {
            int64_t i = A1(idx, q);
            A1(out, q) = (n + 1) - A1(hst, i);
            A1(hst, i) = j;
        }
    free(hst);
}

/* SparseColorArrays.jl:101-118 */
orc_net *orc_netcount_build(int32_t hint, int64_t m, int64_t n, int64_t N,
                            const int64_t *pos, const int64_t *idx)
{
    orc_net *C = (orc_net *)calloc(1, sizeof(orc_net));
    C->self = 0; C->n = n;
    C->pos = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
    memcpy(C->pos, pos, (size_t)(n + 1) * sizeof(int64_t));
    int64_t *idx2 = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    orc_net_link_array(m, n, N, pos, idx, idx2);
    C->lnk = orc_dom_build(hint, n + 1, n, N, pos, idx2, 0, 0, 0);   /* m := n+1 rows */
    free(idx2);
    return C;
}

/* SparseColorArrays.jl:177-222 */
orc_net *orc_selfnetcount_build(int32_t hint, int64_t m, int64_t n, int64_t N,
                                const int64_t *pos, const int64_t *idx)
{
    (void)N;
    orc_net *C = (orc_net *)calloc(1, sizeof(orc_net));
    C->self = 1; C->n = n;
    int64_t *hst = (int64_t *)calloc((size_t)(m > 0 ? m : 1), sizeof(int64_t));
    int64_t *hst2 = (int64_t *)calloc((size_t)(m > 0 ? m : 1), sizeof(int64_t));
    int64_t *pos2 = (int64_t *)calloc((size_t)(n + 1), sizeof(int64_t));
    for (int64_t j = 1; j <= n; j++)
        for (int64_t q = A1(pos, j); q <= A1(pos, j + 1) - 1; q++) {
            int64_t i = A1(idx, q);
            if (A1(hst, i) == 0) A1(hst, i) = j;
            A1(hst2, i) = j;
        }
    for (int64_t i = 1; i <= m; i++)
        if (A1(hst, i) != 0) A1(pos2, A1(hst2, i) + 1) += 1;
    int64_t q = 1;
    for (int64_t j = 1; j <= n + 1; j++) {
        int64_t t = A1(pos2, j);
        A1(pos2, j) = q;
        q += t;
    }
    int64_t N2 = q - 1;
    int64_t *idx2 = (int64_t *)malloc((size_t)(N2 > 0 ? N2 : 1) * sizeof(int64_t));
    for (int64_t i = 1; i <= m; i++)
        if (A1(hst, i) != 0) {
            int64_t j = A1(hst, i), jp = A1(hst2, i);
            int64_t qq = A1(pos2, jp + 1);
            A1(idx2, qq) = (n + 1) - j;
            A1(pos2, jp + 1) = qq + 1;
        }
    /* after the fill pos2[j'+1] has advanced to the start of column j'+1, i.e. pos2 is
     * again a valid (n+1)-entry column pointer whose entry k is the start of column k. */
    /* NB: the reference hands this advanced pos' straight to dominancecount! (:220). */
    C->pos = NULL;
    C->lnk = orc_dom_build(hint, n + 1, n, N2, pos2, idx2, 0, 0, 0);
    /* n+1 "columns" but only n+1 pos entries: queries use j' <= n+1 only. */
    free(hst); free(hst2); free(pos2); free(idx2);
    return C;
}

int64_t orc_net_query(orc_net *C, int64_t j, int64_t jp)
{
    if (C->self) return orc_dom_query(C->lnk, (C->n + 2) - j, jp);          /* :225-229 */
    return (A1(C->pos, jp) - A1(C->pos, j)) - orc_dom_query(C->lnk, (C->n + 2) - j, jp); /* :121-125 */
}

/* Step(net)(_j, _j') SparseColorArrays.jl:127-152, 231-256.
 * moves: 0 Same, 1 Next, 2 Prev, 3 Jump.  Next(j) on the net maps to Prev(n+2-j) on lnk. */
int64_t orc_net_step(orc_net *C, int32_t mj, int64_t j, int32_t mjp, int64_t jp)
{
    int64_t base = C->self ? 0 : (A1(C->pos, jp) - A1(C->pos, j));
    int64_t r;
    if (mj == 0) r = orc_dom_step(C->lnk, 0, (C->n + 2) - j, mjp, jp);
    else if (mj == 1 && mjp == 0) r = orc_dom_step(C->lnk, 2, (C->n + 2) - j, 0, jp);
    else if (mj == 2 && mjp == 0) r = orc_dom_step(C->lnk, 1, (C->n + 2) - j, 0, jp);
    else r = orc_dom_query(C->lnk, (C->n + 2) - j, jp);
    return C->self ? r : base - r;
}

void orc_net_free(orc_net *C)
{
    if (!C) return;
    free(C->pos);
    orc_dom_free(C->lnk);
    free(C);
}

/* ---------------- partwise: PartwiseCounts.jl:1-60 ---------------- */
int64_t orc_partwise(int64_t m, int64_t n, int64_t N, const int64_t *pos, const int64_t *idx,
                     int64_t K, const int64_t *asg,
                     int64_t *pios_out, int64_t *prm_out, int64_t *pos_out, int64_t *idx_out)
{
    (void)m;
    int64_t *Pos = (int64_t *)calloc((size_t)(K + 1), sizeof(int64_t));   /* Πos */
    int64_t *pios = pios_out;                                              /* πos */
    memset(pios, 0, (size_t)(K + 1) * sizeof(int64_t));
    int64_t *hst = (int64_t *)calloc((size_t)(K > 0 ? K : 1), sizeof(int64_t));
    for (int64_t j = 1; j <= n; j++)
        for (int64_t q = A1(pos, j); q <= A1(pos, j + 1) - 1; q++) {
            int64_t i = A1(idx, q);
            int64_t k = A1(asg, i);
            A1(pios, k + 1) += A1(hst, k) != j;
            A1(hst, k) = j;
            A1(Pos, k + 1) += 1;
        }
    int64_t q = 1, jp = 1;
    for (int64_t k = 1; k <= K + 1; k++) {
        int64_t t = A1(Pos, k); A1(Pos, k) = q; q += t;
        int64_t u = A1(pios, k); A1(pios, k) = jp; jp += u;
    }
    int64_t np = jp - 1;
    memset(hst, 0, (size_t)(K > 0 ? K : 1) * sizeof(int64_t));
    for (int64_t j = 1; j <= n; j++)
        for (int64_t qq = A1(pos, j); qq <= A1(pos, j + 1) - 1; qq++) {
            int64_t i = A1(idx, qq);
            int64_t k = A1(asg, i);
            int64_t qp = A1(Pos, k + 1);
            A1(idx_out, qp) = i;
            A1(Pos, k + 1) = qp + 1;
            if (A1(hst, k) != j) {
                int64_t jj = A1(pios, k + 1);
                A1(pos_out, jj) = qp;
                A1(prm_out, jj) = j;
                A1(pios, k + 1) = jj + 1;
            }
            A1(hst, k) = j;
        }
    A1(pos_out, np + 1) = N + 1;
    /* as in the reference, πos has been advanced by one part during the fill:
     * πos[k+1] now equals the original πos[k+1] start of part k+1 -- i.e. it is again
     * the K+1-entry offset array (entry k = first column of part k). */
    free(Pos); free(hst);
    return np;
}
