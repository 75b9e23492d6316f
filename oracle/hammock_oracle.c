/*
 * hammock_oracle.c -- literal CPU restatement of the reference's greedy
 * initial-clustering path.  TEST INFRASTRUCTURE ONLY; see hammock_oracle.h
 * for who may use it and for the parity status ("parity unpinned": the
 * reference has no tests for this path and cannot be run in this image).
 *
 * Kept deliberately literal: same loop bounds, same > / >= as the cited Java
 * lines, no algorithmic shortcuts.  Paths below are relative to
 * /root/reference/src/cz/krejciadam/hammock/.
 */
#include "hammock_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* UniqueSequence.java:23-26 alphabet                                  */
/* ------------------------------------------------------------------ */
static const char HMO_AA[HMO_ALPHABET + 1] = "ARNDCQEGHILKMFPSTWYVBZX*";

int hmo_encode_residue(char c) {
    /* UniqueSequence.java:49 toUpperCase, :51 lookup */
    if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');
    for (int i = 0; i < HMO_ALPHABET; i++)
        if (HMO_AA[i] == c) return i;
    return -1;
}

char hmo_decode_residue(int r) {
    if (r < 0 || r >= HMO_ALPHABET) return '?';
    return HMO_AA[r];
}

/* ------------------------------------------------------------------ */
/* ShiftedScorer.java:48-95                                            */
/* ------------------------------------------------------------------ */
int hmo_shifted_score(const int32_t *M, const uint8_t *seq1, int len1,
                      const uint8_t *seq2, int len2, int max_shift,
                      int shift_penalty, int32_t *score, int32_t *shift) {
    const uint8_t *shorter, *longer;
    int slen, llen;
    int shorter_is_seq2;
    /* :51-57 -- ties make seq2 the shorter one */
    if (len1 >= len2) {
        shorter = seq2; slen = len2;
        longer = seq1;  llen = len1;
        shorter_is_seq2 = 1;
    } else {
        shorter = seq1; slen = len1;
        longer = seq2;  llen = len2;
        shorter_is_seq2 = 0;
    }
    /* :59-62 */
    if (max_shift >= slen) return HMO_ERR_SHIFT_TOO_BIG;

    int best = INT_MIN;   /* :64 */
    int best_shift = 0;   /* :65 */
    int diff = llen - slen; /* :66 */
    for (int s = -max_shift; s <= max_shift + diff; s++) { /* :67 */
        int actual = 0;
        if (s <= 0) { /* :69-72 */
            for (int i = 0; i < slen + s; i++)
                actual += M[shorter[i - s] * HMO_ALPHABET + longer[i]];
        } else { /* :73-77 */
            int lim = slen < (llen - s) ? slen : (llen - s);
            for (int i = 0; i < lim; i++)
                actual += M[shorter[i] * HMO_ALPHABET + longer[i + s]];
        }
        actual += diff * shift_penalty;                       /* :79 */
        if (s < 0) actual += -s * 2 * shift_penalty;          /* :80-82 */
        if (s > diff) actual += (s - diff) * 2 * shift_penalty; /* :83-85 */
        if (actual > best) { /* :86-89, first strict maximum */
            best = actual;
            best_shift = s;
        }
    }
    if (!shorter_is_seq2) best_shift = -best_shift; /* :91-93 */
    *score = best;
    if (shift) *shift = best_shift;
    return HMO_OK;
}

/* ------------------------------------------------------------------ */
/* LocalAlignmentScorer.java:27-101                                    */
/* ------------------------------------------------------------------ */
enum { DIR_LEFT = 0, DIR_UP = 1, DIR_DIAGONAL = 2, DIR_NOWHERE = 3, DIR_NULL = 4 };

int32_t hmo_local_score(const int32_t *M, const uint8_t *seq1, int len1,
                        const uint8_t *seq2, int len2, int gap_open,
                        int gap_extend) {
    int d1 = len1 + 1, d2 = len2 + 1;
    /* :88-101 initializeMatrices: zeros; column 0 UP, line 0 LEFT, [0][0] null */
    int32_t *H = (int32_t *)calloc((size_t)d1 * d2, sizeof(int32_t));
    uint8_t *D = (uint8_t *)malloc((size_t)d1 * d2);
    if (!H || !D) { free(H); free(D); return INT_MIN; }
    memset(D, DIR_NULL, (size_t)d1 * d2);
    for (int i = 1; i < d1; i++) D[i * d2 + 0] = DIR_UP;
    for (int j = 1; j < d2; j++) D[0 * d2 + j] = DIR_LEFT;

    int32_t global_max = 0; /* :32 */
    for (int line = 1; line < d1; line++) {       /* :40 */
        for (int col = 1; col < d2; col++) {      /* :41 */
            int up_pen = (D[(line - 1) * d2 + col] == DIR_UP) ? gap_extend : gap_open;   /* :43-48 */
            int left_pen = (D[line * d2 + col - 1] == DIR_LEFT) ? gap_extend : gap_open; /* :50-55 */
            int up = H[(line - 1) * d2 + col] + up_pen;       /* :57 */
            int left = H[line * d2 + col - 1] + left_pen;     /* :58 */
            int diag = H[(line - 1) * d2 + col - 1] +
                       M[seq1[line - 1] * HMO_ALPHABET + seq2[col - 1]]; /* :59 */
            int ul = up > left ? up : left;
            int mx = diag > ul ? diag : ul; /* :61 */
            if (mx < 0) { /* :63-65 */
                H[line * d2 + col] = 0;
                D[line * d2 + col] = DIR_NOWHERE;
            } else {
                H[line * d2 + col] = mx;                 /* :67 */
                if (mx > global_max) global_max = mx;    /* :68-72 */
                if (mx == left) D[line * d2 + col] = DIR_LEFT;     /* :73-75 */
                if (mx == up) D[line * d2 + col] = DIR_UP;         /* :76-78 */
                if (mx == diag) D[line * d2 + col] = DIR_DIAGONAL; /* :79-81 */
            }
        }
    }
    free(H);
    free(D);
    return global_max; /* :85 */
}

/* one SequenceScorer.sequenceScore call */
typedef struct {
    const int32_t *M;
    const uint8_t *res;
    const uint32_t *off;
    int kind, a, b;
} scorer_t;

static inline int sequence_score(const scorer_t *sc, uint32_t i1, uint32_t i2,
                                 int32_t *out) {
    const uint8_t *s1 = sc->res + sc->off[i1];
    const uint8_t *s2 = sc->res + sc->off[i2];
    int l1 = (int)(sc->off[i1 + 1] - sc->off[i1]);
    int l2 = (int)(sc->off[i2 + 1] - sc->off[i2]);
    if (sc->kind == HMO_SCORER_SHIFTED)
        return hmo_shifted_score(sc->M, s1, l1, s2, l2, sc->a, sc->b, out, NULL);
    *out = hmo_local_score(sc->M, s1, l1, s2, l2, sc->a, sc->b);
    return HMO_OK;
}

int hmo_score_pairs(const int32_t *M, const uint8_t *res, const uint32_t *off,
                    const uint32_t *i, const uint32_t *j, uint64_t n_pairs,
                    int scorer, int a, int b, int32_t *out) {
    scorer_t sc = {M, res, off, scorer, a, b};
    int status = HMO_OK;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t k = 0; k < (int64_t)n_pairs; k++) {
        int st = sequence_score(&sc, i[k], j[k], &out[k]);
        if (st != HMO_OK) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            status = st;
        }
    }
    return status;
}

/* ------------------------------------------------------------------ */
/* Cluster.java:31-74,156-158                                          */
/* ------------------------------------------------------------------ */
typedef struct {
    int32_t id;
    int32_t size;      /* Cluster.size(): sum of member counts, :156 */
    int32_t n;         /* getUniqueSize(), :113 */
    int32_t cap;
    uint32_t *members; /* insertion order, like the ArrayList */
} cluster_t;

static int cluster_insert(cluster_t *c, uint32_t seq, int32_t seq_size) {
    /* Cluster.java:50-63 (the duplicate check at :51 cannot fire for unique
     * sequences and is not modelled) */
    if (c->n == c->cap) {
        int32_t ncap = c->cap < 4 ? 4 : c->cap * 2;
        uint32_t *nm = (uint32_t *)realloc(c->members, (size_t)ncap * sizeof(uint32_t));
        if (!nm) return HMO_ERR_OOM;
        c->members = nm;
        c->cap = ncap;
    }
    c->members[c->n++] = seq;
    c->size += seq_size;
    return HMO_OK;
}

static int cluster_insert_all(cluster_t *dst, const cluster_t *src,
                              const int32_t *size) {
    /* Cluster.java:70-74 */
    for (int32_t k = 0; k < src->n; k++) {
        int st = cluster_insert(dst, src->members[k], size ? size[src->members[k]] : 1);
        if (st) return st;
    }
    return HMO_OK;
}

/* ------------------------------------------------------------------ */
/* ClinkageClusterScorer.java:30-49                                    */
/* ------------------------------------------------------------------ */
static int cluster_score(const scorer_t *sc, int threshold, const cluster_t *cl1,
                         const cluster_t *cl2, int32_t *out, uint64_t *calls) {
    int32_t result = INT_MAX; /* :34 */
    for (int32_t x = 0; x < cl1->n; x++) {       /* :36 */
        for (int32_t y = 0; y < cl2->n; y++) {   /* :37 */
            int32_t round_res;
            int st = sequence_score(sc, cl1->members[x], cl2->members[y], &round_res); /* :38 */
            (*calls)++;
            if (st) return st;
            if (round_res < result) {            /* :39 */
                result = round_res;
                if (result < threshold) {        /* :41-43 */
                    *out = INT_MIN + 1;
                    return HMO_OK;
                }
            }
        }
    }
    *out = result;
    return HMO_OK;
}

/* ------------------------------------------------------------------ */
/* NearestClusterRunner.call, ClinkageSequenceClusterer.java:258-293    */
/* ------------------------------------------------------------------ */
typedef struct {
    cluster_t *cluster; /* may be NULL */
    int32_t score;
} nearest_t;

static int nearest_runner(const scorer_t *sc, int threshold, cluster_t **part,
                          int64_t n_part, const cluster_t *compared,
                          nearest_t *out, uint64_t *calls) {
    int32_t max_score = INT_MIN; /* :260 */
    cluster_t *nearest = NULL;   /* :262 */
    for (int64_t k = 0; k < n_part; k++) { /* :263 */
        cluster_t *i = part[k];
        int32_t score;
        int st = cluster_score(sc, threshold, i, compared, &score, calls); /* :264 */
        if (st) return st;
        if (score < max_score) continue; /* :265-267 */
        if (score > max_score) {         /* :268 */
            if (i != compared) {         /* :269 */
                max_score = score;
                nearest = i;
            }
        } else {                          /* :276 score == max_score */
            if (i != compared) {          /* :277 */
                if (i->size > nearest->size) {           /* :278 */
                    nearest = i;
                } else if (i->size < nearest->size) {    /* :281 nothing */
                } else if (i->id < nearest->id) {        /* :285 */
                    nearest = i;
                }
            }
        }
    }
    out->cluster = nearest;
    out->score = max_score;
    return HMO_OK;
}

/* result kinds of findNearestClusterParallel */
enum { NEAR_NULL = 0, NEAR_DUMMY = 1, NEAR_REAL = 2 };
typedef struct {
    int kind;
    cluster_t *cluster;
    int32_t score;
} found_t;

/* ------------------------------------------------------------------ */
/* findNearestClusterParallel, ClinkageSequenceClusterer.java:137-177,  */
/* setNumberOfParts :186-192, activeClustersParts :202-223              */
/* ------------------------------------------------------------------ */
static int find_nearest(const scorer_t *sc, int threshold, cluster_t **input,
                        int64_t n_input, const cluster_t *compared,
                        int64_t sum_commodity, int n_threads, found_t *found,
                        uint64_t *calls) {
    if (n_input == 0) { /* :138-140: a NON-null dummy */
        found->kind = NEAR_DUMMY;
        found->cluster = NULL;
        found->score = INT_MIN;
        return HMO_OK;
    }
    /* :186-192 */
    int64_t n_parts = (int64_t)n_threads * 4;
    if (n_input < (int64_t)n_threads * 4 + 1) {
        n_parts = n_input - 1 > 1 ? n_input - 1 : 1;
    }
    /* :202-223: parts are consecutive runs of the input iteration order */
    int64_t for_one = sum_commodity / n_parts + 1; /* :208 */
    int64_t *bounds = (int64_t *)malloc((size_t)(n_input + 2) * sizeof(int64_t));
    if (!bounds) return HMO_ERR_OOM;
    int64_t nb = 0;
    bounds[nb++] = 0;
    int64_t portion = for_one; /* :209 */
    int64_t in_current = 0;
    for (int64_t k = 0; k < n_input; k++) { /* :210 */
        in_current++;
        portion -= input[k]->n;             /* :212 */
        if (portion <= 0) {                 /* :213-217 */
            bounds[nb++] = k + 1;
            in_current = 0;
            portion = for_one;
        }
    }
    if (in_current > 0) bounds[nb++] = n_input; /* :219-221 */
    int64_t parts = nb - 1;

    nearest_t *results = (nearest_t *)malloc((size_t)parts * sizeof(nearest_t));
    if (!results) { free(bounds); return HMO_ERR_OOM; }
    int status = HMO_OK;
    uint64_t total_calls = 0;
    /* :144-149 one NearestClusterRunner per part on the pool */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) \
    reduction(+ : total_calls) if (n_threads > 1 && parts > 1)
#endif
    for (int64_t p = 0; p < parts; p++) {
        uint64_t c = 0;
        int st = nearest_runner(sc, threshold, input + bounds[p],
                                bounds[p + 1] - bounds[p], compared, &results[p], &c);
        total_calls += c;
        if (st) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            status = st;
        }
    }
    *calls += total_calls;
    if (status) { free(bounds); free(results); return status; }

    int32_t max_score = INT_MIN + 42; /* :151 */
    nearest_t *nearest = NULL;        /* :152 */
    for (int64_t p = 0; p < parts; p++) { /* :155 (completion order; the
                                             fold is order independent) */
        nearest_t *cur = &results[p];
        if (cur->score < max_score) continue; /* :159-161 */
        if (cur->score > max_score) {         /* :163-165 */
            nearest = cur;
            max_score = cur->score;
        } else {                              /* :166 */
            if (cur->cluster->size > nearest->cluster->size) { /* :167 */
                nearest = cur;
            } else if (cur->cluster->size == nearest->cluster->size &&
                       cur->cluster->id < nearest->cluster->id) { /* :170 */
                nearest = cur;
            }
        }
    }
    if (nearest) {
        found->kind = NEAR_REAL;
        found->cluster = nearest->cluster;
        found->score = nearest->score;
    } else {
        found->kind = NEAR_NULL; /* :176 returns null */
        found->cluster = NULL;
        found->score = 0;
    }
    free(bounds);
    free(results);
    return HMO_OK;
}

/* ------------------------------------------------------------------ */
/* LimitedGreedySequenceClusterer.java:39-120                          */
/* ------------------------------------------------------------------ */
int hmo_greedy_cluster(const int32_t *M, const uint8_t *res,
                       const uint32_t *off, const int32_t *size, uint32_t n,
                       int scorer, int a, int b, int threshold,
                       int max_clusters, int n_threads, int32_t *cluster_id,
                       int32_t *result_order, int32_t *member_rank,
                       hmo_greedy_stats *stats) {
    hmo_greedy_stats local;
    if (!stats) stats = &local;
    memset(stats, 0, sizeof(*stats));
    if (!M || !res || !off || !cluster_id) return HMO_ERR_BAD_ARG;
    if (n_threads < 1) n_threads = 1;
    scorer_t sc = {M, res, off, scorer, a, b};
    int status = HMO_OK;

    cluster_t *pool = (cluster_t *)calloc(n ? n : 1, sizeof(cluster_t));
    cluster_t **initial = (cluster_t **)malloc((size_t)(n ? n : 1) * sizeof(cluster_t *));
    cluster_t **clusters = (cluster_t **)malloc((size_t)(n ? n : 1) * sizeof(cluster_t *));
    cluster_t **orphans = (cluster_t **)malloc((size_t)(n ? n : 1) * sizeof(cluster_t *));
    cluster_t **remaining = (cluster_t **)malloc((size_t)(n ? n : 1) * sizeof(cluster_t *));
    if (!pool || !initial || !clusters || !orphans || !remaining) {
        status = HMO_ERR_OOM;
        goto done;
    }
    /* firstPhase :78-83: one singleton cluster per sequence, id = index */
    int64_t ni = n, nc = 0, no = 0, nr = 0;
    for (uint32_t i = 0; i < n; i++) {
        pool[i].id = (int32_t)i;
        status = cluster_insert(&pool[i], i, size ? size[i] : 1);
        if (status) goto done;
        initial[i] = &pool[i];
    }

    uint64_t calls = 0;
    int64_t sum_commodity_clusters = 0; /* :88, never updated */
    int64_t index = 0;                  /* :89 */
    while (index < ni && nc < max_clusters) { /* :90 */
        cluster_t *x = initial[index];        /* :91 */
        found_t A, B;
        status = find_nearest(&sc, threshold, clusters, nc, x,
                              sum_commodity_clusters, n_threads, &A, &calls); /* :92 */
        if (status) goto done;
        status = find_nearest(&sc, threshold, initial + index + 1, ni - index - 1, x,
                              ni - index - 1, n_threads, &B, &calls);         /* :93 */
        if (status) goto done;
        int absorb = 0; /* x absorbs B.cluster and becomes a cluster */
        if (A.kind != NEAR_NULL) {             /* :94 */
            if (B.kind != NEAR_NULL) {         /* :95 */
                if (A.score >= B.score) {      /* :96 */
                    if (!A.cluster) {          /* :97 NPE: both dummies */
                        stats->crash_case = 2;
                        stats->crash_index = (int32_t)index;
                        status = HMO_ERR_REFERENCE_WOULD_CRASH;
                        goto done;
                    }
                    status = cluster_insert_all(A.cluster, x, size);
                    if (status) goto done;
                } else {
                    absorb = 1;                /* :99-101 */
                }
            } else {
                if (!A.cluster) {              /* :104 NPE: A dummy, B null */
                    stats->crash_case = 1;
                    stats->crash_index = (int32_t)index;
                    status = HMO_ERR_REFERENCE_WOULD_CRASH;
                    goto done;
                }
                status = cluster_insert_all(A.cluster, x, size); /* :104 */
                if (status) goto done;
            }
        } else {
            if (B.kind != NEAR_NULL) {         /* :107 */
                if (!B.cluster) {              /* :108 NPE: A null, B dummy */
                    stats->crash_case = 3;
                    stats->crash_index = (int32_t)index;
                    status = HMO_ERR_REFERENCE_WOULD_CRASH;
                    goto done;
                }
                absorb = 1;                    /* :108-110 */
            } else {
                orphans[no++] = x;             /* :112 */
            }
        }
        if (absorb) {
            status = cluster_insert_all(x, B.cluster, size);
            if (status) goto done;
            clusters[nc++] = x;
            /* initialList.remove(Object): first element equal by id */
            int64_t pos = -1;
            for (int64_t k = 0; k < ni; k++) {
                if (initial[k]->id == B.cluster->id) { pos = k; break; }
            }
            memmove(initial + pos, initial + pos + 1,
                    (size_t)(ni - pos - 1) * sizeof(cluster_t *));
            ni--;
        }
        index++; /* :115 */
    }
    stats->score_calls_phase1 = calls;
    stats->phase1_stop_index = (int32_t)index;
    stats->phase1_clusters = (int32_t)nc;
    stats->phase1_orphans = (int32_t)no;

    /* :117-119 returned list = clusters + orphans + initial[index:].
     * cluster() :43-51 splits it at the first cluster of unique size 1;
     * every element of `clusters` has >= 2 members, so the split point is nc. */
    {
        int64_t n_seq = no + (ni - index);
        cluster_t **actual_sequences = (cluster_t **)malloc((size_t)(n_seq ? n_seq : 1) * sizeof(cluster_t *));
        if (!actual_sequences) { status = HMO_ERR_OOM; goto done; }
        memcpy(actual_sequences, orphans, (size_t)no * sizeof(cluster_t *));
        memcpy(actual_sequences + no, initial + index, (size_t)(ni - index) * sizeof(cluster_t *));

        int64_t sum_commodity = 0; /* :55-58, frozen */
        for (int64_t k = 0; k < nc; k++) sum_commodity += clusters[k]->n;
        calls = 0;
        for (int64_t k = 0; k < n_seq; k++) { /* :59 */
            cluster_t *cl = actual_sequences[k];
            found_t F;
            status = find_nearest(&sc, threshold, clusters, nc, cl, sum_commodity,
                                  n_threads, &F, &calls); /* :60 */
            if (status) { free(actual_sequences); goto done; }
            if (F.kind != NEAR_NULL && F.score >= threshold) { /* :61 (dummy: MIN_VALUE) */
                status = cluster_insert_all(F.cluster, cl, size); /* :62 */
                if (status) { free(actual_sequences); goto done; }
            } else {
                remaining[nr++] = cl; /* :64 */
            }
        }
        free(actual_sequences);
        stats->score_calls_phase2 = calls;
    }

    /* :67-68 */
    {
        int64_t k_out = 0;
        for (int64_t k = 0; k < nc; k++) {
            for (int32_t m = 0; m < clusters[k]->n; m++) {
                cluster_id[clusters[k]->members[m]] = clusters[k]->id;
                /* Cluster.getSequences() order = insertion order (Cluster.java:50-74) */
                if (member_rank) member_rank[clusters[k]->members[m]] = m;
            }
            if (result_order) result_order[k_out] = clusters[k]->id;
            k_out++;
        }
        for (int64_t k = 0; k < nr; k++) {
            for (int32_t m = 0; m < remaining[k]->n; m++) {
                cluster_id[remaining[k]->members[m]] = remaining[k]->id;
                if (member_rank) member_rank[remaining[k]->members[m]] = m;
            }
            if (result_order) result_order[k_out] = remaining[k]->id;
            k_out++;
        }
        stats->n_result_clusters = (int32_t)k_out;
        stats->n_multi = (int32_t)nc;
    }

done:
    if (pool)
        for (uint32_t i = 0; i < n; i++) free(pool[i].members);
    free(pool);
    free(initial);
    free(clusters);
    free(orphans);
    free(remaining);
    return status;
}

/* ------------------------------------------------------------------ */
/* java.util.HashSet<Cluster> iteration order (Java 8+), for the clinkage */
/* clusterer: Cluster.hashCode() = 79 * 7 + id (Cluster.java:178-183),   */
/* HashMap.hash = h ^ (h >>> 16), power-of-two table from 16, load 0.75, */
/* chains in insertion order (tail append, order-preserving resize), no  */
/* shrink on removal.                                                    */
/* ------------------------------------------------------------------ */
/* hmo_set_java_hashset(7) / (6): the same set as Java 7 and earlier kept it (the reference is a Java 1.7 project,          */
/* nbproject/project.properties:45-46): hash(h) = h ^ (h>>>20) ^ (h>>>12), then h ^ (h>>>7) ^ (h>>>4); a new entry goes  */
/* to the HEAD of its chain; transfer() walks the old buckets in order and puts every entry at the head of its new        */
/* bucket.  7 = JDK 7u6+ (resize BEFORE the insert, when size >= threshold and the target bucket is not empty);           */
/* 6 = JDK 6 and JDK 7 before 7u6 (insert, then resize when size++ >= threshold).                                         */
static int g_java_hashset = 8;
void hmo_set_java_hashset(int version) { g_java_hashset = (version == 7 || version == 6) ? version : 8; }
int hmo_get_java_hashset(void) { return g_java_hashset; }

typedef struct {
    int32_t *head, *tail; /* per bucket: first / last id of the chain, -1 = empty */
    int32_t *next;        /* per id */
    uint32_t cap, size, max_id, lowest;
} jset_t;

static uint32_t jset_bucket(const jset_t *s, int32_t id) {
    uint32_t h = (uint32_t)(553 + id);
    if (g_java_hashset == 8) h ^= h >> 16;
    else { h ^= (h >> 20) ^ (h >> 12); h ^= (h >> 7) ^ (h >> 4); }
    return h & (s->cap - 1);
}

static int jset_init(jset_t *s, uint32_t max_id) {
    s->cap = 16; s->size = 0; s->max_id = max_id; s->lowest = 0;
    s->head = (int32_t *)malloc(16 * sizeof(int32_t));
    s->tail = (int32_t *)malloc(16 * sizeof(int32_t));
    s->next = (int32_t *)malloc(((size_t)max_id + 1) * sizeof(int32_t));
    if (!s->head || !s->tail || !s->next) return HMO_ERR_OOM;
    for (uint32_t b = 0; b < 16; b++) s->head[b] = s->tail[b] = -1;
    return HMO_OK;
}

static void jset_free(jset_t *s) { free(s->head); free(s->tail); free(s->next); }

static void jset_append(jset_t *s, int32_t id) {
    uint32_t b = jset_bucket(s, id);
    s->next[id] = -1;
    if (s->head[b] < 0) s->head[b] = id; else s->next[s->tail[b]] = id;
    s->tail[b] = id;
    if (b < s->lowest) s->lowest = b;
}

static void jset_push_head(jset_t *s, int32_t id) { /* Java <= 7: table[i] = new Entry(.., table[i]) */
    uint32_t b = jset_bucket(s, id);
    s->next[id] = s->head[b];
    if (s->head[b] < 0) s->tail[b] = id;
    s->head[b] = id;
    if (b < s->lowest) s->lowest = b;
}

static int jset_resize_old(jset_t *s) { /* Java <= 7 transfer(): old buckets in order, every entry to the head of its new chain */
    uint32_t ocap = s->cap;
    int32_t *order = (int32_t *)malloc(((size_t)s->size + 1) * sizeof(int32_t));
    if (!order) return HMO_ERR_OOM;
    uint32_t k = 0;
    for (uint32_t b = 0; b < ocap; b++)
        for (int32_t id2 = s->head[b]; id2 >= 0; id2 = s->next[id2]) order[k++] = id2;
    s->cap = ocap * 2;
    free(s->head); free(s->tail);
    s->head = (int32_t *)malloc((size_t)s->cap * sizeof(int32_t));
    s->tail = (int32_t *)malloc((size_t)s->cap * sizeof(int32_t));
    if (!s->head || !s->tail) { free(order); return HMO_ERR_OOM; }
    for (uint32_t b = 0; b < s->cap; b++) s->head[b] = s->tail[b] = -1;
    s->lowest = s->cap;
    for (uint32_t q = 0; q < k; q++) jset_push_head(s, order[q]);
    free(order);
    return HMO_OK;
}

static int jset_add(jset_t *s, int32_t id) {
    if (g_java_hashset != 8) {
        const uint32_t threshold = s->cap / 4 * 3;
        int st = HMO_OK;
        if (g_java_hashset == 7) {
            if (s->size >= threshold && s->head[jset_bucket(s, id)] >= 0) st = jset_resize_old(s);
            if (st) return st;
            jset_push_head(s, id);
            s->size++;
        } else {
            jset_push_head(s, id);
            if (s->size++ >= threshold) st = jset_resize_old(s);
        }
        return st;
    }
    jset_append(s, id);
    if (++s->size > s->cap / 4 * 3) { /* resize(): every chain is re-appended in order */
        uint32_t ocap = s->cap;
        int32_t *ohead = s->head;
        int32_t *order = (int32_t *)malloc((size_t)s->size * sizeof(int32_t));
        if (!order) return HMO_ERR_OOM;
        uint32_t k = 0;
        for (uint32_t b = 0; b < ocap; b++)
            for (int32_t id2 = ohead[b]; id2 >= 0; id2 = s->next[id2]) order[k++] = id2;
        s->cap = ocap * 2;
        free(s->head); free(s->tail);
        s->head = (int32_t *)malloc((size_t)s->cap * sizeof(int32_t));
        s->tail = (int32_t *)malloc((size_t)s->cap * sizeof(int32_t));
        if (!s->head || !s->tail) { free(order); return HMO_ERR_OOM; }
        for (uint32_t b = 0; b < s->cap; b++) s->head[b] = s->tail[b] = -1;
        s->lowest = s->cap;
        for (uint32_t q = 0; q < k; q++) jset_append(s, order[q]);
        free(order);
    }
    return HMO_OK;
}

static void jset_remove(jset_t *s, int32_t id) {
    uint32_t b = jset_bucket(s, id);
    int32_t prev = -1;
    for (int32_t cur = s->head[b]; cur >= 0; prev = cur, cur = s->next[cur]) {
        if (cur != id) continue;
        if (prev < 0) s->head[b] = s->next[cur]; else s->next[prev] = s->next[cur];
        if (s->tail[b] == cur) s->tail[b] = prev;
        s->size--;
        return;
    }
}

static int32_t jset_first(jset_t *s) { /* iterator().next(); -1 if empty */
    while (s->lowest < s->cap && s->head[s->lowest] < 0) s->lowest++;
    return s->lowest < s->cap ? s->head[s->lowest] : -1;
}

/* ------------------------------------------------------------------ */
/* ClinkageSequenceClusterer.cluster, ClinkageSequenceClusterer.java:43-124 */
/* ------------------------------------------------------------------ */
/* The CachedClusterScorer / DynamicMatrix pair (CachedClusterScorer.java:23-280) is NOT restated here: the literal
 * Python restatement (oracle/hammock_oracle.py) shows by fuzzing over thread counts and with the cache bypassed that
 * at one pool thread it is a transparent memo of clusterScore.  This C form keeps its own memo of cluster scores
 * (slots re-used on merge, rows merged by element-wise min as :95-106 does) and is checked against the Python one. */
int hmo_clinkage_cluster(const int32_t *M, const uint8_t *res, const uint32_t *off, const int32_t *size, uint32_t n,
                         int max_shift, int shift_penalty, int threshold, int n_threads, int32_t *cluster_id,
                         int32_t *result_order, int32_t *member_rank, hmo_clinkage_stats *stats) {
    hmo_clinkage_stats local;
    if (!stats) stats = &local;
    memset(stats, 0, sizeof(*stats));
    if (!M || !res || !off || !cluster_id) return HMO_ERR_BAD_ARG;
    if (n == 0) return HMO_ERR_REFERENCE_WOULD_CRASH; /* :118 NoSuchElementException */
    if (n_threads < 1) n_threads = 1;
    scorer_t sc = {M, res, off, HMO_SCORER_SHIFTED, max_shift, shift_penalty};
    const uint32_t max_id = 2 * n + 2;
    int status = HMO_OK;
    /* cluster id -> slot (row of the memo), members, sizes */
    cluster_t *cl = (cluster_t *)calloc((size_t)max_id + 1, sizeof(cluster_t));
    int32_t *slot_of = (int32_t *)malloc(((size_t)max_id + 1) * sizeof(int32_t));
    char *alive = (char *)calloc((size_t)max_id + 1, 1);
    int32_t *stack = (int32_t *)malloc(((size_t)n + 2) * sizeof(int32_t));
    char *on_stack = (char *)calloc((size_t)max_id + 1, 1);
    /* memo[slot a][slot b], a > b: INT_MAX = unknown */
    int32_t **memo = (int32_t **)calloc(n, sizeof(int32_t *));
    int32_t *active_ids = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int32_t *scores = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    jset_t active, ready;
    memset(&active, 0, sizeof(active));
    memset(&ready, 0, sizeof(ready));
    if (!cl || !slot_of || !alive || !stack || !on_stack || !memo || !active_ids || !scores) { status = HMO_ERR_OOM; goto done; }
    if ((status = jset_init(&active, max_id)) || (status = jset_init(&ready, max_id))) goto done;
    for (uint32_t a = 0; a < n; a++) {
        memo[a] = (int32_t *)malloc(((size_t)a + 1) * sizeof(int32_t));
        if (!memo[a]) { status = HMO_ERR_OOM; goto done; }
        for (uint32_t b = 0; b <= a; b++) memo[a][b] = INT_MAX;
    }
    int32_t current_id = 1;
    for (uint32_t k = 0; k < n; k++) { /* :50-55 */
        cl[current_id].id = current_id;
        if ((status = cluster_insert(&cl[current_id], k, size ? size[k] : 1))) goto done;
        slot_of[current_id] = (int32_t)k;
        alive[current_id] = 1;
        if ((status = jset_add(&active, current_id))) goto done;
        current_id++;
    }
    uint64_t calls = 0;
    int64_t sp = 0;
    while (active.size > 1) { /* :63 */
        stack[sp++] = jset_first(&active); /* :70-71 */
        on_stack[stack[sp - 1]] = 1;
        while (sp > 0) {                   /* :72 */
            const int32_t top = stack[sp - 1];
            /* findNearestClusterParallel(activeClusters, top, ..) = arg-max over the OTHER active clusters of
             * (clusterScore, size, -id) among scores >= MIN_VALUE + 42 (:151-176,258-293); null if none */
            uint32_t na = 0;
            for (uint32_t b = 0; b < active.cap; b++)
                for (int32_t id = active.head[b]; id >= 0; id = active.next[id]) active_ids[na++] = id;
            int st_any = HMO_OK;
            uint64_t c_any = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads) reduction(+ : c_any) if (n_threads > 1 && na > 256)
#endif
            for (int64_t q = 0; q < (int64_t)na; q++) {
                const int32_t other = active_ids[q];
                if (other == top) { scores[q] = INT_MIN; continue; }
                const int32_t sa = slot_of[other], sb = slot_of[top];
                int32_t *cell = sa > sb ? &memo[sa][sb] : &memo[sb][sa];
                if (*cell == INT_MAX) {
                    uint64_t c = 0;
                    int32_t v;
                    int st = cluster_score(&sc, threshold, &cl[other], &cl[top], &v, &c); /* clusterScore(i, comparedCluster), :263 */
                    c_any += c;
                    if (st) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                        st_any = st;
                    }
                    *cell = v;
                }
                scores[q] = *cell;
            }
            calls += c_any;
            stats->searches++;
            if (st_any) { status = st_any; goto done; }
            int32_t nearest = -1, max_score = INT_MIN;
            for (uint32_t q = 0; q < na; q++) {
                const int32_t other = active_ids[q], s = scores[q];
                if (other == top || s < INT_MIN + 42) continue;
                if (nearest < 0 || s > max_score ||
                    (s == max_score && (cl[other].size > cl[nearest].size ||
                                        (cl[other].size == cl[nearest].size && other < nearest)))) {
                    nearest = other;
                    max_score = s;
                }
            }
            if (nearest < 0 || max_score < threshold) { /* :86-92 */
                sp--;
                on_stack[top] = 0;
                if ((status = jset_add(&ready, top))) goto done;
                jset_remove(&active, top);
                continue;
            }
            if (sp > 1 && stack[sp - 2] == nearest) { /* :96 */
                current_id++;
                sp -= 2;
                on_stack[top] = on_stack[nearest] = 0;
                jset_remove(&active, top);
                jset_remove(&active, nearest);
                /* join (:102): the merged cluster's scores are the element-wise min of the two rows where both are
                 * known; it takes over top's slot, the nearest's slot retires */
                const int32_t st_ = slot_of[top], sn = slot_of[nearest];
                for (uint32_t x = 0; x < n; x++) {
                    if ((int32_t)x == st_ || (int32_t)x == sn) continue;
                    int32_t *a = (int32_t)x > st_ ? &memo[x][st_] : &memo[st_][x];
                    const int32_t b = (int32_t)x > sn ? memo[x][sn] : memo[sn][x];
                    *a = (*a == INT_MAX || b == INT_MAX) ? INT_MAX : (*a < b ? *a : b);
                }
                cluster_t *nc = &cl[current_id];
                nc->id = current_id;
                if ((status = cluster_insert_all(nc, &cl[top], size))) goto done;     /* :105 top's members first */
                if ((status = cluster_insert_all(nc, &cl[nearest], size))) goto done; /* :106 */
                slot_of[current_id] = st_;
                alive[top] = alive[nearest] = 0;
                alive[current_id] = 1;
                if ((status = jset_add(&active, current_id))) goto done;
                stats->merges++;
            } else {
                /* A tie (score, size, smaller id) can send the chain back to a cluster that is still on the stack below
                 * stack[-2].  The reference pushes it again (:113), merges or retires the upper copy and later takes the
                 * stale Cluster object below as `top`: the run then ends in NoSuchElementException (:118) or returns a list
                 * in which a sequence belongs to two clusters (oracle/hammock_oracle.py does exactly that).  Neither fits
                 * cluster_id[n]: flagged like the other inputs on which the reference does not produce a clustering. */
                if (on_stack[nearest]) { status = HMO_ERR_REFERENCE_WOULD_CRASH; goto done; }
                stack[sp++] = nearest; /* :113 */
                on_stack[nearest] = 1;
            }
        }
    }
    if ((status = jset_add(&ready, jset_first(&active)))) goto done; /* :118 */
    {
        int32_t k_out = 0;
        for (uint32_t b = 0; b < ready.cap; b++)
            for (int32_t id = ready.head[b]; id >= 0; id = ready.next[id]) { /* :121-123 HashSet order */
                for (int32_t m = 0; m < cl[id].n; m++) {
                    cluster_id[cl[id].members[m]] = id;
                    if (member_rank) member_rank[cl[id].members[m]] = m;
                }
                if (result_order) result_order[k_out] = id;
                k_out++;
            }
        stats->n_result_clusters = k_out;
    }
    stats->score_calls = calls;
done:
    if (cl)
        for (uint32_t i = 0; i <= max_id; i++) free(cl[i].members);
    free(cl); free(slot_of); free(alive); free(stack); free(on_stack); free(active_ids); free(scores);
    if (memo) { for (uint32_t a = 0; a < n; a++) free(memo[a]); free(memo); }
    if (active.next) jset_free(&active);
    if (ready.next) jset_free(&ready);
    return status;
}

/* ------------------------------------------------------------------ */
/* UniqueSequence.sortSequences, UniqueSequence.java:176-203,238-261    */
/* ------------------------------------------------------------------ */
typedef struct {
    const uint8_t *res;
    const uint32_t *off;
    const int32_t *size;
} sort_ctx_t;

/* String.compareTo on the decoded letters (UniqueSequence.java:103-109) */
static int alpha_compare(const sort_ctx_t *c, uint32_t x, uint32_t y) {
    uint32_t lx = c->off[x + 1] - c->off[x], ly = c->off[y + 1] - c->off[y];
    uint32_t lim = lx < ly ? lx : ly;
    for (uint32_t k = 0; k < lim; k++) {
        int cx = (unsigned char)HMO_AA[c->res[c->off[x] + k]];
        int cy = (unsigned char)HMO_AA[c->res[c->off[y] + k]];
        if (cx != cy) return cx - cy;
    }
    return (int)lx - (int)ly;
}

/* comparator being reversed: 0 = SizeAlphabetic (:238-248), 1 = Alphabetic */
static int base_compare(const sort_ctx_t *c, int order, uint32_t x, uint32_t y) {
    if (order == 0) {
        int sx = c->size ? c->size[x] : 1, sy = c->size ? c->size[y] : 1;
        int r = sx - sy; /* SizeComparator.java:15-20 */
        if (r != 0) return r;
    }
    return alpha_compare(c, x, y);
}

static void merge_sort(const sort_ctx_t *c, int order, uint32_t *a, uint32_t *tmp,
                       int64_t lo, int64_t hi) {
    if (hi - lo < 2) return;
    int64_t mid = lo + (hi - lo) / 2;
    merge_sort(c, order, a, tmp, lo, mid);
    merge_sort(c, order, a, tmp, mid, hi);
    int64_t i = lo, j = mid, k = lo;
    while (i < mid && j < hi) {
        /* Collections.reverseOrder(cmp): reversed(x,y) = cmp(y,x); stable */
        if (base_compare(c, order, a[j], a[i]) > 0) tmp[k++] = a[j++];
        else tmp[k++] = a[i++];
    }
    while (i < mid) tmp[k++] = a[i++];
    while (j < hi) tmp[k++] = a[j++];
    memcpy(a + lo, tmp + lo, (size_t)(hi - lo) * sizeof(uint32_t));
}

int hmo_sort_order(const uint8_t *res, const uint32_t *off, const int32_t *size,
                   uint32_t n, int order, uint32_t *perm) {
    if (order < 0 || order > 2) return HMO_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n; i++) perm[i] = i;
    if (order == 2) return HMO_OK; /* "input", :190 */
    uint32_t *tmp = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    if (!tmp) return HMO_ERR_OOM;
    sort_ctx_t c = {res, off, size};
    merge_sort(&c, order, perm, tmp, 0, n);
    free(tmp);
    return HMO_OK;
}

/* ------------------------------------------------------------------ */
/* synthetic inputs, SURVEY.md 8(d) / BASELINE.md section 4             */
/* ------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t *state) {
    uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static uint64_t pep_hash(const uint8_t *p, int len) {
    uint64_t h = 1469598103934665603ULL ^ (uint64_t)len;
    for (int i = 0; i < len; i++) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}

int hmo_synth(uint64_t seed, uint32_t n, int len_lo, int len_hi, uint8_t *res,
              uint32_t *off) {
    if (len_lo < 1 || len_hi < len_lo || len_hi > 255) return HMO_ERR_BAD_ARG;
    if (len_hi <= 8) {  /* more distinct peptides requested than exist: the loop below would never end */
        uint64_t avail = 0, pw = 1;
        for (int l = 1; l <= len_hi; l++) { pw *= 20; if (l >= len_lo) avail += pw; }
        if ((uint64_t)n > avail) return HMO_ERR_BAD_ARG;
    }
    uint64_t cap = 16;
    while (cap < (uint64_t)n * 2 + 2) cap <<= 1;
    int64_t *table = (int64_t *)malloc(cap * sizeof(int64_t));
    if (!table) return HMO_ERR_OOM;
    for (uint64_t i = 0; i < cap; i++) table[i] = -1;
    uint64_t state = seed;
    uint32_t count = 0;
    uint8_t buf[256];
    off[0] = 0;
    while (count < n) {
        int len = len_lo;
        if (len_hi > len_lo)
            len = len_lo + (int)((splitmix64(&state) >> 33) % (uint64_t)(len_hi - len_lo + 1));
        for (int k = 0; k < len; k++) buf[k] = (uint8_t)((splitmix64(&state) >> 33) % 20);
        uint64_t slot = pep_hash(buf, len) & (cap - 1);
        int dup = 0;
        while (table[slot] >= 0) {
            uint32_t o = (uint32_t)table[slot];
            uint32_t l = off[o + 1] - off[o];
            if ((int)l == len && memcmp(res + off[o], buf, (size_t)len) == 0) { dup = 1; break; }
            slot = (slot + 1) & (cap - 1);
        }
        if (dup) continue;
        table[slot] = count;
        memcpy(res + off[count], buf, (size_t)len);
        off[count + 1] = off[count] + (uint32_t)len;
        count++;
    }
    free(table);
    return HMO_OK;
}
