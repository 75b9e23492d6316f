/*
 * hammock_oracle.h -- CPU restatement of krejciadam/hammock's greedy
 * initial-clustering hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (hammock_amd/, libhammock_hip.so) never
 * links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" by reference-run outputs.  The reference is
 * Java 7; this image has no JDK, so the reference cannot be compiled or run
 * here and it ships no tests, golden vectors or fixtures for this path
 * (SURVEY.md section 4, 8c).  The restatement is pinned only by
 *   (1) the hand-derived known answers of SURVEY.md section 8(c)
 *       (tests/golden/known_answers.json), and
 *   (2) an independently written second restatement (oracle/hammock_oracle.py)
 *       that must agree with this one on random inputs.
 *
 * Every function cites the reference lines it follows.  Paths are relative to
 * /root/reference/src/cz/krejciadam/hammock/.
 */
#ifndef HAMMOCK_ORACLE_H
#define HAMMOCK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMO_ALPHABET 24

/* status codes */
#define HMO_OK 0
#define HMO_ERR_BAD_ARG 1
#define HMO_ERR_SHIFT_TOO_BIG 2 /* DataException, ShiftedScorer.java:59-62 */
#define HMO_ERR_OOM 4
/* the reference would throw NullPointerException:
 * stats.crash_case says which of the three branches */
#define HMO_ERR_REFERENCE_WOULD_CRASH 5

#define HMO_SCORER_SHIFTED 0
#define HMO_SCORER_LOCAL 1

/* UniqueSequence.java:23-26,46-57: char -> residue index over
 * ARNDCQEGHILKMFPSTWYVBZX*, case-folded.  Returns -1 for any other char. */
int hmo_encode_residue(char c);
char hmo_decode_residue(int r);

/* ShiftedScorer.java:48-95 (scoreWithShift) / :98-100 (sequenceScore).
 * M is int[24][24] row-major.  *shift may be NULL. */
int hmo_shifted_score(const int32_t *M, const uint8_t *seq1, int len1,
                      const uint8_t *seq2, int len2, int max_shift,
                      int shift_penalty, int32_t *score, int32_t *shift);

/* LocalAlignmentScorer.java:27-101.  seq1 = lines (rows), seq2 = columns. */
int32_t hmo_local_score(const int32_t *M, const uint8_t *seq1, int len1,
                        const uint8_t *seq2, int len2, int gap_open,
                        int gap_extend);

/* batch helpers (same semantics, used by the parity tests) */
int hmo_score_pairs(const int32_t *M, const uint8_t *res, const uint32_t *off,
                    const uint32_t *i, const uint32_t *j, uint64_t n_pairs,
                    int scorer, int a, int b, int32_t *out);

typedef struct {
    uint64_t score_calls_phase1; /* sequenceScore invocations, firstPhase */
    uint64_t score_calls_phase2; /* ... second loop of cluster() */
    int32_t phase1_stop_index;   /* value of `index` when firstPhase ends */
    int32_t phase1_clusters;     /* actualClusters.size() after firstPhase */
    int32_t phase1_orphans;      /* actualSequences.size() after firstPhase */
    int32_t crash_case;          /* 0 none; 1,2,3 = rows of SURVEY 3.2 table */
    int32_t crash_index;         /* index at which the NPE would be thrown */
    int32_t n_result_clusters;   /* size of the returned List<Cluster> */
    int32_t n_multi;             /* returned clusters with > 1 member */
} hmo_greedy_stats;

/*
 * LimitedGreedySequenceClusterer.java:39-120 driven the way
 * Hammock.java:402-409 drives it, on sequences ALREADY in greedy order
 * (UniqueSequence.sortSequences is hmo_sort_order below).
 *
 *   res/off    : residues (0..23) concatenated, off[n+1]
 *   size       : UniqueSequence.size() per sequence (NULL = all 1)
 *   scorer     : HMO_SCORER_SHIFTED (a = maxShift, b = shiftPenalty) or
 *                HMO_SCORER_LOCAL (a = gapOpen, b = gapExtend)
 *   n_threads  : Hammock.nThreads; partitions follow
 *                ClinkageSequenceClusterer.java:186-223.  Result is
 *                independent of it.
 *   cluster_id : out, [n]  id (= seed index) of the cluster holding seq k
 *   result_order: out, [n] ids of the returned clusters in list order
 *                (first stats->n_result_clusters entries are valid)
 *   member_rank: out, [n] or NULL: position of sequence k inside
 *                Cluster.getSequences() of its cluster = insertion order
 *                (Cluster.java:50-74; seed first, then the sequence absorbed
 *                at LimitedGreedySequenceClusterer.java:99-101/108-110, then
 *                the insertAll calls of :97, :104 and :62 in time order)
 */
int hmo_greedy_cluster(const int32_t *M, const uint8_t *res,
                       const uint32_t *off, const int32_t *size, uint32_t n,
                       int scorer, int a, int b, int threshold,
                       int max_clusters, int n_threads, int32_t *cluster_id,
                       int32_t *result_order, int32_t *member_rank,
                       hmo_greedy_stats *stats);

typedef struct {
    uint64_t score_calls;      /* sequenceScore invocations behind the cluster scores */
    int32_t merges;            /* clusters joined (ClinkageSequenceClusterer.java:96-111) */
    int32_t searches;          /* findNearestClusterParallel calls (:77) */
    int32_t n_result_clusters; /* size of the returned list */
    int32_t reserved;
} hmo_clinkage_stats;

/*
 * ClinkageSequenceClusterer(ShiftedScorer(M, shift_penalty, max_shift), threshold).cluster(sequences)
 * (ClinkageSequenceClusterer.java:43-124 driven as Hammock.java:458-462 drives it; sequences in LOAD order, clinkage
 * mode does not sort).  Exact complete linkage by nearest-neighbour chain; the arbitrary start of every chain is
 * activeClusters.iterator().next() of a java.util.HashSet<Cluster> (Java 8+ iteration order, see hammock_oracle.c).
 *   cluster_id[n]  : id of the returned cluster holding sequence k (singletons: k + 1; merged clusters: n + 2, n + 3,
 *                    ... in merge order, ClinkageSequenceClusterer.java:49-55,97)
 *   result_order[] : ids in the order of the returned list (= iteration order of the HashSet readyClusters)
 *   member_rank[n] : position of sequence k in its cluster's member list (top's members, then the nearest's, :105-106)
 * An empty input is HMO_ERR_REFERENCE_WOULD_CRASH (NoSuchElementException at :118).
 */
/* Which java.util.HashSet iteration order the clinkage restatement emulates for `activeClusters.iterator().next()` and the
 * returned list (ClinkageSequenceClusterer.java:70,118-123): 8 = Java 8 and later (default), 7 = JDK 7u6 ... 7u80,
 * 6 = JDK 6 and JDK 7 before 7u6.  Process-wide. */
void hmo_set_java_hashset(int version);
int hmo_get_java_hashset(void);
int hmo_clinkage_cluster(const int32_t *M, const uint8_t *res, const uint32_t *off, const int32_t *size, uint32_t n,
                         int max_shift, int shift_penalty, int threshold, int n_threads, int32_t *cluster_id,
                         int32_t *result_order, int32_t *member_rank, hmo_clinkage_stats *stats);

/* UniqueSequence.java:176-203 for order "size" / "alphabetic" / "input":
 * writes the permutation (perm[k] = input index of the k-th sequence in
 * greedy order).  order: 0 size, 1 alphabetic, 2 input. */
int hmo_sort_order(const uint8_t *res, const uint32_t *off,
                   const int32_t *size, uint32_t n, int order, uint32_t *perm);

/* SURVEY.md 8(d) synthetic generator: SplitMix64(seed), residue =
 * (next() >> 33) % 20; fixed length if len_lo == len_hi, else length =
 * len_lo + (next() >> 33) % (len_hi - len_lo + 1) drawn before the residues;
 * draws until n distinct peptides.  res must hold n * len_hi bytes. */
int hmo_synth(uint64_t seed, uint32_t n, int len_lo, int len_hi, uint8_t *res,
              uint32_t *off);

#ifdef __cplusplus
}
#endif
#endif
