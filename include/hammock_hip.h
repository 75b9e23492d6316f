/*
 * hammock_hip.h -- C ABI of libhammock_hip.so, the MI355X (gfx950) drop-in for
 * the greedy initial-clustering hot path of krejciadam/hammock v1.2.0.
 *
 * The reference has no FFI seam of its own; its seams are the Java interfaces
 * below (paths relative to src/cz/krejciadam/hammock/).  Each entry point names
 * the reference interface it replaces.  A JNI / cgo / ctypes binding binds
 * exactly these symbols (INTEGRATION.md shows the Java side).
 *
 * Conventions
 *   - plain C types only; every output buffer is caller-allocated
 *   - residues are indices 0..23 over "ARNDCQEGHILKMFPSTWYVBZX*"
 *     (UniqueSequence.java:23-26); the scoring matrix is int[24][24] row-major
 *     exactly as FileIOManager.loadScoringMatrix returns it (FileIOManager.java:46-81)
 *   - score(i, j) always means SequenceScorer.sequenceScore(seq1 = sequence i,
 *     seq2 = sequence j) (SequenceScorer.java:14)
 *   - every function returns an hmk_status; hmk_last_error() gives the text
 *   - there is no CPU fallback: a scoring call on a context without a usable
 *     GPU returns HMK_ERR_DEVICE
 */
#ifndef HAMMOCK_HIP_H
#define HAMMOCK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMK_ABI_VERSION 4
#define HMK_ALPHABET 24
#define HMK_MAX_LEN 32          /* longest sequence the GPU kernels accept */
#define HMK_MAX_SEQUENCES (1u << 24)
#define HMK_EDGE_SHARDS 64      /* output segments of the neighbour kernel (ABI 3: 64, was 16 -- a segment's cursor is ONE address, and a
                                 * pass whose waves drain 4 x 10^5 stages serialises on 16 of them) */

typedef enum {
    HMK_OK = 0,
    HMK_ERR_BAD_ARG = 1,
    HMK_ERR_SHIFT_TOO_BIG = 2,          /* DataException, ShiftedScorer.java:59-62 */
    HMK_ERR_DEVICE = 3,                 /* HIP error / no usable gfx950 device */
    HMK_ERR_OOM = 4,
    HMK_ERR_REFERENCE_WOULD_CRASH = 5,  /* NullPointerException in
                                           LimitedGreedySequenceClusterer.java:97/104/108
                                           (surfaces at Hammock.java:153-157) */
    HMK_ERR_CAPACITY = 6,               /* caller's edge buffer too small; *n_edges = needed */
    HMK_ERR_NO_SEQUENCES = 7
} hmk_status;

typedef struct hmk_ctx hmk_ctx;

/* Edge of the thresholded neighbour graph, packed in one uint64:
 *   bits 63..40  x      (sequence index, 24 bit)
 *   bits 39..16  m      (sequence index, 24 bit)
 *   bits 15..0   score  (int16, two's complement)
 * meaning  sequenceScore(seq1 = m, seq2 = x) = score >= threshold.
 * With a symmetric matrix each unordered pair appears once with x < m. */
#define HMK_EDGE_X(e) ((uint32_t)((e) >> 40))
#define HMK_EDGE_M(e) ((uint32_t)(((e) >> 16) & 0xFFFFFFu))
#define HMK_EDGE_SCORE(e) ((int32_t)(int16_t)((e) & 0xFFFFu))

/* ---- context ---------------------------------------------------------- */

/* Replaces `new ShiftedScorer(scoringMatrix, ..)` / `new LocalAlignmentScorer`
 * construction state (ShiftedScorer.java:28-32, LocalAlignmentScorer.java:20-24):
 * holds the matrix (Hammock.scoringMatrix, Hammock.java:46,1264) on `device`.
 * device >= 0: HIP device ordinal.  device == -1: host-only context (only the
 * host-side calls hmk_greedy_from_edges / hmk_set_sequences work).
 * The context comes warm: its streams, events, pinned blocks, the DMA paths of
 * both directions and the kernels' code objects are set up here (25-35 ms after
 * HIP itself has started), not inside the first clustering call -- the reference
 * constructs its scorer before it starts the clock of "Clustering time"
 * (Hammock.java:402-406), and a host may create the context on another thread
 * while it reads its input (hammock_cli.cpp does).  HMK_LAZY_CONTEXT=1 defers. */
int hmk_create(const int32_t matrix[HMK_ALPHABET * HMK_ALPHABET], int device, hmk_ctx **ctx);
/* The same on several GPUs of one node (BASELINE config 5: "pair-space sharded 8 x MI355X over xGMI"): devices[0] is the
 * root.  hmk_set_sequences uploads to every device; hmk_greedy_cluster scores shard d of n_devices on device d (row
 * blocks dealt cyclically, no collective inside the scoring), gathers the peers' edge segments to the root with direct
 * xGMI peer copies (every peer over its own link; a gather to the one device whose host runs the merge moves 1/n of an
 * all-gather's bytes) and runs the merge tail there.  The result is identical to the single-device call.  Every other
 * entry point works on the root device.  n_devices = 1 is hmk_create.  (One PROCESS per GPU with an RCCL all-gather
 * -- hammock_amd/dist.py -- is the other multi-GPU form; both end in hmk_greedy_from_edges_dev on rank/device 0.) */
int hmk_create_multi(const int32_t matrix[HMK_ALPHABET * HMK_ALPHABET], const int *devices, int n_devices, hmk_ctx **ctx);
int hmk_device_count(const hmk_ctx *ctx); /* devices behind this context (0 for a host-only context) */
void hmk_destroy(hmk_ctx *ctx);
const char *hmk_last_error(const hmk_ctx *ctx); /* ctx may be NULL */
int hmk_abi_version(void);
/* Device time (HIP events, milliseconds) of the scoring kernel(s) the last hmk_score_pairs_* /
 * hmk_score_block_* / hmk_score_with_shift call launched; excludes the host<->device copies. */
double hmk_last_kernel_ms(const hmk_ctx *ctx);

/* The List<UniqueSequence> handed to SequenceClusterer.cluster
 * (SequenceClusterer.java:24), flattened in the caller's (greedy) order:
 * residues concatenated, offsets[n+1], sizes[n] = UniqueSequence.size()
 * (UniqueSequence.java:82-88; NULL = all 1).  Copies everything. */
int hmk_set_sequences(hmk_ctx *ctx, const uint8_t *residues, const uint32_t *offsets,
                      const int32_t *sizes, uint32_t n);

/* ---- pairwise scorers (parity probes of SequenceScorer.sequenceScore) ---- */

/* out[k] = ShiftedScorer(matrix, shift_penalty, max_shift).sequenceScore(i[k], j[k])
 * (ShiftedScorer.java:48-100).  HMK_ERR_SHIFT_TOO_BIG if max_shift >= the
 * shorter length of any pair. */
int hmk_score_pairs_shifted(hmk_ctx *ctx, const uint32_t *i, const uint32_t *j, uint64_t n_pairs,
                            int max_shift, int shift_penalty, int32_t *out);
/* AligningSequenceScorer.scoreWithShift (AligningSequenceScorer.java:11, ShiftedScorer.java:48-95):
 * score[k] and shift[k] = AligningScorerResult.getScore() / getShift() for the pair (i[k], j[k]);
 * the first strict maximum wins (:86-89), the sign follows :91-93. */
int hmk_score_with_shift(hmk_ctx *ctx, const uint32_t *i, const uint32_t *j, uint64_t n_pairs,
                         int max_shift, int shift_penalty, int32_t *score, int32_t *shift);
/* out[k] = LocalAlignmentScorer(matrix, gap_open, gap_extend).sequenceScore(i[k], j[k])
 * (LocalAlignmentScorer.java:27-86); i = seq1 = lines, j = seq2 = columns. */
int hmk_score_pairs_local(hmk_ctx *ctx, const uint32_t *i, const uint32_t *j, uint64_t n_pairs,
                          int gap_open, int gap_extend, int32_t *out);
/* Dense block: out[(r - r0) * (c1 - c0) + (c - c0)] = score(r, c), r0<=r<r1, c0<=c<c1. */
int hmk_score_block_shifted(hmk_ctx *ctx, uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1,
                            int max_shift, int shift_penalty, int32_t *out);
int hmk_score_block_local(hmk_ctx *ctx, uint32_t r0, uint32_t r1, uint32_t c0, uint32_t c1,
                          int gap_open, int gap_extend, int32_t *out);

/* ---- all-vs-all thresholded neighbour graph (the hot kernel) ------------ */

typedef struct {
    uint64_t n_edges;        /* edges produced by this call */
    uint64_t pairs_scored;   /* pairs the kernels evaluated (unordered unless asymmetric) */
    uint32_t n_tiles;        /* workgroups launched */
    uint32_t symmetric;      /* 1: matrix symmetric, triangle only */
    uint32_t classes_u8;     /* (row length, column length) classes on the 8-bit lane path */
    uint32_t classes_u16;    /* ... on the 16-bit lane path */
    uint32_t classes_direct; /* ... on the generic one-cell-at-a-time path */
    uint32_t classes_rows;   /* of the 8-bit lane classes: how many run the row-packed kernels (8 rows per table entry) */
    double kernel_ms;        /* device time of the scoring kernels (HIP events) */
} hmk_neighbor_stats;

/* Scores every pair with ShiftedScorer semantics and returns the pairs with
 * score >= threshold as packed edges (see HMK_EDGE_*).  This is the batch
 * form of what ClinkageClusterScorer.clusterScore's early exit
 * (ClinkageClusterScorer.java:36-48) asks of the scorer pair by pair.
 * The pair space is split row-block-wise into n_parts shards; this call
 * computes shard `part` (single GPU: part = 0, n_parts = 1).
 * edges: host buffer of `capacity` entries; *n_edges receives the count
 * (HMK_ERR_CAPACITY and the needed count if it does not fit). */
int hmk_neighbors_shifted(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold,
                          uint32_t part, uint32_t n_parts, uint64_t *edges, uint64_t capacity,
                          uint64_t *n_edges, hmk_neighbor_stats *stats);

/* The same for LocalAlignmentScorer(matrix, gap_open, gap_extend) (LocalAlignmentScorer.java:27-86): ALL
 * ORDERED pairs (the scorer depends on the argument order) with sequenceScore(seq1 = m, seq2 = x) >= threshold,
 * as edges (x, m, score).  This is the batch form of the stage-2 pre-filter `sequenceScore(first, i) >= 12`
 * (ClustalRunner.java:81-96, Hammock.java:118-121,645-649).  Gap penalties <= 0 and |matrix| <= 127 run the striped
 * register kernels; anything else (the reference restricts neither) runs the literal DP on the same tiles. */
int hmk_neighbors_local(hmk_ctx *ctx, int gap_open, int gap_extend, int threshold, uint32_t part,
                        uint32_t n_parts, uint64_t *edges, uint64_t capacity, uint64_t *n_edges,
                        hmk_neighbor_stats *stats);

/* Device-resident form: asynchronous on `stream` (a hipStream_t, may be NULL),
 * no host synchronisation.  d_edges: device buffer of `capacity` uint64,
 * logically HMK_EDGE_SHARDS segments of capacity / HMK_EDGE_SHARDS entries;
 * d_counts: device uint64[HMK_EDGE_SHARDS], zeroed by the call, receives the
 * number of edges each segment WANTED to hold (may exceed the segment size:
 * overflow, entries beyond the segment are dropped). */
int hmk_neighbors_shifted_dev(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold,
                              uint32_t part, uint32_t n_parts, void *d_edges, uint64_t capacity,
                              void *d_counts, void *stream);
/* Device to device, asynchronous on `stream`: packs the HMK_EDGE_SHARDS segments of a
 * hmk_neighbors_shifted_dev result into one contiguous block d_out[0 .. *d_total) (at most
 * out_capacity entries are written) -- the block a fixed-size all-gather ships to the other ranks. */
int hmk_compact_edges_dev(hmk_ctx *ctx, const void *d_edges, uint64_t capacity, const void *d_counts,
                          void *d_out, uint64_t out_capacity, void *d_total, void *stream);
/* "Row blocks": the 4-byte-per-edge form of a rank's edges that the multi-GPU exchange ships (half the
 * xGMI bytes of the packed edges).  Device to device, asynchronous on `stream`; calls on one context must be
 * stream-ordered (they share context-owned scratch).  Regroups the HMK_EDGE_SHARDS segments by x:
 *   d_row_start  uint32[n + 2]: entries of row x are d_adj[d_row_start[x] .. d_row_start[x + 1]);
 *                [n] = number of edges, [n + 1] = edges whose score - threshold did not fit 0..255
 *                (must be 0 for the block to be usable; fall back to hmk_compact_edges_dev otherwise)
 *   d_adj        uint32[adj_capacity]: m << 8 | (score - threshold); order inside a row is arbitrary.
 * n = the sequences of hmk_set_sequences.  Part of the replacement for the per-candidate
 * scoring loop's hand-over (NearestCluster results, LimitedGreedySequenceClusterer.java:118-180). */
int hmk_pack_rows_dev(hmk_ctx *ctx, const void *d_edges, uint64_t capacity, const void *d_counts, int threshold,
                      void *d_row_start, void *d_adj, uint64_t adj_capacity, void *stream);
/* Inverse: writes the row block's edges as packed 8-byte edges to d_edges_out[0 .. d_row_start[n]). */
int hmk_unpack_rows_dev(hmk_ctx *ctx, const void *d_row_start, const void *d_adj, int threshold, void *d_edges_out,
                        uint64_t out_capacity, void *stream);
/* pairs / tiles of the plan the last hmk_neighbors_shifted[_dev] call used */
int hmk_neighbors_last_plan(hmk_ctx *ctx, hmk_neighbor_stats *stats);

/* ---- greedy clustering -------------------------------------------------- */

typedef struct {
    uint64_t n_edges;            /* neighbour edges consumed */
    int32_t phase1_stop_index;   /* `index` when firstPhase ends (LimitedGreedy..java:90) */
    int32_t phase1_clusters;
    int32_t phase1_orphans;
    int32_t crash_case;          /* 0; or 1,2,3 = which NPE the reference would throw */
    int32_t crash_index;
    int32_t n_result_clusters;   /* size of the returned List<Cluster> */
    int32_t n_multi;             /* clusters with more than one member */
    int32_t reserved;
    double neighbors_ms;         /* GPU scoring (hmk_greedy_cluster only) */
    double greedy_ms;            /* host greedy merge */
} hmk_greedy_stats;

/* Replaces LimitedGreedySequenceClusterer(scorer, threshold, maxClusters).cluster(sequences)
 * (LimitedGreedySequenceClusterer.java:22,39-69) with scorer =
 * ShiftedScorer(matrix, shift_penalty, max_shift), on the sequences of
 * hmk_set_sequences (already in greedy order, UniqueSequence.sortSequences).
 *   cluster_id[n]   : id of the cluster holding sequence k (= index of its seed,
 *                     LimitedGreedySequenceClusterer.java:82)
 *   result_order[n] : ids of the returned clusters in list order (clusters first,
 *                     then singletons); first n_result_clusters entries valid; may be NULL
 *   member_rank[n]  : position of sequence k inside Cluster.getSequences() of its
 *                     cluster (insertion order, seed = 0); may be NULL
 * Returns HMK_ERR_REFERENCE_WOULD_CRASH where the reference throws
 * NullPointerException (stats->crash_case says which). */
int hmk_greedy_cluster(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold,
                       int max_clusters, int32_t *cluster_id, int32_t *result_order,
                       int32_t *member_rank, hmk_greedy_stats *stats);

/* The host-side greedy merge alone, on an edge list produced by
 * hmk_neighbors_shifted (all shards concatenated, any order).
 * symmetric != 0: each edge stands for both directions. */
int hmk_greedy_from_edges(hmk_ctx *ctx, const uint64_t *edges, uint64_t n_edges, int symmetric,
                          int threshold, int max_clusters, int32_t *cluster_id,
                          int32_t *result_order, int32_t *member_rank, hmk_greedy_stats *stats);

/* The same with the edges in device memory (one contiguous block of packed edges, e.g. what the ranks'
 * all-gather left on every GPU): adjacency built on the device, one pinned copy to the host, host merge --
 * the tail of hmk_greedy_cluster without the scoring (LimitedGreedySequenceClusterer.java:39-120 on a
 * precomputed neighbour graph).  Invalid edges (index >= n, self pairs) give HMK_ERR_BAD_ARG. */
int hmk_greedy_from_edges_dev(hmk_ctx *ctx, const void *d_edges, uint64_t n_edges, int symmetric,
                              int max_clusters, int32_t *cluster_id, int32_t *result_order,
                              int32_t *member_rank, hmk_greedy_stats *stats);

/* ---- clinkage mode ------------------------------------------------------- */

typedef struct {
    uint64_t n_edges;            /* neighbour edges consumed */
    int32_t merges;              /* clusters joined (ClinkageSequenceClusterer.java:96-111) */
    int32_t searches;            /* nearest-neighbour searches (:77) */
    int32_t n_result_clusters;   /* size of the returned List<Cluster> */
    int32_t reserved;
    double neighbors_ms;         /* GPU scoring */
    double chain_ms;             /* host nearest-neighbour chain */
} hmk_clinkage_stats;

/* Replaces ClinkageSequenceClusterer(ShiftedScorer(matrix, shift_penalty, max_shift), threshold).cluster(sequences)
 * (ClinkageSequenceClusterer.java:29-33,43-124; driver Hammock.java:449-462; the reference's default initial clustering
 * for up to 10,000 unique sequences, Hammock.java:371-377) on the sequences of hmk_set_sequences, which are in LOAD
 * order -- clinkage mode does not sort.  The whole pair space is scored on the GPU; the nearest-neighbour chain runs
 * on the host over the thresholded graph.
 *   cluster_id[n]   : id of the returned cluster holding sequence k: k + 1 for a sequence left alone, n + 2, n + 3, ...
 *                     for merged clusters in merge order (:49-55,97)
 *   result_order[n] : ids of the returned clusters in list order = iteration order of the java.util.HashSet
 *                     readyClusters (:121-123; Java 8 and later); first n_result_clusters entries valid; may be NULL
 *   member_rank[n]  : position of sequence k inside Cluster.getSequences() (:105-106); may be NULL
 * The cacheSizeLimit of -L/--cache_size_limit never reaches the clusterer in the reference (Hammock.java:459 uses the
 * two-argument constructor), so there is no such parameter.  Needs a symmetric matrix (HMK_ERR_BAD_ARG otherwise: the
 * reference's score cache is keyed by the unordered pair, so with score(a,b) != score(b,a) its result depends on which
 * direction happened to be asked first).  An empty input is HMK_ERR_REFERENCE_WOULD_CRASH (NoSuchElementException, :118).
 * So is an input on which the reference's chain returns to a cluster that is still on its stack (a tie of score, size
 * and id order, :96-113 pushes it again): the reference then works with a stale Cluster object and throws
 * NoSuchElementException or returns a list in which a sequence belongs to two clusters -- there is no cluster_id[] for
 * that (hmk_last_error names the cluster; four 6-mers suffice, tests/test_oracle.py). */
int hmk_clinkage_cluster(hmk_ctx *ctx, int max_shift, int shift_penalty, int threshold, int32_t *cluster_id,
                         int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *stats);

/* Optional.  Sizes the context's grow-only device and pinned buffers for a first hmk_greedy_cluster / hmk_clinkage_cluster
 * call on n_sequences sequences (edge buffer at the first guess of 0.3 % of the pair space, adjacency, CSR and second-loop
 * scratch: 36 GB at 10^6), so that a host which knows the sequence count early -- hammock-hip after it has read its input --
 * can have the allocations done on another thread while it is still sorting and packing the sequences.  Calls on a context
 * are serialised; a later call that needs more grows the buffers as usual.  HMK_OK on a host-only context (nothing to do).
 * The two buffers a call needs LAST (the adjacency and the CSR's bucket records, 2 x 11 GB at 10^6) are obtained on a thread
 * of the library's own after this function has returned -- on some hosts fresh device memory of that size takes 0.3-1.5 s to
 * get --; a clustering call that starts in the meantime scores, hands the band over and runs phase 1 first and enqueues its
 * CSR step when they are there (hmk_destroy joins the thread). */
int hmk_reserve(hmk_ctx *ctx, uint32_t n_sequences);

/* Which java.util.HashSet iteration order hmk_clinkage_cluster / hmk_clinkage_from_edges emulate for
 * `activeClusters.iterator().next()` (ClinkageSequenceClusterer.java:70, the arbitrary start of every chain) and for the
 * returned list (:118-123).  version 8 (default): Java 8 and later; 7: JDK 7u6 ... 7u80 (the reference is a Java 1.7
 * project, nbproject/project.properties:45-46); 6: JDK 6 and JDK 7 before 7u6.  The orders differ in hash spreading, in
 * where a new entry joins its bucket's chain and in what a resize does to a chain; cluster MEMBERSHIP differs only where a
 * tie of score, size and id order lets the chain start decide.  HMK_ERR_BAD_ARG for any other version. */
int hmk_set_java_hashset(hmk_ctx *ctx, int version);


/* The host-side nearest-neighbour chain alone, on the edge list of a symmetric matrix as hmk_neighbors_shifted produces it
 * (each unordered pair once, any order; all shards concatenated).  Works on a host-only context (device = -1). */
int hmk_clinkage_from_edges(hmk_ctx *ctx, const uint64_t *edges, uint64_t n_edges, int32_t *cluster_id,
                            int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *stats);

/* Where the time of the last hmk_greedy_cluster / hmk_greedy_from_edges_dev call of this context went
 * (milliseconds; the span of Hammock.java:406-411 minus the sort).  score_ms and csr_ms are device times (HIP events on
 * the call's stream), the others host wall time.  The parts overlap (phase 1 runs while the rest of the pair space is
 * being scored), so they do not add up to total_ms. */
typedef struct {
    double plan_ms;          /* tiling plan (cached between calls with the same parameters) */
    double score_ms;         /* neighbour kernels, first launch to last edge */
    double csr_ms;           /* edge segments -> CSR adjacency on the device */
    double wait_rows_ms;     /* host waiting for adjacency rows (band hand-over, later fetches) */
    double phase1_ms;        /* firstPhase, LimitedGreedySequenceClusterer.java:77-120 (includes wait_rows_ms) */
    double precheck_ms;      /* device pre-check of the second loop (:59-66) incl. copies */
    double exchange_ms;      /* multi-device calls: routing the edges to the devices that own their rows, until the last block has landed */
    double device_loop_ms;   /* the second loop on the device in optimistic rounds (large inputs), incl. copies */
    double host_precheck_ms; /* host wall time between the end of phase 1 and the sequential part: the wait for the full CSR, the
                              * pre-check and the device-side loop (includes precheck_ms and device_loop_ms) */
    double sequential_ms;    /* the order-dependent loop :59-66 itself */
    double total_ms;
    uint64_t cand_entries;   /* (leftover, feasible cluster) pairs after phase 1 */
    uint64_t band_bytes;     /* bytes of the band, as prepared for phase 1, that crossed PCIe (0: the band went as whole rows, or not at all) */
    uint64_t loop_rounds;    /* rounds of the device-side second loop, 0 if the loop ran on the host */
} hmk_greedy_phases;
int hmk_greedy_last_phases(const hmk_ctx *ctx, hmk_greedy_phases *out);

#ifdef __cplusplus
}
#endif
#endif /* HAMMOCK_HIP_H */
