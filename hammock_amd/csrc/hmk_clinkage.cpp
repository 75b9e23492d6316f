// hmk_clinkage.cpp -- exact complete-linkage clustering (Hammock's `clinkage` mode) on the GPU's thresholded
// neighbour graph.  Pure C++ host code, part of the product path.
//
// It reproduces ClinkageSequenceClusterer.cluster (ClinkageSequenceClusterer.java:43-124): a nearest-neighbour chain
// over the active clusters, where
//
//   clusterScore(c, top) (ClinkageClusterScorer.java:30-49, through CachedClusterScorer.java:38-79, which at one pool
//   thread is a transparent memo -- see oracle/hammock_oracle.py and tests/test_oracle.py) = min over member pairs, or
//   MIN_VALUE + 1 as soon as one pair is below the threshold
//     => c is a candidate for top  <=>  EVERY member pair is an edge of the >= threshold graph; then the score is the
//        minimum edge score.  Complete linkage is reducible, so the candidate list of a merged cluster is the
//        INTERSECTION of its parents' lists with the element-wise minimum score -- exactly what join() computes for the
//        cached rows (CachedClusterScorer.java:95-106).
//
//   findNearestClusterParallel(activeClusters, top, ..) (ClinkageSequenceClusterer.java:137-177,258-293)
//     = arg-max over the candidates of (score, Cluster.size(), smaller id); null if there is none.
//
// Every cluster keeps its candidate list sorted by cluster id.  New ids only grow (currentId++, :97), so appending the
// merged cluster to its candidates' lists keeps them sorted; entries of merged-away clusters are dropped lazily.
// The arbitrary start of every chain is activeClusters.iterator().next() of a java.util.HashSet<Cluster>: its iteration
// order (Java 8+: Cluster.hashCode() = 79 * 7 + id, HashMap.hash = h ^ h >>> 16, power-of-two table from 16, load factor
// 0.75, insertion-ordered chains, no shrinking) is emulated, as is the order of the returned list (the HashSet
// readyClusters, :121-123).
#include "hmk_internal.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

namespace hmk {

namespace {

// java.util.HashSet<Cluster> keyed by cluster id (see the header comment).
//   version 8 (default): Java 8 and later -- hash = h ^ h >>> 16, a new node is APPENDED to its bucket's chain, a resize splits
//                        the chains preserving their order.  A bucket that reaches 8 nodes in a table of 64+ would be
//                        treeified (its iteration order then starts at the tree's root): that is not modelled; it cannot
//                        happen for the consecutive ids used here at load factor 0.75, and `unmodelled()` says so if it did.
//   version 7: JDK 7u6 ... 7u80 (the reference is a Java 1.7 project, nbproject/project.properties:45-46) --
//              hash = h ^ h >>> 20 ^ h >>> 12, then h ^ h >>> 7 ^ h >>> 4; a new entry goes to the HEAD of its chain; addEntry
//              resizes BEFORE inserting when size >= threshold and the target bucket is not empty; transfer() walks the old
//              buckets in order and puts every entry at the head of its new bucket.
//   version 6: JDK 6 and JDK 7 before 7u6 -- as 7, but the entry is inserted first and the table resized when size++ >= threshold.
class JavaClusterSet {
    std::vector<int32_t> head_, tail_, next_;
    uint32_t cap_ = 16, size_ = 0, lowest_ = 0;
    int version_ = 8;
    bool unmodelled_ = false;
    uint32_t bucket(int32_t id) const {
        uint32_t h = (uint32_t)(553 + id);   // Cluster.java:178-183
        if (version_ == 8) h ^= h >> 16;
        else { h ^= (h >> 20) ^ (h >> 12); h ^= (h >> 7) ^ (h >> 4); }
        return h & (cap_ - 1);
    }
    void append(int32_t id) {   // Java 8+
        const uint32_t b = bucket(id);
        next_[id] = -1;
        if (head_[b] < 0) head_[b] = id; else next_[tail_[b]] = id;
        tail_[b] = id;
        lowest_ = std::min(lowest_, b);
        if (cap_ >= 64) {       // TREEIFY_THRESHOLD = 8, MIN_TREEIFY_CAPACITY = 64
            uint32_t len = 0;
            for (int32_t cur = head_[b]; cur >= 0; cur = next_[cur]) len++;
            if (len >= 8) unmodelled_ = true;
        }
    }
    void push_head(int32_t id) {   // Java <= 7: table[i] = new Entry(.., table[i])
        const uint32_t b = bucket(id);
        next_[id] = head_[b];
        if (head_[b] < 0) tail_[b] = id;
        head_[b] = id;
        lowest_ = std::min(lowest_, b);
    }
    void grow() {
        std::vector<int32_t> order;
        order.reserve(size_ + 1);
        for (uint32_t b = 0; b < cap_; b++)
            for (int32_t cur = head_[b]; cur >= 0; cur = next_[cur]) order.push_back(cur);
        cap_ *= 2;
        head_.assign(cap_, -1);
        tail_.assign(cap_, -1);
        lowest_ = cap_;
        for (int32_t id2 : order) { if (version_ == 8) append(id2); else push_head(id2); }
    }
public:
    JavaClusterSet(uint32_t max_id, int version) : head_(16, -1), tail_(16, -1), next_((size_t)max_id + 1, -1), version_(version) {}
    uint32_t size() const { return size_; }
    bool unmodelled() const { return unmodelled_; }
    void add(int32_t id) {
        const uint32_t threshold = cap_ / 4 * 3;
        if (version_ == 8) {
            append(id);
            if (++size_ > threshold) grow();   // resize(): chains are split preserving their order
        } else if (version_ == 7) {
            if (size_ >= threshold && head_[bucket(id)] >= 0) grow();
            push_head(id);
            size_++;
        } else {
            push_head(id);
            if (size_++ >= threshold) grow();
        }
    }
    void remove(int32_t id) {
        const uint32_t b = bucket(id);
        int32_t prev = -1;
        for (int32_t cur = head_[b]; cur >= 0; prev = cur, cur = next_[cur]) {
            if (cur != id) continue;
            if (prev < 0) head_[b] = next_[cur]; else next_[prev] = next_[cur];
            if (tail_[b] == cur) tail_[b] = prev;
            size_--;
            return;
        }
    }
    int32_t first() {   // iterator().next(); -1 if empty
        while (lowest_ < cap_ && head_[lowest_] < 0) lowest_++;
        return lowest_ < cap_ ? head_[lowest_] : -1;
    }
    template <class F> void for_each(F f) const {
        for (uint32_t b = 0; b < cap_; b++)
            for (int32_t cur = head_[b]; cur >= 0; cur = next_[cur]) f(cur);
    }
};

struct CNbr { int32_t id, score; };   // candidate cluster, complete-linkage score

template <class NbrT>
int clinkage_impl(uint32_t n, const int32_t *sizes, const uint64_t *start, const NbrT *adj, int32_t *cluster_id,
                  int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *st, std::string *err, int hashset_version) {
    const auto t0 = std::chrono::steady_clock::now();
    if (n == 0) {   // activeClusters.iterator().next() on an empty set, :118
        if (err) *err = "the reference throws NoSuchElementException here (ClinkageSequenceClusterer.java:118): empty input";
        return HMK_ERR_REFERENCE_WOULD_CRASH;
    }
    const uint32_t max_id = 2 * n + 2;
    std::vector<std::vector<CNbr>> cand((size_t)max_id + 1);
    std::vector<char> alive((size_t)max_id + 1, 0);
    std::vector<int64_t> csize((size_t)max_id + 1, 0);   // Cluster.size()
    // member lists as chains over the sequences: head / tail per cluster, next per sequence (top's members, then the
    // nearest's, :105-106)
    std::vector<int32_t> mhead((size_t)max_id + 1, -1), mtail((size_t)max_id + 1, -1), mnext(n, -1);
    JavaClusterSet active(max_id, hashset_version), ready(max_id, hashset_version);
    int32_t current_id = 1;
    for (uint32_t k = 0; k < n; k++) {   // :50-55: one cluster per sequence, ids from 1 in list order
        const int32_t id = current_id++;
        std::vector<CNbr> &l = cand[id];
        l.reserve(start[k + 1] - start[k]);
        for (uint64_t q = start[k]; q < start[k + 1]; q++) l.push_back(CNbr{(int32_t)adj[q].id() + 1, adj[q].score()});
        std::sort(l.begin(), l.end(), [](const CNbr &a, const CNbr &b) { return a.id < b.id; });
        alive[id] = 1;
        csize[id] = sizes ? sizes[k] : 1;
        mhead[id] = mtail[id] = (int32_t)k;
        active.add(id);
    }
    std::vector<int32_t> stack;
    std::vector<char> on_stack((size_t)max_id + 1, 0);
    std::vector<CNbr> merged;
    while (active.size() > 1) {                       // :63
        stack.push_back(active.first());              // :70-71
        on_stack[stack.back()] = 1;
        while (!stack.empty()) {                      // :72
            const int32_t top = stack.back();
            // nearest neighbour: arg-max (score, size, -id) over the live candidates; dead entries are dropped on the way
            std::vector<CNbr> &l = cand[top];
            size_t w = 0;
            int32_t nearest = -1, best = 0;
            for (size_t q = 0; q < l.size(); q++) {
                const CNbr e = l[q];
                if (!alive[e.id]) continue;
                l[w++] = e;
                if (nearest < 0 || e.score > best ||
                    (e.score == best && (csize[e.id] > csize[nearest] || (csize[e.id] == csize[nearest] && e.id < nearest)))) {
                    nearest = e.id;
                    best = e.score;
                }
            }
            l.resize(w);
            st->searches++;
            if (nearest < 0) {                        // :86-92 (every listed score is >= threshold)
                stack.pop_back();
                on_stack[top] = 0;
                ready.add(top);
                active.remove(top);
                continue;
            }
            if (stack.size() > 1 && stack[stack.size() - 2] == nearest) {   // :96
                current_id++;
                stack.pop_back();
                stack.pop_back();
                on_stack[top] = on_stack[nearest] = 0;
                active.remove(top);
                active.remove(nearest);
                const int32_t nid = current_id;
                // join (:102): candidates of the merged cluster = common candidates, element-wise min score
                merged.clear();
                const std::vector<CNbr> &a = cand[top], &b = cand[nearest];
                for (size_t i = 0, j = 0; i < a.size() && j < b.size();) {
                    if (a[i].id < b[j].id) i++;
                    else if (a[i].id > b[j].id) j++;
                    else {
                        if (alive[a[i].id] && a[i].id != top && a[i].id != nearest)
                            merged.push_back(CNbr{a[i].id, std::min(a[i].score, b[j].score)});
                        i++; j++;
                    }
                }
                alive[top] = alive[nearest] = 0;
                alive[nid] = 1;
                csize[nid] = csize[top] + csize[nearest];
                mnext[mtail[top]] = mhead[nearest];   // :105-106
                mhead[nid] = mhead[top];
                mtail[nid] = mtail[nearest];
                for (const CNbr &e : merged) cand[e.id].push_back(CNbr{nid, e.score});   // nid is the largest id: stays sorted
                cand[nid] = merged;
                std::vector<CNbr>().swap(cand[top]);
                std::vector<CNbr>().swap(cand[nearest]);
                active.add(nid);
                st->merges++;
            } else {
                if (on_stack[nearest]) {
                    // A tie (score, then size, then the smaller id) sent the chain back to a cluster that is still on the
                    // stack, below stack[-2].  The reference pushes it again (:113, no such check), merges or retires the upper
                    // copy, and later takes the stale Cluster object below as `top`: its members sit in another cluster by
                    // then, and the run ends in NoSuchElementException (:118, the active set ran empty) or returns a list in
                    // which a sequence belongs to two clusters -- nothing an int32 cluster_id[n] can reproduce.
                    if (err)
                        *err = "the reference's nearest-neighbour chain returns to cluster " + std::to_string(nearest) +
                               ", which is still on its stack (ClinkageSequenceClusterer.java:96-113 has no check): it goes on with a "
                               "stale Cluster object and throws NoSuchElementException (:118) or returns a sequence in two clusters";
                    return HMK_ERR_REFERENCE_WOULD_CRASH;
                }
                stack.push_back(nearest);             // :113
                on_stack[nearest] = 1;
            }
        }
    }
    ready.add(active.first());                        // :118
    int32_t out = 0;
    ready.for_each([&](int32_t id) {                  // :121-123: the HashSet's iteration order
        int32_t pos = 0;
        for (int32_t k = mhead[id]; k >= 0; k = mnext[k]) {
            cluster_id[k] = id;
            if (member_rank) member_rank[k] = pos;
            pos++;
        }
        if (result_order) result_order[out] = id;
        out++;
    });
    st->n_result_clusters = out;
    st->chain_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (active.unmodelled() || ready.unmodelled()) {   // never seen; a result whose order rests on an unmodelled tree bin is refused
        if (err) *err = "clinkage: a HashSet bucket reached 8 entries (Java 8+ would treeify it): its iteration order is not modelled";
        return HMK_ERR_BAD_ARG;
    }
    return HMK_OK;
}

}  // namespace

int clinkage_from_csr(int hashset_version, uint32_t n, const int32_t *sizes, const uint64_t *start, const Nbr *adj, int32_t *cluster_id,
                      int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *st, std::string *err) {
    return clinkage_impl<Nbr>(n, sizes, start, adj, cluster_id, result_order, member_rank, st, err, hashset_version);
}

int clinkage_from_csr_packed(int hashset_version, uint32_t n, const int32_t *sizes, const uint64_t *start, const NbrPacked *adj, int32_t *cluster_id,
                             int32_t *result_order, int32_t *member_rank, hmk_clinkage_stats *st, std::string *err) {
    return clinkage_impl<NbrPacked>(n, sizes, start, adj, cluster_id, result_order, member_rank, st, err, hashset_version);
}

}  // namespace hmk
