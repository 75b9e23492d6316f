// hammock_host.hpp -- C++ host-side mirror of the reference's Java interfaces for the
// greedy initial-clustering path, on top of the C ABI of include/hammock_hip.h.
//
// The reference is Java; this image has no JDK, so the host side above the C ABI
// is C++ (the Java shim a Hammock maintainer would add is shipped as source under
// hammock_amd/java/).  Class names, argument order and error
// behaviour follow the reference (paths relative to src/cz/krejciadam/hammock/):
//
//   UniqueSequence            UniqueSequence.java:19
//   Cluster                   Cluster.java:21
//   SequenceScorer / AligningSequenceScorer / SequenceClusterer   (interfaces)
//   ShiftedScorer             ShiftedScorer.java:12          sequenceScore runs on the GPU
//   LocalAlignmentScorer      LocalAlignmentScorer.java:10   sequenceScore runs on the GPU
//   HipGreedySequenceClusterer  drop-in for LimitedGreedySequenceClusterer.java:17
//   FileIOManager             loaders / stage-1 writers, FileIOManager.java
//   Logger                    Logger.java
//
// Nothing in this header computes a score on the CPU.
#ifndef HAMMOCK_HOST_HPP
#define HAMMOCK_HOST_HPP

#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/hammock_hip.h"

namespace hammock {

// how many threads the host-side helpers below may use (parsing, object construction, formatting)
inline unsigned hostThreads() {
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(16u, hw ? hw : 1u));
}
// f(t, lo, hi) over [0, n) cut into one contiguous range per thread; the first exception is rethrown
template <class F>
inline void parallelRanges(size_t n, unsigned threads, F f) {
    threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / 4096 + 1));
    if (threads == 1) { f(0u, (size_t)0, n); return; }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> failed(threads);
    for (unsigned t = 0; t < threads; t++)
        pool.emplace_back([&, t]() {
            try { f(t, n * t / threads, n * (t + 1) / threads); } catch (...) { failed[t] = std::current_exception(); }
        });
    for (std::thread &th : pool) th.join();
    for (auto &e : failed) if (e) std::rethrow_exception(e);
}
// f(lo, hi) over [0, n) in chunks of `chunk` items handed out by an atomic cursor (for work whose cost per item is uneven:
// clusters of 1 .. 10^4 members); the first exception is rethrown
template <class F>
inline void parallelChunks(size_t n, size_t chunk, unsigned threads, F f) {
    threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, (n + chunk - 1) / std::max<size_t>(chunk, 1)));
    if (threads <= 1) { if (n) f((size_t)0, n); return; }
    std::atomic<size_t> cursor{0};
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> failed(threads);
    for (unsigned t = 0; t < threads; t++)
        pool.emplace_back([&, t]() {
            try {
                for (;;) {
                    const size_t lo = cursor.fetch_add(chunk);
                    if (lo >= n) break;
                    f(lo, std::min(n, lo + chunk));
                }
            } catch (...) { failed[t] = std::current_exception(); }
        });
    for (std::thread &th : pool) th.join();
    for (auto &e : failed) if (e) std::rethrow_exception(e);
}

// ---- exceptions (HammockException.java and subclasses) ---------------------------------
struct HammockException : std::runtime_error { using std::runtime_error::runtime_error; };
struct DataException : HammockException { using HammockException::HammockException; };
struct FileFormatException : HammockException { using HammockException::HammockException; };
struct CLIException : HammockException { using HammockException::HammockException; };
struct DeviceException : HammockException { using HammockException::HammockException; };
// The reference throws java.lang.NullPointerException here
// (LimitedGreedySequenceClusterer.java:97/104/108, caught at Hammock.java:153-157).
struct NullPointerException : std::runtime_error {
    int crashCase, crashIndex;
    NullPointerException(const std::string &m, int c, int i) : std::runtime_error(m), crashCase(c), crashIndex(i) {}
};

static const char AMINO_ACIDS[25] = "ARNDCQEGHILKMFPSTWYVBZX*";  // UniqueSequence.java:23-26

// The default scoring matrix of the reference is <jar parent>/matrices/blosum62.txt (Hammock.java:45).  The build
// writes that file next to the hammock-hip binary from this table (`hammock-hip dump-matrix`): the NCBI BLOSUM62
// half-bit table over AMINO_ACIDS, row = first residue, column = second.
static const int BLOSUM62[24][24] = {
    { 4, -1, -2, -2,  0, -1, -1,  0, -2, -1, -1, -1, -1, -2, -1,  1,  0, -3, -2,  0, -2, -1,  0, -4},
    {-1,  5,  0, -2, -3,  1,  0, -2,  0, -3, -2,  2, -1, -3, -2, -1, -1, -3, -2, -3, -1,  0, -1, -4},
    {-2,  0,  6,  1, -3,  0,  0,  0,  1, -3, -3,  0, -2, -3, -2,  1,  0, -4, -2, -3,  3,  0, -1, -4},
    {-2, -2,  1,  6, -3,  0,  2, -1, -1, -3, -4, -1, -3, -3, -1,  0, -1, -4, -3, -3,  4,  1, -1, -4},
    { 0, -3, -3, -3,  9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1, -3, -3, -2, -4},
    {-1,  1,  0,  0, -3,  5,  2, -2,  0, -3, -2,  1,  0, -3, -1,  0, -1, -2, -1, -2,  0,  3, -1, -4},
    {-1,  0,  0,  2, -4,  2,  5, -2,  0, -3, -3,  1, -2, -3, -1,  0, -1, -3, -2, -2,  1,  4, -1, -4},
    { 0, -2,  0, -1, -3, -2, -2,  6, -2, -4, -4, -2, -3, -3, -2,  0, -2, -2, -3, -3, -1, -2, -1, -4},
    {-2,  0,  1, -1, -3,  0,  0, -2,  8, -3, -3, -1, -2, -1, -2, -1, -2, -2,  2, -3,  0,  0, -1, -4},
    {-1, -3, -3, -3, -1, -3, -3, -4, -3,  4,  2, -3,  1,  0, -3, -2, -1, -3, -1,  3, -3, -3, -1, -4},
    {-1, -2, -3, -4, -1, -2, -3, -4, -3,  2,  4, -2,  2,  0, -3, -2, -1, -2, -1,  1, -4, -3, -1, -4},
    {-1,  2,  0, -1, -3,  1,  1, -2, -1, -3, -2,  5, -1, -3, -1,  0, -1, -3, -2, -2,  0,  1, -1, -4},
    {-1, -1, -2, -3, -1,  0, -2, -3, -2,  1,  2, -1,  5,  0, -2, -1, -1, -1, -1,  1, -3, -1, -1, -4},
    {-2, -3, -3, -3, -2, -3, -3, -3, -1,  0,  0, -3,  0,  6, -4, -2, -2,  1,  3, -1, -3, -3, -1, -4},
    {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4,  7, -1, -1, -4, -3, -2, -2, -1, -2, -4},
    { 1, -1,  1,  0, -1,  0,  0,  0, -1, -2, -2,  0, -1, -2, -1,  4,  1, -3, -2, -2,  0,  0,  0, -4},
    { 0, -1,  0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1,  1,  5, -2, -2,  0, -1, -1,  0, -4},
    {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1,  1, -4, -3, -2, 11,  2, -3, -4, -3, -2, -4},
    {-2, -2, -2, -3, -2, -1, -2, -3,  2, -1, -1, -2, -1,  3, -3, -2, -2,  2,  7, -1, -3, -2, -1, -4},
    { 0, -3, -3, -3, -1, -2, -2, -3, -3,  3,  1, -2,  1, -1, -2, -2,  0, -3, -1,  4, -3, -2, -1, -4},
    {-2, -1,  3,  4, -3,  0,  1, -1,  0, -3, -4,  0, -3, -3, -2,  0, -1, -4, -3, -3,  4,  1, -1, -4},
    {-1,  0,  0,  1, -3,  3,  4, -2,  0, -3, -3,  1, -1, -3, -1,  0, -1, -3, -2, -2,  1,  4, -1, -4},
    { 0, -1, -1, -1, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -2,  0,  0, -2, -1, -1, -1, -1, -1, -4},
    {-4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4,  1}};
static const char CSV_SEPARATOR = '\t';                          // Hammock.java:35

// ---- small Java-semantics helpers ------------------------------------------------------------
// Integer.decode: sign, then 0x / 0X / # hex, leading-0 octal, else decimal
inline int javaIntegerDecode(const std::string &text) {
    std::string t = text;
    bool neg = false;
    size_t p = 0;
    if (!t.empty() && (t[0] == '-' || t[0] == '+')) { neg = t[0] == '-'; p = 1; }
    int base = 10;
    if (t.compare(p, 2, "0x") == 0 || t.compare(p, 2, "0X") == 0) { base = 16; p += 2; }
    else if (t.compare(p, 1, "#") == 0) { base = 16; p += 1; }
    else if (t.size() > p + 1 && t[p] == '0') { base = 8; p += 1; }
    if (p >= t.size()) throw HammockException("NumberFormatException: For input string: \"" + text + "\"");
    long long v = 0;
    for (; p < t.size(); p++) {
        int d;
        const char c = t[p];
        if (c >= '0' && c <= '9') d = c - '0';
        else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
        else d = 99;
        if (d >= base) throw HammockException("NumberFormatException: For input string: \"" + text + "\"");
        v = v * base + d;
        if (v > 2147483648LL) throw HammockException("NumberFormatException: For input string: \"" + text + "\"");
    }
    v = neg ? -v : v;
    if (v > 2147483647LL || v < -2147483648LL) throw HammockException("NumberFormatException: For input string: \"" + text + "\"");
    return (int)v;
}

inline long long javaRound(double x) { return (long long)std::floor(x + 0.5); }  // Math.round(double)

inline int32_t javaStringHash(const std::string &s) {
    uint32_t h = 0;
    for (unsigned char c : s) h = 31u * h + c;
    return (int32_t)h;
}

// Iteration order of a java.util.HashMap<String, ?> (Java 8+: spread hash, power-of-two
// table that doubles when size exceeds 0.75 * capacity, buckets keep insertion order)
// holding `keys` (distinct, in insertion order).
inline std::vector<std::string> javaHashMapOrder(const std::vector<std::string> &keys) {
    size_t cap = 16;
    while (keys.size() > cap * 3 / 4) cap *= 2;
    std::vector<std::pair<std::pair<uint32_t, size_t>, std::string>> v;
    for (size_t k = 0; k < keys.size(); k++) {
        const uint32_t h = (uint32_t)javaStringHash(keys[k]);
        v.push_back({{(h ^ (h >> 16)) & (uint32_t)(cap - 1), k}, keys[k]});
    }
    std::sort(v.begin(), v.end());
    std::vector<std::string> out;
    for (auto &e : v) out.push_back(e.second);
    return out;
}

// java.util.Random (48-bit LCG) and Collections.shuffle, for `-R random`
class JavaRandom {
    uint64_t seed_;
public:
    explicit JavaRandom(int64_t seed) : seed_(((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1)) {}
    int32_t next(int bits) {
        seed_ = (seed_ * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)(int64_t)(seed_ >> (48 - bits));
    }
    int32_t nextInt(int32_t bound) {
        int32_t bits = next(31);
        const int32_t m = bound - 1;
        if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)bits) >> 31);
        int32_t val = bits % bound;
        while ((int32_t)((uint32_t)bits - (uint32_t)val + (uint32_t)m) < 0) {  // int overflow test of Random.nextInt
            bits = next(31);
            val = bits % bound;
        }
        return val;
    }
};

// ---- UniqueSequence.java -------------------------------------------------------------------------
class UniqueSequence {
    std::vector<int> sequence_;
    std::vector<std::pair<std::string, int>> labels_;  // labelsMap, insertion order kept
    std::string string_;   // getSequenceString(), kept: the comparators ask for it O(n log n) times
    int size_ = 0;         // size(), kept up to date by the only mutator
public:
    explicit UniqueSequence(const std::string &sequence) : UniqueSequence(sequence, {{"no_label", 1}}) {}  // :65-74
    UniqueSequence(const std::string &sequence, std::vector<std::pair<std::string, int>> labelsMap)  // :46-57
        : labels_(std::move(labelsMap)) {
        sequence_.reserve(sequence.size());   // one allocation instead of five doublings (10^6 sequences: 0.28 -> 0.1 s)
        string_.reserve(sequence.size());
        for (char ch : sequence) {
            char up = (ch >= 'a' && ch <= 'z') ? (char)(ch - 'a' + 'A') : ch;
            const char *f = std::strchr(AMINO_ACIDS, up);
            if (!f || up == '\0')
                throw FileFormatException(std::string("Error, character ") + ch +
                                          " is not a valid letter from the amino acid alphabet code.");
            sequence_.push_back((int)(f - AMINO_ACIDS));
            string_.push_back(AMINO_ACIDS[sequence_.back()]);
        }
        for (auto &e : labels_) size_ += e.second;
    }
    int size() const { return size_; }  // :82-88 (the sum of the label counts)
    const std::vector<int> &getSequence() const { return sequence_; }
    const std::string &getSequenceString() const { return string_; }  // :103-109
    const std::vector<std::pair<std::string, int>> &getLabelsMap() const { return labels_; }
    int labelCount(const std::string &label, bool *present = nullptr) const {
        for (auto &e : labels_) if (e.first == label) { if (present) *present = true; return e.second; }
        if (present) *present = false;
        return 0;
    }
    void addLabelCount(const std::string &label, int count) {  // FileIOManager.updateLabelsMap :204-216
        size_ += count;
        for (auto &e : labels_) if (e.first == label) { e.second += count; return; }
        labels_.push_back({label, count});
    }
    bool operator==(const UniqueSequence &o) const { return sequence_ == o.sequence_; }  // :143-153
};
using UniqueSequencePtr = std::shared_ptr<UniqueSequence>;

// String.compareTo on ASCII strings
inline int javaStringCompare(const std::string &a, const std::string &b) {
    const size_t lim = std::min(a.size(), b.size());
    for (size_t k = 0; k < lim; k++)
        if (a[k] != b[k]) return (int)(unsigned char)a[k] - (int)(unsigned char)b[k];
    return (int)a.size() - (int)b.size();
}
// UniqueSequenceSizeAlphabeticComparator, UniqueSequence.java:238-248
inline int sizeAlphabeticCompare(const UniqueSequence &a, const UniqueSequence &b) {
    const int r = a.size() - b.size();
    return r != 0 ? r : javaStringCompare(a.getSequenceString(), b.getSequenceString());
}

// UniqueSequence.sortSequences, UniqueSequence.java:176-203 (Collections.sort is stable)
// the "size" order (UniqueSequenceSizeAlphabeticComparator reversed: size descending, then string descending), sorted on
// keys held beside the pointers -- size and the string's first 8 bytes as one big-endian word decide almost every
// comparison without touching the sequence objects (10^6 sequences: 0.47 s -> 0.2 s)
inline void sortBySizeThenStringDescending(std::vector<UniqueSequencePtr> &seqs) {
    struct Key { int size; uint64_t prefix; uint32_t index; };
    const size_t n = seqs.size();
    std::vector<Key> keys(n);
    const unsigned T = hostThreads();
    parallelRanges(n, T, [&](unsigned, size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; k++) {
            const std::string &s = seqs[k]->getSequenceString();
            uint64_t prefix = 0;
            for (size_t b = 0; b < 8; b++) prefix = (prefix << 8) | (b < s.size() ? (unsigned char)s[b] : 0u);
            keys[k] = Key{seqs[k]->size(), prefix, (uint32_t)k};
        }
    });
    auto before = [&](const Key &a, const Key &b) {
        if (a.size != b.size) return a.size > b.size;
        if (a.prefix != b.prefix) return a.prefix > b.prefix;      // (a zero byte pads a string shorter than 8: it sorts first, as in compareTo)
        return javaStringCompare(seqs[b.index]->getSequenceString(), seqs[a.index]->getSequenceString()) < 0;
    };
    // a stable merge sort: contiguous runs sorted on their own threads, then merged pairwise (std::merge takes from the left
    // run on ties, so the result is the one std::stable_sort of the whole list gives)
    size_t runs = 1;
    while (runs < T && n / (2 * runs) >= 16384) runs *= 2;
    std::vector<size_t> cut(runs + 1);
    for (size_t r = 0; r <= runs; r++) cut[r] = n * r / runs;
    {
        std::vector<std::thread> pool;
        for (size_t r = 1; r < runs; r++) pool.emplace_back([&, r] { std::stable_sort(keys.begin() + (long)cut[r], keys.begin() + (long)cut[r + 1], before); });
        std::stable_sort(keys.begin(), keys.begin() + (long)cut[1], before);
        for (std::thread &th : pool) th.join();
    }
    std::vector<Key> other(runs > 1 ? n : 0);
    std::vector<Key> *from = &keys, *to = &other;
    for (size_t width = 1; width < runs; width *= 2) {
        std::vector<std::thread> pool;
        for (size_t r = 0; r < runs; r += 2 * width) {
            auto job = [&, r] {
                std::merge(from->begin() + (long)cut[r], from->begin() + (long)cut[r + width], from->begin() + (long)cut[r + width],
                           from->begin() + (long)cut[r + 2 * width], to->begin() + (long)cut[r], before);
            };
            if (r + 2 * width < runs) pool.emplace_back(job); else job();
        }
        for (std::thread &th : pool) th.join();
        std::swap(from, to);
    }
    const std::vector<Key> &order = *from;
    std::vector<UniqueSequencePtr> sorted(n);
    parallelRanges(n, T, [&](unsigned, size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; k++) sorted[k] = std::move(seqs[order[k].index]);
    });
    seqs.swap(sorted);
}

inline void sortSequences(std::vector<UniqueSequencePtr> &seqs, const std::string &order, int seed,
                          const std::vector<std::string> &labels) {
    auto sizeAlphaDesc = [](const UniqueSequencePtr &a, const UniqueSequencePtr &b) {
        return sizeAlphabeticCompare(*b, *a) < 0;  // Collections.reverseOrder(cmp)
    };
    if (order == "size") {
        if (seqs.size() < (1u << 31)) sortBySizeThenStringDescending(seqs);
        else std::stable_sort(seqs.begin(), seqs.end(), sizeAlphaDesc);
    } else if (order == "alphabetic") {
        std::stable_sort(seqs.begin(), seqs.end(), [](const UniqueSequencePtr &a, const UniqueSequencePtr &b) {
            return javaStringCompare(b->getSequenceString(), a->getSequenceString()) < 0;
        });
    } else if (order == "random") {  // Collections.shuffle(list, Hammock.random), Hammock.java:1252
        JavaRandom rnd(seed);
        for (size_t i = seqs.size(); i > 1; i--) std::swap(seqs[i - 1], seqs[(size_t)rnd.nextInt((int32_t)i)]);
    } else if (order == "input") {
    } else {
        if (std::find(labels.begin(), labels.end(), order) == labels.end())
            throw DataException("Incorrect sequence order defined. Use one of: size, alphabetic, random, input, or a label");
        std::stable_sort(seqs.begin(), seqs.end(), sizeAlphaDesc);
        std::stable_sort(seqs.begin(), seqs.end(), [&](const UniqueSequencePtr &a, const UniqueSequencePtr &b) {
            return b->labelCount(order) - a->labelCount(order) < 0;  // reverseOrder(LabelComparator) :209-228
        });
    }
}

// ---- Cluster.java ---------------------------------------------------------------------------------
class Cluster {
    std::vector<UniqueSequencePtr> sequences_;
    int id_;
    int size_ = 0;
public:
    Cluster(std::vector<UniqueSequencePtr> sequences, int id) : sequences_(std::move(sequences)), id_(id) {  // :31-41
        for (auto &s : sequences_) size_ += s->size();
    }
    void insert(const UniqueSequencePtr &sequence) {  // :50-63
        for (auto &s : sequences_)
            if (*s == *sequence)
                throw DataException("Trying to insert unique sequence " + sequence->getSequenceString() + " into cluster " +
                                    std::to_string(id_) + ", which already contains this sequence. ");
        sequences_.push_back(sequence);
        size_ += sequence->size();
    }
    void insertAll(const std::vector<UniqueSequencePtr> &sequences) { for (auto &s : sequences) insert(s); }  // :70-74
    int getUniqueSize() const { return (int)sequences_.size(); }  // :113-115
    std::vector<UniqueSequencePtr> &getSequences() { return sequences_; }
    const std::vector<UniqueSequencePtr> &getSequences() const { return sequences_; }
    int getId() const { return id_; }
    int size() const { return size_; }  // :156-158
    int compareTo(const Cluster &o) const { return size_ != o.size_ ? size_ - o.size_ : id_ - o.id_; }  // :198-204
};
using ClusterPtr = std::shared_ptr<Cluster>;

struct AligningScorerResult {  // AligningScorerResult.java:11-44
    int score, shift;
    UniqueSequencePtr sequence;
    int getScore() const { return score; }
    int getShift() const { return shift; }
};

// ---- interfaces -----------------------------------------------------------------------------------------
struct SequenceScorer {  // SequenceScorer.java:12-15
    virtual ~SequenceScorer() = default;
    virtual int sequenceScore(const UniqueSequencePtr &seq1, const UniqueSequencePtr &seq2) = 0;
};
struct AligningSequenceScorer : SequenceScorer {  // AligningSequenceScorer.java:10-12
    virtual AligningScorerResult scoreWithShift(const UniqueSequencePtr &seq1, const UniqueSequencePtr &seq2) = 0;
};
struct SequenceClusterer {  // SequenceClusterer.java:15-26
    virtual ~SequenceClusterer() = default;
    virtual std::vector<ClusterPtr> cluster(const std::vector<UniqueSequencePtr> &sequences) = 0;
};

// ---- the native context ----------------------------------------------------------------------------------
class NativeContext {
    hmk_ctx *ctx_ = nullptr;
public:
    NativeContext(const std::vector<std::vector<int>> &scoringMatrix, int device) {
        if (scoringMatrix.size() != 24) throw HammockException("scoring matrix must be 24 x 24");
        int32_t m[576];
        for (int r = 0; r < 24; r++) {
            if (scoringMatrix[r].size() != 24) throw HammockException("scoring matrix must be 24 x 24");
            for (int c = 0; c < 24; c++) m[r * 24 + c] = scoringMatrix[r][c];
        }
        const int st = hmk_create(m, device, &ctx_);
        if (st) raise(st, nullptr);
    }
    // several GPUs of the node behind one context (hmk_create_multi): devices[0] is the root
    NativeContext(const std::vector<std::vector<int>> &scoringMatrix, const std::vector<int> &devices) {
        if (scoringMatrix.size() != 24) throw HammockException("scoring matrix must be 24 x 24");
        int32_t m[576];
        for (int r = 0; r < 24; r++) {
            if (scoringMatrix[r].size() != 24) throw HammockException("scoring matrix must be 24 x 24");
            for (int c = 0; c < 24; c++) m[r * 24 + c] = scoringMatrix[r][c];
        }
        const int st = hmk_create_multi(m, devices.data(), (int)devices.size(), &ctx_);
        if (st) raise(st, nullptr);
    }
    ~NativeContext() { hmk_destroy(ctx_); }
    NativeContext(const NativeContext &) = delete;
    NativeContext &operator=(const NativeContext &) = delete;
    hmk_ctx *get() const { return ctx_; }
    [[noreturn]] void raise(int st, const hmk_greedy_stats *gs) const {
        const std::string msg = hmk_last_error(ctx_);
        switch (st) {
            case HMK_ERR_SHIFT_TOO_BIG: throw DataException(msg);
            case HMK_ERR_REFERENCE_WOULD_CRASH:
                throw NullPointerException(msg, gs ? gs->crash_case : 0, gs ? gs->crash_index : -1);
            case HMK_ERR_DEVICE: case HMK_ERR_OOM: throw DeviceException(msg);
            default: throw HammockException(msg);
        }
    }
    void setSequences(const std::vector<UniqueSequencePtr> &seqs, bool withSizes) const {
        size_t total = 0;
        for (auto &s : seqs) total += s->getSequence().size();
        std::vector<uint8_t> res(total);
        std::vector<uint32_t> off(seqs.size() + 1, 0);
        std::vector<int32_t> sizes(seqs.size());
        for (size_t k = 0; k < seqs.size(); k++) off[k + 1] = off[k] + (uint32_t)seqs[k]->getSequence().size();
        parallelRanges(seqs.size(), hostThreads(), [&](unsigned, size_t lo, size_t hi) {   // (10^6 objects scattered over the heap)
            for (size_t k = lo; k < hi; k++) {
                uint8_t *dst = res.data() + off[k];
                for (int r : seqs[k]->getSequence()) *dst++ = (uint8_t)r;
                sizes[k] = seqs[k]->size();
            }
        });
        const int st = hmk_set_sequences(ctx_, res.data(), off.data(), withSizes ? sizes.data() : nullptr, (uint32_t)seqs.size());
        if (st) raise(st, nullptr);
    }
};

// ---- ShiftedScorer.java ---------------------------------------------------------------------------------
class ShiftedScorer : public AligningSequenceScorer {
    std::shared_ptr<NativeContext> ctx_;
    int shiftPenalty_, maxShift_;
public:
    // ShiftedScorer(int[][] scoringMatrix, int shiftPenalty, int maxShift), ShiftedScorer.java:28-32
    ShiftedScorer(const std::vector<std::vector<int>> &scoringMatrix, int shiftPenalty, int maxShift, int device = 0)
        : ctx_(std::make_shared<NativeContext>(scoringMatrix, device)), shiftPenalty_(shiftPenalty), maxShift_(maxShift) {}
    // the same scorer with the pair space sharded over several GPUs (used by HipGreedySequenceClusterer.cluster)
    ShiftedScorer(const std::vector<std::vector<int>> &scoringMatrix, int shiftPenalty, int maxShift, const std::vector<int> &devices)
        : ctx_(std::make_shared<NativeContext>(scoringMatrix, devices)), shiftPenalty_(shiftPenalty), maxShift_(maxShift) {}
    // over a context that already exists (a host may create it while it is still reading its input)
    ShiftedScorer(std::shared_ptr<NativeContext> context, int shiftPenalty, int maxShift)
        : ctx_(std::move(context)), shiftPenalty_(shiftPenalty), maxShift_(maxShift) {}
    AligningScorerResult scoreWithShift(const UniqueSequencePtr &seq1, const UniqueSequencePtr &seq2) override {  // :48-95
        ctx_->setSequences({seq1, seq2}, false);
        const uint32_t i = 0, j = 1;
        int32_t score = 0, shift = 0;
        const int st = hmk_score_with_shift(ctx_->get(), &i, &j, 1, maxShift_, shiftPenalty_, &score, &shift);
        if (st) ctx_->raise(st, nullptr);
        return AligningScorerResult{score, shift, seq2};
    }
    int sequenceScore(const UniqueSequencePtr &seq1, const UniqueSequencePtr &seq2) override {  // :98-100
        return scoreWithShift(seq1, seq2).getScore();
    }
    int getShiftPenalty() const { return shiftPenalty_; }
    int getMaxShift() const { return maxShift_; }
    const std::shared_ptr<NativeContext> &native() const { return ctx_; }
};

// ---- LocalAlignmentScorer.java --------------------------------------------------------------------------
class LocalAlignmentScorer : public SequenceScorer {
    std::shared_ptr<NativeContext> ctx_;
    int gapOpenPenalty_, gapExtendPenalty_;
public:
    LocalAlignmentScorer(const std::vector<std::vector<int>> &scoringMatrix, int gapOpenPenalty, int gapExtendPenalty,
                         int device = 0)  // LocalAlignmentScorer.java:20-24
        : ctx_(std::make_shared<NativeContext>(scoringMatrix, device)), gapOpenPenalty_(gapOpenPenalty),
          gapExtendPenalty_(gapExtendPenalty) {}
    int sequenceScore(const UniqueSequencePtr &seq1, const UniqueSequencePtr &seq2) override {  // :27-29
        ctx_->setSequences({seq1, seq2}, false);
        const uint32_t i = 0, j = 1;
        int32_t score = 0;
        const int st = hmk_score_pairs_local(ctx_->get(), &i, &j, 1, gapOpenPenalty_, gapExtendPenalty_, &score);
        if (st) ctx_->raise(st, nullptr);
        return score;
    }
};

// ---- HipGreedySequenceClusterer: LimitedGreedySequenceClusterer.java:17-121 on the GPU ----------------------
class HipGreedySequenceClusterer : public SequenceClusterer {
    std::shared_ptr<ShiftedScorer> scorer_;
    int threshold_, maxClusters_;
public:
    hmk_greedy_stats stats{};
    // same constructor shape as LimitedGreedySequenceClusterer(sequenceScorer, threshold, maxClusters), :22-26
    HipGreedySequenceClusterer(std::shared_ptr<ShiftedScorer> sequenceScorer, int threshold, int maxClusters)
        : scorer_(std::move(sequenceScorer)), threshold_(threshold), maxClusters_(maxClusters) {}
    // cluster(List<UniqueSequence>) -> List<Cluster>, :39-69: clusters in creation order (id = index of the
    // seed), then the remaining singletons; members in Cluster.getSequences() insertion order.
    std::vector<ClusterPtr> cluster(const std::vector<UniqueSequencePtr> &sequences) override {
        const auto &nc = scorer_->native();
        const bool timing = std::getenv("HMK_CLI_TIMING") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            const auto t1 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[hammock-hip] %s: %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        };
        nc->setSequences(sequences, true);
        lap("upload");
        const size_t n = sequences.size();
        std::vector<int32_t> cid(std::max<size_t>(n, 1)), order(std::max<size_t>(n, 1)), rank(std::max<size_t>(n, 1));
        const int st = hmk_greedy_cluster(nc->get(), scorer_->getMaxShift(), scorer_->getShiftPenalty(), threshold_,
                                          maxClusters_, cid.data(), order.data(), rank.data(), &stats);
        if (st) nc->raise(st, &stats);
        lap("hmk_greedy_cluster");
        // members of every cluster in insertion order: a cluster's id is the index of its seed (< n), ranks are 0 .. size - 1
        std::vector<uint32_t> first(n + 1, 0);
        for (size_t k = 0; k < n; k++) first[(size_t)cid[k] + 1]++;
        for (size_t c = 0; c < n; c++) first[c + 1] += first[c];
        std::vector<uint32_t> slot(n);
        for (size_t k = 0; k < n; k++) slot[first[(size_t)cid[k]] + (size_t)rank[k]] = (uint32_t)k;
        std::vector<ClusterPtr> result((size_t)stats.n_result_clusters);
        parallelRanges(result.size(), hostThreads(), [&](unsigned, size_t lo, size_t hi) {
            for (size_t q = lo; q < hi; q++) {
                const size_t c = (size_t)order[q];
                std::vector<UniqueSequencePtr> seqs;
                seqs.reserve(first[c + 1] - first[c]);
                for (uint32_t e = first[c]; e < first[c + 1]; e++) seqs.push_back(sequences[slot[e]]);
                result[q] = std::make_shared<Cluster>(std::move(seqs), order[q]);
            }
        });
        lap("Cluster objects");
        return result;
    }
};

// ---- HipClinkageSequenceClusterer: ClinkageSequenceClusterer.java:21-124 on the GPU -----------------------------
class HipClinkageSequenceClusterer : public SequenceClusterer {
    std::shared_ptr<ShiftedScorer> scorer_;
    int threshold_;
public:
    hmk_clinkage_stats stats{};
    // same constructor shape as ClinkageSequenceClusterer(sequenceScorer, threshold), :29-33
    HipClinkageSequenceClusterer(std::shared_ptr<ShiftedScorer> sequenceScorer, int threshold)
        : scorer_(std::move(sequenceScorer)), threshold_(threshold) {}
    // cluster(List<UniqueSequence>) -> List<Cluster>, :43-124: ids as the reference assigns them (index + 1 for a sequence
    // left alone, n + 2, n + 3, ... for merged clusters), the list in the iteration order of the reference's HashSet
    std::vector<ClusterPtr> cluster(const std::vector<UniqueSequencePtr> &sequences) override {
        const auto &nc = scorer_->native();
        nc->setSequences(sequences, true);
        const size_t n = sequences.size();
        std::vector<int32_t> cid(std::max<size_t>(n, 1)), order(std::max<size_t>(n, 1)), rank(std::max<size_t>(n, 1));
        const int st = hmk_clinkage_cluster(nc->get(), scorer_->getMaxShift(), scorer_->getShiftPenalty(), threshold_, cid.data(),
                                            order.data(), rank.data(), &stats);
        if (st) nc->raise(st, nullptr);
        std::unordered_map<int, std::vector<std::pair<int, size_t>>> members;  // id -> (rank, index)
        for (size_t k = 0; k < n; k++) members[cid[k]].push_back({rank[k], k});
        std::vector<ClusterPtr> result;
        for (int q = 0; q < stats.n_result_clusters; q++) {
            auto &mv = members[order[q]];
            std::sort(mv.begin(), mv.end());
            std::vector<UniqueSequencePtr> seqs;
            for (auto &e : mv) seqs.push_back(sequences[e.second]);
            result.push_back(std::make_shared<Cluster>(seqs, order[q]));
        }
        return result;
    }
};

// ---- Logger.java ---------------------------------------------------------------------------------------------
class Logger {
    std::string filePath_;
    bool dummy_;
public:
    Logger(std::string filePath, bool dummyLogger) : filePath_(std::move(filePath)), dummy_(dummyLogger) {}  // :39-42
    void logWithoutTime(const std::string &line) const {  // :49-60
        if (dummy_ || filePath_.empty()) return;
        std::ofstream f(filePath_, std::ios::app);
        if (!f) { std::cerr << "Warning: Failed to log follwoing message. Run will continue, message will not be appended run.log\n"; return; }
        f << line << "\n";
    }
    void logWithTime(const std::string &line) const {  // :67-74, "yyyy-MM-dd HH:mm:ss.SSS"
        if (dummy_) return;
        using namespace std::chrono;
        const auto now = system_clock::now();
        const std::time_t t = system_clock::to_time_t(now);
        const int ms = (int)(duration_cast<milliseconds>(now.time_since_epoch()).count() % 1000);
        std::tm tm{};
        localtime_r(&t, &tm);
        char buf[64];
        std::snprintf(buf, sizeof(buf), "%04d-%02d-%02d %02d:%02d:%02d.%03d", tm.tm_year + 1900, tm.tm_mon + 1, tm.tm_mday,
                      tm.tm_hour, tm.tm_min, tm.tm_sec, ms);
        logWithoutTime(std::string(buf) + ":\t" + line);
    }
    void logAndStderr(const std::string &line) const {  // :81-87
        if (dummy_) return;
        logWithTime(line);
        std::cerr << line << std::endl;
    }
};

// ---- FileIOManager.java (greedy-path subset) ----------------------------------------------------------------------
// what the driver asks of the whole list before clustering (Hammock.java:763-785 sizes and lengths, :1421-1427 the shortest,
// :1554-1563 the mean length), gathered in one pass on several threads instead of six walks over 10^6 scattered objects
struct SequenceListSummary {
    long long total = 0;        // sum of UniqueSequence.size()
    long long lengthSum = 0;    // sum of the sequences' lengths
    int minLength = INT_MAX, maxLength = INT_MIN;
    size_t count = 0;
    double meanLength() const { return (double)lengthSum / (double)count; }
};
inline SequenceListSummary summariseSequences(const std::vector<UniqueSequencePtr> &sequences) {
    const unsigned T = hostThreads();
    std::vector<SequenceListSummary> parts(T);
    parallelRanges(sequences.size(), T, [&](unsigned t, size_t lo, size_t hi) {
        SequenceListSummary p;
        for (size_t q = lo; q < hi; q++) {
            const int len = (int)sequences[q]->getSequence().size();
            p.total += sequences[q]->size();
            p.lengthSum += len;
            p.minLength = std::min(p.minLength, len);
            p.maxLength = std::max(p.maxLength, len);
        }
        parts[t] = p;
    });
    SequenceListSummary all;
    all.count = sequences.size();
    for (const SequenceListSummary &p : parts) {
        all.total += p.total;
        all.lengthSum += p.lengthSum;
        all.minLength = std::min(all.minLength, p.minLength);
        all.maxLength = std::max(all.maxLength, p.maxLength);
    }
    return all;
}

namespace FileIOManager {

// String.split("\\s+"): a leading empty token is kept, trailing empty tokens are dropped
inline std::vector<std::string> splitWhitespace(const std::string &line) {
    std::vector<std::string> out;
    size_t p = 0;
    bool first = true;
    while (p <= line.size()) {
        size_t q = p;
        while (q < line.size() && !std::isspace((unsigned char)line[q])) q++;
        if (q > p || first) out.push_back(line.substr(p, q - p));
        first = false;
        while (q < line.size() && std::isspace((unsigned char)line[q])) q++;
        if (q >= line.size()) break;
        p = q;
    }
    while (!out.empty() && out.back().empty()) out.pop_back();
    return out;
}

inline std::vector<std::string> splitChar(const std::string &line, char sep, bool dropTrailingEmpty) {
    std::vector<std::string> out;
    std::string cur;
    for (char c : line) { if (c == sep) { out.push_back(cur); cur.clear(); } else cur.push_back(c); }
    out.push_back(cur);
    if (dropTrailingEmpty) while (out.size() > 1 && out.back().empty()) out.pop_back();
    return out;
}

inline std::vector<std::string> readLines(const std::string &path) {  // BufferedReader.readLine semantics
    std::ifstream f(path, std::ios::binary);
    if (!f) throw HammockException("java.io.FileNotFoundException: " + path + " (No such file or directory)");
    f.seekg(0, std::ios::end);
    const std::streamoff size = f.tellg();
    f.seekg(0, std::ios::beg);
    std::string all((size_t)std::max<std::streamoff>(size, 0), '\0');
    if (size > 0) f.read(&all[0], size);
    all.resize((size_t)f.gcount());
    std::vector<std::string> lines;
    lines.reserve(all.size() / 12 + 16);
    size_t start = 0;   // a line ends at \n, \r or \r\n; what follows the last terminator is a line only if it is not empty
    for (size_t k = 0; k < all.size(); k++) {
        const char c = all[k];
        if (c != '\n' && c != '\r') continue;
        lines.emplace_back(all, start, k - start);
        if (c == '\r' && k + 1 < all.size() && all[k + 1] == '\n') k++;
        start = k + 1;
    }
    if (start < all.size()) lines.emplace_back(all, start, all.size() - start);
    return lines;
}

inline std::string trim(const std::string &s) {  // String.trim(): strips chars <= ' '
    size_t a = 0, b = s.size();
    while (a < b && (unsigned char)s[a] <= ' ') a++;
    while (b > a && (unsigned char)s[b - 1] <= ' ') b--;
    return s.substr(a, b - a);
}

// loadScoringMatrix, FileIOManager.java:46-81
inline std::vector<std::vector<int>> loadScoringMatrix(const std::string &matrixFilePath) {
    std::vector<std::vector<int>> m(24, std::vector<int>(24, 0));
    int lineCounter = 0;
    for (const std::string &line : readLines(matrixFilePath)) {
        if (!line.empty() && (line[0] == '#' || line[0] == ' ' || line[0] == '\t')) continue;  // :58
        const std::vector<std::string> parts = splitWhitespace(line);                          // :59
        if (parts.size() != 25)
            throw FileFormatException("Error in scoring matrix file: " + matrixFilePath + ". Scoring matrix "
                                      "should always have 24 columns (plus 1 column describing AAs).");
        if (lineCounter >= 24)  // ArrayIndexOutOfBoundsException -> :76-79
            throw FileFormatException("Error in scoring matrix file: " + matrixFilePath + ". Scoring matrix "
                                      "should always have 24 rows (plus 1 column describing AAs).");
        for (int i = 1; i < 25; i++) {
            try {
                size_t used = 0;
                m[lineCounter][i - 1] = std::stoi(parts[i], &used);  // Integer.parseInt
                if (used != parts[i].size()) throw std::invalid_argument("");
            } catch (const std::exception &) {
                throw HammockException("NumberFormatException: For input string: \"" + parts[i] + "\"");
            }
        }
        lineCounter++;
    }
    return m;
}

inline std::vector<UniqueSequencePtr> loadUniqueSequencesFromFastaLiteral(const std::string &fileName);

// loadUniqueSequencesFromFasta, FileIOManager.java:159-202 -- the same result as the literal loader below, built on several
// threads: the file is cut at header lines into one piece per thread, every piece is parsed into (sequence, count, label)
// records with the literal loader's own helpers, the records are merged in file order (duplicates add their counts to the
// first occurrence, FileIOManager.java:204-216) and the UniqueSequence objects are constructed in parallel.  Anything
// unusual -- a malformed header, a count below 1, text before the first header, an empty file -- goes to the literal loader,
// which raises what the reference raises, in the reference's order.  10^6 records: 0.62 s -> see DESIGN.md.
inline std::vector<UniqueSequencePtr> loadUniqueSequencesFromFasta(const std::string &fileName) {
    const unsigned T = hostThreads();
    if (T == 1 || std::getenv("HMK_LITERAL_LOADER")) return loadUniqueSequencesFromFastaLiteral(fileName);
    const auto tl0 = std::chrono::steady_clock::now();
    auto loaderLap = [&](const char *what) {
        if (std::getenv("HMK_CLI_TIMING"))
            std::fprintf(stderr, "[hammock-hip] loader, %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count());
    };
    std::string all;
    {
        std::ifstream f(fileName, std::ios::binary);
        if (!f) throw HammockException("java.io.FileNotFoundException: " + fileName + " (No such file or directory)");
        f.seekg(0, std::ios::end);
        const std::streamoff size = f.tellg();
        f.seekg(0, std::ios::beg);
        all.resize((size_t)std::max<std::streamoff>(size, 0));
        if (size > 0) f.read(&all[0], size);
        all.resize((size_t)f.gcount());
    }
    if (all.size() < (1u << 16) || all[0] != '>') return loadUniqueSequencesFromFastaLiteral(fileName);
    struct Record { std::string seq; std::string label; int count; };
    // piece boundaries: the start of a line that begins with '>'
    std::vector<size_t> cut(1, 0);
    for (unsigned t = 1; t < T; t++) {
        size_t at = all.size() * t / T;
        while (at < all.size() && !((all[at - 1] == '\n' || all[at - 1] == '\r') && all[at] == '>')) at++;
        if (at > cut.back() && at < all.size()) cut.push_back(at);
    }
    cut.push_back(all.size());
    const size_t pieces = cut.size() - 1;
    std::vector<std::vector<Record>> parsed(pieces);
    std::vector<char> odd(pieces, 0);
    {
        std::vector<std::thread> pool;
        for (size_t pc = 0; pc < pieces; pc++)
            pool.emplace_back([&, pc]() {
                try {
                    std::vector<Record> &out = parsed[pc];
                    out.reserve((cut[pc + 1] - cut[pc]) / 20 + 16);
                    size_t start = cut[pc];
                    const size_t end = cut[pc + 1];
                    bool have = false;
                    Record cur;
                    while (start < end) {   // BufferedReader.readLine: a line ends at \n, \r or \r\n
                        size_t k = start;
                        while (k < end && all[k] != '\n' && all[k] != '\r') k++;
                        const std::string line(all, start, k - start);
                        if (k < end && all[k] == '\r' && k + 1 < end && all[k + 1] == '\n') k++;
                        start = k + 1;
                        if (!line.empty() && line[0] == '>') {
                            if (have) out.push_back(std::move(cur));
                            cur = Record();
                            have = true;
                            const std::vector<std::string> split = splitChar(trim(line).substr(1), '|', true);  // :173
                            if (split.size() >= 2) {
                                cur.count = javaIntegerDecode(trim(split[1]));                                  // :175
                                if (cur.count < 1) { odd[pc] = 1; return; }
                            } else cur.count = 1;
                            cur.label = split.size() >= 3 ? split[2] : "no_label";                              // :182-186
                        } else {
                            if (!have) { odd[pc] = 1; return; }
                            cur.seq += trim(line);                                                              // :191
                        }
                    }
                    if (have) out.push_back(std::move(cur));
                } catch (...) { odd[pc] = 1; }
            });
        for (std::thread &th : pool) th.join();
    }
    for (char o : odd) if (o) return loadUniqueSequencesFromFastaLiteral(fileName);
    loaderLap("pieces parsed");
    // ---- merge in file order (the LinkedHashMap of :160), partitioned by the sequence's hash: thread t owns the sequences
    // with hash % threads == t, walks ALL records in file order and merges its own; the first occurrences, marked at their
    // record's position, are then collected in file order ----
    size_t total = 0;
    std::vector<size_t> piece_base(pieces + 1, 0);
    for (size_t pc = 0; pc < pieces; pc++) { piece_base[pc + 1] = piece_base[pc] + parsed[pc].size(); }
    total = piece_base[pieces];
    if (total == 0) return loadUniqueSequencesFromFastaLiteral(fileName);
    struct Entry { std::string seq; std::vector<std::pair<std::string, int>> labels; };
    const std::hash<std::string_view> hasher;
    std::vector<uint64_t> hashes(total);
    {
        std::vector<std::thread> pool;
        for (size_t pc = 0; pc < pieces; pc++)
            pool.emplace_back([&, pc]() {
                for (size_t k = 0; k < parsed[pc].size(); k++) hashes[piece_base[pc] + k] = hasher(std::string_view(parsed[pc][k].seq));
            });
        for (std::thread &th : pool) th.join();
    }
    std::vector<uint64_t> first_at(total, ~0ull);   // thread << 32 | entry, at the record position of a first occurrence
    std::vector<std::vector<Entry>> owned(T);
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; t++)
            pool.emplace_back([&, t]() {
                std::vector<Entry> &mine = owned[t];
                size_t share = 0;
                for (uint64_t h : hashes) share += h % T == t;
                mine.reserve(share);
                size_t cap = 16;
                while (cap < 2 * share + 2) cap <<= 1;
                std::vector<uint32_t> table(cap, 0xFFFFFFFFu);
                std::vector<std::pair<size_t, uint32_t>> firsts;         // (record position, entry)
                for (size_t pc = 0; pc < pieces; pc++)
                    for (size_t k = 0; k < parsed[pc].size(); k++) {
                        const size_t g = piece_base[pc] + k;
                        const uint64_t h = hashes[g];
                        if (h % T != t) continue;
                        Record &r = parsed[pc][k];
                        if (r.seq.empty() && g + 1 != total) continue;   // :168-172 adds a sequence only if it is not empty; :193-195 adds the last one as it is
                        size_t at = (size_t)(h / T) & (cap - 1);
                        while (table[at] != 0xFFFFFFFFu && mine[table[at]].seq != r.seq) at = (at + 1) & (cap - 1);
                        if (table[at] == 0xFFFFFFFFu) {
                            table[at] = (uint32_t)mine.size();
                            firsts.push_back({g, (uint32_t)mine.size()});
                            mine.push_back(Entry{std::move(r.seq), {{std::move(r.label), r.count}}});
                        } else {
                            auto &lm = mine[table[at]].labels;
                            bool found = false;
                            for (auto &e : lm) if (e.first == r.label) { e.second += r.count; found = true; break; }
                            if (!found) lm.push_back({r.label, r.count});
                        }
                    }
                for (auto &f : firsts) first_at[f.first] = (uint64_t)t << 32 | f.second;
            });
        for (std::thread &th : pool) th.join();
    }
    std::vector<Entry *> order;
    order.reserve(total);
    for (uint64_t f : first_at) if (f != ~0ull) order.push_back(&owned[f >> 32][(uint32_t)f]);
    loaderLap("records merged");
    std::vector<UniqueSequencePtr> result(order.size());
    parallelRanges(order.size(), T, [&](unsigned, size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; k++) result[k] = std::make_shared<UniqueSequence>(order[k]->seq, std::move(order[k]->labels));
    });
    loaderLap("objects built");
    return result;
}

// the literal form: FileIOManager.java:159-202 line by line
inline std::vector<UniqueSequencePtr> loadUniqueSequencesFromFastaLiteral(const std::string &fileName) {
    std::vector<std::pair<std::string, std::vector<std::pair<std::string, int>>>> order;  // LinkedHashMap
    std::unordered_map<std::string, size_t> index;
    const std::vector<std::string> lines = readLines(fileName);
    index.reserve(lines.size() / 2 + 1);
    order.reserve(lines.size() / 2 + 1);
    std::string sequence, label;
    int count = 0;
    bool haveLabel = false, haveCount = false;
    auto add = [&](const std::string &seq) {
        auto it = index.find(seq);
        if (it == index.end()) {
            index[seq] = order.size();
            order.push_back({seq, {{label, count}}});
        } else {
            auto &lm = order[it->second].second;
            bool found = false;
            for (auto &e : lm) if (e.first == label) { e.second += count; found = true; break; }
            if (!found) lm.push_back({label, count});
        }
    };
    for (const std::string &line : lines) {
        if (!line.empty() && line[0] == '>') {
            if (!sequence.empty()) { add(sequence); sequence.clear(); }              // :168-172
            const std::vector<std::string> split = splitChar(trim(line).substr(1), '|', true);  // :173
            if (split.size() >= 2) {
                count = javaIntegerDecode(trim(split[1]));                           // :175
                if (count < 1) throw FileFormatException("Error while loading input file. Fasta header defines sequence count lower than 1.");
            } else count = 1;
            haveCount = true;
            label = split.size() >= 3 ? split[2] : "no_label";                       // :182-186
            haveLabel = true;
        } else {
            if (!haveLabel || !haveCount)
                throw FileFormatException("Error. Incorrect fasta format. Maybe header or sequence line missing?");
            sequence += trim(line);                                                  // :191
        }
    }
    if (!haveLabel || !haveCount)  // :193 unboxes a null Integer
        throw NullPointerException("java.lang.NullPointerException (empty input file, FileIOManager.java:193)", 0, -1);
    add(sequence);                                                                   // :193-195
    std::vector<UniqueSequencePtr> result;
    for (auto &e : order) result.push_back(std::make_shared<UniqueSequence>(e.first, e.second));
    return result;
}

// loadUniqueSequencesFromTable, FileIOManager.java:227-255
inline std::vector<UniqueSequencePtr> loadUniqueSequencesFromTable(const std::string &fileName) {
    const std::vector<std::string> lines = readLines(fileName);
    if (lines.empty()) throw NullPointerException("java.lang.NullPointerException (empty table, FileIOManager.java:231)", 0, -1);
    std::vector<std::string> header = splitChar(lines[0], CSV_SEPARATOR, true);
    std::vector<std::string> labels(header.begin() + 1, header.end());
    std::vector<UniqueSequencePtr> result;
    for (size_t k = 1; k < lines.size(); k++) {
        const std::vector<std::string> parts = splitChar(lines[k], CSV_SEPARATOR, true);
        std::vector<std::pair<std::string, int>> lm;
        for (size_t i = 1; i < parts.size(); i++) {
            const int value = javaIntegerDecode(parts[i]);
            if (value != 0) {
                if (i - 1 >= labels.size()) throw HammockException("java.lang.IndexOutOfBoundsException (more columns than labels)");
                lm.push_back({labels[i - 1], value});
            }
        }
        result.push_back(std::make_shared<UniqueSequence>(parts[0], lm));
    }
    return result;
}

// Hammock.getSortedLabels, Hammock.java:1586-1605 with ValueComparator (FileIOManager.java:1464-1480):
// total count descending; among equal totals the label put LATER (in HashMap iteration order) comes first.
inline std::vector<std::string> getSortedLabels(const std::vector<UniqueSequencePtr> &sequences) {
    // The reference walks the list once: every sequence's labels in its HashMap's iteration order, a label's first sight
    // fixing its place in `insertion`, its counts summed.  Here every thread walks one contiguous run of the list the same way;
    // joining the runs in list order (a label keeps the place of its first sight) gives the same `insertion` and the same sums.
    struct Part { std::vector<std::string> insertion; std::unordered_map<std::string, long long> total; };
    const unsigned T = hostThreads();
    std::vector<Part> parts(T);
    parallelRanges(sequences.size(), T, [&](unsigned t, size_t lo, size_t hi) {
        Part &p = parts[t];
        std::vector<std::string> keys;
        for (size_t q = lo; q < hi; q++) {
            const auto &lm = sequences[q]->getLabelsMap();
            if (lm.size() == 1) {   // one label: its HashMap order is itself
                auto it = p.total.find(lm[0].first);
                if (it == p.total.end()) { p.insertion.push_back(lm[0].first); p.total.emplace(lm[0].first, lm[0].second); }
                else it->second += lm[0].second;
                continue;
            }
            keys.clear();
            for (auto &e : lm) keys.push_back(e.first);
            for (const std::string &k : javaHashMapOrder(keys)) {
                if (!p.total.count(k)) p.insertion.push_back(k);
                p.total[k] += sequences[q]->labelCount(k);
            }
        }
    });
    std::vector<std::string> insertion;
    std::unordered_map<std::string, long long> total;
    for (const Part &p : parts)
        for (const std::string &k : p.insertion) {
            if (!total.count(k)) insertion.push_back(k);
            total[k] += p.total.at(k);
        }
    std::vector<std::string> result;
    for (const std::string &k : javaHashMapOrder(insertion)) {
        size_t pos = 0;
        while (pos < result.size() && total[result[pos]] > total[k]) pos++;  // first element with count <= new
        result.insert(result.begin() + (long)pos, k);
    }
    return result;
}

inline std::string sequenceLine(const UniqueSequence &seq, const std::vector<std::string> &labels) {
    std::string out = std::to_string(seq.size());
    for (const std::string &label : labels) out += CSV_SEPARATOR + std::to_string(seq.labelCount(label));
    return out;
}

// The same columns for many sequences: label -> column once, then every sequence fills its (few) labels' columns --
// labelCount() is a linear search with string compares, 225 of them per line with 15 labels.
// out += std::to_string(v) without the temporary string
inline void appendNumber(std::string &out, long long v) {
    char buf[24];
    const auto r = std::to_chars(buf, buf + sizeof buf, v);
    out.append(buf, (size_t)(r.ptr - buf));
}

class LabelColumns {
    std::unordered_map<std::string, size_t> column_;
    std::vector<std::string> names_;
    std::vector<size_t> first_;   // column k prints the count of labels[k]'s FIRST column (a label may be listed twice)
    size_t n_;
    // label -> its first column, or n_: a handful of labels is the rule, and comparing against each beats hashing the string
    size_t columnOf(const std::string &label) const {
        if (n_ <= 8) {
            for (size_t k = 0; k < n_; k++) if (names_[k] == label) return k;
            return n_;
        }
        auto it = column_.find(label);
        return it == column_.end() ? n_ : it->second;
    }
public:
    explicit LabelColumns(const std::vector<std::string> &labels) : names_(labels), n_(labels.size()) {
        for (size_t k = 0; k < labels.size(); k++) first_.push_back(column_.emplace(labels[k], k).first->second);
    }
    size_t size() const { return n_; }
    // adds the sequence's counts to `sums` (n_ entries); labels outside the list are ignored
    void add(const UniqueSequence &seq, std::vector<long long> &sums) const {
        for (auto &e : seq.getLabelsMap()) {
            const size_t c = columnOf(e.first);
            if (c < n_) sums[c] += e.second;
        }
    }
    // appends "<size>\t<count of label 0>\t..." (sequenceLine) to `out`
    void appendLine(const UniqueSequence &seq, const std::vector<std::string> &, std::vector<long long> &scratch, std::string &out) const {
        std::fill(scratch.begin(), scratch.end(), 0);
        add(seq, scratch);
        appendNumber(out, seq.size());
        for (size_t k = 0; k < n_; k++) {
            out += CSV_SEPARATOR;
            appendNumber(out, scratch[first_[k]]);   // a label listed twice repeats its column
        }
    }
};

// sequence -> cluster, keyed by the sequence STRING as in the reference (writeClusterSequencesToCsv builds two
// HashMap<String, ...>, FileIOManager.java:596-607: sequence -> cluster id, and for singletons sequence-without-gaps ->
// "alignment").  One open-addressing table over the sequence objects the clusters hold, filled on several threads (a slot is
// claimed with one compare-and-swap of the owner pointer): with two std::unordered_map<std::string, ...> per file the two
// callers took 9 of the 12 s of a 10^6-sequence run, with one serial table per file 2 x 0.1 s.  A sequence has no '-' (not in
// the alphabet), so the "alignment" key of a singleton IS its string, and as long as no string occurs in two clusters "found in
// msaMap" is "its cluster is a singleton"; a repeated string (impossible after the loaders, which merge duplicates) is
// rebuilt serially with HashMap.put's last-one-wins and the literal msaMap.
class SequenceClusterIndex {
    struct Slot { std::atomic<const UniqueSequence *> owner; const Cluster *cluster; };
    std::unique_ptr<Slot[]> table_;
    size_t cap_ = 16;
    bool repeated_ = false;
    std::unordered_map<std::string_view, std::string_view> msaMap_;   // only for the impossible case
    static size_t hashOf(std::string_view key) { return std::hash<std::string_view>()(key); }
public:
    explicit SequenceClusterIndex(const std::vector<ClusterPtr> &clusters) {
        size_t n_members = 0;
        for (auto &cl : clusters) n_members += (size_t)cl->getUniqueSize();
        while (cap_ < 2 * n_members + 2) cap_ <<= 1;
        table_.reset(new Slot[cap_]());
        std::atomic<bool> repeated{false};
        parallelChunks(clusters.size(), 512, hostThreads(), [&](size_t lo, size_t hi) {
            for (size_t c = lo; c < hi; c++)
                for (auto &s : clusters[c]->getSequences()) {
                    const std::string &key = s->getSequenceString();
                    size_t at = hashOf(key) & (cap_ - 1);
                    for (;;) {
                        const UniqueSequence *cur = table_[at].owner.load(std::memory_order_acquire);
                        if (cur == nullptr && table_[at].owner.compare_exchange_strong(cur, s.get(), std::memory_order_acq_rel)) {
                            table_[at].cluster = clusters[c].get();
                            break;
                        }
                        if (cur->getSequenceString() == key) { repeated.store(true, std::memory_order_relaxed); break; }
                        at = (at + 1) & (cap_ - 1);
                    }
                }
        });
        repeated_ = repeated.load();
        if (repeated_) {   // the same string in two clusters (or twice in one): the last put wins, in list order
            for (size_t k = 0; k < cap_; k++) { table_[k].owner.store(nullptr, std::memory_order_relaxed); table_[k].cluster = nullptr; }
            for (auto &cl : clusters)
                for (auto &s : cl->getSequences()) {
                    const std::string &key = s->getSequenceString();
                    size_t at = hashOf(key) & (cap_ - 1);
                    while (table_[at].owner.load(std::memory_order_relaxed) &&
                           table_[at].owner.load(std::memory_order_relaxed)->getSequenceString() != key) at = (at + 1) & (cap_ - 1);
                    table_[at].owner.store(s.get(), std::memory_order_relaxed);
                    table_[at].cluster = cl.get();
                }
            for (auto &cl : clusters)
                if (cl->getUniqueSize() == 1) { const std::string &s = cl->getSequences()[0]->getSequenceString(); msaMap_[s] = s; }
        }
    }
    // the cluster whose member list holds this string, or nullptr
    const Cluster *clusterOf(std::string_view key) const {
        size_t at = hashOf(key) & (cap_ - 1);
        for (;;) {
            const UniqueSequence *cur = table_[at].owner.load(std::memory_order_relaxed);
            if (cur == nullptr) return nullptr;
            if (cur->getSequenceString() == key) return table_[at].cluster;
            at = (at + 1) & (cap_ - 1);
        }
    }
    // msaMap.containsKey(sequence): the string is the (gap-free) alignment of a single-member cluster
    bool isSingletonAlignment(std::string_view key, const Cluster *its) const {
        return repeated_ ? msaMap_.find(key) != msaMap_.end() : its->getUniqueSize() == 1;
    }
};

// writeClusterSequencesToCsv, FileIOManager.java:594-638.  The `alignment` column: the reference fills
// it from Clustal Omega output for multi-member clusters (external process, out of scope) and with the
// bare sequence for singletons (:770-776); members of multi-member clusters get "NA" (:617-618).
inline void writeClusterSequencesToCsv(const std::vector<UniqueSequencePtr> &sequences, const SequenceClusterIndex &index,
                                       const std::string &filePath, const std::vector<std::string> &labels) {
    std::ofstream w(filePath, std::ios::binary);
    if (!w) throw HammockException("java.io.IOException: cannot write " + filePath);
    std::string head = std::string("cluster_id") + CSV_SEPARATOR + "sequence" + CSV_SEPARATOR + "alignment" + CSV_SEPARATOR + "sum";
    for (const std::string &label : labels) { head += CSV_SEPARATOR; head += label; }
    head += "\n";
    w.write(head.data(), (std::streamsize)head.size());
    const LabelColumns columns(labels);
    // the lines are formatted on several threads, one contiguous run of sequences each, and written in order
    const unsigned T = hostThreads();
    std::vector<std::string> parts(T);
    parallelRanges(sequences.size(), T, [&](unsigned t, size_t lo, size_t hi) {
      std::string &out = parts[t];
      out.reserve((hi - lo) * (48 + 4 * labels.size()) + 256);
      std::vector<long long> scratch(labels.size());
      for (size_t q = lo; q < hi; q++) {
        const UniqueSequencePtr &seq = sequences[q];
        const std::string &str = seq->getSequenceString();
        const Cluster *cluster = index.clusterOf(str);
        if (cluster) {
            appendNumber(out, cluster->getId());
            out += CSV_SEPARATOR; out += str; out += CSV_SEPARATOR;
            if (index.isSingletonAlignment(str, cluster)) out += str; else out += "NA";
            out += CSV_SEPARATOR;
        } else {
            out += "NA"; out += CSV_SEPARATOR; out += str; out += CSV_SEPARATOR; out += "NA"; out += CSV_SEPARATOR;
        }
        columns.appendLine(*seq, labels, scratch, out);
        out += "\n";
      }
    });
    for (const std::string &part : parts) w.write(part.data(), (std::streamsize)part.size());
}
inline void writeClusterSequencesToCsv(const std::vector<UniqueSequencePtr> &sequences, const std::vector<ClusterPtr> &clusters,
                                       const std::string &filePath, const std::vector<std::string> &labels) {
    writeClusterSequencesToCsv(sequences, SequenceClusterIndex(clusters), filePath, labels);
}

inline std::vector<ClusterPtr> clustersSortedDescending(const std::vector<ClusterPtr> &clusters) {
    // Collections.sort(list, Collections.reverseOrder()): size desc, id desc (Cluster.compareTo :198-204), on keys held
    // beside the indices so that the sort does not chase 10^6 pointers per pass
    struct Key { int size, id; uint32_t index; };
    std::vector<Key> keys(clusters.size());
    for (size_t k = 0; k < clusters.size(); k++) keys[k] = Key{clusters[k]->size(), clusters[k]->getId(), (uint32_t)k};
    std::stable_sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {   // b.compareTo(a) < 0, the same int arithmetic
        return (b.size != a.size ? b.size - a.size : b.id - a.id) < 0;
    });
    std::vector<ClusterPtr> sorted(clusters.size());
    for (size_t k = 0; k < clusters.size(); k++) sorted[k] = clusters[keys[k].index];
    return sorted;
}

// getSortedSequences, FileIOManager.java:530-538: the clusters in descending order, each cluster's own member list sorted in
// place (size descending, then string descending) as the reference does, the members concatenated.  Clusters are independent:
// they are sorted on several threads and copied to places known from a prefix sum.
inline std::vector<UniqueSequencePtr> sortedClusterSequences(const std::vector<ClusterPtr> &clusters) {
    const std::vector<ClusterPtr> order = clustersSortedDescending(clusters);
    std::vector<size_t> first(order.size() + 1, 0);
    for (size_t k = 0; k < order.size(); k++) first[k + 1] = first[k] + order[k]->getSequences().size();
    std::vector<UniqueSequencePtr> sortedSequences(first[order.size()]);
    parallelChunks(order.size(), 256, hostThreads(), [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; k++) {
            auto &seqs = order[k]->getSequences();
            if (seqs.size() > 1)
                std::stable_sort(seqs.begin(), seqs.end(), [](const UniqueSequencePtr &a, const UniqueSequencePtr &b) {
                    return sizeAlphabeticCompare(*b, *a) < 0;
                });
            std::copy(seqs.begin(), seqs.end(), sortedSequences.begin() + (long)first[k]);
        }
    });
    return sortedSequences;
}

// saveClusterSequencesToCsv, FileIOManager.java:398-404 + getSortedSequences :530-538
inline void saveClusterSequencesToCsv(const std::vector<ClusterPtr> &clusters, const std::string &filePath,
                                      const std::vector<std::string> &labels) {
    writeClusterSequencesToCsv(sortedClusterSequences(clusters), clusters, filePath, labels);
}

// saveClusterSequencesToCsvOrdered, FileIOManager.java:371-374
inline void saveClusterSequencesToCsvOrdered(const std::vector<ClusterPtr> &clusters, const std::string &filePath,
                                             const std::vector<std::string> &labels,
                                             const std::vector<UniqueSequencePtr> &orderedSequences) {
    writeClusterSequencesToCsv(orderedSequences, clusters, filePath, labels);
}

// SaveClustersToCsv, FileIOManager.java:649-676
inline void SaveClustersToCsv(const std::vector<ClusterPtr> &clusters, const std::string &filePath,
                              const std::vector<std::string> &labels, bool sortInPlace = true) {
    std::ofstream w(filePath, std::ios::binary);
    if (!w) throw HammockException("java.io.IOException: cannot write " + filePath);
    std::string head = std::string("cluster_id") + CSV_SEPARATOR + "main_sequence" + CSV_SEPARATOR + "sum";
    for (const std::string &label : labels) { head += CSV_SEPARATOR; head += label; }
    head += "\n";
    const LabelColumns columns(labels);
    std::vector<size_t> first_column(labels.size());   // a label listed twice repeats its column
    for (size_t k = 0; k < labels.size(); k++) first_column[k] = (size_t)(std::find(labels.begin(), labels.end(), labels[k]) - labels.begin());
    auto compareTo = [](const UniqueSequence &x, const UniqueSequence &y) {   // UniqueSequence.compareTo :161-171
        if (x.size() != y.size()) return x.size() - y.size();
        return -javaStringCompare(x.getSequenceString(), y.getSequenceString());
    };
    const std::vector<ClusterPtr> order = clustersSortedDescending(clusters);
    // one line per cluster; the clusters are independent, so runs of 1,024 of them are formatted on several threads
    const size_t RUN = 1024;
    std::vector<std::string> parts((order.size() + RUN - 1) / RUN);
    parallelChunks(order.size(), RUN, hostThreads(), [&](size_t lo, size_t hi) {
        std::string &out = parts[lo / RUN];
        std::vector<long long> sums(labels.size());
        for (size_t c = lo; c < hi; c++) {
            const ClusterPtr &cl = order[c];
            auto &seqs = cl->getSequences();
            // Collections.sort(sequences, reverseOrder()) on the cluster's own list as in the reference; the line needs only
            // the list's new head, so with sortInPlace = false (saveInitialClusters below: other writers are reading the
            // lists) the head is found without sorting: the first of the greatest elements, which is where a stable sort puts it
            const UniqueSequence *head_seq = seqs[0].get();
            if (sortInPlace) {
                if (seqs.size() > 1) std::stable_sort(seqs.begin(), seqs.end(), [&](const UniqueSequencePtr &a, const UniqueSequencePtr &b) {
                    return compareTo(*b, *a) < 0;
                });
                head_seq = seqs[0].get();
            } else {
                for (auto &s : seqs) if (compareTo(*s, *head_seq) > 0) head_seq = s.get();
            }
            appendNumber(out, cl->getId());
            out += CSV_SEPARATOR; out += head_seq->getSequenceString(); out += CSV_SEPARATOR; appendNumber(out, cl->size());
            std::fill(sums.begin(), sums.end(), 0);
            for (auto &s : seqs) columns.add(*s, sums);
            for (size_t k = 0; k < labels.size(); k++) { out += CSV_SEPARATOR; appendNumber(out, sums[first_column[k]]); }
            out += "\n";
        }
    });
    w.write(head.data(), (std::streamsize)head.size());
    for (const std::string &part : parts) w.write(part.data(), (std::streamsize)part.size());
}

// The three result files of runGreedyClustering / runClinkageClustering (Hammock.java:429-432, :484-487) written side by side:
// the member lists are sorted once (what saveClusterSequencesToCsv does first) and the string -> cluster table is built once,
// then the three writers only read the clusters (SaveClustersToCsv finds each line's main sequence without sorting).  The
// files are the ones the three calls in a row produce.
inline void saveInitialClusters(const std::vector<ClusterPtr> &clusters, const std::string &sequencesCsv,
                                const std::string &sequencesOrderedCsv, const std::string &clustersCsv,
                                const std::vector<std::string> &labels, const std::vector<UniqueSequencePtr> &orderedSequences) {
    const std::vector<UniqueSequencePtr> sortedSequences = sortedClusterSequences(clusters);   // the only step that changes the lists
    auto summary = std::async(std::launch::async, [&] { SaveClustersToCsv(clusters, clustersCsv, labels, false); });
    std::exception_ptr failed;
    try {
        const SequenceClusterIndex index(clusters);
        auto ordered = std::async(std::launch::async, [&] { writeClusterSequencesToCsv(orderedSequences, index, sequencesOrderedCsv, labels); });
        try {
            writeClusterSequencesToCsv(sortedSequences, index, sequencesCsv, labels);
        } catch (...) {
            failed = std::current_exception();
        }
        try { ordered.get(); } catch (...) { if (!failed) failed = std::current_exception(); }
    } catch (...) {
        if (!failed) failed = std::current_exception();
    }
    try { summary.get(); } catch (...) { if (!failed) failed = std::current_exception(); }
    if (failed) std::rethrow_exception(failed);
}

// saveInputStatistics, FileIOManager.java:709-729 (no newline after the last row)
inline void saveInputStatistics(const std::vector<UniqueSequencePtr> &sequences, const std::vector<std::string> &labels,
                                const std::string &filePath) {
    std::ofstream w(filePath, std::ios::binary);
    if (!w) throw HammockException("java.io.IOException: cannot write " + filePath);
    for (const std::string &label : labels) w << CSV_SEPARATOR << label;
    // per label: the sum of its counts and the number of sequences that carry it -- one pass over the list on several threads
    const unsigned T = hostThreads();
    const size_t L = labels.size();
    std::vector<std::vector<long long>> parts(T, std::vector<long long>(2 * L, 0));
    parallelRanges(sequences.size(), T, [&](unsigned t, size_t lo, size_t hi) {
        std::vector<long long> &p = parts[t];
        for (size_t q = lo; q < hi; q++)
            for (size_t k = 0; k < L; k++) {
                bool present = false;
                p[k] += sequences[q]->labelCount(labels[k], &present);
                p[L + k] += present;
            }
    });
    std::vector<long long> all(2 * L, 0);
    for (auto &p : parts) for (size_t k = 0; k < 2 * L; k++) all[k] += p[k];
    w << "\n" << "total_count";
    for (size_t k = 0; k < L; k++) w << CSV_SEPARATOR << all[k];
    w << "\n" << "unique_count";
    for (size_t k = 0; k < L; k++) w << CSV_SEPARATOR << all[L + k];
}

}  // namespace FileIOManager
}  // namespace hammock

#endif
