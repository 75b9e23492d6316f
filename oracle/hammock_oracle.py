"""Pure-Python restatement of the reference's greedy initial-clustering path.

TEST INFRASTRUCTURE ONLY -- a second, independently structured restatement
(object style, mirroring the Java classes one to one) used to cross-check the C
oracle (oracle/hammock_oracle.c) on small inputs.  Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import it;
the product package ``hammock_amd`` never does.

PARITY STATUS: "parity unpinned" -- the reference (Java 7) cannot be run in
this image and has no tests for this path; see oracle/hammock_oracle.h.

Citations are relative to /root/reference/src/cz/krejciadam/hammock/.
"""
from __future__ import annotations

import re
from collections import OrderedDict
from functools import cmp_to_key

INT_MIN = -(2 ** 31)
INT_MAX = 2 ** 31 - 1

AMINO_ACIDS = "ARNDCQEGHILKMFPSTWYVBZX*"  # UniqueSequence.java:23-26
NAME_TO_NUM = {c: i for i, c in enumerate(AMINO_ACIDS)}


class HammockException(Exception):
    pass


class DataException(HammockException):
    pass


class FileFormatException(HammockException):
    pass


class ReferenceWouldCrash(Exception):
    """The reference dereferences a null Cluster here."""

    def __init__(self, case, index):
        super().__init__(f"NullPointerException case {case} at index {index}")
        self.case = case
        self.index = index


def _java_int_decode(text):
    """Integer.decode: optional sign, then 0x/0X/# hex, leading-0 octal, decimal."""
    t = text
    neg = False
    if t.startswith("-"):
        neg, t = True, t[1:]
    elif t.startswith("+"):
        t = t[1:]
    if t.startswith(("0x", "0X")):
        v = int(t[2:], 16)
    elif t.startswith("#"):
        v = int(t[1:], 16)
    elif t.startswith("0") and len(t) > 1:
        v = int(t[1:], 8)
    else:
        if not re.fullmatch(r"[0-9]+", t):
            raise ValueError(f"NumberFormatException: {text!r}")
        v = int(t)
    return -v if neg else v


class UniqueSequence:
    """UniqueSequence.java:19-171."""

    def __init__(self, sequence, labels_map=None):
        if labels_map is None:  # :65-74
            labels_map = {"no_label": 1}
        self.labels_map = labels_map
        self.sequence = []
        for ch in sequence.upper():  # :49-56
            if ch not in NAME_TO_NUM:
                raise FileFormatException(f"Error, character {ch} is not a valid letter")
            self.sequence.append(NAME_TO_NUM[ch])

    def size(self):  # :82-88
        return sum(self.labels_map.values())

    def get_sequence_string(self):  # :103-109
        return "".join(AMINO_ACIDS[i] for i in self.sequence)

    def __repr__(self):
        return f"UniqueSequence({self.get_sequence_string()}, {self.labels_map})"


def _string_compare(a, b):
    """java.lang.String.compareTo for ASCII strings."""
    for x, y in zip(a, b):
        if x != y:
            return ord(x) - ord(y)
    return len(a) - len(b)


def _alphabetic_cmp(o1, o2):  # UniqueSequence.java:255-261
    return _string_compare(o1.get_sequence_string(), o2.get_sequence_string())


def _size_alphabetic_cmp(o1, o2):  # :238-248
    r = o1.size() - o2.size()
    if r == 0:
        r = _alphabetic_cmp(o1, o2)
    return r


class JavaRandom:
    """java.util.Random (48-bit LCG), enough for Collections.shuffle."""

    def __init__(self, seed):
        self.seed = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def next(self, bits):
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.seed >> (48 - bits)
        if v >= 1 << 31:  # (int) cast of the 48-bit shift result
            v -= 1 << 32
        return v

    def next_int(self, bound):
        r = self.next(31)
        m = bound - 1
        if (bound & m) == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            # overflow check of u - r + m in 32-bit arithmetic
            if u - r + m < 2 ** 31:
                return r
            u = self.next(31)


def java_shuffle(lst, rnd):
    """Collections.shuffle(list, rnd): for i=size..2 swap(i-1, nextInt(i))."""
    for i in range(len(lst), 1, -1):
        j = rnd.next_int(i)
        lst[i - 1], lst[j] = lst[j], lst[i - 1]


def sort_sequences(sequences, order, seed=42, labels=None):
    """UniqueSequence.sortSequences, UniqueSequence.java:176-203 (in place, returns the list)."""
    rev = lambda cmp: cmp_to_key(lambda a, b: cmp(b, a))  # Collections.reverseOrder
    if order == "size":
        sequences.sort(key=rev(_size_alphabetic_cmp))  # list.sort is stable like Collections.sort
    elif order == "alphabetic":
        sequences.sort(key=rev(_alphabetic_cmp))
    elif order == "random":
        java_shuffle(sequences, JavaRandom(seed))
    elif order == "input":
        pass
    else:
        if labels is None or order not in labels:
            raise DataException("Incorrect sequence order defined.")
        sequences.sort(key=rev(_size_alphabetic_cmp))
        sequences.sort(key=rev(lambda a, b: a.labels_map.get(order, 0) - b.labels_map.get(order, 0)))
    return sequences


class Cluster:
    """Cluster.java:21-204 (members, id, size)."""

    def __init__(self, sequences, cid):
        self.sequences = list(sequences)
        self.id = cid
        self._size = sum(s.size() for s in self.sequences)

    def insert(self, seq):  # :50-63
        if any(s.sequence == seq.sequence for s in self.sequences):
            raise DataException("Trying to insert unique sequence twice")
        self.sequences.append(seq)
        self._size += seq.size()

    def insert_all(self, seqs):  # :70-74
        for s in list(seqs):
            self.insert(s)

    def size(self):  # :156-158
        return self._size

    def get_unique_size(self):  # :113-115
        return len(self.sequences)


class ShiftedScorer:
    """ShiftedScorer.java:12-114."""

    def __init__(self, scoring_matrix, shift_penalty, max_shift):
        self.m = scoring_matrix
        self.shift_penalty = shift_penalty
        self.max_shift = max_shift
        self.calls = 0

    def score_with_shift(self, seq1, seq2):
        a, b = seq1.sequence, seq2.sequence
        if len(a) >= len(b):  # :51-57
            shorter, longer, shorter_is_seq2 = b, a, True
        else:
            shorter, longer, shorter_is_seq2 = a, b, False
        if self.max_shift >= len(shorter):  # :59-62
            raise DataException("Shift too big")
        best, best_shift = INT_MIN, 0
        diff = len(longer) - len(shorter)
        for shift in range(-self.max_shift, self.max_shift + diff + 1):  # :67
            score = 0
            if shift <= 0:  # :69-72
                for i in range(len(shorter) + shift):
                    score += self.m[shorter[i - shift]][longer[i]]
            else:  # :73-77
                for i in range(min(len(shorter), len(longer) - shift)):
                    score += self.m[shorter[i]][longer[i + shift]]
            score += diff * self.shift_penalty  # :79
            if shift < 0:
                score += -shift * 2 * self.shift_penalty  # :80-82
            if shift > diff:
                score += (shift - diff) * 2 * self.shift_penalty  # :83-85
            if score > best:  # :86-89
                best, best_shift = score, shift
        if not shorter_is_seq2:
            best_shift = -best_shift  # :91-93
        return best, best_shift

    def sequence_score(self, seq1, seq2):  # :98-100
        self.calls += 1
        return self.score_with_shift(seq1, seq2)[0]


class LocalAlignmentScorer:
    """LocalAlignmentScorer.java:10-155."""

    LEFT, UP, DIAGONAL, NOWHERE = "LEFT", "UP", "DIAGONAL", "NOWHERE"

    def __init__(self, scoring_matrix, gap_open_penalty, gap_extend_penalty):
        self.m = scoring_matrix
        self.gap_open = gap_open_penalty
        self.gap_extend = gap_extend_penalty
        self.calls = 0

    def sequence_score(self, seq1, seq2):  # :27-29
        self.calls += 1
        s1, s2 = seq1.sequence, seq2.sequence
        d1, d2 = len(s1) + 1, len(s2) + 1
        score = [[0] * d2 for _ in range(d1)]  # :88-101
        direction = [[None] * d2 for _ in range(d1)]
        for i in range(1, d1):
            direction[i][0] = self.UP
        for j in range(1, d2):
            direction[0][j] = self.LEFT
        global_max = 0
        for line in range(1, d1):  # :40
            for column in range(1, d2):  # :41
                up_pen = self.gap_extend if direction[line - 1][column] == self.UP else self.gap_open
                left_pen = self.gap_extend if direction[line][column - 1] == self.LEFT else self.gap_open
                up = score[line - 1][column] + up_pen
                left = score[line][column - 1] + left_pen
                diag = score[line - 1][column - 1] + self.m[s1[line - 1]][s2[column - 1]]
                mx = max(diag, max(up, left))
                if mx < 0:  # :63-65
                    score[line][column] = 0
                    direction[line][column] = self.NOWHERE
                else:
                    score[line][column] = mx
                    if mx > global_max:
                        global_max = mx
                    if mx == left:
                        direction[line][column] = self.LEFT
                    if mx == up:
                        direction[line][column] = self.UP
                    if mx == diag:
                        direction[line][column] = self.DIAGONAL
        return global_max


class ClinkageClusterScorer:
    """ClinkageClusterScorer.java:10-50."""

    def __init__(self, scorer, threshold):
        self.scorer = scorer
        self.threshold = threshold

    def cluster_score(self, cl1, cl2):
        result = INT_MAX
        for seq1 in cl1.sequences:
            for seq2 in cl2.sequences:
                r = self.scorer.sequence_score(seq1, seq2)
                if r < result:
                    result = r
                    if result < self.threshold:
                        return INT_MIN + 1
        return result


class NearestCluster:
    def __init__(self, cluster, score):
        self.cluster = cluster
        self.score = score


def _nearest_cluster_runner(database, compared, scorer):
    """NearestClusterRunner.call, ClinkageSequenceClusterer.java:258-293."""
    max_score = INT_MIN
    nearest = None
    for i in database:
        score = scorer.cluster_score(i, compared)
        if score < max_score:
            continue
        if score > max_score:
            if i is not compared:
                max_score = score
                nearest = i
        else:
            if i is not compared:
                if i.size() > nearest.size():
                    nearest = i
                elif i.size() < nearest.size():
                    pass
                elif i.id < nearest.id:
                    nearest = i
    return NearestCluster(nearest, max_score)


def find_nearest_cluster_parallel(input_clusters, compared, scorer, sum_commodity, n_threads=4):
    """ClinkageSequenceClusterer.java:137-223 (parts evaluated serially)."""
    input_clusters = list(input_clusters)
    if not input_clusters:
        return NearestCluster(None, INT_MIN)  # :138-140 non-null dummy
    n_parts = n_threads * 4  # :186-192
    if len(input_clusters) < n_threads * 4 + 1:
        n_parts = max(len(input_clusters) - 1, 1)
    parts, current = [], []  # :202-223
    for_one = sum_commodity // n_parts + 1
    portion = for_one
    for cl in input_clusters:
        current.append(cl)
        portion -= cl.get_unique_size()
        if portion <= 0:
            parts.append(current)
            current = []
            portion = for_one
    if current:
        parts.append(current)
    max_score = INT_MIN + 42  # :151
    nearest = None
    for part in parts:
        cur = _nearest_cluster_runner(part, compared, scorer)
        if cur.score < max_score:
            continue
        if cur.score > max_score:
            nearest = cur
            max_score = cur.score
        else:
            if cur.cluster.size() > nearest.cluster.size():
                nearest = cur
            elif cur.cluster.size() == nearest.cluster.size() and cur.cluster.id < nearest.cluster.id:
                nearest = cur
    return nearest


# ---------------------------------------------------------------------------
# clinkage mode: ClinkageSequenceClusterer.cluster + CachedClusterScorer + DynamicMatrix
# ---------------------------------------------------------------------------
class NoSuchElement(Exception):
    """java.util.NoSuchElementException: activeClusters.iterator().next() on an empty set
    (ClinkageSequenceClusterer.java:118 with an empty input list)."""


def _spread(h):
    """java.util.HashMap.hash (Java 8+): h ^ (h >>> 16) on the 32-bit hashCode."""
    h &= 0xFFFFFFFF
    return h ^ (h >> 16)


class JavaHashSet:
    """Iteration order of java.util.HashSet / HashMap keys, Java 8 and later: power-of-two table starting at 16,
    load factor 0.75, index = spread(hashCode) & (capacity - 1), a bucket's chain in insertion order (new nodes are
    appended, a resize splits a chain preserving relative order), no shrinking on removal.  Tree bins (9 nodes in one
    bucket) never occur for the consecutive integer hash codes used here and are refused.
    hash_of(element) = the element's Java hashCode(); elements compare by `key_of` (Java equals)."""

    def __init__(self, hash_of, key_of=lambda x: x):
        self.hash_of, self.key_of = hash_of, key_of
        self.table = [[] for _ in range(16)]
        self.size = 0

    def _bucket(self, x):
        return self.table[_spread(self.hash_of(x)) & (len(self.table) - 1)]

    def add(self, x):
        b = self._bucket(x)
        k = self.key_of(x)
        if any(self.key_of(y) == k for y in b):
            return False
        b.append(x)
        if len(b) >= 9:
            raise NotImplementedError("HashMap tree bin: iteration order not modelled")
        self.size += 1
        if self.size > len(self.table) * 3 // 4:      # ++size > threshold -> resize()
            old = self.table
            self.table = [[] for _ in range(2 * len(old))]
            for chain in old:
                for y in chain:
                    self._bucket(y).append(y)
        return True

    def remove(self, x):
        b = self._bucket(x)
        k = self.key_of(x)
        for i, y in enumerate(b):
            if self.key_of(y) == k:
                del b[i]
                self.size -= 1
                return True
        return False

    def __contains__(self, x):
        k = self.key_of(x)
        return any(self.key_of(y) == k for y in self._bucket(x))

    def __len__(self):
        return self.size

    def __iter__(self):
        for chain in list(self.table):
            for y in list(chain):
                yield y

    def first(self):
        """iterator().next()"""
        for chain in self.table:
            if chain:
                return chain[0]
        raise NoSuchElement()


class JavaHashSet7(JavaHashSet):
    """The same for Java 7 and earlier -- the reference is a Java 1.7 project (nbproject/project.properties:45-46).
    java.util.HashMap before the Java 8 rewrite: hash(h) = h ^ (h >>> 20) ^ (h >>> 12), then h ^ (h >>> 7) ^ (h >>> 4);
    a new entry goes to the HEAD of its bucket's chain (createEntry); transfer() walks the old buckets in order and puts
    every entry at the head of its new bucket, so a resize reverses the chains it keeps together.
      variant 7 (JDK 7u6 ... 7u80): addEntry resizes BEFORE inserting, when size >= threshold AND the target bucket is not
                                    empty;
      variant 6 (JDK 6, JDK 7 GA ... 7u5): inserts, then resizes when size++ >= threshold.
    Iteration is the same everywhere: buckets in index order, a chain from its head."""

    def __init__(self, hash_of, key_of=lambda x: x, variant=7):
        super().__init__(hash_of, key_of)
        self.variant = variant

    @staticmethod
    def _hash7(h):
        h &= 0xFFFFFFFF
        h ^= (h >> 20) ^ (h >> 12)
        return (h ^ (h >> 7) ^ (h >> 4)) & 0xFFFFFFFF

    def _bucket(self, x):
        return self.table[self._hash7(self.hash_of(x)) & (len(self.table) - 1)]

    def _resize(self):
        old = self.table
        self.table = [[] for _ in range(2 * len(old))]
        for chain in old:
            for y in chain:                       # chain[0] is the head
                self._bucket(y).insert(0, y)      # e.next = newTable[i]; newTable[i] = e

    def add(self, x):
        b = self._bucket(x)
        k = self.key_of(x)
        if any(self.key_of(y) == k for y in b):
            return False
        threshold = len(self.table) * 3 // 4
        if self.variant == 7:
            if self.size >= threshold and b:      # addEntry: (size >= threshold) && (null != table[bucketIndex])
                self._resize()
                b = self._bucket(x)
            b.insert(0, x)
            self.size += 1
        else:
            b.insert(0, x)                        # table[bucketIndex] = new Entry(hash, key, value, e)
            self.size += 1
            if self.size - 1 >= threshold:        # if (size++ >= threshold) resize(2 * table.length)
                self._resize()
        return True


JAVA_HASHSET = 8          # 8: Java 8 and later (default); 7: JDK 7u6+; 6: JDK 6 and JDK 7 before 7u6


def _cluster_set():
    # Cluster.hashCode() = 79 * 7 + id (Cluster.java:178-183); equals compares ids (:185-195)
    if JAVA_HASHSET == 8:
        return JavaHashSet(lambda c: 553 + c.id, lambda c: c.id)
    return JavaHashSet7(lambda c: 553 + c.id, lambda c: c.id, JAVA_HASHSET)


class DynamicMatrix:
    """CachedClusterScorer.java:128-280, literally: a lower-triangular List<List<Integer>> with re-used ("dirty") rows."""

    def __init__(self):  # :133-147
        self.matrix = [[None], [None, None]]
        self.dirty = JavaHashSet(lambda i: i)
        self.dirty.add(1)

    def get_row(self, row_index):  # :154-160
        line = list(self.matrix[row_index])
        for i in range(row_index + 1, len(self.matrix)):
            line.append(self.matrix[i][row_index])   # a null row here is a NullPointerException in the reference
        return line

    def get(self, row_index, column_index):  # :168-175
        line = self.matrix[max(row_index, column_index)]
        return line[min(row_index, column_index)] if line is not None else None

    def set(self, row_index, column_index, value):  # :184-196
        line = self.matrix[max(row_index, column_index)]
        if line is not None:
            line[min(row_index, column_index)] = value

    def remove(self, row_index):  # :202-204
        self.dirty.add(row_index)

    def add_empty(self, matrix_index):  # :213-219
        self.add_row([None] * (matrix_index + 1), matrix_index)

    def add_value(self, index, row_index, value):  # :230-237
        line = [None] * len(self.matrix)
        line[index] = value
        self.add_row(line, row_index)

    def add_row(self, row, row_index):  # :247-260
        self.matrix[row_index] = row[:row_index + 1]
        for i in range(row_index + 1, len(self.matrix)):
            current = self.matrix[i]
            if current is not None and i < len(row):
                current[row_index] = row[i]
            # else: "Do nothing" -- a stale value of an earlier tenant of this index may stay in column row_index

    def get_matrix_index(self):  # :268-278
        if len(self.dirty):
            idx = self.dirty.first()
            self.dirty.remove(idx)
        else:
            idx = len(self.matrix)
            self.matrix.append(None)
        return idx


class CachedClusterScorer:
    """CachedClusterScorer.java:23-126 (single-threaded reading of the synchronized collections)."""

    def __init__(self, default_cluster_scorer, size_limit):
        self.size_limit = size_limit
        self.default = default_cluster_scorer
        self.index_map = {}
        self.matrix = DynamicMatrix()
        self.default_calls = 0

    def _score(self, cl1, cl2):
        self.default_calls += 1
        return self.default.cluster_score(cl1, cl2)

    def cluster_score(self, cl1, cl2):  # :38-79
        if cl1.get_unique_size() < self.size_limit or cl2.get_unique_size() < self.size_limit:
            return self._score(cl1, cl2)
        im, m = self.index_map, self.matrix
        if cl1.id in im:
            if cl2.id in im:
                i1, i2 = im[cl1.id], im[cl2.id]
                score = m.get(i1, i2)
                if score is not None:
                    return score
                score = self._score(cl1, cl2)
                m.set(i1, i2, score)
            else:
                idx = m.get_matrix_index()
                score = self._score(cl1, cl2)
                m.add_value(im[cl1.id], idx, score)
                im[cl2.id] = idx
        else:
            if cl2.id in im:
                idx = m.get_matrix_index()
                score = self._score(cl1, cl2)
                m.add_value(im[cl2.id], idx, score)
                im[cl1.id] = idx
            else:
                first = m.get_matrix_index()
                second = m.get_matrix_index()
                score = self._score(cl1, cl2)
                m.add_empty(first)
                m.add_empty(second)
                im[cl1.id] = first
                im[cl2.id] = second
        return score

    def join(self, cl1, cl2, new_id):  # :82-125
        if cl1.get_unique_size() < self.size_limit or cl2.get_unique_size() < self.size_limit:
            return
        im, m = self.index_map, self.matrix
        if cl1.id in im:
            i1 = im[cl1.id]
            if cl2.id in im:
                i2 = im[cl2.id]
                line1, line2 = m.get_row(i1), m.get_row(i2)
                new_line = []
                for a, b in zip(line1, line2):
                    new_line.append(min(a, b) if a is not None and b is not None else None)
                m.remove(i1)
                m.remove(i2)
                new_index = m.get_matrix_index()
                m.add_row(new_line, new_index)
                del im[cl1.id]
                del im[cl2.id]
                im[new_id] = new_index
            else:
                m.remove(i1)
                del im[cl1.id]
        elif cl2.id in im:
            m.remove(im[cl2.id])
            del im[cl2.id]


def _nearest_over_hash_parts(input_set, compared, scorer, sum_commodity, n_threads):
    """findNearestClusterParallel (ClinkageSequenceClusterer.java:137-223) with the parts as the HashSets the reference
    builds (:202-223), evaluated one after the other: the ORDER of the clusterScore calls is what a single pool thread
    would produce, which matters to the cache's index allocation."""
    if len(input_set) == 0:
        return NearestCluster(None, INT_MIN)
    n_parts = n_threads * 4
    if len(input_set) < n_threads * 4 + 1:
        n_parts = max(len(input_set) - 1, 1)
    parts, current = [], _cluster_set()
    for_one = sum_commodity // n_parts + 1
    portion = for_one
    for cl in input_set:
        current.add(cl)
        portion -= cl.get_unique_size()
        if portion <= 0:
            parts.append(current)
            current = _cluster_set()
            portion = for_one
    if len(current) > 0:
        parts.append(current)
    max_score = INT_MIN + 42
    nearest = None
    for part in parts:
        cur = _nearest_cluster_runner(part, compared, scorer)
        if cur.score < max_score:
            continue
        if cur.score > max_score:
            nearest = cur
            max_score = cur.score
        elif cur.cluster.size() > nearest.cluster.size():
            nearest = cur
        elif cur.cluster.size() == nearest.cluster.size() and cur.cluster.id < nearest.cluster.id:
            nearest = cur
    return nearest


class ClinkageSequenceClusterer:
    """ClinkageSequenceClusterer.java:21-124: exact complete-linkage clustering by nearest-neighbour chain, at
    Hammock.nThreads = n_threads with the pool's tasks run one after the other.  size_limit: the constructor's
    sizeLimit (1 from Hammock.java:459, i.e. every score goes through the cache); a huge value bypasses the cache."""

    def __init__(self, sequence_scorer, threshold, size_limit=1, n_threads=1):
        self.sequence_scorer = sequence_scorer
        self.threshold = threshold
        self.size_limit = size_limit
        self.n_threads = n_threads
        self.stats = {}

    def cluster(self, sequences):  # :43-124
        scorer = CachedClusterScorer(ClinkageClusterScorer(self.sequence_scorer, self.threshold), self.size_limit)
        stack = []
        ready = _cluster_set()
        active = _cluster_set()
        current_id = 1
        for seq in sequences:                                   # :50-55
            active.add(Cluster([seq], current_id))
            current_id += 1
        sum_commodity = sum(c.get_unique_size() for c in active)  # :57-59
        merges = searches = 0
        while len(active) > 1:                                  # :63
            stack.append(active.first())                        # :70-71
            while stack:                                        # :72
                top = stack[-1]
                found = _nearest_over_hash_parts(active, top, scorer, sum_commodity, self.n_threads)  # :77
                searches += 1
                max_score, nearest = INT_MIN, None
                if found is not None:                           # :80-83
                    nearest, max_score = found.cluster, found.score
                if max_score < self.threshold:                  # :86-92
                    stack.pop()
                    ready.add(top)
                    active.remove(top)
                    sum_commodity -= top.get_unique_size()
                    continue
                if len(stack) > 1 and stack[-2].id == nearest.id:   # :96
                    current_id += 1
                    stack.pop()
                    stack.pop()
                    active.remove(top)
                    active.remove(nearest)
                    scorer.join(top, nearest, current_id)       # :102
                    sum_commodity -= top.get_unique_size() + nearest.get_unique_size()
                    merged = top.sequences                      # :105-106: top's own list, then the nearest's members
                    merged.extend(nearest.sequences)
                    new_top = Cluster(merged, current_id)
                    active.add(new_top)
                    sum_commodity += new_top.get_unique_size()
                    merges += 1
                else:
                    stack.append(nearest)                       # :113
        ready.add(active.first())                               # :118 (NoSuchElementException for an empty input)
        self.stats = {"merges": merges, "searches": searches, "default_scorer_calls": scorer.default_calls}
        return list(ready)                                      # :121-123: the HashSet's iteration order


def clinkage_defaults(sequences):
    """threshold and max shift Hammock derives for `clinkage` (Hammock.java:452-455,1415-1434)."""
    thr, x, _ = greedy_defaults(sequences)
    return thr, x


class LimitedGreedySequenceClusterer:
    """LimitedGreedySequenceClusterer.java:17-121."""

    def __init__(self, sequence_scorer, threshold, max_clusters, n_threads=4):
        self.threshold = threshold
        self.max_clusters = max_clusters
        self.sequence_scorer = sequence_scorer
        self.n_threads = n_threads
        self.stats = {}

    def cluster(self, sequences):  # :39-69
        cluster_scorer = ClinkageClusterScorer(self.sequence_scorer, self.threshold)
        clusters = self._first_phase(sequences, self.max_clusters, cluster_scorer)
        index = len(clusters)
        for i, c in enumerate(clusters):
            if c.get_unique_size() == 1:
                index = i
                break
        actual_clusters = list(clusters[:index])
        actual_sequences = list(clusters[index:])
        remaining = []
        sum_commodity = sum(c.get_unique_size() for c in actual_clusters)
        calls0 = self.sequence_scorer.calls
        for cl in actual_sequences:
            found = find_nearest_cluster_parallel(actual_clusters, cl, cluster_scorer, sum_commodity, self.n_threads)
            if found is not None and found.score >= self.threshold:
                found.cluster.insert_all(cl.sequences)  # NPE impossible: dummy score is MIN_VALUE
            else:
                remaining.append(cl)
        self.stats["score_calls_phase2"] = self.sequence_scorer.calls - calls0
        actual_clusters.extend(remaining)
        return actual_clusters

    def _first_phase(self, sequences, max_clusters, cluster_scorer):  # :77-120
        initial = [Cluster([s], i) for i, s in enumerate(sequences)]
        actual_clusters, actual_sequences = [], []
        sum_commodity_clusters = 0
        index = 0
        calls0 = self.sequence_scorer.calls
        while index < len(initial) and len(actual_clusters) < max_clusters:
            compared = initial[index]
            a = find_nearest_cluster_parallel(actual_clusters, compared, cluster_scorer, sum_commodity_clusters, self.n_threads)
            b = find_nearest_cluster_parallel(initial[index + 1:], compared, cluster_scorer, len(initial) - index - 1, self.n_threads)
            if a is not None:
                if b is not None:
                    if a.score >= b.score:
                        if a.cluster is None:
                            raise ReferenceWouldCrash(2, index)  # :97
                        a.cluster.insert_all(compared.sequences)
                    else:
                        compared.insert_all(b.cluster.sequences)
                        actual_clusters.append(compared)
                        initial.remove(b.cluster)
                else:
                    if a.cluster is None:
                        raise ReferenceWouldCrash(1, index)  # :104
                    a.cluster.insert_all(compared.sequences)
            else:
                if b is not None:
                    if b.cluster is None:
                        raise ReferenceWouldCrash(3, index)  # :108
                    compared.insert_all(b.cluster.sequences)
                    actual_clusters.append(compared)
                    initial.remove(b.cluster)
                else:
                    actual_sequences.append(compared)
            index += 1
        self.stats["score_calls_phase1"] = self.sequence_scorer.calls - calls0
        self.stats["phase1_stop_index"] = index
        self.stats["phase1_clusters"] = len(actual_clusters)
        self.stats["phase1_orphans"] = len(actual_sequences)
        return actual_clusters + actual_sequences + initial[index:]


# ---------------------------------------------------------------------------
# loaders (FileIOManager.java)
# ---------------------------------------------------------------------------
def _java_split_ws(line):
    """String.split("\\s+"): leading empty token kept, trailing empties dropped."""
    parts = re.split(r"\s+", line)
    while parts and parts[-1] == "":
        parts.pop()
    return parts


def load_scoring_matrix(path):
    """FileIOManager.loadScoringMatrix, FileIOManager.java:46-81."""
    matrix = [[0] * 24 for _ in range(24)]
    line_counter = 0
    with open(path, "r") as fh:
        for raw in fh.read().splitlines():
            line = raw
            if not (line.startswith("#") or line.startswith(" ") or line.startswith("\t")):
                parts = _java_split_ws(line)
                if len(parts) != 25:
                    raise FileFormatException("Scoring matrix should always have 24 columns")
                if line_counter >= 24:  # ArrayIndexOutOfBounds -> FileFormatException, :76-79
                    raise FileFormatException("Scoring matrix should always have 24 rows")
                for i in range(1, 25):
                    matrix[line_counter][i - 1] = int(parts[i])
                line_counter += 1
                if line_counter > 24:
                    raise FileFormatException("Scoring matrix should always have 24 rows")
    return matrix


def load_unique_sequences_from_fasta(path):
    """FileIOManager.loadUniqueSequencesFromFasta, FileIOManager.java:159-216."""
    sequence_map = OrderedDict()
    sequence = ""
    label = None
    count = None

    def update(seq):
        lm = sequence_map.get(seq)
        if lm is None:
            lm = {label: count}
        else:
            lm[label] = lm.get(label, 0) + count
        sequence_map[seq] = lm

    with open(path, "r") as fh:
        for raw in fh.read().splitlines():
            line = raw
            if line.startswith(">"):
                if len(sequence) > 0:
                    update(sequence)
                    sequence = ""
                split = line.strip()[1:].split("|")
                while len(split) > 1 and split[-1] == "":  # Java split drops trailing empties
                    split.pop()
                if len(split) >= 2:
                    count = _java_int_decode(split[1].strip())
                    if count < 1:
                        raise FileFormatException("Fasta header defines sequence count lower than 1.")
                else:
                    count = 1
                label = split[2] if len(split) >= 3 else "no_label"
            else:
                if label is None or count is None:
                    raise FileFormatException("Incorrect fasta format.")
                sequence += line.strip()
    if label is None or count is None:
        raise ReferenceWouldCrash(0, -1)  # NPE unboxing null count, :193
    update(sequence)
    return [UniqueSequence(k, v) for k, v in sequence_map.items()]


def load_unique_sequences_from_table(path, sep="\t"):
    """FileIOManager.loadUniqueSequencesFromTable, FileIOManager.java:227-255."""
    result = []
    with open(path, "r") as fh:
        lines = fh.read().splitlines()
    labels = lines[0].split(sep)[1:]
    for line in lines[1:]:
        parts = line.split(sep)
        lm = {}
        for i, v in enumerate(parts[1:]):
            value = _java_int_decode(v)
            if value != 0:
                lm[labels[i]] = value
        result.append(UniqueSequence(parts[0], lm))
    return result


def java_round(x):
    """Math.round(double): floor(x + 0.5)."""
    import math
    return int(math.floor(x + 0.5))


def greedy_defaults(sequences):
    """Hammock.java:394-401, :1409-1434: (threshold, max_shift, max_clusters)."""
    lengths = [len(s.sequence) for s in sequences]
    mean = sum(lengths) / len(lengths)
    threshold = java_round(mean * 1.7)
    max_shift = min(java_round(mean / 4), min(lengths) - 1)
    max_clusters = java_round(len(sequences) * 0.025)
    return threshold, max_shift, max_clusters


# ---------------------------------------------------------------------------
# label ordering and stage-1 writers (Hammock.java:1586-1605, FileIOManager.java)
# ---------------------------------------------------------------------------
def java_string_hash(s):
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return h


def java_hashmap_order(keys):
    """Iteration order of a Java 8+ HashMap<String, ?> holding `keys` (insertion order given)."""
    cap = 16
    while len(keys) > cap * 3 // 4:
        cap *= 2
    def bucket(k):
        h = java_string_hash(k)
        return (h ^ (h >> 16)) & (cap - 1)
    return [k for _, _, k in sorted((bucket(k), i, k) for i, k in enumerate(keys))]


def get_sorted_labels(sequences):
    """Hammock.getSortedLabels: TreeMap with ValueComparator (never returns 0): total count
    descending, among equal totals the label put later comes first."""
    insertion, total = [], {}
    for s in sequences:
        for k in java_hashmap_order(list(s.labels_map.keys())):
            if k not in total:
                insertion.append(k)
                total[k] = 0
            total[k] += s.labels_map[k]
    result = []
    for k in java_hashmap_order(insertion):
        pos = 0
        while pos < len(result) and total[result[pos]] > total[k]:
            pos += 1
        result.insert(pos, k)
    return result


def _cluster_compare(a, b):  # Cluster.compareTo, Cluster.java:198-204
    return a.size() - b.size() if a.size() != b.size() else a.id - b.id


def _seq_line(seq, labels):
    return "\t".join([str(seq.size())] + [str(seq.labels_map.get(l, 0)) for l in labels])


def write_cluster_sequences_csv(sequences, clusters, path, labels):
    """writeClusterSequencesToCsv, FileIOManager.java:594-638, without Clustal: singletons carry
    their own sequence as alignment (:770-776), members of multi-member clusters "NA"."""
    seq_cluster, msa = {}, {}
    for cl in clusters:
        for s in cl.sequences:
            seq_cluster[s.get_sequence_string()] = cl
        if cl.get_unique_size() == 1:
            s = cl.sequences[0].get_sequence_string()
            msa[s.replace("-", "")] = s
    with open(path, "w") as fh:
        fh.write("\t".join(["cluster_id", "sequence", "alignment", "sum"] + labels) + "\n")
        for seq in sequences:
            s = seq.get_sequence_string()
            cl = seq_cluster.get(s)
            if cl is not None:
                fh.write(f"{cl.id}\t{s}\t{msa.get(s, 'NA')}\t")
            else:
                fh.write(f"NA\t{s}\tNA\t")
            fh.write(_seq_line(seq, labels) + "\n")


def save_cluster_sequences_csv(clusters, path, labels):
    """saveClusterSequencesToCsv, FileIOManager.java:398-404 (+ getSortedSequences :530-538)."""
    ordered = sorted(clusters, key=cmp_to_key(lambda a, b: _cluster_compare(b, a)))
    seqs = []
    for cl in ordered:
        cl.sequences.sort(key=cmp_to_key(lambda a, b: _size_alphabetic_cmp(b, a)))
        seqs.extend(cl.sequences)
    write_cluster_sequences_csv(seqs, clusters, path, labels)


def save_clusters_csv(clusters, path, labels):
    """SaveClustersToCsv, FileIOManager.java:649-676."""
    def seq_cmp(a, b):  # UniqueSequence.compareTo :161-171
        if a.size() != b.size():
            return a.size() - b.size()
        return -_string_compare(a.get_sequence_string(), b.get_sequence_string())
    ordered = sorted(clusters, key=cmp_to_key(lambda a, b: _cluster_compare(b, a)))
    with open(path, "w") as fh:
        fh.write("\t".join(["cluster_id", "main_sequence", "sum"] + labels) + "\n")
        for cl in ordered:
            cl.sequences.sort(key=cmp_to_key(lambda a, b: seq_cmp(b, a)))
            counts = [sum(s.labels_map.get(l, 0) for s in cl.sequences) for l in labels]
            fh.write("\t".join([str(cl.id), cl.sequences[0].get_sequence_string(), str(cl.size())] +
                               [str(c) for c in counts]) + "\n")


def save_input_statistics(sequences, labels, path):
    """saveInputStatistics, FileIOManager.java:709-729 (no trailing newline)."""
    with open(path, "w") as fh:
        fh.write("".join("\t" + l for l in labels) + "\n")
        fh.write("total_count" + "".join("\t" + str(sum(s.labels_map.get(l, 0) for s in sequences)) for l in labels) + "\n")
        fh.write("unique_count" + "".join("\t" + str(sum(1 for s in sequences if l in s.labels_map)) for l in labels))
