"""The C++ host side (hammock_amd/host/): loaders, orderings and label ranking of the
hammock-hip CLI against the oracle's Python restatement -- CPU only, no scoring."""
import gzip
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT
from oracle import hammock_oracle as po

CLI = os.path.join(ROOT, "hammock_amd", "bin", "hammock-hip")


def cli(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True)


def expected_listing(seqs, order, seed=42):
    labels = po.get_sorted_labels(seqs)
    po.sort_sequences(seqs, order, seed=seed, labels=labels)
    lines = ["labels\t" + "\t".join(labels)]
    for s in seqs:
        lines.append(s.get_sequence_string() + "\t" + "\t".join([str(s.size())] + [str(s.labels_map.get(l, 0)) for l in labels]))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("order", ["size", "alphabetic", "input", "random", "label2"])
def test_fasta_and_table_loaders_and_orders(order):
    fa = os.path.join(GOLDEN, "manual_example.fa")
    r = cli("io-selftest", "sequences", "fasta", fa, order, "7")
    assert r.returncode == 0, r.stderr
    assert r.stdout == expected_listing(po.load_unique_sequences_from_fasta(fa), order, 7)
    tab = os.path.join(GOLDEN, "manual_example.tsv")
    r = cli("io-selftest", "sequences", "tab", tab, order, "7")
    assert r.returncode == 0, r.stderr
    assert r.stdout == expected_listing(po.load_unique_sequences_from_table(tab), order, 7)


def test_antibodies_counts_labels_and_shuffle(tmp_path):
    """74,041 unique 12-mers, 15 labels, counts in the headers: size order, label ranking, Java shuffle."""
    fa = tmp_path / "antibodies.fa"
    with gzip.open(os.path.join(GOLDEN, "antibodies.fa.gz"), "rb") as src:
        fa.write_bytes(src.read())
    seqs = po.load_unique_sequences_from_fasta(str(fa))
    assert len(seqs) == 74041 and sum(s.size() for s in seqs) == 389873
    for order in ("size", "random"):
        r = cli("io-selftest", "sequences", "fasta", str(fa), order, "42")
        assert r.returncode == 0, r.stderr
        assert r.stdout == expected_listing(po.load_unique_sequences_from_fasta(str(fa)), order, 42)


def test_parallel_fasta_loader_equals_literal_loader(tmp_path):
    """loadUniqueSequencesFromFasta cuts a large file at header lines and parses the pieces on several threads; the result must
    be the literal loader's (FileIOManager.java:159-202, HMK_LITERAL_LOADER=1) and the oracle's: duplicates that meet across
    pieces add their counts to the FIRST occurrence, labels keep their first-seen order per sequence, multi-line sequences are
    concatenated, \\r\\n and \\r end lines, a header without a sequence adds nothing -- except the very last one, which adds the
    empty sequence (:193-195).  Malformed input goes to the literal loader and raises what it raises."""
    import numpy as np
    rng = np.random.default_rng(3)
    aa = "ARNDCQEGHILKMFPSTWYV"
    base = ["".join(aa[int(c)] for c in rng.integers(0, 20, size=int(rng.integers(7, 21)))) for _ in range(6000)]
    lines = []
    for k in range(40000):
        s = base[int(rng.integers(0, len(base)))]            # many duplicates, all over the file
        kind = int(rng.integers(0, 6))
        head = f">{k}" if kind == 0 else f">{k}|{1 + int(rng.integers(0, 50))}" if kind == 1 else \
            f"> s{k} | {1 + int(rng.integers(0, 9))} |lab{int(rng.integers(0, 5))}|extra" if kind == 2 else f">{k}|{1 + int(rng.integers(0, 9))}|lab{int(rng.integers(0, 5))}"
        eol = "\r\n" if k % 7 == 0 else "\r" if k % 11 == 0 else "\n"
        if k % 5 == 0 and len(s) > 8:                        # multi-line, lower case, padded
            lines.append(head + eol + "  " + s[:5].lower() + " " + eol + s[5:] + eol)
        elif k % 97 == 0:
            lines.append(head + eol)                          # a header without a sequence
        else:
            lines.append(head + eol + s + eol)
    fa = tmp_path / "big.fa"
    fa.write_bytes("".join(lines).encode())
    assert fa.stat().st_size > (1 << 16)
    want = expected_listing(po.load_unique_sequences_from_fasta(str(fa)), "input", 42)
    fast = cli("io-selftest", "sequences", "fasta", str(fa), "input", "42")
    assert fast.returncode == 0, fast.stderr
    env = dict(os.environ, HMK_LITERAL_LOADER="1")
    literal = subprocess.run([CLI, "io-selftest", "sequences", "fasta", str(fa), "input", "42"], capture_output=True, text=True, env=env)
    assert literal.returncode == 0, literal.stderr
    assert fast.stdout == literal.stdout == want
    # the same file ending in a header: the empty sequence is added by the last record only
    fa2 = tmp_path / "tail.fa"
    fa2.write_bytes("".join(lines).encode() + b">last|3|labz\n")
    a = cli("io-selftest", "sequences", "fasta", str(fa2), "input", "42")
    b = subprocess.run([CLI, "io-selftest", "sequences", "fasta", str(fa2), "input", "42"], capture_output=True, text=True, env=env)
    assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr)
    # a count below 1 in the middle of the file: both loaders fail alike
    fa3 = tmp_path / "bad.fa"
    fa3.write_bytes("".join(lines[:20000]).encode() + b">x|0|l\nACDEFGHIKL\n" + "".join(lines[20000:]).encode())
    a = cli("io-selftest", "sequences", "fasta", str(fa3), "input", "42")
    b = subprocess.run([CLI, "io-selftest", "sequences", "fasta", str(fa3), "input", "42"], capture_output=True, text=True, env=env)
    assert a.returncode != 0 and (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr)


def test_matrix_loader_all_shipped_matrices(tmp_path):
    with open(os.path.join(GOLDEN, "matrices.json")) as fh:
        d = json.load(fh)
    aa = d["alphabet"]
    for name, M in d["matrices"].items():
        p = tmp_path / (name + ".txt")
        with open(p, "w") as fh:
            fh.write("# comment\n   " + "  ".join(aa) + "\n")
            for r, row in enumerate(M):
                fh.write(aa[r] + " " + " ".join("%2d" % v for v in row) + " \n")
        r = cli("io-selftest", "matrix", str(p))
        assert r.returncode == 0, (name, r.stderr)
        assert [[int(v) for v in line.split()] for line in r.stdout.splitlines()] == M
        assert po.load_scoring_matrix(str(p)) == M
    # a 22 x 22 file (gonnet250-shaped) is rejected as by FileIOManager.java:61-64
    p = tmp_path / "gonnet_like.txt"
    with open(p, "w") as fh:
        for r in range(22):
            fh.write("A " + " ".join(["1"] * 22) + "\n")
    assert cli("io-selftest", "matrix", str(p)).returncode == 3
    with pytest.raises(po.FileFormatException):
        po.load_scoring_matrix(str(p))
    # 25 rows -> rejected (:69-72 / :76-79)
    p = tmp_path / "too_many_rows.txt"
    with open(p, "w") as fh:
        for r in range(25):
            fh.write("A " + " ".join(["1"] * 24) + "\n")
    assert cli("io-selftest", "matrix", str(p)).returncode == 3


def test_default_matrix_file_is_blosum62():
    with open(os.path.join(GOLDEN, "matrices.json")) as fh:
        M = json.load(fh)["matrices"]["blosum62"]
    assert po.load_scoring_matrix(os.path.join(ROOT, "hammock_amd", "matrices", "blosum62.txt")) == M


def test_cli_argument_errors(tmp_path):
    assert cli("greedy").returncode == 2                                   # no -i: CLIException
    out = tmp_path / "exists"
    out.mkdir()
    r = cli("greedy", "-i", os.path.join(GOLDEN, "musi.fa"), "-d", str(out))
    assert r.returncode == 2 and "Output directory exists" in r.stderr      # Hammock.java:1212-1216
    r = cli("greedy", "-i", os.path.join(GOLDEN, "musi.fa"), "-d", str(tmp_path / "o2"), "-f", "xml")
    assert r.returncode == 2 and "Parameter -f" in r.stderr
    assert cli("full", "-i", "x").returncode == 2                          # other modes are out of scope


def test_jni_shim_compiles_and_covers_every_native_method(tmp_path):
    """The Java side cannot be built here (no JDK).  The C half of the shim can at least be held to compiling: it is
    compiled with -Wall -Wextra -Werror against tests/jni_stub/jni.h (a declaration-only stand-in, NOT the JDK header)
    and include/hammock_hip.h, and every `static native` method of HipNative.java must have its
    Java_cz_krejciadam_hammock_HipNative_<name> function in the object file."""
    import re
    import subprocess
    obj = str(tmp_path / "hammock_jni.o")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fPIC", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
                           "-I" + os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "hammock_amd", "java", "jni", "hammock_jni.c"), "-o", obj])
    syms = subprocess.check_output(["nm", obj], text=True)
    defined = set(re.findall(r" T (Java_cz_krejciadam_hammock_HipNative_\w+)", syms))
    java = open(os.path.join(ROOT, "hammock_amd", "java", "cz", "krejciadam", "hammock", "HipNative.java")).read()
    natives = re.findall(r"static native [\w\[\]]+ (\w+)\(", java)
    assert len(natives) >= 8
    assert {"Java_cz_krejciadam_hammock_HipNative_" + m for m in natives} == defined
    # every C ABI function the shim calls is one the header declares and the library exports
    called = set(re.findall(r"\b(hmk_\w+)\(", open(os.path.join(ROOT, "hammock_amd", "java", "jni", "hammock_jni.c")).read()))
    from hammock_amd import _native
    assert called <= set(_native.SYMBOLS), called - set(_native.SYMBOLS)


def _write_and_compare(tmp_path, fa, seqs_py, clusters_spec, order="size"):
    """runs `io-selftest writers` on clusters_spec = [(id, [sequence strings])] and returns nothing: asserts that the result
    files of the reference's three calls in a row, of the side-by-side writers and of the oracle's writers are the same bytes"""
    spec = tmp_path / "clusters.tsv"
    spec.write_text("".join(f"{cid}\t{','.join(members)}\n" for cid, members in clusters_spec))
    out = tmp_path / "out"
    for sub in ("serial", "side", "expected"):
        (out / sub).mkdir(parents=True)
    r = cli("io-selftest", "writers", "fasta", str(fa), order, "42", str(spec), str(out))
    assert r.returncode == 0, r.stderr
    labels = po.get_sorted_labels(seqs_py)
    initial = list(seqs_py)
    by_string = {s.get_sequence_string(): s for s in seqs_py}
    cl_list = [po.Cluster([by_string[m] for m in members], cid) for cid, members in clusters_spec]
    exp = out / "expected"
    po.save_cluster_sequences_csv(cl_list, str(exp / "initial_clusters_sequences.tsv"), labels)
    po.write_cluster_sequences_csv(initial, cl_list, str(exp / "initial_clusters_sequences_original_order.tsv"), labels)
    po.save_clusters_csv(cl_list, str(exp / "initial_clusters.tsv"), labels)
    po.save_input_statistics(seqs_py, labels, str(exp / "input_statistics.tsv"))
    assert (out / "input_statistics.tsv").read_bytes() == (exp / "input_statistics.tsv").read_bytes()
    for name in ("initial_clusters_sequences.tsv", "initial_clusters_sequences_original_order.tsv", "initial_clusters.tsv"):
        want = (exp / name).read_bytes()
        assert (out / "serial" / name).read_bytes() == want, ("serial", name)
        assert (out / "side" / name).read_bytes() == want, ("side", name)


def test_result_writers_side_by_side_equal_serial_and_oracle(tmp_path):
    """The three stage-1 files (FileIOManager.java:594-676) for a random assignment of the antibodies set (74,041 sequences, 15
    labels, counts): clusters of 1..40 members with random ids, 5 % of the sequences in no cluster ("NA" lines).  The
    side-by-side writers share one string -> cluster table built on several threads, sort the member lists in parallel and find
    a cluster's main sequence without sorting; the bytes must be those of the three calls in a row and of the oracle."""
    import random
    fa = tmp_path / "antibodies.fa"
    with gzip.open(os.path.join(GOLDEN, "antibodies.fa.gz"), "rb") as src:
        fa.write_bytes(src.read())
    seqs = po.load_unique_sequences_from_fasta(str(fa))
    rnd = random.Random(5)
    names = [s.get_sequence_string() for s in seqs]
    rnd.shuffle(names)
    names = names[: int(len(names) * 0.95)]
    ids = rnd.sample(range(len(seqs)), len(names))
    spec, at = [], 0
    while at < len(names):
        k = 1 if rnd.random() < 0.5 else rnd.randint(2, 40)
        spec.append((ids[len(spec)], names[at:at + k]))
        at += k
    _write_and_compare(tmp_path, fa, seqs, spec)


def test_result_writers_with_a_string_in_two_clusters(tmp_path):
    """Impossible after the loaders but legal for the interface: the same sequence in two clusters (and a singleton whose string
    is also a member elsewhere).  The reference's HashMap.put keeps the LAST cluster for the string and msaMap decides the
    alignment column (FileIOManager.java:596-607); the parallel table detects the repeat and is rebuilt that way."""
    fa = os.path.join(GOLDEN, "musi.fa")
    seqs = po.load_unique_sequences_from_fasta(fa)
    names = [s.get_sequence_string() for s in seqs]
    spec = [(7, names[0:3]), (3, [names[1], names[3]]), (11, [names[4]]), (2, [names[4], names[5]]), (5, [names[0]])]
    spec += [(100 + k, names[k:k + 7]) for k in range(6, len(names) - 7, 7)]
    _write_and_compare(tmp_path, fa, seqs, spec, order="input")
