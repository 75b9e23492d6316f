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
