#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/.

Run HERE (the build container), where /root/reference is mounted:
    python tests/golden/make_golden.py

What it writes, and where each comes from:

  matrices.json        DATA: the 24x24 integer tables of the reference's
                       matrices/*.txt files (public NCBI/BLOSUM/PAM tables),
                       re-encoded as JSON; files the reference loader rejects
                       are listed under "rejected".
  musi.fa              DATA: examples/MUSI/musi.fa (BASELINE.json configs[0]).
  antibodies.fa.gz     DATA: examples/antibodies/antibodies.fa, gzip'ed.
  manual_example.*     DATA typed from manual/manual.tex:146-194 (the only
                       document-level fixture the reference has).
  known_answers.json   the hand-derived pairs of SURVEY.md 8(c) (typed here,
                       not produced by any program).
  musi_greedy_oracle.json   produced by the ORACLE (oracle/hammock_oracle.c),
                       NOT by the reference: a regression pin only.  The Java
                       reference cannot be run in this image ("parity
                       unpinned", see oracle/hammock_oracle.h).

No reference SOURCE text is copied; only data files / data tables.
"""
import gzip
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def parse_matrix_text(path):
    """Independent minimal parser used only to re-encode the data files."""
    rows = []
    with open(path) as fh:
        for line in fh.read().splitlines():
            if line.startswith("#") or line.startswith(" ") or line.startswith("\t"):
                continue
            toks = re.split(r"\s+", line.strip())
            rows.append([int(t) for t in toks[1:]])
    return rows


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (data files)")

    # --- matrices ---------------------------------------------------------
    mats, rejected = {}, {}
    mdir = os.path.join(REF, "matrices")
    for fn in sorted(os.listdir(mdir)):
        if not fn.endswith(".txt"):
            continue
        rows = parse_matrix_text(os.path.join(mdir, fn))
        if len(rows) == 24 and all(len(r) == 24 for r in rows):
            mats[fn[:-4]] = rows
        else:
            rejected[fn[:-4]] = {"rows": len(rows), "cols": sorted({len(r) for r in rows})}
    with open(os.path.join(HERE, "matrices.json"), "w") as fh:
        json.dump({"alphabet": "ARNDCQEGHILKMFPSTWYVBZX*", "matrices": mats, "rejected": rejected}, fh,
                  separators=(",", ":"))
        fh.write("\n")
    print("matrices:", sorted(mats), "rejected:", rejected)

    # --- datasets ---------------------------------------------------------
    shutil.copyfile(os.path.join(REF, "examples/MUSI/musi.fa"), os.path.join(HERE, "musi.fa"))
    with open(os.path.join(REF, "examples/antibodies/antibodies.fa"), "rb") as src, \
            gzip.GzipFile(os.path.join(HERE, "antibodies.fa.gz"), "wb", mtime=0) as dst:
        shutil.copyfileobj(src, dst)

    # --- manual example (manual/manual.tex:146-194) -------------------------
    with open(os.path.join(HERE, "manual_example.fa"), "w") as fh:
        fh.write(">\nWVTAPRSLPVLP\n>4863\nGSWVVDISNVED\n>4628|8\nNYSGNRPLPGIW\n>6642|4|label1\nRSPIVRQLPSLP\n"
                 ">6643|3|label2|something\nRSPIVRQLPSLP\n>664\nRSPIVRQLPSLP\n>4893|1|label2\nAKSRPLPMVGLV\n")
    with open(os.path.join(HERE, "manual_example.tsv"), "w") as fh:
        fh.write("sequence\tlabel1\tlabel2\tno_label\nWVTAPRSLPVLP\t0\t0\t1\nGSWVVDISNVED\t0\t0\t1\n"
                 "NYSGNRPLPGIW\t0\t0\t8\nRSPIVRQLPSLP\t4\t3\t1\nAKSRPLPMVGLV\t0\t1\t0\n")
    with open(os.path.join(HERE, "manual_example_expected.json"), "w") as fh:
        json.dump({"sequences": [
            ["WVTAPRSLPVLP", {"no_label": 1}],
            ["GSWVVDISNVED", {"no_label": 1}],
            ["NYSGNRPLPGIW", {"no_label": 8}],
            ["RSPIVRQLPSLP", {"label1": 4, "label2": 3, "no_label": 1}],
            ["AKSRPLPMVGLV", {"label2": 1}]]}, fh, indent=1)
        fh.write("\n")

    # --- SURVEY.md 8(c) known answers (typed, hand-derived) -----------------
    known = {
        "source": "SURVEY.md section 8(c); rows with hand=true were re-derived by hand from blosum62",
        "shifted_blosum62": [
            {"seq1": "AAAA", "seq2": "AAAA", "X": 1, "p": 0, "per_shift": [12, 16, 12], "score": 16, "hand": True},
            {"seq1": "WVTAPRSLPVLP", "seq2": "WVTAPRSLPVLP", "X": 3, "p": 0,
             "per_shift": [-1, -12, -16, 66, -16, -12, -1], "score": 66, "hand": True},
            {"seq1": "WVTAPRSLPVLP", "seq2": "RSPIVRQLPSLP", "X": 3, "p": 0,
             "per_shift": [9, -13, -15, 16, -22, -4, 2], "score": 16, "hand": True},
            {"seq1": "ACDEFGH", "seq2": "CDEFGHIKLM", "X": 2, "p": -1,
             "per_shift": [-16, 35, -15, -20, -12, -21, -13, -14], "score": 35, "hand": True},
            {"seq1": "YSYKTRGLPAVP", "seq2": "YYYKTRGLPAVP", "X": 3, "p": 0, "score": 59, "hand": True},
            {"seq1": "YYNRRLPDFRVF", "seq2": "YYNRRLPDLRVF", "X": 3, "p": 0, "score": 62, "hand": True},
        ],
        "local_blosum62_open-5_ext-1": [
            {"seq1": "AW", "seq2": "WA", "score": 11, "hand": True},
            {"seq1": "WWWW", "seq2": "WWAAWW", "score": 38, "hand": True},
            {"seq1": "ACDEFGH", "seq2": "CDEFGHIKLM", "score": 40, "hand": True},
            {"seq1": "HEAGAWGHEE", "seq2": "PAWHEAE", "score": 23, "hand": False},
            {"seq1": "WVTAPRSLPVLP", "seq2": "RSPIVRQLPSLP", "score": 27, "hand": False},
            {"seq1": "YFNAPWM", "seq2": "IYWGWCMS", "score": 11, "hand": False, "gotoh": 13},
            {"seq1": "WWISMFWMGDGCDAKEEF", "seq2": "RNWGPQFGCF", "score": 19, "hand": False, "swapped": 18},
        ],
        "musi_greedy_provisional": {
            "n": 2457, "threshold": 20, "max_shift": 3, "shift_penalty": 0, "max_clusters": 61,
            "first_five": ["YYYKTRGLPAVP", "YYNRRLPDLRVF", "YYNRRLPDFRVF", "YWPDLPAVPYDQ", "YWAWGHGFMNLS"],
            "neighbours_ge20_first_three": [389, 77, 39],
            "phase1_stop_index": 67, "phase1_clusters": 61, "phase1_orphans": 0,
            "final_clusters": 61, "final_singletons": 1755,
            "ten_largest_unique_sizes": [23, 21, 20, 20, 19, 19, 19, 18, 17, 17],
            "score_calls_phase1": 162592, "score_calls_phase2": 169039,
        },
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as fh:
        json.dump(known, fh, indent=1)
        fh.write("\n")

    # --- oracle regression pin on MUSI --------------------------------------
    from oracle import c_oracle, hammock_oracle as po
    seqs = po.load_unique_sequences_from_fasta(os.path.join(HERE, "musi.fa"))
    thr, X, maxc = po.greedy_defaults(seqs)
    po.sort_sequences(seqs, "size")
    strings = [s.get_sequence_string() for s in seqs]
    res, off = c_oracle.pack(strings)
    size = [s.size() for s in seqs]
    st, cid, order, stats = c_oracle.greedy_cluster(mats["blosum62"], res, off, size, 0, X, 0, thr, maxc, 1)
    assert st == 0
    with open(os.path.join(HERE, "musi_greedy_oracle.json"), "w") as fh:
        json.dump({"note": "produced by oracle/hammock_oracle.c, NOT by the Java reference",
                   "threshold": thr, "max_shift": X, "max_clusters": maxc,
                   "order": strings, "cluster_id": cid.tolist(), "result_order": order.tolist(),
                   "member_rank": stats.member_rank.tolist(),
                   "score_calls_phase1": int(stats.score_calls_phase1),
                   "score_calls_phase2": int(stats.score_calls_phase2),
                   "phase1_stop_index": int(stats.phase1_stop_index)}, fh, separators=(",", ":"))
        fh.write("\n")
    print("musi: thr", thr, "X", X, "maxc", maxc, "stop", stats.phase1_stop_index, "calls",
          stats.score_calls_phase1, stats.score_calls_phase2, "clusters", stats.n_multi,
          "result", stats.n_result_clusters)


if __name__ == "__main__":
    main()
