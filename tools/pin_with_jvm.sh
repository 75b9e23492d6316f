#!/bin/bash
# The five-minute pin: run the REAL Hammock on the two reference inputs this repository holds and compare its stage-1 result
# files with hammock-hip's.  It needs what this build never had: a JVM and a built Hammock.jar (the reference's own jar -- no
# stub classes, no stand-ins).  It cannot run in the build container or on the GPU boxes of this project (no JDK there), so
# it has never been run; it turns "parity unpinned" into one command for whoever has both.
#
#   tools/pin_with_jvm.sh /path/to/Hammock.jar [java-binary] [extra hammock-hip flags, e.g. --java_hashset 7]
#
# For each of tests/golden/musi.fa and tests/golden/antibodies.fa(.gz) and each of the modes greedy and clinkage:
#   java -jar Hammock.jar <mode> -i <input> -d <out_ref> -t 1          (Hammock.java:392-437, :449-489)
#   hammock-hip <mode> -i <input> -d <out_hip> [flags]
# then the three stage-1 TSVs are compared after the Clustal `alignment` column (column 3 of the two *_sequences files) has been
# cut from both: the reference fills it from Clustal Omega for multi-member clusters (external binary, out of scope), hammock-hip
# writes NA there.  Exit status 0 = every file identical.  clinkage on antibodies.fa (74,041 sequences) is skipped: the
# reference's clinkage mode is meant for <= 10,000 sequences (Hammock.java:371-377) and takes hours there.
#
# If `clinkage` differs while `greedy` agrees, try --java_hashset 7 (JDK 7u6+) or 6 (JDK 6 / early 7): the chain starts and the
# list order of clinkage are java.util.HashSet iteration orders, which Java 8 changed ("clinkage").
set -u
JAR=${1:?usage: tools/pin_with_jvm.sh /path/to/Hammock.jar [java] [hammock-hip flags]}
JAVA=${2:-java}
shift; shift 2>/dev/null || true
FLAGS="$*"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
HIP=$ROOT/hammock_amd/bin/hammock-hip
command -v "$JAVA" >/dev/null || { echo "no JVM ($JAVA): this script needs the real reference, it does not fake one"; exit 2; }
[ -f "$JAR" ] || { echo "no such jar: $JAR"; exit 2; }
[ -x "$HIP" ] || make -C "$ROOT/hammock_amd/csrc" -j4 >/dev/null || exit 2
W=$(mktemp -d)
gunzip -c "$ROOT/tests/golden/antibodies.fa.gz" > "$W/antibodies.fa"
cp "$ROOT/tests/golden/musi.fa" "$W/musi.fa"
cut_alignment() {   # drop column 3 of the *_sequences files, keep the clusters file as it is
    case "$1" in *sequences*) cut -f1,2,4- "$1" ;; *) cat "$1" ;; esac
}
status=0
for input in musi antibodies; do
  for mode in greedy clinkage; do
    [ "$input" = antibodies ] && [ "$mode" = clinkage ] && continue
    ref=$W/ref_${input}_$mode; hip=$W/hip_${input}_$mode
    echo "== $mode on $input.fa"
    "$JAVA" -jar "$JAR" $mode -i "$W/$input.fa" -d "$ref" -t 1 > "$W/ref_${input}_$mode.log" 2>&1 || { echo "   the reference failed: see $W/ref_${input}_$mode.log"; status=1; continue; }
    "$HIP" $mode -i "$W/$input.fa" -d "$hip" $FLAGS > "$W/hip_${input}_$mode.log" 2>&1 || { echo "   hammock-hip failed: see $W/hip_${input}_$mode.log"; status=1; continue; }
    grep -h "Clustering time" "$ref/run.log" "$hip/run.log" | sed 's/^/   /'
    for f in initial_clusters.tsv initial_clusters_sequences.tsv initial_clusters_sequences_original_order.tsv input_statistics.tsv; do
      if cmp -s <(cut_alignment "$ref/$f") <(cut_alignment "$hip/$f"); then echo "   identical: $f"
      else echo "   DIFFERENT: $f   (diff <(cut -f1,2,4- $ref/$f) <(cut -f1,2,4- $hip/$f))"; status=1; fi
    done
  done
done
echo "work directory kept: $W"
[ $status = 0 ] && echo "PINNED: every stage-1 file of the reference is reproduced." || echo "NOT identical: the differences above are the first thing to look at."
exit $status
