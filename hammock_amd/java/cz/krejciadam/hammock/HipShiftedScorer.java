/*
 * ShiftedScorer on the GPU. Same constructor shape and semantics as ShiftedScorer.java:28-32 / :48-100.
 * A batch-backed parity probe: one JNI call per pair is slow by design; the fast path is
 * HipGreedySequenceClusterer, which scores the whole pair space in one call.
 * SOURCE ONLY (no JDK in the build image), see HipNative.java.
 */
package cz.krejciadam.hammock;

public class HipShiftedScorer implements AligningSequenceScorer {

    final int[][] scoringMatrix;
    final int shiftPenalty;
    final int maxShift;
    final long ctx;

    public HipShiftedScorer(int[][] scoringMatrix, int shiftPenalty, int maxShift) {
        this.scoringMatrix = scoringMatrix;
        this.shiftPenalty = shiftPenalty;
        this.maxShift = maxShift;
        this.ctx = HipNative.create(flatten(scoringMatrix), 0);
    }

    static int[] flatten(int[][] m) {
        int[] flat = new int[24 * 24];
        for (int r = 0; r < 24; r++) {
            System.arraycopy(m[r], 0, flat, r * 24, 24);
        }
        return flat;
    }

    static void upload(long ctx, java.util.List<UniqueSequence> sequences) {
        int total = 0;
        for (UniqueSequence s : sequences) {
            total += s.getSequence().length;
        }
        byte[] residues = new byte[total];
        int[] offsets = new int[sequences.size() + 1];
        int[] sizes = new int[sequences.size()];
        int pos = 0;
        for (int k = 0; k < sequences.size(); k++) {
            for (int r : sequences.get(k).getSequence()) {
                residues[pos++] = (byte) r;
            }
            offsets[k + 1] = pos;
            sizes[k] = sequences.get(k).size();
        }
        HipNative.setSequences(ctx, residues, offsets, sizes);
    }

    @Override
    public synchronized AligningScorerResult scoreWithShift(UniqueSequence seq1, UniqueSequence seq2) throws DataException {
        upload(ctx, java.util.Arrays.asList(seq1, seq2));
        int[] r = HipNative.scoreWithShift(ctx, 0, 1, maxShift, shiftPenalty);
        return new AligningScorerResult(r[0], r[1], seq2);
    }

    @Override
    public int sequenceScore(UniqueSequence seq1, UniqueSequence seq2) throws DataException {
        return scoreWithShift(seq1, seq2).getScore();
    }
}
