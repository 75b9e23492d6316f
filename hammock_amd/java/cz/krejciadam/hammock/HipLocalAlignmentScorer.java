/*
 * LocalAlignmentScorer on the GPU (LocalAlignmentScorer.java:20-29), direction-matrix gap rule included.
 * SOURCE ONLY (no JDK in the build image), see HipNative.java.
 */
package cz.krejciadam.hammock;

public class HipLocalAlignmentScorer implements SequenceScorer, AutoCloseable {

    private final int gapOpenPenalty;
    private final int gapExtendPenalty;
    private long ctx;

    /** LocalAlignmentScorer(scoringMatrix, gapOpenPenalty, gapExtendPenalty); GPU = first entry of hammock.hip.devices. */
    public HipLocalAlignmentScorer(int[][] scoringMatrix, int gapOpenPenalty, int gapExtendPenalty) {
        this.gapOpenPenalty = gapOpenPenalty;
        this.gapExtendPenalty = gapExtendPenalty;
        this.ctx = HipNative.create(HipShiftedScorer.flatten(scoringMatrix), HipShiftedScorer.devicesFromProperty()[0]);
    }

    @Override
    public synchronized void close() {
        if (ctx != 0) {
            HipNative.destroy(ctx);
            ctx = 0;
        }
    }

    @Override
    protected void finalize() throws Throwable {
        try {
            close();
        } finally {
            super.finalize();
        }
    }

    @Override
    public synchronized int sequenceScore(UniqueSequence seq1, UniqueSequence seq2) throws DataException {
        HipShiftedScorer.upload(ctx, java.util.Arrays.asList(seq1, seq2));
        return HipNative.scoreLocal(ctx, 0, 1, gapOpenPenalty, gapExtendPenalty);
    }
}
