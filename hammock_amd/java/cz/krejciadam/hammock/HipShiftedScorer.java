/*
 * ShiftedScorer on the GPU. Same constructor shape and semantics as ShiftedScorer.java:28-32 / :48-100.
 * A batch-backed parity probe: one JNI call per pair is slow by design; the fast path is
 * HipGreedySequenceClusterer, which scores the whole pair space in one call.
 * SOURCE ONLY (no JDK in the build image), see HipNative.java.
 */
package cz.krejciadam.hammock;

public class HipShiftedScorer implements AligningSequenceScorer, AutoCloseable {

    final int[][] scoringMatrix;
    final int shiftPenalty;
    final int maxShift;
    long ctx;

    /**
     * Same shape as ShiftedScorer(scoringMatrix, shiftPenalty, maxShift) (ShiftedScorer.java:28-32). The GPU(s) come
     * from the system property hammock.hip.devices, a comma separated list of HIP ordinals (default "0"); more than
     * one shards the pair space over them (hmk_create_multi), the first one runs the merge:
     * java -Dhammock.hip.devices=0,1,2,3,4,5,6,7 -jar Hammock.jar greedy ...
     */
    public HipShiftedScorer(int[][] scoringMatrix, int shiftPenalty, int maxShift) {
        this(scoringMatrix, shiftPenalty, maxShift, devicesFromProperty());
    }

    public HipShiftedScorer(int[][] scoringMatrix, int shiftPenalty, int maxShift, int[] devices) {
        this.scoringMatrix = scoringMatrix;
        this.shiftPenalty = shiftPenalty;
        this.maxShift = maxShift;
        this.ctx = devices.length == 1 ? HipNative.create(flatten(scoringMatrix), devices[0])
                                       : HipNative.createMulti(flatten(scoringMatrix), devices);
    }

    static int[] devicesFromProperty() {
        String[] parts = System.getProperty("hammock.hip.devices", "0").split(",");
        int[] devices = new int[parts.length];
        for (int k = 0; k < parts.length; k++) {
            devices[k] = Integer.decode(parts[k].trim());
        }
        return devices;
    }

    /** Releases the native context (device buffers, streams). The scorer must not be used afterwards. */
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
