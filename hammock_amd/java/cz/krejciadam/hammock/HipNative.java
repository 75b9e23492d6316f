/*
 * JNI binding of libhammock_hip.so (include/hammock_hip.h) for Hammock.
 *
 * SHIPPED AS SOURCE, NOT COMPILED IN THIS REPOSITORY'S BUILD: the build image has no JDK
 * (no javac, no jni.h).  A Hammock maintainer drops the four Hip*.java files next to the
 * other classes of package cz.krejciadam.hammock, compiles jni/hammock_jni.c against
 * $JAVA_HOME/include and puts libhammock_jni.so + libhammock_hip.so on java.library.path.
 * INTEGRATION.md has the build line and the one-line change in Hammock.java.
 */
package cz.krejciadam.hammock;

final class HipNative {

    static {
        System.loadLibrary("hammock_jni"); // links libhammock_hip.so
    }

    private HipNative() {
    }

    /** hmk_create: matrix is int[24][24] flattened row-major (Hammock.scoringMatrix). Returns the context handle. */
    static native long create(int[] matrix576, int device);

    /** hmk_create_multi: the pair space sharded over several GPUs of the node; devices[0] runs the merge. */
    static native long createMulti(int[] matrix576, int[] devices);

    /** hmk_destroy */
    static native void destroy(long ctx);

    /** hmk_set_sequences: residues concatenated, offsets[n + 1], sizes[n] = UniqueSequence.size(). */
    static native void setSequences(long ctx, byte[] residues, int[] offsets, int[] sizes);

    /** hmk_reserve: sizes the result buffers for n sequences ahead of the first cluster() call (optional; e.g. while the input is read). */
    static native void reserve(long ctx, int nSequences);

    /**
     * hmk_set_java_hashset: which java.util.HashSet iteration order clinkageCluster emulates (8 = Java 8 and later,
     * 7 = JDK 7u6 .. 7u80, 6 = JDK 6 and JDK 7 before 7u6). HipClinkageSequenceClusterer passes the version of the running JVM.
     */
    static native void setJavaHashset(long ctx, int version);

    /** hmk_score_with_shift for one pair (i, j): returns {score, shift}. Throws DataException ("Shift too big"). */
    static native int[] scoreWithShift(long ctx, int i, int j, int maxShift, int shiftPenalty) throws DataException;

    /** hmk_score_pairs_local for one pair. */
    static native int scoreLocal(long ctx, int i, int j, int gapOpen, int gapExtend);

    /**
     * hmk_greedy_cluster. Fills clusterId[n], resultOrder[n], memberRank[n]; returns the number of valid
     * entries of resultOrder. Throws NullPointerException where LimitedGreedySequenceClusterer would
     * (LimitedGreedySequenceClusterer.java:97/104/108), DataException for "Shift too big".
     */
    static native int greedyCluster(long ctx, int maxShift, int shiftPenalty, int threshold, int maxClusters,
            int[] clusterId, int[] resultOrder, int[] memberRank) throws DataException;

    /**
     * hmk_clinkage_cluster (sequences in load order). Same outputs as greedyCluster; cluster ids as
     * ClinkageSequenceClusterer assigns them. Throws NoSuchElementException for an empty input
     * (ClinkageSequenceClusterer.java:118), DataException for "Shift too big".
     */
    static native int clinkageCluster(long ctx, int maxShift, int shiftPenalty, int threshold,
            int[] clusterId, int[] resultOrder, int[] memberRank) throws DataException;
}
