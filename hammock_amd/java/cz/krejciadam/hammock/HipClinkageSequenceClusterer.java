/*
 * Drop-in for ClinkageSequenceClusterer (ClinkageSequenceClusterer.java:21-124): same constructor shape, same
 * cluster() contract -- exact complete linkage by nearest-neighbour chain, cluster ids and the order of the returned
 * list as the reference produces them (Java 8+ HashSet iteration order) -- with the whole pair space scored on the GPU.
 * Swap it in at Hammock.java:458-459:
 *
 *   ShiftedScorer -> HipShiftedScorer scorer = new HipShiftedScorer(scoringMatrix, shiftPenalty, maxShift);
 *   clusterer = new HipClinkageSequenceClusterer(scorer, sequenceClusteringThreshold);
 *
 * SOURCE ONLY (no JDK in the build image), see HipNative.java.
 */
package cz.krejciadam.hammock;

import java.util.ArrayList;
import java.util.HashMap;
import java.util.List;
import java.util.Map;
import java.util.concurrent.ExecutionException;

public class HipClinkageSequenceClusterer implements SequenceClusterer {

    private final SequenceScorer anyScorer;
    private final HipShiftedScorer sequenceScorer;   // non-null: the native path
    private final int threshold;

    /** Same signature as ClinkageSequenceClusterer(SequenceScorer, int), ClinkageSequenceClusterer.java:29. */
    public HipClinkageSequenceClusterer(SequenceScorer sequenceScorer, int threshold) {
        this.anyScorer = sequenceScorer;
        this.sequenceScorer = sequenceScorer instanceof HipShiftedScorer ? (HipShiftedScorer) sequenceScorer : null;
        this.threshold = threshold;
    }

    @Override
    public List<Cluster> cluster(List<UniqueSequence> sequences) throws InterruptedException, ExecutionException, DataException {
        if (sequenceScorer == null) {   // not a GPU scorer: the reference's own clusterer
            return new ClinkageSequenceClusterer(anyScorer, threshold).cluster(sequences);
        }
        int n = sequences.size();
        synchronized (sequenceScorer) {
            HipShiftedScorer.upload(sequenceScorer.ctx, sequences);
            HipNative.setJavaHashset(sequenceScorer.ctx, hashSetOrderOfThisJvm());
            int[] clusterId = new int[Math.max(n, 1)];
            int[] resultOrder = new int[Math.max(n, 1)];
            int[] memberRank = new int[Math.max(n, 1)];
            int nResult = HipNative.clinkageCluster(sequenceScorer.ctx, sequenceScorer.maxShift, sequenceScorer.shiftPenalty,
                    threshold, clusterId, resultOrder, memberRank);
            Map<Integer, Integer> uniqueSize = new HashMap<>();
            for (int k = 0; k < n; k++) {
                Integer c = uniqueSize.get(clusterId[k]);
                uniqueSize.put(clusterId[k], c == null ? 1 : c + 1);
            }
            Map<Integer, UniqueSequence[]> members = new HashMap<>();
            for (int k = 0; k < n; k++) {
                UniqueSequence[] slot = members.get(clusterId[k]);
                if (slot == null) {
                    slot = new UniqueSequence[uniqueSize.get(clusterId[k])];
                    members.put(clusterId[k], slot);
                }
                slot[memberRank[k]] = sequences.get(k);
            }
            List<Cluster> result = new ArrayList<>(nResult);
            for (int q = 0; q < nResult; q++) {
                List<UniqueSequence> l = new ArrayList<>();
                for (UniqueSequence s : members.get(resultOrder[q])) {
                    l.add(s);
                }
                result.add(new Cluster(l, resultOrder[q]));
            }
            return result;
        }
    }

    /**
     * The java.util.HashSet iteration order ClinkageSequenceClusterer would see on the JVM this runs in
     * (activeClusters.iterator().next(), ClinkageSequenceClusterer.java:70): 8 for Java 8 and later,
     * 7 for 7u6 .. 7u80, 6 for anything older. Overridden by -Dhammock.hip.java_hashset=6|7|8.
     */
    static int hashSetOrderOfThisJvm() {
        String forced = System.getProperty("hammock.hip.java_hashset");
        if (forced != null) {
            return Integer.parseInt(forced.trim());
        }
        String v = System.getProperty("java.version", "1.8.0");
        if (!v.startsWith("1.")) {
            return 8;                       // 9, 10, 11 ... : the Java 8 HashMap
        }
        if (v.startsWith("1.8")) {
            return 8;
        }
        if (v.startsWith("1.7")) {
            int us = v.indexOf('_');
            int update = 0;
            if (us >= 0) {
                int end = us + 1;
                while (end < v.length() && Character.isDigit(v.charAt(end))) {
                    end++;
                }
                if (end > us + 1) {
                    update = Integer.parseInt(v.substring(us + 1, end));
                }
            }
            return update >= 6 ? 7 : 6;
        }
        return 6;
    }
}
