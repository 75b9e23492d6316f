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
}
