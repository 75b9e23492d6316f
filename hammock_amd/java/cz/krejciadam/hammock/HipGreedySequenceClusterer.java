/*
 * Drop-in for LimitedGreedySequenceClusterer (LimitedGreedySequenceClusterer.java:17-121): same
 * constructor shape, same cluster() contract, the whole pair space scored on one MI355X.
 * Swap it in at Hammock.java:402-403 (two lines, no casts):
 *
 *   AligningSequenceScorer scorer = new HipShiftedScorer(scoringMatrix, shiftPenalty, maxShift);
 *   SequenceClusterer clusterer = new HipGreedySequenceClusterer(scorer, sequenceClusteringThreshold, initialClustersLimit);
 *
 * Any other AligningSequenceScorer is accepted as well and simply runs the reference's own
 * LimitedGreedySequenceClusterer, so the class is a strict superset of the one it replaces.
 *
 * SOURCE ONLY (no JDK in the build image), see HipNative.java.
 */
package cz.krejciadam.hammock;

import java.util.ArrayList;
import java.util.HashMap;
import java.util.List;
import java.util.Map;
import java.util.concurrent.ExecutionException;

public class HipGreedySequenceClusterer implements SequenceClusterer {

    private final AligningSequenceScorer anyScorer;
    private final HipShiftedScorer sequenceScorer;   // non-null: the native path
    private final int threshold;
    private final int maxClusters;

    /** Same signature as LimitedGreedySequenceClusterer(AligningSequenceScorer, int, int), LimitedGreedySequenceClusterer.java:22. */
    public HipGreedySequenceClusterer(AligningSequenceScorer sequenceScorer, int threshold, int maxClusters) {
        this.anyScorer = sequenceScorer;
        this.sequenceScorer = sequenceScorer instanceof HipShiftedScorer ? (HipShiftedScorer) sequenceScorer : null;
        this.threshold = threshold;
        this.maxClusters = maxClusters;
    }

    /**
     * Returns the same List the reference returns: clusters with more than one member first, in
     * creation order with id = index of the seed (LimitedGreedySequenceClusterer.java:82), then the
     * remaining singletons; the Cluster objects hold the caller's UniqueSequence instances in
     * insertion order.
     */
    @Override
    public List<Cluster> cluster(List<UniqueSequence> sequences) throws InterruptedException, ExecutionException, DataException {
        if (sequenceScorer == null) {   // not a GPU scorer: the reference's own clusterer
            return new LimitedGreedySequenceClusterer(anyScorer, threshold, maxClusters).cluster(sequences);
        }
        int n = sequences.size();
        synchronized (sequenceScorer) {
            HipShiftedScorer.upload(sequenceScorer.ctx, sequences);
            int[] clusterId = new int[Math.max(n, 1)];
            int[] resultOrder = new int[Math.max(n, 1)];
            int[] memberRank = new int[Math.max(n, 1)];
            int nResult = HipNative.greedyCluster(sequenceScorer.ctx, sequenceScorer.maxShift, sequenceScorer.shiftPenalty,
                    threshold, maxClusters, clusterId, resultOrder, memberRank);
            Map<Integer, UniqueSequence[]> members = new HashMap<>();
            int[] uniqueSize = new int[Math.max(n, 1)];
            for (int k = 0; k < n; k++) {
                uniqueSize[clusterId[k]]++;
            }
            for (int k = 0; k < n; k++) {
                UniqueSequence[] slot = members.get(clusterId[k]);
                if (slot == null) {
                    slot = new UniqueSequence[uniqueSize[clusterId[k]]];
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
