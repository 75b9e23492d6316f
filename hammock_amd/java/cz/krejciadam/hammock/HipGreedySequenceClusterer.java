/*
 * Drop-in for LimitedGreedySequenceClusterer (LimitedGreedySequenceClusterer.java:17-121): same
 * constructor shape, same cluster() contract, the whole pair space scored on one MI355X.
 * Swap it in at Hammock.java:403:
 *
 *   AligningSequenceScorer scorer = new HipShiftedScorer(scoringMatrix, shiftPenalty, maxShift);
 *   SequenceClusterer clusterer = new HipGreedySequenceClusterer((HipShiftedScorer) scorer,
 *           sequenceClusteringThreshold, initialClustersLimit);
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

    private final HipShiftedScorer sequenceScorer;
    private final int threshold;
    private final int maxClusters;

    public HipGreedySequenceClusterer(HipShiftedScorer sequenceScorer, int threshold, int maxClusters) {
        this.sequenceScorer = sequenceScorer;
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
