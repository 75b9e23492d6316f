"""hammock_amd -- MI355X-native drop-in for the greedy initial-clustering hot
path of krejciadam/hammock (pairwise ShiftedScorer / LocalAlignmentScorer
scoring + LimitedGreedySequenceClusterer), behind the C ABI of
include/hammock_hip.h.  See DESIGN.md and INTEGRATION.md."""
from .api import (AMINO_ACIDS, Cluster, Context, DataException, DeviceError, FileFormatException,
                  HammockException, HipClinkageSequenceClusterer, HipGreedySequenceClusterer, LocalAlignmentScorer, ReferenceWouldCrash,
                  ShiftedScorer, UniqueSequence, edge_fields, encode, pack_edges, pack_sequences)

__all__ = ["AMINO_ACIDS", "Cluster", "Context", "DataException", "DeviceError", "FileFormatException",
           "HammockException", "HipClinkageSequenceClusterer", "HipGreedySequenceClusterer", "LocalAlignmentScorer", "ReferenceWouldCrash",
           "ShiftedScorer", "UniqueSequence", "edge_fields", "encode", "pack_edges", "pack_sequences"]
