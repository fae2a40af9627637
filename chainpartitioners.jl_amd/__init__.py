"""chainpartitioners.jl_amd -- MI355X-native engine for the contiguous sparse-matrix
partitioning hot path of ChainPartitioners.jl (partition_stripe / pack_stripe with the
Dynamic*, BisectCost and Convex methods over the Work / Connectivity / HyperedgeCut /
Block cost oracles).  Host side mirrors the reference's dispatch API; the compute runs in
hand-written HIP kernels behind the C ABI of include/chainpart.h (csrc/).
"""
from .types import *          # noqa: F401,F403
from .types import to_map, to_domain   # noqa: F401
from .models import (PowerWorkModel, ConvexWorkModel, ConcaveWorkModel,   # noqa: F401
                     AffineWorkModel, AffinePrimaryConnectivityModel, AffineSecondaryConnectivityModel, AffineConnectivityModel, AffineHyperedgeCutModel,      # noqa: F401
                     ColumnBlockComponentCostModel, BlockComponentCostModel, VertexCount, FeasibleCost,
                     ConstrainedCost, EquiSplitter, EquiChunker, DynamicTotalSplitter,
                     DynamicBottleneckSplitter, DynamicTotalChunker, DynamicBottleneckChunker,
                     ReferenceTotalSplitter, ReferenceBottleneckSplitter, ReferenceTotalChunker,
                     BisectCostBottleneckSplitter, FlipBisectCostBottleneckSplitter,
                     BisectIndexBottleneckSplitter, FlipBisectIndexBottleneckSplitter,
                     LazyBisectCostBottleneckSplitter, DisjointPartitioner, AlternatingPartitioner,
                     AlternatingNetPartitioner, SymmetricPartitioner,
                     ConvexTotalChunker, ConvexTotalSplitter, ConcaveTotalChunker, ConcaveTotalSplitter)
from . import _lib  # noqa: F401
from .api import (adjointpattern, partition_plaid, partition_stripe, partition_stripe_batch, pack_stripe, pack_stripe_batch, oracle_stripe, bound_stripe, total_value,   # noqa: F401
                  bottleneck_value, netcount, selfnetcount, dominancecount, set_default_backend,
                  get_backend, CPError, Step, Same, Next, Prev, Jump)
