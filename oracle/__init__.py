"""CPU oracle for the k-NN mapping hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / reported CPU baseline (never as the thing shipped).
See nabo_oracle.c for the reference citations and how parity is pinned.
"""
from .oracle import (build, pairwise, knn, snn_edges, snn_weight, max_threads, score_null,  # noqa: F401
                     EUCLIDEAN, MOD_CANBERRA, COSINE)
