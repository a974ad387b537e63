"""nabo_amd -- MI355X-native k-NN mapping hot path behind Nabo's Mapping API.

Public names mirror the reference's `nabo` package for this path (`Mapping`), plus the
array-level entry points of the C ABI (`knn`, `pairwise`, `KnnIndex`)."""
from ._lib import EUCLIDEAN, MOD_CANBERRA, COSINE, NaboError, device_count  # noqa: F401
from ._knn import knn, knn_devices, pairwise, KnnIndex, snn_counts  # noqa: F401
from ._mapping import Mapping, write_dense_pca, expand_graph  # noqa: F401
from ._score import (get_mapping_score, mapping_score_from_edges, mapping_score_null,  # noqa: F401
                     get_mapping_score_null)

__all__ = ["Mapping", "write_dense_pca", "expand_graph", "get_mapping_score", "mapping_score_from_edges", "mapping_score_null", "get_mapping_score_null", "knn", "knn_devices", "pairwise", "KnnIndex", "snn_counts", "device_count", "EUCLIDEAN", "MOD_CANBERRA", "COSINE",
           "NaboError"]
