"""Synthetic PCA-like embeddings used by the tests, the golden generator and bench.py.

Shape follows SURVEY.md section 8(d): cells drawn around 32 shared Gaussian cluster
centres, component variance decaying linearly from 30 to 1 (like PCA scores), float64.
Only elementwise IEEE operations and numpy's PCG64 ``default_rng`` stream are used so
the arrays are bit-identical across numpy 1.26 / 2.x (the golden fixtures store a
sha256 of the generated inputs and the tests re-check it).
"""
import hashlib

import numpy as np

N_CLUSTERS = 32


def pca_like(n, d, seed, centre_seed=7, n_clusters=N_CLUSTERS, dtype=np.float64):
    """n cells x d components, float64, C-contiguous."""
    crng = np.random.default_rng(centre_seed)
    centres = crng.standard_normal((n_clusters, d)) * 2.0
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, n_clusters, size=n)
    z = rng.standard_normal((n, d))
    j = np.arange(d, dtype=np.float64)
    lam = 30.0 - 29.0 * j / float(max(d - 1, 1))
    out = (centres[lab] + z) * np.sqrt(lam)
    return np.ascontiguousarray(out, dtype=dtype)


def digest(a):
    """Short sha256 of an array's bytes (fixtures pin their regenerated inputs with it)."""
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]
