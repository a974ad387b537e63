import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nabo_amd, oracle
from nabo_amd._synth import pca_like
m, n, g, k = 64, 20000, 50, 15
Y = pca_like(n, g, seed=1000 + n + g); X = pca_like(m, g, seed=2000 + m + g)
oi, od = oracle.knn(X, Y, k, 0, nthreads=8)
ix = nabo_amd.KnnIndex(n, g, metric=0).set_ref(Y)
gi, gd = ix.query(X, k)
print(ix.last_kernel(), ix.last_stats())
bad = np.where((gi != oi).any(axis=1))[0]
print("bad rows", bad)
for r in bad[:6]:
    miss = sorted(set(oi[r]) - set(gi[r]))
    print(r, "missing", miss, "tiles", [j // 32 for j in miss], "pos in tile", [j % 32 for j in miss], "rank", [list(oi[r]).index(j) for j in miss])
