import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle import oracle
from upsp_processing_amd import engine
from test_interpolate_gpu import _surface
for ns, nq, k in ((5, 40, 10), (300, 500, 10)):
    src = _surface(ns, ns); data = np.sin(src[:, 0]) + 0.3 * src[:, 1]
    qry = _surface(nq, nq + 1, spread=(8.5, 2.2, 0.35))
    want, wn = oracle.interpolate_idw(src, data, qry, k, 2.0)
    got, gn = engine.interpolate_idw(src, data, qry, k, 2.0, want_neighbors=True)
    gn = gn.cpu().numpy()
    bad = np.nonzero((gn != wn).any(1))[0]
    print(ns, "bad rows", bad.size, "of", nq)
    for b in bad[:3]:
        print(" q", b, qry[b], "\n  gpu", gn[b], "\n  orc", wn[b])
