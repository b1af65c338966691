"""Op-level timing: s5fxp_dense vs s5fxp_dense_csr on a 90 %-pruned encoder-sized layer (N=131072, K=257, M=96)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sparsernns_amd.fxparray import CsrWeight, FxpArray, fxp_matmul, fxp_matmul_csr
rng = np.random.default_rng(0)
for (N, K, M) in ((131072, 257, 96), (131072, 96, 128), (131072, 96, 257)):
    w = rng.integers(-127, 128, size=(K, M)).astype(np.int32); w[rng.random((K, M)) < 0.9] = 0
    x = FxpArray(torch.from_numpy(rng.integers(-32767, 32768, size=(N, K)).astype(np.int32)).cuda(), 16, 10)
    wd, wc = FxpArray(torch.from_numpy(w).cuda(), 8, 7), CsrWeight(w, 8, 7)
    for name, fn in (("dense int32 VALU", lambda: fxp_matmul(x, wd, result_bits=16, result_exp=9)),
                     ("csr", lambda: fxp_matmul_csr(x, wc, result_bits=16, result_exp=9))):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print(f"N={N} K={K} M={M} density={wc.density:.2f} {name:18s} {(time.perf_counter()-t0)/10*1e6:8.1f} us")
