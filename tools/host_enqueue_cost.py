"""How long the host spends enqueuing one forward (Python + ctypes + HIP launches), vs the GPU time per forward."""
import sys, time
sys.path.insert(0, ".")
import torch
import __graft_entry__ as g
g.build()
from sparsernns_amd import synth, _lib
from sparsernns_amd.engine import InflightRunner
from sparsernns_amd.fxparray import RoundingMode, fxp_from_fp
from sparsernns_amd.fxpmodel import build_regression_model
B, L = 32, 4096
md, qc, dims = synth.make_model(0.5, calib_L=1024, state_headroom_bits=1)
model = build_regression_model(md, qc, dims["n_layers"]); eng = model.engine()
fx = fxp_from_fp(synth.make_input(B, L, dims["d_in"], seed=1), bits=qc["encoder"]["inp_bits"], exp=qc["encoder"]["inp_exp"], signed=True, round_mode=RoundingMode.FLOOR)
ys = [torch.empty((B, L, dims["d_out"]), dtype=torch.int32, device="cuda") for _ in range(3)]
r = InflightRunner(eng, 3)
for k in range(6): r.submit(fx.data, fx.bits, fx.exp, ys[k % 3], B, L, check=False)
torch.cuda.synchronize()
n = 60
t0 = time.perf_counter()
for k in range(n): r.submit(fx.data, fx.bits, fx.exp, ys[k % 3], B, L, check=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {(t1 - t0) / n * 1e6:.0f} us per forward; total {(t2 - t0) / n * 1e6:.0f} us per forward")
