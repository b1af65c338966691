"""Experiment: two independent forwards in flight on two HIP streams vs back to back on one."""
import sys, time
import torch
sys.path.insert(0, ".")
import __graft_entry__ as g
g.build()
from sparsernns_amd import synth
from sparsernns_amd.fxparray import RoundingMode, fxp_from_fp
from sparsernns_amd.fxpmodel import build_regression_model

B, L = 32, 4096
md, qc, dims = synth.make_model(0.5, calib_B=2, calib_L=1024, state_headroom_bits=1)
engs = []
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for i in range(NS):
    model = build_regression_model(md, qc, dims["n_layers"])
    eng = model.engine()
    x = synth.make_input(B, L, dims["d_in"], seed=1000 + i)
    fx = fxp_from_fp(x, bits=qc["encoder"]["inp_bits"], exp=qc["encoder"]["inp_exp"], signed=True, round_mode=RoundingMode.FLOOR)
    y = torch.empty((B, L, dims["d_out"]), dtype=torch.int32, device="cuda")
    engs.append((eng, fx, y, model))
streams = [torch.cuda.Stream() for _ in range(NS)]

def run(n, concurrent):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        for i, (eng, fx, y, _) in enumerate(engs):
            s = streams[i] if concurrent else streams[0]
            with torch.cuda.stream(s):
                eng.enqueue(fx.data, fx.bits, fx.exp, y, B, L, flags=1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (NS * n) * 1e3

for c in (False, True, False, True):
    run(3, c)
    print("concurrent" if c else "serial    ", f"{run(20, c):.4f} ms per forward")
for eng, *_ in engs:
    print(eng.check_status()[:1])
