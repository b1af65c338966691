#!/usr/bin/env python3
"""prof_phases.py -- where wave 0 of an encoder workgroup spends its cycles (diagnostic).
  EXTRA_FLAGS=-DS5_PHASE_PROF tools/build_variant.sh phaseprof && S5FXP_LIB=$PWD/tools/bin/phaseprof/libs5fxp.so python tools/prof_phases.py
Marks: csrc/proj_p.hpp PHASE_MARK (each mark waits for the work before it, then reads the shader clock)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import fxp_oracle as O  # noqa: E402
from sparsernns_amd import _lib, synth  # noqa: E402
from sparsernns_amd.fxparray import FxpArray  # noqa: E402
from sparsernns_amd.fxpmodel import build_regression_model  # noqa: E402

md, qc, dims = synth.make_model(0.5, state_headroom_bits=1)
model = build_regression_model(md, qc, dims["n_layers"])
eng = model.engine()
B, L = 32, 4096
x = synth.make_input(B, L, dims["d_in"], seed=1)
fx = O.from_fp(x, qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], True, O.FLOOR)
xa = FxpArray(fx.data, fx.bits, fx.exp)
for _ in range(3):
    eng.forward(xa)
torch.cuda.synchronize()
n = 2048 * 8
buf = (C.c_longlong * n)()
_lib.lib.s5fxp_debug_phase_prof.restype = C.c_int
assert _lib.lib.s5fxp_debug_phase_prof(buf, n) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(2048, 8)
a = a[a.sum(axis=1) > 0]
names = ["loop top / closing barrier", "A: prefetched rows converted", "A: rows requested at the top", "A: tail column, prefetch issue",
         "mid barrier", "B: operand reads + MFMA chain", "B: epilogue + store issue", "-"]
tot = a.sum(axis=1).mean()
print(f"{len(a)} workgroups, mean cycles per workgroup {tot:.0f}")
for i, nm in enumerate(names[:7]):
    print(f"  {nm:34s} {a[:, i].mean():9.0f} cycles  {100 * a[:, i].mean() / tot:5.1f} %")
