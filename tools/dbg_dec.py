import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import cref, fxp_oracle as O
from sparsernns_amd import synth
from sparsernns_amd.fxparray import FxpArray
from sparsernns_amd.fxpmodel import build_regression_model
md, qc, dims = synth.make_model(dim_scale=1.0, calib_L=128)
model = build_regression_model(md, qc, 3)
x = synth.make_input(2, 96, 257, seed=5)
fx = O.from_fp(x, 16, qc['encoder']['inp_exp'], True, O.FLOOR)
ref, rb, re_, rtr = cref.CModel(model.export()).forward(fx.data, 16, fx.exp, trace=True)
eng = model.engine()
y = eng.forward(FxpArray(fx.data, 16, fx.exp)).numpy()
bad = (y != ref)
print('bad total', bad.sum(), 'of', bad.size)
print('bad per column (nonzero):', {int(c): int(v) for c, v in enumerate(bad.sum(axis=(0,1))) if v})
print('bad per frame first 40:', bad.reshape(-1, 257).sum(axis=1)[:40])
print('exps', eng.layer_exponents(), 'dec qc', {k: qc['decoder'][k] for k in ('inp_bits','inp_exp','w_exp','out_exp')})
