import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import cref, fxp_oracle as O
from sparsernns_amd import synth, _lib
from sparsernns_amd.fxpmodel import build_regression_model
md, qc, dims = synth.make_model(dim_scale=1.0, calib_L=256, state_headroom_bits=1)
def mk():
    return build_regression_model(md, qc, dims["n_layers"]).engine()
eng = mk()
os.environ["S5FXP_NO_COMPACT"]="1"; eng2 = mk(); del os.environ["S5FXP_NO_COMPACT"]
cm = cref.CModel(build_regression_model(md, qc, dims["n_layers"]).export())
B,L=3,333
x = synth.make_input(B, L, dims["d_in"], seed=21)
fx = O.from_fp(x, qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], True, O.FLOOR)
ref,_,_,tr = cm.forward(fx.data, fx.bits, fx.exp, trace=True)
print("tops", [max(int(np.abs(t["xs_re"]).max()), int(np.abs(t["xs_im"]).max())) for t in tr], "bounds", [_lib.lib.s5fxp_model_recurrence_xmax(eng._h,i) for i in range(3)])
xd = torch.from_numpy(fx.data).cuda()
for name,e in (("compact",eng),("full",eng2)):
    for flags in (1, 5, 0, 2):
        y = torch.empty((B,L,dims["d_out"]),dtype=torch.int32,device="cuda")
        e.enqueue(xd, fx.bits, fx.exp, y, B, L, flags=flags)
        st = e.lane_status(0).cpu().numpy()
        print(name, "flags",flags,"status0",hex(int(st[0])),"slots",[int(st[8+8*i+6]) for i in range(3)],"rung",[int(st[8+8*i+5]) for i in range(3)],"equal",bool(np.array_equal(y.cpu().numpy(),ref)), "mism", int((y.cpu().numpy()!=ref).sum()))
