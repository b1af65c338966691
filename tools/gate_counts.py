#!/usr/bin/env python3
"""gate_counts.py: with tools/bin/gate_count/libs5fxp.so (tools/variant.py gate_count -DS5_GATE_CHECK=1), run forwards of the
synthetic N-DNS models and read the device-side counters of the gating test (proj_p.hpp gate_check): how many MFMA operand
fragments of the gate kernel's two projections were entirely zero -- what an activation-gating kernel could skip.
Usage (GPU box): S5FXP_LIB=$PWD/tools/bin/gate_count/libs5fxp.so python3 tools/gate_counts.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    import torch
    from sparsernns_amd import _lib, synth
    from sparsernns_amd.fxparray import RoundingMode, fxp_from_fp
    from sparsernns_amd.fxpmodel import build_regression_model

    fn = _lib.lib.s5fxp_debug_gate_counts
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    out = (ctypes.c_ulonglong * 4)()
    for name, ds, sparsity, hb in (("configs[1] dim 0.5 dense", 0.5, 0.0, 1), ("configs[2] dim 0.5 90 % pruned", 0.5, 0.9, 2),
                                   ("configs[3] dim 1.0 90 % pruned", 1.0, 0.9, 2)):
        md, qc, dims = synth.make_model(ds, sparsity=sparsity, calib_L=1024, state_headroom_bits=hb)
        model = build_regression_model(md, qc, dims["n_layers"])
        eng = model.engine()
        B, L = 32, 4096
        x = fxp_from_fp(synth.make_input(B, L, dims["d_in"], seed=1000), bits=qc["encoder"]["inp_bits"], exp=qc["encoder"]["inp_exp"],
                        signed=True, round_mode=RoundingMode.FLOOR)
        assert fn(out, 1) == 0
        eng.forward(x)
        torch.cuda.synchronize()
        assert fn(out, 1) == 0
        c = [int(v) for v in out]
        print(f"{name}: C projection operand fragments {c[0]}, all-zero {c[1]} ({100.0 * c[1] / max(c[0], 1):.3f} %); "
              f"out2 operand fragments {c[2]}, all-zero {c[3]} ({100.0 * c[3] / max(c[2], 1):.3f} %)")


if __name__ == "__main__":
    main()
