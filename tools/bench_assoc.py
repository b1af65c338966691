#!/usr/bin/env python3
"""bench_assoc.py -- the float model's time-parallel scan (csrc/scan_assoc.hpp) alone: launch duration and the fraction
of the 8 TB/s HBM peak on its bytes (8*P in + 8*P out per frame: complex64 Bu read once, states written once).
  python tools/bench_assoc.py            (on an MI355X box)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsernns_amd import ssm  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    for B, L, P in ((1, 1024, 64), (32, 4096, 64), (32, 4096, 128), (128, 4096, 64), (512, 4096, 128)):
        lam = torch.as_tensor((0.99 * np.exp(1j * rng.uniform(-3, 3, P))).astype(np.complex64)).cuda()
        bu = torch.randn(B, L, P, dtype=torch.complex64, device="cuda")
        for _ in range(3):
            ssm.associative_scan(lam, bu)
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ssm.associative_scan(lam, bu)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        byts = 16.0 * B * L * P
        print(f"B={B:4d} L={L} P={P:4d}: {us:8.1f} us per scan (incl. the output allocation)  {byts / us / 1e3:8.1f} GB/s  "
              f"{byts / us / 1e3 / 8000:.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
