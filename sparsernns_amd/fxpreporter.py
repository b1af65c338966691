"""Per-stage error report of the fixed-point forward against float activations.

The counterpart of the reference's verification report (``sparseRNNs/fxpreporter.py:12-24`` ``compute_error``,
``:120-186`` ``Reporter.add_block_raw``, driven by ``sparseRNNs/fxprun.py:553-731``): for every stage the harness
looks at, the dequantised fixed-point intermediate (``xrec``) is compared with the float model's activation of the
same stage (``xhat``) -- absolute error mean / std / max / median and relative error mean / max / median over the
elements whose float value is not zero; complex stages are reported per part.  No plots (the reference's
matplotlib figures are out of scope): the report is a text table, a markdown file and a json.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List

import numpy as np

METRICS = ("abs_error_mean", "abs_error_std", "abs_error_max", "abs_error_med", "rel_error_mean", "rel_error_max",
           "rel_error_med")


def compute_error(xrec: np.ndarray, xhat: np.ndarray) -> Dict[str, float]:
    """fxpreporter.py:12-24.  xrec: reconstructed (fixed point -> float), xhat: the float model's value."""
    xrec, xhat = np.asarray(xrec, dtype=np.float64), np.asarray(xhat, dtype=np.float64)
    err = np.abs(xrec - xhat)
    nz = xhat != 0
    rel = err[nz] / np.abs(xhat[nz])
    return dict(abs_error_mean=float(err.mean()), abs_error_std=float(err.std()), abs_error_max=float(err.max()),
                abs_error_med=float(np.median(err)),
                rel_error_mean=float(rel.mean()) if rel.size else 0.0, rel_error_max=float(rel.max()) if rel.size else 0.0,
                rel_error_med=float(np.median(rel)) if rel.size else 0.0)


class Reporter:
    """Collects one block per stage (``add_block_raw``), prints a line per block, writes report.md / results.json."""

    def __init__(self, folder: str | None = None, title: str = "fixed point vs float, stage by stage", header: dict | None = None):
        self.folder, self.title, self.header = folder, title, dict(header or {})
        self.results_data: List[dict] = []

    @staticmethod
    def _line(m: Dict[str, float], xhat: np.ndarray) -> str:
        relmax = f"{m['rel_error_max']:8.3%}" if m["rel_error_max"] < 1.0 else f"{m['rel_error_max']:10.4f}"
        return (f"abs err: mean {m['abs_error_mean']:.3e}, std {m['abs_error_std']:.3e}, max {m['abs_error_max']:.3e}, "
                f"med {m['abs_error_med']:.3e} -- rel err: mean {m['rel_error_mean']:8.3%}, med {m['rel_error_med']:8.3%}, "
                f"max {relmax} -- xhat: mean {float(np.mean(xhat)):9.4f}, median {float(np.median(xhat)):8.4f}, "
                f"max {float(np.max(xhat)):8.4f}")

    def add_block_raw(self, name: str, xhat, xrec, verbose: bool = True, xhatname: str = "float", xrecname: str = "fxp") -> None:
        xhat, xrec = np.asarray(xhat), np.asarray(xrec)
        assert xhat.shape == xrec.shape, f"{name}: xhat {xhat.shape} and xrec {xrec.shape} must have the same shape"
        parts = [("", xhat, xrec)]
        if np.iscomplexobj(xhat) or np.iscomplexobj(xrec):  # fxpreporter.py:137-171: real and imaginary part separately
            parts = [(" (real)", np.real(xhat), np.real(xrec)), (" (imag)", np.imag(xhat), np.imag(xrec))]
        for suffix, h, r in parts:
            m = compute_error(xrec=r, xhat=h)
            line = self._line(m, h)
            if verbose:
                print(f"{name + suffix:<46} {line}")
            self.results_data.append(dict(name=name + suffix, compared=f"{xrecname} vs {xhatname}", line=line,
                                          xhat_absmax=float(np.max(np.abs(h))) if h.size else 0.0, **m))

    def worst(self, metric: str = "rel_error_med") -> dict:
        return max(self.results_data, key=lambda r: r[metric])

    def markdown(self) -> str:
        out = [f"# {self.title}", ""]
        for k, v in self.header.items():
            out.append(f"- {k}: {v}")
        out += ["", "| stage | " + " | ".join(METRICS) + " |", "|---|" + "---|" * len(METRICS)]
        for r in self.results_data:
            out.append(f"| {r['name']} | " + " | ".join(f"{r[m]:.4g}" for m in METRICS) + " |")
        return "\n".join(out) + "\n"

    def save(self) -> None:
        if not self.folder:
            return
        os.makedirs(self.folder, exist_ok=True)
        with open(os.path.join(self.folder, "report.md"), "w") as f:
            f.write(self.markdown())
        with open(os.path.join(self.folder, "results.json"), "w") as f:
            json.dump(dict(header=self.header, results=[{k: v for k, v in r.items() if k != "line"} for r in self.results_data]), f,
                      indent=1)


def verification_report(model, fx, x_float, activations: dict, reporter: Reporter, seq_len: int | None = None) -> Reporter:
    """The stage list of ``run_verification`` (sparseRNNs/fxprun.py:577-727) on an eager model that has just run ``fx``.

    model: an ``FxpRegressionModel`` built with ``store_intermediates=True`` AFTER ``model(fx)``; fx / x_float: the
    fixed-point and float inputs (L, d_in) of batch item 0; activations: the float model's intermediates as a tree in the
    reference's layout -- ``encoder/layers_i/{input, pre_s5, pre_C, pre_GLU, post_GLU, __call__}``,
    ``encoder/layers_i/mixer/{B_bar, __call__}``, ``encoder/layers_i/out2/__call__``, ``__call__`` -- each the value
    for that one sequence (the reference indexes its recorded lists with ``[0][0, :seq_len]``)."""
    def f(a):
        return np.asarray(a.to_float().cpu().numpy() if hasattr(a, "to_float") else a)

    def cplx(a):  # ComplexFxpArray -> complex64
        return f(a.real) + 1j * f(a.imag)

    def one(a):  # (1, L, ...) or (L, ...) -> (L, ...)
        a = np.asarray(a)
        return a[0] if a.ndim == 3 else a

    T = slice(None, seq_len)
    enc = activations["encoder"]
    reporter.add_block_raw("inputs", xhat=np.asarray(x_float)[T], xrec=one(f(fx))[T])
    n_layers = len(model.encoder.seq_layers)
    for i, layer in enumerate(model.encoder.seq_layers):
        act = enc[f"layers_{i}"]
        li, mi = layer.intermediates, layer.mixer.intermediates
        reporter.add_block_raw("encoder.encoder (post-relu)" if i == 0 else f"encoder.layers_{i}.input",
                               xhat=act["input"][T], xrec=one(f(li["ssm_input"][0]))[T])
        reporter.add_block_raw(f"encoder.layers_{i}.norm", xhat=act["pre_s5"][T], xrec=one(f(li["pre_s5"][0]))[T])
        # fxprun.py:611-628: B @ u is not stored by the float model; it is recomputed from u and B_bar
        bu = mi["Bu_elements"][0]
        reporter.add_block_raw(f"encoder.layers_{i}.mixer.Bu.calc_hat", xhat=(act["pre_s5"][T].astype(np.complex64) @ act["mixer"]["B_bar"].T),
                               xrec=one(cplx(bu))[T], xhatname="float (calc)")
        reporter.add_block_raw(f"encoder.layers_{i}.mixer.xt", xhat=act["pre_C"][T], xrec=one(cplx(mi["xs_relu"][0]))[T])
        reporter.add_block_raw(f"encoder.layers_{i}.mixer.yt", xhat=act["mixer"]["__call__"][T], xrec=one(f(mi["ys"][0]))[T])
        reporter.add_block_raw(f"encoder.layers_{i}.mixer.pre_glu", xhat=act["pre_GLU"][T], xrec=one(f(li["pre_GLU"][0]))[T])
        o2 = act["out2"]["__call__"][T]
        reporter.add_block_raw(f"encoder.layers_{i}.mixer.out2", xhat=o2, xrec=one(f(layer.out2.intermediates["__call__"][0]))[T])
        reporter.add_block_raw(f"encoder.layers_{i}.mixer.out2_sigmoid", xhat=1.0 / (1.0 + np.exp(-o2.astype(np.float64))),
                               xrec=one(f(li["out2_sigmoid"][0]))[T])
        reporter.add_block_raw(f"encoder.layers_{i}.mixer.post_glu", xhat=act["post_GLU"][T], xrec=one(f(li["post_GLU"][0]))[T])
        reporter.add_block_raw(f"encoder.layers_{i}.mixer.residadd", xhat=act["post_GLU"][T] + act["input"][T],
                               xrec=one(f(li["residadd"][0]))[T], xhatname="float (calc)")
        reporter.add_block_raw(f"encoder.layers_{i}.mixer.output", xhat=act["__call__"][T], xrec=one(f(li["output"][0]))[T])
    assert n_layers == len([k for k in enc if k.startswith("layers_")])
    reporter.add_block_raw("decoder", xhat=activations["__call__"][T], xrec=one(f(model.intermediates["output"][0]))[T])
    return reporter
