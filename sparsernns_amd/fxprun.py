"""fxprun-style command line for the MI355X fixed-point S5 path.

Mirrors what the reference's ``sparseRNNs/fxprun.py`` does around the model (``run_validation`` :63-88: float input
-> ``fxp_from_fp`` -> ``model(fxp_x)`` -> ``to_float``; ``run_verification`` :476-731: the same forward with
``store_intermediates`` and a per-layer report), with the pieces that need JAX pickles or the NDNS dataset
replaced by interchange files this repo can read:

  --synthetic            NDNS-shaped random-init model (sparsernns_amd/synth.py) -- no files needed
  --model M.npz --meta M.json
                         an integer model in ``FxpRegressionModel.export()`` layout (keys ``params/...`` in the
                         npz, ``export_qconfig`` in the json: the format of tests/golden/*.npz); a maintainer
                         with JAX writes it from the reference with
                         ``np.savez(path, **{f"params/{k}": v for k, v in flatten(model.export()["params"])})``
  --params P.npz --stats S.npz [--separate_exponents]
                         the reference's calibration output (``sc_calibrated_params.pkl`` / ``sc_cal_stats.pkl``) as npz
                         trees: modeldict and fxp_qconfig are derived exactly as
                         ``fxprun.py:294-397`` derives them (sparsernns_amd/fxputils.py), then the model is built from them
  --inputs X.npy         float32 (B,L,d_in) model inputs (the reference's ``inputs.npy``); default: synthetic
  --verify [--activations A.npz] [--report DIR]
                         ``run_verification`` (fxprun.py:476-731): the op-by-op forward with ``store_intermediates``, checked
                         against the fused engine bit for bit, then every stage the reference's report looks at is compared
                         with the FLOAT model's activation of that stage (abs / rel error, sparsernns_amd/fxpreporter.py).
                         A.npz is the float side as an npz tree (the reference's ``activations_fp.pkl`` through
                         tools/reference_pickles_to_npz.py); without it the float forward of this package runs on the
                         same float parameters (--synthetic, --params)

Flags kept from the reference where they mean the same (fxprun.py:98-269): --quantization, --seq_len, --bsz, --export,
--separate_exponents, and --params_fname / --stats_fname / --inputs_fname / --activations_fname as aliases of the file
options above (here they are paths, not names inside a checkpoint directory).
There is no CPU fallback: without a ROCm GPU and the built libs5fxp.so this exits with an error.
"""
from __future__ import annotations

import argparse
import json
import sys
import time

import numpy as np


def _unflatten(flat: dict) -> dict:
    out: dict = {}
    for k, v in flat.items():
        d = out
        parts = k.split("/")
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return out


def _flatten(tree: dict, prefix: str = "") -> dict:
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(_flatten(v, f"{prefix}{k}/"))
        else:
            out[f"{prefix}{k}"] = v
    return out


def load_export(npz_path: str, meta_path: str) -> dict:
    """{"params", "qconfig"} as Engine wants it, from the interchange pair."""
    z = np.load(npz_path, allow_pickle=False)
    with open(meta_path) as f:
        meta = json.load(f)
    params = _unflatten({k[len("params/"):]: np.asarray(z[k]).astype(np.int32) for k in z.files if k.startswith("params/")})
    return dict(params=params, qconfig=meta["export_qconfig"]), meta


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m sparsernns_amd.fxprun", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--synthetic", action="store_true")
    src.add_argument("--model", type=str, help="integer model, export() layout (.npz)")
    src.add_argument("--params", "--params_fname", dest="params", type=str, help="calibrated float parameters as an npz tree (needs --stats)")
    ap.add_argument("--stats", "--stats_fname", dest="stats", type=str, help="calibration statistics as an npz tree (with --params)")
    ap.add_argument("--separate_exponents", action="store_true", help="per-layer exponents, as the reference's flag (with --params)")
    ap.add_argument("--meta", type=str, help="json beside --model (export_qconfig, input bits/exp)")
    ap.add_argument("--inputs", "--inputs_fname", dest="inputs", type=str, default=None, help="float32 (B,L,d_in) .npy")
    ap.add_argument("--activations", "--activations_fname", dest="activations", type=str, default=None,
                    help="--verify: the float model's intermediates of sequence 0 as an npz tree (fxpreporter.verification_report)")
    ap.add_argument("--report", type=str, default=None, help="--verify: write report.md / results.json into this folder")
    ap.add_argument("--write-activations", type=str, default=None,
                    help="--verify without --activations: save the float intermediates this run computed (npz tree)")
    ap.add_argument("--outputs", type=str, default=None, help="write the float outputs here (.npy)")
    ap.add_argument("--quantization", type=str, default="w8a16")
    ap.add_argument("--dim_scale", type=float, default=0.5)
    ap.add_argument("--sparsity", type=float, default=0.0, help="synthetic: magnitude-prune this fraction of every weight matrix")
    ap.add_argument("--seq_len", type=int, default=4096)
    ap.add_argument("--bsz", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10, help="timed forwards")
    ap.add_argument("--inflight", type=int, default=1, help="batches kept in flight (engine.InflightRunner)")
    ap.add_argument("--verify", action="store_true",
                    help="run_verification: op-by-op forward with store_intermediates, compared with the fused engine and, stage by "
                         "stage, with the float model's activations")
    ap.add_argument("--export", type=str, default=None, help="write the integer model as PREFIX.npz / PREFIX.json")
    ap.add_argument("--check-golden", action="store_true",
                    help="--model only: the npz also holds an integer input `x` and the output `y` some other implementation "
                         "produced for it (the reference's fxpmodel_io.pkl through tools/reference_pickles_to_npz.py export, or "
                         "tests/golden/*.npz): run x and compare bit for bit")
    args = ap.parse_args(argv)

    import torch
    if not torch.cuda.is_available():
        print("fxprun: no ROCm GPU visible -- this path has no CPU fallback", file=sys.stderr)
        return 2
    from . import _lib, synth
    from .engine import Engine, InflightRunner
    from .fxparray import FxpArray, RoundingMode, fxp_from_fp
    from .fxpmodel import build_regression_model

    model = None
    if args.params:
        if not args.stats:
            ap.error("--params needs --stats")
        from . import fxputils
        md, qc = fxputils.derive(fxputils.load_tree_npz(args.params), fxputils.load_tree_npz(args.stats), args.quantization,
                                 separate_exponents=args.separate_exponents)
        dims = dict(n_layers=len([k for k in md["encoder"] if k.startswith("layers_")]))
        model = build_regression_model(md, qc, dims["n_layers"])
        eng = model.engine()
        inp_bits, inp_exp, d_in = int(qc["encoder"]["inp_bits"]), int(qc["encoder"]["inp_exp"]), eng.d_in
    elif args.synthetic:
        md, qc, dims = synth.make_model(args.dim_scale, quantization=args.quantization, sparsity=args.sparsity,
                                        calib_L=min(1024, max(64, args.seq_len)), state_headroom_bits=1)
        model = build_regression_model(md, qc, dims["n_layers"])
        eng = model.engine()
        inp_bits, inp_exp, d_in = qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], dims["d_in"]
    else:
        if not args.meta:
            ap.error("--model needs --meta")
        export, meta = load_export(args.model, args.meta)
        eng = Engine(export)
        inp_bits, inp_exp, d_in = int(meta["x_bits"]), int(meta["x_exp"]), eng.d_in
    print(f"[fxprun] model: d_in={eng.d_in} H={eng.H} P={eng.P} layers={eng.n_layers} d_out={eng.d_out} "
          f"mfma_fast_path={bool(_lib.lib.s5fxp_model_is_fast(eng._h))}")

    if args.check_golden:
        if args.synthetic:
            ap.error("--check-golden needs --model")
        z = np.load(args.model, allow_pickle=False)
        if "x" not in z.files or "y" not in z.files:
            print("[fxprun] --check-golden: the npz holds no x / y", file=sys.stderr)
            return 2
        gx, gy = z["x"].astype(np.int32), z["y"].astype(np.int32)
        got = eng.forward(FxpArray(torch.from_numpy(gx).to(eng.device), inp_bits, inp_exp))
        same = bool(np.array_equal(got.numpy(), gy))
        cfg_ok = ("y_bits" not in meta) or (got.bits, got.exp) == (int(meta["y_bits"]), int(meta["y_exp"]))
        print(f"[fxprun] golden check: {gx.shape} -> {gy.shape}: bit-exact={same} output config matches={cfg_ok}")
        return 0 if (same and cfg_ok) else 1

    if args.inputs:
        x = np.load(args.inputs, allow_pickle=False).astype(np.float32)
        if x.ndim == 2:
            x = x[None]
    else:
        x = synth.make_input(args.bsz, args.seq_len, d_in, seed=args.seed)
    B, L = x.shape[0], x.shape[1]
    # fxprun.py:69-75: signed, FLOOR, the encoder's input configuration
    fx = fxp_from_fp(x, bits=inp_bits, exp=inp_exp, signed=True, round_mode=RoundingMode.FLOOR)

    y = eng.forward(fx)
    st = eng.check_status()
    print(f"[fxprun] output {tuple(y.data.shape)} bits={y.bits} exp={y.exp}  status=0x{int(st[0]):x}")
    for i, e in enumerate(eng.layer_exponents()):
        print(f"[fxprun] layer {i} compute_best exponents: " + ", ".join(f"{k}={v}" for k, v in e.items()))
    if args.outputs:
        np.save(args.outputs, y.to_float().cpu().numpy())

    if args.verify:
        if model is None:
            print("[fxprun] --verify needs the float modeldict (use --synthetic): the op-by-op path is built from it",
                  file=sys.stderr)
            return 2
        eager = build_regression_model(md, qc, dims["n_layers"], store_intermediates=True)
        ye = eager(fx)
        same = bool(torch.equal(ye.data, y.data)) and (ye.bits, ye.exp) == (y.bits, y.exp)
        print(f"[fxprun] verification: op-by-op forward == fused forward: {same}")
        for i, layer in enumerate(eager.encoder.seq_layers):
            names = sorted(layer.intermediates.keys()) + sorted(f"mixer.{k}" for k in layer.mixer.intermediates.keys())
            print(f"[fxprun]   layer {i} intermediates: {', '.join(names)}")
        if not same:
            return 1
        # ---- fixed point vs float, stage by stage (fxprun.py:553-731, fxpreporter.py)
        from . import fxputils
        from .fxpreporter import Reporter, verification_report
        if args.activations:
            acts = fxputils.load_tree_npz(args.activations)
        else:
            acts = {}
            synth.float_forward(md, x[:1], dims["n_layers"], activations=acts)
            if args.write_activations:
                fxputils.save_tree_npz(args.write_activations, acts)
        rep = Reporter(args.report, header=dict(quantization=args.quantization, batch=B, seq_len=L, input_bits=inp_bits, input_exp=inp_exp,
                                                float_side=args.activations or "sparsernns_amd.synth.float_forward"))
        verification_report(eager, FxpArray(fx.data[:1], fx.bits, fx.exp), x[0], acts, rep, seq_len=L)
        rep.save()
        w = rep.worst()
        print(f"[fxprun] verification report: {len(rep.results_data)} stages; largest median relative error {w['rel_error_med']:.3%} "
              f"at {w['name']}" + (f"; written to {args.report}/report.md" if args.report else ""))

    if args.export:
        ex = (model.export() if model is not None else export)
        arrays = {f"params/{k}": np.asarray(v) for k, v in _flatten(ex["params"]).items()}
        np.savez_compressed(args.export + ".npz", **arrays)
        with open(args.export + ".json", "w") as f:
            json.dump(dict(export_qconfig=ex["qconfig"], x_bits=inp_bits, x_exp=inp_exp), f, indent=1, sort_keys=True)
        print(f"[fxprun] wrote {args.export}.npz / .json")

    if args.steps > 0:
        yo = [torch.empty_like(y.data) for _ in range(max(1, args.inflight))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.inflight > 1:
            runner = InflightRunner(eng, args.inflight)
            for k in range(args.steps):
                # the same input every time: the status check at drain() speaks for every step
                runner.submit(fx.data, fx.bits, fx.exp, yo[k % args.inflight], B, L, check=False)
            runner.drain()
        else:
            for _ in range(args.steps):
                eng.enqueue(fx.data, fx.bits, fx.exp, yo[0], B, L)
            torch.cuda.synchronize()
            eng.check_status()
        dt = time.perf_counter() - t0
        print(f"[fxprun] {args.steps} forwards of {B}x{L} frames: {dt / args.steps * 1e3:.3f} ms each, "
              f"{B * L * args.steps / dt:.3e} frames/s ({args.inflight} in flight)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
