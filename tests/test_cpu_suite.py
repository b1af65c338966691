"""CPU-only tests (`-m "not gpu"`): the two oracle halves against each other and against the committed
golden fixtures, the host-side setup logic of the product against the oracle, the C-ABI library
(loads, exports every declared symbol, validates arguments), and the 2-rank protocol over gloo.
No GPU compute is called here.
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import cref
from oracle import fxp_oracle as O
from sparsernns_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
GOLDEN = ["tiny_a", "tiny_b_bnscale", "ndns05_short"]



def _free_port() -> int:
    """A rendezvous port nobody holds right now (a fixed one collides when two test runs share a host)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

def unflatten(flat):
    tree = {}
    for k, v in flat.items():
        node = tree
        parts = k.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return tree


def load_golden(name):
    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    meta = json.load(open(os.path.join(GOLD, f"{name}.json")))
    md = unflatten({k[3:]: z[k] for k in z.files if k.startswith("md/")})
    params = unflatten({k[7:]: z[k].astype(np.int32) for k in z.files if k.startswith("params/")})
    inter = {k[6:]: z[k] for k in z.files if k.startswith("inter/")}
    export = dict(params=params, qconfig=meta["export_qconfig"])
    return md, meta, export, z["x"].astype(np.int32), z["y"].astype(np.int32), inter


# ------------------------------------------------------------------------------------------
# oracle vs golden, oracle vs oracle
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", GOLDEN)
def test_oracles_reproduce_golden(name):
    md, meta, export, x, y, inter = load_golden(name)
    # scalar C half, from the integer model
    yc, yb, ye, tr = cref.CModel(export).forward(x, meta["x_bits"], meta["x_exp"], trace=True)
    assert (yb, ye) == (meta["y_bits"], meta["y_exp"])
    assert np.array_equal(yc, y)
    names = dict(xs_re="mixer.xs_re", bu_im="mixer.Bu_im", ys="mixer.ys", residadd="residadd")
    for i, t in enumerate(tr):
        for ck, gk in names.items():
            key = f"layers_{i}.{gk}"
            if key in inter:
                assert np.array_equal(t[ck], inter[key]), key
        assert t["residadd_exp"] == meta["inter"][f"layers_{i}.residadd"][1]
    # NumPy half, from the float model (small fixtures carry the float parameters)
    if md:
        model = O.RegressionModel(md, meta["qconfig"], meta["dims"]["n_layers"])
        it = {}
        yo = model(O.Fx(x, meta["x_bits"], meta["x_exp"]), it)
        assert np.array_equal(yo.data, y)
        fl = O.flatten_intermediates(it)
        for k, v in inter.items():
            assert np.array_equal(fl[k].data, v), k
            assert [fl[k].bits, fl[k].exp] == meta["inter"][k], k


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_numpy_and_c_oracles_agree_on_random_models(seed):
    rng = np.random.default_rng(seed)
    H = int(rng.choice([8, 12, 16]))
    P = int(rng.choice([4, 6, 8]))
    dims = synth.tiny_dims(H=H, P=P, d_in=int(rng.integers(3, 9)), d_out=int(rng.integers(2, 9)), n_layers=int(rng.integers(1, 4)))
    scale = float(rng.choice([1.0, 30.0, 300.0]))
    md, qc, dims = synth.make_model(dims=dims, seed=100 + seed, bn_scale_bias=bool(seed % 2), input_scale=scale,
                                    bn_stats="random" if seed == 3 else "calibrated")
    B, L = int(rng.integers(1, 4)), int(rng.integers(1, 70))
    x = synth.make_input(B, L, dims["d_in"], seed=seed, scale=scale)
    fx = O.from_fp(x, qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], True, O.FLOOR)
    m = O.RegressionModel(md, qc, dims["n_layers"])
    it = {}
    y = m(fx, it)
    yc, yb, ye, tr = cref.CModel(m.export()).forward(fx.data, fx.bits, fx.exp, trace=True)
    assert (yb, ye) == (y.bits, y.exp)
    assert np.array_equal(yc, y.data)
    fl = O.flatten_intermediates(it)
    names = dict(pre_s5="pre_s5", u="mixer.u", bu_re="mixer.Bu_re", xs_im="mixer.xs_im", ys="mixer.ys", out2="out2",
                 sigmoid="out2_sigmoid", post_glu="post_GLU", residadd="residadd")
    for i in range(dims["n_layers"]):
        for ck, ok in names.items():
            assert np.array_equal(tr[i][ck], fl[f"layers_{i}.{ok}"].data), (i, ck)
    # a 2-D (L, d_in) input is the same computation as a batch of one (fxprun.py:531)
    y1, _, _, _ = cref.CModel(m.export()).forward(fx.data[0], fx.bits, fx.exp)
    assert np.array_equal(y1, y.data[0]) or B > 1  # with B > 1 the batch couples through compute_best


def test_float_log2_path_matches_definition_away_from_powers_of_two():
    """The reference evaluates log(x)/log(2) in float32; away from powers of two that agrees with the
    oracle's correctly rounded log2 (the documented ambiguity is confined to a few ulp above 2^k)."""
    rng = np.random.default_rng(5)
    v = np.exp(rng.uniform(np.log(1e-6), np.log(6e4), 20000)).astype(np.float32)
    frac = np.abs(np.log2(v.astype(np.float64)) - np.rint(np.log2(v.astype(np.float64))))
    v = v[frac > 1e-5]
    jaxlike = np.ceil((np.log(v) / np.float32(np.log(np.float32(2)))).astype(np.float32))
    ours = np.array([O.ceil_log2_f32(t) for t in v], dtype=np.float32)
    assert np.array_equal(jaxlike, ours)


# ------------------------------------------------------------------------------------------
# product host logic vs oracle (no GPU: setup is host NumPy)
# ------------------------------------------------------------------------------------------
def tree_equal(a, b, path=""):
    assert isinstance(a, dict) == isinstance(b, dict), path
    if isinstance(a, dict):
        assert set(a) == set(b), (path, set(a) ^ set(b))
        for k in a:
            tree_equal(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        assert np.array_equal(np.asarray(a), np.asarray(b)), path
    else:
        assert a == b, (path, a, b)


@pytest.mark.parametrize("cfg", [dict(dims=synth.tiny_dims()), dict(dims=synth.tiny_dims(H=12, P=6, n_layers=3), bn_scale_bias=True),
                                 dict(dim_scale=0.5), dict(dim_scale=0.5, sparsity=0.9), dict(dim_scale=1.0, quantization="w8a8", bn_stats="random", input_scale=300.0)])
def test_product_setup_quantises_like_the_oracle(cfg):
    from sparsernns_amd.fxpmodel import build_regression_model

    md, qc, dims = synth.make_model(**cfg)
    prod = build_regression_model(md, qc, dims["n_layers"]).export()
    orc = O.RegressionModel(md, qc, dims["n_layers"]).export()
    tree_equal(prod["params"], orc["params"], "params")
    # the product's qconfig carries the reference's extra *_signed keys; everything the oracle has must match
    def sub(a, b, path=""):
        for k, v in b.items():
            if isinstance(v, dict):
                sub(a[k], v, f"{path}/{k}")
            else:
                assert a[k] == v, (path, k)
    sub(prod["qconfig"], orc["qconfig"])
    if cfg.get("sparsity"):
        w = prod["params"]["encoder"]["encoder"]["weight"]
        assert 0.85 < np.mean(w == 0) < 0.97  # zeros stored densely, as jaxpruner leaves them


def test_model_rejects_what_the_reference_rejects():
    from sparsernns_amd.fxpmodel import FxpSSM, build_regression_model

    md, qc, dims = synth.make_model(dims=synth.tiny_dims())
    with pytest.raises(AssertionError):
        build_regression_model(md, qc, dims["n_layers"], glu_variant="bogus")
    with pytest.raises(NotImplementedError):
        build_regression_model(md, qc, dims["n_layers"], glu_variant="full")
    with pytest.raises(AssertionError):  # fxpmodel.py:430-432
        FxpSSM.init_fn(H=8, P=4, discretization="zoh", associative_scan=True)(
            modeldict=md["encoder"]["layers_0"]["mixer"], fxp_qconfig=qc["blocks"]["ssm"], scope="m", store_intermediates=False)
    from sparsernns_amd.fxpmodel import FxpRegressionModel, QuantizationConfig
    with pytest.raises(NotImplementedError):  # the reference's fused-BN branch cannot run (fxpmodel.py:537-549)
        FxpRegressionModel(modeldict=md, fxp_qconfig=qc, scope="model", mixer_cls=FxpSSM.init_fn(H=8, P=4, discretization="zoh"),
                           n_layers=2, d_model=8, batchnorm=True, prenorm=True, glu_variant="half1", relufication=True,
                           fuse_batchnorm_linear=True, q_config=QuantizationConfig.none(), dropout=0.0, training=False,
                           store_intermediates=False)


def test_synth_qconfig_rules():
    # fxputils.py:137-142: an exact power of two needs one more integer bit
    assert [synth.get_intbits(v) for v in (0.3, 1.0, 1.5, 2.0, 3.99, 4.0)] == [0, 1, 1, 2, 2, 3]
    # utils/quantization.py:352-370 + fxputils.py:404-450: exp = min(fracbits, bits - 1 - intbits)
    e = synth._entry(3.7, 16)
    assert (e["intbits"], e["fracbits"], e["exp"]) == (2, 13, 13)
    e = synth._entry(0.0047, 16)
    assert e["exp"] == 15 and e["fracbits"] == 23
    assert synth.ndns_dims(0.5) == dict(H=96, P=64, blocks=8, block_size=16, n_layers=3, d_in=257, d_out=257)
    assert synth.ndns_dims(1.0)["H"] == 192 and synth.ndns_dims(1.0)["P"] == 128
    lam, V = synth.hippo_dplr(16)
    assert np.allclose(lam.real, -0.5) and np.allclose(V.conj().T @ V, np.eye(16), atol=1e-9)
    with pytest.raises(ValueError):
        # at the NDNS input scale the calibrated 1/sqrt(var + 1e-5) is ~300: no non-negative exponent holds it in 8 bits
        synth.make_model(dim_scale=0.5, quantization="w4a8")
    md, qc, _ = synth.make_model(dim_scale=0.5, quantization="w4a8", input_scale=300.0)  # unit-scale activations fit
    w, a = qc["blocks"]["ssm"]["weights"], qc["blocks"]["ssm"]["activations"]
    assert a["Bu_re"]["exp"] <= a["u"]["exp"] + w["B_re"]["exp"]  # cap_result_exponents: no negative matmul shift
    assert w["B_re"]["exp"] > 3  # full_range: a 4-bit B-bar keeps its small entries (exp = fracbits)


def test_shard_bounds():
    from sparsernns_amd.dist import shard_bounds

    for total in (0, 1, 7, 32, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


# ------------------------------------------------------------------------------------------
# the C ABI: the library loads without a GPU, exports every declared symbol, validates arguments
# ------------------------------------------------------------------------------------------
def test_library_exports_every_symbol_in_the_header():
    from sparsernns_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "s5fxp.h")).read()
    declared = set(re.findall(r"\b(s5fxp_[a-z0-9_]+)\s*\(", hdr)) - {"s5fxp_allreduce_max_fn"}
    raw = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} is declared in include/s5fxp.h but not exported by libs5fxp.so"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert _lib.lib.s5fxp_version() == int(re.search(r"#define S5FXP_VERSION (\d+)", hdr).group(1)) >= 101
    assert _lib.lib.s5fxp_strerror(-2).decode().startswith("negative")


def test_c_abi_argument_validation_without_a_gpu():
    """Every entry point checks its arguments before it touches the device."""
    from sparsernns_amd import _lib
    from sparsernns_amd._lib import lib

    assert lib.s5fxp_from_fp(None, None, 4, 16, 8, 0, None) == _lib.S5FXP_EBADARG
    assert lib.s5fxp_change_cfg(1, 1, 4, 16, 40, 16, 2, None) == _lib.S5FXP_ENEGSHIFT
    assert lib.s5fxp_mul(1, 1, 1, 4, 4, 3, 3, 16, 9, None) == _lib.S5FXP_ENEGSHIFT  # fxparray.py:619-621
    assert lib.s5fxp_add(1, 1, 1, 6, 4, 16, 3, 16, 3, 16, 3, 0, None) == _lib.S5FXP_EBADARG  # 6 % 4 != 0
    assert lib.s5fxp_dense(1, 1, None, 1, 8, 4, 4000, 3, 3, 0, 0, 16, 2, 0, None) == _lib.S5FXP_EUNSUPPORTED
    assert lib.s5fxp_scan(1, 1, 1, 1, 1, 1, 2, 8, 0, 15, 15, 15, 15, 14, 14, 0, None) == _lib.S5FXP_EBADARG
    assert lib.s5fxp_model_blob_bytes(None) == 0
    with pytest.raises(ValueError):
        _lib.check(_lib.S5FXP_ENEGSHIFT, "x")
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.S5FXP_EUNSUPPORTED, "x")
    with pytest.raises(_lib.S5FxpError):
        _lib.check(_lib.S5FXP_EHIP, "x")


def test_model_descriptor_validation_on_the_host():
    """s5fxp_model_blob_bytes runs the same validation as create() and needs no GPU."""
    from sparsernns_amd import _lib
    from sparsernns_amd.engine import Engine
    from sparsernns_amd.fxpmodel import build_regression_model

    md, qc, dims = synth.make_model(dim_scale=0.5)
    export = build_regression_model(md, qc, dims["n_layers"]).export()
    eng = Engine.__new__(Engine)  # descriptor construction only (no device)
    eng._keep = []
    eng.n_layers = 3
    eng._layers = (_lib.LayerDesc * 3)()
    for i in range(3):
        eng._fill_layer(eng._layers[i], export["params"]["encoder"][f"layers_{i}"], export["qconfig"]["encoder"][f"layers_{i}"])
    desc = _lib.ModelDesc()
    desc.n_layers = 3
    desc.encoder = eng._dense(export["params"]["encoder"]["encoder"], export["qconfig"]["encoder"]["encoder"])
    desc.layers = C.cast(eng._layers, C.POINTER(_lib.LayerDesc))
    desc.decoder = eng._dense(export["params"]["decoder"], export["qconfig"]["decoder"])
    nbytes = _lib.lib.s5fxp_model_blob_bytes(C.byref(desc))
    assert nbytes > 4 * (257 * 96 * 2 + 3 * (4 * 64 * 96 + 96 * 96))
    # a negative static shift is what the reference turns into a ValueError: rejected at build time
    eng._layers[1].ssm.y_exp = 31
    assert _lib.lib.s5fxp_model_blob_bytes(C.byref(desc)) == 0
    h = C.c_void_p()
    assert _lib.lib.s5fxp_model_create(C.byref(desc), 1, 1 << 30, 0, None, C.byref(h)) == _lib.S5FXP_ENEGSHIFT


def test_importing_without_the_extension_fails_loudly(tmp_path):
    code = ("import importlib.util, sys, os\n"
            "spec = importlib.util.spec_from_file_location('lib_copy', sys.argv[1])\n"
            "m = importlib.util.module_from_spec(spec)\n"
            "try:\n    spec.loader.exec_module(m)\nexcept ImportError as e:\n    print('LOUD', e)\n")
    src = open(os.path.join(ROOT, "sparsernns_amd", "_lib.py")).read()
    p = tmp_path / "lib_copy.py"
    p.write_text(src)  # next to it there is no libs5fxp.so
    out = subprocess.run([sys.executable, "-c", code, str(p)], capture_output=True, text=True)
    assert "LOUD" in out.stdout and "no CPU fallback" in out.stdout.replace("\n", " ")


# ------------------------------------------------------------------------------------------
# 2-rank protocol over gloo (world_size 2, CPU): batch sharding + exponent all-reduce + output gather
# ------------------------------------------------------------------------------------------
WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from oracle import fxp_oracle as O
from sparsernns_amd import synth
from sparsernns_amd.dist import shard_bounds, make_exponent_allreduce, gather_outputs
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
md, qc, dims = synth.make_model(dims=synth.tiny_dims(H=12, P=6, d_in=7, d_out=9, n_layers=2), bn_scale_bias=True, input_scale=30.0)
B, L = 6, 33
x = synth.make_input(B, L, dims["d_in"], seed=4, scale=30.0)
fx = O.from_fp(x, qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], True, O.FLOOR)
model = O.RegressionModel(md, qc, dims["n_layers"])
full = model(fx).data                                   # one reference run over the whole batch
lo, hi = shard_bounds(B, world, rank)
mine = O.Fx(fx.data[lo:hi], fx.bits, fx.exp)
hook = make_exponent_allreduce()
def exchange(v):                                        # mode A: all_reduce(MAX) of the compute_best maxima
    t = torch.from_numpy(np.array(v, dtype=np.float32))
    hook(t)
    return t.numpy()
O.MAX_EXCHANGE = exchange
y_global = model(mine).data
O.MAX_EXCHANGE = None
y_local = model(mine).data                              # mode B: the shard is its own reference batch
out = gather_outputs(torch.from_numpy(y_global.copy()))
ok_a = bool(np.array_equal(out.numpy(), full))          # N ranks == one run over the concatenated batch
from oracle import cref                                 # mode B == the OTHER oracle half run on the shard alone
ok_b = bool(np.array_equal(y_local, cref.CModel(model.export()).forward(mine.data, mine.bits, mine.exp)[0]))
differs = bool(not np.array_equal(y_local, full[lo:hi]))
print(f"RANK{rank} modeA={ok_a} modeB={ok_b} coupled={differs}", flush=True)
dist.destroy_process_group()
'''


def test_two_rank_protocol_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"RANK{r} modeA=True modeB=True" in o, o


def test_fxprun_cli_fails_loudly_without_gpu_and_reads_the_interchange_format():
    import torch
    from sparsernns_amd import fxprun

    if not torch.cuda.is_available():
        assert fxprun.main(["--synthetic", "--steps", "0"]) == 2  # no CPU fallback
    md, meta, export, x, y, inter = load_golden("tiny_a")
    ex2, meta2 = fxprun.load_export(os.path.join(GOLD, "tiny_a.npz"), os.path.join(GOLD, "tiny_a.json"))
    assert ex2["qconfig"] == export["qconfig"]
    a, b = fxprun._flatten(ex2["params"]), fxprun._flatten(export["params"])
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)


def test_audio_front_and_back_end_match_scipy():
    """STFT / iSTFT / SI-SNR of the denoising loop (train_helpers.py:15-53,1382-1412) against scipy.signal, which
    jax.scipy.signal mirrors; lengths that are and are not a multiple of the hop."""
    import scipy.signal
    import torch
    from sparsernns_amd import audio

    rng = np.random.default_rng(3)
    for T in (4096, 5000, 777):
        x = rng.standard_normal((2, T)).astype(np.float32)
        _, _, Z = scipy.signal.stft(x, nperseg=512, nfft=512, noverlap=384, window="boxcar", return_onesided=True)
        mag, ph = audio.stft_splitter(torch.from_numpy(x))
        assert tuple(mag.shape) == Z.shape
        z = torch.polar(mag, ph).numpy()
        assert np.allclose(z, Z, atol=2e-6), np.abs(z - Z).max()
        _, xr = scipy.signal.istft(Z, nperseg=512, nfft=512, noverlap=384, window="boxcar", input_onesided=True)
        back = audio.stft_mixer(mag, ph).numpy()
        assert back.shape == xr.shape and np.allclose(back, xr, atol=2e-5), np.abs(back - xr).max()
        assert np.allclose(back[:, :T], x, atol=2e-5)
    t = rng.standard_normal((3, 2000)).astype(np.float32)
    e = (t + 0.1 * rng.standard_normal((3, 2000))).astype(np.float32)
    st, se = t - t.mean(-1, keepdims=True), e - e.mean(-1, keepdims=True)
    proj = (st * se).sum(-1, keepdims=True) * st / (st ** 2).sum(-1, keepdims=True)
    want = 10 * np.log10((proj ** 2).sum(-1) / (((se - proj) ** 2).sum(-1) + 1e-8) + 1e-8)
    assert np.allclose(audio.si_snr(torch.from_numpy(t), torch.from_numpy(e)).numpy(), want, atol=1e-4)


def test_generated_scan_asm_is_up_to_date(tmp_path):
    """sparsernns_amd/csrc/scan_quad_asm.inc is generated by tools/gen_scan_asm.py: the committed file must be what
    the generator writes (so an edit of one without the other cannot ship)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_scan_asm", os.path.join(ROOT, "tools", "gen_scan_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    t32, _ = gen.emit("S5_SCAN_ASM", gen.Plan(False))
    t16, _ = gen.emit("S5_SCAN16_ASM", gen.Plan(True))
    t32w, _ = gen.emit("S5_SCAN32W_ASM", gen.Plan(False, wide=True))
    tp, _ = gen.emit_pair()
    tpl = "".join(gen.emit_pairl(D, f"S5_SCANPL{D}_ASM")[0] for D in gen.PAIRL_BLOCKS)
    have = open(os.path.join(ROOT, "sparsernns_amd", "csrc", "scan_quad_asm.inc")).read()
    assert t32 in have and t16 in have and t32w in have and tp in have and tpl in have
    assert f"#define S5_SCAN_ASM_DEPTH {gen.DEPTH}" in have and f"#define S5_SCANP_ASM_DEPTH {gen.PAIR_DEPTH}" in have


def test_config0_float_forward_plumbing_and_fixed_point_tracks_it():
    """BASELINE configs[0]: the fp32 dense forward at B=1, L=1024, dim_scale 0.5 (the reference runs it on JAX-CPU with an
    associative scan, model/ssm.py:54-185; here the NumPy restatement used for calibration).  The w8a16 integer model
    built from it follows it only loosely: every product of the recurrence is FLOORed (fxpmodel.py:155-169), a bias of
    about two LSB per step that a pole at 0.9995 amplifies a thousandfold -- the drift behind synth's
    state_headroom_bits.  The check is therefore plumbing plus a sanity bound, not an accuracy claim."""
    md, qc, dims = synth.make_model(0.5, calib_B=1, calib_L=1024, state_headroom_bits=1)
    x = synth.make_input(1, 1024, dims["d_in"], seed=5)
    yf = synth.float_forward(md, x, dims["n_layers"])
    assert yf.shape == (1, 1024, dims["d_out"]) and np.isfinite(yf).all()
    fx = O.from_fp(x, qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], True, O.FLOOR)
    model = O.RegressionModel(md, qc, dims["n_layers"])
    yq, yb, ye, _ = cref.CModel(model.export()).forward(fx.data, fx.bits, fx.exp)
    yq = yq.astype(np.float64) / (1 << ye)
    err = np.abs(yq - yf).max() / (np.abs(yf).max() + 1e-12)
    corr = np.corrcoef(yq.ravel(), yf.ravel())[0, 1]
    assert corr > 0.5 and err < 2.0, (corr, err)


def test_streaming_carry_in_both_oracle_halves():
    """The streaming carry (fxpmodel.py:147-172: the step function's state is an explicit argument): NumPy half == C half
    chunk by chunk, carries equal after every chunk, and a zero carry is the plain forward."""
    md, qc, dims = synth.make_model(dims=synth.tiny_dims(H=12, P=6, d_in=7, d_out=9, n_layers=2), bn_scale_bias=True, input_scale=30.0)
    m = O.RegressionModel(md, qc, dims["n_layers"])
    cm = cref.CModel(m.export())
    B = 3
    st = m.zero_state((B,))
    stc = np.zeros((dims["n_layers"], 2, B, dims["P"]), dtype=np.int32)
    for i, L in enumerate((16, 1, 23, 40)):
        x = synth.make_input(B, L, dims["d_in"], seed=50 + i, scale=30.0)
        fx = O.from_fp(x, qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], True, O.FLOOR)
        a = m(fx, None, st).data
        b = cm.forward(fx.data, fx.bits, fx.exp, state=stc)[0]
        assert np.array_equal(a, b), i
        assert all(np.array_equal(st[l][c], stc[l, c]) for l in range(dims["n_layers"]) for c in range(2)), i
        if i == 0:
            assert np.array_equal(a, m(fx).data)  # zero carry
    assert np.abs(stc).max() > 0


# ------------------------------------------------------------------------------------------
# qconfig derivation from calibration trees (sparsernns_amd/fxputils.py <- sparseRNNs/fxputils.py:121-134,351-401,453-786)
# ------------------------------------------------------------------------------------------
def _calibrated(dim_scale=0.5, bn_scale_bias=False, seed=1919):
    dims = synth.ndns_dims(dim_scale)
    md = synth.make_float_params(dims, seed, bn_scale_bias)
    stats = {}
    synth.float_forward(md, synth.make_input(2, 128, dims["d_in"], seed=seed + 1), dims["n_layers"], calibrate_bn=True, stats=stats)
    return md, stats, dims


def _same_numbers(a, b, path=""):
    """every bits / exp / intbits / fracbits / absmax of `a` is in `b` with the same value"""
    n = 0
    for k, v in a.items():
        if isinstance(v, dict):
            n += _same_numbers(v, b[k], f"{path}/{k}")
        elif k.endswith(("bits", "exp", "intbits", "fracbits")):
            assert b[k] == v, (path, k, v, b[k])
            n += 1
        elif k.endswith("absmax"):
            assert abs(b[k] - v) <= 1e-6 * abs(v), (path, k, v, b[k])
            n += 1
    return n


def test_fxputils_reproduces_the_synthetic_qconfig_from_checkpoint_shaped_trees(tmp_path):
    from sparsernns_amd import fxputils
    from sparsernns_amd.fxpmodel import build_regression_model

    md, stats, dims = _calibrated()
    want = synth.derive_qconfig(md, stats, dims["n_layers"])
    params, st = synth.reference_trees(md, stats, dims["n_layers"])
    # the interchange format: one npz per tree, '/'-joined paths, no pickles
    fxputils.save_tree_npz(tmp_path / "params.npz", params)
    fxputils.save_tree_npz(tmp_path / "stats.npz", st)
    params2, st2 = fxputils.load_tree_npz(tmp_path / "params.npz"), fxputils.load_tree_npz(tmp_path / "stats.npz")
    md2, qc = fxputils.derive(params2, st2, "w8a16")
    assert _same_numbers(want, qc) > 150
    assert fxputils.precisions_for("w8a16") == synth.W8A16 and fxputils.precisions_for("w8a8") == synth.W8A8
    # load_modeldict's marks: every *scale* leaf is a log2 now, every observer has absmax / intbits
    d = md2["encoder"]["encoder"]
    assert float(d["act_scale"]) == round(float(d["act_scale"])) and d["input_observer"]["intbits"] == int(np.ceil(np.log2(d["input_observer"]["absmax"])))
    # the derived pair builds the same integer model as the synthetic pair (oracle and product setup)
    a = O.RegressionModel(md, want, dims["n_layers"]).export()
    b = O.RegressionModel(md2, qc, dims["n_layers"]).export()
    tree_equal(a["params"], b["params"], "params")
    tree_equal(build_regression_model(md2, qc, dims["n_layers"]).export()["params"], b["params"], "product")


def test_fxputils_separate_exponents_and_batchnorm_scale_quirk():
    from sparsernns_amd import fxputils
    from sparsernns_amd.fxpmodel import build_regression_model

    md, stats, dims = _calibrated(bn_scale_bias=True, seed=7)
    md["encoder"]["layers_1"]["norm"]["scale"][3] = -0.7            # log2 of a negative scale is NaN ...
    params, st = synth.reference_trees(md, stats, dims["n_layers"])
    md2, qc = fxputils.derive(params, st, "w8a16", separate_exponents=True)
    assert set(qc["blocks"]) == {"layers_0", "layers_1", "layers_2"}  # --separate_exponents layout (fxputils.py:386-401)
    sc = md2["encoder"]["layers_1"]["norm"]["scale"]
    want = np.log2(md["encoder"]["layers_1"]["norm"]["scale"].astype(np.float32), where=md["encoder"]["layers_1"]["norm"]["scale"] > 0,
                   out=np.ones_like(sc))
    assert sc[3] == 1.0 and np.allclose(sc, want)                    # ... which the reference replaces by 1.0 (:711-731)
    for i in range(3):
        blk = qc["blocks"][f"layers_{i}"]
        assert set(blk) == {"ssm", "multgate", "out2", "norm"} and set(blk["norm"]) == {"mean", "var", "invsq_var", "scale", "bias"}
        own = max(0, int(np.ceil(np.log2(np.abs(md["encoder"][f"layers_{i}"]["norm"]["mean"]).max()))))
        assert blk["norm"]["mean"]["exp"] == 15 - own
    # per-layer exponents are at least as fine as the shared ones, and the model builds and runs from them
    shared = fxputils.derive(*synth.reference_trees(md, stats, dims["n_layers"]), "w8a16")[1]
    for i in range(3):
        for t, e in qc["blocks"][f"layers_{i}"]["ssm"]["activations"].items():
            assert e["exp"] >= shared["blocks"]["ssm"]["activations"][t]["exp"], (i, t)
    x = synth.make_input(1, 24, dims["d_in"], seed=3)
    fx = O.from_fp(x, qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], True, O.FLOOR)
    model = O.RegressionModel(md2, qc, dims["n_layers"])
    y = model(fx)
    yc = cref.CModel(build_regression_model(md2, qc, dims["n_layers"]).export()).forward(fx.data, fx.bits, fx.exp)[0]
    assert np.array_equal(y.data, yc)
    with pytest.raises(ValueError):
        fxputils.create_fxp_qconfig(md2, agg="mean")
    full, joined = fxputils.create_fxp_qconfig(fxputils.load_modeldict(*synth.reference_trees(md, stats, dims["n_layers"])), agg="set")
    assert isinstance(joined["blocks"]["ssm"]["activations"]["u"]["fracbits"], list) and "layers_0" in full["blocks"]["ssm"]


def test_reporter_metrics_follow_the_reference_definitions():
    """sparseRNNs/fxpreporter.py:12-24: absolute error over everything, relative error over the elements whose float value is
    not zero; complex stages are reported per part (:137-171)."""
    from sparsernns_amd.fxpreporter import Reporter, compute_error

    m = compute_error(xrec=np.array([1.0, 2.5, 0.0, -4.0]), xhat=np.array([1.0, 2.0, 0.0, -5.0]))
    assert m["abs_error_mean"] == 0.375 and m["abs_error_max"] == 1.0 and m["abs_error_med"] == 0.25
    assert abs(m["rel_error_mean"] - 0.15) < 1e-12 and m["rel_error_max"] == 0.25 and abs(m["rel_error_med"] - 0.2) < 1e-12
    r = Reporter(None)
    r.add_block_raw("z", xhat=np.array([1 + 2j, 3 - 1j]), xrec=np.array([1 + 2j, 3 - 2j]), verbose=False)
    assert [b["name"] for b in r.results_data] == ["z (real)", "z (imag)"]
    assert r.results_data[0]["abs_error_max"] == 0.0 and r.results_data[1]["abs_error_max"] == 1.0
    assert "| z (imag) |" in r.markdown()


def test_pickle_converter_reads_array_trees_only(tmp_path):
    """tools/reference_pickles_to_npz.py (the run-elsewhere step in front of fxputils): pickles written HERE by this test --
    NumPy trees shaped like the reference's calibration output -- convert to the npz pair fxputils reads, and a pickle that
    references anything but arrays and containers is refused before it can run."""
    import importlib.util
    import pickle

    from sparsernns_amd import fxputils

    spec = importlib.util.spec_from_file_location("conv", os.path.join(ROOT, "tools", "reference_pickles_to_npz.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    md, stats, dims = _calibrated()
    params, st = synth.reference_trees(md, stats, dims["n_layers"])
    folder = tmp_path / "data"
    folder.mkdir()
    for name, tree in (("sc_calibrated_params.pkl", params), ("sc_cal_stats.pkl", st)):
        with open(folder / name, "wb") as f:
            pickle.dump(tree, f)
    assert conv.main(["calib", str(folder), str(tmp_path / "m")]) == 0
    _, qc = fxputils.derive(fxputils.load_tree_npz(tmp_path / "m.params.npz"), fxputils.load_tree_npz(tmp_path / "m.stats.npz"), "w8a16")
    assert _same_numbers(synth.derive_qconfig(md, stats, dims["n_layers"]), qc) > 150
    evil = pickle.dumps({"x": os.getcwd})  # a global that is not an array constructor
    with pytest.raises(pickle.UnpicklingError):
        conv.load_arrays_only(evil)
    with pytest.raises(pickle.UnpicklingError):  # nor does living under "numpy." make a callable an array constructor
        conv.load_arrays_only(pickle.dumps({"x": np.testing.assert_equal}))
    # activations_fp.pkl (lists of recorded calls, batch first) -> the per-stage tree of the verification report
    acts = {}
    xs = synth.make_input(2, 24, dims["d_in"], seed=3)
    synth.float_forward(md, xs, dims["n_layers"], activations=acts)
    rec = {"__call__": [acts["__call__"][None]], "encoder": {}}
    for name, a in acts["encoder"].items():
        rec["encoder"][name] = {k: [a[k][None]] for k in ("input", "pre_s5", "pre_C", "pre_GLU", "__call__")}
        rec["encoder"][name]["mixer"] = {"B_bar": [a["mixer"]["B_bar"]], "__call__": [(a["mixer"]["__call__"][None], a["pre_C"][None])]}
        rec["encoder"][name]["out2"] = {"__call__": [a["out2"]["__call__"][None]]}
        rec["encoder"][name]["drop"] = {"__call__": [a["pre_GLU"][None], a["post_GLU"][None]]}
    with open(folder / "activations_fp.pkl", "wb") as f:
        pickle.dump(rec, f)
    assert conv.main(["acts", str(folder / "activations_fp.pkl"), str(tmp_path / "a.npz")]) == 0
    tree_equal(fxputils.load_tree_npz(tmp_path / "a.npz"), acts, "activations")
    # the exported-integer-model form
    m = O.RegressionModel(md, synth.derive_qconfig(md, stats, dims["n_layers"]), dims["n_layers"])
    with open(folder / "fxpmodel.pkl", "wb") as f:
        pickle.dump(m.export(), f)
    assert conv.main(["export", str(folder), str(tmp_path / "e")]) == 0
    from sparsernns_amd import fxprun
    ex, meta = fxprun.load_export(str(tmp_path / "e.npz"), str(tmp_path / "e.json"))
    tree_equal(ex["params"], m.export()["params"], "params")


def test_hand_written_wait_counts_match_the_compiled_store_counts():
    """proj_p.hpp k_enc_p / k_dec_p prefetch with loads the compiler cannot see and wait with s_waitcnt vmcnt(N), N = the store
    instructions a wave issues per full tile (scan_quad.hpp vm_wait).  That is only right while the compiler emits exactly
    those stores: tools/check_vmwait.py compiles the device code to assembly (no GPU) and counts."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_vmwait.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
