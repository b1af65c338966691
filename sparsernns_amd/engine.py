"""Fused forward of the fixed-point S5 model: Python driver of ``s5fxp_model_forward``.

``Engine`` takes the INTEGER model in the reference's ``export()`` layout
(sparseRNNs/fxpmodel.py:1441-1458 and the nested exports it gathers), hands it to the C ABI,
and owns the device buffers (parameter blob, workspace, status words) as torch tensors.
The forward itself is a sequence of HIP kernel launches on the current stream with no host
synchronisation inside; data-dependent exponents stay in device memory.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import (DenseDesc, LayerDesc, LayerTrace, ModelDesc, NormDesc, SSMDesc, TRACE_FIELDS, check, lib)
from .fxparray import FxpArray

I32P = _lib.I32P


class Engine:
    def __init__(self, export: dict, flags: int = 0, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("sparsernns_amd.Engine needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self._keep: list = []
        P, Q = export["params"], export["qconfig"]
        self.n_layers = len([k for k in P["encoder"] if k.startswith("layers_")])
        self._layers = (LayerDesc * max(self.n_layers, 1))()
        for i in range(self.n_layers):
            self._fill_layer(self._layers[i], P["encoder"][f"layers_{i}"], Q["encoder"][f"layers_{i}"])
        self._desc = ModelDesc()
        self._desc.n_layers = self.n_layers
        self._desc.encoder = self._dense(P["encoder"]["encoder"], Q["encoder"]["encoder"])
        self._desc.layers = C.cast(self._layers, C.POINTER(LayerDesc))
        self._desc.decoder = self._dense(P["decoder"], Q["decoder"])
        self.d_in, self.H = self._desc.encoder.K, self._desc.encoder.M
        self.P = self._layers[0].ssm.P if self.n_layers else 0
        self.d_out = self._desc.decoder.M
        self.inp_bits, self.inp_exp = self._desc.encoder.inp_bits, self._desc.encoder.inp_exp
        nbytes = lib.s5fxp_model_blob_bytes(C.byref(self._desc))
        if nbytes == 0:
            # let create() report the precise reason
            nbytes = 256
        with torch.cuda.device(self.device):
            self.blob = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            handle = C.c_void_p()
            check(lib.s5fxp_model_create(C.byref(self._desc), self.blob.data_ptr(), nbytes, flags,
                                         torch.cuda.current_stream().cuda_stream, C.byref(handle)), "s5fxp_model_create")
        self._h = handle
        self.out_bits, self.out_exp = lib.s5fxp_model_out_bits(self._h), lib.s5fxp_model_out_exp(self._h)
        # a lane = the per-forward mutable state (status words + workspace); forwards on different lanes may be
        # in flight at the same time on different streams (InflightRunner).  Lane 0 is the default.
        self._status = {0: torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=self.device)}
        # The ladder of recurrence kernels an optimistic forward climbs when ST_REDO comes back: 0 pair kernel (tightest bound
        # on |state|), 1 quad kernel with int16 streams (16 bits), 2 exact 32-bit kernels.  `level` is where forwards start;
        # a rung that failed twice is not tried again for this model.
        self.level = 0
        self._redos = [0, 0]
        self._wsl: Dict[int, tuple] = {}
        self._groups: Dict[int, int] = {}   # groups of the last forward enqueued on a lane
        self._cb_keep = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.s5fxp_model_destroy(h)
            self._h = None

    # -- descriptor construction ---------------------------------------------------------------
    def _ptr(self, a) -> I32P:
        arr = np.ascontiguousarray(np.asarray(a).astype(np.int64).astype(np.int32))
        self._keep.append(arr)
        return arr.ctypes.data_as(I32P)

    def _dense(self, p: dict, q: dict) -> DenseDesc:
        d = DenseDesc()
        d.K, d.M = p["weight"].shape
        d.weight, d.bias = self._ptr(p["weight"]), self._ptr(p["bias"])
        d.w_bits, d.w_exp = int(q["weight_bits"]), int(q["weight_exp"])
        d.b_bits, d.b_exp = int(q["bias_bits"]), int(q["bias_exp"])
        d.inp_bits, d.inp_exp = int(q["inp_bits"]), int(q["inp_exp"])
        d.out_bits, d.out_exp = int(q["out_bits"]), int(q["out_exp"])
        return d

    def _fill_layer(self, L: LayerDesc, lp: dict, lq: dict) -> None:
        L.out2 = self._dense(lp["out2"], lq["out2"])
        m, mq = lp["mixer"], lq["mixer"]
        s: SSMDesc = L.ssm
        s.P, s.H = m["B_real"].shape
        for f, k in (("A_re", "A_real"), ("A_im", "A_imag"), ("B_re", "B_real"), ("B_im", "B_imag"),
                     ("C_re", "C_real"), ("C_im", "C_imag"), ("D", "D")):
            setattr(s, f, self._ptr(m[k]))
            setattr(s, f + "_bits", int(mq[f"{k}_bits"]))
            setattr(s, f + "_exp", int(mq[f"{k}_exp"]))
        for k in ("u", "Bu_re", "Bu_im", "x_re", "x_im", "y"):
            setattr(s, k + "_bits", int(mq[f"{k}_bits"]))
            setattr(s, k + "_exp", int(mq[f"{k}_exp"]))
        n, nq = lp["norm"], lq["norm"]
        bn: NormDesc = L.norm
        bn.minus_mean = self._ptr(-np.asarray(n["mean"], dtype=np.int64))  # export() stores +mean (:949)
        bn.invsq_var = self._ptr(n["invsq_var"])
        bn.mean_bits, bn.mean_exp = int(nq["mean_bits"]), int(nq["mean_exp"])
        bn.invsq_var_bits, bn.invsq_var_exp = int(nq["invsq_var_bits"]), int(nq["invsq_var_exp"])
        if "scale" in n:
            bn.scale, bn.scale_bits, bn.scale_exp = self._ptr(n["scale"]), int(nq["scale_bits"]), int(nq["scale_exp"])
        if "bias" in n:
            bn.bias, bn.bias_bits, bn.bias_exp = self._ptr(n["bias"]), int(nq["bias_bits"]), int(nq["bias_exp"])
        for k in ("l_bits", "l_exp", "r_bits", "r_exp", "res_bits", "res_exp"):
            setattr(L, k, int(lq["multgate"][k]))
        sg = lq["sigmoid"]
        if int(sg.get("x_extra", 3)) != 3 or int(sg.get("n_exp", 3)) != 3:
            raise NotImplementedError("only the reference's 8-entry sigmoid LUT is implemented")
        L.sig_x_exp, L.sig_y_exp = int(sg["x_exp"]), int(sg["y_exp"])
        from .fxpmodel import sigmoid_lut
        lut = sigmoid_lut(L.sig_x_exp, L.sig_y_exp)
        for j in range(8):
            L.lut[j] = int(lut[j])

    # -- forward ---------------------------------------------------------------------------------
    LEVEL_FLAGS = (_lib.FWD_DEFER_REDO, _lib.FWD_DEFER_REDO | _lib.FWD_NO_PAIR, _lib.FWD_EXACT)

    @property
    def redo_seen(self) -> int:  # kept for callers of the two-rung form: 2 = "go straight to the exact kernels"
        return 2 if self.level >= 2 else 0

    @redo_seen.setter
    def redo_seen(self, v: int) -> None:
        if v >= 2:
            self.level = 2

    def note_redo(self, level: int) -> int:
        """An optimistic forward at `level` came back with ST_REDO: returns the rung to repeat it on."""
        if level < 2:
            self._redos[level] += 1
            if self._redos[level] >= 2 and self.level <= level:
                self.level = level + 1
        return min(level + 1, 2)

    def run_ladder(self, launch: Callable[[int], None], check: Callable[[], np.ndarray]) -> None:
        """launch(flags) enqueues the forward, check() returns the status words (after synchronising)."""
        level = self.level
        while True:
            launch(self.LEVEL_FLAGS[level])
            st = check()
            if not (st[0] & _lib.ST_REDO) or level >= 2:
                return
            level = self.note_redo(level)

    @property
    def status(self) -> torch.Tensor:
        return self._status[0]

    def lane_status(self, lane: int, groups: int = 1) -> torch.Tensor:
        """The status words of `lane`: STATUS_WORDS per group, group after group (a plain forward is one group)."""
        st = self._status.get(lane)
        if st is None or st.numel() < groups * _lib.STATUS_WORDS:
            st = self._status[lane] = torch.zeros(groups * _lib.STATUS_WORDS, dtype=torch.int32, device=self.device)
        return st

    def workspace(self, B: int, L: int, lane: int = 0, groups: int = 1) -> torch.Tensor:
        key, ws = self._wsl.get(lane, (None, None))
        if key != (B, L, groups):
            n = groups * lib.s5fxp_workspace_bytes(self._h, B, L)
            ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._wsl[lane] = ((B, L, groups), ws)
        return ws

    def enqueue(self, x: torch.Tensor, x_bits: int, x_exp: int, y: torch.Tensor, B: int, L: int,
                traces: Optional[List[Dict[str, torch.Tensor]]] = None, allreduce: Optional[Callable] = None,
                scan_events: Optional[list] = None, flags: int = 0, lane: int = 0,
                state_in: Optional[torch.Tensor] = None, state_out: Optional[torch.Tensor] = None, groups: int = 1,
                gate_events: Optional[list] = None) -> None:
        """Launches one forward on the current stream; nothing is synchronised.

        groups = G > 1: x and y hold G * B sequences, G independent reference batches of B sequences each (what G calls
        would compute: own exponents, status words and carry per group) enqueued as ONE set of kernel launches
        (include/s5fxp.h, s5fxp_forward_opts::groups).

        flags: _lib.FWD_DEFER_REDO drops the (normally idle) gated exact re-run launches -- the caller must then
        read the status words and repeat with _lib.FWD_EXACT when ST_REDO is set (``forward`` does)."""
        if groups > 1 and (traces is not None or allreduce is not None):
            raise ValueError("a grouped forward takes neither traces nor a cross-rank hook: run the groups one by one")
        ws = self.workspace(B, L, lane, groups)
        self._groups[lane] = groups
        tr = None
        if traces is not None:
            tr = (LayerTrace * self.n_layers)()
            for i, d in enumerate(traces):
                for k in TRACE_FIELDS:
                    if k in d:
                        setattr(tr[i], k, d[k].data_ptr())
        opts = _lib.ForwardOpts()
        if allreduce is not None:
            ws_base = ws.data_ptr()

            def _cb(ctx, dev_ptr, n, stream):  # noqa: ANN001
                try:
                    # the maxima live in the workspace: hand the hook a float32 view of exactly those n words
                    off = int(dev_ptr) - ws_base
                    allreduce(ws[off:off + 4 * int(n)].view(torch.float32))
                    return 0
                except Exception:  # pragma: no cover - surfaced as EHIP by the C side
                    import traceback
                    traceback.print_exc()
                    return 1
            opts.allreduce = _lib.ALLREDUCE_FN(_cb)
        if scan_events is not None:
            assert len(scan_events) == 2 * self.n_layers
            arr = (C.c_void_p * len(scan_events))(*[(e.cuda_event if e is not None else None) for e in scan_events])
            opts.scan_events = C.cast(arr, C.POINTER(C.c_void_p))
            self._ev_keep = arr
        if gate_events is not None:
            assert len(gate_events) == 2 * self.n_layers
            arr2 = (C.c_void_p * len(gate_events))(*[(e.cuda_event if e is not None else None) for e in gate_events])
            opts.gate_events = C.cast(arr2, C.POINTER(C.c_void_p))
            self._ev_keep2 = arr2
        opts.flags = int(flags)
        opts.groups = int(groups)
        want = (self.n_layers, 2, B, self.P) if groups == 1 else (groups, self.n_layers, 2, B, self.P)
        for name, t in (("state_in", state_in), ("state_out", state_out)):
            if t is not None:
                if t.dtype != torch.int32 or tuple(t.shape) != want or not t.is_contiguous() or not t.is_cuda:
                    raise ValueError(f"{name} must be a contiguous int32 device tensor of shape {want}")
                setattr(opts, name, t.data_ptr())
        self._cb_keep = opts
        check(lib.s5fxp_model_forward(self._h, x.data_ptr(), x_bits, x_exp, B, L, y.data_ptr(), ws.data_ptr(),
                                      ws.numel(), self.lane_status(lane, groups).data_ptr(),
                                      C.cast(tr, C.POINTER(LayerTrace)) if tr is not None else None, C.byref(opts),
                                      torch.cuda.current_stream().cuda_stream), "s5fxp_model_forward")

    def check_status(self, lane: int = 0) -> np.ndarray:
        """Reads the status words back (one sync) and raises what the reference would have raised.  After a grouped
        forward word [0] of the returned array carries the error bits of ALL groups (the per-group words follow at
        multiples of STATUS_WORDS)."""
        groups = self._groups.get(lane, 1)
        st = self.lane_status(lane, groups).cpu().numpy()[:groups * _lib.STATUS_WORDS].copy()
        for g in range(1, groups):
            st[0] |= st[g * _lib.STATUS_WORDS]
        if st[0] & _lib.ST_NEGSHIFT:
            raise ValueError("invalid result_exp: a data-dependent shift came out negative (fxparray.py:619-621)")
        if st[0] & _lib.ST_NEGEXP:
            raise ValueError("a compute_best exponent came out negative")
        if st[0] & _lib.ST_WIDE_INPUT:
            raise OverflowError("input FxpArray holds values beyond 24 bits; rebuild the engine with "
                                "flags=MODEL_FORCE_GENERIC")
        return st

    def layer_exponents(self) -> List[Dict[str, int]]:
        st = self.status.cpu().numpy()
        names = ("norm_input_minus_mean", "norm_output_raw", "norm_output_scaled", "norm_output_scaled_bias", "residadd")
        return [{n: int(st[8 + 8 * i + j]) for j, n in enumerate(names)} for i in range(self.n_layers)]

    def forward(self, x: FxpArray, traces: bool = False, allreduce: Optional[Callable] = None, check_status: bool = True):
        """x: FxpArray (B,L,d_in) or (L,d_in).  Returns an FxpArray (and the traces when asked)."""
        data = x.data.contiguous()
        if data.shape[-1] != self.d_in:
            raise ValueError(f"expected last dim {self.d_in}, got {tuple(data.shape)}")
        B, L = (1, data.shape[0]) if data.ndim == 2 else (data.shape[0], data.shape[1])
        y = torch.empty(tuple(data.shape[:-1]) + (self.d_out,), dtype=torch.int32, device=data.device)
        if B * L == 0:
            return (FxpArray(y, self.out_bits, self.out_exp, True), []) if traces else FxpArray(y, self.out_bits, self.out_exp, True)
        tr = None
        if traces:
            tr = []
            for _ in range(self.n_layers):
                d = {}
                for k in TRACE_FIELDS:
                    w = self.P if k in ("Bu_re", "Bu_im", "xs_re", "xs_im") else self.H
                    d[k] = torch.empty(tuple(data.shape[:-1]) + (w,), dtype=torch.int32, device=data.device)
                tr.append(d)
        if not check_status:
            self.enqueue(data, x.bits, x.exp, y, B, L, tr, allreduce)  # self-contained: exact re-run enqueued, gated
        else:
            # the status words are read anyway: run optimistically and repeat with the exact kernels if a state
            # left the fast recurrence's range (never with a multi-rank hook: ranks must enqueue the same work)
            if allreduce:  # self-contained: the gated exact kernels are part of the one enqueue
                self.enqueue(data, x.bits, x.exp, y, B, L, tr, allreduce, flags=0)
                self.check_status()
            else:
                self.run_ladder(lambda fl: self.enqueue(data, x.bits, x.exp, y, B, L, tr, None, flags=fl), self.check_status)
        out = FxpArray(y, self.out_bits, self.out_exp, True)
        return (out, tr) if traces else out


    def layer_forward(self, layer: int, x: FxpArray, traces: bool = False):
        """One ``FxpSequenceLayer.forward`` (sparseRNNs/fxpmodel.py:1110-1161) through ``s5fxp_layer_forward``: x is the
        layer's input (B,L,H) or (L,H) with its own bits / exponent; returns the layer's output FxpArray (its exponent is the
        one the residual compute_best add chose on the device), and the layer's traces when asked."""
        data = x.data.contiguous()
        if data.shape[-1] != self.H:
            raise ValueError(f"expected last dim {self.H}, got {tuple(data.shape)}")
        B, L = (1, data.shape[0]) if data.ndim == 2 else (data.shape[0], data.shape[1])
        y = torch.empty_like(data)
        ws = self.workspace(B, L)
        self._groups[0] = 1
        tr, d = None, None
        if traces:
            tr = (LayerTrace * 1)()
            d = {}
            for k in TRACE_FIELDS:
                wd = self.P if k in ("Bu_re", "Bu_im", "xs_re", "xs_im") else self.H
                d[k] = torch.empty(tuple(data.shape[:-1]) + (wd,), dtype=torch.int32, device=data.device)
                setattr(tr[0], k, d[k].data_ptr())
        e = torch.zeros(1, dtype=torch.int32, device=data.device)
        check(lib.s5fxp_layer_forward(self._h, layer, data.data_ptr(), x.bits, x.exp, B, L, y.data_ptr(), e.data_ptr(), ws.data_ptr(),
                                      ws.numel(), self.lane_status(0).data_ptr(),
                                      C.cast(tr, C.POINTER(LayerTrace)) if tr is not None else None, None,
                                      torch.cuda.current_stream().cuda_stream), "s5fxp_layer_forward")
        self.check_status()
        out = FxpArray(y, lib.s5fxp_model_layer_out_bits(self._h, layer), int(e.item()), True)
        return (out, d) if traces else out

    def forward_batches(self, x: FxpArray, batch: int) -> FxpArray:
        """x: (G * batch, L, d_in) -- G independent reference batches of `batch` sequences each (the reference's
        run_validation loop over a loader, sparseRNNs/fxprun.py:53-88, several batches per call).  Returns what G calls of
        ``forward`` would, as one (G * batch, L, d_out) FxpArray, from ONE set of kernel launches."""
        data = x.data.contiguous()
        if data.ndim != 3 or data.shape[-1] != self.d_in or data.shape[0] % batch:
            raise ValueError(f"expected (G * {batch}, L, {self.d_in}), got {tuple(data.shape)}")
        G, L = data.shape[0] // batch, data.shape[1]
        y = torch.empty(tuple(data.shape[:-1]) + (self.d_out,), dtype=torch.int32, device=data.device)
        if G * batch * L:
            self.run_ladder(lambda fl: self.enqueue(data, x.bits, x.exp, y, batch, L, flags=fl, groups=G), self.check_status)
        return FxpArray(y, self.out_bits, self.out_exp, True)

    # -- streaming ------------------------------------------------------------------------------
    def zero_state(self, B: int) -> torch.Tensor:
        """The carry a sequence starts with: (n_layers, 2, B, P) int32 zeros (re plane, im plane per layer)."""
        return torch.zeros((self.n_layers, 2, B, self.P), dtype=torch.int32, device=self.device)

    def forward_chunk(self, x: FxpArray, state: Optional[torch.Tensor] = None):
        """One chunk of a stream: x (B,L,d_in) or (L,d_in); `state` from zero_state() or the previous call (None: zeros).
        Returns (y, new_state).  What comes out is what the reference computes for THIS chunk when its recurrences
        (sparseRNNs/fxpmodel.py:147-172, the carry is an explicit argument of the step function) start from `state`;
        every chunk is its own compute_best batch.  `state` is not modified."""
        data = x.data.contiguous()
        if data.shape[-1] != self.d_in:
            raise ValueError(f"expected last dim {self.d_in}, got {tuple(data.shape)}")
        B, L = (1, data.shape[0]) if data.ndim == 2 else (data.shape[0], data.shape[1])
        if B * L == 0:
            raise ValueError("empty chunk")
        y = torch.empty(tuple(data.shape[:-1]) + (self.d_out,), dtype=torch.int32, device=data.device)
        new_state = torch.empty((self.n_layers, 2, B, self.P), dtype=torch.int32, device=data.device)
        # `state` is never written: a chunk that comes back with ST_REDO is repeated from it on the next rung
        self.run_ladder(lambda fl: self.enqueue(data, x.bits, x.exp, y, B, L, flags=fl, state_in=state, state_out=new_state),
                        self.check_status)
        return FxpArray(y, self.out_bits, self.out_exp, True), new_state

    def stream(self, B: int = 1) -> "StreamingSession":
        return StreamingSession(self, B)


class StreamingSession:
    """Frame-chunk-at-a-time inference with the SSM states carried between calls (the paper's
    real-time denoising use).  ``push`` takes (B,L,d_in) / (L,d_in) chunks of any length and returns the outputs of
    exactly those frames."""

    def __init__(self, engine: "Engine", B: int = 1):
        self.engine, self.B = engine, B
        self.state = engine.zero_state(B)
        self.frames = 0

    def push(self, x: FxpArray) -> FxpArray:
        y, self.state = self.engine.forward_chunk(x, self.state)
        self.frames += x.data.shape[-2]
        return y

    def reset(self) -> None:
        self.state = self.engine.zero_state(self.B)
        self.frames = 0


class InflightRunner:
    """Keeps up to ``depth`` forwards of one Engine in flight, each on its own HIP stream and lane.

    One layer's recurrence is a latency chain that occupies B*P/16 waves; the projections around it want the
    whole chip.  Within one forward they cannot overlap (every layer's exponents depend on the whole previous
    layer), but the recurrence of one batch overlaps the projections of the others.  Batches are independent,
    so the results are the ones ``Engine.forward`` gives.

    submit() returns at once; a lane is synchronised and its status words are checked (with the exact re-run if
    ST_REDO came back) when the lane is reused or on drain().
    """

    def __init__(self, engine: Engine, depth: int = 3):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.engine, self.depth = engine, depth
        self.streams = [torch.cuda.Stream(device=engine.device) for _ in range(depth)]
        self._pending: List[Optional[tuple]] = [None] * depth
        self._next = 0
        # engine lanes 1..depth: lane 0 (its workspace and status words) stays Engine.forward's, which may run on
        # another stream while batches are in flight here
        self._lane0 = 1

    def submit(self, x: torch.Tensor, x_bits: int, x_exp: int, y: torch.Tensor, B: int, L: int, check: bool = True,
               scan_events: Optional[list] = None, groups: int = 1) -> int:
        """check=False skips the status check of the lane's previous batch (only sound when every batch of the lane
        is the same input, as in bench.py: the last check then speaks for all)."""
        lane = self._next
        self._next = (lane + 1) % self.depth
        if check:
            self._finish(lane)
        s = self.streams[lane]
        s.wait_stream(torch.cuda.current_stream(self.engine.device))
        level = self.engine.level
        with torch.cuda.stream(s):
            self.engine.enqueue(x, x_bits, x_exp, y, B, L, flags=Engine.LEVEL_FLAGS[level], lane=self._lane0 + lane,
                                scan_events=scan_events, groups=groups)
        # the tensors were allocated on another stream: tell the caching allocator that this lane's stream uses them, so
        # that dropping the previous job's references below (check=False) cannot hand their memory out while kernels of
        # this stream still read or write it
        x.record_stream(s)
        y.record_stream(s)
        self._pending[lane] = (x, x_bits, x_exp, y, B, L, level, groups)
        return lane

    def _finish(self, lane: int) -> None:
        job = self._pending[lane]
        if job is None:
            return
        self._pending[lane] = None
        s = self.streams[lane]
        s.synchronize()
        x, x_bits, x_exp, y, B, L, level, groups = job
        st = self.engine.check_status(self._lane0 + lane)
        while (st[0] & _lib.ST_REDO) and level < 2:   # climb the ladder: pair -> quad -> exact
            level = self.engine.note_redo(level)
            with torch.cuda.stream(s):
                self.engine.enqueue(x, x_bits, x_exp, y, B, L, flags=Engine.LEVEL_FLAGS[level], lane=self._lane0 + lane, groups=groups)
            s.synchronize()
            st = self.engine.check_status(self._lane0 + lane)

    def lane_of(self, slot: int) -> int:
        """Engine lane (status words, workspace) of in-flight slot `slot`."""
        return self._lane0 + slot

    def drain(self) -> None:
        for lane in range(self.depth):
            self._finish(lane)
