"""Host-side plan of the CPC train step: buffers in HBM + the launch sequence over the C ABI.

One ``CPCEngine`` = one (batch size, clip length, storage dtype) instance of the path
    AudioEncoder.forward -> context network -> prediction_model -> score/InfoNCE -> backward -> Adam
(reference: audio_model.py:193-211 and contrastive_estimation_training.py:97-162).  The context network is pluggable:
GRUContext (AudioGRUModel), ConvArContext / scalogram_engine.ConvArGridContext (ConvolutionalArModel), AttentionContext
(AttentionModel); scalogram_engine.ScalogramCPCEngine swaps the encoder for the 2-D residual encoder.  PyTorch is used for
device memory, streams and a few strided copies only; every arithmetic step is a HIP kernel behind libcpc_hip.so.

Data layout (all per GPU, resident for the life of the engine):
  act[l], dact[l]   storage dtype, channels-last [B][L_alloc[l]][C_l], zero pad rows, zero guards front/back
  L_alloc[l-1] = stride_l * L_alloc[l]  so that a strided conv is a GEMM whose A rows overlap uniformly
  flat parameters / gradients / Adam moments: f32, one contiguous buffer each (owned by the model)
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch

from . import _hip


_SIDE_STREAMS = {}


def side_stream(device):
    """ONE high-priority side stream per device for the whole process, shared by every engine: HIP multiplexes its streams
    onto a few hardware queues, and an engine created late in a process that had made one stream per engine ended up with a
    side stream sharing a hardware queue with the main stream (measured: the same step 6.9 -> 12.9 ms)."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=-1)
    return _SIDE_STREAMS[key]


def _ceil_div(a, b):
    return -(-a // b)


class EncoderGeometry:
    """Valid and allocated lengths of every encoder layer for clips of ``length`` samples."""

    def __init__(self, length: int, strides: Sequence[int], kernel_sizes: Sequence[int]):
        self.strides = [int(s) for s in strides]
        self.kernels = [int(k) for k in kernel_sizes]
        n = len(self.strides)
        cur = int(length)
        self.valid: List[int] = []
        for s, k in zip(self.strides, self.kernels):
            cur = (cur - k) // s + 1
            if cur <= 0:
                raise ValueError(f"clips of {length} samples are shorter than the encoder's receptive field")
            self.valid.append(cur)
        self.taps = [_ceil_div(k, s) for s, k in zip(self.strides, self.kernels)]    # D_l: output rows per input position
        pad_top = 1
        while True:
            alloc = [0] * n
            alloc[n - 1] = self.valid[n - 1] + pad_top
            for l in range(n - 1, 0, -1):
                alloc[l - 1] = self.strides[l] * alloc[l]
            if all(alloc[l] - self.valid[l] >= max(self.taps[l] - 1, 0) for l in range(n)):
                break
            pad_top += 1
            if pad_top > 4096:
                raise ValueError("could not find a padded layout for this encoder configuration")
        self.alloc = alloc
        self.frames = self.valid[-1]


class CPCEngine:
    def __init__(self, model, batch_size: int, length: int, device, dtype: torch.dtype):
        enc, ar = model.encoder, model.autoregressive_model
        self.model = model
        self.device = torch.device(device)
        self.dt = dtype
        self.code = _hip.dtype_code(dtype)
        self.B, self.L = int(batch_size), int(length)
        self.strides = list(enc.strides)
        self.kernels = list(enc.kernel_sizes)
        self.channels = list(enc.channel_count)
        self.n = len(self.strides)
        self.E = self.channels[-1]
        self.H = int(model.ar_size)
        self.K = int(model.prediction_steps)
        self.V = int(model.visible_steps)
        full = EncoderGeometry(self.L, self.strides, self.kernels)
        if full.frames < self.V + self.K:
            raise ValueError(f"clips give {full.frames} encoder frames, need visible+prediction = {self.V + self.K}")
        # The model only consumes the last V+K encoder frames (audio_model.py:197-198); frame t depends on samples
        # >= t * downsampling only, so the leading frames the reference computes and discards are never computed here:
        # the encoder runs on the clip from sample x_off on.  Results (outputs, loss, gradients) are unchanged.
        self.frames_full = full.frames
        skip = full.frames - (self.V + self.K) if (self.V + self.K) > 0 else 0
        self.x_off = skip * int(enc.downsampling_factor)
        self.L_eff = self.L - self.x_off
        self.geo = EncoderGeometry(self.L_eff, self.strides, self.kernels)
        self.T = self.geo.frames
        assert self.T == full.frames - skip
        if model.enc_size != self.E:
            raise ValueError("enc_size does not match the encoder's last channel count")
        self._check_supported()
        model._flatten_parameters(self.device)
        # high-priority side stream for short kernels that need not sit between the large GEMMs (see _alloc_encoder)
        self.aux = side_stream(self.device)
        self.ctx = make_context(self, ar) if (self.V + self.K) > 0 else None
        self._alloc()

    # ------------------------------------------------------------------------------------------ side stream
    use_aux = os.environ.get("CPC_SIDE_STREAM", "1") != "0"     # False: everything on the launching stream (GraphedStep captures that way; CPC_SIDE_STREAM=0 for A/B runs)

    @contextlib.contextmanager
    def side(self, ev):
        """Work issued inside runs on the high-priority side stream once everything issued so far on the current stream
        has finished (event ``ev``); backward() joins the side stream before it returns."""
        if not self.use_aux:
            yield
            return
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.aux):
            self.aux.wait_event(ev)
            yield

    # ------------------------------------------------------------------------------------------ setup
    def _check_supported(self):
        ch = 8 if self.dt == torch.bfloat16 else 4
        for l in range(1, self.n):
            if (self.kernels[l] * self.channels[l - 1]) % ch or self.channels[l] % 8:
                raise NotImplementedError("HIP conv path needs channel counts that are multiples of 8")
        c0 = self.channels[0]
        if c0 % 8 or 256 % (c0 // 8) or self.kernels[0] > 16 or self.strides[0] > 8:
            raise NotImplementedError("HIP layer-1 kernel: channels in {8,16,...,2048}, kernel <= 16, stride <= 8")
        if self.E % ch:
            raise NotImplementedError("enc_size must be a multiple of 8")

    def _buf(self, rows: int, cols: int, guard_rows: int = 16):
        """Zeroed [rows][cols] storage-dtype buffer with ``guard_rows`` (>= 16) zero rows on both sides; returns (full, view,
        guard_elems).  The guards are what the overlapped-row GEMMs read beyond the array (include/cpc_hip.h, guard contract):
        kernel - stride rows after a convolution input, ceil(kernel / stride) - 1 rows before an output gradient."""
        guard = max(16, int(guard_rows)) * cols
        full = torch.zeros(guard + rows * cols + guard, device=self.device, dtype=self.dt)
        return full, full[guard:guard + rows * cols], guard

    def _alloc(self):
        need = self._alloc_encoder()
        self._alloc_head(need)

    def _alloc_encoder(self):
        """Activation / gradient buffers and weight operands of the AudioEncoder stack; returns the slab sizes it needs."""
        dev, dt = self.device, self.dt
        B, n = self.B, self.n
        La = self.geo.alloc
        self.act, self.dact = [], []
        self._keep = []
        # guard rows: act[l] is read (kernel - stride) rows past its end by layer l+1's forward / weight gradient, dact[l] is
        # read (taps - 1) rows before its start by layer l's data gradient
        self.guard_rows = max([16] + [self.kernels[l] - self.strides[l] for l in range(1, n)] + [t - 1 for t in self.geo.taps[1:]])
        self.guard = []
        for l in range(n):
            for store in (self.act, self.dact):
                full, view, g_el = self._buf(B * La[l], self.channels[l], self.guard_rows)
                self._keep.append(full)
                store.append(view)
            self.guard.append(g_el)
        # Sign-bit mask of layer 1's output (include/cpc_hip.h, cpc_sign_bits): act[0] > 0 as one byte per 8 elements, written by the
        # layer-1 forward kernel and read by the data gradient of layer 2 (the fused one) in place of act[0]: the tile's 8 KiB of
        # bits arrive by LDS-DMA before its K loop, so the epilogue of a tile reads no mask from memory (it used to end in a 32 MB
        # burst per round of tiles; 0.96 GB of the launch's traffic).  Measured (tools/bits_ab.py, B = 256): that launch 1 035 ->
        # 950 us, layer-1 forward +20 us.  For the deeper layers the same trade is even (the forward GEMMs' epilogues pay 2-6 % for
        # writing the bits — one workgroup per CU, nothing to hide an epilogue instruction behind — and a separate kernel costs what
        # the data gradients gain — also when that kernel runs on the side stream while the GRU leaves 240 CUs idle: 4.52 -> 4.57 ms per
        # step), so they keep reading their masks from the activations.  CPC_MASK_BITS=0: plain masks (A/B).
        self.act_bits: List[Optional[torch.Tensor]] = [None] * n
        if (dt == torch.bfloat16 and os.environ.get("CPC_MASK_BITS", "1") != "0" and n >= 2 and self.channels[0] == 512
                and _hip.nt_tile(self.code, B * La[1], self.strides[1] * 512, self.geo.taps[1] * self.channels[1]) == 256
                and self.guard[0] % 128 == 0):
            full = torch.zeros((2 * self.guard[0] + B * La[0] * 512) // 8, device=dev, dtype=torch.uint8)
            self._keep.append(full)
            self.act_bits[0] = full[self.guard[0] // 8:]
        # Bias gradients of layers 2 .. n-1 (indices 1 .. n-2) from the data gradient of the layer above: its epilogue leaves the
        # column sums of every 256-row tile it stores (include/cpc_hip.h, dx_colsum_slabs), and a small reduction replaces the
        # column-sum pass over the 0.06-0.24 GB gradient (beside the main-stream GEMMs those passes cost the step 58 us,
        # CPC_PROBE runs).  bf16 and 256 x 256 tiles only; the top layer's gradient comes from other kernels and keeps its pass.
        self.cs_slabs: List[Optional[torch.Tensor]] = [None] * n       # cs_slabs[l]: sums of dact[l], written by layer l+1's data gradient
        if dt == torch.bfloat16 and os.environ.get("CPC_FUSED_COLSUM", "1") != "0":
            for l in range(2, n):
                cin, cout, s = self.channels[l - 1], self.channels[l], self.strides[l]
                if _hip.nt_tile(self.code, B * La[l], s * cin, self.geo.taps[l] * cout) == 256 and (s * cin) % 8 == 0:
                    nf = int(_hip.lib().cpc_conv_dgrad_colsum_floats(B, cin, s, La[l]))
                    self.cs_slabs[l - 1] = torch.zeros(nf, device=dev, dtype=torch.float32)
        # weight operand layouts (storage dtype)
        self.w_fwd: List[Optional[torch.Tensor]] = [None] * n
        self.w_dgrad: List[Optional[torch.Tensor]] = [None] * n
        self._w_dgrad_alt: List[Optional[torch.Tensor]] = [None] * n      # see prepare_ahead
        for l in range(1, n):
            cin, cout, kw, s = self.channels[l - 1], self.channels[l], self.kernels[l], self.strides[l]
            self.w_fwd[l] = torch.empty(cout * kw * cin, device=dev, dtype=dt)
            self.w_dgrad[l] = torch.empty(s * cin * self.geo.taps[l] * cout, device=dev, dtype=dt)
        need = [1]
        self.nsplit = [1] * n
        for l in range(1, n):
            I, J, M = self.kernels[l] * self.channels[l - 1], self.channels[l], B * La[l]
            self.nsplit[l] = self._pick_split(I, J, M)
            need.append(self.nsplit[l] * I * J)
        self.c1_blocks = max(1, min(8, _ceil_div(self.geo.valid[0], 512)))
        self.c1_item_blocks = min(B, 128)
        need.append(self.c1_item_blocks * self.c1_blocks * (self.kernels[0] + 1) * self.channels[0])
        self.colsum_blocks = 1024
        need.append(self.colsum_blocks * max(self.channels))
        # bf16, wide first layer: the data gradient of layer 2 is fused with the weight / bias gradient of layer 1
        # (cpc_conv_dgrad_conv1): the 1 GB gradient of layer 1's output is neither written nor read back
        self.fuse_c1 = False
        if n >= 2 and dt == torch.bfloat16 and self.channels[0] % 256 == 0 and self.kernels[0] <= 15:
            s1, c0, c1 = self.strides[1], self.channels[0], self.channels[1]
            if _hip.nt_tile(self.code, B * La[1], s1 * c0, self.geo.taps[1] * c1) == 256 and (self.geo.taps[1] * c1) % 64 == 0:
                nf = [int(_hip.lib().cpc_conv_dgrad_conv1_floats(B, c0, s1, La[1], self.kernels[0], w)) for w in (0, 1)]
                self.c1_slabs = torch.empty(nf[0], device=dev, dtype=torch.float32)
                self.c1_tmp = torch.empty(nf[1], device=dev, dtype=torch.float32)
                self.fuse_c1 = True
        # Side stream for the short, latency-bound kernels of the weight-gradient path (bias column sums, slab reductions): they run beside the large GEMMs of the main stream instead of between them.  The big
        # GEMMs all stay on the main stream.  Each layer's weight-gradient slabs get their own buffer so that the next layer's
        # GEMM never waits for the previous reduction.
        self.wslab = [None] + [torch.empty(need[l], device=dev, dtype=torch.float32) for l in range(1, n)]
        self.aux_slabs = torch.empty(self.colsum_blocks * max(self.channels), device=dev, dtype=torch.float32)
        self._ev_d = [torch.cuda.Event() for _ in range(n)]
        self._ev_w = [torch.cuda.Event() for _ in range(n)]
        return need

    def _alloc_head(self, need):
        """Predictor / loss state, the context network's buffers and the shared split-reduction workspace."""
        dev, dt, f32 = self.device, self.dt, torch.float32
        B, H, E, K = self.B, self.H, self.E, self.K
        self.w_p = torch.empty(K * E * H, device=dev, dtype=dt)           # [K*E][H]
        self.w_p_t = torch.empty(H * K * E, device=dev, dtype=dt)         # [H][K*E]
        self.c = torch.empty(B, H, device=dev, dtype=f32)
        self.pred = torch.empty(B * K * E, device=dev, dtype=dt)
        self.ldS = _ceil_div(B, 8) * 8
        self.S = torch.zeros(K * B * self.ldS, device=dev, dtype=f32)
        self.dS = torch.zeros(K * B * self.ldS, device=dev, dtype=dt)
        self.dST = torch.zeros(K * B * self.ldS, device=dev, dtype=dt)
        self.nce_out = torch.zeros(8, device=dev, dtype=f32)
        self.nce_ws = torch.empty(int(_hip.lib().cpc_nce_workspace_floats(B, K)), device=dev, dtype=f32)
        self.dpred = torch.zeros(B * K * E, device=dev, dtype=dt)
        self.dc = torch.zeros(B, H, device=dev, dtype=f32)
        need = list(need)
        need.append(_ceil_div(max(K * E, 1), 256) * B * H)               # split-K slabs of the dc GEMM
        if self.ctx is not None:
            self.ctx.allocate()
            need.append(self.ctx.slab_floats())
        self.slabs = torch.empty(max(need), device=dev, dtype=f32)

    def _pick_split(self, I, J, M, dt=None):
        dt = self.dt if dt is None else dt
        big = dt == torch.bfloat16 and I >= 256 and J >= 256       # 256x256 tiles, one workgroup per CU
        tile = 256 if big else 128
        tiles = _ceil_div(I, tile) * _ceil_div(J, tile)
        blk = 64 if dt == torch.bfloat16 else 32
        # as many splits as fit in ONE round of workgroups (256 CUs x one 256-tile or two 128-tile workgroups): rounding the
        # count up instead (30 tiles x 18 splits = 540 workgroups on 512 slots) leaves a second, almost empty round
        want = max(1, (256 if big else 512) // tiles)
        return max(1, min(want, _ceil_div(M, 8 * blk), 256))

    def _chunk(self, M, nsplit, dt=None):
        blk = 64 if (self.dt if dt is None else dt) == torch.bfloat16 else 32
        return _ceil_div(_ceil_div(M, nsplit), blk) * blk

    # ---------------------------------------------------------------------------------- weight layouts
    def prepare_weights(self):
        """f32 master parameters (reference state_dict shapes) -> storage-dtype GEMM operand layouts.  Skipped when
        prepare_ahead() already rebuilt every operand copy after the last parameter update and nothing has touched the
        parameters since."""
        token, self._ahead_token = getattr(self, "_ahead_token", None), None
        ev = getattr(self, "_ahead_ev", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev[1])      # a side-stream rebuild (_prepare_all_ahead), current or stale, has finished
        if token is not None and token == self._param_state():
            return
        self._prepare_encoder_weights()
        self._prepare_head_weights()

    def _prepare_all_ahead(self):
        """Every operand copy of the next step rebuilt on the side stream, behind everything issued so far (the step's last Adam launch):
        the layout kernels then run beside the head of the next step instead of in front of its first convolution.  prepare_weights()
        waits for their event."""
        if getattr(self, "_ahead_ev", None) is None:
            self._ahead_ev = (torch.cuda.Event(), torch.cuda.Event())
        with self.side(self._ahead_ev[0]):
            self._prepare_encoder_weights()
            self._prepare_head_weights()
            self._ahead_ev[1].record(self.aux)
        self._ahead_token = self._param_state()

    def _prepare_head_weights(self):
        p, code = self.model._param, self.code
        H, E, K = self.H, self.E, self.K
        if self.ctx is None:          # encoder-only engine (stand-alone AudioEncoder call)
            return
        self.ctx.prepare_weights()
        w_p = p["prediction_model.weight"]
        _hip.call("cpc_cast2d", _hip.ptr(w_p), _hip.ptr(self.w_p), K * E, H, H, 1, code)
        _hip.call("cpc_cast2d", _hip.ptr(w_p), _hip.ptr(self.w_p_t), H, K * E, 1, H, code)

    def _prepare_encoder_weights(self):
        for l in range(1, self.n):
            self._prepare_conv(l, True, True)

    def _prepare_conv(self, l, fwd, dgrad, alt=False):
        w = self.model._param[f"encoder.layers.{l}.weight"]
        wd = None
        if dgrad:
            wd = self.w_dgrad[l]
            if alt:          # the current copy is still being read: build the next one in a second buffer (swapped in later)
                if self._w_dgrad_alt[l] is None:
                    self._w_dgrad_alt[l] = torch.empty_like(self.w_dgrad[l])
                wd = self._w_dgrad_alt[l]
        _hip.call("cpc_conv_w_prep", _hip.ptr(w), _hip.ptr(self.w_fwd[l]) if fwd else None, _hip.ptr(wd), self.channels[l],
                  self.channels[l - 1], self.kernels[l], self.strides[l], self.code)

    # Operand copies for the NEXT step, rebuilt as soon as the optimizer has updated their parameters (single-process training):
    # the ten layout kernels (0.1 ms at B = 256) then run on the side stream beside the remaining backward GEMMs instead of in
    # front of the next step's first convolution.
    supports_prepare_ahead = True

    def _param_state(self):
        """Changes whenever the parameters may have changed: torch's version counters catch in-place torch ops on the
        parameters (load_state_dict, torch optimizers), model._raw_updates counts the fused optimizer's raw-pointer updates."""
        m = self.model
        return (sum(p._version for p in m.parameters()) + m._flat_param._version, getattr(m, "_raw_updates", 0),
                m._flat_param.data_ptr())

    def prepare_ahead(self, lo, hi, final):
        """FusedAdam calls this right after it updated flat_param[lo:hi): from the backward pass's grad_ready_hook (on the
        side stream; ``final`` False) and from step() for the head of the buffer (``final`` True, main stream, backward
        complete).  In a hook call the lowest updated encoder layer's data-gradient operand is still being read by that
        layer's data-gradient GEMM on the main stream: its next copy is built in a second buffer, swapped in by the final call."""
        if not self.supports_prepare_ahead or not getattr(self.ctx, "ahead_ok", False) or os.environ.get("CPC_PREPARE_AHEAD", "1") == "0":
            return
        st = getattr(self, "_ahead", None)
        # (one Adam launch over everything, no gradient-ready pieces: the copies are rebuilt here, on the main stream behind Adam.  Doing that on
        # the side stream instead -- _prepare_all_ahead, as the scalogram engine does beside its CQT GEMMs -- was measured SLOWER for this
        # engine: the next step opens with the HBM-bound layer-1 kernel and the layer-2 GEMM, and eleven small high-priority layout kernels
        # beside them cost more than they save: conv_ar 4.202 against 4.186 ms, attention 4.674 against 4.625, interleaved runs on one box)
        if st is None:
            st = self._ahead = {"conv": set(), "head": False, "swap": []}
        off = self.model._offset
        updated = [l for l in range(1, self.n) if lo <= off[f"encoder.layers.{l}.weight"] < hi]
        for l in updated:
            busy = (not final) and l == min(updated)
            self._prepare_conv(l, True, True, alt=busy)
            st["conv"].add(l)
            if busy:
                st["swap"].append(l)
        if lo <= off["prediction_model.weight"] < hi:
            self._prepare_head_weights()
            st["head"] = True
        if final:
            for l in st["swap"]:
                self.w_dgrad[l], self._w_dgrad_alt[l] = self._w_dgrad_alt[l], self.w_dgrad[l]
            complete = st["head"] and st["conv"] == set(range(1, self.n))
            self._ahead = None
            self._ahead_token = self._param_state() if complete else None

    # ------------------------------------------------------------------------------------------ forward
    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("the CPC hot path runs on the GPU only (no CPU fallback): move the batch to the device")
        if x.dtype != torch.float32 or tuple(x.shape) != (self.B, self.L) or not x.is_contiguous():
            raise ValueError(f"expected a contiguous float32 batch of shape ({self.B}, {self.L}), got {tuple(x.shape)} {x.dtype}")

    # ---- the rows only the TARGETS need, beside the GRU recurrence -------------------------------------------------------------
    # The context network reads the first V of the V + K top-layer frames; frames V .. V+K-1 are the targets (audio_model.py:197-198).
    # With unpadded causal convolutions top rows [0, V) need rows [0, n_l) of layer l, n_{l-1} = s_l (n_l - 1) + k_l — a prefix of every
    # layer.  The GRU occupies 16 of the 256 CUs for 0.27 ms (100 dependent steps): the remaining rows of every layer (11 % of the encoder's
    # forward work at V = 100, K = 12) are computed on the side stream while it runs, and the main stream's forward launches are
    # that much shorter.  Same kernels on row ranges, identical results (CPC_TARGET_LANE=0: one launch per layer).
    def _target_lane_rows(self):
        """n_l per layer (rows of layer l that the context network's frames depend on), or None when the split does not apply."""
        if getattr(self, "_tl_rows", 0) != 0:
            return self._tl_rows
        self._tl_rows = None
        if (os.environ.get("CPC_TARGET_LANE", "1") == "0" or not self.use_aux or not isinstance(self.ctx, GRUContext) or self.K <= 0
                or self.V <= 0 or self.n < 2 or self.T != self.V + self.K):
            return None
        La = self.geo.alloc
        n = [0] * self.n
        n[-1] = self.V
        for l in range(self.n - 1, 0, -1):
            n[l - 1] = self.strides[l] * (n[l] - 1) + self.kernels[l]
        if any(n[l] >= La[l] or n[l] <= 0 for l in range(self.n)):
            return None
        # (worth it only where the side lane is small and the main lane still fills the chip)
        if n[-1] * 4 < La[-1] * 3 or self.B * n[-1] < 4096:
            return None
        self._tl_rows = n
        self._tl_ev = (torch.cuda.Event(), torch.cuda.Event())
        return n

    def _row_launch_tile(self, M, N):
        """CPC_GEMM_BIG_TILE for a row-range launch that is ONE nearly full round of 256 x 256 tiles (the launcher's own rule takes the
        256-wide tile from 200 tiles on): layer 2's target rows at the headline size are 196 such tiles — 0.11 ms in one round against
        0.135 ms as 784 128-wide tiles on 512 slots."""
        tiles = _ceil_div(M, 256) * _ceil_div(N, 256)
        lo = int(os.environ.get("CPC_ROW_BIG_TILE_MIN", "160"))
        return _hip.GEMM_BIG_TILE if (self.dt == torch.bfloat16 and lo <= tiles < 200 and N % 256 == 0) else 0

    def _encoder_rows(self, x, lo, hi):
        """Layers 1 .. n on rows [lo[l], hi[l]) of every item (hi[l] = L_alloc[l]: to the end, pad rows included)."""
        p, code, B, La, Lv = self.model._param, self.code, self.B, self.geo.alloc, self.geo.valid
        _hip.call("cpc_conv1_fwd_rows", _hip.ptr(x, self.x_off), _hip.ptr(p["encoder.layers.0.weight"]), _hip.ptr(p.get("encoder.layers.0.bias")),
                  _hip.ptr(self.act[0]), B, self.channels[0], self.strides[0], self.kernels[0], self.L, Lv[0], La[0],
                  1 if self.n > 1 else 0, code, _hip.ptr(self.act_bits[0]), lo[0], hi[0],
                  key="cpc_conv1_fwd" if lo[0] == 0 else "cpc_conv1_fwd_target_rows")
        for l in range(1, self.n):
            cin, cout, kw, s = self.channels[l - 1], self.channels[l], self.kernels[l], self.strides[l]
            r0, rows = lo[l], hi[l] - lo[l]
            M = B * rows
            _hip.gemm_nt(_hip.ptr(self.act[l - 1], r0 * s * cin), _hip.ptr(self.w_fwd[l]), _hip.ptr(self.act[l], r0 * cout), M, cout, kw * cin,
                         s * cin, kw * cin, cout, code, bias=_hip.ptr(p.get(f"encoder.layers.{l}.bias")),
                         a_rpi=rows, a_item=La[l - 1] * cin, c_rpi=rows, c_item=La[l] * cout, c_valid=max(0, min(Lv[l], hi[l]) - r0),
                         flags=(_hip.GEMM_RELU if l < self.n - 1 else 0) | self._row_launch_tile(M, cout))

    def encoder_forward(self, x):
        """AudioEncoder.forward (audio_model.py:36-44): relu(conv) x (n-1), then a bare conv."""
        self._check_input(x)
        p, code, B, La, Lv = self.model._param, self.code, self.B, self.geo.alloc, self.geo.valid
        # (only inside forward(), which joins the lanes again; not under a hipGraph capture, which runs on one stream)
        nrows = self._target_lane_rows() if (self._tl_in_forward and self.use_aux) else None
        if nrows is not None:
            self._encoder_rows(x, [0] * self.n, nrows)                       # what the context network needs: main stream
            ev0, ev1 = self._tl_ev
            ev0.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.aux):
                self.aux.wait_event(ev0)
                self._encoder_rows(x, nrows, list(La))                      # what only the targets need: side stream, beside the GRU
                ev1.record(self.aux)
            self._tl_pending = ev1
            return
        _hip.call("cpc_conv1_fwd", _hip.ptr(x, self.x_off), _hip.ptr(p["encoder.layers.0.weight"]), _hip.ptr(p.get("encoder.layers.0.bias")),
                  _hip.ptr(self.act[0]), B, self.channels[0], self.strides[0], self.kernels[0], self.L, Lv[0], La[0],
                  1 if self.n > 1 else 0, code, _hip.ptr(self.act_bits[0]))
        for l in range(1, self.n):
            _hip.call("cpc_conv_fwd", _hip.ptr(self.act[l - 1]), _hip.ptr(self.w_fwd[l]), _hip.ptr(p.get(f"encoder.layers.{l}.bias")),
                      _hip.ptr(self.act[l]), B, self.channels[l - 1], self.channels[l], self.kernels[l], self.strides[l],
                      La[l], Lv[l], 1 if l < self.n - 1 else 0, C.c_longlong(self.guard[l - 1]), code,
                      key="gemm_nt" + _hip._variant(code, 0, _hip.nt_tile(code, B * La[l], self.channels[l], self.kernels[l] * self.channels[l - 1])),
                      work=2.0 * B * La[l] * self.channels[l] * self.kernels[l] * self.channels[l - 1],
                      shape=("fwd", B * La[l], self.channels[l], self.kernels[l] * self.channels[l - 1]))

    def context_forward(self):
        """c = autoregressive_model(z) over z = frames [T-K-V, T-K) (audio_model.py:198-204), then prediction_model (:208)."""
        code, B, H, E, K = self.code, self.B, self.H, self.E, self.K
        self.ctx.forward()
        ct, coff, cstride = self.ctx.c_operand()
        _hip.gemm_nt(_hip.ptr(ct, coff), _hip.ptr(self.w_p), _hip.ptr(self.pred), B, K * E, H, H, H, K * E, code,
                     a_rpi=1, a_item=cstride)
        ev = getattr(self, "_tl_pending", None)
        if ev is not None:          # the target rows of the encoder (side stream, see encoder_forward) are complete from here on
            torch.cuda.current_stream().wait_event(ev)
            self._tl_pending = None

    def forward(self, x):
        self.prepare_weights()
        self._nbt_batch = []          # BatchNorm `num_batches_tracked` counters of this pass: ONE multi-tensor add instead of a launch each
        self._tl_in_forward = True
        try:
            self.encoder_forward(x)
            self.context_forward()
        finally:
            self._tl_in_forward = False
            batch, self._nbt_batch = self._nbt_batch, None
            if batch:
                torch._foreach_add_(batch, 1)

    _nbt_batch = None
    _tl_in_forward = False

    def count_batch(self, counter):
        """`num_batches_tracked += 1` of a train-mode BatchNorm (twelve of them in a configs[2] step, each a 5 us launch on the main queue):
        collected while forward() runs and added with one launch at its end; outside forward() (stand-alone calls) added at once."""
        if self._nbt_batch is None:
            counter += 1
        else:
            self._nbt_batch.append(counter)

    # views of the forward results in the reference's shapes (storage dtype, no copies)
    def view_top(self):
        return self.act[-1].view(self.B, self.geo.alloc[-1], self.E)

    def outputs(self):
        """(predicted_z (B,K,E), targets (B,E,K), z (B,E,V), c (B,H)) as float32 tensors in the reference's shapes."""
        top = self.view_top()
        T, K, V = self.T, self.K, self.V
        pred = self.pred.view(self.B, K, self.E).float()
        targets = top[:, T - K:T, :].float().transpose(1, 2)
        z = top[:, T - K - V:T - K, :].float().transpose(1, 2)
        zs = getattr(self.ctx, "z_scale", 1.0)       # AttentionModel rescales z in place (attention_model.py:30)
        if zs != 1.0:
            z = z * zs
        return pred, targets, z, self.ctx.c_float().clone()

    # ------------------------------------------------------------------------------------------ loss
    def score_gemm(self, pred=None, top=None, out=None):
        """The score contraction of the default branch (contrastive_estimation_training.py:12-22 restricted to equal steps,
        :116): K batched B x E x B products S[k] = predicted_z[:, k, :] targets[:, :, k]^T.  Returns its algorithmic FLOPs."""
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T, ld = self.geo.alloc[-1], self.T, self.ldS
        pred = self.pred if pred is None else pred
        top = self.act[-1] if top is None else top
        _hip.gemm_nt(_hip.ptr(pred), _hip.ptr(top, (T - K) * E), _hip.ptr(self.S if out is None else out), B, B, E, K * E, Ltop * E, ld,
                     code, a_batch=E, b_batch=E, c_batch=B * ld, batch=K, flags=_hip.GEMM_OUT_F32)
        return 2.0 * K * B * B * E

    def score_gemm_all(self, pred=None, top=None, out=None):
        """The full (B K) x E x (B K) score contraction of score_over_all_timesteps=True (:12-22, :108-114).  Returns its FLOPs."""
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T = self.geo.alloc[-1], self.T
        R = B * K
        ld = _ceil_div(R, 8) * 8
        if getattr(self, "S_all", None) is None:
            self.S_all = torch.zeros(R * ld, device=self.device, dtype=torch.float32)
        pred = self.pred if pred is None else pred
        top = self.act[-1] if top is None else top
        _hip.gemm_nt(_hip.ptr(pred), _hip.ptr(top, (T - K) * E), _hip.ptr(self.S_all if out is None else out), R, R, E, E, E, ld, code,
                     b_rpi=K, b_item=Ltop * E, flags=_hip.GEMM_OUT_F32)
        return 2.0 * R * R * E

    def nce_forward_backward(self, softplus: bool, regularization: float):
        """Equal-step scores, InfoNCE loss + regulariser, and d loss / d (predicted_z, targets).

        contrastive_estimation_training.py:106-122,141 with score_over_all_timesteps=False.  Only the K diagonal
        (B x B) blocks of the reference's (B K)^2 score tensor are ever formed (12x fewer FLOPs)."""
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T, ld = self.geo.alloc[-1], self.T, self.ldS
        top, dtop = self.act[-1], self.dact[-1]
        self.score_gemm()
        _hip.call("cpc_nce_loss", _hip.ptr(self.S), _hip.ptr(self.dS), _hip.ptr(self.dST), _hip.ptr(self.nce_out),
                  _hip.ptr(self.nce_ws), B, K, ld, 1 if softplus else 0, C.c_float(regularization), code)
        self._score_grads(self.dS, self.dST, self.pred, top, self.dpred, dtop)

    def _score_grads(self, W, WT, pred, top, out_pred, out_top):
        """The two contractions behind d loss / d (predicted_z, targets) of the default branch, for any coefficients W[k][b][b']
        (WT: its transpose per k) and operands in the layouts of ``self.pred`` / the top-layer buffer."""
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T, ld = self.geo.alloc[-1], self.T, self.ldS
        # out_pred[b][k][:] = sum_b' W[k][b][b'] * targets[b'][k][:]
        _hip.gemm_tn(_hip.ptr(WT), _hip.ptr(top, (T - K) * E), _hip.ptr(out_pred), B, B, E, ld, Ltop * E, K * E, code,
                     a_batch=B * ld, b_batch=E, c_batch=E, batch=K)
        # out_top rows T-K+k of item b' = sum_b W[k][b][b'] * predicted_z[b][k][:]
        _hip.gemm_tn(_hip.ptr(W), _hip.ptr(pred), _hip.ptr(out_top, (T - K) * E), B, B, E, ld, K * E, Ltop * E, code,
                     a_batch=B * ld, b_batch=E, c_batch=E, batch=K)

    # ---- score_over_all_timesteps=True with the column pass fused into the score GEMM (bf16; include/cpc_hip.h, cpc_score_lse) ----
    def fused_scores_ok(self, rows=None, cols=None):
        """The fused path needs bf16 storage, 256-row / 256-column tiles, E a multiple of 64 and an even K <= 24
        (CPC_FUSED_SCORE=0: the unfused kernels, A/B)."""
        R = self.B * self.K
        rows, cols = (R if rows is None else rows), (R if cols is None else cols)
        return (self.dt == torch.bfloat16 and os.environ.get("CPC_FUSED_SCORE", "1") != "0" and rows % 256 == 0 and cols % 256 == 0
                and self.E % 64 == 0 and self.E >= 128 and self.K % 2 == 0 and self.K <= 24)

    def _nce_all_fused(self, softplus: bool, regularization: float):
        """One persistent MFMA launch forms the (B K) x (B K) scores tile by tile and leaves the f32 scores + per-tile column
        (max, sum-exp) pairs + the diagonal; a merge, one gradient pass (dS and its transpose) and the loss scalars follow.  Replaces: the second (transposed) score GEMM, two f32 score matrices, the 16-way split column pass and the
        two element-wise gradient passes of nce_all_forward_backward."""
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T = self.geo.alloc[-1], self.T
        top, dtop = self.act[-1], self.dact[-1]
        R = B * K
        f = getattr(self, "_fs", None)
        if f is None:
            dev, f32 = self.device, torch.float32
            f = self._fs = SimpleNamespace(
                targ=torch.empty(R, E, device=dev, dtype=self.dt), Sb=torch.empty(R, R, device=dev, dtype=f32),
                pm=torch.empty(R // 256, R, device=dev, dtype=f32), ps=torch.empty(R // 256, R, device=dev, dtype=f32),
                valid=torch.zeros(R, device=dev, dtype=f32), lse=torch.empty(R, device=dev, dtype=f32),
                colp=torch.empty(_ceil_div(R, 256), 2, device=dev, dtype=f32),
                gradp=torch.empty(int(_hip.lib().cpc_nce_fused_grad_blocks(B, R)), device=dev, dtype=f32),
                dS=torch.empty(R, R, device=dev, dtype=self.dt), dST=torch.empty(R, R, device=dev, dtype=self.dt))
        P = _hip.ptr
        f.targ.view(B, K, E).copy_(top.view(B, Ltop, E)[:, T - K:T, :])
        _hip.call("cpc_score_lse", P(self.pred), P(f.targ), P(f.Sb), P(f.pm), P(f.ps), P(f.valid), R, R, E, C.c_longlong(E), C.c_longlong(E),
                  C.c_longlong(R), 0, key="score_lse<bf16,256>", work=2.0 * R * R * E)
        _hip.call("cpc_nce_lse_merge", P(f.pm), P(f.ps), R // 256, R, 1 if softplus else 0, C.c_float(R), P(f.lse), P(f.colp))
        _hip.call("cpc_nce_fused_grad", P(f.Sb), P(f.lse), P(f.dS), P(f.dST), P(f.gradp), B, K, R, C.c_longlong(R), C.c_longlong(R), 0,
                  1 if softplus else 0, C.c_float(regularization), C.c_float(R), C.c_float(B))
        _hip.call("cpc_nce_fused_finalize", P(f.colp), f.colp.shape[0], P(f.valid), R, P(f.gradp), f.gradp.numel(), None, 0, C.c_float(R),
                  C.c_float(B), K, C.c_float(regularization), 1 if softplus else 0, P(self.nce_out))
        self._score_grads_all(f.dS, f.dST, self.pred, top, self.dpred, dtop)

    def nce_all_forward_backward(self, softplus: bool, regularization: float):
        """score_over_all_timesteps=True (contrastive_estimation_training.py:108-114, :141): the full (B K) x (B K) score
        matrix (and its transpose, as a second tiny GEMM, so that both gradient layouts are written coalesced), the
        log-sum-exp over ALL predictions for every (target item, step), and d loss / d (predicted_z, targets).
        bf16 storage and tile-sized problems take the fused route (_nce_all_fused) instead."""
        if self.fused_scores_ok():
            return self._nce_all_fused(softplus, regularization)
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T = self.geo.alloc[-1], self.T
        top, dtop = self.act[-1], self.dact[-1]
        R = B * K
        ld = _ceil_div(R, 8) * 8
        if getattr(self, "ST_all", None) is None:
            f32 = torch.float32
            if getattr(self, "S_all", None) is None:
                self.S_all = torch.zeros(R * ld, device=self.device, dtype=f32)
            self.ST_all = torch.zeros(R * ld, device=self.device, dtype=f32)
            self.dS_all = torch.zeros(R * ld, device=self.device, dtype=self.dt)
            self.dST_all = torch.zeros(R * ld, device=self.device, dtype=self.dt)
            self.nce_all_ws = torch.empty(int(_hip.lib().cpc_nce_all_workspace_floats(B, K)), device=self.device, dtype=f32)
        tg = (T - K) * E
        self.score_gemm_all()
        _hip.gemm_nt(_hip.ptr(top, tg), _hip.ptr(self.pred), _hip.ptr(self.ST_all), R, R, E, E, E, ld, code,
                     a_rpi=K, a_item=Ltop * E, flags=_hip.GEMM_OUT_F32)
        _hip.call("cpc_nce_loss_all", _hip.ptr(self.S_all), _hip.ptr(self.ST_all), _hip.ptr(self.dS_all), _hip.ptr(self.dST_all),
                  _hip.ptr(self.nce_out), _hip.ptr(self.nce_all_ws), B, K, ld, 1 if softplus else 0, C.c_float(regularization), code)
        # d predicted_z[(b,k)][:] = sum_c dS[(b,k)][c] * targets[c][:]  and  d targets[c][:] = sum_r dS[r][c] * predicted_z[r][:]
        # (the latter into rows T-K+k' of item b' of the top-layer gradient).  As NT GEMMs over the long axis — the reduction
        # form would put a 3072-row reduction on 24 workgroups — with the small right-hand operands transposed once (3 MB each).
        self._score_grads_all(self.dS_all, self.dST_all, self.pred, top, self.dpred, dtop)

    def _score_grads_all(self, W, WT, pred, top, out_pred, out_top):
        """The same for the all-timesteps branch: W [(b,k)][(b',k')] and its transpose."""
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T = self.geo.alloc[-1], self.T
        R = B * K
        ld = _ceil_div(R, 8) * 8
        tg = (T - K) * E
        if getattr(self, "targT", None) is None:
            self.targT = torch.zeros(E, ld, device=self.device, dtype=self.dt)
            self.predT = torch.zeros(E, ld, device=self.device, dtype=self.dt)
        self.targT[:, :R].copy_(top.view(B, Ltop, E)[:, T - K:T, :].reshape(R, E).t())
        self.predT[:, :R].copy_(pred.view(R, E).t())
        _hip.gemm_nt(_hip.ptr(W), _hip.ptr(self.targT), _hip.ptr(out_pred), R, E, ld, ld, ld, E, code)
        _hip.gemm_nt(_hip.ptr(WT), _hip.ptr(self.predT), _hip.ptr(out_top, tg), R, E, ld, ld, ld, E, code,
                     c_rpi=K, c_item=Ltop * E, c_valid=K)

    def nce_eval(self, softplus: bool, all_timesteps: bool, sums, workspace):
        """Adds this batch's validation quantities (per-step losses, per-step accuracies, mean score: include/cpc_hip.h,
        cpc_nce_eval) to ``sums`` (2 K + 1 floats) — from the score matrices of the train step (forward() must have run)."""
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T = self.geo.alloc[-1], self.T
        top = self.act[-1]
        tg = (T - K) * E
        if all_timesteps:
            ld = _ceil_div(B * K, 8) * 8
            self.score_gemm_all()
            S = self.S_all
        else:
            ld = self.ldS
            self.score_gemm()
            S = self.S
        _hip.call("cpc_nce_eval", _hip.ptr(S), _hip.ptr(sums), _hip.ptr(workspace), B, K, ld, 1 if softplus else 0,
                  1 if all_timesteps else 0, 1)

    # ------------------------------------------------------------------------------------------ backward
    def _tn_to_grad(self, A, B_, grad, M, I, J, lda, ldb, nsplit, grad_offset=0, scratch=None, **kw):
        """grad[grad_offset + i*J + j] = sum_m A[m][i] B[m][j] via f32 slabs + deterministic reduction."""
        code = self.code
        chunk = self._chunk(M, nsplit)
        scratch = self.slabs if scratch is None else scratch
        _hip.gemm_tn(A, B_, _hip.ptr(scratch), M, I, J, lda, ldb, J, code, nsplit=nsplit, m_chunk=chunk, slab_stride=I * J,
                     flags=_hip.GEMM_OUT_F32, **kw)
        _hip.call("cpc_reduce_slabs", _hip.ptr(scratch), _hip.ptr(grad, grad_offset), I, J, nsplit, I * J, 1, 1, J, 0)

    def _colsum_to_grad(self, X, grad, M, N, code=None, scratch=None):
        nb = min(self.colsum_blocks, max(1, M // 64))
        scratch = self.slabs if scratch is None else scratch
        _hip.call("cpc_colsum", X, _hip.ptr(scratch), M, N, N, nb, self.code if code is None else code)
        _hip.call("cpc_reduce_slabs", _hip.ptr(scratch), _hip.ptr(grad), 1, N, nb, N, 1, 1, 0, 0)

    def backward(self, x, add_dc: Optional[torch.Tensor] = None, add_dz: Optional[torch.Tensor] = None, grad_ready_hook=None):
        """Gradients of everything upstream of (predicted_z, targets, z, c) into the model's flat gradient buffer.

        Expects ``dpred`` and rows [T-K, T) of the top-layer gradient to be filled (by nce_forward_backward or by the
        autograd bridge).  Rows [T-K-V, T-K) are overwritten here with the GRU's input gradient (+ add_dz (B,E,V)).
        ``grad_ready_hook(lo, hi)`` is called as soon as the flat-gradient range [lo, hi) is final, so that a caller can
        start reducing it across ranks while the remaining layers are still being differentiated."""
        g, code = self.model._grad, self.code
        B, V, H, E, K, n = self.B, self.V, self.H, self.E, self.K, self.n
        La, Lv = self.geo.alloc, self.geo.valid
        Ltop, T = La[-1], self.T
        t0 = T - K - V
        self._ahead = None          # prepare_ahead state of a step whose optimizer step never came (nothing was swapped in yet)
        top, dtop = self.act[-1], self.dact[-1]
        # predictor: dW_p = dpred^T c ;  dc = dpred W_p.  Only dc is on the path to the context network's backward pass: the weight
        # gradient is issued later, inside the first side-stream block of the encoder's backward pass (no event of its own)
        ct, coff, cstride = self.ctx.c_operand()

        def dwp_call():
            _hip.gemm_tn(_hip.ptr(self.dpred), _hip.ptr(ct, coff), _hip.ptr(g["prediction_model.weight"]), B, K * E, H,
                         K * E, H, H, code, b_rpi=1, b_item=cstride, flags=_hip.GEMM_OUT_F32)
        if self.use_aux and n > 1:
            self._deferred_side = [dwp_call]
        else:
            dwp_call()
        # (a B x H output with a K*E-long reduction: split the reduction over workgroups, sum the slabs in fixed order)
        ke = K * E
        ksplit = ke // 256 if (ke % 256 == 0 and ke >= 1024 and self.slabs.numel() >= (ke // 256) * B * H) else 1
        if ksplit > 1:
            _hip.gemm_nt(_hip.ptr(self.dpred), _hip.ptr(self.w_p_t), _hip.ptr(self.slabs), B, H, 256, ke, ke, H, code,
                         a_batch=256, b_batch=256, c_batch=B * H, batch=ksplit, flags=_hip.GEMM_OUT_F32)
            _hip.call("cpc_reduce_slabs", _hip.ptr(self.slabs), _hip.ptr(self.dc), B, H, ksplit, B * H, 1, 1, H, 0)
        else:
            _hip.gemm_nt(_hip.ptr(self.dpred), _hip.ptr(self.w_p_t), _hip.ptr(self.dc), B, H, ke, ke, ke, H, code,
                         flags=_hip.GEMM_OUT_F32)
        if add_dc is not None:
            self.dc.add_(add_dc)
        lanes_ok = self.use_aux and getattr(self, "fuse_c1", False) and os.environ.get("CPC_WGRAD_STREAM", "1") != "0" and getattr(self, "_gp_phase", 0) == 0
        self._bl_active = self._bwd_lane() if lanes_ok else None
        if self._bl_active is not None:
            self._backward_target_rows(x, self._bl_active)
        self.ctx.backward(self.dc)      # parameter gradients of the context network + dz into rows [t0, t0+V) of dtop
        if add_dz is not None:
            dtop.view(B, Ltop, E)[:, t0:t0 + V, :].add_(add_dz.transpose(1, 2), alpha=getattr(self.ctx, "z_scale", 1.0))
        self._backward_encoder(x, grad_ready_hook)
        for fn in getattr(self, "_deferred_side", ()):           # an encoder backward without that side block (scalogram engine)
            fn()
        self._deferred_side = ()
        if self.use_aux:
            torch.cuda.current_stream().wait_stream(self.aux)

    def _bwd_lane(self):
        """The row split of the encoder's backward pass that mirrors encoder_forward's target lane, or None.  With kernel = 2 stride in
        every layer above the first, the data gradient at positions >= n_(l-1) of layer l-1 depends on output-gradient rows >= n_l only
        (n_l as in _target_lane_rows): GEMM rows [n_l + 1, L_alloc) of every data gradient — the part of the backward pass behind the TARGET
        frames, whose top-layer gradient is final once the loss kernels are — run on the side stream beside the GRU's backward recurrence
        (_backward_target_rows); the main stream's launches after the recurrence cover rows [0, n_l + 1).  The weight gradients stay
        one launch per layer: split at row n_l the same way (two slab sets, one reduction) they cost the step 0.37 ms — the chip is full
        once the recurrence has ended, and the second slab set is pure extra traffic (round 4, CPC_TARGET_LANE_BWD A/B: 4.52 off, 4.49 on,
        4.89 with the weight gradients split).  Tried on top of this and removed again: the recurrence itself in two step
        ranges with the encoder rows of the later frames beside the first (forward) / the earlier steps beside the encoder's backward pass
        over the late rows (backward) — bit-identical, and 0.0 - 0.1 ms SLOWER: what the hidden recurrence saves, the split launches lose."""
        bl = getattr(self, "_bl", 0)
        if bl != 0:
            return bl
        self._bl = None
        nr = self._target_lane_rows()
        n, B, La = self.n, self.B, self.geo.alloc
        if (nr is None or os.environ.get("CPC_TARGET_LANE_BWD", "1") == "0" or self.dt != torch.bfloat16 or not self.fuse_c1
                or type(self)._backward_encoder is not CPCEngine._backward_encoder
                or any(self.kernels[l] != 2 * self.strides[l] for l in range(1, n))
                or any(B * (La[l] - nr[l] - 1) < 256 for l in range(1, n))):
            return None
        dev, f32 = self.device, torch.float32
        bl = SimpleNamespace(n=nr, ev0=torch.cuda.Event(), ev=[torch.cuda.Event() for _ in range(n)], cs=[None] * n, tiles=[None] * n)
        for l in range(1, n):
            cin, cout, kw, s = self.channels[l - 1], self.channels[l], self.kernels[l], self.strides[l]
            bl.tiles[l] = (_ceil_div(B * (nr[l] + 1), 256), _ceil_div(B * (La[l] - nr[l] - 1), 256))
            if l >= 2 and self.cs_slabs[l - 1] is not None:        # (per-tile column sums: where the one-launch path has them)
                bl.cs[l - 1] = torch.zeros(sum(bl.tiles[l]) * s * cin, device=dev, dtype=f32)
        bl.c1_tile = (self.strides[1] * self.channels[0] // 256) * (self.kernels[0] + 1) * 256          # slab floats of one 256-row tile
        bl.c1 = torch.empty(sum(bl.tiles[1]) * bl.c1_tile, device=dev, dtype=f32)
        self._bl = bl
        return bl

    def _lane_dgrad(self, bl, l, part, x):
        """Data gradient of layer l: GEMM rows [0, n_l + 1) (part 0) or [n_l + 1, L_alloc) (part 1); layer 2's is fused with layer 1's
        weight gradient (cpc_conv_dgrad_conv1_rows)."""
        B, code, La, Lv = self.B, self.code, self.geo.alloc, self.geo.valid
        cin, cout, kw, s = self.channels[l - 1], self.channels[l], self.kernels[l], self.strides[l]
        lo, hi = (0, bl.n[l] + 1) if part == 0 else (bl.n[l] + 1, La[l])
        M = B * (hi - lo)
        tile = 256 if (l == 1 or bl.cs[l - 1] is not None) else _hip.nt_tile(code, M, s * cin, self.geo.taps[l] * cout)
        # (the side stream's launches, a fraction of a round of tiles each beside the recurrence, are booked under their own key)
        tkey = dict(key="gemm_nt" + _hip._variant(code, 0, tile) + ("@target_rows" if part else ""), work=2.0 * M * s * cin * self.geo.taps[l] * cout,
                    shape=("dgrad", M, s * cin, self.geo.taps[l] * cout))
        if l == 1:
            _hip.call("cpc_conv_dgrad_conv1_rows", _hip.ptr(self.dact[1]), _hip.ptr(self.w_dgrad[1]), _hip.ptr(self.act[0]),
                      _hip.ptr(x, self.x_off), _hip.ptr(bl.c1, part * bl.tiles[1][0] * bl.c1_tile), B, cin, cout, kw, s, La[1], self.L,
                      self.kernels[0], self.strides[0], Lv[0], C.c_longlong(self.guard[1]), code, _hip.ptr(self.act_bits[0]), lo, hi,
                      **dict(tkey, key=tkey["key"].replace("gemm_nt", "gemm_nt_conv1")))
        else:
            _hip.call("cpc_conv_dgrad_rows", _hip.ptr(self.dact[l]), _hip.ptr(self.w_dgrad[l]), _hip.ptr(self.act[l - 1]),
                      _hip.ptr(self.dact[l - 1]), B, cin, cout, kw, s, La[l], C.c_longlong(self.guard[l]), code,
                      _hip.ptr(self.act_bits[l - 1]), _hip.ptr(bl.cs[l - 1], part * bl.tiles[l][0] * s * cin), lo, hi, **tkey)

    def _backward_target_rows(self, x, bl):
        """Side stream, issued before the context network's backward pass: everything of the encoder's backward pass that hangs on the
        target frames alone (see _bwd_lane): the data-gradient chain; the main stream's launches wait for its events layer by layer."""
        bl.ev0.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.aux):
            self.aux.wait_event(bl.ev0)
            for l in range(self.n - 1, 0, -1):
                self._lane_dgrad(bl, l, 1, x)
                bl.ev[l].record(self.aux)

    def _backward_encoder(self, x, grad_ready_hook=None):
        """Encoder part of the backward pass: consumes the top-layer gradient, fills the encoder's parameter gradients."""
        g, code = self.model._grad, self.code
        B, n = self.B, self.n
        La, Lv = self.geo.alloc, self.geo.valid
        bl, self._bl_active = getattr(self, "_bl_active", None), None      # target lane (_bwd_lane): this pass covers rows [0, n_l + 1) only
        cs_slabs = self.cs_slabs if bl is None else bl.cs
        # encoder, top layer down to layer 2.  Main stream: weight-gradient GEMM, data-gradient GEMM.  Side stream, after the
        # weight-gradient GEMM: the bias column sum (needs dact[l]) and the slab reduction, see _alloc_encoder.
        for l in range(n - 1, 0, -1):
            cin, cout, kw, s = self.channels[l - 1], self.channels[l], self.kernels[l], self.strides[l]
            bname = f"encoder.layers.{l}.bias"
            flops = 2.0 * B * La[l] * cout * kw * cin
            wg_mode = os.environ.get("CPC_WGRAD_STREAM", "1")

            # The weight-gradient GEMM of layer l and the data-gradient GEMM of layer l both read dact[l] and are independent: the
            # weight gradient is issued on the side stream, beside the data-gradient chain (the tiles of the two kernels interleave on
            # the CUs, the partial last rounds of tiles of one launch are filled by the other, and both read dact[l] while it is in the
            # caches): 4.61 -> 4.55 ms per step at the end of round 2 (4.72 -> 4.69 before the side stream was cleared of the
            # column-sum passes).  Every kernel then runs longer BY ITSELF: bench.py reports the dominant kernel's roofline both as
            # it runs in the step and alone (CPC_WGRAD_STREAM=0: everything on the main stream, the round-1 arrangement).
            def wgrad_call():
                _hip.call("cpc_conv_wgrad", _hip.ptr(self.act[l - 1]), _hip.ptr(self.dact[l]), _hip.ptr(self.wslab[l]), B, cin, cout, kw, s,
                          La[l], self.nsplit[l], C.c_longlong(self.guard[l - 1]), code,
                          key="gemm_tn" + _hip._variant(code, _hip.GEMM_OUT_F32, _hip.tn_tile(code, B * La[l], kw * cin, cout, self.nsplit[l],
                                                                                                 self._chunk(B * La[l], self.nsplit[l]))),
                          work=flops,
                          shape=("wgrad", B * La[l], kw * cin, cout, self.nsplit[l]))
            if wg_mode == "0":
                wgrad_call()
            with self.side(self._ev_w[l]):
                for fn in getattr(self, "_deferred_side", ()):       # work backward() put off the main stream (see there)
                    fn()
                self._deferred_side = ()
                if wg_mode != "0":
                    wgrad_call()
                # (ONE event per layer on the main stream: a recorded event between two GEMMs costs ~6 us of idle queue)
                if bname in g:
                    if cs_slabs[l] is not None:           # per-tile sums left by layer l+1's data gradient
                        _hip.call("cpc_reduce_slabs", _hip.ptr(cs_slabs[l]), _hip.ptr(g[bname]), 1, cout, cs_slabs[l].numel() // cout,
                                  cout, 1, 1, 0, 0)
                    else:
                        self._colsum_to_grad(_hip.ptr(self.dact[l]), g[bname], B * La[l], cout, scratch=self.aux_slabs)
                _hip.call("cpc_reduce_conv_w", _hip.ptr(self.wslab[l]), _hip.ptr(g[f"encoder.layers.{l}.weight"]), cin, cout, kw,
                          self.nsplit[l], kw * cin * cout)
                if grad_ready_hook is not None and l <= 2:
                    # everything from this encoder layer upwards (+ context network, predictor: later in the flat buffer) is final
                    # once the side stream gets here (its wait on _ev_w covers all earlier main-stream work): layers >= 3 and the
                    # head travel under the backward of layers 2 and 1, layer 2 (index 1) under the last data-gradient GEMM
                    off = self.model._offset
                    hi = off[f"encoder.layers.{l + 1}.weight"] if (l == 1 and n > 2) else self.model._flat_grad.numel()
                    grad_ready_hook(off[f"encoder.layers.{l}.weight"], hi)
            tkey = dict(key="gemm_nt" + _hip._variant(code, 0, _hip.nt_tile(code, B * La[l], s * cin, self.geo.taps[l] * cout)),
                        work=2.0 * B * La[l] * s * cin * self.geo.taps[l] * cout,
                        shape=("dgrad", B * La[l], s * cin, self.geo.taps[l] * cout))
            if bl is not None:
                if l < n - 1:
                    torch.cuda.current_stream().wait_event(bl.ev[l + 1])     # rows >= n_l of dact[l]: the side stream's data gradient of layer l+1
                self._lane_dgrad(bl, l, 0, x)
            elif l == 1 and self.fuse_c1:
                _hip.call("cpc_conv_dgrad_conv1", _hip.ptr(self.dact[1]), _hip.ptr(self.w_dgrad[1]), _hip.ptr(self.act[0]),
                          _hip.ptr(x, self.x_off), _hip.ptr(self.c1_slabs), B, cin, cout, kw, s, La[1], self.L, self.kernels[0],
                          self.strides[0], Lv[0], C.c_longlong(self.guard[1]), code, _hip.ptr(self.act_bits[0]),
                          **dict(tkey, key=tkey["key"].replace("gemm_nt", "gemm_nt_conv1")))
            else:
                _hip.call("cpc_conv_dgrad", _hip.ptr(self.dact[l]), _hip.ptr(self.w_dgrad[l]), _hip.ptr(self.act[l - 1]),
                          _hip.ptr(self.dact[l - 1]), B, cin, cout, kw, s, La[l], Lv[l - 1], C.c_longlong(self.guard[l]), code,
                          _hip.ptr(self.act_bits[l - 1]), _hip.ptr(self.cs_slabs[l - 1]), **tkey)
        # layer 1
        c0, k0, s0 = self.channels[0], self.kernels[0], self.strides[0]
        if bl is not None:
            _hip.call("cpc_conv1_fused_reduce_tiles", _hip.ptr(bl.c1), _hip.ptr(self.c1_tmp), _hip.ptr(g["encoder.layers.0.weight"]),
                      _hip.ptr(g.get("encoder.layers.0.bias")), sum(bl.tiles[1]), c0, self.strides[1], k0)
            return
        if self.fuse_c1:
            _hip.call("cpc_conv1_fused_reduce", _hip.ptr(self.c1_slabs), _hip.ptr(self.c1_tmp), _hip.ptr(g["encoder.layers.0.weight"]),
                      _hip.ptr(g.get("encoder.layers.0.bias")), B, c0, self.strides[1], La[1], k0)
            return
        nblk, nbb = self.c1_blocks, self.c1_item_blocks
        _hip.call("cpc_conv1_bwd", _hip.ptr(x, self.x_off), _hip.ptr(self.dact[0]), _hip.ptr(self.slabs), B, c0, s0, k0, self.L, Lv[0], La[0],
                  nblk, nbb, code)
        stride = (k0 + 1) * c0
        _hip.call("cpc_reduce_slabs", _hip.ptr(self.slabs), _hip.ptr(g["encoder.layers.0.weight"]), k0, c0, nbb * nblk, stride, 1, k0, 1, 0)
        if "encoder.layers.0.bias" in g:
            _hip.call("cpc_reduce_slabs", _hip.ptr(self.slabs, k0 * c0), _hip.ptr(g["encoder.layers.0.bias"]), 1, c0, nbb * nblk, stride,
                      1, 1, 0, 0)

    def input_gradient(self):
        """d / d x (B, L) float32 of whatever _backward_encoder() has just differentiated — the transposed convolution of layer 1
        (audio_model.py:38: nn.Conv1d(1, C, k, s), its autograd with respect to the input), for stand-alone encoder calls whose input
        requires grad.  Needs layer 1's output gradient in memory (the unfused route: dact[0]).  With D = ceil(k / s) taps,
          dx[b][t s + r] = sum_{d < D} sum_c dY[b][t - (D - 1 - d)][c] * w[c][r + (D - 1 - d) s]          (r < s; taps beyond k are zero)
        is ONE overlapped-row GEMM over dact[0]: row (b, t) = the D rows ending at t, K = D C, N = s (padded to 8).  Samples in front of
        x_off (frames the model never consumes, not computed at all) and behind the last window get zero."""
        B, C0, k0, s0, D = self.B, self.channels[0], self.kernels[0], self.strides[0], self.geo.taps[0]
        La0, dev = self.geo.alloc[0], self.device
        w = self.model._param["encoder.layers.0.weight"].detach().float().view(C0, k0)
        Bt = torch.zeros(8, D, C0, device=dev, dtype=torch.float32)
        for d in range(D):
            j0 = (D - 1 - d) * s0
            n = max(0, min(s0, k0 - j0))
            if n > 0:
                Bt[:n, d, :] = w[:, j0:j0 + n].t()
        Bt = Bt.view(8, D * C0).to(self.dt).contiguous()
        out = torch.empty(B * La0, 8, device=dev, dtype=torch.float32)
        if s0 > 8 or self.guard[0] < (D - 1) * C0:
            raise NotImplementedError("input gradient: layer-1 stride <= 8")
        _hip.gemm_nt(_hip.ptr(self.dact[0], -(D - 1) * C0), _hip.ptr(Bt), _hip.ptr(out), B * La0, 8, D * C0, C0, D * C0, 8, self.code,
                     flags=_hip.GEMM_OUT_F32)
        dx = torch.zeros(B, self.L, device=dev, dtype=torch.float32)
        n = min(self.L_eff, La0 * s0)
        dx[:, self.x_off:self.x_off + n] = out.view(B, La0, 8)[:, :, :s0].reshape(B, La0 * s0)[:, :n]
        return dx

    # ------------------------------------------------------------------------------------------ whole step
    def loss_and_grads(self, x, softplus: bool, regularization: float, all_timesteps: bool = False, grad_ready_hook=None,
                       global_negatives=None, after_loss=None):
        """Forward + loss + backward; returns the device tensor [loss, max_score, -mean valid, mean lse, reg, NaN indicator of
        this step, sticky NaN flag, -] (no sync; include/cpc_hip.h, cpc_nce_loss).
        ``global_negatives``: a GlobalNegatives object — the loss is then taken over the batches of ALL ranks.
        ``after_loss(nce_out)`` is called once the loss kernels are queued and before the backward pass is: data-parallel runs
        start the reduction of the NaN flag over the ranks there (GradAllReduce.reduce_flag)."""
        self.forward(x)
        if global_negatives is not None:
            global_negatives.forward_backward(softplus, regularization, all_timesteps)
        elif all_timesteps:
            self.nce_all_forward_backward(softplus, regularization)
        else:
            self.nce_forward_backward(softplus, regularization)
        if after_loss is not None:
            after_loss(self.nce_out)
        self.backward(x, grad_ready_hook=grad_ready_hook)
        return self.nce_out

    def nan_flag(self):
        """One-element view of the sticky NaN flag (nce_out[6]): FusedAdam.skip_flag; zero it when a run starts."""
        return self.nce_out[6:7]


class GRUContext:
    """AudioGRUModel as the context network (audio_model.py:47-77): one batched input-projection GEMM for all V steps, the
    persistent GRU kernels for the recurrence, GEMMs for the weight gradients and for dz."""

    ahead_ok = True       # prepare_weights is a pure function of the parameters (CPCEngine.prepare_ahead)

    def __init__(self, eng, ar):
        self.eng = eng
        self.ar = ar
        self.H = int(ar.hidden_size)
        ch = 8 if eng.dt == torch.bfloat16 else 4
        if ar.input_size != eng.E or self.H != eng.H:
            raise ValueError("AudioGRUModel sizes do not match enc_size / ar_size")
        if self.H % 16 or self.H % (4 * ch) or self.H > 256:
            raise NotImplementedError("HIP GRU kernel: hidden size must be a multiple of 32 and <= 256")
        self.prefix = "autoregressive_model.gruCell."

    def allocate(self):
        e = self.eng
        B, V, H, E, dev, dt = e.B, e.V, self.H, e.E, e.device, e.dt
        self.w_ih = torch.empty(3 * H * E, device=dev, dtype=dt)          # [3H][E]
        self.w_ih_t = torch.empty(E * 3 * H, device=dev, dtype=dt)        # [E][3H]
        self.w_hh_frag = torch.empty(3 * H * H, device=dev, dtype=dt)
        self.w_hh_t_frag = torch.empty(3 * H * H, device=dev, dtype=dt)
        self.Gi = torch.empty(B * V * 3 * H, device=dev, dtype=dt)
        self.Hall = torch.empty(B * (V + 1) * H, device=dev, dtype=dt)
        self.tape = torch.zeros(max(1, int(_hip.lib().cpc_gru_tape_elems(B, max(V, 1), H, e.code))), device=dev, dtype=dt)
        self.dG = torch.empty(B * V * 4 * H, device=dev, dtype=dt)       # [dr | du | dn | dn*r] per (item, step)
        self.split_ih = e._pick_split(3 * H, E, B * V)
        self.split_hh = e._pick_split(2 * H, H, B * V)
        self.scratch = torch.empty(self.slab_floats(), device=dev, dtype=torch.float32)     # side-stream workspace
        self._ev = torch.cuda.Event()

    def slab_floats(self):
        e, H = self.eng, self.H
        return max(self.split_ih * 3 * H * e.E, self.split_hh * 2 * H * H, e.colsum_blocks * 4 * H)

    def prepare_weights(self):
        e, H, E, code = self.eng, self.H, self.eng.E, self.eng.code
        p = e.model._param
        w_ih, w_hh = p[self.prefix + "weight_ih"], p[self.prefix + "weight_hh"]
        _hip.call("cpc_cast2d", _hip.ptr(w_ih), _hip.ptr(self.w_ih), 3 * H, E, E, 1, code)
        _hip.call("cpc_cast2d", _hip.ptr(w_ih), _hip.ptr(self.w_ih_t), E, 3 * H, 1, E, code)
        _hip.call("cpc_prep_frag", _hip.ptr(w_hh), _hip.ptr(self.w_hh_frag), 3 * H, H, H, 0, code)
        _hip.call("cpc_prep_frag", _hip.ptr(w_hh), _hip.ptr(self.w_hh_t_frag), H, 3 * H, H, 1, code)

    def forward(self):
        e, H = self.eng, self.H
        p, code, B, V, E, K = e.model._param, e.code, e.B, e.V, e.E, e.K
        Ltop, t0, top = e.geo.alloc[-1], e.T - K - V, e.act[-1]
        _hip.gemm_nt(_hip.ptr(top, t0 * E), _hip.ptr(self.w_ih), _hip.ptr(self.Gi), B * V, 3 * H, E, E, E, 3 * H, code,
                     bias=_hip.ptr(p.get(self.prefix + "bias_ih")), a_rpi=V, a_item=Ltop * E)
        # reset_hidden=False (audio_model.py:69, :75): the last hidden state of the previous call is this call's initial one
        carry = not getattr(self.ar, "reset_hidden", True)
        h0 = self.ar.hidden if carry else None
        if h0 is not None and (tuple(h0.shape) != (B, H) or h0.device != e.c.device):
            raise ValueError(f"carried GRU state has shape {tuple(h0.shape)} on {h0.device}, this call needs ({B}, {H}) on {e.c.device}")
        self.carried = h0 is not None
        if h0 is not None:
            _hip.call("cpc_gru_fwd_h0", _hip.ptr(self.Gi), _hip.ptr(self.w_hh_frag), _hip.ptr(p.get(self.prefix + "bias_hh")),
                      _hip.ptr(h0.detach().float().contiguous()), _hip.ptr(self.Hall), _hip.ptr(self.tape), _hip.ptr(e.c), B, V, H, code)
        else:
            _hip.call("cpc_gru_fwd", _hip.ptr(self.Gi), _hip.ptr(self.w_hh_frag), _hip.ptr(p.get(self.prefix + "bias_hh")),
                      _hip.ptr(self.Hall), _hip.ptr(self.tape), _hip.ptr(e.c), B, V, H, code)
        self.ar.hidden = e.c.detach().clone() if carry else None

    def c_operand(self):
        """(tensor, element offset, item stride): storage-dtype rows of c, one per item (h_V inside the hidden-state buffer)."""
        return self.Hall, self.eng.V * self.H, (self.eng.V + 1) * self.H

    def c_float(self):
        return self.eng.c

    def backward(self, dc):
        e, H = self.eng, self.H
        g, code, B, V, E, K = e.model._grad, e.code, e.B, e.V, e.E, e.K
        Ltop, t0, top, dtop = e.geo.alloc[-1], e.T - K - V, e.act[-1], e.dact[-1]
        phase = getattr(e, "_gp_phase", 0)
        if getattr(self, "carried", False):
            raise RuntimeError("this GRU call started from a state carried over from the previous call (reset_hidden=False): it cannot be "
                               "differentiated -- the reference's autograd cannot either (the previous call's graph is gone after its "
                               "backward()); use reset_hidden=False for inference / streaming, or set model.hidden = None before a train step")
        if phase == 1:               # gradient penalty, pass 1: dc is the adjoint of the summed scores (gp_grads starts from it)
            self._gp_buffers()
            self.gp_dc1.copy_(dc)
        _hip.call("cpc_gru_bwd", _hip.ptr(dc), _hip.ptr(self.tape), _hip.ptr(self.w_hh_t_frag), _hip.ptr(self.dG), B, V, H, code)
        # the GRU's parameter gradients (ten short launches) are off the critical path dG -> dz -> encoder backward: side stream
        with e.side(self._ev):
            self.backward_params()
        # dz -> rows [t0, t0+V) of the top-layer gradient
        _hip.gemm_nt(_hip.ptr(self.dG), _hip.ptr(self.w_ih_t), _hip.ptr(dtop, t0 * E), B * V, E, 3 * H, 4 * H, 3 * H, E, code,
                     c_rpi=V, c_item=Ltop * E, c_valid=V)
        if phase == 3:               # gradient penalty, pass 3: what z gains through the tangent recurrence's coefficients
            dtop.view(B, Ltop, E)[:, t0:t0 + V, :].add_(self.gp_dz.view(B, V, E))

    def backward_params(self):
        """Parameter gradients of the recurrence from dG (all steps), on the current stream."""
        e, H = self.eng, self.H
        g, code, B, V, E, K = e.model._grad, e.code, e.B, e.V, e.E, e.K
        Ltop, t0, top = e.geo.alloc[-1], e.T - K - V, e.act[-1]
        # dG[b][t] = [dr | du | dn | dn*r]: columns [0,3H) are the gradient of the input-projection term, columns [0,2H) and
        # [3H,4H) that of the recurrent term
        g_ih, g_hh = g[self.prefix + "weight_ih"], g[self.prefix + "weight_hh"]
        sl = self.scratch
        e._tn_to_grad(_hip.ptr(self.dG), _hip.ptr(top, t0 * E), g_ih, B * V, 3 * H, E, 4 * H, E, self.split_ih,
                      b_rpi=V, b_item=Ltop * E, scratch=sl)
        e._tn_to_grad(_hip.ptr(self.dG), _hip.ptr(self.Hall), g_hh, B * V, 2 * H, H, 4 * H, H, self.split_hh,
                      b_rpi=V, b_item=(V + 1) * H, scratch=sl)
        e._tn_to_grad(_hip.ptr(self.dG, 3 * H), _hip.ptr(self.Hall), g_hh, B * V, H, H, 4 * H, H, self.split_hh,
                      b_rpi=V, b_item=(V + 1) * H, grad_offset=2 * H * H, scratch=sl)
        if (self.prefix + "bias_ih") in g:
            g_bi, g_bh = g[self.prefix + "bias_ih"], g[self.prefix + "bias_hh"]
            M = B * V
            nb = min(e.colsum_blocks, max(1, M // 64))
            _hip.call("cpc_colsum", _hip.ptr(self.dG), _hip.ptr(sl), M, 4 * H, 4 * H, nb, code)
            _hip.call("cpc_reduce_slabs", _hip.ptr(sl), _hip.ptr(g_bi), 1, 3 * H, nb, 4 * H, 1, 1, 0, 0)
            _hip.call("cpc_reduce_slabs", _hip.ptr(sl), _hip.ptr(g_bh), 1, 2 * H, nb, 4 * H, 1, 1, 0, 0)
            _hip.call("cpc_reduce_slabs", _hip.ptr(sl, 3 * H), _hip.ptr(g_bh, 2 * H), 1, H, nb, 4 * H, 1, 1, 0, 0)

    # ---- Wasserstein gradient penalty (scalogram_engine._gp_step; exact-f32 mode): tangent of c, penalty parts of the gradients
    def _gp_buffers(self):
        if getattr(self, "gp_tape", None) is not None:
            return
        e, H = self.eng, self.H
        B, V, E, dev = e.B, e.V, e.E, e.device
        if e.dt != torch.float32:
            raise NotImplementedError("the gradient penalty runs in the exact-f32 mode (compute_dtype='fp32')")
        f32 = dict(device=dev, dtype=torch.float32)
        self.gp_dc1 = torch.empty(B, H, **f32)
        self.gp_GiT = torch.empty(B * V * 3 * H, **f32)
        self.gp_whh_t = torch.empty(H * 3 * H, **f32)                 # [H][3H]
        self.gp_tape = torch.empty(B * V * 10 * H, **f32)
        self.gp_ct = torch.empty(B, H, **f32)
        self.gp_dA = torch.empty(B * V * 8 * H, **f32)
        self.gp_dz = torch.empty(B * V * E, **f32)
        self.gp_scratch = torch.empty(2 * self.slab_floats(), **f32)

    def tangent(self, top_t):
        """``top_t``: tangent of the encoder's top buffer; returns (tensor, offset, item stride) of the tangent of c.  Runs the
        primal and the tangent recurrence together (cpc_gru_gp_fwd) and keeps what gp_grads' reverse sweep reads."""
        e, H = self.eng, self.H
        p, code, B, V, E, K = e.model._param, e.code, e.B, e.V, e.E, e.K
        Ltop, t0 = e.geo.alloc[-1], e.T - K - V
        self._gp_buffers()
        _hip.gemm_nt(_hip.ptr(top_t, t0 * E), _hip.ptr(self.w_ih), _hip.ptr(self.gp_GiT), B * V, 3 * H, E, E, E, 3 * H, code,
                     a_rpi=V, a_item=Ltop * E)
        _hip.call("cpc_cast2d", _hip.ptr(p[self.prefix + "weight_hh"]), _hip.ptr(self.gp_whh_t), H, 3 * H, 1, H, code)
        _hip.call("cpc_gru_gp_fwd", _hip.ptr(self.Gi), _hip.ptr(self.gp_GiT), _hip.ptr(self.gp_whh_t),
                  _hip.ptr(p[self.prefix + "bias_hh"]), _hip.ptr(self.gp_tape), _hip.ptr(self.gp_ct), B, V, H)
        self._gp_top_t = top_t
        return self.gp_ct, 0, H

    def gp_grads(self, gp_grad):
        """Penalty parts of the GRU's parameter gradients and of dz.  With dA = [delta | nu] from cpc_gru_gp_bwd (delta: adjoints of
        the summed scores, nu: second-order adjoints): dW_ih = delta^T (tangent x) + nu^T x, dW_hh = delta'^T (tangent h) + nu'^T h,
        the biases take the column sums of nu, z takes nu W_ih (added to the top-layer gradient in pass 3)."""
        e, H = self.eng, self.H
        p, code, B, V, E, K = e.model._param, e.code, e.B, e.V, e.E, e.K
        Ltop, t0, top, top_t = e.geo.alloc[-1], e.T - K - V, e.act[-1], self._gp_top_t
        if (self.prefix + "bias_hh") not in p:
            raise NotImplementedError("gradient penalty through a GRU without biases")
        _hip.call("cpc_gru_gp_bwd", _hip.ptr(self.gp_dc1), _hip.ptr(self.gp_tape), _hip.ptr(p[self.prefix + "weight_hh"]),
                  _hip.ptr(self.gp_dA), B, V, H)
        dA, tape, sl = self.gp_dA, self.gp_tape, self.gp_scratch
        M = B * V

        def pair(grad, I, J, nsplit, first, second, grad_offset=0):
            # grad = first product + second product: both into f32 slabs, one fixed-order reduction over all of them
            chunk = e._chunk(M, nsplit)
            for idx, (a_off, b_ptr, ldb, kw) in enumerate((first, second)):
                _hip.gemm_tn(_hip.ptr(dA, a_off), b_ptr, _hip.ptr(sl, idx * nsplit * I * J), M, I, J, 8 * H, ldb, J, code, nsplit=nsplit,
                             m_chunk=chunk, slab_stride=I * J, flags=_hip.GEMM_OUT_F32, **kw)
            _hip.call("cpc_reduce_slabs", _hip.ptr(sl), _hip.ptr(grad, grad_offset), I, J, 2 * nsplit, I * J, 1, 1, J, 0)

        rows = dict(b_rpi=V, b_item=Ltop * E)
        g_ih, g_hh = gp_grad[self.prefix + "weight_ih"], gp_grad[self.prefix + "weight_hh"]
        pair(g_ih, 3 * H, E, self.split_ih, (0, _hip.ptr(top_t, t0 * E), E, rows), (4 * H, _hip.ptr(top, t0 * E), E, rows))
        ht_prev, h_prev = _hip.ptr(tape, 9 * H), _hip.ptr(tape, 4 * H)
        pair(g_hh, 2 * H, H, self.split_hh, (0, ht_prev, 10 * H, {}), (4 * H, h_prev, 10 * H, {}))
        pair(g_hh, H, H, self.split_hh, (3 * H, ht_prev, 10 * H, {}), (7 * H, h_prev, 10 * H, {}), grad_offset=2 * H * H)
        nb = min(e.colsum_blocks, max(1, M // 64))
        _hip.call("cpc_colsum", _hip.ptr(dA), _hip.ptr(sl), M, 8 * H, 8 * H, nb, code)
        g_bi, g_bh = gp_grad[self.prefix + "bias_ih"], gp_grad[self.prefix + "bias_hh"]
        _hip.call("cpc_reduce_slabs", _hip.ptr(sl, 4 * H), _hip.ptr(g_bi), 1, 3 * H, nb, 8 * H, 1, 1, 0, 0)
        _hip.call("cpc_reduce_slabs", _hip.ptr(sl, 4 * H), _hip.ptr(g_bh), 1, 2 * H, nb, 8 * H, 1, 1, 0, 0)
        _hip.call("cpc_reduce_slabs", _hip.ptr(sl, 7 * H), _hip.ptr(g_bh, 2 * H), 1, H, nb, 8 * H, 1, 1, 0, 0)
        _hip.gemm_nt(_hip.ptr(dA, 4 * H), _hip.ptr(self.w_ih_t), _hip.ptr(self.gp_dz), M, E, 3 * H, 8 * H, 3 * H, E, code)


class ConvArContext:
    """ConvolutionalArModel as the context network (audio_model.py:80-161) for the plain configuration (no batch norm, no
    residual branch; e.g. ``ar_conv_default_dict``): per block [MaxPool1d(pool, ceil)] -> Conv1d(k, stride 1) -> ReLU on
    channels-last buffers; the convolutions are the same overlapped-row GEMMs as the encoder's; c = the last position."""

    ahead_ok = True       # prepare_weights is a pure function of the parameters (CPCEngine.prepare_ahead)

    def __init__(self, eng, ar):
        self.eng = eng
        self.kernels = list(ar.kernel_sizes)
        self.strides = list(ar.strides)
        self.pools = list(ar.poolings)
        self.channels = list(ar.channel_count)
        self.nb = len(self.kernels)
        if ar.batch_norm or ar.residual:
            raise NotImplementedError("ConvArContext is the lean path for plain configurations; make_context() routes batch_norm / residual to ConvArGridContext")
        if any(s != 1 for s in self.strides):
            raise NotImplementedError("ConvArContext handles stride-1 convolutions; make_context() routes strided ones to ConvArGridContext")
        if self.channels[0] != eng.E or self.channels[-1] != eng.H:
            raise ValueError("ConvolutionalArModel channel_count does not match enc_size / ar_size")
        ch = 8 if eng.dt == torch.bfloat16 else 4
        if any(c % 8 for c in self.channels) or any((k * c) % ch for k, c in zip(self.kernels, self.channels)):
            raise NotImplementedError("ConvolutionalArModel channel counts must be multiples of 8")
        # lengths: pooled input and conv output of every block
        self.lp, self.lo = [], []
        cur = eng.V
        for l in range(self.nb):
            cur = _ceil_div(cur, self.pools[l])
            self.lp.append(cur)
            cur = cur - self.kernels[l] + 1
            if cur <= 0:
                raise ValueError("visible_steps too short for this ConvolutionalArModel")
            self.lo.append(cur)
        # padded row counts per item: the conv input buffer of block l and its output share one value (stride 1) and keep
        # >= kernel-1 zero pad rows, which is what the overlapped-row data-gradient GEMM needs between items
        self.la = [max(self.lp[l], self.lo[l]) + self.kernels[l] for l in range(self.nb)]
        self.conv_idx = [1 if self.pools[l] > 1 else 0 for l in range(self.nb)]

    def _name(self, l, what):
        return f"autoregressive_model.module_list.{l}.main_modules.{self.conv_idx[l]}.{what}"

    def allocate(self):
        e = self.eng
        B, dev, dt = e.B, e.device, e.dt
        self.x, self.dx, self.y, self.dy, self.w_fwd, self.w_dgrad, self.nsplit = [], [], [], [], [], [], []
        self._keep = []
        for l in range(self.nb):
            cin, cout, kw = self.channels[l], self.channels[l + 1], self.kernels[l]
            for store, cols in ((self.x, cin), (self.dx, cin), (self.y, cout), (self.dy, cout)):
                if l == 0 and self.pools[0] == 1 and store in (self.x, self.dx):
                    store.append(None)        # block 0 without pooling reads z straight out of the encoder's top buffer
                    continue
                full, view, _ = e._buf(B * self.la[l], cols, max(self.kernels))
                self._keep.append(full)
                store.append(view)
            self.w_fwd.append(torch.empty(cout * kw * cin, device=dev, dtype=dt))
            self.w_dgrad.append(torch.empty(cin * kw * cout, device=dev, dtype=dt))
            self.nsplit.append(e._pick_split(kw * cin, cout, B * self.la[l]))
        self.c32 = torch.empty(B, self.channels[-1], device=dev, dtype=torch.float32)
        # side-stream reductions: per-layer weight-gradient slabs, one bias scratch, events
        self.wslab = [torch.empty(self.nsplit[l] * self.kernels[l] * self.channels[l] * self.channels[l + 1], device=dev, dtype=torch.float32)
                      for l in range(self.nb)]
        self.bscratch = torch.empty(e.colsum_blocks * max(self.channels), device=dev, dtype=torch.float32)
        self._ev = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(self.nb)]

    def slab_floats(self):
        return max(self.nsplit[l] * self.kernels[l] * self.channels[l] * self.channels[l + 1] for l in range(self.nb)) + \
            self.eng.colsum_blocks * max(self.channels)

    def prepare_weights(self):
        e = self.eng
        p, code = e.model._param, e.code
        for l in range(self.nb):
            _hip.call("cpc_conv_w_prep", _hip.ptr(p[self._name(l, "weight")]), _hip.ptr(self.w_fwd[l]), _hip.ptr(self.w_dgrad[l]),
                      self.channels[l + 1], self.channels[l], self.kernels[l], 1, code)

    def _block_input(self, l):
        """(pointer to the conv input rows of block l, a_rpi, a_item) — block 0 without pooling reads the encoder output."""
        e = self.eng
        if self.x[l] is None:
            t0 = e.T - e.K - e.V
            return _hip.ptr(e.act[-1], t0 * e.E), self.la[0], e.geo.alloc[-1] * e.E
        return _hip.ptr(self.x[l]), 0, 0

    def forward(self):
        e = self.eng
        p, code, B = e.model._param, e.code, e.B
        for l in range(self.nb):
            cin, cout, kw, la = self.channels[l], self.channels[l + 1], self.kernels[l], self.la[l]
            if self.pools[l] > 1:
                if l == 0:
                    t0 = e.T - e.K - e.V
                    raise NotImplementedError("ConvArContext: no pooling in the first block (make_context() routes that to ConvArGridContext)")
                _hip.call("cpc_maxpool_fwd", _hip.ptr(self.y[l - 1]), _hip.ptr(self.x[l]), B, cin, self.pools[l], self.lo[l - 1],
                          self.la[l - 1], self.lp[l], la, code)
            elif l > 0:
                raise NotImplementedError("ConvArContext: pooling in every later block (make_context() routes other layouts to ConvArGridContext)")
            a, a_rpi, a_item = self._block_input(l)
            _hip.gemm_nt(a, _hip.ptr(self.w_fwd[l]), _hip.ptr(self.y[l]), B * la, cout, kw * cin, cin, kw * cin, cout, code,
                         bias=_hip.ptr(p.get(self._name(l, "bias"))), a_rpi=a_rpi, a_item=a_item, c_rpi=la, c_item=la * cout,
                         c_valid=self.lo[l], flags=_hip.GEMM_RELU)

    def c_operand(self):
        l = self.nb - 1
        return self.y[l], (self.lo[l] - 1) * self.channels[-1], self.la[l] * self.channels[-1]

    def c_float(self):
        l = self.nb - 1
        H = self.channels[-1]
        self.c32.copy_(self.y[l].view(self.eng.B, self.la[l], H)[:, self.lo[l] - 1, :])
        return self.c32

    def backward(self, dc):
        e = self.eng
        g, code, B = e.model._grad, e.code, e.B
        last = self.nb - 1
        H = self.channels[-1]
        # gradient enters at the single position forward() returns; everything else of dy[last] stays zero
        _hip.call("cpc_relu_row_bwd", _hip.ptr(dc), _hip.ptr(self.y[last]), _hip.ptr(self.dy[last]), B, H, self.la[last] * H,
                  (self.lo[last] - 1) * H, code)
        for l in range(last, -1, -1):
            cin, cout, kw, la = self.channels[l], self.channels[l + 1], self.kernels[l], self.la[l]
            bname = self._name(l, "bias")
            if bname in g:
                with e.side(self._ev[l][0]):
                    e._colsum_to_grad(_hip.ptr(self.dy[l]), g[bname], B * la, cout, scratch=self.bscratch)
            a, a_rpi, a_item = self._block_input(l)
            chunk = e._chunk(B * la, self.nsplit[l])
            # The weight-gradient GEMM stays on the main stream (on the high-priority side stream it takes the CUs from the short
            # data-gradient chain it runs beside: measured 3.9 -> 5.3 ms per step); only the short reductions go to the side stream.
            _hip.gemm_tn(a, _hip.ptr(self.dy[l]), _hip.ptr(self.wslab[l]), B * la, kw * cin, cout, cin, cout, cout, code, a_rpi=a_rpi,
                         a_item=a_item, nsplit=self.nsplit[l], m_chunk=chunk, slab_stride=kw * cin * cout, flags=_hip.GEMM_OUT_F32)
            with e.side(self._ev[l][1]):
                _hip.call("cpc_reduce_conv_w", _hip.ptr(self.wslab[l]), _hip.ptr(g[self._name(l, "weight")]), cin, cout, kw,
                          self.nsplit[l], kw * cin * cout)
            D = kw                                   # stride 1: every input position is touched by kw output rows
            dy_shift = _hip.ptr(self.dy[l], -(D - 1) * cout)
            if self.x[l] is None:
                # data gradient straight into rows [t0, t0+V) of the encoder's top-layer gradient (no ReLU on that layer)
                t0 = e.T - e.K - e.V
                _hip.gemm_nt(dy_shift, _hip.ptr(self.w_dgrad[l]), _hip.ptr(e.dact[-1], t0 * e.E), B * la, cin, D * cout, cout,
                             D * cout, cin, code, c_rpi=la, c_item=e.geo.alloc[-1] * e.E, c_valid=e.V, flags=_hip.GEMM_SKIP_PAD_ROWS)
            else:
                _hip.gemm_nt(dy_shift, _hip.ptr(self.w_dgrad[l]), _hip.ptr(self.dx[l]), B * la, cin, D * cout, cout, D * cout, cin,
                             code, mask=_hip.ptr(self.x[l]), c_rpi=la, c_item=la * cin, c_valid=self.lp[l])
                _hip.call("cpc_maxpool_bwd", _hip.ptr(self.y[l - 1]), _hip.ptr(self.dx[l]), _hip.ptr(self.dy[l - 1]), B, cin,
                          self.pools[l], self.lo[l - 1], self.la[l - 1], la, code)


class AttentionContext:
    """AttentionModel as the context network (attention_model.py:38-82): positional encoding, ``num_layers`` post-norm
    transformer encoder layers with a causal mask (transformer.py:262-271), a final LayerNorm, the mean over time and
    ``end_layer``.  Rows are (item, step) with C channels; every Linear is a cpc_gemm_nt / cpc_gemm_tn call, the
    per-(item, head) attention, the residual + LayerNorm and the mean are the kernels of csrc/attn.hip.
    Dropout (train mode, p > 0) uses counter-based masks that the backward regenerates (include/cpc_hip.h, cpc_dropout):
    same distribution as the reference's nn.Dropout, not its random stream."""

    LN_EPS = 1e-5
    SITE_ATTN, SITE_DROP1, SITE_FF, SITE_DROP2 = 0, 1, 2, 3       # dropout sites of layer l: 4*l + ...

    def __init__(self, eng, ar):
        self.eng = eng
        self.C, self.N, self.heads, self.FF = ar.channels, ar.num_layers, ar.num_heads, ar.feedforward_size
        self.out = ar.output_size
        self.S = eng.V
        self.ar = ar
        if self.C != eng.E or self.out != eng.H:
            raise ValueError("AttentionModel channels / output_size do not match enc_size / ar_size")
        if self.S > ar.sequence_length:
            raise ValueError("visible_steps exceeds the AttentionModel's sequence_length (positional table)")
        if self.S > 64 or self.C % self.heads or self.C // self.heads > 64:
            raise NotImplementedError("HIP attention kernel: visible_steps <= 64 and head size <= 64")
        ch = 8 if eng.dt == torch.bfloat16 else 4
        if self.C % 8 or self.FF % 8 or self.out % ch or self.C > 2048:
            raise NotImplementedError("AttentionModel sizes must be multiples of 8 (channels <= 2048)")
        self.z_scale = math.sqrt(self.C)
        self.prefix = "autoregressive_model."
        self.drop_p, self.drop_seed, self._drop_counter, self._l2_scale = 0.0, 0, 0, 1.0
        self.fixed_seed = None          # tests pin the mask seed here

    def _lname(self, l, what):
        return f"{self.prefix}encoder.layers.{l}.{what}"

    def allocate(self):
        e = self.eng
        B, dev, dt, f32 = e.B, e.device, e.dt, torch.float32
        C, FF, S, N, H = self.C, self.FF, self.S, self.N, self.out
        M = B * S
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=dt)
        self.pe = self.ar.positional_encoder.pe[:S, 0, :].detach().to(device=dev, dtype=f32).contiguous()
        self.X = [new(M * C) for _ in range(N + 1)]
        self.qkv = [new(M * 3 * C) for _ in range(N)]
        self.P = [new(B * self.heads * S * S) for _ in range(N)]
        self.att = [new(M * C) for _ in range(N)]
        self.r1 = [new(M * C) for _ in range(N)]
        self.x1 = [new(M * C) for _ in range(N)]
        self.f1 = [new(M * FF) for _ in range(N)]
        self.r2 = [new(M * C) for _ in range(N)]
        self.st1 = [torch.empty(M * 2, device=dev, dtype=f32) for _ in range(N)]
        self.st2 = [torch.empty(M * 2, device=dev, dtype=f32) for _ in range(N)]
        self.stn = torch.empty(M * 2, device=dev, dtype=f32)
        self.ytmp, self.xn = new(M * C), new(M * C)
        self.mean, self.dmean = new(B * C), new(B * C)
        self.c32 = torch.empty(B, H, device=dev, dtype=f32)
        self.ct = self.c32 if dt == f32 else new(B * H)
        self.dct = new(B * H)
        # gradient scratch
        # Gradient buffers that the parameter-gradient kernels read exist twice (layer parity): those kernels run on the side
        # stream while the main stream already differentiates the next layer down, which writes the other set.
        self.gA, self.gB = [new(M * C), new(M * C)], [new(M * C), new(M * C)]
        self.gC, self.gD = new(M * C), new(M * C)
        self.gAd, self.gBd = [new(M * C), new(M * C)], [new(M * C), new(M * C)]   # dropped-out summands (dropout only)
        self.dqkv, self.df1, self.datt = [new(M * 3 * C), new(M * 3 * C)], [new(M * FF), new(M * FF)], new(M * C)
        self.scratch = None
        self._cast_jobs = None
        self._ev = [[torch.cuda.Event() for _ in range(5)] for _ in range(self.N + 1)]
        # weight operands in the storage dtype: [out][in] for the forward GEMMs, [in][out] for the data gradients
        shapes = {"in": (3 * C, C), "o": (C, C), "l1": (FF, C), "l2": (C, FF)}
        self.w = [{k: new(r * c) for k, (r, c) in shapes.items()} for _ in range(N)]
        self.wt = [{k: new(r * c) for k, (r, c) in shapes.items()} for _ in range(N)]
        self.w_end, self.w_end_t = new(H * C), new(C * H)
        # LayerNorm backward: a wave walks its rows one at a time (load -> two wave reductions -> store), so the row loop is latency-
        # bound; 512 workgroups = 2 waves per SIMD overlap the rows (attention_architecture_1 at B = 256: 5.60 ms per step with 256 workgroups, 5.45 with 512 or 1024, 5.57 with 2048)
        self.ln_blocks = max(1, min(int(os.environ.get("CPC_LN_BLOCKS", "512")), M // 4))
        self.split = {k: e._pick_split(r, c, M) for k, (r, c) in shapes.items()}

    def slab_floats(self):
        C, FF = self.C, self.FF
        shapes = {"in": 3 * C * C, "o": C * C, "l1": FF * C, "l2": C * FF}
        return max([self.split[k] * n for k, n in shapes.items()] + [self.ln_blocks * 2 * C,
                                                                     self.eng.colsum_blocks * max(3 * C, FF)])

    _WNAMES = {"in": "self_attn.in_proj_weight", "o": "self_attn.out_proj.weight", "l1": "linear1.weight", "l2": "linear2.weight"}
    _BNAMES = {"in": "self_attn.in_proj_bias", "o": "self_attn.out_proj.bias", "l1": "linear1.bias", "l2": "linear2.bias"}

    def prepare_weights(self):
        e = self.eng
        p, code, C, FF, H = e.model._param, e.code, self.C, self.FF, self.out
        shapes = {"in": (3 * C, C), "o": (C, C), "l1": (FF, C), "l2": (C, FF)}
        if self._cast_jobs is None or self._cast_jobs[1] is not e.model._flat_param:
            # all operand-layout copies (weight and transposed weight of every Linear) as one batched launch; the job table holds
            # raw addresses, which are stable: parameters are views of the model's flat buffer
            jobs = []
            for l in range(self.N):
                for k, (r, c) in shapes.items():
                    src = p[self._lname(l, self._WNAMES[k])].data_ptr()
                    jobs.append((src, self.w[l][k].data_ptr(), r, c, c, 1))
                    jobs.append((src, self.wt[l][k].data_ptr(), c, r, 1, c))
            src = p[self.prefix + "end_layer.weight"].data_ptr()
            jobs.append((src, self.w_end.data_ptr(), H, C, C, 1))
            jobs.append((src, self.w_end_t.data_ptr(), C, H, 1, C))
            self._cast_jobs = (torch.tensor(jobs, dtype=torch.int64, device=e.device), e.model._flat_param, len(jobs))
        _hip.call("cpc_cast2d_batch", _hip.ptr(self._cast_jobs[0]), self._cast_jobs[2], code)
        self._l2_scale = 1.0

    ahead_ok = True       # prepare_weights is a pure function of the parameters (CPCEngine.prepare_ahead); the dropout state is drawn in forward()

    def _begin_forward(self):
        """Dropout state of this forward pass (train mode, p > 0): a fresh mask seed, and the 1 / (1 - p) of the feed-forward dropout folded
        into the operand of the feed-forward data-gradient GEMM (d relu(.) dropout = keep / (1 - p): the keep part comes with the ReLU
        mask of the stored, dropped activation)."""
        dp = float(self.ar.dropout) if (self.ar.training and self.ar.dropout > 0.0) else 0.0
        if dp > 0.0:
            if not dp < 1.0:
                raise ValueError("dropout probability must be < 1")
            self._drop_counter += 1
            base = torch.initial_seed() if self.fixed_seed is None else int(self.fixed_seed)
            self.drop_seed = (base * 0x9E3779B1 + (0 if self.fixed_seed is not None else self._drop_counter)) & 0x7FFFFFFFFFFFFFFF
        want = 1.0 / (1.0 - dp) if dp > 0.0 else 1.0
        if want != self._l2_scale:
            if self._l2_scale != 1.0:          # (train <-> eval without a parameter update in between: rebuild rather than rescale twice)
                self.prepare_weights()
            if want != 1.0:
                for l in range(self.N):
                    self.wt[l]["l2"].mul_(want)
            self._l2_scale = want
        self.drop_p = dp

    def _ln(self, a, b, wname, r_out, y, stats, site=None):
        p = self.eng.model._param
        dp = self.drop_p if site is not None else 0.0
        _hip.call("cpc_add_ln_fwd", _hip.ptr(a), _hip.ptr(b), _hip.ptr(p[wname + ".weight"]), _hip.ptr(p[wname + ".bias"]),
                  _hip.ptr(r_out), _hip.ptr(y), _hip.ptr(stats), self.eng.B * self.S, self.C, self.LN_EPS, dp, self.drop_seed,
                  site or 0, self.eng.code)

    def forward(self):
        e = self.eng
        p, code, B = e.model._param, e.code, e.B
        C, FF, S, H = self.C, self.FF, self.S, self.out
        M = B * S
        Ltop, t0 = e.geo.alloc[-1], e.T - e.K - e.V
        P = _hip.ptr
        self._begin_forward()
        dp, seed = self.drop_p, self.drop_seed
        _hip.call("cpc_pe_scale_fwd", P(e.act[-1], t0 * C), P(self.pe), P(self.X[0]), B, S, C, Ltop * C, self.z_scale, code)
        for l in range(self.N):
            X, w = self.X[l], self.w[l]
            bias = {k: P(p[self._lname(l, n)]) for k, n in self._BNAMES.items()}
            _hip.gemm_nt(P(X), P(w["in"]), P(self.qkv[l]), M, 3 * C, C, C, C, 3 * C, code, bias=bias["in"])
            _hip.call("cpc_attn_fwd", P(self.qkv[l]), P(self.att[l]), P(self.P[l]), B, S, C, self.heads, dp, seed, 4 * l + self.SITE_ATTN,
                      code)
            _hip.gemm_nt(P(self.att[l]), P(w["o"]), P(self.ytmp), M, C, C, C, C, C, code, bias=bias["o"])
            self._ln(X, self.ytmp, self._lname(l, "norm1"), self.r1[l], self.x1[l], self.st1[l], site=4 * l + self.SITE_DROP1)
            _hip.gemm_nt(P(self.x1[l]), P(w["l1"]), P(self.f1[l]), M, FF, C, C, C, FF, code, bias=bias["l1"], flags=_hip.GEMM_RELU)
            if dp > 0.0:
                _hip.call("cpc_dropout", P(self.f1[l]), M * FF, dp, seed, 4 * l + self.SITE_FF, code)
            _hip.gemm_nt(P(self.f1[l]), P(w["l2"]), P(self.ytmp), M, C, FF, FF, FF, C, code, bias=bias["l2"])
            self._ln(self.x1[l], self.ytmp, self._lname(l, "norm2"), self.r2[l], self.X[l + 1], self.st2[l], site=4 * l + self.SITE_DROP2)
        self._ln(self.X[self.N], None, self.prefix + "encoder.norm", None, self.xn, self.stn)
        _hip.call("cpc_mean_time", P(self.xn), P(self.mean), B, S, C, code)
        b_end = P(p[self.prefix + "end_layer.bias"])
        _hip.gemm_nt(P(self.mean), P(self.w_end), P(self.c32), B, H, C, C, C, H, code, bias=b_end, flags=_hip.GEMM_OUT_F32)
        if self.ct is not self.c32:
            _hip.gemm_nt(P(self.mean), P(self.w_end), P(self.ct), B, H, C, C, C, H, code, bias=b_end)

    def c_operand(self):
        return self.ct, 0, self.out

    def c_float(self):
        return self.c32

    def _ln_bwd(self, g1, g2, r, stats, wname, dr, bcast=0, gscale=1.0, dr_b=None, site=0):
        e, C = self.eng, self.C
        g, p = e.model._grad, e.model._param
        nb = self.ln_blocks
        _hip.call("cpc_ln_bwd", _hip.ptr(g1), _hip.ptr(g2), _hip.ptr(r), _hip.ptr(stats), _hip.ptr(p[wname + ".weight"]),
                  _hip.ptr(dr), _hip.ptr(e.slabs), e.B * self.S, C, bcast, gscale, nb, _hip.ptr(dr_b),
                  self.drop_p if dr_b is not None else 0.0, self.drop_seed, site, e.code)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs), _hip.ptr(g[wname + ".weight"]), 1, C, nb, 2 * C, 1, 1, 0, 0)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs, C), _hip.ptr(g[wname + ".bias"]), 1, C, nb, 2 * C, 1, 1, 0, 0)

    # ---- Wasserstein gradient penalty (scalogram_engine._gp_step; exact-f32 mode).  The context is not piecewise linear, so the
    # penalty's gradient is the reverse sweep of the JOINT (primal, tangent) program: tangent() runs the tangent pass and keeps the
    # tangent of every Linear's input; the last pass (_backward_gp) carries two adjoints down the layers — lambda, the real loss's
    # (the ordinary backward, which also takes the second-order terms of LayerNorm and of the attention core on the way:
    # cpc_ln_gp, cpc_attn_gp), and delta, the summed scores' (pass 1 again, seeded with the dc kept from it), whose products with the
    # tangent inputs are the penalty parts of the weight gradients.  tools/gp_attention_algebra.py checks the algebra on the CPU.
    def _gp_buffers(self):
        if getattr(self, "gp", None) is not None:
            return
        e = self.eng
        if e.dt != torch.float32:
            raise NotImplementedError("the gradient penalty runs in the exact-f32 mode (compute_dtype='fp32')")
        B, C, FF, S, N, H = e.B, self.C, self.FF, self.S, self.N, self.out
        M = B * S
        new = lambda *shape: torch.empty(*shape, device=e.device, dtype=torch.float32)
        self.gp = SimpleNamespace(
            dc1=new(B, H), zero_pe=torch.zeros(S * C, device=e.device, dtype=torch.float32),
            Xt=[new(M * C) for _ in range(N + 1)], qkvt=[new(M * 3 * C) for _ in range(N)], attt=[new(M * C) for _ in range(N)],
            r1t=[new(M * C) for _ in range(N)], x1t=[new(M * C) for _ in range(N)], f1t=[new(M * FF) for _ in range(N)],
            r2t=[new(M * C) for _ in range(N)], ytmp=new(M * C), xnt=new(M * C), meant=new(B * C), ct=new(B, H),
            # the delta sweep's own gradient buffers
            dmean=new(B * C), gA=new(M * C), gB=new(M * C), gC=new(M * C), gD=new(M * C), gAd=new(M * C), gBd=new(M * C),
            df1=new(M * FF), dqkv=new(M * 3 * C), datt=new(M * C), slabs=new(max(self.slab_floats(), self.ln_blocks * 2 * C)))

    def tangent(self, top_t):
        """``top_t``: tangent of the encoder's top buffer; returns (tensor, offset, item stride) of the tangent of c."""
        e = self.eng
        p, code, B = e.model._param, e.code, e.B
        C, FF, S, H = self.C, self.FF, self.S, self.out
        M = B * S
        Ltop, t0 = e.geo.alloc[-1], e.T - e.K - e.V
        P = _hip.ptr
        self._gp_buffers()
        gp, dp, seed = self.gp, self.drop_p, self.drop_seed

        def ln_t(at, bt, r, stats, wname, rt_out, yt, site=0):
            _hip.call("cpc_ln_tangent", P(at), P(bt), P(r), P(stats), P(p[wname + ".weight"]), P(rt_out), P(yt), M, C,
                      dp if bt is not None else 0.0, seed, site)
        _hip.call("cpc_pe_scale_fwd", P(top_t, t0 * C), P(gp.zero_pe), P(gp.Xt[0]), B, S, C, Ltop * C, self.z_scale, code)
        for l in range(self.N):
            Xt, w = gp.Xt[l], self.w[l]
            _hip.gemm_nt(P(Xt), P(w["in"]), P(gp.qkvt[l]), M, 3 * C, C, C, C, 3 * C, code)
            _hip.call("cpc_attn_tangent", P(self.qkv[l]), P(gp.qkvt[l]), P(self.P[l]), P(gp.attt[l]), B, S, C, self.heads, dp, seed,
                      4 * l + self.SITE_ATTN)
            _hip.gemm_nt(P(gp.attt[l]), P(w["o"]), P(gp.ytmp), M, C, C, C, C, C, code)
            ln_t(Xt, gp.ytmp, self.r1[l], self.st1[l], self._lname(l, "norm1"), gp.r1t[l], gp.x1t[l], 4 * l + self.SITE_DROP1)
            _hip.gemm_nt(P(gp.x1t[l]), P(w["l1"]), P(gp.f1t[l]), M, FF, C, C, C, FF, code, mask=P(self.f1[l]))
            if dp > 0.0:
                gp.f1t[l].mul_(1.0 / (1.0 - dp))          # the stored f1 is the dropped activation: its zeros carry the keep mask
            _hip.gemm_nt(P(gp.f1t[l]), P(w["l2"]), P(gp.ytmp), M, C, FF, FF, FF, C, code)
            ln_t(gp.x1t[l], gp.ytmp, self.r2[l], self.st2[l], self._lname(l, "norm2"), gp.r2t[l], gp.Xt[l + 1], 4 * l + self.SITE_DROP2)
        ln_t(gp.Xt[self.N], None, self.X[self.N], self.stn, self.prefix + "encoder.norm", None, gp.xnt)
        _hip.call("cpc_mean_time", P(gp.xnt), P(gp.meant), B, S, C, code)
        _hip.gemm_nt(P(gp.meant), P(self.w_end), P(gp.ct), B, H, C, C, C, H, code, flags=_hip.GEMM_OUT_F32)
        return gp.ct, 0, H

    def gp_grads(self, gp_grad):
        self._gp_grad = gp_grad           # filled by the last pass (_backward_gp)

    def _backward_gp(self, dc):
        """The last pass of a gradient-penalty step: the ordinary backward of ``dc`` (lambda) with the second-order terms joining
        on the way down, beside the sweep of the summed scores' adjoint (delta, from the dc of pass 1) that gives the penalty parts
        of the weight gradients.  Everything on the main stream."""
        e = self.eng
        g, p, code, B = e.model._grad, e.model._param, e.code, e.B
        C, FF, S, H = self.C, self.FF, self.S, self.out
        M = B * S
        Ltop, t0 = e.geo.alloc[-1], e.T - e.K - e.V
        P = _hip.ptr
        gp, gpg, dp, seed = self.gp, self._gp_grad, self.drop_p, self.drop_seed
        drop = dp > 0.0
        sc, nb = gp.slabs, self.ln_blocks

        def ln_pair(lam, dl, rt, r, stats, wname, lam_out, lam_out_b, dl_out, dl_out_b, bcast=0, gscale=1.0, site=0):
            # lambda through the LayerNorm (+ its weight / bias gradients), the second-order terms from delta, then delta itself
            self._ln_bwd(lam[0], lam[1], r, stats, wname, lam_out, bcast=bcast, gscale=gscale, dr_b=lam_out_b, site=site)
            _hip.call("cpc_ln_gp", P(dl[0]), P(dl[1]), P(rt), P(r), P(stats), P(p[wname + ".weight"]), P(lam_out), P(lam_out_b), P(sc),
                      M, C, bcast, gscale, nb, dp if lam_out_b is not None else 0.0, seed, site)
            _hip.call("cpc_reduce_slabs", P(sc), P(gpg[wname + ".weight"]), 1, C, nb, C, 1, 1, 0, 0)
            _hip.call("cpc_ln_bwd", P(dl[0]), P(dl[1]), P(r), P(stats), P(p[wname + ".weight"]), P(dl_out), P(sc), M, C, bcast, gscale,
                      nb, P(dl_out_b), dp if dl_out_b is not None else 0.0, seed, site, code)

        def linear_grads(lam_y, dl_y, x, xt, k, l, rows, cols):
            e._colsum_to_grad(P(lam_y), g[self._lname(l, self._BNAMES[k])], M, rows, scratch=sc)
            e._tn_to_grad(P(lam_y), P(x), g[self._lname(l, self._WNAMES[k])], M, rows, cols, rows, cols, self.split[k], scratch=sc)
            e._tn_to_grad(P(dl_y), P(xt), gpg[self._lname(l, self._WNAMES[k])], M, rows, cols, rows, cols, self.split[k], scratch=sc)

        # end_layer and the mean over time
        _hip.call("cpc_cast2d", P(dc), P(self.dct), B, H, H, 1, code)
        _hip.gemm_tn(P(self.dct), P(self.mean), P(g[self.prefix + "end_layer.weight"]), B, H, C, H, C, C, code, flags=_hip.GEMM_OUT_F32)
        e._colsum_to_grad(P(self.dct), g[self.prefix + "end_layer.bias"], B, H)
        _hip.gemm_nt(P(self.dct), P(self.w_end_t), P(self.dmean), B, C, H, H, H, C, code)
        _hip.gemm_tn(P(gp.dc1), P(gp.meant), P(gpg[self.prefix + "end_layer.weight"]), B, H, C, H, C, C, code, flags=_hip.GEMM_OUT_F32)
        _hip.gemm_nt(P(gp.dc1), P(self.w_end_t), P(gp.dmean), B, C, H, H, H, C, code)
        lamA, lamB, lamAd, lamBd = self.gA[0], self.gB[0], self.gAd[0], self.gBd[0]
        ln_pair((self.dmean, None), (gp.dmean, None), gp.Xt[self.N], self.X[self.N], self.stn, self.prefix + "encoder.norm",
                self.gA[1], None, gp.gA, None, bcast=S, gscale=1.0 / S)
        lam, dl = (self.gA[1], None), (gp.gA, None)
        for l in range(self.N - 1, -1, -1):
            wt = self.wt[l]
            # norm2 over r2 = x1 + dropout(f2); delta's input gA is free again once this is done
            ln_pair(lam, dl, gp.r2t[l], self.r2[l], self.st2[l], self._lname(l, "norm2"), lamB, lamBd if drop else None,
                    gp.gB, gp.gBd if drop else None, site=4 * l + self.SITE_DROP2)
            lam_y2, dl_y2 = (lamBd, gp.gBd) if drop else (lamB, gp.gB)
            linear_grads(lam_y2, dl_y2, self.f1[l], gp.f1t[l], "l2", l, C, FF)
            _hip.gemm_nt(P(lam_y2), P(wt["l2"]), P(self.df1[0]), M, FF, C, C, C, FF, code, mask=P(self.f1[l]))
            _hip.gemm_nt(P(dl_y2), P(wt["l2"]), P(gp.df1), M, FF, C, C, C, FF, code, mask=P(self.f1[l]))
            linear_grads(self.df1[0], gp.df1, self.x1[l], gp.x1t[l], "l1", l, FF, C)
            _hip.gemm_nt(P(self.df1[0]), P(wt["l1"]), P(self.gC), M, C, FF, FF, FF, C, code)
            _hip.gemm_nt(P(gp.df1), P(wt["l1"]), P(gp.gC), M, C, FF, FF, FF, C, code)
            # norm1 over r1 = x + dropout(attention output projection)
            ln_pair((lamB, self.gC), (gp.gB, gp.gC), gp.r1t[l], self.r1[l], self.st1[l], self._lname(l, "norm1"), lamA,
                    lamAd if drop else None, gp.gA, gp.gAd if drop else None, site=4 * l + self.SITE_DROP1)
            lam_y, dl_y = (lamAd, gp.gAd) if drop else (lamA, gp.gA)
            linear_grads(lam_y, dl_y, self.att[l], gp.attt[l], "o", l, C, C)
            _hip.gemm_nt(P(lam_y), P(wt["o"]), P(self.datt), M, C, C, C, C, C, code)
            _hip.gemm_nt(P(dl_y), P(wt["o"]), P(gp.datt), M, C, C, C, C, C, code)
            site = 4 * l + self.SITE_ATTN
            _hip.call("cpc_attn_bwd", P(self.qkv[l]), P(self.P[l]), P(self.datt), P(self.dqkv[0]), B, S, C, self.heads, dp, seed, site, code)
            _hip.call("cpc_attn_gp", P(self.qkv[l]), P(gp.qkvt[l]), P(self.P[l]), P(gp.datt), P(self.dqkv[0]), B, S, C, self.heads, dp,
                      seed, site)
            _hip.call("cpc_attn_bwd", P(self.qkv[l]), P(self.P[l]), P(gp.datt), P(gp.dqkv), B, S, C, self.heads, dp, seed, site, code)
            linear_grads(self.dqkv[0], gp.dqkv, self.X[l], gp.Xt[l], "in", l, 3 * C, C)
            _hip.gemm_nt(P(self.dqkv[0]), P(wt["in"]), P(self.gD), M, C, 3 * C, 3 * C, 3 * C, C, code)
            _hip.gemm_nt(P(gp.dqkv), P(wt["in"]), P(gp.gD), M, C, 3 * C, 3 * C, 3 * C, C, code)
            # (lamA is read by the next layer's norm2 and rewritten only by its norm1, after that read)
            lam, dl = (lamA, self.gD), (gp.gA, gp.gD)
        _hip.call("cpc_pe_scale_bwd", P(lam[0]), P(lam[1]), P(e.dact[-1], t0 * C), B, S, C, Ltop * C, self.z_scale, code)

    def backward(self, dc):
        e = self.eng
        phase = getattr(e, "_gp_phase", 0)
        if phase == 1:               # gradient penalty, pass 1: dc is the adjoint of the summed scores (the last pass starts from it)
            self._gp_buffers()
            self.gp.dc1.copy_(dc)
        elif phase == 3:
            if e.use_aux:
                torch.cuda.current_stream().wait_stream(e.aux)
            return self._backward_gp(dc)
        g, code, B = e.model._grad, e.code, e.B
        C, FF, S, H = self.C, self.FF, self.S, self.out
        M = B * S
        Ltop, t0 = e.geo.alloc[-1], e.T - e.K - e.V
        P = _hip.ptr
        # end_layer and the mean over time
        _hip.call("cpc_cast2d", P(dc), P(self.dct), B, H, H, 1, code)
        _hip.gemm_tn(P(self.dct), P(self.mean), P(g[self.prefix + "end_layer.weight"]), B, H, C, H, C, C, code, flags=_hip.GEMM_OUT_F32)
        e._colsum_to_grad(P(self.dct), g[self.prefix + "end_layer.bias"], B, H)
        _hip.gemm_nt(P(self.dct), P(self.w_end_t), P(self.dmean), B, C, H, H, H, C, code)
        if self.scratch is None:
            self.scratch = torch.empty(self.slab_floats(), device=e.device, dtype=torch.float32)     # side-stream workspace
        sc = self.scratch
        top = self.N & 1
        self._ln_bwd(self.dmean, None, self.X[self.N], self.stn, self.prefix + "encoder.norm", self.gA[top], bcast=S, gscale=1.0 / S)
        g1, g2 = self.gA[top], None
        for l in range(self.N - 1, -1, -1):
            wt = self.wt[l]
            gname = lambda k, names: g[self._lname(l, names[k])]
            drop = self.drop_p > 0.0
            s_ = l & 1
            ev = self._ev[l]
            if e.use_aux and l + 2 <= self.N - 1:
                # this layer writes buffer set l & 1, which the side stream last read for layer l + 2
                torch.cuda.current_stream().wait_event(self._ev[l + 2][4])
            gA, gB, gAd_, gBd_, df1, dqkv = self.gA[s_], self.gB[s_], self.gAd[s_], self.gBd[s_], self.df1[s_], self.dqkv[s_]
            # norm2 over r2 = x1 + dropout(f2)
            gBd = gBd_ if drop else gB
            self._ln_bwd(g1, g2, self.r2[l], self.st2[l], self._lname(l, "norm2"), gB, dr_b=gBd_ if drop else None,
                         site=4 * l + self.SITE_DROP2)
            with e.side(ev[0]):
                e._colsum_to_grad(P(gBd), gname("l2", self._BNAMES), M, C, scratch=sc)
                e._tn_to_grad(P(gBd), P(self.f1[l]), gname("l2", self._WNAMES), M, C, FF, C, FF, self.split["l2"], scratch=sc)
            _hip.gemm_nt(P(gBd), P(wt["l2"]), P(df1), M, FF, C, C, C, FF, code, mask=P(self.f1[l]))
            with e.side(ev[1]):
                e._colsum_to_grad(P(df1), gname("l1", self._BNAMES), M, FF, scratch=sc)
                e._tn_to_grad(P(df1), P(self.x1[l]), gname("l1", self._WNAMES), M, FF, C, FF, C, self.split["l1"], scratch=sc)
            _hip.gemm_nt(P(df1), P(wt["l1"]), P(self.gC), M, C, FF, FF, FF, C, code)
            # norm1 over r1 = x + dropout(attention output projection)
            gAd = gAd_ if drop else gA
            self._ln_bwd(gB, self.gC, self.r1[l], self.st1[l], self._lname(l, "norm1"), gA, dr_b=gAd_ if drop else None,
                         site=4 * l + self.SITE_DROP1)
            with e.side(ev[2]):
                e._colsum_to_grad(P(gAd), gname("o", self._BNAMES), M, C, scratch=sc)
                e._tn_to_grad(P(gAd), P(self.att[l]), gname("o", self._WNAMES), M, C, C, C, C, self.split["o"], scratch=sc)
            _hip.gemm_nt(P(gAd), P(wt["o"]), P(self.datt), M, C, C, C, C, C, code)
            _hip.call("cpc_attn_bwd", P(self.qkv[l]), P(self.P[l]), P(self.datt), P(dqkv), B, S, C, self.heads, self.drop_p,
                      self.drop_seed, 4 * l + self.SITE_ATTN, code)
            with e.side(ev[3]):
                e._colsum_to_grad(P(dqkv), gname("in", self._BNAMES), M, 3 * C, scratch=sc)
                e._tn_to_grad(P(dqkv), P(self.X[l]), gname("in", self._WNAMES), M, 3 * C, C, 3 * C, C, self.split["in"], scratch=sc)
                if e.use_aux:
                    ev[4].record(e.aux)
            _hip.gemm_nt(P(dqkv), P(wt["in"]), P(self.gD), M, C, 3 * C, 3 * C, 3 * C, C, code)
            g1, g2 = gA, self.gD
        # positional encoder: dz = sqrt(C) * dx0 into rows [t0, t0+V) of the encoder's top-layer gradient
        _hip.call("cpc_pe_scale_bwd", P(g1), P(g2), P(e.dact[-1], t0 * C), B, S, C, Ltop * C, self.z_scale, code)


class _Float32EngineView:
    """What a context network sees of an engine with bf16 storage when it computes in float32 inside it (Float32Context): ``dt`` /
    ``code`` are float32's, ``act[-1]`` / ``dact[-1]`` float32 shadows of the top-layer buffer and its gradient in the same
    [B][L_top][E] indexing; every other attribute is the engine's own, and the engine's methods run with this object as ``self``
    (so that _tn_to_grad, _colsum_to_grad, _pick_split, _chunk pick the float32 kernels); attribute writes go to the engine."""

    def __init__(self, eng, top32, dtop32):
        d = self.__dict__
        d["_e"], d["dt"], d["code"] = eng, torch.float32, _hip.dtype_code(torch.float32)
        d["act"], d["dact"] = list(eng.act[:-1]) + [top32], list(eng.dact[:-1]) + [dtop32]

    def __getattr__(self, name):
        v = getattr(self._e, name)
        if getattr(v, "__self__", None) is self._e and hasattr(v, "__func__"):
            return v.__func__.__get__(self, type(self))
        return v

    def __setattr__(self, name, value):
        setattr(self._e, name, value)


class Float32Context:
    """An AudioGRUModel / AttentionModel context that computes in float32 inside an engine with bf16 storage.  Built for the
    gradient-penalty engines only (make_context): the penalty's tangent recurrence and reverse sweep of these two networks
    (GRUContext.tangent / gp_grads, AttentionContext.tangent / _backward_gp) carry second-order terms of saturating gates,
    LayerNorm and softmax and exist as float32 kernels; the encoder — where the FLOPs are — keeps bf16 storage.  The frames the
    context reads are converted into a float32 shadow of the top-layer buffer on the way in, dz and c on the way out."""

    def __init__(self, eng, cls, ar):
        self.eng = eng
        B, Ltop, E, H = eng.B, eng.geo.alloc[-1], eng.E, eng.H
        f32 = dict(device=eng.device, dtype=torch.float32)
        self.top32, self.dtop32 = torch.zeros(B * Ltop * E, **f32), torch.zeros(B * Ltop * E, **f32)
        self.top_t32 = None
        self.inner = cls(_Float32EngineView(eng, self.top32, self.dtop32), ar)
        self.c_st = torch.zeros(B * H, device=eng.device, dtype=eng.dt)
        self.ct_st = torch.zeros(B * H, device=eng.device, dtype=eng.dt)
        self.z_scale = getattr(self.inner, "z_scale", 1.0)
        self.ahead_ok = getattr(self.inner, "ahead_ok", False)

    def _z(self, flat):
        e = self.eng
        t0 = e.T - e.K - e.V
        return flat.view(e.B, e.geo.alloc[-1], e.E)[:, t0:t0 + e.V, :]

    def allocate(self):
        self.inner.allocate()

    def slab_floats(self):
        return self.inner.slab_floats()

    def prepare_weights(self):
        self.inner.prepare_weights()

    def forward(self):
        self._z(self.top32).copy_(self._z(self.eng.act[-1]))
        self.inner.forward()
        self.c_st.view(self.eng.B, self.eng.H).copy_(self.inner.c_float())

    def c_operand(self):
        return self.c_st, 0, self.eng.H

    def c_float(self):
        return self.inner.c_float()

    def backward(self, dc):
        self.inner.backward(dc)
        self._z(self.eng.dact[-1]).copy_(self._z(self.dtop32))

    def tangent(self, top_t):
        if self.top_t32 is None:
            self.top_t32 = torch.zeros_like(self.top32)
        self._z(self.top_t32).copy_(self._z(top_t))
        ct, off, stride = self.inner.tangent(self.top_t32)
        B, H = self.eng.B, self.eng.H
        self.ct_st.view(B, H).copy_(ct.reshape(-1)[off:].as_strided((B, H), (stride, 1)))
        return self.ct_st, 0, H

    def gp_grads(self, gp_grad):
        self.inner.gp_grads(gp_grad)


def make_context(eng, ar):
    from .audio_model import AudioGRUModel, ConvolutionalArModel
    from .attention_model import AttentionModel
    if getattr(eng, "gp_capable", False) and eng.dt != torch.float32 and isinstance(ar, (AudioGRUModel, AttentionModel)):
        return Float32Context(eng, GRUContext if isinstance(ar, AudioGRUModel) else AttentionContext, ar)
    if isinstance(ar, AudioGRUModel):
        return GRUContext(eng, ar)
    if isinstance(ar, ConvolutionalArModel):
        # ConvArContext is the lean path for the reference's plain default (no pooling in the first block, pooling in every
        # later one, stride 1); every other configuration runs on the grid kernels
        lean = not (ar.batch_norm or ar.residual) and all(s == 1 for s in ar.strides) and ar.poolings[0] == 1 and \
            all(p > 1 for p in ar.poolings[1:])
        if getattr(eng, "gp_capable", False):
            lean = False          # the gradient penalty's tangent pass lives in the grid implementation
        if not lean:
            from .scalogram_engine import ConvArGridContext
            return ConvArGridContext(eng, ar)
        return ConvArContext(eng, ar)
    from .attention_model import AttentionModel
    if isinstance(ar, AttentionModel):
        return AttentionContext(eng, ar)
    from .scalogram_model import ScalogramResidualEncoder
    if isinstance(ar, ScalogramResidualEncoder):
        from .scalogram_engine import ResNetArContext
        return ResNetArContext(eng, ar)
    raise NotImplementedError(f"no HIP context network for {type(ar).__name__}")


class ContextOnlyEngine(CPCEngine):
    """A context network called on its own (``AudioGRUModel(...)(z)`` etc.): z (B, E, V) -> c (B, H), inference only.
    Reuses the context classes unchanged by presenting z as the encoder's top buffer."""

    def __init__(self, owner, batch_size, enc_size, steps, device, dtype: torch.dtype):
        from types import SimpleNamespace
        self.model = owner
        self.device = torch.device(device)
        self.dt = dtype
        self.code = _hip.dtype_code(dtype)
        self.B, self.E, self.V, self.K = int(batch_size), int(enc_size), int(steps), 0
        self.H = int(owner.ar_size)
        self.T, self.L, self.x_off, self.n = self.V, self.V, 0, 0
        self.colsum_blocks = 1024
        owner._flatten_parameters(self.device)
        self.geo = SimpleNamespace(alloc=[self.V], valid=[self.V])
        self._keep = []
        self.act, self.dact = [], []
        for store in (self.act, self.dact):
            full, view, _ = self._buf(self.B * self.V, self.E)
            self._keep.append(full)
            store.append(view)
        self.aux = side_stream(self.device)
        self.ctx = make_context(self, owner.autoregressive_model)
        self._alloc_head([1])

    def prepare_weights(self):
        self.ctx.prepare_weights()

    def run(self, z):
        if not z.is_cuda:
            raise RuntimeError("the context networks run on the GPU only (no CPU fallback)")
        if tuple(z.shape) != (self.B, self.E, self.V):
            raise ValueError(f"expected input of shape ({self.B}, {self.E}, {self.V}) = (batch, channels, steps), got {tuple(z.shape)}")
        self.act[-1].view(self.B, self.V, self.E).copy_(z.detach().transpose(1, 2))
        self.prepare_weights()
        self.ctx.forward()
        return self.ctx.c_float().clone()

    def backward_context(self, dc):
        """dc (B, H) -> d z (B, E, V) float32; the context network's parameter gradients go to the owner's flat gradient buffer
        (the autograd bridge of a stand-alone call, audio_model._ContextForward)."""
        self._ahead = None
        self.dact[-1].zero_()
        self.dc.copy_(dc)
        self.ctx.backward(self.dc)
        if self.use_aux:
            torch.cuda.current_stream().wait_stream(self.aux)
        return self.dact[-1].view(self.B, self.V, self.E).float().transpose(1, 2)


def standalone_context_forward(ar, z, ar_size):
    """Host side of ``<context network>.forward(z)`` outside an AudioPredictiveCodingModel."""
    from .audio_model import AudioPredictiveCodingModel, _LinearParams
    import torch.nn as nn
    owner = getattr(ar, "_owner", None)
    if owner is None:
        object.__setattr__(ar, "_owner", None)
        owner = AudioPredictiveCodingModel.__new__(AudioPredictiveCodingModel)
        nn.Module.__init__(owner)
        owner.enc_size, owner.ar_size = int(z.shape[1]), int(ar_size)
        owner.visible_steps, owner.prediction_steps = int(z.shape[2]), 0
        owner.encoder = None
        owner.autoregressive_model = ar
        owner.prediction_model = _LinearParams(int(ar_size), 8)
        owner.compute_dtype = getattr(ar, "compute_dtype", torch.float32)
        owner._engines, owner._flat_param, owner._flat_grad, owner._param, owner._grad = {}, None, None, {}, {}
        owner._scalogram = False
        object.__setattr__(ar, "_owner", owner)
    dev = z.device
    if next(owner.prediction_model.parameters()).device != dev:
        owner.prediction_model.to(dev)
    key = ("ctx", tuple(z.shape), owner.compute_dtype, str(dev))
    eng = owner._engines.get(key)
    if eng is None or owner._flat_param is None or any(p.data_ptr() != owner._param[n].data_ptr() for n, p in owner.named_parameters()):
        owner._flatten_parameters(dev)
        eng = ContextOnlyEngine(owner, z.shape[0], z.shape[1], z.shape[2], dev, owner.compute_dtype)
        owner._engines = {key: eng}
    if torch.is_grad_enabled() and (z.requires_grad or any(p_.requires_grad for p_ in ar.parameters())):
        from .audio_model import _ContextForward
        names = [n for n, _ in owner.named_parameters() if n.startswith("autoregressive_model.")]
        return _ContextForward.apply(eng, names, z, *[dict(owner.named_parameters())[n] for n in names])
    return eng.run(z.float())


class GlobalNegatives:
    """InfoNCE over the GLOBAL batch under one process per GPU — the loss the reference's nn.DataParallel wrap computes
    (setup_functions.py:112-115: outputs gathered to one device, scores and loss over all B x N items; SURVEY.md 8a13 / 8f
    rank 3).  Every rank all-gathers predicted_z and targets (2 x B K E storage-dtype elements per rank), computes the same
    global loss and its gradient with respect to all predictions / targets, and back-propagates the slice that belongs to
    its own clips; the parameter gradients of the ranks then ADD UP to the gradient of the global loss (all-reduce sum, no
    division).  The default data-parallel mode keeps per-GPU negatives (BASELINE.json's north star); this one is opt-in
    (``ContrastiveEstimationTrainer.global_negatives``).  Both loss branches."""

    def __init__(self, eng):
        import torch.distributed as dist
        self.dist, self.eng = dist, eng
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        B, K, E, dev, dt, f32 = eng.B, eng.K, eng.E, eng.device, eng.dt, torch.float32
        GB = B * self.world
        self.GB, self.ld = GB, _ceil_div(GB, 8) * 8
        self.local_targ = torch.empty(B * K * E, device=dev, dtype=dt)
        self.pred_all = torch.empty(GB * K * E, device=dev, dtype=dt)
        self.targ_all = torch.empty(GB * K * E, device=dev, dtype=dt)
        self.dpred_all = torch.empty(GB * K * E, device=dev, dtype=dt)
        self.dtarg_all = torch.empty(GB * K * E, device=dev, dtype=dt)
        self.S = torch.zeros(K * GB * self.ld, device=dev, dtype=f32)
        self.dS = torch.zeros(K * GB * self.ld, device=dev, dtype=dt)
        self.dST = torch.zeros(K * GB * self.ld, device=dev, dtype=dt)
        self.out = eng.nce_out          # loss values and NaN flags go straight to the engine's result cell
        self.ws = torch.empty(int(_hip.lib().cpc_nce_workspace_floats(GB, K)), device=dev, dtype=f32)

    def _all_timesteps(self, softplus: bool, regularization: float):
        """The full (GB K) x (GB K) score matrix over the gathered predictions / targets (engine.nce_all_forward_backward on
        the global batch)."""
        e = self.eng
        code, E, K, GB = e.code, e.E, e.K, self.GB
        R = GB * K
        ld = _ceil_div(R, 8) * 8
        if getattr(self, "S_all", None) is None:
            dev, dt, f32 = e.device, e.dt, torch.float32
            self.S_all = torch.zeros(R * ld, device=dev, dtype=f32)
            self.ST_all = torch.zeros(R * ld, device=dev, dtype=f32)
            self.dS_all = torch.zeros(R * ld, device=dev, dtype=dt)
            self.dST_all = torch.zeros(R * ld, device=dev, dtype=dt)
            self.ws_all = torch.empty(int(_hip.lib().cpc_nce_all_workspace_floats(GB, K)), device=dev, dtype=f32)
            self.targT = torch.zeros(E, ld, device=dev, dtype=dt)
            self.predT = torch.zeros(E, ld, device=dev, dtype=dt)
        P = _hip.ptr
        _hip.gemm_nt(P(self.pred_all), P(self.targ_all), P(self.S_all), R, R, E, E, E, ld, code, flags=_hip.GEMM_OUT_F32)
        _hip.gemm_nt(P(self.targ_all), P(self.pred_all), P(self.ST_all), R, R, E, E, E, ld, code, flags=_hip.GEMM_OUT_F32)
        _hip.call("cpc_nce_loss_all", P(self.S_all), P(self.ST_all), P(self.dS_all), P(self.dST_all), P(self.out), P(self.ws_all), GB, K,
                  ld, 1 if softplus else 0, C.c_float(regularization), code)
        self.targT[:, :R].copy_(self.targ_all.view(R, E).t())
        self.predT[:, :R].copy_(self.pred_all.view(R, E).t())
        _hip.gemm_nt(P(self.dS_all), P(self.targT), P(self.dpred_all), R, E, ld, ld, ld, E, code)
        _hip.gemm_nt(P(self.dST_all), P(self.predT), P(self.dtarg_all), R, E, ld, ld, ld, E, code)

    def _all_timesteps_strips(self, softplus: bool, regularization: float):
        """The all-timesteps loss over the global batch WITHOUT the global score matrix on every rank: rank r forms two strips of it,
        (all predictions) x (its own targets) and (its own predictions) x (all targets) — 2 / world of the matrix instead of all of it.
        The column strip holds every row of its columns, so the column log-sum-exps, the regulariser's pair means and this rank's part of
        the loss are complete there (three partial sums + a maximum are all-reduced); its gradient gives d loss / d (own targets).  The row
        strip needs the log-sum-exps of ALL columns (one all-gather of R floats) and gives d loss / d (own predictions).  Fused kernels
        (cpc_score_lse / cpc_nce_fused_*): no f32 score matrix, no second (transposed) GEMM.  At 8 x 256 clips x 12 steps a rank computes
        4 x 77 GFLOP instead of 4 x 618."""
        e, dist = self.eng, self.dist
        code, E, K, B, GB, W = e.code, e.E, e.K, e.B, self.GB, self.world
        Rl, Rg = B * K, GB * K
        lo = self.rank * Rl
        s = getattr(self, "_strips", None)
        if s is None:
            dev, dt, f32 = e.device, e.dt, torch.float32
            s = self._strips = SimpleNamespace(
                Sc=torch.empty(Rg, Rl, device=dev, dtype=f32), dSc=torch.empty(Rg, Rl, device=dev, dtype=dt),
                dScT=torch.empty(Rl, Rg, device=dev, dtype=dt), Sr=torch.empty(Rl, Rg, device=dev, dtype=f32),
                dSr=torch.empty(Rl, Rg, device=dev, dtype=dt),
                pm=torch.empty(Rg // 256, Rl, device=dev, dtype=f32), ps=torch.empty(Rg // 256, Rl, device=dev, dtype=f32),
                pm2=torch.empty(Rl // 256, Rg, device=dev, dtype=f32), ps2=torch.empty(Rl // 256, Rg, device=dev, dtype=f32),
                valid=torch.zeros(Rg, device=dev, dtype=f32), lse_loc=torch.empty(Rl, device=dev, dtype=f32),
                lse_all=torch.empty(Rg, device=dev, dtype=f32), colp=torch.empty(_ceil_div(Rl, 256), 2, device=dev, dtype=f32),
                gradp=torch.empty(int(_hip.lib().cpc_nce_fused_grad_blocks(GB, Rl)), device=dev, dtype=f32),
                sums=torch.zeros(4, device=dev, dtype=f32),
                targT=torch.zeros(E, Rg, device=dev, dtype=dt), predT=torch.zeros(E, Rg, device=dev, dtype=dt))
        P, L, F = _hip.ptr, C.c_longlong, C.c_float
        sp = 1 if softplus else 0
        pred_all, targ_all = self.pred_all.view(Rg, E), self.targ_all.view(Rg, E)
        # column strip: every prediction against this rank's targets
        _hip.call("cpc_score_lse", P(pred_all), P(targ_all, lo * E), P(s.Sc), P(s.pm), P(s.ps), P(s.valid), Rg, Rl, E, L(E), L(E), L(Rl), -lo,
                  key="score_lse<bf16,256>", work=2.0 * Rg * Rl * E)
        _hip.call("cpc_nce_lse_merge", P(s.pm), P(s.ps), Rg // 256, Rl, sp, F(Rg), P(s.lse_loc), P(s.colp))
        _hip.call("cpc_nce_fused_grad", P(s.Sc), P(s.lse_loc), P(s.dSc), P(s.dScT), P(s.gradp), GB, K, Rl, L(Rl), L(Rg), -lo, sp,
                  F(regularization), F(Rg), F(GB))
        _hip.call("cpc_nce_fused_finalize", P(s.colp), s.colp.shape[0], P(s.valid, lo), Rl, P(s.gradp), s.gradp.numel(), P(s.sums), 1, F(Rg),
                  F(GB), K, F(regularization), sp, None)
        dist.all_reduce(s.sums[0:3])
        dist.all_reduce(s.sums[3:4], op=dist.ReduceOp.MAX)
        _hip.call("cpc_nce_fused_finalize", None, 0, None, 0, None, 0, P(s.sums), 2, F(Rg), F(GB), K, F(regularization), sp, P(self.out))
        dist.all_gather([s.lse_all[r * Rl:(r + 1) * Rl] for r in range(W)], s.lse_loc)
        # d loss / d own targets [c][:] = sum over ALL predictions r of dS[r][c] pred[r][:]
        s.predT.copy_(pred_all.t())
        _hip.gemm_nt(P(s.dScT), P(s.predT), P(self.dtarg_all, lo * E), Rl, E, Rg, Rg, Rg, E, code)
        # row strip: this rank's predictions against every target (the same kernel: identical accumulators; its column pairs are not used)
        _hip.call("cpc_score_lse", P(pred_all, lo * E), P(targ_all), P(s.Sr), P(s.pm2), P(s.ps2), None, Rl, Rg, E, L(E), L(E), L(Rg), lo,
                  key="score_lse<bf16,256>", work=2.0 * Rl * Rg * E)
        _hip.call("cpc_nce_fused_grad", P(s.Sr), P(s.lse_all), P(s.dSr), None, None, B, K, Rg, L(Rg), L(0), lo, sp, F(regularization), F(Rg),
                  F(GB))
        s.targT.copy_(targ_all.t())
        _hip.gemm_nt(P(s.dSr), P(s.targT), P(self.dpred_all, lo * E), Rl, E, Rg, Rg, Rg, E, code)

    # ---- Wasserstein gradient penalty with softplus scores under global negatives (scalogram_engine._gp_step): the summed scores run over
    # the GLOBAL score matrix, so the seeds of the penalty's passes are contractions with W1 = sigmoid(s) and W2 = softplus''(s) * (tangent
    # of s) over all ranks' predictions and targets.  Every rank forms the whole matrix (a parity path: the penalty costs three passes
    # through the network anyway) and keeps the rows / columns of its own clips.
    def _gp_sizes(self, all_timesteps):
        K, GB = self.eng.K, self.GB
        if all_timesteps:
            R = GB * K
            return 1, R, _ceil_div(R, 8) * 8
        return K, GB, self.ld

    def _gp_state(self, all_timesteps):
        g = getattr(self, "_gp", None)
        if g is not None and g.key == bool(all_timesteps):
            return g
        e = self.eng
        nb, R, ld = self._gp_sizes(all_timesteps)
        n = nb * R * ld
        f32 = dict(device=e.device, dtype=torch.float32)
        st = dict(device=e.device, dtype=e.dt)
        g = self._gp = SimpleNamespace(key=bool(all_timesteps), S=torch.zeros(n, **f32), St1=torch.zeros(n, **f32), St2=torch.zeros(n, **f32),
                                       W1=torch.zeros(n, **f32), W1T=torch.zeros(n, **f32), W2=torch.zeros(n, **f32), W2T=torch.zeros(n, **f32),
                                       pred_t_all=torch.zeros_like(self.pred_all), targ_t_all=torch.zeros_like(self.targ_all),
                                       local=torch.zeros_like(self.local_targ), out_p=torch.zeros_like(self.pred_all),
                                       out_t=torch.zeros_like(self.targ_all), out_p2=torch.zeros_like(self.pred_all),
                                       out_t2=torch.zeros_like(self.targ_all))
        for name in ("W1", "W1T", "W2", "W2T"):          # GEMM operands in the storage dtype
            setattr(g, name + "s", getattr(g, name) if e.dt == torch.float32 else torch.zeros(n, **st))
        if all_timesteps:
            g.aT, g.bT = torch.zeros(e.E, ld, **st), torch.zeros(e.E, ld, **st)
        return g

    def _gp_gather(self, pred, top, pred_all, targ_all, local):
        e = self.eng
        B, K, E, T, Ltop = e.B, e.K, e.E, e.T, e.geo.alloc[-1]
        n = B * K * E
        local.view(B, K, E).copy_(top.view(B, Ltop, E)[:, T - K:T, :])
        self.dist.all_gather([pred_all[r * n:(r + 1) * n] for r in range(self.world)], pred.contiguous())
        self.dist.all_gather([targ_all[r * n:(r + 1) * n] for r in range(self.world)], local)

    def _gp_scores(self, pa, ta, out, all_timesteps):
        e = self.eng
        P, code, E, K, GB = _hip.ptr, e.code, e.E, e.K, self.GB
        nb, R, ld = self._gp_sizes(all_timesteps)
        if all_timesteps:
            _hip.gemm_nt(P(pa), P(ta), P(out), R, R, E, E, E, ld, code, flags=_hip.GEMM_OUT_F32)
        else:
            _hip.gemm_nt(P(pa), P(ta), P(out), GB, GB, E, K * E, K * E, ld, code, a_batch=E, b_batch=E, c_batch=GB * ld, batch=K,
                         flags=_hip.GEMM_OUT_F32)

    def _gp_contract(self, W, WT, pa, ta, out_p, out_t, all_timesteps):
        """out_p[r] = sum_c W[r][c] ta[c],  out_t[c] = sum_r W[r][c] pa[r]  (W, WT in the storage dtype; all global rows / columns)."""
        e, g = self.eng, self._gp
        P, code, E, K, GB = _hip.ptr, e.code, e.E, e.K, self.GB
        nb, R, ld = self._gp_sizes(all_timesteps)
        if all_timesteps:
            g.aT[:, :R].copy_(ta.view(R, E).t())
            g.bT[:, :R].copy_(pa.view(R, E).t())
            _hip.gemm_nt(P(W), P(g.aT), P(out_p), R, E, ld, ld, ld, E, code)
            _hip.gemm_nt(P(WT), P(g.bT), P(out_t), R, E, ld, ld, ld, E, code)
        else:
            _hip.gemm_tn(P(WT), P(ta), P(out_p), GB, GB, E, ld, K * E, K * E, code, a_batch=GB * ld, b_batch=E, c_batch=E, batch=K)
            _hip.gemm_tn(P(W), P(pa), P(out_t), GB, GB, E, ld, K * E, K * E, code, a_batch=GB * ld, b_batch=E, c_batch=E, batch=K)

    def _gp_coeff(self, g, second, all_timesteps):
        nb, R, ld = self._gp_sizes(all_timesteps)
        P = _hip.ptr
        if second:
            _hip.call("cpc_gp_score_coeff", P(g.S), P(g.St1), P(g.St2), P(g.W2), P(g.W2T), nb, R, R, ld, ld, 1)
            names = ("W2", "W2T")
        else:
            _hip.call("cpc_gp_score_coeff", P(g.S), None, None, P(g.W1), P(g.W1T), nb, R, R, ld, ld, 0)
            names = ("W1", "W1T")
        for name in names:
            dst, src = getattr(g, name + "s"), getattr(g, name)
            if dst is not src:
                dst.copy_(src)

    def gp_softplus_seed(self, all_timesteps):
        """Pass 1: (adjoint of the local predictions [B K E], adjoint of the local targets [B, K, E]) of the summed global scores."""
        e = self.eng
        g = self._gp_state(all_timesteps)
        n = e.B * e.K * e.E
        self._gp_gather(e.pred, e.act[-1], self.pred_all, self.targ_all, self.local_targ)
        self._gp_scores(self.pred_all, self.targ_all, g.S, all_timesteps)
        self._gp_coeff(g, False, all_timesteps)
        self._gp_contract(g.W1s, g.W1Ts, self.pred_all, self.targ_all, g.out_p, g.out_t, all_timesteps)
        lo = self.rank * n
        return g.out_p[lo:lo + n], g.out_t[lo:lo + n].view(e.B, e.K, e.E)

    def gp_softplus_second(self, pred_t, top_t, all_timesteps):
        """Last pass: what the local predictions / targets gain from the tangent pass, nu_p = W1 (tangent targets) + W2 targets and
        nu_t = W1^T (tangent predictions) + W2^T predictions over the global batch (pred_all / targ_all / S are pass 1's)."""
        e = self.eng
        g = self._gp_state(all_timesteps)
        n = e.B * e.K * e.E
        self._gp_gather(pred_t, top_t, g.pred_t_all, g.targ_t_all, g.local)
        self._gp_scores(g.pred_t_all, self.targ_all, g.St1, all_timesteps)
        self._gp_scores(self.pred_all, g.targ_t_all, g.St2, all_timesteps)
        self._gp_coeff(g, True, all_timesteps)
        self._gp_contract(g.W1s, g.W1Ts, g.pred_t_all, g.targ_t_all, g.out_p, g.out_t, all_timesteps)
        self._gp_contract(g.W2s, g.W2Ts, self.pred_all, self.targ_all, g.out_p2, g.out_t2, all_timesteps)
        lo = self.rank * n
        add_p = g.out_p[lo:lo + n].view(e.B, e.K, e.E) + g.out_p2[lo:lo + n].view(e.B, e.K, e.E)
        add_t = g.out_t[lo:lo + n].view(e.B, e.K, e.E) + g.out_t2[lo:lo + n].view(e.B, e.K, e.E)
        return add_p, add_t

    def forward_backward(self, softplus: bool, regularization: float, all_timesteps: bool = False):
        e, dist = self.eng, self.dist
        code, B, E, K, GB, ld = e.code, e.B, e.E, e.K, self.GB, self.ld
        Ltop, T = e.geo.alloc[-1], e.T
        top, dtop = e.act[-1].view(B, Ltop, E), e.dact[-1].view(B, Ltop, E)
        n = B * K * E
        self.local_targ.view(B, K, E).copy_(top[:, T - K:T, :])
        dist.all_gather([self.pred_all[r * n:(r + 1) * n] for r in range(self.world)], e.pred)
        dist.all_gather([self.targ_all[r * n:(r + 1) * n] for r in range(self.world)], self.local_targ)
        P = _hip.ptr
        if all_timesteps:
            if e.fused_scores_ok(rows=B * K, cols=B * K) and os.environ.get("CPC_SCORE_STRIPS", "1") != "0":
                self._all_timesteps_strips(softplus, regularization)
            else:
                self._all_timesteps(softplus, regularization)
            lo = self.rank * n
            e.dpred.copy_(self.dpred_all[lo:lo + n])
            dtop[:, T - K:T, :].copy_(self.dtarg_all[lo:lo + n].view(B, K, E))
            return
        _hip.gemm_nt(P(self.pred_all), P(self.targ_all), P(self.S), GB, GB, E, K * E, K * E, ld, code, a_batch=E, b_batch=E,
                     c_batch=GB * ld, batch=K, flags=_hip.GEMM_OUT_F32)
        _hip.call("cpc_nce_loss", P(self.S), P(self.dS), P(self.dST), P(self.out), P(self.ws), GB, K, ld, 1 if softplus else 0,
                  C.c_float(regularization), code)
        _hip.gemm_tn(P(self.dST), P(self.targ_all), P(self.dpred_all), GB, GB, E, ld, K * E, K * E, code, a_batch=GB * ld, b_batch=E,
                     c_batch=E, batch=K)
        _hip.gemm_tn(P(self.dS), P(self.pred_all), P(self.dtarg_all), GB, GB, E, ld, K * E, K * E, code, a_batch=GB * ld, b_batch=E,
                     c_batch=E, batch=K)
        lo = self.rank * n
        e.dpred.copy_(self.dpred_all[lo:lo + n])
        dtop[:, T - K:T, :].copy_(self.dtarg_all[lo:lo + n].view(B, K, E))


class GradAllReduce:
    """Data-parallel gradient exchange: ONE sum over ranks of the model's flat f32 gradient buffer per step (RCCL over
    xGMI with backend "nccl"), issued in pieces as the backward pass completes them: encoder layers >= 3 + GRU + predictor
    (63 % of the bytes) travel while layers 2 and 1 are still being differentiated, layer 2 (28 %) under the last
    data-gradient GEMM, and only layer 1's 22 KB is left for finish().
    The mean is taken by FusedAdam's ``grad_scale = 1 / world``."""

    def __init__(self, model, optimizer=None, grad_scale=None):
        import torch.distributed as dist
        self.dist = dist
        self.model = model
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.pending = []
        self.split = None
        # With a FusedAdam given, a piece's parameters are updated (and their operand copies for the next step rebuilt, see
        # CPCEngine.prepare_ahead) as soon as its reduction has finished: at the next hook call, on the side stream beside the
        # remaining backward GEMMs, instead of after the whole backward pass.  grad_scale: 1 / world for the mean of the shard
        # gradients (default), 1 where the shard gradients add up (global negatives).
        self.optimizer = optimizer
        self.grad_scale = (1.0 / self.world) if grad_scale is None else float(grad_scale)
        self._plan, self.last_plan = [], []       # flat-gradient ranges in the order their reductions were issued (this step / last finished step)

    def describe(self):
        """The bucket plan of the last finished step, for a run's report (bench.py `data_parallel`): every range of the flat gradient buffer
        with its bytes and where its reduction is issued; ``covers_once`` says that the ranges tile the buffer exactly once."""
        flat = self.model._flat_grad
        esz, total = flat.element_size(), flat.numel()
        plan = list(self.last_plan)
        edges, ok = 0, True
        for lo, hi, _ in sorted(plan):
            ok = ok and lo == edges and hi > lo
            edges = hi
        return {"grad_bytes": total * esz, "world": self.world,
                "buckets": [{"lo": lo, "hi": hi, "bytes": (hi - lo) * esz, "issued": where} for lo, hi, where in plan],
                "covers_once": bool(ok and edges == total)}

    def reduce_flag(self, nce_out):
        """Pass as ``after_loss``: the sticky NaN flag (nce_out[6]) becomes the MAXIMUM over the ranks (and the step's indicator
        nce_out[5] with it), asynchronously, before any Adam piece of this step reads it — so that with per-GPU negatives, where
        every rank has its own loss, all ranks skip the same update and all of them leave train() at the same step."""
        if self.world > 1:
            self.flag_work = self.dist.all_reduce(nce_out[5:7], op=self.dist.ReduceOp.MAX, async_op=True)

    def _apply_finished(self):
        """Adam on every piece whose all-reduce has been issued: the current stream waits for the reduction first."""
        work = getattr(self, "flag_work", None)
        if work is not None:
            work.wait()
            self.flag_work = None
        for work, lo, hi in self.pending:
            work.wait()
            if self.optimizer is not None:
                self.optimizer.update_range(lo, hi, self.grad_scale)
        self.pending = []

    def hook(self, lo, hi):
        """Pass as ``grad_ready_hook``: starts the asynchronous all-reduce of flat_grad[lo:hi]."""
        if self.optimizer is not None:
            self._apply_finished()          # pieces issued at earlier hook calls (reduced while the backward pass went on)
        self.split = lo if self.split is None else min(self.split, lo)
        self._plan.append((lo, hi, "backward hook (overlapped)"))
        self.pending.append((self.dist.all_reduce(self.model._flat_grad[lo:hi], async_op=True), lo, hi))

    def finish(self):
        """Reduces what the hook has not covered and makes the current stream wait for all pieces.  With an optimizer attached
        the caller still calls ``optimizer.step(grad_scale)`` afterwards: it updates what is left (the head of the buffer)."""
        flat = self.model._flat_grad
        rest = flat if self.split is None else flat[:self.split]
        self._apply_finished()
        if rest.numel():
            self._plan.append((0, rest.numel(), "finish() (after the backward pass)"))
            self.dist.all_reduce(rest, async_op=True).wait()
        self.split = None
        self.last_plan, self._plan = self._plan, []


class FusedAdam:
    """torch.optim.Adam (default betas / eps, no weight decay) over the model's flat f32 parameter buffer as one kernel."""

    def __init__(self, model, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, device_step: bool = False):
        self.model = model
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        flat = model._flat_param
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.t = 0
        self._done_lo = None
        self._piece_scale = 1.0
        # after_update(lo, hi, final): called right after flat_param[lo:hi) was updated; set it to the engine's prepare_ahead in
        # single-process training so that the operand copies of the next step are rebuilt off the critical path
        self.after_update = None
        # skip_flag: one-element device tensor (CPCEngine.nan_flag()); while it is non-zero every update is a no-op on the device
        # — the reference's NaN guard returns before backward() / optimizer.step() (contrastive_estimation_training.py:124-133)
        self.skip_flag = None
        # device_step: the step count lives on the device (cpc_adam_dev), so the call's arguments never change and the step
        # can be part of a captured hipGraph
        self.state = torch.zeros(4, device=flat.device, dtype=torch.float32) if device_step else None

    def hook(self, lo, hi):
        """Single-process use as ``grad_ready_hook``: updates flat_param[lo:hi] as soon as the backward pass reports that range
        of the gradient final (the engine calls it from its side stream, beside the remaining data-gradient GEMMs); step()
        then updates what is left.  Nothing the backward pass still runs reads the f32 master parameters of a finished range
        (the GEMMs use the storage-dtype operand copies made at the start of the step).  Not for data-parallel runs, where the
        update has to follow the all-reduce."""
        if self.state is not None:
            raise ValueError("FusedAdam.hook needs the host-side step count (device_step=False)")
        self.update_range(lo, hi, 1.0)

    def update_range(self, lo, hi, grad_scale=1.0):
        """Data-parallel use (GradAllReduce): Adam on flat_param[lo:hi) once that range of the gradient has been reduced over
        the ranks, with the step count of the step in progress; step(grad_scale) then updates the head of the buffer."""
        if self.state is not None:
            raise ValueError("piecewise updates need the host-side step count (device_step=False)")
        self._done_lo = lo if self._done_lo is None else min(self._done_lo, lo)
        self._piece_scale = float(grad_scale)
        self._launch(lo, hi, self.t + 1, grad_scale)
        if self.after_update is not None:
            self.after_update(lo, hi, False)

    def _launch(self, lo, hi, t, grad_scale):
        flat, grad = self.model._flat_param, self.model._flat_grad
        if hi <= lo:
            return
        self.model._raw_updates = getattr(self.model, "_raw_updates", 0) + 1
        _hip.call("cpc_adam", _hip.ptr(flat, lo), _hip.ptr(grad, lo), _hip.ptr(self.m, lo), _hip.ptr(self.v, lo), C.c_longlong(hi - lo),
                  C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps), t,
                  C.c_float(grad_scale), _hip.ptr(self.skip_flag))

    def step(self, grad_scale: float = 1.0):
        self.t += 1
        flat, grad = self.model._flat_param, self.model._flat_grad
        if self._done_lo is not None:          # ranges [done_lo, end) were updated by hook() during the backward pass
            hi, self._done_lo = self._done_lo, None
            if float(grad_scale) != self._piece_scale:
                raise ValueError(f"the pieces of this step were updated with grad_scale {self._piece_scale}, step() got {grad_scale}")
            self._launch(0, hi, self.t, grad_scale)
            if self.after_update is not None:
                self.after_update(0, hi, True)
            return
        self.model._raw_updates = getattr(self.model, "_raw_updates", 0) + 1
        if self.state is not None:
            _hip.call("cpc_adam_dev", _hip.ptr(flat), _hip.ptr(grad), _hip.ptr(self.m), _hip.ptr(self.v), C.c_longlong(flat.numel()),
                      C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                      _hip.ptr(self.state), C.c_float(grad_scale), _hip.ptr(self.skip_flag))
            return
        _hip.call("cpc_adam", _hip.ptr(flat), _hip.ptr(grad), _hip.ptr(self.m), _hip.ptr(self.v), C.c_longlong(flat.numel()),
                  C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps), self.t,
                  C.c_float(grad_scale), _hip.ptr(self.skip_flag))
        if self.after_update is not None:
            self.after_update(0, flat.numel(), True)


class GraphedStep:
    """One whole train step — forward, loss, backward, fused Adam — captured once into a hipGraph and replayed with one call.
    Measured on MI355X at B = 256: 5.04 ms per step replayed (single stream) against 4.9-5.0 ms for the eager step with its side
    stream and 5.5 ms for the eager step on one stream — the graph closes the same gaps between short kernels that the side
    stream hides.  The small-batch step is bound by latency-bound kernels (the GRU's 100 sequential steps each way), not by
    launch overhead.  Kept as an option for hosts with slower launch paths.  Requirements: single process (no collective inside the graph), no host-side per-step state
    (dropout seeds)."""

    def __init__(self, eng, opt, softplus: bool, regularization: float, all_timesteps: bool = False):
        if opt.state is None:
            raise ValueError("GraphedStep needs FusedAdam(device_step=True)")
        ctx = eng.ctx
        if getattr(getattr(ctx, "ar", None), "dropout", 0.0) and ctx.ar.training:
            raise NotImplementedError("a step with dropout draws a new host-side seed per step and cannot be replayed from a graph")
        self.eng, self.opt = eng, opt
        shape = getattr(eng, "in_shape", None) or (eng.B, eng.L)
        self.x = torch.zeros(*shape, device=eng.device, dtype=torch.float32)
        args = dict(softplus=softplus, regularization=regularization, all_timesteps=all_timesteps)
        # no warm-up run: nothing here initialises lazily on first use except buffers, which the capture allocates from the
        # graph's own pool — and a real step on a dummy batch would move BatchNorm's running statistics
        # captured on ONE stream: a capture with the side-stream forks replays slower (6.4 vs 5.0 ms at B = 256), and the graph
        # itself closes the launch gaps the side stream hides in the eager step
        eng.use_aux = False
        eng._ahead_token = None            # the captured step always rebuilds its operand copies
        opt.skip_flag = eng.nan_flag()     # NaN guard: part of the captured update
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = eng.loss_and_grads(self.x, **args)
            opt.step()
        opt.t -= 1                                        # the capture ran step() once on the host side only

    def __call__(self, batch):
        self.x.copy_(batch, non_blocking=True)
        self.graph.replay()
        self.opt.t += 1
        self.eng.model._raw_updates = getattr(self.eng.model, "_raw_updates", 0) + 1
        return self.out


def smoke_check(device):
    """One tiny train step on the GPU, checked against the CPU oracle (used by __graft_entry__.smoke)."""
    from oracle import cpc_oracle as O
    from .audio_model import AudioEncoder, AudioGRUModel, AudioPredictiveCodingModel
    torch.manual_seed(0)
    enc = AudioEncoder({'strides': [5, 4, 2, 2, 2], 'kernel_sizes': [10, 8, 4, 4, 4], 'channel_count': [64] * 5, 'bias': True})
    ar = AudioGRUModel(input_size=64, hidden_size=64)
    model = AudioPredictiveCodingModel(enc, ar, enc_size=64, ar_size=64, visible_steps=10, prediction_steps=4,
                                       compute_dtype="fp32")
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("weight") and "encoder" in name:
                p.mul_(3.0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(device)
    B, L = 8, 465 + 14 * 160
    x = torch.randn(B, L) * 0.5
    eng = model.engine(B, L)
    out = eng.loss_and_grads(x.to(device), softplus=True, regularization=1.0)
    torch.cuda.synchronize()
    tr = O.OracleTrainer(params, 10, 4, score="softplus", regularization=1.0)
    loss, smax, grads = tr.loss_and_grads(x)
    got = float(out[0])
    assert abs(got - float(loss)) < 1e-3 * abs(float(loss)), (got, float(loss))
    for name, gref in grads.items():
        gdev = model._grad[name].detach().cpu()
        err = (gdev - gref).abs().max().item() / (gref.abs().max().item() + 1e-12)
        assert err < 1e-3, (name, err)
    print(f"smoke ok: loss {got:.6f} (oracle {float(loss):.6f})")
