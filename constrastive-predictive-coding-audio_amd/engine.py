"""Host-side plan of the CPC train step: buffers in HBM + the launch sequence over the C ABI.

One ``CPCEngine`` = one (batch size, clip length, storage dtype) instance of the path
    AudioEncoder.forward -> AudioGRUModel.forward -> prediction_model -> score/InfoNCE -> backward -> Adam
for a model built from AudioEncoder + AudioGRUModel (reference: audio_model.py:193-211 and
contrastive_estimation_training.py:97-162).  PyTorch is used for device memory and streams only; every
arithmetic step is a HIP kernel behind libcpc_hip.so.

Data layout (all per GPU, resident for the life of the engine):
  act[l], dact[l]   storage dtype, channels-last [B][L_alloc[l]][C_l], zero pad rows, zero guards front/back
  L_alloc[l-1] = stride_l * L_alloc[l]  so that a strided conv is a GEMM whose A rows overlap uniformly
  flat parameters / gradients / Adam moments: f32, one contiguous buffer each (owned by the model)
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import torch

from . import _hip


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
        self.H = int(ar.hidden_size)
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
        if model.enc_size != self.E or model.ar_size != self.H or ar.input_size != self.E:
            raise ValueError("enc_size / ar_size do not match the encoder and autoregressive model")
        self._check_supported()
        model._flatten_parameters(self.device)
        self._alloc()

    # ------------------------------------------------------------------------------------------ setup
    def _check_supported(self):
        ch = 8 if self.dt == torch.bfloat16 else 4
        for l in range(1, self.n):
            if (self.kernels[l] * self.channels[l - 1]) % ch or self.channels[l] % 8:
                raise NotImplementedError("HIP conv path needs channel counts that are multiples of 8")
        c0 = self.channels[0]
        if c0 % 8 or 256 % (c0 // 8) or self.kernels[0] > 16 or self.strides[0] > 8:
            raise NotImplementedError("HIP layer-1 kernel: channels in {8,16,...,2048}, kernel <= 16, stride <= 8")
        if self.H % 16 or self.H % (4 * ch) or self.H > 256:
            raise NotImplementedError("HIP GRU kernel: hidden size must be a multiple of 32 and <= 256")
        if self.E % ch:
            raise NotImplementedError("enc_size must be a multiple of 8")

    def _buf(self, rows: int, cols: int):
        """Zeroed [rows][cols] storage-dtype buffer with 16 guard rows on both sides; returns (full, view, guard_elems)."""
        guard = 16 * cols
        full = torch.zeros(guard + rows * cols + guard, device=self.device, dtype=self.dt)
        return full, full[guard:guard + rows * cols], guard

    def _alloc(self):
        dev, dt, f32 = self.device, self.dt, torch.float32
        B, V, H, E, K, n = self.B, self.V, self.H, self.E, self.K, self.n
        La = self.geo.alloc
        self.act, self.dact = [], []
        self._keep = []
        for l in range(n):
            for store in (self.act, self.dact):
                full, view, _ = self._buf(B * La[l], self.channels[l])
                self._keep.append(full)
                store.append(view)
        # weight operand layouts (storage dtype)
        self.w_fwd: List[Optional[torch.Tensor]] = [None] * n
        self.w_dgrad: List[Optional[torch.Tensor]] = [None] * n
        for l in range(1, n):
            cin, cout, kw, s = self.channels[l - 1], self.channels[l], self.kernels[l], self.strides[l]
            self.w_fwd[l] = torch.empty(cout * kw * cin, device=dev, dtype=dt)
            self.w_dgrad[l] = torch.empty(s * cin * self.geo.taps[l] * cout, device=dev, dtype=dt)
        self.w_ih = torch.empty(3 * H * E, device=dev, dtype=dt)          # [3H][E]
        self.w_ih_t = torch.empty(E * 3 * H, device=dev, dtype=dt)        # [E][3H]
        self.w_hh_frag = torch.empty(3 * H * H, device=dev, dtype=dt)
        self.w_hh_t_frag = torch.empty(3 * H * H, device=dev, dtype=dt)
        self.w_p = torch.empty(K * E * H, device=dev, dtype=dt)           # [K*E][H]
        self.w_p_t = torch.empty(H * K * E, device=dev, dtype=dt)         # [H][K*E]
        # GRU / predictor / loss state
        self.Gi = torch.empty(B * V * 3 * H, device=dev, dtype=dt)
        self.Hall = torch.empty(B * (V + 1) * H, device=dev, dtype=dt)
        self.tape = torch.zeros(max(1, int(_hip.lib().cpc_gru_tape_elems(B, max(V, 1), H, self.code))), device=dev, dtype=dt)
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
        self.dG = torch.empty(B * V * 4 * H, device=dev, dtype=dt)       # [dr | du | dn | dn*r] per (item, step)
        # split-reduction workspace (f32 slabs), sized for the largest user
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
        need.append(self.colsum_blocks * max(max(self.channels), 4 * H))
        self.split_ih = self._pick_split(3 * H, E, B * V)
        self.split_hh = self._pick_split(3 * H, H, B * V)
        need.append(self.split_ih * 3 * H * E)
        need.append(self.split_hh * 3 * H * H)
        self.slabs = torch.empty(max(need), device=dev, dtype=f32)

    def _pick_split(self, I, J, M):
        big = self.dt == torch.bfloat16 and I >= 256 and J >= 256       # 256x256 tiles, one workgroup per CU
        tile = 256 if big else 128
        tiles = _ceil_div(I, tile) * _ceil_div(J, tile)
        blk = 64 if self.dt == torch.bfloat16 else 32
        want = _ceil_div(256 if big else 512, tiles)
        return max(1, min(want, _ceil_div(M, 8 * blk), 64))

    def _chunk(self, M, nsplit):
        blk = 64 if self.dt == torch.bfloat16 else 32
        return _ceil_div(_ceil_div(M, nsplit), blk) * blk

    # ---------------------------------------------------------------------------------- weight layouts
    def prepare_weights(self):
        """f32 master parameters (reference state_dict shapes) -> storage-dtype GEMM operand layouts."""
        p, code = self.model._param, self.code
        for l in range(1, self.n):
            _hip.call("cpc_conv_w_prep", _hip.ptr(p[f"encoder.layers.{l}.weight"]), _hip.ptr(self.w_fwd[l]),
                      _hip.ptr(self.w_dgrad[l]), self.channels[l], self.channels[l - 1], self.kernels[l], self.strides[l], code)
        H, E, K = self.H, self.E, self.K
        if K == 0:          # encoder-only engine (stand-alone AudioEncoder call)
            return
        w_ih, w_hh = p["autoregressive_model.gruCell.weight_ih"], p["autoregressive_model.gruCell.weight_hh"]
        w_p = p["prediction_model.weight"]
        _hip.call("cpc_cast2d", _hip.ptr(w_ih), _hip.ptr(self.w_ih), 3 * H, E, E, 1, code)
        _hip.call("cpc_cast2d", _hip.ptr(w_ih), _hip.ptr(self.w_ih_t), E, 3 * H, 1, E, code)
        _hip.call("cpc_prep_frag", _hip.ptr(w_hh), _hip.ptr(self.w_hh_frag), 3 * H, H, H, 0, code)
        _hip.call("cpc_prep_frag", _hip.ptr(w_hh), _hip.ptr(self.w_hh_t_frag), H, 3 * H, H, 1, code)
        _hip.call("cpc_cast2d", _hip.ptr(w_p), _hip.ptr(self.w_p), K * E, H, H, 1, code)
        _hip.call("cpc_cast2d", _hip.ptr(w_p), _hip.ptr(self.w_p_t), H, K * E, 1, H, code)

    # ------------------------------------------------------------------------------------------ forward
    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("the CPC hot path runs on the GPU only (no CPU fallback): move the batch to the device")
        if x.dtype != torch.float32 or tuple(x.shape) != (self.B, self.L) or not x.is_contiguous():
            raise ValueError(f"expected a contiguous float32 batch of shape ({self.B}, {self.L}), got {tuple(x.shape)} {x.dtype}")

    def encoder_forward(self, x):
        """AudioEncoder.forward (audio_model.py:36-44): relu(conv) x (n-1), then a bare conv."""
        self._check_input(x)
        p, code, B, La, Lv = self.model._param, self.code, self.B, self.geo.alloc, self.geo.valid
        _hip.call("cpc_conv1_fwd", _hip.ptr(x, self.x_off), _hip.ptr(p["encoder.layers.0.weight"]), _hip.ptr(p.get("encoder.layers.0.bias")),
                  _hip.ptr(self.act[0]), B, self.channels[0], self.strides[0], self.kernels[0], self.L, Lv[0], La[0], code)
        for l in range(1, self.n):
            _hip.call("cpc_conv_fwd", _hip.ptr(self.act[l - 1]), _hip.ptr(self.w_fwd[l]), _hip.ptr(p.get(f"encoder.layers.{l}.bias")),
                      _hip.ptr(self.act[l]), B, self.channels[l - 1], self.channels[l], self.kernels[l], self.strides[l],
                      La[l], Lv[l], 1 if l < self.n - 1 else 0, code,
                      key="gemm_nt" + _hip._variant(code, 0, _hip.nt_tile(code, B * La[l], self.channels[l], self.kernels[l] * self.channels[l - 1])),
                      work=2.0 * B * La[l] * self.channels[l] * self.kernels[l] * self.channels[l - 1],
                      shape=("fwd", B * La[l], self.channels[l], self.kernels[l] * self.channels[l - 1]))

    def context_forward(self):
        """AudioGRUModel.forward over z = frames [T-K-V, T-K) (audio_model.py:198-202, :66-77) + prediction_model (:208)."""
        p, code, B, V, H, E, K = self.model._param, self.code, self.B, self.V, self.H, self.E, self.K
        Ltop = self.geo.alloc[-1]
        t0 = self.T - K - V
        top = self.act[-1]
        _hip.gemm_nt(_hip.ptr(top, t0 * E), _hip.ptr(self.w_ih), _hip.ptr(self.Gi), B * V, 3 * H, E, E, E, 3 * H, code,
                     bias=_hip.ptr(p.get("autoregressive_model.gruCell.bias_ih")), a_rpi=V, a_item=Ltop * E)
        _hip.call("cpc_gru_fwd", _hip.ptr(self.Gi), _hip.ptr(self.w_hh_frag), _hip.ptr(p.get("autoregressive_model.gruCell.bias_hh")),
                  _hip.ptr(self.Hall), _hip.ptr(self.tape), _hip.ptr(self.c), B, V, H, code)
        _hip.gemm_nt(_hip.ptr(self.Hall, V * H), _hip.ptr(self.w_p), _hip.ptr(self.pred), B, K * E, H, H, H, K * E, code,
                     a_rpi=1, a_item=(V + 1) * H)

    def forward(self, x):
        self.prepare_weights()
        self.encoder_forward(x)
        self.context_forward()

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
        return pred, targets, z, self.c.clone()

    # ------------------------------------------------------------------------------------------ loss
    def nce_forward_backward(self, softplus: bool, regularization: float):
        """Equal-step scores, InfoNCE loss + regulariser, and d loss / d (predicted_z, targets).

        contrastive_estimation_training.py:106-122,141 with score_over_all_timesteps=False.  Only the K diagonal
        (B x B) blocks of the reference's (B K)^2 score tensor are ever formed (12x fewer FLOPs)."""
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T, ld = self.geo.alloc[-1], self.T, self.ldS
        top, dtop = self.act[-1], self.dact[-1]
        _hip.gemm_nt(_hip.ptr(self.pred), _hip.ptr(top, (T - K) * E), _hip.ptr(self.S), B, B, E, K * E, Ltop * E, ld, code,
                     a_batch=E, b_batch=E, c_batch=B * ld, batch=K, flags=_hip.GEMM_OUT_F32)
        _hip.call("cpc_nce_loss", _hip.ptr(self.S), _hip.ptr(self.dS), _hip.ptr(self.dST), _hip.ptr(self.nce_out),
                  _hip.ptr(self.nce_ws), B, K, ld, 1 if softplus else 0, C.c_float(regularization), code)
        # d predicted_z[b][k][:] = sum_b' dS[k][b][b'] * targets[b'][k][:]
        _hip.gemm_tn(_hip.ptr(self.dST), _hip.ptr(top, (T - K) * E), _hip.ptr(self.dpred), B, B, E, ld, Ltop * E, K * E, code,
                     a_batch=B * ld, b_batch=E, c_batch=E, batch=K)
        # d targets[b'][k][:] = sum_b dS[k][b][b'] * predicted_z[b][k][:]   -> rows T-K+k of the top-layer gradient
        _hip.gemm_tn(_hip.ptr(self.dS), _hip.ptr(self.pred), _hip.ptr(dtop, (T - K) * E), B, B, E, ld, K * E, Ltop * E, code,
                     a_batch=B * ld, b_batch=E, c_batch=E, batch=K)

    def nce_all_forward_backward(self, softplus: bool, regularization: float):
        """score_over_all_timesteps=True (contrastive_estimation_training.py:108-114, :141): the full (B K) x (B K) score
        matrix (and its transpose, as a second tiny GEMM, so that both gradient layouts are written coalesced), the
        log-sum-exp over ALL predictions for every (target item, step), and d loss / d (predicted_z, targets)."""
        code, B, E, K = self.code, self.B, self.E, self.K
        Ltop, T = self.geo.alloc[-1], self.T
        top, dtop = self.act[-1], self.dact[-1]
        R = B * K
        ld = _ceil_div(R, 8) * 8
        if getattr(self, "S_all", None) is None:
            f32 = torch.float32
            self.S_all = torch.zeros(R * ld, device=self.device, dtype=f32)
            self.ST_all = torch.zeros(R * ld, device=self.device, dtype=f32)
            self.dS_all = torch.zeros(R * ld, device=self.device, dtype=self.dt)
            self.dST_all = torch.zeros(R * ld, device=self.device, dtype=self.dt)
            self.nce_all_ws = torch.empty(int(_hip.lib().cpc_nce_all_workspace_floats(B, K)), device=self.device, dtype=f32)
        tg = (T - K) * E
        _hip.gemm_nt(_hip.ptr(self.pred), _hip.ptr(top, tg), _hip.ptr(self.S_all), R, R, E, E, E, ld, code,
                     b_rpi=K, b_item=Ltop * E, flags=_hip.GEMM_OUT_F32)
        _hip.gemm_nt(_hip.ptr(top, tg), _hip.ptr(self.pred), _hip.ptr(self.ST_all), R, R, E, E, E, ld, code,
                     a_rpi=K, a_item=Ltop * E, flags=_hip.GEMM_OUT_F32)
        _hip.call("cpc_nce_loss_all", _hip.ptr(self.S_all), _hip.ptr(self.ST_all), _hip.ptr(self.dS_all), _hip.ptr(self.dST_all),
                  _hip.ptr(self.nce_out), _hip.ptr(self.nce_all_ws), B, K, ld, 1 if softplus else 0, C.c_float(regularization), code)
        # d predicted_z[(b,k)][:] = sum_c dS[(b,k)][c] * targets[c][:]
        _hip.gemm_tn(_hip.ptr(self.dST_all), _hip.ptr(top, tg), _hip.ptr(self.dpred), R, R, E, ld, E, E, code, b_rpi=K, b_item=Ltop * E)
        # d targets[c][:] = sum_r dS[r][c] * predicted_z[r][:]   -> rows T-K+k' of item b' of the top-layer gradient
        _hip.gemm_tn(_hip.ptr(self.dS_all), _hip.ptr(self.pred), _hip.ptr(dtop, tg), R, R, E, ld, E, E, code, c_rpi=K, c_item=Ltop * E)

    # ------------------------------------------------------------------------------------------ backward
    def _tn_to_grad(self, A, B_, grad, M, I, J, lda, ldb, nsplit, grad_offset=0, **kw):
        """grad[grad_offset + i*J + j] = sum_m A[m][i] B[m][j] via f32 slabs + deterministic reduction."""
        code = self.code
        chunk = self._chunk(M, nsplit)
        _hip.gemm_tn(A, B_, _hip.ptr(self.slabs), M, I, J, lda, ldb, J, code, nsplit=nsplit, m_chunk=chunk, slab_stride=I * J,
                     flags=_hip.GEMM_OUT_F32, **kw)
        _hip.call("cpc_reduce_slabs", _hip.ptr(self.slabs), _hip.ptr(grad, grad_offset), I, J, nsplit, I * J, 1, 1, J, 0)

    def _colsum_to_grad(self, X, grad, M, N):
        nb = min(self.colsum_blocks, max(1, M // 64))
        _hip.call("cpc_colsum", X, _hip.ptr(self.slabs), M, N, N, nb, self.code)
        _hip.call("cpc_reduce_slabs", _hip.ptr(self.slabs), _hip.ptr(grad), 1, N, nb, N, 1, 1, 0, 0)

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
        top, dtop = self.act[-1], self.dact[-1]
        # predictor: dW_p = dpred^T c ;  dc = dpred W_p
        _hip.gemm_tn(_hip.ptr(self.dpred), _hip.ptr(self.Hall, V * H), _hip.ptr(g["prediction_model.weight"]), B, K * E, H,
                     K * E, H, H, code, b_rpi=1, b_item=(V + 1) * H, flags=_hip.GEMM_OUT_F32)
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
        _hip.call("cpc_gru_bwd", _hip.ptr(self.dc), _hip.ptr(self.tape), _hip.ptr(self.w_hh_t_frag), _hip.ptr(self.dG), B, V, H, code)
        # dG[b][t] = [dr | du | dn | dn*r]: columns [0,3H) are the gradient of the input-projection term, columns [0,2H) and
        # [3H,4H) that of the recurrent term
        g_ih, g_hh = g["autoregressive_model.gruCell.weight_ih"], g["autoregressive_model.gruCell.weight_hh"]
        self._tn_to_grad(_hip.ptr(self.dG), _hip.ptr(top, t0 * E), g_ih, B * V, 3 * H, E, 4 * H, E, self.split_ih,
                         b_rpi=V, b_item=Ltop * E)
        self._tn_to_grad(_hip.ptr(self.dG), _hip.ptr(self.Hall), g_hh, B * V, 2 * H, H, 4 * H, H, self.split_hh,
                         b_rpi=V, b_item=(V + 1) * H)
        self._tn_to_grad(_hip.ptr(self.dG, 3 * H), _hip.ptr(self.Hall), g_hh, B * V, H, H, 4 * H, H, self.split_hh,
                         b_rpi=V, b_item=(V + 1) * H, grad_offset=2 * H * H)
        if "autoregressive_model.gruCell.bias_ih" in g:
            g_bi, g_bh = g["autoregressive_model.gruCell.bias_ih"], g["autoregressive_model.gruCell.bias_hh"]
            M = B * V
            nb = min(self.colsum_blocks, max(1, M // 64))
            _hip.call("cpc_colsum", _hip.ptr(self.dG), _hip.ptr(self.slabs), M, 4 * H, 4 * H, nb, code)
            sl = self.slabs
            _hip.call("cpc_reduce_slabs", _hip.ptr(sl), _hip.ptr(g_bi), 1, 3 * H, nb, 4 * H, 1, 1, 0, 0)
            _hip.call("cpc_reduce_slabs", _hip.ptr(sl), _hip.ptr(g_bh), 1, 2 * H, nb, 4 * H, 1, 1, 0, 0)
            _hip.call("cpc_reduce_slabs", _hip.ptr(sl, 3 * H), _hip.ptr(g_bh, 2 * H), 1, H, nb, 4 * H, 1, 1, 0, 0)
        # dz -> rows [t0, t0+V) of the top-layer gradient
        _hip.gemm_nt(_hip.ptr(self.dG), _hip.ptr(self.w_ih_t), _hip.ptr(dtop, t0 * E), B * V, E, 3 * H, 4 * H, 3 * H, E, code,
                     c_rpi=V, c_item=Ltop * E, c_valid=V)
        if add_dz is not None:
            dtop.view(B, Ltop, E)[:, t0:t0 + V, :].add_(add_dz.transpose(1, 2))
        # encoder, top layer down to layer 2
        for l in range(n - 1, 0, -1):
            cin, cout, kw, s = self.channels[l - 1], self.channels[l], self.kernels[l], self.strides[l]
            bname = f"encoder.layers.{l}.bias"
            if bname in g:
                self._colsum_to_grad(_hip.ptr(self.dact[l]), g[bname], B * La[l], cout)
            flops = 2.0 * B * La[l] * cout * kw * cin
            _hip.call("cpc_conv_wgrad", _hip.ptr(self.act[l - 1]), _hip.ptr(self.dact[l]), _hip.ptr(self.slabs), B, cin, cout, kw, s,
                      La[l], self.nsplit[l], code,
                      key="gemm_tn" + _hip._variant(code, _hip.GEMM_OUT_F32, _hip.tn_tile(code, B * La[l], kw * cin, cout, self.nsplit[l],
                                                                                             self._chunk(B * La[l], self.nsplit[l]))),
                      work=flops,
                      shape=("wgrad", B * La[l], kw * cin, cout, self.nsplit[l]))
            _hip.call("cpc_reduce_conv_w", _hip.ptr(self.slabs), _hip.ptr(g[f"encoder.layers.{l}.weight"]), cin, cout, kw,
                      self.nsplit[l], kw * cin * cout)
            if grad_ready_hook is not None and l == 2 and n > 2:
                # everything from encoder layer index 2 upwards (+ GRU, predictor: later in the flat buffer) is final
                lo = self.model._offset["encoder.layers.2.weight"]
                grad_ready_hook(lo, self.model._flat_grad.numel())
            _hip.call("cpc_conv_dgrad", _hip.ptr(self.dact[l]), _hip.ptr(self.w_dgrad[l]), _hip.ptr(self.act[l - 1]),
                      _hip.ptr(self.dact[l - 1]), B, cin, cout, kw, s, La[l], Lv[l - 1], code,
                      key="gemm_nt" + _hip._variant(code, 0, _hip.nt_tile(code, B * La[l], s * cin, self.geo.taps[l] * cout)),
                      work=2.0 * B * La[l] * s * cin * self.geo.taps[l] * cout,
                      shape=("dgrad", B * La[l], s * cin, self.geo.taps[l] * cout))
        # layer 1
        c0, k0, s0 = self.channels[0], self.kernels[0], self.strides[0]
        nblk, nbb = self.c1_blocks, self.c1_item_blocks
        _hip.call("cpc_conv1_bwd", _hip.ptr(x, self.x_off), _hip.ptr(self.dact[0]), _hip.ptr(self.slabs), B, c0, s0, k0, self.L, Lv[0], La[0],
                  nblk, nbb, code)
        stride = (k0 + 1) * c0
        _hip.call("cpc_reduce_slabs", _hip.ptr(self.slabs), _hip.ptr(g["encoder.layers.0.weight"]), k0, c0, nbb * nblk, stride, 1, k0, 1, 0)
        if "encoder.layers.0.bias" in g:
            _hip.call("cpc_reduce_slabs", _hip.ptr(self.slabs, k0 * c0), _hip.ptr(g["encoder.layers.0.bias"]), 1, c0, nbb * nblk, stride,
                      1, 1, 0, 0)

    # ------------------------------------------------------------------------------------------ whole step
    def loss_and_grads(self, x, softplus: bool, regularization: float, all_timesteps: bool = False, grad_ready_hook=None):
        """Forward + loss + backward; returns the device tensor [loss, max_score, -mean valid, mean lse, reg] (no sync)."""
        self.forward(x)
        if all_timesteps:
            self.nce_all_forward_backward(softplus, regularization)
        else:
            self.nce_forward_backward(softplus, regularization)
        self.backward(x, grad_ready_hook=grad_ready_hook)
        return self.nce_out


class GradAllReduce:
    """Data-parallel gradient exchange: ONE sum over ranks of the model's flat f32 gradient buffer per step (RCCL over
    xGMI with backend "nccl"), issued in two pieces so that the larger, earlier-finished piece (encoder layers >= 3,
    GRU, predictor: 63 % of the bytes) travels while encoder layers 2 and 1 are still being differentiated.
    The mean is taken by FusedAdam's ``grad_scale = 1 / world``."""

    def __init__(self, model):
        import torch.distributed as dist
        self.dist = dist
        self.model = model
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.pending = []
        self.split = None

    def hook(self, lo, hi):
        """Pass as ``grad_ready_hook``: starts the asynchronous all-reduce of flat_grad[lo:hi]."""
        self.split = lo
        self.pending.append(self.dist.all_reduce(self.model._flat_grad[lo:hi], async_op=True))

    def finish(self):
        """Reduces what the hook has not covered and makes the current stream wait for all pieces."""
        flat = self.model._flat_grad
        rest = flat if self.split is None else flat[:self.split]
        if rest.numel():
            self.pending.append(self.dist.all_reduce(rest, async_op=True))
        for work in self.pending:
            work.wait()
        self.pending, self.split = [], None


class FusedAdam:
    """torch.optim.Adam (default betas / eps, no weight decay) over the model's flat f32 parameter buffer as one kernel."""

    def __init__(self, model, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.model = model
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        flat = model._flat_param
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.t = 0

    def step(self, grad_scale: float = 1.0):
        self.t += 1
        flat, grad = self.model._flat_param, self.model._flat_grad
        _hip.call("cpc_adam", _hip.ptr(flat), _hip.ptr(grad), _hip.ptr(self.m), _hip.ptr(self.v), C.c_longlong(flat.numel()),
                  C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps), self.t,
                  C.c_float(grad_scale))


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
