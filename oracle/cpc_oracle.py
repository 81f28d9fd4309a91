"""CPU oracle for the CPC-audio train step.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch functional ops on CPU tensors,
float32 or float64) of the reference algorithm of
vincentherrmann/constrastive-predictive-coding-audio for ONE path: the
ContrastiveEstimationTrainer train / validate step.  It is the checker for the HIP
path; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package never imports anything from ``oracle/``.

Pinning: every function here is checked in ``tests/test_oracle_golden.py`` against
golden vectors produced by importing the reference itself in the development
container (``tests/golden/generate_golden.py``; the reference's own tests hold no
numeric vectors — SURVEY.md section 8c).  Each function cites the reference
file:line it follows (paths relative to the reference repository root).

Parameters are passed as a flat ``dict`` keyed by the reference's state_dict names
(``encoder.layers.N.weight`` ...), so a reference checkpoint can be fed directly.
"""
from __future__ import annotations

import math
import random
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

DEFAULT_STRIDES = (5, 4, 2, 2, 2)
DEFAULT_KERNELS = (10, 8, 4, 4, 4)


# --------------------------------------------------------------------------- encoder
def encoder_geometry(strides: Sequence[int], kernel_sizes: Sequence[int]) -> Tuple[int, int]:
    """(downsampling_factor, receptive_field) — audio_model.py:19-25."""
    ds = 1
    for s in strides:
        ds *= int(s)
    rf = int(kernel_sizes[0])
    hop = 1
    for i in range(1, len(strides)):
        hop *= int(strides[i - 1])
        rf += (int(kernel_sizes[i]) - 1) * hop
    return ds, rf


def encoder_layer_lengths(length: int, strides: Sequence[int], kernel_sizes: Sequence[int]) -> List[int]:
    """Output length of every conv layer (no padding, dilation 1) — audio_model.py:28-34."""
    out = []
    cur = int(length)
    for s, k in zip(strides, kernel_sizes):
        cur = (cur - int(k)) // int(s) + 1
        out.append(cur)
    return out


def encoder_forward(x: torch.Tensor, params: Params, strides: Sequence[int] = DEFAULT_STRIDES,
                    prefix: str = "encoder.", return_all: bool = False):
    """relu(conv) on every layer but the last, bare conv on the last — audio_model.py:36-44.

    x: (B, 1, L).  Returns (B, C, T) (and the per-layer activations if return_all).
    """
    acts = []
    n = len(strides)
    for l in range(n):
        w = params[f"{prefix}layers.{l}.weight"]
        b = params.get(f"{prefix}layers.{l}.bias")
        x = F.conv1d(x, w, b, stride=int(strides[l]))
        if l < n - 1:
            x = torch.relu(x)
        acts.append(x)
    return (x, acts) if return_all else x


# ------------------------------------------------------------------------------- GRU
def gru_cell(x: torch.Tensor, h: torch.Tensor, w_ih, w_hh, b_ih, b_hh) -> torch.Tensor:
    """One torch.nn.GRUCell step written out (gate order r, z, n) — audio_model.py:58-60,72."""
    gi = x @ w_ih.t()
    gh = h @ w_hh.t()
    if b_ih is not None:
        gi = gi + b_ih
        gh = gh + b_hh
    hs = h.shape[1]
    r = torch.sigmoid(gi[:, :hs] + gh[:, :hs])
    u = torch.sigmoid(gi[:, hs:2 * hs] + gh[:, hs:2 * hs])
    n = torch.tanh(gi[:, 2 * hs:] + r * gh[:, 2 * hs:])
    return (1.0 - u) * n + u * h


def gru_forward(z: torch.Tensor, params: Params, prefix: str = "autoregressive_model.gruCell.",
                return_trace: bool = False):
    """AudioGRUModel.forward with reset_hidden=True — audio_model.py:66-77.

    z: (B, input_size, steps); hidden starts at zero; returns the last hidden (B, H).
    """
    w_ih = params[prefix + "weight_ih"]
    w_hh = params[prefix + "weight_hh"]
    b_ih = params.get(prefix + "bias_ih")
    b_hh = params.get(prefix + "bias_hh")
    h = torch.zeros(z.shape[0], w_hh.shape[1], dtype=z.dtype)
    trace = []
    for t in range(z.shape[2]):
        h = gru_cell(z[:, :, t], h, w_ih, w_hh, b_ih, b_hh)
        trace.append(h)
    return (h, trace) if return_trace else h


def conv_ar_forward(z: torch.Tensor, params: Params, kernel_sizes: Sequence[int], poolings: Sequence[int],
                    prefix: str = "autoregressive_model.", strides: Optional[Sequence[int]] = None, batch_norm: bool = False,
                    residual: bool = False, training: bool = True) -> torch.Tensor:
    """ConvolutionalArModel.forward — audio_model.py:98-136 (block), :158-161 (model): per block
    [MaxPool1d(pool, ceil_mode=True)] -> Conv1d -> [BatchNorm1d] -> ReLU, plus the optional residual branch
    [MaxPool1d(pool*stride, ceil)] -> [Conv1d 1x1 if the channel count changes], right-aligned and added.  The reference
    adds IN PLACE (``main_x += ...``, :133), which modern autograd rejects in backward; the out-of-place sum below has the
    same value and is what the gradients of the residual variants are defined by.  Returns the last position (B, C_out)."""
    x = z
    strides = [1] * len(kernel_sizes) if strides is None else strides
    for l, (k, pool, stride) in enumerate(zip(kernel_sizes, poolings, strides)):
        pre = f"{prefix}module_list.{l}."
        original = x
        idx = 0
        if pool > 1:
            x = F.max_pool1d(x, pool, ceil_mode=True)
            idx = 1
        x = F.conv1d(x, params[f"{pre}main_modules.{idx}.weight"], params.get(f"{pre}main_modules.{idx}.bias"), stride=stride)
        if batch_norm:
            bn = f"{pre}main_modules.{idx + 1}."
            x = F.batch_norm(x, params[bn + "running_mean"], params[bn + "running_var"], params[bn + "weight"], params[bn + "bias"],
                             training=training, momentum=0.1, eps=1e-5)
            if training:
                params[bn + "num_batches_tracked"] += 1
        x = torch.relu(x)
        if residual:
            r, ridx = original, 0
            if pool * stride > 1:
                r = F.max_pool1d(r, pool * stride, ceil_mode=True)
                ridx = 1
            wname = f"{pre}residual_modules.{ridx}.weight"
            if wname in params:
                r = F.conv1d(r, params[wname], params.get(f"{pre}residual_modules.{ridx}.bias"))
            x = x + r[:, :, -x.shape[2]:]
    return x[:, :, -1]


def positional_encoding(max_len: int, channels: int, max_wavelength: float = 10000.0) -> torch.Tensor:
    """The constant table of PositionalEncoder.__init__ — attention_model.py:17-25 (note the exponent 2i/C with i the
    even channel index itself, and the pi factor).  Returns (max_len, channels) float32."""
    import math
    pe = torch.zeros(max_len, channels)
    for pos in range(max_len):
        for i in range(0, channels, 2):
            a = math.pi * pos / (max_wavelength ** ((2 * i) / channels))
            pe[pos, i] = math.sin(a)
            pe[pos, i + 1] = math.cos(a)
    return pe


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w + b


def attention_forward(z: torch.Tensor, params: Params, num_layers: int, num_heads: int,
                      prefix: str = "autoregressive_model.", dropout_factors=None):
    """AttentionModel.forward in eval mode / dropout 0 — attention_model.py:59-82, with the post-norm encoder layer of
    transformer.py:262-271 (self-attention + residual + LayerNorm, Linear-ReLU-Linear + residual + LayerNorm), the final
    LayerNorm of TransformerEncoder (transformer.py:167-168), the mean over time and end_layer.

    ``dropout_factors`` (train mode with p > 0): {(layer, site): tensor of factors 0 or 1/(1-p)} for the four nn.Dropout
    sites of a layer — 0: attention weights (B*heads, S, S), 1: attention block output (S, B, C), 2: feed-forward hidden
    (S, B, FF), 3: feed-forward output (S, B, C) — i.e. nn.Dropout with its mask given instead of drawn.

    z (B, C, S).  Returns (c (B, out), z * sqrt(C)): PositionalEncoder multiplies its input IN PLACE
    (attention_model.py:30), and that input is a view of the z the model returns, so the returned z is the scaled one."""
    import math
    B, C, S = z.shape
    d = C // num_heads
    z_scaled = z * math.sqrt(C)
    x = z_scaled.permute(2, 0, 1) + positional_encoding(S, C).unsqueeze(1)          # (S, B, C)
    mask = torch.triu(torch.full((S, S), float("-inf")), diagonal=1)
    df = dropout_factors or {}
    drop = lambda t, l, site: t * df[(l, site)].to(t.dtype) if (l, site) in df else t
    for l in range(num_layers):
        pl = f"{prefix}encoder.layers.{l}."
        qkv = x @ params[pl + "self_attn.in_proj_weight"].t() + params[pl + "self_attn.in_proj_bias"]
        q, k, v = (t.reshape(S, B * num_heads, d).transpose(0, 1) for t in qkv.split(C, dim=-1))   # (B*h, S, d)
        scores = (q * d ** -0.5) @ k.transpose(1, 2) + mask
        o = drop(torch.softmax(scores, dim=-1), l, 0) @ v                              # (B*h, S, d)
        o = o.transpose(0, 1).reshape(S, B, C)
        y = o @ params[pl + "self_attn.out_proj.weight"].t() + params[pl + "self_attn.out_proj.bias"]
        x = layer_norm(x + drop(y, l, 1), params[pl + "norm1.weight"], params[pl + "norm1.bias"])
        f = drop(torch.relu(x @ params[pl + "linear1.weight"].t() + params[pl + "linear1.bias"]), l, 2)
        f = f @ params[pl + "linear2.weight"].t() + params[pl + "linear2.bias"]
        x = layer_norm(x + drop(f, l, 3), params[pl + "norm2.weight"], params[pl + "norm2.bias"])
    x = layer_norm(x, params[prefix + "encoder.norm.weight"], params[prefix + "encoder.norm.bias"])
    m = x.sum(0) / S
    c = m @ params[prefix + "end_layer.weight"].t() + params[prefix + "end_layer.bias"]
    return c, z_scaled


# ------------------------------------------------------------------ scalogram front end
def cqt_frequencies(n_bins: int, fmin: float, bins_per_octave: int):
    """librosa.time_frequency.cqt_frequencies (tuning 0): fmin * 2**(k / bins_per_octave).  Third-party, absent here."""
    import numpy as np
    return float(fmin) * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bins_per_octave)


def constant_q_filters(sr, fmin, n_bins, bins_per_octave, filter_scale):
    """Restatement of librosa.filters.constant_q(window='hann', norm=1, pad_fft=True) — the third-party call of
    constant_q_transform.py:108-112 (librosa is not vendored and its version is not pinned by the reference: coefficient
    parity UNPINNED; everything downstream is pinned by running the reference on these coefficients).
    Filter k: exp(2 pi i f_k n / sr) for n in [-l/2, l/2), l = Q sr / f_k, Q = filter_scale / (2**(1/bpo) - 1), times a
    periodic Hann window, L1-normalised, centre-padded to the next power of two.  Returns (complex128 [n_bins, P], lengths)."""
    import numpy as np
    freqs = cqt_frequencies(n_bins, fmin, bins_per_octave)
    q = float(filter_scale) / (2.0 ** (1.0 / bins_per_octave) - 1.0)
    lengths = q * sr / freqs
    sigs = []
    for ilen, freq in zip(lengths, freqs):
        sig = np.exp(np.arange(-ilen // 2, ilen // 2, dtype=np.float64) * 1j * 2 * np.pi * freq / sr)
        n = len(sig)
        sig = sig * (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n))
        sigs.append(sig / np.sum(np.abs(sig)))
    max_len = int(2.0 ** np.ceil(np.log2(max(len(s) for s in sigs))))
    bank = np.zeros((n_bins, max_len), dtype=np.complex128)
    for k, sig in enumerate(sigs):
        lpad = (max_len - len(sig)) // 2
        bank[k, lpad:lpad + len(sig)] = sig
    return bank, lengths


def cqt_forward(x: torch.Tensor, weights: Sequence[torch.Tensor], hop: int) -> torch.Tensor:
    """CQT.forward — constant_q_transform.py:161-172.  weights[g] (2 n_g, 1, size_g): real rows then imaginary rows, sizes
    decreasing.  x (B, 1, L) -> (B, n_bins, T, 2)."""
    k0 = weights[0].shape[-1]
    real, imag = [], []
    for w in weights:
        offset = (k0 - w.shape[-1]) // 2
        res = F.conv1d(x[:, :, offset:-(offset + 1)], w, stride=hop)
        r, i = torch.chunk(res, 2, dim=1)
        real.append(r)
        imag.append(i)
    return torch.stack([torch.cat(real, dim=1), torch.cat(imag, dim=1)], dim=3)


def phase_difference_constants(sr, fmin, n_bins, bins_per_octave, hop):
    """PhaseDifference.__init__ — constant_q_transform.py:272-280: (fixed advance, 1/ln f) per bin, float32."""
    import numpy as np
    freqs = cqt_frequencies(n_bins, fmin, bins_per_octave)
    fixed = (((1.0 * freqs * hop / sr) + 0.5) % 1 - 0.5) * 2 * np.pi
    return torch.from_numpy(fixed).float(), torch.from_numpy(1 / np.log(freqs)).float()


def preprocessing_forward(cq: torch.Tensor, phase_consts=None, offset_zero: bool = False, output_power: float = 1.0,
                          scaling: float = 1.0, pooling=None) -> torch.Tensor:
    """PreprocessingModule.forward after the CQT — scalogram_model.py:77-97 with abs / angle / unwrap /
    PhaseDifference.forward of constant_q_transform.py:36-52, :69-72, :281-285.  cq (B, bins, T, 2)."""
    import math
    offset = 1e-9 if offset_zero else 0.0
    log_offset = -math.log(offset) if offset_zero else 0.0
    norm = scaling / log_offset if offset_zero else scaling
    mag = torch.sqrt(cq[..., 0] ** 2 + cq[..., 1] ** 2)
    if phase_consts is not None:
        fixed, pscale = phase_consts
        amp = torch.log(mag[:, :, 1:] ** 2 + offset) + log_offset
        ang = torch.atan2(cq[..., 1], cq[..., 0])
        pd = ang[:, :, 1:] - ang[:, :, :-1] + fixed.view(1, -1, 1)
        pd = torch.where(pd > math.pi, pd - 2 * math.pi, pd)
        pd = torch.where(pd < -math.pi, pd + 2 * math.pi, pd)
        x = torch.stack([amp, pd * pscale.view(1, -1, 1)], dim=1)
    else:
        x = (torch.log(mag ** 2 + offset) + log_offset).unsqueeze(1)
    if pooling is not None:
        x = F.max_pool2d(x, list(pooling))
    return (x * norm) ** output_power


def scalogram_block_forward(x: torch.Tensor, params: Params, prefix: str, cfg: dict, training: bool, last: bool) -> torch.Tensor:
    """ScalogramEncoderBlock.forward — scalogram_model.py:381-479 (module order :388-425, residual branch :427-441,
    cropped add :453-472) followed by the F.relu ScalogramResidualEncoder.forward applies between blocks (:525-526).
    ``params`` holds weights and BatchNorm buffers under the reference's state_dict keys; running statistics are updated
    in place when ``training``."""
    def conv_bn_relu(h, idx, tag):
        top = cfg.get("top_padding_" + tag)
        if top is not None:
            h = F.pad(h, (0, 0, top, 0))
            idx += 1
        if cfg.get("separable"):
            # Conv2dSeparable (scalogram_model.py:532-544): depthwise k x k (groups = channels, no bias), then 1 x 1 with bias
            wd = params[f"{prefix}main_modules.{idx}.conv.weight"]
            h = F.conv2d(h, wd, None, stride=cfg["stride_" + tag], padding=cfg["padding_" + tag], groups=wd.shape[0])
            h = F.conv2d(h, params[f"{prefix}main_modules.{idx}.conv_1x1.weight"], params.get(f"{prefix}main_modules.{idx}.conv_1x1.bias"))
        else:
            h = F.conv2d(h, params[f"{prefix}main_modules.{idx}.weight"], params.get(f"{prefix}main_modules.{idx}.bias"),
                         stride=cfg["stride_" + tag], padding=cfg["padding_" + tag])
        idx += 1
        if cfg["batch_norm"]:
            bn = f"{prefix}main_modules.{idx}."
            h = F.batch_norm(h, params[bn + "running_mean"], params[bn + "running_var"], params[bn + "weight"], params[bn + "bias"],
                             training=training, momentum=0.1, eps=1e-5)
            if training:
                params[bn + "num_batches_tracked"] += 1
            idx += 1
        if cfg.get("pooling_" + tag, 1) > 1:
            h = F.max_pool2d(h, cfg["pooling_" + tag], ceil_mode=bool(cfg.get("ceil_pooling", False)))
            idx += 1
        return torch.relu(h), idx + 2          # ReLU + ActivationWriter

    main, idx = conv_bn_relu(x, 0, "1")
    main, _ = conv_bn_relu(main, idx, "2")
    if cfg["residual"]:
        res, ridx = x, 0
        pool = cfg["stride_1"] * cfg["stride_2"] * cfg.get("pooling_1", 1) * cfg.get("pooling_2", 1)
        if pool > 1:
            res = F.max_pool2d(res, pool, ceil_mode=True)
            ridx = 1
        if cfg["in_channels"] != cfg["out_channels"]:
            res = F.conv2d(res, params[f"{prefix}residual_modules.{ridx}.weight"], None, padding=cfg["padding_1"] + cfg["padding_2"])
        m_h, m_w = main.shape[2], main.shape[3]
        o_h, o_w = (res.shape[2] - m_h + 1) / 2, (res.shape[3] - m_w + 1) / 2
        if int(o_h) > 0:
            res = res[:, :, -int(o_h + m_h):-int(o_h), :]
        if int(o_w) > 0:
            res = res[:, :, :, -int(o_w + m_w):-int(o_w)]
        main = main + res
    return main if last else torch.relu(main)


def scalogram_encoder_forward(x: torch.Tensor, params: Params, blocks: Sequence[dict], training: bool = True,
                              prefix: str = "encoder.") -> torch.Tensor:
    """ScalogramResidualEncoder.forward — scalogram_model.py:519-529: (B, C, bins, frames) -> (B, E, frames')."""
    for i, cfg in enumerate(blocks):
        x = scalogram_block_forward(x, params, f"{prefix}blocks.{i}.", cfg, training, last=i == len(blocks) - 1)
    return x[:, :, 0, :]


# ------------------------------------------------------------------------- CPC model
def item_length(receptive_field: int, downsampling: int, visible_steps: int, prediction_steps: int) -> int:
    """audio_model.py:187-191."""
    return receptive_field + (visible_steps + prediction_steps) * downsampling


def cpc_forward(x: torch.Tensor, params: Params, visible_steps: int, prediction_steps: int,
                strides: Sequence[int] = DEFAULT_STRIDES, conv_ar=None, attention=None, scalogram=None, training: bool = True,
                ar_resnet=None):
    """AudioPredictiveCodingModel.forward with AudioEncoder + AudioGRUModel — audio_model.py:193-211.
    conv_ar = (kernel_sizes, poolings) selects ConvolutionalArModel, attention = (num_layers, num_heads) AttentionModel.

    Returns (predicted_z (B,K,E), targets (B,E,K), z (B,E,V), c (B,H)); targets are NOT detached.
    """
    if scalogram is not None:       # scalogram = list of block dicts; x is the (B, C, bins, frames) scalogram
        enc = scalogram_encoder_forward(x, params, scalogram, training)
    else:
        enc = encoder_forward(x, params, strides)
    K, V = prediction_steps, visible_steps
    targets = enc[:, :, -K:]
    z = enc[:, :, -(V + K):-K]
    if ar_resnet is not None:       # ScalogramResidualEncoder as the context network (configs ar_resnet_architecture_*): block dicts
        c3 = scalogram_encoder_forward(z.unsqueeze(2), params, ar_resnet, training, prefix="autoregressive_model.")
        c = c3[:, :, 0]             # audio_model.py:203-204
    elif attention is not None:     # (num_layers, num_heads[, dropout_factors])
        c, z = attention_forward(z, params, attention[0], attention[1], dropout_factors=attention[2] if len(attention) > 2 else None)
    elif conv_ar is None:
        c = gru_forward(z, params)
    elif isinstance(conv_ar, dict):     # full ConvolutionalArModel args: kernel_sizes, pooling, stride, batch_norm, residual
        c = conv_ar_forward(z, params, conv_ar["kernel_sizes"], conv_ar["pooling"], strides=conv_ar["stride"],
                            batch_norm=conv_ar["batch_norm"], residual=conv_ar["residual"], training=training)
    else:       # conv_ar = (kernel_sizes, poolings) of a plain ConvolutionalArModel
        c = conv_ar_forward(z, params, conv_ar[0], conv_ar[1])
    w_p = params["prediction_model.weight"]
    predicted = (c @ w_p.t()).view(-1, K, enc.shape[1])
    return predicted, targets, z, c


# ------------------------------------------------------------------- score functions
def linear_scores(predicted_z: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """scores[b,k,b',k'] = sum_e predicted_z[b,k,e] * targets[b',e,k'] — contrastive_estimation_training.py:19-22."""
    return torch.einsum("bke,cel->bkcl", predicted_z, targets)


def softplus_scores(predicted_z: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """softplus of the linear scores — contrastive_estimation_training.py:12-16."""
    return F.softplus(linear_scores(predicted_z, targets))


def difference_scores(predicted_z: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """1 / squared distance — contrastive_estimation_training.py:25-33 (small shapes only)."""
    diff = predicted_z[:, :, :, None, None] - targets.permute(1, 0, 2)[None, None]
    return 1.0 / (diff ** 2).sum(dim=2)


SCORE_FUNCTIONS = {"linear": linear_scores, "softplus": softplus_scores, "difference": difference_scores}


# ------------------------------------------------------------------------------ loss
def _loss_terms(scores4: torch.Tensor, all_timesteps: bool):
    """The (scores, noise_scoring, valid_scores) triple of contrastive_estimation_training.py:108-119.

    Keeps the reference's raw ``view(-1, batch, steps)`` reinterpretation in the default
    branch (:117) so that the per-step numbers ``validate`` reports (:237-241) match too.
    """
    B, K = scores4.shape[0], scores4.shape[1]
    if all_timesteps:
        noise = torch.logsumexp(scores4.reshape(-1, B, K), dim=0)
        valid = torch.diagonal(torch.diagonal(scores4, dim1=0, dim2=2), dim1=0, dim2=1)
        return scores4, noise, valid
    s = torch.diagonal(scores4, dim1=1, dim2=3).permute(0, 2, 1).contiguous()  # (b, k, b')
    noise = torch.logsumexp(s.view(-1, B, K), dim=0)
    valid = torch.diagonal(s, dim1=0, dim2=2).permute(1, 0)
    return s, noise, valid


def info_nce_loss(scores4: torch.Tensor, all_timesteps: bool = False, regularization: float = 1.0):
    """Train-mode loss incl. the regulariser — contrastive_estimation_training.py:106-122,141.

    Returns (loss, max_score) where max_score is what the logger records (:166).
    """
    s, noise, valid = _loss_terms(scores4, all_timesteps)
    loss = torch.mean(-torch.mean(valid - noise, dim=1))
    loss = loss + regularization * torch.mean(torch.mean(s, dim=1) ** 2)
    return loss, s.max()


def validation_terms(scores4: torch.Tensor, all_timesteps: bool = False):
    """Per-step losses and accuracies of one validation batch — contrastive_estimation_training.py:227-247."""
    B, K = scores4.shape[0], scores4.shape[1]
    s, noise, valid = _loss_terms(scores4, all_timesteps)
    prediction_losses = -torch.mean(valid - noise, dim=0)
    n = B * K if all_timesteps else B
    template = torch.arange(n)
    template = template.view(B, K) if all_timesteps else template.unsqueeze(1).repeat(1, K)
    max_score = torch.argmax(s.reshape(B, K, -1), dim=2)
    accuracy = torch.sum(torch.eq(template, max_score), dim=0).to(scores4.dtype) / n
    return prediction_losses, accuracy, s.mean()


def gradient_penalty(score_sum: torch.Tensor, batch: torch.Tensor, factor: float) -> torch.Tensor:
    """Wasserstein gradient penalty — contrastive_estimation_training.py:144-155: the gradient of the summed scores with
    respect to the (preprocessed) input batch, its 2-norm over dim 1 pushed towards 1; differentiable (create_graph)."""
    grad, = torch.autograd.grad(outputs=score_sum, inputs=batch, create_graph=True, retain_graph=True, only_inputs=True)
    return ((grad.norm(2, dim=1) - 1) ** 2).mean() * factor


# ------------------------------------------------------------------------------ Adam
def adam_update(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (no weight decay / amsgrad), as built at
    contrastive_estimation_training.py:83.  In place on p, m, v; ``step`` counts from 1."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


class OracleTrainer:
    """Holds parameters + Adam state and runs the reference train step on CPU.

    Follows the loop body of contrastive_estimation_training.py:97-169 (every model family cpc_forward covers; the optional
    Wasserstein gradient penalty of :144-155 with ``gradient_penalty_factor``).
    """

    def __init__(self, params: Params, visible_steps: int, prediction_steps: int,
                 strides: Sequence[int] = DEFAULT_STRIDES, score: str = "softplus",
                 all_timesteps: bool = False, regularization: float = 1.0, lr: float = 1e-4, conv_ar=None,
                 attention=None, scalogram=None, ar_resnet=None, gradient_penalty_factor: Optional[float] = None):
        is_buffer = lambda k: ("running_" in k) or k.endswith("num_batches_tracked") or k.endswith("positional_encoder.pe")
        self.buffers = {k: v.detach().clone() for k, v in params.items() if is_buffer(k)}
        self.params = {k: v.detach().clone().requires_grad_(True) for k, v in params.items() if not is_buffer(k)}
        self.m = {k: torch.zeros_like(v) for k, v in self.params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.params.items()}
        self.scalogram = scalogram
        self.ar_resnet = ar_resnet
        self.V, self.K = visible_steps, prediction_steps
        self.strides = tuple(strides)
        self.score = SCORE_FUNCTIONS[score]
        self.all_timesteps = all_timesteps
        self.regularization = regularization
        self.lr = lr
        self.conv_ar = conv_ar
        self.attention = attention
        self.gp_factor = gradient_penalty_factor      # None: wasserstein_gradient_penalty=False
        self.t = 0

    def loss_and_grads(self, batch: torch.Tensor):
        """batch (B, L) -> (loss, max_score, grads dict); does not update parameters."""
        for p in self.params.values():
            p.grad = None
        x = batch if self.scalogram is not None else batch.unsqueeze(1)
        if self.gp_factor is not None:
            x = x.detach().clone().requires_grad_(True)        # :102 (the reference only does this behind a preprocessing module)
        pred, targ, _, _ = cpc_forward(x, {**self.params, **self.buffers}, self.V, self.K, self.strides, self.conv_ar, self.attention,
                                       self.scalogram, ar_resnet=self.ar_resnet)
        scores = self.score(pred, targ)
        loss, smax = info_nce_loss(scores, self.all_timesteps, self.regularization)
        if self.gp_factor is not None:
            s, _, _ = _loss_terms(scores, self.all_timesteps)      # ``scores`` as rebound at :116 in the default branch
            loss = loss + gradient_penalty(torch.sum(s), x, self.gp_factor)
        loss.backward()
        return loss.detach(), smax.detach(), {k: p.grad for k, p in self.params.items()}

    def step(self, batch: torch.Tensor):
        loss, smax, grads = self.loss_and_grads(batch)
        self.t += 1
        with torch.no_grad():
            for k, p in self.params.items():
                adam_update(p, grads[k], self.m[k], self.v[k], self.t, self.lr)
        return float(loss), float(smax)


# --------------------------------------------------------------------------- sampler
def file_batch_sampler(index_count_per_file: Sequence[int], batch_size: int, file_batch_size: int = 1,
                       drop_last: bool = True, seed: Optional[int] = None) -> List[List[int]]:
    """The batch index lists FileBatchSampler.__iter__ yields — audio_dataset.py:202-263.

    Uses Python's ``random`` exactly as the reference does (same call order), so the same
    seed / global RNG state gives the same lists.  Note ``len()`` of the reference sampler
    is the number of FILE batches (sum of per-file counts), which is also the index range
    shuffled when file_batch_size == 1 (:236).
    """
    if drop_last:
        per_file = [n // file_batch_size for n in index_count_per_file]
    else:
        per_file = [-(-n // file_batch_size) for n in index_count_per_file]
    total = int(sum(per_file))

    def chunks(seq, n):
        out = []
        for i in range(0, len(seq), n):
            if drop_last and i + n > len(seq):
                break
            out.append(seq[i:i + n])
        return out

    if file_batch_size == 1:
        order = list(range(total))
        if seed is not None:
            random.seed(seed)
        random.shuffle(order)
        return chunks(order, batch_size)

    files, start = [], 0
    for n in index_count_per_file:
        files.append(list(range(start, start + n)))
        start += n
    for i, f in enumerate(files):
        if seed is not None:
            random.seed(seed + i)
        random.shuffle(f)
    groups = []
    for f in files:
        groups.extend(chunks(f, file_batch_size))
    if seed is not None:
        random.seed(seed)
    random.shuffle(groups)
    per_batch = batch_size // file_batch_size
    if per_batch > 1:
        merged = []
        for i in range(0, len(groups), per_batch):
            if drop_last and i + per_batch > len(groups):
                break
            merged.append([x for g in groups[i:i + per_batch] for x in g])
        return merged
    return groups


def deterministic_order(n: int, seed: int = 0) -> List[int]:
    """DeterministicSampler.__iter__ — contrastive_estimation_training.py:373-378."""
    order = list(range(n))
    random.seed(seed)
    random.shuffle(order)
    return order


# ------------------------------------------------------------------- parameter setup
def init_params(channels: Sequence[int] = (512,) * 5, kernel_sizes: Sequence[int] = DEFAULT_KERNELS,
                ar_size: int = 256, prediction_steps: int = 12, seed: int = 0,
                dtype=torch.float32) -> Params:
    """Builds a parameter dict with torch's default initialisers in the reference's
    construction order (AudioEncoder, AudioGRUModel, then the predictor Linear —
    audio_model.py:26-34, :59-61, :174), so that ``torch.manual_seed(seed)`` gives the
    same values as constructing the reference modules in that order."""
    import torch.nn as nn
    torch.manual_seed(seed)
    out: Params = {}
    cin = 1
    for l, (c, k) in enumerate(zip(channels, kernel_sizes)):
        conv = nn.Conv1d(cin, c, k)
        out[f"encoder.layers.{l}.weight"] = conv.weight.detach().to(dtype)
        out[f"encoder.layers.{l}.bias"] = conv.bias.detach().to(dtype)
        cin = c
    cell = nn.GRUCell(cin, ar_size)
    for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        out["autoregressive_model.gruCell." + name] = getattr(cell, name).detach().to(dtype)
    lin = nn.Linear(ar_size, cin * prediction_steps, bias=False)
    out["prediction_model.weight"] = lin.weight.detach().to(dtype)
    return out
