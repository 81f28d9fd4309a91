"""Host-side mirror of the reference's attention context network (reference attention_model.py:9-82 and the vendored
transformer.py:131-171, :223-272): the same constructor arguments, attribute names and ``state_dict`` keys, so checkpoints
move in both directions.  The transformer containers hold the parameters (standard torch containers, default initialisers, the reference's construction
order); inside ``AudioPredictiveCodingModel`` (or ``AttentionModel(z)``) the arithmetic and its gradients run in the HIP kernels
behind ``engine.AttentionContext``; a stand-alone ``TransformerEncoderLayer`` / ``TransformerEncoder`` call runs the same kernels
forward-only.
"""
import copy
import math

import torch
import torch.nn as nn


class PositionalEncoder(nn.Module):
    """Constant sin/cos table ``pe`` (max_seq_len, 1, code_size) — reference attention_model.py:9-26."""

    def __init__(self, code_size, max_seq_len=128, max_wavelength=10000):
        super().__init__()
        self.code_size = code_size
        table = torch.zeros(max_seq_len, code_size)
        for pos in range(max_seq_len):
            for i in range(0, code_size, 2):
                angle = math.pi * pos / (max_wavelength ** ((2 * i) / code_size))
                table[pos, i] = math.sin(angle)
                table[pos, i + 1] = math.cos(angle)
        self.register_buffer('pe', table.unsqueeze(1))

    def forward(self, x):
        """x (steps, batch, code_size) on the GPU -> x * sqrt(code_size) + pe[:steps] (reference attention_model.py:28-35) through
        cpc_pe_scale_fwd.  Inference only when called stand-alone (inside AudioPredictiveCodingModel the same kernel runs as part
        of engine.AttentionContext, where its gradient exists); unlike the reference the caller's tensor is not rescaled in place."""
        from . import _hip
        if not x.is_cuda:
            raise RuntimeError("PositionalEncoder runs on the GPU only (no CPU fallback)")
        S, B, C = x.shape
        xb = x.detach().permute(1, 0, 2).contiguous().float()
        pe = self.pe[:S, 0, :].detach().to(device=x.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, S, C, device=x.device, dtype=torch.float32)
        _hip.call("cpc_pe_scale_fwd", _hip.ptr(xb), _hip.ptr(pe), _hip.ptr(out), B, S, C, S * C, math.sqrt(self.code_size), _hip.F32)
        return out.permute(1, 0, 2).to(x.dtype)


def _is_causal_mask(mask, S):
    """Is ``mask`` the (S, S) additive mask AttentionModel builds (attention_model.py:61-63): -inf above the diagonal, 0 elsewhere?"""
    if mask is None or mask.dim() != 2 or tuple(mask.shape) != (S, S):
        return False
    m = mask.detach().float().cpu()
    upper = torch.triu(torch.ones(S, S, dtype=torch.bool), diagonal=1)
    return bool(torch.isneginf(m[upper]).all()) and bool((m[~upper] == 0).all())


def _standalone_input(module, src, mask):
    """Checks shared by the stand-alone layer / encoder calls; returns the rows (B*S, C) float32, item-major, and (B, S, C)."""
    if not src.is_cuda:
        raise RuntimeError("the transformer layers run on the GPU only (no CPU fallback)")
    if torch.is_grad_enabled() and src.requires_grad:
        raise NotImplementedError("a stand-alone transformer layer call is forward-only (torch.no_grad() or a detached input): the "
                                  "differentiable path is AttentionModel(z) / the whole model (engine.AttentionContext)")
    S, B, C = src.shape
    if not _is_causal_mask(mask, S):
        raise NotImplementedError("the HIP attention kernels implement the causal mask of attention_model.py:61-63 only: pass the "
                                  "(S, S) mask with -inf above the diagonal")
    return src.detach().permute(1, 0, 2).contiguous().float(), B, S, C


def _layer_rows(layer, X, B, S, seed, site0):
    """One post-norm encoder layer (transformer.py:262-271) on rows X (B*S, C) float32 through the C ABI: in_proj GEMM, cpc_attn_fwd,
    out_proj GEMM, cpc_add_ln_fwd (residual + dropout + LayerNorm), linear1 + ReLU GEMM, cpc_dropout, linear2 GEMM, cpc_add_ln_fwd.
    Dropout (train mode, p > 0): the counter-based masks of include/cpc_hip.h with sites site0 ... site0 + 3."""
    from . import _hip
    P, F32 = _hip.ptr, _hip.F32
    attn = layer.self_attn
    C, heads, FF = attn.embed_dim, attn.num_heads, layer.linear1.out_features
    if S > 64 or C % heads or C // heads > 64 or C % 8 or FF % 8 or C > 2048:
        raise NotImplementedError("HIP attention kernels: at most 64 steps, head size <= 64, channel counts multiples of 8 (<= 2048)")
    dp = float(layer.dropout.p) if layer.training else 0.0
    M, dev = B * S, X.device
    w = lambda t: t.detach().float().contiguous()
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    qkv, att, probs, y, x1, f1, out, stats = new(M, 3 * C), new(M, C), new(B * heads, S, S), new(M, C), new(M, C), new(M, FF), new(M, C), new(M, 2)
    w_in, b_in, w_o, b_o = w(attn.in_proj_weight), w(attn.in_proj_bias), w(attn.out_proj.weight), w(attn.out_proj.bias)
    w1, b1, w2, b2 = w(layer.linear1.weight), w(layer.linear1.bias), w(layer.linear2.weight), w(layer.linear2.bias)
    n1w, n1b, n2w, n2b = w(layer.norm1.weight), w(layer.norm1.bias), w(layer.norm2.weight), w(layer.norm2.bias)
    _hip.gemm_nt(P(X), P(w_in), P(qkv), M, 3 * C, C, C, C, 3 * C, F32, bias=P(b_in))
    _hip.call("cpc_attn_fwd", P(qkv), P(att), P(probs), B, S, C, heads, dp, seed, site0 + 0, F32)
    _hip.gemm_nt(P(att), P(w_o), P(y), M, C, C, C, C, C, F32, bias=P(b_o))
    _hip.call("cpc_add_ln_fwd", P(X), P(y), P(n1w), P(n1b), None, P(x1), P(stats), M, C, layer.norm1.eps, dp, seed, site0 + 1, F32)
    _hip.gemm_nt(P(x1), P(w1), P(f1), M, FF, C, C, C, FF, F32, bias=P(b1), flags=_hip.GEMM_RELU)
    if dp > 0.0:
        _hip.call("cpc_dropout", P(f1), M * FF, dp, seed, site0 + 2, F32)
    _hip.gemm_nt(P(f1), P(w2), P(y), M, C, FF, FF, FF, C, F32, bias=P(b2))
    _hip.call("cpc_add_ln_fwd", P(x1), P(y), P(n2w), P(n2b), None, P(out), P(stats), M, C, layer.norm2.eps, dp, seed, site0 + 3, F32)
    return out


def _dropout_seed():
    return (torch.initial_seed() * 0x9E3779B1 + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF


class TransformerEncoderLayer(nn.Module):
    """Post-norm encoder layer: self_attn (in_proj / out_proj), linear1, linear2, norm1, norm2 (reference transformer.py:241-252)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        """src (steps, batch, d_model) on the GPU -> the layer's output, same shape (reference transformer.py:254-272), in float32 on the
        HIP kernels.  Forward only, with the causal ``src_mask`` AttentionModel passes; no key padding mask.  Inside AttentionModel /
        AudioPredictiveCodingModel the same kernels run in engine.AttentionContext, where the gradients exist."""
        if src_key_padding_mask is not None:
            raise NotImplementedError("key padding masks are not part of the HIP path (the reference never passes one)")
        X, B, S, C = _standalone_input(self, src, src_mask)
        out = _layer_rows(self, X, B, S, _dropout_seed() if self.training else 0, 0)
        return out.view(B, S, C).permute(1, 0, 2).to(src.dtype)


class TransformerEncoder(nn.Module):
    """``num_layers`` deep copies of one layer (so all layers start from identical values, as in the reference
    transformer.py:144-148, :339-340) and an optional final norm."""

    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, mask=None, src_key_padding_mask=None):
        """src (steps, batch, d_model) on the GPU through the stack of layers and the final norm (reference transformer.py:150-170);
        forward only, causal ``mask`` (see TransformerEncoderLayer.forward)."""
        from . import _hip
        if src_key_padding_mask is not None:
            raise NotImplementedError("key padding masks are not part of the HIP path (the reference never passes one)")
        X, B, S, C = _standalone_input(self, src, mask)
        seed = _dropout_seed() if self.training else 0
        for l, layer in enumerate(self.layers):
            X = _layer_rows(layer, X, B, S, seed, 4 * l)
        if self.norm is not None:
            out, stats = torch.empty_like(X), torch.empty(B * S, 2, device=X.device, dtype=torch.float32)
            _hip.call("cpc_add_ln_fwd", _hip.ptr(X), None, _hip.ptr(self.norm.weight.detach().float().contiguous()),
                      _hip.ptr(self.norm.bias.detach().float().contiguous()), None, _hip.ptr(out), _hip.ptr(stats), B * S, C, self.norm.eps,
                      0.0, 0, 0, _hip.F32)
            X = out
        return X.view(B, S, C).permute(1, 0, 2).to(src.dtype)


class AttentionModel(nn.Module):
    """Causal transformer encoder over the visible steps, mean over time, ``end_layer`` (reference attention_model.py:38-82).

    ``args_dict`` keys: channels, num_layers, num_heads, feedforward_size, dropout, sequence_length, output_size.
    Note that the reference multiplies z by sqrt(channels) IN PLACE (attention_model.py:30), which also rescales the ``z``
    that AudioPredictiveCodingModel.forward returns; the HIP path reproduces that."""

    def __init__(self, args_dict):
        super().__init__()
        self.channels = int(args_dict['channels'])
        self.num_layers = int(args_dict['num_layers'])
        self.num_heads = int(args_dict['num_heads'])
        self.feedforward_size = int(args_dict['feedforward_size'])
        self.dropout = float(args_dict['dropout'])
        self.sequence_length = int(args_dict['sequence_length'])
        self.output_size = int(args_dict['output_size'])
        self.positional_encoder = PositionalEncoder(self.channels, self.sequence_length)
        layer = TransformerEncoderLayer(self.channels, self.num_heads, self.feedforward_size, self.dropout)
        self.encoder = TransformerEncoder(layer, self.num_layers, nn.LayerNorm(self.channels))
        self.end_layer = nn.Linear(self.channels, self.output_size)

    def forward(self, x):
        """(batch, channels, steps) on the GPU -> (batch, output_size).  Inference only when called stand-alone; unlike the
        reference this does not rescale the caller's tensor in place (attention_model.py:30)."""
        from .engine import standalone_context_forward
        return standalone_context_forward(self, x, self.output_size)
