"""Host-side mirror of the reference's attention context network (reference attention_model.py:9-82 and the vendored
transformer.py:131-171, :223-272): the same constructor arguments, attribute names and ``state_dict`` keys, so checkpoints
move in both directions.  The transformer containers hold the parameters (standard torch containers, default initialisers, the reference's construction
order); inside ``AudioPredictiveCodingModel`` (or ``AttentionModel(z)``) the arithmetic and its gradients run in the HIP kernels
behind ``engine.AttentionContext``; a stand-alone ``TransformerEncoderLayer`` / ``TransformerEncoder`` call runs the same kernels
in float32 and is differentiable like the reference's modules (``_LayerFn`` / ``_NormFn``: autograd functions over the C ABI's
forward and backward entry points).
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
    S, B, C = src.shape
    if not _is_causal_mask(mask, S):
        raise NotImplementedError("the HIP attention kernels implement the causal mask of attention_model.py:61-63 only: pass the "
                                  "(S, S) mask with -inf above the diagonal")
    return src.permute(1, 0, 2).contiguous().float().reshape(B * S, C), B, S, C


def _layer_check(layer, S):
    attn = layer.self_attn
    C, heads, FF = attn.embed_dim, attn.num_heads, layer.linear1.out_features
    if S > 64 or C % heads or C // heads > 64 or C % 8 or FF % 8 or C > 2048:
        raise NotImplementedError("HIP attention kernels: at most 64 steps, head size <= 64, channel counts multiples of 8 (<= 2048)")
    return C, heads, FF


def _colsum(X, M, N):
    """Column sums of X (M, N) float32: per-block partial sums + one fixed-order reduction (cpc_colsum / cpc_reduce_slabs)."""
    from . import _hip
    nb = min(1024, max(1, M // 64))
    slabs = torch.empty(nb * N, device=X.device, dtype=torch.float32)
    out = torch.empty(N, device=X.device, dtype=torch.float32)
    _hip.call("cpc_colsum", _hip.ptr(X), _hip.ptr(slabs), M, N, N, nb, _hip.F32)
    _hip.call("cpc_reduce_slabs", _hip.ptr(slabs), _hip.ptr(out), 1, N, nb, N, 1, 1, 0, 0)
    return out


def _ln_backward(g1, g2, r, stats, weight, M, C, dp, seed, site):
    """cpc_ln_bwd: (dr, dr_b or None, d weight, d bias) of y = LayerNorm(r) * weight + bias for dy = g1 (+ g2); dr_b: the gradient of a
    dropped summand (dr times its dropout factors) when dp > 0."""
    from . import _hip
    P = _hip.ptr
    nb = max(1, min(512, M // 4))
    new = lambda *shape: torch.empty(*shape, device=r.device, dtype=torch.float32)
    dr, slabs, dw, db = new(M, C), new(nb * 2 * C), new(C), new(C)
    dr_b = new(M, C) if dp > 0.0 else None
    _hip.call("cpc_ln_bwd", P(g1), P(g2), P(r), P(stats), P(weight), P(dr), P(slabs), M, C, 0, 1.0, nb, P(dr_b), dp if dr_b is not None else 0.0,
              seed, site, _hip.F32)
    _hip.call("cpc_reduce_slabs", P(slabs), P(dw), 1, C, nb, 2 * C, 1, 1, 0, 0)
    _hip.call("cpc_reduce_slabs", P(slabs, C), P(db), 1, C, nb, 2 * C, 1, 1, 0, 0)
    return dr, dr_b, dw, db


class _LayerFn(torch.autograd.Function):
    """One post-norm encoder layer (transformer.py:262-271) on rows X (B*S, C) float32 through the C ABI: in_proj GEMM, cpc_attn_fwd,
    out_proj GEMM, cpc_add_ln_fwd (residual + dropout + LayerNorm), linear1 + ReLU GEMM, cpc_dropout, linear2 GEMM, cpc_add_ln_fwd;
    backward: the same kernels engine.AttentionContext.backward issues for a layer (cpc_ln_bwd, masked data-gradient GEMMs,
    cpc_attn_bwd, reduction-form GEMMs for the weight gradients).  Dropout (train mode, p > 0): the counter-based masks of
    include/cpc_hip.h with sites site0 ... site0 + 3, regenerated in the backward pass."""

    @staticmethod
    def forward(ctx, X, w_in, b_in, w_o, b_o, w1, b1, w2, b2, n1w, n1b, n2w, n2b, cfg):
        from . import _hip
        B, S, C, heads, FF, dp, seed, site0, eps1, eps2 = cfg
        P, F32 = _hip.ptr, _hip.F32
        M, dev = B * S, X.device
        f = lambda t: t.detach().float().contiguous()
        X, w_in, b_in, w_o, b_o, w1, b1, w2, b2, n1w, n1b, n2w, n2b = (f(t) for t in (X, w_in, b_in, w_o, b_o, w1, b1, w2, b2, n1w, n1b, n2w, n2b))
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        qkv, att, probs, y, r1, x1, f1, r2, out = new(M, 3 * C), new(M, C), new(B * heads, S, S), new(M, C), new(M, C), new(M, C), new(M, FF), new(M, C), new(M, C)
        st1, st2 = new(M, 2), new(M, 2)
        _hip.gemm_nt(P(X), P(w_in), P(qkv), M, 3 * C, C, C, C, 3 * C, F32, bias=P(b_in))
        _hip.call("cpc_attn_fwd", P(qkv), P(att), P(probs), B, S, C, heads, dp, seed, site0 + 0, F32)
        _hip.gemm_nt(P(att), P(w_o), P(y), M, C, C, C, C, C, F32, bias=P(b_o))
        _hip.call("cpc_add_ln_fwd", P(X), P(y), P(n1w), P(n1b), P(r1), P(x1), P(st1), M, C, eps1, dp, seed, site0 + 1, F32)
        _hip.gemm_nt(P(x1), P(w1), P(f1), M, FF, C, C, C, FF, F32, bias=P(b1), flags=_hip.GEMM_RELU)
        if dp > 0.0:
            _hip.call("cpc_dropout", P(f1), M * FF, dp, seed, site0 + 2, F32)
        _hip.gemm_nt(P(f1), P(w2), P(y), M, C, FF, FF, FF, C, F32, bias=P(b2))
        _hip.call("cpc_add_ln_fwd", P(x1), P(y), P(n2w), P(n2b), P(r2), P(out), P(st2), M, C, eps2, dp, seed, site0 + 3, F32)
        ctx.cfg = cfg
        ctx.save_for_backward(X, w_in, w_o, w1, w2, n1w, n2w, qkv, att, probs, r1, x1, f1, r2, st1, st2)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _hip
        B, S, C, heads, FF, dp, seed, site0, eps1, eps2 = ctx.cfg
        X, w_in, w_o, w1, w2, n1w, n2w, qkv, att, probs, r1, x1, f1, r2, st1, st2 = ctx.saved_tensors
        P, F32 = _hip.ptr, _hip.F32
        M, dev = B * S, X.device
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        g = g.detach().float().contiguous()

        def linear_grads(dy, x, rows, cols):          # (d weight (rows, cols), d bias (rows)) of y = x W^T + b
            dw = new(rows, cols)
            _hip.gemm_tn(P(dy), P(x), P(dw), M, rows, cols, rows, cols, cols, F32, flags=_hip.GEMM_OUT_F32)
            return dw, _colsum(dy, M, rows)

        # norm2 over r2 = x1 + dropout(linear2(...))
        gB, gBd, dn2w, dn2b = _ln_backward(g, None, r2, st2, n2w, M, C, dp, seed, site0 + 3)
        gy2 = gBd if gBd is not None else gB
        dw2, db2 = linear_grads(gy2, f1, C, FF)
        # d relu(.) dropout = keep / (1 - p): the keep part comes with the mask of the stored, dropped activation
        w2t = (w2.t() * (1.0 / (1.0 - dp) if dp > 0.0 else 1.0)).contiguous()
        df1 = new(M, FF)
        _hip.gemm_nt(P(gy2), P(w2t), P(df1), M, FF, C, C, C, FF, F32, mask=P(f1))
        dw1, db1 = linear_grads(df1, x1, FF, C)
        gC = new(M, C)
        _hip.gemm_nt(P(df1), P(w1.t().contiguous()), P(gC), M, C, FF, FF, FF, C, F32)
        # norm1 over r1 = x + dropout(out_proj(attention))
        gA, gAd, dn1w, dn1b = _ln_backward(gB, gC, r1, st1, n1w, M, C, dp, seed, site0 + 1)
        gy1 = gAd if gAd is not None else gA
        dwo, dbo = linear_grads(gy1, att, C, C)
        datt, dqkv, gD = new(M, C), new(M, 3 * C), new(M, C)
        _hip.gemm_nt(P(gy1), P(w_o.t().contiguous()), P(datt), M, C, C, C, C, C, F32)
        _hip.call("cpc_attn_bwd", P(qkv), P(probs), P(datt), P(dqkv), B, S, C, heads, dp, seed, site0 + 0, F32)
        dwin, dbin = linear_grads(dqkv, X, 3 * C, C)
        _hip.gemm_nt(P(dqkv), P(w_in.t().contiguous()), P(gD), M, C, 3 * C, 3 * C, 3 * C, C, F32)
        return gA + gD, dwin, dbin, dwo, dbo, dw1, db1, dw2, db2, dn1w, dn1b, dn2w, dn2b, None


class _NormFn(torch.autograd.Function):
    """The encoder's final LayerNorm on rows (cpc_add_ln_fwd / cpc_ln_bwd)."""

    @staticmethod
    def forward(ctx, X, w, b, eps):
        from . import _hip
        X, w, b = (t.detach().float().contiguous() for t in (X, w, b))
        M, C = X.shape
        out, stats = torch.empty_like(X), torch.empty(M, 2, device=X.device, dtype=torch.float32)
        _hip.call("cpc_add_ln_fwd", _hip.ptr(X), None, _hip.ptr(w), _hip.ptr(b), None, _hip.ptr(out), _hip.ptr(stats), M, C, eps, 0.0, 0, 0,
                  _hip.F32)
        ctx.save_for_backward(X, w, stats)
        return out

    @staticmethod
    def backward(ctx, g):
        X, w, stats = ctx.saved_tensors
        M, C = X.shape
        dr, _, dw, db = _ln_backward(g.detach().float().contiguous(), None, X, stats, w, M, C, 0.0, 0, 0)
        return dr, dw, db, None


def _layer_rows(layer, X, B, S, seed, site0):
    """``layer`` applied to rows X (B*S, C) float32 (differentiable with respect to X and the layer's parameters)."""
    C, heads, FF = _layer_check(layer, S)
    attn = layer.self_attn
    dp = float(layer.dropout.p) if layer.training else 0.0
    cfg = (B, S, C, heads, FF, dp, int(seed), int(site0), float(layer.norm1.eps), float(layer.norm2.eps))
    return _LayerFn.apply(X, attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias, layer.linear1.weight,
                          layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias,
                          layer.norm2.weight, layer.norm2.bias, cfg)


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
        HIP kernels, differentiable with respect to ``src`` and the layer's parameters (_LayerFn).  The causal ``src_mask`` AttentionModel
        passes only; no key padding mask."""
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
        differentiable; causal ``mask`` (see TransformerEncoderLayer.forward)."""
        if src_key_padding_mask is not None:
            raise NotImplementedError("key padding masks are not part of the HIP path (the reference never passes one)")
        X, B, S, C = _standalone_input(self, src, mask)
        seed = _dropout_seed() if self.training else 0
        for l, layer in enumerate(self.layers):
            X = _layer_rows(layer, X, B, S, seed, 4 * l)
        if self.norm is not None:
            X = _NormFn.apply(X, self.norm.weight, self.norm.bias, float(self.norm.eps))
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
