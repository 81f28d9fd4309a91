"""Host-side mirror of the reference's attention context network (reference attention_model.py:9-82 and the vendored
transformer.py:131-171, :223-272): the same constructor arguments, attribute names and ``state_dict`` keys, so checkpoints
move in both directions.  The transformer containers here only HOLD parameters (standard torch containers, default initialisers, the
reference's construction order); the arithmetic runs in the HIP kernels behind ``engine.AttentionContext`` when the model is
used as ``AudioPredictiveCodingModel.autoregressive_model`` (or called stand-alone as ``AttentionModel(z)``, inference only).
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


class TransformerEncoderLayer(nn.Module):
    """Post-norm encoder layer parameters: self_attn (in_proj / out_proj), linear1, linear2, norm1, norm2
    (reference transformer.py:241-252)."""

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

    def forward(self, *args, **kwargs):
        raise NotImplementedError("a stand-alone TransformerEncoderLayer call is not part of the HIP path: the layer runs inside "
                                  "AttentionModel (engine.AttentionContext); call AttentionModel(z) or the whole model "
                                  "(INTEGRATION.md, 'Deviations from the reference surface')")


class TransformerEncoder(nn.Module):
    """``num_layers`` deep copies of one layer (so all layers start from identical values, as in the reference
    transformer.py:144-148, :339-340) and an optional final norm."""

    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, *args, **kwargs):
        raise NotImplementedError("a stand-alone TransformerEncoder call is not part of the HIP path: it runs inside AttentionModel "
                                  "(engine.AttentionContext); call AttentionModel(z) or the whole model "
                                  "(INTEGRATION.md, 'Deviations from the reference surface')")


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
