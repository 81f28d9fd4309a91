"""Drop-in model surface: AudioEncoder, AudioGRUModel, AudioPredictiveCodingModel (+ activation taps).

Same constructor arguments, attributes, state_dict keys and forward return values as the reference's
``audio_model.py`` (AudioEncoder :14-44, AudioGRUModel :47-77, AudioPredictiveCodingModel :164-219, ActivationWriter :274-284), but ``forward`` runs the hand-written HIP path (``engine.CPCEngine``) instead of ATen
convolutions / GRUCell loops.  Parameters stay ``nn.Parameter``s in the reference's shapes (so checkpoints and
``model.parameters()``-built optimizers keep working); on the device they are views into one flat f32 buffer.

There is no CPU implementation here: calling ``forward`` with a CPU tensor raises.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

encoder_default_dict = {'strides': [5, 4, 2, 2, 2],
                        'kernel_sizes': [10, 8, 4, 4, 4],
                        'channel_count': [512, 512, 512, 512, 512],
                        'bias': True}

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32,
           torch.bfloat16: torch.bfloat16, torch.float32: torch.float32}


class _ConvParams(nn.Module):
    """Holds ``weight`` (out, in, k) and optional ``bias`` with nn.Conv1d's default initialisation; no forward."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = (kernel_size,), (stride,)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_channels * self.kernel_size[0])
            nn.init.uniform_(self.bias, -bound, bound)


class _GRUCellParams(nn.Module):
    """weight_ih / weight_hh / bias_ih / bias_hh with nn.GRUCell's names, shapes and initialisation; no forward."""

    def __init__(self, input_size, hidden_size, bias=True):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weight_ih = nn.Parameter(torch.empty(3 * hidden_size, input_size))
        self.weight_hh = nn.Parameter(torch.empty(3 * hidden_size, hidden_size))
        if bias:
            self.bias_ih = nn.Parameter(torch.empty(3 * hidden_size))
            self.bias_hh = nn.Parameter(torch.empty(3 * hidden_size))
        else:
            self.register_parameter("bias_ih", None)
            self.register_parameter("bias_hh", None)
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)


class AudioEncoder(nn.Module):
    """Strided 1-D conv stack, relu between layers, none after the last (reference audio_model.py:14-44)."""

    def __init__(self, args_dict=encoder_default_dict):
        super().__init__()
        self.strides = list(args_dict['strides'])
        self.kernel_sizes = list(args_dict['kernel_sizes'])
        self.channel_count = list(args_dict['channel_count'])
        self.num_layers = len(self.strides)
        self.downsampling_factor = np.prod(self.strides)
        self.receptive_field = self.kernel_sizes[0]
        hop = 1
        for i in range(1, self.num_layers):
            hop *= self.strides[i - 1]
            self.receptive_field += (self.kernel_sizes[i] - 1) * hop
        self.layers = nn.ModuleList()
        for l in range(self.num_layers):
            self.layers.append(_ConvParams(1 if l == 0 else self.channel_count[l - 1], self.channel_count[l],
                                           self.kernel_sizes[l], self.strides[l], bias=args_dict['bias']))

    def forward(self, x):
        """x (B, 1, L) on the GPU -> (B, C, T) float32, differentiable with respect to the encoder's parameters and its input (stand-alone
        calls go through the autograd bridge _EncoderForward)."""
        owner = _standalone_owner(self)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            owner.engine_for(x)                           # (flattens the owner's parameters on first use)
            names = [n for n, _ in owner.named_parameters() if n.startswith("encoder.")]
            return _EncoderForward.apply(owner, names, x, *[dict(owner.named_parameters())[n] for n in names])
        return owner.encode(x)


class AudioGRUModel(nn.Module):
    """GRU context network; ``forward(input)`` takes (batch, input_size, steps) and returns the last hidden state
    (reference audio_model.py:47-77).  ``reset_hidden=False`` carries the last hidden state of one call into the next (``self.hidden``,
    as in the reference, :69 / :75): forward only -- a call that starts from a carried state cannot be differentiated (the reference's
    autograd raises there too, the previous call's graph being gone)."""

    def __init__(self, input_size, hidden_size, bias=True, reset_hidden=True):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.gruCell = _GRUCellParams(input_size, hidden_size, bias)
        self.hidden = None
        self.reset_hidden = reset_hidden

    def forward(self, input):
        """(batch, input_size, steps) on the GPU -> last hidden state (batch, hidden_size), differentiable with respect to the
        input and the parameters (stand-alone calls go through the autograd bridge _ContextForward)."""
        from .engine import standalone_context_forward
        return standalone_context_forward(self, input, self.hidden_size)


class ConvolutionalArBlock(nn.Module):
    """[MaxPool1d(pooling, ceil)] -> Conv1d -> [BatchNorm1d] -> ReLU (+ optional residual branch); parameter holder with the
    reference's module indices (audio_model.py:80-136), so ``main_modules.N.weight`` keys line up."""

    def __init__(self, in_channels, out_channels, kernel_size, pooling=1, stride=1, bias=True, residual=False, batch_norm=False,
                 name='ar_block', activation_register=None):
        super().__init__()
        self.name = name
        self.main_modules = nn.ModuleList()
        if pooling > 1:
            self.main_modules.append(nn.MaxPool1d(pooling, ceil_mode=True))
        self.main_modules.append(_ConvParams(in_channels, out_channels, kernel_size, stride, bias=bias))
        if batch_norm:
            self.main_modules.append(nn.BatchNorm1d(out_channels))
        self.main_modules.append(nn.ReLU())
        self.residual = residual
        self.residual_modules = None
        if residual:
            self.residual_modules = nn.ModuleList()
            if pooling * stride > 1:
                self.residual_modules.append(nn.MaxPool1d(pooling * stride, ceil_mode=True))
            if in_channels != out_channels:
                self.residual_modules.append(_ConvParams(in_channels, out_channels, 1, 1, bias=True))
        self.output_activation_writer = ActivationWriter(register=activation_register, name=self.name)

    def forward(self, x):
        raise NotImplementedError("ConvolutionalArBlock runs fused inside AudioPredictiveCodingModel.forward on the HIP path")


class ConvolutionalArModel(nn.Module):
    """Convolutional context network (reference audio_model.py:139-161); ``forward`` would return x[:, :, -1]."""

    def __init__(self, args_dict):
        super().__init__()
        self.kernel_sizes = list(args_dict['kernel_sizes'])
        self.channel_count = list(args_dict['channel_count'])
        self.strides = list(args_dict['stride'])
        self.poolings = list(args_dict['pooling'])
        self.batch_norm = bool(args_dict['batch_norm'])
        self.residual = bool(args_dict['residual'])
        self.module_list = nn.ModuleList()
        for l in range(len(self.kernel_sizes)):
            self.module_list.append(ConvolutionalArBlock(in_channels=self.channel_count[l], out_channels=self.channel_count[l + 1],
                                                         kernel_size=self.kernel_sizes[l], stride=self.strides[l],
                                                         pooling=self.poolings[l], bias=args_dict['bias'],
                                                         batch_norm=self.batch_norm, residual=self.residual,
                                                         name='ar_block_' + str(l),
                                                         activation_register=args_dict.get('activation_register')))
        self.encoding_size = self.channel_count[0]
        self.ar_size = self.channel_count[-1]

    def forward(self, x):
        """(batch, channels, steps) on the GPU -> the last position (batch, ar_size).  Inference only when called
        stand-alone (BatchNorm follows self.training); gradients flow through AudioPredictiveCodingModel."""
        from .engine import standalone_context_forward
        return standalone_context_forward(self, x, self.ar_size)


class AudioPredictiveCodingModel(nn.Module):
    """encoder -> (targets, z) -> autoregressive context c -> W_k c predictions (reference audio_model.py:164-219).

    ``compute_dtype``: "bf16" (default; bf16 storage, f32 accumulate — BASELINE config 2) or "fp32" (exact-f32 MFMA,
    the parity mode)."""

    def __init__(self, encoder, autoregressive_model, enc_size, ar_size, visible_steps=100, prediction_steps=12,
                 activation_register=None, compute_dtype="bf16"):
        super().__init__()
        self.enc_size = enc_size
        self.ar_size = ar_size
        self.visible_steps = visible_steps
        self.prediction_steps = prediction_steps
        self.encoder = encoder
        self.autoregressive_model = autoregressive_model
        self.prediction_model = _LinearParams(ar_size, enc_size * prediction_steps)
        self.activation_register = activation_register
        self.input_activation_writer = ActivationWriter(register=self.activation_register, name='scalogram')
        self.z_activation_writer = ActivationWriter(register=self.activation_register, name='z_code')
        self.c_activation_writer = ActivationWriter(register=self.activation_register, name='c_code')
        self.prediction_activation_writer = ActivationWriter(register=self.activation_register, name='prediction')
        self.compute_dtype = _DTYPES[compute_dtype]
        self._engines = {}
        self._flat_param = None
        self._flat_grad = None
        self._param = {}
        self._grad = {}
        from .attention_model import AttentionModel
        from .scalogram_model import ScalogramResidualEncoder
        self._scalogram = isinstance(encoder, ScalogramResidualEncoder)
        if not isinstance(encoder, (AudioEncoder, ScalogramResidualEncoder)) or \
                not isinstance(autoregressive_model, (AudioGRUModel, ConvolutionalArModel, AttentionModel, ScalogramResidualEncoder)):
            raise NotImplementedError("the HIP path covers AudioEncoder / ScalogramResidualEncoder + AudioGRUModel / "
                                      "ConvolutionalArModel / AttentionModel (SURVEY.md section 8 rows a1-a10)")

    @property
    def item_length(self):
        item_length = self.encoder.receptive_field
        item_length += (self.visible_steps + self.prediction_steps) * self.encoder.downsampling_factor
        return item_length

    def parameter_count(self):
        return sum(int(np.prod(p.shape)) for p in self.parameters())

    # ------------------------------------------------------------------ flat parameter storage
    def _flatten_parameters(self, device):
        """Moves all parameters into ONE f32 buffer on ``device`` (each a 256-byte aligned view) and creates the
        parallel flat gradient buffer the engine writes (``self._grad`` views; ``link_grads`` exposes them as
        ``p.grad``).  Idempotent."""
        device = torch.device(device)
        named = list(self.named_parameters())
        if self._flat_param is not None and self._flat_param.device == device and \
                all(p.data_ptr() == self._param[n].data_ptr() for n, p in named):
            return
        offsets, total = {}, 0
        for n, p in named:
            offsets[n] = total
            total += (p.numel() + 63) // 64 * 64
        flat = torch.zeros(total, device=device, dtype=torch.float32)
        grad = torch.zeros(total, device=device, dtype=torch.float32)
        self._param, self._grad, self._offset = {}, {}, dict(offsets)
        with torch.no_grad():
            for n, p in named:
                view = flat[offsets[n]:offsets[n] + p.numel()].view(p.shape)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                gview = grad[offsets[n]:offsets[n] + p.numel()].view(p.shape)
                self._param[n], self._grad[n] = view, gview
        self._flat_param, self._flat_grad = flat, grad
        self._engines = {}

    def link_grads(self):
        """Makes every parameter's ``.grad`` the engine's gradient view (used by the fused trainer path, where autograd
        is not involved; e.g. for the reference's grad_mean_var helper)."""
        for n, p in self.named_parameters():
            p.grad = self._grad[n]

    def engine(self, batch_size, length, device=None):
        """The cached train-step engine for ``batch_size`` clips of ``length`` samples — or, with a
        ScalogramResidualEncoder, for scalogram batches of shape ``length = (channels, bins, frames)``."""
        from .engine import CPCEngine
        if device is None:
            device = next(self.parameters()).device
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the CPC hot path runs on the GPU only (no CPU fallback): call model.to('cuda') first")
        shape = tuple(int(v) for v in length) if self._scalogram else int(length)
        gp = bool(getattr(self, "gradient_penalty_engine", False)) and self._scalogram
        key = (int(batch_size), shape, self.compute_dtype, str(device), gp)
        eng = self._engines.get(key)
        if eng is None or self._flat_param is None or any(
                p.data_ptr() != self._param[n].data_ptr() for n, p in self.named_parameters()):
            self._flatten_parameters(device)
            if self._scalogram:
                from .scalogram_engine import ScalogramCPCEngine
                eng = ScalogramCPCEngine(self, (int(batch_size),) + shape, device, self.compute_dtype, gradient_penalty=gp)
            else:
                eng = CPCEngine(self, batch_size, length, device, self.compute_dtype)
            self._engines[key] = eng
        return eng

    def engine_for(self, x):
        """Engine matching an input batch: (B, 1, L) waveforms or (B, C, bins, frames) scalograms."""
        if self._scalogram:
            if x.dim() != 4:
                raise ValueError("expected a scalogram batch of shape (batch, channels, bins, frames)")
            return self.engine(x.shape[0], tuple(x.shape[1:]), x.device)
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError("expected input of shape (batch, 1, samples)")
        return self.engine(x.shape[0], x.shape[2], x.device)

    def encode(self, x):
        eng = self.engine_for(x)
        xin = x.detach().float() if x.dim() == 4 else x.detach()[:, 0, :].contiguous().float()
        eng.prepare_weights()
        eng.encoder_forward(xin)
        return eng.view_top()[:, :eng.T, :].float().transpose(1, 2)

    def forward(self, x):
        """x (B, 1, L) -> (predicted_z (B,K,E), targets (B,E,K), z (B,E,V), c (B,H)), autograd-connected to the
        parameters (targets are not detached, as in the reference audio_model.py:197)."""
        x = self.input_activation_writer(x)
        eng = self.engine_for(x)
        params = [p for _, p in self.named_parameters()]
        predicted_z, targets, z, c = _CPCForward.apply(eng, x, *params)
        z = self.z_activation_writer(z)
        c = self.c_activation_writer(c)
        predicted_z = self.prediction_activation_writer(predicted_z)
        return predicted_z, targets, z, c


class _LinearParams(nn.Module):
    """``weight`` (out, in) with nn.Linear's default initialisation, no bias; no forward."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class _CPCForward(torch.autograd.Function):
    """Autograd bridge: forward = engine.forward, backward = engine.backward fed with the incoming gradients."""

    @staticmethod
    def forward(ctx, eng, x, *params):
        xin = x.detach().float() if x.dim() == 4 else x.detach()[:, 0, :].contiguous().float()
        eng.forward(xin)
        ctx.eng, ctx.xin = eng, xin
        ctx.set_materialize_grads(False)
        pred, targets, z, c = eng.outputs()
        return pred, targets, z, c

    @staticmethod
    def backward(ctx, d_pred, d_targets, d_z, d_c):
        eng = ctx.eng
        B, E, K, V, T = eng.B, eng.E, eng.K, eng.V, eng.T
        dtop = eng.dact[-1].view(B, eng.geo.alloc[-1], E)
        if d_pred is None:
            eng.dpred.zero_()
        else:
            eng.dpred.view(B, K, E).copy_(d_pred)
        if d_targets is None:
            dtop[:, T - K:T, :].zero_()
        else:
            dtop[:, T - K:T, :].copy_(d_targets.transpose(1, 2))
        eng.backward(ctx.xin, add_dc=d_c, add_dz=d_z)
        grads = [eng.model._grad[n].clone() for n, _ in eng.model.named_parameters()]
        return (None, None, *grads)


class _EncoderForward(torch.autograd.Function):
    """Autograd bridge of a stand-alone encoder call (reference modules are ordinary differentiable nn.Modules, audio_model.py:36-44):
    forward = the engine's encoder pass, backward = its encoder backward pass fed with the incoming gradient.  An AudioEncoder also
    gives the gradient with respect to its input (CPCEngine.input_gradient: one more overlapped-row GEMM over layer 1's output
    gradient); a ScalogramResidualEncoder does not."""

    @staticmethod
    def forward(ctx, owner, names, x, *params):
        eng = owner.engine_for(x)
        ctx.need_dx = bool(x.requires_grad)
        if ctx.need_dx and getattr(owner, "_scalogram", False):
            raise NotImplementedError("a stand-alone scalogram encoder call gives no gradient with respect to its input: detach the input, "
                                      "or train through AudioPredictiveCodingModel (the gradient penalty's engine forms it there)")
        xin = x.detach().float() if x.dim() == 4 else x.detach()[:, 0, :].contiguous().float()
        eng.prepare_weights()
        eng.encoder_forward(xin)
        # the backward pass reads the engine's activation buffers: stamp this forward pass, so that a later one (which overwrites
        # them) is noticed instead of silently differentiating the wrong activations
        eng._standalone_stamp = getattr(eng, "_standalone_stamp", 0) + 1
        ctx.eng, ctx.xin, ctx.names, ctx.stamp, ctx.x_shape = eng, xin, names, eng._standalone_stamp, tuple(x.shape)
        return eng.view_top()[:, :eng.T, :].float().transpose(1, 2)

    @staticmethod
    def backward(ctx, d_out):
        eng = ctx.eng
        if getattr(eng, "_standalone_stamp", 0) != ctx.stamp:
            raise RuntimeError("another forward pass of this encoder ran before backward(): the engine keeps ONE set of activations "
                               "(call backward() before the next forward, as a train loop does)")
        B, E, T = eng.B, eng.E, eng.T
        dtop = eng.dact[-1].view(B, eng.geo.alloc[-1], E)
        dtop.zero_()
        dtop[:, :T, :].copy_(d_out.transpose(1, 2))
        eng._ahead = None
        fused = getattr(eng, "fuse_c1", False)
        if ctx.need_dx and fused:
            eng.fuse_c1 = False          # the input gradient needs layer 1's output gradient in memory: the unfused layer-2 / layer-1 route
        try:
            eng._backward_encoder(ctx.xin)
        finally:
            if ctx.need_dx and fused:
                eng.fuse_c1 = True
        for fn in getattr(eng, "_deferred_side", ()):
            fn()
        eng._deferred_side = ()
        if eng.use_aux:
            torch.cuda.current_stream().wait_stream(eng.aux)
        dx = eng.input_gradient().view(ctx.x_shape) if ctx.need_dx else None
        return (None, None, dx, *[eng.model._grad[n].clone() for n in ctx.names])


class _ContextForward(torch.autograd.Function):
    """Autograd bridge of a stand-alone context-network call (reference audio_model.py:66-77, :139-161; attention_model.py:72-82):
    gradients with respect to z and to the network's parameters."""

    @staticmethod
    def forward(ctx, eng, names, z, *params):
        ctx.eng, ctx.names = eng, names
        return eng.run(z.float())

    @staticmethod
    def backward(ctx, d_c):
        eng = ctx.eng
        dz = eng.backward_context(d_c.float())
        return (None, None, dz.clone(), *[eng.model._grad[n].clone() for n in ctx.names])


def _standalone_owner(encoder):
    """A stand-alone AudioEncoder call needs an engine; it is hosted by a private model wrapper cached on the encoder."""
    owner = getattr(encoder, "_owner", None)
    if owner is None:
        c = encoder.channel_count[-1] if hasattr(encoder, "channel_count") else encoder.blocks[-1].cfg['out_channels']
        gru = AudioGRUModel(c, 32)
        object.__setattr__(encoder, "_owner", None)
        owner = AudioPredictiveCodingModel.__new__(AudioPredictiveCodingModel)
        nn.Module.__init__(owner)
        owner.enc_size, owner.ar_size = c, 32
        owner.visible_steps, owner.prediction_steps = 0, 0
        owner.encoder = encoder
        owner.autoregressive_model = gru
        owner.prediction_model = _LinearParams(32, c)
        owner.compute_dtype = getattr(encoder, "compute_dtype", torch.float32)
        owner._engines, owner._flat_param, owner._flat_grad, owner._param, owner._grad = {}, None, None, {}, {}
        owner._scalogram = not isinstance(encoder, AudioEncoder)
        object.__setattr__(encoder, "_owner", owner)
    dev = next(encoder.parameters()).device
    if next(owner.autoregressive_model.parameters()).device != dev:
        owner.autoregressive_model.to(dev)
        owner.prediction_model.to(dev)
    return owner


class ActivationWriter(nn.Module):
    """Pass-through tap (reference audio_model.py:274-284).  The hot path never registers anything: ``register`` is None in
    every training configuration, and the module is kept only because it sits in the reference's module tree (ModuleList
    indices, and with them the state_dict keys, count it).  A caller-supplied register object is served through the one
    method the reference's writer uses, ``write_activation(name, value)``; ``ActivationRegister`` below is the host-side
    container the reference's tools hand in (dreaming / activation statistics, SURVEY.md section 2 rows 12-13: outside the
    hot path, kept for import compatibility)."""

    def __init__(self, register, name):
        super().__init__()
        self.register, self.name = register, name

    def forward(self, x):
        if self.register is not None:
            self.register.write_activation(self.name, x)
        return x


class ActivationRegister:
    """Host-side container for tapped activations, same constructor and methods as the reference's (audio_model.py:222-271):
    one ordered dictionary, or one per device index when ``devices`` is given (get_activations then concatenates the per-device
    entries on the first device).  Pure bookkeeping around tensors a caller's ActivationWriter hands in; the train step never
    registers anything."""

    def __init__(self, writing_condition=None, clone_activations=False, batch_filter=None, move_to_cpu=False, devices=None):
        from collections import OrderedDict
        self.devices = devices
        self.activations = OrderedDict() if devices is None else {d: OrderedDict() for d in devices}
        self.active = True
        self.writing_condition, self.clone_activations = writing_condition, clone_activations
        self.batch_filter, self.move_to_cpu = batch_filter, move_to_cpu

    def write_activation(self, name, value):
        if not self.active or (self.writing_condition is not None and not self.writing_condition(value)):
            return
        if self.batch_filter is not None:
            value = value[self.batch_filter]
        slot = self.activations if self.devices is None else self.activations[value.device.index]
        if self.move_to_cpu:
            value = value.cpu()
        slot[name] = value.clone() if self.clone_activations else value

    def get_activations(self):
        if self.devices is None:
            return self.activations
        first = self.devices[0]
        return {key: torch.cat([self.activations[d][key].to(torch.device("cuda", first)) for d in self.devices], dim=0)
                for key in self.activations[first].keys()}


def load_to_cpu(path):
    """torch.load of a whole-module snapshot onto the CPU (reference audio_model.py:287-290)."""
    model = torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)
    model.cpu()
    return model


def cuda0_writing_condition(x):
    """Writing condition: keep what lives on the CPU or on the first GPU (reference audio_model.py:300-307)."""
    return x.device.index == 0 if x.device.type == 'cuda' else True


def num_parameters(model):
    return sum(int(np.prod(p.shape)) for p in model.parameters())
