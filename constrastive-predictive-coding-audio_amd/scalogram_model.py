"""Scalogram front end and 2-D residual encoder behind the reference's names (reference scalogram_model.py:34-102,
:372-529): ``PreprocessingModule`` (CQT -> log power [+ phase difference]) and ``ScalogramResidualEncoder`` /
``ScalogramEncoderBlock`` parameter holders whose arithmetic runs in the HIP kernels of csrc/scalogram.hip.
"""
import math

import torch
import torch.nn as nn

from . import _hip
from .constant_q_transform import CQT, PhaseDifference
from .audio_model import ActivationWriter

cqt_default_dict = {'sample_rate': 16000,
                    'fmin': 30,
                    'n_bins': 256,
                    'bins_per_octave': 32,
                    'filter_scale': 0.5,
                    'hop_length': 128,
                    'trainable_cqt': False}


class PreprocessingModule(nn.Module):
    """(B, 1, L) waveform -> (B, {1,2}, n_bins, frames) scalogram, float32 (reference scalogram_model.py:34-102).

    The result is a strided NCHW *view* of a channels-last buffer [B][frames][bins][channels]; ScalogramResidualEncoder
    consumes that buffer without a copy.  Not differentiable (the reference's CQT is frozen and the trainer only sets
    requires_grad on the output)."""

    def __init__(self, cqt_dict=None, phase=False, output_requires_grad=False, offset_zero=False, output_power=1.,
                 pooling=None, scaling=1.):
        super().__init__()
        self.downsampling_factor = 1
        self.receptive_field = 1
        self.cqt = None
        if cqt_dict is not None:
            self.cqt = CQT(sr=cqt_dict['sample_rate'], fmin=cqt_dict['fmin'], n_bins=cqt_dict['n_bins'],
                           bins_per_octave=cqt_dict['bins_per_octave'], filter_scale=cqt_dict['filter_scale'],
                           hop_length=cqt_dict['hop_length'], trainable=cqt_dict['trainable_cqt'],
                           filters=cqt_dict.get('filters'))
            self.downsampling_factor = cqt_dict['hop_length']
            self.receptive_field = self.cqt.conv_kernel_sizes[0]
        self.phase_diff = None
        if phase:
            self.phase_diff = PhaseDifference(sr=cqt_dict['sample_rate'], fmin=cqt_dict['fmin'], n_bins=cqt_dict['n_bins'],
                                              bins_per_octave=cqt_dict['bins_per_octave'], hop_length=cqt_dict['hop_length'])
        self.output_power = output_power
        if offset_zero:
            self.offset = 1e-9
            self.log_offset = -math.log(self.offset)
            self.normalization_factor = scaling / self.log_offset
        else:
            self.offset = 0
            self.log_offset = 0
            self.normalization_factor = scaling
        self.pooling = pooling
        if pooling is not None:
            self.downsampling_factor *= pooling[1]
        self.output = None

    def forward(self, x):
        if self.cqt is None:
            return x
        ph, pw = (int(self.pooling[0]), int(self.pooling[1])) if self.pooling is not None else (1, 1)
        cq, Tn, ldq = self.cqt.transform(x)
        B, bins = cq.shape[0], self.cqt.n_bins
        phase = self.phase_diff is not None
        W, Cc = (Tn - 1, 2) if phase else (Tn, 1)
        if W < 1:
            raise ValueError("clip too short for a phase-difference scalogram (needs at least two CQT frames)")
        if W // pw < 1 or bins // ph < 1:
            raise ValueError("scalogram smaller than the pooling window")
        out = torch.empty(B, W // pw, bins // ph, Cc, device=cq.device, dtype=torch.float32)
        fixed = self.phase_diff.fixed_phase_diff.detach().to(cq.device).reshape(-1).contiguous() if phase else None
        scale = self.phase_diff.scaling.detach().to(cq.device).reshape(-1).contiguous() if phase else None
        _hip.call("cpc_scalogram_pointwise", _hip.ptr(cq), _hip.ptr(fixed), _hip.ptr(scale), _hip.ptr(out), B, Tn, bins, ldq,
                  1 if phase else 0, float(self.offset), float(self.log_offset), float(self.normalization_factor),
                  float(self.output_power), ph, pw)
        x = out.permute(0, 3, 2, 1)              # (B, channels, bins, frames) view
        self.output = x
        return x


default_encoder_block_dict = {'in_channels': 64,
                              'hidden_channels': None,
                              'out_channels': 64,
                              'kernel_size_1': (3, 3),
                              'kernel_size_2': (3, 3),
                              'top_padding_1': None,
                              'top_padding_2': None,
                              'padding_1': 0,
                              'padding_2': 0,
                              'stride_1': 1,
                              'stride_2': 1,
                              'pooling_1': 1,
                              'pooling_2': 1,
                              'bias': True,
                              'separable': False,
                              'residual': True,
                              'batch_norm': False}


class Conv2dSeparable(nn.Module):
    """Depthwise k x k convolution (groups = in_channels, no bias) followed by a 1 x 1 convolution — parameter holder with the
    reference's sub-module names and construction order (scalogram_model.py:532-544); runs as scalogram_engine._SepConv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        if dilation != 1:
            raise NotImplementedError("dilated separable convolutions are not part of the HIP path")
        self.conv = nn.Conv2d(in_channels=in_channels, out_channels=in_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, dilation=dilation, bias=False, groups=in_channels)
        self.conv_1x1 = nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels

    @property
    def weight(self):
        return self.conv.weight

    def forward(self, x):
        raise NotImplementedError("Conv2dSeparable runs inside AudioPredictiveCodingModel.forward on the HIP path")


class ScalogramEncoderBlock(nn.Module):
    """conv_a -> [BN] -> ReLU -> [top pad] -> conv_b -> [BN] -> ReLU, plus the pooled / 1x1-projected, cropped residual
    branch (reference scalogram_model.py:372-479).  Parameter holder: standard torch modules at the reference's
    ``main_modules`` / ``residual_modules`` indices (so state_dict keys and default initialisation coincide); the
    arithmetic runs in scalogram_engine.ScalogramCPCEngine."""

    def __init__(self, args_dict=default_encoder_block_dict, name='scalogram_block', activation_register=None):
        super().__init__()
        self.name = name
        if args_dict['hidden_channels'] is None:
            args_dict['hidden_channels'] = args_dict['out_channels']
        a = args_dict
        conv_module = Conv2dSeparable if a['separable'] else nn.Conv2d
        self.cfg = {k: a[k] for k in ('in_channels', 'hidden_channels', 'out_channels', 'kernel_size_1', 'kernel_size_2',
                                      'top_padding_1', 'top_padding_2', 'padding_1', 'padding_2', 'stride_1', 'stride_2',
                                      'pooling_1', 'pooling_2', 'bias', 'residual', 'batch_norm')}
        self.cfg['separable'] = bool(a['separable'])
        self.cfg['ceil_pooling'] = bool(a.get('ceil_pooling', False))
        self.main_modules = nn.ModuleList()
        self.index = {}
        for tag, cin, cout in (('1', a['in_channels'], a['hidden_channels']), ('2', a['hidden_channels'], a['out_channels'])):
            if a['top_padding_' + tag] is not None:
                self.main_modules.append(nn.ZeroPad2d((0, 0, a['top_padding_' + tag], 0)))
            self.index['conv_' + tag] = len(self.main_modules)
            self.main_modules.append(conv_module(in_channels=cin, out_channels=cout, kernel_size=a['kernel_size_' + tag],
                                                 bias=a['bias'], padding=a['padding_' + tag], stride=a['stride_' + tag]))
            if a['batch_norm']:
                self.index['bn_' + tag] = len(self.main_modules)
                self.main_modules.append(nn.BatchNorm2d(cout))
            if a['pooling_' + tag] > 1:
                self.main_modules.append(nn.MaxPool2d(kernel_size=a['pooling_' + tag], ceil_mode=self.cfg['ceil_pooling']))
            self.main_modules.append(nn.ReLU())
            self.main_modules.append(ActivationWriter(register=activation_register, name=self.name + '_main_conv_' + tag))
        self.residual = a['residual']
        if self.residual:
            self.residual_modules = nn.ModuleList()
            self.res_pool = a['stride_1'] * a['stride_2'] * a['pooling_1'] * a['pooling_2']
            if self.res_pool > 1:
                self.residual_modules.append(nn.MaxPool2d(kernel_size=self.res_pool, ceil_mode=True))
            if a['in_channels'] != a['out_channels']:
                self.index['res_conv'] = len(self.residual_modules)
                self.residual_modules.append(nn.Conv2d(in_channels=a['in_channels'], out_channels=a['out_channels'], kernel_size=1,
                                                       padding=a['padding_1'] + a['padding_2'], bias=False))
        self.output_activation_writer = ActivationWriter(register=activation_register, name=self.name + '_main_conv_2')

    def forward(self, x):
        raise NotImplementedError("ScalogramEncoderBlock runs inside AudioPredictiveCodingModel.forward on the HIP path")


class ScalogramResidualEncoder(nn.Module):
    """Stack of ScalogramEncoderBlocks with ReLU between blocks; the output is the single remaining frequency row
    (reference scalogram_model.py:488-529)."""

    def __init__(self, args_dict, preprocessing_module=None, verbose=0):
        super().__init__()
        self.verbose = verbose
        self.phase = args_dict['phase']
        if self.phase:
            args_dict['blocks'][0]['in_channels'] = 2
        if preprocessing_module is None:
            self.receptive_field = 1
            self.downsampling_factor = 1
        else:
            self.receptive_field = preprocessing_module.receptive_field
            self.downsampling_factor = preprocessing_module.downsampling_factor
        self.blocks = nn.ModuleList()
        for i, block_dict in enumerate(args_dict['blocks']):
            self.blocks.append(ScalogramEncoderBlock(block_dict, name='scalogram_block_' + str(i),
                                                     activation_register=args_dict.get('activation_register')))
            self.receptive_field += (block_dict['kernel_size_1'][1] - 1) * self.downsampling_factor
            self.downsampling_factor *= block_dict['pooling_1'] * block_dict['stride_1']
            self.receptive_field += (block_dict['kernel_size_2'][1] - 1) * self.downsampling_factor
            self.downsampling_factor *= block_dict['pooling_2'] * block_dict['stride_2']

    def forward(self, x):
        """x (B, C, bins, frames) on the GPU -> (B, E, frames') float32 (the reference's x[:, :, 0, :]); BatchNorm follows
        self.training.  Differentiable with respect to the encoder's parameters, stand-alone too (audio_model._EncoderForward)."""
        from .audio_model import _EncoderForward, _standalone_owner
        if x.dim() == 3:
            x = x.unsqueeze(2)
        owner = _standalone_owner(self)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            owner.engine_for(x)
            names = [n for n, _ in owner.named_parameters() if n.startswith("encoder.")]
            return _EncoderForward.apply(owner, names, x, *[dict(owner.named_parameters())[n] for n in names])
        return owner.encode(x)
