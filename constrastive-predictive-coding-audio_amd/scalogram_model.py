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
        if self.pooling is not None:
            raise NotImplementedError("scalogram_pooling is not part of the HIP path yet")
        cq, Tn, ldq = self.cqt.transform(x)
        B, bins = cq.shape[0], self.cqt.n_bins
        phase = self.phase_diff is not None
        W, Cc = (Tn - 1, 2) if phase else (Tn, 1)
        if W < 1:
            raise ValueError("clip too short for a phase-difference scalogram (needs at least two CQT frames)")
        out = torch.empty(B, W, bins, Cc, device=cq.device, dtype=torch.float32)
        fixed = self.phase_diff.fixed_phase_diff.detach().to(cq.device).reshape(-1).contiguous() if phase else None
        scale = self.phase_diff.scaling.detach().to(cq.device).reshape(-1).contiguous() if phase else None
        _hip.call("cpc_scalogram_pointwise", _hip.ptr(cq), _hip.ptr(fixed), _hip.ptr(scale), _hip.ptr(out), B, Tn, bins, ldq,
                  1 if phase else 0, float(self.offset), float(self.log_offset), float(self.normalization_factor),
                  float(self.output_power))
        x = out.permute(0, 3, 2, 1)              # (B, channels, bins, frames) view
        self.output = x
        return x
