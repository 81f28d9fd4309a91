"""Constant-Q filter bank front end on the HIP path (reference constant_q_transform.py:94-172, :268-286).

``CQT`` keeps the reference's structure — one strided 1-D filter bank per octave group, real rows then imaginary rows in
``conv_modules.N.weight`` (2*n_g, 1, size_g), not trainable — so state_dicts interchange.  Its forward is one f32
``cpc_gemm_nt`` per group over overlapped waveform rows (row t of clip b = x[b, offset_g + t*hop : ... + size_g]).

Filter design: the reference calls ``librosa.filters.constant_q`` (third party, not vendored, version unpinned — SURVEY.md
8c).  ``constant_q_filters`` below restates that published design (Hann-windowed complex exponentials of length
Q*sr/f_k, L1-normalised, centre-padded to a power of two); coefficient parity with any particular librosa release is
UNPINNED.  Everything downstream of the coefficients is pinned: pass ``filters=(complex [n_bins, len], lengths)`` from
librosa, or load a reference state_dict, and the outputs follow the reference exactly.
"""
import math
import os

import numpy as np
import ctypes as C

import torch
import torch.nn as nn

from . import _hip

pi = np.pi


def cqt_frequencies(n_bins, fmin, bins_per_octave=12):
    """Centre frequencies fmin * 2**(k / bins_per_octave) (librosa.time_frequency.cqt_frequencies, tuning 0)."""
    return float(fmin) * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bins_per_octave)


def constant_q_filters(sr, fmin, n_bins, bins_per_octave, filter_scale):
    """Restated librosa.filters.constant_q(window='hann', pad_fft=True, norm=1): returns (complex128 [n_bins, P], lengths)
    with P the next power of two >= the longest filter."""
    freqs = cqt_frequencies(n_bins, fmin, bins_per_octave)
    q = float(filter_scale) / (2.0 ** (1.0 / bins_per_octave) - 1.0)
    lengths = q * sr / freqs
    sigs = []
    for ilen, freq in zip(lengths, freqs):
        n = np.arange(-ilen // 2, ilen // 2, dtype=np.float64)
        sig = np.exp(n * 1j * 2 * np.pi * freq / sr)
        n_min, n_max = int(np.floor(len(sig))), int(np.ceil(len(sig)))
        k = np.arange(n_min, dtype=np.float64)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n_min)        # periodic Hann
        if len(window) < n_max:
            window = np.pad(window, (0, n_max - len(window)))
        sig = sig * window
        sig = sig / np.sum(np.abs(sig))
        sigs.append(sig)
    max_len = int(2.0 ** np.ceil(np.log2(max(len(s) for s in sigs))))
    bank = np.zeros((n_bins, max_len), dtype=np.complex128)
    for k, sig in enumerate(sigs):
        lpad = (max_len - len(sig)) // 2
        bank[k, lpad:lpad + len(sig)] = sig
    return bank, lengths


class _FilterBank(nn.Module):
    """Holds ``weight`` (2*n, 1, size) like the reference's bias-free nn.Conv1d; ``stride`` = hop length."""

    def __init__(self, weight, stride):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=False)
        self.stride = (stride,)
        self.kernel_size = (weight.shape[-1],)


class CQT(nn.Module):
    def __init__(self, sr=16000, fmin=30, n_bins=256, bins_per_octave=32, filter_scale=1., hop_length=128, trainable=False,
                 filters=None):
        super().__init__()
        self.sr, self.fmin, self.n_bins = sr, fmin, n_bins
        self.bins_per_octave, self.filter_scale, self.hop_length = bins_per_octave, filter_scale, hop_length
        if filters is None:
            filters = constant_q_filters(sr, fmin, n_bins, bins_per_octave, filter_scale)
        cqt_filters, lengths = filters
        self.cqt_filter_lengths = lengths
        # one filter bank per octave: bins whose power-of-two rounded length is the same (reference :113-128)
        self.conv_kernel_sizes, self.conv_index_ranges = [], []
        current, last_change = None, 0
        for i, l in enumerate(lengths):
            size = 2 ** math.ceil(np.log2(l))
            if current is not None and size >= current:
                continue
            self.conv_kernel_sizes.append(size)
            current = size
            if i != 0:
                self.conv_index_ranges.append(range(last_change, i))
            last_change = i
        self.conv_index_ranges.append(range(last_change, len(lengths)))
        total = cqt_filters.shape[-1]
        self.conv_modules = nn.ModuleList()
        for size, rng in zip(self.conv_kernel_sizes, self.conv_index_ranges):
            off = (total - size) // 2
            part = cqt_filters[rng, off:total - off] if off > 0 else cqt_filters[rng, :]
            w = torch.cat([torch.from_numpy(np.real(part).copy()), torch.from_numpy(np.imag(part).copy())], dim=0).float()
            self.conv_modules.append(_FilterBank(w.unsqueeze(1), hop_length))
        # "fp32": exact-f32 MFMA GEMMs;  "bf16x3": waveform and filters split into bf16 (hi, lo) pairs, three bf16 products per
        # tap accumulated in f32 (error ~2^-16 relative per product) at several times the rate
        self.precision = "fp32"
        self._trainable = False
        if trainable:
            raise NotImplementedError("trainable CQT filters are not part of the HIP path (every reference config freezes them)")
        self._operands = None
        self._split_buf = None

    @property
    def trainable(self):
        return self._trainable

    def frames(self, length):
        return (length - 1 - self.conv_kernel_sizes[0]) // self.hop_length + 1

    # ------------------------------------------------------------------ device operands
    def _prepare(self, device):
        key = (str(device), self.precision) + tuple((m.weight.data_ptr(), m.weight._version) for m in self.conv_modules)
        if self._operands is not None and self._operands[0] == key:
            return self._operands[1]
        ops = []
        banks = []                  # per octave group: (interleaved (re, im) rows f32 [2n][size], first bin, size)
        for m, rng in zip(self.conv_modules, self.conv_index_ranges):
            w = m.weight.detach().to(device=device, dtype=torch.float32)[:, 0, :]          # (2n, size): real rows, imag rows
            n = len(rng)
            inter = torch.zeros(2 * n, w.shape[1], device=device, dtype=torch.float32)
            inter[0::2] = w[:n]
            inter[1::2] = w[n:]
            banks.append((inter, rng.start, w.shape[1]))
        if self.precision == "bf16x3":
            # Octave groups are MERGED while their output columns fit one 128-wide GEMM tile: the shorter filters are zero-padded
            # (centred, exactly as the reference centres every group inside the longest filter, :136-140 / :165-166) to the length
            # of the first group of the merge.  A 40- or 64-column launch occupies the 128-wide tile as a full one does, so the merged
            # launch takes the time of its first group alone and the others' launches disappear (configs[2]: 2.2 -> 1.5 ms of CQT).
            merged, i = [], 0
            while i < len(banks):
                rows, start, size = [banks[i][0]], banks[i][1], banks[i][2]
                count = rows[0].shape[0]
                j = i + 1
                while j < len(banks) and count + banks[j][0].shape[0] <= 128 and os.environ.get("CPC_CQT_MERGE", "1") != "0":
                    wj, _, sj = banks[j]
                    pad = torch.zeros(wj.shape[0], size, device=device, dtype=torch.float32)
                    o = (size - sj) // 2
                    pad[:, o:o + sj] = wj
                    rows.append(pad)
                    count += wj.shape[0]
                    j += 1
                merged.append((torch.cat(rows, dim=0), start, size))
                i = j
            for inter, start, size in merged:
                # Only the taps some filter of the bank actually has: a constant-Q filter is its window's length, centred in the
                # power-of-two frame (the longest of the default bank: 12 060 of 16 384 taps) -- the zero margins are dropped from the
                # contraction (bounds rounded outwards to multiples of 128 taps: the K-stage, also of a half when K is split; CPC_CQT_TRIM=0: the full frames).
                lo, hi = 0, size
                if os.environ.get("CPC_CQT_TRIM", "1") != "0":
                    nz = (inter != 0).any(dim=0).nonzero()
                    if nz.numel():
                        lo = int(nz.min()) // 128 * 128
                        hi = min(size, (int(nz.max()) + 128) // 128 * 128)
                inter = inter[:, lo:hi]
                npad = (inter.shape[0] + 7) // 8 * 8
                full = torch.zeros(npad, hi - lo, device=device, dtype=torch.float32)
                full[:inter.shape[0]] = inter
                wh = full.to(torch.bfloat16)
                wl = (full - wh.float()).to(torch.bfloat16)
                ops.append((torch.stack([wh, wh, wl], dim=-1).reshape(npad, 3 * (hi - lo)).contiguous(), npad, start, size, lo, hi - lo))
        else:
            for inter, start, size in banks:
                npad = (inter.shape[0] + 3) // 4 * 4
                full = torch.zeros(npad, size, device=device, dtype=torch.float32)
                full[:inter.shape[0]] = inter
                ops.append((full.contiguous(), npad, start, size, 0, size))
        self._operands = (key, ops)
        return ops

    def transform(self, x):
        """x (B, 1, L) or (B, L) f32 on the GPU -> (cq f32 [B][Tn][ldq] with (re, im) interleaved per bin, Tn, ldq)."""
        if x.dim() == 3:
            x = x[:, 0, :]
        if x.device.type != "cuda":
            raise RuntimeError("the CQT runs on the GPU only (no CPU fallback)")
        x = x.detach().float().contiguous()
        B, L = x.shape
        if L % 8:
            x = torch.nn.functional.pad(x, (0, 8 - L % 8))
        Tn = self.frames(L)
        if Tn < 1:
            raise ValueError(f"clip of {L} samples is shorter than the longest CQT filter ({self.conv_kernel_sizes[0]})")
        ldq = 2 * self.n_bins + 8
        cq = torch.empty(B, Tn, ldq, device=x.device, dtype=torch.float32)
        k0, hop = self.conv_kernel_sizes[0], self.hop_length
        if self.precision == "bf16x3":
            x3 = torch.empty(B, 3 * x.shape[1], device=x.device, dtype=torch.bfloat16)
            _hip.call("cpc_split3_bf16", _hip.ptr(x), _hip.ptr(x3), x.numel())
            for filt, npad, start, size, lo, taps in self._prepare(x.device):
                offset = (k0 - size) // 2 + lo
                M, K = B * Tn, 3 * taps
                tiles = -(-M // 128)
                if K >= 16384 and tiles % 512 and tiles % 512 <= 256:
                    # One 128-row tile per workgroup, two workgroups per CU: 630 tiles would run as one full round plus a
                    # quarter-full one.  The long filters are split in two halves of K (two "batches" into a scratch buffer,
                    # added afterwards): 1 260 half-length workgroups fill 2.5 rounds instead.
                    if self._split_buf is None or self._split_buf.shape != (2, M, npad):
                        self._split_buf = torch.empty(2, M, npad, device=x.device, dtype=torch.float32)
                    _hip.gemm_nt(_hip.ptr(x3, 3 * offset), _hip.ptr(filt), _hip.ptr(self._split_buf), M, npad, K // 2, 3 * hop, K, npad,
                                 _hip.BF16, a_rpi=Tn, a_item=3 * x.shape[1], a_batch=K // 2, b_batch=K // 2, c_batch=M * npad, batch=2,
                                 flags=_hip.GEMM_OUT_F32)
                    # cq[m][2 start + j] = half 0 + half 1 (a strided torch.add here was the last torch arithmetic of the scalogram step)
                    _hip.call("cpc_reduce_slabs", _hip.ptr(self._split_buf), _hip.ptr(cq, 2 * start), M, npad, 2, C.c_longlong(M * npad), 1,
                              C.c_longlong(1), C.c_longlong(ldq), C.c_longlong(0))
                    continue
                _hip.gemm_nt(_hip.ptr(x3, 3 * offset), _hip.ptr(filt), _hip.ptr(cq, 2 * start), M, npad, K, 3 * hop,
                             K, ldq, _hip.BF16, a_rpi=Tn, a_item=3 * x.shape[1], flags=_hip.GEMM_OUT_F32)
            return cq, Tn, ldq
        if self.precision != "fp32":
            raise ValueError("CQT.precision must be 'fp32' or 'bf16x3'")
        for filt, npad, start, size, lo, taps in self._prepare(x.device):
            offset = (k0 - size) // 2 + lo
            _hip.gemm_nt(_hip.ptr(x, offset), _hip.ptr(filt), _hip.ptr(cq, 2 * start), B * Tn, npad, taps, hop, taps, ldq, _hip.F32,
                         a_rpi=Tn, a_item=x.shape[1])
        return cq, Tn, ldq

    def forward(self, x):
        """(B, 1, L) -> (B, n_bins, frames, 2) like the reference (a strided view of the interleaved buffer)."""
        cq, Tn, ldq = self.transform(x)
        return cq[:, :, :2 * self.n_bins].view(cq.shape[0], Tn, self.n_bins, 2).permute(0, 2, 1, 3)


class PhaseDifference(nn.Module):
    """Per-bin expected phase advance and 1/ln(f) scaling (reference :268-280); applied inside cpc_scalogram_pointwise."""

    def __init__(self, sr=16000, fmin=30, n_bins=256, bins_per_octave=32, hop_length=128):
        super().__init__()
        freqs = cqt_frequencies(n_bins, fmin, bins_per_octave)
        fixed = (((1.0 * freqs * hop_length / sr) + 0.5) % 1 - 0.5) * 2 * np.pi
        self.fixed_phase_diff = nn.Parameter(torch.from_numpy(fixed).float().view(1, -1, 1), requires_grad=False)
        self.scaling = nn.Parameter(torch.from_numpy(1 / np.log(freqs)).float().view(1, -1, 1), requires_grad=False)
