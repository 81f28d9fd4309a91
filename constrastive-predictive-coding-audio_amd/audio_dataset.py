"""Batch composition for the trainer: FileBatchSampler (index semantics of the reference's audio_dataset.py:202-263)
and a synthetic stand-in for AudioDataset that implements the protocol the trainer uses
(``dataset[i] -> (item_length,) float tensor``, ``get_example_count_per_file()``; reference :155-168).

File I/O (torchaudio / mutagen decoding, reference :16-153) is outside the hot path and not rebuilt.
The sampler draws from Python's ``random`` in the same call order as the reference, so a given seed (or a given
global RNG state when seed is None) yields bit-identical index lists.
"""
from __future__ import annotations

import math
import random
from typing import List, Optional, Sequence

import torch
import torch.utils.data


def _split(seq, n, drop_last):
    out = []
    for start in range(0, len(seq), n):
        if drop_last and start + n > len(seq):
            break
        out.append(seq[start:start + n])
    return out


class FileBatchSampler(torch.utils.data.Sampler):
    """Batches of example indices; with file_batch_size > 1 every batch is made of runs of that many examples taken
    from one file each (examples are numbered file after file)."""

    def __init__(self, index_count_per_file, batch_size, file_batch_size=1, drop_last=True, seed=None, verbose=True):
        self.index_count_per_file = list(index_count_per_file)
        self.indices_in_file: List[List[int]] = []
        first = 0
        for count in self.index_count_per_file:
            self.indices_in_file.append(list(range(first, first + count)))
            first += count
        self.batch_size = batch_size
        self.file_batch_size = file_batch_size
        self.drop_last = drop_last
        self.seed = seed
        rounding = math.floor if drop_last else math.ceil
        self.batches_per_file = [rounding(n / file_batch_size) for n in self.index_count_per_file]
        if verbose:
            print("minimum batches per file:", min(self.batches_per_file),
                  "maximum batches per file:", max(self.batches_per_file))

    def __len__(self):
        # As in the reference: the number of FILE batches (also the index range shuffled when file_batch_size == 1).
        return int(sum(self.batches_per_file))

    def __iter__(self):
        if self.file_batch_size == 1:
            order = list(range(len(self)))
            if self.seed is not None:
                random.seed(self.seed)
            random.shuffle(order)
            return iter(_split(order, self.batch_size, self.drop_last))
        # NB: like the reference, the per-file lists are shuffled in place and keep their order between passes
        for i, indices in enumerate(self.indices_in_file):
            if self.seed is not None:
                random.seed(self.seed + i)
            random.shuffle(indices)
        runs = []
        for indices in self.indices_in_file:
            runs.extend(_split(indices, self.file_batch_size, self.drop_last))
        if self.seed is not None:
            random.seed(self.seed)
        random.shuffle(runs)
        runs_per_batch = self.batch_size // self.file_batch_size
        if runs_per_batch > 1:
            merged = []
            for start in range(0, len(runs), runs_per_batch):
                if self.drop_last and start + runs_per_batch > len(runs):
                    break
                merged.append([i for run in runs[start:start + runs_per_batch] for i in run])
            return iter(merged)
        return iter(runs)


class SyntheticAudioDataset(torch.utils.data.Dataset):
    """``n_items`` clips of ``item_length`` samples of seeded white noise.

    ``device`` != None keeps the whole set resident in HBM (``device_data``) so that the trainer indexes it on the
    device instead of going through a DataLoader (the benchmark contract: inputs already in HBM)."""

    def __init__(self, n_items: int, item_length: int, seed: int = 0, scale: float = 1.0,
                 counts: Optional[Sequence[int]] = None, device=None):
        g = torch.Generator().manual_seed(seed)
        self.data = torch.randn(n_items, item_length, generator=g) * scale
        self.counts = list(counts) if counts is not None else [n_items]
        assert sum(self.counts) == n_items
        self._item_length = item_length
        self.device_data = self.data.to(device) if device is not None else None

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, idx):
        return self.data[idx]

    def get_example_count_per_file(self):
        return list(self.counts)


class TensorAudioDataset(SyntheticAudioDataset):
    """Wraps an existing (n_items, item_length) tensor (e.g. golden fixture data) in the same protocol."""

    def __init__(self, data: torch.Tensor, counts: Optional[Sequence[int]] = None, device=None):
        self.data = data
        self.counts = list(counts) if counts is not None else [data.shape[0]]
        self._item_length = data.shape[1]
        self.device_data = data.to(device) if device is not None else None
