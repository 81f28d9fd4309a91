"""Drop-in trainer surface: score functions, ContrastiveEstimationTrainer, DeterministicSampler, grad_mean_var.

Mirrors the reference's ``contrastive_estimation_training.py`` (score functions :12-33, trainer :36-269,
DeterministicSampler :363-382, grad_mean_var :385-391).  ``train`` has two routes:

* fused (the hot path): AudioEncoder + AudioGRUModel model, softplus/linear score (either ``score_over_all_timesteps``
  setting), Adam.  Forward, InfoNCE loss, analytic backward and the Adam update all run as
  HIP kernels (engine.CPCEngine); with torch.distributed initialised, one process per GPU, the flat gradient buffer is
  all-reduced over RCCL before the update (per-GPU in-batch negatives).
* generic: any other score function or optimizer: the model forward and backward run on the HIP path through the autograd
  bridge, the score function is the caller's, and the loss (:106-122, :141) and its gradient come from the same loss kernels
  (``_InfoNCE``).  ``validate`` takes its per-step losses / accuracies from ``cpc_nce_eval`` on both routes.
"""
from __future__ import annotations

import math
import os
import random

import torch
import torch.nn.functional as F
import torch.optim
import torch.utils.data

from .audio_model import *          # noqa: F401,F403  (the reference re-exports the model names from here)
from .audio_dataset import FileBatchSampler


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} runs on the GPU only (libcpc_hip.so; there is no CPU fallback): move the tensors to the device")


class InputAhead:
    """The preprocessing module of batch i + 1 (CQT GEMMs + the pointwise scalogram kernel: 1.3 of the 12 ms of a BASELINE configs[2]
    step) issued on the side stream while step i runs, instead of on the main stream in front of step i + 1's encoder.  The reference
    preprocesses inside the step (:99-103); the scalogram does not depend on the parameters, so the result is the same tensor, one
    step early: ``submit(batch)`` queues it behind everything the main stream has been given so far (so that the batch itself is
    complete) and returns at once, ``take()`` makes the main stream wait for the oldest submitted batch and returns
    (batch, model input).  Buffers: the module allocates a fresh output per call; record_stream keeps the caching allocator from
    handing a block to the other stream while it is still being read."""

    def __init__(self, fn, device):
        from .engine import side_stream
        self.fn, self.device = fn, torch.device(device)
        self.aux = side_stream(self.device)
        if os.environ.get("CPC_PREPROCESS_STREAM", "side") == "own":          # A/B: a stream of its own at the default priority
            self.aux = _own_stream(self.device)
        self.pending = []

    def submit(self, batch):
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.aux):
            self.aux.wait_event(ready)
            x = self.fn(batch)
            done = torch.cuda.Event()
            done.record(self.aux)
        batch.record_stream(self.aux)
        self.pending.append((batch, x, done))

    def take(self):
        batch, x, done = self.pending.pop(0)
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        if isinstance(x, torch.Tensor):
            x.record_stream(main)
        return batch, x


_OWN_STREAMS = {}


def _own_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _OWN_STREAMS:
        _OWN_STREAMS[key] = torch.cuda.Stream(device=device)
    return _OWN_STREAMS[key]


def _with_next(it):
    """(item, next item or None) pairs of an iterator."""
    it = iter(it)
    cur = next(it, None)
    while cur is not None:
        nxt = next(it, None)
        yield cur, nxt
        cur = nxt


class _ScoreContraction(torch.autograd.Function):
    """scores[b, k, b', k'] = sum_e predicted_z[b, k, e] * targets[b', e, k'] as ONE cpc_gemm_nt call (f32, exact-f32 MFMA) over
    the (B K) x E operands; the gradients are the two matching contractions (cpc_gemm_nt / cpc_gemm_tn)."""

    @staticmethod
    def forward(ctx, predicted_z, targets):
        from . import _hip
        _need_gpu(predicted_z, "the score contraction")
        B, K, E = predicted_z.shape
        if tuple(targets.shape) != (B, E, K) or E % 4 or (B * K) % 4:
            raise ValueError("score contraction: expected predicted_z (B, K, E) and targets (B, E, K) with E and B*K multiples of 4")
        R = B * K
        A = predicted_z.detach().reshape(R, E).float().contiguous()
        Tt = targets.detach().permute(0, 2, 1).reshape(R, E).float().contiguous()
        S = torch.empty(R, R, device=A.device, dtype=torch.float32)
        _hip.gemm_nt(_hip.ptr(A), _hip.ptr(Tt), _hip.ptr(S), R, R, E, E, E, R, _hip.F32)
        ctx.save_for_backward(A, Tt)
        ctx.shape = (B, K, E)
        return S.view(B, K, B, K)

    @staticmethod
    def backward(ctx, d_scores):
        from . import _hip
        A, Tt = ctx.saved_tensors
        B, K, E = ctx.shape
        R = B * K
        dS = d_scores.reshape(R, R).float().contiguous()
        dA = torch.empty(R, E, device=dS.device, dtype=torch.float32)
        dT = torch.empty(R, E, device=dS.device, dtype=torch.float32)
        TtT = Tt.t().contiguous()                                                        # [E][R]
        _hip.gemm_nt(_hip.ptr(dS), _hip.ptr(TtT), _hip.ptr(dA), R, E, R, R, R, E, _hip.F32)          # dA = dS Tt
        _hip.gemm_tn(_hip.ptr(dS), _hip.ptr(A), _hip.ptr(dT), R, R, E, R, E, E, _hip.F32, flags=_hip.GEMM_OUT_F32)   # dT = dS^T A
        return dA.view(B, K, E), dT.view(B, K, E).permute(0, 2, 1)


def linear_score_function(predicted_z, targets):
    """scores[b, k, b', k'] = sum_e predicted_z[b,k,e] * targets[b',e,k']  (reference :19-22)."""
    return _ScoreContraction.apply(predicted_z, targets)


def softplus_score_function(predicted_z, targets):
    """softplus of the linear scores (reference :12-16)."""
    return F.softplus(_ScoreContraction.apply(predicted_z, targets))


def difference_score_function(predicted_z, targets):
    """1 / squared distance between every prediction and every target (reference :25-33); O(B^2 K^2 E) memory."""
    diff = predicted_z.unsqueeze(3).unsqueeze(4) - targets.permute(1, 0, 2).unsqueeze(0).unsqueeze(1)
    return 1 / torch.sum(diff ** 2, dim=2)


def _score_layout(scores4, all_timesteps):
    """The 4-D score tensor of ANY score function in the layout the loss kernels read: the (B K) x (B K) matrix, or in the default
    branch its K equal-step blocks S[k][b][b'] = scores[b, k, b', k] with rows padded to a multiple of 8 floats."""
    B, K = scores4.shape[0], scores4.shape[1]
    if all_timesteps:
        return scores4.reshape(B * K, B * K).float().contiguous(), B * K
    ld = -(-B // 8) * 8
    S = torch.zeros(K, B, ld, device=scores4.device, dtype=torch.float32)
    S[:, :, :B] = torch.diagonal(scores4, dim1=1, dim2=3).permute(2, 0, 1)
    return S, ld


class _InfoNCE(torch.autograd.Function):
    """Loss of the train step (reference :108-122, :141) from a 4-D score tensor, through the same kernels as the fused route
    (cpc_nce_loss / cpc_nce_loss_all with the score function already applied): returns (loss incl. regulariser, max score,
    NaN indicator of the loss before the regulariser); the backward hands d loss / d scores back to autograd."""

    @staticmethod
    def forward(ctx, scores4, all_timesteps, regularization):
        import ctypes as C
        from . import _hip
        _need_gpu(scores4, "the InfoNCE loss")
        B, K = scores4.shape[0], scores4.shape[1]
        S, ld = _score_layout(scores4.detach(), all_timesteps)
        dev, f32 = S.device, torch.float32
        out = torch.zeros(8, device=dev, dtype=f32)
        dS, dST = torch.zeros_like(S), torch.zeros_like(S)
        if all_timesteps:
            ws = torch.empty(int(_hip.lib().cpc_nce_all_workspace_floats(B, K)), device=dev, dtype=f32)
            ST = S.t().contiguous()
            _hip.call("cpc_nce_loss_all", _hip.ptr(S), _hip.ptr(ST), _hip.ptr(dS), _hip.ptr(dST), _hip.ptr(out), _hip.ptr(ws), B, K, ld, 0,
                      C.c_float(regularization), _hip.F32)
        else:
            ws = torch.empty(int(_hip.lib().cpc_nce_workspace_floats(B, K)), device=dev, dtype=f32)
            _hip.call("cpc_nce_loss", _hip.ptr(S), _hip.ptr(dS), _hip.ptr(dST), _hip.ptr(out), _hip.ptr(ws), B, K, ld, 0,
                      C.c_float(regularization), _hip.F32)
        ctx.save_for_backward(dS)
        ctx.meta = (B, K, bool(all_timesteps), scores4.dtype)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, d_loss, _d_out):
        (dS,) = ctx.saved_tensors
        B, K, all_t, dtype = ctx.meta
        if all_t:
            d = dS.view(B, K, B, K) * d_loss
        else:
            d = torch.zeros(B, K, B, K, device=dS.device, dtype=torch.float32)
            torch.diagonal(d, dim1=1, dim2=3).copy_(dS[:, :, :B].permute(1, 2, 0) * d_loss)
        return d.to(dtype), None, None


class ContrastiveEstimationTrainer:
    def __init__(self, model, dataset, logger=None, device=None,
                 regularization=1., validation_set=None, test_task_set=None, prediction_noise=0.01,
                 optimizer=torch.optim.Adam,
                 file_batch_size=1,
                 score_over_all_timesteps=False,
                 score_function=softplus_score_function,
                 wasserstein_gradient_penalty=False,
                 gradient_penalty_factor=10.,
                 preprocessing=None,
                 ar_size=256,
                 prediction_steps=16):
        self.model = model
        self.ar_size = ar_size
        self.prediction_steps = prediction_steps
        self.dataset = dataset
        self.logger = logger
        self.device = device
        self.regularization = regularization
        self.validation_set = validation_set
        self.test_task_set = test_task_set
        self.training_step = 0
        self.print_out_scores = False
        self.prediction_noise = prediction_noise
        self.optimizer = optimizer
        self.file_batch_size = file_batch_size
        self.score_over_all_timesteps = score_over_all_timesteps
        self.score_function = score_function
        self.wasserstein_gradient_penalty = wasserstein_gradient_penalty
        self.gradient_penalty_factor = gradient_penalty_factor
        self.preprocessing = preprocessing
        # Not in the reference: how often loss / max-score are read back to the host (the reference reads them every
        # step, :124 and :165-166).  Values are delivered to the logger in order, at most this many steps late.
        self.host_sync_interval = 1
        self.host_sync_lag = 1          # steps the loss readback trails the launches by (0: read every step's loss at once)
        # Not in the reference: replay the whole step from a captured hipGraph (single process, fused path, no preprocessing
        # module, no dropout).  Measured neutral on MI355X (launches are already hidden); off by default.
        self.use_graph = False
        # Not in the reference's signature: under torch.distributed take the InfoNCE loss over the batches of ALL ranks — what
        # the reference's nn.DataParallel wrap computes — instead of per-GPU negatives (engine.GlobalNegatives).
        self.global_negatives = False
        # Not in the reference: the preprocessing module of the NEXT batch runs on the side stream beside the current step (InputAhead)
        self.preprocess_ahead = True
        self.verbose = True
        if wasserstein_gradient_penalty:
            # reference :144-158.  Its penalty differentiates the summed scores with respect to the PREPROCESSED batch, which only
            # requires grad behind a preprocessing module (:100-102): without one the reference itself fails in autograd.grad.
            if preprocessing is None:
                raise ValueError("wasserstein_gradient_penalty needs a preprocessing module (the reference takes the penalty's "
                                 "gradient with respect to the preprocessed batch, contrastive_estimation_training.py:100-102, :147)")
            if score_function not in (linear_score_function, softplus_score_function) or optimizer is not torch.optim.Adam:
                raise NotImplementedError("the gradient penalty on the HIP path covers linear_score_function / softplus_score_function "
                                          "+ Adam; see DESIGN.md section 8")
            model.gradient_penalty_engine = True      # engines built from now on also give the gradient w.r.t. the scalogram
        if self.verbose:
            print("use score function", self.score_function)

    # ------------------------------------------------------------------------------------------ helpers
    def _device(self):
        if self.device is not None:
            return torch.device(self.device)
        return next(self.model.parameters()).device

    def _model_input(self, batch):
        """(B, L) device batch -> what the model is called with: (B, 1, L), or the scalogram when a preprocessing module
        (scalogram_model.PreprocessingModule) is set — reference :99-103, :221-223, :289-291."""
        x = batch.unsqueeze(1)
        if self.preprocessing is not None:
            x = self.preprocessing(x)
        return x

    def _fused(self):
        return self.score_function in (softplus_score_function, linear_score_function) and self.optimizer is torch.optim.Adam

    @staticmethod
    def _world():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def _batches(self, dataset, sampler, device, num_workers, pin_memory, rank, world):
        """Yields device batches (B, L).  A dataset exposing ``device_data`` (an (N, L) tensor already in HBM) is
        indexed on the device; anything else goes through a torch DataLoader as in the reference (:87-91)."""
        resident = getattr(dataset, "device_data", None)
        if world > 1:
            import torch.distributed as dist
            lists = [list(b) for b in iter(sampler)] if rank == 0 else None
            box = [lists]
            dist.broadcast_object_list(box, src=0)
            per = len(box[0][0]) // world if box[0] else 0
            index_lists = [b[rank * per:(rank + 1) * per] for b in box[0]]
        else:
            index_lists = None
        if resident is not None:
            it = index_lists if index_lists is not None else iter(sampler)
            for idx in it:
                yield resident[torch.as_tensor(list(idx), device=resident.device)]
            return
        loader = torch.utils.data.DataLoader(dataset, batch_sampler=index_lists if index_lists is not None else sampler,
                                             num_workers=num_workers, pin_memory=pin_memory)
        if torch.device(device).type != "cuda":
            for batch in iter(loader):
                yield batch.to(device=device, non_blocking=True)
            return
        # Host dataset: double-buffered upload.  Batch i + 1 travels pinned host -> HBM on a copy stream while step i computes
        # (21 MB per step at B = 256: 0.3-0.4 ms of PCIe time that would otherwise sit in front of every step).
        copy = torch.cuda.Stream(device=device)
        it = iter(loader)

        def fetch():
            batch = next(it, None)
            if batch is None:
                return None
            with torch.cuda.stream(copy):
                dev_batch = batch.to(device=device, non_blocking=True)
                done = torch.cuda.Event()
                done.record(copy)
            return dev_batch, done, batch          # the pinned source stays referenced until its copy has been waited for

        nxt = fetch()
        while nxt is not None:
            dev_batch, done, _src = nxt
            nxt = fetch()
            cur = torch.cuda.current_stream(device)
            cur.wait_event(done)
            dev_batch.record_stream(cur)
            yield dev_batch

    # ------------------------------------------------------------------------------------------ train
    def train(self, batch_size=32, epochs=10, lr=0.0001, continue_training_at_step=0, num_workers=1, max_steps=None,
              profile=False):
        """Same contract as the reference's train (:74-176): returns ``prof`` (None unless profile=True) when max_steps is
        reached, None on a NaN loss or when the epochs are exhausted.  ``batch_size`` is the per-process batch; under
        torch.distributed the sampler draws batch_size * world_size indices and every rank takes its slice."""
        device = self._device()
        rank, world = self._world()
        self.model.train()
        fused = self._fused()
        if fused:
            from .engine import FusedAdam, GlobalNegatives, GradAllReduce, GraphedStep
            self.model._flatten_parameters(device)
            graphed = bool(self.use_graph) and world == 1 and self.preprocessing is None
            optimizer = FusedAdam(self.model, lr=lr, device_step=graphed)
            self.last_optimizer = optimizer          # (inspection only: tests read its step count after a NaN return)
            graph_steps = {}
            glob_neg = {}
            self.model.link_grads()
            sync = GradAllReduce(self.model, optimizer=optimizer) if world > 1 else None
        else:
            self.model._flatten_parameters(device)
            optimizer = self.optimizer(self.model.parameters(), lr=lr)
        sampler = FileBatchSampler(index_count_per_file=self.dataset.get_example_count_per_file(),
                                   batch_size=batch_size * world, file_batch_size=self.file_batch_size, drop_last=True,
                                   verbose=self.verbose)
        self.training_step = continue_training_at_step
        pending = []          # (step, device scalars) not yet read back
        guarded = set()       # engines whose sticky NaN flag was cleared for this run

        on_gpu = torch.device(device).type == "cuda"
        ring, ring_pos = [], [0]

        def stash(step, vals):
            """Queues a step's (loss, max score) for the logger.  On the GPU they travel to a pinned host buffer right behind the
            step's own kernels and an event marks their arrival: reading them later does not wait for LATER steps' work, which a
            synchronous read of a device tensor — queued behind everything launched since — would."""
            if not on_gpu:
                pending.append((step, vals.detach()[:6].clone(), None))
                return
            need = self.host_sync_interval + self.host_sync_lag + 2
            while len(ring) < need:
                ring.append(torch.empty(6, dtype=torch.float32, pin_memory=True))
            buf = ring[ring_pos[0] % len(ring)]
            ring_pos[0] += 1
            buf.copy_(vals.detach()[:6].float(), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pending.append((step, buf, ev))

        def flush(keep=0):
            """Reads back all pending steps but the ``keep`` most recent ones (in order); returns the step of a NaN loss."""
            n = max(len(pending) - keep, 0)
            for step, vals, ev in pending[:n]:
                if ev is not None:
                    ev.synchronize()
                row = vals.tolist()
                loss_v, score_v, nan_v = float(row[0]), float(row[1]), float(row[5])        # cpc_nce_loss: out[0], out[1], out[5]
                # reference order (:124-133 before :164-169): a NaN loss ends the run before anything is logged for that step
                if nan_v != 0.0 or math.isnan(loss_v):
                    pending.clear()
                    return step
                if self.logger is not None:
                    self.logger.loss_meter.update(loss_v)
                    self.logger.score_meter.update(score_v)
                    self.logger.log(step)
                elif self.verbose:
                    print("loss at step step " + str(step) + ":", loss_v)
            del pending[:n]
            return None

        # The host learns of a NaN loss host_sync_lag steps late; the steps launched meanwhile change nothing on the device that Adam
        # owns (their updates are skipped there), but their forward passes move the BatchNorm running statistics, and the host-side
        # step count advances.  Both are put back to where the reference leaves them (it returns inside the NaN step, after that step's
        # forward pass, :124-133): the buffers are snapshotted at the START of every step (one multi-tensor copy) and the snapshot of
        # step `nan + 1` is restored.  What cannot be taken back: the sampler has handed out the later batches.
        bn_bufs = [b for n_, b in self.model.named_buffers() if "running_" in n_ or n_.endswith("num_batches_tracked")]
        # (sorted by dtype, copied one dtype at a time: a mixed list — float32 statistics and int64 counters — takes _foreach_copy_'s
        # per-tensor route, 40 device-to-device copies and 0.13 ms per configs[2] step; a uniform list is one multi-tensor kernel)
        bn_bufs.sort(key=lambda b: str(b.dtype))
        bn_groups = []
        for i_, b_ in enumerate(bn_bufs):
            if not bn_groups or bn_bufs[bn_groups[-1][0]].dtype != b_.dtype:
                bn_groups.append([i_, i_ + 1])
            else:
                bn_groups[-1][1] = i_ + 1

        def copy_groups(dst, src):
            for lo_, hi_ in bn_groups:
                torch._foreach_copy_(dst[lo_:hi_], src[lo_:hi_])
        snap_ring, snap_pos, snaps = [], [0], {}
        if fused and bn_bufs:          # the ring exists before the loop: no allocation inside the hot loop; models without BatchNorm keep none
            snap_ring = [[torch.empty_like(b) for b in bn_bufs] for _ in range(self.host_sync_interval + self.host_sync_lag + 2)]

        def snapshot(step):
            if not fused:
                return
            depth = self.host_sync_interval + self.host_sync_lag + 2
            if bn_bufs:
                while len(snap_ring) < depth:          # (host_sync_interval / host_sync_lag raised while training)
                    snap_ring.append([torch.empty_like(b) for b in bn_bufs])
                dst = snap_ring[snap_pos[0] % depth]
                snap_pos[0] += 1
                copy_groups(dst, bn_bufs)
            else:
                dst = None
            snaps[step] = (dst, getattr(optimizer, "t", None))
            for old_step in [k for k in snaps if k < step - depth + 1]:
                del snaps[old_step]

        def nan_return(step):
            later = snaps.get(step + 1)
            if later is not None and later[0] is not None:          # statistics as they were after the NaN step's own forward pass
                copy_groups(bn_bufs, later[0])
            here = snaps.get(step)
            if here is not None and here[1] is not None and hasattr(optimizer, "t"):
                optimizer.t = here[1]                               # no update has happened since the start of the NaN step
            self.training_step = step          # the reference leaves train() inside step `step`, before its update (:124-133)
            print("nan loss")
            print("returned with nan loss at step", step)
            return None

        # A full collection of Python's cyclic garbage collector walks every tracked object of the process (38 - 43 ms with the module
        # trees of a model alive) and lands in whichever step crosses its allocation threshold — a GPU that is only a few steps of work
        # ahead of the host idles through it.  What is alive now stays alive for the whole run: collect once and freeze it, later
        # collections only look at what the steps allocate (bench.py, round 4: this is the 41 ms the round-3 driver run lost once).
        import gc
        gc.collect()
        gc.freeze()
        prof = None
        for current_epoch in range(epochs):
            if self.verbose:
                print("epoch", current_epoch)
            ctx = torch.autograd.profiler.profile(use_device="cuda", enabled=profile)
            with ctx as prof_ctx:
                ahead = None
                if (fused and not graphed and self.preprocessing is not None and self.preprocess_ahead and device.type == "cuda"
                        and os.environ.get("CPC_PREPROCESS_AHEAD", "1") != "0"):
                    ahead = InputAhead(self._model_input, device)
                batches = self._batches(self.dataset, sampler, device, num_workers, True, rank, world)
                # (the sampler is read one batch ahead only where that batch is preprocessed ahead)
                for batch, next_batch in (_with_next(batches) if ahead is not None else ((b_, None) for b_ in batches)):
                    snapshot(self.training_step)
                    if fused and graphed:
                        eng = self.model.engine(batch.shape[0], batch.shape[1], device)
                        key = (batch.shape[0], batch.shape[1])
                        if key not in graph_steps:
                            graph_steps[key] = GraphedStep(eng, optimizer, self.score_function is softplus_score_function,
                                                           float(self.regularization), bool(self.score_over_all_timesteps))
                        if id(eng) not in guarded:
                            eng.nan_flag().zero_()
                            guarded.add(id(eng))
                        vals = graph_steps[key](batch)
                    elif fused:
                        if self.preprocessing is not None:
                            if ahead is not None:
                                if not ahead.pending:              # first step of the epoch: nothing was submitted beside a previous one
                                    ahead.submit(batch)
                                _, x_eng = ahead.take()
                            else:
                                x_eng = self._model_input(batch)
                            eng = self.model.engine_for(x_eng)
                            if ahead is not None and next_batch is not None:
                                # the engine runs it behind its encoder's forward pass, where its main queue turns latency-bound
                                eng.side_job = lambda nb=next_batch: ahead.submit(nb)
                        else:
                            x_eng = batch.contiguous()
                            eng = self.model.engine(batch.shape[0], batch.shape[1], device)
                        gneg = None
                        if self.global_negatives and world > 1:
                            gneg = glob_neg.get(id(eng))
                            if gneg is None:
                                gneg = glob_neg[id(eng)] = GlobalNegatives(eng)
                        # the operand copies of the next step are rebuilt as soon as Adam has updated their parameters
                        # (engine.CPCEngine.prepare_ahead; under data parallelism Adam follows each reduced gradient piece)
                        optimizer.after_update = eng.prepare_ahead
                        # NaN guard on the device (reference :124-133 returns before backward() / optimizer.step()): the loss
                        # kernel raises the engine's sticky flag, every Adam launch of this and of later steps is a no-op while it
                        # is up, and the host leaves train() when the step's indicator arrives (host_sync_lag steps later)
                        if id(eng) not in guarded:
                            eng.nan_flag().zero_()
                            guarded.add(id(eng))
                        optimizer.skip_flag = eng.nan_flag()
                        if sync is not None:      # per-GPU negatives: mean of the shard gradients; global negatives: they add up
                            sync.grad_scale = 1.0 if gneg is not None else 1.0 / world
                        if self.wasserstein_gradient_penalty:
                            # three passes through the network (scalogram_engine.ScalogramCPCEngine._gp_step); the parameter
                            # gradients are complete only at the end, so Adam runs once, after them
                            out = eng.loss_and_grads(x_eng, softplus=self.score_function is softplus_score_function,
                                                     regularization=float(self.regularization),
                                                     all_timesteps=bool(self.score_over_all_timesteps), global_negatives=gneg,
                                                     after_loss=sync.reduce_flag if sync is not None else None,
                                                     gradient_penalty=float(self.gradient_penalty_factor))
                        else:
                            out = eng.loss_and_grads(x_eng, softplus=self.score_function is softplus_score_function,
                                                     regularization=float(self.regularization),
                                                     all_timesteps=bool(self.score_over_all_timesteps),
                                                     grad_ready_hook=sync.hook if sync is not None else getattr(optimizer, "hook", None),
                                                     global_negatives=gneg, after_loss=sync.reduce_flag if sync is not None else None)
                        if sync is not None:
                            sync.finish()
                        # per-GPU negatives: mean of the shard gradients; global negatives: the shard gradients add up
                        optimizer.step(grad_scale=1.0 if gneg is not None else 1.0 / world)
                        vals = out
                    else:
                        vals = self._generic_step(batch, batch.shape[0], optimizer, world)
                    stash(self.training_step, vals)
                    if not fused:            # this route has already read the loss (NaN check in front of backward(), as the reference)
                        nan_step = flush()
                        if nan_step is not None:
                            return nan_return(nan_step)
                    # the readback trails the launches by host_sync_lag steps: the host waits for step i - 1's loss while step i
                    # already runs (reading step i's loss right away left the GPU idle for 0.3 ms of every 5.1 ms step while the
                    # host prepared the next one); every step is still logged, in order, and a NaN loss still ends the run
                    if len(pending) >= self.host_sync_interval + self.host_sync_lag:
                        nan_step = flush(keep=self.host_sync_lag)
                        if nan_step is not None:
                            return nan_return(nan_step)
                    self.training_step += 1
                    if max_steps is not None and self.training_step >= max_steps:
                        nan_step = flush()
                        if nan_step is not None:
                            return nan_return(nan_step)
                        return prof_ctx if profile else None
            prof = prof_ctx if profile else None
        nan_step = flush()
        if nan_step is not None:
            return nan_return(nan_step)
        return None

    def _generic_step(self, batch, batch_size, optimizer, world):
        """Any score function / optimizer: model forward and backward through the autograd bridge (HIP), the score function as the
        caller wrote it, the loss and its gradient through the loss kernels (_InfoNCE).  This route reads the loss every step, so
        the NaN guard sits where the reference has it: in front of backward() and optimizer.step() (:124-133)."""
        predicted_z, targets, _, _ = self.model(self._model_input(batch))
        scores = self.score_function(predicted_z, targets)
        loss, out = _InfoNCE.apply(scores, bool(self.score_over_all_timesteps), float(self.regularization))
        nan = out[5:6].clone()
        if world > 1:          # every rank has its own loss: all ranks leave at the same step
            import torch.distributed as dist
            dist.all_reduce(nan, op=dist.ReduceOp.MAX)
        vals = torch.cat([out[:5], nan])
        if float(nan.item()) != 0.0:
            return vals
        self.model.zero_grad()
        loss.backward()
        if world > 1:
            for p in self.model.parameters():
                dist.all_reduce(p.grad)
                p.grad.div_(world)
        optimizer.step()
        return vals

    # ------------------------------------------------------------------------------------------ validate
    def validate(self, batch_size=64, num_workers=1, max_steps=None):
        """Reference validate (:178-269): per-step loss, per-step arg-max accuracy, mean score and the mutual-information lower
        bound log(n) - loss over the validation set (eval mode, FileBatchSampler(seed=0, file_batch_size=8)).  The per-batch
        quantities come from cpc_nce_eval on the train step's own score matrices and are summed on the device; the host reads
        2 K + 1 numbers once, after the last batch."""
        import ctypes as C
        from . import _hip
        if self.validation_set is None:
            print("No validation set")
            return 0, 0
        device = self._device()
        K, all_t = self.prediction_steps, bool(self.score_over_all_timesteps)
        counts = self.validation_set.get_example_count_per_file()
        n_batches = sum(1 for _ in FileBatchSampler(counts, batch_size, 8, True, seed=0, verbose=False))
        steps = n_batches if max_steps is None else min(max_steps, n_batches)
        sampler = FileBatchSampler(index_count_per_file=counts, batch_size=batch_size, file_batch_size=8, drop_last=True, seed=0,
                                   verbose=self.verbose)
        sums = torch.zeros(2 * K + 1, device=device, dtype=torch.float32)
        ws = torch.empty(int(_hip.lib().cpc_nce_eval_workspace_floats(batch_size, K)), device=device, dtype=torch.float32)
        kernel_scores = self.score_function in (softplus_score_function, linear_score_function)
        self.model.eval()
        done = 0
        with torch.no_grad():
            for batch in self._batches(self.validation_set, sampler, device, num_workers, False, 0, 1):
                if done >= steps:
                    break
                x = self._model_input(batch)
                if kernel_scores:
                    eng = self.model.engine_for(x)
                    eng.forward(x.float() if x.dim() == 4 else x[:, 0, :].contiguous().float())
                    eng.nce_eval(self.score_function is softplus_score_function, all_t, sums, ws)
                else:
                    predicted_z, targets, _, _ = self.model(x)
                    S, ld = _score_layout(self.score_function(predicted_z, targets), all_t)
                    _hip.call("cpc_nce_eval", _hip.ptr(S), _hip.ptr(sums), _hip.ptr(ws), batch_size, K, ld, 0, 1 if all_t else 0, 1)
                done += 1
        self.model.train()
        sums = sums / max(steps, 1)
        n = batch_size * K if all_t else batch_size
        step_losses, step_accuracy = sums[:K].clone(), sums[K:2 * K].clone()
        return step_losses, step_accuracy, float(sums[2 * K]), math.log(n) - step_losses

    def calc_test_task_data(self, batch_size=64, num_workers=1):
        """Context vectors c of every item of the test-task set (reference :271-303)."""
        if self.test_task_set is None:
            print("No test task set")
        device = self._device()
        num_items = len(self.test_task_set)
        self.model.eval()
        task_data = torch.zeros(num_items, self.ar_size)
        task_labels = torch.zeros(num_items, dtype=torch.long)
        loader = torch.utils.data.DataLoader(self.test_task_set, batch_size=batch_size, num_workers=num_workers)
        with torch.no_grad():
            for step, (batch, labels) in enumerate(iter(loader)):
                _, _, _, c = self.model(self._model_input(batch.to(device)))
                task_data[step * batch_size:step * batch_size + c.shape[0], :] = c.cpu()
                task_labels[step * batch_size:step * batch_size + c.shape[0]] = labels
        self.model.train()
        return task_data.numpy(), task_labels.numpy()

    def test_task(self, task_data, task_labels, evaluation_ratio=0.2):
        """Downstream probe of the context vectors (reference :305-350): a 128-64 ReLU MLP classifier trained with Adam
        (lr 1e-3, batch 64, 10 epochs) on a seeded 80/20 split; returns the evaluation accuracy after the last epoch.
        This is evaluation tooling outside the train-step hot path: a few thousand 256-vectors through
        a three-layer MLP, run with stock torch modules on ``self.device``."""
        num_items = task_data.shape[0]
        order = list(range(num_items))
        random.seed(0)
        random.shuffle(order)
        n_eval = int(num_items * evaluation_ratio)
        eval_idx, train_idx = order[:n_eval], order[n_eval:]
        files = getattr(self.test_task_set, "files", None)
        n_classes = len(files) if files is not None else int(task_labels.max()) + 1
        device = self._device()
        probe = torch.nn.Sequential(torch.nn.Linear(self.ar_size, 128), torch.nn.ReLU(), torch.nn.Linear(128, 64), torch.nn.ReLU(),
                                    torch.nn.Linear(64, n_classes)).to(device)
        to_dev = lambda a, idx: torch.from_numpy(a[idx]).to(device)
        x_train, y_train = to_dev(task_data, train_idx), to_dev(task_labels, train_idx)
        x_eval, y_eval = to_dev(task_data, eval_idx), to_dev(task_labels, eval_idx)
        opt = torch.optim.Adam(probe.parameters(), lr=1e-3)
        accuracy = 0.0
        for epoch in range(10):
            for lo in range(0, y_train.shape[0], 64):
                loss = torch.nn.functional.cross_entropy(probe(x_train[lo:lo + 64]), y_train[lo:lo + 64])
                probe.zero_grad()
                loss.backward()
                opt.step()
            with torch.no_grad():
                hits = torch.eq(torch.argmax(probe(x_eval), dim=1), y_eval)
            accuracy = torch.sum(hits).item() / max(len(eval_idx), 1)
            if self.verbose:
                print("task accuracy after epoch", epoch, ":", accuracy)
        return accuracy


class DeterministicSampler(torch.utils.data.Sampler):
    """Shuffles range(len(data_source)) with a fixed seed: same order on every pass (reference :363-382)."""

    def __init__(self, data_source, seed=0):
        self.data_source = data_source
        self.seed = seed

    def __iter__(self):
        order = list(range(len(self.data_source)))
        random.seed(self.seed)
        random.shuffle(order)
        return iter(order)

    def __len__(self):
        return len(self.data_source)


def grad_mean_var(module):
    """{parameter name: [mean(grad), var(grad)]} (reference :385-391)."""
    out = {}
    for name, p in module.named_parameters():
        if p.grad is not None:
            out[name] = [torch.mean(p.grad).item(), torch.var(p.grad).item()]
    return out
