"""CPU-side tests: host logic of the product package, the C-ABI surface (symbols only — no compute without a GPU),
and the multi-process (gloo, world size 2) data-parallel plumbing."""
import json
import numpy as np
import os
import random
import re
import subprocess
import sys

import pytest
import torch

import cpc_audio_amd
from cpc_audio_amd import _hip
from cpc_audio_amd.audio_dataset import FileBatchSampler, SyntheticAudioDataset
from cpc_audio_amd.audio_model import (AudioEncoder, AudioGRUModel, AudioPredictiveCodingModel, encoder_default_dict)
from cpc_audio_amd.contrastive_estimation_training import ContrastiveEstimationTrainer, DeterministicSampler
from cpc_audio_amd.engine import EncoderGeometry
from oracle import cpc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_file_batch_sampler_bit_exact(golden_dir):
    s = json.load(open(os.path.join(golden_dir, "samplers.json")))
    for case in s["file_batch_sampler"]:
        if case["seed"] is None:
            random.seed(case["global_seed"])
        sampler = FileBatchSampler(case["counts"], case["batch_size"], case["file_batch_size"], case["drop_last"],
                                   case["seed"], verbose=False)
        assert len(sampler) == case["len"]
        assert [list(b) for b in iter(sampler)] == case["batches"], case
    d = s["deterministic_sampler"]
    assert list(iter(DeterministicSampler(list(range(d["n"])), seed=d["seed"]))) == d["order"]


def test_encoder_attributes_match_reference_test():
    enc = AudioEncoder({'strides': [5, 4, 2, 2, 2], 'kernel_sizes': [10, 8, 4, 4, 4], 'channel_count': [32] * 5, 'bias': False})
    assert enc.downsampling_factor == 160 and enc.receptive_field == 465     # reference tests/test_audioEncoder.py:20,28
    assert all(l.bias is None for l in enc.layers)


def test_geometry():
    g = EncoderGeometry(20480, [5, 4, 2, 2, 2], [10, 8, 4, 4, 4])
    assert g.valid == [4095, 1022, 510, 254, 126]
    assert g.alloc == [4096, 1024, 512, 256, 128]
    for length in (465 + 16 * 160, 4800, 12465, 18385, 3222):
        g = EncoderGeometry(length, [5, 4, 2, 2, 2], [10, 8, 4, 4, 4])
        assert g.valid == O.encoder_layer_lengths(length, O.DEFAULT_STRIDES, O.DEFAULT_KERNELS)
        for l in range(5):
            assert g.alloc[l] >= g.valid[l] + g.taps[l] - 1
            if l:
                assert g.alloc[l - 1] == g.strides[l] * g.alloc[l]
    with pytest.raises(ValueError):
        EncoderGeometry(300, [5, 4, 2, 2, 2], [10, 8, 4, 4, 4])


def test_model_surface_and_state_dict_keys():
    torch.manual_seed(0)
    enc = AudioEncoder(encoder_default_dict)
    ar = AudioGRUModel(input_size=512, hidden_size=256)
    model = AudioPredictiveCodingModel(enc, ar, enc_size=512, ar_size=256, visible_steps=100, prediction_steps=12)
    assert model.parameter_count() == 7414784            # measured on the reference
    assert model.item_length == 18385
    keys = list(model.state_dict().keys())
    expect = [f"encoder.layers.{l}.{n}" for l in range(5) for n in ("weight", "bias")]
    expect += ["autoregressive_model.gruCell." + n for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    expect += ["prediction_model.weight"]
    assert keys == expect
    # same seed + same construction order => same initial values as the oracle's init (i.e. as the reference modules)
    ref = O.init_params(seed=0)
    for k, v in model.state_dict().items():
        assert torch.equal(v, ref[k]), k
    # no CPU fallback: a CPU batch must fail loudly
    with pytest.raises(RuntimeError):
        model(torch.zeros(2, 1, 18385))


def test_c_abi_exports_every_declared_symbol():
    """libcpc_hip.so loads on a CPU-only host and exports exactly what include/cpc_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "cpc_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|long long)\s+(cpc_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    handle = _hip.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
    assert declared == set(_hip.EXPORTED_SYMBOLS)
    assert handle.cpc_abi_version() == 8
    nm = subprocess.run(["nm", "-D", _hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (cpc_\w+)", nm))
    assert exported == declared


def test_over_read_contract_is_checked_before_any_launch():
    """include/cpc_hip.h, over-read / guard contract: calls whose overlapped-row views would read beyond what the caller says
    is readable return CPC_EINVAL (-22) from the argument check — no kernel is launched, so this runs without a GPU."""
    import ctypes as C
    lib = _hip.lib()
    P = C.c_void_p(0x1000)        # never dereferenced: the calls below are refused before any launch
    B, Cin, Cout, kw, stride, La = 2, 64, 64, 8, 4, 10
    need_tail, need_head = (kw - stride) * Cin, (-(-kw // stride) - 1) * Cout
    s = C.c_void_p(0)
    assert lib.cpc_conv_fwd(P, P, None, P, B, Cin, Cout, kw, stride, La, La - 1, 1, C.c_longlong(need_tail - 1), _hip.BF16, s) == -22
    assert lib.cpc_conv_wgrad(P, P, P, B, Cin, Cout, kw, stride, La, 1, C.c_longlong(0), _hip.BF16, s) == -22
    assert lib.cpc_conv_dgrad(P, P, None, P, B, Cin, Cout, kw, stride, La, La * stride, C.c_longlong(need_head - 1), _hip.BF16, None, None, s) == -22
    assert lib.cpc_conv_dgrad_conv1(P, P, P, P, P, B, 256, Cout, kw, stride, La, 100, 10, 5, 10, C.c_longlong(need_head - 1), _hip.BF16, None, s) == -22
    # cpc_gemm_nt with stated extents: M rows of K = 512 at lda = 256 need (M - 1) * 256 + 512 readable elements
    M, N, K, lda = 40, 64, 512, 256
    args = _hip.GemmNTArgs(P, P, P, None, None, M, N, K, lda, K, N, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, _hip.BF16, M * lda, 0)
    assert lib.cpc_gemm_nt(C.byref(args), s) == -22            # an [M][lda] array is (K - lda) elements short
    args = _hip.GemmNTArgs(P, P, P, None, None, M, N, K, lda, K, N, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, _hip.BF16, 0, N * K - 1)
    assert lib.cpc_gemm_nt(C.byref(args), s) == -22
    # CPC_GEMM_KRANGE_EXACT: K ranges that cut out NON-zero pieces need every tile inside one range index; 96 rows per index against
    # 128-row tiles is refused (without the flag the same launch is admitted: a tile then runs the union of its rows' ranges)
    args = _hip.GemmNTArgs(P, P, P, None, None, 96 * 8, 128, 256, 256, 256, 128, 96, 96 * 256, 0, 0, 0, 0, 0, 0, 0, 0, 1, _hip.GEMM_KRANGE_EXACT,
                           _hip.BF16, 0, 0, 0, 0, 0, 0, P, 0, 0, 0)
    assert lib.cpc_gemm_nt(C.byref(args), s) == -22


def test_engine_guard_rows_cover_every_over_read():
    """EncoderGeometry / CPCEngine._buf: the guard in front of and behind every activation buffer is at least what the
    overlapped-row GEMMs of its consumers read there, also for kernels much wider than their stride."""
    from cpc_audio_amd.engine import EncoderGeometry
    for strides, kernels in (([5, 4, 2, 2, 2], [10, 8, 4, 4, 4]), ([5, 2, 1], [10, 40, 33])):
        geo = EncoderGeometry(4000, strides, kernels)
        need = max([16] + [kernels[l] - strides[l] for l in range(1, len(strides))] + [t - 1 for t in geo.taps[1:]])
        assert need >= max(k - s for k, s in zip(kernels[1:], strides[1:]))
        assert need >= max(-(-k // s) - 1 for k, s in zip(kernels[1:], strides[1:]))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_hip.HipLibraryMissing):
        _hip.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "constrastive-predictive-coding-audio_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py") and fn != "engine.py":
            assert "oracle" not in open(os.path.join(pkg, fn)).read(), fn
    # engine.py may reference the oracle only inside smoke_check (the __graft_entry__.smoke() checker)
    src = open(os.path.join(pkg, "engine.py")).read()
    head, _, tail = src.partition("def smoke_check")
    assert "oracle" not in head


WORKER = r'''
import os, sys, random, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cpc_audio_amd.audio_dataset import FileBatchSampler, SyntheticAudioDataset
from cpc_audio_amd.contrastive_estimation_training import ContrastiveEstimationTrainer
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
ds = SyntheticAudioDataset(32, 50, seed=3)
random.seed(1000 + rank)            # deliberately different RNG state per rank: rank 0's lists must win
sampler = FileBatchSampler(ds.get_example_count_per_file(), batch_size=4 * world, verbose=False)
tr = ContrastiveEstimationTrainer.__new__(ContrastiveEstimationTrainer)
batches = [b.clone() for b in tr._batches(ds, sampler, torch.device("cpu"), 0, False, rank, world)]
mine = torch.stack(batches)                                  # (n_batches, 4, 50)
gathered = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(gathered, mine)
# gradient averaging as the trainer does it: all-reduce(sum) of the flat buffer, then scale by 1/world
flat = torch.full((10,), float(rank + 1))
dist.all_reduce(flat)
flat *= 1.0 / world
if rank == 0:
    glob = torch.cat(gathered, dim=1)                        # (n_batches, 8, 50): the global batches
    rows = glob.reshape(-1, 50)
    # every global batch consists of distinct dataset rows, and the two shards are disjoint
    for b in range(glob.shape[0]):
        idx = [int((ds.data == glob[b, i]).all(1).nonzero()[0]) for i in range(glob.shape[1])]
        assert len(set(idx)) == 8, idx
    assert glob.shape[0] == 32 // 8
    assert torch.allclose(flat, torch.full((10,), 1.5))
    print("DP-OK")
dist.destroy_process_group()
'''


def test_data_parallel_plumbing_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DP-OK" in outs[0]


WORKER_PIECES = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cpc_audio_amd.engine import GradAllReduce
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()

class Model:
    pass

import time
total = float(sum(range(1, world + 1)))           # rank r holds r + 1 everywhere
class Opt:
    """Stands in for FusedAdam: records the pieces and checks that a piece is REDUCED when its update is issued, and that the
    NaN flag has been reduced over the ranks before the first update of the step reads it."""
    def __init__(self, model, flag=None, want_flag=None):
        self.model, self.calls, self.flag, self.want_flag = model, [], flag, want_flag
    def update_range(self, lo, hi, scale):
        g = self.model._flat_grad[lo:hi]
        assert torch.equal(g, torch.full_like(g, total)), (lo, hi, g)
        if self.flag is not None:
            assert self.flag[5:7].tolist() == self.want_flag, self.flag
        self.calls.append((lo, hi, scale))

m = Model()
for step in range(3):
    m._flat_grad = torch.full((100,), float(rank + 1))
    # step 1: only the LAST rank sees a NaN loss (per-GPU negatives: every rank has its own loss) -> every rank must see the flag
    nce_out = torch.zeros(8)
    if step == 1 and rank == world - 1:
        nce_out[5:7] = 1.0
    opt = Opt(m, nce_out, [1.0, 1.0] if step == 1 else [0.0, 0.0])
    sync = GradAllReduce(m, optimizer=opt)
    assert sync.world == world and sync.grad_scale == 1.0 / world
    sync.reduce_flag(nce_out)
    time.sleep(0.05 * ((rank * 7 + step * 3) % world))      # ranks reach their hooks at different times
    sync.hook(60, 100)
    assert opt.calls == []                       # nothing is applied before a later hook / finish
    time.sleep(0.03 * ((rank * 5 + step) % world))
    sync.hook(20, 60)
    assert opt.calls == [(60, 100, 1.0 / world)]
    time.sleep(0.02 * ((world - rank + step) % world))
    sync.finish()
    assert opt.calls == [(60, 100, 1.0 / world), (20, 60, 1.0 / world)]
    assert torch.equal(m._flat_grad, torch.full((100,), total))        # head [0, 20) reduced by finish(); step() updates it
    assert sync.pending == [] and sync.split is None
    assert nce_out[5:7].tolist() == ([1.0, 1.0] if step == 1 else [0.0, 0.0])
    # the bucket plan a data-parallel bench line reports (bench.py `data_parallel`): every range once, nothing twice
    d = sync.describe()
    assert [(b["lo"], b["hi"]) for b in d["buckets"]] == [(60, 100), (20, 60), (0, 20)] and d["covers_once"], d
    assert d["grad_bytes"] == 400 and sum(b["bytes"] for b in d["buckets"]) == 400 and d["world"] == world
# without an optimizer: plain overlapped reduction
m._flat_grad = torch.full((100,), float(rank + 1))
sync = GradAllReduce(m)
sync.hook(50, 100)
sync.finish()
assert torch.equal(m._flat_grad, torch.full((100,), total))
assert sync.describe()["covers_once"]
# the report bench.py builds from it: one entry per rank and the plan
sys.path.insert(0, sys.argv[1])
import bench
rep = bench.data_parallel_report(sync, [[5.0, 4.9, 6.0, 0.8, 0.05]] * world, 4)
for key in ("per_rank_ms_per_step", "per_rank_ms_per_step_median", "per_rank_ms_per_step_max", "per_rank_host_enqueue_ms_per_step",
            "allreduce_exposed_ms_per_step"):
    assert len(rep[key]) == world, key
assert rep["grad_bytes"] == 400 and rep["covers_once"] and len(rep["buckets"]) == 2
if rank == 0:
    print("PIECES-OK")
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_gradient_pieces_are_updated_only_after_their_reduction_gloo(tmp_path, world):
    """engine.GradAllReduce with an optimizer attached (the data-parallel train step), 2 and 4 ranks that reach their hooks at
    different times: Adam on a piece of the flat gradient is issued at the NEXT hook call / in finish(), after that piece's
    all-reduce, never before; the NaN flag of ONE rank (per-GPU negatives) reaches every rank before any update reads it."""
    script = tmp_path / "worker_pieces.py"
    script.write_text(WORKER_PIECES)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29641 + world), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "PIECES-OK" in outs[0]


def test_cqt_filter_design_matches_oracle_restatement():
    """Product and oracle hold independent restatements of the (absent, unpinned) librosa filter design; they must agree
    bit for bit, and the octave grouping must be the one the reference derives for its default bank."""
    import numpy as np
    from cpc_audio_amd import constant_q_transform as cq
    for args in ((16000, 30, 256, 32, 0.5), (16000, 560, 24, 8, 0.5), (44100, 30, 292, 32, 0.5)):
        a, la = cq.constant_q_filters(*args)
        b, lb = O.constant_q_filters(*args)
        assert a.shape == b.shape and np.array_equal(a, b) and np.array_equal(la, lb)
    m = cq.CQT(sr=16000, fmin=30, n_bins=256, bins_per_octave=32, filter_scale=0.5, hop_length=128)
    assert m.conv_kernel_sizes == [16384, 8192, 4096, 2048, 1024, 512, 256, 128, 64]
    assert [len(r) for r in m.conv_index_ranges] == [19, 32, 32, 32, 32, 32, 32, 32, 13]
    assert list(m.state_dict().keys()) == [f"conv_modules.{i}.weight" for i in range(9)]
    assert m.frames(97024) == 630


def test_downstream_probe_learns_a_separable_task():
    """ContrastiveEstimationTrainer.test_task (reference :305-350): seeded split, MLP probe; on linearly separable context
    vectors it must reach high accuracy (CPU; the probe is stock torch, outside the hot path)."""
    import numpy as np
    from cpc_audio_amd.contrastive_estimation_training import ContrastiveEstimationTrainer
    tr = ContrastiveEstimationTrainer.__new__(ContrastiveEstimationTrainer)
    tr.ar_size, tr.device, tr.verbose = 16, "cpu", False
    tr.test_task_set = type("S", (), {"files": ["a", "b", "c"]})()
    tr.model = torch.nn.Linear(1, 1)
    g = np.random.default_rng(0)
    labels = g.integers(0, 3, size=600)
    centers = g.normal(size=(3, 16)) * 3
    data = (centers[labels] + g.normal(size=(600, 16))).astype(np.float32)
    torch.manual_seed(0)
    acc = tr.test_task(data, labels.astype(np.int64))
    assert acc > 0.9


def test_reference_whole_module_snapshot_loads_without_the_reference(golden_dir):
    """The reference's snapshots are whole-module pickles naming its classes (setup_functions.py:134-164); the reader replaces
    classes of modules that are not importable by stand-ins and returns the state_dict, which loads strictly into this package's
    model of the same configuration."""
    import sys
    from cpc_audio_amd.checkpoint import load_reference_snapshot, reference_state_dict
    assert "audio_model" not in sys.modules or "reference" not in (getattr(sys.modules["audio_model"], "__file__", "") or "")
    path = os.path.join(golden_dir, "reference_snapshot_small.pt")
    meta = json.load(open(os.path.join(golden_dir, "reference_snapshot_small.json")))
    ref = np.load(os.path.join(golden_dir, "reference_snapshot_small.npz"))
    sd = reference_state_dict(path)
    assert set(sd.keys()) == set(ref.files)
    for k in ref.files:
        assert torch.equal(sd[k], torch.from_numpy(ref[k])), k
    c = meta["channels"]
    enc = AudioEncoder({'strides': [5, 4, 2, 2, 2], 'kernel_sizes': [10, 8, 4, 4, 4], 'channel_count': [c] * 5, 'bias': True})
    model = AudioPredictiveCodingModel(enc, AudioGRUModel(c, meta["ar_size"]), enc_size=c, ar_size=meta["ar_size"],
                                       visible_steps=meta["V"], prediction_steps=meta["K"], compute_dtype="fp32")
    load_reference_snapshot(model, path)
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), torch.from_numpy(ref[k])), k


def test_config_presets_equal_the_reference_effective_values(golden_dir):
    """cpc_audio_amd.configs against the reference's preset dictionaries AS THEY ARE AFTER IMPORT (tests/golden/effective_configs.json,
    written by generate_golden.py from the reference's own config modules: its dict.copy() aliasing makes e.g. architecture 5 = 6 = 7
    and strips architecture 2 of its residual branches)."""
    from cpc_audio_amd import configs
    ref = json.load(open(os.path.join(golden_dir, "effective_configs.json")))

    def plain(v):
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        if isinstance(v, (int, float, bool, str)) or v is None:
            return v
        return getattr(v, "__name__", type(v).__name__)

    names = [n for n in ref if hasattr(configs, n)]
    for must in ("scalogram_resnet_architecture_1", "scalogram_resnet_architecture_2", "scalogram_resnet_architecture_2_wo_res",
                 "scalogram_resnet_architecture_3", "scalogram_resnet_architecture_4", "scalogram_resnet_architecture_7",
                 "scalogram_resnet_architecture_8", "scalogram_resnet_architecture_9", "ar_conv_default_dict", "ar_conv_architecture_3",
                 "ar_conv_architecture_5", "attention_architecture_1", "attention_architecture_2", "cqt_default_dict", "cqt_high_res_dict"):
        assert must in names, must
    for n in names:
        mine, theirs = plain(getattr(configs, n)), ref[n]
        if "blocks" in theirs:
            # the survey's note on architecture 7: the classification config flips BatchNorm of its LAST block off after import
            assert len(mine["blocks"]) == len(theirs["blocks"]), n
            for i, (a, b) in enumerate(zip(mine["blocks"], theirs["blocks"])):
                b = dict(b)
                if n.startswith("scalogram_resnet") or n.startswith("ar_resnet"):
                    if i == 0 and theirs.get("phase"):
                        b["in_channels"] = a["in_channels"]           # the constructor overwrites it with 2 (scalogram_model.py:495-496)
                assert a == b, (n, i, a, b)
            mine = {k: v for k, v in mine.items() if k != "blocks"}
            theirs = {k: v for k, v in theirs.items() if k != "blocks"}
        for k, v in theirs.items():
            if k in mine:
                assert mine[k] == v, (n, k, mine[k], v)


def test_bench_starts_its_own_ranks_when_no_launcher_did(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: the N ranks are started as a child torch.distributed.run
    (never an exec of a process that touched the GPU), rendezvous on 127.0.0.1, and rank 0's JSON line is relayed.  Here the
    command line is checked and the relay is exercised with a stand-in child."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    # the relay: a child that prints noise and one JSON line
    fake = tmp_path / "child.py"
    fake.write_text("import sys\nprint('W0 noise')\nprint('{\"metric\": \"x\", \"n_gpus\": 4}')\nsys.exit(0)\n")
    src = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.launch_command = lambda n, argv: [sys.executable, {str(fake)!r}]; sys.exit(bench.self_launch(4))"
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == '{"metric": "x", "n_gpus": 4}' and "W0 noise" in r.stderr
