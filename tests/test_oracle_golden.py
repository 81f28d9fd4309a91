"""Pins the CPU oracle (oracle/cpc_oracle.py) against golden vectors produced by the reference
itself (tests/golden/generate_golden.py) and against the invariants the reference's own
tests assert (tests/test_audioEncoder.py:19-48 of the reference)."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import cpc_oracle as O


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def _params(d, prefix="param/"):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in d.items() if k.startswith(prefix)}


def _close(a, b, rtol=2e-5, atol=2e-6):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def test_encoder_geometry_matches_reference_test():
    ds, rf = O.encoder_geometry(O.DEFAULT_STRIDES, O.DEFAULT_KERNELS)
    assert ds == 160 and rf == 465
    assert O.encoder_layer_lengths(4800, O.DEFAULT_STRIDES, O.DEFAULT_KERNELS)[-1] == 28
    assert O.encoder_layer_lengths(20480, O.DEFAULT_STRIDES, O.DEFAULT_KERNELS) == [4095, 1022, 510, 254, 126]
    assert O.item_length(rf, ds, 100, 12) == 18385


def test_encoder_reference_case(golden_dir):
    g = _load(golden_dir, "encoder_ref_test.npz")
    p = _params(g)
    y = O.encoder_forward(torch.from_numpy(g["x"]), p)
    assert list(y.shape) == [7, 32, 28]
    _close(y, g["y"])
    # receptive field probe with all-0.1 weights
    p01 = {k: torch.full_like(v, 0.1) for k, v in p.items()}
    for name, idx in (("inside", 464), ("outside", 465)):
        t = torch.zeros(7, 1, 2000)
        t[:, :, idx] = 1.0
        out = O.encoder_forward(t, p01)
        _close(out, g[f"probe_{name}"])
    assert g["probe_inside"][0, 0, 0] != 0 and g["probe_outside"][0, 0, 0] == 0


def test_gru_sequence(golden_dir):
    g = _load(golden_dir, "gru.npz")
    p = {k: v.clone().requires_grad_(True) for k, v in _params(g).items()}
    z = torch.from_numpy(g["z"]).requires_grad_(True)
    h, trace = O.gru_forward(z, p, return_trace=True)
    _close(h, g["h"])
    _close(torch.stack(trace, 1), g["trace"])
    (h * torch.from_numpy(g["dh"])).sum().backward()
    _close(z.grad, g["dz"], rtol=1e-4, atol=1e-6)
    for k, v in p.items():
        _close(v.grad, g["grad/" + k], rtol=1e-4, atol=1e-5)


def test_small_model_forward_and_scores(golden_dir):
    g = _load(golden_dir, "small_model.npz")
    meta = json.load(open(os.path.join(golden_dir, "small_model.json")))
    p = _params(g)
    x = torch.from_numpy(g["batch"]).unsqueeze(1)
    enc, acts = O.encoder_forward(x, p, return_all=True)
    for l in range(5):
        _close(acts[l], g[f"fwd/enc{l}"], rtol=1e-4, atol=1e-5)
    pz, tg, z, c = O.cpc_forward(x, p, meta["V"], meta["K"])
    _close(pz, g["fwd/predicted_z"], rtol=1e-4, atol=1e-5)
    _close(tg, g["fwd/targets"], rtol=1e-4, atol=1e-5)
    _close(z, g["fwd/z"], rtol=1e-4, atol=1e-5)
    _close(c, g["fwd/c"], rtol=1e-4, atol=1e-5)
    # score functions on the reference's own forward outputs
    rpz, rtg = torch.from_numpy(g["fwd/predicted_z"]), torch.from_numpy(g["fwd/targets"])
    _close(O.linear_scores(rpz, rtg), g["scores/linear"], rtol=1e-5, atol=1e-5)
    _close(O.softplus_scores(rpz, rtg), g["scores/softplus"], rtol=1e-5, atol=1e-5)
    _close(O.difference_scores(rpz, rtg), g["scores/difference"], rtol=1e-4, atol=1e-7)
    # the scores are not degenerate
    assert np.abs(g["scores/linear"]).mean() > 0.5


def test_small_model_train_steps(golden_dir):
    g = _load(golden_dir, "small_model.npz")
    meta = json.load(open(os.path.join(golden_dir, "small_model.json")))
    p0 = _params(g)
    data = torch.from_numpy(g["data"])
    for run in meta["runs"]:
        tr = O.OracleTrainer(p0, meta["V"], meta["K"], score=run["score"], all_timesteps=run["all_timesteps"],
                             regularization=run["reg"], lr=run["lr"])
        # batch composition: the sampler must reproduce the reference's index lists from the same seed
        random.seed(run["python_seed"])
        batches = []
        while len(batches) < run["steps"]:      # one sampler pass per epoch, RNG state carried over
            batches.extend(O.file_batch_sampler([meta["n_items"]], meta["B"]))
        assert batches[:run["steps"]] == run["batches"]
        for i in range(run["steps"]):
            batch = data[batches[i]]
            if i == 0 and run["steps"] == 1:
                loss, smax, grads = tr.loss_and_grads(batch)
                for k in [k for k in g if k.startswith(run["tag"] + "/grad/")]:
                    name = k.split("/grad/")[1]
                    ref = torch.from_numpy(g[k])
                    scale = ref.abs().max().item() + 1e-12
                    _close(grads[name] / scale, ref / scale, rtol=2e-4, atol=2e-5)
            loss, smax = tr.step(batch)
            assert abs(loss - run["loss"][i]) <= 2e-5 * max(1.0, abs(run["loss"][i])), (run, i, loss)
            assert abs(smax - run["max_score"][i]) <= 1e-4 * max(1.0, abs(run["max_score"][i]))
        after = [k for k in g if k.startswith(run["tag"] + "/param_after/")]
        for k in after:
            # Adam's update lr*m/(sqrt(v)+eps) is ill-conditioned where |g| ~ eps=1e-8: an element whose
            # gradient is that small may move by up to lr per step on rounding noise alone.  So: every
            # element within lr*steps, and all but a handful within float32 rounding.
            name = k.split("/param_after/")[1]
            got, ref = tr.params[name].detach(), torch.from_numpy(g[k])
            err = (got - ref).abs()
            assert err.max().item() <= run["lr"] * run["steps"]
            tight = err <= 2e-6 + 1e-4 * ref.abs()
            assert tight.float().mean().item() >= 0.999, (name, tight.float().mean().item())


def test_validate(golden_dir):
    g = _load(golden_dir, "validate.npz")
    meta = json.load(open(os.path.join(golden_dir, "validate.json")))
    p = _params(g)
    data = torch.from_numpy(g["data"])
    B, K, V = meta["B"], meta["K"], meta["V"]
    for run in meta["runs"]:
        batches = O.file_batch_sampler(meta["counts"], B, file_batch_size=8, drop_last=True, seed=0)
        assert [i for b in batches for i in b] == run["accessed"]
        fn = O.SCORE_FUNCTIONS[run["score"]]
        tot_l, tot_a, tot_s = torch.zeros(K), torch.zeros(K), 0.0
        for b in batches:
            pz, tg, _, _ = O.cpc_forward(data[b].unsqueeze(1), p, V, K)
            l, a, s = O.validation_terms(fn(pz, tg), run["all_timesteps"])
            tot_l += l
            tot_a += a
            tot_s += float(s)
        n = B * K if run["all_timesteps"] else B
        tot_l /= len(batches)
        tot_a /= len(batches)
        _close(tot_l, g[run["tag"] + "/losses"], rtol=1e-4, atol=1e-5)
        _close(tot_a, g[run["tag"] + "/accuracy"], rtol=0, atol=1e-6)
        _close(np.log(n) - tot_l, g[run["tag"] + "/mi"], rtol=1e-4, atol=1e-5)
        assert abs(tot_s / len(batches) - run["mean_score"]) < 1e-4


def test_samplers(golden_dir):
    s = json.load(open(os.path.join(golden_dir, "samplers.json")))
    for case in s["file_batch_sampler"]:
        if case["seed"] is None:
            random.seed(case["global_seed"])
        got = O.file_batch_sampler(case["counts"], case["batch_size"], case["file_batch_size"],
                                   case["drop_last"], case["seed"])
        assert got == case["batches"], case
    d = s["deterministic_sampler"]
    assert O.deterministic_order(d["n"], d["seed"]) == d["order"]
    # the two known answers recorded in SURVEY.md 8c
    assert O.file_batch_sampler([10], 4, 1, True, 0) == [[7, 8, 1, 5], [3, 4, 2, 0]]
    assert O.file_batch_sampler([6, 5], 4, 2, True, 0) == [[5, 3, 1, 0], [4, 2, 10, 6]]


def test_init_params_matches_reference_construction_order(golden_dir):
    """cfg1 fixture was produced from torch.manual_seed(0) + reference constructors; the oracle's
    init_params must give a model whose first-step loss equals the recorded one (checked in the
    slow test below) — here only the cheap shape / count check."""
    p = O.init_params()
    assert sum(v.numel() for v in p.values()) == 7414784


@pytest.mark.slow
def test_cfg1_trajectory(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "cfg1_trajectory.json")))
    g = torch.Generator().manual_seed(meta["data_seed"])
    data = torch.randn(meta["n_items"], meta["L"], generator=g)
    run = meta["runs"][0]
    p = O.init_params(seed=meta["model_seed"])
    tr = O.OracleTrainer(p, meta["V"], meta["K"], score=run["score"], regularization=run["reg"], lr=meta["lr"])
    random.seed(run["python_seed"])
    batches = O.file_batch_sampler([meta["n_items"]], meta["B"])
    assert batches[:5] == run["batches"]
    for i in range(2):
        loss, _ = tr.step(data[batches[i]])
        assert abs(loss - run["loss"][i]) < 2e-5 * abs(run["loss"][i])


def test_conv_ar_model(golden_dir):
    """AudioEncoder + ConvolutionalArModel (no batch norm / residual): forward, losses and gradients vs the reference."""
    g = _load(golden_dir, "conv_ar_model.npz")
    meta = json.load(open(os.path.join(golden_dir, "conv_ar_model.json")))
    p0 = _params(g)
    conv_ar = (meta["ar"]["kernel_sizes"], meta["ar"]["pooling"])
    data = torch.from_numpy(g["data"])
    pz, tg, z, c = O.cpc_forward(data[:meta["B"]].unsqueeze(1), p0, meta["V"], meta["K"], conv_ar=conv_ar)
    _close(c, g["fwd/c"], rtol=1e-4, atol=1e-5)
    _close(pz, g["fwd/predicted_z"], rtol=1e-4, atol=1e-5)
    for run in meta["runs"]:
        tr = O.OracleTrainer(p0, meta["V"], meta["K"], score=run["score"], all_timesteps=run["all_timesteps"],
                             regularization=run["reg"], lr=run["lr"], conv_ar=conv_ar)
        for i, idx in enumerate(run["batches"]):
            batch = data[idx]
            if run["steps"] == 1:
                loss, smax, grads = tr.loss_and_grads(batch)
                for k in [k for k in g if k.startswith(run["tag"] + "/grad/")]:
                    name = k.split("/grad/")[1]
                    ref = torch.from_numpy(g[k])
                    scale = ref.abs().max().item() + 1e-12
                    _close(grads[name] / scale, ref / scale, rtol=2e-4, atol=2e-5)
            loss, smax = tr.step(batch)
            assert abs(loss - run["loss"][i]) <= 5e-5 * max(1.0, abs(run["loss"][i])), (run["tag"], i, loss)


def test_attention_model(golden_dir):
    """AudioEncoder + AttentionModel (dropout 0): positional table, forward (incl. the in-place scaled z), losses and
    gradients vs the reference (attention_model.py:38-82)."""
    g = _load(golden_dir, "attention_model.npz")
    meta = json.load(open(os.path.join(golden_dir, "attention_model.json")))
    pe_key = "autoregressive_model.positional_encoder.pe"
    pe = g["param/" + pe_key]
    assert np.array_equal(O.positional_encoding(pe.shape[0], pe.shape[2]).numpy(), pe[:, 0, :])
    p0 = {k: v for k, v in _params(g).items() if k != pe_key}
    attention = (meta["ar"]["num_layers"], meta["ar"]["num_heads"])
    data = torch.from_numpy(g["data"])
    pz, tg, z, c = O.cpc_forward(data[:meta["B"]].unsqueeze(1), p0, meta["V"], meta["K"], attention=attention)
    _close(z, g["fwd/z"], rtol=1e-4, atol=1e-5)
    _close(tg, g["fwd/targets"], rtol=1e-4, atol=1e-5)
    _close(c, g["fwd/c"], rtol=1e-4, atol=2e-5)
    _close(pz, g["fwd/predicted_z"], rtol=1e-4, atol=2e-5)
    for run in meta["runs"]:
        tr = O.OracleTrainer(p0, meta["V"], meta["K"], score=run["score"], all_timesteps=run["all_timesteps"],
                             regularization=run["reg"], lr=run["lr"], attention=attention)
        for i, idx in enumerate(run["batches"]):
            batch = data[idx]
            if run["steps"] == 1:
                loss, smax, grads = tr.loss_and_grads(batch)
                for k in [k for k in g if k.startswith(run["tag"] + "/grad/")]:
                    name = k.split("/grad/")[1]
                    ref = torch.from_numpy(g[k])
                    scale = ref.abs().max().item() + 1e-12
                    _close(grads[name] / scale, ref / scale, rtol=2e-4, atol=3e-5)
            loss, smax = tr.step(batch)
            assert abs(loss - run["loss"][i]) <= 1e-4 * max(1.0, abs(run["loss"][i])), (run["tag"], i, loss)


def test_cqt_and_preprocessing(golden_dir):
    """CQT.forward and PreprocessingModule.forward restatements vs the reference run on the same filter coefficients
    (the coefficients themselves come from the restated librosa design: unpinned, see the oracle's header)."""
    g = _load(golden_dir, "cqt_small.npz")
    meta = json.load(open(os.path.join(golden_dir, "cqt_small.json")))
    c = meta["cqt"]
    bank, lengths = O.constant_q_filters(c["sample_rate"], c["fmin"], c["n_bins"], c["bins_per_octave"], c["filter_scale"])
    weights = [torch.from_numpy(g[f"weight/{i}"]) for i in range(len(meta["kernel_sizes"]))]
    for (lo, hi), size, w in zip(meta["index_ranges"], meta["kernel_sizes"], weights):
        off = (bank.shape[1] - size) // 2
        part = bank[lo:hi, off:bank.shape[1] - off] if off else bank[lo:hi]
        assert np.array_equal(np.concatenate([part.real, part.imag]).astype(np.float32), w[:, 0].numpy())
    x = torch.from_numpy(g["x"])
    cq = O.cqt_forward(x, weights, c["hop_length"])
    _close(cq, g["cqt"], rtol=1e-5, atol=1e-6)
    consts = O.phase_difference_constants(c["sample_rate"], c["fmin"], c["n_bins"], c["bins_per_octave"], c["hop_length"])
    assert np.array_equal(consts[0].numpy(), g["fixed_phase_diff"].reshape(-1))
    assert np.array_equal(consts[1].numpy(), g["scaling"].reshape(-1))
    for name, kw in meta["variants"].items():
        got = O.preprocessing_forward(cq, consts if kw.get("phase") else None, kw.get("offset_zero", False),
                                      kw.get("output_power", 1.0), kw.get("scaling", 1.0), kw.get("pooling"))
        ref = torch.from_numpy(g["pre/" + name])
        if kw.get("phase"):
            # the wrapped phase difference may legitimately flip by 2 pi * scaling where it sits within rounding of +-pi
            bad = (got - ref).abs() > 1e-4
            assert bad.float().mean().item() < 1e-3
            got = torch.where(bad, ref, got)
        _close(got, ref, rtol=1e-4, atol=1e-4)


def _scalogram_blocks(meta):
    blocks = [dict(b) for b in meta["blocks"]]
    if meta.get("phase", True):
        blocks[0]["in_channels"] = 2      # ScalogramResidualEncoder.__init__ with phase=True (scalogram_model.py:494-495)
    for b in blocks:
        for k in ("kernel_size_1", "kernel_size_2"):
            b[k] = tuple(b[k])
    return blocks


@pytest.mark.parametrize("fixture", ["scalogram_model", "scalogram_model_b", "scalogram_model_sep", "scalogram_model_gp", "scalogram_model_c"])
def test_scalogram_model(golden_dir, fixture):
    """PreprocessingModule + ScalogramResidualEncoder + GRU (fixture gp: + BatchNorm ConvolutionalArModel, every run with the
    Wasserstein gradient penalty): forward in train / eval mode, running statistics, trainer losses and all gradients vs the
    reference.  Fixture a: architecture-7 traits (phase channel, strided 3x3 + tall kernels
    with top padding); fixture b: architecture-8/9 traits (pooled power scalogram, tall first kernel, padded kernels,
    stride in the second convolution, identity and padded-projection residuals)."""
    g = _load(golden_dir, fixture + ".npz")
    meta = json.load(open(os.path.join(golden_dir, fixture + ".json")))
    blocks = _scalogram_blocks(meta)
    c = meta["cqt"]
    bank, _ = O.constant_q_filters(c["sample_rate"], c["fmin"], c["n_bins"], c["bins_per_octave"], c["filter_scale"])
    sizes, ranges = [256, 128, 64, 32], [(0, 3), (3, 11), (11, 19), (19, 24)]
    weights = []
    for size, (lo, hi) in zip(sizes, ranges):
        off = (bank.shape[1] - size) // 2
        part = bank[lo:hi, off:bank.shape[1] - off] if off else bank[lo:hi]
        weights.append(torch.from_numpy(np.concatenate([part.real, part.imag]).astype(np.float32)).unsqueeze(1))
    consts = O.phase_difference_constants(c["sample_rate"], c["fmin"], c["n_bins"], c["bins_per_octave"], c["hop_length"])

    pk = meta.get("pre", {"phase": True})

    def preprocess(wave):
        return O.preprocessing_forward(O.cqt_forward(wave.unsqueeze(1), weights, c["hop_length"]), consts if pk.get("phase") else None,
                                       pk.get("offset_zero", False), pk.get("output_power", 1.0), pk.get("scaling", 1.0), pk.get("pooling"))

    data = torch.from_numpy(g["data"])
    B, V, K = meta["B"], meta["V"], meta["K"]
    scal = preprocess(data[:B])
    ref_scal = torch.from_numpy(g["scalogram"])
    bad = (scal - ref_scal).abs() > 1e-4
    assert bad.float().mean().item() < 1e-3
    scal = ref_scal
    p0 = _params(g)
    for mode in ("eval", "train"):
        p = {k: v.clone() for k, v in p0.items()}
        with torch.no_grad():
            pz, tg, z, cc = O.cpc_forward(scal, p, V, K, scalogram=blocks, conv_ar=meta.get("ar"), training=mode == "train")
        _close(z, g[mode + "/z"], rtol=1e-4, atol=1e-5)
        _close(cc, g[mode + "/c"], rtol=1e-4, atol=1e-5)
        _close(pz, g[mode + "/predicted_z"], rtol=1e-4, atol=1e-5)
        if mode == "train":
            for k in [k for k in g if k.startswith("after_train_fwd/")]:
                _close(p[k.split("/", 1)[1]], g[k], rtol=1e-5, atol=1e-6)
    for run in meta["runs"]:
        tr = O.OracleTrainer(p0, V, K, score=run["score"], all_timesteps=run["all_timesteps"], regularization=run["reg"], lr=run["lr"],
                             scalogram=blocks, conv_ar=meta.get("ar"), gradient_penalty_factor=run.get("gp"))
        for i, idx in enumerate(run["batches"]):
            batch = preprocess(data[idx])
            if run["steps"] == 1:
                saved = {k: v.clone() for k, v in tr.buffers.items()}
                loss, smax, grads = tr.loss_and_grads(batch)
                tr.buffers = saved
                gkeys = [k for k in g if k.startswith(run["tag"] + "/grad/")]
                gmax = max(1.0, max(float(np.abs(g[k]).max()) for k in gkeys))
                for k in gkeys:
                    name = k.split("/grad/")[1]
                    ref = torch.from_numpy(g[k])
                    scale = ref.abs().max().item() + 1e-12
                    if scale < 1e-5 * gmax:   # conv biases in front of a BatchNorm: mathematically zero gradient (rounding noise)
                        assert grads[name].abs().max().item() < 1e-4 * gmax
                        continue
                    _close(grads[name] / scale, ref / scale, rtol=1e-3, atol=2e-4)
            loss, smax = tr.step(batch)
            assert abs(loss - run["loss"][i]) <= 2e-4 * max(1.0, abs(run["loss"][i])), (run["tag"], i, loss)
        for k in [k for k in g if k.startswith(run["tag"] + "/after/")]:
            _close(tr.buffers[k.split("/after/")[1]], g[k], rtol=1e-3, atol=2e-4)      # after 4 Adam steps


def test_conv_ar_batchnorm_residual(golden_dir):
    """ConvolutionalArModel with BatchNorm1d (forward, running statistics, trainer losses, gradients vs the reference) and
    with BatchNorm1d + residual branches (forward in train / eval mode vs the reference; the reference cannot run its own
    backward for this variant, see conv_ar_forward)."""
    g = _load(golden_dir, "conv_ar_bn.npz")
    meta = json.load(open(os.path.join(golden_dir, "conv_ar_bn.json")))
    data = torch.from_numpy(g["data"])
    B, V, K = meta["B"], meta["V"], meta["K"]
    for name, info in meta["variants"].items():
        p0 = _params(g, prefix=name + "/param/")
        for mode in ("eval", "train"):
            p = {k: v.clone() for k, v in p0.items()}
            with torch.no_grad():
                pz, tg, z, c = O.cpc_forward(data[:B].unsqueeze(1), p, V, K, conv_ar=info["ar"], training=mode == "train")
            _close(c, g[f"{name}/{mode}/c"], rtol=1e-4, atol=2e-5)
            _close(pz, g[f"{name}/{mode}/predicted_z"], rtol=1e-4, atol=2e-5)
            if mode == "train":
                for k in [k for k in g if k.startswith(name + "/after_train_fwd/")]:
                    _close(p[k.split("/after_train_fwd/")[1]], g[k], rtol=1e-5, atol=1e-6)
        for run in info["runs"]:
            tr = O.OracleTrainer(p0, V, K, score=run["score"], all_timesteps=run["all_timesteps"], regularization=run["reg"],
                                 lr=run["lr"], conv_ar=info["ar"])
            for i, idx in enumerate(run["batches"]):
                batch = data[idx]
                if run["steps"] == 1:
                    saved = {k: v.clone() for k, v in tr.buffers.items()}
                    loss, smax, grads = tr.loss_and_grads(batch)
                    tr.buffers = saved
                    for k in [k for k in g if k.startswith(f"{name}/{run['tag']}/grad/")]:
                        pname = k.split("/grad/")[1]
                        ref = torch.from_numpy(g[k])
                        scale = ref.abs().max().item() + 1e-12
                        if scale < 1e-6:
                            assert grads[pname].abs().max().item() < 1e-5
                            continue
                        _close(grads[pname] / scale, ref / scale, rtol=1e-3, atol=2e-4)
                loss, smax = tr.step(batch)
                assert abs(loss - run["loss"][i]) <= 2e-4 * max(1.0, abs(run["loss"][i])), (name, run["tag"], i, loss)


def test_ar_resnet_context(golden_dir):
    """ScalogramResidualEncoder as the context network (pooling inside the blocks, ceil mode): forward in train / eval mode,
    trainer losses and gradients vs the reference."""
    g = _load(golden_dir, "ar_resnet_model.npz")
    meta = json.load(open(os.path.join(golden_dir, "ar_resnet_model.json")))
    blocks = [dict(b, kernel_size_1=tuple(b["kernel_size_1"]), kernel_size_2=tuple(b["kernel_size_2"])) for b in meta["blocks"]]
    data = torch.from_numpy(g["data"])
    B, V, K = meta["B"], meta["V"], meta["K"]
    p0 = _params(g)
    for mode in ("eval", "train"):
        p = {k: v.clone() for k, v in p0.items()}
        with torch.no_grad():
            pz, tg, z, c = O.cpc_forward(data[:B].unsqueeze(1), p, V, K, ar_resnet=blocks, training=mode == "train")
        _close(c, g[mode + "/c"], rtol=1e-4, atol=2e-5)
        _close(pz, g[mode + "/predicted_z"], rtol=1e-4, atol=2e-5)
    for run in meta["runs"]:
        tr = O.OracleTrainer(p0, V, K, score=run["score"], all_timesteps=run["all_timesteps"], regularization=run["reg"], lr=run["lr"],
                             ar_resnet=blocks)
        for i, idx in enumerate(run["batches"]):
            batch = data[idx]
            if run["steps"] == 1:
                saved = {k: v.clone() for k, v in tr.buffers.items()}
                loss, smax, grads = tr.loss_and_grads(batch)
                tr.buffers = saved
                for k in [k for k in g if k.startswith(run["tag"] + "/grad/")]:
                    name = k.split("/grad/")[1]
                    ref = torch.from_numpy(g[k])
                    scale = ref.abs().max().item() + 1e-12
                    if scale < 1e-6:
                        assert grads[name].abs().max().item() < 1e-4
                        continue
                    _close(grads[name] / scale, ref / scale, rtol=1e-3, atol=2e-4)
            loss, smax = tr.step(batch)
            assert abs(loss - run["loss"][i]) <= 2e-4 * max(1.0, abs(run["loss"][i])), (run["tag"], i, loss)
