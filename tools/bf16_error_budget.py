"""Where does the bf16 loss error of BASELINE configs[2] come from?  The exact-f32 engine is run with ONE group of tensors at a time
rounded to bf16 (what bf16 storage does to it), and the relative change of the loss is printed per group.

    python tools/bf16_error_budget.py [--batch 128]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cpc_audio_amd import configs  # noqa: E402
from cpc_audio_amd.audio_model import AudioPredictiveCodingModel, ConvolutionalArModel  # noqa: E402
from cpc_audio_amd.scalogram_model import PreprocessingModule, ScalogramResidualEncoder, cqt_default_dict  # noqa: E402


def rnd(t):
    t.copy_(t.bfloat16().float())


def rnd_centred(g):
    """What storing (value - per-channel mean) in bf16 would do: the candidate fix for the residual stream (docs/DESIGN_HISTORY_r1-r3.md section 13)."""
    v = g.t.view(-1, g.C)
    nz = v != 0
    c = (v.sum(0) / nz.sum(0).clamp(min=1)).unsqueeze(0)
    v.copy_(torch.where(nz, (v - c).bfloat16().float() + c, v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--arch", default="7", choices=["7", "9"], help="7: BASELINE configs[2]; 9: the reference's script default e29")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, V, K = args.batch, 60, 16
    torch.manual_seed(0)
    if args.arch == "7":
        pre = PreprocessingModule(cqt_dict=cqt_default_dict, phase=True)
        enc = ScalogramResidualEncoder(args_dict=configs.fresh(configs.scalogram_resnet_architecture_7), preprocessing_module=pre)
        ar = ConvolutionalArModel(configs.fresh(configs.ar_conv_architecture_3))
        softplus, all_t, reg = True, False, 1.0
    else:
        V = 43
        cfg = configs.fresh(configs.scalogram_resnet_architecture_9)
        pre = PreprocessingModule(cqt_dict=configs.cqt_high_res_dict, phase=cfg['phase'], offset_zero=cfg['scalogram_offset_zero'],
                                  output_power=cfg['scalogram_output_power'], pooling=cfg['scalogram_pooling'], scaling=cfg['scalogram_scaling'])
        enc = ScalogramResidualEncoder(args_dict=cfg, preprocessing_module=pre)
        ar = ConvolutionalArModel(configs.fresh(configs.ar_conv_architecture_5))
        softplus, all_t, reg = False, True, 0.0
    model = AudioPredictiveCodingModel(enc, ar, enc_size=512, ar_size=256, visible_steps=V, prediction_steps=K, compute_dtype="fp32")
    pre, model = pre.to(dev), model.to(dev)
    wave = (torch.randn(B, model.item_length, generator=torch.Generator().manual_seed(11)) * 0.1).to(dev)
    x32 = pre(wave.unsqueeze(1)).clone()
    pre.cqt.precision = "bf16x3"
    x3 = pre(wave.unsqueeze(1)).clone()
    eng = model.engine_for(x32)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}

    zref = [None]

    def run(x=x32):
        model.load_state_dict(state)
        eng.forward(x)
        if all_t:
            eng.nce_all_forward_backward(softplus, reg)
        else:
            eng.nce_forward_backward(softplus, reg)
        z = eng.view_top()[:, :eng.T, :].float().clone()
        if zref[0] is None:
            zref[0] = z
        run.zerr = float((z - zref[0]).norm() / zref[0].norm())
        return float(eng.nce_out[0])

    base = run()
    print(f"exact-f32 loss {base:.6f}     (columns: relative loss change, relative L2 change of the encoder output z)")
    print(f"{'bf16x3 CQT':50s} {abs(run(x3) - base) / base:.2e}   z {run.zerr:.2e}")

    def with_patch(name, patches):
        saved = []
        for obj, attr, post in patches:
            orig = getattr(obj, attr)
            saved.append((obj, attr, orig))

            def wrapped(*a, _orig=orig, _post=post, **k):
                r = _orig(*a, **k)
                _post()
                return r
            setattr(obj, attr, wrapped)
        try:
            v = run()
        finally:
            for obj, attr, orig in saved:
                setattr(obj, attr, orig)
        print(f"{name:50s} {abs(v - base) / base:.2e}   z {run.zerr:.2e}   ({v:.5f})")

    blocks = eng.blocks
    # weights: the operand copies made by prepare()
    def w_post(c):
        def f():
            for n in ("w_fwd", "w_t", "w_dgrad"):
                t = getattr(c, n, None)
                if t is not None:
                    rnd(t)
        return f
    convs = [c for b in blocks for c in (b.conv_a, b.conv_b, b.res_conv) if c is not None]          # (block 0 of a stem engine has neither conv_a nor res_conv)
    with_patch("all encoder conv weights", [(c, "prepare", w_post(c)) for c in convs if not c.in_f32])
    for i, b in enumerate(blocks):
        if b.conv_a is not None and not b.conv_a.in_f32:
            with_patch(f"block {i} conv_a output (pre-BN / activation)", [(b.conv_a, "forward", lambda g=b.conv_a.y0: rnd(g.t))])
        with_patch(f"block {i} conv_b output (pre-BN / activation)", [(b.conv_b, "forward", lambda g=b.conv_b.y0: rnd(g.t))])
        if b.bn_a is not None:
            with_patch(f"block {i} BN_a output (activation a)", [(b.bn_a, "forward", lambda g=b.bn_a.a: rnd(g.t))])
        if b.bn_b is not None:
            with_patch(f"block {i} BN_b output (main)", [(b.bn_b, "forward", lambda g=b.bn_b.a: rnd(g.t))])
        if b.res_conv is not None and not b.res_conv.in_f32:
            with_patch(f"block {i} residual projection output", [(b.res_conv, "forward", lambda g=b.res_conv.y0: rnd(g.t))])
        with_patch(f"block {i} output (after the residual add)", [(b, "forward", lambda g=b.out: rnd(g.t))])
    with_patch("all encoder pre-BN conv outputs", [(c, "forward", lambda g=c.y0: rnd(g.t)) for b in blocks for c, bn in ((b.conv_a, b.bn_a), (b.conv_b, b.bn_b)) if bn is not None and c is not None and not c.in_f32])
    with_patch("all encoder BN outputs", [(bn, "forward", lambda g=bn.a: rnd(g.t)) for b in blocks for bn in (b.bn_a, b.bn_b) if bn is not None])
    ctx = eng.ctx
    ar_blocks = getattr(ctx, "blocks", [])
    with_patch("context: all conv outputs", [(b.conv, "forward", lambda g=b.conv.y0: rnd(g.t)) for b in ar_blocks])
    with_patch("context: all BN outputs", [(b.bn, "forward", lambda g=b.bn.a: rnd(g.t)) for b in ar_blocks if b.bn is not None])
    with_patch("context: all block outputs", [(b, "forward", lambda g=b.out: rnd(g.t)) for b in ar_blocks])
    with_patch("context: conv weights", [(c, "prepare", w_post(c)) for b in ar_blocks for c in (b.conv, b.res_conv) if c is not None])
    stream = [(b, "forward", lambda g=b.out: rnd(g.t)) for b in blocks[:-1]] + \
             [(b.res_conv, "forward", lambda g=b.res_conv.y0: rnd(g.t)) for b in blocks if b.res_conv is not None and not b.res_conv.in_f32]
    rest = [(c, "prepare", w_post(c)) for c in convs if not c.in_f32] + \
           [(c, "forward", lambda g=c.y0: rnd(g.t)) for b in blocks for c in (b.conv_a, b.conv_b) if c is not None and not c.in_f32] + \
           [(bn, "forward", lambda g=bn.a: rnd(g.t)) for b in blocks for bn in (b.bn_a, b.bn_b) if bn is not None] + \
           [(blocks[-1], "forward", lambda g=blocks[-1].out: rnd(g.t))]
    with_patch("ONLY the residual stream (block outputs, projections)", stream)
    with_patch("everything in the encoder EXCEPT the residual stream", rest)
    stream_c = [(b, "forward", lambda g=b.out: rnd_centred(g)) for b in blocks[:-1]] + \
               [(b.res_conv, "forward", lambda g=b.res_conv.y0: rnd_centred(g)) for b in blocks if b.res_conv is not None and not b.res_conv.in_f32]
    with_patch("ONLY the residual stream, stored CENTRED per channel", stream_c)
    for i, b in enumerate(blocks[:-1]):
        with_patch(f"block {i} output, stored centred per channel", [(b, "forward", lambda g=b.out: rnd_centred(g))])
    with_patch("the encoder with a CENTRED residual stream", rest[:-1] + stream_c + [(blocks[-1], "forward", lambda g=blocks[-1].out: rnd(g.t))])
    pre_bn = [(c, bn) for b in blocks for c, bn in ((b.conv_a, b.bn_a), (b.conv_b, b.bn_b)) if bn is not None and c is not None and not c.in_f32]
    with_patch("all encoder pre-BN conv outputs, stored CENTRED per channel", [(c, "forward", lambda g=c.y0: rnd_centred(g)) for c, _ in pre_bn])
    rest_c = [(c, "prepare", w_post(c)) for c in convs if not c.in_f32] + \
             [(c, "forward", (lambda g=c.y0: rnd_centred(g)) if any(c is cc for cc, _ in pre_bn) else (lambda g=c.y0: rnd(g.t)))
              for b in blocks for c in (b.conv_a, b.conv_b) if c is not None and not c.in_f32] + \
             [(bn, "forward", lambda g=bn.a: rnd(g.t)) for b in blocks for bn in (b.bn_a, b.bn_b) if bn is not None]
    with_patch("the encoder with CENTRED stream AND centred pre-BN outputs", rest_c + stream_c + [(blocks[-1], "forward", lambda g=blocks[-1].out: rnd(g.t))])
    with_patch("... and unrounded weights", rest_c[len([c for c in convs if not c.in_f32]):] + stream_c + [(blocks[-1], "forward", lambda g=blocks[-1].out: rnd(g.t))])
    with_patch("everything above at once", [(c, "prepare", w_post(c)) for c in convs if not c.in_f32] +
               [(c, "forward", lambda g=c.y0: rnd(g.t)) for c in convs if not c.in_f32] +
               [(bn, "forward", lambda g=bn.a: rnd(g.t)) for b in blocks for bn in (b.bn_a, b.bn_b) if bn is not None] +
               [(b, "forward", lambda g=b.out: rnd(g.t)) for b in blocks] +
               [(b.conv, "forward", lambda g=b.conv.y0: rnd(g.t)) for b in ar_blocks] +
               [(b.bn, "forward", lambda g=b.bn.a: rnd(g.t)) for b in ar_blocks if b.bn is not None] +
               [(b, "forward", lambda g=b.out: rnd(g.t)) for b in ar_blocks])


if __name__ == "__main__":
    main()
