"""GPU unit tests of every C-ABI entry point against plain PyTorch CPU references (float64 where cheap).

Tolerances: f32 mode 2e-5 relative to the operand scale (exact-f32 MFMA, different summation order);
bf16 mode 1e-2 (inputs rounded to 8 significant bits, f32 accumulation).
"""
import ctypes as C
import math
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cpc_audio_amd import _hip  # noqa: E402
from oracle import cpc_oracle as O  # noqa: E402

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return 3e-5 if dt == torch.float32 else 1.2e-2


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def dev(t, dt=None):
    t = t.to(DEV)
    return t.to(dt) if dt is not None else t


def rounded(t, dt):
    """The value the device sees after storage in dt (so that references use identical inputs)."""
    return t.to(dt).double()


# --------------------------------------------------------------------------------------- gemm_nt
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 72, 96), (256, 128, 128), (130, 260, 40), (17, 8, 8)])
def test_gemm_nt_plain(dt, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    Bt = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    ref = rounded(A, dt) @ rounded(Bt, dt).T
    code = _hip.dtype_code(dt)
    dA, dB, db = dev(A, dt), dev(Bt, dt), dev(bias)
    out = torch.full((M, N), float("nan"), device=DEV, dtype=dt)
    _hip.gemm_nt(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(out), M, N, K, K, K, N, code)
    assert rel_err(out, ref) < tol(dt)
    # bias + relu, f32 output
    out32 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float32)
    _hip.gemm_nt(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(out32), M, N, K, K, K, N, code, bias=_hip.ptr(db),
                 flags=_hip.GEMM_RELU | _hip.GEMM_OUT_F32)
    ref2 = torch.relu(ref + bias.double())
    assert rel_err(out32, ref2) < tol(dt)
    assert not torch.isnan(out32).any()


@pytest.mark.parametrize("M,N,K", [(1100, 300, 128), (2048, 512, 1024), (1024, 264, 64), (1500, 136, 192)])
def test_gemm_nt_tile_variants_agree(M, N, K):
    """bf16: 256x256-tile kernel vs 128x128-tile kernel vs generic kernel on the same operands (incl. ragged M, N)."""
    g = torch.Generator().manual_seed(M + N + K)
    A, Bt = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    mask = torch.randn(M, N, generator=g)
    dA, dB, db, dM = dev(A, torch.bfloat16), dev(Bt, torch.bfloat16), dev(bias), dev(mask, torch.bfloat16)
    ref = torch.relu(rounded(A, torch.bfloat16) @ rounded(Bt, torch.bfloat16).T + bias.double())
    ref = torch.where(rounded(mask, torch.bfloat16) > 0, ref, torch.zeros_like(ref))
    outs = []
    for extra in (0, _hip.GEMM_SMALL_TILE, _hip.GEMM_FORCE_GENERIC, _hip.GEMM_NARROW_EPI, _hip.GEMM_NARROW_EPI | _hip.GEMM_SMALL_TILE,
                  _hip.GEMM_NO_DMA, _hip.GEMM_NO_DMA | _hip.GEMM_SMALL_TILE):
        out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
        _hip.gemm_nt(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(out), M, N, K, K, K, N, _hip.BF16, bias=_hip.ptr(db), mask=_hip.ptr(dM),
                     flags=_hip.GEMM_RELU | extra)
        assert rel_err(out, ref) < tol(torch.bfloat16)
        outs.append(out)
    # same accumulation order per output element in all three kernels -> bitwise identical results
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("M,N,K,lda,rpi", [(256 * 80, 1024, 256, 256, 0), (256 * 79 + 100, 1000, 512, 256, 0),
                                             (256 * 60 + 8, 1024, 512, 128, 97), (256 * 130, 512, 1024, 512, 640),
                                             # more than 512 tiles, M % 32 == 0, N % 256 == 0: the unmasked launches take the PERSISTENT kernel
                                             # (gemm_nt_persist_kernel: one workgroup per CU walks the tiles), incl. a ragged last M tile,
                                             # overlapped rows (tap-innermost K order) and pad-row handling
                                             (256 * 140, 1024, 512, 512, 0), (256 * 150 + 96, 1024, 1024, 256, 0), (256 * 300, 512, 128, 128, 320)])
def test_gemm_nt_register_epilogue_agrees(M, N, K, lda, rpi):
    """The register epilogue of the 256x256 kernel (16-byte stores straight from the accumulators, permuted B rows) against
    the LDS-staged epilogue (CPC_GEMM_NO_PERS) and the f64 reference: bias + relu, relu-backward mask, pad-row zeroing /
    skipping, ragged M and N, overlapped rows (lda < K: tap-innermost K order).  Bitwise equal: same sums in the same order."""
    g = torch.Generator().manual_seed(M + N + K)
    bf = torch.bfloat16
    X = torch.randn((M - 1) * lda + K + 64, generator=g)
    Bt = torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g)
    dX, dB, db = dev(X, bf), dev(Bt, bf), dev(bias)
    Xr = rounded(X, bf)
    A = Xr.as_strided((M, K), (lda, 1))
    ref0 = A @ rounded(Bt, bf).T
    ldc = N
    valid = rpi - 3 if rpi else 0
    rowsel = (torch.arange(M) % rpi < valid) if rpi else torch.ones(M, dtype=torch.bool)
    mask = torch.randn(M, N, generator=g)
    dM = dev(mask, bf)
    cases = [("bias_relu", dict(bias=_hip.ptr(db), flags=_hip.GEMM_RELU), torch.relu(ref0 + bias.double())),
             ("mask", dict(mask=_hip.ptr(dM)), torch.where(rounded(mask, bf) > 0, ref0, torch.zeros_like(ref0))),
             ("plain", dict(), ref0)]
    for name, kw, ref in cases:
        for skip in ((0, _hip.GEMM_SKIP_PAD_ROWS) if rpi else (0,)):
            outs = []
            for extra in (_hip.GEMM_DIRECT_MASK, _hip.GEMM_NO_PERS):
                out = torch.full((M, N), 7.0, device=DEV, dtype=bf)
                k2 = dict(kw)
                k2["flags"] = k2.get("flags", 0) | extra | skip
                _hip.gemm_nt(_hip.ptr(dX), _hip.ptr(dB), _hip.ptr(out), M, N, K, lda, K, ldc, _hip.BF16, c_rpi=rpi, c_item=rpi * ldc,
                             c_valid=valid, **k2)
                outs.append(out)
            r = ref.clone()
            if rpi:
                r[~rowsel] = 7.0 if skip else 0.0
            assert rel_err(outs[0], r) < tol(bf), (name, skip)
            assert torch.equal(outs[0], outs[1]), (name, skip)


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_addressing_mask_batch(dt):
    """Overlapping rows (strided-conv view), item addressing on A and C, pad-row zeroing, relu mask, batch strides."""
    g = torch.Generator().manual_seed(3)
    code = _hip.dtype_code(dt)
    # A: 3 items of 40 rows x 16 channels, conv k=8 (2 rows... here: K = 2*16 spans two consecutive rows), stride 1 row
    items, rows, ch = 3, 40, 16
    X = torch.randn(items * rows * ch + 64, generator=g)          # trailing guard
    Bt = torch.randn(24, 2 * ch, generator=g)
    dX, dB = dev(X, dt), dev(Bt, dt)
    M = items * rows
    valid = 37
    out = torch.full((items, rows, 24), float("nan"), device=DEV, dtype=dt)
    _hip.gemm_nt(_hip.ptr(dX), _hip.ptr(dB), _hip.ptr(out), M, 24, 2 * ch, ch, 2 * ch, 24, code,
                 c_rpi=rows, c_item=rows * 24, c_valid=valid)
    Xr = rounded(X, dt)
    Arows = torch.stack([Xr[m * ch: m * ch + 2 * ch] for m in range(M)])
    ref = (Arows @ rounded(Bt, dt).T).view(items, rows, 24)
    ref[:, valid:, :] = 0
    assert rel_err(out, ref) < tol(dt)
    assert (out[:, valid:, :] == 0).all()
    # relu-backward mask + item addressing on A (windows of 5 rows starting at row 7 of each item)
    win, start = 5, 7
    mask = torch.randn(items * win, 24, generator=g)
    dM = dev(mask, dt)
    out2 = torch.full((items * win, 24), float("nan"), device=DEV, dtype=dt)
    _hip.gemm_nt(_hip.ptr(dX, start * ch), _hip.ptr(dB), _hip.ptr(out2), items * win, 24, 2 * ch, ch, 2 * ch, 24, code,
                 a_rpi=win, a_item=rows * ch, mask=_hip.ptr(dM))
    rows_idx = [i * rows + start + w for i in range(items) for w in range(win)]
    ref2 = Arows[rows_idx] @ rounded(Bt, dt).T
    ref2 = torch.where(rounded(mask, dt) > 0, ref2, torch.zeros_like(ref2))
    assert rel_err(out2, ref2) < tol(dt)
    # batched (the score contraction pattern): A_k = P[:, k, :], Bt_k = Tg[:, k, :]
    Bn, Kn, E = 20, 3, 32
    P = torch.randn(Bn, Kn, E, generator=g)
    Tg = torch.randn(Bn, 7, E, generator=g)       # rows of 7 frames, use frames 4..6
    dP, dT = dev(P, dt), dev(Tg, dt)
    S = torch.full((Kn, Bn, Bn), float("nan"), device=DEV, dtype=torch.float32)
    _hip.gemm_nt(_hip.ptr(dP), _hip.ptr(dT, 4 * E), _hip.ptr(S), Bn, Bn, E, Kn * E, 7 * E, Bn, code,
                 a_batch=E, b_batch=E, c_batch=Bn * Bn, batch=Kn, flags=_hip.GEMM_OUT_F32)
    refS = torch.einsum("bke,cke->kbc", rounded(P, dt), rounded(Tg, dt)[:, 4:7])
    assert rel_err(S, refS) < tol(dt)


@pytest.mark.parametrize("dt,nb,ncol,RB,N,K", [(torch.bfloat16, 3, 100, 2, 72, 256), (torch.float32, 3, 100, 2, 72, 128),
                                                 (torch.bfloat16, 5, 5100, 2, 256, 512), (torch.bfloat16, 17, 3000, 1, 256, 384)])
def test_gemm_nt_bands_and_k_ranges(dt, nb, ncol, RB, N, K):
    """cpc_gemm_nt_args.a_rpi2 / c_rpi2 / k_ranges: rows ordered (band, column, row in band) over a column-major grid, each band
    summing only its own stage range of K, with a relu-backward mask addressed like C; against an f64 reference.  Outside its band's
    range a row holds zeros where its tile straddles two bands (the tile runs the union of the ranges) and NaN where the tile lies in
    one band: a launch that ignored the ranges would produce NaN there.  128- and 256-tile kernels."""
    g = torch.Generator().manual_seed(nb * 1000 + ncol + K)
    code, bk = _hip.dtype_code(dt), (64 if dt == torch.bfloat16 else 32)
    lda = K + 64                               # rows do not overlap: each can carry its own pattern
    rows_col = nb * RB                         # rows per column that this launch writes
    a_item = (rows_col + 2) * lda              # column stride of the source grid (a little more than its rows)
    nst = K // bk
    lo = torch.randint(0, nst, (nb,), generator=g)
    hi = torch.minimum(lo + 1 + torch.randint(0, nst, (nb,), generator=g), torch.tensor(nst))
    ranges = torch.stack([lo, hi], 1).to(torch.int32).contiguous()
    M = nb * ncol * RB
    m = torch.arange(M)
    band, col, rl = m // (ncol * RB), (m // RB) % ncol, m % RB
    a_off = band * (RB * lda) + col * a_item + rl * lda
    c_row = col * rows_col + band * RB + rl                            # C as [ncol][rows_col][N]
    tile = _hip.nt_tile(code, M, N, K, 0, 1)
    t0 = (m // tile) * tile
    one_band = (t0 // (ncol * RB)) == (torch.clamp(t0 + tile - 1, max=M - 1) // (ncol * RB))
    kk = torch.arange(K)
    inside = (kk[None, :] >= (lo[band] * bk)[:, None]) & (kk[None, :] < (hi[band] * bk)[:, None])
    vals = torch.randn(M, K, generator=g)
    rows_ref = torch.where(inside, rounded(vals, dt), torch.zeros(1, dtype=torch.float64))
    rows_dev = torch.where(inside, vals, torch.where(one_band[:, None], torch.tensor(float("nan")), torch.tensor(0.0)))
    X = torch.zeros(ncol * a_item + K + 64)
    X[a_off[:, None] + kk[None, :]] = rows_dev
    Bt = torch.randn(N, K, generator=g) * 0.2
    mask = torch.randn(ncol * rows_col, N, generator=g)
    dX, dB, dM, dR = dev(X, dt), dev(Bt, dt), dev(mask, dt), dev(ranges)
    out = torch.full((ncol * rows_col, N), float("nan"), device=DEV, dtype=dt)
    _hip.gemm_nt(_hip.ptr(dX), _hip.ptr(dB), _hip.ptr(out), M, N, K, lda, K, N, code, mask=_hip.ptr(dM),
                 a_rpi=RB, a_item=a_item, a_rpi2=ncol, a_item2=RB * lda, c_rpi=RB, c_item=rows_col * N, c_valid=RB, c_rpi2=ncol, c_item2=RB * N,
                 k_ranges=_hip.ptr(dR))
    ref = torch.zeros(ncol * rows_col, N, dtype=torch.float64)
    ref[c_row] = rows_ref @ rounded(Bt, dt).T
    ref = torch.where(rounded(mask, dt) > 0, ref, torch.zeros_like(ref))
    assert one_band.any() and not one_band.all()
    assert not torch.isnan(out.float()).any()
    assert rel_err(out, ref) < tol(dt)
    # the generic kernel has neither: refused, never ignored
    with pytest.raises(RuntimeError):
        _hip.gemm_nt(_hip.ptr(dX), _hip.ptr(dB), _hip.ptr(out), M, N, K, lda, K, N, code, a_rpi=RB, a_item=a_item, a_rpi2=ncol, a_item2=RB * lda,
                     c_rpi=RB, c_item=rows_col * N, c_valid=RB, c_rpi2=ncol, c_item2=RB * N, k_ranges=_hip.ptr(dR), flags=_hip.GEMM_FORCE_GENERIC)


@pytest.mark.parametrize("dt,B,W,H,C,cout,kh,kw,sh,sw", [(torch.bfloat16, 3, 21, 19, 32, 72, 3, 3, 2, 2), (torch.float32, 2, 12, 11, 16, 40, 2, 2, 1, 1),
                                                           (torch.bfloat16, 48, 80, 63, 32, 256, 3, 3, 2, 2), (torch.bfloat16, 8, 41, 34, 128, 256, 3, 3, 2, 2)])
def test_gemm_nt_gathered_rows_are_a_conv2d(dt, B, W, H, C, cout, kh, kw, sh, sw):
    """cpc_gemm_nt_args.k_taps / k_tap_stride / k_tap_stride_a: the window of an nn.Conv2d read straight from a channels-last grid
    [B][W][Ha][C] (piece j of a GEMM row = the kh rows x C channels of kernel column j, one grid column apart; K padded per piece to the
    stage size with zero weights) against F.conv2d.  One launch, batch = B; 128- and 256-tile kernels."""
    g = torch.Generator().manual_seed(B * 100 + W + C)
    code, bk = _hip.dtype_code(dt), (64 if dt == torch.bfloat16 else 32)
    Ha = H + 3                                                   # a few allocated rows more than valid ones, as the grids have
    x = torch.randn(B, W, Ha, C, generator=g)
    w = torch.randn(cout, C, kh, kw, generator=g) * 0.2
    Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    seg = (kh * C + bk - 1) // bk * bk
    Bt = torch.zeros(cout, kw, seg)
    Bt[:, :, :kh * C] = w.permute(0, 3, 2, 1).reshape(cout, kw, kh * C)          # [co][dw][(dh, c)]
    guard = torch.zeros(kw * Ha * C + seg)                                        # the last rows' pieces read beyond the grid
    dX, dB = dev(torch.cat([x.reshape(-1), guard]), dt), dev(Bt, dt)
    out = torch.full((B, Wo, Ho, cout), float("nan"), device=DEV, dtype=dt)
    _hip.gemm_nt(_hip.ptr(dX), _hip.ptr(dB), _hip.ptr(out), Wo * Ho, cout, kw * seg, sh * C, kw * seg, cout, code,
                 a_rpi=Ho, a_item=sw * Ha * C, a_batch=W * Ha * C, c_batch=Wo * Ho * cout, batch=B,
                 k_taps=kw, k_tap_stride=seg, k_tap_stride_a=Ha * C)
    xr = rounded(x, dt)[:, :, :H, :].permute(0, 3, 2, 1)                           # (B, C, H, W)
    ref = F.conv2d(xr, rounded(w, dt), stride=(sh, sw)).permute(0, 3, 2, 1)      # (B, Wo, Ho, cout)
    assert not torch.isnan(out.float()).any()
    assert rel_err(out, ref) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
def test_conv_w_prep_batch_and_group(dt):
    """cpc_conv_w_prep_plan / _batch: several convolutions' operand layouts in one launch, bitwise equal to one cpc_conv_w_prep each
    (different shapes, one without a data-gradient operand).  cpc_conv_w_prep_group: the G shifted kernel copies of a tall (kh,1)
    convolution computed G rows per GEMM row, against their definition."""
    g = torch.Generator().manual_seed(5)
    code = _hip.dtype_code(dt)
    shapes = [(64, 32, 5, 1), (96, 40, 8, 4), (32, 64, 3, 1), (128, 128, 10, 5)]
    ws = [torch.randn(co, ci, k, generator=g).to(DEV) for co, ci, k, s_ in shapes]
    single, batched, jobs = [], [], []
    for w, (co, ci, k, st) in zip(ws, shapes):
        D = -(-k // st)
        f1, d1 = torch.zeros(co * k * ci, device=DEV, dtype=dt), torch.zeros(st * ci * D * co, device=DEV, dtype=dt)
        f2, d2 = torch.full_like(f1, 3.0), torch.full_like(d1, 3.0)
        _hip.call("cpc_conv_w_prep", _hip.ptr(w), _hip.ptr(f1), _hip.ptr(d1), co, ci, k, st, code)
        single.append((f1, d1))
        batched.append((f2, d2))
        jobs.append((w, f2, d2, co, ci, k, st))
    _hip.ConvPrepBatch(jobs, DEV).run(code)
    for (f1, d1), (f2, d2) in zip(single, batched):
        assert torch.equal(f1, f2) and torch.equal(d1, d2)
    # grouped tall kernel
    cout, cin, kh, G = 32, 16, 7, 4
    Rw = Rd = kh + G - 1 + 2
    w = torch.randn(cout, cin, kh, generator=g)
    bias = torch.randn(cout, generator=g)
    wf = torch.full((G, cout, Rw, cin), 9.0, device=DEV, dtype=dt)
    wd = torch.full((G, cin, Rd, cout), 9.0, device=DEV, dtype=dt)
    bg = torch.full((G * cout,), 9.0, device=DEV)
    dw_, db_ = w.to(DEV), bias.to(DEV)
    _hip.call("cpc_conv_w_prep_group", _hip.ptr(dw_), _hip.ptr(db_), _hip.ptr(wf), _hip.ptr(wd), _hip.ptr(bg), cout, cin, kh, G, Rw, Rd, code)
    rf, rd = torch.zeros(G, cout, Rw, cin), torch.zeros(G, cin, Rd, cout)
    for dh in range(G):
        rf[dh, :, dh:dh + kh, :] = w.permute(0, 2, 1)
        rd[dh, :, dh:dh + kh, :] = w.permute(1, 2, 0).flip(1)
    assert torch.equal(wf.cpu(), rf.to(dt)) and torch.equal(wd.cpu(), rd.to(dt))
    assert torch.equal(bg.cpu(), bias.repeat(G))


@pytest.mark.parametrize("B,W,H,C,cout,kh,kw,sh,sw,nsplit", [(3, 21, 19, 32, 72, 3, 3, 2, 2, 1), (6, 41, 34, 32, 128, 3, 3, 2, 2, 4), (4, 12, 9, 128, 256, 2, 2, 1, 1, 2)])
def test_gemm_tn_two_level_rows_are_a_conv2d_weight_gradient(B, W, H, C, cout, kh, kw, sh, sw, nsplit):
    """cpc_gemm_tn_args.a_rpi2 / a_item2 (bf16 LDS-DMA kernel): the reduction rows of the A operand are the windows of an nn.Conv2d read
    straight from a channels-last grid [B][W][Ha][C] — (clip, output column, output row), one batch entry per kernel column — against
    torch's conv2d_weight; split slabs included."""
    g = torch.Generator().manual_seed(B * 10 + W)
    bf = torch.bfloat16
    Ha = H + 2
    x = torch.randn(B, W, Ha, C, generator=g)
    Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    dy = torch.randn(B, Wo, Ho + 1, cout, generator=g)               # one spare row per column, as the gradient grids have
    M, I = B * Wo * Ho, kh * C
    chunk = -(-(-(-M // nsplit)) // 64) * 64
    dX = dev(torch.cat([x.reshape(-1), torch.zeros(kw * Ha * C + I)]), bf)
    dY = dev(dy, bf)
    slabs = torch.full((nsplit, kw, I, cout), float("nan"), device=DEV)
    _hip.gemm_tn(_hip.ptr(dX), _hip.ptr(dY), _hip.ptr(slabs), M, I, cout, sh * C, cout, cout, _hip.BF16, a_rpi=Ho, a_item=sw * Ha * C, a_rpi2=Wo,
                 a_item2=W * Ha * C, a_batch=Ha * C, b_rpi=Ho, b_item=(Ho + 1) * cout, c_batch=I * cout, batch=kw, nsplit=nsplit, m_chunk=chunk,
                 slab_stride=kw * I * cout, flags=_hip.GEMM_OUT_F32)
    got = slabs.sum(0).view(kw, kh, C, cout).permute(3, 2, 1, 0).double().cpu()          # [co][c][dh][dw]
    xr = rounded(x, bf)[:, :, :H, :].permute(0, 3, 2, 1)                                  # (B, C, H, W)
    dyr = rounded(dy, bf)[:, :, :Ho, :].permute(0, 3, 2, 1)                                # (B, cout, Ho, Wo)
    ref = torch.nn.grad.conv2d_weight(xr, (cout, C, kh, kw), dyr, stride=(sh, sw))
    assert not torch.isnan(got).any()
    assert rel_err(got, ref) < 1e-4


# --------------------------------------------------------------------------------------- gemm_tn
@pytest.mark.parametrize("dt,flags", [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, _hip.GEMM_TN_NO_TR)])
@pytest.mark.parametrize("M,I,J", [(1000, 136, 72), (64, 128, 128), (129, 8, 264)])
def test_gemm_tn(dt, flags, M, I, J):
    g = torch.Generator().manual_seed(M + I)
    A = torch.randn(M, I, generator=g)
    B = torch.randn(M, J, generator=g)
    code = _hip.dtype_code(dt)
    dA, dB = dev(A, dt), dev(B, dt)
    ref = rounded(A, dt).T @ rounded(B, dt)
    out = torch.full((I, J), float("nan"), device=DEV, dtype=torch.float32)
    _hip.gemm_tn(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(out), M, I, J, I, J, J, code, flags=flags | _hip.GEMM_OUT_F32)
    assert rel_err(out, ref) < tol(dt)
    # split over m into slabs + deterministic reduce with the conv-weight permutation (i = (j, c) -> out[co][c][j])
    nsplit = 3
    blk = 64 if dt == torch.bfloat16 else 32
    chunk = -(-M // nsplit)
    chunk = -(-chunk // blk) * blk
    slabs = torch.full((nsplit, I, J), float("nan"), device=DEV, dtype=torch.float32)
    _hip.gemm_tn(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(slabs), M, I, J, I, J, J, code, nsplit=nsplit, m_chunk=chunk,
                 slab_stride=I * J, flags=flags | _hip.GEMM_OUT_F32)
    assert rel_err(slabs.sum(0), ref) < tol(dt)
    if I % 8 == 0:
        cin, kw = I // 4, 4          # i = j*cin + c
        outp = torch.full((J, cin, kw), float("nan"), device=DEV, dtype=torch.float32)
        _hip.call("cpc_reduce_slabs", _hip.ptr(slabs), _hip.ptr(outp), I, J, nsplit, I * J, cin, cin * kw, 1, kw)
        refp = ref.view(kw, cin, J).permute(2, 1, 0)
        assert rel_err(outp, refp) < tol(dt)
    # direct storage-dtype output
    outT = torch.full((I, J), float("nan"), device=DEV, dtype=dt)
    _hip.gemm_tn(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(outT), M, I, J, I, J, J, code, flags=flags)
    assert rel_err(outT, ref) < tol(dt)


@pytest.mark.parametrize("M,I,J,nsplit,ldb_pad", [(5003, 24, 32, 7, 0), (777, 8, 64, 3, 8), (64, 32, 32, 1, 0), (3000, 20, 16, 5, 0)])
def test_gemm_tn_skinny_f32(M, I, J, nsplit, ldb_pad):
    """The f32 path for tiny outputs over many rows (first-layer weight gradients of the scalogram encoder) agrees with the
    generic kernel; A rows may be padded (lda > I), B rows may use item addressing."""
    g = torch.Generator().manual_seed(M)
    lda = (I + 3) // 4 * 4 + 4
    A = torch.randn(M, lda, generator=g)
    rpi = 11
    items = -(-M // rpi)
    Bfull = torch.randn(items, rpi + 2, J + ldb_pad, generator=g)       # item stride (rpi + 2) rows, row stride J + pad
    Bm = Bfull[:, :rpi, :J].reshape(-1, J)[:M]
    ref = A[:, :I].double().T @ Bm.double()
    dA, dB = dev(A), dev(Bfull)
    chunk = ((M + nsplit - 1) // nsplit + 31) // 32 * 32
    outs = []
    for extra in (0, _hip.GEMM_FORCE_GENERIC):
        slabs = torch.full((nsplit, I, J), float("nan"), device=DEV)
        _hip.gemm_tn(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(slabs), M, I, J, lda, J + ldb_pad, J, _hip.F32, b_rpi=rpi,
                     b_item=(rpi + 2) * (J + ldb_pad), nsplit=nsplit, m_chunk=chunk if nsplit > 1 else 0, slab_stride=I * J,
                     flags=_hip.GEMM_OUT_F32 | extra)
        assert rel_err(slabs.sum(0), ref) < 3e-5
        outs.append(slabs)


@pytest.mark.parametrize("M,I,J,nsplit", [(4096, 264, 256, 2), (2050, 512, 520, 1), (8192, 4096, 512, 4)])
def test_gemm_tn_tile_variants_agree(M, I, J, nsplit):
    g = torch.Generator().manual_seed(M + I)
    A, B = torch.randn(M, I, generator=g), torch.randn(M, J, generator=g)
    dA, dB = dev(A, torch.bfloat16), dev(B, torch.bfloat16)
    ref = rounded(A, torch.bfloat16).T @ rounded(B, torch.bfloat16)
    chunk = -(-(-(-M // nsplit)) // 64) * 64 if nsplit > 1 else 0
    chunk = ((M + nsplit - 1) // nsplit + 63) // 64 * 64
    outs = []
    for extra in (0, _hip.GEMM_SMALL_TILE, _hip.GEMM_FORCE_GENERIC):
        slabs = torch.full((nsplit, I, J), float("nan"), device=DEV)
        _hip.gemm_tn(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(slabs), M, I, J, I, J, J, _hip.BF16, nsplit=nsplit, m_chunk=chunk,
                     slab_stride=I * J, flags=_hip.GEMM_OUT_F32 | extra)
        assert rel_err(slabs.sum(0), ref) < 2e-3
        outs.append(slabs)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dt", DTYPES)
def test_colsum(dt):
    g = torch.Generator().manual_seed(1)
    for M, N in [(1000, 64), (333, 768), (50, 2048), (77, 8)]:
        X = torch.randn(M, N, generator=g)
        dX = dev(X, dt)
        nb = 7
        slabs = torch.full((nb, N), float("nan"), device=DEV)
        _hip.call("cpc_colsum", _hip.ptr(dX), _hip.ptr(slabs), M, N, N, nb, _hip.dtype_code(dt))
        ref = rounded(X, dt).sum(0)
        assert rel_err(slabs.sum(0), ref) < 1e-5
        out = torch.full((N,), float("nan"), device=DEV)
        _hip.call("cpc_reduce_slabs", _hip.ptr(slabs), _hip.ptr(out), 1, N, nb, N, 1, 1, 0, 0)
        assert rel_err(out, ref) < 1e-5


# --------------------------------------------------------------------------------------- conv layer 1
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C,kw,stride", [(64, 10, 5), (512, 10, 5), (32, 7, 3)])
def test_conv1_fwd_bwd(dt, C, kw, stride):
    g = torch.Generator().manual_seed(C)
    B, L = 3, 1234
    x = torch.randn(B, L, generator=g)
    w = torch.randn(C, 1, kw, generator=g) * 0.3
    bias = torch.randn(C, generator=g) * 0.1
    Lv = (L - kw) // stride + 1
    La = Lv + 3
    code = _hip.dtype_code(dt)
    dx, dw, db = dev(x), dev(w), dev(bias)
    y = torch.full((B, La, C), float("nan"), device=DEV, dtype=dt)
    ybits = torch.full((B * La * C // 8,), 0xA5, device=DEV, dtype=torch.uint8) if C == 512 else None
    _hip.call("cpc_conv1_fwd", _hip.ptr(dx), _hip.ptr(dw), _hip.ptr(db), _hip.ptr(y), B, C, stride, kw, L, Lv, La, 1, code, _hip.ptr(ybits))
    if ybits is not None:
        # sign-bit mask (taken from the f32 values; for this data the same as of the stored ones): bit e of byte i = y[8 i + e] > 0
        want = (y.reshape(-1, 8) > 0).to(torch.int32) * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))
        assert torch.equal(ybits.to(torch.int32), want.sum(1))
        yp = torch.full_like(y, float("nan"))
        _hip.call("cpc_conv1_fwd", _hip.ptr(dx), _hip.ptr(dw), _hip.ptr(db), _hip.ptr(yp), B, C, stride, kw, L, Lv, La, 1, code, None)
        assert torch.equal(yp, y)                      # the row order inside a block differs, the values do not
    xr = x.double().unsqueeze(1).requires_grad_(False)
    wr = w.double().requires_grad_(True)
    br = bias.double().requires_grad_(True)
    ref = torch.relu(F.conv1d(xr, wr, br, stride=stride))            # (B, C, Lv)
    assert rel_err(y[:, :Lv].float().transpose(1, 2), ref) < (1e-5 if dt == torch.float32 else 6e-3)
    assert (y[:, Lv:] == 0).all()
    # backward: dy random (as if already relu-masked)
    dy = torch.randn(B, La, C, generator=g)
    dy[:, Lv:] = 0
    ddy = dev(dy, dt)
    nblk, nbb = 3, 2
    slabs = torch.full((nbb * nblk, kw + 1, C), float("nan"), device=DEV)
    _hip.call("cpc_conv1_bwd", _hip.ptr(dx), _hip.ptr(ddy), _hip.ptr(slabs), B, C, stride, kw, L, Lv, La, nblk, nbb, code)
    pre = F.conv1d(xr, wr, br, stride=stride)
    (pre * rounded(dy, dt)[:, :Lv].transpose(1, 2)).sum().backward()
    got = slabs.sum(0)
    assert rel_err(got[:kw].T, wr.grad[:, 0, :]) < 2e-5
    assert rel_err(got[kw], br.grad) < 2e-5


# --------------------------------------------------------------------------------------- conv layers >= 2
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Cin,Cout,kw,stride", [(64, 64, 8, 4), (32, 64, 4, 2), (64, 32, 4, 2)])
def test_conv_fwd_dgrad_wgrad(dt, Cin, Cout, kw, stride):
    g = torch.Generator().manual_seed(Cin + kw)
    code = _hip.dtype_code(dt)
    B = 3
    Lin_valid = 67
    Lout_valid = (Lin_valid - kw) // stride + 1
    Lout_alloc = Lout_valid + 2
    Lin_alloc = Lout_alloc * stride
    assert Lin_alloc >= Lin_valid + 1
    guard = 16 * max(Cin, Cout)
    x = torch.relu(torch.randn(B, Lin_alloc, Cin, generator=g))
    x[:, Lin_valid:] = 0
    w = torch.randn(Cout, Cin, kw, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g) * 0.1

    def padded(t):
        buf = torch.zeros(guard + t.numel() + guard, device=DEV, dtype=dt)
        buf[guard:guard + t.numel()] = t.reshape(-1).to(DEV).to(dt)
        return buf

    xbuf = padded(x)
    dw_, db_ = dev(w), dev(bias)
    wf = torch.empty(Cout * kw * Cin, device=DEV, dtype=dt)
    D = -(-kw // stride)
    wd = torch.empty(stride * Cin * D * Cout, device=DEV, dtype=dt)
    _hip.call("cpc_conv_w_prep", _hip.ptr(dw_), _hip.ptr(wf), _hip.ptr(wd), Cout, Cin, kw, stride, code)
    assert rel_err(wf.float().view(Cout, kw, Cin), rounded(w, dt).permute(0, 2, 1)) == 0
    ybuf = torch.full((guard + B * Lout_alloc * Cout + guard,), float("nan"), device=DEV, dtype=dt)
    for relu in (1, 0):
        _hip.call("cpc_conv_fwd", _hip.ptr(xbuf, guard), _hip.ptr(wf), _hip.ptr(db_), _hip.ptr(ybuf, guard), B, Cin, Cout, kw,
                  stride, Lout_alloc, Lout_valid, relu, C.c_longlong(guard), code)
        y = ybuf[guard:guard + B * Lout_alloc * Cout].view(B, Lout_alloc, Cout)
        xr = rounded(x, dt)[:, :Lin_valid].transpose(1, 2)
        ref = F.conv1d(xr, rounded(w, dt), bias.double(), stride=stride)
        if relu:
            ref = torch.relu(ref)
        assert rel_err(y[:, :Lout_valid].float().transpose(1, 2), ref) < tol(dt)
        assert (y[:, Lout_valid:] == 0).all()
    # gradients
    dy = torch.randn(B, Lout_alloc, Cout, generator=g)
    dy[:, Lout_valid:] = 0
    dybuf = padded(dy)
    dxbuf = torch.full((guard + B * Lin_alloc * Cin + guard,), float("nan"), device=DEV, dtype=dt)
    _hip.call("cpc_conv_dgrad", _hip.ptr(dybuf, guard), _hip.ptr(wd), _hip.ptr(xbuf, guard), _hip.ptr(dxbuf, guard), B, Cin, Cout,
              kw, stride, Lout_alloc, Lin_valid, C.c_longlong(guard), code, None, None)
    xin = rounded(x, dt)[:, :Lin_valid].transpose(1, 2).clone().requires_grad_(True)
    wr = rounded(w, dt).clone().requires_grad_(True)
    out = F.conv1d(xin, wr, None, stride=stride)
    (out * rounded(dy, dt)[:, :Lout_valid].transpose(1, 2)).sum().backward()
    dx = dxbuf[guard:guard + B * Lin_alloc * Cin].view(B, Lin_alloc, Cin)
    ref_dx = xin.grad.transpose(1, 2) * (rounded(x, dt)[:, :Lin_valid] > 0)
    assert rel_err(dx[:, :Lin_valid], ref_dx) < tol(dt)
    assert (dx[:, Lin_valid:] == 0).all()
    nsplit = 2
    slabs = torch.full((nsplit, kw * Cin, Cout), float("nan"), device=DEV)
    _hip.call("cpc_conv_wgrad", _hip.ptr(xbuf, guard), _hip.ptr(dybuf, guard), _hip.ptr(slabs), B, Cin, Cout, kw, stride, Lout_alloc,
              nsplit, C.c_longlong(guard), code)
    got = slabs.sum(0).view(kw, Cin, Cout).permute(2, 1, 0)
    assert rel_err(got, wr.grad) < tol(dt)
    wg = torch.full((Cout, Cin, kw), float("nan"), device=DEV)
    _hip.call("cpc_reduce_conv_w", _hip.ptr(slabs), _hip.ptr(wg), Cin, Cout, kw, nsplit, kw * Cin * Cout)
    assert rel_err(wg, wr.grad) < tol(dt)


@pytest.mark.parametrize("Cin,B,La1", [(256, 8, 1600), (512, 5, 1312)])
def test_conv_dgrad_fused_with_layer1_weight_gradient(Cin, B, La1):
    """cpc_conv_dgrad_conv1 + cpc_conv1_fused_reduce against cpc_conv_dgrad followed by the plain sums over the stored tile."""
    dt, code = torch.bfloat16, _hip.BF16
    g = torch.Generator().manual_seed(Cin + B)
    Cout, kw, stride, kw1, s1 = 64, 8, 4, 10, 5
    D = -(-kw // stride)
    La0 = stride * La1
    Lv0 = La0 - 7
    ldx = (La0 - 1) * s1 + kw1 + 3
    guard = 16 * max(Cin, Cout)
    xwave = torch.randn(B, ldx, generator=g)
    act0 = torch.randn(B, La0, Cin, generator=g)                     # layer-1 output (its sign is the ReLU mask)
    act0[:, Lv0:] = 0
    dy = torch.randn(B, La1, Cout, generator=g)
    dy[:, La1 - 3:] = 0
    w = torch.randn(Cout, Cin, kw, generator=g) * 0.05

    def padded(t):
        buf = torch.zeros(guard + t.numel() + guard, device=DEV, dtype=dt)
        buf[guard:guard + t.numel()] = t.reshape(-1).to(DEV).to(dt)
        return buf

    abuf, dybuf = padded(act0), padded(dy)
    wf = torch.empty(Cout * kw * Cin, device=DEV, dtype=dt)
    wd = torch.empty(stride * Cin * D * Cout, device=DEV, dtype=dt)
    _hip.call("cpc_conv_w_prep", _hip.ptr(w.to(DEV)), _hip.ptr(wf), _hip.ptr(wd), Cout, Cin, kw, stride, code)
    dxbuf = torch.zeros(guard + B * La0 * Cin + guard, device=DEV, dtype=dt)
    _hip.call("cpc_conv_dgrad", _hip.ptr(dybuf, guard), _hip.ptr(wd), _hip.ptr(abuf, guard), _hip.ptr(dxbuf, guard), B, Cin, Cout,
              kw, stride, La1, Lv0, C.c_longlong(guard), code, None, None)
    G = dxbuf[guard:guard + B * La0 * Cin].view(B, La0, Cin).double().cpu()[:, :Lv0]
    win = xwave.double().unfold(1, kw1, s1)[:, :Lv0]                 # (B, Lv0, kw1)
    ref_w = torch.einsum("btc,btj->cj", G, win)
    ref_b = G.sum((0, 1))
    n_sl = int(_hip.lib().cpc_conv_dgrad_conv1_floats(B, Cin, stride, La1, kw1, 0))
    n_tmp = int(_hip.lib().cpc_conv_dgrad_conv1_floats(B, Cin, stride, La1, kw1, 1))
    slabs = torch.full((n_sl,), float("nan"), device=DEV)
    tmp = torch.full((n_tmp,), float("nan"), device=DEV)
    xd = xwave.to(DEV)
    _hip.call("cpc_conv_dgrad_conv1", _hip.ptr(dybuf, guard), _hip.ptr(wd), _hip.ptr(abuf, guard), _hip.ptr(xd), _hip.ptr(slabs), B, Cin,
              Cout, kw, stride, La1, ldx, kw1, s1, Lv0, C.c_longlong(guard), code, None)
    dw = torch.full((Cin, 1, kw1), float("nan"), device=DEV)
    db = torch.full((Cin,), float("nan"), device=DEV)
    _hip.call("cpc_conv1_fused_reduce", _hip.ptr(slabs), _hip.ptr(tmp), _hip.ptr(dw), _hip.ptr(db), B, Cin, stride, La1, kw1)
    assert rel_err(dw[:, 0, :], ref_w) < 2e-4          # same bf16 tile in both paths; x enters as a (hi, lo) bf16 pair
    assert rel_err(db, ref_b) < 2e-4
    # the same with the mask as sign bits (one byte per 8 elements, mirroring the activation buffer): identical slabs, and the
    # plain data gradient likewise identical
    bits = torch.zeros(abuf.numel() // 8, device=DEV, dtype=torch.uint8)
    _hip.call("cpc_sign_bits", _hip.ptr(abuf), _hip.ptr(bits), C.c_longlong(abuf.numel()), code)
    slabs2 = torch.full((n_sl,), float("nan"), device=DEV)
    _hip.call("cpc_conv_dgrad_conv1", _hip.ptr(dybuf, guard), _hip.ptr(wd), None, _hip.ptr(xd), _hip.ptr(slabs2), B, Cin,
              Cout, kw, stride, La1, ldx, kw1, s1, Lv0, C.c_longlong(guard), code, _hip.ptr(bits, guard // 8))
    assert torch.equal(slabs2, slabs)
    dxbuf2 = torch.zeros_like(dxbuf)
    nf = int(_hip.lib().cpc_conv_dgrad_colsum_floats(B, Cin, stride, La1))
    cs = torch.full((nf,), float("nan"), device=DEV)
    _hip.call("cpc_conv_dgrad", _hip.ptr(dybuf, guard), _hip.ptr(wd), None, _hip.ptr(dxbuf2, guard), B, Cin, Cout,
              kw, stride, La1, Lv0, C.c_longlong(guard), code, _hip.ptr(bits, guard // 8), _hip.ptr(cs))
    assert torch.equal(dxbuf2, dxbuf)
    # ... and the per-tile column sums of what it stored: summed over tiles and phases = the column sums of the stored gradient
    got = cs.view(-1, Cin).double().sum(0).cpu()
    want = dxbuf[guard:guard + B * La0 * Cin].view(-1, Cin).double().sum(0).cpu()
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_sign_bits(dt):
    """cpc_sign_bits: bit e of byte i = element 8 i + e is > 0 (zeros of both signs, negatives and NaN with the sign set: 0)."""
    g = torch.Generator().manual_seed(5)
    n = 32 * 12345
    x = torch.randn(n, generator=g)
    x[::7] = 0.0
    x[3::11] = -0.0
    x[5::13] = torch.relu(x[5::13])
    xd = x.to(DEV).to(dt)
    bits = torch.full((n // 8,), 0xA5, device=DEV, dtype=torch.uint8)
    _hip.call("cpc_sign_bits", _hip.ptr(xd), _hip.ptr(bits), C.c_longlong(n), _hip.dtype_code(dt))
    want = ((xd.view(-1, 8) > 0).to(torch.int32) * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(1)
    assert torch.equal(bits.to(torch.int32), want)
    assert _hip.lib().cpc_sign_bits(_hip.ptr(xd), _hip.ptr(bits), C.c_longlong(n - 8), _hip.dtype_code(dt), _hip.stream_ptr()) == -22


@pytest.mark.parametrize("N,K,relu,rpi", [(32, 8, 0, 0), (32, 32, 1, 0), (16, 20, 0, 0), (32, 20, 1, 37), (8, 4, 0, 0)])
def test_gemm_nt_skinny_f32(N, K, relu, rpi):
    """The f32 NT path for tiny N and K over many rows (first scalogram convolutions) against torch, incl. the padded-row
    output mapping (rows >= c_valid of every item are written as zeros)."""
    g = torch.Generator().manual_seed(N * 100 + K)
    M = 37 * 150 if rpi else 6001
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    dA, dW, db = dev(A), dev(W), dev(bias)
    ref = A.double() @ W.double().t() + bias.double()
    if relu:
        ref = torch.relu(ref)
    if rpi:
        items, alloc, valid = M // rpi, rpi + 3, rpi - 2
        out = torch.full((items, alloc, N), float("nan"), device=DEV)
        _hip.gemm_nt(_hip.ptr(dA), _hip.ptr(dW), _hip.ptr(out), M, N, K, K, K, N, _hip.F32, bias=_hip.ptr(db), c_rpi=rpi, c_item=alloc * N,
                     c_valid=valid, flags=_hip.GEMM_RELU if relu else 0)
        got = out[:, :rpi].reshape(M, N)
        refv = ref.view(items, rpi, N).clone()
        refv[:, valid:] = 0
        assert rel_err(got, refv.view(M, N)) < 1e-5
        assert torch.isnan(out[:, rpi:]).all()
    else:
        out = torch.full((M, N), float("nan"), device=DEV)
        _hip.gemm_nt(_hip.ptr(dA), _hip.ptr(dW), _hip.ptr(out), M, N, K, K, K, N, _hip.F32, bias=_hip.ptr(db),
                     flags=_hip.GEMM_RELU if relu else 0)
        assert rel_err(out, ref) < 1e-5


# --------------------------------------------------------------------------------------- GRU
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,V,H", [(7, 13, 64), (20, 5, 32), (16, 3, 256)])
def test_gru_fwd_bwd(dt, B, V, H):
    g = torch.Generator().manual_seed(B + H)
    code = _hip.dtype_code(dt)
    w_hh = torch.randn(3 * H, H, generator=g) / math.sqrt(H)
    b_hh = torch.randn(3 * H, generator=g) * 0.1
    Gi = torch.randn(B, V, 3 * H, generator=g)
    dc = torch.randn(B, H, generator=g)
    dW, db, dGi_in, ddc = dev(w_hh), dev(b_hh), dev(Gi), dev(dc)
    wfrag = torch.empty(3 * H * H, device=DEV, dtype=dt)
    wTfrag = torch.empty(3 * H * H, device=DEV, dtype=dt)
    _hip.call("cpc_prep_frag", _hip.ptr(dW), _hip.ptr(wfrag), 3 * H, H, H, 0, code)
    _hip.call("cpc_prep_frag", _hip.ptr(dW), _hip.ptr(wTfrag), H, 3 * H, H, 1, code)
    Hall = torch.full((B, V + 1, H), float("nan"), device=DEV, dtype=dt)
    tape = torch.zeros(_hip.lib().cpc_gru_tape_elems(B, V, H, code), device=DEV, dtype=dt)
    c = torch.full((B, H), float("nan"), device=DEV)
    dGi_T = dev(Gi, dt)           # the input projection is stored in the storage dtype
    _hip.call("cpc_gru_fwd", _hip.ptr(dGi_T), _hip.ptr(wfrag), _hip.ptr(db), _hip.ptr(Hall), _hip.ptr(tape), _hip.ptr(c), B, V, H, code)
    # reference (float64) with the weights as the device sees them
    wr = rounded(w_hh, dt).requires_grad_(True)
    br = b_hh.double().requires_grad_(True)
    gir = rounded(Gi, dt).requires_grad_(True)
    h = torch.zeros(B, H, dtype=torch.float64)
    hs = [h]
    for t in range(V):
        gh = h @ wr.T + br
        r = torch.sigmoid(gir[:, t, :H] + gh[:, :H])
        u = torch.sigmoid(gir[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gir[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - u) * n + u * h
        hs.append(h)
    t_f = 2e-5 if dt == torch.float32 else 2e-2
    assert rel_err(c, h) < t_f
    assert rel_err(Hall, torch.stack(hs, 1)) < t_f
    (h * dc.double()).sum().backward()
    dG = torch.full((B, V, 4 * H), float("nan"), device=DEV, dtype=dt)
    _hip.call("cpc_gru_bwd", _hip.ptr(ddc), _hip.ptr(tape), _hip.ptr(wTfrag), _hip.ptr(dG), B, V, H, code)
    dGi = dG[:, :, :3 * H]
    dGh = torch.cat([dG[:, :, :2 * H], dG[:, :, 3 * H:]], dim=2)
    t_b = 5e-5 if dt == torch.float32 else 4e-2
    assert rel_err(dGi, gir.grad) < t_b
    # dGh: gradient wrt (h W_hh^T + b_hh): its column sums are the b_hh gradient
    assert rel_err(dGh.double().sum((0, 1)), br.grad) < t_b
    # dW_hh = sum_t dGh_t^T h_{t-1}
    dWhh = torch.einsum("bvg,bvh->gh", dGh.double().cpu(), Hall[:, :V].double().cpu())
    assert rel_err(dWhh, wr.grad) < t_b


# --------------------------------------------------------------------------------------- InfoNCE
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("softplus", [0, 1])
@pytest.mark.parametrize("B,K,reg", [(6, 4, 1.0), (40, 12, 0.01), (33, 3, 0.5)])
def test_nce_loss(dt, softplus, B, K, reg):
    g = torch.Generator().manual_seed(B * 3 + K)
    S = torch.randn(K, B, B, generator=g) * 3.0
    S[0, 0, 0] = 25.0                                  # exercises the softplus threshold branch
    code = _hip.dtype_code(dt)
    ld = (B + 7) // 8 * 8                               # leading dimension padded as the engine does
    Sp = torch.full((K, B, ld), 7.0)                    # pad columns hold junk the kernel must ignore
    Sp[:, :, :B] = S
    dS_in = dev(Sp)
    dSp = torch.full((K, B, ld), float("nan"), device=DEV, dtype=dt)
    dSTp = torch.full((K, B, ld), float("nan"), device=DEV, dtype=dt)
    out = torch.full((8,), float("nan"), device=DEV)
    ws = torch.empty(_hip.lib().cpc_nce_workspace_floats(B, K), device=DEV)
    _hip.call("cpc_nce_loss", _hip.ptr(dS_in), _hip.ptr(dSp), _hip.ptr(dSTp), _hip.ptr(out), _hip.ptr(ws), B, K, ld, softplus,
              C.c_float(reg), code)
    assert (dSp[:, :, B:] == 0).all() and (dSTp[:, :, B:] == 0).all()
    dS, dST = dSp[:, :, :B], dSTp[:, :, :B]
    # oracle on the 4-D score tensor whose equal-step diagonal is S (other entries irrelevant in this branch)
    lin = S.double().requires_grad_(True)
    full = torch.zeros(B, K, B, K, dtype=torch.float64)
    for k in range(K):
        full[:, k, :, k] = lin[k]
    sc = F.softplus(full) if softplus else full
    loss, smax = O.info_nce_loss(sc, all_timesteps=False, regularization=reg)
    loss.backward()
    assert abs(out[0].item() - loss.item()) < 2e-5 * max(1.0, abs(loss.item()))
    assert abs(out[1].item() - smax.item()) < 1e-5 * max(1.0, abs(smax.item()))
    t = 2e-5 if dt == torch.float32 else 1e-2
    assert rel_err(dS, lin.grad) < t
    assert rel_err(dST, lin.grad.transpose(1, 2)) < t


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("softplus", [0, 1])
@pytest.mark.parametrize("B,K,reg", [(6, 4, 1.0), (10, 3, 0.01), (32, 12, 0.5)])
def test_nce_loss_all_timesteps(dt, softplus, B, K, reg):
    g = torch.Generator().manual_seed(B * 5 + K)
    R = B * K
    S = torch.randn(R, R, generator=g) * 3.0
    S[0, 0] = 25.0
    ld = (R + 7) // 8 * 8
    Sp = torch.full((R, ld), 7.0); Sp[:, :R] = S
    STp = torch.full((R, ld), 7.0); STp[:, :R] = S.T
    code = _hip.dtype_code(dt)
    dSp = torch.zeros((R, ld), device=DEV, dtype=dt)
    dSTp = torch.zeros((R, ld), device=DEV, dtype=dt)
    out = torch.full((8,), float("nan"), device=DEV)
    ws = torch.empty(_hip.lib().cpc_nce_all_workspace_floats(B, K), device=DEV)
    dSin, dSTin = dev(Sp), dev(STp)          # keep the device copies alive across the (asynchronous) launch
    _hip.call("cpc_nce_loss_all", _hip.ptr(dSin), _hip.ptr(dSTin), _hip.ptr(dSp), _hip.ptr(dSTp), _hip.ptr(out), _hip.ptr(ws),
              B, K, ld, softplus, C.c_float(reg), code)
    lin = S.double().requires_grad_(True)
    full = lin.view(B, K, B, K)
    sc = F.softplus(full) if softplus else full
    loss, smax = O.info_nce_loss(sc, all_timesteps=True, regularization=reg)
    loss.backward()
    assert abs(out[0].item() - loss.item()) < 2e-5 * max(1.0, abs(loss.item()))
    assert abs(out[1].item() - smax.item()) < 1e-5 * max(1.0, abs(smax.item()))
    t = 2e-5 if dt == torch.float32 else 1e-2
    assert rel_err(dSp[:, :R], lin.grad) < t
    assert rel_err(dSTp[:, :R], lin.grad.T) < t


def _fused_scores(Pm, Tm, softplus, reg, K, diag_off, n_rows_total, n_items_total, lse_in=None, want_T=True):
    """cpc_score_lse -> cpc_nce_lse_merge -> cpc_nce_fused_grad on device copies of Pm [M][E] / Tm [N][E] (bf16)."""
    M, E = Pm.shape
    N = Tm.shape[0]
    bf = torch.bfloat16
    dP, dT = dev(Pm, bf), dev(Tm, bf)
    Sb = torch.full((M, N), float("nan"), device=DEV)
    pm = torch.full((M // 256, N), float("nan"), device=DEV)
    ps = torch.full((M // 256, N), float("nan"), device=DEV)
    valid = torch.zeros(M, device=DEV)
    L_, F_ = C.c_longlong, C.c_float
    _hip.call("cpc_score_lse", _hip.ptr(dP), _hip.ptr(dT), _hip.ptr(Sb), _hip.ptr(pm), _hip.ptr(ps), _hip.ptr(valid), M, N, E, L_(E), L_(E), L_(N),
              diag_off)
    lse = torch.full((N,), float("nan"), device=DEV)
    colp = torch.full(((N + 255) // 256, 2), float("nan"), device=DEV)
    _hip.call("cpc_nce_lse_merge", _hip.ptr(pm), _hip.ptr(ps), M // 256, N, softplus, F_(n_rows_total), _hip.ptr(lse), _hip.ptr(colp))
    use = lse if lse_in is None else dev(lse_in.float())
    items = M // K
    dS = torch.full((M, N), float("nan"), device=DEV, dtype=bf)
    dST = torch.full((N, M), float("nan"), device=DEV, dtype=bf) if want_T else None
    gradp = torch.full((int(_hip.lib().cpc_nce_fused_grad_blocks(items, N)),), float("nan"), device=DEV)
    _hip.call("cpc_nce_fused_grad", _hip.ptr(Sb), _hip.ptr(use), _hip.ptr(dS), _hip.ptr(dST), _hip.ptr(gradp), items, K, N, L_(N), L_(M), diag_off,
              softplus, F_(reg), F_(n_rows_total), F_(n_items_total))
    torch.cuda.synchronize()
    return SimpleNamespace(Sb=Sb, pm=pm, ps=ps, valid=valid, lse=lse, colp=colp, dS=dS, dST=dST, gradp=gradp)


@pytest.mark.parametrize("softplus", [0, 1])
@pytest.mark.parametrize("items,K,E,reg", [(64, 8, 128, 1.0), (32, 16, 192, 0.01), (64, 12, 512, 0.5)])
def test_fused_all_timesteps_score_path(softplus, items, K, E, reg):
    """score_over_all_timesteps=True through the fused kernels (bf16): the score GEMM whose epilogue leaves the column log-sum-exp
    pairs, the merge, the gradient pass on the bf16 scores and the loss scalars — against the oracle's loss on the same (bf16-rounded)
    operands and autograd's gradient (contrastive_estimation_training.py:12-22, :108-114, :141)."""
    R = items * K
    g = torch.Generator().manual_seed(items + K + E)
    bf = torch.bfloat16
    Pm = (torch.randn(R, E, generator=g) * (2.0 / math.sqrt(E))).to(bf)
    Tm = (torch.randn(R, E, generator=g) * 2.0).to(bf)
    Pm[0] = Tm[0] * 0.25                                  # one large own-target score (softplus threshold branch, the column maximum)
    r = _fused_scores(Pm.float(), Tm.float(), softplus, reg, K, 0, R, items)
    lin = (Pm.double() @ Tm.double().T).requires_grad_(True)
    sc = lin.view(items, K, items, K)
    sc = F.softplus(sc) if softplus else sc
    loss, smax = O.info_nce_loss(sc, all_timesteps=True, regularization=reg)
    loss.backward()
    # the scores as stored (f32), their diagonal and the column log-sum-exps (taken from the f32 accumulators)
    assert rel_err(r.Sb, lin) < 2e-6
    assert (r.valid.double().cpu() - torch.diagonal(lin.detach())).abs().max().item() < 1e-4 * lin.detach().abs().max().item()
    ref_lse = torch.logsumexp(sc.detach().reshape(R, R), dim=0)
    assert (r.lse.double().cpu() - ref_lse).abs().max().item() < 2e-5 * max(1.0, ref_lse.abs().max().item())
    out = torch.full((8,), float("nan"), device=DEV)
    out[6] = 0.0
    F_ = C.c_float
    _hip.call("cpc_nce_fused_finalize", _hip.ptr(r.colp), r.colp.shape[0], _hip.ptr(r.valid), R, _hip.ptr(r.gradp), r.gradp.numel(), None, 0,
              F_(R), F_(items), K, F_(reg), softplus, _hip.ptr(out))
    assert abs(out[0].item() - loss.item()) < 2e-4 * max(1.0, abs(loss.item())), (out[0].item(), loss.item())
    assert abs(out[1].item() - smax.item()) < 1e-5 * max(1.0, abs(smax.item()))
    assert out[5].item() == 0.0 and out[6].item() == 0.0
    assert rel_err(r.dS, lin.grad) < 1.5e-2
    assert torch.equal(r.dST, r.dS.T.contiguous())      # the transposed copy is the same numbers
    # the two-stage form a rank of a data-parallel run uses: partial sums, then the scalars from the (all-reduced) sums
    sums = torch.zeros(4, device=DEV)
    out2 = torch.full((8,), float("nan"), device=DEV)
    out2[6] = 0.0
    _hip.call("cpc_nce_fused_finalize", _hip.ptr(r.colp), r.colp.shape[0], _hip.ptr(r.valid), R, _hip.ptr(r.gradp), r.gradp.numel(), _hip.ptr(sums), 1,
              F_(R), F_(items), K, F_(reg), softplus, None)
    _hip.call("cpc_nce_fused_finalize", None, 0, None, 0, None, 0, _hip.ptr(sums), 2, F_(R), F_(items), K, F_(reg), softplus, _hip.ptr(out2))
    assert torch.equal(out[:6], out2[:6])


@pytest.mark.parametrize("softplus", [0, 1])
def test_fused_score_strips_equal_the_whole_matrix(softplus):
    """The strips a rank of a global-negatives run forms — (all predictions) x (own targets) and (own predictions) x (all targets),
    engine.GlobalNegatives._all_timesteps_strips — give the same log-sum-exps, loss sums and gradient blocks as the whole score matrix."""
    items, K, E, W, reg = 64, 8, 128, 2, 0.7
    R = items * K
    Rl, il = R // W, items // W
    g = torch.Generator().manual_seed(99)
    bf = torch.bfloat16
    Pm = (torch.randn(R, E, generator=g) * (2.0 / math.sqrt(E))).to(bf).float()
    Tm = (torch.randn(R, E, generator=g) * 2.0).to(bf).float()
    whole = _fused_scores(Pm, Tm, softplus, reg, K, 0, R, items)
    tot = torch.zeros(4, device=DEV)
    for rank in range(W):
        lo = rank * Rl
        col = _fused_scores(Pm, Tm[lo:lo + Rl], softplus, reg, K, -lo, R, items)
        assert torch.equal(col.lse, whole.lse[lo:lo + Rl])
        assert torch.equal(col.dS, whole.dS[:, lo:lo + Rl]) and torch.equal(col.dST, whole.dST[lo:lo + Rl])
        assert torch.equal(col.valid[lo:lo + Rl], whole.valid[lo:lo + Rl])
        row = _fused_scores(Pm[lo:lo + Rl], Tm, softplus, reg, K, lo, R, items, lse_in=whole.lse, want_T=False)
        assert torch.equal(row.dS, whole.dS[lo:lo + Rl])
        sums = torch.zeros(4, device=DEV)
        F_ = C.c_float
        _hip.call("cpc_nce_fused_finalize", _hip.ptr(col.colp), col.colp.shape[0], _hip.ptr(col.valid, lo), Rl, _hip.ptr(col.gradp),
                  col.gradp.numel(), _hip.ptr(sums), 1, F_(R), F_(items), K, F_(reg), softplus, None)
        tot[:3] += sums[:3]
        tot[3] = torch.maximum(tot[3], sums[3]) if rank else sums[3]
    ref = torch.zeros(4, device=DEV)
    _hip.call("cpc_nce_fused_finalize", _hip.ptr(whole.colp), whole.colp.shape[0], _hip.ptr(whole.valid), R, _hip.ptr(whole.gradp),
              whole.gradp.numel(), _hip.ptr(ref), 1, C.c_float(R), C.c_float(items), K, C.c_float(reg), softplus, None)
    assert torch.allclose(tot, ref, rtol=2e-6, atol=0.0), (tot, ref)


def test_fused_score_path_refuses_what_it_cannot_tile():
    a = torch.zeros(512, 128, device=DEV, dtype=torch.bfloat16)
    pm = torch.zeros(2, 512, device=DEV)
    with pytest.raises(_hip.HipCallError):          # 300 rows: not a whole number of 256-row tiles
        _hip.call("cpc_score_lse", _hip.ptr(a), _hip.ptr(a), None, _hip.ptr(pm), _hip.ptr(pm), None, 300, 512, 128, C.c_longlong(128),
                  C.c_longlong(128), C.c_longlong(512), 0)
    with pytest.raises(_hip.HipCallError):          # odd K
        _hip.call("cpc_nce_fused_grad", _hip.ptr(pm), _hip.ptr(pm), _hip.ptr(a), None, None, 64, 7, 128, C.c_longlong(128), C.c_longlong(0), 0, 0,
                  C.c_float(1.0), C.c_float(448.0), C.c_float(64.0))


# --------------------------------------------------------------------------------------- Adam
def test_adam_matches_torch():
    g = torch.Generator().manual_seed(0)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    p = dev(p0.clone())
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        ref_p.grad = grad.clone()
        opt.step()
        dg = dev(grad * 2.0)
        _hip.call("cpc_adam", _hip.ptr(p), _hip.ptr(dg), _hip.ptr(m), _hip.ptr(v), C.c_longlong(n), C.c_float(1e-3),
                  C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), step, C.c_float(0.5), None)
        assert (p.cpu() - ref_p.detach()).abs().max().item() < 2e-6
    # skip flag (the NaN guard): raised -> parameters and both moments keep their bits; lowered -> the update happens
    flag = torch.ones(1, device=DEV)
    before = [t.clone() for t in (p, m, v)]
    state = torch.zeros(4, device=DEV)
    for name, extra in (("cpc_adam", (4, C.c_float(1.0))), ("cpc_adam_dev", (_hip.ptr(state), C.c_float(1.0)))):
        _hip.call(name, _hip.ptr(p), _hip.ptr(dg), _hip.ptr(m), _hip.ptr(v), C.c_longlong(n), C.c_float(1e-3),
                  C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), *extra, _hip.ptr(flag))
        assert all(torch.equal(a, b) for a, b in zip(before, (p, m, v))), name
    assert torch.equal(state, torch.zeros(4, device=DEV))          # the device-side step count did not advance either
    flag.zero_()
    _hip.call("cpc_adam", _hip.ptr(p), _hip.ptr(dg), _hip.ptr(m), _hip.ptr(v), C.c_longlong(n), C.c_float(1e-3),
              C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), 4, C.c_float(1.0), _hip.ptr(flag))
    assert not torch.equal(before[0], p)


def test_unsupported_shapes_fail_loudly():
    a = torch.zeros(64, 12, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_hip.HipCallError):
        _hip.gemm_nt(_hip.ptr(a), _hip.ptr(a), _hip.ptr(a), 64, 64, 12, 12, 12, 64, _hip.BF16)   # K % 8 != 0


# --------------------------------------------------------------------------------------- attention context kernels
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,S,C,heads", [(3, 60, 64, 8), (2, 64, 128, 2), (5, 7, 32, 4), (3, 60, 512, 8), (2, 37, 128, 2), (2, 5, 64, 1)])
def test_attention_fwd_bwd(dt, B, S, C, heads):
    """cpc_attn_fwd / cpc_attn_bwd vs autograd of the definition (causal softmax(q k^T / sqrt(d)) v per head)."""
    g = torch.Generator().manual_seed(S * 3 + C)
    d = C // heads
    qkv = rounded(torch.randn(B * S, 3 * C, generator=g), dt).requires_grad_(True)
    dout = rounded(torch.randn(B * S, C, generator=g), dt)
    q, k, v = (t.reshape(B, S, heads, d).permute(0, 2, 1, 3) for t in qkv.split(C, dim=1))
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(d) + torch.triu(torch.full((S, S), float("-inf"), dtype=torch.double), 1)
    P = torch.softmax(sc, -1)
    out = (P @ v).permute(0, 2, 1, 3).reshape(B * S, C)
    out.backward(dout)
    code = _hip.dtype_code(dt)
    d_qkv, d_dout = dev(qkv.detach().float(), dt), dev(dout.float(), dt)
    o = torch.full((B * S, C), float("nan"), device=DEV, dtype=dt)
    Pd = torch.full((B * heads, S, S), float("nan"), device=DEV, dtype=dt)
    _hip.call("cpc_attn_fwd", _hip.ptr(d_qkv), _hip.ptr(o), _hip.ptr(Pd), B, S, C, heads, 0.0, 0, 0, code)
    assert rel_err(o, out) < tol(dt)
    assert rel_err(Pd, P.reshape(B * heads, S, S)) < tol(dt)
    dq = torch.full((B * S, 3 * C), float("nan"), device=DEV, dtype=dt)
    _hip.call("cpc_attn_bwd", _hip.ptr(d_qkv), _hip.ptr(Pd), _hip.ptr(d_dout), _hip.ptr(dq), B, S, C, heads, 0.0, 0, 0, code)
    assert rel_err(dq, qkv.grad) < (1e-4 if dt == torch.float32 else 2.5e-2)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,S,C,heads", [(3, 60, 512, 8), (2, 37, 128, 2), (3, 60, 64, 8)])
def test_attention_fwd_bwd_with_dropout(dt, B, S, C, heads):
    """The same with dropout on the attention weights (nn.MultiheadAttention's, p = 0.3): the device masks are a function of
    (seed, site, index) and are materialised with cpc_dropout_mask for the reference.  Head size 64 in bf16 runs on the matrix-pipe
    kernels, the other cases on the vector kernels."""
    g = torch.Generator().manual_seed(S * 5 + C)
    d, p_drop, seed, site = C // heads, 0.3, 987654321, 4
    qkv = rounded(torch.randn(B * S, 3 * C, generator=g), dt).requires_grad_(True)
    dout = rounded(torch.randn(B * S, C, generator=g), dt)
    mask = torch.empty(B * heads * S * S, device=DEV)
    _hip.call("cpc_dropout_mask", _hip.ptr(mask), mask.numel(), p_drop, seed, site)
    m = mask.cpu().double().reshape(B, heads, S, S)
    q, k, v = (t.reshape(B, S, heads, d).permute(0, 2, 1, 3) for t in qkv.split(C, dim=1))
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(d) + torch.triu(torch.full((S, S), float("-inf"), dtype=torch.double), 1)
    P = torch.softmax(sc, -1)
    out = ((P * m) @ v).permute(0, 2, 1, 3).reshape(B * S, C)
    out.backward(dout)
    code = _hip.dtype_code(dt)
    d_qkv, d_dout = dev(qkv.detach().float(), dt), dev(dout.float(), dt)
    o = torch.full((B * S, C), float("nan"), device=DEV, dtype=dt)
    Pd = torch.full((B * heads, S, S), float("nan"), device=DEV, dtype=dt)
    _hip.call("cpc_attn_fwd", _hip.ptr(d_qkv), _hip.ptr(o), _hip.ptr(Pd), B, S, C, heads, p_drop, seed, site, code)
    assert rel_err(o, out) < tol(dt)
    assert rel_err(Pd, P.reshape(B * heads, S, S)) < tol(dt)          # the saved weights stay undropped
    dq = torch.full((B * S, 3 * C), float("nan"), device=DEV, dtype=dt)
    _hip.call("cpc_attn_bwd", _hip.ptr(d_qkv), _hip.ptr(Pd), _hip.ptr(d_dout), _hip.ptr(dq), B, S, C, heads, p_drop, seed, site, code)
    assert rel_err(dq, qkv.grad) < (1e-4 if dt == torch.float32 else 2.5e-2)


def test_attention_unsupported_shapes():
    x = torch.zeros(16, device=DEV)
    p = _hip.ptr(x)
    for B, S, C, heads in ((1, 65, 64, 8), (1, 8, 256, 2), (1, 8, 60, 8)):
        with pytest.raises(_hip.HipCallError):
            _hip.call("cpc_attn_fwd", p, p, p, B, S, C, heads, 0.0, 0, 0, 0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,C,bcast", [(180, 64, 0), (37, 512, 0), (120, 96, 60)])
def test_add_layernorm_fwd_bwd(dt, M, C, bcast):
    """cpc_add_ln_fwd / cpc_ln_bwd vs autograd of F.layer_norm(a + b); with bcast the incoming gradient is a per-item row
    scaled by 1/bcast (the mean over time folded into the final norm's backward)."""
    g = torch.Generator().manual_seed(M + C)
    a = rounded(torch.randn(M, C, generator=g), dt).requires_grad_(True)
    b = rounded(torch.randn(M, C, generator=g) * 0.5, dt)
    w = (1 + 0.3 * torch.randn(C, generator=g)).double().requires_grad_(True)
    bias = (0.2 * torch.randn(C, generator=g)).double().requires_grad_(True)
    code = _hip.dtype_code(dt)
    y_ref = F.layer_norm(a + b, (C,), w, bias, 1e-5)
    if bcast:
        gm = rounded(torch.randn(M // bcast, C, generator=g), dt)
        dy = gm.repeat_interleave(bcast, 0) / bcast
        g1, g2 = gm, None
    else:
        g1 = rounded(torch.randn(M, C, generator=g), dt)
        g2 = rounded(torch.randn(M, C, generator=g), dt)
        dy = g1 + g2
    y_ref.backward(dy)
    da, db, dw, dbias = dev(a.detach().float(), dt), dev(b.float(), dt), dev(w.detach().float()), dev(bias.detach().float())
    r = torch.full((M, C), float("nan"), device=DEV, dtype=dt)
    y = torch.full((M, C), float("nan"), device=DEV, dtype=dt)
    stats = torch.full((M, 2), float("nan"), device=DEV)
    _hip.call("cpc_add_ln_fwd", _hip.ptr(da), _hip.ptr(db), _hip.ptr(dw), _hip.ptr(dbias), _hip.ptr(r), _hip.ptr(y), _hip.ptr(stats),
              M, C, 1e-5, 0.0, 0, 0, code)
    assert rel_err(y, y_ref) < tol(dt)
    assert rel_err(r, a.detach() + b) < tol(dt)
    nb = 5
    slabs = torch.full((nb, 2, C), float("nan"), device=DEV)
    dr = torch.full((M, C), float("nan"), device=DEV, dtype=dt)
    dg1, dg2 = dev(g1.float(), dt), (dev(g2.float(), dt) if g2 is not None else None)
    _hip.call("cpc_ln_bwd", _hip.ptr(dg1), _hip.ptr(dg2), _hip.ptr(r), _hip.ptr(stats), _hip.ptr(dw), _hip.ptr(dr), _hip.ptr(slabs),
              M, C, bcast, (1.0 / bcast) if bcast else 1.0, nb, None, 0.0, 0, 0, code)
    t = 1e-4 if dt == torch.float32 else 2.5e-2
    assert rel_err(dr, a.grad) < t
    assert rel_err(slabs.sum(0)[0], w.grad) < t
    assert rel_err(slabs.sum(0)[1], bias.grad) < t


@pytest.mark.parametrize("B,V,E,H", [(3, 20, 64, 256), (5, 7, 32, 48), (2, 100, 16, 96)])
def test_gru_gradient_penalty_kernels(B, V, E, H):
    """cpc_gru_gp_fwd / cpc_gru_gp_bwd (f32) at hidden sizes up to the kernels' limit: the gradient of the directional derivative
    D = <xt, d S / d x>, S = <wc, GRU(x)>, with respect to the GRU's parameters and its input, assembled from the kernels' outputs
    as engine.GRUContext.gp_grads does, against autograd's double backward in float64."""
    g = torch.Generator().manual_seed(B * 1000 + H)
    f64 = lambda *sh, s=1.0: (torch.randn(*sh, generator=g) * s).float().double()
    Wih, Whh = f64(3 * H, E, s=E ** -0.5), f64(3 * H, H, s=H ** -0.5)
    bih, bhh = f64(3 * H, s=0.3), f64(3 * H, s=0.3)
    x, xt, wc = f64(B, V, E), f64(B, V, E), f64(B, H)
    params = [t.clone().requires_grad_(True) for t in (Wih, Whh, bih, bhh, x)]
    pWih, pWhh, pbih, pbhh, px = params
    h = torch.zeros(B, H, dtype=torch.double)
    for t in range(V):
        gi, gh = px[:, t] @ pWih.T + pbih, h @ pWhh.T + pbhh
        r, z = torch.sigmoid(gi[:, :H] + gh[:, :H]), torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
    gx, = torch.autograd.grad((wc * h).sum(), px, create_graph=True)
    ref = torch.autograd.grad((gx * xt).sum(), params)
    # device: the projections as the engine's GEMMs give them, then the two kernels
    Gi = dev((x.reshape(B * V, E) @ Wih.T + bih).float().contiguous())
    GiT = dev((xt.reshape(B * V, E) @ Wih.T).float().contiguous())
    WT, W, d_bhh, d_wc = dev(Whh.T.float().contiguous()), dev(Whh.float().contiguous()), dev(bhh.float()), dev(wc.float().contiguous())
    tape = torch.full((B, V, 10, H), float("nan"), device=DEV)
    ct = torch.full((B, H), float("nan"), device=DEV)
    _hip.call("cpc_gru_gp_fwd", _hip.ptr(Gi), _hip.ptr(GiT), _hip.ptr(WT), _hip.ptr(d_bhh), _hip.ptr(tape), _hip.ptr(ct), B, V, H)
    dA = torch.full((B, V, 8, H), float("nan"), device=DEV)
    _hip.call("cpc_gru_gp_bwd", _hip.ptr(d_wc), _hip.ptr(tape), _hip.ptr(W), _hip.ptr(dA), B, V, H)
    assert torch.isfinite(tape).all() and torch.isfinite(dA).all() and torch.isfinite(ct).all()
    tp, da = tape.double().cpu(), dA.double().cpu().reshape(B * V, 8 * H)
    hp, htp = tp[:, :, 4].reshape(B * V, H), tp[:, :, 9].reshape(B * V, H)
    X, XT = x.reshape(B * V, E), xt.reshape(B * V, E)
    d3, v3 = da[:, :3 * H], da[:, 4 * H:7 * H]
    dh, vh = torch.cat([da[:, :2 * H], da[:, 3 * H:4 * H]], 1), torch.cat([da[:, 4 * H:6 * H], da[:, 7 * H:]], 1)
    got = [d3.T @ XT + v3.T @ X, dh.T @ htp + vh.T @ hp, v3.sum(0), vh.sum(0), (v3 @ Wih).reshape(B, V, E)]
    for name, a, b in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh", "x"), got, ref):
        l2 = ((a - b).norm() / (b.norm() + 1e-30)).item()
        assert l2 < 2e-4, (name, l2)
    # the tangent of the last hidden state: directional derivative of the GRU output along xt (central difference in float64)
    def gru64(xx):
        hh = torch.zeros(B, H, dtype=torch.double)
        for t in range(V):
            gi, gh = xx[:, t] @ Wih.T + bih, hh @ Whh.T + bhh
            r, z = torch.sigmoid(gi[:, :H] + gh[:, :H]), torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            hh = (1 - z) * torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]) + z * hh
        return hh
    eps = 1e-6
    fd = (gru64(x + eps * xt) - gru64(x - eps * xt)) / (2 * eps)
    assert rel_err(ct, fd) < 1e-4


@pytest.mark.parametrize("p_drop", [0.0, 0.3])
@pytest.mark.parametrize("B,S,C,heads", [(3, 10, 64, 8), (2, 64, 128, 2), (2, 7, 32, 4)])
def test_attention_gradient_penalty_kernels(B, S, C, heads, p_drop):
    """cpc_attn_tangent / cpc_attn_gp (f32) against the float64 formulas that tools/gp_attention_algebra.py checks against
    autograd's double backward; the dropout factors of the attention weights are materialised with cpc_dropout_mask."""
    g = torch.Generator().manual_seed(S + C + heads)
    d, M, seed, site = C // heads, B * S, 1234567, 5
    f64 = lambda *sh: torch.randn(*sh, generator=g).float().double()
    qkv, qkvt, dO, lam = f64(M, 3 * C), f64(M, 3 * C), f64(M, C), f64(M, 3 * C)
    hd = lambda t: t.reshape(B, S, heads, d).permute(0, 2, 1, 3)
    un = lambda t: t.permute(0, 2, 1, 3).reshape(M, C)
    q, k, v = (hd(t) for t in qkv.split(C, 1))
    qt, kt, vt = (hd(t) for t in qkvt.split(C, 1))
    causal = torch.tril(torch.ones(S, S)).bool()
    sc = 1.0 / math.sqrt(d)
    P = torch.softmax(((q @ k.transpose(-1, -2)) * sc).masked_fill(~causal, float("-inf")), -1)
    mask = torch.ones(B * heads * S * S, device=DEV)
    if p_drop:
        _hip.call("cpc_dropout_mask", _hip.ptr(mask), mask.numel(), p_drop, seed, site)
    m = mask.cpu().double().reshape(B, heads, S, S)
    u = ((qt @ k.transpose(-1, -2) + q @ kt.transpose(-1, -2)) * sc).masked_fill(~causal, 0.0)
    mrow = (P * u).sum(-1, keepdim=True)
    pt = P * (u - mrow)
    out_t = un((pt * m) @ v + (P * m) @ vt)
    a = m * (hd(dO) @ v.transpose(-1, -2))
    cc = (P * a).sum(-1, keepdim=True)
    dS = P * (a - cc)
    w = m * (hd(dO) @ vt.transpose(-1, -2)) + (a - cc) * (u - mrow)
    sig = P * (w - (P * w).sum(-1, keepdim=True))
    src = torch.cat([un(sc * (sig @ k + dS @ kt)), un(sc * (sig.transpose(-1, -2) @ q + dS.transpose(-1, -2) @ qt)),
                     un((pt * m).transpose(-1, -2) @ hd(dO))], 1)
    d_qkv, d_qkvt, d_dO, d_P = dev(qkv.float()), dev(qkvt.float()), dev(dO.float()), dev(P.reshape(B * heads, S, S).float())
    o = torch.full((M, C), float("nan"), device=DEV)
    _hip.call("cpc_attn_tangent", _hip.ptr(d_qkv), _hip.ptr(d_qkvt), _hip.ptr(d_P), _hip.ptr(o), B, S, C, heads, p_drop, seed, site)
    assert rel_err(o, out_t) < 1e-5
    acc = dev(lam.float())
    _hip.call("cpc_attn_gp", _hip.ptr(d_qkv), _hip.ptr(d_qkvt), _hip.ptr(d_P), _hip.ptr(d_dO), _hip.ptr(acc), B, S, C, heads, p_drop,
              seed, site)
    assert rel_err(acc.double().cpu() - lam, src) < 1e-4


@pytest.mark.parametrize("p_drop", [0.0, 0.3])
@pytest.mark.parametrize("M,C,bcast", [(180, 64, 0), (37, 512, 0), (120, 96, 60), (24, 1024, 0)])
def test_layernorm_gradient_penalty_kernels(M, C, bcast, p_drop):
    """cpc_ln_tangent / cpc_ln_gp (f32) against the float64 formulas of tools/gp_attention_algebra.py (tangent of
    LayerNorm(a + dropout(b)); second-order term on the input and the penalty part of the weight gradient)."""
    g = torch.Generator().manual_seed(M + C)
    seed, site = 424242, 3
    f64 = lambda *sh: torch.randn(*sh, generator=g).float().double()
    a, b, at, bt, lam = f64(M, C), f64(M, C) * 0.5, f64(M, C), f64(M, C), f64(M, C)
    w, bias = (1 + 0.3 * f64(C)), 0.2 * f64(C)
    mask = torch.ones(M * C, device=DEV)
    if p_drop:
        _hip.call("cpc_dropout_mask", _hip.ptr(mask), mask.numel(), p_drop, seed, site)
    m = mask.cpu().double().reshape(M, C)
    da, db, dw, dbias = dev(a.float()), dev(b.float()), dev(w.float()), dev(bias.float())
    r = torch.empty(M, C, device=DEV)
    y = torch.empty(M, C, device=DEV)
    stats = torch.empty(M, 2, device=DEV)
    _hip.call("cpc_add_ln_fwd", _hip.ptr(da), _hip.ptr(db), _hip.ptr(dw), _hip.ptr(dbias), _hip.ptr(r), _hip.ptr(y), _hip.ptr(stats),
              M, C, 1e-5, p_drop, seed, site, _hip.F32)
    rr = a + b * m
    assert rel_err(r, rr) < 1e-6
    mu = rr.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((rr - mu) ** 2).mean(-1, keepdim=True) + 1e-5)
    xh = (rr - mu) * rstd
    proj = lambda t: t - t.mean(-1, keepdim=True) - xh * (xh * t).mean(-1, keepdim=True)
    rt = at + bt * m
    yt_ref = w * rstd * proj(rt)
    rt_out = torch.full((M, C), float("nan"), device=DEV)
    yt = torch.full((M, C), float("nan"), device=DEV)
    d_at, d_bt = dev(at.float()), dev(bt.float())
    _hip.call("cpc_ln_tangent", _hip.ptr(d_at), _hip.ptr(d_bt), _hip.ptr(r), _hip.ptr(stats), _hip.ptr(dw),
              _hip.ptr(rt_out), _hip.ptr(yt), M, C, p_drop, seed, site)
    assert rel_err(rt_out, rt) < 1e-6 and rel_err(yt, yt_ref) < 1e-5
    if bcast:
        gm = f64(M // bcast, C)
        dy, g1, g2, gscale = gm.repeat_interleave(bcast, 0) / bcast, gm, None, 1.0 / bcast
    else:
        g1, g2, gscale = f64(M, C), f64(M, C), 1.0
        dy = g1 + g2
    p = dy * w
    pr, pp = proj(rt), proj(p)
    aa, bb = (p * xh).mean(-1, keepdim=True), (xh * rt).mean(-1, keepdim=True)
    src = -(rstd ** 2) * (xh * (p * pr).mean(-1, keepdim=True) + bb * pp + aa * pr)
    gw = (dy * rstd * pr).sum(0)
    nb = 5
    slabs = torch.full((nb, C), float("nan"), device=DEV)
    acc, acc_b = dev(lam.float()), dev(lam.float())
    d_g1, d_g2 = dev(g1.float()), (dev(g2.float()) if g2 is not None else None)
    _hip.call("cpc_ln_gp", _hip.ptr(d_g1), _hip.ptr(d_g2), _hip.ptr(rt_out), _hip.ptr(r),
              _hip.ptr(stats), _hip.ptr(dw), _hip.ptr(acc), _hip.ptr(acc_b), _hip.ptr(slabs), M, C, bcast, gscale, nb, p_drop, seed, site)
    assert rel_err(acc.double().cpu() - lam, src) < 1e-4
    assert rel_err(acc_b.double().cpu() - lam, src * m) < 1e-4
    assert rel_err(slabs.sum(0), gw) < 1e-4


@pytest.mark.parametrize("dt", DTYPES)
def test_positional_scale_and_time_mean(dt):
    B, S, C, Ltop, t0 = 3, 11, 32, 20, 4
    g = torch.Generator().manual_seed(2)
    top = rounded(torch.randn(B, Ltop, C, generator=g), dt)
    pe = O.positional_encoding(S, C)
    code = _hip.dtype_code(dt)
    d_top = dev(top.float(), dt)
    x0 = torch.full((B * S, C), float("nan"), device=DEV, dtype=dt)
    scale = math.sqrt(C)
    _hip.call("cpc_pe_scale_fwd", _hip.ptr(d_top, t0 * C), _hip.ptr(dev(pe)), _hip.ptr(x0), B, S, C, Ltop * C, scale, code)
    ref = (top[:, t0:t0 + S] * scale + pe.double()).reshape(B * S, C)
    assert rel_err(x0, ref) < tol(dt)
    g1 = rounded(torch.randn(B * S, C, generator=g), dt)
    g2 = rounded(torch.randn(B * S, C, generator=g), dt)
    dtop = torch.zeros(B, Ltop, C, device=DEV, dtype=dt)
    d1, d2 = dev(g1.float(), dt), dev(g2.float(), dt)
    _hip.call("cpc_pe_scale_bwd", _hip.ptr(d1), _hip.ptr(d2), _hip.ptr(dtop, t0 * C), B, S, C, Ltop * C, scale, code)
    assert rel_err(dtop[:, t0:t0 + S], ((g1 + g2) * scale).reshape(B, S, C)) < tol(dt)
    assert dtop[:, :t0].abs().max().item() == 0 and dtop[:, t0 + S:].abs().max().item() == 0
    m = torch.full((B, C), float("nan"), device=DEV, dtype=dt)
    _hip.call("cpc_mean_time", _hip.ptr(x0), _hip.ptr(m), B, S, C, code)
    assert rel_err(m, x0.double().cpu().reshape(B, S, C).mean(1)) < tol(dt)


# --------------------------------------------------------------------------------------- seeded shape fuzzing
def _rand_shapes(seed, n):
    import random as _r
    rng = _r.Random(seed)
    for _ in range(n):
        yield rng


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_random_shapes(dt):
    """40 seeded random problems through cpc_gemm_nt: ragged M / N, K in and out of the fast path's granularity, item
    addressing on A and C, pad-row zeroing, bias / relu / mask epilogues, f32 or storage-dtype output."""
    import random as _r
    rng = _r.Random(1234 + (dt == torch.bfloat16))
    ch = 8 if dt == torch.bfloat16 else 4
    code = _hip.dtype_code(dt)
    for case in range(40):
        M = rng.choice([1, 7, 64, 130, 257, 513, 1030, 2300])
        N = rng.choice([4, 8, 36, 64, 128, 200, 260, 520]) // 4 * 4
        K = rng.choice([ch, 2 * ch, 64, 96, 128, 8 * ch * 3, 320])
        K = K // ch * ch
        use_items = rng.random() < 0.5 and M >= 8
        rpi = rng.choice([3, 5, 16, 40]) if use_items else 0
        g = torch.Generator().manual_seed(case)
        lda = K + rng.choice([0, ch, 4 * ch])
        A = torch.randn(M + 4, lda, generator=g)
        Bt = torch.randn(N, K, generator=g)
        bias = torch.randn(N, generator=g)
        mask = torch.randn(M, N, generator=g)
        relu, use_bias, use_mask = rng.random() < 0.4, rng.random() < 0.5, rng.random() < 0.4
        of32 = rng.random() < 0.3 and not use_mask
        ref = rounded(A[:M, :K], dt) @ rounded(Bt, dt).T
        if use_bias:
            ref = ref + bias.double()
        if relu:
            ref = torch.relu(ref)
        if use_mask:
            ref = torch.where(rounded(mask, dt) > 0, ref, torch.zeros_like(ref))
        c_valid = 0
        if rpi:
            c_valid = rng.randint(1, rpi)
            rows = torch.arange(M) % rpi
            ref = torch.where((rows < c_valid).unsqueeze(1), ref, torch.zeros_like(ref))
        dA, dB, db, dm = dev(A, dt), dev(Bt, dt), dev(bias), dev(mask, dt)
        items = -(-M // rpi) if rpi else 0
        item_rows = rpi + 2
        ldc = N + rng.choice([0, 8])
        if rpi:
            out = torch.full((items, item_rows, ldc), float("nan"), device=DEV, dtype=torch.float32 if of32 else dt)
            dmask = torch.zeros(items, item_rows, ldc, device=DEV, dtype=dt)
            idx = torch.arange(M, device=DEV)
            dmask[idx // rpi, idx % rpi, :N] = dm
        else:
            out = torch.full((M, ldc), float("nan"), device=DEV, dtype=torch.float32 if of32 else dt)
            dmask = torch.zeros(M, ldc, device=DEV, dtype=dt)
            dmask[:, :N] = dm
        _hip.gemm_nt(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(out), M, N, K, lda, K, ldc, code, bias=_hip.ptr(db) if use_bias else None,
                     mask=_hip.ptr(dmask) if use_mask else None, c_rpi=rpi, c_item=item_rows * ldc if rpi else 0, c_valid=c_valid,
                     flags=(_hip.GEMM_RELU if relu else 0) | (_hip.GEMM_OUT_F32 if of32 else 0))
        if rpi:
            idx = torch.arange(M, device=DEV)
            got = out[idx // rpi, idx % rpi, :N]
        else:
            got = out[:, :N]
        scale = ref.abs().max().item() + 1e-9
        err = (got.double().cpu() - ref).abs().max().item() / scale
        assert err < tol(dt) * 2, (case, M, N, K, rpi, c_valid, relu, use_bias, use_mask, of32, err)


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_tn_random_shapes(dt):
    """30 seeded random problems through cpc_gemm_tn: ragged I (padded A rows), J multiples of the chunk, split reductions,
    item addressing on B."""
    import random as _r
    rng = _r.Random(99 + (dt == torch.bfloat16))
    ch = 8 if dt == torch.bfloat16 else 4
    code = _hip.dtype_code(dt)
    for case in range(30):
        M = rng.choice([5, 64, 100, 333, 1024, 2500, 4100])
        I = rng.choice([8, 20, 24, 72, 128, 264, 520])
        J = rng.choice([ch, 32, 64, 136, 256, 264]) // ch * ch
        nsplit = rng.choice([1, 1, 2, 3, 7])
        g = torch.Generator().manual_seed(1000 + case)
        lda = (I + ch - 1) // ch * ch + rng.choice([0, ch])
        A = torch.randn(M, lda, generator=g)
        rpi = rng.choice([0, 0, 9, 32])
        if rpi:
            items = -(-M // rpi)
            Bfull = torch.randn(items, rpi + 1, J, generator=g)
            Bm = Bfull[:, :rpi].reshape(-1, J)[:M]
        else:
            Bfull = torch.randn(M, J, generator=g)
            Bm = Bfull
        ref = rounded(A[:, :I], dt).T @ rounded(Bm, dt)
        dA, dB = dev(A, dt), dev(Bfull, dt)
        blk = 64 if dt == torch.bfloat16 else 32
        chunk = ((M + nsplit - 1) // nsplit + blk - 1) // blk * blk
        slabs = torch.full((nsplit, I, J), float("nan"), device=DEV)
        _hip.gemm_tn(_hip.ptr(dA), _hip.ptr(dB), _hip.ptr(slabs), M, I, J, lda, J, J, code, b_rpi=rpi, b_item=(rpi + 1) * J if rpi else 0,
                     nsplit=nsplit, m_chunk=chunk if nsplit > 1 else 0, slab_stride=I * J, flags=_hip.GEMM_OUT_F32)
        scale = ref.abs().max().item() + 1e-9
        err = (slabs.sum(0).double().cpu() - ref).abs().max().item() / scale
        assert err < tol(dt) * 2, (case, M, I, J, nsplit, rpi, err)
