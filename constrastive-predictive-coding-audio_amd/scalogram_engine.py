"""Train-step engine for the scalogram model family (BASELINE configs[2]): ScalogramResidualEncoder in front of the same
context network / predictor / InfoNCE head as the waveform model.

Activations are channels-last "grids" [B][W = time][H = frequency][C] (csrc/scalogram.hip): the reference's tall (k,1)
kernels (scalogram_model.py:390-395 with e.g. kernel_size_2 = (64,1)) become overlapped-row GEMMs over the frequency axis,
every other kernel shape runs as im2col + GEMM + col2im; BatchNorm2d uses batch statistics in train mode, running
statistics in eval mode.  The encoder's output (one frequency row left) IS the [B][frames][E] buffer the context networks
read, so everything downstream is shared with engine.CPCEngine.
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace
from typing import List, Optional

import torch

from . import _hip
from .engine import CPCEngine, Float32Context, _ceil_div, make_context, side_stream


class Grid:
    """Zero-initialised [B][W][Ha][C] buffer with guard rows on both ends; valid rows are [top, top + H) of every column."""

    def __init__(self, B, W, H, C_, device, dtype, top=0, tail=0, guard_rows=96, lazy=False):
        self.B, self.W, self.H, self.C, self.top = int(B), int(W), int(H), int(C_), int(top)
        self.Ha = self.top + self.H + int(tail)
        self.guard_rows = int(guard_rows)
        self.dtype = dtype
        self.code = _hip.dtype_code(dtype)
        self.rows = self.B * self.W * self.Ha
        self.device = device
        self.full = self.t = None
        if not lazy:          # (lazy: a grid that normally aliases the caller's memory gets its own only when allocate() is called)
            self.allocate()
        self.desc = (C.c_int * 6)(self.B, self.W, self.H, self.Ha, self.top, self.C)
        # the same memory seen as a grid whose top padding rows are ordinary (zero) data rows
        self.padded_desc = (C.c_int * 6)(self.B, self.W, self.top + self.H, self.Ha, 0, self.C)

    def allocate(self):
        guard = self.guard_rows * self.C
        self.full = torch.zeros(guard + self.rows * self.C + guard, device=self.device, dtype=self.dtype)
        self.t = self.full[guard:guard + self.rows * self.C]
        return self.t

    def ptr(self, offset_elems=0):
        return _hip.ptr(self.t, offset_elems)

    def like(self, device, dtype=None, guard_rows=96):
        dtype = self.dtype if dtype is None else dtype
        return Grid(self.B, self.W, self.H, self.C, device, dtype, top=self.top, tail=self.Ha - self.top - self.H, guard_rows=guard_rows)

    @property
    def count(self):
        return self.B * self.W * self.H


def _desc(t, desc):
    return C.cast(desc, C.c_void_p)


def _accumulate(dst: Grid, src: Grid):
    """dst += src over two grids of one geometry (two branches' data gradients meeting, scalogram_model.py:462-476 under autograd)."""
    n = dst.t.numel()
    if src.t.numel() != n or src.dtype != dst.dtype or n % 4:
        raise ValueError("accumulate: the two grids must share one geometry and storage type")
    _hip.call("cpc_accumulate", dst.ptr(), src.ptr(), C.c_longlong(n), dst.code)


def _twin(eng, grid: Grid) -> Grid:
    """The grid that holds the TANGENT of ``grid`` during a gradient-penalty step (same geometry, dtype and guards), made on
    first use.  Keyed by identity, so grids that alias in the primal graph alias in the tangent graph too."""
    tw = eng._twins.get(id(grid))
    if tw is None:
        tw = eng._twins[id(grid)] = grid.like(grid.t.device, guard_rows=grid.guard_rows)
    return tw


def _col_group(cout, kw=1, stride=(1, 1), pad=0):
    """How many consecutive output rows a tall-kernel GEMM computes per GEMM row: with fewer than 128 output channels the
    128-wide MFMA tile would be mostly empty, so G = 128 / C_out rows are produced side by side (their windows overlap in
    all but G-1 input rows; the weight operand holds G shifted copies of the kernel, +(G-1)/k extra FLOPs)."""
    width = int(os.environ.get("CPC_COL_WIDTH", "256"))      # 256 x 256 tiles of the large GEMM kernels (128: 20.5 -> 22.2 ms per configs[2] step)
    if kw != 1 or stride != (1, 1) or pad != 0 or cout >= width or width % cout:
        return 1
    return width // cout


def _col_ok(kw, sh, sw, pad, cin, in_f32):
    """Does a convolution run as an overlapped-row GEMM (mode 'col')?  (k,1) kernel, stride 1, no padding."""
    return kw == 1 and sh == 1 and sw == 1 and pad == 0 and cin % 8 == 0 and not in_f32


def _prepare_convs(eng, holder, convs):
    """prepare() of a list of convolutions (objects with the _Conv interface): the plain overlapped-row ones (mode 'col', one output row per
    GEMM row) share ONE cpc_conv_w_prep_batch launch — eleven launches of 20 us each for ar_conv_architecture_3 otherwise; the job table is
    planned once per parameter buffer (``holder`` keeps it)."""
    batch = [c for c in convs if isinstance(c, _Conv) and c.mode == 'col' and c.G == 1]
    if len(batch) < 2 or os.environ.get("CPC_PREP_BATCH", "1") == "0":
        batch = []
    for c in convs:
        if c not in batch:
            c.prepare()
    if not batch:
        return
    key = eng.model._flat_param.data_ptr()
    st = getattr(holder, "_prep_batch", None)
    if st is None or st[0] != key:
        p = eng.model._param
        st = holder._prep_batch = (key, _hip.ConvPrepBatch([(p[c.wname], c.w_fwd, c.w_dgrad, c.cout, c.cin, c.kh, 1) for c in batch], eng.device))
    st[1].run(eng.code)


class _Conv:
    """One nn.Conv2d of the encoder on grids.  mode 'col': (k,1) kernel, stride 1, no padding -> overlapped-row GEMMs;
    mode 'win': im2col + GEMM (+ col2im for the data gradient)."""

    def __init__(self, eng, wname, bname, mod, gin: Grid, in_f32=False, need_dgrad=True, relu=False, out_pad=None):
        """``out_pad`` = (top, tail, guard_rows): geometry of the output grid when the NEXT convolution reads it directly (a block
        without BatchNorm whose second kernel has top padding, scalogram_model.py:411-412); mode 'win' only."""
        self.eng, self.wname, self.bname, self.gin, self.in_f32, self.need_dgrad, self.relu = eng, wname, bname, gin, in_f32, need_dgrad, relu
        # a convolution that reads float32 input (the scalogram, or its pooled copy) runs entirely in float32: its output
        # is normalised / added downstream, and bf16 would quantise away the signal riding on the log-amplitude offset
        dev, dt = eng.device, (torch.float32 if in_f32 else eng.dt)
        self.dt, self.code = dt, _hip.dtype_code(dt)
        self.cin, self.cout = mod.in_channels, mod.out_channels
        self.kh, self.kw = mod.kernel_size
        self.sh, self.sw = mod.stride
        ph, pw = mod.padding
        if ph != pw:
            raise NotImplementedError("asymmetric Conv2d padding")
        self.pad = ph
        if self.cin != gin.C:
            raise ValueError(f"{wname}: expects {self.cin} input channels, the incoming activation has {gin.C}")
        hin = gin.top + gin.H
        self.Ho = (hin + 2 * self.pad - self.kh) // self.sh + 1
        self.Wo = (gin.W + 2 * self.pad - self.kw) // self.sw + 1
        if self.Ho < 1 or self.Wo < 1:
            raise ValueError(f"{wname}: input {hin} x {gin.W} is smaller than the kernel")
        self.mode = 'col' if _col_ok(self.kw, self.sh, self.sw, self.pad, self.cin, in_f32) else 'win'
        self.gather = self.gather_w = self.parity = False
        B = gin.B
        if self.cout % 8:
            raise NotImplementedError("scalogram encoder channel counts must be multiples of 8")
        if self.mode == 'col':
            if out_pad is not None and (out_pad[0] or out_pad[1]):
                raise AssertionError("an overlapped-row convolution writes its input's row geometry (use _CastRelu for other outputs)")
            self.y0 = Grid(B, gin.W, self.Ho, self.cout, dev, dt, top=0, tail=gin.Ha - self.Ho, guard_rows=self.kh + 16)
            self.K = self.kh * self.cin
            self.M = gin.rows
            G = _col_group(self.cout)
            self.G = G if (G > 1 and gin.Ha % G == 0 and dt == torch.bfloat16) else 1
            if self.G > 1:
                G = self.G
                self.Rw = _ceil_div((self.kh + G - 1) * self.cin, 64) * 64 // self.cin      # window rows incl. K padding to 64
                self.Rd = _ceil_div((self.kh + G - 1) * self.cout, 64) * 64 // self.cout
                if (self.Rw * self.cin) % 64 or (self.Rd * self.cout) % 64:
                    self.G = 1
            # Only the rows that exist are computed.  A valid convolution with a tall kernel has Ho = Ha - kh + 1 output rows per column
            # (127 of 190, 34 of 63, 2 of 16 in scalogram_resnet_architecture_7), and only the rows [top, top + H) of the input carry a
            # gradient: the GEMMs address "rows per column" (a_rpi / c_rpi) instead of running over every allocated row of the grid
            # (CPC_COL_VALID=0: all rows, the round-2 form -- 33 ... 47 % more GEMM work on the tall kernels).
            self.valid_rows = os.environ.get("CPC_COL_VALID", "1") != "0"
            self.ncol = B * gin.W
            Gq = self.G
            self.Hg = _ceil_div(self.Ho, Gq)                                   # output (super-)rows per column
            self.r0 = gin.top // Gq                                            # first input (super-)row that carries a gradient
            self.nr = _ceil_div(gin.top + gin.H, Gq) - self.r0                 # ... and how many
            if self.G > 1:
                G = self.G
                self.w_fwd = torch.zeros(G, self.cout, self.Rw, self.cin, device=dev, dtype=dt)       # [(dh,co)][(r,c)]
                self.w_dgrad = torch.zeros(G, self.cin, self.Rd, self.cout, device=dev, dtype=dt)     # [(dr,c)][(q,co)]
                self.bias_g = torch.zeros(G * self.cout, device=dev, dtype=torch.float32)
                self.Mw = self.ncol * self.Hg if self.valid_rows else self.M // G          # rows of the weight-gradient reduction
                self.nsplit = eng._pick_split(self.Rw * self.cin, G * self.cout, self.Mw)
                self.slab = self.nsplit * self.Rw * self.cin * G * self.cout
            else:
                self.w_fwd = torch.empty(self.cout * self.K, device=dev, dtype=dt)
                self.w_dgrad = torch.empty(self.cin * self.kh * self.cout, device=dev, dtype=dt)
                self.Mw = self.ncol * self.Hg if self.valid_rows else self.M
                self.nsplit = eng._pick_split(self.K, self.cout, self.Mw)
                self.slab = self.nsplit * self.K * self.cout
            rb = int(os.environ.get(f"CPC_DGRAD_BAND_K{self.kh}", os.environ.get("CPC_DGRAD_BAND", "1")))      # (per kernel height for A/B runs)
            self.bands = self._bands(rb) if (self.valid_rows and need_dgrad) else None
        else:
            o_top, o_tail, o_guard = out_pad if out_pad is not None else (0, 0, 96)
            # Data gradient of a 3x3 stride-2 convolution WITHOUT the im2col-gradient matrix (_dgrad_parity below; CPC_DGRAD_PARITY=0: off): needs
            # one zero row below every output column (the row "H_out" that the last input rows' windows reach)
            self.parity = (need_dgrad and (self.kh, self.kw, self.sh, self.sw, self.pad) == (3, 3, 2, 2, 0) and not in_f32 and
                           os.environ.get("CPC_DGRAD_PARITY", "1") != "0" and (2 * self.cout) % (64 if dt == torch.bfloat16 else 32) == 0 and
                           self.cin % 8 == 0 and gin.Ha >= 2 * (self.Ho + 1) and gin.W >= 2 * self.Wo + 1)
            if self.parity and out_pad is None:
                o_tail = 1
            elif self.parity and o_tail < 1:
                self.parity = False
            self.y0 = Grid(B, self.Wo, self.Ho, self.cout, dev, dt, top=o_top, tail=o_tail, guard_rows=o_guard)
            if self.parity:
                self._parity_setup(B, dev, dt)
            self.K = self.kh * self.kw * self.cin
            # K padded so that the fast GEMM paths apply (K-stage of 64 bf16 / 32 f32 elements); tiny K stays at a multiple of 8
            kq = 64 if dt == torch.bfloat16 else 32
            self.Kp = _ceil_div(self.K, kq) * kq if self.K > kq // 2 else _ceil_div(self.K, 8) * 8
            self.M = B * self.Wo * self.Ho
            # Forward and weight gradient without an im2col matrix (bf16; CPC_CONV_GATHER=0: off, =1: also the float32 forward): a GEMM row is the
            # window read straight from the grid as kw pieces — piece dw = the kh rows x C_in channels of kernel column dw, contiguous in the
            # channels-last grid, the next kernel column one grid column (Ha C_in elements) further (cpc_gemm_nt_args.k_taps / k_tap_stride_a;
            # each piece padded to the stage size with zero weights, which over-reads into the rows below: zeros or activations, never beyond
            # the grid's guard), one launch with batch = B; the weight gradient is a TN GEMM whose reduction rows are the same windows
            # (cpc_gemm_tn_args.a_rpi2, one batch entry per kernel column).  12.67 -> 12.38 ms per configs[2] step.  The forward half alone
            # (the im2col pass kept for the weight gradient, on the side stream) had measured 13.78 against 13.62.
            bkq = 64 if dt == torch.bfloat16 else 32
            self.seg = _ceil_div(self.kh * self.cin, bkq) * bkq
            self.gather = (self.pad == 0 and not in_f32 and self.cin % 8 == 0 and (os.environ.get("CPC_CONV_GATHER", "") == "1" or (dt == torch.bfloat16 and os.environ.get("CPC_CONV_GATHER", "1") != "0")) and
                           gin.guard_rows * self.cin >= self.seg)
            if self.gather:
                self.w_imp = torch.zeros(self.cout, self.kw, self.seg, device=dev, dtype=dt)
            self.gather_w = self.gather and dt == torch.bfloat16          # (the two-level TN kernel is the bf16 LDS-DMA one)
            # (neither matrix exists where the forward and the weight gradient read the grid directly and the data gradient takes the parity route)
            self.col = torch.empty(self.M * self.Kp, device=dev, dtype=dt) if not self.gather_w else None
            self.dcol = torch.empty(self.M * self.Kp, device=dev, dtype=dt) if (need_dgrad and not self.parity) else None
            self.w_fwd = torch.zeros(self.cout, self.Kp, device=dev, dtype=dt)
            self.w_t = torch.zeros(self.Kp, self.cout, device=dev, dtype=dt)
            self.nsplit = eng._pick_split(self.Kp, self.cout, self.M, dt)
            if self.Kp * self.cout <= 8192:
                # a tiny weight matrix reduced over millions of rows (first convolution: 18 x 32 over 5 M positions): the
                # reduction, not the tile, has to fill the chip
                self.nsplit = max(self.nsplit, min(2048, self.M // 2048))
            self.slab = self.nsplit * self.Kp * self.cout
        self.dy0: Optional[Grid] = None        # set by the owner (may alias another gradient grid)
        # the train-mode BatchNorm behind this convolution, if any: its input gradient has zero mean per channel BY CONSTRUCTION
        # (dx = gamma rstd (g - <g> - xhat <g xhat>)), so the bias gradient is exactly zero -- the reference's autograd sums rounding
        # noise of relative size 1e-7 there.  The column-sum pass over the gradient grid is skipped and zero is written.
        self.bn_after = None

    def _parity_setup(self, B, dev, dt):
        """Data gradient of a 3x3 stride-2 convolution as two overlapped-row GEMMs on the output-gradient grid (no im2col-gradient matrix, no
        col2im pass).  With R = a // 2 over the allocated input rows a and dY rows outside [0, H_out) zero:
            dX[w][2R]   = sum_dw  dY[wo][R] W[0][dw] + dY[wo][R-1] W[2][dw],      dX[w][2R+1] = sum_dw dY[wo][R] W[1][dw],     w = 2 wo + dw
        so an EVEN input column w = 2 wo' takes dY columns wo'-1 (dw = 2) and wo' (dw = 0), an ODD one w = 2 wo' + 1 only column wo' (dw = 1).
        Per input column and super-row R the 2 C_in values (rows 2R, 2R+1) are one GEMM row: A = the dY rows (R-1, R) of one column (2 C_out
        contiguous elements, overlapped rows) — for even columns two such pieces, one grid column apart (cpc_gemm_nt_args.k_taps).  The first
        and the last even column have only one piece inside their clip: the rows are ordered (column, clip, R) through the second addressing
        level, so that a tile lies in one column, and k_ranges cuts the missing piece out (a piece outside the clip belongs to the
        neighbouring clip).  Rows per (column, clip) are padded to a whole number of tiles per column where B * (H_out + 1) is not one."""
        cin, cout, Ho, Wo = self.cin, self.cout, self.Ho, self.Wo
        bk = 64 if dt == torch.bfloat16 else 32
        Hs = Ho + 1
        # G row pairs per GEMM row where 2 C_in columns would leave most of a 128-wide tile empty (C_in = 32: G = 2, the window is then the
        # three dY rows R-1 .. R+1 per piece): half the tiles, each full — 0.52 -> 0.3x ms for the even columns of block 1 of architecture 7
        G = 1
        width = int(os.environ.get("CPC_DGRAD_PARITY_GROUP", "128"))          # 0: no grouping; 256: up to the 256-wide tile
        while width and 2 * cin * G * 2 <= width and Hs % (2 * G) == 0:
            G *= 2
        Hg = Hs // G
        tile = 256 if 2 * cin * G >= 256 else 128
        Hp = Hg
        while (B * Hp) % tile and Hp < Hg + 16:
            Hp += 1
        if (B * Hp) % tile or Hp * 4 > Hg * 5 or ((G + 1) * cout) % bk:
            self.parity = False                         # small batches of odd heights: the im2col route
            return
        self.Hs, self.Hp, self.Hg, self.Gp = Hs, Hp, Hg, G
        spp = (G + 1) * cout // bk
        r = [0, 2 * spp] * (Wo + 1)
        r[0], r[1] = spp, 2 * spp                      # column 0: only dY column 0 (piece 1)
        r[2 * Wo], r[2 * Wo + 1] = 0, spp              # column 2 Wo: only dY column Wo - 1 (piece 0)
        self.par_ranges = torch.tensor(r, dtype=torch.int32, device=dev)
        self.w_even = torch.zeros(2 * G, cin, 2, G + 1, cout, device=dev, dtype=dt)      # [(g, r, c)][(piece, q, co)]
        self.w_odd = torch.zeros(2 * G, cin, G + 1, cout, device=dev, dtype=dt)          # [(g, r, c)][(q, co)]
        n, kp = 2.0 * B * Hg * 2 * G * cin, (G + 1) * cout
        self.par_work = (n * ((Wo - 1) * 2 * kp + 2 * kp), n * Wo * kp)

    def _parity_prepare(self, w4):
        wt = w4.permute(2, 3, 1, 0)                    # [dh][dw][c][co]
        for dst, dws in ((self.w_even, (2, 0)), (self.w_odd, (1,))):
            for pi, dw in enumerate(dws):
                d = dst[:, :, pi] if dst is self.w_even else dst
                for g in range(self.Gp):               # row pair R = G R' + g reads the window rows q = g (dY[R-1]) and g + 1 (dY[R])
                    d[2 * g, :, g, :].copy_(wt[2, dw])             # row 2R   <- dY[R-1] W[2]
                    d[2 * g, :, g + 1, :].copy_(wt[0, dw])         # row 2R   <- dY[R]   W[0]
                    d[2 * g + 1, :, g + 1, :].copy_(wt[1, dw])     # row 2R+1 <- dY[R]   W[1]

    def _dgrad_parity(self, dy0: Grid, din: Grid, mask_input):
        gin, cin, cout, Wo, Hg, Hp, G, code = self.gin, self.cin, self.cout, self.Wo, self.Hg, self.Hp, self.Gp, self.code
        B, P = gin.B, _hip.ptr
        if dy0.Ha - dy0.top - self.Ho < 1 or din.Ha != gin.Ha or dy0.W != Wo or dy0.guard_rows < G * (Hp - Hg) + 2:
            raise AssertionError("parity data gradient: the output-gradient grid has no zero row below its columns")
        col_o, col_i = dy0.Ha * cout, din.Ha * cin
        a0 = (dy0.top - 1) * cout                      # row R - 1 of R = 0
        N, kp = 2 * G * cin, (G + 1) * cout
        skip = _hip.GEMM_SKIP_PAD_ROWS if Hp > Hg else 0
        mk = (lambda off: gin.ptr(off)) if mask_input else (lambda off: None)
        # even input columns 2 wo', wo' = 0 .. Wo: pieces = dY columns wo' - 1 and wo'
        _hip.gemm_nt(dy0.ptr(a0 - col_o), P(self.w_even), din.ptr(), (Wo + 1) * B * Hp, N, 2 * kp, G * cout, 2 * kp, N, code,
                     mask=mk(0), a_rpi=Hp, a_item=Wo * col_o, a_rpi2=B, a_item2=col_o, c_rpi=Hp, c_item=gin.W * col_i, c_valid=Hg, c_rpi2=B,
                     c_item2=2 * col_i, k_taps=2, k_tap_stride=kp, k_tap_stride_a=col_o, k_ranges=P(self.par_ranges),
                     flags=skip | _hip.GEMM_KRANGE_EXACT,          # the cut-out piece is the neighbouring clip's data, not zeros: no tile may straddle two columns
                     work=self.par_work[0])
        # odd input columns 2 wo' + 1, wo' = 0 .. Wo - 1: dY column wo'
        _hip.gemm_nt(dy0.ptr(a0), P(self.w_odd), din.ptr(col_i), B * Wo * Hp, N, kp, G * cout, kp, N, code, mask=mk(col_i),
                     a_rpi=Hp, a_item=col_o, a_rpi2=Wo, a_item2=Wo * col_o, c_rpi=Hp, c_item=2 * col_i, c_valid=Hg, c_rpi2=Wo,
                     c_item2=gin.W * col_i, flags=skip | _hip.GEMM_LINEAR_K, work=self.par_work[1])
        if gin.W > 2 * Wo + 1:                          # input columns no window reaches: zero (the residual branch adds into this grid)
            din.t.view(B, gin.W, col_i)[:, 2 * Wo + 1:, :].zero_()

    def _bands(self, RB):
        """Data gradient of a tall kernel in bands of RB (super-)rows: the GEMM rows are ordered (band, column, row within the band), so
        that a 256-row tile lies in one band (or two), and every band runs only the part of its window of output-gradient rows that
        lies inside the column -- input row a receives W[j] dY[a - j] for 0 <= a - j < Ho only, the rest of the window is the zero
        rows above and below (cpc_gemm_nt_args.k_ranges; (30,1) kernel on 63 rows: 53 % of the MACs of the full windows remain).
        Returns (first super-row, number of bands, RB, ranges tensor, FLOPs) or None (RB = 0, or the geometry does not fit)."""
        gin, G, kh = self.gin, self.G, self.kh
        if RB <= 0:
            return None
        rows_alloc = gin.Ha // G
        nrb = _ceil_div(self.nr, RB) * RB
        bk = 64 if self.dt == torch.bfloat16 else 32
        Kd = (self.Rd if G > 1 else kh) * self.cout
        if nrb > rows_alloc or Kd % bk:
            return None
        r0b = min(self.r0, rows_alloc - nrb)
        nst, ranges, stages = Kd // bk, [], 0
        for i in range(nrb // RB):
            first, last = r0b + i * RB, r0b + (i + 1) * RB - 1
            lo, hi = max(0, (kh - 1) - G * last), min(kh + G - 1, self.Ho + kh - 1 - G * first)      # window rows [lo, hi) of the band
            lo_s, hi_s = (lo * self.cout // bk, min(nst, _ceil_div(hi * self.cout, bk))) if hi > lo else (0, 1)
            ranges += [lo_s, hi_s]
            stages += hi_s - lo_s
        if stages * 10 > 9 * nst * (nrb // RB) * self.nr // nrb:           # less than 10 % to gain: the plain column order reuses L2 better
            return None
        t = torch.tensor(ranges, dtype=torch.int32, device=self.eng.device)
        return r0b, nrb // RB, RB, t, 2.0 * self.ncol * RB * stages * bk * G * self.cin

    # ------------------------------------------------------------------
    def prepare(self):
        p, code = self.eng.model._param, self.code
        w = p[self.wname]
        if self.mode == 'col' and self.G > 1:
            # forward: output row G R + dh uses window rows dh .. dh+kh-1;  data gradient: input row G R + dr receives tap j from
            # dY window row q = dr + kh-1 - j   (one launch; 3 G small torch copies before)
            _hip.call("cpc_conv_w_prep_group", _hip.ptr(w), _hip.ptr(p[self.bname]) if self.bname else None, _hip.ptr(self.w_fwd),
                      _hip.ptr(self.w_dgrad), _hip.ptr(self.bias_g) if self.bname else None, self.cout, self.cin, self.kh, self.G, self.Rw,
                      self.Rd, code)
        elif self.mode == 'col':
            _hip.call("cpc_conv_w_prep", _hip.ptr(w), _hip.ptr(self.w_fwd), _hip.ptr(self.w_dgrad), self.cout, self.cin, self.kh, 1, code)
        else:
            w4 = w.detach().view(self.cout, self.cin, self.kh, self.kw)
            flat = w4.permute(0, 2, 3, 1).reshape(self.cout, self.K)                  # [co][(dh*kw + dw)*C + c]
            if self.gather:
                self.w_imp[:, :, :self.kh * self.cin].copy_(w4.permute(0, 3, 2, 1).reshape(self.cout, self.kw, self.kh * self.cin))   # [co][dw][(dh, c)]
            else:
                self.w_fwd[:, :self.K].copy_(flat)
            if self.parity:
                self._parity_prepare(w4)
            else:
                self.w_t[:self.K, :].copy_(flat.t())

    def forward(self, tangent=False):
        """tangent=True (gradient-penalty step): the same GEMM on the TANGENT of the input, written to the tangent of the output —
        no bias, and where the convolution carries the block's ReLU the PRIMAL output is the mask (a tangent passes where the
        activation was positive)."""
        e, gin, y0 = self.eng, self.gin, self.y0
        p, code = e.model._param, self.code
        bias = _hip.ptr(p.get(self.bname)) if self.bname else None
        flags = _hip.GEMM_RELU if self.relu else 0
        col, mask, mask_w = getattr(self, "col", None), None, None
        if tangent:
            if self.mode == 'win' and not self.gather:
                if getattr(self, "col_t", None) is None:
                    self.col_t = torch.empty_like(self.col)     # the primal im2col matrix is still needed by the weight gradient
                col = self.col_t
            if self.relu:
                mask, mask_w = y0.ptr(), y0.ptr(y0.top * self.cout)
            gin, y0, bias, flags = _twin(e, gin), _twin(e, y0), None, 0
        if self.mode == 'col' and self.G > 1:
            G = self.G
            Kg, Ng = self.Rw * self.cin, G * self.cout
            Hg = _ceil_div(self.Ho, G)
            bias_g = (_hip.ptr(self.bias_g) if self.bname else None) if not tangent else None
            if self.valid_rows:
                _hip.gemm_nt(gin.ptr(), _hip.ptr(self.w_fwd), y0.ptr(), self.ncol * Hg, Ng, Kg, G * self.cin, Kg, Ng, code, bias=bias_g, mask=mask,
                             a_rpi=Hg, a_item=gin.Ha * self.cin, c_rpi=Hg, c_item=y0.Ha * self.cout, c_valid=Hg, flags=flags)
            else:
                _hip.gemm_nt(gin.ptr(), _hip.ptr(self.w_fwd), y0.ptr(), self.M // G, Ng, Kg, G * self.cin, Kg, Ng, code, bias=bias_g, mask=mask,
                             c_rpi=gin.Ha // G, c_item=y0.Ha * self.cout, c_valid=Hg, flags=flags)
            if Hg * G > self.Ho:          # rows of the last super-row beyond the valid output
                y0.t.view(-1, y0.Ha, self.cout)[:, self.Ho:Hg * G, :] = 0
        elif self.mode == 'col' and self.valid_rows:
            _hip.gemm_nt(gin.ptr(), _hip.ptr(self.w_fwd), y0.ptr(), self.ncol * self.Ho, self.cout, self.K, self.cin, self.K, self.cout, code,
                         bias=bias, mask=mask, a_rpi=self.Ho, a_item=gin.Ha * self.cin, c_rpi=self.Ho, c_item=y0.Ha * self.cout,
                         c_valid=self.Ho, flags=flags)
        elif self.mode == 'col':
            _hip.gemm_nt(gin.ptr(), _hip.ptr(self.w_fwd), y0.ptr(), self.M, self.cout, self.K, self.cin, self.K, self.cout, code,
                         bias=bias, mask=mask, c_rpi=gin.Ha, c_item=y0.Ha * self.cout, c_valid=self.Ho, flags=flags)
        elif self.gather:
            Kg = self.kw * self.seg
            taps = dict(k_taps=self.kw, k_tap_stride=self.seg, k_tap_stride_a=gin.Ha * self.cin) if self.kw > 1 else {}
            # rows (clip, output column, output row) in ONE launch through the second row level (a launch per clip, batch = B, left the last
            # blocks' tiles nearly empty: 76 rows per clip in block 3 of architecture 7)
            _hip.gemm_nt(gin.ptr(), _hip.ptr(self.w_imp), y0.ptr(y0.top * self.cout), gin.B * self.Wo * self.Ho, self.cout, Kg, self.sh * self.cin, Kg,
                         self.cout, code, bias=bias, mask=mask_w, a_rpi=self.Ho, a_item=self.sw * gin.Ha * self.cin, a_rpi2=self.Wo,
                         a_item2=gin.W * gin.Ha * self.cin, c_rpi=self.Ho, c_item=y0.Ha * self.cout, c_valid=self.Ho, flags=flags, **taps)
        else:
            _hip.call("cpc_im2col2d", gin.ptr(), _hip.ptr(col), _desc(gin, gin.padded_desc), self.kh, self.kw, self.sh, self.sw,
                      self.pad, self.pad, self.Ho, self.Wo, self.Kp, 1 if self.in_f32 else 0, code)
            _hip.gemm_nt(_hip.ptr(col), _hip.ptr(self.w_fwd), y0.ptr(y0.top * self.cout), self.M, self.cout, self.Kp, self.Kp, self.Kp,
                         self.cout, code, bias=bias, mask=mask_w, c_rpi=self.Ho, c_item=y0.Ha * self.cout, c_valid=self.Ho, flags=flags)

    def _scratch(self):
        e = self.eng
        if getattr(self, "_wslab", None) is None:
            self._wslab = torch.empty(max(self.slab, 1), device=e.device, dtype=torch.float32)
            self._bscratch = torch.empty(e.colsum_blocks * self.cout, device=e.device, dtype=torch.float32)
            self._ev = (torch.cuda.Event(), torch.cuda.Event())
        return self._wslab

    def _wgrad(self, gin: Grid, col, dy0: Grid, gw, gb):
        """gw (the weight's gradient view, reference layout) = correlation of the input ``gin`` (or its im2col matrix ``col``)
        with the output gradient ``dy0``; gb (or None) = its column sums.  The short reductions (bias column sum, slab sums) run on
        the engine's side stream beside the next GEMMs of the main stream; the slabs live in a buffer of this convolution."""
        e, code = self.eng, self.code
        wslab = self._scratch()
        if gb is not None:
            with e.side(self._ev[0]):
                e._colsum_to_grad(dy0.ptr(), gb, dy0.rows, self.cout, code, scratch=self._bscratch)
        # The weight-gradient GEMM reads the convolution's input and its output gradient, both final at this point, and nothing the
        # main stream does next depends on it: it goes to the side stream with its slab reduction, beside the data-gradient GEMM and
        # the HBM-bound BatchNorm / pooling passes of the layer below (the main queue was busy 18.4 of 18.5 ms per configs[2] step
        # with the side queue idle for 16.8 of them).  CPC_WGRAD_STREAM=0 / a gradient-penalty step: GEMM on the main stream.
        side_gemm = e.use_aux and getattr(e, "_gp_phase", 0) == 0 and os.environ.get("CPC_WGRAD_STREAM", "1") != "0"

        def staged(gemm, reduce):
            if side_gemm:
                with e.side(self._ev[1]):
                    gemm()
                    reduce()
            else:
                gemm()
                with e.side(self._ev[1]):
                    reduce()

        if self.mode == 'col' and self.G > 1:
            G = self.G
            Kg, Ng, Mg = self.Rw * self.cin, G * self.cout, self.Mw
            chunk = e._chunk(Mg, self.nsplit)
            rows = dict(a_rpi=self.Hg, a_item=gin.Ha * self.cin, b_rpi=self.Hg, b_item=dy0.Ha * self.cout) if self.valid_rows else {}

            def reduce():
                # slab[(r,c)][(dh,co)] = sum_R X[G R + r][c] dY[G R + dh][co]  ->  dW[co][c][j] = sum_dh slab[(j+dh, c)][(dh, co)]
                # (one kernel: split sum, the G diagonals and the transposition into the reference layout; it was a torch sum + G - 1 adds + a copy)
                L = C.c_longlong
                _hip.call("cpc_reduce_conv_w2d", _hip.ptr(wslab), _hip.ptr(gw), self.nsplit, L(Kg * Ng), self.cout, self.cin, self.kh, 1, L(0),
                          L(self.cin * G * self.cout), L(G * self.cout), G, L(self.cout))

            staged(lambda: _hip.gemm_tn(gin.ptr(), dy0.ptr(), _hip.ptr(wslab), Mg, Kg, Ng, G * self.cin, Ng, Ng, code, nsplit=self.nsplit,
                                        m_chunk=chunk, slab_stride=Kg * Ng, flags=_hip.GEMM_OUT_F32, **rows), reduce)
        elif self.mode == 'col':
            chunk = e._chunk(self.Mw, self.nsplit)
            rows = dict(a_rpi=self.Ho, a_item=gin.Ha * self.cin, b_rpi=self.Ho, b_item=dy0.Ha * self.cout) if self.valid_rows else {}
            staged(lambda: _hip.gemm_tn(gin.ptr(), dy0.ptr(), _hip.ptr(wslab), self.Mw, self.K, self.cout, self.cin, self.cout, self.cout, code,
                                        nsplit=self.nsplit, m_chunk=chunk, slab_stride=self.K * self.cout, flags=_hip.GEMM_OUT_F32, **rows),
                   lambda: _hip.call("cpc_reduce_conv_w", _hip.ptr(wslab), _hip.ptr(gw), self.cin, self.cout, self.kh, self.nsplit,
                                     self.K * self.cout))
        else:
            chunk = e._chunk(self.M, self.nsplit, self.dt)
            taps = self.kh * self.kw

            if self.gather_w:
                # the forward pass wrote no im2col matrix, and none is needed here: the rows of the reduction are the windows read straight
                # from the grid (clip, output column, output row: cpc_gemm_tn_args.a_rpi2), one batch entry per kernel column
                I, kc = self.kh * self.cin, self.kh * self.cin * self.cout

                def gemm_g():
                    _hip.gemm_tn(gin.ptr(), dy0.ptr(dy0.top * self.cout), _hip.ptr(wslab), self.M, I, self.cout, self.sh * self.cin, self.cout,
                                 self.cout, code, a_rpi=self.Ho, a_item=self.sw * gin.Ha * self.cin, a_rpi2=self.Wo,
                                 a_item2=gin.W * gin.Ha * self.cin, a_batch=gin.Ha * self.cin, b_rpi=self.Ho, b_item=dy0.Ha * self.cout,
                                 c_batch=kc, batch=self.kw, nsplit=self.nsplit, m_chunk=chunk, slab_stride=self.kw * kc, flags=_hip.GEMM_OUT_F32)

                def reduce_g():          # slabs [z][dw][dh][c][co] -> dW[co][c][dh][dw]
                    L = C.c_longlong
                    _hip.call("cpc_reduce_conv_w2d", _hip.ptr(wslab), _hip.ptr(gw), self.nsplit, L(self.kw * kc), self.cout, self.cin, self.kh, self.kw,
                              L(kc), L(self.cin * self.cout), L(self.cout), 1, L(0))
                staged(gemm_g, reduce_g)
                return

            def gemm():
                if self.gather:      # (float32: the forward pass wrote no im2col matrix; build it here, from the grid this gradient is taken against)
                    _hip.call("cpc_im2col2d", gin.ptr(), _hip.ptr(self.col), _desc(gin, gin.padded_desc), self.kh, self.kw, self.sh, self.sw,
                              self.pad, self.pad, self.Ho, self.Wo, self.Kp, 0, code)
                _hip.gemm_tn(_hip.ptr(self.col if self.gather else col), dy0.ptr(dy0.top * self.cout), _hip.ptr(wslab), self.M, self.Kp, self.cout, self.Kp,
                             self.cout, self.cout, code, b_rpi=self.Ho, b_item=dy0.Ha * self.cout, nsplit=self.nsplit,
                             m_chunk=chunk, slab_stride=self.Kp * self.cout, flags=_hip.GEMM_OUT_F32)
            staged(gemm,
                   lambda: _hip.call("cpc_reduce_slabs", _hip.ptr(wslab), _hip.ptr(gw), self.K, self.cout, self.nsplit, self.Kp * self.cout,
                                     self.cin, self.cin * taps, 1, taps))

    def gp_wgrad(self, gp_grad):
        """Penalty part of the weight gradient: (tangent of the input) x (gradient of the summed scores w.r.t. the output, which
        the first backward pass of the step left in dy0); nothing for the bias, which the tangent does not see."""
        e = self.eng
        self._wgrad(_twin(e, self.gin), getattr(self, "col_t", None), self.dy0, gp_grad[self.wname], None)

    def _dgrad_bands(self, dy0: Grid, dst: Grid, Kd, mask_input):
        G, gin = self.G, self.gin
        r0b, nb, RB, ranges, flops = self.bands
        _hip.gemm_nt(dy0.ptr((r0b * G - (self.kh - 1)) * self.cout), _hip.ptr(self.w_dgrad), dst.ptr(r0b * G * self.cin), nb * self.ncol * RB,
                     G * self.cin, Kd, G * self.cout, Kd, G * self.cin, self.code, mask=gin.ptr(r0b * G * self.cin) if mask_input else None,
                     a_rpi=RB, a_item=dy0.Ha * self.cout, a_rpi2=self.ncol, a_item2=RB * G * self.cout,
                     c_rpi=RB, c_item=dst.Ha * self.cin, c_valid=RB, c_rpi2=self.ncol, c_item2=RB * G * self.cin,
                     k_ranges=_hip.ptr(ranges), work=flops)

    def backward(self, din: Optional[Grid], accumulate=False, mask_input=False):
        """dy0 (gradient of the convolution output) -> bias / weight gradients, and the input gradient into ``din``
        (``mask_input``: multiplied by gin > 0, the ReLU that produced the input)."""
        e, gin, dy0 = self.eng, self.gin, self.dy0
        g, code = e.model._grad, self.code
        gb = g[self.bname] if (self.bname and self.bname in g) else None
        if gb is not None and self.bn_after is not None and self.bn_after.trained and os.environ.get("CPC_BN_BIAS_COLSUM", "0") != "1":
            gb.zero_()
            gb = None
        self._wgrad(gin, getattr(self, "col", None), dy0, g[self.wname], gb)
        if din is None or not self.need_dgrad:
            return
        if self.mode == 'col' and self.G > 1:
            G = self.G
            Mg = self.M // G
            dst = din
            if accumulate:
                if getattr(self, "_din_tmp", None) is None:
                    self._din_tmp = din.like(e.device, e.dt)
                dst = self._din_tmp
            Kd = self.Rd * self.cout
            if self.bands is not None:
                self._dgrad_bands(dy0, dst, Kd, mask_input)
            elif self.valid_rows:
                r0, nr = self.r0, self.nr
                _hip.gemm_nt(dy0.ptr((r0 * G - (self.kh - 1)) * self.cout), _hip.ptr(self.w_dgrad), dst.ptr(r0 * G * self.cin), self.ncol * nr,
                             G * self.cin, Kd, G * self.cout, Kd, G * self.cin, code, mask=gin.ptr(r0 * G * self.cin) if mask_input else None,
                             a_rpi=nr, a_item=dy0.Ha * self.cout, c_rpi=nr, c_item=dst.Ha * self.cin, c_valid=nr)
            else:
                _hip.gemm_nt(dy0.ptr(-(self.kh - 1) * self.cout), _hip.ptr(self.w_dgrad), dst.ptr(), Mg, G * self.cin, Kd, G * self.cout,
                             Kd, G * self.cin, code, mask=gin.ptr() if mask_input else None)
            if accumulate:
                _accumulate(din, dst)
        elif self.mode == 'col':
            D = self.kh
            dst = din
            if accumulate:          # the overlapped-row GEMM overwrites: go through a scratch grid of the same layout
                if getattr(self, "_din_tmp", None) is None:
                    self._din_tmp = din.like(e.device, e.dt)
                dst = self._din_tmp
            if self.bands is not None:
                self._dgrad_bands(dy0, dst, D * self.cout, mask_input)
            elif self.valid_rows:
                r0, nr = self.r0, self.nr
                _hip.gemm_nt(dy0.ptr((r0 - (D - 1)) * self.cout), _hip.ptr(self.w_dgrad), dst.ptr(r0 * self.cin), self.ncol * nr, self.cin,
                             D * self.cout, self.cout, D * self.cout, self.cin, code, mask=gin.ptr(r0 * self.cin) if mask_input else None,
                             a_rpi=nr, a_item=dy0.Ha * self.cout, c_rpi=nr, c_item=dst.Ha * self.cin, c_valid=nr)
            else:
                _hip.gemm_nt(dy0.ptr(-(D - 1) * self.cout), _hip.ptr(self.w_dgrad), dst.ptr(), self.M, self.cin, D * self.cout, self.cout,
                             D * self.cout, self.cin, code, mask=gin.ptr() if mask_input else None)
            if accumulate:
                _accumulate(din, dst)
        elif self.parity and not accumulate:
            self._dgrad_parity(dy0, din, mask_input)
        else:
            if self.parity:
                raise NotImplementedError("accumulating data gradient of a parity-route convolution")
            # (an output-bound launch: with its weight-gradient GEMM beside it the pair moves 2.25 GB in 0.75 ms for block 1 of
            # scalogram_resnet_architecture_7; 128-wide tiles or the LDS-staged epilogue change nothing)
            _hip.gemm_nt(dy0.ptr(dy0.top * self.cout), _hip.ptr(self.w_t), _hip.ptr(self.dcol), self.M, self.Kp, self.cout, self.cout,
                         self.cout, self.Kp, code, a_rpi=self.Ho, a_item=dy0.Ha * self.cout)
            _hip.call("cpc_col2im2d", _hip.ptr(self.dcol), din.ptr(), _desc(din, din.padded_desc), self.kh, self.kw, self.sh, self.sw,
                      self.pad, self.pad, self.Ho, self.Wo, self.Kp, 1 if accumulate else 0, code)
            if mask_input:
                _hip.call("cpc_relu_mask", din.ptr(), gin.ptr(), din.rows * din.C, code)


class _SepConv:
    """Conv2dSeparable on grids (scalogram_model.py:532-544): the depthwise k x k convolution runs on the im2col matrix of the
    input (cpc_im2col2d + cpc_dw_fwd; backward cpc_dw_bwd_col + cpc_col2im2d and cpc_dw_bwd_w), the 1 x 1 convolution is an
    ordinary _Conv on the depthwise output.  Same interface as _Conv."""

    def __init__(self, eng, prefix, mod, gin: Grid, in_f32=False, need_dgrad=True, relu=False, bias=True, out_pad=None):
        self.eng, self.gin, self.in_f32, self.need_dgrad = eng, gin, in_f32, need_dgrad
        dev, dt = eng.device, (torch.float32 if in_f32 else eng.dt)
        self.dt, self.code = dt, _hip.dtype_code(dt)
        dw = mod.conv
        self.wname = prefix + ".conv.weight"
        self.C = dw.in_channels
        self.kh_dw, self.kw = dw.kernel_size
        self.sh, self.sw = dw.stride
        ph, pw = dw.padding
        if ph != pw:
            raise NotImplementedError("asymmetric Conv2d padding")
        self.pad = ph
        if self.C != gin.C:
            raise ValueError(f"{self.wname}: expects {self.C} input channels, the incoming activation has {gin.C}")
        hin = gin.top + gin.H
        self.Ho = (hin + 2 * self.pad - self.kh_dw) // self.sh + 1
        self.Wo = (gin.W + 2 * self.pad - self.kw) // self.sw + 1
        if self.Ho < 1 or self.Wo < 1:
            raise ValueError(f"{self.wname}: input {hin} x {gin.W} is smaller than the kernel")
        self.taps = self.kh_dw * self.kw
        self.K = self.taps * self.C
        self.Kp = _ceil_div(self.K, 8) * 8
        self.M = gin.B * self.Wo * self.Ho
        self.col = torch.empty(self.M * self.Kp, device=dev, dtype=dt)
        self.dcol = torch.empty(self.M * self.Kp, device=dev, dtype=dt) if need_dgrad else None
        self.mid = Grid(gin.B, self.Wo, self.Ho, self.C, dev, dt)
        self.d_mid = self.mid.like(dev)
        self.pw = _Conv(eng, prefix + ".conv_1x1.weight", prefix + ".conv_1x1.bias" if bias else None, mod.conv_1x1, self.mid,
                        in_f32=in_f32, need_dgrad=True, relu=relu)
        if out_pad is not None and (out_pad[0] or out_pad[1]):
            raise AssertionError("a separable convolution writes a plain grid (use _CastRelu for padded outputs)")
        self.y0, self.cout, self.kh = self.pw.y0, self.pw.cout, 1
        self.nb = max(1, min(256, self.M // 256))
        self.slab = max(self.pw.slab, self.nb * self.C * self.taps)

    @property
    def dy0(self):
        return self.pw.dy0

    @dy0.setter
    def dy0(self, g):
        self.pw.dy0 = g

    def prepare(self):
        self.pw.prepare()

    def forward(self):
        e, gin, mid = self.eng, self.gin, self.mid
        w = e.model._param[self.wname]
        _hip.call("cpc_im2col2d", gin.ptr(), _hip.ptr(self.col), _desc(gin, gin.padded_desc), self.kh_dw, self.kw, self.sh, self.sw,
                  self.pad, self.pad, self.Ho, self.Wo, self.Kp, 1 if self.in_f32 else 0, self.code)
        _hip.call("cpc_dw_fwd", _hip.ptr(self.col), _hip.ptr(w), mid.ptr(mid.top * self.C), C.c_longlong(self.M), self.C, self.taps,
                  self.Kp, self.Ho, C.c_longlong(mid.Ha * self.C), self.code)
        self.pw.forward()

    def backward(self, din: Optional[Grid], accumulate=False, mask_input=False):
        e, gin, d_mid = self.eng, self.gin, self.d_mid
        w = e.model._param[self.wname]
        self.pw.backward(d_mid)
        off = d_mid.top * self.C
        _hip.call("cpc_dw_bwd_w", _hip.ptr(self.col), d_mid.ptr(off), _hip.ptr(e.slabs), C.c_longlong(self.M), self.C, self.taps, self.Kp,
                  self.Ho, C.c_longlong(d_mid.Ha * self.C), self.nb, self.code)
        e.model._grad[self.wname].view(-1).copy_(e.slabs[:self.nb * self.C * self.taps].view(self.nb, -1).sum(0))
        if din is not None and self.need_dgrad:
            _hip.call("cpc_dw_bwd_col", d_mid.ptr(off), _hip.ptr(w), _hip.ptr(self.dcol), C.c_longlong(self.M), self.C, self.taps, self.Kp,
                      self.Ho, C.c_longlong(d_mid.Ha * self.C), self.code)
            _hip.call("cpc_col2im2d", _hip.ptr(self.dcol), din.ptr(), _desc(din, din.padded_desc), self.kh_dw, self.kw, self.sh, self.sw,
                      self.pad, self.pad, self.Ho, self.Wo, self.Kp, 1 if accumulate else 0, self.code)
            if mask_input:
                _hip.call("cpc_relu_mask", din.ptr(), gin.ptr(), din.rows * din.C, self.code)


def _make_conv(eng, base, mod, gin, bias, **kw):
    """_Conv for an nn.Conv2d at state_dict prefix ``base``, _SepConv for a Conv2dSeparable."""
    if hasattr(mod, "conv_1x1"):
        return _SepConv(eng, base, mod, gin, bias=bias, **kw)
    return _Conv(eng, base + ".weight", base + ".bias" if bias else None, mod, gin, **kw)


class _BatchNorm:
    """nn.BatchNorm2d + ReLU between a convolution output grid y0 and the activation grid a."""

    def __init__(self, eng, prefix, mod, y0: Grid, a: Grid):
        self.eng, self.prefix, self.mod, self.y0, self.a = eng, prefix, mod, y0, a
        self.C = y0.C
        self.x_f32 = 1 if (y0.dtype == torch.float32 and eng.dt != torch.float32) else 0
        self.stats = torch.zeros(2, self.C, device=eng.device, dtype=torch.float32)
        # streaming reductions: enough workgroups in flight to reach the HBM rate (512 left the statistics pass at 2.4 TB/s)
        # (small grids too: the BatchNorm1d layers of ar_conv_architecture_3 have 14 336 rows of 512 channels -- with rows // 512
        # workgroups their backward reduction ran on 28 of the 256 CUs, 130 us per launch)
        self.nb = max(1, min(2048, y0.rows // 32))
        self.nb_bwd = max(1, min(2048, y0.rows // 32))
        self.slab = self.nb_bwd * 2 * self.C
        self.dy0: Optional[Grid] = None
        self.trained = True
        # the activation's ReLU mask as sign bits (one byte per 8 channels), written by the normalisation pass and read by the two
        # backward passes instead of the activation itself: 4 of their 14 bytes per element (bf16 grids; not in gradient-penalty
        # engines, whose passes call these kernels on other operands)
        self.abits = None
        if (eng.dt == torch.bfloat16 and not self.x_f32 and self.C % 8 == 0 and not getattr(eng, "gp_capable", False) and
                os.environ.get("CPC_BN_BITS", "1") != "0"):
            self.abits = torch.zeros(a.rows * a.C // 8, device=eng.device, dtype=torch.uint8)

    def apply_residual(self, res: Grid, out: Grid, oh, ow, relu_out, r_f32, obits=None):
        """The apply pass of forward(apply=False), fused with the block's cropped residual add and the ReLU between blocks
        (cpc_bn_apply_residual): the activation grid ``a`` is not written, only its sign bits (and, ``obits``, those of the block output)."""
        e = self.eng
        p = e.model._param
        _hip.call("cpc_bn_apply_residual", self.y0.ptr(), _desc(self.y0, self.y0.desc), res.ptr(), _desc(res, res.desc), out.ptr(), _desc(out, out.desc),
                  _hip.ptr(self.stats), _hip.ptr(p[self.prefix + ".weight"]), _hip.ptr(p[self.prefix + ".bias"]), oh, ow, 1, relu_out, r_f32,
                  _hip.ptr(self.abits), _desc(self.a, self.a.desc), _hip.ptr(obits), e.code)

    def backward_res(self, d_out: Grid, obits, d_res: Optional[Grid], oh, ow):
        """backward() with the block's residual add folded in (cpc_bn_bwd_reduce_res / _apply_res): the gradient of the block output d_out,
        masked by the output's sign bits, is what both passes read; the apply pass also writes the residual operand's gradient."""
        e = self.eng
        p, g, code = e.model._param, e.model._grad, e.code
        gw, gb = g[self.prefix + ".weight"], g[self.prefix + ".bias"]
        _hip.call("cpc_bn_bwd_reduce_res", d_out.ptr(), _desc(d_out, d_out.desc), _hip.ptr(obits), _hip.ptr(self.abits), _desc(self.a, self.a.desc),
                  self.y0.ptr(), _desc(self.y0, self.y0.desc), _hip.ptr(self.stats), _hip.ptr(e.slabs), self.nb_bwd, code)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs), _hip.ptr(gw), 1, self.C, self.nb_bwd, 2 * self.C, 1, 1, 0, 0)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs, self.C), _hip.ptr(gb), 1, self.C, self.nb_bwd, 2 * self.C, 1, 1, 0, 0)
        _hip.call("cpc_bn_bwd_apply_res", d_out.ptr(), _desc(d_out, d_out.desc), _hip.ptr(obits), _hip.ptr(self.abits), _desc(self.a, self.a.desc),
                  self.y0.ptr(), self.dy0.ptr(), _desc(self.y0, self.y0.desc), _hip.ptr(self.stats), _hip.ptr(p[self.prefix + ".weight"]),
                  _hip.ptr(gw), _hip.ptr(gb), float(self.y0.count), 1 if self.trained else 0, d_res.ptr() if d_res is not None else None,
                  _desc(d_res, d_res.desc) if d_res is not None else None, oh, ow, code)

    def forward(self, apply=True):
        e, mod = self.eng, self.mod
        p, code = e.model._param, e.code
        self.trained = bool(mod.training or not mod.track_running_stats)
        if self.trained:
            _hip.call("cpc_bn_stats", self.y0.ptr(), _hip.ptr(e.slabs), self.y0.rows, self.C, self.nb, self.y0.code)
            rm = mod.running_mean if mod.track_running_stats else None
            rv = mod.running_var if mod.track_running_stats else None
            momentum = 0.1 if mod.momentum is None else float(mod.momentum)
            _hip.call("cpc_bn_finalize", _hip.ptr(e.slabs), self.nb, self.C, float(self.y0.count), float(mod.eps), momentum,
                      _hip.ptr(self.stats), _hip.ptr(rm), _hip.ptr(rv))
            if mod.track_running_stats and mod.num_batches_tracked is not None:
                e.count_batch(mod.num_batches_tracked)
        else:
            self.stats[0].copy_(mod.running_mean)
            self.stats[1].copy_(torch.rsqrt(mod.running_var + mod.eps))
        if not apply:
            return
        if self.abits is not None:
            _hip.call("cpc_bn_apply_bits", self.y0.ptr(), _desc(self.y0, self.y0.desc), self.a.ptr(), _desc(self.a, self.a.desc),
                      _hip.ptr(self.stats), _hip.ptr(p[self.prefix + ".weight"]), _hip.ptr(p[self.prefix + ".bias"]), 1, _hip.ptr(self.abits), code)
        else:
            _hip.call("cpc_bn_apply", self.y0.ptr(), _desc(self.y0, self.y0.desc), self.a.ptr(), _desc(self.a, self.a.desc), _hip.ptr(self.stats),
                      _hip.ptr(p[self.prefix + ".weight"]), _hip.ptr(p[self.prefix + ".bias"]), 1, self.x_f32, code)

    def backward(self, da: Grid):
        e = self.eng
        p, g, code = e.model._param, e.model._grad, e.code
        gw, gb = g[self.prefix + ".weight"], g[self.prefix + ".bias"]
        if self.abits is not None:
            _hip.call("cpc_bn_bwd_reduce_bits", da.ptr(), _hip.ptr(self.abits), _desc(self.a, self.a.desc), self.y0.ptr(),
                      _desc(self.y0, self.y0.desc), _hip.ptr(self.stats), _hip.ptr(e.slabs), self.nb_bwd, code)
        else:
            _hip.call("cpc_bn_bwd_reduce", da.ptr(), self.a.ptr(), _desc(self.a, self.a.desc), self.y0.ptr(), _desc(self.y0, self.y0.desc),
                      _hip.ptr(self.stats), _hip.ptr(e.slabs), 1, self.nb_bwd, self.x_f32, code)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs), _hip.ptr(gw), 1, self.C, self.nb_bwd, 2 * self.C, 1, 1, 0, 0)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs, self.C), _hip.ptr(gb), 1, self.C, self.nb_bwd, 2 * self.C, 1, 1, 0, 0)
        if self.abits is not None:
            _hip.call("cpc_bn_bwd_apply_bits", da.ptr(), _hip.ptr(self.abits), _desc(self.a, self.a.desc), self.y0.ptr(), self.dy0.ptr(),
                      _desc(self.y0, self.y0.desc), _hip.ptr(self.stats), _hip.ptr(p[self.prefix + ".weight"]), _hip.ptr(gw), _hip.ptr(gb),
                      float(self.y0.count), 1 if self.trained else 0, code)
        else:
            _hip.call("cpc_bn_bwd_apply", da.ptr(), self.a.ptr(), _desc(self.a, self.a.desc), self.y0.ptr(), self.dy0.ptr(),
                      _desc(self.y0, self.y0.desc), _hip.ptr(self.stats), _hip.ptr(p[self.prefix + ".weight"]), _hip.ptr(gw), _hip.ptr(gb),
                      float(self.y0.count), 1, 1 if self.trained else 0, self.x_f32, code)
        gp = getattr(e, "_gp_phase", 0)
        if gp == 1:          # first backward pass of a gradient-penalty step (seeds: the summed scores): keep sum q xhat
            if getattr(self, "s2", None) is None:
                self.s2 = torch.empty_like(gw)
            self.s2.copy_(gw)
        elif gp == 3:        # last pass: the second-order terms of this BatchNorm join the adjoint of its input (gp_terms)
            _accumulate(self.dy0, self.xterm)

    # ---- Wasserstein gradient penalty: tangent pass and second-order terms
    def tangent(self):
        """Tangent of relu(BatchNorm(.)) in train mode: with the batch statistics depending on the input, the tangent of the
        normalisation is the BatchNorm BACKWARD formula applied to the tangent, y. = (gamma / sigma) P a.  (P u = u - <u> - xhat <xhat u>),
        then the primal ReLU mask.  Keeps y. (before the mask) and the sums <a.>, <xhat a.> for gp_terms."""
        e = self.eng
        p, code = e.model._param, e.code
        if not self.trained:
            raise NotImplementedError("gradient penalty through an eval-mode BatchNorm")
        # First-stage BatchNorm of a bf16 engine (x_f32): its input grid, that grid's tangent and the tangent y. are all float32, so
        # the two reductions / the projection run on the float32 kernels; only the step into the (bf16) activation tangent converts.
        fcode, fx = (_hip.F32, 0) if self.x_f32 else (code, self.x_f32)
        y0, a = self.y0, self.a
        y0_t, a_t = _twin(e, y0), _twin(e, a)
        C_ = self.C
        if getattr(self, "yt", None) is None:
            dev = e.device
            self.yt = y0.like(dev, guard_rows=y0.guard_rows)
            self.xterm = y0.like(dev, guard_rows=y0.guard_rows)
            self.t_dgamma = torch.zeros(C_, device=dev, dtype=torch.float32)
            self.t_dbeta = torch.zeros(C_, device=dev, dtype=torch.float32)
            self.s1 = torch.zeros(C_, device=dev, dtype=torch.float32)
            self.ident = torch.cat([torch.zeros(C_), torch.ones(C_)]).to(dev)             # "statistics" of an identity normalisation
            self.ones, self.zeros = torch.ones(C_, device=dev), torch.zeros(C_, device=dev)
            self.coef = torch.zeros(3 * C_, device=dev, dtype=torch.float32)
            self.inv_gamma_stats = torch.zeros(2 * C_, device=dev, dtype=torch.float32)
        # sums over the batch of a. and xhat a. (no mask: the tangent enters the normalisation itself)
        _hip.call("cpc_bn_bwd_reduce", y0_t.ptr(), None, _desc(y0, y0.desc), y0.ptr(), _desc(y0, y0.desc), _hip.ptr(self.stats),
                  _hip.ptr(e.slabs), 0, self.nb_bwd, fx, fcode)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs), _hip.ptr(self.t_dgamma), 1, C_, self.nb_bwd, 2 * C_, 1, 1, 0, 0)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs, C_), _hip.ptr(self.t_dbeta), 1, C_, self.nb_bwd, 2 * C_, 1, 1, 0, 0)
        # y. = gamma rstd (a. - <a.> - xhat <xhat a.>)
        _hip.call("cpc_bn_bwd_apply", y0_t.ptr(), None, _desc(y0, y0.desc), y0.ptr(), self.yt.ptr(), _desc(y0, y0.desc), _hip.ptr(self.stats),
                  _hip.ptr(p[self.prefix + ".weight"]), _hip.ptr(self.t_dgamma), _hip.ptr(self.t_dbeta), float(y0.count), 0, 1,
                  fx, fcode)
        # into the activation's geometry (identity "normalisation": mean 0, rstd 1, gamma 1, beta 0, no ReLU), then the primal mask
        _hip.call("cpc_bn_apply", self.yt.ptr(), _desc(y0, y0.desc), a_t.ptr(), _desc(a, a.desc), _hip.ptr(self.ident), _hip.ptr(self.ones),
                  _hip.ptr(self.zeros), 0, self.x_f32, code)
        _hip.call("cpc_relu_mask", a_t.ptr(), a.ptr(), a.rows * a.C, code)

    def gp_terms(self, da: Grid, gp_grad):
        """After the first backward pass (``da`` = its adjoint at this BatchNorm's output, before the ReLU mask; dy0 = at its input)
        and the tangent pass: the penalty's gradient for the scale, gamma_bar = sum q y./gamma (q = masked da), and the terms the
        input's adjoint gains in the last pass:  -(gamma/sigma) (<q a^.> xhat + <q xhat> a^.) - (<xhat a.>/sigma) delta_in."""
        e = self.eng
        p, code = e.model._param, e.code
        C_, n = self.C, float(self.y0.count)
        gamma = p[self.prefix + ".weight"].detach()
        rstd = self.stats[1]
        # S1 = sum q a^.  with a^. = y. / gamma: the backward reduction run on x := y., "statistics" (0, 1/gamma)
        self.inv_gamma_stats[:C_].zero_()
        self.inv_gamma_stats[C_:].copy_(1.0 / gamma)
        _hip.call("cpc_bn_bwd_reduce", da.ptr(), self.a.ptr(), _desc(self.a, self.a.desc), self.yt.ptr(), _desc(self.y0, self.y0.desc),
                  _hip.ptr(self.inv_gamma_stats), _hip.ptr(e.slabs), 1, self.nb_bwd, self.x_f32, code)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs), _hip.ptr(self.s1), 1, C_, self.nb_bwd, 2 * C_, 1, 1, 0, 0)
        gp_grad[self.prefix + ".weight"].copy_(self.s1)
        self.coef[:C_].copy_(-(gamma * rstd) * self.s1 / n)                  # on xhat
        self.coef[C_:2 * C_].copy_(-rstd * self.s2 / n)                      # on y. (= gamma a^.)
        self.coef[2 * C_:].copy_(-rstd * self.t_dgamma / n)                  # on delta_in
        # (first-stage BatchNorm of a bf16 engine: x, y., the adjoint and the result are all float32 grids)
        fcode, fx = (_hip.F32, 0) if self.x_f32 else (code, self.x_f32)
        _hip.call("cpc_bn_gp_cross", self.y0.ptr(), self.yt.ptr(), self.dy0.ptr(), self.xterm.ptr(), _desc(self.y0, self.y0.desc),
                  _hip.ptr(self.stats), _hip.ptr(self.coef), fx, fcode)


class _CastRelu:
    """ReLU between a convolution output grid y0 and an activation grid ``a`` of another geometry or dtype, for blocks WITHOUT
    BatchNorm (scalogram_model.py:399-400 skipped, :406 kept): the float32 output of a first block's convolution going to the
    bf16 activation the next convolution reads, or an overlapped-row convolution's output going to a grid with top padding
    (:411-412).  Same interface as _BatchNorm; runs on the normalisation kernels with the identity as "statistics"."""

    def __init__(self, eng, y0: Grid, a: Grid):
        self.eng, self.y0, self.a = eng, y0, a
        self.C = y0.C
        self.x_f32 = 1 if (y0.dtype == torch.float32 and eng.dt != torch.float32) else 0
        dev = eng.device
        self.ident = torch.cat([torch.zeros(self.C), torch.ones(self.C)]).to(dev)      # mean 0, 1/sigma 1
        self.ones, self.zeros = torch.ones(self.C, device=dev), torch.zeros(self.C, device=dev)
        self.slab = 0
        self.dy0: Optional[Grid] = None
        self.trained = False

    def forward(self):
        _hip.call("cpc_bn_apply", self.y0.ptr(), _desc(self.y0, self.y0.desc), self.a.ptr(), _desc(self.a, self.a.desc), _hip.ptr(self.ident),
                  _hip.ptr(self.ones), _hip.ptr(self.zeros), 1, self.x_f32, self.eng.code)

    def backward(self, da: Grid):
        _hip.call("cpc_bn_bwd_apply", da.ptr(), self.a.ptr(), _desc(self.a, self.a.desc), self.y0.ptr(), self.dy0.ptr(),
                  _desc(self.y0, self.y0.desc), _hip.ptr(self.ident), _hip.ptr(self.ones), _hip.ptr(self.zeros), _hip.ptr(self.zeros),
                  float(self.y0.count), 1, 0, self.x_f32, self.eng.code)

    def tangent(self):
        e = self.eng
        a_t = _twin(e, self.a)
        _hip.call("cpc_bn_apply", _twin(e, self.y0).ptr(), _desc(self.y0, self.y0.desc), a_t.ptr(), _desc(self.a, self.a.desc),
                  _hip.ptr(self.ident), _hip.ptr(self.ones), _hip.ptr(self.zeros), 0, self.x_f32, e.code)
        _hip.call("cpc_relu_mask", a_t.ptr(), self.a.ptr(), self.a.rows * self.a.C, e.code)

    def gp_terms(self, da: Grid, gp_grad):
        pass          # piecewise linear, no parameters: nothing of second order


class _Stem:
    """First convolution of the encoder + its train-mode BatchNorm + ReLU on the float32 scalogram (scalogram_model.py:392-406 of
    block 0) through csrc/stem.hip: the convolution's output is recomputed in every pass instead of being stored (DESIGN.md,
    scalogram family).  Stands where conv_a and bn_a stand in a _Block whose input gradient is not needed."""

    def __init__(self, eng, wname, bname, conv_mod, bn_prefix, bn_mod, gin: Grid, a: Grid, Ho, Wo):
        self.eng, self.wname, self.bname, self.prefix, self.mod, self.gin, self.a = eng, wname, bname, bn_prefix, bn_mod, gin, a
        self.C = conv_mod.out_channels
        kh, kw = conv_mod.kernel_size
        sh, sw = conv_mod.stride
        ph, pw = conv_mod.padding
        self.taps = gin.C * kh * kw
        self.conv = (C.c_int * 9)(self.C, kh, kw, sh, sw, ph, pw, Ho, Wo)
        self.count = float(gin.B * Wo * Ho)
        # two workgroups per CU stay resident (registers): one round of them, each walking its share of the output columns
        self.nb = max(1, min(512, gin.B * Wo))
        self.slab = self.nb * max(2 * self.C, self.C * self.taps)
        self.stats = torch.zeros(2, self.C, device=eng.device, dtype=torch.float32)
        self.zeros = torch.zeros(self.C, device=eng.device, dtype=torch.float32)
        self.trained = True

    def _args(self):
        p = self.eng.model._param
        return (self.gin.ptr(), _desc(self.gin, self.gin.desc), _hip.ptr(p[self.wname]), _hip.ptr(p.get(self.bname)) if self.bname else None,
                C.cast(self.conv, C.c_void_p))

    def forward(self):
        e, mod = self.eng, self.mod
        p = e.model._param
        self.trained = bool(mod.training or not mod.track_running_stats)
        if self.trained:
            _hip.call("cpc_stem_stats", *self._args(), _hip.ptr(e.slabs), self.nb)
            rm = mod.running_mean if mod.track_running_stats else None
            rv = mod.running_var if mod.track_running_stats else None
            momentum = 0.1 if mod.momentum is None else float(mod.momentum)
            _hip.call("cpc_bn_finalize", _hip.ptr(e.slabs), self.nb, self.C, self.count, float(mod.eps), momentum, _hip.ptr(self.stats),
                      _hip.ptr(rm), _hip.ptr(rv))
            if mod.track_running_stats and mod.num_batches_tracked is not None:
                e.count_batch(mod.num_batches_tracked)
        else:
            self.stats[0].copy_(mod.running_mean)
            self.stats[1].copy_(torch.rsqrt(mod.running_var + mod.eps))
        _hip.call("cpc_stem_apply", *self._args(), _hip.ptr(self.stats), _hip.ptr(p[self.prefix + ".weight"]), _hip.ptr(p[self.prefix + ".bias"]),
                  self.a.ptr(), _desc(self.a, self.a.desc), self.nb, e.code)

    def backward(self, da: Grid):
        """da: gradient of the activation (before its ReLU mask) -> BatchNorm scale / shift gradients and the convolution's weight
        gradient; the convolution's output gradient is formed inside the weight-gradient kernel and never stored."""
        e = self.eng
        p, g = e.model._param, e.model._grad
        gw, gb = g[self.prefix + ".weight"], g[self.prefix + ".bias"]
        a = self.a
        _hip.call("cpc_stem_bwd_reduce", *self._args(), _hip.ptr(self.stats), da.ptr(), a.ptr(), _desc(a, a.desc), _hip.ptr(e.slabs), self.nb,
                  e.code)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs), _hip.ptr(gw), 1, self.C, self.nb, 2 * self.C, 1, 1, 0, 0)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs, self.C), _hip.ptr(gb), 1, self.C, self.nb, 2 * self.C, 1, 1, 0, 0)
        dg, db = (gw, gb) if self.trained else (self.zeros, self.zeros)
        _hip.call("cpc_stem_bwd_wgrad", *self._args(), _hip.ptr(self.stats), _hip.ptr(p[self.prefix + ".weight"]), _hip.ptr(dg), _hip.ptr(db),
                  self.count, da.ptr(), a.ptr(), _desc(a, a.desc), _hip.ptr(e.slabs), self.nb, e.code)
        n = self.C * self.taps
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs), _hip.ptr(g[self.wname]), 1, n, self.nb, n, 1, 1, 0, 0)
        if self.bname and self.bname in g:
            if self.trained:          # exactly zero in front of a train-mode BatchNorm (its input gradient has zero mean per channel)
                g[self.bname].zero_()
            else:                     # running statistics: dy = gamma rstd g, summed over the positions
                g[self.bname].copy_(p[self.prefix + ".weight"].detach() * self.stats[1] * gb)


class _StemResidual:
    """Residual branch of block 0 on the float32 input: [MaxPool2d] -> 1x1 Conv2d (no bias, no padding) -> cropped add
    (scalogram_model.py:434-446, :462-472) with the projection applied inside the add (csrc/stem.hip): neither the projected
    float32 grid nor its gradient is stored."""

    def __init__(self, eng, wname, xp: Grid, main: Grid, out: Grid, oh, ow, relu):
        self.eng, self.wname, self.xp, self.main, self.out, self.oh, self.ow, self.relu = eng, wname, xp, main, out, oh, ow, relu
        self.nb = max(1, min(1024, main.B * main.W))
        self.n = main.C * xp.C
        self.slab = self.nb * self.n

    def forward(self, bn=None, obits=None):
        """``bn``: the block's second BatchNorm whose apply pass was left out (forward(apply=False)): normalise + ReLU inside this add
        (cpc_stem_residual_bn_add), the normalised branch is kept as sign bits only (``obits``: and the block output's sign bits)."""
        e, m, xp, o = self.eng, self.main, self.xp, self.out
        p = e.model._param
        if bn is not None:
            _hip.call("cpc_stem_residual_bn_add", bn.y0.ptr(), _desc(bn.y0, bn.y0.desc), xp.ptr(), _desc(xp, xp.desc), _hip.ptr(p[self.wname]),
                      o.ptr(), _desc(o, o.desc), self.oh, self.ow, 1 if self.relu else 0, _hip.ptr(bn.stats), _hip.ptr(p[bn.prefix + ".weight"]),
                      _hip.ptr(p[bn.prefix + ".bias"]), _hip.ptr(bn.abits), _desc(bn.a, bn.a.desc), _hip.ptr(obits), e.code)
            return
        _hip.call("cpc_stem_residual_add", m.ptr(), _desc(m, m.desc), xp.ptr(), _desc(xp, xp.desc), _hip.ptr(p[self.wname]),
                  o.ptr(), _desc(o, o.desc), self.oh, self.ow, 1 if self.relu else 0, e.code)

    def backward_bits(self, d_out: Grid, obits):
        """The projection's weight gradient alone, the ReLU mask from the output's sign bits (cpc_stem_residual_wgrad_bits): the masked
        gradient of the main branch is not stored — the second BatchNorm's fused backward passes read d_out and the same bits."""
        e, xp, o, m = self.eng, self.xp, self.out, self.main
        _hip.call("cpc_stem_residual_wgrad_bits", d_out.ptr(), _hip.ptr(obits), _desc(o, o.desc), _desc(m, m.desc), xp.ptr(), _desc(xp, xp.desc),
                  _hip.ptr(e.slabs), self.oh, self.ow, self.nb, e.code)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs), _hip.ptr(e.model._grad[self.wname]), 1, self.n, self.nb, self.n, 1, 1, 0, 0)

    def backward(self, d_out: Grid, d_main: Grid):
        e, xp, o = self.eng, self.xp, self.out
        _hip.call("cpc_stem_residual_bwd", d_out.ptr(), o.ptr(), _desc(o, o.desc), d_main.ptr(), _desc(d_main, d_main.desc), xp.ptr(),
                  _desc(xp, xp.desc), _hip.ptr(e.slabs), self.oh, self.ow, 1 if self.relu else 0, self.nb, e.code)
        _hip.call("cpc_reduce_slabs", _hip.ptr(e.slabs), _hip.ptr(e.model._grad[self.wname]), 1, self.n, self.nb, self.n, 1, 1, 0, 0)


class _Block:
    """One ScalogramEncoderBlock (scalogram_model.py:372-479) on grids."""

    def __init__(self, eng, idx, blk, gin: Grid, in_f32, last, next_top, prefix="encoder.", need_input_grad=None):
        self.eng, self.idx, self.blk, self.gin, self.in_f32, self.last = eng, idx, blk, gin, in_f32, last
        dev, dt = eng.device, eng.dt
        cfg = blk.cfg
        pre = f"{prefix}blocks.{idx}."
        self.need_input_grad = (idx != 0) if need_input_grad is None else bool(need_input_grad)
        first = not self.need_input_grad
        mm = blk.main_modules
        ceil = cfg.get('ceil_pooling', False)
        pooled = lambda n, p: (_ceil_div(n, p) if ceil else n // p) if p > 1 else n
        self.pool1, self.pool2 = int(cfg.get('pooling_1', 1)), int(cfg.get('pooling_2', 1))
        has_bn = cfg['batch_norm']
        bias = cfg['bias']
        top2 = cfg['top_padding_2'] or 0
        if (cfg['top_padding_1'] or 0) != gin.top:
            raise AssertionError("block input grid was not allocated with this block's top_padding_1")
        # ---- main branch
        i1, i2 = blk.index['conv_1'], blk.index['conv_2']
        k1, k2 = tuple(cfg['kernel_size_1']), tuple(cfg['kernel_size_2'])
        s1, p1 = int(cfg['stride_1']), int(cfg['padding_1'])
        Hc, Wc = (gin.top + gin.H + 2 * p1 - k1[0]) // s1 + 1, (gin.W + 2 * p1 - k1[1]) // s1 + 1
        # [MaxPool2d(pooling_1)] sits between the BatchNorm and the ReLU in the reference (scalogram_model.py:401-405); max
        # pooling commutes with the monotone ReLU, so here the fused BatchNorm + ReLU output is pooled
        Ha1, Wa1 = pooled(Hc, self.pool1), pooled(Wc, self.pool1)
        if Hc < 1 or Wc < 1 or Ha1 < 1 or Wa1 < 1:
            raise ValueError(f"block {idx}: nothing left of a {gin.top + gin.H} x {gin.W} input after convolution 1 / pooling_1")
        G2 = _col_group(cfg['out_channels'], k2[1], (cfg['stride_2'],) * 2, cfg['padding_2'])
        a_geom = dict(top=top2, tail=(-(top2 + Ha1)) % G2, guard_rows=k2[0] + 16)
        # Without BatchNorm the convolution output (ReLU in the GEMM epilogue) IS the activation the second convolution reads, written
        # straight into a grid with that convolution's top padding (:411-412) -- unless it has to change dtype (a first block's float32
        # convolution in bf16 mode) or comes from a kernel that cannot address such a grid: then a _CastRelu pass stands where the
        # BatchNorm would.
        cast = (not has_bn) and in_f32 and dt != torch.float32
        direct = (not has_bn) and (not cast) and self.pool1 == 1 and bool(a_geom['top'] or a_geom['tail'])
        if direct and (cfg.get('separable') or _col_ok(k1[1], s1, s1, p1, cfg['in_channels'], in_f32)):
            direct, cast = False, bool(top2)        # (a missing tail only costs the row grouping of the next GEMM)
        # block 0 in a training engine (no input gradient): convolution + BatchNorm + ReLU through the recomputing kernels of csrc/stem.hip
        self.stem = None
        if (in_f32 and first and has_bn and self.pool1 == 1 and not cfg.get('separable') and gin.top == 0 and
                os.environ.get("CPC_STEM", "1") != "0" and
                _hip.lib().cpc_stem_supported(gin.C, cfg['hidden_channels'], k1[0], k1[1], s1, gin.H, p1) == 1):
            self.a_a = Grid(gin.B, Wa1, Ha1, cfg['hidden_channels'], dev, dt, **a_geom)
            self.stem = _Stem(eng, f"{pre}main_modules.{i1}.weight", f"{pre}main_modules.{i1}.bias" if bias else None, mm[i1],
                              f"{pre}main_modules.{blk.index['bn_1']}", mm[blk.index['bn_1']], gin, self.a_a, Hc, Wc)
            self.conv_a = self.bn_a = self.a_full = None
        else:
            self.conv_a = _make_conv(eng, f"{pre}main_modules.{i1}", mm[i1], gin, bias, in_f32=in_f32, need_dgrad=not first,
                                     relu=not has_bn, out_pad=(a_geom['top'], a_geom['tail'], a_geom['guard_rows']) if direct else None)
        ya = self.conv_a.y0 if self.conv_a is not None else None
        assert ya is None or (ya.H, ya.W) == (Hc, Wc)
        self.a_full = None
        if self.stem is not None:
            pass
        elif has_bn:
            self.a_a = Grid(ya.B, Wa1, Ha1, ya.C, dev, dt, **a_geom)
            if self.pool1 > 1:
                self.a_full = Grid(ya.B, ya.W, ya.H, ya.C, dev, dt)
            self.bn_a = _BatchNorm(eng, f"{pre}main_modules.{blk.index['bn_1']}", mm[blk.index['bn_1']], ya,
                                   self.a_full if self.pool1 > 1 else self.a_a)
            (self.conv_a.pw if isinstance(self.conv_a, _SepConv) else self.conv_a).bn_after = self.bn_a
        else:
            self.bn_a = None
            if self.pool1 > 1:
                self.a_full = Grid(ya.B, ya.W, ya.H, ya.C, dev, dt) if cast else ya
                self.a_a = Grid(ya.B, Wa1, Ha1, ya.C, dev, dt, **a_geom)
            else:
                self.a_a = Grid(ya.B, Wa1, Ha1, ya.C, dev, dt, **a_geom) if cast else ya
            if cast:
                self.bn_a = _CastRelu(eng, ya, self.a_full if self.pool1 > 1 else self.a_a)
        self.conv_b = _make_conv(eng, f"{pre}main_modules.{i2}", mm[i2], self.a_a, bias, relu=not has_bn)
        yb = self.conv_b.y0
        Hb2, Wb2 = pooled(yb.H, self.pool2), pooled(yb.W, self.pool2)
        if Hb2 < 1 or Wb2 < 1:
            raise ValueError(f"block {idx}: pooling_2 leaves nothing of a {yb.H} x {yb.W} activation")
        self.main_full = None
        if has_bn:
            self.main = Grid(yb.B, Wb2, Hb2, yb.C, dev, dt)
            if self.pool2 > 1:
                self.main_full = Grid(yb.B, yb.W, yb.H, yb.C, dev, dt)
            self.bn_b = _BatchNorm(eng, f"{pre}main_modules.{blk.index['bn_2']}", mm[blk.index['bn_2']], yb,
                                   self.main_full if self.pool2 > 1 else self.main)
            (self.conv_b.pw if isinstance(self.conv_b, _SepConv) else self.conv_b).bn_after = self.bn_b
        else:
            self.bn_b = None
            if self.pool2 > 1:
                self.main_full = yb
                self.main = Grid(yb.B, Wb2, Hb2, yb.C, dev, dt)
            else:
                self.main = yb
        # ---- residual branch
        self.res_conv = self.rp = None
        if blk.residual:
            src, src_f32 = gin, in_f32
            if gin.top:
                raise NotImplementedError("residual branch with top_padding_1")
            if blk.res_pool > 1:
                p_ = blk.res_pool
                self.rp = Grid(gin.B, _ceil_div(gin.W, p_), _ceil_div(gin.H, p_), gin.C, dev, torch.float32 if in_f32 else dt)
                src, src_f32 = self.rp, in_f32
            self.stem_res = None
            if ('res_conv' in blk.index and src_f32 and first and blk.residual_modules[blk.index['res_conv']].padding == (0, 0) and
                    src.C <= 2 and os.environ.get("CPC_STEM", "1") != "0" and cfg['out_channels'] % 8 == 0 and
                    256 % (cfg['out_channels'] // 8) == 0 and cfg['out_channels'] <= 128):
                ri = blk.index['res_conv']
                self.stem_res = f"{pre}residual_modules.{ri}.weight"
                self.res = src                      # 1x1, no padding: the projected grid would have the pooled input's extents
            elif 'res_conv' in blk.index:
                ri = blk.index['res_conv']
                self.res_conv = _Conv(eng, f"{pre}residual_modules.{ri}.weight", None, blk.residual_modules[ri], src, in_f32=src_f32,
                                      need_dgrad=not first)
                self.res = self.res_conv.y0
            else:
                if src_f32:
                    raise NotImplementedError("identity residual on the float32 scalogram input")
                self.res = src
            m = self.main
            o_h, o_w = (self.res.H - m.H + 1) / 2, (self.res.W - m.W + 1) / 2
            self.oh = self.res.H - int(o_h + m.H) if int(o_h) > 0 else 0
            self.ow = self.res.W - int(o_w + m.W) if int(o_w) > 0 else 0
            if (int(o_h) <= 0 and self.res.H != m.H) or (int(o_w) <= 0 and self.res.W != m.W):
                raise ValueError(f"block {idx}: residual {self.res.H}x{self.res.W} cannot be cropped onto main {m.H}x{m.W}")
            # (an even number of allocated rows: the parity-route data gradient of the next block's 3x3 stride-2 convolution writes row pairs)
            self.out = Grid(m.B, m.W, m.H, m.C, dev, dt, top=next_top, tail=0 if last else (next_top + m.H) % 2)
            self.r_f32 = 1 if (self.res.dtype == torch.float32 and dt != torch.float32) else 0
            if self.stem_res is not None:
                self.stem_res = _StemResidual(eng, self.stem_res, self.res, m, self.out, self.oh, self.ow, relu=not last)
        else:
            self.stem_res = None
            if next_top:
                raise NotImplementedError("top_padding_1 after a block without residual branch")
            self.out = self.main
        self.slab = max([c.slab for c in (self.conv_a, self.conv_b, self.res_conv, self.stem, self.stem_res) if c is not None] +
                        [b.slab for b in (self.bn_a, self.bn_b) if b is not None])

    def allocate_grads(self, d_in: Optional[Grid]):
        """Gradient grids; ``d_in`` is the gradient grid of the block input (None for the first block)."""
        dev, dt = self.eng.device, self.eng.dt
        self.d_in = d_in
        self.d_out = self.out.like(dev)
        if self.blk.residual:
            self.d_main = self.main.like(dev, guard_rows=self.conv_b.kh + 16)
            self.d_res = self.res.like(dev) if self.stem_res is None else None
            self.d_rp = self.rp.like(dev) if (self.rp is not None and self.res_conv is not None) else None
            if self.res_conv is not None:
                self.res_conv.dy0 = self.d_res
        else:
            self.d_main = self.d_out
        # with pooling the gradient first goes back through the pooling to the full-resolution activation
        self.d_main_full = self.main_full.like(dev, guard_rows=self.conv_b.kh + 16) if self.pool2 > 1 else None
        if self.bn_b is not None:
            self.bn_b.dy0 = self.conv_b.y0.like(dev, guard_rows=self.conv_b.kh + 16)
            self.conv_b.dy0 = self.bn_b.dy0
        else:
            self.conv_b.dy0 = self.d_main_full if self.pool2 > 1 else self.d_main
        self.d_a = self.a_a.like(dev, guard_rows=self.conv_b.kh + 16)
        if self.stem is not None:
            self.d_a_full = None
            return
        self.d_a_full = self.a_full.like(dev, guard_rows=self.conv_a.kh + 16) if self.pool1 > 1 else None
        if self.bn_a is not None:
            self.bn_a.dy0 = self.conv_a.y0.like(dev, guard_rows=self.conv_a.kh + 16)
            self.conv_a.dy0 = self.bn_a.dy0
        else:
            self.conv_a.dy0 = self.d_a_full if self.pool1 > 1 else self.d_a

    def convs(self):
        return [c for c in (self.conv_a, self.conv_b, self.res_conv) if c is not None]

    def prepare(self):
        for c in self.convs():
            c.prepare()

    def forward(self):
        e, code = self.eng, self.eng.code
        if self.stem is not None:
            self.stem.forward()
        else:
            self.conv_a.forward()
        if self.bn_a is not None:
            self.bn_a.forward()
        if self.pool1 > 1:
            _hip.call("cpc_maxpool2d_fwd", self.a_full.ptr(), _desc(self.a_full, self.a_full.desc), self.a_a.ptr(),
                      _desc(self.a_a, self.a_a.desc), self.pool1, 0, code)
        self.conv_b.forward()
        # second BatchNorm + ReLU, residual add and the ReLU between blocks in one pass where nothing else reads the normalised branch
        # (CPC_BN_RESIDUAL=0: two passes; the gradient penalty's tangent pass reads it)
        fuse = (self.bn_b is not None and self.blk.residual and self.pool2 == 1 and self.bn_b.abits is not None and
                not getattr(e, "gp_capable", False) and self.main.C % 8 == 0 and os.environ.get("CPC_BN_RESIDUAL", "1") != "0")
        self._fused_bwd = False
        if self.bn_b is not None:
            self.bn_b.forward(apply=not fuse)
        if self.pool2 > 1:
            _hip.call("cpc_maxpool2d_fwd", self.main_full.ptr(), _desc(self.main_full, self.main_full.desc), self.main.ptr(),
                      _desc(self.main, self.main.desc), self.pool2, 0, code)
        if self.blk.residual:
            if self.rp is not None:
                _hip.call("cpc_maxpool2d_fwd", self.gin.ptr(), _desc(self.gin, self.gin.desc), self.rp.ptr(), _desc(self.rp, self.rp.desc),
                          self.blk.res_pool, 1 if self.in_f32 else 0, self.rp.code)
            if self.res_conv is not None:
                self.res_conv.forward()
            if self.stem_res is not None:
                # (fused backward for the first block: needs the ReLU behind the add, i.e. a block that is not the last one)
                self._fused_bwd = fuse and self.stem_res.relu and os.environ.get("CPC_BN_RESIDUAL_BWD", "1") != "0"
                if self._fused_bwd and getattr(self, "obits", None) is None:
                    self.obits = torch.zeros(self.out.rows * self.out.C // 8, device=e.device, dtype=torch.uint8)
                self.stem_res.forward(self.bn_b if fuse else None, self.obits if self._fused_bwd else None)
            elif fuse:
                # (the backward pass folds the residual add into the BatchNorm's passes where the residual operand is a bf16 grid: it then
                # needs the sign bits of the block output in place of the output itself)
                self._fused_bwd = (not self.r_f32) and os.environ.get("CPC_BN_RESIDUAL_BWD", "1") != "0"
                if self._fused_bwd and not self.last and getattr(self, "obits", None) is None:
                    self.obits = torch.zeros(self.out.rows * self.out.C // 8, device=e.device, dtype=torch.uint8)
                self.bn_b.apply_residual(self.res, self.out, self.oh, self.ow, 0 if self.last else 1, self.r_f32,
                                         self.obits if (self._fused_bwd and not self.last) else None)
            else:
                _hip.call("cpc_residual_add", self.main.ptr(), _desc(self.main, self.main.desc), self.res.ptr(), _desc(self.res, self.res.desc),
                          self.out.ptr(), _desc(self.out, self.out.desc), self.oh, self.ow, 0 if self.last else 1, self.r_f32, code)

    # ---- Wasserstein gradient penalty
    def tangent(self):
        """The block applied to the tangent of its input: bias-free convolutions, BatchNorm tangents, pooling and ReLU selections
        taken from the primal pass."""
        e, code = self.eng, self.eng.code
        T = lambda grid: _twin(e, grid)
        self.conv_a.forward(tangent=True)
        if self.bn_a is not None:
            self.bn_a.tangent()
        if self.pool1 > 1:
            _hip.call("cpc_maxpool2d_select", self.a_full.ptr(), T(self.a_full).ptr(), _desc(self.a_full, self.a_full.desc),
                      T(self.a_a).ptr(), _desc(self.a_a, self.a_a.desc), self.pool1, 0, code)
        self.conv_b.forward(tangent=True)
        if self.bn_b is not None:
            self.bn_b.tangent()
        if self.pool2 > 1:
            _hip.call("cpc_maxpool2d_select", self.main_full.ptr(), T(self.main_full).ptr(), _desc(self.main_full, self.main_full.desc),
                      T(self.main).ptr(), _desc(self.main, self.main.desc), self.pool2, 0, code)
        if self.blk.residual:
            if self.rp is not None:
                _hip.call("cpc_maxpool2d_select", self.gin.ptr(), T(self.gin).ptr(), _desc(self.gin, self.gin.desc), T(self.rp).ptr(),
                          _desc(self.rp, self.rp.desc), self.blk.res_pool, 1 if self.in_f32 else 0, self.rp.code)
            if self.res_conv is not None:
                self.res_conv.forward(tangent=True)
            out_t = T(self.out)
            _hip.call("cpc_residual_add", T(self.main).ptr(), _desc(self.main, self.main.desc), T(self.res).ptr(),
                      _desc(self.res, self.res.desc), out_t.ptr(), _desc(self.out, self.out.desc), self.oh, self.ow, 0, self.r_f32, code)
            if not self.last:
                _hip.call("cpc_relu_mask", out_t.ptr(), self.out.ptr(), self.out.rows * self.out.C, code)

    def gp_grads(self, gp_grad):
        """Penalty parts of this block's parameter gradients + the BatchNorms' second-order terms (between the first backward
        pass, whose adjoints are still in the gradient grids, and the last one)."""
        if isinstance(self.conv_a, _SepConv) or isinstance(self.conv_b, _SepConv):
            raise NotImplementedError("gradient penalty through Conv2dSeparable")
        self.conv_a.gp_wgrad(gp_grad)
        self.conv_b.gp_wgrad(gp_grad)
        if self.res_conv is not None:
            self.res_conv.gp_wgrad(gp_grad)
        if self.bn_b is not None:
            self.bn_b.gp_terms(self.d_main_full if self.pool2 > 1 else self.d_main, gp_grad)
        if self.bn_a is not None:
            self.bn_a.gp_terms(self.d_a_full if self.pool1 > 1 else self.d_a, gp_grad)

    def backward(self):
        e, code = self.eng, self.eng.code
        first = not self.need_input_grad
        fused_bwd = getattr(self, "_fused_bwd", False)
        if self.stem_res is not None and fused_bwd:
            self.stem_res.backward_bits(self.d_out, self.obits)
        elif self.stem_res is not None:
            self.stem_res.backward(self.d_out, self.d_main)
        elif fused_bwd:
            pass          # (the residual add's backward runs inside the second BatchNorm's passes below)
        elif self.blk.residual:
            # (d_res was zeroed when it was allocated: the cropped add's backward overwrites the same interior every step and never
            # touches the border)
            _hip.call("cpc_residual_add_bwd", self.d_out.ptr(), self.out.ptr(), _desc(self.out, self.out.desc), self.d_main.ptr(),
                      _desc(self.d_main, self.d_main.desc), self.d_res.ptr(), _desc(self.d_res, self.d_res.desc), self.oh, self.ow,
                      0 if self.last else 1, self.r_f32, code)

        def unpool(full, d_full, pooled, d_pooled, p):
            """gradient of the pooled activation -> gradient of the full-resolution one (zero outside the windows)"""
            d_full.t.zero_()
            _hip.call("cpc_maxpool2d_bwd", full.ptr(), d_full.ptr(), _desc(full, full.desc), d_pooled.ptr(), _desc(pooled, pooled.desc),
                      p, 1, code)
            return d_full

        # second convolution
        g_b, act_b = self.d_main, self.main
        if self.pool2 > 1:
            g_b, act_b = unpool(self.main_full, self.d_main_full, self.main, self.d_main, self.pool2), self.main_full
        if fused_bwd:
            self.bn_b.backward_res(self.d_out, None if self.last else self.obits, self.d_res if self.stem_res is None else None, self.oh, self.ow)
        elif self.bn_b is not None:
            self.bn_b.backward(g_b)
        else:
            _hip.call("cpc_relu_mask", g_b.ptr(), act_b.ptr(), g_b.rows * g_b.C, code)
        self.conv_b.backward(self.d_a, mask_input=self.bn_a is None and self.stem is None and self.pool1 == 1)
        if self.stem is not None:          # (block 0: no input gradient, and its residual branch was handled above)
            self.stem.backward(self.d_a)
            if self.blk.residual and self.stem_res is None and self.res_conv is not None:
                self.res_conv.backward(None)
            return
        # first convolution
        g_a = self.d_a
        if self.pool1 > 1:
            g_a = unpool(self.a_full, self.d_a_full, self.a_a, self.d_a, self.pool1)
            if self.bn_a is None:
                _hip.call("cpc_relu_mask", g_a.ptr(), self.a_full.ptr(), g_a.rows * g_a.C, code)
        if self.bn_a is not None:
            self.bn_a.backward(g_a)
        self.conv_a.backward(None if first else self.d_in)
        # residual branch (adds into the block-input gradient after the main branch wrote it)
        if self.blk.residual and self.stem_res is None:
            if self.res_conv is not None:
                if self.rp is not None:
                    self.res_conv.backward(None if first else self.d_rp)
                    if not first:
                        _hip.call("cpc_maxpool2d_bwd", self.gin.ptr(), self.d_in.ptr(), _desc(self.gin, self.gin.desc), self.d_rp.ptr(),
                                  _desc(self.rp, self.rp.desc), self.blk.res_pool, 1, self.rp.code)
                else:
                    self.res_conv.backward(None if first else self.d_in, accumulate=True)
            elif not first:
                if self.rp is not None:
                    _hip.call("cpc_maxpool2d_bwd", self.gin.ptr(), self.d_in.ptr(), _desc(self.gin, self.gin.desc), self.d_res.ptr(),
                              _desc(self.rp, self.rp.desc), self.blk.res_pool, 1, self.rp.code)
                else:
                    _accumulate(self.d_in, self.d_res)


class ScalogramCPCEngine(CPCEngine):
    """CPCEngine whose encoder is a ScalogramResidualEncoder fed with (B, C, bins, frames) scalograms."""

    def __init__(self, model, in_shape, device, dtype: torch.dtype, gradient_penalty: bool = False):
        """``gradient_penalty``: also provide the gradient with respect to the input scalogram and the tangent pass the
        Wasserstein gradient penalty needs (loss_and_grads(..., gradient_penalty=factor))."""
        enc, ar = model.encoder, model.autoregressive_model
        self.gp_capable = bool(gradient_penalty)
        self._twins, self._gp_phase = {}, 0
        self.model = model
        self.device = torch.device(device)
        self.dt = dtype
        self.code = _hip.dtype_code(dtype)
        B, Cin, Hin, Win = (int(v) for v in in_shape)
        self.B, self.in_shape = B, (B, Cin, Hin, Win)
        self.L = Win
        self.E = int(model.enc_size)
        self.H = int(model.ar_size)
        self.K = int(model.prediction_steps)
        self.V = int(model.visible_steps)
        self.x_off = 0
        self.n = 0
        self.colsum_blocks = 1024
        model._flatten_parameters(self.device)
        # ---- encoder graph
        blocks = list(enc.blocks)
        top0 = blocks[0].cfg['top_padding_1'] or 0
        if top0:
            raise NotImplementedError("top_padding_1 on the first scalogram block")
        # (its own 165 MB at configs[2] only if a batch ever arrives in another layout: PreprocessingModule's output is read in place)
        self.x_grid = Grid(B, Win, Hin, Cin, self.device, torch.float32, lazy=True)
        self._x_own = None
        self.blocks: List[_Block] = []
        gin, in_f32 = self.x_grid, True
        for i, blk in enumerate(blocks):
            last = i == len(blocks) - 1
            next_top = 0 if last else (blocks[i + 1].cfg['top_padding_1'] or 0)
            b = _Block(self, i, blk, gin, in_f32, last, next_top, need_input_grad=True if (i == 0 and self.gp_capable) else None)
            self.blocks.append(b)
            gin, in_f32 = b.out, False
        out = self.blocks[-1].out
        if out.C != self.E:
            raise ValueError(f"the scalogram encoder ends with {out.C} channels, the model's enc_size is {self.E}")
        # The reference returns x[:, :, 0, :] (scalogram_model.py:529): the FIRST frequency row of whatever is left.  With one row
        # left (architectures 1, 3-9) the last grid IS the [B][frames][E] buffer the context networks read; with more
        # (scalogram_resnet_architecture_2 ends with two) row 0 is copied out, and its gradient copied back beside zeros.
        self.top_grid = self.d_top_grid = None
        if out.H != 1 or out.Ha != 1:
            self.top_grid = Grid(B, out.W, 1, self.E, self.device, dtype)
            self.d_top_grid = self.top_grid.like(self.device)
        self.T = out.W
        if self.T < self.V + self.K:
            raise ValueError(f"scalogram gives {self.T} encoder frames, need visible+prediction = {self.V + self.K}")
        d_in = self.x_grid.like(self.device) if self.gp_capable else None      # gradient w.r.t. the scalogram (penalty only)
        self.d_x = d_in
        for b in self.blocks:
            b.allocate_grads(d_in)
            d_in = b.d_out
        self.geo = SimpleNamespace(alloc=[self.T], valid=[self.T])
        if self.top_grid is not None:
            self.act, self.dact = [self.top_grid.t], [self.d_top_grid.t]
        else:
            self.act, self.dact = [out.t], [self.blocks[-1].d_out.t]
        self.aux = side_stream(self.device)      # side stream, see engine.CPCEngine
        self.ctx = make_context(self, ar) if (self.V + self.K) > 0 else None
        need = [b.slab for b in self.blocks] + [self.colsum_blocks * max(max(b.a_a.C, b.conv_b.cout) for b in self.blocks)]
        self._alloc_head(need)

    # Operand copies for the NEXT step: rebuilt on the side stream right after the step's last Adam launch (FusedAdam.after_update), beside
    # the next batch's CQT GEMMs, instead of on the main stream in front of the first convolution (0.4 ms of small launches per configs[2]
    # step).  The next prepare_weights() only waits for their event; any other change of the parameters (load_state_dict, a torch
    # optimizer, the NaN guard's restore) changes _param_state() and the copies are rebuilt in place as before.  CPC_PREPARE_AHEAD=0: off.
    supports_prepare_ahead = True

    def prepare_ahead(self, lo, hi, final):
        if final and self.ctx is not None and getattr(self.ctx, "ahead_ok", False) and self.use_aux and os.environ.get("CPC_PREPARE_AHEAD", "1") != "0":
            self._prepare_all_ahead()

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("the CPC hot path runs on the GPU only (no CPU fallback): move the batch to the device")
        if tuple(x.shape) != self.in_shape:
            raise ValueError(f"expected a scalogram batch of shape {self.in_shape}, got {tuple(x.shape)}")

    def _prepare_encoder_weights(self):
        _prepare_convs(self, self, [c for b in self.blocks for c in b.convs()])

    def encoder_forward(self, x):
        """x (B, C, bins, frames) float32 — typically PreprocessingModule's permuted view, whose memory already is the
        channels-last grid; any other layout is re-laid out once."""
        self._check_input(x)
        cl = x.detach().permute(0, 3, 2, 1)
        if cl.is_contiguous() and cl.dtype == torch.float32 and self.x_grid.top == 0 and self.x_grid.Ha == self.x_grid.H:
            # PreprocessingModule's output IS the channels-last grid: read it in place (every kernel that reads the input grid stays
            # inside its B x W x H x C elements; the 165 MB copy of a configs[2] batch is gone)
            self.x_grid.t = cl.reshape(-1)
            # the backward pass recomputes block 0 from this memory (stem kernels, residual weight gradient): an in-place write to the
            # caller's tensor between forward and backward would silently change the gradients -- its version counter is checked there
            self._x_alias = (x, x._version)
        else:
            self._x_alias = None
            if self._x_own is None:
                self._x_own = self.x_grid.allocate()
            self.x_grid.t = self._x_own
            self.x_grid.t.view(cl.shape).copy_(cl)
        # ``side_job``: a callable the caller left for this step (the trainer's InputAhead: the NEXT batch's CQT + scalogram kernels on the
        # side stream), run once, behind the encoder's last block: from there to the first large data gradient (context network, loss,
        # the upper blocks' backward passes) the main queue holds short, latency-bound launches.  Measured on configs[2] (30 steps after
        # 15, one box): off 12.08 - 12.17 ms, in front of the step 12.25, after block 0 / 1 / 2 / 3: 12.15 / 12.04 / 12.00 / 11.84, behind
        # the loss kernels 11.86; on a stream of its own at the default priority 11.88 - 12.25.  CPC_SIDE_JOB_BLOCK: after which block.
        job, self.side_job = getattr(self, "side_job", None), None
        at = min(int(os.environ.get("CPC_SIDE_JOB_BLOCK", str(len(self.blocks) - 1))), len(self.blocks) - 1)
        if job is not None and at < 0:
            job()
        for i, b in enumerate(self.blocks):
            b.forward()
            if job is not None and i == at:
                job()
        if self.top_grid is not None:
            self.top_grid.t.view(self.B, self.T, self.E).copy_(self._row0(self.blocks[-1].out))

    @staticmethod
    def _row0(grid):
        return grid.t.view(grid.B, grid.W, grid.Ha, grid.C)[:, :, grid.top, :]

    def _top_tangent(self):
        """Tangent of the encoder's top buffer during a gradient-penalty step."""
        out_t = _twin(self, self.blocks[-1].out)
        if self.top_grid is None:
            return out_t.t
        top_t = _twin(self, self.top_grid)
        top_t.t.view(self.B, self.T, self.E).copy_(self._row0(out_t))
        return top_t.t

    def _backward_encoder(self, x, grad_ready_hook=None):
        alias = getattr(self, "_x_alias", None)
        if alias is not None and alias[0]._version != alias[1]:
            raise RuntimeError("the scalogram batch was modified in place between the forward and the backward pass: the engine reads it in "
                               "place (no copy) and recomputes its first block from it — pass a tensor you do not write to, or a clone")
        if self.top_grid is not None:
            d_out = self.blocks[-1].d_out
            d_out.t.zero_()
            self._row0(d_out).copy_(self.d_top_grid.t.view(self.B, self.T, self.E))
        for b in reversed(self.blocks):
            b.backward()

    # ------------------------------------------------------------------ Wasserstein gradient penalty
    def loss_and_grads(self, x, softplus: bool, regularization: float, all_timesteps: bool = False, grad_ready_hook=None,
                       global_negatives=None, after_loss=None, gradient_penalty=None):
        if gradient_penalty is None:
            return super().loss_and_grads(x, softplus, regularization, all_timesteps, grad_ready_hook, global_negatives, after_loss)
        return self._gp_step(x, softplus, regularization, all_timesteps, float(gradient_penalty), global_negatives, after_loss)

    def _gp_softplus_buffers(self, all_timesteps):
        key = bool(all_timesteps)
        if getattr(self, "_gp_sp", None) is not None and self._gp_sp.key == key:
            return
        B, K = self.B, self.K
        n = (B * K) * ((B * K + 7) // 8 * 8) if all_timesteps else K * B * self.ldS
        new = lambda m: torch.zeros(m, device=self.device, dtype=torch.float32)
        self._gp_sp = SimpleNamespace(key=key, W1=new(n), W1T=new(n), W2=new(n), W2T=new(n), St1=new(n), St2=new(n),
                                      pred_a=torch.zeros_like(self.pred), pred_b=torch.zeros_like(self.pred),
                                      top_a=torch.zeros_like(self.dact[-1]), top_b=torch.zeros_like(self.dact[-1]))
        # cpc_gp_score_coeff writes float32 coefficients; the contractions behind them take GEMM operands in the storage dtype
        sp = self._gp_sp
        for name in ("W1", "W1T", "W2", "W2T"):
            setattr(sp, name + "s", getattr(sp, name) if self.dt == torch.float32 else torch.zeros(n, device=self.device, dtype=self.dt))

    def _gp_coeff(self, *names):
        sp = self._gp_sp
        for name in names:
            dst, src = getattr(sp, name + "s"), getattr(sp, name)
            if dst is not src:
                dst.copy_(src)

    def _gp_step(self, x, softplus, regularization, all_timesteps, factor, global_negatives, after_loss):
        """One train step with the Wasserstein gradient penalty (contrastive_estimation_training.py:141-161):
        loss = InfoNCE + regulariser + factor * mean((|d sum(scores) / d x|_2 over channels - 1)^2), x the scalogram batch.
        d penalty / d theta = d/d theta of the directional derivative of sum(scores) along v = d penalty / d (input gradient):
          pass 1  backward of the summed scores (adjoints delta of every activation, g = delta at the input);
          tangent pass of v (bias-free convolutions, BatchNorm tangents, primal ReLU / pooling selections);
          penalty weight gradients (tangent input) x delta per convolution, BatchNorm second-order terms;
          pass 3  the ordinary backward of the real loss, seeded additionally with sum(tangent targets) on the predictions and
                  sum(tangent predictions) on the targets, the BatchNorm terms joining in on the way down.
        DESIGN.md section 8 has the derivation; tests compare with the reference's own double backward (tests/golden/scalogram_model_gp)."""
        if not self.gp_capable:
            raise RuntimeError("this engine was built without gradient-penalty support (model.gradient_penalty_engine = True first)")
        gn = global_negatives
        world = gn.world if gn is not None else 1

        def over_ranks(t):
            """Sum of a (small) seed tensor over the ranks: with global negatives the summed scores S run over the GLOBAL batch, so the
            constant seeds of pass 1 / the tangent seeds of pass 3 are sums over every rank's targets / predictions."""
            if gn is None:
                return t
            f = t.float().contiguous()
            gn.dist.all_reduce(f)
            return f.to(t.dtype)
        if self.dt != torch.float32:
            # bf16 storage: tangent grids and penalty weight-gradient GEMMs in bf16 like the primal ones, the float32 first stage kept.
            # Grid-based context networks (ConvolutionalArModel / ScalogramResidualEncoder contexts: the reference's e22-e26 and its
            # script default e29) run in bf16 too; a GRU or attention context computes in float32 inside the bf16 engine
            # (engine.Float32Context: their second-order sweeps are float32 kernels).
            if not isinstance(self.ctx, (ConvArGridContext, ResNetArContext, Float32Context)):
                raise NotImplementedError(f"gradient penalty with bf16 storage: no route for {type(self.ctx).__name__} "
                                          f"(compute_dtype='fp32' runs it)")
        if not hasattr(self.ctx, "tangent"):
            raise NotImplementedError(f"no gradient-penalty tangent pass for {type(self.ctx).__name__}")
        model, code = self.model, self.code
        B, E, K, V, T, H = self.B, self.E, self.K, self.V, self.T, self.H
        Ltop = self.geo.alloc[-1]
        top = self.act[-1].view(B, Ltop, E)
        dtop = self.dact[-1].view(B, Ltop, E)
        if getattr(self, "gp_flat", None) is None:
            self.gp_flat = torch.zeros_like(model._flat_grad)
            self.gp_grad = {n: self.gp_flat[model._offset[n]:model._offset[n] + p_.numel()].view(p_.shape) for n, p_ in model.named_parameters()}
            self.gp_partial = torch.zeros(256, device=self.device, dtype=torch.float32)
            self.pred_t = torch.zeros_like(self.pred)
        self.forward(x)                                                                   # pass 0
        # ---- pass 1: adjoints of S = sum of the scores the loss is built from (all (b,k,b',k') pairs, or the equal-step ones)
        pred3 = self.pred.view(B, K, E)
        tg = top[:, T - K:T, :]
        if all_timesteps:
            seed_p = over_ranks(tg.sum((0, 1), keepdim=True)).expand(B, K, E)
            seed_t = over_ranks(pred3.sum((0, 1), keepdim=True)).expand(B, K, E)
        else:
            seed_p = over_ranks(tg.sum(0, keepdim=True)).expand(B, K, E)
            seed_t = over_ranks(pred3.sum(0, keepdim=True)).expand(B, K, E)
        self.dact[-1].zero_()
        if softplus and gn is not None:
            # softplus scores over the GLOBAL score matrix (engine.GlobalNegatives.gp_softplus_*)
            sd_p, sd_t = gn.gp_softplus_seed(all_timesteps)
            self.dpred.copy_(sd_p)
            dtop[:, T - K:T, :].copy_(sd_t)
        elif softplus:
            # softplus scores: the summed scores are sum softplus(s), the seeds carry W1 = sigmoid(s) (cpc_gp_score_coeff)
            self._gp_softplus_buffers(all_timesteps)
            sp = self._gp_sp
            if all_timesteps:
                R = B * K
                ldR = (R + 7) // 8 * 8
                self.score_gemm_all()
                _hip.call("cpc_gp_score_coeff", _hip.ptr(self.S_all), None, None, _hip.ptr(sp.W1), _hip.ptr(sp.W1T), 1, R, R, ldR, ldR, 0)
                self._gp_coeff("W1", "W1T")
                self._score_grads_all(sp.W1s, sp.W1Ts, self.pred, self.act[-1], self.dpred, self.dact[-1])
            else:
                self.score_gemm()
                _hip.call("cpc_gp_score_coeff", _hip.ptr(self.S), None, None, _hip.ptr(sp.W1), _hip.ptr(sp.W1T), K, B, B, self.ldS, self.ldS, 0)
                self._gp_coeff("W1", "W1T")
                self._score_grads(sp.W1s, sp.W1Ts, self.pred, self.act[-1], self.dpred, self.dact[-1])
        else:
            self.dpred.view(B, K, E).copy_(seed_p)
            dtop[:, T - K:T, :].copy_(seed_t)
        self._gp_phase = 1
        self.backward(x)
        # ---- direction v = d penalty / d g and the penalty's value
        x_t = _twin(self, self.x_grid)
        npix = self.x_grid.B * self.x_grid.W * self.x_grid.H
        nb = min(256, max(1, npix // 256))
        self.gp_partial.zero_()
        # (global negatives: the penalty is the mean over the GLOBAL batch, 1 / world of this rank's mean)
        _hip.call("cpc_gp_direction", self.d_x.ptr(), x_t.ptr(), C.c_longlong(npix), self.x_grid.C, factor / world, _hip.ptr(self.gp_partial), nb)
        # ---- tangent pass
        self._gp_phase = 2
        for b in self.blocks:
            b.tangent()
        top_t = self._top_tangent()
        ct, coff, cstride = self.ctx.tangent(top_t)
        _hip.gemm_nt(_hip.ptr(ct, coff), _hip.ptr(self.w_p), _hip.ptr(self.pred_t), B, K * E, H, H, H, K * E, code, a_rpi=1, a_item=cstride)
        # ---- penalty parts of the parameter gradients (pass-1 adjoints are still in the gradient grids)
        self.gp_flat.zero_()
        for b in self.blocks:
            b.gp_grads(self.gp_grad)
        self.ctx.gp_grads(self.gp_grad)
        # predictor: W_bar += (pass-1 adjoint of the predictions)^T x (tangent of c)
        _hip.gemm_tn(_hip.ptr(self.dpred), _hip.ptr(ct, coff), _hip.ptr(self.gp_grad["prediction_model.weight"]), B, K * E, H,
                     K * E, H, H, code, b_rpi=1, b_item=cstride, flags=_hip.GEMM_OUT_F32)
        if self.use_aux:       # the slab reductions of the penalty's weight gradients (side stream) read buffers pass 3 reuses
            torch.cuda.current_stream().wait_stream(self.aux)
        # ---- pass 3: the real loss, plus the penalty's seeds on the primal stream
        top_t3 = top_t.view(B, Ltop, E)
        tg_t, pred_t3 = top_t3[:, T - K:T, :], self.pred_t.view(B, K, E)
        if softplus and gn is not None:
            add_p, add_t = gn.gp_softplus_second(self.pred_t, top_t, all_timesteps)
        elif softplus:
            # nu_p = W1 (tangent targets) + W2 targets,  nu_t = W1^T (tangent predictions) + W2^T predictions,
            # W2 = softplus''(s) * (tangent of s),  tangent of s = (tangent predictions) targets^T + predictions (tangent targets)^T
            sp = self._gp_sp
            sp.top_a.zero_(), sp.top_b.zero_()
            if all_timesteps:
                R = B * K
                ldR = (R + 7) // 8 * 8
                self.score_gemm_all(pred=self.pred_t, out=sp.St1)
                self.score_gemm_all(top=top_t, out=sp.St2)
                _hip.call("cpc_gp_score_coeff", _hip.ptr(self.S_all), _hip.ptr(sp.St1), _hip.ptr(sp.St2), _hip.ptr(sp.W2), _hip.ptr(sp.W2T),
                          1, R, R, ldR, ldR, 1)
                self._gp_coeff("W2", "W2T")
                self._score_grads_all(sp.W1s, sp.W1Ts, self.pred_t, top_t, sp.pred_a, sp.top_a)
                self._score_grads_all(sp.W2s, sp.W2Ts, self.pred, self.act[-1], sp.pred_b, sp.top_b)
            else:
                self.score_gemm(pred=self.pred_t, out=sp.St1)
                self.score_gemm(top=top_t, out=sp.St2)
                _hip.call("cpc_gp_score_coeff", _hip.ptr(self.S), _hip.ptr(sp.St1), _hip.ptr(sp.St2), _hip.ptr(sp.W2), _hip.ptr(sp.W2T),
                          K, B, B, self.ldS, self.ldS, 1)
                self._gp_coeff("W2", "W2T")
                self._score_grads(sp.W1s, sp.W1Ts, self.pred_t, top_t, sp.pred_a, sp.top_a)
                self._score_grads(sp.W2s, sp.W2Ts, self.pred, self.act[-1], sp.pred_b, sp.top_b)
            add_p = sp.pred_a.view(B, K, E) + sp.pred_b.view(B, K, E)
            add_t = sp.top_a.view(B, Ltop, E)[:, T - K:T, :] + sp.top_b.view(B, Ltop, E)[:, T - K:T, :]
        elif all_timesteps:
            add_p = over_ranks(tg_t.sum((0, 1), keepdim=True)).expand(B, K, E)
            add_t = over_ranks(pred_t3.sum((0, 1), keepdim=True)).expand(B, K, E)
        else:
            add_p = over_ranks(tg_t.sum(0, keepdim=True)).expand(B, K, E)
            add_t = over_ranks(pred_t3.sum(0, keepdim=True)).expand(B, K, E)
        add_p, add_t = add_p.clone(), add_t.clone()
        self.dact[-1].zero_()
        if gn is not None:
            gn.forward_backward(softplus, regularization, all_timesteps)
        elif all_timesteps:
            self.nce_all_forward_backward(softplus, regularization)
        else:
            self.nce_forward_backward(softplus, regularization)
        if after_loss is not None:
            after_loss(self.nce_out)
        self.dpred.view(B, K, E).add_(add_p)
        dtop[:, T - K:T, :].add_(add_t)
        self._gp_phase = 3
        self.backward(x)
        self._gp_phase = 0
        model._flat_grad.add_(self.gp_flat)
        pen = (self.gp_partial.sum() * (factor / (npix * world))).reshape(1)
        if gn is not None:
            gn.dist.all_reduce(pen)
        self.nce_out[0:1].add_(pen)
        return self.nce_out


# =====================================================================================================================
# ConvolutionalArModel with BatchNorm1d and / or residual branches (audio_model.py:80-161) on the same grid machinery: a
# sequence [B][L][C] is a grid with W = 1, Conv1d(k) a tall (k,1) kernel, MaxPool1d(ceil) the p x p pooling restricted to
# one column, BatchNorm1d the same per-channel statistics.
class _ArBlock:
    def __init__(self, eng, idx, blk, gin: Grid, kernel, stride, pool, cin, cout, batch_norm, residual, bias):
        self.eng, self.idx, self.gin, self.pool, self.stride = eng, idx, gin, pool, stride
        dev, dt = eng.device, eng.dt
        pre = f"autoregressive_model.module_list.{idx}."
        ci = 1 if pool > 1 else 0
        self.xp = Grid(gin.B, 1, _ceil_div(gin.H, pool), cin, dev, dt) if pool > 1 else gin
        conv_mod = SimpleNamespace(in_channels=cin, out_channels=cout, kernel_size=(kernel, 1), stride=(stride, 1), padding=(0, 0))
        self.conv = _Conv(eng, f"{pre}main_modules.{ci}.weight", f"{pre}main_modules.{ci}.bias" if bias else None, conv_mod, self.xp,
                          relu=not batch_norm)
        y0 = self.conv.y0
        if batch_norm:
            self.main = Grid(y0.B, 1, y0.H, cout, dev, dt)
            self.bn = _BatchNorm(eng, f"{pre}main_modules.{ci + 1}", blk.main_modules[ci + 1], y0, self.main)
            self.conv.bn_after = self.bn
        else:
            self.main, self.bn = y0, None
        self.residual = residual
        self.res_conv = self.rp = None
        if residual:
            rpool = pool * stride
            if rpool > 1:
                self.rp = self.xp if stride == 1 else Grid(gin.B, 1, _ceil_div(gin.H, rpool), cin, dev, dt)
            src = self.rp if self.rp is not None else gin
            if cin != cout:
                ri = 1 if rpool > 1 else 0
                rmod = SimpleNamespace(in_channels=cin, out_channels=cout, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0))
                self.res_conv = _Conv(eng, f"{pre}residual_modules.{ri}.weight", f"{pre}residual_modules.{ri}.bias", rmod, src)
                self.res = self.res_conv.y0
            else:
                self.res = src
            self.oh = self.res.H - self.main.H           # right-aligned crop x[:, :, -len:]  (audio_model.py:133)
            if self.oh < 0:
                raise ValueError(f"ConvolutionalArBlock {idx}: residual branch shorter than the main branch")
            self.out = Grid(y0.B, 1, self.main.H, cout, dev, dt)
        else:
            self.out = self.main
        self.slab = max([c.slab for c in (self.conv, self.res_conv) if c is not None] + ([self.bn.slab] if self.bn else [0]))

    def allocate_grads(self, d_in: Grid):
        dev = self.eng.device
        self.d_in = d_in
        self.d_out = self.out.like(dev)
        self.d_xp = self.xp.like(dev) if self.xp is not self.gin else d_in
        if self.residual:
            self.d_main = self.main.like(dev, guard_rows=self.conv.kh + 16)
            self.d_res = self.res.like(dev)
            if self.res_conv is not None:
                self.res_conv.dy0 = self.d_res
            if self.rp is not None and self.rp is not self.xp:
                self.d_rp = self.rp.like(dev)
        else:
            self.d_main = self.d_out
        if self.bn is not None:
            self.bn.dy0 = self.conv.y0.like(dev, guard_rows=self.conv.kh + 16)
            self.conv.dy0 = self.bn.dy0
        else:
            self.conv.dy0 = self.d_main

    def convs(self):
        return [c for c in (self.conv, self.res_conv) if c is not None]

    def prepare(self):
        for c in self.convs():
            c.prepare()

    def forward(self):
        code = self.eng.code
        if self.pool > 1:
            _hip.call("cpc_maxpool2d_fwd", self.gin.ptr(), _desc(self.gin, self.gin.desc), self.xp.ptr(), _desc(self.xp, self.xp.desc),
                      self.pool, 0, code)
        self.conv.forward()
        # BatchNorm1d + ReLU and the residual add in one pass where nothing else reads the normalised branch (see _Block.forward)
        fuse = (self.bn is not None and self.residual and self.bn.abits is not None and not getattr(self.eng, "gp_capable", False) and
                self.main.C % 8 == 0 and os.environ.get("CPC_BN_RESIDUAL", "1") != "0")
        if self.bn is not None:
            self.bn.forward(apply=not fuse)
        if self.residual:
            if self.rp is not None and self.rp is not self.xp:
                _hip.call("cpc_maxpool2d_fwd", self.gin.ptr(), _desc(self.gin, self.gin.desc), self.rp.ptr(), _desc(self.rp, self.rp.desc),
                          self.pool * self.stride, 0, code)
            if self.res_conv is not None:
                self.res_conv.forward()
            self._fused_bwd = fuse and os.environ.get("CPC_BN_RESIDUAL_BWD", "1") != "0"
            if fuse:
                self.bn.apply_residual(self.res, self.out, self.oh, 0, 0, 0)
            else:
                _hip.call("cpc_residual_add", self.main.ptr(), _desc(self.main, self.main.desc), self.res.ptr(), _desc(self.res, self.res.desc),
                          self.out.ptr(), _desc(self.out, self.out.desc), self.oh, 0, 0, 0, code)

    def tangent(self):
        e, code = self.eng, self.eng.code
        T = lambda grid: _twin(e, grid)
        if self.pool > 1:
            _hip.call("cpc_maxpool2d_select", self.gin.ptr(), T(self.gin).ptr(), _desc(self.gin, self.gin.desc), T(self.xp).ptr(),
                      _desc(self.xp, self.xp.desc), self.pool, 0, code)
        self.conv.forward(tangent=True)
        if self.bn is not None:
            self.bn.tangent()
        if self.residual:
            if self.rp is not None and self.rp is not self.xp:
                _hip.call("cpc_maxpool2d_select", self.gin.ptr(), T(self.gin).ptr(), _desc(self.gin, self.gin.desc), T(self.rp).ptr(),
                          _desc(self.rp, self.rp.desc), self.pool * self.stride, 0, code)
            if self.res_conv is not None:
                self.res_conv.forward(tangent=True)
            _hip.call("cpc_residual_add", T(self.main).ptr(), _desc(self.main, self.main.desc), T(self.res).ptr(),
                      _desc(self.res, self.res.desc), T(self.out).ptr(), _desc(self.out, self.out.desc), self.oh, 0, 0, 0, code)

    def gp_grads(self, gp_grad):
        self.conv.gp_wgrad(gp_grad)
        if self.res_conv is not None:
            self.res_conv.gp_wgrad(gp_grad)
        if self.bn is not None:
            self.bn.gp_terms(self.d_main, gp_grad)

    def backward(self):
        code = self.eng.code
        fused_bwd = self.residual and getattr(self, "_fused_bwd", False)
        if self.residual and not fused_bwd:
            # (d_res was zeroed when it was allocated; the cropped add's backward overwrites the same interior every step)
            _hip.call("cpc_residual_add_bwd", self.d_out.ptr(), self.out.ptr(), _desc(self.out, self.out.desc), self.d_main.ptr(),
                      _desc(self.d_main, self.d_main.desc), self.d_res.ptr(), _desc(self.d_res, self.d_res.desc), self.oh, 0, 0, 0, code)
        if fused_bwd:      # the add's backward inside the BatchNorm's passes (no ReLU behind the add in a ConvolutionalArBlock: no output mask)
            self.bn.backward_res(self.d_out, None, self.d_res, self.oh, 0)
        elif self.bn is not None:
            self.bn.backward(self.d_main)
        else:
            _hip.call("cpc_relu_mask", self.d_main.ptr(), self.main.ptr(), self.d_main.rows * self.d_main.C, code)
        self.conv.backward(self.d_xp)                                  # writes the gradient of the (pooled) block input
        shared = self.residual and (self.rp is self.xp or self.rp is None)
        if self.residual and shared:                                   # residual source is the conv input itself
            if self.res_conv is not None:
                self.res_conv.backward(self.d_xp, accumulate=True)
            else:
                _accumulate(self.d_xp, self.d_res)
        if self.pool > 1:
            _hip.call("cpc_maxpool2d_bwd", self.gin.ptr(), self.d_in.ptr(), _desc(self.gin, self.gin.desc), self.d_xp.ptr(),
                      _desc(self.xp, self.xp.desc), self.pool, 0, code)
        if self.residual and not shared:                               # own pooling (stride > 1)
            if self.res_conv is not None:
                self.res_conv.backward(self.d_rp)
                src = self.d_rp
            else:
                src = self.d_res
            _hip.call("cpc_maxpool2d_bwd", self.gin.ptr(), self.d_in.ptr(), _desc(self.gin, self.gin.desc), src.ptr(),
                      _desc(self.rp, self.rp.desc), self.pool * self.stride, 1, code)


class ConvArGridContext:
    """ConvolutionalArModel with batch_norm and / or residual (e.g. ar_conv_architecture_2/3) as the context network."""

    ahead_ok = True       # prepare_weights is a pure function of the parameters (ScalogramCPCEngine.prepare_ahead)

    def __init__(self, eng, ar):
        self.eng, self.ar = eng, ar
        self.channels = list(ar.channel_count)
        if self.channels[0] != eng.E or self.channels[-1] != eng.H:
            raise ValueError("ConvolutionalArModel channel_count does not match enc_size / ar_size")
        if any(c % 8 for c in self.channels):
            raise NotImplementedError("ConvolutionalArModel channel counts must be multiples of 8")

    def allocate(self):
        e, ar = self.eng, self.ar
        self.x0 = Grid(e.B, 1, e.V, e.E, e.device, e.dt)
        self.blocks = []
        gin = self.x0
        for l in range(len(ar.kernel_sizes)):
            b = _ArBlock(e, l, ar.module_list[l], gin, ar.kernel_sizes[l], ar.strides[l], ar.poolings[l], self.channels[l],
                         self.channels[l + 1], ar.batch_norm, ar.residual, ar.module_list[l].main_modules[1 if ar.poolings[l] > 1 else 0].bias is not None)
            self.blocks.append(b)
            gin = b.out
        self.d_x0 = self.x0.like(e.device)
        d_in = self.d_x0
        for b in self.blocks:
            b.allocate_grads(d_in)
            d_in = b.d_out
        self.c32 = torch.empty(e.B, self.channels[-1], device=e.device, dtype=torch.float32)

    def slab_floats(self):
        return max(b.slab for b in self.blocks) + self.eng.colsum_blocks * max(self.channels)

    def prepare_weights(self):
        _prepare_convs(self.eng, self, [c for b in self.blocks for c in b.convs()])

    def _z_rows(self, buf):
        e = self.eng
        t0 = e.T - e.K - e.V
        return buf.view(e.B, e.geo.alloc[-1], e.E)[:, t0:t0 + e.V, :]

    def forward(self):
        e = self.eng
        self.x0.t.view(e.B, e.V, e.E).copy_(self._z_rows(e.act[-1]))
        for b in self.blocks:
            b.forward()

    def c_operand(self):
        out = self.blocks[-1].out
        return out.t, (out.top + out.H - 1) * out.C, out.Ha * out.C

    def c_float(self):
        out = self.blocks[-1].out
        self.c32.copy_(out.t.view(out.B, out.Ha, out.C)[:, out.top + out.H - 1, :])
        return self.c32

    # ---- Wasserstein gradient penalty: tangent of c, penalty parts of the parameter gradients
    def tangent(self, top_t):
        """``top_t``: tangent of the encoder's top buffer; returns (tensor, offset, item stride) of the tangent of c."""
        e = self.eng
        _twin(e, self.x0).t.view(e.B, e.V, e.E).copy_(self._z_rows(top_t))
        for b in self.blocks:
            b.tangent()
        out = self.blocks[-1].out
        return _twin(e, out).t, (out.top + out.H - 1) * out.C, out.Ha * out.C

    def gp_grads(self, gp_grad):
        for b in self.blocks:
            b.gp_grads(gp_grad)

    def backward(self, dc):
        e = self.eng
        last = self.blocks[-1]
        d = last.d_out
        # (only the last position carries a gradient; the rest of the grid was zeroed when it was allocated and nothing writes it)
        d.t.view(d.B, d.Ha, d.C)[:, d.top + d.H - 1, :] = dc.to(d.t.dtype)
        for b in reversed(self.blocks):
            b.backward()
        self._z_rows(e.dact[-1]).copy_(self.d_x0.t.view(e.B, e.V, e.E))


class ResNetArContext:
    """ScalogramResidualEncoder used as the autoregressive model (reference configs ar_resnet_architecture_1/2: blocks with
    (1,k) kernels, pooling_1 = 2 with ceil mode, batch norm, residual branches): z (B, E, V) is the grid [B][W = V][H = 1][E];
    the model takes the first remaining time step of the (B, C, T') result (audio_model.py:203-204)."""

    ahead_ok = True       # prepare_weights is a pure function of the parameters (ScalogramCPCEngine.prepare_ahead)

    def __init__(self, eng, ar):
        self.eng, self.ar = eng, ar
        if ar.phase:
            raise ValueError("a ScalogramResidualEncoder used as context network must have phase=False")

    def allocate(self):
        e, ar = self.eng, self.ar
        blocks = list(ar.blocks)
        if blocks[0].cfg['in_channels'] != e.E or blocks[-1].cfg['out_channels'] != e.H:
            raise ValueError("context network block channels do not match enc_size / ar_size")
        if blocks[0].cfg['top_padding_1']:
            raise NotImplementedError("top_padding_1 on the first block")
        self.x0 = Grid(e.B, e.V, 1, e.E, e.device, e.dt)
        self.blocks, gin = [], self.x0
        for i, blk in enumerate(blocks):
            last = i == len(blocks) - 1
            next_top = 0 if last else (blocks[i + 1].cfg['top_padding_1'] or 0)
            b = _Block(e, i, blk, gin, False, last, next_top, prefix="autoregressive_model.", need_input_grad=True)
            self.blocks.append(b)
            gin = b.out
        out = self.blocks[-1].out
        if out.H != 1:
            raise NotImplementedError("the context network must end with one row")
        self.d_x0 = self.x0.like(e.device)
        d_in = self.d_x0
        for b in self.blocks:
            b.allocate_grads(d_in)
            d_in = b.d_out
        self.c32 = torch.empty(e.B, out.C, device=e.device, dtype=torch.float32)

    def slab_floats(self):
        return max(b.slab for b in self.blocks) + self.eng.colsum_blocks * max(max(b.a_a.C, b.conv_b.cout) for b in self.blocks)

    def prepare_weights(self):
        _prepare_convs(self.eng, self, [c for b in self.blocks for c in b.convs()])

    def _z_rows(self, buf):
        e = self.eng
        t0 = e.T - e.K - e.V
        return buf.view(e.B, e.geo.alloc[-1], e.E)[:, t0:t0 + e.V, :]

    def forward(self):
        e = self.eng
        self.x0.t.view(e.B, e.V, e.E).copy_(self._z_rows(e.act[-1]))
        for b in self.blocks:
            b.forward()

    def _first_step(self, grid):
        return grid.t.view(grid.B, grid.W, grid.Ha, grid.C)[:, 0, grid.top, :]

    def c_operand(self):
        out = self.blocks[-1].out
        return out.t, out.top * out.C, out.W * out.Ha * out.C

    # ---- Wasserstein gradient penalty (see ConvArGridContext)
    def tangent(self, top_t):
        e = self.eng
        _twin(e, self.x0).t.view(e.B, e.V, e.E).copy_(self._z_rows(top_t))
        for b in self.blocks:
            b.tangent()
        out = self.blocks[-1].out
        return _twin(e, out).t, out.top * out.C, out.W * out.Ha * out.C

    def gp_grads(self, gp_grad):
        for b in self.blocks:
            b.gp_grads(gp_grad)

    def c_float(self):
        self.c32.copy_(self._first_step(self.blocks[-1].out))
        return self.c32

    def backward(self, dc):
        e = self.eng
        d = self.blocks[-1].d_out
        # (only the first time step carries a gradient; the rest of the grid was zeroed when it was allocated and nothing writes it)
        self._first_step(d).copy_(dc)
        for b in reversed(self.blocks):
            b.backward()
        self._z_rows(e.dact[-1]).copy_(self.d_x0.t.view(e.B, e.V, e.E))
