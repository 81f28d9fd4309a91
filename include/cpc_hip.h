/* cpc_hip.h — C ABI of the MI355X-native CPC-audio train-step library (libcpc_hip.so).
 *
 * The reference (vincentherrmann/constrastive-predictive-coding-audio) is pure Python/PyTorch and has no FFI of its
 * own; the seam this library sits behind is the ATen call each reference line makes.  Every entry point below names
 * the reference statement(s) it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, a hipStream_t passed as void*; no torch types.
 *   - every call is stream-ordered and asynchronous, never allocates, never synchronises, never throws;
 *     returns 0 on success, -22 (EINVAL) for unsupported shapes/arguments, -5 (EIO) if the launch failed.
 *   - dtype: CPC_F32 (0) = exact-f32 parity mode (v_mfma_f32_16x16x4_f32), CPC_BF16 (1) = bf16 storage with f32
 *     accumulation (v_mfma_f32_16x16x32_bf16).  "T" below means that storage type.
 *   - activations are channels-last: act[item][position][channel], positions padded per item to L_alloc rows
 *     (pad rows hold zeros); see DESIGN.md "Data layout in HBM".
 */
#ifndef CPC_HIP_H
#define CPC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CPC_F32 0
#define CPC_BF16 1

#define CPC_GEMM_RELU 1      /* NT epilogue: relu */
#define CPC_GEMM_OUT_F32 2   /* store the result as f32 even when dtype is bf16 */
#define CPC_GEMM_TN_NO_TR 4  /* TN/bf16 only: scalar LDS reads instead of ds_read_b64_tr_b16 (A/B check) */
#define CPC_GEMM_FORCE_GENERIC 8 /* use the generic (any-shape) kernel even where the double-buffered fast path applies */
#define CPC_GEMM_NARROW_EPI 32   /* NT/bf16: per-lane 8-byte stores instead of the LDS-staged full-row epilogue (A/B check) */
#define CPC_GEMM_NO_DMA 64       /* NT fast path: register-staged global->LDS copies instead of LDS-DMA (A/B check) */
#define CPC_GEMM_SKIP_PAD_ROWS 128 /* NT: rows with (m % c_rpi) >= c_valid are left untouched instead of zeroed */
#define CPC_GEMM_LINEAR_K 256     /* NT: visit K in storage order even for overlapped-row A operands (A/B check; default for lda < K,
                                    K % lda == 0 is tap-innermost: identical sums in a different order, each input byte fetched once) */
#define CPC_GEMM_NO_PERS 512      /* NT/bf16: LDS-staged epilogue instead of the register epilogue (A/B check) */
#define CPC_GEMM_DIRECT_MASK 1024 /* NT/bf16: register epilogue also for launches with a mask (A/B check; default: LDS-staged there) */
#define CPC_GEMM_KRANGE_EXACT 2048 /* with k_ranges: the ranges cut out pieces that are NOT zero, so no tile may straddle two range indices — the
                                    * launch returns CPC_EINVAL unless a_rpi * max(a_rpi2, 1) is a multiple of the tile height (128 / 256) the
                                    * launcher picks.  Without the flag a tile runs the union of its rows' ranges (correct for known-zero cuts only). */
#define CPC_GEMM_SMALL_TILE 16   /* keep the 128x128 tile where the 256x256 one would be chosen (A/B check) */
#define CPC_GEMM_BIG_TILE 0x100000 /* NT/bf16: the 256x256 tile also where fewer than 200 of them exist (M >= 256): a launch that is ONE nearly
                                    * full round of such tiles (layer 2's target rows at the headline size: 196) runs 0.11 instead of 0.135 ms */

/* 8 (round 4): the fused all-timesteps score path (cpc_score_lse, cpc_nce_lse_merge, cpc_nce_fused_grad(_blocks), cpc_nce_fused_finalize); cpc_reduce_conv_w2d; cpc_accumulate; the row-range launches cpc_conv1_fwd_rows, cpc_conv_dgrad_rows, cpc_conv_dgrad_conv1_rows, cpc_conv1_fused_reduce_tiles.
 * 7 (round 3, second half): cpc_gemm_nt_args grew the second row level (a_rpi2 / c_rpi2), k_ranges and the gathered-row taps (k_taps,
 * k_tap_stride, k_tap_stride_a); new entry points cpc_conv_w_prep_group / _plan / _batch, cpc_bn_apply_residual, cpc_bn_bwd_reduce_res / _apply_res, cpc_stem_residual_bn_add,
 * cpc_stem_residual_wgrad_bits; cpc_gemm_tn_args grew a_rpi2 / a_item2.
 * 6: the stem kernels (cpc_stem_*), sign-bit BatchNorm passes (cpc_bn_*_bits), cpc_gru_fwd_h0.
 * 5 since the gradient-penalty entry points (cpc_gru_gp_*, cpc_ln_tangent / cpc_ln_gp, cpc_attn_tangent / cpc_attn_gp,
 * cpc_gp_score_coeff) were added; 4: per-tile column sums of the data gradients; 3: sign-bit masks; 2: over-read contract. */
int cpc_abi_version(void);

/* Row addressing used by both GEMMs: row m of an operand starts at element
 *   rpi == 0 :  m * ld
 *   rpi  > 0 :  (m / rpi) * item + (m % rpi) * ld        ("items" of rpi rows, e.g. a window of frames per clip)   */

/* C[m][n] = epi( sum_k A[m][k] * Bt[n][k] ), epi = (+bias[n]) -> (relu) -> (mask[m][n] > 0 ? . : 0) -> (pad row -> 0).
 * Replaces, in channels-last layout: F.conv1d forward of AudioEncoder layers 2..5 (audio_model.py:38-41, A rows
 * overlap: lda = stride*C_in, K = kernel*C_in), their data gradient (autograd of the same lines), nn.GRUCell's
 * input projection for all steps at once (audio_model.py:72), prediction_model (audio_model.py:208) and the
 * tensordot of the score functions restricted to equal steps (contrastive_estimation_training.py:13-14, :20-21). */
typedef struct cpc_gemm_nt_args {
    const void* A;  const void* Bt;  void* C;
    const float* bias;            /* [N] f32 or NULL */
    const void* mask;             /* T, addressed like C, or NULL */
    int M, N, K;
    long long lda, ldb, ldc;
    int a_rpi; long long a_item;
    int b_rpi; long long b_item;
    int c_rpi; long long c_item; int c_valid;   /* rows with (m % c_rpi) >= c_valid are stored as zeros */
    long long a_batch, b_batch, c_batch; int batch;
    int flags; int dtype;
    /* OVER-READ CONTRACT.  Row m of A is read as the K elements starting at its row address, whatever lda is: with
     * overlapped rows (lda < K, the strided-convolution view) the last row therefore reads (K - lda) elements beyond the end
     * of an [M][lda] array, and rows are clamped to M - 1 but never shortened.  ALL of [A, A + last_row_offset + K) (and the
     * same for Bt) must be readable device memory: an allocation that ends exactly at M*lda elements may end at the end of a
     * mapped segment, and the over-read then faults (this happened once).  a_extent / b_extent:
     * number of elements readable from A / Bt (all batches); when > 0 the call returns CPC_EINVAL if any row would end beyond
     * it; 0 = not checked, the caller vouches. */
    long long a_extent, b_extent;
    /* Second row level and K ranges (LDS-DMA kernels, K a multiple of 64 bf16 / 32 f32 elements; CPC_EINVAL elsewhere): with
     * a_rpi2 != 0 row m of A sits at (m / (a_rpi a_rpi2)) a_item2 + ((m / a_rpi) % a_rpi2) a_item + (m % a_rpi) lda, likewise C and the
     * mask with c_rpi2 / c_item2: the rows of a grid ordered (band of rows, column, row within the band).  k_ranges: device int
     * pairs [lo, hi), hi > lo, in stages of 64 bf16 / 32 f32 elements: the part of the K axis that is not known to be zero for the
     * rows of band i = m / (a_rpi a_rpi2) (a_rpi2 == 0: of item m / a_rpi); a tile runs the union of the ranges of its rows.  The
     * extent check above does not know the second level: pass a_extent = 0 with a_rpi2.  All zero = off. */
    int a_rpi2; long long a_item2;
    int c_rpi2; long long c_item2;
    const int* k_ranges;
    /* Gathered rows (same kernels): with k_taps > 1 the K axis is k_taps pieces of k_tap_stride = K / k_taps elements (a multiple of 64 bf16 /
     * 32 f32), dense in Bt, while piece j of a row of A starts j * k_tap_stride_a elements after the row address — the window of an nn.Conv2d
     * read straight from a channels-last grid (piece = kh rows x C channels of one kernel column; k_tap_stride_a = one grid column), no
     * im2col matrix.  The over-read contract applies to every piece.  k_taps = 0: off (overlapped rows with lda < K are detected by the launcher
     * and visited tap-innermost on their own). */
    int k_taps; long long k_tap_stride; long long k_tap_stride_a;
} cpc_gemm_nt_args;
int cpc_gemm_nt(const cpc_gemm_nt_args* args, void* stream);

/* C[i][j] = sum_m A[m][i] * B[m][j]  (reduction over the row index of both operands), optionally split over m into
 * nsplit f32 slabs C + s*slab_stride that cpc_reduce_slabs sums deterministically.
 * Replaces the weight gradients autograd computes for conv1d / GRUCell / Linear (loss.backward(),
 * contrastive_estimation_training.py:161) and the two score-gradient contractions. */
typedef struct cpc_gemm_tn_args {
    const void* A;  const void* B;  void* C;
    int M, I, J;
    long long lda, ldb, ldc;
    int a_rpi; long long a_item;
    int b_rpi; long long b_item;
    long long a_batch, b_batch, c_batch; int batch;
    int nsplit; int m_chunk; long long slab_stride;
    int flags; int dtype;
    int c_rpi; long long c_item;          /* output row i at the item address (nsplit == 1 only); 0 = plain i*ldc */
    /* second row level of A (bf16 LDS-DMA kernel; CPC_EINVAL elsewhere): row m at (m / (a_rpi a_rpi2)) a_item2 + ((m / a_rpi) % a_rpi2) a_item
     * + (m % a_rpi) lda — with a_batch stepping over the kernel columns, the windows of a strided nn.Conv2d read straight from a channels-last grid
     * (rows = clip, output column, output row): the weight gradient without an im2col matrix.  0 = one level. */
    int a_rpi2; long long a_item2;
} cpc_gemm_tn_args;
int cpc_gemm_tn(const cpc_gemm_tn_args* args, void* stream);

/* out[j*s_j + (i / cdiv)*s_hi + (i % cdiv)*s_lo] = sum_z slabs[z*slab_stride + i*J + j]  (f32; fixed summation order). */
int cpc_reduce_slabs(const float* slabs, float* out, int I, int J, int nslab, long long slab_stride, int cdiv,
                     long long s_j, long long s_hi, long long s_lo, void* stream);

/* Conv weight gradient slabs (cpc_conv_wgrad) -> reference layout: out[co][c][j] = sum_z slabs[z][j*cin + c][co]. */
int cpc_reduce_conv_w(const float* slabs, float* out, int cin, int cout, int kw, int nslab, long long slab_stride,
                      void* stream);

/* Weight gradient of an nn.Conv2d in the reference's [cout][cin][kh][kw] layout from split-K slabs of the grid GEMMs (ABI 8):
 *   out[((co cin + c) kh + dh) kw + dw] = sum_z sum_{g < G} slabs[z slab_stride + dw s_dw + (dh + g) s_dh + c s_c + g s_g + co]
 * G = 1: slabs [kw][kh][cin][cout] (windows gathered from the grid, one GEMM batch entry per kernel column); G > 1: the row-grouped tall (k,1)
 * kernels, slab [(r, c)][(g, co)], whose G diagonals r = dh + g make up dW (scalogram_model.py:392-417's nn.Conv2d autograd). */
int cpc_reduce_conv_w2d(const float* slabs, float* out, int nslab, long long slab_stride, int cout, int cin, int kh, int kw, long long s_dw,
                        long long s_dh, long long s_c, int G, long long s_g, void* stream);

/* slabs[blk][n] = partial column sums of X[M][N] (T) — bias gradients; reduce with cpc_reduce_slabs(I=1). */
int cpc_colsum(const void* X, float* slabs, int M, int N, long long ldx, int nblocks, int dtype, void* stream);

/* AudioEncoder layer 1 (C_in = 1): y[b][t][co] = act(bias[co] + sum_j x[b][t*stride+j] * w[co][j]), act = relu when
 * relu != 0 (audio_model.py:38-41 for l == 0: the last layer of the stack has no activation, so a one-layer encoder passes
 * relu = 0).  x: f32 [B][ldx]; w: f32 [C][kw] (reference layout); y: T [B][L_alloc][C]. */
/* y_bits (may be NULL; C = 512 only, 16-byte aligned): the sign-bit mask of y, see cpc_sign_bits — written from the f32 values
 * before they are rounded to the storage type (what the reference's ReLU mask is taken from). */
int cpc_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int C, int stride, int kw,
                  long long ldx, int L_valid, int L_alloc, int relu, int dtype, void* y_bits, void* stream);
/* The same for positions [row_lo, row_hi) of every item only (0 <= row_lo < row_hi <= L_alloc; ABI 8): the train step computes the rows
 * the context network needs first and the rows only the targets need beside the GRU recurrence (engine.CPCEngine.encoder_forward). */
int cpc_conv1_fwd_rows(const float* x, const float* w, const float* bias, void* y, int B, int C, int stride, int kw, long long ldx,
                       int L_valid, int L_alloc, int relu, int dtype, void* y_bits, int row_lo, int row_hi, void* stream);
/* SIGN-BIT MASKS: bits[i] bit e = x[8 i + e] > 0 (for bf16: of the stored value), n elements (a multiple of 32; x 16-byte, bits
 * 4-byte aligned).  The ReLU-backward mask of a data gradient at 1/16 of the bytes of the activation it is taken from:
 * cpc_conv_dgrad / cpc_conv_dgrad_conv1 take it as x_act_bits in place of x_act — the same decision per element, identical
 * results — addressed by the element offset / 8, so a bits buffer mirrors its activation buffer (guard rows included). */
int cpc_sign_bits(const void* x, void* bits, long long n, int dtype, void* stream);
/* Weight/bias gradient of layer 1: slabs[nblk_b*nblk_t][(kw+1)][C] partials (row kw = bias); reduce with
 * cpc_reduce_slabs.  nblk_t blocks split the positions, nblk_b (<= B) blocks stride over the items. */
int cpc_conv1_bwd(const float* x, const void* dy, float* slabs, int B, int C, int stride, int kw, long long ldx,
                  int L_valid, int L_alloc, int nblk_t, int nblk_b, int dtype, void* stream);

/* Data gradient of encoder layer 2 FUSED with the weight/bias gradient of layer 1 (audio_model.py:38-39 differentiated, l = 1
 * and l = 0): the masked gradient tile of layer 1's output never goes to memory; each 256 x 256 tile leaves
 * slabs[tile][j][c] = sum_rows G[row][c] * x[b][t*stride1 + j] (j < kw1) and [kw1][c] = sum_rows G[row][c].
 * bf16 only, Cin % 256 == 0, kw1 <= 15; CPC_EINVAL otherwise (callers then use cpc_conv_dgrad + cpc_conv1_bwd).
 * x: f32 waveform (first sample the encoder reads), ldx samples per item.  Sizes of slabs / tmp (floats):
 * cpc_conv_dgrad_conv1_floats(..., what = 0 / 1).  cpc_conv1_fused_reduce sums the slabs in a fixed order into
 * dw [Cin][1][kw1] (reference layout) and db [Cin] (may be NULL). */
long long cpc_conv_dgrad_conv1_floats(int B, int Cin, int stride, int Lout_alloc, int kw1, int what);
int cpc_conv_dgrad_conv1(const void* dy, const void* w_dgrad, const void* x_act, const float* x, float* slabs, int B, int Cin,
                         int Cout, int kw, int stride, int Lout_alloc, long long ldx, int kw1, int stride1, int L1_valid,
                         long long dy_head, int dtype, const void* x_act_bits, void* stream);
int cpc_conv1_fused_reduce(const float* slabs, float* tmp, float* dw, float* db, int B, int Cin, int stride, int Lout_alloc,
                           int kw1, void* stream);

/* Conv1d (layers >= 2) in channels-last layout, expressed through the GEMMs above.
 *   fwd  : y[(b,t)][co] = relu?(bias + sum_{j,c} x[(b, t*stride + j)][c] * w[co][c][j])       audio_model.py:38-41
 *   dgrad: dx[(b,p)][c]  = (x_act > 0 ?) sum_{t,co: t*stride + j = p} dy[(b,t)][co] * w[co][c][j]
 *   wgrad: slabs of dw[(j,c)][co] = sum_{b,t} x[(b, t*stride+j)][c] * dy[(b,t)][co]
 * w_fwd / w_dgrad are the operand layouts cpc_conv_w_prep produces (either output pointer may be NULL: that layout is then
 * not written).  Lout_alloc * stride == Lin_alloc is required.
 * GUARD CONTRACT (the over-read contract of cpc_gemm_nt_args for these views): the forward and the weight gradient read
 * max(0, kw - stride) * Cin elements BEYOND the B * Lin_alloc * Cin elements of x (the window of the last row); the data
 * gradient reads (ceil(kw / stride) - 1) * Cout elements BEFORE dy.  The caller states how many elements are readable there
 * (x_tail after the end of x, dy_head in front of dy; zeros expected in dy_head, values in x_tail only ever meet pad rows);
 * CPC_EINVAL when that is less than the call needs.
 * x_act_bits: the sign-bit mask cpc_sign_bits makes of x_act (NULL: the mask is read from x_act).  Only the bf16 256 x 256-tile
 * kernel knows the format (>= 200 such tiles, stride * Cin a multiple of 256): CPC_EINVAL where the launch would take another
 * kernel, never a silent fall-back.  With x_act_bits, x_act may be NULL. */
int cpc_conv_fwd(const void* x, const void* w_fwd, const float* bias, void* y, int B, int Cin, int Cout, int kw,
                 int stride, int Lout_alloc, int Lout_valid, int relu, long long x_tail, int dtype, void* stream);
/* dx_colsum_slabs (may be NULL): f32 [ceil(B * Lout_alloc / 256)][stride * Cin] = per 256-row tile of the launch the column sums
 * of the dx it stores (cpc_conv_dgrad_colsum_floats floats); read as [tiles * stride][Cin] and summed by cpc_reduce_slabs this is the
 * bias gradient of the layer below — d loss / d bias = sum over (b, t) of dx — without another pass over dx.  bf16, 256 x 256-tile
 * kernel with a mask only (as x_act_bits: CPC_EINVAL elsewhere). */
long long cpc_conv_dgrad_colsum_floats(int B, int Cin, int stride, int Lout_alloc);
int cpc_conv_dgrad(const void* dy, const void* w_dgrad, const void* x_act, void* dx, int B, int Cin, int Cout, int kw,
                   int stride, int Lout_alloc, int Lin_valid, long long dy_head, int dtype, const void* x_act_bits,
                   float* dx_colsum_slabs, void* stream);
int cpc_conv_wgrad(const void* x, const void* dy, float* slabs, int B, int Cin, int Cout, int kw, int stride,
                   int Lout_alloc, int nsplit, long long x_tail, int dtype, void* stream);

/* The data-gradient launches above on rows [row_lo, row_hi) of every item only (ABI 8).  A data-gradient row q produces the input positions
 * q stride .. q stride + stride - 1 from the output-gradient rows q - D + 1 .. q (D = ceil(kw / stride)).  With kw = 2 stride the rows
 * [n_l + 1, L_alloc) of a data gradient need output-gradient rows >= n_l only: the part of the backward pass behind the TARGET frames runs
 * beside the GRU's backward recurrence (engine.CPCEngine._bwd_lane), the rest afterwards.  dx_colsum_slabs / the fused launch's slabs are
 * indexed by the launch's own 256-row tiles (ceil(B rows / 256) of them: pass an offset pointer to a second launch;
 * cpc_conv1_fused_reduce_tiles sums num_row_tiles of them). */
int cpc_conv_dgrad_rows(const void* dy, const void* w_dgrad, const void* x_act, void* dx, int B, int Cin, int Cout, int kw, int stride,
                        int Lout_alloc, long long dy_head, int dtype, const void* x_act_bits, float* dx_colsum_slabs, int row_lo, int row_hi,
                        void* stream);
int cpc_conv_dgrad_conv1_rows(const void* dy, const void* w_dgrad, const void* x_act, const float* x, float* slabs, int B, int Cin,
                              int Cout, int kw, int stride, int Lout_alloc, long long ldx, int kw1, int stride1, int L1_valid,
                              long long dy_head, int dtype, const void* x_act_bits, int row_lo, int row_hi, void* stream);
int cpc_conv1_fused_reduce_tiles(const float* slabs, float* tmp, float* dw, float* db, int num_row_tiles, int Cin, int stride, int kw1,
                                 void* stream);
int cpc_conv_w_prep(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int kw, int stride, int dtype,
                    void* stream);
/* cpc_conv_w_prep for a LIST of convolutions in one launch (a context network's eleven 5 x 512 x 512 kernels: eleven launches of 20 us in
 * front of every step otherwise).  The caller fills w / w_fwd / w_dgrad (either output may be NULL) / Cout / Cin / kw / stride of every job in
 * HOST memory; cpc_conv_w_prep_plan fills the launch geometry (D, tco, gx, first) and returns the grid size and dynamic LDS bytes; the table is
 * then copied to device memory once (the operand addresses are stable) and cpc_conv_w_prep_batch launched with it whenever the weights changed. */
typedef struct cpc_conv_prep_job {
    const float* w; void* w_fwd; void* w_dgrad;
    int Cout, Cin, kw, stride;
    int D, tco, gx, first;        /* filled by cpc_conv_w_prep_plan */
} cpc_conv_prep_job;
int cpc_conv_w_prep_plan(cpc_conv_prep_job* jobs, int njobs, int* total_blocks, int* lds_bytes);
int cpc_conv_w_prep_batch(const cpc_conv_prep_job* jobs_dev, int njobs, int total_blocks, int lds_bytes, int dtype, void* stream);
/* Operands of a tall (kh,1) nn.Conv2d (scalogram_model.py:393-412, the (64,1) / (30,1) / (15,1) second kernels of the residual blocks)
 * when G output rows are computed per GEMM row (C_out < 256: G = 256 / C_out rows side by side fill the 256-wide tile): G shifted copies
 * of the kernel in a window of Rw / Rd >= kh + G - 1 rows, zeros elsewhere.  w f32 [Cout][Cin][kh] (reference layout);
 *   w_fwd  T [G][Cout][Rw][Cin]: [dh][co][r][c] = w[co][c][r - dh];     w_dgrad T [G][Cin][Rd][Cout]: [dr][c][q][co] = w[co][c][kh-1-(q-dr)];
 *   bias_g f32 [G][Cout] = bias repeated (bias, bias_g may both be NULL). */
int cpc_conv_w_prep_group(const float* w, const float* bias, void* w_fwd, void* w_dgrad, float* bias_g, int Cout, int Cin, int kh, int G,
                          int Rw, int Rd, int dtype, void* stream);

/* ConvolutionalArBlock's MaxPool1d(pool, ceil_mode=True) over positions (audio_model.py:98-99) on channels-last
 * activations [B][L_alloc][C], forward and backward (gradient to the first maximal element of each window). */
int cpc_maxpool_fwd(const void* in, void* out, int B, int C, int pool, int Lin_valid, int Lin_alloc, int Lout_valid,
                    int Lout_alloc, int dtype, void* stream);
int cpc_maxpool_bwd(const void* in, const void* dout, void* din, int B, int C, int pool, int Lin_valid, int Lin_alloc,
                    int Lout_alloc, int dtype, void* stream);
/* dy[b*item_stride + row_off + c] = y[...] > 0 ? dc[b][c] : 0 — the gradient entering the last ReLU of
 * ConvolutionalArModel at the single position its forward returns (audio_model.py:161). */
int cpc_relu_row_bwd(const float* dc, const void* y, void* dy, int B, int C, long long item_stride, long long row_off,
                     int dtype, void* stream);

/* ---- AttentionModel as the context network (attention_model.py:38-82; layers: transformer.py:223-272) ----
 * Everything works on channels-last rows x[(b,t)][C], t < S.  The linear maps are cpc_gemm_nt / cpc_gemm_tn calls.
 *
 * PositionalEncoder.forward (attention_model.py:28-35): x0[(b,t)] = top[b*item_stride + t*C ...] * scale + pe[t]
 * (pe f32 [S][C]); its backward writes scale * (g1 + g2) (g2 may be NULL) into the same rows of dtop. */
int cpc_pe_scale_fwd(const void* top, const float* pe, void* x0, int B, int S, int C, long long item_stride, float scale,
                     int dtype, void* stream);
int cpc_pe_scale_bwd(const void* g1, const void* g2, void* dtop, int B, int S, int C, long long item_stride, float scale,
                     int dtype, void* stream);
/* Dropout (transformer.py:243-252 modules, active in train mode): every mask is a pure function of (seed, site, element
 * index) — factor 1/(1-p) where a 64-bit mix of the three is >= p*2^32, else 0 — so the backward entry points regenerate
 * it from the same (drop_p, seed, site) instead of reading a stored mask.  drop_p = 0 disables it.  The reference draws
 * its masks from torch's generator stream; only the distribution can be shared, so dropout parity is statistical.
 * cpc_dropout: x[i] *= factor(i) in place (the feed-forward dropout);  cpc_dropout_mask: mask[i] = factor(i) as f32. */
int cpc_dropout(void* x, long long n, float drop_p, unsigned long long seed, unsigned site, int dtype, void* stream);
int cpc_dropout_mask(float* mask, long long n, float drop_p, unsigned long long seed, unsigned site, void* stream);
/* nn.MultiheadAttention core with the causal mask of attention_model.py:61-63, one (item, head) per workgroup:
 *   qkv T [(b,t)][3C] (q | k | v, head h at columns h*C/heads);  out T [(b,t)][C];  P T [B*heads][S][S] softmax rows (saved,
 *   before dropout; element index of the dropout mask = its offset in P).
 * Limits: S <= 64, C/heads <= 64 (-EINVAL otherwise).  The backward gives dqkv in the layout of qkv.
 * bf16 with C/heads == 64 (the reference's attention architectures) runs on the matrix pipe: scores, P V, dP, dq, dk, dv as 16x16x32
 * MFMAs with P / ds rounded to bf16 for the second products; other head sizes and f32 use vector kernels (f32 accumulation throughout).
 * CPC_ATTN_MFMA=0 in the environment forces the vector kernels (A/B). */
int cpc_attn_fwd(const void* qkv, void* out, void* P, int B, int S, int C, int heads, float drop_p, unsigned long long seed,
                 unsigned site, int dtype, void* stream);
int cpc_attn_bwd(const void* qkv, const void* P, const void* dout, void* dqkv, int B, int S, int C, int heads, float drop_p,
                 unsigned long long seed, unsigned site, int dtype, void* stream);
/* r = a + dropout(b) (b may be NULL; r_out may be NULL), y = LayerNorm(r) * w + bias (transformer.py:262-271, eps inside
 * the sqrt); stats f32 [M][2] = (mean, rstd) saved for the backward; dropout element index = m*C + c. */
int cpc_add_ln_fwd(const void* a, const void* b, const float* w, const float* bias, void* r_out, void* y, float* stats, int M,
                   int C, float eps, float drop_p, unsigned long long seed, unsigned site, int dtype, void* stream);
/* LayerNorm backward: dy = g1 * gscale (+ g2); with bcast > 0 row m reads g1 row m / bcast (the mean over time of
 * attention_model.py:79 folded in, gscale = 1/S).  dr = gradient of r; dr_b (may be NULL) = dr * dropout factor = gradient of
 * the summand b; slabs f32 [nblocks][2][C] hold per-block partial sums of (dw, dbias), to be summed by cpc_reduce_slabs.
 * C*32 bytes of LDS must fit 64 KB. */
int cpc_ln_bwd(const void* g1, const void* g2, const void* r, const float* stats, const float* w, void* dr, float* slabs, int M,
               int C, int bcast, float gscale, int nblocks, void* dr_b, float drop_p, unsigned long long seed, unsigned site,
               int dtype, void* stream);
/* The Wasserstein gradient penalty through AttentionModel (contrastive_estimation_training.py:144-158: loss.backward() through
 * torch.autograd.grad(..., create_graph=True), here for attention_model.py:72-82 / transformer.py:262-271).  f32 only.  "Tangent":
 * the directional derivative along the penalty's direction; "delta": the adjoint of the SUMMED SCORES (pass 1 of DESIGN.md section 8).
 * cpc_ln_tangent: rt = at + dropout(bt) (bt, rt_out may be NULL), yt = w * rstd * (rt - <rt> - xh <xh rt>), xh = (r - mean) * rstd with
 *   (mean, rstd) = stats of the primal cpc_add_ln_fwd.
 * cpc_ln_gp: the second-order terms of LayerNorm.  dy = g1 * gscale (+ g2) is delta at the output (bcast as in cpc_ln_bwd), rt the
 *   tangent of the input: dr += -rstd^2 (xh <p P rt> + <xh rt> P p + <p xh> P rt), p = dy * w, P u = u - <u> - xh <xh u> (what the
 *   input's adjoint of the LAST pass gains; dr_b, may be NULL, gains the same times the dropout factor); slabs f32 [nblocks][C] hold
 *   per-block partial sums of the penalty part of the weight gradient, sum dy * rstd * P rt.  C <= 1024.
 * cpc_attn_tangent: out_t = (pt m) v + (P m) vt, pt = P (u - <P,u>), u = scale (qt k^T + q kt^T), m the dropout factors; qkvt in
 *   the layout of qkv.
 * cpc_attn_gp: dqkv += the second-order terms of the attention core, dout = delta at its output (formulas in csrc/attn.hip). */
int cpc_ln_tangent(const float* at, const float* bt, const float* r, const float* stats, const float* w, float* rt_out, float* yt,
                   int M, int C, float drop_p, unsigned long long seed, unsigned site, void* stream);
int cpc_ln_gp(const float* g1, const float* g2, const float* rt, const float* r, const float* stats, const float* w, float* dr,
              float* dr_b, float* slabs, int M, int C, int bcast, float gscale, int nblocks, float drop_p, unsigned long long seed,
              unsigned site, void* stream);
int cpc_attn_tangent(const float* qkv, const float* qkvt, const float* P, float* out_t, int B, int S, int C, int heads, float drop_p,
                     unsigned long long seed, unsigned site, void* stream);
int cpc_attn_gp(const float* qkv, const float* qkvt, const float* P, const float* dout, float* dqkv, int B, int S, int C, int heads,
                float drop_p, unsigned long long seed, unsigned site, void* stream);
/* out[b][c] = mean_t x[(b,t)][c]  (attention_model.py:79) */
int cpc_mean_time(const void* x, void* out, int B, int S, int C, int dtype, void* stream);

/* ---- scalogram front end (constant_q_transform.py, scalogram_model.py:34-102) ----
 * CQT.forward (constant_q_transform.py:161-172) is one f32 cpc_gemm_nt per octave group over overlapped waveform rows
 * (lda = hop) into cq f32 [B][Tn][ldq] with (re, im) interleaved per bin.  This call is the rest of
 * PreprocessingModule.forward: |z|^2 -> log(. + offset) + log_offset, and with phase != 0 the wrapped phase advance
 * (atan2 difference along time + fixed_pd[bin], single wrap into (-pi, pi], * pd_scale[bin]); then * norm and ** power.
 * out f32 channels-last [B][W/pw][bins/ph][Cc]: phase: W = Tn-1, Cc = 2 (amp of frame w+1, phase difference); else W = Tn,
 * Cc = 1.  ph, pw: F.max_pool2d(x, [ph, pw]) of scalogram_model.py:90-91 (floor mode), taken before the scaling; 1, 1 = none. */
int cpc_scalogram_pointwise(const float* cq, const float* fixed_pd, const float* pd_scale, float* out, int B, int Tn, int bins,
                            long long ldq, int phase, float offset, float log_offset, float norm, float power, int ph, int pw,
                            void* stream);

/* ---- 2-D residual encoder on channels-last "grids" (ScalogramEncoderBlock / ScalogramResidualEncoder,
 * scalogram_model.py:372-529) ----
 * A grid is described by int[6] = {B, W, H, Ha, top, C}: element (b, w, h, c) at ((b*W + w)*Ha + top + h)*C + c, with
 * h the frequency axis, w the time axis; Ha >= top + H rows are allocated per (b, w) column and rows outside
 * [top, top+H) are zero (ZeroPad2d top padding, scalogram_model.py:388-389, is a non-zero `top`).
 * Convolutions: tall (k,1) stride-1 kernels are cpc_gemm_nt / cpc_gemm_tn over overlapped rows of a grid; every other
 * kernel shape goes through cpc_im2col2d + plain GEMMs + cpc_col2im2d.
 *
 * col T [B*Wo*Ho][Kp], column (dh*kw + dw)*C + c = in(b, wo*sw + dw - pw, ho*sh + dh - ph, c) (zero outside the grid and
 * for columns >= kh*kw*C); in_f32: the input grid is f32 (the scalogram itself), else T. */
int cpc_im2col2d(const void* in, void* col, const int* grid, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp,
                 int in_f32, int dtype, void* stream);
/* The adjoint: din(b,w,h,c) (+= if accumulate) sum of the dcol entries that read it. */
int cpc_col2im2d(const void* dcol, void* din, const int* grid, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp,
                 int accumulate, int dtype, void* stream);
/* Depthwise (groups = channels) convolution of Conv2dSeparable (scalogram_model.py:532-544) on the im2col matrix of its input
 * (cpc_im2col2d: tap t of channel c of output row m at col[m][t*C + c]); w f32 [C][taps] is the reference's [C][1][kh][kw].
 * Output / gradient rows m live at (m / rpi) * item + (m % rpi) * C (rpi == 0: m * C), i.e. inside a grid.
 *   cpc_dw_fwd     : y[m][c] = sum_t col[m][t*C + c] * w[c][t]
 *   cpc_dw_bwd_col : dcol[m][t*C + c] = dy[m][c] * w[c][t]      (cpc_col2im2d then gives the input gradient)
 *   cpc_dw_bwd_w   : slabs[blk][c][t] = partial sums over rows of col[m][t*C + c] * dy[m][c]  (reduce with cpc_reduce_slabs) */
int cpc_dw_fwd(const void* col, const float* w, void* y, long long M, int C, int taps, int Kp, int rpi, long long item, int dtype,
               void* stream);
int cpc_dw_bwd_col(const void* dy, const float* w, void* dcol, long long M, int C, int taps, int Kp, int rpi, long long item, int dtype,
                   void* stream);
int cpc_dw_bwd_w(const void* col, const void* dy, float* slabs, long long M, int C, int taps, int Kp, int rpi, long long item,
                 int nblocks, int dtype, void* stream);
/* nn.BatchNorm2d / BatchNorm1d with batch statistics (scalogram_model.py:398-399, audio_model.py:102-103):
 * cpc_bn_stats: slabs f32 [nblocks][2][C] partial (sum, sum of squares) over the rows of x T [rows][C] (pad rows are zero);
 * cpc_bn_finalize: stats f32 [2][C] = (mean, 1/sqrt(biased var + eps)) over `count` elements per channel, and, when
 *   run_mean / run_var are given, torch's running update (momentum, unbiased variance);
 * cpc_bn_apply: out = act((x - mean) * rstd * gamma + beta) on the valid rows of two grids of equal shape. C a multiple of 4, at most 1024.
 * x_f32 (here and in the backward): x / dx are float32 grids although dtype is bf16 — the first convolution of the
 * encoder reads the float32 scalogram and keeps its pre-normalisation output in float32 (log-amplitudes have a large
 * common offset that bf16 storage would quantise away before the normalisation removes it). */
int cpc_bn_stats(const void* x, float* slabs, long long rows, int C, int nblocks, int dtype, void* stream);
int cpc_bn_finalize(const float* slabs, int nslab, int C, double count, float eps, float momentum, float* stats, float* run_mean,
                    float* run_var, void* stream);
/* cpc_bn_apply_residual (bf16, C a multiple of 8): out = act_out(act_in(BatchNorm(x)) + res(w + ow, h + oh)) — the block's second BatchNorm + ReLU,
 * the cropped residual add (scalogram_model.py:447-472) and the ReLU between blocks (:525-526) in one pass; the normalised branch is not stored,
 * only its sign bits (bits, may be NULL; addressed by the element offsets of the activation grid ga it stands for, as cpc_bn_apply_bits writes them)
 * for the BatchNorm's backward pass; obits (may be NULL): the sign bits of out, addressed like out, for cpc_bn_bwd_*_res.  Same results as cpc_bn_apply followed
 * by cpc_residual_add, bit for bit.  r_f32: res is a float32 grid. */
int cpc_bn_apply_residual(const void* x, const int* gx, const void* res, const int* gr, void* out, const int* go, const float* stats,
                          const float* gamma, const float* beta, int oh, int ow, int relu_in, int relu_out, int r_f32, unsigned char* bits,
                          const int* ga, unsigned char* obits, int dtype, void* stream);
/* ... and its backward: the cropped residual add's backward (scalogram_model.py:462-472 under autograd) folded into the second BatchNorm's two
 * backward passes.  dout: gradient of the block output on grid gd; obits (may be NULL: no ReLU behind the add): sign bits of the block output,
 * written by cpc_bn_apply_residual, addressed like dout; abits: sign bits of the normalised branch addressed like its activation grid ga.
 * g = dout [out > 0] [bn_out > 0].  cpc_bn_bwd_reduce_res: slabs as cpc_bn_bwd_reduce.  cpc_bn_bwd_apply_res: dx as cpc_bn_bwd_apply, and
 * dres (may be NULL) = dout [out > 0] at (w + ow, h + oh) of grid gr: the residual operand's gradient.  Neither the masked gradient of the main
 * branch nor a separate residual-add backward pass exists.  bf16, C a multiple of 8. */
int cpc_bn_bwd_reduce_res(const void* dout, const int* gd, const unsigned char* obits, const unsigned char* abits, const int* ga, const void* x,
                          const int* gx, const float* stats, float* slabs, int nblocks, int dtype, void* stream);
int cpc_bn_bwd_apply_res(const void* dout, const int* gd, const unsigned char* obits, const unsigned char* abits, const int* ga, const void* x,
                         void* dx, const int* gx, const float* stats, const float* gamma, const float* dgamma, const float* dbeta, double count,
                         int train, void* dres, const int* gr, int oh, int ow, int dtype, void* stream);
int cpc_bn_apply(const void* x, const int* gx, void* out, const int* go, const float* stats, const float* gamma, const float* beta,
                 int relu, int x_f32, int dtype, void* stream);
/* Backward: g = dy * (y > 0) if relu.  cpc_bn_bwd_reduce: slabs [nblocks][2][C] partials of (sum g*xhat, sum g) = (dgamma,
 * dbeta) to be summed by cpc_reduce_slabs; cpc_bn_bwd_apply: dx = gamma*rstd*(g - dbeta/count - xhat*dgamma/count) with
 * train != 0, g*gamma*rstd otherwise (running statistics). */
int cpc_bn_bwd_reduce(const void* dy, const void* y, const int* gy, const void* x, const int* gx, const float* stats, float* slabs,
                      int relu, int nblocks, int x_f32, int dtype, void* stream);
int cpc_bn_bwd_apply(const void* dy, const void* y, const int* gy, const void* x, void* dx, const int* gx, const float* stats,
                     const float* gamma, const float* dgamma, const float* dbeta, double count, int relu, int train, int x_f32,
                     int dtype, void* stream);
/* The same three passes with the activation's ReLU mask as SIGN BITS (one byte per 8 channels of a position, bit e = channel 8 i + e
 * is > 0, indexed by the element offset in the activation grid / 8): cpc_bn_apply_bits also writes them, the two backward passes read
 * them INSTEAD of the activation (2 bytes per element less in each; a ReLU's backward needs nothing else of it).  bf16 grids with C a
 * multiple of 8 only (CPC_EINVAL otherwise); relu is implied in the backward passes. */
int cpc_bn_apply_bits(const void* x, const int* gx, void* out, const int* go, const float* stats, const float* gamma, const float* beta,
                      int relu, void* out_bits, int dtype, void* stream);
int cpc_bn_bwd_reduce_bits(const void* dy, const void* y_bits, const int* gy, const void* x, const int* gx, const float* stats,
                           float* slabs, int nblocks, int dtype, void* stream);
int cpc_bn_bwd_apply_bits(const void* dy, const void* y_bits, const int* gy, const void* x, void* dx, const int* gx, const float* stats,
                          const float* gamma, const float* dgamma, const float* dbeta, double count, int train, int dtype, void* stream);
/* nn.MaxPool2d(kernel = stride = p): ceil mode (residual branches, scalogram_model.py:434-436; window clipped at the border)
 * or floor mode (main-branch pooling, :401-403; remainder dropped) according to the extents of the output grid; the backward
 * routes dout to the first maximum of each window (+= if accumulate; positions outside every window are not written). */
int cpc_maxpool2d_fwd(const void* in, const int* gi, void* out, const int* go, int p, int in_f32, int dtype, void* stream);
int cpc_maxpool2d_bwd(const void* in, void* din, const int* gi, const void* dout, const int* go, int p, int accumulate, int dtype,
                      void* stream);
/* out = act(a + r(w + ow, h + oh)) — the cropped residual add (scalogram_model.py:453-472) with the inter-block ReLU
 * (:525-526) folded in; backward: da = g, dr(w + ow, h + oh) = g with g = dout * (out > 0) if relu (dr is cleared by the caller).
 * r_f32: the residual grid r / dr is float32 although dtype is bf16 (first block, see x_f32 above). */
int cpc_residual_add(const void* a, const int* ga, const void* r, const int* gr, void* out, const int* go, int oh, int ow, int relu,
                     int r_f32, int dtype, void* stream);
int cpc_residual_add_bwd(const void* dout, const void* out, const int* go, void* da, const int* ga, void* dr, const int* gr, int oh,
                         int ow, int relu, int r_f32, int dtype, void* stream);

/* dst bf16 [3n]: (hi, lo, hi) per f32 sample, hi = bf16(x), lo = bf16(x - hi).  With filter rows laid out (wh, wh, wl) the CQT
 * filter bank runs as bf16 MFMA GEMMs with f32-grade accuracy (xh wh + xl wh + xh wl; the dropped lo*lo term is ~2^-16
 * relative) — the "bf16x3" precision of constant_q_transform.CQT. */
int cpc_split3_bf16(const float* src, void* dst, long long n, void* stream);

/* FIRST BLOCK OF THE SCALOGRAM ENCODER ("stem", ABI version 6): nn.Conv2d on the float32 scalogram (1-4 input channels) + train-mode
 * nn.BatchNorm2d + ReLU (scalogram_model.py:392-406 of the first ScalogramEncoderBlock) WITHOUT storing the convolution's output:
 * every pass recomputes it from the input columns (bit-identically), so the 32-wide float32 tensor of 5 M positions that im2col + GEMM
 * moved three times per direction never exists.  x f32 grid gx = {B, W, H, Ha, top = 0, Cin}; w f32 [Cout][Cin][kh][kw] and bias
 * f32 [Cout] (or NULL) are the reference's parameters; conv = {Cout, kh, kw, sh, sw, ph, pw, Ho, Wo} (int[9]).
 *   cpc_stem_supported : 1 when the window shape / channel counts have kernels (Cout a multiple of 4 up to 64 dividing 1024;
 *                        (Cin, kh, kw, sh) one of (1|2, 3,3, 1|2), (1|2, 5,1, 1), (1|2, 2,2, 1): compile-time shapes, the weights and a
 *                        lane's input window live in registers), else 0 (callers take the im2col route)
 *   cpc_stem_stats     : slabs f32 [nblocks][2][Cout] partial (sum y, sum y^2)                       -> cpc_bn_finalize
 *   cpc_stem_apply     : out (T grid go, any row geometry) = relu((y - mean) rstd gamma + beta)
 *   cpc_stem_bwd_reduce: slabs f32 [nblocks][2][Cout] partial (sum g xhat, sum g), g = da * (a > 0) -> cpc_reduce_slabs (dgamma, dbeta)
 *   cpc_stem_bwd_wgrad : slabs f32 [nblocks][Cout][Cin*kh*kw] partial sums of dy (x) window, dy = gamma rstd (g - dbeta/n - xhat dgamma/n)
 *                        formed per position and never stored -> cpc_reduce_slabs gives d loss / d w in the reference's layout.
 *                        (The bias gradient of a convolution in front of a train-mode BatchNorm is exactly zero.)
 * Residual branch of the same block (scalogram_model.py:434-446, :462-472): xp f32 grid gp = the max-pooled input (cpc_maxpool2d_fwd),
 * wr f32 [Cout][Cin] the 1x1 projection (no bias, padding 0; Cin 1 or 2, Cout a multiple of 8 (bf16) / 4 (f32) dividing 2048 / 1024):
 *   cpc_stem_residual_add: out = act(main + wr xp(w + ow, h + oh))
 *   cpc_stem_residual_bwd: g = dout * (out > 0 if relu); dmain = g; slabs f32 [nblocks][Cout][Cin] partial sums of g (x) xp. */
int cpc_stem_supported(int cin, int cout, int kh, int kw, int sh, int hin, int ph);
int cpc_stem_stats(const float* x, const int* gx, const float* w, const float* bias, const int* conv, float* slabs, int nblocks,
                   void* stream);
int cpc_stem_apply(const float* x, const int* gx, const float* w, const float* bias, const int* conv, const float* stats,
                   const float* gamma, const float* beta, void* out, const int* go, int nblocks, int dtype, void* stream);
int cpc_stem_bwd_reduce(const float* x, const int* gx, const float* w, const float* bias, const int* conv, const float* stats,
                        const void* da, const void* a, const int* ga, float* slabs, int nblocks, int dtype, void* stream);
int cpc_stem_bwd_wgrad(const float* x, const int* gx, const float* w, const float* bias, const int* conv, const float* stats,
                       const float* gamma, const float* dgamma, const float* dbeta, double count, const void* da, const void* a,
                       const int* ga, float* slabs, int nblocks, int dtype, void* stream);
/* cpc_stem_residual_bn_add (bf16): cpc_bn_apply (+ReLU) of the block's second BatchNorm and cpc_stem_residual_add in one pass — y is the
 * BatchNorm's INPUT grid; out = act(bf16(relu(BatchNorm(y))) + wr xp(w + ow, h + oh)); bits (may be NULL): the sign bits of the normalised branch,
 * addressed like its activation grid ga (what the BatchNorm's backward pass reads); obits (may be NULL): the sign bits of out, addressed like out.
 * Bit-identical to the two passes. */
int cpc_stem_residual_bn_add(const void* y, const int* gy, const float* xp, const int* gp, const float* wr, void* out, const int* go, int oh, int ow,
                             int relu, const float* stats, const float* gamma, const float* beta, unsigned char* bits, const int* ga,
                             unsigned char* obits, int dtype, void* stream);
/* cpc_stem_residual_wgrad_bits (bf16): the projection's weight-gradient slabs of cpc_stem_residual_bwd with the ReLU mask taken from obits (the
 * sign bits of out that cpc_stem_residual_bn_add wrote, addressed like out) and WITHOUT storing the masked gradient: cpc_bn_bwd_*_res read dout
 * and the same bits.  gm: the main branch's grid (geometry only). */
int cpc_stem_residual_wgrad_bits(const void* dout, const unsigned char* obits, const int* go, const int* gm, const float* xp, const int* gp,
                                 float* slabs, int oh, int ow, int nblocks, int dtype, void* stream);
int cpc_stem_residual_add(const void* main_, const int* gm, const float* xp, const int* gp, const float* wr, void* out, const int* go,
                          int oh, int ow, int relu, int dtype, void* stream);
int cpc_stem_residual_bwd(const void* dout, const void* out, const int* go, void* dmain, const int* gm, const float* xp, const int* gp,
                          float* slabs, int oh, int ow, int relu, int nblocks, int dtype, void* stream);

/* ---- Wasserstein gradient penalty (contrastive_estimation_training.py:144-155: the gradient of the summed scores with respect
 * to the preprocessed batch, its 2-norm over the channel axis pushed towards 1, differentiated again by loss.backward()) ----
 * The penalty's parameter gradient is the parameter gradient of the directional derivative of the summed scores along
 * v = d penalty / d (input gradient); it is computed as a tangent (forward-mode) pass of v through the network plus extra
 * weight-gradient GEMMs and, at every train-mode BatchNorm, the second-order terms below.  The tangent
 * pass reuses the forward entry points with bias-free operands and the PRIMAL activations as ReLU masks; these three calls are
 * what it needs beyond them:
 * cpc_maxpool2d_select: out = element of `sel` at the first maximum of `in` in every p x p window (a max pooling applied to a
 *   tangent, selecting where the primal pooling selected); grids as for cpc_maxpool2d_fwd, `in` and `sel` share gi.
 * cpc_gp_direction: g f32 [npix][C] = gradient of the summed scores w.r.t. the channels-last scalogram; writes
 *   v = factor * 2 (|g| - 1) / |g| * g / npix (norm over the C channels of a pixel, :153) and per-block partial sums of
 *   (|g| - 1)^2 (penalty = factor * sum / npix).
 * cpc_bn_gp_cross: out = coef[c] * xhat + coef[C + c] * yt + coef[2C + c] * delta on the valid positions of grid gx, with
 *   xhat = (x - stats[c]) * stats[C + c]: the terms a train-mode BatchNorm adds to the adjoint of its input under the penalty. */
int cpc_maxpool2d_select(const void* in, const void* sel, const int* gi, void* out, const int* go, int p, int in_f32, int dtype,
                         void* stream);
int cpc_gp_direction(const float* g, float* v, long long npix, int C, float factor, float* partial, int nblocks, void* stream);
int cpc_bn_gp_cross(const void* x, const void* yt, const void* delta, void* out, const int* gx, const float* stats, const float* coef,
                    int x_f32, int dtype, void* stream);

/* g[i] = y[i] > 0 ? g[i] : 0 for i < n (n % 4 == 0): ReLU backward on whole buffers where no fused epilogue applies. */
int cpc_relu_mask(void* g, const void* y, long long n, int dtype, void* stream);

/* a[i] += b[i] for i < n (n % 4 == 0; ABI 8): the sum in f32, rounded once to the storage type.  Where two branches' data gradients meet
 * and the GEMM that produces the second cannot write into the first's grid (scalogram_model.py:462-476 under autograd). */
int cpc_accumulate(void* a, const void* b, long long n, int dtype, void* stream);

/* dst[r][c] = (T) src[r*sr + c*sc] — cast / transpose of a master weight into an operand layout. */
int cpc_cast2d(const float* src, void* dst, int R, int C, long long sr, long long sc, int dtype, void* stream);

/* Many cpc_cast2d jobs in one launch: jobs is a DEVICE array of njobs records {const float* src; void* dst; int64 R, C, sr, sc}
 * (six 64-bit fields each) with dst[r][c] = (T) src[r*sr + c*sc] — the per-step operand-layout copies of a context network's
 * nn.Linear weights (attention_model.py:59-66 keeps them f32; the GEMMs read storage-dtype copies). */
int cpc_cast2d_batch(const void* jobs, int njobs, int dtype, void* stream);
/* MFMA fragment order of a [R][Kd] operand (transpose: logical[n][k] = src[k*ld + n]) for the GRU kernels. */
int cpc_prep_frag(const float* src, void* dst, int R, int Kd, long long ld, int transpose, int dtype, void* stream);

/* AudioGRUModel.forward's python loop over nn.GRUCell (audio_model.py:66-77) as one persistent launch.
 *   Gi    T   [B][V][3H]  = x_t W_ih^T + b_ih for all steps (cpc_gemm_nt), gate order r, z, n
 *   Wfrag T   cpc_prep_frag(weight_hh [3H][H]);  bhh f32 [3H]
 *   Hall  T   [B][V+1][H] hidden states (Hall[:,0] = 0)
 *   tape  T   cpc_gru_tape_elems(B,V,H,dtype) elements: saved activations (r, z, n, W_hn h + b_hn, h_{t-1}) in a layout
 *             private to the fwd/bwd kernel pair (lane-fragment order for the weight-resident bf16 kernels)
 *   c_out f32 [B][H] last hidden state (what AudioGRUModel.forward returns). */
long long cpc_gru_tape_elems(int B, int V, int H, int dtype);
int cpc_gru_fwd(const void* Gi, const void* Wfrag, const float* bhh, void* Hall, void* tape, float* c_out, int B, int V,
                int H, int dtype, void* stream);
/* The same with an initial hidden state h0 f32 [B][H] (NULL = zeros): AudioGRUModel(reset_hidden=False) carries the last hidden state
 * of one call into the next (audio_model.py:69, :75).  Forward only: the reference's autograd cannot differentiate a second call
 * through the first one's graph either. */
int cpc_gru_fwd_h0(const void* Gi, const void* Wfrag, const float* bhh, const float* h0, void* Hall, void* tape, float* c_out, int B,
                   int V, int H, int dtype, void* stream);
/* Backward through time: dc f32 [B][H] -> dG T [B][V][4H] = [d r_pre | d z_pre | d n_pre | d n_pre * r]: columns [0,3H) are
 * the gradient w.r.t. the input-projection term, columns [0,2H) and [3H,4H) the gradient w.r.t. h W_hh^T + b_hh.
 * WTfrag = cpc_prep_frag(weight_hh, transpose=1) ([H][3H] logical). */
int cpc_gru_bwd(const float* dc, const void* tape, const void* WTfrag, void* dG, int B, int V, int H, int dtype,
                void* stream);

/* The Wasserstein gradient penalty through AudioGRUModel (contrastive_estimation_training.py:144-158: loss.backward() through
 * torch.autograd.grad(..., create_graph=True), here for the GRUCell loop of audio_model.py:66-77).  f32 only, H <= 256.
 * cpc_gru_gp_fwd: primal and tangent recurrence together.  Gi as for cpc_gru_fwd (f32); GiT [B][V][3H] = (tangent of x_t) W_ih^T
 * (no bias); WT [H][3H] = weight_hh transposed; bhh [3H]; tape f32 [B][V][10][H] (r, z, n, W_hn h + b_hn, h_{t-1}, then the
 * tangents of the pre-activations of r and z, of W_hn h, of the pre-activation of n, and of h_{t-1}); ct_out [B][H] = tangent of
 * the last hidden state.
 * cpc_gru_gp_bwd: reverse sweep of the joint program seeded with dc [B][H], the adjoint of the SUMMED SCORES at the last hidden
 * state; W = weight_hh [3H][H].  dA f32 [B][V][8][H]: columns [0,4H) = [d r_pre | d z_pre | d n_pre | d n_pre * r] of the summed
 * scores (to be contracted with the TANGENT inputs / hidden states for the weight gradients), columns [4H,8H) the same four of
 * the second-order adjoint (contracted with the primal inputs / hidden states; [4H,7H) W_ih is what the encoder's top-layer
 * gradient gains). */
int cpc_gru_gp_fwd(const float* Gi, const float* GiT, const float* WT, const float* bhh, float* tape, float* ct_out, int B,
                   int V, int H, void* stream);
int cpc_gru_gp_bwd(const float* dc, const float* tape, const float* W, float* dA, int B, int V, int H, void* stream);

/* A/B switch: on != 0 forces the weight-streaming GRU kernels where the weight-resident bf16 ones would be used
 * (H in {32,64,128,256}); returns the previous setting.  Not stream-ordered (host-side flag). */
int cpc_gru_set_streaming(int on);
/* Tuning knobs for A/B measurements (tools/): key 1 = start stagger of the 256x256 NT GEMM in 1/64 of a tile time (0 = off);
 * keys 4, 5 = timing probes of that kernel's K loop (results are garbage, tools/nt_probe.py); key 6 = output stores of the NT fast
 * kernels: 2 (default) written through the L2 at system scope, 1 at agent scope, 0 plain stores — same results in every mode.
 * Returns the previous value, CPC_EINVAL for an unknown key.  Not part of the product path.  The probe keys 4 / 5 are refused
 * (CPC_EINVAL + a line on stderr) unless the process runs with CPC_ENABLE_PROBES=1, and announce themselves on stderr. */
int cpc_debug_set(int key, int value);

/* InfoNCE loss of ContrastiveEstimationTrainer.train, default branch score_over_all_timesteps=False
 * (contrastive_estimation_training.py:116-122, :141) on the equal-step scores S[k][b][b'] (f32, rows of ld >= B
 * floats; dS / dST use the same ld and get zeros in the pad columns), with the score
 * function folded in (softplus != 0: softplus_score_function :12-16, else linear_score_function :19-22).
 *   out f32[8]: out[0] = loss (incl. regulariser), out[1] = max score (logger value, :166), out[2..4] = -mean valid, mean lse,
 *   reg; out[5] = 1 if the loss before the regulariser (out[2] + out[3]) is NaN — the value the reference's NaN guard tests
 *   (:124) — else 0; out[6] = sticky NaN flag: set to 1 together with out[5], never cleared by the library (the caller zeroes it
 *   when a run starts; cpc_adam's `skip` argument).  cpc_nce_loss_all writes the same eight values.
 *   dS[k][b][b'], dST[k][b'][b] (T): d loss / d linear score.  workspace: cpc_nce_workspace_floats(B,K) f32. */
long long cpc_nce_workspace_floats(int B, int K);
int cpc_nce_loss(const float* S, void* dS, void* dST, float* out, float* workspace, int B, int K, int ld, int softplus,
                 float regularization, int dtype, void* stream);

/* Wasserstein gradient penalty with softplus_score_function (contrastive_estimation_training.py:12-16 under :144-158): the
 * coefficients the penalty's seeds carry.  S (and the tangent scores St1 + St2, St2 may be NULL): nmat f32 matrices [rows][ld].
 * mode 0: w = softplus'(s) = sigmoid(s) (1 beyond torch's threshold 20); mode 1: w = softplus''(s) * (St1 + St2).
 * W [nmat][rows][ld] and its transpose WT [nmat][cols][ldT], both f32. */
int cpc_gp_score_coeff(const float* S, const float* St1, const float* St2, float* W, float* WT, int nmat, int rows, int cols, int ld,
                       int ldT, int mode, void* stream);

/* Same loss with score_over_all_timesteps=True (contrastive_estimation_training.py:108-114, :141): S is the full
 * (B*K) x (B*K) score matrix, row (b,k) = prediction, column (b',k') = target, ST its transpose (both f32, rows of ld
 * floats, produced by two cpc_gemm_nt calls); dS / dST (T, same ld) receive d loss / d linear score and its transpose. */
long long cpc_nce_all_workspace_floats(int B, int K);
int cpc_nce_loss_all(const float* S, const float* ST, void* dS, void* dST, float* out, float* workspace, int B, int K, int ld,
                     int softplus, float regularization, int dtype, void* stream);

/* score_over_all_timesteps=True with the column pass FUSED into the score contraction (bf16 storage; ABI 8).  The reference forms the
 * (B K) x (B K) score tensor (contrastive_estimation_training.py:12-22) and takes logsumexp over its first two axes (:109-110); here the
 * f32 score matrix is never written:
 *   cpc_score_lse      scores s[r][c] = <P[r][:], T[c][:]> (P [M][ldp], T [N][ldt] bf16, E a multiple of 64 and >= 128, M and N multiples of
 *                      256) on 256 x 256 tiles of a persistent MFMA kernel whose epilogue leaves, per M tile and column, the online pair
 *                      pm / ps [M / 256][N] = (max, sum exp(s - max)) over the tile's rows, taken from the f32 accumulators; valid[r] =
 *                      s[r][r + diag_off] where that column exists (a prediction's own target; may be NULL); S [M][lds] f32 = the scores,
 *                      for cpc_nce_fused_grad (NULL: not stored) — f32 because that pass takes exp(score - lse): a bf16 copy of a
 *                      score of 100 is off by up to 0.4, its softmax weight by half (measured: gradient cosine 0.48).
 *   cpc_nce_lse_merge  lse[c] = log( sum over ALL rows of exp(score(s[r][c])) ) from the pairs: for softplus scores exp(softplus(s)) =
 *                      1 + exp(s), so lse = log(nrows_total + sum ps exp(pm)) — no softplus is evaluated; colp [ceil(ncols / 256)][2] =
 *                      {sum of the block's lse, max of its pm}.
 *   cpc_nce_fused_grad d loss / d linear score (bf16, dS [items K][ld]; and its transpose, dST [ncols][ldT], or NULL) from S and lse, as cpc_nce_loss_all forms it:
 *                      (exp(sp - lse[c]) - [c == r + diag_off]) / n_rows_total + 2 reg / (n_items_total^2 K^2) mean_k sp[(b,k)][c], times the score
 *                      function's derivative; rows r = (item, k), items K of them, K even and <= 24, ncols a multiple of 8; gradp
 *                      [cpc_nce_fused_grad_blocks(items, ncols)] = partial sums of (mean_k sp)^2 (the regulariser :141).
 *   cpc_nce_fused_finalize   the eight values cpc_nce_loss_all writes to `out`.  mode 0: from the partials (colp, valid, gradp); mode 1:
 *                      the partials reduced to sums[4] = {sum valid, sum lse, sum m^2, max s} only — a rank that holds a strip of the
 *                      global score matrix all-reduces them (SUM, SUM, SUM, MAX) — and mode 2: out from sums.  n_rows_total /
 *                      n_items_total: rows (predictions) and items of the WHOLE score matrix.
 * Rectangular problems (M != N, diag_off != 0) are the strips of engine.GlobalNegatives: a rank computes all predictions x its own
 * targets and its own predictions x all targets instead of the whole global matrix. */
int cpc_score_lse(const void* P, const void* T, float* S, float* pm, float* ps, float* valid, int M, int N, int E, long long ldp,
                  long long ldt, long long lds, int diag_off, void* stream);
int cpc_nce_lse_merge(const float* pm, const float* ps, int nparts, int ncols, int softplus, float nrows_total, float* lse, float* colp,
                      void* stream);
long long cpc_nce_fused_grad_blocks(int items, int ncols);
int cpc_nce_fused_grad(const float* S, const float* lse, void* dS, void* dST, float* gradp, int items, int K, int ncols, long long ld,
                       long long ldT, int diag_off, int softplus, float regularization, float n_rows_total, float n_items_total, void* stream);
int cpc_nce_fused_finalize(const float* colp, int ncolp, const float* valid, int nvalid, const float* gradp, int ngrad, float* sums, int mode,
                           float n_rows_total, float n_items_total, int K, float regularization, int softplus, float* out, void* stream);

/* The per-batch quantities of ContrastiveEstimationTrainer.validate (contrastive_estimation_training.py:227-247) from the same
 * score matrices the train step uses (S of cpc_nce_loss when all_timesteps == 0, S of cpc_nce_loss_all otherwise; softplus as
 * there): out[0..K) = prediction_losses per step (:237-241, including the reference's reading of the (k, b') log-sum-exps as a
 * (B, K) matrix in flat order in the default branch), out[K..2K) = prediction_accuracy per step (:245-247: arg max over the
 * targets of every prediction equal to its own target; first maximum on ties), out[2K] = mean score (:249).  accumulate != 0
 * adds to out instead of overwriting it (validate sums over batches and divides once).  workspace:
 * cpc_nce_eval_workspace_floats(B, K) f32. */
long long cpc_nce_eval_workspace_floats(int B, int K);
int cpc_nce_eval(const float* S, float* out, float* workspace, int B, int K, int ld, int softplus, int all_timesteps, int accumulate,
                 void* stream);

/* torch.optim.Adam.step with default betas/eps semantics over one flat f32 buffer
 * (contrastive_estimation_training.py:83, :162).  step counts from 1; g is multiplied by grad_scale first.
 * skip (device pointer to one float, or NULL): the reference's NaN guard returns BEFORE backward() / optimizer.step()
 * (contrastive_estimation_training.py:124-133), so a NaN loss leaves the parameters at their last good values.  Here the update
 * is issued without waiting for the host to read the loss: while *skip != 0 — pass out + 6 of cpc_nce_loss / cpc_nce_loss_all,
 * the sticky NaN flag — the call changes nothing (p, m, v and the device-side step count keep their values). */
int cpc_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
             int step, float grad_scale, const float* skip, void* stream);
/* The same update with the step count kept on the device: state f32[4] = {step count (int bits), lr/(1-b1^t), 1/sqrt(1-b2^t), -},
 * zero-initialised by the caller; every call advances the count first.  Nothing in the argument list changes from step to
 * step, so a train step captured in a hipGraph can be replayed. */
int cpc_adam_dev(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float* state,
                 float grad_scale, const float* skip, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CPC_HIP_H */
