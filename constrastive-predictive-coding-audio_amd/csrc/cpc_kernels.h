// Internal launcher interface between the kernel translation units and the C ABI (cpc_abi.hip).
#pragma once
#include <hip/hip_runtime.h>

// gemm flags
#define GEMM_RELU 1        // NT: relu in the epilogue
#define GEMM_OUT_F32 2     // output stored as f32 regardless of the storage dtype
#define GEMM_TN_NO_TR 4    // TN/bf16: use scalar LDS reads instead of ds_read_b64_tr_b16 (debug / A-B check)
#define GEMM_SMALL_TILE 16     // NT: keep the 128x128 tile even where the 256x256 one would be chosen (A-B check)
#define GEMM_NARROW_EPI 32     // NT/bf16: per-lane 8-byte stores instead of the LDS-staged full-row epilogue (A-B check)
#define GEMM_WIDE_EPI 0x10000  // internal: set by the launcher when the LDS-staged epilogue applies
#define GEMM_EPI_CONV1 0x20000 // internal: fused layer-1 weight-gradient epilogue (see GemmNT::c1_*)
#define GEMM_NO_DMA 64         // NT fast path: register-staged global->LDS copies instead of LDS-DMA (A-B check)
#define GEMM_SKIP_PAD_ROWS 128  // NT: rows with (m % c_rpi) >= c_valid are not stored at all (default: stored as zeros)
#define GEMM_NO_PERS 512        // NT/bf16: LDS-staged epilogue instead of the register epilogue (A-B check; see DIRECT in gemm.hip)
#define GEMM_DIRECT_MASK 1024   // NT/bf16: register epilogue also for masked launches (A-B check)
#define GEMM_KRANGE_EXACT 2048  // NT fast path with k_ranges: the ranges cut out pieces that are NOT zero (a neighbouring item's data), so a tile must lie in ONE
                                // range index: the launch is refused unless a_rpi * max(a_rpi2, 1) is a multiple of the tile height the launcher picks
#define GEMM_LINEAR_K 256       // NT fast path: visit K in storage order even for overlapped-row operands (A-B check, see GemmNT::k_taps)
#define GEMM_BIG_TILE 0x100000  // internal: the 256x256 tile also where fewer than 200 of them exist (a row-range launch that needs the per-tile column sums)
#define GEMM_WT_AGENT 0x40000   // internal (cpc_debug_set key 6): output stores of the NT fast kernels write through at agent scope (sc1)
#define GEMM_WT_SYSTEM 0x80000  // ... at system scope (sc0 sc1): the default
#define GEMM_FORCE_GENERIC 8   // use the register-staged generic kernel even when the LDS-DMA fast path applies (A-B check)

struct GemmNT {
    const void* A;        // [M][K], row m at row_off(m, a_rpi, a_item, lda); rows may overlap (strided-conv view)
    const void* Bt;       // [N][K]
    void* C;              // [M][N]
    const float* bias;    // [N] or null
    const void* mask;     // same addressing/type as C's storage dtype; out = mask > 0 ? out : 0  (relu backward); or null
    // The same mask as one byte per 8 elements (bit e of byte i = element 8 i + e of the mask array is > 0; cpc_sign_bits makes it),
    // addressed by the C element offset / 8.  When set it replaces `mask` in the 256x256 bf16 LDS-staged epilogues: the tile's 8 KiB of
    // bits are fetched by LDS-DMA before the K loop starts, so the epilogue reads no mask from memory at all.
    const unsigned char* mask_bits = nullptr;
    // Per-tile column sums of the stored result (256x256 bf16 LDS-staged epilogue): slabs[ceil(M / 256)][N] f32, slab mt = the sums over the
    // rows of M-tile mt (rows beyond M and skipped pad rows count as zero).  Reduced by reduce_slabs: the bias gradient of the layer below
    // a conv data gradient without a separate pass over the gradient it just wrote.
    float* colsum_slabs = nullptr;
    int M, N, K;
    long long lda, ldb, ldc;
    int a_rpi; long long a_item;
    int b_rpi; long long b_item;
    int c_rpi; long long c_item; int c_valid;   // rows with (m % c_rpi) >= c_valid are written as zeros (c_rpi != 0)
    long long a_batch, b_batch, c_batch;
    int flags;
    // Second row level (fast LDS-DMA kernels): with a_rpi2 != 0 row m of A sits at
    //   (m / (a_rpi a_rpi2)) * a_item2 + ((m / a_rpi) % a_rpi2) * a_item + (m % a_rpi) * lda,   likewise C (and the mask) with c_*2.
    // "Bands": the rows of a grid ordered (band of rows, column, row within the band) — every row of a tile then lies in one or two
    // bands, and k_ranges can cut the K loop to what that band needs.
    int a_rpi2 = 0; long long a_item2 = 0;
    int c_rpi2 = 0; long long c_item2 = 0;
    // k_ranges (device, int pairs, or null): the K stages [lo, hi) (units of 64 bf16 / 32 f32 elements) that are not known zeros for
    // level-2 index i = m / (a_rpi a_rpi2) (a_rpi2 == 0: item index m / a_rpi), hi > lo.  A tile runs the union of its rows' ranges.
    // The data gradient of a tall (k,1) convolution reads a window of output-gradient rows of which, near the top and bottom of a
    // column, most lie outside the column (zeros): 47 % of the MACs of the (30,1) kernel on 63 rows.
    const int* k_ranges = nullptr;
    int m_off = 0;        // internal: first row of this launch (fast kernels; a launch may cover rows [m_off, M) only)
    // internal, fast kernels: order in which the K axis is visited.  Stage s (BK elements) covers the K offsets
    //   (s / k_taps) * BK + (s % k_taps) * k_tap_stride ... + BK        (k_taps <= 1: plain s * BK)
    // for BOTH operands — a permutation of the reduction index, invisible in the sum.  The launcher sets it for overlapped-row
    // A operands (lda < K, K % lda == 0: the strided-conv views) to k_tap_stride = lda, k_taps = K / lda: row m+1 at tap j then
    // reads the bytes row m read at tap j+1 ONE stage earlier (an L2 hit) instead of lda / BK stages earlier (by then evicted:
    // every activation byte crossed the fabric K / lda times; profiles/r01h_traffic.json, conv forward 2.16x).
    int k_taps = 0; long long k_tap_stride = 0;
    // k_tap_stride_a (0 = k_tap_stride): with k_taps > 1 the A operand's tap j starts k_tap_stride_a elements after tap j-1 while the B operand
    // and the K axis stay dense (K = k_taps * k_tap_stride).  A row of A is then k_taps SEPARATE pieces of k_tap_stride elements: the window
    // of a 2-D convolution read straight from a channels-last grid (piece = the kh rows x C channels of one kernel column, contiguous in
    // the grid; the next kernel column is one grid column = Ha * C elements further) — the im2col matrix is never written.
    long long k_tap_stride_a = 0;
    // internal: the taps were given by the caller (gathered rows): stages are visited in storage order (all of piece 0, then piece 1, ...), the
    // order k_ranges counts in, instead of tap-innermost (there is no reuse between the pieces of neighbouring rows to win)
    bool k_taps_linear = false;
    // internal, 256x256 LDS-DMA kernel: start stagger.  The workgroups of the first round (one per CU) start (phase * stagger)
    // sleeps of 4096 cycles late, phase = (block / 8) % 8, so that the CUs do not all sit in their epilogues (a 32 MB store
    // burst per round of tiles) and prologues at the same time; 0 = off.  Set by the launcher from the tile's K extent.
    int stagger = 0;
    // internal, GEMM_EPI_CONV1 (data gradient of encoder layer 2 fused with the weight gradient of layer 1): the masked result
    // tile G[(b,q)][(r,c)] = d loss / d act1[b][q*c1_sub + r][c] is NOT stored; instead
    //   c1_slabs[tile][j][c] = sum_rows G[row][c] * x[b][t*c1_stride + j]  (j < c1_kw),   [c1_kw][c] = sum_rows G[row][c]
    // with t = q*c1_sub + r (rows with t >= c1_valid contribute nothing); tile = mt * numN + nt, 256 columns per tile.
    const float* c1_x = nullptr; long long c1_ldx = 0; float* c1_slabs = nullptr;
    int c1_rpi = 0, c1_sub = 0, c1_stride = 0, c1_kw = 0, c1_valid = 0;
    int c1_row0 = 0;      // row m of the launch is row c1_row0 + m % c1_rpi of item m / c1_rpi (a launch over a row range of every item)
    // internal, launch_score_lse (persistent 256x256 bf16 kernel, LSE epilogue): per M tile and column the pair (max, sum exp(s - max)) over
    // the tile's rows -> lse_pm / lse_ps [M / 256][N]; lse_valid[r] = s[r][r + lse_diag_off] where that column exists (or null)
    float* lse_pm = nullptr; float* lse_ps = nullptr; float* lse_valid = nullptr; int lse_diag_off = 0;
};

struct GemmTN {
    const void* A;        // [M][I]
    const void* B;        // [M][J]
    void* C;              // [I][J]  (+ split * slab_stride)
    int M, I, J;
    long long lda, ldb, ldc;
    int a_rpi; long long a_item;
    int b_rpi; long long b_item;
    long long a_batch, b_batch, c_batch;
    int c_rpi; long long c_item;      // output row i at row_off(i, c_rpi, c_item, ldc)  (c_rpi == 0: plain i * ldc)
    // second row level of the A operand (LDS-DMA kernel): row m at (m / (a_rpi a_rpi2)) a_item2 + ((m / a_rpi) % a_rpi2) a_item + (m % a_rpi) lda —
    // the windows of a strided 2-D convolution read straight from a channels-last grid (clip, output column, output row), no im2col matrix
    int a_rpi2 = 0; long long a_item2 = 0;
    int m_chunk;          // reduction rows per split
    long long slab_stride;
    int flags;
};

// tuning knobs (cpc_debug_set): key 1 = stagger of the 256x256 NT kernel in 1/64 of a tile time (default see gemm.hip)
extern int g_nt_stagger64;
extern int g_nt_probe_taps;     // key 5: taps of the chunk-major A operand of probe 32
extern int g_nt_wt;             // key 6: output stores of the NT fast kernels written through the L2 (1: sc1, 2: sc0 sc1)
extern int g_nt_probe;          // key 4: timing probes of the 256x256 NT kernel (DBG in gemm.hip; the results are garbage)
int launch_gemm_nt(const GemmNT& p, int dtype, int batch, hipStream_t stream);
int launch_score_lse(const void* P, const void* Tg, void* Sb, float* pm, float* ps, float* valid, int M, int N, int E, long long ldp,
                     long long ldt, long long lds_, int diag_off, hipStream_t stream);
int launch_gemm_tn(const GemmTN& p, int dtype, int nsplit, int batch, hipStream_t stream);
int launch_reduce_slabs(const float* slabs, float* out, int I, int J, int nslab, long long slab_stride, int cdiv,
                        long long s_j, long long s_hi, long long s_lo, hipStream_t stream);
int launch_dw_fwd(const void* col, const float* w, void* y, long long M, int C, int taps, int Kp, int rpi, long long item, int dtype,
                  hipStream_t st);
int launch_dw_bwd_col(const void* dy, const float* w, void* dcol, long long M, int C, int taps, int Kp, int rpi, long long item,
                      int dtype, hipStream_t st);
int launch_dw_bwd_w(const void* col, const void* dy, float* slabs, long long M, int C, int taps, int Kp, int rpi, long long item,
                    int nblocks, int dtype, hipStream_t st);
int launch_cast2d_batch(const void* jobs, int njobs, int dtype, hipStream_t stream);
int launch_colsum(const void* X, float* slabs, int M, int N, long long ldx, int dtype, int nblocks, hipStream_t stream);

int launch_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int C, int stride, int kw,
                     long long ldx, int L_valid, int L_alloc, int relu, int dtype, unsigned char* y_bits, hipStream_t stream,
                     int row_lo = 0, int row_hi = -1);
int launch_conv1_bwd(const float* x, const void* dy, float* slabs, int B, int C, int stride, int kw, long long ldx,
                     int L_valid, int L_alloc, int nblk_t, int nblk_b, int dtype, hipStream_t stream);
int launch_conv1_fused_reduce(const float* slabs, float* tmp, float* dw, float* db, int numM, int cin, int sub, int kw,
                              hipStream_t stream);
int launch_reduce_conv_w2d(const float* slabs, float* out, int nslab, long long slab_stride, int cout, int cin, int kh, int kw,
                           long long s_dw, long long s_dh, long long s_c, int G, long long s_g, hipStream_t stream);
int launch_reduce_conv_w(const float* slabs, float* out, int cin, int cout, int kw, int nslab, long long slab_stride,
                         hipStream_t stream);
long long gru_tape_elems(int B, int V, int H, int dtype);
int launch_gru_fwd(const void* Gi, const void* Wfrag, const float* bhh, void* Hall, void* tape, float* c_out, int B,
                   int V, int H, int dtype, hipStream_t stream, const float* h0 = nullptr);
int launch_gru_bwd(const float* dc, const void* tape, const void* WTfrag, void* dG, int B, int V, int H, int dtype,
                   hipStream_t stream);
int launch_gru_gp_fwd(const float* Gi, const float* GiT, const float* WT, const float* bhh, float* tape, float* ct_out, int B,
                      int V, int H, hipStream_t stream);
int launch_gru_gp_bwd(const float* dc, const float* tape, const float* W, float* dA, int B, int V, int H, hipStream_t stream);
int launch_prep_frag(const float* src, void* dst, int R, int Kd, long long ld, int transpose, int dtype, hipStream_t stream);
int launch_nce_lse_merge(const float* pm, const float* ps, int nparts, int ncols, int softplus, float nrows, float* lse, float* colp,
                         hipStream_t stream);
long long nce_fused_grad_blocks(int items, int ncols);
int launch_nce_fused_grad(const void* Sb, const float* lse, void* dS, void* dST, float* gradp, int items, int K, int ncols, long long ld,
                          long long ldT, int diag_off, int softplus, float reg, float n_rows, float n_items, hipStream_t stream);
int launch_nce_fused_finalize(const float* colp, int ncolp, const float* valid, int nvalid, const float* gradp, int ngrad, float* sums, int mode,
                              float n_rows, float n_items, int K, float reg, int softplus, float* out, hipStream_t stream);
long long nce_workspace_floats(int B, int K);
long long nce_all_workspace_floats(int B, int K);
int launch_nce_all(const float* S, const float* ST, void* dS, void* dST, float* out, float* workspace, int B, int K, int ld,
                   int softplus, float reg, int dtype, hipStream_t stream);
int launch_nce(const float* S, void* dS, void* dST, float* out, float* workspace, int B, int K, int ld, int softplus, float reg,
               int dtype, hipStream_t stream);
int launch_gp_score_coeff(const float* S, const float* St1, const float* St2, float* W, float* WT, int nmat, int rows, int cols, int ld,
                          int ldT, int mode, hipStream_t stream);
long long nce_eval_workspace_floats(int B, int K);
int launch_nce_eval(const float* S, float* out, float* workspace, int B, int K, int ld, int softplus, int all_timesteps,
                    int accumulate, hipStream_t stream);
int launch_sign_bits(const void* x, unsigned char* bits, long long n, int dtype, hipStream_t stream);
int launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step,
                float grad_scale, const float* skip, hipStream_t stream);
int conv_w_prep_plan(void* jobs_host, int njobs, int* total_blocks, int* lds_bytes);
int launch_conv_w_prep_batch(const void* jobs_dev, int njobs, int total_blocks, int lds_bytes, int dtype, hipStream_t stream);
int launch_conv_w_prep_group(const float* W, const float* bias, void* fwd, void* dgrd, float* bias_g, int Cout, int Cin, int kh, int G,
                             int Rw, int Rd, int dtype, hipStream_t stream);
int launch_conv_w_prep(const float* W, void* fwd, void* dgrd, int Cout, int Cin, int kw, int stride, int dtype,
                       hipStream_t stream);
int launch_cast2d(const float* src, void* dst, int R, int C, long long sr, long long sc, int dtype, hipStream_t stream);

int launch_maxpool_fwd(const void* in, void* out, int B, int C, int pool, int Lin_valid, int Lin_alloc, int Lout_valid,
                       int Lout_alloc, int dtype, hipStream_t stream);
int launch_maxpool_bwd(const void* in, const void* dout, void* din, int B, int C, int pool, int Lin_valid, int Lin_alloc,
                       int Lout_alloc, int dtype, hipStream_t stream);
int launch_relu_row_bwd(const float* dc, const void* y, void* dy, int B, int C, long long item_stride, long long row_off, int dtype,
                        hipStream_t stream);

// attention context network (attn.hip)
int launch_pe_scale_fwd(const void* top, const float* pe, void* x0, int B, int S, int C, long long item_stride, float scale,
                        int dtype, hipStream_t stream);
int launch_pe_scale_bwd(const void* g1, const void* g2, void* dtop, int B, int S, int C, long long item_stride, float scale,
                        int dtype, hipStream_t stream);
int launch_attn_fwd(const void* qkv, void* out, void* P, int B, int S, int C, int heads, float drop_p, unsigned long long seed,
                    unsigned site, int dtype, hipStream_t stream);
int launch_attn_bwd(const void* qkv, const void* P, const void* dout, void* dqkv, int B, int S, int C, int heads, float drop_p,
                    unsigned long long seed, unsigned site, int dtype, hipStream_t stream);
int launch_add_ln_fwd(const void* a, const void* b, const float* w, const float* bias, void* r_out, void* y, float* stats, int M,
                      int C, float eps, float drop_p, unsigned long long seed, unsigned site, int dtype, hipStream_t stream);
int launch_ln_bwd(const void* g1, const void* g2, const void* r, const float* stats, const float* w, void* dr, float* slabs, int M,
                  int C, int bcast, float gscale, int nblocks, void* dr_b, float drop_p, unsigned long long seed, unsigned site,
                  int dtype, hipStream_t stream);
int launch_ln_tangent(const float* at, const float* bt, const float* r, const float* stats, const float* w, float* rt_out, float* yt,
                      int M, int C, float drop_p, unsigned long long seed, unsigned site, hipStream_t st);
int launch_ln_gp(const float* g1, const float* g2, const float* rt, const float* r, const float* stats, const float* w, float* dr,
                 float* dr_b, float* slabs, int M, int C, int bcast, float gscale, int nblocks, float drop_p, unsigned long long seed,
                 unsigned site, hipStream_t st);
int launch_attn_tangent(const float* qkv, const float* qkvt, const float* P, float* out_t, int B, int S, int C, int heads,
                        float drop_p, unsigned long long seed, unsigned site, hipStream_t st);
int launch_attn_gp(const float* qkv, const float* qkvt, const float* P, const float* dout, float* dqkv, int B, int S, int C, int heads,
                   float drop_p, unsigned long long seed, unsigned site, hipStream_t st);
int launch_dropout(void* x, long long n, float drop_p, unsigned long long seed, unsigned site, int dtype, hipStream_t stream);
int launch_dropout_mask(float* mask, long long n, float drop_p, unsigned long long seed, unsigned site, hipStream_t stream);
int launch_mean_time(const void* x, void* out, int B, int S, int C, int dtype, hipStream_t stream);

// scalogram front end / 2-D encoder (scalogram.hip)
int launch_scalogram_pointwise(const float* cq, const float* fixed_pd, const float* pd_scale, float* out, int B, int Tn, int bins,
                               long long ldq, int phase, float offset, float log_offset, float norm, float power, int ph, int pw,
                               hipStream_t stream);
int launch_im2col2d(const void* in, void* col, const int* grid, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp,
                    int in_f32, int dtype, hipStream_t stream);
int launch_col2im2d(const void* dcol, void* din, const int* grid, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo,
                    int Kp, int accumulate, int dtype, hipStream_t stream);
int launch_bn_stats(const void* x, float* slabs, long long rows, int C, int nblocks, int dtype, hipStream_t stream);
int launch_bn_finalize(const float* slabs, int nslab, int C, double count, float eps, float momentum, float* stats, float* run_mean,
                       float* run_var, hipStream_t stream);
int launch_bn_apply_residual(const void* x, const int* gx, const void* res, const int* gr, void* out, const int* go, const float* stats,
                             const float* gamma, const float* beta, int oh, int ow, int relu_in, int relu_out, int r_f32, unsigned char* bits,
                             const int* ga, int dtype, hipStream_t st, unsigned char* obits = nullptr);
int launch_bn_bwd_reduce_res(const void* dout, const int* gd, const unsigned char* obits, const unsigned char* abits, const int* ga,
                             const void* x, const int* gx, const float* stats, float* slabs, int nblocks, int dtype, hipStream_t st);
int launch_bn_bwd_apply_res(const void* dout, const int* gd, const unsigned char* obits, const unsigned char* abits, const int* ga, const void* x,
                            void* dx, const int* gx, const float* stats, const float* gamma, const float* dgamma, const float* dbeta,
                            double count, int train, void* dres, const int* gr, int oh, int ow, int dtype, hipStream_t st);
int launch_bn_apply(const void* x, const int* gx, void* out, const int* go, const float* stats, const float* gamma, const float* beta,
                    int relu, int x_f32, int dtype, hipStream_t stream, unsigned char* bits = nullptr);
int launch_bn_bwd_reduce(const void* dy, const void* y, const int* gy, const void* x, const int* gx, const float* stats, float* slabs,
                         int relu, int nblocks, int x_f32, int dtype, hipStream_t stream, const unsigned char* ybits = nullptr);
int launch_bn_bwd_apply(const void* dy, const void* y, const int* gy, const void* x, void* dx, const int* gx, const float* stats,
                        const float* gamma, const float* dgamma, const float* dbeta, double count, int relu, int train, int x_f32,
                        int dtype, hipStream_t stream, const unsigned char* ybits = nullptr);
int launch_maxpool2d_fwd(const void* in, const int* gi, void* out, const int* go, int p, int in_f32, int dtype, hipStream_t stream);
int launch_maxpool2d_bwd(const void* in, void* din, const int* gi, const void* dout, const int* go, int p, int accumulate, int dtype,
                         hipStream_t stream);
int launch_residual_add(const void* a, const int* ga, const void* r, const int* gr, void* out, const int* go, int oh, int ow, int relu,
                        int r_f32, int dtype, hipStream_t stream);
int launch_residual_add_bwd(const void* dout, const void* out, const int* go, void* da, const int* ga, void* dr, const int* gr, int oh,
                            int ow, int relu, int r_f32, int dtype, hipStream_t stream);
int launch_maxpool2d_select(const void* in, const void* sel, const int* gi, void* out, const int* go, int p, int in_f32, int dtype,
                            hipStream_t st);
int launch_gp_direction(const float* g, float* v, long long npix, int C, float factor, float* partial, int nblocks, hipStream_t st);
int launch_bn_gp_cross(const void* x, const void* yt, const void* delta, void* out, const int* gx, const float* stats, const float* coef,
                       int x_f32, int dtype, hipStream_t st);
int launch_relu_mask(void* g, const void* y, long long n, int dtype, hipStream_t stream);
int launch_accumulate(void* a, const void* b, long long n, int dtype, hipStream_t stream);
int launch_split3_bf16(const float* src, void* dst, long long n, hipStream_t stream);
int launch_adam_dev(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float* state,
                    float grad_scale, const float* skip, hipStream_t stream);

// first block of the scalogram encoder on the float32 input (stem.hip): the convolution is recomputed, never stored
int launch_stem_supported(int cin, int cout, int kh, int kw, int sh, int hin, int ph);
int launch_stem_stats(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw, int ph,
                      int pw, int Ho, int Wo, float* slabs, int nblocks, hipStream_t stream);
int launch_stem_apply(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw, int ph,
                      int pw, int Ho, int Wo, const float* stats, const float* gamma, const float* beta, void* out, const int* go,
                      int nblocks, int dtype, hipStream_t stream);
int launch_stem_bwd_reduce(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw,
                           int ph, int pw, int Ho, int Wo, const float* stats, const void* da, const void* a, const int* ga, float* slabs,
                           int nblocks, int dtype, hipStream_t stream);
int launch_stem_bwd_wgrad(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw,
                          int ph, int pw, int Ho, int Wo, const float* stats, const float* gamma, const float* dgamma, const float* dbeta,
                          double count, const void* da, const void* a, const int* ga, float* slabs, int nblocks, int dtype,
                          hipStream_t stream);
int launch_stem_residual_bn_add(const void* y, const int* gy, const float* xp, const int* gp, const float* wr, void* out, const int* go, int oh,
                                int ow, int relu, const float* stats, const float* gamma, const float* beta, unsigned char* bits, const int* ga,
                                int dtype, hipStream_t st, unsigned char* obits = nullptr);
int launch_stem_residual_add(const void* main_, const int* gm, const float* xp, const int* gp, const float* wr, void* out, const int* go,
                             int oh, int ow, int relu, int dtype, hipStream_t stream);
int launch_stem_residual_bwd(const void* dout, const void* out, const int* go, void* dmain, const int* gm, const float* xp, const int* gp,
                             float* slabs, int oh, int ow, int relu, int nblocks, int dtype, hipStream_t stream, const unsigned char* obits = nullptr);
