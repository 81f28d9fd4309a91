// extern "C" boundary of libcpc_hip.so: argument checking + translation of the domain-level calls (conv / GRU / NCE /
// Adam) onto the kernel launchers.  See include/cpc_hip.h for the contract.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "cpc_common.h"
#include "cpc_kernels.h"
#include "../../include/cpc_hip.h"

static_assert(CPC_GEMM_FORCE_GENERIC == GEMM_FORCE_GENERIC && CPC_GEMM_SMALL_TILE == GEMM_SMALL_TILE, "flag mismatch");
static_assert(CPC_GEMM_SKIP_PAD_ROWS == GEMM_SKIP_PAD_ROWS && CPC_GEMM_NO_DMA == GEMM_NO_DMA, "flag mismatch");
static_assert(CPC_GEMM_RELU == GEMM_RELU && CPC_GEMM_OUT_F32 == GEMM_OUT_F32 && CPC_GEMM_TN_NO_TR == GEMM_TN_NO_TR, "flag mismatch");
static_assert(CPC_GEMM_LINEAR_K == GEMM_LINEAR_K && CPC_GEMM_NO_PERS == GEMM_NO_PERS && CPC_GEMM_DIRECT_MASK == GEMM_DIRECT_MASK, "flag mismatch");
static_assert(CPC_F32 == CPC_DTYPE_F32 && CPC_BF16 == CPC_DTYPE_BF16, "dtype mismatch");

static inline int esize(int dtype) { return dtype == CPC_DTYPE_BF16 ? 2 : 4; }

extern "C" {

int cpc_abi_version(void) { return 8; }

int cpc_gemm_nt(const cpc_gemm_nt_args* a, void* stream) {
    if (!a || !a->A || !a->Bt || !a->C) return CPC_EINVAL;
    GemmNT p = {};
    p.A = a->A; p.Bt = a->Bt; p.C = a->C; p.bias = a->bias; p.mask = a->mask;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc;
    p.a_rpi = a->a_rpi; p.a_item = a->a_item;
    p.b_rpi = a->b_rpi; p.b_item = a->b_item;
    p.c_rpi = a->c_rpi; p.c_item = a->c_item; p.c_valid = a->c_valid;
    p.a_batch = a->a_batch; p.b_batch = a->b_batch; p.c_batch = a->c_batch;
    p.flags = a->flags;
    p.a_rpi2 = a->a_rpi2; p.a_item2 = a->a_item2; p.c_rpi2 = a->c_rpi2; p.c_item2 = a->c_item2; p.k_ranges = a->k_ranges;
    if (a->a_rpi2 && a->a_extent > 0) return CPC_EINVAL;
    if (a->k_taps < 0 || a->k_taps == 1 || (a->k_taps > 1 && (a->k_tap_stride <= 0 || a->k_tap_stride_a < 0 || a->a_extent > 0))) return CPC_EINVAL;
    p.k_taps = a->k_taps; p.k_tap_stride = a->k_taps > 1 ? a->k_tap_stride : 0; p.k_tap_stride_a = a->k_taps > 1 ? a->k_tap_stride_a : 0;
    p.k_taps_linear = a->k_taps > 1;
    if (a->dtype == CPC_DTYPE_F32) p.flags |= GEMM_OUT_F32;
    if (a->mask && (p.flags & GEMM_OUT_F32) && a->dtype != CPC_DTYPE_F32) return CPC_EINVAL;
    // Optional extent check (see the over-read contract in cpc_hip.h): the last row of the last batch ends at ..._end elements.
    const int nb = a->batch > 0 ? a->batch : 1;
    if (a->a_extent > 0 && a->M > 0) {
        const long long last = a->a_rpi ? (long long)((a->M - 1) / a->a_rpi) * a->a_item + (long long)((a->M - 1) % a->a_rpi) * a->lda
                                        : (long long)(a->M - 1) * a->lda;
        if (last + a->K + (long long)(nb - 1) * a->a_batch > a->a_extent) return CPC_EINVAL;
    }
    if (a->b_extent > 0 && a->N > 0) {
        const long long last = a->b_rpi ? (long long)((a->N - 1) / a->b_rpi) * a->b_item + (long long)((a->N - 1) % a->b_rpi) * a->ldb
                                        : (long long)(a->N - 1) * a->ldb;
        if (last + a->K + (long long)(nb - 1) * a->b_batch > a->b_extent) return CPC_EINVAL;
    }
    return launch_gemm_nt(p, a->dtype, nb, (hipStream_t)stream);
}

int cpc_gemm_tn(const cpc_gemm_tn_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C) return CPC_EINVAL;
    GemmTN p;
    p.A = a->A; p.B = a->B; p.C = a->C;
    p.M = a->M; p.I = a->I; p.J = a->J;
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc;
    p.a_rpi = a->a_rpi; p.a_item = a->a_item;
    p.b_rpi = a->b_rpi; p.b_item = a->b_item;
    p.a_batch = a->a_batch; p.b_batch = a->b_batch; p.c_batch = a->c_batch;
    p.c_rpi = a->c_rpi; p.c_item = a->c_item;
    p.a_rpi2 = a->a_rpi2; p.a_item2 = a->a_item2;
    const int nsplit = a->nsplit > 0 ? a->nsplit : 1;
    p.m_chunk = nsplit > 1 ? a->m_chunk : a->M;
    p.slab_stride = a->slab_stride;
    p.flags = a->flags;
    if (a->dtype == CPC_DTYPE_F32) p.flags |= GEMM_OUT_F32;
    const int blk = a->dtype == CPC_DTYPE_BF16 ? 64 : 32;
    if (nsplit > 1 && (p.m_chunk % blk)) return CPC_EINVAL;
    return launch_gemm_tn(p, a->dtype, nsplit, a->batch > 0 ? a->batch : 1, (hipStream_t)stream);
}

int cpc_reduce_slabs(const float* slabs, float* out, int I, int J, int nslab, long long slab_stride, int cdiv, long long s_j,
                     long long s_hi, long long s_lo, void* stream) {
    if (!slabs || !out) return CPC_EINVAL;
    return launch_reduce_slabs(slabs, out, I, J, nslab, slab_stride, cdiv, s_j, s_hi, s_lo, (hipStream_t)stream);
}

int cpc_colsum(const void* X, float* slabs, int M, int N, long long ldx, int nblocks, int dtype, void* stream) {
    if (!X || !slabs) return CPC_EINVAL;
    return launch_colsum(X, slabs, M, N, ldx, dtype, nblocks, (hipStream_t)stream);
}

int cpc_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int C, int stride, int kw, long long ldx,
                  int L_valid, int L_alloc, int relu, int dtype, void* y_bits, void* stream) {
    if (!x || !w || !y) return CPC_EINVAL;
    if ((long long)(L_valid - 1) * stride + kw > ldx) return CPC_EINVAL;
    return launch_conv1_fwd(x, w, bias, y, B, C, stride, kw, ldx, L_valid, L_alloc, relu, dtype, (unsigned char*)y_bits, (hipStream_t)stream);
}

int cpc_conv1_fwd_rows(const float* x, const float* w, const float* bias, void* y, int B, int C, int stride, int kw, long long ldx,
                       int L_valid, int L_alloc, int relu, int dtype, void* y_bits, int row_lo, int row_hi, void* stream) {
    if (!x || !w || !y) return CPC_EINVAL;
    if ((long long)(L_valid - 1) * stride + kw > ldx) return CPC_EINVAL;
    return launch_conv1_fwd(x, w, bias, y, B, C, stride, kw, ldx, L_valid, L_alloc, relu, dtype, (unsigned char*)y_bits, (hipStream_t)stream,
                            row_lo, row_hi);
}

int cpc_sign_bits(const void* x, void* bits, long long n, int dtype, void* stream) {
    if (!x || !bits || n <= 0 || n % 32) return CPC_EINVAL;
    return launch_sign_bits(x, (unsigned char*)bits, n, dtype, (hipStream_t)stream);
}

int cpc_conv1_bwd(const float* x, const void* dy, float* slabs, int B, int C, int stride, int kw, long long ldx, int L_valid,
                  int L_alloc, int nblk_t, int nblk_b, int dtype, void* stream) {
    if (!x || !dy || !slabs) return CPC_EINVAL;
    if ((long long)(L_valid - 1) * stride + kw > ldx) return CPC_EINVAL;
    return launch_conv1_bwd(x, dy, slabs, B, C, stride, kw, ldx, L_valid, L_alloc, nblk_t, nblk_b, dtype, (hipStream_t)stream);
}

int cpc_reduce_conv_w(const float* slabs, float* out, int cin, int cout, int kw, int nslab, long long slab_stride, void* stream) {
    if (!slabs || !out) return CPC_EINVAL;
    return launch_reduce_conv_w(slabs, out, cin, cout, kw, nslab, slab_stride, (hipStream_t)stream);
}

int cpc_reduce_conv_w2d(const float* slabs, float* out, int nslab, long long slab_stride, int cout, int cin, int kh, int kw, long long s_dw,
                        long long s_dh, long long s_c, int G, long long s_g, void* stream) {
    return launch_reduce_conv_w2d(slabs, out, nslab, slab_stride, cout, cin, kh, kw, s_dw, s_dh, s_c, G, s_g, (hipStream_t)stream);
}

int cpc_conv_fwd(const void* x, const void* w_fwd, const float* bias, void* y, int B, int Cin, int Cout, int kw, int stride,
                 int Lout_alloc, int Lout_valid, int relu, long long x_tail, int dtype, void* stream) {
    if (!x || !w_fwd || !y || B <= 0 || Lout_alloc <= 0 || Lout_valid > Lout_alloc || kw <= 0 || stride <= 0) return CPC_EINVAL;
    if (x_tail < (long long)std::max(0, kw - stride) * Cin) return CPC_EINVAL;    // the last row reads kw - stride positions past the end
    GemmNT p = {};
    p.A = x; p.Bt = w_fwd; p.C = y; p.bias = bias; p.mask = nullptr;
    p.M = B * Lout_alloc; p.N = Cout; p.K = kw * Cin;
    p.lda = (long long)stride * Cin; p.ldb = p.K; p.ldc = Cout;
    p.c_rpi = Lout_alloc; p.c_item = (long long)Lout_alloc * Cout; p.c_valid = Lout_valid;
    p.flags = (relu ? GEMM_RELU : 0) | (dtype == CPC_DTYPE_F32 ? GEMM_OUT_F32 : 0);
    return launch_gemm_nt(p, dtype, 1, (hipStream_t)stream);
}

int cpc_conv_dgrad(const void* dy, const void* w_dgrad, const void* x_act, void* dx, int B, int Cin, int Cout, int kw,
                   int stride, int Lout_alloc, int Lin_valid, long long dy_head, int dtype, const void* x_act_bits, float* dx_colsum_slabs,
                   void* stream) {
    if (!dy || !w_dgrad || !dx || B <= 0 || Lout_alloc <= 0 || kw <= 0 || stride <= 0) return CPC_EINVAL;
    (void)Lin_valid;   // positions >= Lin_valid receive zeros because the pad rows of dy are zero
    const int D = (kw + stride - 1) / stride;
    if (dy_head < (long long)(D - 1) * Cout) return CPC_EINVAL;                   // the first row starts D - 1 positions before dy
    GemmNT p = {};
    p.A = (const char*)dy - (long long)(D - 1) * Cout * esize(dtype);   // D-1 zero guard rows precede the buffer
    p.Bt = w_dgrad; p.C = dx; p.bias = nullptr; p.mask = x_act;
    p.mask_bits = (const unsigned char*)x_act_bits;
    p.colsum_slabs = dx_colsum_slabs;
    p.M = B * Lout_alloc; p.N = stride * Cin; p.K = D * Cout;
    p.lda = Cout; p.ldb = p.K; p.ldc = (long long)stride * Cin;
    p.flags = dtype == CPC_DTYPE_F32 ? GEMM_OUT_F32 : 0;
    return launch_gemm_nt(p, dtype, 1, (hipStream_t)stream);
}

long long cpc_conv_dgrad_colsum_floats(int B, int Cin, int stride, int Lout_alloc) {
    return (((long long)B * Lout_alloc + 255) / 256) * stride * Cin;
}

long long cpc_conv_dgrad_conv1_floats(int B, int Cin, int stride, int Lout_alloc, int kw1, int what) {
    const long long numM = ((long long)B * Lout_alloc + 255) / 256, numN = (long long)stride * Cin / 256;
    if (what == 0) return numM * numN * (kw1 + 1) * 256;      // slabs
    return 128LL * (kw1 + 1) * Cin;                           // reduction workspace (C1F_CHUNKS partial sums)
}

int cpc_conv_dgrad_conv1(const void* dy, const void* w_dgrad, const void* x_act, const float* x, float* slabs, int B, int Cin,
                         int Cout, int kw, int stride, int Lout_alloc, long long ldx, int kw1, int stride1, int L1_valid,
                         long long dy_head, int dtype, const void* x_act_bits, void* stream) {
    if (!dy || !w_dgrad || !(x_act || x_act_bits) || !x || !slabs || B <= 0 || Lout_alloc <= 0 || dtype != CPC_DTYPE_BF16) return CPC_EINVAL;
    if (Cin % 256 || kw1 < 1 || kw1 > 15 || stride1 < 1 || L1_valid < 1 || kw <= 0 || stride <= 0) return CPC_EINVAL;
    const int D = (kw + stride - 1) / stride;
    if (dy_head < (long long)(D - 1) * Cout) return CPC_EINVAL;
    GemmNT p = {};
    p.A = (const char*)dy - (long long)(D - 1) * Cout * esize(dtype);
    p.Bt = w_dgrad; p.C = slabs /* not written */; p.bias = nullptr; p.mask = x_act;
    p.mask_bits = (const unsigned char*)x_act_bits;
    p.M = B * Lout_alloc; p.N = stride * Cin; p.K = D * Cout;
    p.lda = Cout; p.ldb = p.K; p.ldc = (long long)stride * Cin;
    p.flags = GEMM_EPI_CONV1;
    p.c1_x = x; p.c1_ldx = ldx; p.c1_slabs = slabs;
    p.c1_rpi = Lout_alloc; p.c1_sub = stride; p.c1_stride = stride1; p.c1_kw = kw1; p.c1_valid = L1_valid;
    return launch_gemm_nt(p, dtype, 1, (hipStream_t)stream);
}

int cpc_conv1_fused_reduce(const float* slabs, float* tmp, float* dw, float* db, int B, int Cin, int stride, int Lout_alloc,
                           int kw1, void* stream) {
    if (B <= 0 || Lout_alloc <= 0) return CPC_EINVAL;
    const int numM = (int)(((long long)B * Lout_alloc + 255) / 256);
    return launch_conv1_fused_reduce(slabs, tmp, dw, db, numM, Cin, stride, kw1, (hipStream_t)stream);
}

int cpc_conv_wgrad(const void* x, const void* dy, float* slabs, int B, int Cin, int Cout, int kw, int stride, int Lout_alloc,
                   int nsplit, long long x_tail, int dtype, void* stream) {
    if (!x || !dy || !slabs || B <= 0 || Lout_alloc <= 0 || nsplit <= 0 || kw <= 0 || stride <= 0) return CPC_EINVAL;
    if (x_tail < (long long)std::max(0, kw - stride) * Cin) return CPC_EINVAL;
    GemmTN p = {};
    p.A = x; p.B = dy; p.C = slabs;
    p.M = B * Lout_alloc; p.I = kw * Cin; p.J = Cout;
    p.lda = (long long)stride * Cin; p.ldb = Cout; p.ldc = Cout;
    const int blk = dtype == CPC_DTYPE_BF16 ? 64 : 32;
    int chunk = (p.M + nsplit - 1) / nsplit;
    chunk = (chunk + blk - 1) / blk * blk;
    p.m_chunk = chunk;
    p.slab_stride = (long long)p.I * p.J;
    p.flags = GEMM_OUT_F32;
    return launch_gemm_tn(p, dtype, nsplit, 1, (hipStream_t)stream);
}

// ---- the data-gradient launches on rows [row_lo, row_hi) of every item (engine.CPCEngine: the rows only the targets need are
// differentiated beside the GRU's backward recurrence).  A data-gradient GEMM row q produces input positions q s .. q s + s - 1 from output-
// gradient rows q - D + 1 .. q.
static bool rows_ok(int row_lo, int row_hi, int Lout_alloc) { return row_lo >= 0 && row_lo < row_hi && row_hi <= Lout_alloc; }

int cpc_conv_dgrad_rows(const void* dy, const void* w_dgrad, const void* x_act, void* dx, int B, int Cin, int Cout, int kw, int stride,
                        int Lout_alloc, long long dy_head, int dtype, const void* x_act_bits, float* dx_colsum_slabs, int row_lo, int row_hi,
                        void* stream) {
    if (!dy || !w_dgrad || !dx || B <= 0 || Lout_alloc <= 0 || kw <= 0 || stride <= 0 || !rows_ok(row_lo, row_hi, Lout_alloc)) return CPC_EINVAL;
    const int D = (kw + stride - 1) / stride;
    if (dy_head < (long long)(D - 1) * Cout) return CPC_EINVAL;
    const int rows = row_hi - row_lo;
    const long long es = esize(dtype), c0 = (long long)row_lo * stride * Cin;          // element offset of the first output row in dx / x_act
    GemmNT p = {};
    p.A = (const char*)dy + ((long long)row_lo - (D - 1)) * Cout * es;
    p.Bt = w_dgrad; p.C = (char*)dx + c0 * es; p.bias = nullptr;
    p.mask = x_act ? (const char*)x_act + c0 * es : nullptr;
    p.mask_bits = x_act_bits ? (const unsigned char*)x_act_bits + c0 / 8 : nullptr;
    p.colsum_slabs = dx_colsum_slabs;
    p.M = B * rows; p.N = stride * Cin; p.K = D * Cout;
    p.lda = Cout; p.ldb = p.K; p.ldc = (long long)stride * Cin;
    p.a_rpi = rows; p.a_item = (long long)Lout_alloc * Cout;
    p.c_rpi = rows; p.c_item = (long long)Lout_alloc * stride * Cin; p.c_valid = rows;
    p.flags = (dtype == CPC_DTYPE_F32 ? GEMM_OUT_F32 : 0) | (dx_colsum_slabs ? GEMM_BIG_TILE : 0);
    return launch_gemm_nt(p, dtype, 1, (hipStream_t)stream);
}

int cpc_conv_dgrad_conv1_rows(const void* dy, const void* w_dgrad, const void* x_act, const float* x, float* slabs, int B, int Cin,
                              int Cout, int kw, int stride, int Lout_alloc, long long ldx, int kw1, int stride1, int L1_valid,
                              long long dy_head, int dtype, const void* x_act_bits, int row_lo, int row_hi, void* stream) {
    if (!dy || !w_dgrad || !(x_act || x_act_bits) || !x || !slabs || B <= 0 || Lout_alloc <= 0 || dtype != CPC_DTYPE_BF16) return CPC_EINVAL;
    if (Cin % 256 || kw1 < 1 || kw1 > 15 || stride1 < 1 || L1_valid < 1 || kw <= 0 || stride <= 0 || !rows_ok(row_lo, row_hi, Lout_alloc)) return CPC_EINVAL;
    const int D = (kw + stride - 1) / stride;
    if (dy_head < (long long)(D - 1) * Cout) return CPC_EINVAL;
    const int rows = row_hi - row_lo;
    const long long es = esize(dtype), c0 = (long long)row_lo * stride * Cin;
    GemmNT p = {};
    p.A = (const char*)dy + ((long long)row_lo - (D - 1)) * Cout * es;
    p.Bt = w_dgrad; p.C = slabs /* not written */; p.bias = nullptr;
    p.mask = x_act ? (const char*)x_act + c0 * es : nullptr;
    p.mask_bits = x_act_bits ? (const unsigned char*)x_act_bits + c0 / 8 : nullptr;
    p.M = B * rows; p.N = stride * Cin; p.K = D * Cout;
    p.lda = Cout; p.ldb = p.K; p.ldc = (long long)stride * Cin;
    p.a_rpi = rows; p.a_item = (long long)Lout_alloc * Cout;
    p.c_rpi = rows; p.c_item = (long long)Lout_alloc * stride * Cin; p.c_valid = rows;
    p.flags = GEMM_EPI_CONV1 | GEMM_BIG_TILE;
    p.c1_x = x; p.c1_ldx = ldx; p.c1_slabs = slabs;
    p.c1_rpi = rows; p.c1_row0 = row_lo; p.c1_sub = stride; p.c1_stride = stride1; p.c1_kw = kw1; p.c1_valid = L1_valid;
    return launch_gemm_nt(p, dtype, 1, (hipStream_t)stream);
}

int cpc_conv1_fused_reduce_tiles(const float* slabs, float* tmp, float* dw, float* db, int num_row_tiles, int Cin, int stride, int kw1,
                                 void* stream) {
    if (num_row_tiles <= 0) return CPC_EINVAL;
    return launch_conv1_fused_reduce(slabs, tmp, dw, db, num_row_tiles, Cin, stride, kw1, (hipStream_t)stream);
}

static_assert(sizeof(cpc_conv_prep_job) == 56, "cpc_conv_prep_job layout (three pointers + eight ints)");
int cpc_conv_w_prep_plan(cpc_conv_prep_job* jobs, int njobs, int* total_blocks, int* lds_bytes) {
    return conv_w_prep_plan(jobs, njobs, total_blocks, lds_bytes);
}
int cpc_conv_w_prep_batch(const cpc_conv_prep_job* jobs_dev, int njobs, int total_blocks, int lds_bytes, int dtype, void* stream) {
    return launch_conv_w_prep_batch(jobs_dev, njobs, total_blocks, lds_bytes, dtype, (hipStream_t)stream);
}
int cpc_conv_w_prep_group(const float* w, const float* bias, void* w_fwd, void* w_dgrad, float* bias_g, int Cout, int Cin, int kh, int G,
                          int Rw, int Rd, int dtype, void* stream) {
    return launch_conv_w_prep_group(w, bias, w_fwd, w_dgrad, bias_g, Cout, Cin, kh, G, Rw, Rd, dtype, (hipStream_t)stream);
}
int cpc_conv_w_prep(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int kw, int stride, int dtype, void* stream) {
    if (!w || (!w_fwd && !w_dgrad)) return CPC_EINVAL;
    return launch_conv_w_prep(w, w_fwd, w_dgrad, Cout, Cin, kw, stride, dtype, (hipStream_t)stream);
}

int cpc_maxpool_fwd(const void* in, void* out, int B, int C, int pool, int Lin_valid, int Lin_alloc, int Lout_valid, int Lout_alloc,
                    int dtype, void* stream) {
    if (!in || !out) return CPC_EINVAL;
    return launch_maxpool_fwd(in, out, B, C, pool, Lin_valid, Lin_alloc, Lout_valid, Lout_alloc, dtype, (hipStream_t)stream);
}

int cpc_maxpool_bwd(const void* in, const void* dout, void* din, int B, int C, int pool, int Lin_valid, int Lin_alloc, int Lout_alloc,
                    int dtype, void* stream) {
    if (!in || !dout || !din) return CPC_EINVAL;
    return launch_maxpool_bwd(in, dout, din, B, C, pool, Lin_valid, Lin_alloc, Lout_alloc, dtype, (hipStream_t)stream);
}

int cpc_relu_row_bwd(const float* dc, const void* y, void* dy, int B, int C, long long item_stride, long long row_off, int dtype,
                     void* stream) {
    if (!dc || !y || !dy) return CPC_EINVAL;
    return launch_relu_row_bwd(dc, y, dy, B, C, item_stride, row_off, dtype, (hipStream_t)stream);
}

int cpc_pe_scale_fwd(const void* top, const float* pe, void* x0, int B, int S, int C, long long item_stride, float scale,
                     int dtype, void* stream) {
    if (!top || !pe || !x0) return CPC_EINVAL;
    return launch_pe_scale_fwd(top, pe, x0, B, S, C, item_stride, scale, dtype, (hipStream_t)stream);
}

int cpc_pe_scale_bwd(const void* g1, const void* g2, void* dtop, int B, int S, int C, long long item_stride, float scale,
                     int dtype, void* stream) {
    if (!g1 || !dtop) return CPC_EINVAL;
    return launch_pe_scale_bwd(g1, g2, dtop, B, S, C, item_stride, scale, dtype, (hipStream_t)stream);
}

int cpc_attn_fwd(const void* qkv, void* out, void* P, int B, int S, int C, int heads, float drop_p, unsigned long long seed,
                 unsigned site, int dtype, void* stream) {
    if (!qkv || !out || !P) return CPC_EINVAL;
    return launch_attn_fwd(qkv, out, P, B, S, C, heads, drop_p, seed, site, dtype, (hipStream_t)stream);
}

int cpc_attn_bwd(const void* qkv, const void* P, const void* dout, void* dqkv, int B, int S, int C, int heads, float drop_p,
                 unsigned long long seed, unsigned site, int dtype, void* stream) {
    if (!qkv || !P || !dout || !dqkv) return CPC_EINVAL;
    return launch_attn_bwd(qkv, P, dout, dqkv, B, S, C, heads, drop_p, seed, site, dtype, (hipStream_t)stream);
}

int cpc_add_ln_fwd(const void* a, const void* b, const float* w, const float* bias, void* r_out, void* y, float* stats, int M,
                   int C, float eps, float drop_p, unsigned long long seed, unsigned site, int dtype, void* stream) {
    if (!a || !y || !stats) return CPC_EINVAL;
    return launch_add_ln_fwd(a, b, w, bias, r_out, y, stats, M, C, eps, drop_p, seed, site, dtype, (hipStream_t)stream);
}

int cpc_ln_bwd(const void* g1, const void* g2, const void* r, const float* stats, const float* w, void* dr, float* slabs, int M,
               int C, int bcast, float gscale, int nblocks, void* dr_b, float drop_p, unsigned long long seed, unsigned site,
               int dtype, void* stream) {
    if (!g1 || !r || !stats || !w || !dr || !slabs) return CPC_EINVAL;
    return launch_ln_bwd(g1, g2, r, stats, w, dr, slabs, M, C, bcast, gscale, nblocks, dr_b, drop_p, seed, site, dtype,
                         (hipStream_t)stream);
}

int cpc_ln_tangent(const float* at, const float* bt, const float* r, const float* stats, const float* w, float* rt_out, float* yt,
                   int M, int C, float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (!at || !r || !stats || !w || !yt) return CPC_EINVAL;
    return launch_ln_tangent(at, bt, r, stats, w, rt_out, yt, M, C, drop_p, seed, site, (hipStream_t)stream);
}

int cpc_ln_gp(const float* g1, const float* g2, const float* rt, const float* r, const float* stats, const float* w, float* dr,
              float* dr_b, float* slabs, int M, int C, int bcast, float gscale, int nblocks, float drop_p, unsigned long long seed,
              unsigned site, void* stream) {
    if (!g1 || !rt || !r || !stats || !w || !dr || !slabs) return CPC_EINVAL;
    return launch_ln_gp(g1, g2, rt, r, stats, w, dr, dr_b, slabs, M, C, bcast, gscale, nblocks, drop_p, seed, site, (hipStream_t)stream);
}

int cpc_attn_tangent(const float* qkv, const float* qkvt, const float* P, float* out_t, int B, int S, int C, int heads, float drop_p,
                     unsigned long long seed, unsigned site, void* stream) {
    if (!qkv || !qkvt || !P || !out_t) return CPC_EINVAL;
    return launch_attn_tangent(qkv, qkvt, P, out_t, B, S, C, heads, drop_p, seed, site, (hipStream_t)stream);
}

int cpc_attn_gp(const float* qkv, const float* qkvt, const float* P, const float* dout, float* dqkv, int B, int S, int C, int heads,
                float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (!qkv || !qkvt || !P || !dout || !dqkv) return CPC_EINVAL;
    return launch_attn_gp(qkv, qkvt, P, dout, dqkv, B, S, C, heads, drop_p, seed, site, (hipStream_t)stream);
}

int cpc_dropout(void* x, long long n, float drop_p, unsigned long long seed, unsigned site, int dtype, void* stream) {
    if (!x) return CPC_EINVAL;
    return launch_dropout(x, n, drop_p, seed, site, dtype, (hipStream_t)stream);
}

int cpc_dropout_mask(float* mask, long long n, float drop_p, unsigned long long seed, unsigned site, void* stream) {
    if (!mask) return CPC_EINVAL;
    return launch_dropout_mask(mask, n, drop_p, seed, site, (hipStream_t)stream);
}

int cpc_mean_time(const void* x, void* out, int B, int S, int C, int dtype, void* stream) {
    if (!x || !out) return CPC_EINVAL;
    return launch_mean_time(x, out, B, S, C, dtype, (hipStream_t)stream);
}

int cpc_scalogram_pointwise(const float* cq, const float* fixed_pd, const float* pd_scale, float* out, int B, int Tn, int bins,
                            long long ldq, int phase, float offset, float log_offset, float norm, float power, int ph, int pw,
                            void* stream) {
    if (!cq || !out) return CPC_EINVAL;
    return launch_scalogram_pointwise(cq, fixed_pd, pd_scale, out, B, Tn, bins, ldq, phase, offset, log_offset, norm, power, ph, pw,
                                      (hipStream_t)stream);
}

int cpc_im2col2d(const void* in, void* col, const int* grid, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp,
                 int in_f32, int dtype, void* stream) {
    if (!in || !col) return CPC_EINVAL;
    return launch_im2col2d(in, col, grid, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp, in_f32, dtype, (hipStream_t)stream);
}

int cpc_col2im2d(const void* dcol, void* din, const int* grid, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp,
                 int accumulate, int dtype, void* stream) {
    if (!dcol || !din) return CPC_EINVAL;
    return launch_col2im2d(dcol, din, grid, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp, accumulate, dtype, (hipStream_t)stream);
}

int cpc_dw_fwd(const void* col, const float* w, void* y, long long M, int C, int taps, int Kp, int rpi, long long item, int dtype,
               void* stream) {
    if (!col || !w || !y) return CPC_EINVAL;
    return launch_dw_fwd(col, w, y, M, C, taps, Kp, rpi, item, dtype, (hipStream_t)stream);
}
int cpc_dw_bwd_col(const void* dy, const float* w, void* dcol, long long M, int C, int taps, int Kp, int rpi, long long item, int dtype,
                   void* stream) {
    if (!dy || !w || !dcol) return CPC_EINVAL;
    return launch_dw_bwd_col(dy, w, dcol, M, C, taps, Kp, rpi, item, dtype, (hipStream_t)stream);
}
int cpc_dw_bwd_w(const void* col, const void* dy, float* slabs, long long M, int C, int taps, int Kp, int rpi, long long item,
                 int nblocks, int dtype, void* stream) {
    if (!col || !dy || !slabs) return CPC_EINVAL;
    return launch_dw_bwd_w(col, dy, slabs, M, C, taps, Kp, rpi, item, nblocks, dtype, (hipStream_t)stream);
}

int cpc_bn_stats(const void* x, float* slabs, long long rows, int C, int nblocks, int dtype, void* stream) {
    if (!x || !slabs) return CPC_EINVAL;
    return launch_bn_stats(x, slabs, rows, C, nblocks, dtype, (hipStream_t)stream);
}

int cpc_bn_finalize(const float* slabs, int nslab, int C, double count, float eps, float momentum, float* stats, float* run_mean,
                    float* run_var, void* stream) {
    if (!slabs || !stats) return CPC_EINVAL;
    return launch_bn_finalize(slabs, nslab, C, count, eps, momentum, stats, run_mean, run_var, (hipStream_t)stream);
}

int cpc_bn_apply_residual(const void* x, const int* gx, const void* res, const int* gr, void* out, const int* go, const float* stats,
                          const float* gamma, const float* beta, int oh, int ow, int relu_in, int relu_out, int r_f32, unsigned char* bits,
                          const int* ga, unsigned char* obits, int dtype, void* stream) {
    return launch_bn_apply_residual(x, gx, res, gr, out, go, stats, gamma, beta, oh, ow, relu_in, relu_out, r_f32, bits, ga, dtype, (hipStream_t)stream,
                                    obits);
}
int cpc_bn_bwd_reduce_res(const void* dout, const int* gd, const unsigned char* obits, const unsigned char* abits, const int* ga, const void* x,
                          const int* gx, const float* stats, float* slabs, int nblocks, int dtype, void* stream) {
    return launch_bn_bwd_reduce_res(dout, gd, obits, abits, ga, x, gx, stats, slabs, nblocks, dtype, (hipStream_t)stream);
}
int cpc_bn_bwd_apply_res(const void* dout, const int* gd, const unsigned char* obits, const unsigned char* abits, const int* ga, const void* x,
                         void* dx, const int* gx, const float* stats, const float* gamma, const float* dgamma, const float* dbeta, double count,
                         int train, void* dres, const int* gr, int oh, int ow, int dtype, void* stream) {
    return launch_bn_bwd_apply_res(dout, gd, obits, abits, ga, x, dx, gx, stats, gamma, dgamma, dbeta, count, train, dres, gr, oh, ow, dtype,
                                   (hipStream_t)stream);
}
int cpc_bn_apply(const void* x, const int* gx, void* out, const int* go, const float* stats, const float* gamma, const float* beta,
                 int relu, int x_f32, int dtype, void* stream) {
    if (!x || !out || !stats || !gamma || !beta) return CPC_EINVAL;
    return launch_bn_apply(x, gx, out, go, stats, gamma, beta, relu, x_f32, dtype, (hipStream_t)stream);
}

int cpc_bn_bwd_reduce(const void* dy, const void* y, const int* gy, const void* x, const int* gx, const float* stats, float* slabs,
                      int relu, int nblocks, int x_f32, int dtype, void* stream) {
    if (!dy || !x || !stats || !slabs || (relu && !y)) return CPC_EINVAL;
    return launch_bn_bwd_reduce(dy, y, gy, x, gx, stats, slabs, relu, nblocks, x_f32, dtype, (hipStream_t)stream);
}

int cpc_bn_bwd_apply(const void* dy, const void* y, const int* gy, const void* x, void* dx, const int* gx, const float* stats,
                     const float* gamma, const float* dgamma, const float* dbeta, double count, int relu, int train, int x_f32,
                     int dtype, void* stream) {
    if (!dy || !x || !dx || !stats || !gamma || (relu && !y)) return CPC_EINVAL;
    return launch_bn_bwd_apply(dy, y, gy, x, dx, gx, stats, gamma, dgamma, dbeta, count, relu, train, x_f32, dtype, (hipStream_t)stream);
}

int cpc_maxpool2d_fwd(const void* in, const int* gi, void* out, const int* go, int p, int in_f32, int dtype, void* stream) {
    if (!in || !out) return CPC_EINVAL;
    return launch_maxpool2d_fwd(in, gi, out, go, p, in_f32, dtype, (hipStream_t)stream);
}

int cpc_maxpool2d_bwd(const void* in, void* din, const int* gi, const void* dout, const int* go, int p, int accumulate, int dtype,
                      void* stream) {
    if (!in || !din || !dout) return CPC_EINVAL;
    return launch_maxpool2d_bwd(in, din, gi, dout, go, p, accumulate, dtype, (hipStream_t)stream);
}

int cpc_residual_add(const void* a, const int* ga, const void* r, const int* gr, void* out, const int* go, int oh, int ow, int relu,
                     int r_f32, int dtype, void* stream) {
    if (!a || !r || !out) return CPC_EINVAL;
    return launch_residual_add(a, ga, r, gr, out, go, oh, ow, relu, r_f32, dtype, (hipStream_t)stream);
}

int cpc_residual_add_bwd(const void* dout, const void* out, const int* go, void* da, const int* ga, void* dr, const int* gr, int oh,
                         int ow, int relu, int r_f32, int dtype, void* stream) {
    if (!dout || !da || !dr || (relu && !out)) return CPC_EINVAL;
    return launch_residual_add_bwd(dout, out, go, da, ga, dr, gr, oh, ow, relu, r_f32, dtype, (hipStream_t)stream);
}

int cpc_stem_supported(int cin, int cout, int kh, int kw, int sh, int hin, int ph) { return launch_stem_supported(cin, cout, kh, kw, sh, hin, ph); }

int cpc_stem_stats(const float* x, const int* gx, const float* w, const float* bias, const int* conv, float* slabs, int nblocks,
                   void* stream) {
    if (!x || !gx || !w || !conv || !slabs) return CPC_EINVAL;
    return launch_stem_stats(x, gx, w, bias, conv[0], conv[1], conv[2], conv[3], conv[4], conv[5], conv[6], conv[7], conv[8], slabs, nblocks,
                             (hipStream_t)stream);
}

int cpc_stem_apply(const float* x, const int* gx, const float* w, const float* bias, const int* conv, const float* stats,
                   const float* gamma, const float* beta, void* out, const int* go, int nblocks, int dtype, void* stream) {
    if (!x || !gx || !w || !conv || !stats || !gamma || !beta || !out || !go) return CPC_EINVAL;
    return launch_stem_apply(x, gx, w, bias, conv[0], conv[1], conv[2], conv[3], conv[4], conv[5], conv[6], conv[7], conv[8], stats, gamma, beta,
                             out, go, nblocks, dtype, (hipStream_t)stream);
}

int cpc_stem_bwd_reduce(const float* x, const int* gx, const float* w, const float* bias, const int* conv, const float* stats,
                        const void* da, const void* a, const int* ga, float* slabs, int nblocks, int dtype, void* stream) {
    if (!x || !gx || !w || !conv || !stats || !da || !a || !ga || !slabs) return CPC_EINVAL;
    return launch_stem_bwd_reduce(x, gx, w, bias, conv[0], conv[1], conv[2], conv[3], conv[4], conv[5], conv[6], conv[7], conv[8], stats, da, a,
                                  ga, slabs, nblocks, dtype, (hipStream_t)stream);
}

int cpc_stem_bwd_wgrad(const float* x, const int* gx, const float* w, const float* bias, const int* conv, const float* stats,
                       const float* gamma, const float* dgamma, const float* dbeta, double count, const void* da, const void* a,
                       const int* ga, float* slabs, int nblocks, int dtype, void* stream) {
    if (!x || !gx || !w || !conv || !stats || !gamma || !dgamma || !dbeta || !da || !a || !ga || !slabs) return CPC_EINVAL;
    return launch_stem_bwd_wgrad(x, gx, w, bias, conv[0], conv[1], conv[2], conv[3], conv[4], conv[5], conv[6], conv[7], conv[8], stats, gamma,
                                 dgamma, dbeta, count, da, a, ga, slabs, nblocks, dtype, (hipStream_t)stream);
}

int cpc_stem_residual_bn_add(const void* y, const int* gy, const float* xp, const int* gp, const float* wr, void* out, const int* go, int oh, int ow,
                             int relu, const float* stats, const float* gamma, const float* beta, unsigned char* bits, const int* ga,
                             unsigned char* obits, int dtype, void* stream) {
    if (!y || !xp || !wr || !out) return CPC_EINVAL;
    return launch_stem_residual_bn_add(y, gy, xp, gp, wr, out, go, oh, ow, relu, stats, gamma, beta, bits, ga, dtype, (hipStream_t)stream, obits);
}
int cpc_stem_residual_wgrad_bits(const void* dout, const unsigned char* obits, const int* go, const int* gm, const float* xp, const int* gp,
                                 float* slabs, int oh, int ow, int nblocks, int dtype, void* stream) {
    if (!dout || !obits || !go || !gm || !xp || !gp || !slabs) return CPC_EINVAL;
    return launch_stem_residual_bwd(dout, nullptr, go, nullptr, gm, xp, gp, slabs, oh, ow, 1, nblocks, dtype, (hipStream_t)stream, obits);
}
int cpc_stem_residual_add(const void* main_, const int* gm, const float* xp, const int* gp, const float* wr, void* out, const int* go,
                          int oh, int ow, int relu, int dtype, void* stream) {
    if (!main_ || !gm || !xp || !gp || !wr || !out || !go) return CPC_EINVAL;
    return launch_stem_residual_add(main_, gm, xp, gp, wr, out, go, oh, ow, relu, dtype, (hipStream_t)stream);
}

int cpc_stem_residual_bwd(const void* dout, const void* out, const int* go, void* dmain, const int* gm, const float* xp, const int* gp,
                          float* slabs, int oh, int ow, int relu, int nblocks, int dtype, void* stream) {
    if (!dout || !go || !dmain || !gm || !xp || !gp || !slabs || (relu && !out)) return CPC_EINVAL;
    return launch_stem_residual_bwd(dout, out, go, dmain, gm, xp, gp, slabs, oh, ow, relu, nblocks, dtype, (hipStream_t)stream);
}

int cpc_bn_apply_bits(const void* x, const int* gx, void* out, const int* go, const float* stats, const float* gamma, const float* beta,
                      int relu, void* out_bits, int dtype, void* stream) {
    if (!x || !out || !stats || !gamma || !beta || !out_bits) return CPC_EINVAL;
    return launch_bn_apply(x, gx, out, go, stats, gamma, beta, relu, 0, dtype, (hipStream_t)stream, (unsigned char*)out_bits);
}

int cpc_bn_bwd_reduce_bits(const void* dy, const void* y_bits, const int* gy, const void* x, const int* gx, const float* stats,
                           float* slabs, int nblocks, int dtype, void* stream) {
    if (!dy || !y_bits || !x || !stats || !slabs) return CPC_EINVAL;
    return launch_bn_bwd_reduce(dy, nullptr, gy, x, gx, stats, slabs, 1, nblocks, 0, dtype, (hipStream_t)stream, (const unsigned char*)y_bits);
}

int cpc_bn_bwd_apply_bits(const void* dy, const void* y_bits, const int* gy, const void* x, void* dx, const int* gx, const float* stats,
                          const float* gamma, const float* dgamma, const float* dbeta, double count, int train, int dtype, void* stream) {
    if (!dy || !y_bits || !x || !dx || !stats || !gamma) return CPC_EINVAL;
    return launch_bn_bwd_apply(dy, nullptr, gy, x, dx, gx, stats, gamma, dgamma, dbeta, count, 1, train, 0, dtype, (hipStream_t)stream,
                               (const unsigned char*)y_bits);
}

int cpc_maxpool2d_select(const void* in, const void* sel, const int* gi, void* out, const int* go, int p, int in_f32, int dtype,
                         void* stream) {
    if (!in || !sel || !out) return CPC_EINVAL;
    return launch_maxpool2d_select(in, sel, gi, out, go, p, in_f32, dtype, (hipStream_t)stream);
}

int cpc_gp_direction(const float* g, float* v, long long npix, int C, float factor, float* partial, int nblocks, void* stream) {
    if (!g || !v || !partial) return CPC_EINVAL;
    return launch_gp_direction(g, v, npix, C, factor, partial, nblocks, (hipStream_t)stream);
}

int cpc_bn_gp_cross(const void* x, const void* yt, const void* delta, void* out, const int* gx, const float* stats, const float* coef,
                    int x_f32, int dtype, void* stream) {
    if (!x || !yt || !delta || !out || !stats || !coef) return CPC_EINVAL;
    return launch_bn_gp_cross(x, yt, delta, out, gx, stats, coef, x_f32, dtype, (hipStream_t)stream);
}

int cpc_relu_mask(void* g, const void* y, long long n, int dtype, void* stream) {
    if (!g || !y) return CPC_EINVAL;
    return launch_relu_mask(g, y, n, dtype, (hipStream_t)stream);
}

int cpc_accumulate(void* a, const void* b, long long n, int dtype, void* stream) {
    if (!a || !b) return CPC_EINVAL;
    return launch_accumulate(a, b, n, dtype, (hipStream_t)stream);
}

int cpc_split3_bf16(const float* src, void* dst, long long n, void* stream) {
    if (!src || !dst) return CPC_EINVAL;
    return launch_split3_bf16(src, dst, n, (hipStream_t)stream);
}

int cpc_adam_dev(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float* state,
                 float grad_scale, const float* skip, void* stream) {
    if (!p || !g || !m || !v || !state) return CPC_EINVAL;
    return launch_adam_dev(p, g, m, v, n, lr, b1, b2, eps, state, grad_scale, skip, (hipStream_t)stream);
}

int cpc_cast2d(const float* src, void* dst, int R, int C, long long sr, long long sc, int dtype, void* stream) {
    if (!src || !dst) return CPC_EINVAL;
    return launch_cast2d(src, dst, R, C, sr, sc, dtype, (hipStream_t)stream);
}

int cpc_cast2d_batch(const void* jobs, int njobs, int dtype, void* stream) {
    return launch_cast2d_batch(jobs, njobs, dtype, (hipStream_t)stream);
}

int cpc_prep_frag(const float* src, void* dst, int R, int Kd, long long ld, int transpose, int dtype, void* stream) {
    if (!src || !dst) return CPC_EINVAL;
    return launch_prep_frag(src, dst, R, Kd, ld, transpose, dtype, (hipStream_t)stream);
}

long long cpc_gru_tape_elems(int B, int V, int H, int dtype) { return gru_tape_elems(B, V, H, dtype); }

int cpc_gru_fwd(const void* Gi, const void* Wfrag, const float* bhh, void* Hall, void* tape, float* c_out, int B, int V,
                int H, int dtype, void* stream) {
    if (!Gi || !Wfrag || !Hall || !tape || !c_out) return CPC_EINVAL;
    return launch_gru_fwd(Gi, Wfrag, bhh, Hall, tape, c_out, B, V, H, dtype, (hipStream_t)stream);
}

int cpc_gru_fwd_h0(const void* Gi, const void* Wfrag, const float* bhh, const float* h0, void* Hall, void* tape, float* c_out, int B,
                   int V, int H, int dtype, void* stream) {
    if (!Gi || !Wfrag || !Hall || !tape || !c_out) return CPC_EINVAL;
    return launch_gru_fwd(Gi, Wfrag, bhh, Hall, tape, c_out, B, V, H, dtype, (hipStream_t)stream, h0);
}

int cpc_gru_bwd(const float* dc, const void* tape, const void* WTfrag, void* dG, int B, int V, int H, int dtype, void* stream) {
    if (!dc || !tape || !WTfrag || !dG) return CPC_EINVAL;
    return launch_gru_bwd(dc, tape, WTfrag, dG, B, V, H, dtype, (hipStream_t)stream);
}

int cpc_gru_gp_fwd(const float* Gi, const float* GiT, const float* WT, const float* bhh, float* tape, float* ct_out, int B,
                   int V, int H, void* stream) {
    if (!Gi || !GiT || !WT || !bhh || !tape || !ct_out) return CPC_EINVAL;
    return launch_gru_gp_fwd(Gi, GiT, WT, bhh, tape, ct_out, B, V, H, (hipStream_t)stream);
}

int cpc_gru_gp_bwd(const float* dc, const float* tape, const float* W, float* dA, int B, int V, int H, void* stream) {
    if (!dc || !tape || !W || !dA) return CPC_EINVAL;
    return launch_gru_gp_bwd(dc, tape, W, dA, B, V, H, (hipStream_t)stream);
}

extern int g_gru_force_streaming;
extern int g_gru_waves;
extern int g_gru_debug;
int cpc_gru_set_streaming(int on) {
    const int old = g_gru_force_streaming;
    if (on == 8 || on == 16) { g_gru_waves = on; return old; }      // 8 / 16: waves per workgroup at H = 256 (tuning knob)
    if (on >= 100 && on < 108) { g_gru_debug = on - 100; return old; } // 100 + bits: timing experiments (see gru.hip)
    g_gru_force_streaming = on ? 1 : 0;
    return old;
}

int cpc_debug_set(int key, int value) {
    // Keys 4 / 5 select timing-probe variants of the NT K loop whose RESULTS ARE GARBAGE: honoured only in a process that opted in
    // with CPC_ENABLE_PROBES=1, and announced on stderr once, so that a stray call cannot silently corrupt a training run.
    if (key == 4 || key == 5) {
        const char* on = getenv("CPC_ENABLE_PROBES");
        if (!on || on[0] != '1') {
            fprintf(stderr, "[cpc_hip] cpc_debug_set(%d, %d) refused: timing probes need CPC_ENABLE_PROBES=1 (their results are garbage)\n", key, value);
            return CPC_EINVAL;
        }
        static bool warned = false;
        if (!warned && value != 0) {
            fprintf(stderr, "[cpc_hip] WARNING: NT-GEMM timing probe active (cpc_debug_set %d = %d): results are garbage until it is reset\n", key, value);
            warned = true;
        }
    }
    if (key == 1) { const int old = g_nt_stagger64; g_nt_stagger64 = value; return old; }
    if (key == 4) { const int old = g_nt_probe; g_nt_probe = value; return old; }
    if (key == 6) { const int old = g_nt_wt; g_nt_wt = value; return old; }
    if (key == 5) { const int old = g_nt_probe_taps; g_nt_probe_taps = value > 0 ? value : 1; return old; }
    return CPC_EINVAL;
}

long long cpc_nce_workspace_floats(int B, int K) { return nce_workspace_floats(B, K); }

int cpc_nce_loss(const float* S, void* dS, void* dST, float* out, float* workspace, int B, int K, int ld, int softplus,
                 float regularization, int dtype, void* stream) {
    if (!S || !dS || !dST || !out || !workspace) return CPC_EINVAL;
    return launch_nce(S, dS, dST, out, workspace, B, K, ld, softplus, regularization, dtype, (hipStream_t)stream);
}

int cpc_gp_score_coeff(const float* S, const float* St1, const float* St2, float* W, float* WT, int nmat, int rows, int cols, int ld,
                       int ldT, int mode, void* stream) {
    if (!S || !W || !WT) return CPC_EINVAL;
    return launch_gp_score_coeff(S, St1, St2, W, WT, nmat, rows, cols, ld, ldT, mode, (hipStream_t)stream);
}

long long cpc_nce_all_workspace_floats(int B, int K) { return nce_all_workspace_floats(B, K); }

int cpc_nce_loss_all(const float* S, const float* ST, void* dS, void* dST, float* out, float* workspace, int B, int K, int ld,
                     int softplus, float regularization, int dtype, void* stream) {
    if (!S || !ST || !dS || !dST || !out || !workspace) return CPC_EINVAL;
    return launch_nce_all(S, ST, dS, dST, out, workspace, B, K, ld, softplus, regularization, dtype, (hipStream_t)stream);
}

int cpc_score_lse(const void* P, const void* T, float* S, float* pm, float* ps, float* valid, int M, int N, int E, long long ldp,
                  long long ldt, long long lds, int diag_off, void* stream) {
    return launch_score_lse(P, T, S, pm, ps, valid, M, N, E, ldp, ldt, lds, diag_off, (hipStream_t)stream);
}

int cpc_nce_lse_merge(const float* pm, const float* ps, int nparts, int ncols, int softplus, float nrows_total, float* lse, float* colp,
                      void* stream) {
    return launch_nce_lse_merge(pm, ps, nparts, ncols, softplus, nrows_total, lse, colp, (hipStream_t)stream);
}

long long cpc_nce_fused_grad_blocks(int items, int ncols) { return nce_fused_grad_blocks(items, ncols); }

int cpc_nce_fused_grad(const float* S, const float* lse, void* dS, void* dST, float* gradp, int items, int K, int ncols, long long ld,
                       long long ldT, int diag_off, int softplus, float regularization, float n_rows_total, float n_items_total, void* stream) {
    return launch_nce_fused_grad(S, lse, dS, dST, gradp, items, K, ncols, ld, ldT, diag_off, softplus, regularization, n_rows_total,
                                 n_items_total, (hipStream_t)stream);
}

int cpc_nce_fused_finalize(const float* colp, int ncolp, const float* valid, int nvalid, const float* gradp, int ngrad, float* sums, int mode,
                           float n_rows_total, float n_items_total, int K, float regularization, int softplus, float* out, void* stream) {
    return launch_nce_fused_finalize(colp, ncolp, valid, nvalid, gradp, ngrad, sums, mode, n_rows_total, n_items_total, K, regularization,
                                     softplus, out, (hipStream_t)stream);
}

long long cpc_nce_eval_workspace_floats(int B, int K) { return nce_eval_workspace_floats(B, K); }

int cpc_nce_eval(const float* S, float* out, float* workspace, int B, int K, int ld, int softplus, int all_timesteps, int accumulate,
                 void* stream) {
    if (!S || !out || !workspace) return CPC_EINVAL;
    return launch_nce_eval(S, out, workspace, B, K, ld, softplus, all_timesteps, accumulate, (hipStream_t)stream);
}

int cpc_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, int step,
             float grad_scale, const float* skip, void* stream) {
    if (!p || !g || !m || !v) return CPC_EINVAL;
    return launch_adam(p, g, m, v, n, lr, beta1, beta2, eps, step, grad_scale, skip, (hipStream_t)stream);
}

}  // extern "C"
