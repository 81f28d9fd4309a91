// Encoder layer 1 (C_in = 1): the genuinely HBM-bound piece of the conv stack.
//   fwd : y[b][t][co] = relu(bias[co] + sum_j x[b][t*s + j] * w[co][j])      x f32 [B][Lx], y T channels-last
//   bwd : dw[co][j] = sum_{b,t} dy[b][t][co] * x[b][t*s + j],  db[co] = sum dy   (dy already relu-masked)
// A lane owns 8 consecutive output channels (one 16-byte bf16 store per position); the raw waveform window of the
// workgroup's positions is staged once in LDS and broadcast-read; weights live in registers.
#include <cstdlib>
#include "cpc_common.h"
#include "cpc_kernels.h"

namespace {

constexpr int C1_MAXK = 16;      // max kernel size of layer 1 handled (reference default: 10)
constexpr int C1_POS = 64;       // output positions per workgroup pass (backward)

template <typename T>
__device__ __forceinline__ void store8(T* dst, const float* v);
template <>
__device__ __forceinline__ void store8<bf16_t>(bf16_t* dst, const float* v) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    // non-temporal: the layer-1 output (956 MB at B = 256) is far larger than L2 + Infinity Cache and is next read from the top;
    // as ordinary stores its dirty lines are still being written back while the layer-2 GEMM starts (step 4.62 -> 4.58 ms)
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(__builtin_bit_cast(v4u, o), (v4u*)dst);
}
template <>
__device__ __forceinline__ void store8<float>(float* dst, const float* v) {
    *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
    *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}
template <typename T>
__device__ __forceinline__ void load8(const T* src, float* v);
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t* src, float* v) {
    bf16x8 i = *(const bf16x8*)src;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)i[e];
}
template <>
__device__ __forceinline__ void load8<float>(const float* src, float* v) {
    f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// grid: (ceil(L_alloc / C1_FPOS), B).  Threads: lanes_per_row = C/8 lanes cover one output row; 256/lanes_per_row rows
// are produced per pass.  WROW (C = 512): a row is exactly one wave, so the row index, its validity and its address offset are
// wave-uniform (scalar registers and branches instead of per-lane ones).  BITS (needs WROW): y_bits, the sign-bit mask of y
// (include/cpc_hip.h at cpc_sign_bits) — the waves then take CONTIGUOUS blocks of rows (wave w rows [w RPW, (w+1) RPW)) instead of
// interleaved ones, so that a wave's bytes are one contiguous piece of memory: collected in its own slice of LDS and written as
// 16-byte chunks at the end, no block barrier, no byte store per row.
constexpr int C1_FPOS = 256;

template <typename T, int KW, bool WROW, bool BITS>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ y, int C,
                                                        int stride, int kw_rt, long long ldx, int L_valid, int L_alloc, int relu,
                                                        unsigned char* __restrict__ y_bits, int row_lo, int row_hi) {
    static_assert(!BITS || WROW, "sign bits: one wave per row");
    const int kw = KW > 0 ? KW : kw_rt;
    __shared__ float xs[C1_FPOS * 8 + C1_MAXK + 8];     // stride <= 8 supported
    __shared__ __attribute__((aligned(16))) unsigned char bimg[BITS ? C1_FPOS * 64 : 16];
    const int b = blockIdx.y;
    const int t0 = row_lo + blockIdx.x * C1_FPOS;          // (rows [row_lo, row_hi) of every item: cpc_conv1_fwd_rows)
    const int tid = threadIdx.x;
    const int lpr = WROW ? 64 : C / 8;
    const int cg = tid % lpr, nrl = 256 / lpr;
    const int rl = WROW ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid / lpr;
    constexpr int RPW = C1_FPOS / 4;

    // stage the input window of positions [t0, t0 + C1_FPOS): samples [t0*stride, (t0+C1_FPOS-1)*stride + kw)
    const int npos = min(C1_FPOS, L_valid - t0);          // valid positions in this block (may be <= 0)
    const int nsamp = npos > 0 ? (npos - 1) * stride + kw : 0;
    const float* xb = x + (long long)b * ldx + (long long)t0 * stride;
    for (int i = tid; i < nsamp; i += 256) xs[i] = xb[i];

    float wr[(KW > 0 ? KW : C1_MAXK)][8];
    float br[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) br[e] = bias ? bias[cg * 8 + e] : 0.f;
#pragma unroll
    for (int j = 0; j < (KW > 0 ? KW : C1_MAXK); ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) wr[j][e] = j < kw ? w[(cg * 8 + e) * kw + j] : 0.f;
    __syncthreads();

    T* yb = y + ((long long)b * L_alloc + t0) * C + cg * 8;
    const int nrows = min(C1_FPOS, row_hi - t0);
    const int niter = BITS ? RPW : (nrows - rl + nrl - 1) / nrl;
    // Pad rows are stored from their own branch (no per-row zero fill of v).  Tried against this loop and not faster: one v_max_f32 per
    // value instead of relu_f's compare / select pair (with a NaN put-back); persistent kernels that stream the rows in memory order
    // (scalar loads of the input samples: 370 us; one vector load per 4-row group + v_readlane: 240 us; this kernel: 199-212 us for
    // 956 MB, a plain fill of the buffer 137-144).  The arithmetic is not what binds it: with one tap instead of ten it takes 194 us.
    for (int k = 0; k < niter; ++k) {
        const int r = BITS ? rl * RPW + k : rl + k * nrl;
        if (r >= nrows) break;
        unsigned b8 = 0;
        if (r < npos) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = br[e];
#pragma unroll
            for (int j = 0; j < (KW > 0 ? KW : C1_MAXK); ++j) {
                const float xv = j < kw ? xs[r * stride + j] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(xv, wr[j][e], v[e]);
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
            }
            store8<T>(yb + (long long)r * C, v);
            if constexpr (BITS) {
                if (relu) {                                    // relu_f output > 0 <=> its bit pattern is not zero
#pragma unroll
                    for (int e = 7; e >= 0; --e) b8 = (b8 << 1) | relu_positive_bit(v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) b8 |= (v[e] > 0.f ? 1u << e : 0u);
                }
            }
        } else {
            const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // pad rows are written as zeros
            store8<T>(yb + (long long)r * C, z);
        }
        if constexpr (BITS) bimg[r * 64 + cg] = (unsigned char)b8;
    }
    if constexpr (BITS) {
        // wave rl wrote rows [rl RPW, ...) itself; the LDS executes a wave's operations in order
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int r0 = rl * RPW, nb = (min(nrows, r0 + RPW) - r0) * 64;
        unsigned char* dst = y_bits + ((long long)b * L_alloc + t0 + r0) * 64;
        for (int i = cg * 16; i < nb; i += 64 * 16) *(uint4*)(dst + i) = *(const uint4*)(bimg + r0 * 64 + i);
    }
}

// grid: (nblk_t, nblk_b): block handles positions [bx*tpb, (bx+1)*tpb) of items by, by+nblk_b, ... and writes a partial
// slab slabs[(by*nblk_t + bx)][(kw+1)][C]  (row kw = bias grad).
template <typename T, int KW>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                        float* __restrict__ slabs, int B, int C, int stride, int kw_rt,
                                                        long long ldx, int L_valid, int L_alloc, int tpb) {
    const int kw = KW > 0 ? KW : kw_rt;
    constexpr int KA = (KW > 0 ? KW : C1_MAXK);
    __shared__ float xs[C1_POS * 8 + C1_MAXK + 8];
    __shared__ float red[256 * 8];                     // cross-row-lane reduction scratch
    const int tid = threadIdx.x;
    const int lpr = C / 8;
    const int cg = tid % lpr, rl = tid / lpr, nrl = 256 / lpr;
    const int t_begin = blockIdx.x * tpb;
    const int t_end = min(L_valid, t_begin + tpb);

    float acc[KA + 1][8];
#pragma unroll
    for (int j = 0; j <= KA; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;

    for (int b = blockIdx.y; b < B; b += gridDim.y)
    for (int t0 = t_begin; t0 < t_end; t0 += C1_POS) {
        const int npos = min(C1_POS, t_end - t0);
        const int nsamp = (npos - 1) * stride + kw;
        const float* xb = x + (long long)b * ldx + (long long)t0 * stride;
        __syncthreads();
        for (int i = tid; i < nsamp; i += 256) xs[i] = xb[i];
        __syncthreads();
        const T* dyb = dy + ((long long)b * L_alloc + t0) * C + cg * 8;
        for (int r = rl; r < npos; r += nrl) {
            float g[8];
            load8<T>(dyb + (long long)r * C, g);
#pragma unroll
            for (int j = 0; j < KA; ++j) {
                const float xv = j < kw ? xs[r * stride + j] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][e] = fmaf(xv, g[e], acc[j][e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[KA][e] += g[e];
        }
    }
    // reduce over the row lanes through LDS (one accumulator row at a time), then write the slab
    float* slab = slabs + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (long long)(kw + 1) * C;
#pragma unroll
    for (int j = 0; j <= KA; ++j) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = acc[j][e];
        __syncthreads();
        const int out_row = j < KA ? j : kw;       // accumulator row KA is the bias gradient -> slab row kw
        if (rl == 0 && (j == KA || j < kw)) {
            float s8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s8[e] = acc[j][e];
            for (int r = 1; r < nrl; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) s8[e] += red[(tid + r * lpr) * 8 + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) slab[(long long)out_row * C + cg * 8 + e] = s8[e];
        }
    }
}

}  // namespace

static bool c1_ok(int C, int stride, int kw) {
    if (C < 8 || C % 8) return false;
    const int lpr = C / 8;
    if (lpr > 256 || (256 % lpr)) return false;
    if (stride < 1 || stride > 8 || kw < 1 || kw > C1_MAXK) return false;
    return true;
}

int launch_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int C, int stride, int kw,
                     long long ldx, int L_valid, int L_alloc, int relu, int dtype, unsigned char* y_bits, hipStream_t stream,
                     int row_lo, int row_hi) {
    if (!c1_ok(C, stride, kw) || B <= 0 || L_valid <= 0 || L_alloc < L_valid) return CPC_EINVAL;
    if (row_hi < 0) row_hi = L_alloc;
    if (row_lo < 0 || row_lo >= row_hi || row_hi > L_alloc) return CPC_EINVAL;
    if (y_bits && (C != 512 || (uintptr_t)y_bits % 16)) return CPC_EINVAL;        // sign bits: one wave per row (see the kernel)
    // 256 output positions per workgroup: the 88 weight / bias registers of a thread are loaded once per workgroup
    dim3 grid((row_hi - row_lo + C1_FPOS - 1) / C1_FPOS, B);
#define LAUNCH(T, KWT) \
    do { \
        if (y_bits) hipLaunchKernelGGL((conv1_fwd_kernel<T, KWT, true, true>), grid, dim3(256), 0, stream, x, w, bias, (T*)y, C, stride, kw, ldx, L_valid, L_alloc, relu, y_bits, row_lo, row_hi); \
        else if (C == 512) hipLaunchKernelGGL((conv1_fwd_kernel<T, KWT, true, false>), grid, dim3(256), 0, stream, x, w, bias, (T*)y, C, stride, kw, ldx, L_valid, L_alloc, relu, y_bits, row_lo, row_hi); \
        else hipLaunchKernelGGL((conv1_fwd_kernel<T, KWT, false, false>), grid, dim3(256), 0, stream, x, w, bias, (T*)y, C, stride, kw, ldx, L_valid, L_alloc, relu, y_bits, row_lo, row_hi); \
    } while (0)
    if (dtype == CPC_DTYPE_BF16) {
        if (kw == 10) LAUNCH(bf16_t, 10); else LAUNCH(bf16_t, 0);
    } else if (dtype == CPC_DTYPE_F32) {
        if (kw == 10) LAUNCH(float, 10); else LAUNCH(float, 0);
    } else {
        return CPC_EINVAL;
    }
#undef LAUNCH
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// slabs: [nblk_b * nblk_t][(kw+1)][C] f32.  Reduce with reduce_slabs (I = kw+1, J = C).
int launch_conv1_bwd(const float* x, const void* dy, float* slabs, int B, int C, int stride, int kw, long long ldx,
                     int L_valid, int L_alloc, int nblk_t, int nblk_b, int dtype, hipStream_t stream) {
    if (!c1_ok(C, stride, kw) || B <= 0 || L_valid <= 0 || L_alloc < L_valid || nblk_t <= 0 || nblk_b <= 0 || nblk_b > B)
        return CPC_EINVAL;
    int tpb = (L_valid + nblk_t - 1) / nblk_t;
    tpb = (tpb + C1_POS - 1) / C1_POS * C1_POS;
    dim3 grid(nblk_t, nblk_b);
#define LAUNCH(T, KWT) \
    hipLaunchKernelGGL((conv1_bwd_kernel<T, KWT>), grid, dim3(256), 0, stream, x, (const T*)dy, slabs, B, C, stride, kw, ldx, L_valid, L_alloc, tpb)
    if (dtype == CPC_DTYPE_BF16) {
        if (kw == 10) LAUNCH(bf16_t, 10); else LAUNCH(bf16_t, 0);
    } else if (dtype == CPC_DTYPE_F32) {
        if (kw == 10) LAUNCH(float, 10); else LAUNCH(float, 0);
    } else {
        return CPC_EINVAL;
    }
#undef LAUNCH
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}


// ---- reduction of the fused data-gradient / layer-1 weight-gradient slabs (gemm.hip, GEMM_EPI_CONV1) ----
// slabs: [numM][numN][(kw+1)][256] with numN = sub * (cin / 256), N-tile nt = r * (cin/256) + chalf.
// Phase 1: tmp[z][j][c] = sum over the M-panels of chunk z and over r of slab[mt][r*(cin/256) + c/256][j][c % 256];
// phase 2: dw[c][j] (reference layout [C][1][kw]) / db[c] = sum_z tmp[z][j][c].  Fixed summation order: deterministic.
#define C1F_CHUNKS 128
// block (z, chalf): 256 threads walk the (kw+1) x 256 floats of a slab tile as float4 slots (16 bytes per lane, coalesced)
__global__ __launch_bounds__(256) void conv1_fused_reduce1_kernel(const float* __restrict__ slabs, float* __restrict__ tmp,
                                                                  int numM, int cin, int sub, int kw) {
    const int nch = cin / 256, numN = sub * nch;
    const int z = blockIdx.x, chalf = blockIdx.y;
    const int per = (numM + C1F_CHUNKS - 1) / C1F_CHUNKS;
    const int m_lo = z * per, m_hi = min(numM, m_lo + per);
    const long long tile = (long long)(kw + 1) * 256;
    const int nslot = (kw + 1) * 64;                       // float4 slots per tile
    // grid.z walks the slots (one slot per thread): three times the workgroups in flight, the same summation order per slot
    const int sl = blockIdx.z * 256 + threadIdx.x;
    if (sl >= nslot) return;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* base = slabs + (long long)chalf * tile + (long long)sl * 4;
    for (int mt = m_lo; mt < m_hi; ++mt) {
        if (sub == 4) {
            f32x4 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = *(const f32x4*)(base + ((long long)mt * numN + r * nch) * tile);
#pragma unroll
            for (int r = 0; r < 4; ++r) s += v[r];
        } else {
            for (int r = 0; r < sub; ++r) s += *(const f32x4*)(base + ((long long)mt * numN + r * nch) * tile);
        }
    }
    const int j = sl / 64, c4 = sl % 64;
    *(f32x4*)(tmp + ((long long)z * (kw + 1) + j) * cin + chalf * 256 + c4 * 4) = s;
}
__global__ __launch_bounds__(256) void conv1_fused_reduce2_kernel(const float* __restrict__ tmp, float* __restrict__ dw,
                                                                  float* __restrict__ db, int cin, int kw) {
    // 64 columns x 4 groups of C1F_CHUNKS / 4 chunks per workgroup: a group's loads are all in flight together (one thread per column
    // walking the 128 chunks one dependent load after the other took 13 us on 22 workgroups); fixed summation order: within a
    // group chunk by chunk, then group 0 .. 3
    __shared__ float part[4][64];
    const int j = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), zg = threadIdx.x >> 6;
    constexpr int ZG = C1F_CHUNKS / 4;
    float s = 0.f;
    if (c < cin) {
        float v[ZG];
#pragma unroll
        for (int u = 0; u < ZG; ++u) v[u] = tmp[((long long)(zg * ZG + u) * (kw + 1) + j) * cin + c];
#pragma unroll
        for (int u = 0; u < ZG; ++u) s += v[u];
    }
    part[zg][threadIdx.x & 63] = s;
    __syncthreads();
    if (zg == 0 && c < cin) {
        s = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        if (j < kw) dw[(long long)c * kw + j] = s;
        else if (db) db[c] = s;
    }
}

int launch_conv1_fused_reduce(const float* slabs, float* tmp, float* dw, float* db, int numM, int cin, int sub, int kw,
                              hipStream_t stream) {
    if (!slabs || !tmp || !dw || numM <= 0 || cin <= 0 || cin % 256 || sub <= 0 || kw <= 0) return CPC_EINVAL;
    hipLaunchKernelGGL(conv1_fused_reduce1_kernel, dim3(C1F_CHUNKS, cin / 256, ((kw + 1) * 64 + 255) / 256), dim3(256), 0, stream, slabs, tmp, numM, cin, sub, kw);
    hipLaunchKernelGGL(conv1_fused_reduce2_kernel, dim3((cin + 63) / 64, kw + 1), dim3(256), 0, stream, tmp, dw, db, cin, kw);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
