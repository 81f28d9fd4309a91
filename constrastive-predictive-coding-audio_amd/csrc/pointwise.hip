// Small streaming kernels of the train step: fused Adam over the flat parameter buffer, and the per-step re-layout of
// the f32 master weights (kept in the reference's state_dict shapes) into the storage-dtype GEMM operand layouts.
#include "cpc_common.h"
#include "cpc_kernels.h"
#include <algorithm>
#include <cstdlib>

namespace {

// torch.optim.Adam (no weight decay, no amsgrad):  m = m + (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).   g is pre-multiplied by grad_scale (1 / world size for DP means).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float step_size, float b1, float b2,
                                                   float eps, float inv_bc2_sqrt, float grad_scale, const float* __restrict__ skip) {
    // NaN guard (contrastive_estimation_training.py:124-133 returns BEFORE backward() / optimizer.step()): the loss kernel raises
    // *skip when the loss is NaN and this update becomes a no-op — parameters and moments keep their last good values
    if (skip && skip[0] != 0.f) return;
    const long long n4 = n / 4;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        // gradient and moments are touched once per step: non-temporal, so that 130 MB of them per step do not push the activations and
        // gradients the backward GEMMs are working on out of the Infinity Cache (the parameters are read again by the layout kernels)
        f32x4 pp = ((f32x4*)p)[i], gg = __builtin_nontemporal_load((const f32x4*)g + i), mm = __builtin_nontemporal_load((f32x4*)m + i),
              vv = __builtin_nontemporal_load((f32x4*)v + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gg[e] * grad_scale;
            mm[e] = mm[e] + (ge - mm[e]) * (1.f - b1);
            vv[e] = vv[e] * b2 + (1.f - b2) * ge * ge;
            const float denom = sqrtf(vv[e]) * inv_bc2_sqrt + eps;
            pp[e] = pp[e] - step_size * (mm[e] / denom);
        }
        ((f32x4*)p)[i] = pp; __builtin_nontemporal_store(mm, (f32x4*)m + i); __builtin_nontemporal_store(vv, (f32x4*)v + i);
    }
    // tail
    for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float ge = g[i] * grad_scale;
        const float mm = m[i] + (ge - m[i]) * (1.f - b1);
        const float vv = v[i] * b2 + (1.f - b2) * ge * ge;
        m[i] = mm; v[i] = vv;
        p[i] = p[i] - step_size * (mm / (sqrtf(vv) * inv_bc2_sqrt + eps));
    }
}

// Conv weight W[co][c][tap] (f32, reference layout) ->
//   fwd  [co][(j, c)]            : gemm_nt Bt operand of the forward conv       (K = kw * Cin)
//   dgrd [(r, c)][(dd, co)]      : gemm_nt Bt operand of the data gradient      (K = D * Cout), tap = r + (D-1-dd)*stride,
//                                  zero where tap >= kw.  D = ceil(kw / stride).
// A workgroup takes a 32 (co) x 32 (c) x kw tile through LDS: W is read along (c, tap) — contiguous — and both layouts are written
// in 64-byte runs (32 c of one (co, j); 32 co of one (r, c, dd)).  The element-per-thread form (4-byte gathers at a stride of kw
// floats, 2 048 workgroups) ran 50-100 us per layer on the side stream beside the backward GEMMs.
template <typename T>
__device__ __forceinline__ void conv_w_prep_tile(const float* __restrict__ W, T* __restrict__ fwd, T* __restrict__ dgrd, int Cout, int Cin,
                                                 int kw, int stride, int D, int tco, int bx, int by, float* tile) {
    const int rs = 32 * kw + 1;                        // tile [tco][32 * kw + 1], tco = 32 (a smaller power of two for very wide kernels: LDS)
    const int c0 = bx * 32, co0 = by * tco;
    const int tid = threadIdx.x;
    const int nc = min(32, Cin - c0), nco = min(tco, Cout - co0);
    for (int i = tid; i < tco * 32 * kw; i += 256) {
        const int col = i / (32 * kw), rem = i % (32 * kw);          // rem = cl * kw + tap: contiguous in W for one co
        float v = 0.f;
        if (col < nco && rem < nc * kw) v = W[((long long)(co0 + col) * Cin + c0) * kw + rem];
        tile[col * rs + rem] = v;
    }
    __syncthreads();
    if (fwd) {
        for (int i = tid; i < tco * kw * 32; i += 256) {
            const int cl = i % 32, j = (i / 32) % kw, col = i / (32 * kw);
            if (col < nco && cl < nc) fwd[((long long)(co0 + col) * kw + j) * Cin + c0 + cl] = from_f32<T>(tile[col * rs + cl * kw + j]);
        }
    }
    if (dgrd) {
        for (int i = tid; i < stride * 32 * D * tco; i += 256) {
            const int col = i % tco, dd = (i / tco) % D, cl = (i / (tco * D)) % 32, r = i / (tco * D * 32);
            const int tap = r + (D - 1 - dd) * stride;
            if (col < nco && cl < nc)
                dgrd[(((long long)r * Cin + c0 + cl) * D + dd) * Cout + co0 + col] = from_f32<T>(tap < kw ? tile[col * rs + cl * kw + tap] : 0.f);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_w_prep_kernel(const float* __restrict__ W, T* __restrict__ fwd,
                                                          T* __restrict__ dgrd, int Cout, int Cin, int kw, int stride, int D, int tco) {
    extern __shared__ float tile[];
    conv_w_prep_tile<T>(W, fwd, dgrd, Cout, Cin, kw, stride, D, tco, blockIdx.x, blockIdx.y, tile);
}

// The same for a list of convolutions in ONE launch (a context network's eleven kernels of 5 x 512 x 512 took eleven launches of 20 us on
// the queue in front of the step).  jobs[j] (device) = {W, fwd, dgrd, Cout, Cin, kw, stride, D, tco, gx, first}: the job's workgroups are
// [first, first + gx * gy) of the grid, (bx, by) = (local % gx, local / gx).
struct ConvPrepJob { const float* W; void* fwd; void* dgrd; int Cout, Cin, kw, stride, D, tco, gx, first; };
template <typename T>
__global__ __launch_bounds__(256) void conv_w_prep_batch_kernel(const ConvPrepJob* __restrict__ jobs, int njobs) {
    extern __shared__ float tile[];
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].first) ++j;
    const ConvPrepJob q = jobs[j];
    const int local = blockIdx.x - q.first;
    conv_w_prep_tile<T>(q.W, (T*)q.fwd, (T*)q.dgrd, q.Cout, q.Cin, q.kw, q.stride, q.D, q.tco, local % q.gx, local / q.gx, tile);
}

// dst[r][c] = (T) src[r * sr + c * sc]   (dst contiguous [R][C])
template <typename T>
__global__ __launch_bounds__(256) void cast2d_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int C,
                                                     long long sr, long long sc) {
    const long long total = (long long)R * C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long long r = idx / C;
        dst[idx] = from_f32<T>(src[r * sr + (long long)c * sc]);
    }
}

// The same through a 32 x 32 LDS tile: the source is read along whichever of its axes has unit stride (sc == 1: rows; otherwise along r,
// e.g. the transposed copy sr == 1), the destination is written along c.  One element per thread and iteration in cast2d_kernel with
// 64-bit div / mod and — for a transposed copy — a 4-byte read per 2 KiB of stride: 21 - 30 us for the 0.4 M elements of a GRU input
// projection; the tile form is bound by the launch.
template <typename T>
__global__ __launch_bounds__(256) void cast2d_tile_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int C, long long sr,
                                                          long long sc) {
    __shared__ float t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (sc == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            t[ty + 8 * k][tx] = (r < R && c < C) ? src[(long long)r * sr + c] : 0.f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            t[tx][ty + 8 * k] = (r < R && c < C) ? src[(long long)r * sr + (long long)c * sc] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < R && c < C) dst[(long long)r * C + c] = from_f32<T>(t[ty + 8 * k][tx]);
    }
}

// Many cast2d jobs in one launch (the operand-layout copies of a context network: 26 small matrices for a 3-layer transformer):
// jobs[y] = {src, dst, R, C, sr, sc}, all 64-bit, in device memory; grid (x, number of jobs).
struct CastJob { const float* src; void* dst; long long R, C, sr, sc; };
template <typename T>
__global__ __launch_bounds__(256) void cast2d_batch_kernel(const CastJob* __restrict__ jobs) {
    const CastJob j = jobs[blockIdx.y];
    T* dst = (T*)j.dst;
    const long long total = j.R * j.C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long c = idx % j.C, r = idx / j.C;
        dst[idx] = from_f32<T>(j.src[r * j.sr + c * j.sc]);
    }
}

// MaxPool1d(pool, ceil_mode=True) over positions of a channels-last activation (ConvolutionalArBlock, audio_model.py:98-99).
// out[b][p][c] = max_{i < pool, p*pool+i < Lin_valid} in[b][p*pool+i][c];  pad rows (p >= Lout_valid) get zeros.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int C, int pool,
                                                          int Lin_valid, int Lin_alloc, int Lout_valid, int Lout_alloc) {
    const int c4n = C / 4;
    const long long total = (long long)B * Lout_alloc * c4n;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c4 = (int)(idx % c4n);
        const int pp = (int)((idx / c4n) % Lout_alloc);
        const int b = (int)(idx / ((long long)c4n * Lout_alloc));
        f32x4 m = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (pp < Lout_valid) {
            const T* src = in + ((long long)b * Lin_alloc + (long long)pp * pool) * C + c4 * 4;
            m = load4(src);
            for (int i = 1; i < pool && pp * pool + i < Lin_valid; ++i) {
                const f32x4 v = load4(src + (long long)i * C);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        store4(out + ((long long)b * Lout_alloc + pp) * C + c4 * 4, m);
    }
}

// Backward of the pooling: the gradient of a window goes to its FIRST maximal element (torch semantics); everything
// else, and the pad rows, get zeros.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ in, const T* __restrict__ dout, T* __restrict__ din,
                                                          int B, int C, int pool, int Lin_valid, int Lin_alloc, int Lout_alloc) {
    const int c4n = C / 4;
    const long long total = (long long)B * Lin_alloc * c4n;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c4 = (int)(idx % c4n);
        const int x = (int)((idx / c4n) % Lin_alloc);
        const int b = (int)(idx / ((long long)c4n * Lin_alloc));
        f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (x < Lin_valid) {
            const int pp = x / pool, me = x % pool;
            const T* src = in + ((long long)b * Lin_alloc + (long long)pp * pool) * C + c4 * 4;
            const f32x4 d = load4(dout + ((long long)b * Lout_alloc + pp) * C + c4 * 4);
            f32x4 best = load4(src);
            int arg[4] = {0, 0, 0, 0};
            for (int i = 1; i < pool && pp * pool + i < Lin_valid; ++i) {
                const f32x4 v = load4(src + (long long)i * C);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e]) { best[e] = v[e]; arg[e] = i; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = arg[e] == me ? d[e] : 0.f;
        }
        store4(din + ((long long)b * Lin_alloc + x) * C + c4 * 4, g);
    }
}

// dy[b][row][c] = y[b][row][c] > 0 ? dc[b][c] : 0   (gradient entering the last ReLU of the conv context network at the one
// position ConvolutionalArModel.forward returns, audio_model.py:161); dy is otherwise left untouched.
template <typename T>
__global__ __launch_bounds__(256) void relu_row_bwd_kernel(const float* __restrict__ dc, const T* __restrict__ y, T* __restrict__ dy,
                                                           int B, int C, long long item_stride, long long row_off) {
    const int total = B * C;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int b = idx / C, c = idx % C;
        const long long o = (long long)b * item_stride + row_off + c;
        dy[o] = from_f32<T>(to_f32(y[o]) > 0.f ? dc[idx] : 0.f);
    }
}

}  // namespace

// Step counter kept on the device (so that a captured hipGraph can be replayed): state[0] = step count (as float bits of an
// int), state[1] = lr / (1 - b1^t), state[2] = 1 / sqrt(1 - b2^t).
__global__ void adam_tick_kernel(float* __restrict__ state, float lr, float b1, float b2, const float* __restrict__ skip) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (skip && skip[0] != 0.f) return;
    int t = __float_as_int(state[0]) + 1;
    state[0] = __int_as_float(t);
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    state[1] = (float)((double)lr / bc1);
    state[2] = (float)(1.0 / sqrt(bc2));
}
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long long n, const float* __restrict__ state, float b1,
                                                       float b2, float eps, float grad_scale, const float* __restrict__ skip) {
    if (skip && skip[0] != 0.f) return;
    const float step_size = state[1], inv_bc2_sqrt = state[2];
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float ge = g[i] * grad_scale;
        const float mm = m[i] + (ge - m[i]) * (1.f - b1);
        const float vv = v[i] * b2 + (1.f - b2) * ge * ge;
        m[i] = mm; v[i] = vv;
        p[i] = p[i] - step_size * (mm / (sqrtf(vv) * inv_bc2_sqrt + eps));
    }
}

int launch_adam_dev(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float* state,
                    float grad_scale, const float* skip, hipStream_t stream) {
    if (n <= 0 || !state) return CPC_EINVAL;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, state, lr, b1, b2, skip);
    const int blocks = (int)min((long long)2048, (n + 255) / 256);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n, (const float*)state, b1, b2, eps, grad_scale,
                       skip);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step,
                float grad_scale, const float* skip, hipStream_t stream) {
    if (n <= 0 || step < 1) return CPC_EINVAL;
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const int blocks = (int)min((long long)2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n, step_size, b1, b2, eps, inv_bc2_sqrt,
                       grad_scale, skip);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// Operands of a tall (kh,1) convolution computed G output rows per GEMM row (scalogram_engine._col_group): G shifted copies of the kernel
// in a window of Rw (forward) / Rd (data gradient) rows, zero elsewhere.  W f32 [co][c][kh];
//   fwd  [dh][co][r][c]  = W[co][c][r - dh]              for dh <= r < dh + kh      (output row G R + dh reads window rows dh .. dh+kh-1)
//   dgrd [dr][c][q][co]  = W[co][c][kh - 1 - (q - dr)]   for dr <= q < dr + kh      (input row G R + dr receives tap j from dY window row dr + kh-1 - j)
//   bias_g [dh][co] = bias[co]   (bias may be null)
// One thread per output element, the fastest index of each layout along the lanes (the weights are a few MB; this replaces 3 G small
// copy / flip launches per convolution and step).
template <typename T>
__global__ __launch_bounds__(256) void conv_w_prep_group_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                                T* __restrict__ fwd, T* __restrict__ dgrd, float* __restrict__ bias_g,
                                                                int Cout, int Cin, int kh, int G, int Rw, int Rd) {
    const long long nf = (long long)G * Cout * Rw * Cin, nd = (long long)G * Cin * Rd * Cout;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < nf) {
        const int c = (int)(i % Cin), r = (int)((i / Cin) % Rw), co = (int)((i / ((long long)Cin * Rw)) % Cout), dh = (int)(i / ((long long)Cin * Rw * Cout));
        const int j = r - dh;
        fwd[i] = from_f32<T>((j >= 0 && j < kh) ? W[((long long)co * Cin + c) * kh + j] : 0.f);
    } else if (i < nf + nd) {
        const long long k = i - nf;
        const int co = (int)(k % Cout), q = (int)((k / Cout) % Rd), c = (int)((k / ((long long)Cout * Rd)) % Cin), dr = (int)(k / ((long long)Cout * Rd * Cin));
        const int j = kh - 1 - (q - dr);
        dgrd[k] = from_f32<T>((j >= 0 && j < kh) ? W[((long long)co * Cin + c) * kh + j] : 0.f);
    } else if (bias && i < nf + nd + (long long)G * Cout) {
        const long long k = i - nf - nd;
        bias_g[k] = bias[k % Cout];
    }
}

int launch_conv_w_prep_group(const float* W, const float* bias, void* fwd, void* dgrd, float* bias_g, int Cout, int Cin, int kh, int G,
                             int Rw, int Rd, int dtype, hipStream_t stream) {
    if (Cout <= 0 || Cin <= 0 || kh <= 0 || G <= 0 || Rw < kh + G - 1 || Rd < kh + G - 1 || !W || !fwd || !dgrd || (bias && !bias_g))
        return CPC_EINVAL;
    const long long n = (long long)G * Cout * Rw * Cin + (long long)G * Cin * Rd * Cout + (long long)G * Cout;
    const long long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return CPC_EINVAL;
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((conv_w_prep_group_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, stream, W, bias, (bf16_t*)fwd, (bf16_t*)dgrd,
                           bias_g, Cout, Cin, kh, G, Rw, Rd);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((conv_w_prep_group_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, W, bias, (float*)fwd, (float*)dgrd,
                           bias_g, Cout, Cin, kh, G, Rw, Rd);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// Host side of the batched form: fills the launch geometry of every job (D, tco, gx, first workgroup) and returns the grid size and the
// dynamic LDS the launch needs; the caller uploads the table once (operand addresses are stable) and launches with those.
int conv_w_prep_plan(void* jobs_host, int njobs, int* total_blocks, int* lds_bytes) {
    if (!jobs_host || njobs <= 0 || njobs > 4096 || !total_blocks || !lds_bytes) return CPC_EINVAL;
    ConvPrepJob* jobs = (ConvPrepJob*)jobs_host;
    long long first = 0;
    size_t lds = 0;
    for (int j = 0; j < njobs; ++j) {
        ConvPrepJob& q = jobs[j];
        if (q.Cout <= 0 || q.Cin <= 0 || q.kw <= 0 || q.stride <= 0 || q.kw > 511 || !q.W || (!q.fwd && !q.dgrd)) return CPC_EINVAL;
        q.D = (q.kw + q.stride - 1) / q.stride;
        int tco = 32;
        while (tco > 1 && (long long)tco * (32 * q.kw + 1) > 16384) tco >>= 1;
        q.tco = tco;
        q.gx = (q.Cin + 31) / 32;
        q.first = (int)first;
        first += (long long)q.gx * ((q.Cout + tco - 1) / tco);
        if (first > 0x7fffffffLL) return CPC_EINVAL;
        lds = std::max(lds, (size_t)tco * (32 * q.kw + 1) * sizeof(float));
    }
    *total_blocks = (int)first;
    *lds_bytes = (int)lds;
    return CPC_OK;
}

int launch_conv_w_prep_batch(const void* jobs_dev, int njobs, int total_blocks, int lds_bytes, int dtype, hipStream_t stream) {
    if (!jobs_dev || njobs <= 0 || total_blocks <= 0 || lds_bytes <= 0 || lds_bytes > 65536) return CPC_EINVAL;
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((conv_w_prep_batch_kernel<bf16_t>), dim3(total_blocks), dim3(256), (size_t)lds_bytes, stream, (const ConvPrepJob*)jobs_dev, njobs);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((conv_w_prep_batch_kernel<float>), dim3(total_blocks), dim3(256), (size_t)lds_bytes, stream, (const ConvPrepJob*)jobs_dev, njobs);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_conv_w_prep(const float* W, void* fwd, void* dgrd, int Cout, int Cin, int kw, int stride, int dtype,
                       hipStream_t stream) {
    if (Cout <= 0 || Cin <= 0 || kw <= 0 || stride <= 0) return CPC_EINVAL;
    const int D = (kw + stride - 1) / stride;
    if (kw > 511) return CPC_EINVAL;
    int tco = 32;                                                            // LDS tile of at most 64 KiB: fewer co for very wide kernels
    while (tco > 1 && (long long)tco * (32 * kw + 1) > 16384) tco >>= 1;
    // (a 512 x 512 x 8 weight is 256 tiles of 32 x 32 channels: one workgroup per CU, each walking 96 elements per thread through index
    // arithmetic with nothing to hide its latency behind — 65 us for 2 M elements; 4 output channels per tile, eight times the workgroups: 25 us; 45 / 31 / 27 / 25 / 25 us at 32 / 16 / 8 / 4 / 2)
    static const int tco_max = [] { const char* v = getenv("CPC_W_PREP_TCO"); return v ? atoi(v) : 4; }();
    while (tco > tco_max && (long long)((Cin + 31) / 32) * ((Cout + tco - 1) / tco) < 2048) tco >>= 1;
    const dim3 grid((Cin + 31) / 32, (Cout + tco - 1) / tco);
    const size_t lds = (size_t)tco * (32 * kw + 1) * sizeof(float);
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((conv_w_prep_kernel<bf16_t>), grid, dim3(256), lds, stream, W, (bf16_t*)fwd, (bf16_t*)dgrd, Cout,
                           Cin, kw, stride, D, tco);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((conv_w_prep_kernel<float>), grid, dim3(256), lds, stream, W, (float*)fwd, (float*)dgrd, Cout, Cin,
                           kw, stride, D, tco);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_cast2d(const float* src, void* dst, int R, int C, long long sr, long long sc, int dtype, hipStream_t stream) {
    if (R <= 0 || C <= 0) return CPC_EINVAL;
    const long long n = (long long)R * C;
    const int blocks = (int)min((long long)2048, (n + 255) / 256);
    static const int tiled = [] { const char* v = getenv("CPC_CAST2D_TILE"); return v ? atoi(v) : 1; }();
    if (tiled && n >= 4096 && (C + 31) / 32 <= 65535 && (R + 31) / 32 <= 65535) {
        const dim3 grid((C + 31) / 32, (R + 31) / 32);
        if (dtype == CPC_DTYPE_BF16)
            hipLaunchKernelGGL((cast2d_tile_kernel<bf16_t>), grid, dim3(256), 0, stream, src, (bf16_t*)dst, R, C, sr, sc);
        else if (dtype == CPC_DTYPE_F32)
            hipLaunchKernelGGL((cast2d_tile_kernel<float>), grid, dim3(256), 0, stream, src, (float*)dst, R, C, sr, sc);
        else
            return CPC_EINVAL;
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((cast2d_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, src, (bf16_t*)dst, R, C, sr, sc);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((cast2d_kernel<float>), dim3(blocks), dim3(256), 0, stream, src, (float*)dst, R, C, sr, sc);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_cast2d_batch(const void* jobs, int njobs, int dtype, hipStream_t stream) {
    if (!jobs || njobs <= 0 || njobs > 65535) return CPC_EINVAL;
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((cast2d_batch_kernel<bf16_t>), dim3(64, njobs), dim3(256), 0, stream, (const CastJob*)jobs);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((cast2d_batch_kernel<float>), dim3(64, njobs), dim3(256), 0, stream, (const CastJob*)jobs);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_maxpool_fwd(const void* in, void* out, int B, int C, int pool, int Lin_valid, int Lin_alloc, int Lout_valid,
                       int Lout_alloc, int dtype, hipStream_t stream) {
    if (B <= 0 || C <= 0 || C % 4 || pool < 1 || Lin_valid <= 0 || Lin_alloc < Lin_valid || Lout_valid <= 0 || Lout_alloc < Lout_valid ||
        (long long)(Lout_valid - 1) * pool >= Lin_valid)
        return CPC_EINVAL;
    const long long total = (long long)B * Lout_alloc * (C / 4);
    const int blocks = (int)min((long long)2048, (total + 255) / 256);
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, B, C, pool,
                           Lin_valid, Lin_alloc, Lout_valid, Lout_alloc);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)in, (float*)out, B, C, pool,
                           Lin_valid, Lin_alloc, Lout_valid, Lout_alloc);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_maxpool_bwd(const void* in, const void* dout, void* din, int B, int C, int pool, int Lin_valid, int Lin_alloc,
                       int Lout_alloc, int dtype, hipStream_t stream) {
    if (B <= 0 || C <= 0 || C % 4 || pool < 1 || Lin_valid <= 0 || Lin_alloc < Lin_valid || Lout_alloc <= 0 ||
        (Lin_valid - 1) / pool >= Lout_alloc)
        return CPC_EINVAL;
    const long long total = (long long)B * Lin_alloc * (C / 4);
    const int blocks = (int)min((long long)2048, (total + 255) / 256);
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)in, (const bf16_t*)dout,
                           (bf16_t*)din, B, C, pool, Lin_valid, Lin_alloc, Lout_alloc);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)in, (const float*)dout,
                           (float*)din, B, C, pool, Lin_valid, Lin_alloc, Lout_alloc);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_relu_row_bwd(const float* dc, const void* y, void* dy, int B, int C, long long item_stride, long long row_off, int dtype,
                        hipStream_t stream) {
    if (B <= 0 || C <= 0) return CPC_EINVAL;
    const int blocks = min(1024, (B * C + 255) / 256);
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((relu_row_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, dc, (const bf16_t*)y, (bf16_t*)dy, B, C,
                           item_stride, row_off);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((relu_row_bwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, dc, (const float*)y, (float*)dy, B, C,
                           item_stride, row_off);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// ---- sign-bit masks (include/cpc_hip.h, cpc_sign_bits): out[i] bit e = x[8 i + e] > 0.  A block turns 8192 elements into 1 KiB:
// four passes in which thread t takes the 8 elements 2048 q + 8 t (16-byte loads, contiguous over the wave) and writes byte 256 q + t.
template <typename T>
__global__ __launch_bounds__(256) void sign_bits_kernel(const T* __restrict__ x, unsigned char* __restrict__ out, long long n8) {
    const long long nchunk = (n8 + 1023) / 1024;
    for (long long c = blockIdx.x; c < nchunk; c += gridDim.x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long i = c * 1024 + q * 256 + threadIdx.x;          // 8-element group
            if (i >= n8) break;
            unsigned b8 = 0;
            if constexpr (sizeof(T) == 2) {
                const uint4 v = *(const uint4*)(x + i * 8);
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // bf16 > 0  <=>  not zero and sign clear  <=>  (h - 1) < 0x7fff
                    b8 |= (((w4[e] & 0xffffu) - 1u) < 0x7fffu ? 1u : 0u) << (2 * e);
                    b8 |= (((w4[e] >> 16) - 1u) < 0x7fffu ? 1u : 0u) << (2 * e + 1);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) b8 |= ((float)x[i * 8 + e] > 0.f ? 1u : 0u) << e;
            }
            out[i] = (unsigned char)b8;
        }
    }
}

int launch_sign_bits(const void* x, unsigned char* bits, long long n, int dtype, hipStream_t stream) {
    if (n % 32 || ((uintptr_t)x % 16) || ((uintptr_t)bits % 4)) return CPC_EINVAL;
    const long long n8 = n / 8;
    const unsigned grid = (unsigned)std::min<long long>((n8 + 1023) / 1024, 256 * 32);
    if (dtype == CPC_DTYPE_BF16) hipLaunchKernelGGL(sign_bits_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, bits, n8);
    else if (dtype == CPC_DTYPE_F32) hipLaunchKernelGGL(sign_bits_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, bits, n8);
    else return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
