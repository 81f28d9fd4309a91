// Scalogram front end and 2-D residual encoder kernels (constant_q_transform.py / scalogram_model.py of the reference).
// The CQT filter bank itself is a set of overlapped-row f32 GEMMs (cpc_gemm_nt); this file holds what is not a GEMM:
//   * the pointwise chain  |.|^2 -> log -> phase difference -> scale -> power            (PreprocessingModule.forward)
//   * im2col / col2im for the windowed 2-D convolutions, on channels-last [B][W][H][C] buffers
//   * BatchNorm (batch statistics) forward / backward, MaxPool2d(ceil), cropped residual add.
#include "cpc_common.h"
#include "cpc_kernels.h"
#include <cstdlib>
#include <algorithm>

namespace {

constexpr float PI_F = 3.14159265358979323846f;

// cq  f32 [B][Tn][ldq]: (re, im) interleaved per bin.   out f32 [B][Wp][Hp][Cc]: W = Tn - 1 and Cc = 2 with phase, W = Tn
// and Cc = 1 without; Wp = W / pw, Hp = bins / ph (F.max_pool2d(x, [ph, pw]), floor mode, applied between the log / phase
// stage and the scaling).  scalogram_model.py:77-97, constant_q_transform.py:36-52, :69-72, :281-285.
__global__ __launch_bounds__(256) void scalogram_pointwise_kernel(const float* __restrict__ cq, const float* __restrict__ fixed_pd,
                                                                  const float* __restrict__ pd_scale, float* __restrict__ out,
                                                                  int B, int Tn, int bins, long long ldq, int phase, float offset,
                                                                  float log_offset, float norm, float power, int ph, int pw) {
    const int W = phase ? Tn - 1 : Tn;
    const int Wp = W / pw, Hp = bins / ph;
    const long long total = (long long)B * Wp * Hp;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int hb = (int)(idx % Hp);
        const int wb = (int)((idx / Hp) % Wp);
        const int b = (int)(idx / ((long long)Hp * Wp));
        float amp_m = -INFINITY, pd_m = -INFINITY;
        for (int dw = 0; dw < pw; ++dw)
            for (int dh = 0; dh < ph; ++dh) {
                const int bin = hb * ph + dh, w = wb * pw + dw;
                const int t1 = phase ? w + 1 : w;
                const float2 z1 = *(const float2*)(cq + ((long long)b * Tn + t1) * ldq + 2 * bin);
                const float mag = sqrtf(z1.x * z1.x + z1.y * z1.y);
                amp_m = fmaxf(amp_m, logf(mag * mag + offset) + log_offset);
                if (phase) {
                    const float2 z0 = *(const float2*)(cq + ((long long)b * Tn + w) * ldq + 2 * bin);
                    float pd = atan2f(z1.y, z1.x) - atan2f(z0.y, z0.x) + fixed_pd[bin];
                    if (pd > PI_F) pd -= 2.f * PI_F;
                    if (pd < -PI_F) pd += 2.f * PI_F;
                    pd_m = fmaxf(pd_m, pd * pd_scale[bin]);
                }
            }
        float amp = amp_m * norm;
        if (power != 1.f) amp = powf(amp, power);
        if (!phase) {
            out[idx] = amp;
            continue;
        }
        float pd = pd_m * norm;
        if (power != 1.f) pd = powf(pd, power);
        *(float2*)(out + idx * 2) = make_float2(amp, pd);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Channels-last 2-D activations: element (b, w, h, c) of a "grid" lives at ((b*W + w)*Ha + top + h)*C + c — the frequency
// axis h is the fast row axis (so that the reference's tall (k,1) kernels are overlapped-row GEMMs), Ha >= top + H rows
// are allocated per (b, w) column, rows outside [top, top + H) stay zero.
struct Grid {
    int B, W, H, Ha, top, C;
};
__device__ __forceinline__ long long grid_off(const Grid& g, int b, int w, int h) {
    return (((long long)b * g.W + w) * g.Ha + g.top + h) * g.C;
}

// col[((b*Wo + wo)*Ho + ho)][(dh*kw + dw)*C + c] = in(b, wo*sw + dw - pw, ho*sh + dh - ph, c), zero outside / for k >= kh*kw*C
template <typename TI, typename T>
__global__ __launch_bounds__(256) void im2col2d_kernel(const TI* __restrict__ in, T* __restrict__ col, Grid g, int kh, int kw,
                                                       int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp) {
    const unsigned total = (unsigned)((long long)g.B * Wo * Ho * Kp);
    const int K = kh * kw * g.C;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int k = (int)(idx % Kp);
        const unsigned row = idx / Kp;
        float v = 0.f;
        if (k < K) {
            const int c = k % g.C, tap = k / g.C, dw = tap % kw, dh = tap / kw;
            const int ho = (int)(row % Ho), wo = (int)((row / Ho) % Wo), b = (int)(row / (unsigned)(Ho * Wo));
            const int h = ho * sh + dh - ph, w = wo * sw + dw - pw;
            if (h >= 0 && h < g.H && w >= 0 && w < g.W) v = to_f32(in[grid_off(g, b, w, h) + c]);
        }
        col[idx] = from_f32<T>(v);
    }
}

// din(b, w, h, c) (+)= sum over taps of dcol[row(b, wo, ho)][(dh*kw + dw)*C + c] with wo*sw + dw - pw == w, ho*sh + dh - ph == h
template <typename T>
__global__ __launch_bounds__(256) void col2im2d_kernel(const T* __restrict__ dcol, T* __restrict__ din, Grid g, int kh, int kw,
                                                       int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp, int accumulate) {
    const unsigned total = (unsigned)((long long)g.B * g.W * g.H * g.C);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c = (int)(idx % g.C);
        const int h = (int)((idx / g.C) % g.H);
        const int w = (int)((idx / (unsigned)(g.C * g.H)) % g.W);
        const int b = (int)(idx / (unsigned)(g.C * g.H * g.W));
        float acc = 0.f;
        for (int dh = 0; dh < kh; ++dh) {
            const int hn = h + ph - dh;
            if (hn < 0 || hn % sh) continue;
            const int ho = hn / sh;
            if (ho >= Ho) continue;
            for (int dw = 0; dw < kw; ++dw) {
                const int wn = w + pw - dw;
                if (wn < 0 || wn % sw) continue;
                const int wo = wn / sw;
                if (wo >= Wo) continue;
                acc += to_f32(dcol[(((long long)b * Wo + wo) * Ho + ho) * Kp + (dh * kw + dw) * g.C + c]);
            }
        }
        const long long o = grid_off(g, b, w, h) + c;
        if (accumulate) acc += to_f32(din[o]);
        din[o] = from_f32<T>(acc);
    }
}

// 16-byte versions (C and Kp multiples of 16 B / sizeof(T), same element type on both sides): one thread moves a whole
// chunk of 8 bf16 / 4 f32 channels of one tap.
template <typename T>
__global__ __launch_bounds__(256) void im2col2d_vec_kernel(const T* __restrict__ in, T* __restrict__ col, Grid g, int kh, int kw,
                                                           int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp) {
    constexpr int V = 16 / (int)sizeof(T);
    const int cpr = Kp / V, cpt = g.C / V;                      // chunks per col row, per tap
    const unsigned total = (unsigned)((long long)g.B * Wo * Ho * cpr);
    const int K = kh * kw * g.C;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int q = (int)(idx % cpr);
        const unsigned row = idx / cpr;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q * V < K) {
            const int tap = q / cpt, c0 = (q - tap * cpt) * V, dw = tap % kw, dh = tap / kw;
            const int ho = (int)(row % Ho), wo = (int)((row / Ho) % Wo), b = (int)(row / (unsigned)(Ho * Wo));
            const int h = ho * sh + dh - ph, w = wo * sw + dw - pw;
            if (h >= 0 && h < g.H && w >= 0 && w < g.W) v = *(const uint4*)(in + grid_off(g, b, w, h) + c0);
        }
        *(uint4*)(col + (long long)row * Kp + q * V) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void col2im2d_vec_kernel(const T* __restrict__ dcol, T* __restrict__ din, Grid g, int kh, int kw,
                                                           int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp, int accumulate) {
    constexpr int V = 16 / (int)sizeof(T);
    const int cv = g.C / V;
    const unsigned total = (unsigned)((long long)g.B * g.W * g.H * cv);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c0 = (int)(idx % cv) * V;
        const int h = (int)((idx / cv) % g.H);
        const int w = (int)((idx / (unsigned)(cv * g.H)) % g.W);
        const int b = (int)(idx / (unsigned)(cv * g.H * g.W));
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int dh = 0; dh < kh; ++dh) {
            const int hn = h + ph - dh;
            if (hn < 0 || hn % sh) continue;
            const int ho = hn / sh;
            if (ho >= Ho) continue;
            for (int dw = 0; dw < kw; ++dw) {
                const int wn = w + pw - dw;
                if (wn < 0 || wn % sw) continue;
                const int wo = wn / sw;
                if (wo >= Wo) continue;
                const uint4 raw = *(const uint4*)(dcol + (((long long)b * Wo + wo) * Ho + ho) * Kp + (dh * kw + dw) * g.C + c0);
                const T* pv = (const T*)&raw;
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += to_f32(pv[e]);
            }
        }
        const long long o = grid_off(g, b, w, h) + c0;
        uint4 outv;
        T* po = (T*)&outv;
        if (accumulate) {
            const uint4 old = *(const uint4*)(din + o);
            const T* pold = (const T*)&old;
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += to_f32(pold[e]);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) po[e] = from_f32<T>(acc[e]);
        *(uint4*)(din + o) = outv;
    }
}

// Depthwise convolution (Conv2dSeparable's grouped nn.Conv2d, scalogram_model.py:532-544) on the im2col matrix: tap t of
// channel c of output row m sits at col[m][t*C + c], the weight at w[c*taps + t] (reference layout [C][1][kh][kw]).
//   fwd   : y[row m][c] = sum_t col[m][t*C + c] * w[c][t]                   y rows at row_off(m, rpi, item, C)
//   bwd_d : dcol[m][t*C + c] = dy[m][c] * w[c][t]                           (col2im then gives the input gradient)
//   bwd_w : slabs[blk][c][t] = sum over the block's rows of col[m][t*C + c] * dy[m][c]
template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const T* __restrict__ col, const float* __restrict__ w, T* __restrict__ y,
                                                     long long M, int C, int taps, int Kp, int rpi, long long item) {
    const long long total = M * C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long long m = idx / C;
        float acc = 0.f;
        for (int t = 0; t < taps; ++t) acc = fmaf(to_f32(col[m * Kp + (long long)t * C + c]), w[c * taps + t], acc);
        y[row_off((int)m, rpi, item, C) + c] = from_f32<T>(acc);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_col_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dcol,
                                                         long long M, int C, int taps, int Kp, int rpi, long long item) {
    const long long total = M * C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long long m = idx / C;
        const float g = to_f32(dy[row_off((int)m, rpi, item, C) + c]);
        for (int t = 0; t < taps; ++t) dcol[m * Kp + (long long)t * C + c] = from_f32<T>(g * w[c * taps + t]);
        if (c == 0)
            for (int k = taps * C; k < Kp; ++k) dcol[m * Kp + k] = from_f32<T>(0.f);          // K padding of the row
    }
}
template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_w_kernel(const T* __restrict__ col, const T* __restrict__ dy, float* __restrict__ slabs,
                                                       long long M, int C, int taps, int Kp, int rpi, long long item,
                                                       long long rows_per_block) {
    // thread -> (c, t) pairs; rows of this block are walked serially (coalesced along c)
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int o = threadIdx.x; o < C * taps; o += 256) {
        const int t = o / C, c = o % C;
        float acc = 0.f;
        for (long long m = r0; m < r1; ++m)
            acc = fmaf(to_f32(col[m * Kp + (long long)t * C + c]), to_f32(dy[row_off((int)m, rpi, item, C) + c]), acc);
        slabs[((long long)blockIdx.x * C + c) * taps + t] = acc;
    }
}

// Per-block partial sums over rows of x[rows][C]: slabs[blk][0][c] = sum x, slabs[blk][1][c] = sum x^2  (BatchNorm statistics;
// pad rows of a grid are zero and drop out).
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ slabs, long long rows, int C,
                                                       long long rows_per_block) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    const int c4n = C / 4;
    // thread t owns column group t % c4n and row phase t / c4n; threads beyond nrp * c4n idle (host guarantees C <= 1024)
    const int cg = threadIdx.x % c4n, rp = threadIdx.x / c4n, nrp = 256 / c4n;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (rp < nrp) {
        // four independent row loads per trip (a single 8-byte load per thread in flight leaves the kernel at half the HBM rate)
        long long r = r0 + rp;
        const T* px = x + cg * 4;
        for (; r + 3LL * nrp < r1; r += 4LL * nrp) {
            const f32x4 v0 = load4(px + r * C), v1 = load4(px + (r + nrp) * C), v2 = load4(px + (r + 2LL * nrp) * C),
                        v3 = load4(px + (r + 3LL * nrp) * C);
            s1 += v0; s2 += v0 * v0;
            s1 += v1; s2 += v1 * v1;
            s1 += v2; s2 += v2 * v2;
            s1 += v3; s2 += v3 * v3;
        }
        for (; r < r1; r += nrp) {
            const f32x4 v = load4(px + r * C);
            s1 += v;
            s2 += v * v;
        }
    }
    __shared__ float red[2048];          // [nrp][2][C]: 256 / (C/4) * 2 * C = 2048 floats for every C
    if (rp < nrp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(rp * 2 + 0) * C + cg * 4 + e] = s1[e];
            red[(rp * 2 + 1) * C + cg * 4 + e] = s2[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float acc = 0.f;
        for (int q = 0; q < nrp; ++q) acc += red[q * 2 * C + i];          // fixed order: deterministic
        slabs[(long long)blockIdx.x * 2 * C + i] = acc;
    }
}

// mean / rstd from the partial sums (fixed summation order, double accumulation); optional running-statistics update with
// torch's conventions (momentum on the batch mean and the UNBIASED batch variance).   stats[0][c] = mean, stats[1][c] = rstd
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ slabs, int nslab, int C, double count, float eps,
                                                           float momentum, float* __restrict__ stats, float* __restrict__ run_mean,
                                                           float* __restrict__ run_var) {
    // 16 channels x 64 slab lanes per block: lane zl sums slabs zl, zl+64, ... (four loads in flight per trip: the kernel is a chain of
    // load latencies, 2 048 slabs on 16 lanes with one load in flight took 25 us per call, twelve calls per configs[2] step); the 64
    // partial sums of a channel are then added in lane order (a fixed order, so the result does not depend on scheduling)
    __shared__ double red[2][64][16];
    const int cl = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const float* ps = slabs + c;
        const long long st = 2LL * C;
        int z = zl;
        for (; z + 192 < nslab; z += 256) {
            const float a0 = ps[z * st], a1 = ps[(z + 64) * st], a2 = ps[(z + 128) * st], a3 = ps[(z + 192) * st];
            const float b0 = ps[z * st + C], b1 = ps[(z + 64) * st + C], b2 = ps[(z + 128) * st + C], b3 = ps[(z + 192) * st + C];
            s1 += (double)a0; s1 += (double)a1; s1 += (double)a2; s1 += (double)a3;
            s2 += (double)b0; s2 += (double)b1; s2 += (double)b2; s2 += (double)b3;
        }
        for (; z < nslab; z += 64) {
            s1 += (double)ps[z * st];
            s2 += (double)ps[z * st + C];
        }
    }
    red[0][zl][cl] = s1;
    red[1][zl][cl] = s2;
    __syncthreads();
    if (zl != 0 || c >= C) return;
    for (int r = 1; r < 64; ++r) {
        s1 += red[0][r][cl];
        s2 += red[1][r][cl];
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * mean);
        run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unbiased);
    }
}

// out(b,w,h,c) = act((x(b,w,h,c) - mean[c]) * rstd[c] * gamma[c] + beta[c]) on the valid rows of two grids of equal B, W, H, C
template <typename TX, typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TX* __restrict__ x, Grid gx, T* __restrict__ out, Grid go,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu) {
    const int c4n = gx.C / 4;
    const unsigned total = (unsigned)((long long)gx.B * gx.W * gx.H * c4n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c4 = (int)(idx % c4n);
        const int h = (int)((idx / c4n) % gx.H);
        const unsigned col = idx / (unsigned)(c4n * gx.H);
        const int w = (int)(col % gx.W), b = (int)(col / gx.W);
        const f32x4 v = load4(x + grid_off(gx, b, w, h) + c4 * 4);
        const f32x4 mu = *(const f32x4*)(stats + c4 * 4), rs = *(const f32x4*)(stats + gx.C + c4 * 4);
        const f32x4 ga = *(const f32x4*)(gamma + c4 * 4), be = *(const f32x4*)(beta + c4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (v[e] - mu[e]) * rs[e] * ga[e] + be[e];
            if (relu) o[e] = relu_f(o[e]);
        }
        store4(out + grid_off(go, b, w, h) + c4 * 4, o);
    }
}

// Backward reductions: g = dy * (y > 0 if relu);  slabs[blk][0][c] = sum g * xhat,  slabs[blk][1][c] = sum g
template <typename TX, typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y, Grid gy,
                                                            const TX* __restrict__ x, Grid gx, const float* __restrict__ stats,
                                                            float* __restrict__ slabs, int relu, long long cols_per_block) {
    const int C = gx.C, c4n = C / 4;
    const int cg = threadIdx.x % c4n, rp = threadIdx.x / c4n, nrp = 256 / c4n;
    const long long ncol = (long long)gx.B * gx.W;
    const long long q0 = (long long)blockIdx.x * cols_per_block, q1 = min(ncol, q0 + cols_per_block);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (rp < nrp) {
        const f32x4 mu = *(const f32x4*)(stats + cg * 4), rs = *(const f32x4*)(stats + C + cg * 4);
        // rows in batches of four: all twelve loads of a batch are issued before the first is used
        const unsigned rend = (unsigned)(q1 * gx.H);
        for (unsigned r = (unsigned)(q0 * gx.H) + rp; r < rend; r += 4u * nrp) {
            f32x4 g4[4], y4[4], x4[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned ru = r + (unsigned)(u * nrp);
                ok[u] = ru < rend;
                const unsigned rc = ok[u] ? ru : r;
                const unsigned q = rc / (unsigned)gx.H;
                const int h = (int)(rc - q * gx.H), w = (int)(q % gx.W), b = (int)(q / gx.W);
                const long long oy = grid_off(gy, b, w, h) + cg * 4;
                g4[u] = load4(dy + oy);
                y4[u] = relu ? load4(y + oy) : (f32x4){1.f, 1.f, 1.f, 1.f};
                x4[u] = load4(x + grid_off(gx, b, w, h) + cg * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = y4[u][e] > 0.f ? g4[u][e] : 0.f;
                    s1[e] += g * (x4[u][e] - mu[e]) * rs[e];
                    s2[e] += g;
                }
            }
        }
    }
    __shared__ float red[2048];          // [nrp][2][C]: 256 / (C/4) * 2 * C = 2048 floats for every C
    if (rp < nrp) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(rp * 2 + 0) * C + cg * 4 + e] = s1[e];
            red[(rp * 2 + 1) * C + cg * 4 + e] = s2[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float acc = 0.f;
        for (int q = 0; q < nrp; ++q) acc += red[q * 2 * C + i];          // fixed order: deterministic
        slabs[(long long)blockIdx.x * 2 * C + i] = acc;
    }
}

// dx = gamma * rstd * (g - dbeta / n - xhat * dgamma / n)   (train: batch statistics)   or   g * gamma * rstd   (eval)
template <typename TX, typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, Grid gy,
                                                           const TX* __restrict__ x, TX* __restrict__ dx, Grid gx,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           float inv_count, int relu, int train) {
    const int C = gx.C, c4n = C / 4;
    const unsigned total = (unsigned)((long long)gx.B * gx.W * gx.H * c4n);
    // When the grid stride is a multiple of C/4 a thread keeps the same four channels for its whole walk: their coefficients
    // are formed once (dx = k1 * g - k2 - k3 * (x - mean)) instead of sixteen scalar loads per element.
    const bool fixed = ((gridDim.x * 256u) % (unsigned)c4n) == 0u;
    f32x4 k1 = {0.f, 0.f, 0.f, 0.f}, k2 = k1, k3 = k1, mu = k1;
    auto coeffs = [&](int c4) {          // (16-byte loads, as in bn_bwd_apply8_kernel)
        const int c = c4 * 4;
        const f32x4 rs4 = *(const f32x4*)(stats + C + c), ga4 = *(const f32x4*)(gamma + c);
        f32x4 db4 = {0.f, 0.f, 0.f, 0.f}, dg4 = db4;
        if (train) { db4 = *(const f32x4*)(dbeta + c); dg4 = *(const f32x4*)(dgamma + c); }
        mu = *(const f32x4*)(stats + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float rs = rs4[e], ga = ga4[e];
            k1[e] = ga * rs;
            k2[e] = train ? ga * rs * db4[e] * inv_count : 0.f;
            k3[e] = train ? ga * rs * rs * dg4[e] * inv_count : 0.f;
        }
    };
    if (fixed) coeffs((int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)c4n));
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c4 = (int)(idx % c4n);
        const int h = (int)((idx / c4n) % gx.H);
        const unsigned col = idx / (unsigned)(c4n * gx.H);
        const int w = (int)(col % gx.W), b = (int)(col / gx.W);
        const long long oy = grid_off(gy, b, w, h) + c4 * 4, ox = grid_off(gx, b, w, h) + c4 * 4;
        f32x4 g = load4(dy + oy);
        const f32x4 xv = load4(x + ox);
        if (relu) {
            const f32x4 yy = load4(y + oy);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
        }
        if (!fixed) coeffs(c4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = k1[e] * g[e] - k2[e] - k3[e] * (xv[e] - mu[e]);
        store4(dx + ox, o);
    }
}


// ---- 16-byte variants of the four BatchNorm kernels for bf16 tensors (8 channels per thread: one dwordx4 load per tensor and row
// instead of a dwordx2; C a multiple of 8).  Same sums in the same per-thread order over rows; the reduction over row phases is
// in fixed order as above.
struct f32x8 { f32x4 lo, hi; };
__device__ __forceinline__ f32x8 load8(const bf16_t* src) {
    const uint4 u = *(const uint4*)src;
    f32x8 o;
    o.lo = (f32x4){__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                   __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
    o.hi = (f32x4){__builtin_bit_cast(float, u.z << 16), __builtin_bit_cast(float, u.z & 0xffff0000u),
                   __builtin_bit_cast(float, u.w << 16), __builtin_bit_cast(float, u.w & 0xffff0000u)};
    return o;
}
__device__ __forceinline__ void store8(bf16_t* dst, const f32x4& lo, const f32x4& hi) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)lo[e]; o[4 + e] = (bf16_t)hi[e]; }
    *(bf16x8*)dst = o;
}

__global__ __launch_bounds__(256) void bn_stats8_kernel(const bf16_t* __restrict__ x, float* __restrict__ slabs, long long rows, int C,
                                                        long long rows_per_block) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    const int c8n = C / 8;
    const int cg = threadIdx.x % c8n, rp = threadIdx.x / c8n, nrp = 256 / c8n;
    f32x4 s1l = {0.f, 0.f, 0.f, 0.f}, s1h = s1l, s2l = s1l, s2h = s1l;
    if (rp < nrp) {
        long long r = r0 + rp;
        const bf16_t* px = x + cg * 8;
        for (; r + 3LL * nrp < r1; r += 4LL * nrp) {
            const f32x8 v0 = load8(px + r * C), v1 = load8(px + (r + nrp) * C), v2 = load8(px + (r + 2LL * nrp) * C),
                        v3 = load8(px + (r + 3LL * nrp) * C);
            s1l += v0.lo; s1h += v0.hi; s2l += v0.lo * v0.lo; s2h += v0.hi * v0.hi;
            s1l += v1.lo; s1h += v1.hi; s2l += v1.lo * v1.lo; s2h += v1.hi * v1.hi;
            s1l += v2.lo; s1h += v2.hi; s2l += v2.lo * v2.lo; s2h += v2.hi * v2.hi;
            s1l += v3.lo; s1h += v3.hi; s2l += v3.lo * v3.lo; s2h += v3.hi * v3.hi;
        }
        for (; r < r1; r += nrp) {
            const f32x8 v = load8(px + r * C);
            s1l += v.lo; s1h += v.hi; s2l += v.lo * v.lo; s2h += v.hi * v.hi;
        }
    }
    __shared__ float red[4096];          // [nrp][2][C]: 256 / (C/8) * 2 * C = 4096 floats for every C
    if (rp < nrp) {
        *(f32x4*)(red + (rp * 2 + 0) * C + cg * 8) = s1l; *(f32x4*)(red + (rp * 2 + 0) * C + cg * 8 + 4) = s1h;
        *(f32x4*)(red + (rp * 2 + 1) * C + cg * 8) = s2l; *(f32x4*)(red + (rp * 2 + 1) * C + cg * 8 + 4) = s2h;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float acc = 0.f;
        for (int q = 0; q < nrp; ++q) acc += red[q * 2 * C + i];
        slabs[(long long)blockIdx.x * 2 * C + i] = acc;
    }
}

__global__ __launch_bounds__(256) void bn_apply8_kernel(const bf16_t* __restrict__ x, Grid gx, bf16_t* __restrict__ out, Grid go,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int relu, unsigned char* __restrict__ bits) {
    const int c8n = gx.C / 8;
    const unsigned total = (unsigned)((long long)gx.B * gx.W * gx.H * c8n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c8 = (int)(idx % c8n);
        const int h = (int)((idx / c8n) % gx.H);
        const unsigned col = idx / (unsigned)(c8n * gx.H);
        const int w = (int)(col % gx.W), b = (int)(col / gx.W);
        const f32x8 v = load8(x + grid_off(gx, b, w, h) + c8 * 8);
        f32x4 o[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int c = c8 * 8 + hf * 4;
            const f32x4 mu = *(const f32x4*)(stats + c), rs = *(const f32x4*)(stats + gx.C + c);
            const f32x4 ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
            const f32x4 vv = hf ? v.hi : v.lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[hf][e] = (vv[e] - mu[e]) * rs[e] * ga[e] + be[e];
                if (relu) o[hf][e] = relu_f(o[hf][e]);
            }
        }
        const long long oo = grid_off(go, b, w, h) + c8 * 8;
        store8(out + oo, o[0], o[1]);
        if (bits) {          // sign-bit mask of the stored activation (bit e = element e of the 8 is > 0): what the backward passes need of it
            unsigned m = 0u;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 4; ++e) m |= ((float)(bf16_t)o[hf][e] > 0.f ? 1u : 0u) << (hf * 4 + e);
            bits[oo >> 3] = (unsigned char)m;
        }
    }
}

// 16-byte (8-channel) forms of the cropped residual add and its backward for bf16 grids (the 8-byte forms ran at half the HBM rate the
// BatchNorm passes reach on the same tensors); same values.
__global__ __launch_bounds__(256) void residual_add8_kernel(const bf16_t* __restrict__ a, Grid ga, const bf16_t* __restrict__ r, Grid gr,
                                                            bf16_t* __restrict__ out, Grid go, int oh, int ow, int relu) {
    const int c8n = go.C / 8;
    const unsigned total = (unsigned)((long long)go.B * go.W * go.H * c8n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c8 = (int)(idx % c8n);
        const int h = (int)((idx / c8n) % go.H);
        const unsigned col = idx / (unsigned)(c8n * go.H);
        const int w = (int)(col % go.W), b = (int)(col / go.W);
        const f32x8 x = load8(a + grid_off(ga, b, w, h) + c8 * 8), y = load8(r + grid_off(gr, b, w + ow, h + oh) + c8 * 8);
        f32x4 lo = x.lo + y.lo, hi = x.hi + y.hi;
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = relu_f(lo[e]); hi[e] = relu_f(hi[e]); }
        }
        store8(out + grid_off(go, b, w, h) + c8 * 8, lo, hi);
    }
}
__global__ __launch_bounds__(256) void residual_add_bwd8_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out, Grid go,
                                                                bf16_t* __restrict__ da, Grid ga, bf16_t* __restrict__ dr, Grid gr, int oh, int ow,
                                                                int relu) {
    const int c8n = go.C / 8;
    const unsigned total = (unsigned)((long long)go.B * go.W * go.H * c8n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c8 = (int)(idx % c8n);
        const int h = (int)((idx / c8n) % go.H);
        const unsigned col = idx / (unsigned)(c8n * go.H);
        const int w = (int)(col % go.W), b = (int)(col / go.W);
        const long long oo = grid_off(go, b, w, h) + c8 * 8;
        uint4 g = *(const uint4*)(dout + oo);
        if (relu) {          // bf16 > 0  <=>  sign bit clear and not zero: the gradient's 16-bit halves are kept or cleared, no conversion
            const uint4 y = *(const uint4*)(out + oo);
            const unsigned yw[4] = {y.x, y.y, y.z, y.w};
            unsigned gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned l16 = yw[e] & 0xffffu, h16 = yw[e] >> 16;
                gw[e] &= ((l16 != 0u && l16 < 0x8000u) ? 0xffffu : 0u) | ((h16 != 0u && h16 < 0x8000u) ? 0xffff0000u : 0u);
            }
            g = make_uint4(gw[0], gw[1], gw[2], gw[3]);
        }
        *(uint4*)(da + grid_off(ga, b, w, h) + c8 * 8) = g;
        *(uint4*)(dr + grid_off(gr, b, w + ow, h + oh) + c8 * 8) = g;
    }
}

// out = act_out(act_in(BatchNorm(x)) + res(w + ow, h + oh)): the second BatchNorm + ReLU of a ScalogramEncoderBlock, the cropped residual add and
// the ReLU between blocks (scalogram_model.py:418-428, :447-472, :525-526) in one pass.  The normalised main branch is never stored: its
// backward pass needs the BatchNorm's input, its statistics and the SIGN BITS of its output (bits, one byte per 8 elements at the element
// offsets of the x grid, as cpc_bn_apply_bits writes them).  The normalised value is rounded to bf16 before the add, as the two-pass route
// stores it: same results bit for bit.  bf16, 8 channels per thread; TR: the residual's type.
__device__ __forceinline__ f32x8 load8(const float* src) {
    f32x8 o;
    o.lo = *(const f32x4*)src;
    o.hi = *(const f32x4*)(src + 4);
    return o;
}
template <typename TR>
__global__ __launch_bounds__(256) void bn_apply_residual8_kernel(const bf16_t* __restrict__ x, Grid gx, const TR* __restrict__ r, Grid gr,
                                                                 bf16_t* __restrict__ out, Grid go, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, int oh, int ow,
                                                                 int relu_in, int relu_out, unsigned char* __restrict__ bits, Grid ga,
                                                                 unsigned char* __restrict__ obits) {
    const int c8n = gx.C / 8;
    const unsigned total = (unsigned)((long long)gx.B * gx.W * gx.H * c8n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c8 = (int)(idx % c8n);
        const int h = (int)((idx / c8n) % gx.H);
        const unsigned col = idx / (unsigned)(c8n * gx.H);
        const int w = (int)(col % gx.W), b = (int)(col / gx.W);
        const f32x8 v = load8(x + grid_off(gx, b, w, h) + c8 * 8);
        const f32x8 rv = load8(r + grid_off(gr, b, w + ow, h + oh) + c8 * 8);
        f32x4 o[2];
        unsigned m = 0u;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int c = c8 * 8 + hf * 4;
            const f32x4 mu = *(const f32x4*)(stats + c), rs = *(const f32x4*)(stats + gx.C + c);
            const f32x4 ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
            const f32x4 vv = hf ? v.hi : v.lo, rr = hf ? rv.hi : rv.lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = (vv[e] - mu[e]) * rs[e] * ga[e] + be[e];
                if (relu_in) a = relu_f(a);
                a = (float)(bf16_t)a;
                m |= (a > 0.f ? 1u : 0u) << (hf * 4 + e);
                o[hf][e] = a + rr[e];
                if (relu_out) o[hf][e] = relu_f(o[hf][e]);
            }
        }
        const long long oo = grid_off(go, b, w, h) + c8 * 8;
        store8(out + oo, o[0], o[1]);
        if (bits) bits[(grid_off(ga, b, w, h) + c8 * 8) >> 3] = (unsigned char)m;      // (addressed like the activation grid it stands for)
        if (obits) {          // sign bits of the block output (the ReLU between blocks), addressed like out: what the fused backward passes read
            unsigned mo = 0u;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 4; ++e) mo |= ((float)(bf16_t)o[hf][e] > 0.f ? 1u : 0u) << (hf * 4 + e);
            obits[oo >> 3] = (unsigned char)mo;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, Grid gy,
                                                             const bf16_t* __restrict__ x, Grid gx, const float* __restrict__ stats,
                                                             float* __restrict__ slabs, int relu, long long cols_per_block,
                                                             const unsigned char* __restrict__ ybits, Grid gd,
                                                             const unsigned char* __restrict__ obits) {
    // gd: the grid dy lives on (the activation grid gy, or — the block's residual add folded in — the block-output grid, whose ReLU mask
    // comes as sign bits obits addressed like dy: g = dy [out > 0] [y > 0])
    const int C = gx.C, c8n = C / 8;
    const int cg = threadIdx.x % c8n, rp = threadIdx.x / c8n, nrp = 256 / c8n;
    const long long ncol = (long long)gx.B * gx.W;
    const long long q0 = (long long)blockIdx.x * cols_per_block, q1 = min(ncol, q0 + cols_per_block);
    f32x4 s1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, s2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (rp < nrp && q0 < q1) {
        f32x4 mu[2], rs[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            mu[hf] = *(const f32x4*)(stats + cg * 8 + hf * 4);
            rs[hf] = *(const f32x4*)(stats + C + cg * 8 + hf * 4);
        }
        const unsigned rend = (unsigned)(q1 * gx.H);
        for (unsigned r = (unsigned)(q0 * gx.H) + rp; r < rend; r += 2u * nrp) {
            f32x8 g8[2], y8[2], x8[2];
            unsigned mb[2] = {0xffu, 0xffu}, mo[2] = {0xffu, 0xffu};
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned ru = r + (unsigned)(u * nrp);
                ok[u] = ru < rend;
                const unsigned rc = ok[u] ? ru : r;
                const unsigned q = rc / (unsigned)gx.H;
                const int h = (int)(rc - q * gx.H), w = (int)(q % gx.W), b = (int)(q / gx.W);
                const long long oy = grid_off(gy, b, w, h) + cg * 8, od = grid_off(gd, b, w, h) + cg * 8;
                g8[u] = load8(dy + od);
                if (obits) mo[u] = obits[od >> 3];
                if (relu) {
                    if (ybits) mb[u] = ybits[oy >> 3];
                    else y8[u] = load8(y + oy);
                }
                x8[u] = load8(x + grid_off(gx, b, w, h) + cg * 8);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const f32x4 gg = hf ? g8[u].hi : g8[u].lo, yy = hf ? y8[u].hi : y8[u].lo, xx = hf ? x8[u].hi : x8[u].lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool pos = ybits ? ((mb[u] >> (hf * 4 + e)) & 1u) != 0u : yy[e] > 0.f;
                        const float g = ((!relu || pos) && ((mo[u] >> (hf * 4 + e)) & 1u)) ? gg[e] : 0.f;
                        s1[hf][e] += g * (xx[e] - mu[hf][e]) * rs[hf][e];
                        s2[hf][e] += g;
                    }
                }
            }
        }
    }
    __shared__ float red[4096];
    if (rp < nrp) {
        *(f32x4*)(red + (rp * 2 + 0) * C + cg * 8) = s1[0]; *(f32x4*)(red + (rp * 2 + 0) * C + cg * 8 + 4) = s1[1];
        *(f32x4*)(red + (rp * 2 + 1) * C + cg * 8) = s2[0]; *(f32x4*)(red + (rp * 2 + 1) * C + cg * 8 + 4) = s2[1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float acc = 0.f;
        for (int q = 0; q < nrp; ++q) acc += red[q * 2 * C + i];
        slabs[(long long)blockIdx.x * 2 * C + i] = acc;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, Grid gy,
                                                            const bf16_t* __restrict__ x, bf16_t* __restrict__ dx, Grid gx,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                            float inv_count, int relu, int train, const unsigned char* __restrict__ ybits, Grid gd,
                                                            const unsigned char* __restrict__ obits, bf16_t* __restrict__ dres, Grid gr, int oh,
                                                            int ow) {
    // gd / obits as in bn_bwd_reduce8_kernel; dres (may be null): the gradient of the cropped residual operand, dy [out > 0] at (w + ow, h + oh)
    const int C = gx.C, c8n = C / 8;
    const unsigned total = (unsigned)((long long)gx.B * gx.W * gx.H * c8n);
    const bool fixed = ((gridDim.x * 256u) % (unsigned)c8n) == 0u;
    f32x4 k1[2], k2[2], k3[2], mu[2];
    // (16-byte loads of a thread's eight channels: it was forty scalar loads per thread in front of its first position, and a thread of
    // a small grid has one or two positions: 32 us for the 48 MB of a ConvolutionalArModel block at B = 256)
    auto coeffs = [&](int c8) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int c = c8 * 8 + hf * 4;
            const f32x4 rs4 = *(const f32x4*)(stats + C + c), ga4 = *(const f32x4*)(gamma + c);
            f32x4 db4 = {0.f, 0.f, 0.f, 0.f}, dg4 = db4;
            if (train) { db4 = *(const f32x4*)(dbeta + c); dg4 = *(const f32x4*)(dgamma + c); }
            mu[hf] = *(const f32x4*)(stats + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float rs = rs4[e], ga = ga4[e];
                k1[hf][e] = ga * rs;
                k2[hf][e] = train ? ga * rs * db4[e] * inv_count : 0.f;
                k3[hf][e] = train ? ga * rs * rs * dg4[e] * inv_count : 0.f;
            }
        }
    };
    if (fixed) coeffs((int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)c8n));
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c8 = (int)(idx % c8n);
        const int h = (int)((idx / c8n) % gx.H);
        const unsigned col = idx / (unsigned)(c8n * gx.H);
        const int w = (int)(col % gx.W), b = (int)(col / gx.W);
        const long long oy = grid_off(gy, b, w, h) + c8 * 8, ox = grid_off(gx, b, w, h) + c8 * 8, od = grid_off(gd, b, w, h) + c8 * 8;
        uint4 graw = *(const uint4*)(dy + od);
        unsigned mo = 0xffu;
        if (obits) {
            mo = obits[od >> 3];
            unsigned gw[4] = {graw.x, graw.y, graw.z, graw.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) gw[e] &= ((mo >> (2 * e)) & 1u ? 0xffffu : 0u) | ((mo >> (2 * e + 1)) & 1u ? 0xffff0000u : 0u);
            graw = make_uint4(gw[0], gw[1], gw[2], gw[3]);
        }
        if (dres) *(uint4*)(dres + grid_off(gr, b, w + ow, h + oh) + c8 * 8) = graw;
        f32x8 g8;
        g8.lo = (f32x4){__builtin_bit_cast(float, graw.x << 16), __builtin_bit_cast(float, graw.x & 0xffff0000u),
                        __builtin_bit_cast(float, graw.y << 16), __builtin_bit_cast(float, graw.y & 0xffff0000u)};
        g8.hi = (f32x4){__builtin_bit_cast(float, graw.z << 16), __builtin_bit_cast(float, graw.z & 0xffff0000u),
                        __builtin_bit_cast(float, graw.w << 16), __builtin_bit_cast(float, graw.w & 0xffff0000u)};
        const f32x8 x8 = load8(x + ox);
        f32x8 y8;
        unsigned mb = 0xffu;
        if (relu) {
            if (ybits) mb = ybits[oy >> 3];
            else y8 = load8(y + oy);
        }
        if (!fixed) coeffs(c8);
        f32x4 o[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x4 gg = hf ? g8.hi : g8.lo, yy = hf ? y8.hi : y8.lo, xx = hf ? x8.hi : x8.lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool pos = ybits ? ((mb >> (hf * 4 + e)) & 1u) != 0u : yy[e] > 0.f;
                const float g = (!relu || pos) ? gg[e] : 0.f;
                o[hf][e] = k1[hf][e] * g - k2[hf][e] - k3[hf][e] * (xx[e] - mu[hf][e]);
            }
        }
        store8(dx + ox, o[0], o[1]);
    }
}

// MaxPool2d(kernel = stride = p, ceil_mode=True, no padding) on grids (residual branches, scalogram_model.py:434-436).
template <typename TI, typename T>
__global__ __launch_bounds__(256) void maxpool2d_fwd_kernel(const TI* __restrict__ in, Grid gi, T* __restrict__ out, Grid go, int p) {
    const unsigned total = (unsigned)((long long)go.B * go.W * go.H * go.C);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c = (int)(idx % go.C);
        const int ho = (int)((idx / go.C) % go.H);
        const int wo = (int)((idx / (unsigned)(go.C * go.H)) % go.W);
        const int b = (int)(idx / (unsigned)(go.C * go.H * go.W));
        float m = -INFINITY;
        for (int dh = 0; dh < p; ++dh)
            for (int dw = 0; dw < p; ++dw) {
                const int h = ho * p + dh, w = wo * p + dw;
                if (h < gi.H && w < gi.W) m = fmaxf(m, to_f32(in[grid_off(gi, b, w, h) + c]));
            }
        out[grid_off(go, b, wo, ho) + c] = from_f32<T>(m);
    }
}

// bf16, C % 8 == 0: eight channels (16 bytes) per thread — the scalar kernels move 2 bytes per lane
__global__ __launch_bounds__(256) void maxpool2d_fwd_vec8_kernel(const bf16_t* __restrict__ in, Grid gi, bf16_t* __restrict__ out, Grid go,
                                                                 int p) {
    const int c8n = go.C / 8;
    const unsigned total = (unsigned)((long long)go.B * go.W * go.H * c8n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c0 = (int)(idx % c8n) * 8;
        const int ho = (int)((idx / c8n) % go.H);
        const int wo = (int)((idx / (unsigned)(c8n * go.H)) % go.W);
        const int b = (int)(idx / (unsigned)(c8n * go.H * go.W));
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        for (int dh = 0; dh < p; ++dh)
            for (int dw = 0; dw < p; ++dw) {
                const int h = ho * p + dh, w = wo * p + dw;
                if (h < gi.H && w < gi.W) {
                    const bf16x8 v = *(const bf16x8*)(in + grid_off(gi, b, w, h) + c0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
                }
            }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)m[e];
        *(bf16x8*)(out + grid_off(go, b, wo, ho) + c0) = o;
    }
}

__global__ __launch_bounds__(256) void maxpool2d_bwd_vec8_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ din, Grid gi,
                                                                 const bf16_t* __restrict__ dout, Grid go, int p, int accumulate) {
    const int c8n = go.C / 8;
    const unsigned total = (unsigned)((long long)go.B * go.W * go.H * c8n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c0 = (int)(idx % c8n) * 8;
        const int ho = (int)((idx / c8n) % go.H);
        const int wo = (int)((idx / (unsigned)(c8n * go.H)) % go.W);
        const int b = (int)(idx / (unsigned)(c8n * go.H * go.W));
        float m[8];
        int best[8];                      // window position dh * p + dw of the first maximum, per channel
#pragma unroll
        for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; best[e] = -1; }
        for (int dh = 0; dh < p; ++dh)
            for (int dw = 0; dw < p; ++dw) {
                const int h = ho * p + dh, w = wo * p + dw;
                if (h < gi.H && w < gi.W) {
                    const bf16x8 v = *(const bf16x8*)(in + grid_off(gi, b, w, h) + c0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e];
                        if (f > m[e]) { m[e] = f; best[e] = dh * p + dw; }
                    }
                }
            }
        const bf16x8 g = *(const bf16x8*)(dout + grid_off(go, b, wo, ho) + c0);
        for (int dh = 0; dh < p; ++dh)
            for (int dw = 0; dw < p; ++dw) {
                const int h = ho * p + dh, w = wo * p + dw;
                if (h < gi.H && w < gi.W) {
                    const long long o = grid_off(gi, b, w, h) + c0;
                    bf16x8 old;
                    if (accumulate) old = *(const bf16x8*)(din + o);
                    bf16x8 r;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float add = best[e] == dh * p + dw ? (float)g[e] : 0.f;
                        r[e] = (bf16_t)(accumulate ? (float)old[e] + add : add);
                    }
                    *(bf16x8*)(din + o) = r;
                }
            }
    }
}

// din(window) += dout at the first position holding the window maximum (torch's tie rule: first in (h, w) scan order)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2d_bwd_kernel(const T* __restrict__ in, T* __restrict__ din, Grid gi,
                                                            const T* __restrict__ dout, Grid go, int p, int accumulate) {
    const unsigned total = (unsigned)((long long)go.B * go.W * go.H * go.C);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c = (int)(idx % go.C);
        const int ho = (int)((idx / go.C) % go.H);
        const int wo = (int)((idx / (unsigned)(go.C * go.H)) % go.W);
        const int b = (int)(idx / (unsigned)(go.C * go.H * go.W));
        float m = -INFINITY;
        int bh = -1, bw = -1;
        for (int dh = 0; dh < p; ++dh)
            for (int dw = 0; dw < p; ++dw) {
                const int h = ho * p + dh, w = wo * p + dw;
                if (h < gi.H && w < gi.W) {
                    const float v = to_f32(in[grid_off(gi, b, w, h) + c]);
                    if (v > m) { m = v; bh = h; bw = w; }
                }
            }
        const float g = to_f32(dout[grid_off(go, b, wo, ho) + c]);
        for (int dh = 0; dh < p; ++dh)
            for (int dw = 0; dw < p; ++dw) {
                const int h = ho * p + dh, w = wo * p + dw;
                if (h < gi.H && w < gi.W) {
                    const long long o = grid_off(gi, b, w, h) + c;
                    const float add = (h == bh && w == bw) ? g : 0.f;
                    din[o] = from_f32<T>(accumulate ? to_f32(din[o]) + add : add);
                }
            }
    }
}

// out = act(main + res(w + ow, h + oh))   (ScalogramEncoderBlock.forward's cropped residual add, scalogram_model.py:453-472,
// and the F.relu between blocks, :525-526)
template <typename TR, typename T>
__global__ __launch_bounds__(256) void residual_add_kernel(const T* __restrict__ a, Grid ga, const TR* __restrict__ r, Grid gr,
                                                           T* __restrict__ out, Grid go, int oh, int ow, int relu) {
    const int c4n = go.C / 4;
    const unsigned total = (unsigned)((long long)go.B * go.W * go.H * c4n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c4 = (int)(idx % c4n);
        const int h = (int)((idx / c4n) % go.H);
        const unsigned col = idx / (unsigned)(c4n * go.H);
        const int w = (int)(col % go.W), b = (int)(col / go.W);
        f32x4 v = load4(a + grid_off(ga, b, w, h) + c4 * 4) + load4(r + grid_off(gr, b, w + ow, h + oh) + c4 * 4);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
        }
        store4(out + grid_off(go, b, w, h) + c4 * 4, v);
    }
}

// g = dout * (out > 0 if relu);  da = g;  dr(w + ow, h + oh) = g  (dr must be zero elsewhere: the caller clears it)
template <typename TR, typename T>
__global__ __launch_bounds__(256) void residual_add_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ out, Grid go,
                                                               T* __restrict__ da, Grid ga, TR* __restrict__ dr, Grid gr, int oh,
                                                               int ow, int relu) {
    const int c4n = go.C / 4;
    const unsigned total = (unsigned)((long long)go.B * go.W * go.H * c4n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c4 = (int)(idx % c4n);
        const int h = (int)((idx / c4n) % go.H);
        const unsigned col = idx / (unsigned)(c4n * go.H);
        const int w = (int)(col % go.W), b = (int)(col / go.W);
        const long long oo = grid_off(go, b, w, h) + c4 * 4;
        f32x4 g = load4(dout + oo);
        if (relu) {
            const f32x4 y = load4(out + oo);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
        }
        store4(da + grid_off(ga, b, w, h) + c4 * 4, g);
        store4(dr + grid_off(gr, b, w + ow, h + oh) + c4 * 4, g);
    }
}

// dst[3 i .. 3 i + 2] = (hi, lo, hi) with hi = bf16(x_i), lo = bf16(x_i - hi): against weights laid out (wh, wh, wl) a bf16
// MFMA GEMM then accumulates xh wh + xl wh + xh wl = x w up to the dropped lo*lo term (~2^-16 relative) in f32.
__global__ __launch_bounds__(256) void split3_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = src[i];
        const bf16_t hi = (bf16_t)x;
        const bf16_t lo = (bf16_t)(x - (float)hi);
        dst[3 * i] = hi;
        dst[3 * i + 1] = lo;
        dst[3 * i + 2] = hi;
    }
}

// g[i] = y[i] > 0 ? g[i] : 0 over flat buffers (ReLU backward where no fused epilogue applies)
template <typename T>
__global__ __launch_bounds__(256) void relu_mask_kernel(T* __restrict__ g, const T* __restrict__ y, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 gv = load4(g + i * 4);
        const f32x4 yv = load4(y + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = yv[e] > 0.f ? gv[e] : 0.f;
        store4(g + i * 4, gv);
    }
}

// a[i] += b[i] over flat buffers (f32 sum, one rounding to the storage type: what `a += b` gives the reference's autograd when two branches
// meet): the data gradient of a residual projection joining the main branch's where the GEMM cannot write into the same grid
template <typename T>
__global__ __launch_bounds__(256) void accumulate_kernel(T* __restrict__ a, const T* __restrict__ b, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        store4(a + i * 4, load4(a + i * 4) + load4(b + i * 4));
}

// ---- Wasserstein gradient penalty (contrastive_estimation_training.py:144-155) ----
// out = the element of `sel` at the position of the FIRST maximum of `in` in each pooling window: a max pooling applied to a
// tangent vector, selecting where the primal pooling selected (in and sel share one grid geometry).
template <typename TI, typename T>
__global__ __launch_bounds__(256) void maxpool2d_select_kernel(const TI* __restrict__ in, const TI* __restrict__ sel, Grid gi,
                                                               T* __restrict__ out, Grid go, int p) {
    const unsigned total = (unsigned)((long long)go.B * go.W * go.H * go.C);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c = (int)(idx % go.C);
        const int ho = (int)((idx / go.C) % go.H);
        const int wo = (int)((idx / (unsigned)(go.C * go.H)) % go.W);
        const int b = (int)(idx / (unsigned)(go.C * go.H * go.W));
        float m = -INFINITY, pick = 0.f;
        for (int dh = 0; dh < p; ++dh)
            for (int dw = 0; dw < p; ++dw) {
                const int h = ho * p + dh, w = wo * p + dw;
                if (h < gi.H && w < gi.W) {
                    const long long o = grid_off(gi, b, w, h) + c;
                    const float v = to_f32(in[o]);
                    if (v > m) { m = v; pick = to_f32(sel[o]); }
                }
            }
        out[grid_off(go, b, wo, ho) + c] = from_f32<T>(pick);
    }
}

// g: gradient of the summed scores w.r.t. the scalogram, f32 channels-last [npix][C] (the reference's dim 1 = channels is the
// innermost axis here).  v = d penalty / d g = factor * 2 (|g| - 1) / |g| * g / npix per pixel; partial[block] = sum (|g| - 1)^2.
__global__ __launch_bounds__(256) void gp_direction_kernel(const float* __restrict__ g, float* __restrict__ v, long long npix, int C,
                                                           float factor, float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        float n2 = 0.f;
        for (int c = 0; c < C; ++c) { const float t = g[i * C + c]; n2 += t * t; }
        const float n = sqrtf(n2);
        acc += (n - 1.f) * (n - 1.f);
        // |g| = 0: torch's norm backward yields 0 there (sub-gradient), and so does this
        const float k = n > 0.f ? factor * 2.f * (n - 1.f) / (n * (float)npix) : 0.f;
        for (int c = 0; c < C; ++c) v[i * C + c] = k * g[i * C + c];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// Second-order terms of a train-mode BatchNorm under the penalty: out = A[c] * xhat + Bc[c] * yt + Cc[c] * delta over the valid
// positions of the grid (xhat = (x - mean[c]) * rstd[c]; coef = [A | Bc | Cc], 3 C floats), zero elsewhere untouched.
template <typename TX, typename T>
__global__ __launch_bounds__(256) void bn_gp_cross_kernel(const TX* __restrict__ x, const T* __restrict__ yt, const TX* __restrict__ delta,
                                                          TX* __restrict__ out, Grid gx, const float* __restrict__ stats,
                                                          const float* __restrict__ coef) {
    const int C = gx.C;
    const unsigned total = (unsigned)((long long)gx.B * gx.W * gx.H * C);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c = (int)(idx % C);
        const int h = (int)((idx / C) % gx.H);
        const unsigned col = idx / (unsigned)(C * gx.H);
        const int w = (int)(col % gx.W), b = (int)(col / gx.W);
        const long long o = grid_off(gx, b, w, h) + c;
        const float xh = (to_f32(x[o]) - stats[c]) * stats[C + c];
        out[o] = from_f32<TX>(coef[c] * xh + coef[C + c] * to_f32(yt[o]) + coef[2 * C + c] * to_f32(delta[o]));
    }
}

}  // namespace

int launch_scalogram_pointwise(const float* cq, const float* fixed_pd, const float* pd_scale, float* out, int B, int Tn, int bins,
                               long long ldq, int phase, float offset, float log_offset, float norm, float power, int ph, int pw,
                               hipStream_t st) {
    if (B <= 0 || bins <= 0 || Tn <= (phase ? 1 : 0) || ldq < 2 * bins || (ldq & 1)) return CPC_EINVAL;
    if (phase && (!fixed_pd || !pd_scale)) return CPC_EINVAL;
    const int W = phase ? Tn - 1 : Tn;
    if (ph < 1 || pw < 1 || W / pw < 1 || bins / ph < 1) return CPC_EINVAL;
    const long long total = (long long)B * (W / pw) * (bins / ph);
    const int blocks = (int)std::min<long long>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(scalogram_pointwise_kernel, dim3(blocks), dim3(256), 0, st, cq, fixed_pd, pd_scale, out, B, Tn, bins, ldq,
                       phase, offset, log_offset, norm, power, ph, pw);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

static bool grid_ok(const int* g) {      // {B, W, H, Ha, top, C};  the kernels index elements with 32 bits
    return g && g[0] > 0 && g[1] > 0 && g[2] > 0 && g[4] >= 0 && g[3] >= g[4] + g[2] && g[5] > 0 &&
           (long long)g[0] * g[1] * g[2] * g[5] < (1ll << 31);
}
static Grid mk(const int* g) { return Grid{g[0], g[1], g[2], g[3], g[4], g[5]}; }
static int blocks_for(long long total) { return (int)std::min<long long>(8192, (total + 255) / 256); }

#define DISPATCH2(dtype, KERNEL_BF16, KERNEL_F32) \
    do {                                            \
        if ((dtype) == CPC_DTYPE_BF16) { KERNEL_BF16; } \
        else if ((dtype) == CPC_DTYPE_F32) { KERNEL_F32; } \
        else return CPC_EINVAL;                     \
    } while (0)

int launch_im2col2d(const void* in, void* col, const int* g, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp,
                    int in_f32, int dtype, hipStream_t st) {
    if (!grid_ok(g) || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 || Ho <= 0 || Wo <= 0) return CPC_EINVAL;
    if (Kp < kh * kw * g[5] || (long long)g[0] * Wo * Ho * Kp >= (1ll << 31)) return CPC_EINVAL;
    if ((Ho - 1) * sh + kh - ph > g[2] + ph || (Wo - 1) * sw + kw - pw > g[1] + pw) return CPC_EINVAL;   // windows inside the padded input
    const Grid gg = mk(g);
    const int vch = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if ((!in_f32 || dtype == CPC_DTYPE_F32) && g[5] % vch == 0 && Kp % vch == 0 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)col % 16 == 0)) {
        const int nbv = blocks_for((long long)g[0] * Wo * Ho * (Kp / vch));
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((im2col2d_vec_kernel<bf16_t>), dim3(nbv), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)col, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp),
                  hipLaunchKernelGGL((im2col2d_vec_kernel<float>), dim3(nbv), dim3(256), 0, st, (const float*)in, (float*)col, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp));
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    const int nb = blocks_for((long long)g[0] * Wo * Ho * Kp);
    if (in_f32) {
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((im2col2d_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, st, (const float*)in, (bf16_t*)col, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp),
                  hipLaunchKernelGGL((im2col2d_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)in, (float*)col, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp));
    } else {
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((im2col2d_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)col, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp),
                  hipLaunchKernelGGL((im2col2d_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)in, (float*)col, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp));
    }
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_col2im2d(const void* dcol, void* din, const int* g, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, int Kp,
                    int accumulate, int dtype, hipStream_t st) {
    if (!grid_ok(g) || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 || Ho <= 0 || Wo <= 0) return CPC_EINVAL;
    if (Kp < kh * kw * g[5]) return CPC_EINVAL;
    const Grid gg = mk(g);
    const int vch = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if (g[5] % vch == 0 && Kp % vch == 0 && ((uintptr_t)dcol % 16 == 0) && ((uintptr_t)din % 16 == 0)) {
        const int nbv = blocks_for((long long)g[0] * g[1] * g[2] * (g[5] / vch));
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((col2im2d_vec_kernel<bf16_t>), dim3(nbv), dim3(256), 0, st, (const bf16_t*)dcol, (bf16_t*)din, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp, accumulate),
                  hipLaunchKernelGGL((col2im2d_vec_kernel<float>), dim3(nbv), dim3(256), 0, st, (const float*)dcol, (float*)din, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp, accumulate));
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    const int nb = blocks_for((long long)g[0] * g[1] * g[2] * g[5]);
    DISPATCH2(dtype,
              hipLaunchKernelGGL((col2im2d_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dcol, (bf16_t*)din, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp, accumulate),
              hipLaunchKernelGGL((col2im2d_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)dcol, (float*)din, gg, kh, kw, sh, sw, ph, pw, Ho, Wo, Kp, accumulate));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

static bool bn_c_ok(int C) { return C > 0 && C % 4 == 0 && C / 4 <= 256; }    // multiples of 4 up to 1024

int launch_dw_fwd(const void* col, const float* w, void* y, long long M, int C, int taps, int Kp, int rpi, long long item, int dtype,
                  hipStream_t st) {
    if (M <= 0 || C <= 0 || taps <= 0 || Kp < taps * C || M >= (1ll << 31)) return CPC_EINVAL;
    const int nb = blocks_for(M * C);
    DISPATCH2(dtype,
              hipLaunchKernelGGL((dw_fwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)col, w, (bf16_t*)y, M, C, taps, Kp, rpi, item),
              hipLaunchKernelGGL((dw_fwd_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)col, w, (float*)y, M, C, taps, Kp, rpi, item));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
int launch_dw_bwd_col(const void* dy, const float* w, void* dcol, long long M, int C, int taps, int Kp, int rpi, long long item,
                      int dtype, hipStream_t st) {
    if (M <= 0 || C <= 0 || taps <= 0 || Kp < taps * C || M >= (1ll << 31)) return CPC_EINVAL;
    const int nb = blocks_for(M * C);
    DISPATCH2(dtype,
              hipLaunchKernelGGL((dw_bwd_col_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, w, (bf16_t*)dcol, M, C, taps, Kp, rpi, item),
              hipLaunchKernelGGL((dw_bwd_col_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)dy, w, (float*)dcol, M, C, taps, Kp, rpi, item));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
int launch_dw_bwd_w(const void* col, const void* dy, float* slabs, long long M, int C, int taps, int Kp, int rpi, long long item,
                    int nblocks, int dtype, hipStream_t st) {
    if (M <= 0 || C <= 0 || taps <= 0 || Kp < taps * C || nblocks <= 0 || M >= (1ll << 31)) return CPC_EINVAL;
    const long long rpb = (M + nblocks - 1) / nblocks;
    DISPATCH2(dtype,
              hipLaunchKernelGGL((dw_bwd_w_kernel<bf16_t>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)col, (const bf16_t*)dy, slabs, M, C, taps, Kp, rpi, item, rpb),
              hipLaunchKernelGGL((dw_bwd_w_kernel<float>), dim3(nblocks), dim3(256), 0, st, (const float*)col, (const float*)dy, slabs, M, C, taps, Kp, rpi, item, rpb));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_bn_stats(const void* x, float* slabs, long long rows, int C, int nblocks, int dtype, hipStream_t st) {
    if (rows <= 0 || !bn_c_ok(C) || nblocks <= 0) return CPC_EINVAL;
    const long long rpb = (rows + nblocks - 1) / nblocks;
    if (dtype == CPC_DTYPE_BF16 && C % 8 == 0) {
        hipLaunchKernelGGL(bn_stats8_kernel, dim3(nblocks), dim3(256), 0, st, (const bf16_t*)x, slabs, rows, C, rpb);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    DISPATCH2(dtype,
              hipLaunchKernelGGL((bn_stats_kernel<bf16_t>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)x, slabs, rows, C, rpb),
              hipLaunchKernelGGL((bn_stats_kernel<float>), dim3(nblocks), dim3(256), 0, st, (const float*)x, slabs, rows, C, rpb));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_bn_finalize(const float* slabs, int nslab, int C, double count, float eps, float momentum, float* stats, float* run_mean,
                       float* run_var, hipStream_t st) {
    if (nslab <= 0 || C <= 0 || count <= 0 || (run_mean == nullptr) != (run_var == nullptr)) return CPC_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, st, slabs, nslab, C, count, eps, momentum, stats,
                       run_mean, run_var);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

static bool same_shape(const int* a, const int* b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[5] == b[5]; }

int launch_bn_apply_residual(const void* x, const int* gx, const void* res, const int* gr, void* out, const int* go, const float* stats,
                             const float* gamma, const float* beta, int oh, int ow, int relu_in, int relu_out, int r_f32, unsigned char* bits,
                             const int* ga, int dtype, hipStream_t st, unsigned char* obits) {
    if (bits && (!grid_ok(ga) || !same_shape(gx, ga))) return CPC_EINVAL;
    if (dtype != CPC_DTYPE_BF16 || !grid_ok(gx) || !grid_ok(gr) || !grid_ok(go) || !same_shape(gx, go) || gx[5] % 8 || gr[5] != gx[5] ||
        gr[0] != gx[0] || oh < 0 || ow < 0 || gr[2] < gx[2] + oh || gr[1] < gx[1] + ow || !x || !res || !out || !stats || !gamma || !beta)
        return CPC_EINVAL;
    const int nb8 = blocks_for((long long)gx[0] * gx[1] * gx[2] * (gx[5] / 8));
    if (r_f32)
        hipLaunchKernelGGL((bn_apply_residual8_kernel<float>), dim3(nb8), dim3(256), 0, st, (const bf16_t*)x, mk(gx), (const float*)res, mk(gr),
                           (bf16_t*)out, mk(go), stats, gamma, beta, oh, ow, relu_in, relu_out, bits, bits ? mk(ga) : mk(gx), obits);
    else
        hipLaunchKernelGGL((bn_apply_residual8_kernel<bf16_t>), dim3(nb8), dim3(256), 0, st, (const bf16_t*)x, mk(gx), (const bf16_t*)res, mk(gr),
                           (bf16_t*)out, mk(go), stats, gamma, beta, oh, ow, relu_in, relu_out, bits, bits ? mk(ga) : mk(gx), obits);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_bn_apply(const void* x, const int* gx, void* out, const int* go, const float* stats, const float* gamma, const float* beta,
                    int relu, int x_f32, int dtype, hipStream_t st, unsigned char* bits) {
    if (bits && !(dtype == CPC_DTYPE_BF16 && !x_f32 && gx && gx[5] % 8 == 0)) return CPC_EINVAL;
    if (!grid_ok(gx) || !grid_ok(go) || !same_shape(gx, go) || gx[5] % 4) return CPC_EINVAL;
    const int nb = blocks_for((long long)gx[0] * gx[1] * gx[2] * (gx[5] / 4));
    if (dtype == CPC_DTYPE_BF16 && !x_f32 && gx[5] % 8 == 0) {
        const int nb8 = blocks_for((long long)gx[0] * gx[1] * gx[2] * (gx[5] / 8));
        hipLaunchKernelGGL(bn_apply8_kernel, dim3(nb8), dim3(256), 0, st, (const bf16_t*)x, mk(gx), (bf16_t*)out, mk(go), stats, gamma, beta, relu, bits);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16 && x_f32)
        hipLaunchKernelGGL((bn_apply_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, st, (const float*)x, mk(gx), (bf16_t*)out, mk(go), stats, gamma, beta, relu);
    else
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, mk(gx), (bf16_t*)out, mk(go), stats, gamma, beta, relu),
                  hipLaunchKernelGGL((bn_apply_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)x, mk(gx), (float*)out, mk(go), stats, gamma, beta, relu));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// The block's residual add folded into its second BatchNorm's backward passes (bf16, sign-bit masks): dout lives on the block-output grid
// gd, obits (may be null: no ReLU behind the add) are the sign bits of the block output addressed like dout, abits those of the normalised
// branch addressed like its activation grid ga.  The apply pass also writes the residual operand's gradient dres (cropped at (ow, oh) of gr).
int launch_bn_bwd_reduce_res(const void* dout, const int* gd, const unsigned char* obits, const unsigned char* abits, const int* ga,
                             const void* x, const int* gx, const float* stats, float* slabs, int nblocks, int dtype, hipStream_t st) {
    if (dtype != CPC_DTYPE_BF16 || !abits || !grid_ok(gx) || !grid_ok(ga) || !grid_ok(gd) || !same_shape(gx, ga) || !same_shape(gx, gd) ||
        gx[5] % 8 || !bn_c_ok(gx[5]) || nblocks <= 0 || !dout || !x || !stats || !slabs)
        return CPC_EINVAL;
    const long long ncol = (long long)gx[0] * gx[1];
    const long long cpb = (ncol + nblocks - 1) / nblocks;
    hipLaunchKernelGGL(bn_bwd_reduce8_kernel, dim3(nblocks), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)nullptr, mk(ga), (const bf16_t*)x,
                       mk(gx), stats, slabs, 1, cpb, abits, mk(gd), obits);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_bn_bwd_apply_res(const void* dout, const int* gd, const unsigned char* obits, const unsigned char* abits, const int* ga, const void* x,
                            void* dx, const int* gx, const float* stats, const float* gamma, const float* dgamma, const float* dbeta,
                            double count, int train, void* dres, const int* gr, int oh, int ow, int dtype, hipStream_t st) {
    if (dtype != CPC_DTYPE_BF16 || !abits || !grid_ok(gx) || !grid_ok(ga) || !grid_ok(gd) || !same_shape(gx, ga) || !same_shape(gx, gd) ||
        gx[5] % 8 || !bn_c_ok(gx[5]) || count <= 0 || !dout || !x || !dx || !stats || !gamma || !dgamma || !dbeta)
        return CPC_EINVAL;
    if (dres && (!grid_ok(gr) || gr[0] != gx[0] || gr[5] != gx[5] || oh < 0 || ow < 0 || gr[2] < gx[2] + oh || gr[1] < gx[1] + ow)) return CPC_EINVAL;
    const int nb8 = blocks_for((long long)gx[0] * gx[1] * gx[2] * (gx[5] / 8));
    hipLaunchKernelGGL(bn_bwd_apply8_kernel, dim3(nb8), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)nullptr, mk(ga), (const bf16_t*)x,
                       (bf16_t*)dx, mk(gx), stats, gamma, dgamma, dbeta, (float)(1.0 / count), 1, train, abits, mk(gd), obits, (bf16_t*)dres,
                       dres ? mk(gr) : mk(gx), oh, ow);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_bn_bwd_reduce(const void* dy, const void* y, const int* gy, const void* x, const int* gx, const float* stats, float* slabs,
                         int relu, int nblocks, int x_f32, int dtype, hipStream_t st, const unsigned char* ybits) {
    if (ybits && !(dtype == CPC_DTYPE_BF16 && !x_f32 && gx && gx[5] % 8 == 0)) return CPC_EINVAL;
    if (!grid_ok(gx) || !grid_ok(gy) || !same_shape(gx, gy) || !bn_c_ok(gx[5]) || nblocks <= 0) return CPC_EINVAL;
    const long long ncol = (long long)gx[0] * gx[1];
    const long long cpb = (ncol + nblocks - 1) / nblocks;
    if (dtype == CPC_DTYPE_BF16 && !x_f32 && gx[5] % 8 == 0) {
        hipLaunchKernelGGL(bn_bwd_reduce8_kernel, dim3(nblocks), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, mk(gy), (const bf16_t*)x, mk(gx), stats, slabs, relu, cpb, ybits, mk(gy), nullptr);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16 && x_f32)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, bf16_t>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, mk(gy), (const float*)x, mk(gx), stats, slabs, relu, cpb);
    else
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, bf16_t>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, mk(gy), (const bf16_t*)x, mk(gx), stats, slabs, relu, cpb),
                  hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, float>), dim3(nblocks), dim3(256), 0, st, (const float*)dy, (const float*)y, mk(gy), (const float*)x, mk(gx), stats, slabs, relu, cpb));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_bn_bwd_apply(const void* dy, const void* y, const int* gy, const void* x, void* dx, const int* gx, const float* stats,
                        const float* gamma, const float* dgamma, const float* dbeta, double count, int relu, int train, int x_f32,
                        int dtype, hipStream_t st, const unsigned char* ybits) {
    if (ybits && !(dtype == CPC_DTYPE_BF16 && !x_f32 && gx && gx[5] % 8 == 0)) return CPC_EINVAL;
    if (!grid_ok(gx) || !grid_ok(gy) || !same_shape(gx, gy) || gx[5] % 4 || count <= 0) return CPC_EINVAL;
    if (train && (!dgamma || !dbeta)) return CPC_EINVAL;
    const int nb = blocks_for((long long)gx[0] * gx[1] * gx[2] * (gx[5] / 4));
    const float inv = (float)(1.0 / count);
    if (dtype == CPC_DTYPE_BF16 && !x_f32 && gx[5] % 8 == 0) {
        const int nb8 = blocks_for((long long)gx[0] * gx[1] * gx[2] * (gx[5] / 8));
        hipLaunchKernelGGL(bn_bwd_apply8_kernel, dim3(nb8), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, mk(gy), (const bf16_t*)x, (bf16_t*)dx, mk(gx), stats, gamma, dgamma, dbeta, inv, relu, train, ybits, mk(gy), nullptr, nullptr, mk(gy), 0, 0);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16 && x_f32)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, mk(gy), (const float*)x, (float*)dx, mk(gx), stats, gamma, dgamma, dbeta, inv, relu, train);
    else
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, mk(gy), (const bf16_t*)x, (bf16_t*)dx, mk(gx), stats, gamma, dgamma, dbeta, inv, relu, train),
                  hipLaunchKernelGGL((bn_bwd_apply_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)dy, (const float*)y, mk(gy), (const float*)x, (float*)dx, mk(gx), stats, gamma, dgamma, dbeta, inv, relu, train));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// output extents: ceil mode (windows clipped at the border) or floor mode (the remainder is dropped), per axis
static bool pool_ok(const int* gi, const int* go, int p) {
    if (!(grid_ok(gi) && grid_ok(go) && p >= 1 && gi[0] == go[0] && gi[5] == go[5])) return false;
    const bool w_ok = go[1] == (gi[1] + p - 1) / p || (go[1] == gi[1] / p && go[1] > 0);
    const bool h_ok = go[2] == (gi[2] + p - 1) / p || (go[2] == gi[2] / p && go[2] > 0);
    return w_ok && h_ok;
}

int launch_maxpool2d_fwd(const void* in, const int* gi, void* out, const int* go, int p, int in_f32, int dtype, hipStream_t st) {
    if (!pool_ok(gi, go, p)) return CPC_EINVAL;
    const int nb = blocks_for((long long)go[0] * go[1] * go[2] * go[5]);
    if (!in_f32 && dtype == CPC_DTYPE_BF16 && go[5] % 8 == 0) {
        const int nbv = blocks_for((long long)go[0] * go[1] * go[2] * (go[5] / 8));
        hipLaunchKernelGGL(maxpool2d_fwd_vec8_kernel, dim3(nbv), dim3(256), 0, st, (const bf16_t*)in, mk(gi), (bf16_t*)out, mk(go), p);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (in_f32) {
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((maxpool2d_fwd_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, st, (const float*)in, mk(gi), (bf16_t*)out, mk(go), p),
                  hipLaunchKernelGGL((maxpool2d_fwd_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)in, mk(gi), (float*)out, mk(go), p));
    } else {
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((maxpool2d_fwd_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)in, mk(gi), (bf16_t*)out, mk(go), p),
                  hipLaunchKernelGGL((maxpool2d_fwd_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)in, mk(gi), (float*)out, mk(go), p));
    }
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_maxpool2d_bwd(const void* in, void* din, const int* gi, const void* dout, const int* go, int p, int accumulate, int dtype,
                         hipStream_t st) {
    if (!pool_ok(gi, go, p)) return CPC_EINVAL;
    const int nb = blocks_for((long long)go[0] * go[1] * go[2] * go[5]);
    if (dtype == CPC_DTYPE_BF16 && go[5] % 8 == 0) {
        const int nbv = blocks_for((long long)go[0] * go[1] * go[2] * (go[5] / 8));
        hipLaunchKernelGGL(maxpool2d_bwd_vec8_kernel, dim3(nbv), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)din, mk(gi),
                           (const bf16_t*)dout, mk(go), p, accumulate);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    DISPATCH2(dtype,
              hipLaunchKernelGGL((maxpool2d_bwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)din, mk(gi), (const bf16_t*)dout, mk(go), p, accumulate),
              hipLaunchKernelGGL((maxpool2d_bwd_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)in, (float*)din, mk(gi), (const float*)dout, mk(go), p, accumulate));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

static bool crop_ok(const int* ga, const int* gr, const int* go, int oh, int ow) {
    return grid_ok(ga) && grid_ok(gr) && grid_ok(go) && same_shape(ga, go) && gr[0] == go[0] && gr[5] == go[5] && go[5] % 4 == 0 &&
           oh >= 0 && ow >= 0 && oh + go[2] <= gr[2] && ow + go[1] <= gr[1];
}

static bool res8_enabled() {          // CPC_RES8=0: the 8-byte residual kernels (A/B switch)
    static const bool on = [] { const char* v = getenv("CPC_RES8"); return !(v && v[0] == '0'); }();
    return on;
}

int launch_residual_add(const void* a, const int* ga, const void* r, const int* gr, void* out, const int* go, int oh, int ow, int relu,
                        int r_f32, int dtype, hipStream_t st) {
    if (!crop_ok(ga, gr, go, oh, ow)) return CPC_EINVAL;
    const int nb = blocks_for((long long)go[0] * go[1] * go[2] * (go[5] / 4));
    if (dtype == CPC_DTYPE_BF16 && !r_f32 && go[5] % 8 == 0 && res8_enabled()) {
        const int nb8 = blocks_for((long long)go[0] * go[1] * go[2] * (go[5] / 8));
        hipLaunchKernelGGL(residual_add8_kernel, dim3(nb8), dim3(256), 0, st, (const bf16_t*)a, mk(ga), (const bf16_t*)r, mk(gr), (bf16_t*)out, mk(go),
                           oh, ow, relu);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16 && r_f32)
        hipLaunchKernelGGL((residual_add_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)a, mk(ga), (const float*)r, mk(gr), (bf16_t*)out, mk(go), oh, ow, relu);
    else
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((residual_add_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)a, mk(ga), (const bf16_t*)r, mk(gr), (bf16_t*)out, mk(go), oh, ow, relu),
                  hipLaunchKernelGGL((residual_add_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)a, mk(ga), (const float*)r, mk(gr), (float*)out, mk(go), oh, ow, relu));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_residual_add_bwd(const void* dout, const void* out, const int* go, void* da, const int* ga, void* dr, const int* gr, int oh,
                            int ow, int relu, int r_f32, int dtype, hipStream_t st) {
    if (!crop_ok(ga, gr, go, oh, ow)) return CPC_EINVAL;
    const int nb = blocks_for((long long)go[0] * go[1] * go[2] * (go[5] / 4));
    if (dtype == CPC_DTYPE_BF16 && !r_f32 && go[5] % 8 == 0 && res8_enabled()) {
        const int nb8 = blocks_for((long long)go[0] * go[1] * go[2] * (go[5] / 8));
        hipLaunchKernelGGL(residual_add_bwd8_kernel, dim3(nb8), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out, mk(go), (bf16_t*)da, mk(ga),
                           (bf16_t*)dr, mk(gr), oh, ow, relu);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16 && r_f32)
        hipLaunchKernelGGL((residual_add_bwd_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out, mk(go), (bf16_t*)da, mk(ga), (float*)dr, mk(gr), oh, ow, relu);
    else
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((residual_add_bwd_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out, mk(go), (bf16_t*)da, mk(ga), (bf16_t*)dr, mk(gr), oh, ow, relu),
                  hipLaunchKernelGGL((residual_add_bwd_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)dout, (const float*)out, mk(go), (float*)da, mk(ga), (float*)dr, mk(gr), oh, ow, relu));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_relu_mask(void* g, const void* y, long long n, int dtype, hipStream_t st) {
    if (n <= 0 || n % 4) return CPC_EINVAL;
    const int nb = blocks_for(n / 4);
    DISPATCH2(dtype,
              hipLaunchKernelGGL((relu_mask_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (bf16_t*)g, (const bf16_t*)y, n / 4),
              hipLaunchKernelGGL((relu_mask_kernel<float>), dim3(nb), dim3(256), 0, st, (float*)g, (const float*)y, n / 4));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_accumulate(void* a, const void* b, long long n, int dtype, hipStream_t st) {
    if (n <= 0 || n % 4) return CPC_EINVAL;
    const int nb = blocks_for(n / 4);
    DISPATCH2(dtype,
              hipLaunchKernelGGL((accumulate_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (bf16_t*)a, (const bf16_t*)b, n / 4),
              hipLaunchKernelGGL((accumulate_kernel<float>), dim3(nb), dim3(256), 0, st, (float*)a, (const float*)b, n / 4));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_split3_bf16(const float* src, void* dst, long long n, hipStream_t st) {
    if (n <= 0) return CPC_EINVAL;
    hipLaunchKernelGGL(split3_bf16_kernel, dim3(blocks_for(n)), dim3(256), 0, st, src, (bf16_t*)dst, n);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}


int launch_maxpool2d_select(const void* in, const void* sel, const int* gi, void* out, const int* go, int p, int in_f32, int dtype,
                            hipStream_t st) {
    if (!pool_ok(gi, go, p)) return CPC_EINVAL;
    const int nb = blocks_for((long long)go[0] * go[1] * go[2] * go[5]);
    if (in_f32) {
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((maxpool2d_select_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, st, (const float*)in, (const float*)sel, mk(gi), (bf16_t*)out, mk(go), p),
                  hipLaunchKernelGGL((maxpool2d_select_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)in, (const float*)sel, mk(gi), (float*)out, mk(go), p));
    } else {
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((maxpool2d_select_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)in, (const bf16_t*)sel, mk(gi), (bf16_t*)out, mk(go), p),
                  hipLaunchKernelGGL((maxpool2d_select_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)in, (const float*)sel, mk(gi), (float*)out, mk(go), p));
    }
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_gp_direction(const float* g, float* v, long long npix, int C, float factor, float* partial, int nblocks, hipStream_t st) {
    if (npix <= 0 || C <= 0 || nblocks <= 0) return CPC_EINVAL;
    hipLaunchKernelGGL(gp_direction_kernel, dim3(nblocks), dim3(256), 0, st, g, v, npix, C, factor, partial);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_bn_gp_cross(const void* x, const void* yt, const void* delta, void* out, const int* gx, const float* stats, const float* coef,
                       int x_f32, int dtype, hipStream_t st) {
    if (!grid_ok(gx)) return CPC_EINVAL;
    const int nb = blocks_for((long long)gx[0] * gx[1] * gx[2] * gx[5]);
    if (dtype == CPC_DTYPE_BF16 && x_f32)
        hipLaunchKernelGGL((bn_gp_cross_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, st, (const float*)x, (const bf16_t*)yt, (const float*)delta, (float*)out, mk(gx), stats, coef);
    else
        DISPATCH2(dtype,
                  hipLaunchKernelGGL((bn_gp_cross_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)yt, (const bf16_t*)delta, (bf16_t*)out, mk(gx), stats, coef),
                  hipLaunchKernelGGL((bn_gp_cross_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)yt, (const float*)delta, (float*)out, mk(gx), stats, coef));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
