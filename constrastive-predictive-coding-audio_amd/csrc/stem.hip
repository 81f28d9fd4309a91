// First block of the scalogram encoder ("stem"): Conv2d on the raw float32 scalogram (1 or 2 input channels) + train-mode
// BatchNorm2d + ReLU  (reference scalogram_model.py:392-406, first ScalogramEncoderBlock), and the block's residual branch
// MaxPool2d -> 1x1 Conv2d on the same float32 input (:434-446) with the cropped add (:462-472).
//
// The convolution has 18 taps or so and 32 outputs over 5 M positions: as im2col + GEMM it moves a 32-wide float32 matrix three
// times per direction (0.65 GB each at BASELINE configs[2]) and runs at 10-30 TF/s.  Here its output is never stored: every pass
// that needs it (statistics, normalise + ReLU, the two backward passes of the BatchNorm, the weight gradient) RECOMPUTES it from the
// input columns held in LDS -- 576 multiply-adds per position against the 128 bytes it would take to load the stored value.  The
// recomputation runs the same instruction sequence in every kernel (conv_rows), so all passes see bit-identical pre-normalisation
// values.  The weight gradient is taken of the BatchNorm's input gradient as it is formed (never stored either), and the bias
// gradient of a convolution in front of a train-mode BatchNorm is exactly zero.
//
// One workgroup = one output column (b, wo): Ho rows x Cout channels; lane = (channel quad, group of 4 consecutive rows).
#include "cpc_common.h"
#include "cpc_kernels.h"
#include <algorithm>

namespace {

struct Grid {
    int B, W, H, Ha, top, C;
};
__device__ __forceinline__ long long grid_off(const Grid& g, int b, int w, int h) {
    return (((long long)b * g.W + w) * g.Ha + g.top + h) * g.C;
}

struct StemConv {
    const float* x;      // input grid, float32
    Grid gx;
    const float* w;      // [Cout][Cin][kh][kw] (the reference's Conv2d weight)
    const float* bias;   // [Cout] or null
    int Cout, kh, kw, sh, sw, ph, pw, Ho, Wo;
};

constexpr int R = 4;              // output rows per lane
constexpr int MAX_XS = 4096;      // floats of input columns in LDS (kw * (Hin + 2 ph) * Cin)
constexpr int MAX_WL = 2304;      // floats of weights in LDS (taps * Cout)

// weights -> LDS as wl[t][co], t = (c*kh + dh)*kw + dw  (so that a lane's 4 output channels are one 16-byte read)
__device__ __forceinline__ void stage_weights(const StemConv& p, float* wl) {
    const int T = p.gx.C * p.kh * p.kw;
    for (int i = threadIdx.x; i < T * p.Cout; i += 256) {
        const int co = i % p.Cout, t = i / p.Cout;
        wl[i] = p.w[co * T + t];
    }
}

// the kw input columns of output column (b, wo) -> LDS as xs[dw][(ph + h) * Cin + c], zero where the window leaves the grid
__device__ __forceinline__ void stage_columns(const StemConv& p, int b, int wo, float* xs) {
    const int Cin = p.gx.C, Hp = p.gx.H + 2 * p.ph, colf = Hp * Cin;
    for (int i = threadIdx.x; i < p.kw * colf; i += 256) {
        const int dw = i / colf, j = i - dw * colf;
        const int h = j / Cin - p.ph, c = j - (j / Cin) * Cin;
        const int w = wo * p.sw - p.pw + dw;
        float v = 0.f;
        if (w >= 0 && w < p.gx.W && h >= 0 && h < p.gx.H) v = p.x[grid_off(p.gx, b, w, h) + c];
        xs[i] = v;
    }
}

// acc[r] = bias + sum over taps of w * x for output rows ho0 + r (r < R; rows beyond Ho compute on clamped addresses and are
// discarded by the caller), channels 4 cq .. 4 cq + 3.  Fixed order: t = (c, dh, dw) ascending.
__device__ __forceinline__ void conv_rows(const StemConv& p, const float* wl, const float* xs, int cq, int ho0, f32x4 acc[R]) {
    const int Cin = p.gx.C, colf = (p.gx.H + 2 * p.ph) * Cin;
    const f32x4 b4 = p.bias ? *(const f32x4*)(p.bias + cq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    int rowbase[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] = b4;
        rowbase[r] = min(ho0 + r, p.Ho - 1) * p.sh * Cin;
    }
    int t = 0;
    for (int c = 0; c < Cin; ++c)
        for (int dh = 0; dh < p.kh; ++dh)
            for (int dw = 0; dw < p.kw; ++dw, ++t) {
                const f32x4 w4 = *(const f32x4*)(wl + t * p.Cout + cq * 4);
                const float* col = xs + dw * colf + dh * Cin + c;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float xv = col[rowbase[r]];
                    acc[r] += w4 * xv;
                }
            }
}

// block-wide sum of per-lane f32x4 pairs over the lanes that share a channel quad (fixed order), result by the lanes rg == 0
template <int NV>
__device__ __forceinline__ void reduce_quads(f32x4 (&v)[NV], float* red, int cqn, int cq, int rg, int nrg) {
    // red: [nrg][NV][cqn*4] floats
    const int C = cqn * 4;
#pragma unroll
    for (int k = 0; k < NV; ++k) *(f32x4*)(red + (rg * NV + k) * C + cq * 4) = v[k];
    __syncthreads();
    if (rg == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < nrg; ++q) s += *(const f32x4*)(red + (q * NV + k) * C + cq * 4);
            v[k] = s;
        }
    }
    __syncthreads();
}

// ---- pass 1: per-workgroup partial sums of y and y^2 (BatchNorm statistics), slabs[blk][2][Cout]
__global__ __launch_bounds__(256) void stem_stats_kernel(StemConv p, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float wl[MAX_WL];
    __shared__ __attribute__((aligned(16))) float xs[MAX_XS];
    __shared__ __attribute__((aligned(16))) float red[2048];
    const int cqn = p.Cout / 4, cq = threadIdx.x % cqn, rg = threadIdx.x / cqn, nrg = 256 / cqn;
    stage_weights(p, wl);
    f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ncol = p.gx.B * p.Wo;
    for (int q = blockIdx.x; q < ncol; q += gridDim.x) {
        __syncthreads();
        stage_columns(p, q / p.Wo, q % p.Wo, xs);
        __syncthreads();
        for (int ho0 = rg * R; ho0 < p.Ho; ho0 += nrg * R) {
            f32x4 acc[R];
            conv_rows(p, wl, xs, cq, ho0, acc);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (ho0 + r < p.Ho) {
                    s[0] += acc[r];
                    s[1] += acc[r] * acc[r];
                }
        }
    }
    __syncthreads();
    reduce_quads<2>(s, red, cqn, cq, rg, nrg);
    if (rg == 0) {
        *(f32x4*)(slabs + (long long)blockIdx.x * 2 * p.Cout + cq * 4) = s[0];
        *(f32x4*)(slabs + (long long)blockIdx.x * 2 * p.Cout + p.Cout + cq * 4) = s[1];
    }
}

// ---- pass 2: a = relu((y - mean) * rstd * gamma + beta) into the activation grid (storage dtype, its own row geometry)
template <typename T>
__global__ __launch_bounds__(256) void stem_apply_kernel(StemConv p, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ out, Grid go) {
    __shared__ __attribute__((aligned(16))) float wl[MAX_WL];
    __shared__ __attribute__((aligned(16))) float xs[MAX_XS];
    const int cqn = p.Cout / 4, cq = threadIdx.x % cqn, rg = threadIdx.x / cqn, nrg = 256 / cqn;
    stage_weights(p, wl);
    const f32x4 mu = *(const f32x4*)(stats + cq * 4), rs = *(const f32x4*)(stats + p.Cout + cq * 4);
    const f32x4 ga = *(const f32x4*)(gamma + cq * 4), be = *(const f32x4*)(beta + cq * 4);
    const f32x4 k = rs * ga;
    const int ncol = p.gx.B * p.Wo;
    for (int q = blockIdx.x; q < ncol; q += gridDim.x) {
        const int b = q / p.Wo, wo = q % p.Wo;
        __syncthreads();
        stage_columns(p, b, wo, xs);
        __syncthreads();
        for (int ho0 = rg * R; ho0 < p.Ho; ho0 += nrg * R) {
            f32x4 acc[R];
            conv_rows(p, wl, xs, cq, ho0, acc);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (ho0 + r < p.Ho) {
                    f32x4 o = (acc[r] - mu) * k + be;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = relu_f(o[e]);
                    store4(out + grid_off(go, b, wo, ho0 + r) + cq * 4, o);
                }
        }
    }
}

// ---- backward pass 1: partial sums of g * xhat and g, g = da * (a > 0);  slabs[blk][2][Cout]
template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_reduce_kernel(StemConv p, const float* __restrict__ stats, const T* __restrict__ da,
                                                              const T* __restrict__ a, Grid ga_, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float wl[MAX_WL];
    __shared__ __attribute__((aligned(16))) float xs[MAX_XS];
    __shared__ __attribute__((aligned(16))) float red[2048];
    const int cqn = p.Cout / 4, cq = threadIdx.x % cqn, rg = threadIdx.x / cqn, nrg = 256 / cqn;
    stage_weights(p, wl);
    const f32x4 mu = *(const f32x4*)(stats + cq * 4), rs = *(const f32x4*)(stats + p.Cout + cq * 4);
    f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ncol = p.gx.B * p.Wo;
    for (int q = blockIdx.x; q < ncol; q += gridDim.x) {
        const int b = q / p.Wo, wo = q % p.Wo;
        __syncthreads();
        stage_columns(p, b, wo, xs);
        __syncthreads();
        for (int ho0 = rg * R; ho0 < p.Ho; ho0 += nrg * R) {
            f32x4 g4[R], a4[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const long long o = grid_off(ga_, b, wo, min(ho0 + r, p.Ho - 1)) + cq * 4;
                g4[r] = load4(da + o);
                a4[r] = load4(a + o);
            }
            f32x4 acc[R];
            conv_rows(p, wl, xs, cq, ho0, acc);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (ho0 + r < p.Ho) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = a4[r][e] > 0.f ? g4[r][e] : 0.f;
                        s[0][e] += g * (acc[r][e] - mu[e]) * rs[e];
                        s[1][e] += g;
                    }
                }
        }
    }
    __syncthreads();
    reduce_quads<2>(s, red, cqn, cq, rg, nrg);
    if (rg == 0) {
        *(f32x4*)(slabs + (long long)blockIdx.x * 2 * p.Cout + cq * 4) = s[0];
        *(f32x4*)(slabs + (long long)blockIdx.x * 2 * p.Cout + p.Cout + cq * 4) = s[1];
    }
}

// ---- backward pass 2: dy = gamma rstd (g - dbeta / n - xhat dgamma / n) formed per position and contracted with the input window at
// once: slabs[blk][co][t] partial sums of dy[co] * x[tap t]  (the convolution's weight gradient in the reference's layout).
// The Cin * kh * kw accumulators per channel live in registers, so the window shape is a template parameter (the loops over taps
// unroll and every accumulator has a fixed register).
template <typename T, int CIN, int KH, int KW>
__global__ __launch_bounds__(256) void stem_bwd_wgrad_kernel(StemConv p, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                             float inv_count, const T* __restrict__ da, const T* __restrict__ a, Grid ga_,
                                                             float* __restrict__ slabs) {
    constexpr int TAPS = CIN * KH * KW;
    __shared__ __attribute__((aligned(16))) float wl[MAX_WL];
    __shared__ __attribute__((aligned(16))) float xs[MAX_XS];
    __shared__ __attribute__((aligned(16))) float red[4096];          // one chunk of 4 taps at a time: [nrg][4][Cout]
    const int cqn = p.Cout / 4, cq = threadIdx.x % cqn, rg = threadIdx.x / cqn, nrg = 256 / cqn;
    const int colf = (p.gx.H + 2 * p.ph) * CIN;
    stage_weights(p, wl);
    const f32x4 mu = *(const f32x4*)(stats + cq * 4), rs = *(const f32x4*)(stats + p.Cout + cq * 4);
    f32x4 k1, k2, k3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cq * 4 + e;
        k1[e] = gamma[c] * rs[e];
        k2[e] = k1[e] * dbeta[c] * inv_count;
        k3[e] = k1[e] * rs[e] * dgamma[c] * inv_count;
    }
    f32x4 dw[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) dw[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ncol = p.gx.B * p.Wo;
    for (int q = blockIdx.x; q < ncol; q += gridDim.x) {
        const int b = q / p.Wo, wo = q % p.Wo;
        __syncthreads();
        stage_columns(p, b, wo, xs);
        __syncthreads();
        for (int ho0 = rg * R; ho0 < p.Ho; ho0 += nrg * R) {
            f32x4 g4[R], a4[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const long long o = grid_off(ga_, b, wo, min(ho0 + r, p.Ho - 1)) + cq * 4;
                g4[r] = load4(da + o);
                a4[r] = load4(a + o);
            }
            f32x4 acc[R];
            conv_rows(p, wl, xs, cq, ho0, acc);
            int rowbase[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = a4[r][e] > 0.f ? g4[r][e] : 0.f;
                    acc[r][e] = (ho0 + r < p.Ho) ? k1[e] * g - k2[e] - k3[e] * (acc[r][e] - mu[e]) : 0.f;      // dy
                }
                rowbase[r] = min(ho0 + r, p.Ho - 1) * p.sh * CIN;
            }
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int dh = 0; dh < KH; ++dh)
#pragma unroll
                    for (int dwi = 0; dwi < KW; ++dwi) {
                        const float* col = xs + dwi * colf + dh * CIN + c;
#pragma unroll
                        for (int r = 0; r < R; ++r) dw[(c * KH + dh) * KW + dwi] += acc[r] * col[rowbase[r]];
                    }
        }
    }
    // reduction over the row groups, 4 taps at a time (fixed order)
    for (int t0 = 0; t0 < TAPS; t0 += 4) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
            if (t >= t0 && t < t0 + 4) *(f32x4*)(red + (rg * 4 + (t - t0)) * p.Cout + cq * 4) = dw[t];
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * p.Cout; i += 256) {
            const int u = i / p.Cout, co = i - u * p.Cout;
            if (t0 + u >= TAPS) continue;
            float s = 0.f;
            for (int qg = 0; qg < nrg; ++qg) s += red[(qg * 4 + u) * p.Cout + co];
            slabs[(long long)blockIdx.x * p.Cout * TAPS + co * TAPS + t0 + u] = s;
        }
    }
}

// ---- residual branch of the first block: out = act(main + W_r xp(w + ow, h + oh)), xp the pooled float32 input (Cin channels),
// W_r [Cout][Cin] the 1x1 projection (no bias, no padding).
template <typename T>
__global__ __launch_bounds__(256) void stem_residual_add_kernel(const T* __restrict__ main_, Grid gm, const float* __restrict__ xp, Grid gp,
                                                                const float* __restrict__ wr, T* __restrict__ out, Grid go, int oh, int ow,
                                                                int relu) {
    const int c4n = gm.C / 4, Cin = gp.C;
    const unsigned total = (unsigned)((long long)gm.B * gm.W * gm.H * c4n);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const int c4 = (int)(idx % c4n);
        const int h = (int)((idx / c4n) % gm.H);
        const unsigned col = idx / (unsigned)(c4n * gm.H);
        const int w = (int)(col % gm.W), b = (int)(col / gm.W);
        f32x4 v = load4(main_ + grid_off(gm, b, w, h) + c4 * 4);
        const float* xr = xp + grid_off(gp, b, w + ow, h + oh);
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = xr[ci];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += wr[(c4 * 4 + e) * Cin + ci] * xv;
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
        }
        store4(out + grid_off(go, b, w, h) + c4 * 4, v);
    }
}

// backward: g = dout * (out > 0 if relu);  dmain = g;  slabs[blk][co][ci] partial sums of g[co] * xp[ci]  (gradient of W_r)
template <typename T>
__global__ __launch_bounds__(256) void stem_residual_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ out, Grid go,
                                                                T* __restrict__ dmain, Grid gm, const float* __restrict__ xp, Grid gp,
                                                                float* __restrict__ slabs, int oh, int ow, int relu, long long cols_per_block) {
    __shared__ __attribute__((aligned(16))) float red[4096];
    const int C = gm.C, c4n = C / 4, Cin = gp.C;          // Cin <= 4
    const int cg = threadIdx.x % c4n, rp = threadIdx.x / c4n, nrp = 256 / c4n;
    const long long ncol = (long long)gm.B * gm.W;
    const long long q0 = (long long)blockIdx.x * cols_per_block, q1 = min(ncol, q0 + cols_per_block);
    f32x4 s[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (rp < nrp && q0 < q1) {
        const unsigned rend = (unsigned)(q1 * gm.H);
        for (unsigned r = (unsigned)(q0 * gm.H) + rp; r < rend; r += nrp) {
            const unsigned q = r / (unsigned)gm.H;
            const int h = (int)(r - q * gm.H), w = (int)(q % gm.W), b = (int)(q / gm.W);
            const long long oo = grid_off(go, b, w, h) + cg * 4;
            f32x4 g = load4(dout + oo);
            if (relu) {
                const f32x4 y = load4(out + oo);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
            }
            store4(dmain + grid_off(gm, b, w, h) + cg * 4, g);
            const float* xr = xp + grid_off(gp, b, w + ow, h + oh);
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
                if (ci < Cin) s[ci] += g * xr[ci];
        }
    }
    // red[rp][ci][C]
    if (rp < nrp) {
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
            if (ci < Cin) *(f32x4*)(red + (rp * Cin + ci) * C + cg * 4) = s[ci];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Cin * C; i += 256) {
        const int ci = i / C, co = i - ci * C;
        float acc = 0.f;
        for (int qg = 0; qg < nrp; ++qg) acc += red[(qg * Cin + ci) * C + co];
        slabs[(long long)blockIdx.x * C * Cin + co * Cin + ci] = acc;
    }
}

bool grid_ok(const int* g) {
    return g && g[0] > 0 && g[1] > 0 && g[2] > 0 && g[4] >= 0 && g[3] >= g[4] + g[2] && g[5] > 0 &&
           (long long)g[0] * g[1] * g[2] * g[5] < (1ll << 31);
}
Grid mk(const int* g) { return Grid{g[0], g[1], g[2], g[3], g[4], g[5]}; }

bool stem_ok(const int* gx, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo) {
    if (!grid_ok(gx) || gx[4] != 0) return false;
    const int Cin = gx[5], T = Cin * kh * kw;
    if (Cin > 4 || Cout % 4 || Cout < 4 || Cout > 64 || 256 % (Cout / 4)) return false;
    if (kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0) return false;
    if (T * Cout > MAX_WL || kw * (gx[2] + 2 * ph) * Cin > MAX_XS) return false;
    if (Ho != (gx[2] + 2 * ph - kh) / sh + 1 || Wo != (gx[1] + 2 * pw - kw) / sw + 1 || Ho < 1 || Wo < 1) return false;
    return true;
}

StemConv mkconv(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                int Ho, int Wo) {
    StemConv p;
    p.x = x; p.gx = mk(gx); p.w = w; p.bias = bias;
    p.Cout = Cout; p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.Ho = Ho; p.Wo = Wo;
    return p;
}

}  // namespace

int launch_stem_stats(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw, int ph,
                      int pw, int Ho, int Wo, float* slabs, int nblocks, hipStream_t st) {
    if (!stem_ok(gx, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo) || nblocks <= 0) return CPC_EINVAL;
    hipLaunchKernelGGL(stem_stats_kernel, dim3(nblocks), dim3(256), 0, st, mkconv(x, gx, w, bias, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo), slabs);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_stem_apply(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw, int ph,
                      int pw, int Ho, int Wo, const float* stats, const float* gamma, const float* beta, void* out, const int* go,
                      int nblocks, int dtype, hipStream_t st) {
    if (!stem_ok(gx, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo) || !grid_ok(go) || nblocks <= 0) return CPC_EINVAL;
    if (go[0] != gx[0] || go[1] != Wo || go[2] != Ho || go[5] != Cout) return CPC_EINVAL;
    const StemConv p = mkconv(x, gx, w, bias, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo);
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((stem_apply_kernel<bf16_t>), dim3(nblocks), dim3(256), 0, st, p, stats, gamma, beta, (bf16_t*)out, mk(go));
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((stem_apply_kernel<float>), dim3(nblocks), dim3(256), 0, st, p, stats, gamma, beta, (float*)out, mk(go));
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_stem_bwd_reduce(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw,
                           int ph, int pw, int Ho, int Wo, const float* stats, const void* da, const void* a, const int* ga, float* slabs,
                           int nblocks, int dtype, hipStream_t st) {
    if (!stem_ok(gx, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo) || !grid_ok(ga) || nblocks <= 0) return CPC_EINVAL;
    if (ga[0] != gx[0] || ga[1] != Wo || ga[2] != Ho || ga[5] != Cout) return CPC_EINVAL;
    const StemConv p = mkconv(x, gx, w, bias, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo);
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((stem_bwd_reduce_kernel<bf16_t>), dim3(nblocks), dim3(256), 0, st, p, stats, (const bf16_t*)da, (const bf16_t*)a, mk(ga), slabs);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((stem_bwd_reduce_kernel<float>), dim3(nblocks), dim3(256), 0, st, p, stats, (const float*)da, (const float*)a, mk(ga), slabs);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// window shapes with a register-resident weight-gradient kernel: (Cin, kh, kw)
static bool stem_wgrad_shape_ok(int cin, int kh, int kw) {
    return (cin == 2 && kh == 3 && kw == 3) || (cin == 1 && kh == 3 && kw == 3) || (cin == 1 && kh == 5 && kw == 1) ||
           (cin == 2 && kh == 5 && kw == 1) || (cin == 1 && kh == 2 && kw == 2) || (cin == 2 && kh == 2 && kw == 2);
}

template <typename T>
static int stem_wgrad_dispatch(int cin, int kh, int kw, const StemConv& p, const float* stats, const float* gamma, const float* dgamma,
                               const float* dbeta, float inv, const T* da, const T* a, const Grid& ga, float* slabs, int nblocks,
                               hipStream_t st) {
#define STEM_CASE(CI, KH, KW)                                                                                                          \
    if (cin == CI && kh == KH && kw == KW) {                                                                                          \
        hipLaunchKernelGGL((stem_bwd_wgrad_kernel<T, CI, KH, KW>), dim3(nblocks), dim3(256), 0, st, p, stats, gamma, dgamma, dbeta, inv, da, \
                           a, ga, slabs);                                                                                             \
        return CPC_OK;                                                                                                                \
    }
    STEM_CASE(2, 3, 3)
    STEM_CASE(1, 3, 3)
    STEM_CASE(1, 5, 1)
    STEM_CASE(2, 5, 1)
    STEM_CASE(1, 2, 2)
    STEM_CASE(2, 2, 2)
#undef STEM_CASE
    return CPC_EINVAL;
}

int launch_stem_supported(int cin, int cout, int kh, int kw, int hin, int ph) {
    if (cin < 1 || cin > 4 || cout % 4 || cout < 4 || cout > 64 || 256 % (cout / 4)) return 0;
    if (cin * kh * kw * cout > MAX_WL || kw * (hin + 2 * ph) * cin > MAX_XS) return 0;
    return stem_wgrad_shape_ok(cin, kh, kw) ? 1 : 0;
}

int launch_stem_bwd_wgrad(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw,
                          int ph, int pw, int Ho, int Wo, const float* stats, const float* gamma, const float* dgamma, const float* dbeta,
                          double count, const void* da, const void* a, const int* ga, float* slabs, int nblocks, int dtype, hipStream_t st) {
    if (!stem_ok(gx, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo) || !grid_ok(ga) || nblocks <= 0 || count <= 0) return CPC_EINVAL;
    if (ga[0] != gx[0] || ga[1] != Wo || ga[2] != Ho || ga[5] != Cout) return CPC_EINVAL;
    if (!stem_wgrad_shape_ok(gx[5], kh, kw)) return CPC_EINVAL;
    const StemConv p = mkconv(x, gx, w, bias, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo);
    const float inv = (float)(1.0 / count);
    int rc;
    if (dtype == CPC_DTYPE_BF16)
        rc = stem_wgrad_dispatch<bf16_t>(gx[5], kh, kw, p, stats, gamma, dgamma, dbeta, inv, (const bf16_t*)da, (const bf16_t*)a, mk(ga), slabs, nblocks, st);
    else if (dtype == CPC_DTYPE_F32)
        rc = stem_wgrad_dispatch<float>(gx[5], kh, kw, p, stats, gamma, dgamma, dbeta, inv, (const float*)da, (const float*)a, mk(ga), slabs, nblocks, st);
    else
        return CPC_EINVAL;
    if (rc != CPC_OK) return rc;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

static bool residual_ok(const int* gm, const int* gp, const int* go, int oh, int ow) {
    if (!grid_ok(gm) || !grid_ok(gp) || !grid_ok(go)) return false;
    if (gm[0] != go[0] || gm[1] != go[1] || gm[2] != go[2] || gm[5] != go[5] || gp[0] != gm[0]) return false;
    if (gm[5] % 4 || gm[5] > 1024 || 256 % (gm[5] / 4) || gp[5] > 4 || oh < 0 || ow < 0) return false;
    return gm[1] + ow <= gp[1] && gm[2] + oh <= gp[2];
}

int launch_stem_residual_add(const void* main_, const int* gm, const float* xp, const int* gp, const float* wr, void* out, const int* go,
                             int oh, int ow, int relu, int dtype, hipStream_t st) {
    if (!residual_ok(gm, gp, go, oh, ow)) return CPC_EINVAL;
    const int nb = (int)std::min<long long>(8192, ((long long)gm[0] * gm[1] * gm[2] * (gm[5] / 4) + 255) / 256);
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((stem_residual_add_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)main_, mk(gm), xp, mk(gp), wr, (bf16_t*)out, mk(go), oh, ow, relu);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((stem_residual_add_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)main_, mk(gm), xp, mk(gp), wr, (float*)out, mk(go), oh, ow, relu);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_stem_residual_bwd(const void* dout, const void* out, const int* go, void* dmain, const int* gm, const float* xp, const int* gp,
                             float* slabs, int oh, int ow, int relu, int nblocks, int dtype, hipStream_t st) {
    if (!residual_ok(gm, gp, go, oh, ow) || nblocks <= 0) return CPC_EINVAL;
    if (256 / (gm[5] / 4) * gp[5] * gm[5] > 4096) return CPC_EINVAL;
    const long long ncol = (long long)gm[0] * gm[1];
    const long long cpb = (ncol + nblocks - 1) / nblocks;
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((stem_residual_bwd_kernel<bf16_t>), dim3(nblocks), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out, mk(go), (bf16_t*)dmain, mk(gm), xp, mk(gp), slabs, oh, ow, relu, cpb);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((stem_residual_bwd_kernel<float>), dim3(nblocks), dim3(256), 0, st, (const float*)dout, (const float*)out, mk(go), (float*)dmain, mk(gm), xp, mk(gp), slabs, oh, ow, relu, cpb);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
