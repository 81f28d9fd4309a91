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
// (grids hold fewer than 2^31 elements, checked by the launchers: 32-bit arithmetic for the per-row accesses of the hot loops)
__device__ __forceinline__ unsigned grid_off32(const Grid& g, int b, int w, int h) {
    return (unsigned)(((b * g.W + w) * g.Ha + g.top + h) * g.C);
}

struct StemConv {
    const float* x;      // input grid, float32
    Grid gx;
    const float* w;      // [Cout][Cin][kh][kw] (the reference's Conv2d weight)
    const float* bias;   // [Cout] or null
    int Cout, kh, kw, sh, sw, ph, pw, Ho, Wo;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int R = 4;              // output rows per lane
constexpr int MAX_XS = 1664;      // floats of ONE buffer of input columns in LDS (kw * (Hin + 2 ph + R * sh) * Cin); two buffers
constexpr int MAX_WL = 2304;      // floats of weights in LDS (taps * Cout)
constexpr int MAXPF = 6;          // input-column floats a thread stages per output column (256 threads: kw * (Hin + 2 ph) * Cin <= 1536:
                                  // three columns of 256 bins x 2 channels; every slot costs a predicated load per column)

// Window shape as template parameters (0 = taken from the arguments at run time).  With a compile-time shape the tap loops unroll,
// the weights of a lane's 4 channels live in registers and a lane's input window is read from LDS once per column of taps.
template <int CIN_, int KH_, int KW_, int SH_>
struct Win {
    static constexpr bool FIXED = CIN_ > 0;
    static constexpr int T = FIXED ? CIN_ * KH_ * KW_ : 1;
    static constexpr int WROWS = FIXED ? ((R - 1) * SH_ + KH_) * CIN_ : 1;      // floats of a lane's window in one input column
    __device__ static __forceinline__ int cin(const StemConv& p) { return FIXED ? CIN_ : p.gx.C; }
    __device__ static __forceinline__ int kh(const StemConv& p) { return FIXED ? KH_ : p.kh; }
    __device__ static __forceinline__ int kw(const StemConv& p) { return FIXED ? KW_ : p.kw; }
    __device__ static __forceinline__ int sh(const StemConv& p) { return FIXED ? SH_ : p.sh; }
};

// LDS column stride in floats: the padded input height plus R * sh rows of slack (zero), so that the window of a lane whose last
// rows lie beyond Ho stays inside the buffer
// (a multiple of 4 floats: a lane's window starts at ho0 * sh * Cin with ho0 a multiple of R = 4, so its reads are 16-byte aligned)
template <class W>
__device__ __forceinline__ int col_stride(const StemConv& p) { return ((p.gx.H + 2 * p.ph + R * W::sh(p)) * W::cin(p) + 3) & ~3; }

// a lane's window of one input column: WROWS consecutive floats from a 16-byte aligned LDS address
template <int N>
__device__ __forceinline__ void load_window(const float* col, float (&xin)[N]) {
    const f32x4* c4 = (const f32x4*)__builtin_assume_aligned(col, 16);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const f32x4 v = c4[i];
        xin[4 * i] = v[0]; xin[4 * i + 1] = v[1]; xin[4 * i + 2] = v[2]; xin[4 * i + 3] = v[3];
    }
#pragma unroll
    for (int i = N / 4 * 4; i < N; ++i) xin[i] = col[i];
}

// weights -> LDS as wl[t][co], t = (c*kh + dh)*kw + dw  (a lane's 4 output channels are one 16-byte read)
__device__ __forceinline__ void stage_weights(const StemConv& p, float* wl) {
    const int T = p.gx.C * p.kh * p.kw;
    for (int i = threadIdx.x; i < T * p.Cout; i += 256) {
        const int co = i % p.Cout, t = i / p.Cout;
        wl[i] = p.w[co * T + t];
    }
}

// Staging of the kw input columns of output column (b, wo) as xs[dw][(ph + h) * Cin + c] (zero where the window leaves the grid):
// every thread owns up to MAXPF floats of a column set; the NEXT column set is loaded into registers while the current one is
// being convolved, and written to the other LDS buffer afterwards.
template <class W>
struct ColumnStage {
    // per slot: LDS index (12 bits; 0xfff = slot unused) | (offset from the input column's first element + 1) << 12 (0 = padding
    // row) | dw << 28.  The per-column work is a handful of 32-bit operations per slot and no branch: addresses are clamped into
    // the grid and the value is selected afterwards.
    int code[MAXPF];
    float v[MAXPF];
    int npf;
    __device__ __forceinline__ void init(const StemConv& p) {
        const int Cin = W::cin(p), Hp = p.gx.H + 2 * p.ph, colf = Hp * Cin, cs = col_stride<W>(p), n = W::kw(p) * colf;
        npf = (n + 255) / 256;
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) {
            const int i = threadIdx.x + k * 256;
            code[k] = 0xfff;
            if (i < n) {
                const int dw = i / colf, j = i - dw * colf, h = j / Cin - p.ph, c = j - (j / Cin) * Cin;
                const int rel1 = (h >= 0 && h < p.gx.H) ? h * Cin + c + 1 : 0;
                code[k] = (dw * cs + j) | (rel1 << 12) | (dw << 28);
            }
        }
    }
    // (b, wo) are uniform over the workgroup
    __device__ __forceinline__ void fetch(const StemConv& p, int b, int wo) {
        const int w0 = wo * p.sw - p.pw, HaC = p.gx.Ha * W::cin(p);
        const float* xb = p.x + ((long long)b * p.gx.W * p.gx.Ha + p.gx.top) * W::cin(p);
        // (a predicated load per slot: the value stays in flight until commit(); a branch-free clamp-and-select form made the
        // compiler wait for the loads right here, and the kernels ran 1.5x slower)
        // (and plain, unpredicated loads for the unpadded case were SUNK by the compiler to their use in commit(), behind the column's
        // arithmetic: no prefetch at all, 1.8x slower.  The branch around each load is what keeps it up here.)
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) {
            const int w = w0 + (int)((unsigned)code[k] >> 28), rel1 = (code[k] >> 12) & 0xffff;
            v[k] = (k < npf && rel1 > 0 && w >= 0 && w < p.gx.W) ? xb[w * HaC + rel1 - 1] : 0.f;
        }
    }
    __device__ __forceinline__ void commit(float* xs) const {
#pragma unroll
        for (int k = 0; k < MAXPF; ++k)
            if (k < npf && (code[k] & 0xfff) != 0xfff) xs[code[k] & 0xfff] = v[k];
    }
};

// output column index -> (b, wo), advanced by a fixed step without divisions
struct ColIdx {
    int q, b, wo;
    __device__ __forceinline__ ColIdx(int q0, int Wo) : q(q0), b(q0 / Wo), wo(q0 % Wo) {}
    __device__ __forceinline__ void step(int dq, int db, int dwo, int Wo) {
        q += dq; b += db; wo += dwo;
        if (wo >= Wo) { wo -= Wo; ++b; }
    }
};

// four channels of one position in the storage type, unconverted (half the registers of f32x4 for bf16 while a load is in flight)
template <typename T> struct Raw4;
template <> struct Raw4<float> {
    f32x4 v;
    __device__ __forceinline__ void load(const float* p) { v = *(const f32x4*)p; }
    __device__ __forceinline__ f32x4 get() const { return v; }
};
template <> struct Raw4<bf16_t> {
    uint2 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *(const uint2*)p; }
    __device__ __forceinline__ f32x4 get() const {
        return (f32x4){__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                       __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u)};
    }
};

// the channels a lane owns (4 as f32x4, or 2 as f32x2 for the kernel whose accumulators fill the register file) in the storage type
template <typename T, typename VT> struct RawV;
template <typename T> struct RawV<T, f32x4> : Raw4<T> {};
template <> struct RawV<float, f32x2> {
    f32x2 v;
    __device__ __forceinline__ void load(const float* p) { v = *(const f32x2*)p; }
    __device__ __forceinline__ f32x2 get() const { return v; }
};
template <> struct RawV<bf16_t, f32x2> {
    unsigned v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *(const unsigned*)p; }
    __device__ __forceinline__ f32x2 get() const { return (f32x2){__builtin_bit_cast(float, v << 16), __builtin_bit_cast(float, v & 0xffff0000u)}; }
};
template <typename VT> __device__ __forceinline__ VT vzero();
template <> __device__ __forceinline__ f32x4 vzero<f32x4>() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
template <> __device__ __forceinline__ f32x2 vzero<f32x2>() { return (f32x2){0.f, 0.f}; }

// acc[r] = bias + sum over taps of w * x for output rows ho0 + r (r < R; rows beyond Ho read the zero slack rows and are discarded
// by the caller), channels 4 cq .. 4 cq + 3.  One fixed order of summation per window shape: every pass sees identical values.
template <class W, bool WREG, typename VT>
__device__ __forceinline__ void conv_rows(const StemConv& p, const float* wl, const VT (&wreg)[W::T], const float* xs, int cq,
                                          int ho0, VT acc[R]) {
    constexpr int CPL = sizeof(VT) / 4;          // channels per lane
    const int Cin = W::cin(p), KH = W::kh(p), KW = W::kw(p), SH = W::sh(p), cs = col_stride<W>(p);
    const VT b4 = p.bias ? *(const VT*)(p.bias + cq * CPL) : vzero<VT>();
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = b4;
    if constexpr (W::FIXED) {
#pragma unroll
        for (int dw = 0; dw < KW; ++dw) {
            float xin[W::WROWS];
            load_window<W::WROWS>(xs + dw * cs + ho0 * SH * Cin, xin);
#pragma unroll
            for (int c = 0; c < Cin; ++c)
#pragma unroll
                for (int dh = 0; dh < KH; ++dh) {
                    const int t = (c * KH + dh) * KW + dw;
                    const VT w4 = WREG ? wreg[t] : *(const VT*)(wl + t * p.Cout + cq * CPL);
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] += w4 * xin[(r * SH + dh) * Cin + c];
                }
        }
    } else {
        for (int dw = 0; dw < KW; ++dw)
            for (int c = 0; c < Cin; ++c)
                for (int dh = 0; dh < KH; ++dh) {
                    const VT w4 = *(const VT*)(wl + ((c * KH + dh) * KW + dw) * p.Cout + cq * CPL);
                    const float* col = xs + dw * cs + (ho0 * SH + dh) * Cin + c;
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] += w4 * col[r * SH * Cin];
                }
    }
}

// The loop every pass runs: columns q = blockIdx.x, + gridDim.x, ...; double-buffered staging; body(b, wo, ho0, acc) per lane and row group
// pre(b, wo, ho0): issued BEFORE the convolution of a row group (the global loads of the backward passes, whose latency then
// runs under the convolution's arithmetic); WREG: weights in registers (false: 16-byte LDS reads, for the kernel that needs the
// registers for its accumulators).
template <class W, bool WREG, typename VT, class Pre, class Body>
__device__ __forceinline__ void stem_columns(const StemConv& p, float* wl, float* xs2, VT (&wreg)[W::T], int cq, int rg, int nrg, Pre pre,
                                             Body body) {
    stage_weights(p, wl);
    for (int i = threadIdx.x; i < 2 * MAX_XS; i += 256) xs2[i] = 0.f;
    // Two column sets are in flight in registers (sa: the next column, sb: the one after): a set has the convolution of TWO columns
    // to arrive in (a load from HBM takes 1-2 us under load, a column's arithmetic 0.5-1 us).
    ColumnStage<W> sa, sb;
    sa.init(p);
    sb.npf = sa.npf;
#pragma unroll
    for (int k = 0; k < MAXPF; ++k) sb.code[k] = sa.code[k];
    const int ncol = p.gx.B * p.Wo, G = gridDim.x, Gb = G / p.Wo, Gw = G % p.Wo;
    ColIdx cur(blockIdx.x, p.Wo), nxt(blockIdx.x, p.Wo);          // the column being convolved / the newest one requested
    if (cur.q < ncol) sa.fetch(p, cur.b, cur.wo);
    __syncthreads();
    if constexpr (W::FIXED && WREG) {
#pragma unroll
        for (int t = 0; t < W::T; ++t) wreg[t] = *(const VT*)(wl + t * p.Cout + cq * (int)(sizeof(VT) / 4));
    }
    if (cur.q < ncol) sa.commit(xs2);
    nxt.step(G, Gb, Gw, p.Wo);
    if (nxt.q < ncol) sa.fetch(p, nxt.b, nxt.wo);
    __syncthreads();
    auto column = [&](const ColIdx& c, const float* xs) {
        for (int ho0 = rg * R; ho0 < p.Ho; ho0 += nrg * R) {
            pre(c.b, c.wo, ho0);
            VT acc[R];
            conv_rows<W, WREG, VT>(p, wl, wreg, xs, cq, ho0, acc);
            body(c.b, c.wo, ho0, acc, xs);
        }
    };
    // invariant at the top: LDS buffer 0 holds column cur.q, sa holds (or is loading) column cur.q + G = nxt.q
    while (cur.q < ncol) {
        nxt.step(G, Gb, Gw, p.Wo);                                // cur + 2G
        if (nxt.q < ncol) sb.fetch(p, nxt.b, nxt.wo);
        column(cur, xs2);
        if (cur.q + G < ncol) sa.commit(xs2 + MAX_XS);
        __syncthreads();
        cur.step(G, Gb, Gw, p.Wo);
        if (cur.q >= ncol) break;
        nxt.step(G, Gb, Gw, p.Wo);                                // (old cur) + 3G
        if (nxt.q < ncol) sa.fetch(p, nxt.b, nxt.wo);
        column(cur, xs2 + MAX_XS);
        if (cur.q + G < ncol) sb.commit(xs2);
        __syncthreads();
        cur.step(G, Gb, Gw, p.Wo);
        // (buffer 0 now holds cur.q; sa holds cur.q + G = nxt.q: the invariant again)
    }
}

// block-wide sum of per-lane f32x4 values over the lanes that share a channel quad (fixed order), result in the lanes rg == 0
template <int NV>
__device__ __forceinline__ void reduce_quads(f32x4 (&v)[NV], float* red, int cqn, int cq, int rg, int nrg) {
    const int C = cqn * 4;          // red: [nrg][NV][C] floats
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

#define STEM_SHARED                                                    \
    __shared__ __attribute__((aligned(16))) float wl[MAX_WL];          \
    __shared__ __attribute__((aligned(16))) float xs2[2 * MAX_XS];     \
    const int cqn = p.Cout / 4, cq = threadIdx.x % cqn, rg = threadIdx.x / cqn, nrg = 256 / cqn; \
    f32x4 wreg[W::T];

// ---- pass 1: per-workgroup partial sums of y and y^2 (BatchNorm statistics), slabs[blk][2][Cout]
template <class W>
__global__ __launch_bounds__(256) void stem_stats_kernel(StemConv p, float* __restrict__ slabs) {
    STEM_SHARED
    __shared__ __attribute__((aligned(16))) float red[2048];
    f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    stem_columns<W, true, f32x4>(p, wl, xs2, wreg, cq, rg, nrg, [](int, int, int) {}, [&](int, int, int ho0, f32x4* acc, const float*) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (ho0 + r < p.Ho) {
                s[0] += acc[r];
                s[1] += acc[r] * acc[r];
            }
    });
    reduce_quads<2>(s, red, cqn, cq, rg, nrg);
    if (rg == 0) {
        *(f32x4*)(slabs + (long long)blockIdx.x * 2 * p.Cout + cq * 4) = s[0];
        *(f32x4*)(slabs + (long long)blockIdx.x * 2 * p.Cout + p.Cout + cq * 4) = s[1];
    }
}

// ---- pass 2: a = relu((y - mean) * rstd * gamma + beta) into the activation grid (storage dtype, its own row geometry)
template <class W, typename T>
__global__ __launch_bounds__(256) void stem_apply_kernel(StemConv p, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ out, Grid go) {
    STEM_SHARED
    const f32x4 mu = *(const f32x4*)(stats + cq * 4), rs = *(const f32x4*)(stats + p.Cout + cq * 4);
    const f32x4 ga = *(const f32x4*)(gamma + cq * 4), be = *(const f32x4*)(beta + cq * 4);
    const f32x4 k = rs * ga;
    stem_columns<W, true, f32x4>(p, wl, xs2, wreg, cq, rg, nrg, [](int, int, int) {}, [&](int b, int wo, int ho0, f32x4* acc, const float*) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (ho0 + r < p.Ho) {
                f32x4 o = (acc[r] - mu) * k + be;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = relu_f(o[e]);
                store4(out + (grid_off32(go, b, wo, ho0 + r) + cq * 4), o);
            }
    });
}

// ---- backward pass 1: partial sums of g * xhat and g, g = da * (a > 0);  slabs[blk][2][Cout]
template <class W, typename T>
__global__ __launch_bounds__(256) void stem_bwd_reduce_kernel(StemConv p, const float* __restrict__ stats, const T* __restrict__ da,
                                                              const T* __restrict__ a, Grid ga_, float* __restrict__ slabs) {
    STEM_SHARED
    __shared__ __attribute__((aligned(16))) float red[2048];
    const f32x4 mu = *(const f32x4*)(stats + cq * 4), rs = *(const f32x4*)(stats + p.Cout + cq * 4);
    f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    Raw4<T> gr[R], ar[R];
    auto pre = [&](int b, int wo, int ho0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned o = grid_off32(ga_, b, wo, min(ho0 + r, p.Ho - 1)) + cq * 4;
            gr[r].load(da + o);
            ar[r].load(a + o);
        }
    };
    stem_columns<W, true, f32x4>(p, wl, xs2, wreg, cq, rg, nrg, pre, [&](int, int, int ho0, f32x4* acc, const float*) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (ho0 + r < p.Ho) {
                const f32x4 g4 = gr[r].get(), a4 = ar[r].get();
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = a4[e] > 0.f ? g4[e] : 0.f;
                    s[0][e] += g * (acc[r][e] - mu[e]) * rs[e];
                    s[1][e] += g;
                }
            }
    });
    reduce_quads<2>(s, red, cqn, cq, rg, nrg);
    if (rg == 0) {
        *(f32x4*)(slabs + (long long)blockIdx.x * 2 * p.Cout + cq * 4) = s[0];
        *(f32x4*)(slabs + (long long)blockIdx.x * 2 * p.Cout + p.Cout + cq * 4) = s[1];
    }
}

// ---- backward pass 2: dy = gamma rstd (g - dbeta / n - xhat dgamma / n) formed per position and contracted with the input window at
// once: slabs[blk][co][t] partial sums of dy[co] * x[tap t]  (the convolution's weight gradient in the reference's layout).
// The Cin * kh * kw accumulators per channel live in registers: compile-time window shapes only.
template <class W, typename T>
__global__ __launch_bounds__(256) void stem_bwd_wgrad_kernel(StemConv p, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                             float inv_count, const T* __restrict__ da, const T* __restrict__ a, Grid ga_,
                                                             float* __restrict__ slabs) {
    static_assert(W::FIXED, "the weight-gradient kernel needs a compile-time window shape");
    constexpr int TAPS = W::T;
    // TWO channels per lane here (f32x2: one packed FMA per tap and row): with four, the 18 x 4 accumulators beside the window and the
    // convolution's own registers left one wave per SIMD and every LDS / global access exposed (0.51 ms at configs[2]).
    typedef f32x2 VT;
    __shared__ __attribute__((aligned(16))) float wl[MAX_WL];
    __shared__ __attribute__((aligned(16))) float xs2[2 * MAX_XS];
    __shared__ __attribute__((aligned(16))) float red[2048];          // one chunk of 4 taps at a time: [nrg][4][Cout] = 2 048 floats
    const int cqn = p.Cout / 2, cq = threadIdx.x % cqn, rg = threadIdx.x / cqn, nrg = 256 / cqn;
    VT wreg[W::T];
    const int Cin = W::cin(p), KH = W::kh(p), KW = W::kw(p), SH = W::sh(p), cs = col_stride<W>(p);
    const VT mu = *(const VT*)(stats + cq * 2), rs = *(const VT*)(stats + p.Cout + cq * 2);
    VT k1, k2, k3;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = cq * 2 + e;
        k1[e] = gamma[c] * rs[e];
        k2[e] = k1[e] * dbeta[c] * inv_count;
        k3[e] = k1[e] * rs[e] * dgamma[c] * inv_count;
    }
    VT dwa[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) dwa[t] = vzero<VT>();
    RawV<T, VT> gr[R], ar[R];
    auto pre = [&](int b, int wo, int ho0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned o = grid_off32(ga_, b, wo, min(ho0 + r, p.Ho - 1)) + cq * 2;
            gr[r].load(da + o);
            ar[r].load(a + o);
        }
    };
    stem_columns<W, false, VT>(p, wl, xs2, wreg, cq, rg, nrg, pre, [&](int, int, int ho0, VT* acc, const float* xs) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = ho0 + r < p.Ho;
            const VT g4 = gr[r].get(), a4 = ar[r].get();
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float g = a4[e] > 0.f ? g4[e] : 0.f;
                acc[r][e] = ok ? k1[e] * g - k2[e] - k3[e] * (acc[r][e] - mu[e]) : 0.f;      // dy
            }
        }
#pragma unroll
        for (int dw = 0; dw < KW; ++dw) {
            float xin[W::WROWS];
            load_window<W::WROWS>(xs + dw * cs + ho0 * SH * Cin, xin);
#pragma unroll
            for (int c = 0; c < Cin; ++c)
#pragma unroll
                for (int dh = 0; dh < KH; ++dh)
#pragma unroll
                    for (int r = 0; r < R; ++r) dwa[(c * KH + dh) * KW + dw] += acc[r] * xin[(r * SH + dh) * Cin + c];
        }
    });
    // reduction over the row groups, 4 taps at a time (fixed order)
    for (int t0 = 0; t0 < TAPS; t0 += 4) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
            if (t >= t0 && t < t0 + 4) *(VT*)(red + (rg * 4 + (t - t0)) * p.Cout + cq * 2) = dwa[t];
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
// W_r [Cout][Cin] the 1x1 projection (no bias, no padding).  A thread owns CPT consecutive channels (16 bytes of the storage type)
// of one position and keeps their Cin weights in registers while it walks the grid (the walk's stride is a multiple of the channel
// groups per position).
template <typename T> struct Cpt { static constexpr int N = 16 / sizeof(T); };

template <typename T, int N>
__device__ __forceinline__ void loadn(const T* src, float (&v)[N]) {
    if constexpr (sizeof(T) == 2) {
        const uint4 u = *(const uint4*)src;
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, w[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xffff0000u);
        }
    } else {
        const f32x4 f = *(const f32x4*)src;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f[e];
    }
}
template <typename T, int N>
__device__ __forceinline__ void storen(T* dst, const float (&v)[N]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
        *(bf16x8*)dst = o;
    } else {
        *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
    }
}

// BN (bf16): main_ is the INPUT of the block's second BatchNorm (grid gm); the value added to the projection is relu(BatchNorm(main_)) rounded to
// bf16 — what cpc_bn_apply would have stored — and only its sign bits are kept (bits, addressed like the activation grid ga; may be null):
// cpc_bn_apply + this kernel's plain form in one pass, same results bit for bit.
template <typename T, int CIN, bool BN = false>
__global__ __launch_bounds__(256) void stem_residual_add_kernel(const T* __restrict__ main_, Grid gm, const float* __restrict__ xp, Grid gp,
                                                                const float* __restrict__ wr, T* __restrict__ out, Grid go, int oh, int ow,
                                                                int relu, const float* __restrict__ stats = nullptr,
                                                                const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr,
                                                                unsigned char* __restrict__ bits = nullptr, Grid ga = Grid(),
                                                                unsigned char* __restrict__ obits = nullptr) {
    constexpr int N = Cpt<T>::N;
    const int cgn = gm.C / N;
    const unsigned total = (unsigned)((long long)gm.B * gm.W * gm.H * cgn);
    const unsigned first = blockIdx.x * 256u + threadIdx.x;
    const int cg = (int)(first % (unsigned)cgn);          // the launcher makes gridDim.x * 256 a multiple of cgn
    float w[N][CIN];
#pragma unroll
    for (int e = 0; e < N; ++e)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) w[e][ci] = wr[(cg * N + e) * CIN + ci];
    for (unsigned idx = first; idx < total; idx += gridDim.x * 256u) {
        const unsigned pos = idx / (unsigned)cgn;
        const int h = (int)(pos % (unsigned)gm.H);
        const unsigned col = pos / (unsigned)gm.H;
        const int wq = (int)(col % (unsigned)gm.W), b = (int)(col / (unsigned)gm.W);
        float v[N];
        loadn<T, N>(main_ + grid_off(gm, b, wq, h) + cg * N, v);
        if constexpr (BN) {
            unsigned m = 0u;
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const int c = cg * N + e;
                v[e] = (float)(bf16_t)relu_f((v[e] - stats[c]) * stats[gm.C + c] * gamma[c] + beta[c]);
                m |= (v[e] > 0.f ? 1u : 0u) << e;
            }
            if (bits) bits[(grid_off(ga, b, wq, h) + cg * N) >> 3] = (unsigned char)m;
        }
        const float* xr = xp + grid_off(gp, b, wq + ow, h + oh);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float xv = xr[ci];
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] += w[e][ci] * xv;
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = relu_f(v[e]);
        }
        const long long oo = grid_off(go, b, wq, h) + cg * N;
        storen<T, N>(out + oo, v);
        if constexpr (BN) {
            if (obits) {          // sign bits of the stored block output (for the backward passes that read them instead of out)
                unsigned mo = 0u;
#pragma unroll
                for (int e = 0; e < N; ++e) mo |= ((float)(bf16_t)v[e] > 0.f ? 1u : 0u) << e;
                obits[oo >> 3] = (unsigned char)mo;
            }
        }
    }
}

// backward: g = dout * (out > 0 if relu);  dmain = g;  slabs[blk][co][ci] partial sums of g[co] * xp[ci]  (gradient of W_r)
template <typename T, int CIN>
__global__ __launch_bounds__(256) void stem_residual_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ out, Grid go,
                                                                T* __restrict__ dmain, Grid gm, const float* __restrict__ xp, Grid gp,
                                                                float* __restrict__ slabs, int oh, int ow, int relu, long long cols_per_block,
                                                                const unsigned char* __restrict__ obits = nullptr) {
    // obits (bf16, N = 8): the ReLU mask as the sign bits of out (addressed like out) instead of out itself; dmain may then be null — the
    // masked gradient is not stored (cpc_bn_bwd_*_res read dout and the same bits) and the pass only gathers the projection's gradient
    constexpr int N = Cpt<T>::N;
    __shared__ __attribute__((aligned(16))) float red[8192];          // [nrp][CIN][C]: 256 / (C/N) * CIN * C <= 8192
    const int C = gm.C, cgn = C / N;
    const int cg = threadIdx.x % cgn, rp = threadIdx.x / cgn, nrp = 256 / cgn;
    const long long ncol = (long long)gm.B * gm.W;
    const long long q0 = (long long)blockIdx.x * cols_per_block, q1 = min(ncol, q0 + cols_per_block);
    float s[CIN][N];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int e = 0; e < N; ++e) s[ci][e] = 0.f;
    if (rp < nrp && q0 < q1) {
        const unsigned rend = (unsigned)(q1 * gm.H);
        for (unsigned r = (unsigned)(q0 * gm.H) + rp; r < rend; r += nrp) {
            const unsigned q = r / (unsigned)gm.H;
            const int h = (int)(r - q * gm.H), w = (int)(q % gm.W), b = (int)(q / gm.W);
            const long long oo = grid_off(go, b, w, h) + cg * N;
            float g[N];
            loadn<T, N>(dout + oo, g);
            if (relu) {
                if (obits) {
                    const unsigned mo = obits[oo >> 3];
#pragma unroll
                    for (int e = 0; e < N; ++e) g[e] = ((mo >> e) & 1u) ? g[e] : 0.f;
                } else {
                    float y[N];
                    loadn<T, N>(out + oo, y);
#pragma unroll
                    for (int e = 0; e < N; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
                }
            }
            if (dmain) storen<T, N>(dmain + grid_off(gm, b, w, h) + cg * N, g);
            const float* xr = xp + grid_off(gp, b, w + ow, h + oh);
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float xv = xr[ci];
#pragma unroll
                for (int e = 0; e < N; ++e) s[ci][e] += g[e] * xv;
            }
        }
    }
    if (rp < nrp) {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int e = 0; e < N; ++e) red[(rp * CIN + ci) * C + cg * N + e] = s[ci][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CIN * C; i += 256) {
        const int ci = i / C, co = i - ci * C;
        float acc = 0.f;
        for (int qg = 0; qg < nrp; ++qg) acc += red[(qg * CIN + ci) * C + co];
        slabs[(long long)blockIdx.x * C * CIN + co * CIN + ci] = acc;
    }
}

bool grid_ok(const int* g) {
    return g && g[0] > 0 && g[1] > 0 && g[2] > 0 && g[4] >= 0 && g[3] >= g[4] + g[2] && g[5] > 0 &&
           (long long)g[0] * g[1] * g[2] * g[5] < (1ll << 31);
}
Grid mk(const int* g) { return Grid{g[0], g[1], g[2], g[3], g[4], g[5]}; }

// window shapes (Cin, kh, kw, stride in h) with kernels
#define STEM_SHAPES(X) X(2, 3, 3, 2) X(1, 3, 3, 2) X(2, 3, 3, 1) X(1, 3, 3, 1) X(1, 5, 1, 1) X(2, 5, 1, 1) X(1, 2, 2, 1) X(2, 2, 2, 1)

bool shape_ok(int cin, int kh, int kw, int sh) {
#define X(CI, KH, KW, SH) if (cin == CI && kh == KH && kw == KW && sh == SH) return true;
    STEM_SHAPES(X)
#undef X
    return false;
}

bool stem_ok(const int* gx, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo) {
    if (!grid_ok(gx) || gx[4] != 0) return false;
    const int Cin = gx[5];
    if (!shape_ok(Cin, kh, kw, sh)) return false;
    if (Cout % 4 || Cout < 4 || Cout > 64 || 256 % (Cout / 4)) return false;
    if (sw < 1 || ph < 0 || pw < 0) return false;
    if (Cin * kh * kw * Cout > MAX_WL || kw * (((gx[2] + 2 * ph + R * sh) * Cin + 3) & ~3) > MAX_XS || kw * (gx[2] + 2 * ph) * Cin > MAXPF * 256) return false;
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

int launch_stem_supported(int cin, int cout, int kh, int kw, int sh, int hin, int ph) {
    if (cout % 4 || cout < 4 || cout > 64 || 256 % (cout / 4) || !shape_ok(cin, kh, kw, sh)) return 0;
    if (cin * kh * kw * cout > MAX_WL || kw * (((hin + 2 * ph + R * sh) * cin + 3) & ~3) > MAX_XS || kw * (hin + 2 * ph) * cin > MAXPF * 256) return 0;
    return 1;
}

int launch_stem_stats(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw, int ph,
                      int pw, int Ho, int Wo, float* slabs, int nblocks, hipStream_t st) {
    if (!stem_ok(gx, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo) || nblocks <= 0) return CPC_EINVAL;
    const StemConv p = mkconv(x, gx, w, bias, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo);
    const int cin = gx[5];
#define X(CI, KH, KW, SH)                                                                                                  \
    if (cin == CI && kh == KH && kw == KW && sh == SH)                                                                     \
        hipLaunchKernelGGL((stem_stats_kernel<Win<CI, KH, KW, SH>>), dim3(nblocks), dim3(256), 0, st, p, slabs);
    STEM_SHAPES(X)
#undef X
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

template <typename T>
static void stem_apply_t(const StemConv& p, int cin, const float* stats, const float* gamma, const float* beta, T* out, const Grid& go,
                         int nblocks, hipStream_t st) {
#define X(CI, KH, KW, SH)                                                                                                  \
    if (cin == CI && p.kh == KH && p.kw == KW && p.sh == SH)                                                               \
        hipLaunchKernelGGL((stem_apply_kernel<Win<CI, KH, KW, SH>, T>), dim3(nblocks), dim3(256), 0, st, p, stats, gamma, beta, out, go);
    STEM_SHAPES(X)
#undef X
}

int launch_stem_apply(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw, int ph,
                      int pw, int Ho, int Wo, const float* stats, const float* gamma, const float* beta, void* out, const int* go,
                      int nblocks, int dtype, hipStream_t st) {
    if (!stem_ok(gx, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo) || !grid_ok(go) || nblocks <= 0) return CPC_EINVAL;
    if (go[0] != gx[0] || go[1] != Wo || go[2] != Ho || go[5] != Cout) return CPC_EINVAL;
    const StemConv p = mkconv(x, gx, w, bias, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo);
    if (dtype == CPC_DTYPE_BF16)
        stem_apply_t<bf16_t>(p, gx[5], stats, gamma, beta, (bf16_t*)out, mk(go), nblocks, st);
    else if (dtype == CPC_DTYPE_F32)
        stem_apply_t<float>(p, gx[5], stats, gamma, beta, (float*)out, mk(go), nblocks, st);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

template <typename T>
static void stem_bwd_reduce_t(const StemConv& p, int cin, const float* stats, const T* da, const T* a, const Grid& ga, float* slabs,
                              int nblocks, hipStream_t st) {
#define X(CI, KH, KW, SH)                                                                                                  \
    if (cin == CI && p.kh == KH && p.kw == KW && p.sh == SH)                                                               \
        hipLaunchKernelGGL((stem_bwd_reduce_kernel<Win<CI, KH, KW, SH>, T>), dim3(nblocks), dim3(256), 0, st, p, stats, da, a, ga, slabs);
    STEM_SHAPES(X)
#undef X
}

int launch_stem_bwd_reduce(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw,
                           int ph, int pw, int Ho, int Wo, const float* stats, const void* da, const void* a, const int* ga, float* slabs,
                           int nblocks, int dtype, hipStream_t st) {
    if (!stem_ok(gx, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo) || !grid_ok(ga) || nblocks <= 0) return CPC_EINVAL;
    if (ga[0] != gx[0] || ga[1] != Wo || ga[2] != Ho || ga[5] != Cout) return CPC_EINVAL;
    const StemConv p = mkconv(x, gx, w, bias, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo);
    if (dtype == CPC_DTYPE_BF16)
        stem_bwd_reduce_t<bf16_t>(p, gx[5], stats, (const bf16_t*)da, (const bf16_t*)a, mk(ga), slabs, nblocks, st);
    else if (dtype == CPC_DTYPE_F32)
        stem_bwd_reduce_t<float>(p, gx[5], stats, (const float*)da, (const float*)a, mk(ga), slabs, nblocks, st);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

template <typename T>
static void stem_bwd_wgrad_t(const StemConv& p, int cin, const float* stats, const float* gamma, const float* dgamma, const float* dbeta,
                             float inv, const T* da, const T* a, const Grid& ga, float* slabs, int nblocks, hipStream_t st) {
#define X(CI, KH, KW, SH)                                                                                                  \
    if (cin == CI && p.kh == KH && p.kw == KW && p.sh == SH)                                                               \
        hipLaunchKernelGGL((stem_bwd_wgrad_kernel<Win<CI, KH, KW, SH>, T>), dim3(nblocks), dim3(256), 0, st, p, stats, gamma, dgamma, dbeta, \
                           inv, da, a, ga, slabs);
    STEM_SHAPES(X)
#undef X
}

int launch_stem_bwd_wgrad(const float* x, const int* gx, const float* w, const float* bias, int Cout, int kh, int kw, int sh, int sw,
                          int ph, int pw, int Ho, int Wo, const float* stats, const float* gamma, const float* dgamma, const float* dbeta,
                          double count, const void* da, const void* a, const int* ga, float* slabs, int nblocks, int dtype, hipStream_t st) {
    if (!stem_ok(gx, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo) || !grid_ok(ga) || nblocks <= 0 || count <= 0) return CPC_EINVAL;
    if (ga[0] != gx[0] || ga[1] != Wo || ga[2] != Ho || ga[5] != Cout) return CPC_EINVAL;
    const StemConv p = mkconv(x, gx, w, bias, Cout, kh, kw, sh, sw, ph, pw, Ho, Wo);
    const float inv = (float)(1.0 / count);
    if (dtype == CPC_DTYPE_BF16)
        stem_bwd_wgrad_t<bf16_t>(p, gx[5], stats, gamma, dgamma, dbeta, inv, (const bf16_t*)da, (const bf16_t*)a, mk(ga), slabs, nblocks, st);
    else if (dtype == CPC_DTYPE_F32)
        stem_bwd_wgrad_t<float>(p, gx[5], stats, gamma, dgamma, dbeta, inv, (const float*)da, (const float*)a, mk(ga), slabs, nblocks, st);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

static bool residual_ok(const int* gm, const int* gp, const int* go, int oh, int ow, int dtype) {
    if (!grid_ok(gm) || !grid_ok(gp) || !grid_ok(go)) return false;
    if (gm[0] != go[0] || gm[1] != go[1] || gm[2] != go[2] || gm[5] != go[5] || gp[0] != gm[0]) return false;
    const int n = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if (gm[5] % n || gm[5] / n > 256 || 256 % (gm[5] / n) || gp[5] < 1 || gp[5] > 2 || oh < 0 || ow < 0) return false;
    return gm[1] + ow <= gp[1] && gm[2] + oh <= gp[2];
}

int launch_stem_residual_bn_add(const void* y, const int* gy, const float* xp, const int* gp, const float* wr, void* out, const int* go, int oh,
                                int ow, int relu, const float* stats, const float* gamma, const float* beta, unsigned char* bits, const int* ga,
                                int dtype, hipStream_t st, unsigned char* obits) {
    if (dtype != CPC_DTYPE_BF16 || !residual_ok(gy, gp, go, oh, ow, dtype) || !stats || !gamma || !beta) return CPC_EINVAL;
    if (bits && (!grid_ok(ga) || ga[0] != gy[0] || ga[1] != gy[1] || ga[2] != gy[2] || ga[5] != gy[5])) return CPC_EINVAL;
    const int cgn = gy[5] / 8;
    const long long total = (long long)gy[0] * gy[1] * gy[2] * cgn;
    const int nb = (int)std::min<long long>(8192, (total + 255) / 256);
#define RES_BN_ADD(CI) hipLaunchKernelGGL((stem_residual_add_kernel<bf16_t, CI, true>), dim3(nb), dim3(256), 0, st, (const bf16_t*)y, mk(gy), xp, mk(gp), wr, (bf16_t*)out, mk(go), oh, ow, relu, stats, gamma, beta, bits, bits ? mk(ga) : mk(gy), obits)
    if (gp[5] == 1) RES_BN_ADD(1); else RES_BN_ADD(2);
#undef RES_BN_ADD
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_stem_residual_add(const void* main_, const int* gm, const float* xp, const int* gp, const float* wr, void* out, const int* go,
                             int oh, int ow, int relu, int dtype, hipStream_t st) {
    if (!residual_ok(gm, gp, go, oh, ow, dtype)) return CPC_EINVAL;
    const int n = dtype == CPC_DTYPE_BF16 ? 8 : 4, cgn = gm[5] / n;
    const long long total = (long long)gm[0] * gm[1] * gm[2] * cgn;
    // the walk's stride (gridDim.x * 256) is a multiple of cgn (cgn divides 256), so a thread keeps its channel group
    const int nb = (int)std::min<long long>(8192, (total + 255) / 256);
#define RES_ADD(T, CI) hipLaunchKernelGGL((stem_residual_add_kernel<T, CI>), dim3(nb), dim3(256), 0, st, (const T*)main_, mk(gm), xp, mk(gp), wr, (T*)out, mk(go), oh, ow, relu)
    if (dtype == CPC_DTYPE_BF16) { if (gp[5] == 1) RES_ADD(bf16_t, 1); else RES_ADD(bf16_t, 2); }
    else if (dtype == CPC_DTYPE_F32) { if (gp[5] == 1) RES_ADD(float, 1); else RES_ADD(float, 2); }
    else return CPC_EINVAL;
#undef RES_ADD
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_stem_residual_bwd(const void* dout, const void* out, const int* go, void* dmain, const int* gm, const float* xp, const int* gp,
                             float* slabs, int oh, int ow, int relu, int nblocks, int dtype, hipStream_t st, const unsigned char* obits) {
    if (!residual_ok(gm, gp, go, oh, ow, dtype) || nblocks <= 0) return CPC_EINVAL;
    if ((obits && dtype != CPC_DTYPE_BF16) || (!obits && (!out || !dmain))) return CPC_EINVAL;
    const int n = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if (256 / (gm[5] / n) * gp[5] * gm[5] > 8192) return CPC_EINVAL;
    const long long ncol = (long long)gm[0] * gm[1];
    const long long cpb = (ncol + nblocks - 1) / nblocks;
#define RES_BWD(T, CI) hipLaunchKernelGGL((stem_residual_bwd_kernel<T, CI>), dim3(nblocks), dim3(256), 0, st, (const T*)dout, (const T*)out, mk(go), (T*)dmain, mk(gm), xp, mk(gp), slabs, oh, ow, relu, cpb, obits)
    if (dtype == CPC_DTYPE_BF16) { if (gp[5] == 1) RES_BWD(bf16_t, 1); else RES_BWD(bf16_t, 2); }
    else if (dtype == CPC_DTYPE_F32) { if (gp[5] == 1) RES_BWD(float, 1); else RES_BWD(float, 2); }
    else return CPC_EINVAL;
#undef RES_BWD
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
