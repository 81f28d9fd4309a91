// Kernels of the attention context network (AttentionModel, attention_model.py:38-82 with the vendored post-norm
// TransformerEncoderLayer, transformer.py:223-272) that are not GEMMs: positional encoding, causal multi-head attention
// for short sequences (<= 64 steps: the whole (item, head) problem lives in one workgroup's LDS), residual + LayerNorm,
// mean over time.  All work on channels-last rows x[(b, t)][C]; arithmetic in f32, storage type T.
#include <cstdlib>
#include "cpc_common.h"
#include "cpc_kernels.h"

namespace {

constexpr int ATT_S = 64;       // max sequence length handled by the attention kernels
constexpr int ATT_D = 64;       // max head dimension

// Dropout masks are a pure function of (seed, site, element index), so the backward pass regenerates them instead of
// storing them: keep  <=>  hash >= thresh, thresh = p * 2^32.  (The reference draws its masks from torch's generator
// stream, which no other device or launch geometry can reproduce; only the distribution is shared.)
struct Drop {
    unsigned long long seed;
    unsigned site, thresh;      // thresh == 0: no dropout
    float inv_keep;             // 1 / (1 - p)
};
__device__ __forceinline__ unsigned drop_hash(unsigned long long seed, unsigned site, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)site + 1ull) + idx * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32);
}
__device__ __forceinline__ float drop_factor(const Drop& d, unsigned long long idx) {
    if (d.thresh == 0u) return 1.f;
    return drop_hash(d.seed, d.site, idx) >= d.thresh ? d.inv_keep : 0.f;
}

// x0[(b,t)][c] = z[b][t0+t][c] * scale + pe[t][c]      (PositionalEncoder.forward, attention_model.py:28-35)
template <typename T>
__global__ __launch_bounds__(256) void pe_scale_fwd_kernel(const T* __restrict__ top, const float* __restrict__ pe, T* __restrict__ x0,
                                                           int B, int S, int C, long long item_stride, float scale) {
    const int c4n = C / 4;
    const long long total = (long long)B * S * c4n;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c4 = (int)(idx % c4n);
        const int t = (int)((idx / c4n) % S);
        const int b = (int)(idx / ((long long)c4n * S));
        const f32x4 v = load4(top + (long long)b * item_stride + (long long)t * C + c4 * 4);
        const f32x4 p = *(const f32x4*)(pe + (long long)t * C + c4 * 4);
        store4(x0 + ((long long)b * S + t) * C + c4 * 4, v * scale + p);
    }
}

// dz[b][t0+t][c] = scale * (g1 + g2)[(b,t)][c]
template <typename T>
__global__ __launch_bounds__(256) void pe_scale_bwd_kernel(const T* __restrict__ g1, const T* __restrict__ g2, T* __restrict__ dtop,
                                                           int B, int S, int C, long long item_stride, float scale) {
    const int c4n = C / 4;
    const long long total = (long long)B * S * c4n;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c4 = (int)(idx % c4n);
        const int t = (int)((idx / c4n) % S);
        const int b = (int)(idx / ((long long)c4n * S));
        const long long o = ((long long)b * S + t) * C + c4 * 4;
        f32x4 v = load4(g1 + o);
        if (g2) v += load4(g2 + o);
        store4(dtop + (long long)b * item_stride + (long long)t * C + c4 * 4, v * scale);
    }
}

// Causal multi-head self-attention for one (item, head) per workgroup.
//   qkv [(b,t)][3C]: q | k | v column blocks, head h at columns h*d of each;   out [(b,t)][C];   P [(b*heads+h)][S][S]
// Every global access and every LDS operand read moves four consecutive elements (d % 4 == 0; the first version moved single
// 2-byte elements and spent most of its time in the memory instructions, not in the 2 x S x S x d multiply-adds).
constexpr int ATT_LD = ATT_D + 4;      // LDS row stride of the operand tiles: rows stay 8-byte aligned in bf16

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, T* __restrict__ P, int S,
                                                       int C, int heads, float scale, Drop dr) {
    // operand tiles in the storage type (bf16: half the LDS, so three workgroups share a CU), scores in f32
    __shared__ __attribute__((aligned(16))) T q[ATT_S][ATT_LD], k[ATT_S][ATT_LD], v[ATT_S][ATT_LD];
    __shared__ float p[ATT_S][ATT_S + 1];
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads, d = C / heads, tid = threadIdx.x;
    const int d4 = d / 4;
    for (int idx = tid; idx < S * d4; idx += 256) {
        const int t = idx / d4, c = (idx % d4) * 4;
        const T* row = qkv + ((long long)b * S + t) * 3 * C + h * d + c;
        store4(&q[t][c], load4(row));
        store4(&k[t][c], load4(row + C));
        store4(&v[t][c], load4(row + 2 * C));
    }
    __syncthreads();
    for (int idx = tid; idx < S * S; idx += 256) {
        const int i = idx / S, j = idx % S;
        float s = -INFINITY;
        if (j <= i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < d; c += 4) acc += load4(&q[i][c]) * load4(&k[j][c]);
            s = (acc[0] + acc[1] + acc[2] + acc[3]) * scale;
        }
        p[i][j] = s;
    }
    __syncthreads();
    {
        // row softmax: four neighbouring lanes share a row (columns part, part + 4, ...) and combine by lane exchange — with
        // one thread per row a single wave walked 3 x S dependent LDS reads while the other three waited at the barrier
        const int i = tid >> 2, part = tid & 3;          // 256 threads = ATT_S rows x 4
        if (i < S) {
            float mx = -INFINITY;
            for (int j = part; j <= i; j += 4) mx = fmaxf(mx, p[i][j]);
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            float sum = 0.f;
            for (int j = part; j <= i; j += 4) { const float e = expf(p[i][j] - mx); p[i][j] = e; sum += e; }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            const float inv = 1.f / sum;
            for (int j = part; j < S; j += 4) p[i][j] = j <= i ? p[i][j] * inv : 0.f;
        }
    }
    __syncthreads();
    if (S % 4 == 0) {
        for (int idx = tid; idx < S * S / 4; idx += 256) {
            const int i = (idx * 4) / S, j = (idx * 4) % S;
            store4(P + (long long)bh * S * S + idx * 4, (f32x4){p[i][j], p[i][j + 1], p[i][j + 2], p[i][j + 3]});
        }
    } else {
        for (int idx = tid; idx < S * S; idx += 256) P[(long long)bh * S * S + idx] = from_f32<T>(p[idx / S][idx % S]);
    }
    if (dr.thresh) {        // nn.MultiheadAttention's dropout on the attention weights (saved P stays undropped)
        __syncthreads();
        for (int idx = tid; idx < S * S; idx += 256) p[idx / S][idx % S] *= drop_factor(dr, (unsigned long long)bh * S * S + idx);
        __syncthreads();
    }
    for (int idx = tid; idx < S * d4; idx += 256) {
        const int i = idx / d4, c = (idx % d4) * 4;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j <= i; ++j) o += p[i][j] * load4(&v[j][c]);
        store4(out + ((long long)b * S + i) * C + h * d + c, o);
    }
}

// ---- The same forward on the matrix pipe (bf16 storage, head size 64, S <= 64): both products are 64 x 64 x 64, one 16-row band per
// wave.  scores^T fragments come out as (4 consecutive columns j, row i = lane % 16) per lane, so a row's softmax needs two lane
// exchanges; P goes to LDS as bf16 (the A operand of the second product, and what is saved for the backward pass), V is read
// transposed (ds_read_b64_tr_b16).  Operand tiles with 144-byte rows: every 16-byte chunk aligned, the 16 rows of a fragment read
// on different banks.  Rows >= S are zero.
__device__ __forceinline__ uint4 attn_frag_tr(const unsigned char* tile, int rowb, int cb, int ks, int lane) {
    // (the TN GEMM's fragment read: reduction index along the tile's ROWS, 16 columns cb .. cb + 15)
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const unsigned char* a1 = tile + (ks * 32 + 4 * g + q) * rowb + (cb + 4 * pp) * 2;
    const unsigned char* a2 = a1 + 16 * rowb;
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a2);
    uint2 lo = __builtin_bit_cast(uint2, v1), hi = __builtin_bit_cast(uint2, v2);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            bf16_t* __restrict__ P, int S, int C, int heads, float scale, Drop dr) {
    constexpr int LD = 72, D = 64;
    __shared__ __attribute__((aligned(16))) bf16_t q[ATT_S][LD], k[ATT_S][LD], v[ATT_S][LD], pt[ATT_S][LD];
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < ATT_S * 8; idx += 256) {
        const int t = idx >> 3, ch = (idx & 7) * 8;
        uint4 zq = make_uint4(0, 0, 0, 0), zk = zq, zv = zq;
        if (t < S) {
            const bf16_t* row = qkv + ((long long)b * S + t) * 3 * C + h * D + ch;
            zq = *(const uint4*)row; zk = *(const uint4*)(row + C); zv = *(const uint4*)(row + 2 * C);
        }
        *(uint4*)&q[t][ch] = zq; *(uint4*)&k[t][ch] = zk; *(uint4*)&v[t][ch] = zv;
    }
    __syncthreads();
    const int g = lane >> 4, il = lane & 15, i = wave * 16 + il;
    // scores: acc[ct][r] = s[i][j = 16 ct + 4 g + r]
    f32x4 sc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        sc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 fq = *(const uint4*)&q[i][32 * ks + 8 * g];
            const uint4 fk = *(const uint4*)&k[16 * ct + il][32 * ks + 8 * g];
            mfma_chunk<bf16_t>(sc[ct], fk, fq);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * ct + 4 * g + r;
            sc[ct][r] = (j <= i && j < S) ? sc[ct][r] * scale : -INFINITY;
            mx = fmaxf(mx, sc[ct][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = sc[ct][r] == -INFINITY ? 0.f : expf(sc[ct][r] - mx);
            sc[ct][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;                 // (column 0 is never masked for a row < S; rows >= S are not stored)
    const bool rowok = i < S;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int j0 = 16 * ct + 4 * g;
        bf16x4 pv, pd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pr = rowok ? sc[ct][r] * inv : 0.f;
            pv[r] = (bf16_t)pr;
            pd[r] = (bf16_t)(pr * drop_factor(dr, (unsigned long long)bh * S * S + (unsigned long long)i * S + j0 + r));
        }
        if (rowok && j0 < S) {                   // the saved P stays undropped
            bf16_t* dst = P + (long long)bh * S * S + (long long)i * S + j0;
            if (S % 4 == 0) *(bf16x4*)dst = pv;
            else
                for (int r = 0; r < 4 && j0 + r < S; ++r) dst[r] = pv[r];
        }
        *(bf16x4*)&pt[i][j0] = pd;
    }
    __builtin_amdgcn_wave_barrier();             // a wave reads back only the 16 rows it wrote itself
    // out[i][c] = sum_j pd[i][j] v[j][c]: acc[cb][r] = out[i][16 cb + 4 g + r]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // the transposed read hands lane group g the reduction indices 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3 of the k-step:
            // the P fragment takes the same ones
            const uint2 plo = *(const uint2*)&pt[i][32 * ks + 4 * g], phi = *(const uint2*)&pt[i][32 * ks + 16 + 4 * g];
            const uint4 fp = make_uint4(plo.x, plo.y, phi.x, phi.y);
            const uint4 fv = attn_frag_tr((const unsigned char*)&v[0][0], LD * 2, 16 * cb, ks, lane);
            mfma_chunk<bf16_t>(o, fv, fp);
        }
        if (rowok) store4(out + ((long long)b * S + i) * C + h * D + 16 * cb + 4 * g, o);
    }
}

// Backward: dqkv [(b,t)][3C] from dout [(b,t)][C], the saved P and qkv.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ P, const T* __restrict__ dout,
                                                       T* __restrict__ dqkv, int S, int C, int heads, float scale, Drop dr) {
    __shared__ __attribute__((aligned(16))) T q[ATT_S][ATT_LD], k[ATT_S][ATT_LD], v[ATT_S][ATT_LD], go[ATT_S][ATT_LD];
    __shared__ float p[ATT_S][ATT_S + 1], ds[ATT_S][ATT_S + 1];
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads, d = C / heads, tid = threadIdx.x;
    const int d4 = d / 4;
    for (int idx = tid; idx < S * d4; idx += 256) {
        const int t = idx / d4, c = (idx % d4) * 4;
        const T* row = qkv + ((long long)b * S + t) * 3 * C + h * d + c;
        store4(&q[t][c], load4(row));
        store4(&k[t][c], load4(row + C));
        store4(&v[t][c], load4(row + 2 * C));
        store4(&go[t][c], load4(dout + ((long long)b * S + t) * C + h * d + c));
    }
    if (S % 4 == 0) {
        for (int idx = tid; idx < S * S / 4; idx += 256) {
            const f32x4 pv = load4(P + (long long)bh * S * S + idx * 4);
            const int i = (idx * 4) / S, j = (idx * 4) % S;
#pragma unroll
            for (int e = 0; e < 4; ++e) p[i][j + e] = pv[e];
        }
    } else {
        for (int idx = tid; idx < S * S; idx += 256) p[idx / S][idx % S] = to_f32(P[(long long)bh * S * S + idx]);
    }
    __syncthreads();
    // dP[i][j] = m[i][j] * sum_c dO[i][c] V[j][c]   (m = dropout factor of the attention weights, 1 without dropout)
    // The kernel is bound by LDS reads, so a thread takes a 2 x 2 block of (i, j) pairs: four operand reads per sixteen multiply-adds
    // instead of eight.  (Rows >= S of the tiles are never used for a stored value.)
    {
        const int hb = (S + 1) / 2;
        for (int idx = tid; idx < hb * hb; idx += 256) {
            const int i0 = (idx / hb) * 2, j0 = (idx % hb) * 2;
            if (j0 > i0 + 1) {                       // block above the diagonal: zeros
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2)
                        if (i0 + a < S && j0 + b2 < S) ds[i0 + a][j0 + b2] = 0.f;
                continue;
            }
            f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a01 = a00, a10 = a00, a11 = a00;
            for (int c = 0; c < d; c += 4) {
                const f32x4 g0 = load4(&go[i0][c]), g1 = load4(&go[i0 + 1][c]);
                const f32x4 v0 = load4(&v[j0][c]), v1 = load4(&v[j0 + 1][c]);
                a00 += g0 * v0; a01 += g0 * v1; a10 += g1 * v0; a11 += g1 * v1;
            }
            const float r[2][2] = {{a00[0] + a00[1] + a00[2] + a00[3], a01[0] + a01[1] + a01[2] + a01[3]},
                                   {a10[0] + a10[1] + a10[2] + a10[3], a11[0] + a11[1] + a11[2] + a11[3]}};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const int i = i0 + a, j = j0 + b2;
                    if (i < S && j < S)
                        ds[i][j] = j <= i ? r[a][b2] * drop_factor(dr, (unsigned long long)bh * S * S + (unsigned long long)i * S + j) : 0.f;
                }
        }
    }
    __syncthreads();
    {
        const int i = tid >> 2, part = tid & 3;          // four lanes per row, as in the forward softmax
        if (i < S) {
            float dot = 0.f;
            for (int j = part; j <= i; j += 4) dot = fmaf(ds[i][j], p[i][j], dot);
            dot += __shfl_xor(dot, 1, 64);
            dot += __shfl_xor(dot, 2, 64);
            for (int j = part; j < S; j += 4) ds[i][j] = j <= i ? p[i][j] * (ds[i][j] - dot) * scale : 0.f;    // d (raw q.k score)
        }
    }
    __syncthreads();
    // Four rows t0 .. t0+3 per thread and column group: one operand vector read serves four rows (five LDS reads per sixteen
    // multiply-adds instead of eight); per row the terms are added in the same order as before.
    const int tb = (S + 3) / 4;
    for (int idx = tid; idx < tb * d4; idx += 256) {
        const int t0 = (idx / d4) * 4, c = (idx % d4) * 4;
        f32x4 dq[4], dk[4], dv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { dq[r] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[r] = dq[r]; dv[r] = dq[r]; }
        const int jend = min(t0 + 3, S - 1);
        for (int j = 0; j <= jend; ++j) {
            const f32x4 kv = load4(&k[j][c]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j <= t0 + r && t0 + r < S) dq[r] += ds[t0 + r][j] * kv;
        }
        for (int i = t0; i < S; ++i) {
            const f32x4 qv = load4(&q[i][c]), gv = load4(&go[i][c]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + r;
                if (i >= t && t < S) {
                    dk[r] += ds[i][t] * qv;
                    dv[r] += (p[i][t] * drop_factor(dr, (unsigned long long)bh * S * S + (unsigned long long)i * S + t)) * gv;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (t0 + r >= S) continue;
            T* row = dqkv + ((long long)b * S + t0 + r) * 3 * C + h * d + c;
            store4(row, dq[r]);
            store4(row + C, dk[r]);
            store4(row + 2 * C, dv[r]);
        }
    }
}

// ---- The backward on the matrix pipe (bf16, head size 64, S <= 64).  dP = dO V^T comes out like the forward's scores (row i = the
// wave's band + lane % 16, four consecutive columns per lane and tile), so the softmax backward stays in registers; ds and P m go to LDS
// as bf16 tiles [i][j], and the three output products read them through the transposed fragment read: dq = ds K (ds as rows), dk =
// ds^T Q and dv = (P m)^T dO (reduction along the tiles' rows on both sides).  Rows >= S are zero.
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ P,
                                                            const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv, int S, int C,
                                                            int heads, float scale, Drop dr) {
    constexpr int LD = 72, D = 64;
    __shared__ __attribute__((aligned(16))) bf16_t q[ATT_S][LD], k[ATT_S][LD], v[ATT_S][LD], go[ATT_S][LD], dst[ATT_S][LD], pmt[ATT_S][LD];
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < ATT_S * 8; idx += 256) {
        const int t = idx >> 3, ch = (idx & 7) * 8;
        uint4 zq = make_uint4(0, 0, 0, 0), zk = zq, zv = zq, zg = zq;
        if (t < S) {
            const bf16_t* row = qkv + ((long long)b * S + t) * 3 * C + h * D + ch;
            zq = *(const uint4*)row; zk = *(const uint4*)(row + C); zv = *(const uint4*)(row + 2 * C);
            zg = *(const uint4*)(dout + ((long long)b * S + t) * C + h * D + ch);
        }
        *(uint4*)&q[t][ch] = zq; *(uint4*)&k[t][ch] = zk; *(uint4*)&v[t][ch] = zv; *(uint4*)&go[t][ch] = zg;
    }
    __syncthreads();
    const int g = lane >> 4, il = lane & 15, i = wave * 16 + il;
    const bool rowok = i < S;
    // dP[i][j] = sum_c dO[i][c] V[j][c]: acc[ct][r] <-> j = 16 ct + 4 g + r
    f32x4 dp[4], pr[4];
    float dot = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        dp[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 fg_ = *(const uint4*)&go[i][32 * ks + 8 * g];
            const uint4 fv = *(const uint4*)&v[16 * ct + il][32 * ks + 8 * g];
            mfma_chunk<bf16_t>(dp[ct], fv, fg_);
        }
        const int j0 = 16 * ct + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + r;
            float pv = 0.f;
            if (rowok && j <= i) pv = (float)P[(long long)bh * S * S + (long long)i * S + j];
            const float mf = drop_factor(dr, (unsigned long long)bh * S * S + (unsigned long long)i * S + j);
            pr[ct][r] = pv;
            dp[ct][r] = (rowok && j <= i) ? dp[ct][r] * mf : 0.f;
            dot = fmaf(dp[ct][r], pv, dot);
        }
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int j0 = 16 * ct + 4 * g;
        bf16x4 dsv, pmv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dsv[r] = (bf16_t)(pr[ct][r] * (dp[ct][r] - dot) * scale);
            pmv[r] = (bf16_t)(pr[ct][r] * drop_factor(dr, (unsigned long long)bh * S * S + (unsigned long long)i * S + j0 + r));
        }
        *(bf16x4*)&dst[i][j0] = dsv;
        *(bf16x4*)&pmt[i][j0] = pmv;
    }
    __syncthreads();
    // the wave's band of 16 rows t = 16 wave + lane % 16 of dq (rows i), dk and dv (rows j); acc[cb][r] <-> c = 16 cb + 4 g + r
    const unsigned char* kb = (const unsigned char*)&k[0][0];
    const unsigned char* qb = (const unsigned char*)&q[0][0];
    const unsigned char* gb = (const unsigned char*)&go[0][0];
    const unsigned char* db = (const unsigned char*)&dst[0][0];
    const unsigned char* pb = (const unsigned char*)&pmt[0][0];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // dq: reduction over j, ds read as rows with the index pattern of the transposed read (forward kernel)
            const uint2 dlo = *(const uint2*)&dst[i][32 * ks + 4 * g], dhi = *(const uint2*)&dst[i][32 * ks + 16 + 4 * g];
            mfma_chunk<bf16_t>(aq, attn_frag_tr(kb, LD * 2, 16 * cb, ks, lane), make_uint4(dlo.x, dlo.y, dhi.x, dhi.y));
            // dk, dv: reduction over i on both sides
            mfma_chunk<bf16_t>(ak, attn_frag_tr(qb, LD * 2, 16 * cb, ks, lane), attn_frag_tr(db, LD * 2, 16 * wave, ks, lane));
            mfma_chunk<bf16_t>(av, attn_frag_tr(gb, LD * 2, 16 * cb, ks, lane), attn_frag_tr(pb, LD * 2, 16 * wave, ks, lane));
        }
        if (rowok) {
            bf16_t* row = dqkv + ((long long)b * S + i) * 3 * C + h * D + 16 * cb + 4 * g;
            store4(row, aq);
            store4(row + C, ak);
            store4(row + 2 * C, av);
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// r = a + b (b may be null);  y = (r - mean) * rstd * w + bias   — one wave per row, torch.nn.LayerNorm (eps inside the sqrt)
template <typename T>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b2, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ r_out, T* __restrict__ y,
                                                         float* __restrict__ stats, int M, int C, float eps, Drop dr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4n = C / 4;
    // r = a + dropout(b): the sum is formed once (dropout factors are per element), kept in r_out when given
    auto rsum = [&](int m, int c4) {
        f32x4 v = load4(a + (long long)m * C + c4 * 4);
        if (b2) {
            f32x4 w2 = load4(b2 + (long long)m * C + c4 * 4);
            if (dr.thresh) {
#pragma unroll
                for (int e = 0; e < 4; ++e) w2[e] *= drop_factor(dr, (unsigned long long)m * C + c4 * 4 + e);
            }
            v += w2;
        }
        return v;
    };
    // C <= 1024: a lane's (at most four) column groups of the row stay in registers over the three passes — the row is read ONCE (it was
    // read in every pass: three dependent memory round trips per row and wave; 23 us per launch at 15 360 x 512).  Same values, same
    // order of summation: identical results.
    constexpr int LNF_IT = 4;
    if (c4n <= 64 * LNF_IT) {
        for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
            f32x4 v[LNF_IT];
            float s1 = 0.f;
#pragma unroll
            for (int u = 0; u < LNF_IT; ++u) {
                const int c4 = lane + 64 * u;
                if (c4 < c4n) {
                    v[u] = rsum(m, c4);
                    if (r_out) store4(r_out + (long long)m * C + c4 * 4, v[u]);
                    s1 += v[u][0] + v[u][1] + v[u][2] + v[u][3];
                }
            }
            const float mean = wave_sum(s1) / (float)C;
            float s2 = 0.f;
#pragma unroll
            for (int u = 0; u < LNF_IT; ++u) {
                if (lane + 64 * u < c4n) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s2 += (v[u][e] - mean) * (v[u][e] - mean);
                }
            }
            const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
            if (lane == 0) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
#pragma unroll
            for (int u = 0; u < LNF_IT; ++u) {
                const int c4 = lane + 64 * u;
                if (c4 < c4n) {
                    const f32x4 wv = *(const f32x4*)(w + c4 * 4), bv = *(const f32x4*)(bias + c4 * 4);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (v[u][e] - mean) * rstd * wv[e] + bv[e];
                    store4(y + (long long)m * C + c4 * 4, o);
                }
            }
        }
        return;
    }
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        float s1 = 0.f;
        for (int c4 = lane; c4 < c4n; c4 += 64) {
            const f32x4 v = rsum(m, c4);
            if (r_out) store4(r_out + (long long)m * C + c4 * 4, v);
            s1 += v[0] + v[1] + v[2] + v[3];
        }
        const float mean = wave_sum(s1) / (float)C;
        float s2 = 0.f;
        for (int c4 = lane; c4 < c4n; c4 += 64) {
            const f32x4 v = rsum(m, c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s2 += (v[e] - mean) * (v[e] - mean);
        }
        const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
        if (lane == 0) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
        for (int c4 = lane; c4 < c4n; c4 += 64) {
            const f32x4 v = rsum(m, c4);
            const f32x4 wv = *(const f32x4*)(w + c4 * 4), bv = *(const f32x4*)(bias + c4 * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * wv[e] + bv[e];
            store4(y + (long long)m * C + c4 * 4, o);
        }
    }
}

// LayerNorm backward.  dy = g1 (+ g2); with bcast > 0 the gradient row of m is g1[m / bcast] * gscale (mean over time).
// dr = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),  dxhat = dy * w.
// Per-block partial sums of dw = sum dy * xhat and db = sum dy go to slabs[blk][2][C].
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ g1, const T* __restrict__ g2, const T* __restrict__ r,
                                                     const float* __restrict__ stats, const float* __restrict__ w,
                                                     T* __restrict__ dr, float* __restrict__ slabs, int M, int C, int bcast,
                                                     float gscale, T* __restrict__ dr_b, Drop drp) {
    extern __shared__ __attribute__((aligned(16))) float acc[];      // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4n = C / 4;
    // A lane owns the column groups lane, lane + 64, ... (at most LN_IT of them: C <= 1024): the weight / bias gradient sums
    // of its columns stay in registers over all rows of the wave, and a row's operands are loaded once for both passes.
    constexpr int LN_IT = 4;
    f32x4 gw[LN_IT], gb[LN_IT], wv[LN_IT];
#pragma unroll
    for (int u = 0; u < LN_IT; ++u) {
        gw[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gb[u] = gw[u];
        const int c4 = lane + 64 * u;
        wv[u] = c4 < c4n ? *(const f32x4*)(w + c4 * 4) : gw[u];
    }
    // The operands of the wave's NEXT row are requested before this row is reduced (a row was: loads -> two wave reductions -> stores, one
    // memory round trip after the other, 7 - 8 rows per wave), and they are waited for in front of this row's stores: loads and stores
    // share the wait counter, a wait behind the stores would also wait for their acknowledgement.
    const int mstep = gridDim.x * 4;
    f32x4 na[LN_IT], nb[LN_IT], nr[LN_IT];
    float nmean = 0.f, nrstd = 0.f;
    auto fetch = [&](int m) {
        nmean = stats[2 * m];
        nrstd = stats[2 * m + 1];
        const long long grow = bcast > 0 ? (long long)(m / bcast) * C : (long long)m * C;
#pragma unroll
        for (int u = 0; u < LN_IT; ++u) {
            const int c4 = lane + 64 * u;
            if (c4 < c4n) {
                na[u] = load4(g1 + grow + c4 * 4);
                if (g2) nb[u] = load4(g2 + (long long)m * C + c4 * 4);
                nr[u] = load4(r + (long long)m * C + c4 * 4);
            }
        }
    };
    int m0 = blockIdx.x * 4 + wave;
    if (m0 < M) fetch(m0);
    for (int m = m0; m < M; m += mstep) {
        const float mean = nmean, rstd = nrstd;
        f32x4 ca[LN_IT], cb[LN_IT], cr[LN_IT];
#pragma unroll
        for (int u = 0; u < LN_IT; ++u) { ca[u] = na[u]; cb[u] = nb[u]; cr[u] = nr[u]; }
        if (m + mstep < M) fetch(m + mstep);
        f32x4 dy[LN_IT], xh[LN_IT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < LN_IT; ++u) {
            const int c4 = lane + 64 * u;
            if (c4 < c4n) {
                dy[u] = ca[u] * gscale;
                if (g2) dy[u] += cb[u];
                const f32x4 rv = cr[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[u][e] = (rv[e] - mean) * rstd;
                    const float dxh = dy[u][e] * wv[u][e];
                    s1 += dxh;
                    s2 += dxh * xh[u][e];
                    gw[u][e] += dy[u][e] * xh[u][e];
                    gb[u][e] += dy[u][e];
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the next row's operands (and the previous row's stores) before this row's stores
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < LN_IT; ++u) {
            const int c4 = lane + 64 * u;
            if (c4 < c4n) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (dy[u][e] * wv[u][e] - m1 - xh[u][e] * m2);
                store4(dr + (long long)m * C + c4 * 4, o);
                if (dr_b) {         // gradient of the dropped-out summand b of r = a + dropout(b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= drop_factor(drp, (unsigned long long)m * C + c4 * 4 + e);
                    store4(dr_b + (long long)m * C + c4 * 4, o);
                }
            }
        }
    }
    float* aw = acc + (long long)wave * 2 * C;
#pragma unroll
    for (int u = 0; u < LN_IT; ++u) {
        const int c4 = lane + 64 * u;
        if (c4 < c4n) {
            *(f32x4*)(aw + c4 * 4) = gw[u];
            *(f32x4*)(aw + C + c4 * 4) = gb[u];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        slabs[(long long)blockIdx.x * 2 * C + i] = acc[i] + acc[2 * C + i] + acc[4 * C + i] + acc[6 * C + i];
}

// m[b][c] = (1/S) sum_t x[(b,t)][c]
template <typename T>
__global__ __launch_bounds__(256) void mean_time_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int S, int C) {
    const int total = B * C;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int b = idx / C, c = idx % C;
        float s = 0.f;
        for (int t = 0; t < S; ++t) s += to_f32(x[((long long)b * S + t) * C + c]);
        out[idx] = from_f32<T>(s / (float)S);
    }
}

// x[i] *= dropout factor (in place);  mask[i] = factor (f32, for tests)
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(T* __restrict__ x, long long n, Drop dr) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        x[i] = from_f32<T>(to_f32(x[i]) * drop_factor(dr, (unsigned long long)i));
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, long long n, Drop dr) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        mask[i] = drop_factor(dr, (unsigned long long)i);
}

}  // namespace

static Drop make_drop(float p, unsigned long long seed, unsigned site) {
    Drop d;
    d.seed = seed;
    d.site = site;
    d.thresh = p > 0.f ? (unsigned)fmin(4294967295.0, (double)p * 4294967296.0) : 0u;
    d.inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    return d;
}

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32)     \
    do {                                           \
        if ((dtype) == CPC_DTYPE_BF16) { CALL_BF16; } \
        else if ((dtype) == CPC_DTYPE_F32) { CALL_F32; } \
        else return CPC_EINVAL;                    \
    } while (0)

int launch_pe_scale_fwd(const void* top, const float* pe, void* x0, int B, int S, int C, long long item_stride, float scale,
                        int dtype, hipStream_t st) {
    if (B <= 0 || S <= 0 || C <= 0 || C % 4) return CPC_EINVAL;
    const int blocks = (int)min((long long)2048, ((long long)B * S * (C / 4) + 255) / 256);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((pe_scale_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)top, pe, (bf16_t*)x0, B, S, C, item_stride, scale),
               hipLaunchKernelGGL((pe_scale_fwd_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)top, pe, (float*)x0, B, S, C, item_stride, scale));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_pe_scale_bwd(const void* g1, const void* g2, void* dtop, int B, int S, int C, long long item_stride, float scale,
                        int dtype, hipStream_t st) {
    if (B <= 0 || S <= 0 || C <= 0 || C % 4) return CPC_EINVAL;
    const int blocks = (int)min((long long)2048, ((long long)B * S * (C / 4) + 255) / 256);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((pe_scale_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)g1, (const bf16_t*)g2, (bf16_t*)dtop, B, S, C, item_stride, scale),
               hipLaunchKernelGGL((pe_scale_bwd_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)g1, (const float*)g2, (float*)dtop, B, S, C, item_stride, scale));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

static bool attn_ok(int B, int S, int C, int heads) {
    return B > 0 && S > 0 && S <= ATT_S && heads > 0 && C % heads == 0 && C / heads <= ATT_D && (C / heads) % 4 == 0;
}

int launch_attn_fwd(const void* qkv, void* out, void* P, int B, int S, int C, int heads, float drop_p, unsigned long long seed,
                    unsigned site, int dtype, hipStream_t st) {
    if (!attn_ok(B, S, C, heads) || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop dr = make_drop(drop_p, seed, site);
    const float scale = 1.f / sqrtf((float)(C / heads));
    static const bool use_mfma = !(getenv("CPC_ATTN_MFMA") && atoi(getenv("CPC_ATTN_MFMA")) == 0);
    if (use_mfma && dtype == CPC_DTYPE_BF16 && C / heads == 64 && C % 8 == 0 && ((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 8 == 0) &&
        ((uintptr_t)P % 8 == 0)) {
        hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(B * heads), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out, (bf16_t*)P, S, C, heads,
                           scale, dr);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), dim3(B * heads), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out, (bf16_t*)P, S, C, heads, scale, dr),
               hipLaunchKernelGGL((attn_fwd_kernel<float>), dim3(B * heads), dim3(256), 0, st, (const float*)qkv, (float*)out, (float*)P, S, C, heads, scale, dr));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_attn_bwd(const void* qkv, const void* P, const void* dout, void* dqkv, int B, int S, int C, int heads, float drop_p,
                    unsigned long long seed, unsigned site, int dtype, hipStream_t st) {
    if (!attn_ok(B, S, C, heads) || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop dr = make_drop(drop_p, seed, site);
    const float scale = 1.f / sqrtf((float)(C / heads));
    static const bool use_mfma = !(getenv("CPC_ATTN_MFMA") && atoi(getenv("CPC_ATTN_MFMA")) == 0);
    if (use_mfma && dtype == CPC_DTYPE_BF16 && C / heads == 64 && C % 8 == 0 && ((uintptr_t)qkv % 16 == 0) && ((uintptr_t)dout % 16 == 0) &&
        ((uintptr_t)dqkv % 8 == 0)) {
        hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(B * heads), dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)P, (const bf16_t*)dout,
                           (bf16_t*)dqkv, S, C, heads, scale, dr);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((attn_bwd_kernel<bf16_t>), dim3(B * heads), dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)P, (const bf16_t*)dout, (bf16_t*)dqkv, S, C, heads, scale, dr),
               hipLaunchKernelGGL((attn_bwd_kernel<float>), dim3(B * heads), dim3(256), 0, st, (const float*)qkv, (const float*)P, (const float*)dout, (float*)dqkv, S, C, heads, scale, dr));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_add_ln_fwd(const void* a, const void* b, const float* w, const float* bias, void* r_out, void* y, float* stats, int M,
                      int C, float eps, float drop_p, unsigned long long seed, unsigned site, int dtype, hipStream_t st) {
    if (M <= 0 || C <= 0 || C % 4 || !w || !bias || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop dr = make_drop(drop_p, seed, site);
    const int blocks = min(2048, (M + 3) / 4);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((add_ln_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, w, bias, (bf16_t*)r_out, (bf16_t*)y, stats, M, C, eps, dr),
               hipLaunchKernelGGL((add_ln_fwd_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)a, (const float*)b, w, bias, (float*)r_out, (float*)y, stats, M, C, eps, dr));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_ln_bwd(const void* g1, const void* g2, const void* r, const float* stats, const float* w, void* dr, float* slabs, int M,
                  int C, int bcast, float gscale, int nblocks, void* dr_b, float drop_p, unsigned long long seed, unsigned site,
                  int dtype, hipStream_t st) {
    if (M <= 0 || C <= 0 || C % 4 || C > 1024 || nblocks <= 0 || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop drp = make_drop(drop_p, seed, site);
    const size_t shm = (size_t)4 * 2 * C * sizeof(float);
    if (shm > 64 * 1024) return CPC_EINVAL;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((ln_bwd_kernel<bf16_t>), dim3(nblocks), dim3(256), shm, st, (const bf16_t*)g1, (const bf16_t*)g2, (const bf16_t*)r, stats, w, (bf16_t*)dr, slabs, M, C, bcast, gscale, (bf16_t*)dr_b, drp),
               hipLaunchKernelGGL((ln_bwd_kernel<float>), dim3(nblocks), dim3(256), shm, st, (const float*)g1, (const float*)g2, (const float*)r, stats, w, (float*)dr, slabs, M, C, bcast, gscale, (float*)dr_b, drp));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_mean_time(const void* x, void* out, int B, int S, int C, int dtype, hipStream_t st) {
    if (B <= 0 || S <= 0 || C <= 0) return CPC_EINVAL;
    const int blocks = min(1024, (B * C + 255) / 256);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((mean_time_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, B, S, C),
               hipLaunchKernelGGL((mean_time_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)out, B, S, C));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_dropout(void* x, long long n, float drop_p, unsigned long long seed, unsigned site, int dtype, hipStream_t st) {
    if (n <= 0 || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop dr = make_drop(drop_p, seed, site);
    const int blocks = (int)min((long long)4096, (n + 255) / 256);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (bf16_t*)x, n, dr),
               hipLaunchKernelGGL((dropout_kernel<float>), dim3(blocks), dim3(256), 0, st, (float*)x, n, dr));
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_dropout_mask(float* mask, long long n, float drop_p, unsigned long long seed, unsigned site, hipStream_t st) {
    if (n <= 0 || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop dr = make_drop(drop_p, seed, site);
    const int blocks = (int)min((long long)4096, (n + 255) / 256);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, st, mask, n, dr);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// =====================================================================================================================
// Wasserstein gradient penalty through the attention context (contrastive_estimation_training.py:144-158 differentiated through
// attention_model.py:72-82 / transformer.py:262-271).  The penalty's parameter gradient is the reverse sweep of the joint (primal,
// tangent) program (tools/gp_attention_algebra.py checks the algebra against autograd's double backward):
//   ln_tangent / attn_tangent     the tangent pass through residual + LayerNorm and through the (item, head) attention;
//   ln_gp / attn_gp               what the primal inputs of those two gain through the tangent program's coefficients, given the
//                                 adjoint delta of the SUMMED SCORES at their output (added to the adjoints of the last pass), and
//                                 the penalty part of the LayerNorm weight gradient.
// f32 only (the penalty runs in the exact-f32 mode); parity kernels: operands straight from global memory, S x S matrices in LDS.
namespace {

// rt = at + dropout(bt);  yt = w * rstd * (rt - <rt> - xh <xh rt>),  xh = (r - mean) * rstd      — one wave per row
__global__ __launch_bounds__(256) void ln_tangent_kernel(const float* __restrict__ at, const float* __restrict__ bt,
                                                         const float* __restrict__ r, const float* __restrict__ stats,
                                                         const float* __restrict__ w, float* __restrict__ rt_out,
                                                         float* __restrict__ yt, int M, int C, Drop dr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        const float mean = stats[2 * m], rstd = stats[2 * m + 1];
        const long long o = (long long)m * C;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            float v = at[o + c];
            if (bt) v += bt[o + c] * drop_factor(dr, (unsigned long long)o + c);
            if (rt_out) rt_out[o + c] = v;
            s1 += v;
            s2 += v * (r[o + c] - mean) * rstd;
        }
        const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
        for (int c = lane; c < C; c += 64) {
            float v = at[o + c];
            if (bt) v += bt[o + c] * drop_factor(dr, (unsigned long long)o + c);
            const float xh = (r[o + c] - mean) * rstd;
            yt[o + c] = w[c] * rstd * (v - m1 - xh * m2);
        }
    }
}

// dy = g1 * gscale (+ g2) (g1 row m / bcast when bcast > 0), p = dy * w, Pu = u - <u> - xh <xh u>:
//   dr[m] += -rstd^2 (xh <p P rt> + <xh rt> P p + <p xh> P rt)      (and dr_b[m] += the same times the dropout factor)
//   slabs[blk][c] = sum over the block's rows of dy * rstd * P rt     (penalty part of the LayerNorm weight gradient)
__global__ __launch_bounds__(256) void ln_gp_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                    const float* __restrict__ rt, const float* __restrict__ r,
                                                    const float* __restrict__ stats, const float* __restrict__ w,
                                                    float* __restrict__ dr, float* __restrict__ dr_b, float* __restrict__ slabs, int M,
                                                    int C, int bcast, float gscale, Drop drp) {
    extern __shared__ __attribute__((aligned(16))) float acc[];      // [4 waves][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int IT = 16;           // C <= 1024
    float gw[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) gw[u] = 0.f;
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        const float mean = stats[2 * m], rstd = stats[2 * m + 1];
        const long long o = (long long)m * C, go = bcast > 0 ? (long long)(m / bcast) * C : o;
        float s_rt = 0.f, s_xrt = 0.f, s_p = 0.f, s_px = 0.f, s_prt = 0.f;
        for (int c = lane; c < C; c += 64) {
            float dy = g1[go + c] * gscale;
            if (g2) dy += g2[o + c];
            const float p = dy * w[c], xh = (r[o + c] - mean) * rstd, t = rt[o + c];
            s_rt += t; s_xrt += xh * t; s_p += p; s_px += p * xh; s_prt += p * t;
        }
        const float inv = 1.f / (float)C;
        const float m_rt = wave_sum(s_rt) * inv, b = wave_sum(s_xrt) * inv, m_p = wave_sum(s_p) * inv, a = wave_sum(s_px) * inv;
        const float ppr = wave_sum(s_prt) * inv - m_p * m_rt - a * b;      // <p P rt>
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int c = lane + 64 * u;
            if (c < C) {
                float dy = g1[go + c] * gscale;
                if (g2) dy += g2[o + c];
                const float p = dy * w[c], xh = (r[o + c] - mean) * rstd, t = rt[o + c];
                const float pr = t - m_rt - xh * b, pp = p - m_p - xh * a;
                const float src = -rstd * rstd * (xh * ppr + b * pp + a * pr);
                dr[o + c] += src;
                if (dr_b) dr_b[o + c] += src * drop_factor(drp, (unsigned long long)o + c);
                gw[u] += dy * rstd * pr;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const int c = lane + 64 * u;
        if (c < C) acc[wave * C + c] = gw[u];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        slabs[(long long)blockIdx.x * C + c] = acc[c] + acc[C + c] + acc[2 * C + c] + acc[3 * C + c];
}

__device__ __forceinline__ float dot_rows(const float* a, const float* b, int d) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < d; c += 4) acc += *(const f32x4*)(a + c) * *(const f32x4*)(b + c);
    return acc[0] + acc[1] + acc[2] + acc[3];
}

// Tangent of the (item, head) attention: u = scale (qt k^T + q kt^T), pt = P (u - <P, u>), out_t = (pt m) v + (P m) vt, m the
// dropout factors of the attention weights.
__global__ __launch_bounds__(256) void attn_tangent_kernel(const float* __restrict__ qkv, const float* __restrict__ qkvt,
                                                           const float* __restrict__ P, float* __restrict__ out_t, int S, int C,
                                                           int heads, float scale, Drop dr) {
    __shared__ float u[ATT_S][ATT_S + 1], pm[ATT_S][ATT_S + 1];
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads, d = C / heads, tid = threadIdx.x;
    const float* base = qkv + (long long)b * S * 3 * C + h * d;
    const float* tbase = qkvt + (long long)b * S * 3 * C + h * d;
    const float* Pb = P + (long long)bh * S * S;
    auto row = [&](const float* p0, int t, int which) { return p0 + (long long)t * 3 * C + which * C; };
    for (int idx = tid; idx < S * S; idx += 256) {
        const int i = idx / S, j = idx % S;
        u[i][j] = j <= i ? scale * (dot_rows(row(tbase, i, 0), row(base, j, 1), d) + dot_rows(row(base, i, 0), row(tbase, j, 1), d)) : 0.f;
    }
    __syncthreads();
    {
        const int i = tid >> 2, part = tid & 3;
        if (i < S) {
            float mrow = 0.f;
            for (int j = part; j <= i; j += 4) mrow = fmaf(Pb[i * S + j], u[i][j], mrow);
            mrow += __shfl_xor(mrow, 1, 64);
            mrow += __shfl_xor(mrow, 2, 64);
            for (int j = part; j < S; j += 4) {
                const float pv = j <= i ? Pb[i * S + j] : 0.f;
                const float mf = drop_factor(dr, (unsigned long long)bh * S * S + (unsigned long long)i * S + j);
                u[i][j] = pv * (u[i][j] - mrow) * mf;
                pm[i][j] = pv * mf;
            }
        }
    }
    __syncthreads();
    const int d4 = d / 4;
    for (int idx = tid; idx < S * d4; idx += 256) {
        const int i = idx / d4, c = (idx % d4) * 4;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j <= i; ++j) o += u[i][j] * *(const f32x4*)(row(base, j, 2) + c) + pm[i][j] * *(const f32x4*)(row(tbase, j, 2) + c);
        *(f32x4*)(out_t + ((long long)b * S + i) * C + h * d + c) = o;
    }
}

// Second-order terms of the attention: with a = m (dO v^T), cc = <P, a>, dS = P (a - cc), u and pt as in attn_tangent,
// e = m (dO vt^T), w = e + (a - cc)(u - <P,u>), sig = P (w - <P, w>):
//   dq += scale (sig k + dS kt),  dk += scale (sig^T q + dS^T qt),  dv += (pt m)^T dO.       dO: adjoint of the summed scores.
__global__ __launch_bounds__(256) void attn_gp_kernel(const float* __restrict__ qkv, const float* __restrict__ qkvt,
                                                      const float* __restrict__ P, const float* __restrict__ dout,
                                                      float* __restrict__ dqkv, int S, int C, int heads, float scale, Drop dr) {
    __shared__ float u[ATT_S][ATT_S + 1], a[ATT_S][ATT_S + 1], e[ATT_S][ATT_S + 1];
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads, d = C / heads, tid = threadIdx.x;
    const float* base = qkv + (long long)b * S * 3 * C + h * d;
    const float* tbase = qkvt + (long long)b * S * 3 * C + h * d;
    const float* go = dout + (long long)b * S * C + h * d;
    const float* Pb = P + (long long)bh * S * S;
    auto row = [&](const float* p0, int t, int which) { return p0 + (long long)t * 3 * C + which * C; };
    for (int idx = tid; idx < S * S; idx += 256) {
        const int i = idx / S, j = idx % S;
        float uv = 0.f, av = 0.f, ev = 0.f;
        if (j <= i) {
            const float mf = drop_factor(dr, (unsigned long long)bh * S * S + idx);
            uv = scale * (dot_rows(row(tbase, i, 0), row(base, j, 1), d) + dot_rows(row(base, i, 0), row(tbase, j, 1), d));
            av = mf * dot_rows(go + (long long)i * C, row(base, j, 2), d);
            ev = mf * dot_rows(go + (long long)i * C, row(tbase, j, 2), d);
        }
        u[i][j] = uv; a[i][j] = av; e[i][j] = ev;
    }
    __syncthreads();
    {
        const int i = tid >> 2, part = tid & 3;
        if (i < S) {
            float mrow = 0.f, cc = 0.f;
            for (int j = part; j <= i; j += 4) { const float pv = Pb[i * S + j]; mrow = fmaf(pv, u[i][j], mrow); cc = fmaf(pv, a[i][j], cc); }
            mrow += __shfl_xor(mrow, 1, 64); mrow += __shfl_xor(mrow, 2, 64);
            cc += __shfl_xor(cc, 1, 64); cc += __shfl_xor(cc, 2, 64);
            float pw = 0.f;
            for (int j = part; j <= i; j += 4) {
                const float wv = e[i][j] + (a[i][j] - cc) * (u[i][j] - mrow);
                e[i][j] = wv;
                pw = fmaf(Pb[i * S + j], wv, pw);
            }
            pw += __shfl_xor(pw, 1, 64); pw += __shfl_xor(pw, 2, 64);
            for (int j = part; j < S; j += 4) {
                const float pv = j <= i ? Pb[i * S + j] : 0.f;
                const float mf = drop_factor(dr, (unsigned long long)bh * S * S + (unsigned long long)i * S + j);
                const float uu = u[i][j], aa = a[i][j], ww = e[i][j];
                u[i][j] = pv * (uu - mrow) * mf;       // pt m
                a[i][j] = pv * (aa - cc);              // dS
                e[i][j] = pv * (ww - pw);              // sig
            }
        }
    }
    __syncthreads();
    const int d4 = d / 4;
    for (int idx = tid; idx < S * d4; idx += 256) {
        const int t = idx / d4, c = (idx % d4) * 4;
        f32x4 sq = {0.f, 0.f, 0.f, 0.f}, sk = sq, sv = sq;
        for (int j = 0; j <= t; ++j)
            sq += e[t][j] * *(const f32x4*)(row(base, j, 1) + c) + a[t][j] * *(const f32x4*)(row(tbase, j, 1) + c);
        for (int i = t; i < S; ++i) {
            sk += e[i][t] * *(const f32x4*)(row(base, i, 0) + c) + a[i][t] * *(const f32x4*)(row(tbase, i, 0) + c);
            sv += u[i][t] * *(const f32x4*)(go + (long long)i * C + c);
        }
        float* o = dqkv + ((long long)b * S + t) * 3 * C + h * d + c;
        *(f32x4*)o += sq * scale;
        *(f32x4*)(o + C) += sk * scale;
        *(f32x4*)(o + 2 * C) += sv;
    }
}

}  // namespace

int launch_ln_tangent(const float* at, const float* bt, const float* r, const float* stats, const float* w, float* rt_out, float* yt,
                      int M, int C, float drop_p, unsigned long long seed, unsigned site, hipStream_t st) {
    if (M <= 0 || C <= 0 || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop dr = make_drop(drop_p, seed, site);
    hipLaunchKernelGGL(ln_tangent_kernel, dim3(min(2048, (M + 3) / 4)), dim3(256), 0, st, at, bt, r, stats, w, rt_out, yt, M, C, dr);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_ln_gp(const float* g1, const float* g2, const float* rt, const float* r, const float* stats, const float* w, float* dr,
                 float* dr_b, float* slabs, int M, int C, int bcast, float gscale, int nblocks, float drop_p, unsigned long long seed,
                 unsigned site, hipStream_t st) {
    if (M <= 0 || C <= 0 || C > 1024 || nblocks <= 0 || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop drp = make_drop(drop_p, seed, site);
    hipLaunchKernelGGL(ln_gp_kernel, dim3(nblocks), dim3(256), (size_t)4 * C * sizeof(float), st, g1, g2, rt, r, stats, w, dr, dr_b,
                       slabs, M, C, bcast, gscale, drp);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_attn_tangent(const float* qkv, const float* qkvt, const float* P, float* out_t, int B, int S, int C, int heads,
                        float drop_p, unsigned long long seed, unsigned site, hipStream_t st) {
    if (!attn_ok(B, S, C, heads) || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop dr = make_drop(drop_p, seed, site);
    hipLaunchKernelGGL(attn_tangent_kernel, dim3(B * heads), dim3(256), 0, st, qkv, qkvt, P, out_t, S, C, heads,
                       1.f / sqrtf((float)(C / heads)), dr);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_attn_gp(const float* qkv, const float* qkvt, const float* P, const float* dout, float* dqkv, int B, int S, int C, int heads,
                   float drop_p, unsigned long long seed, unsigned site, hipStream_t st) {
    if (!attn_ok(B, S, C, heads) || drop_p < 0.f || drop_p >= 1.f) return CPC_EINVAL;
    const Drop dr = make_drop(drop_p, seed, site);
    hipLaunchKernelGGL(attn_gp_kernel, dim3(B * heads), dim3(256), 0, st, qkv, qkvt, P, dout, dqkv, S, C, heads,
                       1.f / sqrtf((float)(C / heads)), dr);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
