// Fused three-gate GRU recurrence (AudioGRUModel's context network), forward and backward through time.
//
// The input projection x_t * W_ih^T + b_ih has no recurrence and is one batched gemm_nt over all V steps (done by the
// caller).  What is sequential is h_{t-1} * W_hh^T and the gate math; that runs here as ONE persistent launch for all
// V steps: a workgroup owns 16 batch rows, keeps h (f32 master in registers, storage-dtype copy in LDS as the MFMA
// operand) on chip for the whole sequence and streams W_hh, pre-arranged in MFMA fragment order, from L2.
//
// MFMA orientation: D[row = gate column][col = batch row]; a lane therefore holds the r, u and n pre-activations of the
// same (batch row, hidden unit) quadruple in three accumulators and does the gate math without any exchange.
#include "cpc_common.h"
#include "cpc_kernels.h"

namespace {

constexpr int GRU_MAXJT = 4;      // hidden tiles (16 units) per wave -> H <= 256

template <typename T>
__device__ __forceinline__ uint4 ldg16(const T* p) { return *(const uint4*)p; }

// grid: ceil(B/16); block 256.
template <typename T>
__global__ __launch_bounds__(256) void gru_fwd_kernel(const float* __restrict__ Gi, const T* __restrict__ Wfrag,
                                                      const float* __restrict__ bhh, T* __restrict__ Hall,
                                                      T* __restrict__ gates, float* __restrict__ c_out, int B, int V,
                                                      int H) {
    constexpr int CH = Elem<T>::CH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int rowb = H * (int)sizeof(T) + 16;             // h tile row stride in bytes (16 B pad)
    unsigned char* hbuf[2] = {smem, smem + 16 * rowb};
    float* bsh = (float*)(smem + 2 * 16 * rowb);           // b_hh [3H]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int b0 = blockIdx.x * 16;
    const int b = b0 + frow;
    const bool b_ok = b < B;
    const int ntile = H / 16;
    const int KC = H / (4 * CH);

    for (int i = tid; i < 3 * H; i += 256) bsh[i] = bhh ? bhh[i] : 0.f;
    for (int i = tid; i < 16 * rowb / 4; i += 256) ((unsigned int*)hbuf[0])[i] = 0u;    // h_0 = 0
    // Hall[:, 0, :] = 0
    for (int i = tid; i < 16 * H; i += 256) {
        const int rb = i / H, j = i % H;
        if (b0 + rb < B) Hall[((long long)(b0 + rb) * (V + 1)) * H + j] = from_f32<T>(0.f);
    }
    __syncthreads();

    float hprev[GRU_MAXJT][4];
#pragma unroll
    for (int q = 0; q < GRU_MAXJT; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) hprev[q][e] = 0.f;

    for (int t = 0; t < V; ++t) {
        const unsigned char* hcur = hbuf[t & 1];
        unsigned char* hnext = hbuf[(t + 1) & 1];
        // input-projection terms of this step (issued early; consumed after the MFMA loop)
        f32x4 gi[GRU_MAXJT][3];
#pragma unroll
        for (int q = 0; q < GRU_MAXJT; ++q) {
            const int jt = wave + 4 * q;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                gi[q][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (jt < ntile && b_ok)
                    gi[q][g] = *(const f32x4*)(Gi + ((long long)b * V + t) * 3 * H + g * H + jt * 16 + fg * 4);
            }
        }
        f32x4 acc[GRU_MAXJT][3];
#pragma unroll
        for (int q = 0; q < GRU_MAXJT; ++q) {
            const int jt = wave + 4 * q;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                acc[q][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (jt < ntile) acc[q][g] = *(const f32x4*)(bsh + g * H + jt * 16 + fg * 4);
            }
        }

        for (int kc = 0; kc < KC; ++kc) {
            const uint4 hf = *(const uint4*)(hcur + frow * rowb + (kc * 4 + fg) * 16);
#pragma unroll
            for (int q = 0; q < GRU_MAXJT; ++q) {
                const int jt = wave + 4 * q;
                if (jt < ntile) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const int nt = g * ntile + jt;
                        const uint4 wf = ldg16(Wfrag + ((long long)(nt * KC + kc) * 64 + lane) * CH);
                        mfma_chunk<T>(acc[q][g], wf, hf);
                    }
                }
            }
        }
        // gate math for (b, j = jt*16 + fg*4 + e)
#pragma unroll
        for (int q = 0; q < GRU_MAXJT; ++q) {
            const int jt = wave + 4 * q;
            if (jt < ntile) {
                const int j = jt * 16 + fg * 4;
                f32x4 r4, u4, n4, q4, h4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = fast_sigmoid(gi[q][0][e] + acc[q][0][e]);
                    const float u = fast_sigmoid(gi[q][1][e] + acc[q][1][e]);
                    const float qq = acc[q][2][e];
                    const float n = fast_tanh(gi[q][2][e] + r * qq);
                    const float hn = (1.f - u) * n + u * hprev[q][e];
                    hprev[q][e] = hn;
                    r4[e] = r; u4[e] = u; n4[e] = n; q4[e] = qq; h4[e] = hn;
                }
                store4((T*)(hnext + frow * rowb) + j, h4);
                if (b_ok) {
                    store4(Hall + ((long long)b * (V + 1) + (t + 1)) * H + j, h4);
                    T* gp = gates + (((long long)b * V + t) * 4) * H + j;
                    store4(gp, r4);
                    store4(gp + H, u4);
                    store4(gp + 2 * H, n4);
                    store4(gp + 3 * H, q4);
                    if (t == V - 1) *(f32x4*)(c_out + (long long)b * H + j) = h4;
                }
            }
        }
        __syncthreads();
    }
}

// grid: ceil(B/16); block 256.  WTfrag: fragment-ordered W_hh^T ([H][3H] logical: rows = hidden unit, k = gate column).
template <typename T>
__global__ __launch_bounds__(256) void gru_bwd_kernel(const float* __restrict__ dc, const T* __restrict__ Hall,
                                                      const T* __restrict__ gates, const T* __restrict__ WTfrag,
                                                      T* __restrict__ dGi, T* __restrict__ dGh, int B, int V, int H) {
    constexpr int CH = Elem<T>::CH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int rowb = 3 * H * (int)sizeof(T) + 16;
    unsigned char* gcur = smem;                            // dgh tile [16][3H] (+16 B pad per row)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int b0 = blockIdx.x * 16;
    const int b = b0 + frow;
    const bool b_ok = b < B;
    const int ntile = H / 16;
    const int KC = 3 * H / (4 * CH);

    float dh[GRU_MAXJT][4];
#pragma unroll
    for (int q = 0; q < GRU_MAXJT; ++q) {
        const int jt = wave + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) dh[q][e] = 0.f;
        if (jt < ntile && b_ok) {
            const f32x4 v = *(const f32x4*)(dc + (long long)b * H + jt * 16 + fg * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dh[q][e] = v[e];
        }
    }

    for (int t = V - 1; t >= 0; --t) {
        float keep[GRU_MAXJT][4];
#pragma unroll
        for (int q = 0; q < GRU_MAXJT; ++q) {
            const int jt = wave + 4 * q;
            if (jt < ntile) {
                const int j = jt * 16 + fg * 4;
                f32x4 hp = (f32x4){0.f, 0.f, 0.f, 0.f}, r4 = hp, u4 = hp, n4 = hp, q4 = hp;
                if (b_ok) {
                    hp = load4(Hall + ((long long)b * (V + 1) + t) * H + j);
                    const T* gp = gates + (((long long)b * V + t) * 4) * H + j;
                    r4 = load4(gp); u4 = load4(gp + H); n4 = load4(gp + 2 * H); q4 = load4(gp + 3 * H);
                }
                f32x4 dr4, du4, dn4, dnr4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = dh[q][e];
                    const float dn = d * (1.f - u4[e]);
                    const float du = d * (hp[e] - n4[e]);
                    const float dn_pre = dn * (1.f - n4[e] * n4[e]);
                    const float du_pre = du * u4[e] * (1.f - u4[e]);
                    const float dr_pre = dn_pre * q4[e] * r4[e] * (1.f - r4[e]);
                    dr4[e] = dr_pre; du4[e] = du_pre; dn4[e] = dn_pre; dnr4[e] = dn_pre * r4[e];
                    keep[q][e] = d * u4[e];
                }
                T* grow = (T*)(gcur + frow * rowb);
                store4(grow + j, dr4);
                store4(grow + H + j, du4);
                store4(grow + 2 * H + j, dnr4);
                if (b_ok) {
                    T* gi = dGi + ((long long)b * V + t) * 3 * H + j;
                    T* gh = dGh + ((long long)b * V + t) * 3 * H + j;
                    store4(gi, dr4); store4(gi + H, du4); store4(gi + 2 * H, dn4);
                    store4(gh, dr4); store4(gh + H, du4); store4(gh + 2 * H, dnr4);
                }
            }
        }
        __syncthreads();
        // dh_{t-1} = dh * u + dgh * W_hh
        f32x4 acc[GRU_MAXJT];
#pragma unroll
        for (int q = 0; q < GRU_MAXJT; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kc = 0; kc < KC; ++kc) {
            const uint4 gf = *(const uint4*)(gcur + frow * rowb + (kc * 4 + fg) * 16);
#pragma unroll
            for (int q = 0; q < GRU_MAXJT; ++q) {
                const int jt = wave + 4 * q;
                if (jt < ntile) {
                    const uint4 wf = ldg16(WTfrag + ((long long)(jt * KC + kc) * 64 + lane) * CH);
                    mfma_chunk<T>(acc[q], wf, gf);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < GRU_MAXJT; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) dh[q][e] = keep[q][e] + acc[q][e];
        __syncthreads();      // all reads of the dgh tile are done before the next step overwrites it
    }
}

// ----------------------------------------------------------------------------------------------- weight-resident bf16
// bf16 path for H = 32*KC in {32, 64, 128, 256}: W_hh never leaves the CU during the sequence.  8 waves per workgroup
// (two per SIMD, so one wave's LDS / MFMA latency hides under the other's); the r and u gate fragments of a wave's
// hidden tiles live in its registers (H = 256: 128 VGPRs per lane), the n gate fragments in LDS (128 KiB), so a step
// costs 48 MFMAs + gate math per wave instead of a 393 KB L2 read per workgroup.
template <int NJT, int KC>
__global__ __launch_bounds__(512) void gru_fwd_res_kernel(const float* __restrict__ Gi, const bf16_t* __restrict__ Wfrag,
                                                             const float* __restrict__ bhh, bf16_t* __restrict__ Hall,
                                                             bf16_t* __restrict__ gates, float* __restrict__ c_out, int B,
                                                             int V) {
    constexpr int H = 32 * KC, NT = H / 16;
    constexpr int ROWB = H * 2 + 16;
    constexpr int WN_BYTES = 8 * NJT * KC * 1024;
    constexpr bool FULL = (NT == 8 * NJT);               // every (wave, q) pair maps to a real tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[WN_BYTES + 2 * 16 * ROWB + 3 * H * 4];
    unsigned char* wn = smem;
    unsigned char* hbuf0 = smem + WN_BYTES;
    unsigned char* hbuf1 = hbuf0 + 16 * ROWB;
    float* bsh = (float*)(hbuf1 + 16 * ROWB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int b0 = blockIdx.x * 16;
    const int b = b0 + frow;
    const bool b_ok = b < B;

    uint4 wr[NJT][2][KC];
#pragma unroll
    for (int q = 0; q < NJT; ++q) {
        const int jt = wave + 8 * q;
        const bool on = FULL || jt < NT;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
                wr[q][g][kc] = on ? ldg16(Wfrag + ((long long)((g * NT + jt) * KC + kc) * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
            const uint4 w2 = on ? ldg16(Wfrag + ((long long)((2 * NT + jt) * KC + kc) * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
            *(uint4*)(wn + (((wave * NJT + q) * KC + kc) * 64 + lane) * 16) = w2;
        }
    }
    for (int i = tid; i < 3 * H; i += 512) bsh[i] = bhh ? bhh[i] : 0.f;
    for (int i = tid; i < 16 * ROWB / 4; i += 512) ((unsigned int*)hbuf0)[i] = 0u;
    for (int i = tid; i < 16 * H; i += 512) {
        const int rb = i / H, j = i % H;
        if (b0 + rb < B) Hall[((long long)(b0 + rb) * (V + 1)) * H + j] = (bf16_t)0.f;
    }
    __syncthreads();

    float hprev[NJT][4];
#pragma unroll
    for (int q = 0; q < NJT; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) hprev[q][e] = 0.f;

    for (int t = 0; t < V; ++t) {
        const unsigned char* hcur = (t & 1) ? hbuf1 : hbuf0;
        unsigned char* hnext = (t & 1) ? hbuf0 : hbuf1;
        // input-projection terms of this step: issued first, consumed after the MFMA loop (the SIMD's other wave and
        // the MFMAs cover their latency)
        f32x4 gi[NJT][3];
        f32x4 acc[NJT][3];
#pragma unroll
        for (int q = 0; q < NJT; ++q) {
            const int jt = wave + 8 * q;
            const bool on = FULL || jt < NT;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                gi[q][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[q][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (on) {
                    if (b_ok) gi[q][g] = *(const f32x4*)(Gi + ((long long)b * V + t) * 3 * H + g * H + jt * 16 + fg * 4);
                    acc[q][g] = *(const f32x4*)(bsh + g * H + jt * 16 + fg * 4);
                }
            }
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const uint4 hf = *(const uint4*)(hcur + frow * ROWB + (kc * 4 + fg) * 16);
#pragma unroll
            for (int q = 0; q < NJT; ++q) {
                const uint4 w2 = *(const uint4*)(wn + (((wave * NJT + q) * KC + kc) * 64 + lane) * 16);
                mfma_chunk<bf16_t>(acc[q][0], wr[q][0][kc], hf);
                mfma_chunk<bf16_t>(acc[q][1], wr[q][1][kc], hf);
                mfma_chunk<bf16_t>(acc[q][2], w2, hf);
            }
        }
#pragma unroll
        for (int q = 0; q < NJT; ++q) {
            const int jt = wave + 8 * q;
            if (FULL || jt < NT) {
                const int j = jt * 16 + fg * 4;
                f32x4 r4, u4, n4, q4, h4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = fast_sigmoid(gi[q][0][e] + acc[q][0][e]);
                    const float u = fast_sigmoid(gi[q][1][e] + acc[q][1][e]);
                    const float qq = acc[q][2][e];
                    const float n = fast_tanh(gi[q][2][e] + r * qq);
                    const float hn = (1.f - u) * n + u * hprev[q][e];
                    hprev[q][e] = hn;
                    r4[e] = r; u4[e] = u; n4[e] = n; q4[e] = qq; h4[e] = hn;
                }
                store4((bf16_t*)(hnext + frow * ROWB) + j, h4);
                if (b_ok) {
                    store4(Hall + ((long long)b * (V + 1) + (t + 1)) * H + j, h4);
                    bf16_t* gp = gates + (((long long)b * V + t) * 4) * H + j;
                    store4(gp, r4);
                    store4(gp + H, u4);
                    store4(gp + 2 * H, n4);
                    store4(gp + 3 * H, q4);
                    if (t == V - 1) *(f32x4*)(c_out + (long long)b * H + j) = h4;
                }
            }
        }
        __syncthreads();
    }
}

// Backward twin: W_hh^T fragments for the r,u gate columns (k < 2H) in registers, the n gate columns in LDS.
template <int NJT, int KC>
__global__ __launch_bounds__(512) void gru_bwd_res_kernel(const float* __restrict__ dc, const bf16_t* __restrict__ Hall,
                                                             const bf16_t* __restrict__ gates, const bf16_t* __restrict__ WTfrag,
                                                             bf16_t* __restrict__ dGi, bf16_t* __restrict__ dGh, int B, int V) {
    constexpr int H = 32 * KC, NT = H / 16;
    constexpr int KC3 = 3 * KC, KCR = 2 * KC, KCL = KC;
    constexpr int ROWB = 3 * H * 2 + 16;
    constexpr int WL_BYTES = 8 * NJT * KCL * 1024;
    constexpr bool FULL = (NT == 8 * NJT);
    __shared__ __attribute__((aligned(16))) unsigned char smem[WL_BYTES + 16 * ROWB];
    unsigned char* wl = smem;
    unsigned char* gcur = smem + WL_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int b0 = blockIdx.x * 16;
    const int b = b0 + frow;
    const bool b_ok = b < B;

    uint4 wr[NJT][KCR];
    float dh[NJT][4];
#pragma unroll
    for (int q = 0; q < NJT; ++q) {
        const int jt = wave + 8 * q;
        const bool on = FULL || jt < NT;
#pragma unroll
        for (int kc = 0; kc < KCR; ++kc)
            wr[q][kc] = on ? ldg16(WTfrag + ((long long)(jt * KC3 + kc) * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int kc = 0; kc < KCL; ++kc) {
            const uint4 w2 = on ? ldg16(WTfrag + ((long long)(jt * KC3 + KCR + kc) * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
            *(uint4*)(wl + (((wave * NJT + q) * KCL + kc) * 64 + lane) * 16) = w2;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) dh[q][e] = 0.f;
        if (on && b_ok) {
            const f32x4 v = *(const f32x4*)(dc + (long long)b * H + jt * 16 + fg * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dh[q][e] = v[e];
        }
    }
    __syncthreads();

    // saved activations of step t-1 are prefetched while step t runs (bf16x4 = 8 bytes each)
    bf16x4 sv[NJT][5], svn[NJT][5];
#pragma unroll
    for (int q = 0; q < NJT; ++q) {
        const int jt = wave + 8 * q;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            sv[q][k] = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            svn[q][k] = sv[q][k];
        }
        if ((FULL || jt < NT) && b_ok) {
            const int j = jt * 16 + fg * 4;
            sv[q][0] = *(const bf16x4*)(Hall + ((long long)b * (V + 1) + (V - 1)) * H + j);
            const bf16_t* gp = gates + (((long long)b * V + (V - 1)) * 4) * H + j;
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[q][1 + k] = *(const bf16x4*)(gp + k * H);
        }
    }

    for (int t = V - 1; t >= 0; --t) {
        float keep[NJT][4];
#pragma unroll
        for (int q = 0; q < NJT; ++q) {
            const int jt = wave + 8 * q;
            if (FULL || jt < NT) {
                const int j = jt * 16 + fg * 4;
                if (b_ok && t > 0) {
                    svn[q][0] = *(const bf16x4*)(Hall + ((long long)b * (V + 1) + (t - 1)) * H + j);
                    const bf16_t* gpn = gates + (((long long)b * V + (t - 1)) * 4) * H + j;
#pragma unroll
                    for (int k = 0; k < 4; ++k) svn[q][1 + k] = *(const bf16x4*)(gpn + k * H);
                }
                f32x4 hp, r4, u4, n4, q4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hp[e] = (float)sv[q][0][e]; r4[e] = (float)sv[q][1][e]; u4[e] = (float)sv[q][2][e];
                    n4[e] = (float)sv[q][3][e]; q4[e] = (float)sv[q][4][e];
                }
                f32x4 dr4, du4, dn4, dnr4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = dh[q][e];
                    const float dn = d * (1.f - u4[e]);
                    const float du = d * (hp[e] - n4[e]);
                    const float dn_pre = dn * (1.f - n4[e] * n4[e]);
                    const float du_pre = du * u4[e] * (1.f - u4[e]);
                    const float dr_pre = dn_pre * q4[e] * r4[e] * (1.f - r4[e]);
                    dr4[e] = dr_pre; du4[e] = du_pre; dn4[e] = dn_pre; dnr4[e] = dn_pre * r4[e];
                    keep[q][e] = d * u4[e];
                }
                bf16_t* grow = (bf16_t*)(gcur + frow * ROWB);
                store4(grow + j, dr4);
                store4(grow + H + j, du4);
                store4(grow + 2 * H + j, dnr4);
                if (b_ok) {
                    bf16_t* gi = dGi + ((long long)b * V + t) * 3 * H + j;
                    bf16_t* gh = dGh + ((long long)b * V + t) * 3 * H + j;
                    store4(gi, dr4); store4(gi + H, du4); store4(gi + 2 * H, dn4);
                    store4(gh, dr4); store4(gh + H, du4); store4(gh + 2 * H, dnr4);
                }
            }
        }
        __syncthreads();
        f32x4 acc[NJT];
#pragma unroll
        for (int q = 0; q < NJT; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KCR; ++kc) {
            const uint4 gf = *(const uint4*)(gcur + frow * ROWB + (kc * 4 + fg) * 16);
#pragma unroll
            for (int q = 0; q < NJT; ++q) mfma_chunk<bf16_t>(acc[q], wr[q][kc], gf);
        }
#pragma unroll
        for (int kc = 0; kc < KCL; ++kc) {
            const uint4 gf = *(const uint4*)(gcur + frow * ROWB + ((KCR + kc) * 4 + fg) * 16);
#pragma unroll
            for (int q = 0; q < NJT; ++q) {
                const uint4 w2 = *(const uint4*)(wl + (((wave * NJT + q) * KCL + kc) * 64 + lane) * 16);
                mfma_chunk<bf16_t>(acc[q], w2, gf);
            }
        }
#pragma unroll
        for (int q = 0; q < NJT; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dh[q][e] = keep[q][e] + acc[q][e];
#pragma unroll
            for (int k = 0; k < 5; ++k) sv[q][k] = svn[q][k];
        }
        __syncthreads();
    }
}

// dst (fragment order, T) from a row-major f32 matrix.  Logical operand Wn[n][k], n < R, k < Kd:
//   transpose == 0: Wn[n][k] = src[n * ld + k];   transpose == 1: Wn[n][k] = src[k * ld + n]
// dst[((nt * KC + kc) * 64 + lane) * CH + e] = Wn[nt*16 + (lane&15)][kc*4*CH + (lane>>4)*CH + e]
template <typename T>
__global__ __launch_bounds__(256) void prep_frag_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int Kd,
                                                        long long ld, int transpose) {
    constexpr int CH = Elem<T>::CH;
    const int KC = Kd / (4 * CH);
    const long long total = (long long)R * Kd;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int e = (int)(idx % CH);
        const int lane = (int)((idx / CH) % 64);
        const long long blk = idx / (CH * 64);
        const int kc = (int)(blk % KC), nt = (int)(blk / KC);
        const int n = nt * 16 + (lane & 15);
        const int k = kc * 4 * CH + (lane >> 4) * CH + e;
        const float v = transpose ? src[(long long)k * ld + n] : src[(long long)n * ld + k];
        dst[idx] = from_f32<T>(v);
    }
}

}  // namespace

// Debug / A-B switch: 1 = always use the weight-streaming kernels (set through cpc_gru_set_streaming).
int g_gru_force_streaming = 0;

static bool gru_ok(int B, int V, int H, int dtype) {
    const int ch = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if (B <= 0 || V <= 0 || H <= 0) return false;
    if (H % 16 || H % (4 * ch) || H / 16 > 4 * GRU_MAXJT) return false;
    return dtype == CPC_DTYPE_BF16 || dtype == CPC_DTYPE_F32;
}

int launch_gru_fwd(const float* Gi, const void* Wfrag, const float* bhh, void* Hall, void* gates, float* c_out, int B,
                   int V, int H, int dtype, hipStream_t stream) {
    if (!gru_ok(B, V, H, dtype)) return CPC_EINVAL;
    const int esz = dtype == CPC_DTYPE_BF16 ? 2 : 4;
    const size_t shm = 2 * 16 * (size_t)(H * esz + 16) + 3 * H * sizeof(float);
    dim3 grid((B + 15) / 16);
    if (dtype == CPC_DTYPE_BF16 && !g_gru_force_streaming && (H == 32 || H == 64 || H == 128 || H == 256)) {
#define GRU_F(NJT, KC) \
    hipLaunchKernelGGL((gru_fwd_res_kernel<NJT, KC>), grid, dim3(512), 0, stream, Gi, (const bf16_t*)Wfrag, bhh, (bf16_t*)Hall, \
                       (bf16_t*)gates, c_out, B, V)
        if (H == 256) GRU_F(2, 8); else if (H == 128) GRU_F(1, 4); else if (H == 64) GRU_F(1, 2); else GRU_F(1, 1);
#undef GRU_F
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((gru_fwd_kernel<bf16_t>), grid, dim3(256), shm, stream, Gi, (const bf16_t*)Wfrag, bhh,
                           (bf16_t*)Hall, (bf16_t*)gates, c_out, B, V, H);
    else
        hipLaunchKernelGGL((gru_fwd_kernel<float>), grid, dim3(256), shm, stream, Gi, (const float*)Wfrag, bhh,
                           (float*)Hall, (float*)gates, c_out, B, V, H);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_gru_bwd(const float* dc, const void* Hall, const void* gates, const void* WTfrag, void* dGi, void* dGh, int B,
                   int V, int H, int dtype, hipStream_t stream) {
    if (!gru_ok(B, V, H, dtype)) return CPC_EINVAL;
    const int esz = dtype == CPC_DTYPE_BF16 ? 2 : 4;
    const size_t shm = 16 * (size_t)(3 * H * esz + 16);
    if (shm > 64 * 1024) return CPC_EINVAL;
    dim3 grid((B + 15) / 16);
    if (dtype == CPC_DTYPE_BF16 && !g_gru_force_streaming && (H == 32 || H == 64 || H == 128 || H == 256)) {
#define GRU_B(NJT, KC) \
    hipLaunchKernelGGL((gru_bwd_res_kernel<NJT, KC>), grid, dim3(512), 0, stream, dc, (const bf16_t*)Hall, (const bf16_t*)gates, \
                       (const bf16_t*)WTfrag, (bf16_t*)dGi, (bf16_t*)dGh, B, V)
        if (H == 256) GRU_B(2, 8); else if (H == 128) GRU_B(1, 4); else if (H == 64) GRU_B(1, 2); else GRU_B(1, 1);
#undef GRU_B
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((gru_bwd_kernel<bf16_t>), grid, dim3(256), shm, stream, dc, (const bf16_t*)Hall,
                           (const bf16_t*)gates, (const bf16_t*)WTfrag, (bf16_t*)dGi, (bf16_t*)dGh, B, V, H);
    else
        hipLaunchKernelGGL((gru_bwd_kernel<float>), grid, dim3(256), shm, stream, dc, (const float*)Hall,
                           (const float*)gates, (const float*)WTfrag, (float*)dGi, (float*)dGh, B, V, H);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_prep_frag(const float* src, void* dst, int R, int Kd, long long ld, int transpose, int dtype, hipStream_t stream) {
    const int ch = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if (R <= 0 || Kd <= 0 || R % 16 || Kd % (4 * ch)) return CPC_EINVAL;
    const long long total = (long long)R * Kd;
    const int blocks = (int)min((long long)1024, (total + 255) / 256);
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((prep_frag_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, src, (bf16_t*)dst, R, Kd, ld, transpose);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((prep_frag_kernel<float>), dim3(blocks), dim3(256), 0, stream, src, (float*)dst, R, Kd, ld, transpose);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
