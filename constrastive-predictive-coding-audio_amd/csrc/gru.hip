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
__global__ __launch_bounds__(256) void gru_fwd_kernel(const T* __restrict__ Gi, const T* __restrict__ Wfrag,
                                                      const float* __restrict__ bhh, T* __restrict__ Hall,
                                                      T* __restrict__ tape, float* __restrict__ c_out, int B, int V,
                                                      int H, const float* __restrict__ h0) {
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
    __syncthreads();          // (the loop below rewrites the tile with a carried state, by other threads)
    // Hall[:, 0, :] = 0
    // (h0: the state a reset_hidden=False model carries over from its previous call, audio_model.py:69, :75; NULL = zeros)
    for (int i = tid; i < 16 * H; i += 256) {
        const int rb = i / H, j = i % H;
        const float v = (h0 && b0 + rb < B) ? h0[(long long)(b0 + rb) * H + j] : 0.f;
        ((T*)(hbuf[0] + rb * rowb))[j] = from_f32<T>(v);
        if (b0 + rb < B) Hall[((long long)(b0 + rb) * (V + 1)) * H + j] = from_f32<T>(v);
    }
    __syncthreads();

    float hprev[GRU_MAXJT][4];
#pragma unroll
    for (int q = 0; q < GRU_MAXJT; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int jt = wave + 4 * q;
            hprev[q][e] = (h0 && b_ok && jt < ntile) ? h0[(long long)b * H + jt * 16 + fg * 4 + e] : 0.f;
        }

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
                    gi[q][g] = load4(Gi + ((long long)b * V + t) * 3 * H + g * H + jt * 16 + fg * 4);
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
                f32x4 r4, u4, n4, q4, h4, hp4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = fast_sigmoid(gi[q][0][e] + acc[q][0][e]);
                    const float u = fast_sigmoid(gi[q][1][e] + acc[q][1][e]);
                    const float qq = acc[q][2][e];
                    const float n = fast_tanh(gi[q][2][e] + r * qq);
                    hp4[e] = hprev[q][e];
                    const float hn = (1.f - u) * n + u * hprev[q][e];
                    hprev[q][e] = hn;
                    r4[e] = r; u4[e] = u; n4[e] = n; q4[e] = qq; h4[e] = hn;
                }
                store4((T*)(hnext + frow * rowb) + j, h4);
                if (b_ok) {
                    store4(Hall + ((long long)b * (V + 1) + (t + 1)) * H + j, h4);
                    T* gp = tape + (((long long)b * V + t) * 5) * H + j;      // streaming tape: [b][t][r,u,n,q,h_prev][H]
                    store4(gp, r4);
                    store4(gp + H, u4);
                    store4(gp + 2 * H, n4);
                    store4(gp + 3 * H, q4);
                    store4(gp + 4 * H, hp4);
                    if (t == V - 1) *(f32x4*)(c_out + (long long)b * H + j) = h4;
                }
            }
        }
        __syncthreads();
    }
}

// grid: ceil(B/16); block 256.  WTfrag: fragment-ordered W_hh^T ([H][3H] logical: rows = hidden unit, k = gate column).
template <typename T>
__global__ __launch_bounds__(256) void gru_bwd_kernel(const float* __restrict__ dc, const T* __restrict__ tape,
                                                      const T* __restrict__ WTfrag, T* __restrict__ dG, int B, int V, int H) {
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
                    const T* gp = tape + (((long long)b * V + t) * 5) * H + j;
                    r4 = load4(gp); u4 = load4(gp + H); n4 = load4(gp + 2 * H); q4 = load4(gp + 3 * H);
                    hp = load4(gp + 4 * H);
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
                    T* go = dG + ((long long)b * V + t) * 4 * H + j;          // [dr | du | dn | dn*r]
                    store4(go, dr4); store4(go + H, du4); store4(go + 2 * H, dn4); store4(go + 3 * H, dnr4);
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
// bf16 path for H = 32*KC in {32, 64, 128, 256}: W_hh never leaves the CU during the sequence and every global access
// of the recurrence is a full-row / 16-byte-per-lane coalesced one (a step is latency-bound on B/16 workgroups; the
// first version of these kernels spent 80 % of a step in 8-byte scattered loads and stores).
//   * 8 waves per workgroup; wave w owns the hidden tiles w, w+8: their r/u-gate MFMA fragments live in registers
//     (H = 256: 136 VGPRs per lane), the n-gate fragments in LDS (112 KiB).
//   * forward: the input-projection rows of the NEXT step are loaded row-contiguous into registers at the top of a
//     step and parked in an LDS tile once this step's gate math has read the current one; h_{t+1} goes to HBM from
//     the LDS tile that also feeds the next step's MFMAs; the saved activations go to a "tape" in lane-fragment
//     order (1 KiB per wave-instruction), which the backward kernel reads back with the same lane mapping.
//   * backward: gate gradients are assembled in an LDS tile [16][dr | du | dn | dn*r]; that tile is both the MFMA
//     operand of dh_{t-1} = dh*u + dgh W_hh and the source of the coalesced row stores of dG[b][t][0:4H).
constexpr int GRU_NW = 8;
template <int KC> struct GruCfg {
    static constexpr int H = 32 * KC, NT = H / 16;
    static constexpr int NJT = (NT + GRU_NW - 1) / GRU_NW;
    static constexpr bool FULL = (NT == GRU_NW * NJT);
    static constexpr int KCL = KC == 8 ? 7 : KC;          // k-chunks of the LDS-resident weight part (n gate)
    static constexpr int SLOTS = 2 * NJT + 1;             // tape: (r,u), (n,q) per tile + one slot of h_prev
};
__host__ __device__ __forceinline__ long long gru_tape_elems_res(int B, int V, int KC) {
    const int njt = ((32 * KC / 16) + GRU_NW - 1) / GRU_NW;
    return (long long)((B + 15) / 16) * V * GRU_NW * (2 * njt + 1) * 64 * 8;
}

__device__ __forceinline__ uint4 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)a[e]; o[4 + e] = (bf16_t)b[e]; }
    return __builtin_bit_cast(uint4, o);
}
__device__ __forceinline__ void unpack8(const uint4& v, f32x4& a, f32x4& b) {
    const bf16x8 i = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = (float)i[e]; b[e] = (float)i[4 + e]; }
}

template <int KC>
__global__ __launch_bounds__(64 * GRU_NW) void gru_fwd_res_kernel(const bf16_t* __restrict__ Gi, const bf16_t* __restrict__ Wfrag,
                                                                  const float* __restrict__ bhh, bf16_t* __restrict__ Hall,
                                                                  bf16_t* __restrict__ tape, float* __restrict__ c_out, int B,
                                                                  int V, const float* __restrict__ h0) {
    typedef GruCfg<KC> Cfg;
    constexpr int H = Cfg::H, NT = Cfg::NT, NJT = Cfg::NJT, KCL = Cfg::KCL, KCRN = KC - KCL, NW = GRU_NW, NTHR = 64 * NW;
    constexpr bool FULL = Cfg::FULL;
    constexpr int ROWB = H * 2 + 16;                      // h tile row stride
    constexpr int GROW = 3 * H * 2 + 16;                  // input-projection tile row stride
    constexpr int WN_BYTES = NW * NJT * KCL * 1024;
    constexpr int GCHUNKS = 16 * 3 * H / 8;               // 16-byte chunks of one input-projection tile
    constexpr int GPT = (GCHUNKS + NTHR - 1) / NTHR;      // chunks per thread
    __shared__ __attribute__((aligned(16))) unsigned char smem[WN_BYTES + 2 * 16 * ROWB + 16 * GROW + 3 * H * 4];
    unsigned char* wn = smem;
    unsigned char* hbuf0 = smem + WN_BYTES;
    unsigned char* hbuf1 = hbuf0 + 16 * ROWB;
    unsigned char* gtile = hbuf1 + 16 * ROWB;
    float* bsh = (float*)(gtile + 16 * GROW);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int b0 = blockIdx.x * 16;
    const int b = b0 + frow;
    const bool b_ok = b < B;

    uint4 wr[NJT][2][KC];
    uint4 wrn[NJT][KCRN > 0 ? KCRN : 1];
#pragma unroll
    for (int q = 0; q < NJT; ++q) {
        const int jt = wave + NW * q;
        const bool on = FULL || jt < NT;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
                wr[q][g][kc] = on ? ldg16(Wfrag + ((long long)((g * NT + jt) * KC + kc) * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
            const uint4 w2 = on ? ldg16(Wfrag + ((long long)((2 * NT + jt) * KC + kc) * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
            if (kc < KCL) *(uint4*)(wn + (((wave * NJT + q) * KCL + kc) * 64 + lane) * 16) = w2;
            else wrn[q][kc - KCL < (KCRN > 0 ? KCRN : 1) ? kc - KCL : 0] = w2;
        }
    }
    for (int i = tid; i < 3 * H; i += NTHR) bsh[i] = bhh ? bhh[i] : 0.f;
    for (int i = tid; i < 16 * ROWB / 4; i += NTHR) ((unsigned int*)hbuf0)[i] = 0u;
    if (h0) __syncthreads();          // (uniform: the loop below rewrites the tile with the carried state, by other threads)
    // Hall[:, 0, :] = 0 and the input-projection tile of step 0 (row-contiguous 16-byte chunks)
    for (int i = tid; i < 16 * H / 8; i += NTHR) {
        const int rb = i / (H / 8), cc = i % (H / 8);
        uint4 hv = make_uint4(0, 0, 0, 0);
        if (h0 && b0 + rb < B) {          // carried state (reset_hidden=False): the bf16 copy the MFMAs read, and Hall[:, 0]
            const float* hp = h0 + (long long)(b0 + rb) * H + cc * 8;
            hv = pack8(*(const f32x4*)hp, *(const f32x4*)(hp + 4));
            *(uint4*)(hbuf0 + rb * ROWB + cc * 16) = hv;
        }
        if (b0 + rb < B) *(uint4*)(Hall + ((long long)(b0 + rb) * (V + 1)) * H + cc * 8) = hv;
    }
    uint4 gpre[GPT];
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
        const int i = tid + u * NTHR;
        gpre[u] = make_uint4(0, 0, 0, 0);
        if (i < GCHUNKS) {
            const int rb = i / (3 * H / 8), cc = i % (3 * H / 8);
            if (b0 + rb < B) gpre[u] = *(const uint4*)(Gi + ((long long)(b0 + rb) * V) * 3 * H + cc * 8);
            *(uint4*)(gtile + rb * GROW + cc * 16) = gpre[u];
        }
    }
    __syncthreads();

    float hprev[NJT][4];
#pragma unroll
    for (int q = 0; q < NJT; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int jt = wave + NW * q;
            hprev[q][e] = (h0 && b_ok && (FULL || jt < NT)) ? h0[(long long)b * H + jt * 16 + fg * 4 + e] : 0.f;
        }
    const long long tape_bt = (long long)blockIdx.x * V;
    // Nothing may be in flight when the loop starts: with the (conditional) loads of h0 still counted as pending at the loop header, the
    // compiler's wait-count pass — which has one counter for loads and stores and cannot count through conditional requests — put
    // s_waitcnt vmcnt(0) in front of every step's first LDS read, i.e. right behind the requests for the next step's input-projection
    // rows: a full HBM round trip (and the acknowledgement of the previous step's tape stores) on the critical path of each of the V steps.
    // (The builtin, not inline asm: the pass reads S_WAITCNT instructions, not asm strings.  0x0F70 = vmcnt(0), the other counters open.)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    if (V > 1) {          // the rows of step 1 (those of step 0 are in the LDS tile already)
#pragma unroll
        for (int u = 0; u < GPT; ++u) {
            const int i = tid + u * NTHR;
            if (i < GCHUNKS) {
                const int rb = i / (3 * H / 8), cc = i % (3 * H / 8);
                if (b0 + rb < B) gpre[u] = *(const uint4*)(Gi + ((long long)(b0 + rb) * V + 1) * 3 * H + cc * 8);
            }
        }
    }

    for (int t = 0; t < V; ++t) {
        const unsigned char* hcur = (t & 1) ? hbuf1 : hbuf0;
        unsigned char* hnext = (t & 1) ? hbuf0 : hbuf1;
        // (gpre holds the input-projection rows of step t + 1, requested at the end of step t - 1; they are parked in LDS after this step's gate math)
        f32x4 acc[NJT][3];
#pragma unroll
        for (int q = 0; q < NJT; ++q) {
            const int jt = wave + NW * q;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                acc[q][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (FULL || jt < NT) acc[q][g] = *(const f32x4*)(bsh + g * H + jt * 16 + fg * 4);
            }
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const uint4 hf = *(const uint4*)(hcur + frow * ROWB + (kc * 4 + fg) * 16);
#pragma unroll
            for (int q = 0; q < NJT; ++q) {
                uint4 w2;
                if (kc < KCL) w2 = *(const uint4*)(wn + (((wave * NJT + q) * KCL + kc) * 64 + lane) * 16);
                else w2 = wrn[q][kc - KCL < (KCRN > 0 ? KCRN : 1) ? kc - KCL : 0];
                mfma_chunk<bf16_t>(acc[q][0], wr[q][0][kc], hf);
                mfma_chunk<bf16_t>(acc[q][1], wr[q][1][kc], hf);
                mfma_chunk<bf16_t>(acc[q][2], w2, hf);
            }
        }
        // The next step's rows (requested at the top of this step, a matrix phase ago) and the previous step's stores are waited for HERE,
        // before this step's first store is issued: loads and stores share the counter, so a wait placed behind the tape stores below
        // (where the compiler would put it: in front of the LDS writes of gpre at the end of the step) waits for their acknowledgement too.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        f32x4 hp_keep[NJT];
#pragma unroll
        for (int q = 0; q < NJT; ++q) {
            const int jt = wave + NW * q;
            hp_keep[q] = (f32x4){hprev[q][0], hprev[q][1], hprev[q][2], hprev[q][3]};
            if (FULL || jt < NT) {
                const int j = jt * 16 + fg * 4;
                const bf16_t* grow = (const bf16_t*)(gtile + frow * GROW);
                const f32x4 gr = load4(grow + j), gu = load4(grow + H + j), gn = load4(grow + 2 * H + j);
                f32x4 r4, u4, n4, q4, h4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = fast_sigmoid(gr[e] + acc[q][0][e]);
                    const float u = fast_sigmoid(gu[e] + acc[q][1][e]);
                    const float qq = acc[q][2][e];
                    const float n = fast_tanh(gn[e] + r * qq);
                    const float hn = (1.f - u) * n + u * hprev[q][e];
                    hprev[q][e] = hn;
                    r4[e] = r; u4[e] = u; n4[e] = n; q4[e] = qq; h4[e] = hn;
                }
                store4((bf16_t*)(hnext + frow * ROWB) + j, h4);
                bf16_t* tp = tape + (((tape_bt + t) * NW + wave) * Cfg::SLOTS + 2 * q) * 64 * 8 + lane * 8;
                *(uint4*)tp = pack8(r4, u4);
                *(uint4*)(tp + 64 * 8) = pack8(n4, q4);
                if (b_ok && t == V - 1) *(f32x4*)(c_out + (long long)b * H + j) = h4;
            }
        }
        {
            bf16_t* tp = tape + (((tape_bt + t) * NW + wave) * Cfg::SLOTS + 2 * NJT) * 64 * 8 + lane * 8;
            *(uint4*)tp = pack8(hp_keep[0], hp_keep[NJT > 1 ? 1 : 0]);
        }
        __syncthreads();            // h_{t+1} tile complete; every wave is done with the input-projection tile of step t
        {
            // h_{t+1}: LDS tile -> HBM as whole rows (16 rows x H bf16)
            for (int i = tid; i < 16 * H / 8; i += NTHR) {
                const int rb = i / (H / 8), cc = i % (H / 8);
                if (b0 + rb < B)
                    *(uint4*)(Hall + ((long long)(b0 + rb) * (V + 1) + t + 1) * H + cc * 8) = *(const uint4*)(hnext + rb * ROWB + cc * 16);
            }
            if (t + 1 < V) {
#pragma unroll
                for (int u = 0; u < GPT; ++u) {
                    const int i = tid + u * NTHR;
                    if (i < GCHUNKS) {
                        const int rb = i / (3 * H / 8), cc = i % (3 * H / 8);
                        *(uint4*)(gtile + rb * GROW + cc * 16) = gpre[u];
                    }
                }
            }
            // the rows of step t + 2: requested here, a whole step (barrier, matrix phase) before the wait in front of step t + 1's stores
            if (t + 2 < V) {
#pragma unroll
                for (int u = 0; u < GPT; ++u) {
                    const int i = tid + u * NTHR;
                    if (i < GCHUNKS) {
                        const int rb = i / (3 * H / 8), cc = i % (3 * H / 8);
                        if (b0 + rb < B) gpre[u] = *(const uint4*)(Gi + ((long long)(b0 + rb) * V + t + 2) * 3 * H + cc * 8);
                    }
                }
            }
        }
        __syncthreads();            // input-projection tile of step t+1 visible
    }
}

template <int KC>
__global__ __launch_bounds__(64 * GRU_NW) void gru_bwd_res_kernel(const float* __restrict__ dc, const bf16_t* __restrict__ tape,
                                                                  const bf16_t* __restrict__ WTfrag, bf16_t* __restrict__ dG,
                                                                  int B, int V, int dbg) {
    typedef GruCfg<KC> Cfg;
    constexpr int H = Cfg::H, NT = Cfg::NT, NJT = Cfg::NJT, NW = GRU_NW, NTHR = 64 * NW;
    constexpr bool FULL = Cfg::FULL;
    constexpr int KC3 = 3 * KC, KCL = Cfg::KCL, KCR = KC3 - KCL;
    constexpr int ROWB = 4 * H * 2 + 16;                   // [dr | du | dn | dn*r]
    constexpr int WL_BYTES = NW * NJT * KCL * 1024;
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
        const int jt = wave + NW * q;
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
    const long long tape_bt = (long long)blockIdx.x * V;
    // tape of step t: slots (r,u) and (n,q) per tile + h_prev; prefetched one step ahead
    uint4 sv[Cfg::SLOTS], svn[Cfg::SLOTS];
    {
        const bf16_t* tp = tape + ((tape_bt + (V - 1)) * NW + wave) * Cfg::SLOTS * 64 * 8 + lane * 8;
#pragma unroll
        for (int k = 0; k < Cfg::SLOTS; ++k) { sv[k] = *(const uint4*)(tp + k * 64 * 8); svn[k] = sv[k]; }
    }
    __syncthreads();

    for (int t = V - 1; t >= 0; --t) {
        if (t > 0 && !(dbg & 2)) {
            const bf16_t* tp = tape + ((tape_bt + (t - 1)) * NW + wave) * Cfg::SLOTS * 64 * 8 + lane * 8;
#pragma unroll
            for (int k = 0; k < Cfg::SLOTS; ++k) svn[k] = *(const uint4*)(tp + k * 64 * 8);
        }
        float keep[NJT][4];
        f32x4 hp_all[2];
        unpack8(sv[2 * NJT], hp_all[0], hp_all[1]);
#pragma unroll
        for (int q = 0; q < NJT; ++q) {
            const int jt = wave + NW * q;
            if (FULL || jt < NT) {
                const int j = jt * 16 + fg * 4;
                f32x4 r4, u4, n4, q4;
                unpack8(sv[2 * q], r4, u4);
                unpack8(sv[2 * q + 1], n4, q4);
                const f32x4 hp = hp_all[q & 1];
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
                store4(grow + 2 * H + j, dn4);
                store4(grow + 3 * H + j, dnr4);
            }
        }
        __syncthreads();
        // the gradient tile goes to HBM as whole rows (16 rows x 4H bf16): read from LDS here, stored behind the matrix phase (below)
        constexpr int GCH = 16 * 4 * H / 8, GST = (GCH + NTHR - 1) / NTHR;
        uint4 gst[GST];
#pragma unroll
        for (int u = 0; u < GST; ++u) {
            const int i = tid + u * NTHR;
            gst[u] = make_uint4(0, 0, 0, 0);
            if (GCH % NTHR == 0 || i < GCH) gst[u] = *(const uint4*)(gcur + (i / (4 * H / 8)) * ROWB + (i % (4 * H / 8)) * 16);
        }
        f32x4 acc[NJT];
#pragma unroll
        for (int q = 0; q < NJT; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 4)) {
#pragma unroll
            for (int kc = 0; kc < KC3; ++kc) {
                // k-chunk kc of dgh = (dr, du, dn*r): columns [0, 2H) of the tile, then [3H, 4H)
                const int chunk = (kc < 2 * KC ? kc * 4 : (3 * H / 8) + (kc - 2 * KC) * 4) + fg;
                const uint4 gf = *(const uint4*)(gcur + frow * ROWB + chunk * 16);
#pragma unroll
                for (int q = 0; q < NJT; ++q) {
                    uint4 w;
                    if (kc < KCR) w = wr[q][kc < KCR ? kc : 0];
                    else w = *(const uint4*)(wl + (((wave * NJT + q) * KCL + (kc - KCR)) * 64 + lane) * 16);
                    mfma_chunk<bf16_t>(acc[q], w, gf);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NJT; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) dh[q][e] = keep[q][e] + acc[q][e];
        // The tape of step t - 1 (requested at the top of this step, a gate pass and a matrix phase ago) is waited for BEFORE this step's
        // stores are issued: loads and stores share one counter, and the wait the compiler would place in front of `sv = svn` below would
        // also wait for the acknowledgement of the stores issued just before it (gru_fwd_res_kernel has the same arrangement).
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), the other counters open
        asm volatile("" ::: "memory");
        if (!(dbg & 1)) {
#pragma unroll
            for (int u = 0; u < GST; ++u) {
                const int i = tid + u * NTHR;
                const int rb = i / (4 * H / 8), cc = i % (4 * H / 8);
                if ((GCH % NTHR == 0 || i < GCH) && b0 + rb < B) *(uint4*)(dG + ((long long)(b0 + rb) * V + t) * 4 * H + cc * 8) = gst[u];
            }
        }
#pragma unroll
        for (int k = 0; k < Cfg::SLOTS; ++k) sv[k] = svn[k];
        __syncthreads();
    }
}

// dst (fragment order, T) from a row-major f32 matrix.  Logical operand Wn[n][k], n < R, k < Kd:
//   transpose == 0: Wn[n][k] = src[n * ld + k];   transpose == 1: Wn[n][k] = src[k * ld + n]
// dst[((nt * KC + kc) * 64 + lane) * CH + e] = Wn[nt*16 + (lane&15)][kc*4*CH + (lane>>4)*CH + e]
template <typename T>
__global__ __launch_bounds__(256) void prep_frag_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int Kd,
                                                        long long ld, int transpose) {
    // one thread per lane-fragment: its CH elements (consecutive k) are read together and stored as ONE 16-byte piece (it was one element
    // per thread and iteration behind three 64-bit div / mod pairs: 27 - 42 us for the 0.2 M elements of W_hh)
    constexpr int CH = Elem<T>::CH;
    const int KC = Kd / (4 * CH);
    const int total = R * (Kd / CH);
    for (int f = blockIdx.x * 256 + threadIdx.x; f < total; f += gridDim.x * 256) {
        const int lane = f & 63, blk = f >> 6;
        const int kc = blk % KC, nt = blk / KC;
        const int n = nt * 16 + (lane & 15);
        const int k = kc * 4 * CH + (lane >> 4) * CH;
        __attribute__((aligned(16))) T out[CH];          // CH * sizeof(T) = 16 bytes for both storage types
#pragma unroll
        for (int e = 0; e < CH; ++e)
            out[e] = from_f32<T>(transpose ? src[(long long)(k + e) * ld + n] : src[(long long)n * ld + k + e]);
        *(uint4*)(dst + (long long)f * CH) = *(const uint4*)out;
    }
}

}  // namespace

// Debug / A-B switch: 1 = always use the weight-streaming kernels (set through cpc_gru_set_streaming).
int g_gru_force_streaming = 0;
int g_gru_debug = 0;             // timing experiments only (bit0: no gradient stores, bit1: no prefetch, bit2: no MFMA phase)
int g_gru_waves = 8;   // (unused since the tape layout fixes 8 waves)            // waves per workgroup of the weight-resident kernels at H = 256 (8 or 16)

static bool gru_ok(int B, int V, int H, int dtype) {
    const int ch = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if (B <= 0 || V <= 0 || H <= 0) return false;
    if (H % 16 || H % (4 * ch) || H / 16 > 4 * GRU_MAXJT) return false;
    return dtype == CPC_DTYPE_BF16 || dtype == CPC_DTYPE_F32;
}

long long gru_tape_elems(int B, int V, int H, int dtype) {
    if (dtype == CPC_DTYPE_BF16 && !g_gru_force_streaming && (H == 32 || H == 64 || H == 128 || H == 256))
        return gru_tape_elems_res(B, V, H / 32);
    return (long long)B * V * 5 * H;
}

int launch_gru_fwd(const void* Gi, const void* Wfrag, const float* bhh, void* Hall, void* tape, float* c_out, int B,
                   int V, int H, int dtype, hipStream_t stream, const float* h0) {
    if (!gru_ok(B, V, H, dtype)) return CPC_EINVAL;
    const int esz = dtype == CPC_DTYPE_BF16 ? 2 : 4;
    const size_t shm = 2 * 16 * (size_t)(H * esz + 16) + 3 * H * sizeof(float);
    dim3 grid((B + 15) / 16);
    if (dtype == CPC_DTYPE_BF16 && !g_gru_force_streaming && (H == 32 || H == 64 || H == 128 || H == 256)) {
#define GRU_F(KC) \
    hipLaunchKernelGGL((gru_fwd_res_kernel<KC>), grid, dim3(64 * GRU_NW), 0, stream, (const bf16_t*)Gi, (const bf16_t*)Wfrag, bhh, \
                       (bf16_t*)Hall, (bf16_t*)tape, c_out, B, V, h0)
        if (H == 256) GRU_F(8); else if (H == 128) GRU_F(4); else if (H == 64) GRU_F(2); else GRU_F(1);
#undef GRU_F
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((gru_fwd_kernel<bf16_t>), grid, dim3(256), shm, stream, (const bf16_t*)Gi, (const bf16_t*)Wfrag, bhh,
                           (bf16_t*)Hall, (bf16_t*)tape, c_out, B, V, H, h0);
    else
        hipLaunchKernelGGL((gru_fwd_kernel<float>), grid, dim3(256), shm, stream, (const float*)Gi, (const float*)Wfrag, bhh,
                           (float*)Hall, (float*)tape, c_out, B, V, H, h0);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_gru_bwd(const float* dc, const void* tape, const void* WTfrag, void* dG, int B, int V, int H, int dtype,
                   hipStream_t stream) {
    if (!gru_ok(B, V, H, dtype)) return CPC_EINVAL;
    const int esz = dtype == CPC_DTYPE_BF16 ? 2 : 4;
    const size_t shm = 16 * (size_t)(3 * H * esz + 16);
    if (shm > 64 * 1024) return CPC_EINVAL;
    dim3 grid((B + 15) / 16);
    if (dtype == CPC_DTYPE_BF16 && !g_gru_force_streaming && (H == 32 || H == 64 || H == 128 || H == 256)) {
#define GRU_B(KC) \
    hipLaunchKernelGGL((gru_bwd_res_kernel<KC>), grid, dim3(64 * GRU_NW), 0, stream, dc, (const bf16_t*)tape, (const bf16_t*)WTfrag, \
                       (bf16_t*)dG, B, V, g_gru_debug)
        if (H == 256) GRU_B(8); else if (H == 128) GRU_B(4); else if (H == 64) GRU_B(2); else GRU_B(1);
#undef GRU_B
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((gru_bwd_kernel<bf16_t>), grid, dim3(256), shm, stream, dc, (const bf16_t*)tape, (const bf16_t*)WTfrag,
                           (bf16_t*)dG, B, V, H);
    else
        hipLaunchKernelGGL((gru_bwd_kernel<float>), grid, dim3(256), shm, stream, dc, (const float*)tape, (const float*)WTfrag,
                           (float*)dG, B, V, H);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_prep_frag(const float* src, void* dst, int R, int Kd, long long ld, int transpose, int dtype, hipStream_t stream) {
    const int ch = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if (R <= 0 || Kd <= 0 || R % 16 || Kd % (4 * ch)) return CPC_EINVAL;
    const long long total = (long long)R * (Kd / ch);          // one thread per 16-byte fragment piece
    if (total > 0x7fffffffLL) return CPC_EINVAL;
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

// =====================================================================================================================
// Wasserstein gradient penalty through the GRU context (contrastive_estimation_training.py:144-158 differentiated through
// audio_model.py:66-77).  The penalty's parameter gradient is the gradient of a directional derivative of the summed scores
// (DESIGN.md section 8): reverse mode over the JOINT program (primal recurrence, tangent recurrence).  gru_gp_fwd_kernel runs
// the two recurrences together and keeps everything the reverse sweep needs; gru_gp_bwd_kernel carries two adjoints per hidden
// unit down the sequence: G, the adjoint of the summed scores (equal to the adjoint of the tangent variables, the tangent
// program being linear with the primal Jacobians), and N, the adjoint the primal variables gain through the coefficients of the
// tangent program (the second derivatives of sigmoid / tanh and the products r*q, z*h, (1-z)*n).  f32 only (the penalty runs in
// the exact-f32 mode); one workgroup per batch item, one thread per hidden unit, weights streamed from L2 every step: a
// parity path of ~B workgroups, not a tuned one (a penalty step costs about three plain steps anyway).
// tape f32 [B][V][10][H]: r, z, n, q = W_hn h + b_hn, h_{t-1}, tangents of the pre-activations of r and z, of q, of the
// pre-activation of n, and of h_{t-1}.
#define GRU_GP_SLOTS 10

__global__ __launch_bounds__(256) void gru_gp_fwd_kernel(const float* __restrict__ Gi, const float* __restrict__ GiT,
                                                         const float* __restrict__ WT, const float* __restrict__ bhh,
                                                         float* __restrict__ tape, float* __restrict__ ct_out, int V, int H) {
    extern __shared__ float gp_lds[];
    float* hs = gp_lds;          // h_{t-1}
    float* hts = gp_lds + H;     // its tangent
    const int b = blockIdx.x, j = threadIdx.x;
    const bool on = j < H;
    if (on) { hs[j] = 0.f; hts[j] = 0.f; }
    const float br = on ? bhh[j] : 0.f, bz = on ? bhh[H + j] : 0.f, bn = on ? bhh[2 * H + j] : 0.f;
    float h = 0.f, ht = 0.f;
    for (int t = 0; t < V; ++t) {
        __syncthreads();
        float ar = br, az = bz, q = bn, art = 0.f, azt = 0.f, qt = 0.f;
        if (on) {
            for (int i = 0; i < H; ++i) {
                const float* w = WT + (size_t)i * 3 * H + j;
                const float hv = hs[i], tv = hts[i];
                const float wr = w[0], wz = w[H], wn = w[2 * H];
                ar = fmaf(wr, hv, ar); az = fmaf(wz, hv, az); q = fmaf(wn, hv, q);
                art = fmaf(wr, tv, art); azt = fmaf(wz, tv, azt); qt = fmaf(wn, tv, qt);
            }
        }
        __syncthreads();
        if (on) {
            const size_t row = ((size_t)b * V + t) * 3 * H;
            ar += Gi[row + j]; az += Gi[row + H + j];
            art += GiT[row + j]; azt += GiT[row + H + j];
            const float r = 1.f / (1.f + expf(-ar)), z = 1.f / (1.f + expf(-az));
            const float n = tanhf(Gi[row + 2 * H + j] + r * q);
            const float rt = r * (1.f - r) * art, zt = z * (1.f - z) * azt;
            const float ant = GiT[row + 2 * H + j] + rt * q + r * qt;
            const float nt = (1.f - n * n) * ant;
            float* tp = tape + ((size_t)b * V + t) * GRU_GP_SLOTS * H + j;
            tp[0] = r; tp[H] = z; tp[2 * H] = n; tp[3 * H] = q; tp[4 * H] = h;
            tp[5 * H] = art; tp[6 * H] = azt; tp[7 * H] = qt; tp[8 * H] = ant; tp[9 * H] = ht;
            const float hn = (1.f - z) * n + z * h;
            ht = zt * (h - n) + (1.f - z) * nt + z * ht;
            h = hn;
            hs[j] = h; hts[j] = ht;
        }
    }
    if (on) ct_out[(size_t)b * H + j] = ht;
}

// dA f32 [B][V][8][H] = [d r_pre | d z_pre | d n_pre | d q] of the summed scores (the adjoints of the tangent pre-activations),
// then the same four for the second-order adjoint N.  W: weight_hh in the reference's layout [3H][H].
__global__ __launch_bounds__(256) void gru_gp_bwd_kernel(const float* __restrict__ dc, const float* __restrict__ tape,
                                                         const float* __restrict__ W, float* __restrict__ dA, int V, int H) {
    extern __shared__ float gp_lds[];
    float* dv = gp_lds;              // [3][H]: d r_pre, d z_pre, d q
    float* vv = gp_lds + 3 * H;      // [3][H]: the same of N
    const int b = blockIdx.x, i = threadIdx.x;
    const bool on = i < H;
    float G = on ? dc[(size_t)b * H + i] : 0.f, N = 0.f;
    for (int t = V - 1; t >= 0; --t) {
        float gz = 0.f, v_h = 0.f;
        if (on) {
            const float* tp = tape + ((size_t)b * V + t) * GRU_GP_SLOTS * H + i;
            const float r = tp[0], z = tp[H], n = tp[2 * H], q = tp[3 * H], hp = tp[4 * H];
            const float art = tp[5 * H], azt = tp[6 * H], qt = tp[7 * H], ant = tp[8 * H], htp = tp[9 * H];
            const float sr = r * (1.f - r), sz = z * (1.f - z), sn = 1.f - n * n;
            const float rt = sr * art, zt = sz * azt, nt = sn * ant;
            // adjoints of the summed scores
            const float d_n = G * (1.f - z), d_z = G * (hp - n);
            const float d_an = d_n * sn, d_r = d_an * q, d_q = d_an * r, d_ar = d_r * sr, d_az = d_z * sz;
            // what the primal variables gain through the tangent program's coefficients
            const float s_h = G * zt;
            const float s_n = -G * zt - 2.f * n * d_n * ant;
            const float s_z = G * (htp - nt) + d_z * (1.f - 2.f * z) * azt;
            const float s_q = d_an * rt;
            const float s_r = d_an * qt + d_r * (1.f - 2.f * r) * art;
            const float v_n = N * (1.f - z) + s_n, v_z = N * (hp - n) + s_z;
            v_h = N * z + s_h;
            const float v_an = v_n * sn, v_r = v_an * q + s_r, v_q = v_an * r + s_q, v_ar = v_r * sr, v_az = v_z * sz;
            float* o = dA + ((size_t)b * V + t) * 8 * H + i;
            o[0] = d_ar; o[H] = d_az; o[2 * H] = d_an; o[3 * H] = d_q;
            o[4 * H] = v_ar; o[5 * H] = v_az; o[6 * H] = v_an; o[7 * H] = v_q;
            dv[i] = d_ar; dv[H + i] = d_az; dv[2 * H + i] = d_q;
            vv[i] = v_ar; vv[H + i] = v_az; vv[2 * H + i] = v_q;
            gz = G * z;
        }
        __syncthreads();
        if (on) {
            float accG = gz, accN = v_h;
            for (int g = 0; g < 3; ++g)
                for (int j = 0; j < H; ++j) {
                    const float w = W[((size_t)g * H + j) * H + i];
                    accG = fmaf(w, dv[g * H + j], accG);
                    accN = fmaf(w, vv[g * H + j], accN);
                }
            G = accG; N = accN;
        }
        __syncthreads();
    }
}

int launch_gru_gp_fwd(const float* Gi, const float* GiT, const float* WT, const float* bhh, float* tape, float* ct_out, int B,
                      int V, int H, hipStream_t stream) {
    if (B <= 0 || V <= 0 || H <= 0 || H > 256) return CPC_EINVAL;
    hipLaunchKernelGGL(gru_gp_fwd_kernel, dim3(B), dim3((H + 63) / 64 * 64), 2 * H * sizeof(float), stream, Gi, GiT, WT, bhh, tape,
                       ct_out, V, H);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_gru_gp_bwd(const float* dc, const float* tape, const float* W, float* dA, int B, int V, int H, hipStream_t stream) {
    if (B <= 0 || V <= 0 || H <= 0 || H > 256) return CPC_EINVAL;
    hipLaunchKernelGGL(gru_gp_bwd_kernel, dim3(B), dim3((H + 63) / 64 * 64), 6 * H * sizeof(float), stream, dc, tape, W, dA, V, H);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
