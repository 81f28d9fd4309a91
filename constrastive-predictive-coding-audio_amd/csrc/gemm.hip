// MFMA GEMM family for the CPC-audio hot path (gfx950 / CDNA4).
//
// All activations live channels-last ("NLC": [item][position][channel]).  In that layout a strided,
// unpadded Conv1d IS a GEMM whose A operand has overlapping rows:
//     out[(b,t), co] = sum_{(j,c)} X[(b,t)*s*C + (j,c)] * W2[co][(j,c)]          (forward,  "NT")
//     dX[(b,q), (r,c)] = sum_{(d,co)} dY[(b,q-D+1)*C + (d,co)] * W3[(r,c)][(d,co)] (data grad, "NT")
//     dW2[(j,c), co]  = sum_{(b,t)} X[(b,t)*s*C + (j,c)] * dY[(b,t), co]          (weight grad, "TN")
// so no im2col buffer ever exists.  The same two kernels serve the GRU projections, the predictor and the
// InfoNCE score contraction.
//
//   gemm_nt : C[m][n] = epi( sum_k A[m][k] * Bt[n][k] )       both operands K-contiguous
//   gemm_tn : C[i][j] =      sum_m A[m][i] * B[m][j]           both operands reduction-strided
//
// Storage type T is bf16 (v_mfma_f32_16x16x32_bf16) or f32 (v_mfma_f32_16x16x4_f32, exact-f32 parity mode);
// accumulation is always f32.  128x128 output tile per 256-thread workgroup (4 waves, 64x64 each).
#include <algorithm>
#include "cpc_common.h"
#include "cpc_kernels.h"
#include <cstdlib>

// ---------------------------------------------------------------------------------------------------- NT
namespace {

constexpr int BM = 128, BN = 128;

// LDS tile: 128 rows x 128 bytes (8 chunks of 16 B); chunk index XOR-swizzled with (row & 7) so that the
// ds_read_b128 fragment reads (16 rows x same chunk) spread over all banks.
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <typename T, typename TO>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT p) {
    constexpr int CH = Elem<T>::CH;
    constexpr int BK = 8 * CH;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128];
    unsigned char* ldsA = lds;
    unsigned char* ldsB = lds + 128 * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: the blocks of one XCD (bid % 8) walk the N tiles of one M tile back to back, so the
    // A panel of that M tile is fetched into that XCD's L2 once.
    const int numM = (p.M + BM - 1) / BM, numN = (p.N + BN - 1) / BN;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int mt = (slot / numN) * 8 + xcd, nt = slot % numN;
    if (mt >= numM) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const T* Ab = (const T*)p.A + (long long)blockIdx.z * p.a_batch;
    const T* Bb = (const T*)p.Bt + (long long)blockIdx.z * p.b_batch;

    // Staging: 1024 chunks per operand tile, 4 per thread.  chunk c = tid + 256*i -> row = c >> 3, ch = c & 7.
    const int ch = tid & 7;
    long long a_off[4], b_off[4];
    bool a_ok[4], b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int m = m0 + row, n = n0 + row;
        a_ok[i] = m < p.M;
        b_ok[i] = n < p.N;
        a_off[i] = a_ok[i] ? row_off(m, p.a_rpi, p.a_item, p.lda) + ch * CH : 0;
        b_off[i] = b_ok[i] ? row_off(n, p.b_rpi, p.b_item, p.ldb) + ch * CH : 0;
    }
    uint4 ra[4], rb[4];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto gload = [&](int k0) {
        const bool kin = (k0 + ch * CH) < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = (a_ok[i] && kin) ? *(const uint4*)(Ab + a_off[i] + k0) : zero4;
            rb[i] = (b_ok[i] && kin) ? *(const uint4*)(Bb + b_off[i] + k0) : zero4;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            *(uint4*)(ldsA + lds_off(row, ch)) = ra[i];
            *(uint4*)(ldsB + lds_off(row, ch)) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4;
    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();          // previous tile's fragment reads are done
        lstore();
        __syncthreads();
        if (t + 1 < nk) gload((t + 1) * BK);   // next tile's global loads fly under this tile's MFMAs
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra_ = wm * 64 + i * 16 + frow;
                const int rb_ = wn * 64 + i * 16 + frow;
                fa[i] = *(const uint4*)(ldsA + lds_off(ra_, kk * 4 + fg));
                fb[i] = *(const uint4*)(ldsB + lds_off(rb_, kk * 4 + fg));
            }
            // Transposed product D[n][m]: the Bt rows are the MFMA "A" operand, the A rows its "B" operand, so that
            // a lane ends up holding 4 consecutive n for one m (vector stores along the contiguous output axis).
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mfma_chunk<T>(acc[i][j], fb[j], fa[i]);
        }
    }

    // Epilogue.  lane: m = m0 + wm*64 + i*16 + (lane&15);  n = n0 + wn*64 + j*16 + (lane>>4)*4 + {0..3}
    TO* Cb = (TO*)p.C + (long long)blockIdx.z * p.c_batch;
    const T* Mb = (const T*)p.mask;
    const bool relu = p.flags & GEMM_RELU;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + frow;
        if (m >= p.M) continue;
        const long long coff = row_off(m, p.c_rpi, p.c_item, p.ldc);
        const bool row_valid = (p.c_rpi == 0) || ((m % p.c_rpi) < p.c_valid);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fg * 4;
            if (n >= p.N) continue;      // a group straddling N stores zeros into the (caller-provided) pad columns
            f32x4 v = acc[i][j];
            if (p.bias) {
                const f32x4 bv = *(const f32x4*)(p.bias + n);
                v += bv;
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
            }
            if (Mb) {
                const f32x4 mk = load4(Mb + (long long)blockIdx.z * p.c_batch + coff + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
            }
            if (!row_valid) {
                if (p.flags & GEMM_SKIP_PAD_ROWS) continue;
                v = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            store4(Cb + coff + n, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------- TN
// bf16: 64 reduction rows per stage, LDS rows of 128 columns + 16 pad (288 B) so that the 8 rows one half-wave
// touches in a transposed read start 8 banks apart.
// f32 : 32 reduction rows per stage, LDS rows of 128 columns + 16 pad (576 B).
template <typename T> struct TnCfg;
template <> struct TnCfg<bf16_t> { static constexpr int BKM = 64, ROWB = 288, CPR = 16; };   // chunks per row
template <> struct TnCfg<float> { static constexpr int BKM = 32, ROWB = 576, CPR = 32; };

// Fragment for the 16 columns starting at tile column cb, k-step ks.
// bf16: ds_read_b64_tr_b16 hardware-transposed reads.  Within each 16-lane group the instruction reads a
// 4-row x 16-column block: lane 4q+p supplies the address of row q, columns 4p..4p+3, and lane i receives
// column i of the 4 rows.  Group g takes rows 4g..4g+3 (first read) and 16+4g..16+4g+3 (second read) of the
// 32-row k-step: the same permutation of the reduction index for both operands.
__device__ __forceinline__ uint4 tn_frag_bf16(const unsigned char* tile, int cb, int ks, int lane, bool use_tr) {
    const int g = lane >> 4, idx = lane & 15;
    if (use_tr) {
        const int q = idx >> 2, pp = idx & 3;
        const int r1 = ks * 32 + 4 * g + q;
        const unsigned char* a1 = tile + r1 * 288 + (cb + 4 * pp) * 2;
        const unsigned char* a2 = a1 + 16 * 288;
        s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
        s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a2);
        uint2 lo = __builtin_bit_cast(uint2, v1), hi = __builtin_bit_cast(uint2, v2);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    // reference path (scalar LDS reads), same element order
    unsigned int e[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int r = ks * 32 + (jj < 4 ? 4 * g + jj : 16 + 4 * g + (jj - 4));
        e[jj] = *(const unsigned short*)(tile + r * 288 + (cb + idx) * 2);
    }
    return make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN p) {
    constexpr int CH = Elem<T>::CH;
    constexpr int BKM = TnCfg<T>::BKM, ROWB = TnCfg<T>::ROWB, CPR = TnCfg<T>::CPR;
    constexpr bool IS_BF16 = sizeof(T) == 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BKM * ROWB];
    unsigned char* ldsA = lds;
    unsigned char* ldsB = lds + BKM * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int numJ = (p.J + 127) / 128;
    const int it = blockIdx.x / numJ, jt = blockIdx.x % numJ;
    const int i0 = it * 128, j0 = jt * 128;
    const int split = blockIdx.y;
    const int m_begin = split * p.m_chunk;
    const int m_end = min(p.M, m_begin + p.m_chunk);

    const T* Ab = (const T*)p.A + (long long)blockIdx.z * p.a_batch;
    const T* Bb = (const T*)p.B + (long long)blockIdx.z * p.b_batch;

    // staging: BKM rows x CPR chunks = 1024 chunks per operand, 4 per thread
    constexpr int RPP = 256 / CPR;           // rows covered per pass of 256 threads
    const int ch = tid % CPR, r0 = tid / CPR;
    const bool a_col_ok = (i0 + ch * CH) < p.I;
    const bool b_col_ok = (j0 + ch * CH) < p.J;
    uint4 ra[4], rb[4];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto gload = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mb + r0 + RPP * i;
            const bool ok = m < m_end;
            ra[i] = (ok && a_col_ok) ? *(const uint4*)(Ab + row_off(m, p.a_rpi, p.a_item, p.lda) + i0 + ch * CH) : zero4;
            rb[i] = (ok && b_col_ok) ? *(const uint4*)(Bb + row_off(m, p.b_rpi, p.b_item, p.ldb) + j0 + ch * CH) : zero4;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + RPP * i;
            *(uint4*)(ldsA + row * ROWB + ch * 16) = ra[i];
            *(uint4*)(ldsB + row * ROWB + ch * 16) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool use_tr = !(p.flags & GEMM_TN_NO_TR);
    const int fidx = lane & 15, fg = lane >> 4;
    if (m_begin < m_end) gload(m_begin);
    for (int mb = m_begin; mb < m_end; mb += BKM) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (mb + BKM < m_end) gload(mb + BKM);
        if constexpr (IS_BF16) {
#pragma unroll
            for (int ks = 0; ks < BKM / 32; ++ks) {
                uint4 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = tn_frag_bf16(ldsA, wi * 64 + i * 16, ks, lane, use_tr);
                    fb[i] = tn_frag_bf16(ldsB, wj * 64 + i * 16, ks, lane, use_tr);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mfma_chunk<bf16_t>(acc[i][j], fb[j], fa[i]);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BKM / 4; ++ks) {
                float fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = *(const float*)(ldsA + (ks * 4 + fg) * ROWB + (wi * 64 + i * 16 + fidx) * 4);
                    fb[i] = *(const float*)(ldsB + (ks * 4 + fg) * ROWB + (wj * 64 + i * 16 + fidx) * 4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
    }

    // D[row=j][col=i]: lane holds i = i0 + wi*64 + it*16 + (lane&15), j = j0 + wj*64 + jt*16 + (lane>>4)*4 + {0..3}
    TO* Cb = (TO*)p.C + (long long)blockIdx.z * p.c_batch + (long long)split * p.slab_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ii = i0 + wi * 64 + i * 16 + fidx;
        if (ii >= p.I) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jj = j0 + wj * 64 + j * 16 + fg * 4;
            if (jj >= p.J) continue;
            store4(Cb + row_off(ii, p.c_rpi, p.c_item, p.ldc) + jj, acc[i][j]);
        }
    }
}


// ------------------------------------------------------------------------------------------- fast paths
// Register-staged, LDS double-buffered variants with ONE barrier per stage, used whenever the reduction extent is a
// whole number of stages (NT: K % BK == 0).  The next stage's global loads are issued before the MFMAs of the current
// stage and written to the other LDS buffer after them (issue-early / write-late), so HBM/L2 latency hides under the
// matrix work.  Out-of-range tile rows / columns are CLAMPED to the last valid one instead of zero-filled wherever they
// only feed outputs that are never stored.

// XCD-aware tile order for the NT kernels.  Blocks are dealt round-robin over the 8 XCDs (block b -> XCD b % 8, used for
// speed only; each XCD has its own L2).  XCD x owns the M-panels x, x+8, ... and runs the N-tiles of one panel back to back,
// so an activation panel crosses the fabric once and its other N-tiles hit in that XCD's L2; the weight operand (2-4 MB, read
// by every panel) is served by L2 / Infinity Cache.  Measured against "one N-tile per XCD, weights L2-resident, panels re-read
// by every XCD" on the conv GEMMs of the headline config: forward (2 N-tiles) +2.4 %, data gradient (8 N-tiles) +5.9 %.
__host__ __device__ __forceinline__ long long nt_grid_blocks(int numM, int numN) { return 8LL * ((numM + 7) / 8) * numN; }
__device__ __forceinline__ void nt_tile_of_block(int bid, int numM, int numN, int& mt, int& nt) {
    const int xcd = bid & 7, slot = bid >> 3;
    mt = (slot / numN) * 8 + xcd;
    nt = slot % numN;
}

// Tile configuration: WM x WN waves, each owning a (TI*16) x (TJ*16) output sub-tile.
//   <2,2,4,4>: 128x128 tile, 256 threads, 2 x 32 KiB LDS  (two workgroups per CU)
//   <2,4,8,4>: 256x256 tile, 512 threads, 2 x 64 KiB LDS  (one workgroup per CU): half the LDS write traffic and 3/4 of
//              the LDS read traffic per MFMA of the small tile — the small tile is LDS-bound (ds_write_b128 ~79 B/clk/CU).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ds_read_b128 the compiler does not see (see the DMA variant below); OFF is the instruction's immediate offset.
template <int OFF>
__device__ __forceinline__ u32x4 lds_read16_asm(unsigned addr) {
    u32x4 d;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
    return d;
}
__device__ __forceinline__ uint4 as_uint4(const u32x4& v) { return make_uint4(v[0], v[1], v[2], v[3]); }

// DMA = true: operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write traffic —
// the register-staged variant is bound by the ds_write_b128 rate).  hipcc drains vmcnt(0) in front of every LDS read it
// can see while an LDS-DMA is in flight, which would serialise load and MFMA; so here the fragment reads are inline-asm
// ds_read_b128 with hand-counted lgkmcnt waits (each wait statement names the fragments it releases as "+v" operands
// and is followed by sched_barrier(0), so no MFMA can be scheduled above it).
__device__ __forceinline__ uint4 tn_frag_rb(const unsigned char* tile, int rowb, int cb, int ks, int lane);

// DIRECT = true (bf16 in / out, LDS-DMA): the epilogue goes from the accumulator registers straight to memory — no LDS image,
// no barrier, and the workgroup ends with its stores in flight, so the CU's next workgroup starts its first loads while they
// drain (the LDS-staged epilogue holds the CU until a 32 MB-per-round store burst has been accepted).  The B rows are dealt to
// the MFMA rows in a permuted order — applied to the SOURCE rows of the LDS-DMA, so the LDS image and the fragment reads are
// unchanged — such that a lane holds 8 consecutive output columns of a row in the accumulators of the tile pair (2p, 2p+1):
// 16-byte stores, 64 contiguous bytes per row and instruction.  Same sums in the same order as the other variants.
// tile row r of the B operand (consumed by wave column r / 64 as MFMA tile j = (r / 16) % 4, MFMA row q = r % 16) holds output
// column 64 (r / 64) + 32 (j / 2) + 8 (q / 4) + 4 (j % 2) + q % 4 of the tile
__device__ __forceinline__ int direct_b_col(int r) {
    const int j = (r >> 4) & 3, q = r & 15;
    return (r & ~63) + 32 * (j >> 1) + 8 * (q >> 2) + 4 * (j & 1) + (q & 3);
}

// Output store of the fast kernels: written through the XCD's L2 at once (default, GEMM_WT_SYSTEM) or plain.  A large output left dirty
// in the L2s is written back when the kernel ends, before the next one may start; with the stores written through as they are issued
// the step takes 4.469 instead of 4.478-4.490 ms (tools/wt_ab.py, interleaved A/B on one box; agent scope: 4.474-4.479).
__device__ __forceinline__ void store_out16(uint4* dst, uint4 v, int flags) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u d = {v.x, v.y, v.z, v.w};
    if (flags & GEMM_WT_SYSTEM) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(d) : "memory");
    else if (flags & GEMM_WT_AGENT) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(d) : "memory");
    else *dst = v;
}

template <typename T, typename TO, int WM, int WN, int TI, int TJ, bool DMA, bool C1 = false, bool DIRECT = false, int DBG = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_fast_kernel(GemmNT p) {
    // DBG: timing probes for tools/nt_probe.py (cpc_debug_set key 4; never in the product path, the results are garbage):
    // 1 = K loop without its LDS-DMA requests, 2 = without its MFMAs, 16 = only the B tile is requested, 32 = the A operand stored
    // stage-major ([K / 64][M][64], a_item = elements per stage: a stage's A tile is then ONE dense 32 KiB range)
    static_assert(!DIRECT || (DMA && !C1 && sizeof(T) == 2 && sizeof(TO) == 2 && TJ == 4), "direct epilogue: bf16 LDS-DMA variants");
    constexpr int CH = Elem<T>::CH;
    constexpr int BK = 8 * CH;
    constexpr int TBM = WM * TI * 16, TBN = WN * TJ * 16, NTHR = 64 * WM * WN;
    constexpr int ATILE = TBM * 128, BTILE = TBN * 128, STAGE = ATILE + BTILE;
    constexpr int NA = TBM * 8 / NTHR, NB = TBN * 8 / NTHR, RSTEP = NTHR / 8;     // staged chunks per thread, row step
    constexpr int EPI_RS = TBN * 2 + 16;                                            // bf16 output tile row stride in LDS
    constexpr int EPI_BYTES = sizeof(TO) == 2 ? TBM * EPI_RS + (C1 ? 2 * TBM * 32 : 0) : 0;
    constexpr int CORE_BYTES = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
    // 256x256 bf16 LDS-staged epilogues: room for the tile's sign-bit mask (GemmNT::mask_bits), TBM rows x TBN / 8 bytes, fetched by
    // LDS-DMA before the K loop starts so that the epilogue finds it in LDS
    constexpr bool BITS_IN_LDS = DMA && !DIRECT && sizeof(TO) == 2 && TI == 8 && TBN == 256;
    constexpr int BITS_OFF = CORE_BYTES;
    // ... and, without the fused layer-1 epilogue, for the cross-thread reduction of the tile's column sums (GemmNT::colsum_slabs)
    constexpr bool CS_IN_LDS = BITS_IN_LDS && !C1;
    constexpr int CS_OFF = CORE_BYTES + (BITS_IN_LDS ? TBM * (TBN / 8) : 0);
    constexpr int LDS_BYTES = CS_OFF + (CS_IN_LDS ? 16 * (TBN / 8) * 8 * 4 : 0);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int numM = (p.M - p.m_off + TBM - 1) / TBM, numN = (p.N + TBN - 1) / TBN;
    int mt, nt;
    nt_tile_of_block(blockIdx.x, numM, numN, mt, nt);
    if (mt >= numM) return;
    const int m0 = p.m_off + mt * TBM, n0 = nt * TBN;

    const T* Ab = (const T*)p.A + (long long)blockIdx.z * p.a_batch;
    const T* Bb = (const T*)p.Bt + (long long)blockIdx.z * p.b_batch;
    // GemmNT::k_ranges: the K stages this tile's rows need (the union over the bands / items it touches)
    int nk_tile = p.K / (8 * Elem<T>::CH), k_lo = 0;
    if (p.k_ranges) {
        const int per = p.a_rpi * (p.a_rpi2 > 0 ? p.a_rpi2 : 1);
        const int i_lo = m0 / per, i_hi = min(m0 + TBM - 1, p.M - 1) / per;
        int lo = p.k_ranges[2 * i_lo], hi = p.k_ranges[2 * i_lo + 1];
        for (int i = i_lo + 1; i <= i_hi; ++i) {
            lo = min(lo, p.k_ranges[2 * i]);
            hi = max(hi, p.k_ranges[2 * i + 1]);
        }
        if (p.k_taps > 1) {
            k_lo = lo;                            // gathered rows (storage order over the pieces): the loop starts inside piece lo / stages-per-piece
        } else {
            Ab += (long long)lo * (8 * Elem<T>::CH);
            Bb += (long long)lo * (8 * Elem<T>::CH);
        }
        nk_tile = hi - lo;
    }
    if constexpr (DMA && TI == 8) {
        // start stagger of the first round of workgroups (GemmNT::stagger)
        if (p.stagger > 0 && blockIdx.x < 256 && blockIdx.z == 0) {
            const int nsleep = ((blockIdx.x >> 3) & 7) * p.stagger;
            for (int i = 0; i < nsleep; ++i) __builtin_amdgcn_s_sleep(64);
        }
    }

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4;
    int offA[2][TI], offB[2][TJ];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < TI; ++i) offA[kk][i] = lds_off(wm * TI * 16 + i * 16 + frow, kk * 4 + fg);
#pragma unroll
        for (int j = 0; j < TJ; ++j) offB[kk][j] = ATILE + lds_off(wn * TJ * 16 + j * 16 + frow, kk * 4 + fg);
    }
    const int nk = nk_tile;
    // K visiting order (GemmNT::k_taps): element offset of stage 1, and of the stage the loop fetches next (kb + kj * tstride)
    const int taps = p.k_taps > 1 ? p.k_taps : 1;
    const long long tstride = p.k_taps > 1 ? p.k_tap_stride : 0;
    // (GemmNT::k_tap_stride_a: the A operand's taps lie further apart than the K axis says — rows gathered from a grid, see there)
    const long long tstride_a = p.k_taps > 1 && p.k_tap_stride_a ? p.k_tap_stride_a : tstride;
    // Position of the next stage to request: tap kj, offset kb within the tap.  Overlapped rows (launcher-detected) are visited tap-innermost
    // (kj runs fastest); rows gathered from separate pieces (GemmNT::k_taps_linear, taps given by the caller) in storage order (kb runs
    // fastest), which is the order k_ranges counts stages in.
    const bool klin = taps > 1 && p.k_taps_linear;
    int kj = 0;
    long long kb = 0;
    if (klin && k_lo) {
        const int spp = (int)(tstride / BK);
        kj = k_lo / spp;
        kb = (long long)(k_lo % spp) * BK;
    }
#define NT_KSTEP()                                                       \
    do {                                                                 \
        if (klin) { kb += BK; if (kb == tstride) { kb = 0; ++kj; } }     \
        else if (++kj == taps) { kj = 0; kb += BK; }                     \
    } while (0)
    const long long koff0 = kb + kj * tstride, koff0a = kb + kj * tstride_a;
    NT_KSTEP();
    const long long koff1 = kb + kj * tstride, koff1a = kb + kj * tstride_a;
    NT_KSTEP();
    if constexpr (!DMA) {
        // staging: thread -> chunk tid&7 of tile rows (tid>>3) + RSTEP*i, i = 0..3 (rows clamped into range).  Named scalars
        // on purpose: arrays here end up in scratch / LDS-promoted allocas with hipcc 7.2.
        static_assert(NA == 4 && NB == 4, "staging code below is written for 4 chunks per operand per thread");
        const int ch = tid & 7, srow = tid >> 3;
        const T* ga0 = Ab + row_off2(min(m0 + srow, p.M - 1), p.a_rpi, p.a_item, p.lda, p.a_rpi2, p.a_item2) + ch * CH;
        const T* ga1 = Ab + row_off2(min(m0 + srow + RSTEP, p.M - 1), p.a_rpi, p.a_item, p.lda, p.a_rpi2, p.a_item2) + ch * CH;
        const T* ga2 = Ab + row_off2(min(m0 + srow + 2 * RSTEP, p.M - 1), p.a_rpi, p.a_item, p.lda, p.a_rpi2, p.a_item2) + ch * CH;
        const T* ga3 = Ab + row_off2(min(m0 + srow + 3 * RSTEP, p.M - 1), p.a_rpi, p.a_item, p.lda, p.a_rpi2, p.a_item2) + ch * CH;
        const T* gb0 = Bb + row_off(min(n0 + srow, p.N - 1), p.b_rpi, p.b_item, p.ldb) + ch * CH;
        const T* gb1 = Bb + row_off(min(n0 + srow + RSTEP, p.N - 1), p.b_rpi, p.b_item, p.ldb) + ch * CH;
        const T* gb2 = Bb + row_off(min(n0 + srow + 2 * RSTEP, p.N - 1), p.b_rpi, p.b_item, p.ldb) + ch * CH;
        const T* gb3 = Bb + row_off(min(n0 + srow + 3 * RSTEP, p.N - 1), p.b_rpi, p.b_item, p.ldb) + ch * CH;
        // RSTEP is a multiple of 8, so the swizzle term (row & 7) is the same for the 4 rows: one offset + constants
        const int so = lds_off(srow, ch);
        uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define NT_GLOAD(ka, k0)                                                               \
        do {                                                                               \
            ra0 = *(const uint4*)(ga0 + (ka)); rb0 = *(const uint4*)(gb0 + (k0));          \
            ra1 = *(const uint4*)(ga1 + (ka)); rb1 = *(const uint4*)(gb1 + (k0));          \
            ra2 = *(const uint4*)(ga2 + (ka)); rb2 = *(const uint4*)(gb2 + (k0));          \
            ra3 = *(const uint4*)(ga3 + (ka)); rb3 = *(const uint4*)(gb3 + (k0));          \
        } while (0)
#define NT_LSTORE(base)                                                                \
        do {                                                                               \
            unsigned char* da = (base) + so;                                               \
            unsigned char* db = da + ATILE;                                                \
            *(uint4*)(da) = ra0;                   *(uint4*)(db) = rb0;                    \
            *(uint4*)(da + RSTEP * 128) = ra1;     *(uint4*)(db + RSTEP * 128) = rb1;      \
            *(uint4*)(da + 2 * RSTEP * 128) = ra2; *(uint4*)(db + 2 * RSTEP * 128) = rb2;  \
            *(uint4*)(da + 3 * RSTEP * 128) = ra3; *(uint4*)(db + 3 * RSTEP * 128) = rb3;  \
        } while (0)

        NT_GLOAD(koff0a, koff0);
        NT_LSTORE(lds);
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const unsigned char* cur = lds + (t & 1) * STAGE;
            const bool more = t + 1 < nk;
            if (more) {
                if (t == 0) NT_GLOAD(koff1a, koff1);
                else {
                    NT_GLOAD(kb + kj * tstride_a, kb + kj * tstride);
                    NT_KSTEP();
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fa[TI], fb[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) fa[i] = *(const uint4*)(cur + offA[kk][i]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) fb[j] = *(const uint4*)(cur + offB[kk][j]);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) mfma_chunk<T>(acc[i][j], fb[j], fa[i]);
            }
            if (more) NT_LSTORE(lds + ((t + 1) & 1) * STAGE);
            __syncthreads();
        }
#undef NT_GLOAD
#undef NT_LSTORE

    } else {
        static_assert(TJ == 4 && (TI == 4 || TI == 8), "fragment wait statements below are written for these tiles");
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        // LDS-DMA staging: wave-instruction j = wave*4 + i fills tile rows 8j .. 8j+7 (1 KiB, lane-linear); lane -> row
        // 8j + (lane>>3), LDS chunk lane&7, SOURCE chunk (lane&7) ^ (row&7) = (lane&7) ^ (lane>>3)  (swizzle on the source)
        const int srow = lane >> 3, sch = (lane & 7) ^ (lane >> 3);
        const int r0 = (wave_u * 4) * 8 + srow;
        const T* ga0 = Ab + row_off2(min(m0 + r0, p.M - 1), p.a_rpi, p.a_item, p.lda, p.a_rpi2, p.a_item2) + sch * CH;
        const T* ga1 = Ab + row_off2(min(m0 + r0 + 8, p.M - 1), p.a_rpi, p.a_item, p.lda, p.a_rpi2, p.a_item2) + sch * CH;
        const T* ga2 = Ab + row_off2(min(m0 + r0 + 16, p.M - 1), p.a_rpi, p.a_item, p.lda, p.a_rpi2, p.a_item2) + sch * CH;
        const T* ga3 = Ab + row_off2(min(m0 + r0 + 24, p.M - 1), p.a_rpi, p.a_item, p.lda, p.a_rpi2, p.a_item2) + sch * CH;
        const int br0 = DIRECT ? direct_b_col(r0) : r0, br1 = DIRECT ? direct_b_col(r0 + 8) : r0 + 8;
        const int br2 = DIRECT ? direct_b_col(r0 + 16) : r0 + 16, br3 = DIRECT ? direct_b_col(r0 + 24) : r0 + 24;
        const T* gb0 = Bb + row_off(min(n0 + br0, p.N - 1), p.b_rpi, p.b_item, p.ldb) + sch * CH;
        const T* gb1 = Bb + row_off(min(n0 + br1, p.N - 1), p.b_rpi, p.b_item, p.ldb) + sch * CH;
        const T* gb2 = Bb + row_off(min(n0 + br2, p.N - 1), p.b_rpi, p.b_item, p.ldb) + sch * CH;
        const T* gb3 = Bb + row_off(min(n0 + br3, p.N - 1), p.b_rpi, p.b_item, p.ldb) + sch * CH;
        // (timing probe DBG 64: the B operand as if stored stage-major, [N / 256][K / 64][256][64] — a stage's B tile one dense 32 KiB)
        const T* gd0 = Bb + (long long)(n0 / 256) * 256 * p.K + br0 * 64 + sch * CH;
        const T* gd1 = Bb + (long long)(n0 / 256) * 256 * p.K + br1 * 64 + sch * CH;
        const T* gd2 = Bb + (long long)(n0 / 256) * 256 * p.K + br2 * 64 + sch * CH;
        const T* gd3 = Bb + (long long)(n0 / 256) * 256 * p.K + br3 * 64 + sch * CH;
        typedef __attribute__((address_space(3))) unsigned char lds_byte;
        lds_byte* const lds3 = (lds_byte*)lds;
        const unsigned wdst = wave_u * 4096;                   // this wave's 4 KiB slice of an operand tile
#define NT_DMA1(g, dst)                                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g),                         \
                                     (__attribute__((address_space(3))) void*)(lds3 + (dst)), 16, 0, 0)
#define NT_DMA_STAGE(buf, ka, k0)                                                                                \
    do {                                                                                                         \
        const unsigned da = (buf) * STAGE + wdst, db = da + ATILE;                                               \
        NT_DMA1(ga0 + ((DBG & 32) ? (long long)((buf) / taps) * p.a_item + ((buf) % taps) * 64 : (ka)), da);            \
        NT_DMA1(ga1 + ((DBG & 32) ? (long long)((buf) / taps) * p.a_item + ((buf) % taps) * 64 : (ka)), da + 1024);     \
        NT_DMA1(ga2 + ((DBG & 32) ? (long long)((buf) / taps) * p.a_item + ((buf) % taps) * 64 : (ka)), da + 2048);     \
        NT_DMA1(ga3 + ((DBG & 32) ? (long long)((buf) / taps) * p.a_item + ((buf) % taps) * 64 : (ka)), da + 3072);     \
        if constexpr (DBG & 64) {                                                                                \
            NT_DMA1(gd0 + (long long)(k0) * 256, db);        NT_DMA1(gd1 + (long long)(k0) * 256, db + 1024);    \
            NT_DMA1(gd2 + (long long)(k0) * 256, db + 2048); NT_DMA1(gd3 + (long long)(k0) * 256, db + 3072);    \
        } else {                                                                                                 \
            NT_DMA1(gb0 + (k0), db);        NT_DMA1(gb1 + (k0), db + 1024);                                      \
            NT_DMA1(gb2 + (k0), db + 2048); NT_DMA1(gb3 + (k0), db + 3072);                                      \
        }                                                                                                        \
    } while (0)
        const unsigned lds_u32 = (unsigned)(unsigned long long)(lds3);
        // fragment addresses: rows i*16 apart differ by 2048 B with the same swizzle term -> one base per (operand, kk)
        const unsigned aA0 = lds_u32 + offA[0][0], aA1 = lds_u32 + offA[1][0];
        const unsigned aB0 = lds_u32 + offB[0][0], aB1 = lds_u32 + offB[1][0];

#define NT_READ_SET(fa, fb, aA, aB, cur)                                                                         \
    do {                                                                                                         \
        fb[0] = lds_read16_asm<0>(aB + (cur)); fb[1] = lds_read16_asm<2048>(aB + (cur));                         \
        fb[2] = lds_read16_asm<4096>(aB + (cur)); fb[3] = lds_read16_asm<6144>(aB + (cur));                      \
        fa[0] = lds_read16_asm<0>(aA + (cur)); fa[1] = lds_read16_asm<2048>(aA + (cur));                         \
        fa[2] = lds_read16_asm<4096>(aA + (cur)); fa[3] = lds_read16_asm<6144>(aA + (cur));                      \
        if constexpr (TI == 8) {                                                                                 \
            fa[4] = lds_read16_asm<8192>(aA + (cur)); fa[5] = lds_read16_asm<10240>(aA + (cur));                 \
            fa[6] = lds_read16_asm<12288>(aA + (cur)); fa[7] = lds_read16_asm<14336>(aA + (cur));                \
        }                                                                                                        \
    } while (0)
        // Software pipeline over the two k-halves of a stage (fragment register sets 0 and 1), with every LDS read and every
        // LDS-DMA piece issued BETWEEN two MFMAs instead of in bursts (an in-order wave that issues 8 DMA pieces + 12 reads in
        // a row keeps the matrix pipe idle for ~1k cycles, and after a barrier its SIMD partner does the same at the same
        // time):
        //   block 0 (t): MFMA(set 0, stage t), one read of set 1 (stage t) after each of the first TI+TJ MFMAs
        //   wait set 1, DMA(t+1) landed, barrier
        //   block 1 (t): MFMA(set 1, stage t); after the first 8 MFMAs one DMA piece of stage t+2 each (into the slot of
        //                stage t: all its fragments are in registers everywhere), after the next TI+TJ one read of set 0 of
        //                stage t+1 each
        //   wait set 0
        // sched_barrier(0) after every MFMA pins that order.
#define NT_READ1(idx, fa, fb, aA, aB, cur)                                                                       \
    do {                                                                                                         \
        switch (idx) {                                                                                           \
        case 0: fb[0] = lds_read16_asm<0>(aB + (cur)); break;                                                    \
        case 1: fb[1] = lds_read16_asm<2048>(aB + (cur)); break;                                                 \
        case 2: fb[2] = lds_read16_asm<4096>(aB + (cur)); break;                                                 \
        case 3: fb[3] = lds_read16_asm<6144>(aB + (cur)); break;                                                 \
        case 4: fa[0] = lds_read16_asm<0>(aA + (cur)); break;                                                    \
        case 5: fa[1] = lds_read16_asm<2048>(aA + (cur)); break;                                                 \
        case 6: fa[2] = lds_read16_asm<4096>(aA + (cur)); break;                                                 \
        case 7: fa[3] = lds_read16_asm<6144>(aA + (cur)); break;                                                 \
        case 8: if constexpr (TI == 8) fa[4] = lds_read16_asm<8192>(aA + (cur)); break;                          \
        case 9: if constexpr (TI == 8) fa[5] = lds_read16_asm<10240>(aA + (cur)); break;                         \
        case 10: if constexpr (TI == 8) fa[6] = lds_read16_asm<12288>(aA + (cur)); break;                        \
        case 11: if constexpr (TI == 8) fa[7] = lds_read16_asm<14336>(aA + (cur)); break;                        \
        default: break;                                                                                          \
        }                                                                                                        \
    } while (0)
#define NT_DMA_PIECE(idx, buf, ka, k0)                                                                           \
    do {                                                                                                         \
        const unsigned da = (buf) * STAGE + wdst, db = da + ATILE;                                               \
        switch (idx) {                                                                                           \
        case 0: if constexpr (!(DBG & 16)) NT_DMA1(ga0 + ((DBG & 32) ? kA_ : (ka)), da); break;                                       \
        case 1: NT_DMA1(gb0 + (k0), db); break;                                                                  \
        case 2: if constexpr (!(DBG & 16)) NT_DMA1(ga1 + ((DBG & 32) ? kA_ : (ka)), da + 1024); break;                                \
        case 3: NT_DMA1(gb1 + (k0), db + 1024); break;                                                           \
        case 4: if constexpr (!(DBG & 16)) NT_DMA1(ga2 + ((DBG & 32) ? kA_ : (ka)), da + 2048); break;                                \
        case 5: NT_DMA1(gb2 + (k0), db + 2048); break;                                                           \
        case 6: if constexpr (!(DBG & 16)) NT_DMA1(ga3 + ((DBG & 32) ? kA_ : (ka)), da + 3072); break;                                \
        case 7: NT_DMA1(gb3 + (k0), db + 3072); break;                                                           \
        default: break;                                                                                          \
        }                                                                                                        \
    } while (0)
#define NT_WAIT_SET(fa, fb)                                                                                      \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if constexpr (TI == 8)                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                  \
                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), \
                           "+v"(fa[7]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));                    \
        else                                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                  \
                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), \
                           "+v"(fb[3]));                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
        constexpr int NFRAG = TI + TJ;
        // one iteration; DO_DMA: stage t+2 exists, DO_READ: stage t+1 exists (compile-time so that the steady-state body has
        // no branches between the MFMAs)
#define NT_ITER(DO_DMA, DO_READ)                                                                                 \
    do {                                                                                                         \
        const unsigned cur = (t & 1) * STAGE, nxt = ((t + 1) & 1) * STAGE;                                       \
        const long long k2 = kb + kj * tstride, k2a = kb + kj * tstride_a;                                       \
        /* DBG 32: A stored chunk-major, [K / (64 taps)][rows][64]; the taps of a chunk are 64 elements (one row) apart */ \
        const long long kA_ = (long long)((t + 2) / taps) * p.a_item + (long long)((t + 2) % taps) * 64;         \
        (void)kA_;                                                                                               \
        if (DO_DMA) NT_KSTEP();                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) {                                                         \
            _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                     \
                if constexpr (!(DBG & 2)) mfma_chunk<T>(acc[i][j], as_uint4(fb0[j]), as_uint4(fa0[i]));          \
                NT_READ1(i * TJ + j, fa1, fb1, aA1, aB1, cur);                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
        }                                                                                                        \
        NT_WAIT_SET(fa1, fb1);                                                                                   \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
        __syncthreads();                                                                                         \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) {                                                         \
            _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                     \
                if constexpr (!(DBG & 2)) mfma_chunk<T>(acc[i][j], as_uint4(fb1[j]), as_uint4(fa1[i]));          \
                if constexpr (!(DBG & 1)) { if (DO_DMA) NT_DMA_PIECE(i * TJ + j, t & 1, k2a, k2); }                   \
                if (DO_READ) NT_READ1(i * TJ + j - (DO_DMA ? 8 : 0), fa0, fb0, aA0, aB0, nxt);                   \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
        }                                                                                                        \
        if (DO_READ) NT_WAIT_SET(fa0, fb0);                                                                      \
    } while (0)
        static_assert(TI * TJ >= 8 + NFRAG, "block 1 must have room for 8 DMA pieces and one fragment set");
        if constexpr (BITS_IN_LDS) {
            if (p.mask_bits) {
                // tile rows r, 32 bytes each: one DMA instruction = 8 rows x (8 lanes x 4 B); wave w fetches rows 32 w .. 32 w + 31.
                // Older than every stage request of this tile, so the first vmcnt(0) of the K loop covers it.
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = (wave_u * 4 + k) * 8 + (lane >> 3);
                    const int m = min(m0 + r, p.M - 1);
                    const unsigned char* src = p.mask_bits + (((long long)blockIdx.z * p.c_batch + row_off2(m, p.c_rpi, p.c_item, p.ldc, p.c_rpi2, p.c_item2) + n0) >> 3) + (lane & 7) * 4;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(lds3 + BITS_OFF + (wave_u * 4 + k) * 256), 4, 0, 0);
                }
            }
        }
        NT_DMA_STAGE(0, koff0a, koff0);
        if constexpr (C1) {
            // Fused layer-1 weight gradient: beside the tile image its epilogue needs an image of the waveform windows of the tile's
            // rows.  That image lives BEHIND the K loop's stage buffers, so it is built here, behind the requests of the first stage and before the loop
            // (its global loads and the hi / lo split cost the epilogue nothing): xw[2][TBM][16] bf16 = (hi, lo) parts of
            // x[b][t*stride + j] for slots j < kw, 1.0 in slot kw (bias gradient), 0 elsewhere / for rows that do not count
            static_assert(NTHR == 2 * TBM, "two threads per tile row");
            unsigned char* xw = lds + TBM * EPI_RS;
            const int r = tid >> 1, half = tid & 1;
            const int m = m0 + r;
            const int cinN = p.N / p.c1_sub, rsel = n0 / cinN;
            bool ok = m < p.M;
            long long xo = 0;
            if (ok) {
                const int b = m / p.c1_rpi, t = (p.c1_row0 + m % p.c1_rpi) * p.c1_sub + rsel;
                ok = t < p.c1_valid;
                xo = (long long)b * p.c1_ldx + (long long)t * p.c1_stride;
            }
            unsigned hw[4], lw[4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                unsigned short h2[2], l2[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = half * 8 + e2 * 2 + u;
                    float xv = 0.f;
                    if (ok) xv = j < p.c1_kw ? p.c1_x[xo + j] : (j == p.c1_kw ? 1.f : 0.f);
                    const bf16_t hb = (bf16_t)xv;
                    const bf16_t lb = (bf16_t)(xv - (float)hb);
                    h2[u] = __builtin_bit_cast(unsigned short, hb);
                    l2[u] = __builtin_bit_cast(unsigned short, lb);
                }
                hw[e2] = (unsigned)h2[0] | ((unsigned)h2[1] << 16);
                lw[e2] = (unsigned)l2[0] | ((unsigned)l2[1] << 16);
            }
            *(uint4*)(xw + r * 32 + half * 16) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            *(uint4*)(xw + TBM * 32 + r * 32 + half * 16) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nk > 1) NT_DMA_STAGE(1, koff1a, koff1);
        u32x4 fa0[TI], fb0[TJ], fa1[TI], fb1[TJ];
        NT_READ_SET(fa0, fb0, aA0, aB0, 0u);
        NT_WAIT_SET(fa0, fb0);
        int t = 0;
        for (; t + 2 < nk; ++t) NT_ITER(true, true);
        if (t + 1 < nk) {
            NT_ITER(false, true);
            ++t;
        }
        NT_ITER(false, false);
    }


    TO* Cb = (TO*)p.C + (long long)blockIdx.z * p.c_batch;
    const T* Mb = (const T*)p.mask;
    const bool relu = p.flags & GEMM_RELU;
    if constexpr (DIRECT) {
        // lane: rows m0 + (wm*TI + i)*16 + frow, columns n0 + wn*64 + 32 pr + 8 fg + (0..7) from acc[i][2 pr] | acc[i][2 pr + 1]
        const int nb = n0 + wn * 64 + 8 * fg;
        const bool full = (m0 + TBM <= p.M) && (n0 + TBN <= p.N);
        const T* Mz = Mb ? Mb + (long long)blockIdx.z * p.c_batch : nullptr;
        long long off[TI];
        bool rv[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int m = min(m0 + (wm * TI + i) * 16 + frow, p.M - 1);
            off[i] = row_off2(m, p.c_rpi, p.c_item, p.ldc, p.c_rpi2, p.c_item2) + min(nb, p.N - 8);
            rv[i] = (p.c_rpi == 0) || ((m % p.c_rpi) < p.c_valid);
        }
        const long long o1 = (nb + 32 < p.N) ? 32 : 0;                  // pair 1 beyond N: any valid address (value unused)
        uint4 mk[TI][2];
        if (Mz) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                mk[i][0] = *(const uint4*)(Mz + off[i]);
                mk[i][1] = *(const uint4*)(Mz + off[i] + o1);
            }
        }
        f32x4 bs[4];
        if (p.bias) {
            const int c0 = min(nb, p.N - 8), c1 = min(nb + 32, p.N - 8);
            bs[0] = *(const f32x4*)(p.bias + c0); bs[1] = *(const f32x4*)(p.bias + c0 + 4);
            bs[2] = *(const f32x4*)(p.bias + c1); bs[3] = *(const f32x4*)(p.bias + c1 + 4);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4 lo = acc[i][2 * pr], hi = acc[i][2 * pr + 1];
                if (p.bias) { lo += bs[2 * pr]; hi += bs[2 * pr + 1]; }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lo[e] = relu_f(lo[e]); hi[e] = relu_f(hi[e]); }
                }
                bf16x4 pl, ph;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pl[e] = (bf16_t)lo[e]; ph[e] = (bf16_t)hi[e]; }
                const uint2 ul = __builtin_bit_cast(uint2, pl), uh = __builtin_bit_cast(uint2, ph);
                unsigned vw[4] = {ul.x, ul.y, uh.x, uh.y};
                if (Mz) {
                    // bf16 > 0  <=>  sign bit clear and not zero
                    const unsigned mw4[4] = {mk[i][pr].x, mk[i][pr].y, mk[i][pr].z, mk[i][pr].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned l16 = mw4[e] & 0xffffu, h16 = mw4[e] >> 16;
                        const unsigned keep = ((l16 != 0u && l16 < 0x8000u) ? 0xffffu : 0u) | ((h16 != 0u && h16 < 0x8000u) ? 0xffff0000u : 0u);
                        vw[e] &= keep;
                    }
                }
                if (!rv[i]) {
                    if (p.flags & GEMM_SKIP_PAD_ROWS) continue;
                    vw[0] = 0u; vw[1] = 0u; vw[2] = 0u; vw[3] = 0u;
                }
                uint4* dst = (uint4*)(Cb + off[i] + 32 * pr);
                if (full || (m0 + (wm * TI + i) * 16 + frow < p.M && nb + 32 * pr < p.N)) {
                    store_out16(dst, make_uint4(vw[0], vw[1], vw[2], vw[3]), p.flags);
                }
            }
        }
        return;
    }
    if constexpr (sizeof(TO) == 2) {
        if (p.flags & GEMM_WIDE_EPI) {
            // Storage-dtype output through LDS: the accumulator fragments (4 consecutive columns per lane) are written
            // to a row-major tile image, then every thread moves whole 16-byte chunks of full rows, so the mask read and
            // the store are 512-byte row segments instead of 32-byte ones.  (The K loop ended on a barrier.)
            constexpr int CPR = TBN / 8;                 // 16-byte chunks per tile row
            constexpr int RPP = NTHR / CPR;              // rows per pass
            constexpr int NPASS = TBM / RPP;
            const int cc = tid % CPR, rr = tid / CPR;
            const int n = n0 + cc * 8;
            // ReLU-backward mask: all of a thread's mask chunks are requested here, before the accumulators go to LDS, so the
            // tile pays the HBM latency once and under the LDS writes (the fragment registers of the K loop are free by now)
            // With GemmNT::mask_bits the mask comes as ONE BYTE per 8 elements (bit e = element e of the chunk is > 0; written by
            // the kernel that produced the activation): 1/16 of the bytes, and the 2 x 32 MB burst (mask read + store) that every
            // round of tiles ends in loses its read half.  Same decision per element, bit-identical results.
            uint4 mk[NPASS];
            unsigned mbits[NPASS];
            float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool want_cs = CS_IN_LDS && p.colsum_slabs != nullptr;
            const unsigned char* Mbits = p.mask_bits;
            // (BITS_IN_LDS: the bits already sit in LDS and are applied while the accumulators are written to the tile image below)
            const bool bits_early = BITS_IN_LDS && Mbits != nullptr;
            if (Mbits && n < p.N) {
                if constexpr (!BITS_IN_LDS) {
#pragma unroll
                    for (int q = 0; q < NPASS; ++q) {
                        const int m = min(m0 + rr + q * RPP, p.M - 1);
                        mbits[q] = Mbits[((long long)blockIdx.z * p.c_batch + row_off2(m, p.c_rpi, p.c_item, p.ldc, p.c_rpi2, p.c_item2) + n) >> 3];
                    }
                }
            } else if (Mb && n < p.N) {
#pragma unroll
                for (int q = 0; q < NPASS; ++q) {
                    const int m = min(m0 + rr + q * RPP, p.M - 1);
                    mk[q] = *(const uint4*)(Mb + (long long)blockIdx.z * p.c_batch + row_off2(m, p.c_rpi, p.c_item, p.ldc, p.c_rpi2, p.c_item2) + n);
                }
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int ml = (wm * TI + i) * 16 + frow;
                // sign bits of this row's 64 columns of the wave (8 bytes of the tile's bit image); the nibble of the 4 columns a
                // lane holds of MFMA tile j starts at bit (j & 1) * 16 + 4 fg of the low (j < 2) or high word
                unsigned mlo = 0, mhi = 0;
                if constexpr (BITS_IN_LDS) {
                    if (bits_early) {
                        const uint2 mb8 = *(const uint2*)(lds + BITS_OFF + ml * (TBN / 8) + wn * 8);
                        mlo = mb8.x >> (4 * fg); mhi = mb8.y >> (4 * fg);
                    }
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int nl = (wn * TJ + j) * 16 + fg * 4;
                    f32x4 v = acc[i][j];
                    if (p.bias) v += *(const f32x4*)(p.bias + min(n0 + nl, p.N - 4));
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
                    }
                    if constexpr (BITS_IN_LDS) {
                        if (bits_early) {
                            bf16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
                            uint2 w = __builtin_bit_cast(uint2, o);
                            const unsigned nib = ((j < 2 ? mlo : mhi) >> ((j & 1) * 16)) & 0xfu;
                            w.x &= (nib & 1u ? 0xffffu : 0u) | (nib & 2u ? 0xffff0000u : 0u);
                            w.y &= (nib & 4u ? 0xffffu : 0u) | (nib & 8u ? 0xffff0000u : 0u);
                            *(uint2*)((TO*)(lds + ml * EPI_RS) + nl) = w;
                            continue;
                        }
                    }
                    store4((TO*)(lds + ml * EPI_RS) + nl, v);
                }
            }
            __syncthreads();
            if (n < p.N) {
                if constexpr (C1) {
                    // fused layer-1 weight gradient: the masked tile goes back to its LDS image (no global store) ...
                    if (!bits_early) {
#pragma unroll
                    for (int q = 0; q < NPASS; ++q) {
                        const int r = rr + q * RPP;
                        uint4 v = *(const uint4*)(lds + r * EPI_RS + cc * 16);
                        unsigned vw[4] = {v.x, v.y, v.z, v.w};
                        if (Mbits) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                vw[e] &= ((mbits[q] >> (2 * e)) & 1u ? 0xffffu : 0u) | ((mbits[q] >> (2 * e + 1)) & 1u ? 0xffff0000u : 0u);
                        } else {
                            const unsigned mw[4] = {mk[q].x, mk[q].y, mk[q].z, mk[q].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned lo = mw[e] & 0xffffu, hi = mw[e] >> 16;
                                const unsigned keep = ((lo != 0u && lo < 0x8000u) ? 0xffffu : 0u) |
                                                      ((hi != 0u && hi < 0x8000u) ? 0xffff0000u : 0u);
                                vw[e] &= keep;
                            }
                        }
                        *(uint4*)(lds + r * EPI_RS + cc * 16) = make_uint4(vw[0], vw[1], vw[2], vw[3]);
                    }
                    }
                }
                if constexpr (!C1) {
#pragma unroll
                for (int q = 0; q < NPASS; ++q) {
                    const int r = rr + q * RPP;
                    const int m = m0 + r;
                    if (m >= p.M) break;
                    const long long coff = row_off2(m, p.c_rpi, p.c_item, p.ldc, p.c_rpi2, p.c_item2);
                    uint4 v = *(const uint4*)(lds + r * EPI_RS + cc * 16);
                    if (Mbits) {
                        if (!bits_early) {
                            unsigned vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                vw[e] &= ((mbits[q] >> (2 * e)) & 1u ? 0xffffu : 0u) | ((mbits[q] >> (2 * e + 1)) & 1u ? 0xffff0000u : 0u);
                            v = make_uint4(vw[0], vw[1], vw[2], vw[3]);
                        }
                    } else if (Mb) {
                        // bf16 > 0  <=>  sign bit clear and not zero
                        const unsigned mw[4] = {mk[q].x, mk[q].y, mk[q].z, mk[q].w};
                        unsigned vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned lo = mw[e] & 0xffffu, hi = mw[e] >> 16;
                            const unsigned keep = ((lo != 0u && lo < 0x8000u) ? 0xffffu : 0u) |
                                                  ((hi != 0u && hi < 0x8000u) ? 0xffff0000u : 0u);
                            vw[e] &= keep;
                        }
                        v = make_uint4(vw[0], vw[1], vw[2], vw[3]);
                    }
                    const bool row_valid = (p.c_rpi == 0) || ((m % p.c_rpi) < p.c_valid);
                    if (!row_valid) {
                        if (p.flags & GEMM_SKIP_PAD_ROWS) continue;
                        v = make_uint4(0, 0, 0, 0);
                    }
                    store_out16((uint4*)(Cb + coff + n), v, p.flags);
                    if constexpr (CS_IN_LDS) {
                        if (want_cs) {              // column sums of what was just stored (the bf16 values, as a pass over C would see them)
                            const unsigned u4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                cs[2 * e] += __builtin_bit_cast(float, u4[e] << 16);
                                cs[2 * e + 1] += __builtin_bit_cast(float, u4[e] & 0xffff0000u);
                            }
                        }
                    }
                }
                }
            }
            if constexpr (CS_IN_LDS) {
                // GemmNT::colsum_slabs: the tile's 256 column sums — 16 row groups per column chunk through LDS, fixed order — to
                // slabs[mt][n]: the bias gradient of the layer below without another pass over this kernel's 0.06-0.24 GB output
                if (want_cs) {
                    float* red = (float*)(lds + CS_OFF);
                    *(f32x4*)(red + (rr * CPR + cc) * 8) = (f32x4){cs[0], cs[1], cs[2], cs[3]};
                    *(f32x4*)(red + (rr * CPR + cc) * 8 + 4) = (f32x4){cs[4], cs[5], cs[6], cs[7]};
                    __syncthreads();
                    if (tid < TBN && n0 + tid < p.N) {
                        float t = 0.f;
#pragma unroll
                        for (int g = 0; g < RPP; ++g) t += red[(g * CPR + (tid >> 3)) * 8 + (tid & 7)];
                        p.colsum_slabs[(long long)mt * p.N + n0 + tid] = t;
                    }
                }
            }
            if constexpr (C1) {
                // slab[j][c] = sum over the tile's rows of xw[row][j] * G[row][c]: a (16 x TBM) x (TBM x TBN) product on the
                // matrix pipe with both operands read transposed from their LDS images (the TN kernel's fragment reads);
                // wave w owns the column blocks 2w and 2w+1, so no sums cross waves.  x = hi + lo keeps ~16 mantissa bits.
                static_assert(TBN / 16 == 2 * WM * WN, "two 16-column blocks per wave");
                __syncthreads();
                const unsigned char* xw = lds + TBM * EPI_RS;
                f32x4 d0 = (f32x4){0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
                for (int ks = 0; ks < TBM / 32; ++ks) {
                    const uint4 fh = tn_frag_rb(xw, 32, 0, ks, lane);
                    const uint4 fl = tn_frag_rb(xw + TBM * 32, 32, 0, ks, lane);
                    const uint4 g0 = tn_frag_rb(lds, EPI_RS, (wave * 2) * 16, ks, lane);
                    const uint4 g1 = tn_frag_rb(lds, EPI_RS, (wave * 2 + 1) * 16, ks, lane);
                    mfma_chunk<bf16_t>(d0, g0, fh);
                    mfma_chunk<bf16_t>(d0, g0, fl);
                    mfma_chunk<bf16_t>(d1, g1, fh);
                    mfma_chunk<bf16_t>(d1, g1, fl);
                }
                const int slot = lane & 15, c4 = (lane >> 4) * 4;
                if (slot <= p.c1_kw) {
                    float* sl = p.c1_slabs + ((long long)mt * numN + nt) * (p.c1_kw + 1) * TBN + (long long)slot * TBN;
                    // (written through like the tile stores: 82 MB of slabs per launch, read by the reduction right behind this kernel)
                    store_out16((uint4*)(sl + (wave * 2) * 16 + c4), __builtin_bit_cast(uint4, d0), p.flags);
                    store_out16((uint4*)(sl + (wave * 2 + 1) * 16 + c4), __builtin_bit_cast(uint4, d1), p.flags);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int m = m0 + wm * TI * 16 + i * 16 + frow;
        if (m >= p.M) continue;
        const long long coff = row_off2(m, p.c_rpi, p.c_item, p.ldc, p.c_rpi2, p.c_item2);
        const bool row_valid = (p.c_rpi == 0) || ((m % p.c_rpi) < p.c_valid);
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int n = n0 + wn * TJ * 16 + j * 16 + fg * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (p.bias) v += *(const f32x4*)(p.bias + n);
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
            }
            if (Mb) {
                const f32x4 mk = load4(Mb + (long long)blockIdx.z * p.c_batch + coff + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
            }
            if (!row_valid) {
                if (p.flags & GEMM_SKIP_PAD_ROWS) continue;
                v = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            store4(Cb + coff + n, v);
        }
    }
}

// PERSISTENT form of the 256x256 bf16 LDS-DMA kernel with the register epilogue (no mask, no K ranges, one GEMM per launch): one workgroup
// per CU walks tiles blockIdx.x, + gridDim.x, ... and the stage stream never drains between them — while the last two K stages of a tile run,
// their LDS-DMA slots already receive stages 0 and 1 of the NEXT tile (its row pointers replace the current ones as soon as the current
// tile's last stage has been requested), the next tile's first fragments are read in the last iteration, and the epilogue's stores leave
// while the next tile's loads are in flight.  What it removes is the fixed cost of a tile (workgroup launch, first-stage latency, drain):
// 8.9 us of the 20 us a tile of the 24 576 x 512 x 24 576 score contraction takes, a tenth of a K = 2 048 convolution tile.
// Same sums in the same order as gemm_nt_fast_kernel: bit-identical results.  The macros are that kernel's (same local names).
// LSE = true: the InfoNCE score contraction with its column pass fused into the epilogue (GemmNT::lse_*): besides (or instead of) the bf16
// scores the tile leaves, for each of its 256 columns, the online log-sum-exp pair (max, sum exp(s - max)) over its 256 rows, taken from the
// f32 accumulators, and the diagonal scores s[r][r + lse_diag_off]; cpc_nce_lse_merge combines the pairs of a column over the M tiles.  The
// scores themselves are stored as f32 (optional): ONE matrix instead of the unfused path's two (scores and transposed scores).
// one DPP move of a float within its row of 16 lanes (all lanes enabled, bound_ctrl: out-of-row sources read 0 — none here)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int DBG, bool LSE = false>
__global__ __launch_bounds__(512) void gemm_nt_persist_kernel(GemmNT p) {
    typedef bf16_t T;
    typedef bf16_t TO;
    constexpr int WN = 4, TI = 8, TJ = 4;
    constexpr int CH = 8, BK = 64, TBM = 256, TBN = 256;
    constexpr int ATILE = TBM * 128, BTILE = TBN * 128, STAGE = ATILE + BTILE;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE + (LSE ? 2 * TBN * 8 : 0)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int numM = (p.M + TBM - 1) / TBM, numN = (p.N + TBN - 1) / TBN;
    const int total = (int)nt_grid_blocks(numM, numN);
    const int frow = lane & 15, fg = lane >> 4;
    int offA[2][TI], offB[2][TJ];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < TI; ++i) offA[kk][i] = lds_off(wm * TI * 16 + i * 16 + frow, kk * 4 + fg);
#pragma unroll
        for (int j = 0; j < TJ; ++j) offB[kk][j] = ATILE + lds_off(wn * TJ * 16 + j * 16 + frow, kk * 4 + fg);
    }
    const int nk = p.K / BK;
    const int taps = p.k_taps > 1 ? p.k_taps : 1;
    const long long tstride = p.k_taps > 1 ? p.k_tap_stride : 0;
    const long long tstride_a = p.k_taps > 1 && p.k_tap_stride_a ? p.k_tap_stride_a : tstride;
    const bool klin = taps > 1 && p.k_taps_linear;
    int kj = 0;
    long long kb = 0;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int srow = lane >> 3, sch = (lane & 7) ^ (lane >> 3);
    const int r0 = (wave_u * 4) * 8 + srow;
    // (LSE: f32 output — the plain column order, in which the four lanes of a row group store 64 contiguous bytes per row; the 8-column
    // permutation of the bf16 register epilogue would leave every 32-byte sector half written by each store instruction: 1.68 instead of
    // 1.1 ms for the 2.4 GB of the 24 576^2 score matrix)
    const int br0 = LSE ? r0 : direct_b_col(r0);
    const T* Ab = (const T*)p.A;
    const T* Bb = (const T*)p.Bt;
    // The four rows a lane requests of each operand per stage are a FIXED distance apart (the launcher admits only launches where a 32-row
    // group of A lies in one item and inside M, and plain B rows with N a multiple of 256): one pointer per operand in registers instead
    // of four — with eight 64-bit pointers the loop spills (the epilogue's operands now live across it).
    const T *ga0, *gb0;
    const long long a8 = 8LL * p.lda, b4 = 4LL * p.ldb, b16 = 16LL * p.ldb;          // (direct_b_col: rows r0 + 8 k sit 0, 16, 4, 20 B rows after r0's)
#define ga1 (ga0 + a8)
#define ga2 (ga0 + 2 * a8)
#define ga3 (ga0 + 3 * a8)
    const long long bo1 = LSE ? 8LL * p.ldb : b16, bo2 = LSE ? 16LL * p.ldb : b4, bo3 = LSE ? 24LL * p.ldb : b16 + b4;
#define gb1 (gb0 + bo1)
#define gb2 (gb0 + bo2)
#define gb3 (gb0 + bo3)
    const T *gd0 = Bb, *gd1 = Bb, *gd2 = Bb, *gd3 = Bb;          // (timing-probe operands of the shared macros; never used with DBG = 0)
    // tile index -> first row / column (false: a padding block of the 8-XCD grid, no tile)
#define PS_TILE_OF(bid, m0_, n0_, ok_)                                   \
    do {                                                                 \
        int mt_, nt_;                                                    \
        nt_tile_of_block((bid), numM, numN, mt_, nt_);                   \
        m0_ = mt_ * TBM;                                                 \
        n0_ = nt_ * TBN;                                                 \
        ok_ = mt_ < numM;                                                \
    } while (0)
#define PS_NEXT_TILE(bid, m0_, n0_)                                      \
    do {                                                                 \
        bool ok_ = false;                                                \
        while ((bid) < total) {                                          \
            PS_TILE_OF(bid, m0_, n0_, ok_);                              \
            if (ok_) break;                                              \
            (bid) += gridDim.x;                                          \
        }                                                                \
    } while (0)
#define PS_SET_ROWS(m0_, n0_)                                                                                    \
    do {                                                                                                         \
        ga0 = Ab + row_off(min((m0_) + wave_u * 32, p.M - 32) + srow, p.a_rpi, p.a_item, p.lda) + sch * CH;      \
        gb0 = Bb + (long long)((n0_) + br0) * p.ldb + sch * CH;                                                  \
    } while (0)
    int m0 = 0, n0 = 0;
    int tile = blockIdx.x;
    PS_NEXT_TILE(tile, m0, n0);
    if (tile >= total) return;
    PS_SET_ROWS(m0, n0);
    // start stagger (GemmNT::stagger): the workgroups never resynchronise, so a phase shift given here keeps the CUs' store bursts (128 KiB
    // per tile and CU, 32 MB per round if all CUs finish together) apart for the whole launch
    if (p.stagger > 0) {
        const int nsleep = ((blockIdx.x >> 3) & 7) * p.stagger;
        for (int i = 0; i < nsleep; ++i) __builtin_amdgcn_s_sleep(64);
    }

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    lds_byte* const lds3 = (lds_byte*)lds;
    const unsigned wdst = wave_u * 4096;
    const unsigned lds_u32 = (unsigned)(unsigned long long)(lds3);
    const unsigned aA0 = lds_u32 + offA[0][0], aA1 = lds_u32 + offA[1][0];
    const unsigned aB0 = lds_u32 + offB[0][0], aB1 = lds_u32 + offB[1][0];
    const long long koff0 = 0, koff0a = 0;
    NT_KSTEP();
    const long long koff1 = kb + kj * tstride, koff1a = kb + kj * tstride_a;
    NT_KSTEP();
    NT_DMA_STAGE(0, koff0a, koff0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    NT_DMA_STAGE(1, koff1a, koff1);
    u32x4 fa0[TI], fb0[TJ], fa1[TI], fb1[TJ];
    NT_READ_SET(fa0, fb0, aA0, aB0, 0u);
    NT_WAIT_SET(fa0, fb0);
    TO* const Cb = (TO*)p.C;
    const bool relu = p.flags & GEMM_RELU;
    int t = 0;                                   // running stage count over all tiles of this workgroup: the LDS slot of a stage is its parity
    for (;;) {
        int m1 = m0, n1 = n0;
        int nxt_tile = tile + gridDim.x;
        PS_NEXT_TILE(nxt_tile, m1, n1);
        const bool has_next = nxt_tile < total;
        if (!has_next) { m1 = m0; n1 = n0; }      // the last tile requests its own first stages once more (never used) so that the loop has ONE form
        for (int s_ = 0; s_ < nk; ++s_, ++t) {
            if (s_ == nk - 2) {
                // every request of this tile is out: the row pointers become the next tile's, the K position starts over, and the last two
                // iterations request its stages 0 and 1 (and read its first fragments) exactly as any other iteration would
                PS_SET_ROWS(m1, n1);
                kb = 0;
                kj = 0;
            }
            NT_ITER(true, true);
        }
        // register epilogue of gemm_nt_fast_kernel's DIRECT form, without a mask; the accumulators are cleared as they are consumed
        const int nb = n0 + wn * 64 + 8 * fg;
        const bool full = (m0 + TBM <= p.M) && (n0 + TBN <= p.N);
        if constexpr (LSE) {
            // Column q = 4 j + e of this lane is tile column wn 64 + 16 j + 4 fg + e (acc[i][j][e]; plain column order, see br0); its rows are
            // (wm 8 + i) 16 + frow.  Pass 1: the column maxima over the lane's 8 rows, then over the 16 lanes of the row group (lanes that
            // differ in frow only); pass 2: sum of exp(s - max) the same way; the two wm halves meet in LDS.
            float cm[16], cs[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { cm[q] = -INFINITY; cs[q] = 0.f; }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) cm[q] = fmaxf(cm[q], acc[i][q >> 2][q & 3]);
            // (reductions over the 16 lanes of a row group as DPP moves — quad permutes, then half-row and row mirrors — on the vector
            // pipe; __shfl_xor goes through the LDS crossbar: 128 ds_bpermute per lane and tile)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                cm[q] = fmaxf(cm[q], dpp_f32<0xB1>(cm[q]));          // quad_perm [1,0,3,2]
                cm[q] = fmaxf(cm[q], dpp_f32<0x4E>(cm[q]));          // quad_perm [2,3,0,1]
                cm[q] = fmaxf(cm[q], dpp_f32<0x141>(cm[q]));         // row_half_mirror
                cm[q] = fmaxf(cm[q], dpp_f32<0x140>(cm[q]));         // row_mirror
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) cs[q] += __expf(acc[i][q >> 2][q & 3] - cm[q]);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                cs[q] += dpp_f32<0xB1>(cs[q]);
                cs[q] += dpp_f32<0x4E>(cs[q]);
                cs[q] += dpp_f32<0x141>(cs[q]);
                cs[q] += dpp_f32<0x140>(cs[q]);
            }
            float* red = (float*)(lds + 2 * STAGE);
            if (frow == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int col = wn * 64 + 16 * (q >> 2) + 4 * fg + (q & 3);
                    red[(wm * TBN + col) * 2] = cm[q];
                    red[(wm * TBN + col) * 2 + 1] = cs[q];
                }
            }
            // the scores a prediction gives its own target (the "valid" scores of the loss): rows whose column r + lse_diag_off is in this tile
            if (p.lse_valid != nullptr && m0 + p.lse_diag_off < n0 + TBN && m0 + TBM + p.lse_diag_off > n0) {
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const int r = m0 + (wm * TI + i) * 16 + frow;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int c = n0 + wn * 64 + 16 * (q >> 2) + 4 * fg + (q & 3);
                        if (c == r + p.lse_diag_off) p.lse_valid[r] = acc[i][q >> 2][q & 3];
                    }
                }
            }
            __syncthreads();
            if (tid < TBN) {
                const float ma = red[tid * 2], sa = red[tid * 2 + 1], mb = red[(TBN + tid) * 2], sb = red[(TBN + tid) * 2 + 1];
                const float mm = fmaxf(ma, mb);
                const long long o = (long long)(m0 / TBM) * p.N + n0 + tid;
                p.lse_pm[o] = mm;
                p.lse_ps[o] = sa * __expf(ma - mm) + sb * __expf(mb - mm);
            }
            // (the next tile's K loop has barriers between this read of `red` and the next tile's writes to it)
        }
        if constexpr (LSE) {
            // the scores themselves, f32 (the gradient pass takes exp(score - lse): a bf16 copy of a score of 100 is off by up to 0.4, i.e.
            // its softmax weight by half): 16 bytes per lane, the four lanes of a row group write 64 contiguous bytes
            if (Cb != nullptr) {
                float* const Cf = (float*)p.C;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const long long off = (long long)(m0 + (wm * TI + i) * 16 + frow) * p.ldc + n0 + wn * 64 + 4 * fg;
#pragma unroll
                    for (int j = 0; j < TJ; ++j) store_out16((uint4*)(Cf + off + 16 * j), __builtin_bit_cast(uint4, acc[i][j]), p.flags);
                }
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < (LSE ? 0 : TI); ++i) {
            const int mrow = m0 + (wm * TI + i) * 16 + frow;
            const int m = min(mrow, p.M - 1);
            const long long off = row_off(m, p.c_rpi, p.c_item, p.ldc) + min(nb, p.N - 8);
            const bool rv = (p.c_rpi == 0) || ((m % p.c_rpi) < p.c_valid);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4 lo = acc[i][2 * pr], hi = acc[i][2 * pr + 1];
                acc[i][2 * pr] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[i][2 * pr + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
                    const int c_ = min(nb + 32 * pr, p.N - 8);
                    lo += *(const f32x4*)(p.bias + c_);
                    hi += *(const f32x4*)(p.bias + c_ + 4);
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lo[e] = relu_f(lo[e]); hi[e] = relu_f(hi[e]); }
                }
                bf16x4 pl, ph;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pl[e] = (bf16_t)lo[e]; ph[e] = (bf16_t)hi[e]; }
                const uint2 ul = __builtin_bit_cast(uint2, pl), uh = __builtin_bit_cast(uint2, ph);
                unsigned vw[4] = {ul.x, ul.y, uh.x, uh.y};
                if (!rv) {
                    if (p.flags & GEMM_SKIP_PAD_ROWS) continue;
                    vw[0] = 0u; vw[1] = 0u; vw[2] = 0u; vw[3] = 0u;
                }
                if (full || (mrow < p.M && nb + 32 * pr < p.N))
                    store_out16((uint4*)(Cb + off + 32 * pr), make_uint4(vw[0], vw[1], vw[2], vw[3]), p.flags);
            }
        }
        if (!has_next) break;
        tile = nxt_tile;
        m0 = m1;
        n0 = n1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the last tile's surplus requests land before the LDS is released)
}
#undef PS_TILE_OF
#undef PS_NEXT_TILE
#undef PS_SET_ROWS
#undef ga1
#undef ga2
#undef ga3
#undef gb1
#undef gb2
#undef gb3
#undef NT_ITER
#undef NT_KSTEP
#undef NT_READ1
#undef NT_DMA_PIECE
#undef NT_READ_SET
#undef NT_WAIT_SET
#undef NT_DMA1
#undef NT_DMA_STAGE

// TN fast path (bf16): stage = 64 reduction rows x (tile width) columns per operand, LDS rows padded by 32 B so that the
// 8 rows a half-wave touches in one transposed read start 8 banks apart, double-buffered, one barrier per stage.
// Reduction rows beyond the split's end are zeroed in registers (they would otherwise add to the sums); out-of-range
// columns are clamped.  Tile configurations as for the NT kernel: <2,2,4,4> = 128x128, <2,4,8,4> = 256x256.
__device__ __forceinline__ uint4 tn_frag_rb(const unsigned char* tile, int rowb, int cb, int ks, int lane) {
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const unsigned char* a1 = tile + (ks * 32 + 4 * g + q) * rowb + (cb + 4 * pp) * 2;
    const unsigned char* a2 = a1 + 16 * rowb;
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a2);
    uint2 lo = __builtin_bit_cast(uint2, v1), hi = __builtin_bit_cast(uint2, v2);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

template <typename TO, int WI, int WJ, int TI, int TJ, bool PLAIN>
__global__ __launch_bounds__(64 * WI * WJ) void gemm_tn_fast_kernel(GemmTN p) {
    typedef bf16_t T;
    constexpr int BKM = 64;
    constexpr int TBI = WI * TI * 16, TBJ = WJ * TJ * 16, NTHR = 64 * WI * WJ;
    constexpr int ROWA = TBI * 2 + 32, ROWB_ = TBJ * 2 + 32;
    constexpr int ATILE = BKM * ROWA, BTILE = BKM * ROWB_, STAGE = ATILE + BTILE;
    constexpr int CPRA = TBI / 8, CPRB = TBJ / 8;                  // 16-byte chunks per tile row
    static_assert(NTHR / CPRA == 16 && NTHR / CPRB == 16, "staging below assumes 16 rows per pass, 4 passes");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / WJ, wj = wave % WJ;
    // grid (nsplit, tiles): consecutive blocks are the splits of one tile, so with nsplit a multiple of 8 the tiles of one
    // split — which read the same reduction rows of both operands — run together on one XCD and share them through its L2
    const int numJ = (p.J + TBJ - 1) / TBJ;
    const int it = blockIdx.y / numJ, jt = blockIdx.y % numJ;
    const int i0 = it * TBI, j0 = jt * TBJ;
    const int split = blockIdx.x;
    const int m_begin = split * p.m_chunk;
    const int m_end = min(p.M, m_begin + p.m_chunk);

    const T* Ab = (const T*)p.A + (long long)blockIdx.z * p.a_batch;
    const T* Bb = (const T*)p.B + (long long)blockIdx.z * p.b_batch;

    const int cha = tid % CPRA, chb = tid % CPRB, srow = tid / CPRA;
    const long long acol = min(i0 + cha * 8, p.I - 8);
    const long long bcol = min(j0 + chb * 8, p.J - 8);
    const int sda = srow * ROWA + cha * 16, sdb = ATILE + srow * ROWB_ + chb * 16;
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    // PLAIN (both operands are ordinary row-major matrices: the conv weight gradients): row m of a stage is a fixed offset
    // from the stage's first row.  Stages that lie entirely inside [m_begin, m_end) — all but possibly the last — load without
    // clamping or masking, so their eight global loads stay in flight under the MFMAs of the current stage; only a ragged
    // last stage clamps the row index and zeroes the rows beyond the split's end (they would otherwise add to the sums).
    const T* const pa = Ab + acol;
    const T* const pb = Bb + bcol;
    const long long a16 = 16LL * p.lda, b16 = 16LL * p.ldb;
#define TN_LOAD_RAGGED(RA, RB, mrow)                                                                         \
    do {                                                                                                     \
        const int m_ = (mrow);                                                                               \
        const int mc_ = min(m_, p.M - 1);                                                                    \
        RA = *(const uint4*)(pa + (PLAIN ? (long long)mc_ * p.lda : row_off(mc_, p.a_rpi, p.a_item, p.lda))); \
        RB = *(const uint4*)(pb + (PLAIN ? (long long)mc_ * p.ldb : row_off(mc_, p.b_rpi, p.b_item, p.ldb))); \
        if (m_ >= m_end) { RA = zero4; RB = zero4; }                                                         \
    } while (0)
#define TN_GLOAD(st)                                                                                         \
    do {                                                                                                     \
        const int mb_ = m_begin + (st) * BKM + srow;                                                         \
        if ((st) < nfull) {                                                                                  \
            if constexpr (PLAIN) {                                                                           \
                const T* qa = pa + (long long)mb_ * p.lda;                                                   \
                const T* qb = pb + (long long)mb_ * p.ldb;                                                   \
                ra0 = *(const uint4*)(qa);           rb0 = *(const uint4*)(qb);                              \
                ra1 = *(const uint4*)(qa + a16);     rb1 = *(const uint4*)(qb + b16);                        \
                ra2 = *(const uint4*)(qa + 2 * a16); rb2 = *(const uint4*)(qb + 2 * b16);                    \
                ra3 = *(const uint4*)(qa + 3 * a16); rb3 = *(const uint4*)(qb + 3 * b16);                    \
            } else {                                                                                         \
                ra0 = *(const uint4*)(pa + row_off(mb_, p.a_rpi, p.a_item, p.lda));                          \
                rb0 = *(const uint4*)(pb + row_off(mb_, p.b_rpi, p.b_item, p.ldb));                          \
                ra1 = *(const uint4*)(pa + row_off(mb_ + 16, p.a_rpi, p.a_item, p.lda));                     \
                rb1 = *(const uint4*)(pb + row_off(mb_ + 16, p.b_rpi, p.b_item, p.ldb));                     \
                ra2 = *(const uint4*)(pa + row_off(mb_ + 32, p.a_rpi, p.a_item, p.lda));                     \
                rb2 = *(const uint4*)(pb + row_off(mb_ + 32, p.b_rpi, p.b_item, p.ldb));                     \
                ra3 = *(const uint4*)(pa + row_off(mb_ + 48, p.a_rpi, p.a_item, p.lda));                     \
                rb3 = *(const uint4*)(pb + row_off(mb_ + 48, p.b_rpi, p.b_item, p.ldb));                     \
            }                                                                                                \
        } else {                                                                                             \
            TN_LOAD_RAGGED(ra0, rb0, mb_);      TN_LOAD_RAGGED(ra1, rb1, mb_ + 16);                          \
            TN_LOAD_RAGGED(ra2, rb2, mb_ + 32); TN_LOAD_RAGGED(ra3, rb3, mb_ + 48);                          \
        }                                                                                                    \
    } while (0)
#define TN_LSTORE(base)                                                                      \
    do {                                                                                     \
        unsigned char* da = (base) + sda;                                                    \
        unsigned char* db = (base) + sdb;                                                    \
        *(uint4*)(da) = ra0;             *(uint4*)(db) = rb0;                                \
        *(uint4*)(da + 16 * ROWA) = ra1; *(uint4*)(db + 16 * ROWB_) = rb1;                   \
        *(uint4*)(da + 32 * ROWA) = ra2; *(uint4*)(db + 32 * ROWB_) = rb2;                   \
        *(uint4*)(da + 48 * ROWA) = ra3; *(uint4*)(db + 48 * ROWB_) = rb3;                   \
    } while (0)

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nst = (m_end - m_begin + BKM - 1) / BKM;
    const int nfull = (m_end - m_begin) / BKM;            // stages with all 64 rows inside the split
    if (nst > 0) {
        TN_GLOAD(0);
        TN_LSTORE(lds);
    }
    __syncthreads();
    for (int t = 0; t < nst; ++t) {
        const unsigned char* ldsA = lds + (t & 1) * STAGE;
        const unsigned char* ldsB = ldsA + ATILE;
        const bool more = t + 1 < nst;
        if (more) TN_GLOAD(t + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fa[TI], fb[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[i] = tn_frag_rb(ldsA, ROWA, (wi * TI + i) * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[j] = tn_frag_rb(ldsB, ROWB_, (wj * TJ + j) * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) mfma_chunk<bf16_t>(acc[i][j], fb[j], fa[i]);
        }
        if (more) TN_LSTORE(lds + ((t + 1) & 1) * STAGE);
        __syncthreads();
    }
#undef TN_LOAD_RAGGED
#undef TN_GLOAD
#undef TN_LSTORE

    const int fidx = lane & 15, fg = lane >> 4;
    TO* Cb = (TO*)p.C + (long long)blockIdx.z * p.c_batch + (long long)split * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int ii = i0 + (wi * TI + i) * 16 + fidx;
        if (ii >= p.I) continue;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int jj = j0 + (wj * TJ + j) * 16 + fg * 4;
            if (jj >= p.J) continue;
            store4(Cb + row_off(ii, p.c_rpi, p.c_item, p.ldc) + jj, acc[i][j]);
        }
    }
}

// TN with LDS-DMA staging (bf16, ordinary row-major operands): the same tile shapes and split layout as gemm_tn_fast_kernel, but
// the operand stages go global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write traffic — the register-
// staged kernel is bound by its LDS writes) and the transposing fragment reads are inline-asm ds_read_b64_tr_b16 issued between
// the MFMAs with hand-counted waits, exactly as in gemm_nt_fast_kernel (hipcc would drain vmcnt(0) before every LDS read it sees).
// LDS rows are unpadded (a DMA piece is 1 KiB of consecutive LDS): the 32-byte column granule g of tile row r sits at granule
// g ^ (r & 7), applied to the per-lane SOURCE address of the DMA and undone by the fragment reads — the 8 rows x 32 bytes a
// half-wave touches in one transposed read cover all 64 banks.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr_asm(unsigned addr) {
    u32x2 d;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
    return d;
}

// ROWS: operands addressed as items of rpi rows (the valid rows of each column of a grid); the row -> offset division runs on
// a float reciprocal with one correction step (8 per lane and stage, beside 64 MFMAs per wave).
__device__ __forceinline__ void divmod_rcp(int m, int d, float inv, int& q, int& r) {
    q = (int)(((float)m + 0.5f) * inv);
    r = m - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}
__device__ __forceinline__ long long row_off_rcp(int m, int rpi, float inv, long long item, long long ld, int rpi2 = 0, float inv2 = 0.f,
                                                 long long item2 = 0) {
    if (rpi == 0) return (long long)m * ld;
    int q, r;
    divmod_rcp(m, rpi, inv, q, r);
    if (rpi2 == 0) return (long long)q * item + (long long)r * ld;
    int q2, r2;
    divmod_rcp(q, rpi2, inv2, q2, r2);
    return (long long)q2 * item2 + (long long)r2 * item + (long long)r * ld;
}

template <typename TO, int WI, int WJ, int TI, int TJ, bool ROWS>
__global__ __launch_bounds__(64 * WI * WJ) void gemm_tn_dma_kernel(GemmTN p) {
    typedef bf16_t T;
    constexpr int BKM = 64, NW = WI * WJ;
    constexpr int TBI = WI * TI * 16, TBJ = WJ * TJ * 16;
    constexpr int RBA = TBI * 2, RBB = TBJ * 2;                         // bytes per LDS row
    constexpr int ATILE = BKM * RBA, BTILE = BKM * RBB, STAGE = ATILE + BTILE;
    static_assert(ATILE / 1024 == 4 * NW && BTILE / 1024 == 4 * NW, "four DMA pieces per operand per wave");
    static_assert(TJ == 4 && (TI == 4 || TI == 8), "read schedule below is written for these tiles");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wi = wave / WJ, wj = wave % WJ;
    const int numJ = (p.J + TBJ - 1) / TBJ;
    const int it = blockIdx.y / numJ, jt = blockIdx.y % numJ;
    const int i0 = it * TBI, j0 = jt * TBJ;
    const int split = blockIdx.x;
    const int m_begin = split * p.m_chunk;
    const int m_end = min(p.M, m_begin + p.m_chunk);
    const int nst = (m_end - m_begin + BKM - 1) / BKM;
    const T* Ab = (const T*)p.A + (long long)blockIdx.z * p.a_batch;
    const T* Bb = (const T*)p.B + (long long)blockIdx.z * p.b_batch;

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nst > 0) {
        // ---- DMA sources: wave w fills pieces 4w .. 4w+3 of each operand tile; piece pc = LDS bytes [1024 pc, +1024)
        constexpr int CPA = RBA / 16, CPB = RBB / 16;                   // 16-byte chunks per row
        constexpr int RPA = 1024 / RBA, RPB = 1024 / RBB;               // rows per piece
        const int rra = lane / CPA, dca = lane % CPA, rrb = lane / CPB, dcb = lane % CPB;
        const T* ga[4];
        const T* gb[4];
        int rowa[4], rowb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rowa[k] = (wave_u * 4 + k) * RPA + rra;
            rowb[k] = (wave_u * 4 + k) * RPB + rrb;
            const int sca = (((dca >> 1) ^ (rowa[k] & 7)) << 1) | (dca & 1);
            const int scb = (((dcb >> 1) ^ (rowb[k] & 7)) << 1) | (dcb & 1);
            ga[k] = Ab + min(i0 + sca * 8, p.I - 8);
            gb[k] = Bb + min(j0 + scb * 8, p.J - 8);
        }
        typedef __attribute__((address_space(3))) unsigned char lds_byte;
        lds_byte* const lds3 = (lds_byte*)lds;
        const unsigned wdst = wave_u * 4096;
        const float inva = ROWS && p.a_rpi ? 1.0f / (float)p.a_rpi : 0.f, invb = ROWS && p.b_rpi ? 1.0f / (float)p.b_rpi : 0.f;
        const float inva2 = ROWS && p.a_rpi2 ? 1.0f / (float)p.a_rpi2 : 0.f;
        auto offa = [&](int m) -> long long {
            if constexpr (ROWS) return row_off_rcp(m, p.a_rpi, inva, p.a_item, p.lda, p.a_rpi2, inva2, p.a_item2);
            else return (long long)m * p.lda;
        };
        auto offb = [&](int m) -> long long {
            if constexpr (ROWS) return row_off_rcp(m, p.b_rpi, invb, p.b_item, p.ldb);
            else return (long long)m * p.ldb;
        };
#define TN_DMA1(g, dst)                                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g),                         \
                                     (__attribute__((address_space(3))) void*)(lds3 + (dst)), 16, 0, 0)
        // piece k of stage st into slot buf (rows beyond the matrix are clamped; rows beyond the split are zeroed after landing)
#define TN_DMA_PIECE(idx, buf, st)                                                                               \
    do {                                                                                                         \
        const int mb_ = m_begin + (st) * BKM;                                                                    \
        const unsigned da = (buf) * STAGE + wdst, db = da + ATILE;                                               \
        switch (idx) {                                                                                           \
        case 0: TN_DMA1(ga[0] + offa(min(mb_ + rowa[0], p.M - 1)), da); break;                      \
        case 1: TN_DMA1(gb[0] + offb(min(mb_ + rowb[0], p.M - 1)), db); break;                      \
        case 2: TN_DMA1(ga[1] + offa(min(mb_ + rowa[1], p.M - 1)), da + 1024); break;               \
        case 3: TN_DMA1(gb[1] + offb(min(mb_ + rowb[1], p.M - 1)), db + 1024); break;               \
        case 4: TN_DMA1(ga[2] + offa(min(mb_ + rowa[2], p.M - 1)), da + 2048); break;               \
        case 5: TN_DMA1(gb[2] + offb(min(mb_ + rowb[2], p.M - 1)), db + 2048); break;               \
        case 6: TN_DMA1(ga[3] + offa(min(mb_ + rowa[3], p.M - 1)), da + 3072); break;               \
        case 7: TN_DMA1(gb[3] + offb(min(mb_ + rowb[3], p.M - 1)), db + 3072); break;               \
        default: break;                                                                                          \
        }                                                                                                        \
    } while (0)
        // rows of stage st beyond the split's end: zero them in LDS once the stage has landed (they would add to the sums)
        auto zero_ragged = [&](int buf, int st) {
            const int valid = m_end - (m_begin + st * BKM);              // rows [valid, 64) are out
            if (valid >= BKM) return;
            for (int q = tid; q < BKM * (CPA + CPB); q += 64 * NW) {
                const bool isa = q < BKM * CPA;
                const int qq = isa ? q : q - BKM * CPA;
                const int r = isa ? qq / CPA : qq / CPB, c = isa ? qq % CPA : qq % CPB;
                if (r >= valid)
                    *(uint4*)(lds + buf * STAGE + (isa ? 0 : ATILE) + r * (isa ? RBA : RBB) + c * 16) = make_uint4(0, 0, 0, 0);
            }
        };

        // ---- fragment read addresses (bytes within a stage)
        const int g = lane >> 4, idx16 = lane & 15, q4 = idx16 >> 2, pp = idx16 & 3;
        const int rloc = 4 * g + q4, r7 = rloc & 7;
        const unsigned lds_u32 = (unsigned)(unsigned long long)(lds3);
        unsigned adA[TI], adB[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) adA[i] = lds_u32 + rloc * RBA + (((wi * TI + i) ^ r7) << 5) + pp * 8;
#pragma unroll
        for (int j = 0; j < TJ; ++j) adB[j] = lds_u32 + ATILE + rloc * RBB + (((wj * TJ + j) ^ r7) << 5) + pp * 8;
        // fragment set of one k-step ks: fb[j] / fa[i] as (low rows, high rows) 8-byte halves
        u32x2 fbl0[TJ], fbh0[TJ], fal0[TI], fah0[TI], fbl1[TJ], fbh1[TJ], fal1[TI], fah1[TI];
        constexpr int NRD = 2 * (TI + TJ);                              // reads per fragment set
#define TN_READ1(idx, KS, fbl, fbh, fal, fah, cur)                                                               \
    do {                                                                                                         \
        const int i_ = (idx);                                                                                    \
        if (i_ >= 0 && i_ < 2 * TJ) {                                                                            \
            if (i_ & 1) fbh[i_ >> 1] = lds_read_tr_asm<(KS) * 32 * RBB + 16 * RBB>(adB[i_ >> 1] + (cur));        \
            else fbl[i_ >> 1] = lds_read_tr_asm<(KS) * 32 * RBB>(adB[i_ >> 1] + (cur));                          \
        } else if (i_ >= 2 * TJ && i_ < NRD) {                                                                   \
            const int a_ = (i_ - 2 * TJ) >> 1;                                                                   \
            if (i_ & 1) fah[a_] = lds_read_tr_asm<(KS) * 32 * RBA + 16 * RBA>(adA[a_] + (cur));                  \
            else fal[a_] = lds_read_tr_asm<(KS) * 32 * RBA>(adA[a_] + (cur));                                    \
        }                                                                                                        \
    } while (0)
#define TN_WAIT_ALL()                                                                                            \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
#define TN_FRAG(l, h) make_uint4((l)[0], (l)[1], (h)[0], (h)[1])
        // one stage: block 0 = MFMAs of k-step 0 with the reads of k-step 1 between them; wait, DMA(t+1) landed, barrier;
        // block 1 = MFMAs of k-step 1 with the DMA pieces of stage t+2 and the k-step-0 reads of stage t+1 between them
#define TN_ITER(DO_DMA, DO_READ)                                                                                 \
    do {                                                                                                         \
        const unsigned cur = (t & 1) * STAGE, nxt = ((t + 1) & 1) * STAGE;                                       \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) {                                                         \
            _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                     \
                mfma_chunk<bf16_t>(acc[i][j], TN_FRAG(fbl0[j], fbh0[j]), TN_FRAG(fal0[i], fah0[i]));             \
                TN_READ1(i * TJ + j, 1, fbl1, fbh1, fal1, fah1, cur);                                            \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
        }                                                                                                        \
        TN_WAIT_ALL();                                                                                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
        if (DO_READ) zero_ragged((t + 1) & 1, t + 1);                                                            \
        __syncthreads();                                                                                         \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) {                                                         \
            _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                     \
                mfma_chunk<bf16_t>(acc[i][j], TN_FRAG(fbl1[j], fbh1[j]), TN_FRAG(fal1[i], fah1[i]));             \
                if (DO_DMA) TN_DMA_PIECE(i * TJ + j, t & 1, t + 2);                                              \
                if (DO_READ) TN_READ1(TI == 8 ? i * TJ + j - (DO_DMA ? 8 : 0) : i * TJ + j, 0, fbl0, fbh0, fal0, fah0, nxt); \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
        }                                                                                                        \
        if (DO_READ) TN_WAIT_ALL();                                                                              \
    } while (0)
        static_assert(TI * TJ >= 8 + NRD || TI * TJ == 16, "block 1 must hold 8 DMA pieces and one fragment set");
        // prologue: stage 0 (and 1) in flight, k-step-0 fragments of stage 0 in registers
#pragma unroll
        for (int k = 0; k < 8; ++k) TN_DMA_PIECE(k, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        zero_ragged(0, 0);
        __syncthreads();
        if (nst > 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) TN_DMA_PIECE(k, 1, 1);
        }
#pragma unroll
        for (int k = 0; k < NRD; ++k) TN_READ1(k, 0, fbl0, fbh0, fal0, fah0, 0u);
        TN_WAIT_ALL();
        int t = 0;
        for (; t + 2 < nst; ++t) TN_ITER(true, true);
        if (t + 1 < nst) {
            TN_ITER(false, true);
            ++t;
        }
        TN_ITER(false, false);
#undef TN_ITER
#undef TN_FRAG
#undef TN_WAIT_ALL
#undef TN_READ1
#undef TN_DMA_PIECE
#undef TN_DMA1
    }

    const int fidx = lane & 15, fg = lane >> 4;
    TO* Cb = (TO*)p.C + (long long)blockIdx.z * p.c_batch + (long long)split * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int ii = i0 + (wi * TI + i) * 16 + fidx;
        if (ii >= p.I) continue;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int jj = j0 + (wj * TJ + j) * 16 + fg * 4;
            if (jj >= p.J) continue;
            store4(Cb + row_off(ii, p.c_rpi, p.c_item, p.ldc) + jj, acc[i][j]);
        }
    }
}

// out[perm(i, j)] = (accumulate ? out : 0) + sum_z slab[z][i][j];   perm(i,j) = j*s_j + (i / cdiv)*s_hi + (i % cdiv)*s_lo
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                           int I, int J, int nslab, long long slab_stride,
                                                           int cdiv, long long s_j, long long s_hi, long long s_lo) {
    const long long total4 = (long long)I * J / 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long long)gridDim.x * 256) {
        const long long e = idx * 4;
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < nslab; ++z) s += *(const f32x4*)(slabs + (long long)z * slab_stride + e);
        const int i = (int)(e / J), j = (int)(e % J);
        const long long base = (long long)(i / cdiv) * s_hi + (long long)(i % cdiv) * s_lo;
        float* dst = out + base + (long long)j * s_j;
        if (s_j == 1 && ((unsigned long long)dst & 15ull) == 0) {          // rows of a matrix: one 16-byte store
            *(f32x4*)dst = s;
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[(long long)q * s_j] = s[q];
    }
}

// Same reduction for SMALL outputs and many slabs (bias gradients, layer-1 weights, BatchNorm sums): 256 threads =
// (256 / ZL) output float4 x ZL slab lanes, lanes combined through LDS in a fixed order, so that the serial chain per thread
// is nslab / ZL  (ZL = 16, or 64 for 512 slabs and more).
template <int ZL>
__global__ __launch_bounds__(256) void reduce_slabs_small_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                 int I, int J, int nslab, long long slab_stride, int cdiv,
                                                                 long long s_j, long long s_hi, long long s_lo) {
    constexpr int OL = 256 / ZL;
    __shared__ float red[ZL][OL][4];
    const int ol = threadIdx.x % OL, zl = threadIdx.x / OL;
    const long long total4 = (long long)I * J / 4;
    const long long o4 = (long long)blockIdx.x * OL + ol;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (o4 < total4)
        for (int z = zl; z < nslab; z += ZL) s += *(const f32x4*)(slabs + (long long)z * slab_stride + o4 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) red[zl][ol][q] = s[q];
    __syncthreads();
    if (zl == 0 && o4 < total4) {
        for (int r = 1; r < ZL; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += red[r][ol][q];
        const long long e = o4 * 4;
        const int i = (int)(e / J), j = (int)(e % J);
        const long long base = (long long)(i / cdiv) * s_hi + (long long)(i % cdiv) * s_lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[base + (long long)(j + q) * s_j] = s[q];
    }
}

// Conv weight gradient: out[co][c][jj] = sum_z slabs[z][jj*cin + c][co]   (J = cout contiguous in the slabs).
// A 32(c) x 32(co) x kw tile goes through LDS so that both the slab reads (along co) and the weight writes (along
// (c, jj) for one co) are coalesced.  grid (cin/32, cout/32), kw <= 8.
__global__ __launch_bounds__(256) void reduce_conv_w_kernel(const float* __restrict__ slabs, float* __restrict__ out, int cin,
                                                            int cout, int kw, int nslab, long long slab_stride) {
    __shared__ float t[8][32][33];
    const int c0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    {
        // a thread owns FOUR consecutive output channels of one input channel (16-byte loads, four slabs in flight: it was one float per
        // load, 256 dependent-ish loads per thread, 58 us for the 64 MB of layer 2's slabs); per element the same four accumulators over
        // the slabs z = k mod 4 and the same final sum: identical results
        const int q = threadIdx.x & 7, cl = threadIdx.x >> 3;
        for (int jj = 0; jj < kw; ++jj) {
            const long long off = (long long)(jj * cin + c0 + cl) * cout + co0 + 4 * q;
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
            int z = 0;
            for (; z + 3 < nslab; z += 4) {
                const f32x4 a0 = *(const f32x4*)(slabs + (long long)z * slab_stride + off);
                const f32x4 a1 = *(const f32x4*)(slabs + (long long)(z + 1) * slab_stride + off);
                const f32x4 a2 = *(const f32x4*)(slabs + (long long)(z + 2) * slab_stride + off);
                const f32x4 a3 = *(const f32x4*)(slabs + (long long)(z + 3) * slab_stride + off);
                s0 += a0; s1 += a1; s2 += a2; s3 += a3;
            }
            for (; z < nslab; ++z) s0 += *(const f32x4*)(slabs + (long long)z * slab_stride + off);
            const f32x4 r4 = (s0 + s1) + (s2 + s3);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[jj][cl][4 * q + e] = r4[e];
        }
    }
    __syncthreads();
    const int per_co = 32 * kw;
    for (int idx = threadIdx.x; idx < 32 * per_co; idx += 256) {
        const int col = idx / per_co, rem = idx % per_co;
        const int cl = rem / kw, jj = rem % kw;
        out[(long long)(co0 + col) * cin * kw + (long long)(c0 + cl) * kw + jj] = t[jj][cl][col];
    }
}

// Weight gradient of an nn.Conv2d from split-K slabs in the layouts the grid convolutions produce (scalogram_engine._Conv._wgrad):
//   out[((co cin + c) kh + dh) kw + dw] = sum_z sum_{g < G} slab[z][dw s_dw + (dh + g) s_dh + c s_c + g s_g + co]
// G = 1: slabs [kw][kh][cin][cout] of the gathered-window GEMM (one batch entry per kernel column).  G > 1: the row-grouped tall (k,1)
// kernels, slab [(r, c)][(g, co)] = sum_R X[G R + r][c] dY[G R + g][co]: dW[co][c][j] collects the G diagonals r = j + g.
// One thread per output element, co fastest (contiguous slab reads); fixed summation order (z outer, g inner), four accumulators.
__global__ __launch_bounds__(256) void reduce_conv_w2d_kernel(const float* __restrict__ slabs, float* __restrict__ out, int nslab,
                                                              long long slab_stride, int cout, int cin, int kh, int kw, long long s_dw,
                                                              long long s_dh, long long s_c, int G, long long s_g) {
    const long long total = (long long)cout * cin * kh * kw;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int co = (int)(t % cout);
    long long r = t / cout;
    const int dw = (int)(r % kw); r /= kw;
    const int dh = (int)(r % kh);
    const int c = (int)(r / kh);
    const float* base = slabs + dw * s_dw + dh * s_dh + c * s_c + co;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int z = 0; z < nslab; ++z) {
        const float* p = base + (long long)z * slab_stride;
        int g = 0;
        for (; g + 3 < G; g += 4) {
            a0 += p[(long long)g * (s_dh + s_g)];
            a1 += p[(long long)(g + 1) * (s_dh + s_g)];
            a2 += p[(long long)(g + 2) * (s_dh + s_g)];
            a3 += p[(long long)(g + 3) * (s_dh + s_g)];
        }
        for (; g < G; ++g) a0 += p[(long long)g * (s_dh + s_g)];
    }
    out[(((long long)co * cin + c) * kh + dh) * kw + dw] = (a0 + a1) + (a2 + a3);
}

// Column sums of a [M][N] T matrix into per-block partial slabs [gridDim.x][N] (f32); reduced by reduce_slabs.
// A thread owns a group of CW columns — 16 bytes of a row: 8 bf16 or 4 f32 (VEC16), else 4 columns — and walks the rows of its row
// lane four at a time, all four loads in flight before the first add (a wave then has 4 KiB in flight instead of the 512 B of
// one 8-byte load per lane: 1.26 M x 128 bf16 took 209 us = 1.5 TB/s in that form); with fewer than 256 groups several row lanes
// share a group and are combined through LDS in a fixed order.
template <typename T, bool VEC16>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ slabs, int M, int N,
                                                     long long ldx, int rows_per_block) {
    constexpr int CW = VEC16 ? 16 / (int)sizeof(T) : 4;
    __shared__ float red[256 * CW];
    const int ncg = N / CW;
    const int tid = threadIdx.x;
    const int m_begin = blockIdx.x * rows_per_block;
    const int m_end = min(M, m_begin + rows_per_block);
    auto load_cw = [&](const T* src, float* v) {
        if constexpr (VEC16 && sizeof(T) == 2) {
            const uint4 w = *(const uint4*)src;
            const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = __builtin_bit_cast(float, u[e] << 16);
                v[2 * e + 1] = __builtin_bit_cast(float, u[e] & 0xffff0000u);
            }
        } else {
            const f32x4 w = load4(src);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = w[e];
        }
    };
    for (int cg0 = 0; cg0 < ncg; cg0 += 256) {
        const int lpr = min(ncg - cg0, 256);     // column groups handled in this pass
        const int nrl = 256 / lpr;               // row lanes per column group
        const int cg = cg0 + tid % lpr;
        const int rl = tid / lpr;
        const bool active = rl < nrl;
        float s[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) s[e] = 0.f;
        if (active) {
            const T* col = X + cg * CW;
            int m = m_begin + rl;
            for (; m + 3 * nrl < m_end; m += 4 * nrl) {
                float v[4][CW];
#pragma unroll
                for (int u = 0; u < 4; ++u) load_cw(col + (long long)(m + u * nrl) * ldx, v[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < CW; ++e) s[e] += v[u][e];
            }
            for (; m < m_end; m += nrl) {
                float v[CW];
                load_cw(col + (long long)m * ldx, v);
#pragma unroll
                for (int e = 0; e < CW; ++e) s[e] += v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < CW; ++e) red[tid * CW + e] = s[e];
        __syncthreads();
        if (rl == 0) {
            for (int r = 1; r < nrl; ++r)
#pragma unroll
                for (int e = 0; e < CW; ++e) s[e] += red[(tid + r * lpr) * CW + e];
#pragma unroll
            for (int e = 0; e < CW; ++e) slabs[(long long)blockIdx.x * N + cg * CW + e] = s[e];
        }
        __syncthreads();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ launchers
// f32 TN for tiny outputs reduced over very many rows (C[i][j] = sum_m A[m][i] B[m][j] with I <= 32, J <= 64: the weight
// gradient of the scalogram encoder's first convolution, 18 x 32 over 5 M positions).  A 128-wide MFMA tile would be almost
// empty; here a workgroup streams its slab of rows through LDS 64 at a time and every thread keeps 4 consecutive j of one
// i in registers (VALU fma; the kernel is bound by reading A and B once from HBM).
__global__ __launch_bounds__(256) void gemm_tn_skinny_f32_kernel(GemmTN p) {
    __shared__ __attribute__((aligned(16))) float sa[64][32 + 4];
    __shared__ __attribute__((aligned(16))) float sb[64][64 + 4];
    const int tid = threadIdx.x;
    const int split = blockIdx.x;
    const int m_begin = split * p.m_chunk, m_end = min(p.M, m_begin + p.m_chunk);
    const float* Ab = (const float*)p.A;
    const float* Bb = (const float*)p.B;
    const int I4 = (p.I + 3) / 4, J4 = p.J / 4;
    const int oi = tid / J4, oj4 = tid % J4;                 // output (i = oi, j = 4*oj4 ..) for tid < I * J4
    const bool owner = oi < p.I;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int mb = m_begin; mb < m_end; mb += 64) {
        __syncthreads();
        for (int q = tid; q < 64 * I4; q += 256) {
            const int r = q / I4, c4 = q % I4, m = mb + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end) v = *(const f32x4*)(Ab + row_off(m, p.a_rpi, p.a_item, p.lda) + c4 * 4);      // lda >= 4*I4 (host)
            *(f32x4*)&sa[r][c4 * 4] = v;
        }
        for (int q = tid; q < 64 * J4; q += 256) {
            const int r = q / J4, c4 = q % J4, m = mb + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end) v = *(const f32x4*)(Bb + row_off(m, p.b_rpi, p.b_item, p.ldb) + c4 * 4);
            *(f32x4*)&sb[r][c4 * 4] = v;
        }
        __syncthreads();
        if (owner) {
#pragma unroll 8
            for (int r = 0; r < 64; ++r) {
                const float a = sa[r][oi];
                const f32x4 b = *(const f32x4*)&sb[r][oj4 * 4];
                acc += a * b;
            }
        }
    }
    if (owner) *(f32x4*)((float*)p.C + (long long)split * p.slab_stride + (long long)oi * p.ldc + oj4 * 4) = acc;
}

// f32 NT for tiny N and K over very many rows (C[m][n] = bias[n] + sum_k A[m][k] Bt[n][k] with N in {8, 16, 32}, K <= 32: the
// first convolution and the 1x1 residual projection of the scalogram encoder, 32 outputs from 18 / 2 inputs at 5 M positions).
// Bound by reading A and writing C once; a 128 x 128 MFMA tile is almost empty there (2.7 TF/s).  N/8 lanes share a row: each
// reads the row's K floats (L1 broadcast) and owns 8 consecutive outputs; the weights sit transposed in LDS.
__global__ __launch_bounds__(256) void gemm_nt_skinny_f32_kernel(GemmNT p) {
    __shared__ __attribute__((aligned(16))) float wt[32][32];
    const int tid = threadIdx.x, K = p.K, N = p.N, NG = N / 8;
    const float* Ab = (const float*)p.A;
    const float* Bb = (const float*)p.Bt;
    float* Cb = (float*)p.C;
    for (int i = tid; i < N * K; i += 256) {
        const int n = i / K, k = i % K;
        wt[k][n] = Bb[(long long)n * p.ldb + k];
    }
    __syncthreads();
    const int lg = tid % NG, rl = tid / NG, rpp = 256 / NG;
    float b8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b8[j] = p.bias ? p.bias[lg * 8 + j] : 0.f;
    const bool relu = p.flags & GEMM_RELU;
    for (long long m = (long long)blockIdx.x * rpp + rl; m < p.M; m += (long long)gridDim.x * rpp) {
        const float* ar = Ab + m * p.lda;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = b8[j];
        for (int k4 = 0; k4 < K / 4; ++k4) {
            const f32x4 a = *(const f32x4*)(ar + 4 * k4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 w0 = *(const f32x4*)&wt[4 * k4 + e][lg * 8], w1 = *(const f32x4*)&wt[4 * k4 + e][lg * 8 + 4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = fmaf(a[e], w0[j], acc[j]);
                    acc[4 + j] = fmaf(a[e], w1[j], acc[4 + j]);
                }
            }
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = relu_f(acc[j]);
        }
        const bool row_valid = (p.c_rpi == 0) || ((int)(m % p.c_rpi) < p.c_valid);
        if (!row_valid) {
            if (p.flags & GEMM_SKIP_PAD_ROWS) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        }
        float* cr = Cb + row_off((int)m, p.c_rpi, p.c_item, p.ldc) + lg * 8;
        *(f32x4*)cr = (f32x4){acc[0], acc[1], acc[2], acc[3]};
        *(f32x4*)(cr + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
    }
}

// start stagger of the 256x256 kernel (GemmNT::stagger) in 1/64 of a tile time: applied to launches with a ReLU-backward mask and
// at least three rounds of tiles, whose every tile ends in a 2 x 32 MB burst (mask read + store) when all CUs run in step
// (tools/nt_ab.py --stagger: layer-3 data gradient 285 -> 278 us, layer-2 1036 -> 1021 us; no gain without a mask)
int g_nt_stagger64 = 32;
int g_nt_probe = 0, g_nt_probe_taps = 1;
int g_nt_wt = 2;       // output stores of the NT fast kernels: 0 plain, 1 written through at agent scope, 2 at system scope (default)

static bool nt_persist_enabled() {          // CPC_NT_PERSIST=0: one workgroup per tile everywhere (A/B switch)
    static const bool on = [] { const char* v = getenv("CPC_NT_PERSIST"); return !(v && v[0] == '0'); }();
    return on;
}

int launch_gemm_nt(const GemmNT& p, int dtype, int batch, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || batch <= 0) return CPC_EINVAL;
    const int ch = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if (p.K % ch || p.lda % ch || p.ldb % ch || p.ldc % 4) return CPC_EINVAL;
    if ((p.N % 4) && (p.bias || p.mask)) return CPC_EINVAL;       // vector epilogue loads need whole groups of 4
    if (p.ldc < (p.N + 3) / 4 * 4 && p.c_rpi == 0 && p.M > 1) return CPC_EINVAL;
    if (p.a_rpi && p.a_item % ch) return CPC_EINVAL;
    if (p.b_rpi && p.b_item % ch) return CPC_EINVAL;
    if (p.c_rpi && p.c_item % 4) return CPC_EINVAL;
    const bool of32 = p.flags & GEMM_OUT_F32;
    if (dtype == CPC_DTYPE_F32 && batch == 1 && (p.N == 8 || p.N == 16 || p.N == 32) && p.K <= 32 && p.K % 4 == 0 && !p.mask &&
        p.a_rpi == 0 && p.b_rpi == 0 && p.M >= 4096 && p.ldc % 4 == 0 && (p.c_rpi == 0 || p.c_item % 4 == 0) &&
        !(p.flags & GEMM_FORCE_GENERIC) && p.m_off == 0 && ((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.C % 16 == 0)) {
        const long long rows_pp = 256 / (p.N / 8);
        const int blocks = (int)std::min<long long>(256 * 8, (p.M + rows_pp - 1) / rows_pp);
        hipLaunchKernelGGL(gemm_nt_skinny_f32_kernel, dim3(blocks), dim3(256), 0, stream, p);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    const bool fast = (p.K % (8 * ch) == 0) && !(p.flags & GEMM_FORCE_GENERIC);
    // two-level rows and K ranges exist in the fast kernels only; a K range needs items to index and the storage K order
    const bool banded = p.a_rpi2 || p.c_rpi2 || p.k_ranges;
    if (banded && (!fast || (p.flags & (GEMM_EPI_CONV1 | GEMM_NO_DMA)) || p.m_off || (p.k_taps > 1 && !p.k_taps_linear) || p.colsum_slabs)) return CPC_EINVAL;
    if ((p.a_rpi2 && (!p.a_rpi || p.a_item2 % ch)) || (p.c_rpi2 && (!p.c_rpi || p.c_item2 % 4)) || (p.k_ranges && !p.a_rpi)) return CPC_EINVAL;
    // 256x256 tiles only where they fill the chip: below ~200 of them (e.g. the 3072 x 3072 all-timesteps score matrix: 144)
    // four times as many 128x128 tiles keep more CUs busy
    const long long big_tiles = (long long)((p.M - p.m_off + 255) / 256) * ((p.N + 255) / 256);
    const bool big = fast && dtype == CPC_DTYPE_BF16 && p.N >= 256 && !(p.flags & GEMM_SMALL_TILE) &&
                     ((p.M >= 1024 && (big_tiles >= 200 || big_tiles * batch >= 200)) || ((p.flags & GEMM_BIG_TILE) && p.M >= 256));
    GemmNT q = p;
    if (g_nt_wt == 1) q.flags |= GEMM_WT_AGENT; else if (g_nt_wt == 2) q.flags |= GEMM_WT_SYSTEM;
    // overlapped-row A operand (strided-conv view): visit K tap-innermost, see GemmNT::k_taps
    const int bk = 8 * ch;
    if (fast && !p.k_ranges && p.k_taps == 0 && p.lda > 0 && p.lda < p.K && p.K % p.lda == 0 && p.lda % bk == 0 && !(p.flags & GEMM_LINEAR_K)) {
        q.k_taps = (int)(p.K / p.lda);
        q.k_tap_stride = p.lda;
    }
    if (g_nt_probe == 32) { q.k_taps = g_nt_probe_taps; q.k_tap_stride = p.K / g_nt_probe_taps; }      // (timing probe: see DBG 32)
    if (q.k_taps > 1 && (!fast || (long long)q.k_taps * q.k_tap_stride != p.K || q.k_tap_stride % bk || q.k_tap_stride_a % ch)) return CPC_EINVAL;
    if (q.k_tap_stride_a && q.k_taps <= 1) return CPC_EINVAL;
    if (fast && dtype == CPC_DTYPE_BF16 && !of32 && !(p.flags & GEMM_NARROW_EPI) && p.N % 8 == 0 && p.ldc % 8 == 0 &&
        p.c_item % 8 == 0 && p.c_item2 % 8 == 0 && p.c_batch % 8 == 0 && ((uintptr_t)p.C % 16 == 0) && (!p.mask || (uintptr_t)p.mask % 16 == 0))
        q.flags |= GEMM_WIDE_EPI;
    if (big && g_nt_stagger64 > 0 && (p.mask || p.mask_bits) && big_tiles * batch >= 3 * 256) {
        // a tile takes about nk * 3600 + 20000 cycles; the largest phase (7) starts g_nt_stagger64 / 64 of that late
        const long long tile_cycles = (long long)(p.K / bk) * 3600 + 20000;
        q.stagger = (int)std::max<long long>(1, tile_cycles * g_nt_stagger64 / 64 / 7 / 4096);
    }
    const int tbm = big ? 256 : BM, tbn = big ? 256 : BN;
    const int numM = (p.M - p.m_off + tbm - 1) / tbm;
    const int numN = (p.N + tbn - 1) / tbn;
    if (!fast && p.m_off) return CPC_EINVAL;
    if ((p.flags & GEMM_KRANGE_EXACT) && p.k_ranges) {
        const long long per = (long long)p.a_rpi * (p.a_rpi2 > 0 ? p.a_rpi2 : 1);
        if (per <= 0 || per % tbm) return CPC_EINVAL;         // a tile would straddle two range indices (see the flag)
    }
    const long long blocks = nt_grid_blocks(numM, numN);
    if (blocks > 0x7fffffffLL) return CPC_EINVAL;
    dim3 grid((unsigned)blocks, 1, batch);
    const bool dma = !(p.flags & GEMM_NO_DMA);       // LDS-DMA staging (default) vs register staging (A-B check)
#define NT_LAUNCH(TT, TOO, WMM, WNN, TII, TJJ, NTH, ARG)                                                              \
    do {                                                                                                             \
        if (dma) hipLaunchKernelGGL((gemm_nt_fast_kernel<TT, TOO, WMM, WNN, TII, TJJ, true>), grid, dim3(NTH), 0, stream, ARG);  \
        else hipLaunchKernelGGL((gemm_nt_fast_kernel<TT, TOO, WMM, WNN, TII, TJJ, false>), grid, dim3(NTH), 0, stream, ARG);     \
    } while (0)
    if (p.flags & GEMM_EPI_CONV1) {
        // fused layer-1 weight gradient: only the 256x256 LDS-DMA kernel with the LDS-staged epilogue has that variant
        if (!(big && dma && (q.flags & GEMM_WIDE_EPI) && (p.mask || p.mask_bits) && batch == 1 && p.c1_x && p.c1_slabs && p.c1_sub > 0 &&
              p.N % p.c1_sub == 0 && (p.N / p.c1_sub) % 256 == 0 && p.c1_kw >= 1 && p.c1_kw <= 15 && p.c1_rpi > 0))
            return CPC_EINVAL;
        hipLaunchKernelGGL((gemm_nt_fast_kernel<bf16_t, bf16_t, 2, 4, 8, 4, true, true>), grid, dim3(512), 0, stream, q);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    // register epilogue (see DIRECT above): bf16 in / out, LDS-DMA staging, whole 16-byte column groups
    // (measured, tools/nt_ab.py: conv forward launches +4.5 ... 9 %; with a ReLU-backward mask the 64-byte row segments of the
    // mask reads cost what the missing LDS round trip saves, so masked launches keep the LDS-staged epilogue)
    const bool direct = fast && dma && dtype == CPC_DTYPE_BF16 && (q.flags & GEMM_WIDE_EPI) && !(p.flags & GEMM_NO_PERS) &&
                        (!p.bias || (uintptr_t)p.bias % 16 == 0) && !p.mask_bits && (!p.mask || (p.flags & GEMM_DIRECT_MASK));
    // per-tile column sums: 256x256 bf16 LDS-staged epilogue only (a launch with a mask and without GEMM_DIRECT_MASK)
    if (p.colsum_slabs && !(big && dma && dtype == CPC_DTYPE_BF16 && !of32 && (q.flags & GEMM_WIDE_EPI) && !direct && batch == 1 && p.m_off == 0 &&
                            !(p.flags & GEMM_EPI_CONV1)))
        return CPC_EINVAL;
    // the byte-per-8-elements masks exist in the 256x256 bf16 kernels with the LDS-staged epilogue only
    if (p.mask_bits && !(big && dma && dtype == CPC_DTYPE_BF16 && !of32 && (q.flags & GEMM_WIDE_EPI) && p.N % 256 == 0 && p.ldc % 32 == 0 &&
                         p.c_item % 32 == 0 && p.c_batch % 32 == 0 && (uintptr_t)p.mask_bits % 4 == 0))
        return CPC_EINVAL;
    if (dtype == CPC_DTYPE_BF16) {
        if (big) {
            if (of32) NT_LAUNCH(bf16_t, float, 2, 4, 8, 4, 512, q);
            else if (direct && g_nt_probe == 1) hipLaunchKernelGGL((gemm_nt_fast_kernel<bf16_t, bf16_t, 2, 4, 8, 4, true, false, true, 1>), grid, dim3(512), 0, stream, q);
            else if (direct && g_nt_probe == 2) hipLaunchKernelGGL((gemm_nt_fast_kernel<bf16_t, bf16_t, 2, 4, 8, 4, true, false, true, 2>), grid, dim3(512), 0, stream, q);
            else if (direct && g_nt_probe == 16) hipLaunchKernelGGL((gemm_nt_fast_kernel<bf16_t, bf16_t, 2, 4, 8, 4, true, false, true, 16>), grid, dim3(512), 0, stream, q);
            else if (direct && g_nt_probe == 32) hipLaunchKernelGGL((gemm_nt_fast_kernel<bf16_t, bf16_t, 2, 4, 8, 4, true, false, true, 32>), grid, dim3(512), 0, stream, q);
            else if (direct && g_nt_probe == 64) hipLaunchKernelGGL((gemm_nt_fast_kernel<bf16_t, bf16_t, 2, 4, 8, 4, true, false, true, 64>), grid, dim3(512), 0, stream, q);
            // (short K only: a tile of up to 16 stages spends a third of its time outside the K loop.  The convolution launches of the train
            // step, K = 2 048 / 4 096, were measured SLOWER this way — 4.675 against 4.605 ms per configs[1] step, interleaved — the dispatcher's
            // dynamic tile order balances the CUs better than a fixed walk, and their fixed cost is under a tenth of a tile.)
            else if (direct && nt_persist_enabled() && !p.mask && !banded && batch == 1 && p.m_off == 0 && p.K / bk >= 2 && p.K / bk <= 16 && blocks > 2 * 256 &&
                     p.M % 32 == 0 && (p.a_rpi == 0 || p.a_rpi % 32 == 0) && p.b_rpi == 0 && p.N % 256 == 0) {
                static const int ps = [] { const char* v = getenv("CPC_NT_PERSIST_STAGGER"); return v ? atoi(v) : 64; }();
                const long long tile_cycles = (long long)(p.K / bk) * 3600 + 20000;
                q.stagger = ps > 0 ? (int)std::max<long long>(1, tile_cycles * ps / 64 / 7 / 4096) : 0;
                hipLaunchKernelGGL((gemm_nt_persist_kernel<0>), dim3(256), dim3(512), 0, stream, q);      // one workgroup per CU walks the tiles
            }
            else if (direct) hipLaunchKernelGGL((gemm_nt_fast_kernel<bf16_t, bf16_t, 2, 4, 8, 4, true, false, true>), grid, dim3(512), 0, stream, q);
            else NT_LAUNCH(bf16_t, bf16_t, 2, 4, 8, 4, 512, q);
        } else if (fast) {
            if (of32) NT_LAUNCH(bf16_t, float, 2, 2, 4, 4, 256, q);
            else if (direct) hipLaunchKernelGGL((gemm_nt_fast_kernel<bf16_t, bf16_t, 2, 2, 4, 4, true, false, true>), grid, dim3(256), 0, stream, q);
            else NT_LAUNCH(bf16_t, bf16_t, 2, 2, 4, 4, 256, q);
        } else {
            if (of32) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, float>), grid, dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, p);
        }
    } else if (dtype == CPC_DTYPE_F32) {
        if (fast) NT_LAUNCH(float, float, 2, 2, 4, 4, 256, q);
        else hipLaunchKernelGGL((gemm_nt_kernel<float, float>), grid, dim3(256), 0, stream, p);
    } else {
        return CPC_EINVAL;
    }
#undef NT_LAUNCH
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// Score contraction with the column log-sum-exp partials fused into its epilogue (bf16; include/cpc_hip.h, cpc_score_lse).
int launch_score_lse(const void* P, const void* Tg, void* Sb, float* pm, float* ps, float* valid, int M, int N, int E, long long ldp,
                     long long ldt, long long lds_, int diag_off, hipStream_t stream) {
    if (!P || !Tg || !pm || !ps || M <= 0 || N <= 0 || E < 128) return CPC_EINVAL;
    if (M % 256 || N % 256 || E % 64 || ldp % 8 || ldt % 8 || ldp < E || ldt < E) return CPC_EINVAL;
    if ((uintptr_t)P % 16 || (uintptr_t)Tg % 16) return CPC_EINVAL;
    if (Sb && (lds_ % 4 || lds_ < N || (uintptr_t)Sb % 16)) return CPC_EINVAL;
    GemmNT q = {};
    q.A = P; q.Bt = Tg; q.C = Sb;
    q.M = M; q.N = N; q.K = E;
    q.lda = ldp; q.ldb = ldt; q.ldc = Sb ? lds_ : N;
    q.lse_pm = pm; q.lse_ps = ps; q.lse_valid = valid; q.lse_diag_off = diag_off;
    if (g_nt_wt == 1) q.flags |= GEMM_WT_AGENT; else if (g_nt_wt == 2) q.flags |= GEMM_WT_SYSTEM;
    const long long blocks = nt_grid_blocks(M / 256, N / 256);
    if (blocks > 0x7fffffffLL) return CPC_EINVAL;
    static const int ps_ = [] { const char* v = getenv("CPC_NT_PERSIST_STAGGER"); return v ? atoi(v) : 64; }();
    const long long tile_cycles = (long long)(E / 64) * 3600 + 20000;
    q.stagger = (ps_ > 0 && blocks > 2 * 256) ? (int)std::max<long long>(1, tile_cycles * ps_ / 64 / 7 / 4096) : 0;
    hipLaunchKernelGGL((gemm_nt_persist_kernel<0, true>), dim3((unsigned)std::min<long long>(256, blocks)), dim3(512), 0, stream, q);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_gemm_tn(const GemmTN& p, int dtype, int nsplit, int batch, hipStream_t stream) {
    if (p.M <= 0 || p.I <= 0 || p.J <= 0 || batch <= 0 || nsplit <= 0) return CPC_EINVAL;
    const int ch = dtype == CPC_DTYPE_BF16 ? 8 : 4;
    if (p.J % ch || p.lda % ch || p.ldb % ch || p.ldc % 4) return CPC_EINVAL;
    if ((p.I % ch) && p.lda < (p.I + ch - 1) / ch * ch) return CPC_EINVAL;   // ragged I needs padded A rows
    if (p.a_rpi && p.a_item % ch) return CPC_EINVAL;
    if (p.b_rpi && p.b_item % ch) return CPC_EINVAL;
    if (nsplit > 1 && (p.m_chunk <= 0 || (long long)p.m_chunk * nsplit < p.M)) return CPC_EINVAL;
    if (nsplit > 1 && !(p.flags & GEMM_OUT_F32)) return CPC_EINVAL;   // slabs are f32
    if (p.c_rpi && (p.c_item % 4 || nsplit > 1)) return CPC_EINVAL;
    const bool of32 = p.flags & GEMM_OUT_F32;
    if (p.a_rpi2 && (!p.a_rpi || p.a_item2 % ch)) return CPC_EINVAL;
    const int eff_chunk = nsplit > 1 ? p.m_chunk : p.M;
    if (dtype == CPC_DTYPE_F32 && batch == 1 && p.I <= 32 && p.J <= 64 && p.I * (p.J / 4) <= 256 && p.c_rpi == 0 &&
        p.lda >= (p.I + 3) / 4 * 4 && !(p.flags & GEMM_FORCE_GENERIC)) {
        GemmTN q = p;
        q.m_chunk = eff_chunk;
        hipLaunchKernelGGL(gemm_tn_skinny_f32_kernel, dim3(nsplit), dim3(256), 0, stream, q);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    const bool fast = dtype == CPC_DTYPE_BF16 && !(p.flags & (GEMM_FORCE_GENERIC | GEMM_TN_NO_TR)) && (p.I % 8 == 0) &&
                      p.I >= 8 && p.J >= 8;
    const bool big = fast && p.I >= 256 && p.J >= 256 && eff_chunk >= 1024 && !(p.flags & GEMM_SMALL_TILE);
    const int tb = big ? 256 : 128;
    const int numI = (p.I + tb - 1) / tb, numJ = (p.J + tb - 1) / tb;
    dim3 grid(numI * numJ, nsplit, batch);
    if (fast) {
        GemmTN q = p;
        q.m_chunk = eff_chunk;
        grid = dim3(nsplit, numI * numJ, batch);
        const bool plain = p.a_rpi == 0 && p.b_rpi == 0;
        const bool tdma = (plain || (p.a_item % 8 == 0 && p.b_item % 8 == 0 && p.M < (1 << 22))) && !(p.flags & GEMM_NO_DMA) && p.lda % 8 == 0 && p.ldb % 8 == 0 && p.I % 8 == 0 && p.J % 8 == 0 &&
                          ((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.B % 16 == 0) && p.a_batch % 8 == 0 && p.b_batch % 8 == 0;
        if (p.a_rpi2 && !tdma) return CPC_EINVAL;          // the second row level exists in the LDS-DMA kernel only
#define TN_LAUNCH(WII, WJJ, TII, TJJ, NTH)                                                                               \
    do {                                                                                                                 \
        if (tdma && plain) {                                                                                             \
            if (of32) hipLaunchKernelGGL((gemm_tn_dma_kernel<float, WII, WJJ, TII, TJJ, false>), grid, dim3(NTH), 0, stream, q);   \
            else hipLaunchKernelGGL((gemm_tn_dma_kernel<bf16_t, WII, WJJ, TII, TJJ, false>), grid, dim3(NTH), 0, stream, q);      \
        } else if (tdma) {                                                                                               \
            if (of32) hipLaunchKernelGGL((gemm_tn_dma_kernel<float, WII, WJJ, TII, TJJ, true>), grid, dim3(NTH), 0, stream, q);    \
            else hipLaunchKernelGGL((gemm_tn_dma_kernel<bf16_t, WII, WJJ, TII, TJJ, true>), grid, dim3(NTH), 0, stream, q);       \
        } else if (plain) {                                                                                              \
            if (of32) hipLaunchKernelGGL((gemm_tn_fast_kernel<float, WII, WJJ, TII, TJJ, true>), grid, dim3(NTH), 0, stream, q);   \
            else hipLaunchKernelGGL((gemm_tn_fast_kernel<bf16_t, WII, WJJ, TII, TJJ, true>), grid, dim3(NTH), 0, stream, q);      \
        } else {                                                                                                         \
            if (of32) hipLaunchKernelGGL((gemm_tn_fast_kernel<float, WII, WJJ, TII, TJJ, false>), grid, dim3(NTH), 0, stream, q);  \
            else hipLaunchKernelGGL((gemm_tn_fast_kernel<bf16_t, WII, WJJ, TII, TJJ, false>), grid, dim3(NTH), 0, stream, q);     \
        }                                                                                                                \
    } while (0)
        if (big) {
            TN_LAUNCH(2, 4, 8, 4, 512);
        } else {
            TN_LAUNCH(2, 2, 4, 4, 256);
        }
#undef TN_LAUNCH
    } else if (p.a_rpi2) {
        return CPC_EINVAL;
    } else if (dtype == CPC_DTYPE_BF16) {
        if (of32) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, float>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, p);
    } else if (dtype == CPC_DTYPE_F32) {
        hipLaunchKernelGGL((gemm_tn_kernel<float, float>), grid, dim3(256), 0, stream, p);
    } else {
        return CPC_EINVAL;
    }
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_reduce_slabs(const float* slabs, float* out, int I, int J, int nslab, long long slab_stride, int cdiv,
                        long long s_j, long long s_hi, long long s_lo, hipStream_t stream) {
    if (I <= 0 || J <= 0 || J % 4 || nslab <= 0 || cdiv <= 0) return CPC_EINVAL;
    const long long total4 = (long long)I * J / 4;
    if (total4 <= 512 && nslab >= 512) {
        // very small outputs from very many slabs (BatchNorm sums and bias gradients of the scalogram model: 32 ... 512 values from
        // 1024 ... 2048 slabs): one workgroup per output float4 with 256 slab lanes — with 64 lanes such a launch was two to 32
        // workgroups whose threads each walked 16 ... 32 dependent loads (scalogram step: 2.9 ms in these reductions)
        hipLaunchKernelGGL(reduce_slabs_small_kernel<256>, dim3((unsigned)total4), dim3(256), 0, stream, slabs, out, I, J, nslab, slab_stride,
                           cdiv, s_j, s_hi, s_lo);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    if (total4 <= 16384 && nslab >= 16) {     // (from 16 slabs: the predictor's data gradient, 24 slabs of 256 x 256: 10 -> 5 us)
        if (nslab >= 128)            // 64 slab lanes: a serial chain of nslab / 64 loads per thread (228-912 slabs: the per-tile column sums)
            hipLaunchKernelGGL(reduce_slabs_small_kernel<64>, dim3((unsigned)((total4 + 3) / 4)), dim3(256), 0, stream, slabs, out, I,
                               J, nslab, slab_stride, cdiv, s_j, s_hi, s_lo);
        else
            hipLaunchKernelGGL(reduce_slabs_small_kernel<16>, dim3((unsigned)((total4 + 15) / 16)), dim3(256), 0, stream, slabs, out, I,
                               J, nslab, slab_stride, cdiv, s_j, s_hi, s_lo);
        CPC_CHECK_LAUNCH();
        return CPC_OK;
    }
    const int blocks = (int)min((long long)2048, (total4 + 255) / 256);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream, slabs, out, I, J, nslab, slab_stride,
                       cdiv, s_j, s_hi, s_lo);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_reduce_conv_w2d(const float* slabs, float* out, int nslab, long long slab_stride, int cout, int cin, int kh, int kw,
                           long long s_dw, long long s_dh, long long s_c, int G, long long s_g, hipStream_t stream) {
    if (!slabs || !out || nslab <= 0 || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || G <= 0) return CPC_EINVAL;
    const long long total = (long long)cout * cin * kh * kw;
    if ((total + 255) / 256 > 0x7fffffffLL) return CPC_EINVAL;
    hipLaunchKernelGGL(reduce_conv_w2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, slabs, out, nslab, slab_stride, cout, cin,
                       kh, kw, s_dw, s_dh, s_c, G, s_g);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_colsum(const void* X, float* slabs, int M, int N, long long ldx, int dtype, int nblocks, hipStream_t stream) {
    if (M <= 0 || N <= 0 || N % 4 || nblocks <= 0 || ldx % 4) return CPC_EINVAL;
    const int rpb = (M + nblocks - 1) / nblocks;
    // 16-byte loads where every row chunk is 16-byte aligned (bf16: N and ldx multiples of 8)
    if (dtype == CPC_DTYPE_BF16) {
        if (N % 8 == 0 && ldx % 8 == 0 && (uintptr_t)X % 16 == 0)
            hipLaunchKernelGGL((colsum_kernel<bf16_t, true>), dim3(nblocks), dim3(256), 0, stream, (const bf16_t*)X, slabs, M, N, ldx, rpb);
        else
            hipLaunchKernelGGL((colsum_kernel<bf16_t, false>), dim3(nblocks), dim3(256), 0, stream, (const bf16_t*)X, slabs, M, N, ldx, rpb);
    } else if (dtype == CPC_DTYPE_F32) {
        if ((uintptr_t)X % 16 == 0)
            hipLaunchKernelGGL((colsum_kernel<float, true>), dim3(nblocks), dim3(256), 0, stream, (const float*)X, slabs, M, N, ldx, rpb);
        else
            hipLaunchKernelGGL((colsum_kernel<float, false>), dim3(nblocks), dim3(256), 0, stream, (const float*)X, slabs, M, N, ldx, rpb);
    } else {
        return CPC_EINVAL;
    }
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_reduce_conv_w(const float* slabs, float* out, int cin, int cout, int kw, int nslab, long long slab_stride,
                         hipStream_t stream) {
    if (cin <= 0 || cout <= 0 || kw <= 0 || nslab <= 0) return CPC_EINVAL;
    if (cin % 32 || cout % 32 || kw > 8 || (uintptr_t)slabs % 16 || slab_stride % 4)      // generic permuting reduce
        return launch_reduce_slabs(slabs, out, kw * cin, cout, nslab, slab_stride, cin, (long long)cin * kw, 1, kw, stream);
    hipLaunchKernelGGL(reduce_conv_w_kernel, dim3(cin / 32, cout / 32), dim3(256), 0, stream, slabs, out, cin, cout, kw, nslab,
                       slab_stride);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
