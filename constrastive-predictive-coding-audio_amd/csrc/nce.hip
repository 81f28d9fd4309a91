// InfoNCE loss of ContrastiveEstimationTrainer.train (default branch: same-step scores only), forward + analytic
// gradient with respect to the linear scores.
//
//   S[k][b][b'] = <predicted_z[b,k,:], targets[b',:,k]>      (computed by a batched gemm_nt, f32)
//   sp = softplus(S) (softplus score) or S (linear score)
//   loss = -mean_{b,k} sp[k][b][b] + mean_{k,b'} logsumexp_b sp[k][b][b'] + reg * mean_{b,b'} (mean_k sp[k][b][b'])^2
//   dL/dS = dsp * (softplus ? sigmoid(S) : 1),
//   dsp[k][b][b'] = (softmax_b(sp[k][:,b'])[b] - [b == b']) / (B K) + 2 reg / (B^2 K) * mean_k sp[k][b][b']
//
// The softmax runs over the PREDICTION axis b for every target b' (columns), as the reference does.
#include "cpc_common.h"
#include "cpc_kernels.h"

namespace {

constexpr int NCE_CW = 8;       // columns per workgroup of the equal-step loss's column pass (nce_col_mean_kernel)

__device__ __forceinline__ float score_tf(float x, int softplus) {
    if (!softplus) return x;
    return x > 20.f ? x : log1pf(expf(x));       // torch.nn.functional.softplus, beta = 1, threshold = 20
}
__device__ __forceinline__ float score_grad(float x, int softplus) {
    if (!softplus) return 1.f;
    return x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
}

// logsumexp over b of every (k, b') column.  grid (ceil(B/32), K, nsplit); block 256 = 32 columns x 8 row lanes, each lane keeps
// an online (max, sum) over its rows of the split's row range, combined through LDS.  With nsplit == 1 it writes lse[k][b'] and
// one partial sum of lse per block; otherwise the per-split (max, sum) pairs go to pm / ps [split][k][b'] for nce_col_merge_kernel
// (the all-timesteps matrix has 3072 rows but only 96 column blocks: the rows must be split to fill the chip).
// (body with explicit block coordinates: nce_col_kernel calls it with its own, nce_col_mean_kernel with a slice of a 1-D grid)
template <int CW>
__device__ __forceinline__ void nce_col_body(const float* __restrict__ S, float* __restrict__ lse, float* __restrict__ partial, int B,
                                             int K, int ld, int softplus, int rows_per_split, float* __restrict__ pm,
                                             float* __restrict__ ps, int bx, int by, int bz, int nbx, int nbz) {
    // CW columns x RL row lanes per workgroup.  The softplus / exp arithmetic is what this path costs (some hundred instructions per
    // score): with 32 columns per workgroup the equal-step loss (K x B columns of B rows) ran on 96 workgroups, a lane walking 32
    // rows; 8 columns x 32 row lanes spread the same work over four times as many waves (nce_col_mean_kernel: 18.6 -> 13.3 us at
    // B = 256, K = 12; more loads in flight per lane had changed nothing: it is the arithmetic, not the latency).
    constexpr int RL = 256 / CW;
    __shared__ float smx[RL][CW], ssum[RL][CW];
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int k = by, bp = bx * CW + tx;
    const int r0 = bz * rows_per_split, r1 = min(B, r0 + rows_per_split);
    float mx = -INFINITY, sum = 0.f;
    if (bp < B) {
        const float* col = S + (long long)k * B * ld + bp;
        // eight rows per trip: the loads are independent of the running (max, sum), so they are in flight together
        for (int base = r0 + ty; base < r1; base += 8 * RL) {
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = (base + RL * u < r1) ? col[(long long)(base + RL * u) * ld] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (base + RL * u < r1) {
                    const float v = score_tf(vv[u], softplus);
                    if (v > mx) { sum = sum * expf(mx - v) + 1.f; mx = v; }
                    else sum += expf(v - mx);
                }
            }
        }
    }
    smx[ty][tx] = mx;
    ssum[ty][tx] = sum;
    __syncthreads();
    if (ty == 0) {
        float m = mx;
#pragma unroll
        for (int r = 1; r < RL; ++r) m = fmaxf(m, smx[r][tx]);
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < RL; ++r) tot += (smx[r][tx] == -INFINITY) ? 0.f : ssum[r][tx] * expf(smx[r][tx] - m);
        if (nbz > 1) {
            if (bp < B) {
                pm[((long long)bz * K + k) * B + bp] = m;
                ps[((long long)bz * K + k) * B + bp] = tot;
            }
            return;
        }
        const float l = (bp < B) ? m + logf(tot) : 0.f;
        if (bp < B) lse[k * B + bp] = l;
        // sum the CW columns of this block: shuffle reduction over the first CW lanes
        float acc = l;
        for (int o = CW / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, CW);
        if (tx == 0) partial[by * nbx + bx] = acc;
    }
}

__global__ __launch_bounds__(256) void nce_col_kernel(const float* __restrict__ S, float* __restrict__ lse,
                                                      float* __restrict__ partial, int B, int K, int ld, int softplus,
                                                      int rows_per_split, float* __restrict__ pm, float* __restrict__ ps) {
    nce_col_body<32>(S, lse, partial, B, K, ld, softplus, rows_per_split, pm, ps, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.z);
}

// Merges the per-split (max, sum) pairs of a column in split order: lse[c] and one partial sum of lse per block of 256 columns.
__global__ __launch_bounds__(256) void nce_col_merge_kernel(const float* __restrict__ pm, const float* __restrict__ ps, int nsplit,
                                                            int ncols, float* __restrict__ lse, float* __restrict__ partial) {
    __shared__ float red[256];
    const int c = blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (c < ncols) {
        float m = -INFINITY;
        for (int z = 0; z < nsplit; ++z) m = fmaxf(m, pm[(long long)z * ncols + c]);
        float tot = 0.f;
        for (int z = 0; z < nsplit; ++z) {
            const float mz = pm[(long long)z * ncols + c];
            tot += (mz == -INFINITY) ? 0.f : ps[(long long)z * ncols + c] * expf(mz - m);
        }
        l = m + logf(tot);
        lse[c] = l;
    }
    red[threadIdx.x] = l;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// m[b][b'] = mean_k sp[k][b][b'] (the regulariser's mean over the prediction steps, contrastive_estimation_training.py:141)
// for every pair once, plus per-block partials {sum of the valid scores sp[k][b][b], sum m^2, max sp}.  (The gradient kernel
// used to recompute this mean in each of its K blocks per tile: 12 x the softplus work, 42 of the loss's 65 us.)
template <int KT>
__device__ __forceinline__ void nce_mean_body(const float* __restrict__ S, float* __restrict__ mean, float* __restrict__ partial, int B,
                                              int K_rt, int ld, int softplus, int bx) {
    __shared__ float red[3][256];
    const int K = KT > 0 ? KT : K_rt;
    const long long idx = (long long)bx * 256 + threadIdx.x;
    float valid = 0.f, msq = 0.f, mx = -INFINITY;
    if (idx < (long long)B * B) {
        const int b = (int)(idx / B), bp = (int)(idx % B);
        const float* sp0 = S + (long long)b * ld + bp;
        float m = 0.f;
        if (KT > 0) {
            float raw[KT > 0 ? KT : 1];
#pragma unroll
            for (int k = 0; k < KT; ++k) raw[k] = sp0[(long long)k * B * ld];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float sp = score_tf(raw[k], softplus);
                m += sp;
                mx = fmaxf(mx, sp);
                if (b == bp) valid += sp;
            }
        } else {
            for (int k = 0; k < K; ++k) {
                const float sp = score_tf(sp0[(long long)k * B * ld], softplus);
                m += sp;
                mx = fmaxf(mx, sp);
                if (b == bp) valid += sp;
            }
        }
        m /= (float)K;
        msq = m * m;
        mean[(long long)b * ld + bp] = m;
    }
    red[0][threadIdx.x] = valid;
    red[1][threadIdx.x] = msq;
    red[2][threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
            red[2][threadIdx.x] = fmaxf(red[2][threadIdx.x], red[2][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[bx * 3 + 0] = red[0][0];
        partial[bx * 3 + 1] = red[1][0];
        partial[bx * 3 + 2] = red[2][0];
    }
}

// The column log-sum-exps and the pair means read the same scores and do not depend on each other: ONE launch, the first
// ncb * K workgroups take the columns, the rest the pairs (each launch of this latency-bound path costs 5-9 us by itself).
template <int KT>
__global__ __launch_bounds__(256) void nce_col_mean_kernel(const float* __restrict__ S, float* __restrict__ lse, float* __restrict__ colp,
                                                           float* __restrict__ mean, float* __restrict__ pairp, int B, int K, int ld,
                                                           int softplus, int ncb) {
    const int ncol = ncb * K;
    if ((int)blockIdx.x < ncol)
        nce_col_body<NCE_CW>(S, lse, colp, B, K, ld, softplus, B, nullptr, nullptr, blockIdx.x % ncb, blockIdx.x / ncb, 0, ncb, 1);
    else
        nce_mean_body<KT>(S, mean, pairp, B, K, ld, softplus, blockIdx.x - ncol);
}

// out[0] = loss, out[1] = max score, out[2] = -mean valid, out[3] = mean lse, out[4] = reg term (already scaled)
__device__ __forceinline__ void nce_finalize_body(const float* __restrict__ col_partial, int ncol, const float* __restrict__ grad_partial,
                                                  int ngrad, float* __restrict__ out, int B, int K, float reg) {
    // fixed-shape tree reduction: the summation order depends only on (ncol, ngrad), not on timing
    __shared__ float red[4][256];
    float lse_sum = 0.f, valid = 0.f, msq = 0.f, mx = -INFINITY;
    for (int i = threadIdx.x; i < ncol; i += 256) lse_sum += col_partial[i];
    for (int i = threadIdx.x; i < ngrad; i += 256) {
        valid += grad_partial[i * 3 + 0];
        msq += grad_partial[i * 3 + 1];
        mx = fmaxf(mx, grad_partial[i * 3 + 2]);
    }
    red[0][threadIdx.x] = lse_sum; red[1][threadIdx.x] = valid; red[2][threadIdx.x] = msq; red[3][threadIdx.x] = mx;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s2];
            red[1][threadIdx.x] += red[1][threadIdx.x + s2];
            red[2][threadIdx.x] += red[2][threadIdx.x + s2];
            red[3][threadIdx.x] = fmaxf(red[3][threadIdx.x], red[3][threadIdx.x + s2]);
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    lse_sum = red[0][0]; valid = red[1][0]; msq = red[2][0]; mx = red[3][0];
    const float bk = (float)B * (float)K;
    const float t_valid = -valid / bk, t_lse = lse_sum / bk, t_reg = reg * msq / ((float)B * (float)B);
    out[0] = t_valid + t_lse + t_reg;
    out[1] = mx;
    out[2] = t_valid;
    out[3] = t_lse;
    out[4] = t_reg;
    // NaN guard of the reference (contrastive_estimation_training.py:124-133: isnan of the loss BEFORE the regulariser, then
    // `return` before backward() / optimizer.step()): out[5] = this step's indicator, out[6] = sticky (stays raised until the host
    // clears it) — cpc_adam / cpc_adam_dev skip their update while it is raised
    const float lb = t_valid + t_lse;
    const float bad = (lb != lb) ? 1.f : 0.f;
    out[5] = bad;
    if (bad != 0.f) out[6] = 1.f;
}

// One thread per (b, b') pair, 32x32 pairs per block (blockDim = 32x8, 4 rows per thread).
// Writes dS[k][b][b'] and dST[k][b'][b] (storage dtype T) and per-block partials {sum valid, sum m^2, max sp}.
// Writes dS[k][b][b'] and dST[k][b'][b] (storage dtype T); grid (tiles, tiles, K).
// The loss scalars (nce_finalize_body) depend on the partials of the launch before this one only, so workgroup (0, 0, 0) also
// does that reduction: no launch of its own.
template <typename T>
__global__ __launch_bounds__(256) void nce_grad_kernel(const float* __restrict__ S, const float* __restrict__ lse,
                                                       const float* __restrict__ mean, T* __restrict__ dS, T* __restrict__ dST,
                                                       int B, int K, int ld, int softplus, float reg, const float* __restrict__ colp,
                                                       int ncol, const float* __restrict__ pairp, int npair, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // ty 0..7
    const int bp0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const float inv_bk = 1.f / ((float)B * (float)K);
    const float reg_c = 2.f * reg / ((float)B * (float)B * (float)K);
    const int k = blockIdx.z;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + ty + 8 * r, bp = bp0 + tx;
        float g = 0.f;
        if (b < B && bp < B) {
            const float x = S[((long long)k * B + b) * ld + bp];
            const float sp = score_tf(x, softplus);
            float dsp = expf(sp - lse[k * B + bp]) * inv_bk + reg_c * mean[(long long)b * ld + bp];
            if (b == bp) dsp -= inv_bk;
            g = dsp * score_grad(x, softplus);
        }
        if (b < B && bp < ld) dS[((long long)k * B + b) * ld + bp] = from_f32<T>(g);
        tile[ty + 8 * r][tx] = g;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int bp = bp0 + ty + 8 * r, b = b0 + tx;
        if (b < ld && bp < B) dST[((long long)k * B + bp) * ld + b] = from_f32<T>(tile[tx][ty + 8 * r]);
    }
    if (out != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) nce_finalize_body(colp, ncol, pairp, npair, out, B, K, reg);
}

// score_over_all_timesteps=True branch (contrastive_estimation_training.py:108-114, :141): S is the full R x R score
// matrix, R = B*K, row r = (b,k) a prediction, column c = (b',k') a target.  lse[c] (over ALL rows) comes from
// nce_col_kernel(K=1, B=R).  Gradient w.r.t. the linear scores:
//   dsp[r][c] = (exp(sp[r][c] - lse[c]) - [r == c]) / R + 2 reg / (B B K K) * m[b][c],   m[b][c] = mean_k sp[(b,k)][c]
// The kernel is element-wise given lse, so it runs once on S (writing dS) and once on S^T (writing dS^T) with the
// roles of the two strides swapped — no transposes.  Element (r=(b,k), c) of the input sits at c*sc + (b*K+k)*sr.
// FAST_C: consecutive threads walk c (S, sc = 1) or b (S^T, sr = 1; a thread then owns K contiguous elements).
// KT > 0: K as a compile-time constant — the K scores of a pair are loaded once, all loads in flight (with a run-time K the
// pair's 2 K dependent-latency loads made the launch latency-bound: 85 us for 38 MB at B = 256, K = 12).
template <typename T, bool FAST_C, int KT>
__global__ __launch_bounds__(256) void nce_all_grad_kernel(const float* __restrict__ S, const float* __restrict__ lse,
                                                           T* __restrict__ dS, float* __restrict__ partial, int B, int K_rt,
                                                           long long sc, long long sr, int softplus, float reg,
                                                           int want_partials) {
    __shared__ float red[3][256];
    const int K = KT > 0 ? KT : K_rt;
    const int R = B * K;
    const long long total = (long long)B * R;                 // (b, c) pairs
    const float inv_r = 1.f / (float)R;
    const float reg_c = 2.f * reg / ((float)B * (float)B * (float)K * (float)K);
    float valid = 0.f, msq = 0.f, mx = -INFINITY;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int b = FAST_C ? (int)(idx / R) : (int)(idx % B);
        const int c = FAST_C ? (int)(idx % R) : (int)(idx / B);
        const float* base = S + (long long)c * sc + (long long)b * K * sr;
        float m = 0.f;
        const float l = lse[c];
        T* obase = dS + (long long)c * sc + (long long)b * K * sr;
        if (KT > 0) {
            float raw[KT > 0 ? KT : 1], spv[KT > 0 ? KT : 1];
#pragma unroll
            for (int k = 0; k < KT; ++k) raw[k] = base[(long long)k * sr];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                spv[k] = score_tf(raw[k], softplus);
                m += spv[k];
                mx = fmaxf(mx, spv[k]);
                if (b * K + k == c) valid += spv[k];
            }
            m /= (float)K;
            msq += m * m;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                float dsp = expf(spv[k] - l) * inv_r + reg_c * m;
                if (b * K + k == c) dsp -= inv_r;
                obase[(long long)k * sr] = from_f32<T>(dsp * score_grad(raw[k], softplus));
            }
            continue;
        }
        for (int k = 0; k < K; ++k) {
            const float sp = score_tf(base[(long long)k * sr], softplus);
            m += sp;
            mx = fmaxf(mx, sp);
            if (b * K + k == c) valid += sp;
        }
        m /= (float)K;
        msq += m * m;
        for (int k = 0; k < K; ++k) {
            const float x = base[(long long)k * sr];
            const float sp = score_tf(x, softplus);
            float dsp = expf(sp - l) * inv_r + reg_c * m;
            if (b * K + k == c) dsp -= inv_r;
            obase[(long long)k * sr] = from_f32<T>(dsp * score_grad(x, softplus));
        }
    }
    if (!want_partials) return;
    red[0][threadIdx.x] = valid;
    red[1][threadIdx.x] = msq;
    red[2][threadIdx.x] = mx;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s2];
            red[1][threadIdx.x] += red[1][threadIdx.x + s2];
            red[2][threadIdx.x] = fmaxf(red[2][threadIdx.x], red[2][threadIdx.x + s2]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 3 + 0] = red[0][0];
        partial[blockIdx.x * 3 + 1] = red[1][0];
        partial[blockIdx.x * 3 + 2] = red[2][0];
    }
}

// all-timesteps finalize: out[0] = loss, out[1] = max score, out[2] = -mean valid, out[3] = mean lse, out[4] = reg term
__global__ __launch_bounds__(256) void nce_all_finalize_kernel(const float* __restrict__ col_partial, int ncol, const float* __restrict__ grad_partial,
                                        int ngrad, float* __restrict__ out, int B, int K, float reg) {
    // fixed-shape tree reduction: the summation order depends only on (ncol, ngrad), not on timing
    __shared__ float red[4][256];
    float lse_sum = 0.f, valid = 0.f, msq = 0.f, mx = -INFINITY;
    for (int i = threadIdx.x; i < ncol; i += 256) lse_sum += col_partial[i];
    for (int i = threadIdx.x; i < ngrad; i += 256) {
        valid += grad_partial[i * 3 + 0];
        msq += grad_partial[i * 3 + 1];
        mx = fmaxf(mx, grad_partial[i * 3 + 2]);
    }
    red[0][threadIdx.x] = lse_sum; red[1][threadIdx.x] = valid; red[2][threadIdx.x] = msq; red[3][threadIdx.x] = mx;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s2];
            red[1][threadIdx.x] += red[1][threadIdx.x + s2];
            red[2][threadIdx.x] += red[2][threadIdx.x + s2];
            red[3][threadIdx.x] = fmaxf(red[3][threadIdx.x], red[3][threadIdx.x + s2]);
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    lse_sum = red[0][0]; valid = red[1][0]; msq = red[2][0]; mx = red[3][0];
    const float r = (float)B * (float)K;
    const float t_valid = -valid / r, t_lse = lse_sum / r, t_reg = reg * msq / ((float)B * (float)B * (float)K);
    out[0] = t_valid + t_lse + t_reg;
    out[1] = mx;
    out[2] = t_valid;
    out[3] = t_lse;
    out[4] = t_reg;
    // NaN guard of the reference (contrastive_estimation_training.py:124-133: isnan of the loss BEFORE the regulariser, then
    // `return` before backward() / optimizer.step()): out[5] = this step's indicator, out[6] = sticky (stays raised until the host
    // clears it) — cpc_adam / cpc_adam_dev skip their update while it is raised
    const float lb = t_valid + t_lse;
    const float bad = (lb != lb) ? 1.f : 0.f;
    out[5] = bad;
    if (bad != 0.f) out[6] = 1.f;
}

// ---- validation quantities (ContrastiveEstimationTrainer.validate, contrastive_estimation_training.py:227-247) ----
// One wave per score row: sum of the transformed scores, first arg max over the row's columns, and the row's "valid" score
// (the column `diag`).  Default branch: row r = (k, b) of S[k][b][b'], diag = b; all-timesteps branch: row r = (b, k) of the
// R x R matrix, diag = r.
__global__ __launch_bounds__(256) void nce_row_eval_kernel(const float* __restrict__ S, int rows, int cols, int ld, int diag_mod,
                                                           int softplus, float* __restrict__ rsum, int* __restrict__ rarg,
                                                           float* __restrict__ rvalid) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* row = S + (long long)r * ld;
    float sum = 0.f, best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
        const float v = score_tf(row[c], softplus);
        sum += v;
        if (v > best || (v != v && best == best)) { best = v; arg = c; }          // a NaN wins, as in torch.argmax
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        sum += __shfl_xor(sum, o, 64);
        const bool take = (ob > best) || (ob != ob && best == best) || (ob == best && oa < arg);
        if (take) { best = ob; arg = oa; }
    }
    if (lane == 0) {
        rsum[r] = sum;
        rarg[r] = arg;
        rvalid[r] = score_tf(row[diag_mod > 0 ? r % diag_mod : r], softplus);
    }
}

// out[0..K) (+)= per-step losses, out[K..2K) (+)= per-step accuracies, out[2K] (+)= mean score of this batch.
// Default branch: the reference forms noise = logsumexp(s.view(-1, B, K), 0) from the CONTIGUOUS (b, k, b') tensor, i.e. it
// reads the (k, b') array of column log-sum-exps as a (B, K) matrix in flat order, and averages
// valid[j][l] - noise[j][l] over j: lse[j*K + l] with lse stored flat as [k*B + b'].
__global__ __launch_bounds__(256) void nce_eval_finalize_kernel(const float* __restrict__ lse, const float* __restrict__ rsum,
                                                                const int* __restrict__ rarg, const float* __restrict__ rvalid,
                                                                float* __restrict__ out, int B, int K, int all_t, int accumulate) {
    __shared__ float red[256];
    const int R = B * K;
    for (int k = 0; k < K; ++k) {
        float dl = 0.f, hit = 0.f;
        for (int j = threadIdx.x; j < B; j += 256) {
            if (all_t) {
                const int r = j * K + k;
                dl += rvalid[r] - lse[r];
                hit += rarg[r] == r ? 1.f : 0.f;
            } else {
                dl += rvalid[k * B + j] - lse[j * K + k];
                hit += rarg[k * B + j] == j ? 1.f : 0.f;
            }
        }
        red[threadIdx.x] = dl;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
            __syncthreads();
        }
        const float loss_k = -red[0] / (float)B;
        __syncthreads();
        red[threadIdx.x] = hit;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
            __syncthreads();
        }
        const float acc_k = red[0] / (float)(all_t ? R : B);
        __syncthreads();
        if (threadIdx.x == 0) {
            out[k] = (accumulate ? out[k] : 0.f) + loss_k;
            out[K + k] = (accumulate ? out[K + k] : 0.f) + acc_k;
        }
    }
    float tot = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) tot += rsum[r];
    red[threadIdx.x] = tot;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = red[0] / ((float)R * (float)(all_t ? R : B));
        out[2 * K] = (accumulate ? out[2 * K] : 0.f) + mean;
    }
}

// ---- score_over_all_timesteps = True with the column pass fused into the score GEMM (bf16; launch_score_lse in gemm.hip) ----
// The GEMM leaves, per M tile and column, the online log-sum-exp pair (pm, ps) = (max, sum exp(s - max)) of the LINEAR scores s over
// the tile's 256 rows.  For softplus scores exp(softplus(s)) = 1 + exp(s) (torch's threshold 20 changes that by < e^-20), so
//   lse[c] = log( [softplus] * nrows + sum_tiles ps * exp(pm) )
// needs no softplus at all.  colp[block] = {sum of the block's lse, max of its pm}.
__global__ __launch_bounds__(256) void nce_lse_merge_kernel(const float* __restrict__ pm, const float* __restrict__ ps, int nparts, int ncols,
                                                            int softplus, float nrows, float* __restrict__ lse, float* __restrict__ colp) {
    __shared__ float red[2][256];
    const int c = blockIdx.x * 256 + threadIdx.x;
    float l = 0.f, mx = -INFINITY;
    if (c < ncols) {
        for (int i = 0; i < nparts; ++i) mx = fmaxf(mx, pm[(long long)i * ncols + c]);
        const float ref = softplus ? fmaxf(mx, 0.f) : mx;
        float tot = softplus ? nrows * expf(-ref) : 0.f;
        for (int i = 0; i < nparts; ++i) tot += ps[(long long)i * ncols + c] * expf(pm[(long long)i * ncols + c] - ref);
        l = ref + logf(tot);
        lse[c] = l;
    }
    red[0][threadIdx.x] = l;
    red[1][threadIdx.x] = mx;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s2];
            red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + s2]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        colp[blockIdx.x * 2] = red[0][0];
        colp[blockIdx.x * 2 + 1] = red[1][0];
    }
}

// d loss / d linear score from the f32 scores the fused GEMM stored, and its transpose: Sb [items K][ld] (row (b, k) = a prediction, column
// c = a target), lse [ncols]:
//   dS[r][c] = score'(s) * ( (exp(sp - lse[c]) - [c == r + diag_off]) / n_rows + 2 reg / (n_items^2 K^2) * m[b][c] ),   m = mean_k sp[(b,k)][c]
// (n_rows / n_items: of the WHOLE problem — a rank of a data-parallel run holds a strip of it).  Workgroup = 16 items x 64 columns, 128 threads:
// thread = (item, 8 columns), 16-byte loads / stores along the rows; the transposed copy goes through an LDS image [64][16 K] and leaves
// as 16 K contiguous elements per column.  gradp[block] = sum of m^2 over the block's (item, column) pairs (the regulariser's value).
constexpr int NFG_IT = 16, NFG_CW = 64;
// KT > 0: K as a compile-time constant — the K rows of a thread are loaded once, all loads in flight (with a run-time K the rows are read
// twice, one dependent load after the other: 117 us instead of 3x us for the 38 MB of B = 256, K = 12).
template <int KT>
__global__ __launch_bounds__(256) void nce_fused_grad_kernel(const float* __restrict__ Sb, const float* __restrict__ lse, bf16_t* __restrict__ dS,
                                                             bf16_t* __restrict__ dST, float* __restrict__ gradp, int items, int K_rt, int ncols,
                                                             long long ld, long long ldT, int diag_off, int softplus, float inv_r, float reg_c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    __shared__ float red[256];
    const int K = KT > 0 ? KT : K_rt;
    const int rs = NFG_IT * K * 2 + 16;                       // LDS image row stride (bytes): one column's 16 K gradients + pad
    const int tid = threadIdx.x, cg = tid & 15, it = tid >> 4;          // thread = (item, 4 columns)
    const int c0 = blockIdx.x * NFG_CW + cg * 4, item = blockIdx.y * NFG_IT + it;
    const bool on = item < items && c0 < ncols;
    float msq = 0.f;
    if (on) {
        const f32x4 l4 = *(const f32x4*)(lse + c0);
        f32x4 m4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* src = Sb + (long long)item * K * ld + c0;
        bf16_t* dst = dS + (long long)item * K * ld + c0;
        const int r0 = item * K + diag_off;                    // the column of row (item, 0)'s own target
        f32x4 v[KT > 0 ? KT : 1];
        if constexpr (KT > 0) {
#pragma unroll
            for (int k = 0; k < KT; ++k) v[k] = *(const f32x4*)(src + (long long)k * ld);
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) m4[e] += score_tf(v[k][e], softplus);
        } else {
            for (int k = 0; k < K; ++k) {
                const f32x4 x = *(const f32x4*)(src + (long long)k * ld);
#pragma unroll
                for (int e = 0; e < 4; ++e) m4[e] += score_tf(x[e], softplus);
            }
        }
        const float inv_k = 1.f / (float)K;
#pragma unroll
        for (int e = 0; e < 4; ++e) { m4[e] *= inv_k; msq += m4[e] * m4[e]; }
        auto grad4 = [&](const f32x4& x, int k) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = __expf(score_tf(x[e], softplus) - l4[e]) * inv_r + reg_c * m4[e];
                if (c0 + e == r0 + k) d -= inv_r;
                o[e] = (bf16_t)(d * score_grad(x[e], softplus));
            }
            return o;
        };
        auto pair = [&](const f32x4& xa, const f32x4& xb, int k) {
            const bf16x4 oa = grad4(xa, k), ob = grad4(xb, k + 1);
            *(bf16x4*)(dst + (long long)k * ld) = oa;
            *(bf16x4*)(dst + (long long)(k + 1) * ld) = ob;
            if (dST != nullptr) {
                // (element e of row k in the low half, of row k + 1 in the high half; taken from the packed words: hipcc 7.2 gave element 0
                // for every e when the elements were bit-cast one by one — tools/dbg_grad.py, three columns of four wrong)
                const uint2 ua = __builtin_bit_cast(uint2, oa), ub = __builtin_bit_cast(uint2, ob);
                const unsigned wa[2] = {ua.x, ua.y}, wb[2] = {ub.x, ub.y};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned lo = (wa[e >> 1] >> (16 * (e & 1))) & 0xffffu, hi = (wb[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                    *(unsigned*)(tsm + (cg * 4 + e) * rs + (it * K + k) * 2) = lo | (hi << 16);          // image[column][row pair]
                }
            }
        };
        if constexpr (KT > 0) {
#pragma unroll
            for (int k = 0; k < KT; k += 2) pair(v[k], v[k + 1], k);
        } else {
            for (int k = 0; k < K; k += 2)
                pair(*(const f32x4*)(src + (long long)k * ld), *(const f32x4*)(src + (long long)(k + 1) * ld), k);
        }
    }
    if (dST != nullptr) {
        __syncthreads();
        // column cl of the tile: NFG_IT K contiguous elements of row c of dS^T, starting at element blockIdx.y NFG_IT K
        const int cpr = NFG_IT * K / 8;                       // 16-byte chunks per column
        const int rows_here = min(NFG_IT, items - (int)blockIdx.y * NFG_IT) * K;
        for (int q = tid; q < NFG_CW * cpr; q += 256) {
            const int cl = q / cpr, ch = q % cpr;
            const int c = blockIdx.x * NFG_CW + cl;
            if (c < ncols && ch * 8 < rows_here)
                *(uint4*)(dST + (long long)c * ldT + (long long)blockIdx.y * NFG_IT * K + ch * 8) = *(const uint4*)(tsm + cl * rs + ch * 16);
        }
    }
    red[tid] = msq;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
    }
    if (tid == 0 && gradp != nullptr) gradp[blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

// Loss scalars of the fused path.  mode 0: reduce the partials and write out[0..6] (single process); mode 1: reduce only ->
// sums[0..3] = {sum of the valid scores, sum of lse, sum of m^2, max linear score} (a rank's share: all-reduce them, SUM / SUM / SUM / MAX);
// mode 2: out[0..6] from sums.  out as cpc_nce_loss_all writes it.
__global__ __launch_bounds__(256) void nce_fused_finalize_kernel(const float* __restrict__ colp, int ncolp, const float* __restrict__ valid, int nvalid,
                                                                 const float* __restrict__ gradp, int ngrad, float* __restrict__ sums, int mode,
                                                                 float n_rows, float n_items, int K, float reg, int softplus,
                                                                 float* __restrict__ out) {
    __shared__ float red[4][256];
    float sv = 0.f, sl = 0.f, sm = 0.f, mx = -INFINITY;
    if (mode != 2) {
        for (int i = threadIdx.x; i < nvalid; i += 256) sv += score_tf(valid[i], softplus);
        for (int i = threadIdx.x; i < ncolp; i += 256) { sl += colp[2 * i]; mx = fmaxf(mx, colp[2 * i + 1]); }
        for (int i = threadIdx.x; i < ngrad; i += 256) sm += gradp[i];
        red[0][threadIdx.x] = sv; red[1][threadIdx.x] = sl; red[2][threadIdx.x] = sm; red[3][threadIdx.x] = mx;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (threadIdx.x < s2) {
                red[0][threadIdx.x] += red[0][threadIdx.x + s2];
                red[1][threadIdx.x] += red[1][threadIdx.x + s2];
                red[2][threadIdx.x] += red[2][threadIdx.x + s2];
                red[3][threadIdx.x] = fmaxf(red[3][threadIdx.x], red[3][threadIdx.x + s2]);
            }
            __syncthreads();
        }
        sv = red[0][0]; sl = red[1][0]; sm = red[2][0]; mx = red[3][0];
    } else {
        sv = sums[0]; sl = sums[1]; sm = sums[2]; mx = sums[3];
    }
    if (threadIdx.x != 0) return;
    if (mode == 1) {
        sums[0] = sv; sums[1] = sl; sums[2] = sm; sums[3] = mx;
        return;
    }
    const float t_valid = -sv / n_rows, t_lse = sl / n_rows, t_reg = reg * sm / (n_items * n_rows);
    out[0] = t_valid + t_lse + t_reg;
    out[1] = score_tf(mx, softplus);
    out[2] = t_valid;
    out[3] = t_lse;
    out[4] = t_reg;
    const float lb = t_valid + t_lse;             // the reference's NaN guard tests the loss before the regulariser (:124)
    const float bad = (lb != lb) ? 1.f : 0.f;
    out[5] = bad;
    if (bad != 0.f) out[6] = 1.f;
}

}  // namespace

int launch_nce_lse_merge(const float* pm, const float* ps, int nparts, int ncols, int softplus, float nrows, float* lse, float* colp,
                         hipStream_t stream) {
    if (!pm || !ps || !lse || !colp || nparts <= 0 || ncols <= 0) return CPC_EINVAL;
    hipLaunchKernelGGL(nce_lse_merge_kernel, dim3((ncols + 255) / 256), dim3(256), 0, stream, pm, ps, nparts, ncols, softplus, nrows, lse, colp);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

long long nce_fused_grad_blocks(int items, int ncols) {
    return (long long)((ncols + NFG_CW - 1) / NFG_CW) * ((items + NFG_IT - 1) / NFG_IT);
}

int launch_nce_fused_grad(const void* Sb, const float* lse, void* dS, void* dST, float* gradp, int items, int K, int ncols, long long ld,
                          long long ldT, int diag_off, int softplus, float reg, float n_rows, float n_items, hipStream_t stream) {
    if (!Sb || !lse || !dS || items <= 0 || K < 2 || K % 2 || K > 24 || ncols <= 0 || ncols % 4 || ld % 8 || ld < ncols) return CPC_EINVAL;
    if (dST && (ldT % 8 || ldT < (long long)items * K)) return CPC_EINVAL;
    if ((uintptr_t)Sb % 16 || (uintptr_t)dS % 16 || (uintptr_t)dST % 16 || (uintptr_t)lse % 16) return CPC_EINVAL;
    const dim3 grid((ncols + NFG_CW - 1) / NFG_CW, (items + NFG_IT - 1) / NFG_IT);
    const size_t lds_bytes = dST ? (size_t)NFG_CW * (NFG_IT * K * 2 + 16) : 0;
    const float inv_r = 1.f / n_rows, reg_c = 2.f * reg / (n_items * n_items * (float)K * (float)K);
#define NFG_LAUNCH(KT) \
    hipLaunchKernelGGL(nce_fused_grad_kernel<KT>, grid, dim3(256), lds_bytes, stream, (const float*)Sb, lse, (bf16_t*)dS, (bf16_t*)dST, gradp, items, K, \
                       ncols, ld, ldT, diag_off, softplus, inv_r, reg_c)
    if (K == 12) NFG_LAUNCH(12); else if (K == 16) NFG_LAUNCH(16); else if (K == 8) NFG_LAUNCH(8); else NFG_LAUNCH(0);
#undef NFG_LAUNCH
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

int launch_nce_fused_finalize(const float* colp, int ncolp, const float* valid, int nvalid, const float* gradp, int ngrad, float* sums, int mode,
                              float n_rows, float n_items, int K, float reg, int softplus, float* out, hipStream_t stream) {
    if (mode < 0 || mode > 2 || (mode != 2 && (!colp || !valid || !gradp)) || (mode != 0 && !sums) || (mode != 1 && !out)) return CPC_EINVAL;
    hipLaunchKernelGGL(nce_fused_finalize_kernel, dim3(1), dim3(256), 0, stream, colp, ncolp, valid, nvalid, gradp, ngrad, sums, mode, n_rows, n_items,
                       K, reg, softplus, out);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// workspace: lse [K*B] + col partials [ceil(K*B/256)] + grad partials [3 * ceil(B/32)^2]   (f32)
// workspace: lse [K][B] + column partials [K * ceil(B/32)] + pair partials [3 * ceil(B*B/256)] + pair means [B][ld <= B + 7]
long long nce_workspace_floats(int B, int K) {
    const long long nmb = ((long long)B * B + 255) / 256;
    return (long long)K * B + (long long)K * ((B + NCE_CW - 1) / NCE_CW) + 3 * nmb + (long long)B * (B + 8);
}

int launch_nce(const float* S, void* dS, void* dST, float* out, float* workspace, int B, int K, int ld, int softplus, float reg,
               int dtype, hipStream_t stream) {
    if (B <= 0 || K <= 0 || ld < B || ld > B + 7) return CPC_EINVAL;
    float* lse = workspace;
    const int ncb = (B + NCE_CW - 1) / NCE_CW;
    const int ncol = K * ncb;
    float* colp = lse + (long long)K * B;
    float* gradp = colp + ncol;
    const int nmb = (int)(((long long)B * B + 255) / 256);
    float* mean = gradp + 3LL * nmb;
    const int nb = (ld + 31) / 32;
    // two launches: column log-sum-exps + pair means, then gradients + loss scalars (each launch of this path is latency-bound)
    const dim3 g1(ncol + nmb);
    if (K == 12) hipLaunchKernelGGL(nce_col_mean_kernel<12>, g1, dim3(256), 0, stream, S, lse, colp, mean, gradp, B, K, ld, softplus, ncb);
    else if (K == 16) hipLaunchKernelGGL(nce_col_mean_kernel<16>, g1, dim3(256), 0, stream, S, lse, colp, mean, gradp, B, K, ld, softplus, ncb);
    else hipLaunchKernelGGL(nce_col_mean_kernel<0>, g1, dim3(256), 0, stream, S, lse, colp, mean, gradp, B, K, ld, softplus, ncb);
    if (dtype == CPC_DTYPE_BF16)
        hipLaunchKernelGGL((nce_grad_kernel<bf16_t>), dim3(nb, nb, K), dim3(256), 0, stream, S, lse, mean, (bf16_t*)dS, (bf16_t*)dST,
                           B, K, ld, softplus, reg, colp, ncol, gradp, nmb, out);
    else if (dtype == CPC_DTYPE_F32)
        hipLaunchKernelGGL((nce_grad_kernel<float>), dim3(nb, nb, K), dim3(256), 0, stream, S, lse, mean, (float*)dS, (float*)dST, B, K,
                           ld, softplus, reg, colp, ncol, gradp, nmb, out);
    else
        return CPC_EINVAL;
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// ---- score_over_all_timesteps = True ----
static const int NCE_ALL_BLOCKS = 3072;
static const int NCE_ALL_SPLITS = 16;
// workspace: lse [R] + col partials [ceil(R/32)] + grad partials [3 * NCE_ALL_BLOCKS] + per-split (max, sum) [2][NCE_ALL_SPLITS][R]
long long nce_all_workspace_floats(int B, int K) {
    const long long R = (long long)B * K;
    return R + (R + 31) / 32 + 3LL * NCE_ALL_BLOCKS + 2LL * NCE_ALL_SPLITS * R;
}

int launch_nce_all(const float* S, const float* ST, void* dS, void* dST, float* out, float* workspace, int B, int K, int ld,
                   int softplus, float reg, int dtype, hipStream_t stream) {
    const int R = B * K;
    if (B <= 0 || K <= 0 || ld < R) return CPC_EINVAL;
    float* lse = workspace;
    const int ncb = (R + 31) / 32;
    float* colp = lse + R;
    float* gradp = colp + ncb;
    float* pm = gradp + 3LL * NCE_ALL_BLOCKS;
    float* ps = pm + (long long)NCE_ALL_SPLITS * R;
    const int nsplit = R >= 8 * NCE_ALL_SPLITS ? NCE_ALL_SPLITS : 1;
    int n_colp = ncb;
    if (nsplit > 1) {
        const int rps = ((R + nsplit - 1) / nsplit + 7) / 8 * 8;
        hipLaunchKernelGGL(nce_col_kernel, dim3(ncb, 1, nsplit), dim3(256), 0, stream, S, lse, colp, R, 1, ld, softplus, rps, pm, ps);
        n_colp = (R + 255) / 256;
        hipLaunchKernelGGL(nce_col_merge_kernel, dim3(n_colp), dim3(256), 0, stream, pm, ps, nsplit, R, lse, colp);
    } else {
        hipLaunchKernelGGL(nce_col_kernel, dim3(ncb, 1, 1), dim3(256), 0, stream, S, lse, colp, R, 1, ld, softplus, R, (float*)nullptr,
                           (float*)nullptr);
    }
    const long long total = (long long)B * R;
    const int blocks = (int)min((long long)NCE_ALL_BLOCKS, (total + 255) / 256);
#define NCE_ALL_GRAD(T, KT) \
    do { \
        hipLaunchKernelGGL((nce_all_grad_kernel<T, true, KT>), dim3(blocks), dim3(256), 0, stream, S, lse, (T*)dS, gradp, B, K, 1LL, \
                           (long long)ld, softplus, reg, 1); \
        hipLaunchKernelGGL((nce_all_grad_kernel<T, false, KT>), dim3(blocks), dim3(256), 0, stream, ST, lse, (T*)dST, gradp, B, K, \
                           (long long)ld, 1LL, softplus, reg, 0); \
    } while (0)
    if (dtype == CPC_DTYPE_BF16) {
        if (K == 12) NCE_ALL_GRAD(bf16_t, 12); else if (K == 16) NCE_ALL_GRAD(bf16_t, 16); else NCE_ALL_GRAD(bf16_t, 0);
    } else if (dtype == CPC_DTYPE_F32) {
        if (K == 12) NCE_ALL_GRAD(float, 12); else if (K == 16) NCE_ALL_GRAD(float, 16); else NCE_ALL_GRAD(float, 0);
    } else {
        return CPC_EINVAL;
    }
#undef NCE_ALL_GRAD
    hipLaunchKernelGGL(nce_all_finalize_kernel, dim3(1), dim3(256), 0, stream, colp, n_colp, gradp, blocks, out, B, K, reg);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}


// ---- validation ----
// workspace: lse [R] + column partials + per-split (max, sum) [2][NCE_ALL_SPLITS][R] + row sums / arg max / valid [3][R]   (R = B*K)
long long nce_eval_workspace_floats(int B, int K) {
    const long long R = (long long)B * K;
    return R + (R + 31) / 32 + K * (long long)((B + 31) / 32) + 2LL * NCE_ALL_SPLITS * R + 3 * R + 64;
}

int launch_nce_eval(const float* S, float* out, float* workspace, int B, int K, int ld, int softplus, int all_timesteps,
                    int accumulate, hipStream_t stream) {
    const int R = B * K;
    if (B <= 0 || K <= 0 || (all_timesteps ? ld < R : (ld < B || ld > B + 7))) return CPC_EINVAL;
    float* lse = workspace;
    float* colp = lse + R;
    float* pm = colp + (R + 31) / 32 + (long long)K * ((B + 31) / 32);
    float* ps = pm + (long long)NCE_ALL_SPLITS * R;
    float* rsum = ps + (long long)NCE_ALL_SPLITS * R;
    int* rarg = (int*)(rsum + R);
    float* rvalid = rsum + 2LL * R;
    if (all_timesteps) {
        const int ncb = (R + 31) / 32;
        const int nsplit = R >= 8 * NCE_ALL_SPLITS ? NCE_ALL_SPLITS : 1;
        if (nsplit > 1) {
            const int rps = ((R + nsplit - 1) / nsplit + 7) / 8 * 8;
            hipLaunchKernelGGL(nce_col_kernel, dim3(ncb, 1, nsplit), dim3(256), 0, stream, S, lse, colp, R, 1, ld, softplus, rps, pm, ps);
            hipLaunchKernelGGL(nce_col_merge_kernel, dim3((R + 255) / 256), dim3(256), 0, stream, pm, ps, nsplit, R, lse, colp);
        } else {
            hipLaunchKernelGGL(nce_col_kernel, dim3(ncb, 1, 1), dim3(256), 0, stream, S, lse, colp, R, 1, ld, softplus, R, (float*)nullptr,
                               (float*)nullptr);
        }
        hipLaunchKernelGGL(nce_row_eval_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, S, R, R, ld, 0, softplus, rsum, rarg, rvalid);
    } else {
        hipLaunchKernelGGL(nce_col_kernel, dim3((B + 31) / 32, K, 1), dim3(256), 0, stream, S, lse, colp, B, K, ld, softplus, B,
                           (float*)nullptr, (float*)nullptr);
        hipLaunchKernelGGL(nce_row_eval_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, S, R, B, ld, B, softplus, rsum, rarg, rvalid);
    }
    hipLaunchKernelGGL(nce_eval_finalize_kernel, dim3(1), dim3(256), 0, stream, lse, rsum, rarg, rvalid, out, B, K, all_timesteps ? 1 : 0,
                       accumulate ? 1 : 0);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}

// Wasserstein gradient penalty with softplus scores (contrastive_estimation_training.py:12-16 under :144-158): the summed scores
// are sum softplus(s), so the seeds of the penalty's passes carry the coefficients sigmoid(s) (first order) and
// sigmoid'(s) * (tangent of s) (second order) instead of the constant 1 and 0 of linear scores.
//   mode 0:  w = d softplus(s) / d s          mode 1:  w = d^2 softplus(s) / d s^2 * (St1 + St2)     (St2 may be NULL)
// S, St*: nmat matrices [rows][ld] f32;  W [nmat][rows][ld], WT [nmat][cols][ldT] (the transpose), both f32.
__global__ __launch_bounds__(256) void gp_score_coeff_kernel(const float* __restrict__ S, const float* __restrict__ St1,
                                                             const float* __restrict__ St2, float* __restrict__ W,
                                                             float* __restrict__ WT, int rows, int cols, int ld, int ldT, int mode) {
    const int m = blockIdx.y;
    const long long total = (long long)rows * cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx / cols), c = (int)(idx % cols);
        const long long o = ((long long)m * rows + r) * ld + c;
        const float x = S[o];
        const float sg = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
        float w = sg;
        if (mode == 1) {
            float t = St1[o];
            if (St2) t += St2[o];
            w = x > 20.f ? 0.f : sg * (1.f - sg) * t;
        }
        W[o] = w;
        WT[((long long)m * cols + c) * ldT + r] = w;
    }
}

int launch_gp_score_coeff(const float* S, const float* St1, const float* St2, float* W, float* WT, int nmat, int rows, int cols, int ld,
                          int ldT, int mode, hipStream_t stream) {
    if (nmat <= 0 || rows <= 0 || cols <= 0 || ld < cols || ldT < rows || (mode != 0 && mode != 1) || (mode == 1 && !St1)) return CPC_EINVAL;
    const int blocks = (int)std::min<long long>(1024, ((long long)rows * cols + 255) / 256);
    hipLaunchKernelGGL(gp_score_coeff_kernel, dim3(blocks, nmat), dim3(256), 0, stream, S, St1, St2, W, WT, rows, cols, ld, ldT, mode);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
