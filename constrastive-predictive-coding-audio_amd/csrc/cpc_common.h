// Shared device/host helpers for the CPC-audio HIP path (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define CPC_DTYPE_F32 0
#define CPC_DTYPE_BF16 1

#define CPC_OK 0
#define CPC_EINVAL (-22)
#define CPC_EIO (-5)

// Elements per 16-byte chunk for a storage type.
template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int CH = 4; };
template <> struct Elem<bf16_t> { static constexpr int CH = 8; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32, RNE, NaN-safe

// relu as torch computes it: a NaN stays a NaN (fmaxf(NaN, 0) = 0 would swallow it, and with it the trainer's NaN guard,
// contrastive_estimation_training.py:124-133)
// ... and -0 becomes +0, so that a relu output is > 0 exactly when its bit pattern is not zero (relu_positive_bit: what the
// sign-bit masks of include/cpc_hip.h are made of, two integer operations per element)
__device__ __forceinline__ float relu_f(float v) { return v <= 0.f ? 0.f : v; }
__device__ __forceinline__ unsigned relu_positive_bit(float relu_out) {
    const unsigned u = __builtin_bit_cast(unsigned, relu_out);
    return u < 1u ? u : 1u;
}

// Store 4 consecutive values held as f32 into a T* (8 B for bf16, 16 B for f32). dst must be aligned to that size.
__device__ __forceinline__ void store4(float* dst, f32x4 v) { *(f32x4*)dst = v; }
__device__ __forceinline__ void store4(bf16_t* dst, f32x4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    *(bf16x4*)dst = o;
}
__device__ __forceinline__ f32x4 load4(const float* src) { return *(const f32x4*)src; }
__device__ __forceinline__ f32x4 load4(const bf16_t* src) {
    bf16x4 i = *(const bf16x4*)src;
    f32x4 o;
    o[0] = (float)i[0]; o[1] = (float)i[1]; o[2] = (float)i[2]; o[3] = (float)i[3];
    return o;
}

// Row index -> element offset for "items of rpi rows": (m / rpi) * item + (m % rpi) * ld; rpi == 0 means plain m * ld.
__device__ __forceinline__ long long row_off(int m, int rpi, long long item, long long ld) {
    if (rpi == 0) return (long long)m * ld;
    int q = m / rpi;
    return (long long)q * item + (long long)(m - q * rpi) * ld;
}

// ... with a second level: items of rpi rows grouped rpi2 to a band (rpi2 == 0: one level)
__device__ __forceinline__ long long row_off2(int m, int rpi, long long item, long long ld, int rpi2, long long item2) {
    if (rpi2 == 0) return row_off(m, rpi, item, ld);
    const int q = m / rpi, q2 = q / rpi2;
    return (long long)q2 * item2 + (long long)(q - q2 * rpi2) * item + (long long)(m - q * rpi) * ld;
}

// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): a GRU step is
// latency-bound on 16 workgroups, and libm's expf / tanhf / IEEE division cost more than the step's 96 MFMAs.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}

// ---- MFMA on one 16-byte operand chunk per lane (D = A*B + C, 16x16 output tile) ----
template <typename T>
__device__ __forceinline__ void mfma_chunk(f32x4& acc, const uint4& a_op, const uint4& b_op);

// bf16: one 16x16x32 MFMA consumes the whole 16-byte chunk (8 k-values) of each operand.
template <>
__device__ __forceinline__ void mfma_chunk<bf16_t>(f32x4& acc, const uint4& a_op, const uint4& b_op) {
    bf16x8 a = __builtin_bit_cast(bf16x8, a_op);
    bf16x8 b = __builtin_bit_cast(bf16x8, b_op);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
// f32: a 16-byte chunk holds 4 k-values; four 16x16x4 MFMAs, MFMA e taking element e of every lane's chunk.
// (The k order inside the tile is permuted identically for both operands, which a dot product does not see.)
template <>
__device__ __forceinline__ void mfma_chunk<float>(f32x4& acc, const uint4& a_op, const uint4& b_op) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a_op.x), __builtin_bit_cast(float, b_op.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a_op.y), __builtin_bit_cast(float, b_op.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a_op.z), __builtin_bit_cast(float, b_op.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a_op.w), __builtin_bit_cast(float, b_op.w), acc, 0, 0, 0);
}


#define CPC_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return CPC_EIO;               \
    } while (0)
