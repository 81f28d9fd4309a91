// Scalogram front end and 2-D residual encoder kernels (constant_q_transform.py / scalogram_model.py of the reference).
// The CQT filter bank itself is a set of overlapped-row f32 GEMMs (cpc_gemm_nt); this file holds what is not a GEMM:
//   * the pointwise chain  |.|^2 -> log -> phase difference -> scale -> power            (PreprocessingModule.forward)
//   * im2col / col2im for the windowed 2-D convolutions, on channels-last [B][W][H][C] buffers
//   * BatchNorm (batch statistics) forward / backward, MaxPool2d(ceil), cropped residual add.
#include "cpc_common.h"
#include "cpc_kernels.h"

namespace {

constexpr float PI_F = 3.14159265358979323846f;

// cq  f32 [B][Tn][ldq]: (re, im) interleaved per bin.   out f32 [B][W][bins][Cc], W = Tn - 1 and Cc = 2 with phase,
// W = Tn and Cc = 1 without.  scalogram_model.py:77-97, constant_q_transform.py:36-52, :69-72, :281-285.
__global__ __launch_bounds__(256) void scalogram_pointwise_kernel(const float* __restrict__ cq, const float* __restrict__ fixed_pd,
                                                                  const float* __restrict__ pd_scale, float* __restrict__ out,
                                                                  int B, int Tn, int bins, long long ldq, int phase, float offset,
                                                                  float log_offset, float norm, float power) {
    const int W = phase ? Tn - 1 : Tn;
    const long long total = (long long)B * W * bins;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int bin = (int)(idx % bins);
        const int w = (int)((idx / bins) % W);
        const int b = (int)(idx / ((long long)bins * W));
        const int t1 = phase ? w + 1 : w;
        const float2 z1 = *(const float2*)(cq + ((long long)b * Tn + t1) * ldq + 2 * bin);
        const float mag = sqrtf(z1.x * z1.x + z1.y * z1.y);
        float amp = (logf(mag * mag + offset) + log_offset) * norm;
        if (power != 1.f) amp = powf(amp, power);
        if (!phase) {
            out[idx] = amp;
            continue;
        }
        const float2 z0 = *(const float2*)(cq + ((long long)b * Tn + w) * ldq + 2 * bin);
        float pd = atan2f(z1.y, z1.x) - atan2f(z0.y, z0.x) + fixed_pd[bin];
        if (pd > PI_F) pd -= 2.f * PI_F;
        if (pd < -PI_F) pd += 2.f * PI_F;
        pd = pd * pd_scale[bin] * norm;
        if (power != 1.f) pd = powf(pd, power);
        *(float2*)(out + idx * 2) = make_float2(amp, pd);
    }
}

}  // namespace

int launch_scalogram_pointwise(const float* cq, const float* fixed_pd, const float* pd_scale, float* out, int B, int Tn, int bins,
                               long long ldq, int phase, float offset, float log_offset, float norm, float power,
                               hipStream_t st) {
    if (B <= 0 || bins <= 0 || Tn <= (phase ? 1 : 0) || ldq < 2 * bins || (ldq & 1)) return CPC_EINVAL;
    if (phase && (!fixed_pd || !pd_scale)) return CPC_EINVAL;
    const long long total = (long long)B * (phase ? Tn - 1 : Tn) * bins;
    const int blocks = (int)min((long long)4096, (total + 255) / 256);
    hipLaunchKernelGGL(scalogram_pointwise_kernel, dim3(blocks), dim3(256), 0, st, cq, fixed_pd, pd_scale, out, B, Tn, bins, ldq,
                       phase, offset, log_offset, norm, power);
    CPC_CHECK_LAUNCH();
    return CPC_OK;
}
