// Stand-alone probe (never part of libcpc_hip.so): which wave -> address mapping lets a pure store stream of the layer-1 forward's
// size (B x L_alloc rows of 1 KiB = 956 MB at B = 256) reach the rate of a plain fill?  DESIGN.md, layer-1 forward.
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/store_probe && /tmp/store_probe [rows_per_item] [items]
// Every mode writes the same bytes (16 B per lane, 1 KiB per wave-instruction); `work` = dependent FMAs per row between stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st16(void* p, v4u v) {
    if (NT) __builtin_nontemporal_store(v, (v4u*)p);
    else *(v4u*)p = v;
}

__device__ __forceinline__ v4u payload(float seed, int work) {
    float a = seed;
    for (int i = 0; i < work; ++i) a = __builtin_fmaf(a, 1.0001f, 0.5f);
    const unsigned u = __builtin_bit_cast(unsigned, a);
    return (v4u){u, u ^ 1u, u ^ 2u, u ^ 3u};
}

// mode A: one 16-byte store per thread, blocks in memory order (what torch's vectorized fill does)
template <bool NT>
__global__ __launch_bounds__(256) void fill_like(unsigned char* dst, long long nbytes, int per_thread, int work) {
    const long long base = ((long long)blockIdx.x * per_thread) * 4096 + threadIdx.x * 16;
    for (int i = 0; i < per_thread; ++i) {
        const long long off = base + (long long)i * 4096;
        if (off < nbytes) st16<NT>(dst + off, payload((float)threadIdx.x, work));
    }
}

// mode B: the layer-1 forward's mapping: grid (ceil(rows / RPB), items); wave w of a block writes rows [w RPB/4, (w+1) RPB/4) of the
// block's RPB rows, one 1 KiB row per wave-instruction.  INTERLEAVE: wave w takes rows w, w + 4, ...
template <bool NT, int RPB, bool INTERLEAVE>
__global__ __launch_bounds__(256) void conv1_like(unsigned char* dst, int rows, int work) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * RPB;
    unsigned char* base = dst + ((long long)blockIdx.y * rows + r0) * 1024 + lane * 16;
    const int n = min(RPB, rows - r0);
    for (int k = 0; k < RPB / 4; ++k) {
        const int r = INTERLEAVE ? k * 4 + wave : wave * (RPB / 4) + k;
        if (r < n) st16<NT>(base + (long long)r * 1024, payload((float)(r + lane), work));
    }
}

// mode B2: the same mapping with a cap on the stores a wave keeps in flight (s_waitcnt vmcnt(CAP) after every store; CAP < 0: none) and an
// optional pause (s_sleep) between stores; dynamic LDS limits the blocks per CU
template <int RPB, int CAP, int SLEEP, bool INTERLEAVE>
__global__ __launch_bounds__(256) void conv1_capped(unsigned char* dst, int rows) {
    extern __shared__ unsigned char dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * RPB;
    unsigned char* base = dst + ((long long)blockIdx.y * rows + r0) * 1024 + lane * 16;
    const int n = min(RPB, rows - r0);
    if (threadIdx.x == 0 && rows < 0) dyn[0] = 1;
    for (int k = 0; k < RPB / 4; ++k) {
        const int r = INTERLEAVE ? k * 4 + wave : wave * (RPB / 4) + k;
        if (r < n) st16<true>(base + (long long)r * 1024, payload((float)(r + lane), 0));
        if (CAP == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (CAP == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (CAP == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (CAP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (CAP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// mode B3: the layer-1 mapping (wave = RPB / 4 contiguous rows) with the waves DE-PHASED: wave w of block (bx, by) starts at row
// (phase mod RPB / 4) of its range and wraps around, so that the ~4 000 streams of the chip sit at different offsets modulo 64 KiB
template <int RPB, int MODE>
__global__ __launch_bounds__(256) void conv1_rot(unsigned char* dst, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * RPB;
    unsigned char* base = dst + ((long long)blockIdx.y * rows + r0) * 1024 + lane * 16;
    const int n = min(RPB, rows - r0);
    constexpr int RPW = RPB / 4;
    const unsigned id = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
    const int rot = MODE == 0 ? 0 : MODE == 1 ? (int)((id * 17u) % RPW) : MODE == 2 ? (int)((id * 2654435761u >> 16) % RPW) : (int)((wave * (RPW / 4) + blockIdx.x * 5 + blockIdx.y * 3) % RPW);
    for (int k = 0; k < RPW; ++k) {
        const int r = wave * RPW + (k + rot) % RPW;
        if (r < n) st16<true>(base + (long long)r * 1024, payload((float)(r + lane), 0));
    }
}

// mode C: persistent, compact write front: `grid` blocks; block g writes chunks g, g + grid, ... of CH rows (wave w: rows w CH/4 ...)
template <bool NT, int CH>
__global__ __launch_bounds__(256) void front_like(unsigned char* dst, long long total_rows, int work) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nchunk = (total_rows + CH - 1) / CH;
    for (long long c = blockIdx.x; c < nchunk; c += gridDim.x) {
#pragma unroll
        for (int k = 0; k < CH / 4; ++k) {
            const long long r = c * CH + wave * (CH / 4) + k;
            if (r < total_rows) st16<NT>(dst + r * 1024 + lane * 16, payload((float)(k + lane), work));
        }
    }
}



#include <functional>
static float time_ms(hipStream_t s, int reps, const std::function<void()>& fn) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) fn();
    hipStreamSynchronize(s);
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0, s);
        fn();
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 3648, items = argc > 2 ? atoi(argv[2]) : 256;
    const long long total_rows = (long long)rows * items, nbytes = total_rows * 1024;
    unsigned char* buf = nullptr;
    if (hipMalloc(&buf, nbytes + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipStream_t s;
    hipStreamCreate(&s);
    printf("store probe: %d items x %d rows x 1 KiB = %.1f MB\n", items, rows, nbytes / 1e6);
    auto report = [&](const char* name, float ms) { printf("%-64s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, nbytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    report("hipMemsetAsync", time_ms(s, 9, [&] { hipMemsetAsync(buf, 0, nbytes, s); }));
    {
        dim3 g256((rows + 255) / 256, items), g128((rows + 127) / 128, items);
#define CAPPED(RPB, CAP, SLEEP, IL, LDS, G, NAME) report(NAME, time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_capped<RPB, CAP, SLEEP, IL>), G, dim3(256), LDS, s, buf, rows); }))
        printf("-- layer-1 mapping (wave = RPB / 4 contiguous rows), rows per block\n");
        for (int pass = 0; pass < 2; ++pass) {
            report("  16 rows per block", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_rot<16, 0>), dim3((rows + 15) / 16, items), dim3(256), 0, s, buf, rows); }));
            report("  32 rows per block", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_rot<32, 0>), dim3((rows + 31) / 32, items), dim3(256), 0, s, buf, rows); }));
            report("  64 rows per block", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_rot<64, 0>), dim3((rows + 63) / 64, items), dim3(256), 0, s, buf, rows); }));
            report(" 128 rows per block", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_rot<128, 0>), dim3((rows + 127) / 128, items), dim3(256), 0, s, buf, rows); }));
            report(" 256 rows per block", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_rot<256, 0>), dim3((rows + 255) / 256, items), dim3(256), 0, s, buf, rows); }));
            report(" 512 rows per block", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_rot<512, 0>), dim3((rows + 511) / 512, items), dim3(256), 0, s, buf, rows); }));
        }
        for (int pass = 0; pass < 1; ++pass) {
            printf("-- pass %d: identical mappings from three kernels, then rotations\n", pass);
            report("conv1_like<nt,256>", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_like<true, 256, false>), g256, dim3(256), 0, s, buf, rows, 0); }));
            CAPPED(256, -1, 0, false, 0, g256, "conv1_capped<256, no cap>");
            report("conv1_rot<256, none>", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_rot<256, 0>), g256, dim3(256), 0, s, buf, rows); }));
            report("conv1_rot<256, 17 id>", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_rot<256, 1>), g256, dim3(256), 0, s, buf, rows); }));
            report("conv1_rot<128, none>", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_rot<128, 0>), g128, dim3(256), 0, s, buf, rows); }));
            report("conv1_like<nt,16 rows interleaved>", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_like<true, 16, true>), dim3((rows + 15) / 16, items), dim3(256), 0, s, buf, rows, 0); }));
            report("fill-like 4 KiB nt", time_ms(s, 9, [&] { hipLaunchKernelGGL(fill_like<true>, dim3((nbytes + 4095) / 4096), dim3(256), 0, s, buf, nbytes, 1, 0); }));
            CAPPED(256, -1, 0, false, 0, g256, "conv1_capped<256, no cap> again");
        }
    }
    for (int work : std::vector<int>{}) {
        printf("-- %d dependent FMAs per row between stores\n", work);
        const long long nblk = (nbytes + 4095) / 4096;
        report("fill-like, 4 KiB per block, plain", time_ms(s, 9, [&] { hipLaunchKernelGGL(fill_like<false>, dim3(nblk), dim3(256), 0, s, buf, nbytes, 1, work); }));
        report("fill-like, 4 KiB per block, nt", time_ms(s, 9, [&] { hipLaunchKernelGGL(fill_like<true>, dim3(nblk), dim3(256), 0, s, buf, nbytes, 1, work); }));
        report("fill-like, 64 KiB per block (16 passes), nt", time_ms(s, 9, [&] { hipLaunchKernelGGL(fill_like<true>, dim3((nblk + 15) / 16), dim3(256), 0, s, buf, nbytes, 16, work); }));
        report("conv1 mapping, 256 rows per block, wave = 64 contiguous rows, nt", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_like<true, 256, false>), dim3((rows + 255) / 256, items), dim3(256), 0, s, buf, rows, work); }));
        report("conv1 mapping, 256 rows per block, wave = 64 contiguous rows, plain", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_like<false, 256, false>), dim3((rows + 255) / 256, items), dim3(256), 0, s, buf, rows, work); }));
        report("conv1 mapping, 256 rows per block, interleaved rows, nt", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_like<true, 256, true>), dim3((rows + 255) / 256, items), dim3(256), 0, s, buf, rows, work); }));
        report("conv1 mapping, 64 rows per block, interleaved rows, nt", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_like<true, 64, true>), dim3((rows + 63) / 64, items), dim3(256), 0, s, buf, rows, work); }));
        report("conv1 mapping, 16 rows per block, interleaved rows, nt", time_ms(s, 9, [&] { hipLaunchKernelGGL((conv1_like<true, 16, true>), dim3((rows + 15) / 16, items), dim3(256), 0, s, buf, rows, work); }));
        for (int grid : {512, 1024, 2048, 4096}) {
            char nm[96];
            snprintf(nm, sizeof nm, "persistent compact front, %d blocks, 4-row chunks, nt", grid);
            report(nm, time_ms(s, 9, [&] { hipLaunchKernelGGL((front_like<true, 4>), dim3(grid), dim3(256), 0, s, buf, total_rows, work); }));
            snprintf(nm, sizeof nm, "persistent compact front, %d blocks, 16-row chunks, nt", grid);
            report(nm, time_ms(s, 9, [&] { hipLaunchKernelGGL((front_like<true, 16>), dim3(grid), dim3(256), 0, s, buf, total_rows, work); }));
        }
        report("persistent compact front, 2048 blocks, 16-row chunks, plain", time_ms(s, 9, [&] { hipLaunchKernelGGL((front_like<false, 16>), dim3(2048), dim3(256), 0, s, buf, total_rows, work); }));
        report("persistent compact front, 2048 blocks, 64-row chunks, nt", time_ms(s, 9, [&] { hipLaunchKernelGGL((front_like<true, 64>), dim3(2048), dim3(256), 0, s, buf, total_rows, work); }));
    }
    hipFree(buf);
    return 0;
}
