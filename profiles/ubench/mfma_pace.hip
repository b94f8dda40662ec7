// mfma_pace.hip -- what fraction of a SIMD's matrix-pipe cycles can the MFMA actor's instruction pattern fill?  (round 5)
// v_mfma_f32_32x32x2_f32 is 64 cycles per SIMD; rollout_mlp_kernel keeps the pipe busy 0.82-0.86 of the time at two waves per SIMD
// (profiles/r05/mlp_*_sq.txt), and neither its chunk barriers nor its LDS-DMA bursts are the rest (mlp_barrier_bound.txt,
// mlp_spread_fill_ab.txt).  This times bare loops of the same shape -- one accumulator chain per wave, the A operand from an
// LDS ring read eight records ahead, the B operand cycling through sixteen registers, a v_max_f32 per MFMA in the head part --
// at one and two waves per SIMD, and prints cycles per MFMA per SIMD (64 = the pipe never idles).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_pace mfma_pace.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int CHUNK = 144, CHUNKS = 2048;          // MFMAs per loop body, loop trips: 294 912 MFMAs per wave

// MODE 0: operands in registers; 1: + A operand through an LDS ring, eight reads in flight; 2: 1 + v_max_f32 on the B operand;
// 3: 2 + a block barrier per chunk
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(float *out, const float *w, unsigned long long *cyc)
{
    __shared__ float s_w[2][148 * 64];
    const unsigned lane = threadIdx.x & 63u;
    for (int i = threadIdx.x; i < 2 * 148 * 64; i += 256) (&s_w[0][0])[i] = w[i & 1023];
    __syncthreads();
    f32x16 h[8];
    for (int m = 0; m < 8; ++m)
        for (int r = 0; r < 16; ++r) h[m][r] = (float)((lane + m * 16 + r) & 15) * 0.001f;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int buf = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();      // shader cycles of THIS compute unit
    for (int c = 0; c < CHUNKS; ++c) {
        if constexpr (MODE == 3) __syncthreads();
        const float *wb = &s_w[buf][lane];
        float ring[8];
        if constexpr (MODE >= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ring[j] = wb[j * 64];
        }
#pragma unroll
        for (int i = 0; i < CHUNK; ++i) {
            float a = 0.25f;
            if constexpr (MODE >= 1) {
                a = ring[i % 8];
                if (i + 8 < CHUNK) ring[i % 8] = wb[(i + 8) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            float b = h[(i / 16) & 7][i % 16];
            if constexpr (MODE >= 2) { if (i >= 128) b = fmaxf(b, 0.0f); }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if constexpr (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
        }
        buf ^= 1;
    }
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    asm volatile("s_nop 0" :: "v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static int run(const char *name, int blocks_per_cu, int cus, float *out, const float *w, unsigned long long *cyc)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = cus * blocks_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, w, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, w, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double wave_cycles = (double)h[h.size() / 2];                      // median wave: s_memtime ticks = shader cycles
    const double mfma_per_simd = (double)CHUNK * CHUNKS * blocks_per_cu;     // four waves of a block = one per SIMD
    const double c = wave_cycles / mfma_per_simd;
    printf("%-58s %d wave(s) per SIMD: %6.2f cycles per MFMA per SIMD  (pipe busy %.3f; %.2f ms, %.0f MHz)\n", name, blocks_per_cu,
           c, 64.0 / c, ms, wave_cycles / (ms * 1e3));
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float *out, *w;
    CHECK(hipMalloc(&out, (size_t)cus * 2 * 256 * 4));
    CHECK(hipMalloc(&w, 1024 * 4));
    CHECK(hipMemset(w, 0, 1024 * 4));
    unsigned long long *cyc;
    CHECK(hipMalloc(&cyc, (size_t)cus * 2 * 4 * 8));
    printf("%d CUs\n", cus);
    for (int b = 1; b <= 2; ++b) {
        if (run<0>("operands in registers", b, cus, out, w, cyc)) return 1;
        if (run<1>("A operand through an LDS ring (eight reads ahead)", b, cus, out, w, cyc)) return 1;
        if (run<2>("... + v_max_f32 on the B operand of the last 16", b, cus, out, w, cyc)) return 1;
        if (run<3>("... + a block barrier per 144 MFMAs", b, cus, out, w, cyc)) return 1;
    }
    return 0;
}
