// store_vs_valu.hip -- does ISSUING global stores cost the SIMD vector-ALU time?
// Build: hipcc --offload-arch=gfx950 -O3 -o store_vs_valu store_vs_valu.hip ; run on the GPU box.
// PowerGrid's LDS-resident rollout pays ~0.45 ms per 250 steps for issuing eight 1 KiB trajectory stores per wave-step
// although the stores may stay in cache (profiles/r03/pg_store_probe.txt) and although three other waves share its SIMD:
// the cost looks serialized with the vector work.  This isolates the question.  One 1 024-thread block per CU (four
// waves per SIMD).  Every wave runs STEPS "steps"; a step = V dependent-free v_fma_f32 (four accumulators) followed by
// K 1 KiB streaming stores into the wave's own 8 KiB region (reused every step: cache-resident, no HBM drain).
//   mode 0  every wave: V fma, no stores                      -> the vector floor
//   mode 1  every wave: V fma + K stores (PowerGrid's shape)  -> what interleaving costs
//   mode 2  waves 0-2 of a SIMD: V fma only; wave 3: only stores, 4 K per step (the same bytes per SIMD)  -> a "recorder wave"
// Prints the kernel time per step for each; (1) - (0) is the store cost when every wave stores, (2) - (0) when one does.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024, 1) k(float *out, float *sink, int steps, int V, int K)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;      // wave w runs on SIMD w % 4
    const bool recorder = MODE == 2 && (wave >> 2) == 3;                    // the fourth wave of each SIMD
    v4f *mine = reinterpret_cast<v4f *>(out) + ((size_t)blockIdx.x * 16 + wave) * 2048;   // 32 KiB per wave (4 K stores of 1 KiB)
    float a0 = lane, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f;
    const float m = 1.0000001f, c = 1e-9f;
    for (int s = 0; s < steps; ++s) {
        if (!recorder) {
            for (int i = 0; i < V; i += 4) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c);
                a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            }
        }
        const int nst = MODE == 0 ? 0 : (MODE == 1 ? K : (recorder ? 4 * K : 0));
        v4f v = {a0, a1, a2, a3};
        for (int j = 0; j < nst; ++j) __builtin_nontemporal_store(v, mine + lane + 64 * (j & 31));
    }
    if (a0 + a1 + a2 + a3 == 12345.678f) sink[0] = a0;                      // keep the arithmetic alive
}

int main()
{
    const int cus = 256, steps = 400, V = 1144, K = 8;                      // V ~ PowerGrid's vector instructions per wave-step
    float *out, *sink;
    CHECK(hipMalloc(&out, (size_t)cus * 16 * 32768)); CHECK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms[3];
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            CHECK(hipEventRecord(e0));
            for (int n = 0; n < 4; ++n) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K);
            }
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
            ms[mode] /= 4;
        }
    printf("4 waves per SIMD, %d fma per wave-step, %d x 1 KiB stores per wave-step (cache-resident targets), %d steps\n", V, K, steps);
    printf("mode 0 (no stores)                 %.3f us per step\n", ms[0] * 1e3 / steps);
    printf("mode 1 (every wave stores)         %.3f us per step   (+%.3f us)\n", ms[1] * 1e3 / steps, (ms[1] - ms[0]) * 1e3 / steps);
    printf("mode 2 (3 fma waves + 1 recorder)  %.3f us per step   (3/4 of the fma work; stores of a whole SIMD on one wave)\n", ms[2] * 1e3 / steps);
    printf("   mode 0 scaled to 3 waves' work: %.3f us per step\n", ms[0] * 0.75 * 1e3 / steps);
    return 0;
}
