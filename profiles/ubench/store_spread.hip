// store_spread.hip -- WHERE in a wave's step its stores are issued: does that change what they cost?
// Build: hipcc --offload-arch=gfx950 -O3 -o store_spread store_spread.hip ; run on the GPU box.
// store_vs_valu.hip (round 3) showed: four waves per SIMD, 1 144 fma per wave-step, eight 1 KiB stores per wave-step issued
// back to back at the END of the step cost +1.83 us per step -- exactly 128 KiB per CU and step through a ~31 B/clk store
// path, ADDED to the vector time instead of hidden behind it.  Added means the waves of a CU reach their store burst
// together (same instruction count per step): while they queue for the store path nobody computes, and while they
// compute the path idles -- a convoy.  This asks whether the placement of the stores breaks it:
//   mode 0  no stores                                              -> the vector floor
//   mode 1  every wave: V fma, then K stores back to back          -> PowerGrid's shape (round 3)
//   mode 3  every wave: one store after every V / K fma            -> the same stores spread evenly through the step
//   mode 4  every wave: the K-store burst after (i / 4) V fma of the step, i = the wave's index on its SIMD (0-3)
//           -> bursts as in mode 1, the four waves of a SIMD a quarter step out of phase BY CONSTRUCTION (a start-up
//           delay, profiles/r03/pg_stagger.txt, did nothing: the queue re-forms the convoy)
//   mode 5  every wave: two half bursts (K / 2 stores) half a step apart
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

#define FMA_CHUNK(n) for (int i_ = 0; i_ < (n); i_ += 4) { a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c); }

template <int MODE>
__global__ void __launch_bounds__(1024, 1) k(float *out, float *sink, int steps, int V, int K)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;      // wave w runs on SIMD w % 4
    const int phase = (int)(wave >> 2);                                     // index of the wave on its SIMD
    v4f *mine = reinterpret_cast<v4f *>(out) + ((size_t)blockIdx.x * 16 + wave) * 2048;
    float a0 = lane, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f;
    const float m = 1.0000001f, c = 1e-9f;
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0) { FMA_CHUNK(V) }
        if (MODE == 1) {
            FMA_CHUNK(V)
            v4f v = {a0, a1, a2, a3};
            for (int j = 0; j < K; ++j) __builtin_nontemporal_store(v, mine + lane + 64 * j);
        }
        if (MODE == 3) {
            for (int j = 0; j < K; ++j) {
                FMA_CHUNK(V / K)
                v4f v = {a0, a1, a2, a3};
                __builtin_nontemporal_store(v, mine + lane + 64 * j);
            }
        }
        if (MODE == 4) {
            FMA_CHUNK(phase * (V / 4))
            v4f v = {a0, a1, a2, a3};
            for (int j = 0; j < K; ++j) __builtin_nontemporal_store(v, mine + lane + 64 * j);
            FMA_CHUNK(V - phase * (V / 4))
        }
        if (MODE == 5) {
            FMA_CHUNK(V / 2)
            v4f v = {a0, a1, a2, a3};
            for (int j = 0; j < K / 2; ++j) __builtin_nontemporal_store(v, mine + lane + 64 * j);
            FMA_CHUNK(V / 2)
            v4f w = {a0, a1, a2, a3};
            for (int j = K / 2; j < K; ++j) __builtin_nontemporal_store(w, mine + lane + 64 * j);
        }
    }
    if (a0 + a1 + a2 + a3 == 12345.678f) sink[0] = a0;
}

int main()
{
    const int cus = 256, steps = 400, V = 1144, K = 8;
    float *out, *sink;
    CHECK(hipMalloc(&out, (size_t)cus * 16 * 32768)); CHECK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int modes[5] = {0, 1, 3, 4, 5};
    float ms[5];
    for (int rep = 0; rep < 2; ++rep)
        for (int q = 0; q < 5; ++q) {
            CHECK(hipEventRecord(e0));
            for (int n = 0; n < 4; ++n) {
                switch (modes[q]) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K); break;
                default: hipLaunchKernelGGL(k<5>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K); break;
                }
            }
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms[q], e0, e1));
            ms[q] /= 4;
        }
    printf("4 waves per SIMD, %d fma per wave-step, %d x 1 KiB stores per wave-step (cache-resident targets), %d steps\n", V, K, steps);
    const char *names[5] = {"mode 0 (no stores)", "mode 1 (burst at the end of the step)", "mode 3 (one store every V/K fma)",
                            "mode 4 (burst, SIMD's waves a quarter step apart)", "mode 5 (two half bursts)"};
    for (int q = 0; q < 5; ++q)
        printf("%-52s %.3f us per step   (+%.3f us)\n", names[q], ms[q] * 1e3 / steps, (ms[q] - ms[0]) * 1e3 / steps);
    return 0;
}
