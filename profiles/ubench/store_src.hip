// store_src.hip -- does the REGISTER FILE the store data comes from matter for what a 1 KiB store costs the issuing SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 -o store_src store_src.hip ; run on the GPU box.
// store_vs_valu.hip / store_spread.hip (rounds 3-4): with four waves per SIMD and ~1 150 vector instructions per wave-step, eight
// 1 KiB streaming stores per wave-step cost ~1 us per step of SIMD time wherever in the step they are issued -- ~60 cycles per
// store, about what reading 64 lanes x 16 bytes of store data out of the vector register file takes.  gfx90a and later let DS
// and VMEM instructions name ACCUMULATION registers (a[..]) as their data: do stores fed from AGPRs cost the vector pipe less?
//   mode 0  no stores                                    -> the vector floor
//   mode 1  K stores per step, data in v[..] (compiler)   -> rounds 3-4's figure
//   mode 2  K stores per step, data in a[..] (inline asm) -> the question
//   mode 3  K stores per step, data in v[..] via the same inline asm (control for the asm form)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024, 1) k(float *out, float *sink, int steps, int V, int K)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    v4f *mine = reinterpret_cast<v4f *>(out) + ((size_t)blockIdx.x * 16 + wave) * 2048;   // 32 KiB per wave, reused every step
    {                                                          // wave-uniform: make the compiler keep it in scalar registers
        const uint64_t u = (uint64_t)mine;
        // (the casts to uint32_t matter: the builtin returns int, and a low half >= 2^31 sign-extended into the high half sent the
        // first version of this file to an unmapped address on its second run -- "Memory access fault", gpurun_out/r05_s5)
        mine = (v4f *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(u >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)u));
    }
    const uint32_t voff = lane * 16u;
    float a0 = lane, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f;
    const float m = 1.0000001f, c = 1e-9f;
    v4f d = {(float)lane, 1.0f, 2.0f, 3.0f};                 // loop-invariant store data (the real kernel's comes from LDS reads)
    for (int s = 0; s < steps; ++s) {
        for (int i = 0; i < V; i += 4) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c);
            a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        }
        if (MODE == 1) {
            v4f v = {a0, a1, a2, a3};
            for (int j = 0; j < K; ++j) __builtin_nontemporal_store(v, mine + lane + 64 * j);
        } else if (MODE == 2) {
            for (int j = 0; j < K; ++j) {
                const v4f *p = mine + 64 * j;
                asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(voff), "a"(d), "s"(p) : "memory");
            }
        } else if (MODE == 3) {
            for (int j = 0; j < K; ++j) {
                const v4f *p = mine + 64 * j;
                asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(voff), "v"(d), "s"(p) : "memory");
            }
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
    float ms[4];
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            CHECK(hipEventRecord(e0));
            for (int n = 0; n < 4; ++n) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(cus), dim3(1024), 0, 0, out, sink, steps, V, K);
            }
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
            ms[mode] /= 4;
        }
    printf("4 waves per SIMD, %d fma per wave-step, %d x 1 KiB streaming stores per wave-step (cache-resident targets), %d steps\n", V, K, steps);
    const char *name[4] = {"no stores", "data in VGPRs (compiler)", "data in AGPRs (asm)", "data in VGPRs (asm)"};
    for (int mode = 0; mode < 4; ++mode)
        printf("mode %d  %-28s %.3f us per step   (+%.3f us)\n", mode, name[mode], ms[mode] * 1e3 / steps, (ms[mode] - ms[0]) * 1e3 / steps);
    return 0;
}
