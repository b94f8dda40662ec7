// store_addr.hip -- does the ADDRESSING FORM of a 1 KiB wave store change what issuing it costs?
// Build: hipcc --offload-arch=gfx950 -O3 -o store_addr store_addr.hip ; run on the GPU box.
// The fused rollouts with trajectory outputs are bound by store issue (profiles/ubench/store_vs_valu.hip: ~31 B/clk/CU
// where the vector memory path is specified at 64 B/clk/CU).  The kernels store through 64-bit per-lane addresses
// (global_store_dwordx4 v[a:a+1], v[d:d+3], off).  This measures the same store stream -- every wave writes K 1 KiB rows
// per step into its own 32 KiB region, reused every step (cache-resident: issue cost, not HBM drain) -- in three forms:
//   flat   global_store_dwordx4 v[addr64], v[data], off            64-bit address per lane
//   saddr  global_store_dwordx4 v_off32, v[data], s[base:base+1]   wave-uniform base + 32-bit lane offset
//   buffer buffer_store_dwordx4 v[data], v_off32, s[srd:srd+3], 0 offen   descriptor + 32-bit lane offset
// each plain and with the nt (streaming) bit, with 1, 2 and 4 waves per SIMD, with and without V independent fma between
// the stores.  Prints bytes per clock per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ void __launch_bounds__(1024, 1) k(float *out, float *sink, int steps, int V, int K)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *region = reinterpret_cast<char *>(out) + ((size_t)blockIdx.x * 16 + wave) * 32768;     // wave-uniform, in SGPRs
    float a0 = lane, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f;
    const float m = 1.0000001f, c = 1e-9f;
    const uint32_t off = lane * 16u;
    __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(region, 0, 32768, 0x00020000);
    for (int s = 0; s < steps; ++s) {
        for (int i = 0; i < V; i += 4) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c);
            a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        }
        v4f v = {a0, a1, a2, a3};
        for (int j = 0; j < K; ++j) {
            const uint32_t o = off + 1024u * (uint32_t)(j & 31);
            if constexpr (MODE == 0) {
                v4f *p = reinterpret_cast<v4f *>(region + o);
                if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
            } else if constexpr (MODE == 1) {
                if constexpr (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(o), "v"(v), "s"(region) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(o), "v"(v), "s"(region) : "memory");
            } else {
                v4u u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                __builtin_amdgcn_raw_buffer_store_b128(u, srd, (int)o, 0, NT ? 2 : 0);
            }
        }
    }
    if (a0 + a1 + a2 + a3 == 12345.678f) sink[0] = a0;
}

typedef void (*kern_t)(float *, float *, int, int, int);

int main()
{
    const int cus = 256, steps = 400, K = 8;
    float *out, *sink;
    CHECK(hipMalloc(&out, (size_t)cus * 16 * 32768)); CHECK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const double mhz = prop.clockRate / 1000.0;
    const char *names[6] = {"flat", "flat nt", "saddr", "saddr nt", "buffer", "buffer nt"};
    kern_t ks[6] = {k<0, false>, k<0, true>, k<1, false>, k<1, true>, k<2, false>, k<2, true>};
    printf("%d CUs, %d x 1 KiB stores per wave-step into a reused 32 KiB region per wave; clock %.0f MHz\n", cus, K, mhz);
    for (int V : {0, 1144}) for (int wps : {1, 2, 4}) {
        printf("V = %d fma per wave-step, %d wave(s) per SIMD:\n", V, wps);
        for (int m = 0; m < 6; ++m) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                for (int n = 0; n < 4; ++n) hipLaunchKernelGGL(ks[m], dim3(cus), dim3(256 * wps), 0, 0, out, sink, steps, V, K);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double us_step = ms / 4 * 1e3 / steps;
            const double bytes_cu_step = 4.0 * wps * K * 1024.0;
            printf("   %-10s %7.3f us per step   %6.1f B/clk/CU\n", names[m], us_step, bytes_cu_step / (us_step * mhz));
        }
    }
    return 0;
}
