// lds_gather.hip -- cost of a random 16-byte-per-lane table gather from LDS (the generator's normal-transform table:
// 768 entries of four float coefficients), as one ds_read_b128 from an array of structures, as two ds_read_b64, as
// four ds_read_b32 from a structure of arrays.  8 waves per CU (2 per SIMD) all gathering, like the PowerGrid rollout.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_gather lds_gather.hip ; prints LDS-pipe cycles per gathered entry.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int N = 768, ITERS = 2048;

template <int MODE>
__global__ void __launch_bounds__(512) gather(float *out, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) float tab[4 * N];
    for (int i = threadIdx.x; i < 4 * N; i += blockDim.x) tab[i] = (float)i;
    __syncthreads();
    uint32_t x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    float acc = 0.0f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t idx = (x >> 8) % N;
            if constexpr (MODE == 0) {                      // AoS, one b128
                const float4 v = reinterpret_cast<const float4 *>(tab)[idx];
                acc += v.x + v.y * v.z + v.w;
            } else if constexpr (MODE == 1) {               // two arrays of float2
                const float2 a = reinterpret_cast<const float2 *>(tab)[idx];
                const float2 b = reinterpret_cast<const float2 *>(tab + 2 * N)[idx];
                acc += a.x + a.y * b.x + b.y;
            } else if constexpr (MODE == 2) {               // SoA, four b32
                acc += tab[idx] + tab[N + idx] * tab[2 * N + idx] + tab[3 * N + idx];
            } else {                                        // no gather: the arithmetic alone
                acc += (float)idx + (float)(idx + 1) * (float)(idx + 2) + (float)(idx + 3);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[4] = {"AoS ds_read_b128", "2 x ds_read_b64", "SoA 4 x ds_read_b32", "no gather (arithmetic only)"};
    float base_ms = 0;
    for (int m = 3; m >= 0; --m) {
        for (int rep = 0; rep < 2; ++rep) {
            auto launch = [&]() {
                if (m == 0) hipLaunchKernelGGL(gather<0>, dim3(256), dim3(512), 0, 0, out, 1u);
                if (m == 1) hipLaunchKernelGGL(gather<1>, dim3(256), dim3(512), 0, 0, out, 1u);
                if (m == 2) hipLaunchKernelGGL(gather<2>, dim3(256), dim3(512), 0, 0, out, 1u);
                if (m == 3) hipLaunchKernelGGL(gather<3>, dim3(256), dim3(512), 0, 0, out, 1u);
            };
            launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int l = 0; l < 5; ++l) launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 5;
            if (m == 3) base_ms = ms;
            // per CU: 8 waves x ITERS x 8 gathers; time x 2.1e6 cycles/ms / gathers = cycles per wave-gather on the CU's one LDS pipe
            const double gathers = 8.0 * ITERS * 8;
            printf("%-30s %8.3f ms   %.1f cycles per wave-gather (at 2.1 GHz, whole kernel)   %.1f net of the arithmetic\n", names[m], ms,
                   ms * 2.1e6 / gathers, (ms - base_ms) * 2.1e6 / gathers);
        }
    }
    return 0;
}
