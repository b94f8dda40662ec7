// store_pattern.hip -- what sustained write rate does the rollout's OUTPUT PATTERN reach, with nothing else going on?
// Build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip ; run on the GPU box.
// The recorder wave of split_rollout_kernel writes, per step and per 64 lanes: 3 x 1 KB of the row-major
// observation block [T][B][12] (lane-contiguous 16-byte pieces) + 256 B of the reward row + 256 B of the flag row.
// This kernel issues exactly those stores (same addresses, same order) for B = 65 536, T = 250 from N waves per
// CU, optionally paced with s_sleep, with streaming (nt) or plain stores, and optionally with the block -> lanes
// assignment permuted; back-to-back launches over a fresh 1.1 GB region each, HIP events around them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int S = 12, B = 65536, T = 250;

template <bool NT>
__device__ __forceinline__ void st(v4f *p, v4f v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT>
__device__ __forceinline__ void st(float *p, float v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// waves_per_group: how many waves share the stores of one 64-lane group's step (1: one wave issues all 5; 3: one
// wave per 1 KB piece + the rows on the first).  pace: s_sleep units (64 clocks each) per step.
template <bool NT>
__global__ void __launch_bounds__(1024) pattern(float *obs, float *rew, float *fl, int pace, int groups_per_block, int split, int steps)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned group = wave % groups_per_block, part = wave / groups_per_block;     // part < split
    const unsigned g = blockIdx.x * groups_per_block + group;                          // 64-lane group, [0, 1024)
    v4f v = {1.0f * g, 2.0f, 3.0f, (float)lane};
    for (int t = 0; t < steps; ++t) {
        v4f *oo = reinterpret_cast<v4f *>(obs + ((size_t)t * B + g * 64u) * S);
        if (split == 1) {
            st<NT>(oo + lane, v); st<NT>(oo + lane + 64, v); st<NT>(oo + lane + 128, v);
            st<NT>(rew + (size_t)t * B + g * 64u + lane, v.x); st<NT>(fl + (size_t)t * B + g * 64u + lane, v.w);
        } else {
            st<NT>(oo + lane + 64 * part, v);
            if (part == 0) { st<NT>(rew + (size_t)t * B + g * 64u + lane, v.x); st<NT>(fl + (size_t)t * B + g * 64u + lane, v.w); }
        }
        v.y += 1.0f;
        for (int k = 0; k < pace; ++k) __builtin_amdgcn_s_sleep(8);      // 8 x 64 clocks
    }
}

int main(int argc, char **argv)
{
    const int launches = 24;
    const size_t obs_n = (size_t)T * B * S, row_n = (size_t)T * B;
    float *obs, *rew, *fl;
    CHECK(hipMalloc(&obs, obs_n * 4)); CHECK(hipMalloc(&rew, row_n * 4)); CHECK(hipMalloc(&fl, row_n * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = (double)(obs_n + 2 * row_n) * 4;
    struct Cfg { const char *name; bool nt; int pace, gpb, split; } cfgs[] = {
        {"nt, unpaced, 4 groups/block (1 wave per group)", true, 0, 4, 1},
        {"plain, unpaced, 4 groups/block", false, 0, 4, 1},
        {"nt, paced ~0.8 us/step, 4 groups/block", true, 3, 4, 1},
        {"nt, paced ~0.55 us/step, 4 groups/block", true, 2, 4, 1},
        {"plain, paced ~0.8 us/step, 4 groups/block", false, 3, 4, 1},
        {"nt, unpaced, 3 waves per group", true, 0, 4, 3},
        {"nt, paced ~0.8 us/step, 3 waves per group", true, 3, 4, 3},
        {"nt, unpaced, 1 group/block (1024 blocks)", true, 0, 1, 1},
    };
    for (const Cfg &c : cfgs) {
        const int blocks = 1024 / c.gpb, threads = 64 * c.gpb * c.split;
        for (int rep = 0; rep < 2; ++rep) {
            for (int w = 0; w < 3; ++w) {
                if (c.nt) hipLaunchKernelGGL(pattern<true>, dim3(blocks), dim3(threads), 0, 0, obs, rew, fl, c.pace, c.gpb, c.split, T);
                else hipLaunchKernelGGL(pattern<false>, dim3(blocks), dim3(threads), 0, 0, obs, rew, fl, c.pace, c.gpb, c.split, T);
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int l = 0; l < launches; ++l) {
                if (c.nt) hipLaunchKernelGGL(pattern<true>, dim3(blocks), dim3(threads), 0, 0, obs, rew, fl, c.pace, c.gpb, c.split, T);
                else hipLaunchKernelGGL(pattern<false>, dim3(blocks), dim3(threads), 0, 0, obs, rew, fl, c.pace, c.gpb, c.split, T);
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-52s %7.1f us per launch  %5.2f TB/s\n", c.name, ms * 1e3 / launches, bytes * launches / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
