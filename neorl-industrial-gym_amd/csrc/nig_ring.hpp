// nig_ring.hpp -- LDS ring counters shared by the cooperating-wave kernels (nig_split.hpp, nig_split_policy.hpp, the
// paired PowerGrid form of nig_pg_lds.hpp).  Included by nig_kernels.hpp.
#pragma once

namespace nig {

// ASSUMPTION the rings rest on: the data slots are written and read with plain LDS accesses and ordered
// against the counter only by (a) the hardware rule that the DS operations of ONE wave execute in issue order on
// gfx950's LDS pipeline -- so "write data, then write counter" and "read counter, then read data" need no wait in
// between -- and (b) wavefront-scope fences that pin the COMPILER's order of those accesses.  Under the HIP / LLVM
// memory model the slots are formally racing (the fences are not workgroup-scope release / acquire); a workgroup-scope
// fence would insert an s_waitcnt lgkmcnt(0) per post and per wait, i.e. on the integrator's critical path.  The rule
// holds for this target only (not in threadgroup-split mode, not necessarily on another architecture), hence:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "nig_split.hpp relies on in-order LDS execution within a wave as implemented on gfx950; re-validate before building for another target"
#endif
// tests/test_gpu_split.py + tests/test_gpu_round3.py::test_three_wave_form_rows_against_oracle_trajectories would show a
// reordering as a bit mismatch (the spin loops could also hang: they carry no timeout on purpose -- a rollout of 10^5
// steps is legitimate -- so a toolchain change must be re-validated with those tests under a `timeout`).
//
// The ring counters are accessed through LDS-address-space pointers: a volatile access through a generic pointer
// is compiled to a system-coherent FLAT operation with a vmcnt(0) wait behind it.
using lds_u32_t = __attribute__((address_space(3))) uint32_t;

__device__ __forceinline__ uint32_t split_peek(lds_u32_t *cnt)     // the load only: no wait for its result here
{
    return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// spin until the counter (wave-uniform address) has reached `want`; returns the value seen
__device__ __forceinline__ uint32_t split_wait(lds_u32_t *cnt, uint32_t want)
{
    uint32_t v = __builtin_amdgcn_readfirstlane(split_peek(cnt));
    while (v < want) {
        __builtin_amdgcn_s_sleep(1);
        v = __builtin_amdgcn_readfirstlane(split_peek(cnt));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return v;
}
__device__ __forceinline__ void split_post(lds_u32_t *cnt, uint32_t v, unsigned lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) __hip_atomic_store(cnt, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace nig
