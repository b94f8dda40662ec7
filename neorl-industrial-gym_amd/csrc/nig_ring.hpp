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
// reordering as a bit mismatch (the production spin loops carry no timeout on purpose -- a rollout of 10^5 steps is
// legitimate; a toolchain change is re-validated with those tests on the NIG_RING_SPIN_LIMIT variant below, where a slip
// is an error code, and by tests/test_ring_isa.py, which checks the order of the DS pairs in the generated ISA).
//
// The ring counters are accessed through LDS-address-space pointers: a volatile access through a generic pointer
// is compiled to a system-coherent FLAT operation with a vmcnt(0) wait behind it.
using lds_u32_t = __attribute__((address_space(3))) uint32_t;

// -DNIG_RING_MARKERS (never linked: tests/test_ring_isa.py compiles the kernels to assembly with it): comment-only asm
// statements at the fences of every post / wait, so the test can find the sites in the generated ISA and check that the
// DS operations sit on the side of the fence the protocol needs.  The production build has no markers.
#ifdef NIG_RING_MARKERS
#define NIG_RING_MARK(what) asm volatile("; NIG_RING_MARK " what)
#else
#define NIG_RING_MARK(what) ((void)0)
#endif

__device__ __forceinline__ uint32_t split_peek(lds_u32_t *cnt)     // the load only: no wait for its result here
{
    return __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#ifdef NIG_RING_SPIN_LIMIT
// TEST-ONLY build (profiles/mkvariant.sh ... "-DNIG_RING_SPIN_LIMIT=<polls>"): a slip in a ring
// protocol -- or a toolchain that reorders a data / counter pair -- must surface as an ERROR, not as a hung GPU.  A wait
// that has polled NIG_RING_SPIN_LIMIT times gives up: it records which counter it was waiting on in the ring's spare sync
// word (every sync block is 16 bytes: three counters + this word), every later wait of the block's waves on that ring
// sees the word and falls through at once, every role runs its loop to the end (on garbage), reports the word to the
// handle's error slot when it leaves (ring_report) and nig_rollout / nig_rollout_policy return NIG_ERR_HIP naming the
// ring.  The production build has none of this: its waits are the bare spin above.
__device__ __forceinline__ lds_u32_t *ring_abort_word(lds_u32_t *cnt)
{
    return (lds_u32_t *)((((size_t)cnt) & ~(size_t)15) + 12);
}
__device__ __forceinline__ uint32_t split_wait(lds_u32_t *cnt, uint32_t want)
{
    lds_u32_t *const ab = ring_abort_word(cnt);
    uint32_t v = __builtin_amdgcn_readfirstlane(split_peek(cnt));
    uint32_t polls = 0u;
    while (v < want) {
        if (__builtin_amdgcn_readfirstlane(split_peek(ab)) != 0u) break;               // a partner gave up: fall through
        if (++polls > (uint32_t)(NIG_RING_SPIN_LIMIT)) {
            // code: 0x100 | index of the counter in its sync block (0-2) | the count waited for << 16 (low 15 bits)
            const uint32_t code = 0x100u | (uint32_t)((((size_t)cnt) >> 2) & 3u) | ((want & 0x7FFFu) << 16);
            __hip_atomic_store(ab, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        v = __builtin_amdgcn_readfirstlane(split_peek(cnt));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return v;
}
// a role leaving its loop: hand a recorded time-out to the host (p.ring_err: a word of the handle's workspace)
__device__ __forceinline__ void ring_report(uint32_t *err, lds_u32_t *sync, unsigned lane)
{
    const uint32_t a = __builtin_amdgcn_readfirstlane(split_peek(ring_abort_word(sync)));
    if (a != 0u && lane == 0u && err != nullptr) atomicOr(err, a);
}
// fault injection for the variant's own test: with HF_DIAG_RING_FAULT set a producing role stops posting after 7 steps
#define NIG_RING_FAULT_GUARD(hflags, step) if (((hflags) & HF_DIAG_RING_FAULT) != 0u && (int)(step) >= 7) {} else
#define NIG_RING_REPORT(err, sync, lane) ring_report((err), (sync), (lane))
#else
// spin until the counter (wave-uniform address) has reached `want`; returns the value seen
__device__ __forceinline__ uint32_t split_wait(lds_u32_t *cnt, uint32_t want)
{
    uint32_t v = __builtin_amdgcn_readfirstlane(split_peek(cnt));
    while (v < want) {
        __builtin_amdgcn_s_sleep(1);
        v = __builtin_amdgcn_readfirstlane(split_peek(cnt));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    NIG_RING_MARK("WAIT_END");
    return v;
}
#define NIG_RING_FAULT_GUARD(hflags, step)
#define NIG_RING_REPORT(err, sync, lane) ((void)0)
#endif
__device__ __forceinline__ void split_post(lds_u32_t *cnt, uint32_t v, unsigned lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    NIG_RING_MARK("POST_BEGIN");
    if (lane == 0) __hip_atomic_store(cnt, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    NIG_RING_MARK("POST_END");
}

}  // namespace nig
