// nig_kernels.hpp -- HIP kernel templates (gfx950) of libnig.so, one instantiation set per environment.
//
// One wavefront lane per environment instance.  State, actions, noise and outputs are
// structure-of-arrays ([row][lane], row pitch ld) so every global access of a wave is one
// fully coalesced 256-byte row segment.  The step kernel fuses the whole of
// IndustrialEnv.step (environments/base.py:157-213): clip -> constraint checks on the
// pre-state -> dynamics -> reward -> penalties -> counters -> done/truncation -> critical
// shutdown -> (optional) episode tally flush and in-kernel auto-reset.  No MFMA: these are
// elementwise ODE updates (HBM-bound, DESIGN.md "Roofline").
//
// Translation units: every environment's kernels are instantiated in a file of their own
// (env_*.hip: `NIG_DEFINE_ENV_LAUNCH(Env, name)`), the C ABI and the env-independent kernels
// live in nig_api.hip; _build.py compiles them in parallel and links libnig.so.
// Build flags: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <type_traits>

#include "../../include/nig.h"
#include "nig_envs.hpp"

namespace nig {

constexpr int BLOCK = 256;

// One 16-byte store per call.  A HIP float4 assignment is scalarised and re-merged by hipcc, which can
// pick 12+16+16+4-byte pieces for a 48-byte row (misaligned dwordx4: -20 % on the row-major
// trajectory); a native vector store stays one aligned global_store_dwordx4.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(float *dst16, float a, float b, float c, float d)
{
    v4f v = {a, b, c, d};
    *reinterpret_cast<v4f *>(dst16) = v;
}
// Per-step rollout outputs are written once and read by nobody on the device: streaming (nt) stores
// keep them from evicting the action ring and the generator table from L2 (+5..16 % on the fused
// rollout).  Only for stores that cover whole lines per instruction -- nt on the lane-strided 16-byte
// pieces of an untransposed row-major row HALVED the 1M-lane rate (no write-combining in L2).
template <class T>
__device__ __forceinline__ void stream_store(T *dst, T v)
{
#ifdef NIG_DIAG_STORE_POLICY           // (diagnostic builds only, profiles/r05: another cache policy for the 16-byte trajectory stores --
    // 1 = sc1 (write-through, dropped from L2), 2 = sc0 sc1, 3 = nt sc1; the production nt keeps the line in L2)
    if constexpr (sizeof(T) == 16) {
#if NIG_DIAG_STORE_POLICY == 1
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
#elif NIG_DIAG_STORE_POLICY == 2
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
#else
        asm volatile("global_store_dwordx4 %0, %1, off nt sc1" :: "v"(dst), "v"(v) : "memory");
#endif
        return;
    }
#endif
    __builtin_nontemporal_store(v, dst);
}
constexpr int REDUCE_BLOCKS = 256;
constexpr int64_t POLICY_BYTES = 2048;     // device copy of nig_policy at the workspace tail

struct StepArgs {
    // library-owned
    float *state; uint32_t *ctr; long long *life_viol; double *ep_ret; double *tally;
    uint32_t ld; uint32_t B;          // 32-bit on purpose: row offsets k*ld stay in scalar registers
    uint32_t ld_state;                // pitch of the state rows (== ld unless the caller bound its own array)
    // caller-owned
    const float *actions; uint32_t ld_act;
    const double *actions64;          // nig_step64: the same rows as float64 (actions is then unused)
    const double *step_noise; const double *reset_noise; uint32_t ld_noise;
    float *reward; double *reward64; uint32_t *flags; float *final_obs; uint32_t ld_obs;
    // scalars
    uint64_t env0; uint32_t seed_lo, seed_hi;
    const uint32_t *t_ptr; uint32_t t_off;   // launch counter t = (t_ptr ? *t_ptr : 0) + t_off (graph replay keeps t on the device)
    int max_steps; float dt32; double dt; uint32_t hflags; uint32_t cmask;
    int n_en;                         // enabled built-in constraints = SafetyMetrics.total_constraints of every step (base.py:115)
    // host side only (which kernel form a launch takes, nig_tune): thresholds in effect for this handle's device
    uint32_t split_blocks, wide_min_blocks;
    uint32_t *ring_err;               // device word a timed-out ring wait is reported in (NIG_RING_SPIN_LIMIT builds only; nig_ring.hpp)
    // nig_step_host: a second copy of every lane's post-step state rows, [S][ld_mirror], written by the step kernel itself
    // (the host-buffer entry points used to launch a row-gather kernel behind every step: one launch less per env.step)
    float *mirror; uint32_t ld_mirror;
};
// internal bit of StepArgs::hflags (above the public NIG_F_* bits): some lane of the handle may hold
// NIG_CTR_DONE although the handle auto-resets (never reset, left out by reset(mask), set by
// nig_set_state); cleared by a full nig_reset.  Lets the rollout kernel keep its no-freeze fast path.
constexpr uint32_t HF_MAY_HOLD_DONE = 0x10000u;
// test-only (NIG_RING_SPIN_LIMIT builds, nig_ring.hpp): producing roles stop posting after 7 steps
constexpr uint32_t HF_DIAG_RING_FAULT = 0x20000u;
}  // namespace nig
#include "nig_ring.hpp"        // LDS ring counters of the cooperating-wave kernels (used from rollout_body's RING form on)
namespace nig {

// IndustrialEnv.step for one lane, entirely in registers (base.py:157-213): action clip, constraint
// check on the pre-state and dynamics, then post_core = reward / penalties / termination on the
// finished transition.
// AT = float, or double when the caller hands float64 actions over (nig_step64): base.py:167 clips a float64 array
// against the float32 bounds without casting, and the envs' arithmetic follows NumPy's promotion from there.
template <class Env, class AT>
__device__ __forceinline__ void clip_action(AT (&a)[Env::A])
{
#pragma unroll
    for (int k = 0; k < Env::A; ++k) {            // base.py:167 np.clip(action, -1, 1) == min(max(x,lo),hi)
        AT x = a[k];
        if constexpr (std::is_same<AT, float>::value) {
            // NumPy's maximum / minimum hand a NaN on, and so do gfx950's v_maximum3_f32 / v_minimum3_f32 (IEEE 754-2019
            // maximum / minimum): two instructions where compare + select pairs are four.  (Neither limit is a zero,
            // so the sign of a zero result never comes from a limit; an action inside the limits is returned as it is.)
            x = __builtin_elementwise_maximum(x, -1.0f);
            x = __builtin_elementwise_minimum(x, 1.0f);
        } else {
            x = (x < (AT)-1) ? (AT)-1 : x;
            x = (x > (AT)1) ? (AT)1 : x;
        }
        a[k] = x;
    }
}

// the reward's type at the end of the reference's arithmetic: float64 as soon as the action is float64
template <class Env, class AT> using reward_of = std::conditional_t<std::is_same<AT, double>::value, double, typename Env::reward_t>;

template <class Env, class R>
__device__ __forceinline__ void post_finish(R r, bool term, uint32_t vb, int step_pre, int max_steps, StepResult<Env, R> &out);

template <class Env, class AT>
__device__ __forceinline__ void post_core(const float (&n)[Env::S], const AT (&a)[Env::A], uint32_t vb,
                                          int step_pre, int max_steps, StepResult<Env, reward_of<Env, AT>> &out)
{
    post_finish<Env, reward_of<Env, AT>>(Env::reward(n, a) /* base.py:176 */, Env::done(n) /* base.py:190 */, vb, step_pre, max_steps, out);
}

// the env-independent rest of IndustrialEnv.step once the reward and the env's own termination test are known
template <class Env, class R>
__device__ __forceinline__ void post_finish(R r, bool term, uint32_t vb, int step_pre, int max_steps, StepResult<Env, R> &out)
{
#pragma unroll
    for (int k = 0; k < 3; ++k)                   // base.py:179-183, constraint order
        r = (vb & (1u << k)) ? (R)(r + (R)Env::penalty(k)) : r;
    const int nviol = __popc(vb);
    const int ncrit = __popc(vb & Env::CRIT_MASK);
    const bool trunc = (step_pre + 1) >= max_steps;   // base.py:191
    if (ncrit > 0) { term = true; r = r - (R)1000; }  // base.py:195-198
    out.reward = r; out.viol_bits = vb; out.nviol = nviol; out.ncrit = ncrit;
    out.terminated = term; out.truncated = trunc; out.shutdown = ncrit > 0;   // info['critical_shutdown'], base.py:210
}

template <class Env, class NZ, class AT>
__device__ __forceinline__ void step_core(const float (&s)[Env::S], AT (&a)[Env::A],
                                          const NZ (&nz)[Env::KS > 0 ? Env::KS : 1], int step_pre,
                                          int max_steps, float dt32, double dt, uint32_t cmask,
                                          float (&n)[Env::S], StepResult<Env, reward_of<Env, AT>> &out)
{
    if constexpr (Env::CUSTOM_STEP) {             // the Advanced envs override step() wholesale
        Env::custom_step(s, a, step_pre, max_steps, dt32, n, out);
        out.viol_bits &= cmask;
        out.nviol = __popc(out.viol_bits);
        return;
    } else {
        clip_action<Env, AT>(a);
        const uint32_t vb = Env::violated(s, a) & cmask;   // base.py:170 (and again :180, same inputs); cmask: base.py:224-228
        Env::dynamics(s, a, nz, dt32, dt, n);         // base.py:173
        post_core<Env, AT>(n, a, vb, step_pre, max_steps, out);
    }
}

// The per-lane flag word of one step (include/nig.h NIG_FLAG_*).
template <class Env, class R>
__device__ __forceinline__ uint32_t pack_flags(const StepResult<Env, R> &res, int step)
{
    uint32_t f = (res.terminated ? NIG_FLAG_TERMINATED : 0u) | (res.truncated ? NIG_FLAG_TRUNCATED : 0u) |
                 ((res.viol_bits & 7u) << NIG_FLAG_VIOL_SHIFT) | (((uint32_t)res.nviol & 3u) << NIG_FLAG_NVIOL_SHIFT) |
                 ((uint32_t)res.ncrit << NIG_FLAG_NCRIT_SHIFT) | (res.shutdown ? NIG_FLAG_SHUTDOWN : 0u) |
                 ((uint32_t)step << NIG_FLAG_STEP_SHIFT);
    if constexpr (Env::CUSTOM_STEP)                // only the Advanced envs carry a 4th condition / a count of 4
        f |= ((res.viol_bits & 8u) ? NIG_FLAG_VIOL3 : 0u) | (((uint32_t)res.nviol & 4u) ? NIG_FLAG_NVIOL_HI : 0u);
    return f;
}

// Per-lane key of the counter-based generator: (global env index, launch counter t).
__device__ __forceinline__ RngKey make_key(uint64_t gi, uint32_t t, uint32_t seed_lo, uint32_t seed_hi,
                                           const float4 *tab = nullptr)
{
    RngKey k;
    k.env_lo = (uint32_t)gi; k.env_hi = (uint32_t)(gi >> 32);
    k.t = t; k.seed_lo = seed_lo; k.seed_hi = seed_hi; k.tab = tab;
    return k;
}

// Env hooks that only some envs have, callable from generic lambdas (where a discarded
// `if constexpr` branch is still name-checked because Env is not the lambda's own parameter).
template <class Env>
__device__ __forceinline__ u32x4 pair_block(const RngKey &k)
{
    if constexpr (Env::SHARED_STEP_BLOCK) return Env::step_block(k);
    else return u32x4{0u, 0u, 0u, 0u};
}
template <class Env, class NZ>
__device__ __forceinline__ void pair_noise(uint32_t w0, uint32_t w1, const float4 *tab, NZ (&n)[Env::KS > 0 ? Env::KS : 1])
{
    if constexpr (Env::SHARED_STEP_BLOCK) Env::step_noise(w0, w1, tab, n);
}
template <class Env>
__device__ __forceinline__ void pair_fetch(uint32_t w0, uint32_t w1, const float4 *tab, ProbitFetch (&f)[Env::KS > 0 ? Env::KS : 1])
{
    if constexpr (Env::SHARED_STEP_BLOCK) Env::step_noise_fetch(w0, w1, tab, f);
}
template <class Env, class NZ>
__device__ __forceinline__ void pair_eval(const ProbitFetch (&f)[Env::KS > 0 ? Env::KS : 1], NZ (&n)[Env::KS > 0 ? Env::KS : 1])
{
    if constexpr (Env::SHARED_STEP_BLOCK) Env::step_noise_eval(f, n);
}
template <class Env, class NZ>
__device__ __forceinline__ void draw_one(const RngKey &k, NZ (&n)[Env::KS > 0 ? Env::KS : 1])
{
    if constexpr (Env::KS > 0) Env::draw_step(k, n);
}

// Stage the 12 KiB probit table (normal transform of the generator) in LDS.  Every thread of the block
// must pass through here before any early exit.
#define NIG_STAGE_PROBIT(tab)                                                         \
    __shared__ float4 tab[768];                                                       \
    for (int i_ = (int)threadIdx.x; i_ < 768; i_ += BLOCK) tab[i_] = NIG_PROBIT[i_];  \
    __syncthreads()

// Episode bookkeeping of one finished episode (utils.py:120-125), lane-private column of the tally.
// All 11 rows are loaded before any is stored: one memory round trip instead of eleven dependent ones.
__device__ __forceinline__ void flush_tally(double *T, uint32_t ld, double ret, int step, uint32_t viol_ep, int ncrit, int n_en)
{
    double v[NIG_T_ROWS];
#pragma unroll
    for (int r = 0; r < NIG_T_ROWS; ++r) v[r] = T[(size_t)r * ld];
    const double len = (double)step;
    v[NIG_T_EPISODES] += 1.0;
    v[NIG_T_RET_SUM] += ret;
    v[NIG_T_RET_SQ] += ret * ret;
    v[NIG_T_RET_MIN] = fmin(v[NIG_T_RET_MIN], ret);
    v[NIG_T_RET_MAX] = fmax(v[NIG_T_RET_MAX], ret);
    v[NIG_T_LEN_SUM] += len;
    v[NIG_T_LEN_SQ] += len * len;
    v[NIG_T_VIOL] += (double)viol_ep;
    v[NIG_T_CRIT] += (double)ncrit;            // a critical step always ends the episode
    v[NIG_T_SHUTDOWN] += (ncrit > 0) ? 1.0 : 0.0;
    v[NIG_T_SUCCESS] += (ret > 0.0) ? 1.0 : 0.0;
    v[NIG_T_SATISFIED] += (double)(n_en * step - (int)viol_ep);   // sum over the episode's steps of constraints_satisfied
    v[NIG_T_CONSTRAINTS] += (double)(n_en * step);
#pragma unroll
    for (int r = 0; r < NIG_T_ROWS; ++r) T[(size_t)r * ld] = v[r];
}

// The same bookkeeping as no-return float64 atomics into the lane's own column (global_atomic_add / min / max_f64,
// executed at the memory side): nothing is loaded, nothing is waited for.  The step kernel used flush_tally, i.e. 13
// loads, a wait and 13 stores behind the step of every finishing lane -- a third dependent memory round trip on the
// critical path of a launch that is latency-bound at the headline batch (profiles/r03/step_api_probe.py: the tally cost
// 0.8 us of a 5.1 us launch).  Each row is one IEEE operation on the same operands as in flush_tally, and a lane's column
// is touched by that lane only (a kernel boundary orders consecutive steps), so the rows hold the same bits.
__device__ __forceinline__ void flush_tally_atomic(double *T, uint32_t ld, double ret, int step, uint32_t viol_ep, int ncrit, int n_en)
{
    typedef __attribute__((address_space(1))) double gdouble;
    auto add = [&](int r, double x) { (void)__builtin_amdgcn_global_atomic_fadd_f64((gdouble *)(T + (size_t)r * ld), x); };
    const double len = (double)step;
    add(NIG_T_EPISODES, 1.0);
    add(NIG_T_RET_SUM, ret);
    add(NIG_T_RET_SQ, ret * ret);
    (void)__builtin_amdgcn_global_atomic_fmin_f64((gdouble *)(T + (size_t)NIG_T_RET_MIN * ld), ret);
    (void)__builtin_amdgcn_global_atomic_fmax_f64((gdouble *)(T + (size_t)NIG_T_RET_MAX * ld), ret);
    add(NIG_T_LEN_SUM, len);
    add(NIG_T_LEN_SQ, len * len);
    add(NIG_T_VIOL, (double)viol_ep);
    add(NIG_T_CRIT, (double)ncrit);
    add(NIG_T_SHUTDOWN, (ncrit > 0) ? 1.0 : 0.0);
    add(NIG_T_SUCCESS, (ret > 0.0) ? 1.0 : 0.0);
    add(NIG_T_SATISFIED, (double)(n_en * step - (int)viol_ep));
    add(NIG_T_CONSTRAINTS, (double)(n_en * step));
}

// Which of the two the step kernel uses: the atomics unless the env says otherwise.  They execute at the memory side at
// ~1.3 TB/s chip-wide (MI355X_MICROARCH.md "Global float atomics"): nothing for an env whose lanes finish rarely
// (ChemicalReactor 0.3 % per step, RobotAssembly 2.4 %), but PowerGrid finishes 18 % of its lanes every step -- 19 bytes
// of atomic traffic per env-step, ~16 % of its step launch at 262 144 lanes -- so it keeps the load / store flush.
template <class E, class = void> struct tally_atomic : std::true_type {};
template <class E> struct tally_atomic<E, std::void_t<decltype(E::TALLY_ATOMIC)>> : std::bool_constant<E::TALLY_ATOMIC> {};

// Register-resident partial tally of one lane for the duration of a fused rollout.
struct LaneTally {
    double ret_sum, ret_sq, ret_min, ret_max, len_sq;
    int episodes, len_sum, viol, crit, shutdown, success;
    long long life;
    __device__ __forceinline__ void clear()
    {
        ret_sum = 0.0; ret_sq = 0.0; ret_min = __builtin_inf(); ret_max = -__builtin_inf(); len_sq = 0.0;
        episodes = 0; len_sum = 0; viol = 0; crit = 0; shutdown = 0; success = 0; life = 0;
    }
    __device__ __forceinline__ void episode(double ret, int step, uint32_t viol_ep, int ncrit)
    {
        const double len = (double)step;
        episodes += 1; ret_sum += ret; ret_sq += ret * ret;
        ret_min = fmin(ret_min, ret); ret_max = fmax(ret_max, ret);
        len_sum += step; len_sq += len * len;
        viol += (int)viol_ep; crit += ncrit; shutdown += (ncrit > 0) ? 1 : 0; success += (ret > 0.0) ? 1 : 0;
    }
    // merge into the lane's column of the global tally (same fp64 operation order per row as
    // flush_tally would have produced when at most one episode finished; sums of several
    // episodes are added as one partial -- integer rows exact, fp rows within 1 ulp of fp64)
    __device__ __forceinline__ void merge(double *T, uint32_t ld, int n_en) const
    {
        double v[NIG_T_ROWS];
#pragma unroll
        for (int r = 0; r < NIG_T_ROWS; ++r) v[r] = T[(size_t)r * ld];
        v[NIG_T_EPISODES] += (double)episodes;
        v[NIG_T_RET_SUM] += ret_sum;
        v[NIG_T_RET_SQ] += ret_sq;
        v[NIG_T_RET_MIN] = fmin(v[NIG_T_RET_MIN], ret_min);
        v[NIG_T_RET_MAX] = fmax(v[NIG_T_RET_MAX], ret_max);
        v[NIG_T_LEN_SUM] += (double)len_sum;
        v[NIG_T_LEN_SQ] += len_sq;
        v[NIG_T_VIOL] += (double)viol;
        v[NIG_T_CRIT] += (double)crit;
        v[NIG_T_SHUTDOWN] += (double)shutdown;
        v[NIG_T_SUCCESS] += (double)success;
        v[NIG_T_SATISFIED] += (double)((long long)n_en * len_sum - viol);   // every step of a finished episode has n_en constraints
        v[NIG_T_CONSTRAINTS] += (double)((long long)n_en * len_sum);
#pragma unroll
        for (int r = 0; r < NIG_T_ROWS; ++r) T[(size_t)r * ld] = v[r];
    }
};

// Wave-cooperative reset (envs with COOP_RESET): the lanes of `m` (ballot of the finishing lanes of this wave)
// get their initial states from work items (finishing lane, generator block) spread over all 64 lanes; an item
// writes the state rows its block feeds into column `owner` of the wave-private LDS image img[RESET_ROWS][64], the owners
// read their column back.  No block barrier.  DS operations of one wave execute in order, so the reads see the
// writes issued before them without a wait in between; the fences only pin the compiler's ordering.
// `lane_gi0` = global env index of the wave's lane 0, `t` = launch counter of the step that finished.
template <class Env>
__device__ __forceinline__ void coop_reset(unsigned long long m, bool mine, unsigned lane, float *img, unsigned char *lst,
                                           uint64_t lane_gi0, uint32_t t, uint32_t seed_lo, uint32_t seed_hi,
                                           const float4 *tab, float (&n)[Env::S])
{
    constexpr int ITEMS = Env::RESET_ITEMS;                   // work items per finishing lane (a power of two, or 6)
    static_assert((ITEMS & (ITEMS - 1)) == 0 || ITEMS == 6, "item index -> (lane, item): shift, or the divide-by-6 below");
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));   // finishing lanes below this one (v_mbcnt: no per-lane mask register)
    if (mine) lst[rank] = (unsigned char)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const int total = __popcll(m) * ITEMS;
    for (int i = (int)lane; i < total; i += 64) {
        // The item index is laundered: in the first pass it equals the lane index, a loop invariant of the ROLLOUT loop
        // around this call, and hipcc then hoists every per-block constant select of reset_item (standard deviations,
        // offsets, row numbers: ~30 registers for PowerGrid) out of that loop and keeps them alive across the whole step.
        int ii = i;
        asm volatile("" : "+v"(ii));
        unsigned li, item;                     // ii = li * ITEMS + item (ii < 64 * ITEMS)
        if constexpr (ITEMS == 6) { li = ((unsigned)ii * 171u) >> 10; item = (unsigned)ii - 6u * li; }     // exact for ii < 515
        else { li = (unsigned)ii / (unsigned)ITEMS; item = (unsigned)ii % (unsigned)ITEMS; }
        const unsigned owner = lst[li];
        Env::reset_item(make_key(lane_gi0 + owner, t, seed_lo, seed_hi, tab), item, img, owner);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (mine) Env::reset_readback(img, lane, n);
}

// One launch = IndustrialEnv.step for every lane.
//
// Memory shape: every row pointer is block-uniform (SGPR base) and the lane adds a 32-bit offset,
// so each access is "global_load_dword v, v_off, s[base]" over one contiguous 1 KiB row segment
// per block.  All loads (counter, state rows, action rows, injected noise) are issued up front in
// one batch -- a lane that turns out to be finished just discards them -- so the kernel has one
// memory round trip before the arithmetic, not two.
//
// Auto-reset: lanes that finish are COMPACTED across the 256-lane block through LDS and their
// initial states are produced by the first ceil(n/64) waves at full lane utilisation (with 18 % of
// PowerGrid lanes finishing per step every wave would otherwise run the whole reset path for a
// handful of active lanes).
// ACT64: the action rows are float64 (nig_step64; CR / PG / RA only: the envs whose NumPy arithmetic then changes).
// BLK: threads per block.  256, or Env::STEP_BLOCK for big fast-mode batches (PowerGrid: 512 -- the 12 KiB generator
// table is then shared by eight waves and two blocks = 16 waves fit a CU next to their reset images, so a
// 262 144-lane batch is resident in ONE round instead of 1.33: 30 -> 24 us per step; small batches keep 256 to
// spread over all CUs).
// HELP (auto-reset handles of COOP_RESET envs in fast mode, batches that leave one wave per SIMD: launch_step): the block
// is launched with 2 BLK threads.  Threads BLK .. 2 BLK - 1 are HELPER waves: helper h draws and builds the initial state
// lane h would restart from if it finished in this step -- it depends on the lane's generator key only, not on the state
// -- into LDS while the lane's own wave is still waiting for its loads and stepping; one block barrier later a finishing
// lane just picks its row up.  At 65 536 lanes every launch has some wave with a finishing lane, so the launch always
// paid the reset path behind its step (ballot, work list, two to eight generator blocks, their table look-ups -- a
// third dependent memory round trip -- and the read-back: 1.6 of ChemicalReactor's 4.9 us per replayed launch,
// profiles/r03/step_api_probe.txt); now that path runs beside the step instead of behind it, on issue slots the lone
// wave of a SIMD leaves empty.  Same draw_init + init as reset_kernel and as the cooperative reset: same values.
template <class Env, bool PARITY, bool ACT64 = false, int BLK = 256, bool HELP = false>
__global__ void __launch_bounds__(HELP ? 2 * BLK : BLK, (ACT64 || HELP ? 2 : (BLK / 256) * Env::STEP_WAVES)) step_kernel(const StepArgs p)
{
    constexpr int BLOCK = BLK;             // shadows the file-wide constant inside this kernel: LANES per block
    constexpr int S = Env::S, A = Env::A, KS = Env::KS, KR = Env::KR;
    constexpr int KSN = KS > 0 ? KS : 1;
    constexpr int NWAVE = BLOCK / 64;
    constexpr bool COOP = Env::COOP_RESET && !PARITY;          // wave-cooperative auto-reset (fast mode): coop_reset above
    static_assert(!HELP || (COOP && !ACT64), "helper waves: fast-mode float32 steps of envs with a cooperative reset");
    using act_t = std::conditional_t<ACT64, double, float>;
    __shared__ unsigned short s_list[COOP ? 1 : BLOCK];
    __shared__ int s_cnt[COOP ? 1 : NWAVE];
    __shared__ float s_img[COOP && !HELP ? NWAVE * Env::RESET_ROWS * 64 : 1];
    __shared__ unsigned char s_wlist[COOP && !HELP ? BLOCK : 1];
    __shared__ float s_new[HELP ? S * BLOCK : 1];              // [S][BLOCK]: the initial states the helpers prepared
    constexpr bool HELP_TALLY = HELP && !tally_atomic<Env>::value;
    __shared__ double s_fin_ret[HELP_TALLY ? BLOCK : 1];       // what a finished episode leaves for the tally when the helpers flush it
    __shared__ uint32_t s_fin_viol[HELP_TALLY ? BLOCK : 1], s_fin_word[HELP_TALLY ? BLOCK : 1];   // word: step | ncrit << 20 | finished << 31

    const bool helper = HELP && threadIdx.x >= (unsigned)BLOCK;
    const unsigned tid = HELP ? (threadIdx.x & (unsigned)(BLOCK - 1)) : threadIdx.x;    // lane of the block (helper: the lane it works for)
    const uint32_t base = blockIdx.x * BLOCK;                  // block-uniform
    const bool in_range = base + tid < p.B;
    const uint32_t t_now = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off;     // (the pointer chase costs 0.03-0.05 us of the launch: measured with a build that skipped it)
    // The generator's table: staged in LDS when a lane looks up many normals per launch; an env with a couple of draws
    // per step reads its entries straight from the 12 KiB global table (L2-resident) -- staging 12 KiB per block plus a
    // block barrier costs more than two or three 16-byte loads per lane.  With helper waves THEY stage it, first thing.
    constexpr bool STAGE_TABLE = PARITY ? false : (KS > 4 || (HELP && KS > 0));
    __shared__ float4 s_probit_[STAGE_TABLE ? 768 : 1];
    const float4 *const s_probit = STAGE_TABLE ? s_probit_ : NIG_PROBIT;
    if constexpr (HELP) {
        if (helper) {
            __builtin_amdgcn_s_setprio(0);
            if constexpr (STAGE_TABLE) {
                for (int i_ = (int)tid; i_ < 768; i_ += BLOCK) s_probit_[i_] = NIG_PROBIT[i_];
                __syncthreads();
            }
#ifdef NIG_DIAG_HELP_SKIP              // (diagnostic builds only, wrong restart states: what is left when the helpers cost nothing?)
            if (false) {
#else
            if (in_range) {
#endif
                double rn[KR > 0 ? KR : 1];
                Env::draw_init(make_key(p.env0 + (uint64_t)(base + tid), t_now, p.seed_lo, p.seed_hi, s_probit), rn);
                float r0[S];
                Env::init(rn, r0);
#pragma unroll
                for (int k = 0; k < S; ++k) s_new[k * BLOCK + tid] = r0[k];
            }
            __syncthreads();
            // ... and, for an env whose tally is not kept with atomics (PowerGrid), takes the episode tally of the lanes that
            // finished off their waves after the barrier: 13 loads, a wait and 13 stores the stepping wave no longer sits
            // through (8.84 -> 8.57 us per launch).  No-return atomics stay with the stepping wave, which issues them earlier
            // than a helper could (ChemicalReactor: 4.05 us there, 4.23 us from the helper).
            if constexpr (HELP_TALLY) {
                if (p.tally != nullptr && in_range) {
                    const uint32_t w = s_fin_word[tid];
                    if (w >> 31)
                        flush_tally(p.tally + base + tid, p.ld, s_fin_ret[tid], (int)(w & 0xFFFFFu), s_fin_viol[tid], (int)((w >> 20) & 0x7FFu), p.n_en);
                }
            }
            return;
        }
        __builtin_amdgcn_s_setprio(2);
    }

    // ---- one batch of loads -------------------------------------------------------------
    const uint32_t *ctr_row = p.ctr + base;
    const float *st_row = p.state + base;
    const act_t *act_row;
    if constexpr (ACT64) act_row = p.actions64 + base; else act_row = p.actions + base;
    uint32_t ctr = NIG_CTR_DONE;
    float s[S], n[S];
    act_t a[A];
    using nz_t = std::conditional_t<PARITY, double, typename Env::fast_noise_t>;   // injected draws are fp64
    nz_t nz[KSN];
    double ret_prev = 0.0;                 // the running episode return (utils.py:99), read with the batch: a load behind the
                                           // step's arithmetic would be one more memory round trip on the launch's critical path
    // No branch around the loads: a lane beyond the batch reads lane 0's rows of its block (which exist) and is masked out
    // below.  Loads inside a conditional block make the waitcnt pass wait for them where the block ends -- before the
    // generator's arithmetic, which needs none of them -- instead of at their first use.
    const unsigned li = in_range ? tid : 0u;
    {
        const uint32_t c_ld = ctr_row[li];
        ctr = in_range ? c_ld : NIG_CTR_DONE;
        // (the running return without a branch as well.  A handle without the tally has no return row; the always-present
        // lifetime-violation row stands in, its value unused)
        ret_prev = (p.tally ? p.ep_ret + base : reinterpret_cast<const double *>(p.life_viol + base))[li];
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = (st_row + k * p.ld_state)[li];
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = (act_row + k * p.ld_act)[li];
        if constexpr (PARITY && KS > 0) {
            const double *nz_row = p.step_noise + base;
#pragma unroll
            for (int k = 0; k < KS; ++k) nz[k] = (nz_row + k * p.ld_noise)[li];
        }
    }
    // the table is staged only now: the state / action loads above are already in flight
    if constexpr (STAGE_TABLE) {
        if constexpr (!HELP) {
            for (int i_ = (int)threadIdx.x; i_ < 768; i_ += BLOCK) s_probit_[i_] = NIG_PROBIT[i_];
        }
        __syncthreads();
    }
    const bool active = in_range && !(ctr & NIG_CTR_DONE);     // base.py:159-160: finished lanes wait for reset

    const RngKey key = make_key(p.env0 + (uint64_t)(base + tid), t_now, p.seed_lo, p.seed_hi, s_probit);
    if constexpr (KS > 0) {
        if constexpr (!PARITY) Env::draw_step(key, nz);
    } else {
        nz[0] = (nz_t)0;
    }

    // ---- IndustrialEnv.step in registers --------------------------------------------------
    const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
    StepResult<Env, reward_of<Env, act_t>> res;
    step_core<Env>(s, a, nz, step_pre, p.max_steps, p.dt32, p.dt, p.cmask, n, res);

    const int step = step_pre + 1;
    const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;   // base.py:182
    const bool done = res.terminated || res.truncated;
    uint32_t fl = pack_flags<Env>(res, step);
    uint32_t nctr = (uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT);
    const bool autoreset = (p.hflags & NIG_F_AUTORESET) != 0;
    const bool need_reset = active && done && autoreset;

    // utils.py:99  episode_return += reward.  Computed for every lane, used by the tally's: a use inside the conditional
    // blocks below would let the compiler sink the LOAD of the running return down there, behind the step (one more round trip)
    double ret;
    if constexpr (Env::RET_F32 && !ACT64) ret = (double)((float)ret_prev + res.reward);   // float32 accumulation (CR, float32 rewards)
    else ret = ret_prev + (double)res.reward;
    asm volatile("" :: "v"(ret));                 // (a use the sinking pass cannot move the load past)
    if (active) {
        if (done) {
            // base.py:183 total_violations (never reset): the lane's own counter, added to with a no-return atomic -- a
            // load + add + store would put a dependent memory round trip behind the step in every launch that finishes a lane
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(p.life_viol + base) + tid, (unsigned long long)viol_ep,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.tally) {
                if constexpr (HELP_TALLY) { s_fin_ret[tid] = ret; s_fin_viol[tid] = viol_ep; }      // flushed by the lane's helper, after the barrier
                else if constexpr (tally_atomic<Env>::value) flush_tally_atomic(p.tally + base + tid, p.ld, ret, step, viol_ep, res.ncrit, p.n_en);
                else flush_tally(p.tally + base + tid, p.ld, ret, step, viol_ep, res.ncrit, p.n_en);
                ret = 0.0;
            }
            if (p.final_obs) {
                float *fo = p.final_obs + base;
#pragma unroll
                for (int k = 0; k < S; ++k) (fo + k * p.ld_obs)[tid] = n[k];
            }
            if (autoreset) { nctr = 0u; fl |= NIG_FLAG_DID_RESET; }
            else nctr |= NIG_CTR_DONE;
        }
        if constexpr (!COOP) {
            if (!need_reset) {                    // a resetting lane's state is written by the compacted pass below
                float *so = p.state + base;
#pragma unroll
                for (int k = 0; k < S; ++k) (so + k * p.ld_state)[tid] = n[k];
                if (p.mirror) {
#pragma unroll
                    for (int k = 0; k < S; ++k) (p.mirror + base + k * p.ld_mirror)[tid] = n[k];
                }
            }
        }
        (p.ctr + base)[tid] = nctr;
        if (p.tally) (p.ep_ret + base)[tid] = ret;
        if (p.reward) (p.reward + base)[tid] = (float)res.reward;
        if (p.reward64) (p.reward64 + base)[tid] = (double)res.reward;
        if (p.flags) (p.flags + base)[tid] = fl;
    } else if (in_range) {
        if (p.flags) (p.flags + base)[tid] = NIG_FLAG_INACTIVE | ((ctr & NIG_CTR_STEP_MASK) << NIG_FLAG_STEP_SHIFT);
        if (p.reward) (p.reward + base)[tid] = 0.0f;
        if (p.reward64) (p.reward64 + base)[tid] = 0.0;
        if (p.mirror) {                           // a frozen lane: its state as it stands
#pragma unroll
            for (int k = 0; k < S; ++k) (p.mirror + base + k * p.ld_mirror)[tid] = s[k];
        }
    }

    if constexpr (COOP) {
        if constexpr (HELP) {
            if constexpr (HELP_TALLY) {
                if (in_range) s_fin_word[tid] = (uint32_t)step | ((uint32_t)res.ncrit << 20) | ((active && done) ? 0x80000000u : 0u);
            }
            __syncthreads();                      // the helpers' rows are in LDS
            if (need_reset) {
#pragma unroll
                for (int k = 0; k < S; ++k) n[k] = s_new[k * BLOCK + tid];
            }
        } else {
        // every wave renews its own finishing lanes (all 64 lanes work, whatever their own state), then stores
        const unsigned long long m = __ballot(need_reset);
        if (m != 0ull)
            coop_reset<Env>(m, need_reset, tid & 63u, s_img + (tid >> 6) * (Env::RESET_ROWS * 64), s_wlist + (tid >> 6) * 64,
                            p.env0 + (uint64_t)(base + (tid & ~63u)), t_now, p.seed_lo, p.seed_hi, s_probit, n);
        }
        if (active) {
            float *so = p.state + base;
#pragma unroll
            for (int k = 0; k < S; ++k) (so + k * p.ld_state)[tid] = n[k];
            if (p.mirror) {
#pragma unroll
                for (int k = 0; k < S; ++k) (p.mirror + base + k * p.ld_mirror)[tid] = n[k];
            }
        }
        return;
    }
    // ---- compacted auto-reset: IndustrialEnv.reset (base.py:133-155) for the finished lanes ----
    if (!autoreset) return;                       // block-uniform
    const unsigned wave = tid >> 6, lane = tid & 63u;
    const unsigned long long m = __ballot(need_reset);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    if (need_reset) s_list[wave * 64 + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tid;
    __syncthreads();
    int cnt[NWAVE], total = 0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) { cnt[w] = s_cnt[w]; total += cnt[w]; }
    for (int j = (int)tid; j < total; j += BLOCK) {
        int w = 0, r = j;
#pragma unroll
        for (int q = 0; q < NWAVE - 1; ++q) { const bool nxt = (w == q) && (r >= cnt[q]); r = nxt ? r - cnt[q] : r; w = nxt ? q + 1 : w; }
        const unsigned tl = s_list[w * 64 + r];   // block-local index of the lane being reset
        double rn[KR > 0 ? KR : 1];
        if constexpr (PARITY) {
            const double *rn_row = p.reset_noise + base;
#pragma unroll
            for (int k = 0; k < KR; ++k) rn[k] = (rn_row + k * p.ld_noise)[tl];
        } else {
            Env::draw_init(make_key(p.env0 + (uint64_t)(base + tl), t_now, p.seed_lo, p.seed_hi, s_probit), rn);
        }
        float r0[S];
        Env::init(rn, r0);
        float *so = p.state + base;
#pragma unroll
        for (int k = 0; k < S; ++k) (so + k * p.ld_state)[tl] = r0[k];
        if (p.mirror) {
#pragma unroll
            for (int k = 0; k < S; ++k) (p.mirror + base + k * p.ld_mirror)[tl] = r0[k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused multi-step rollout: n_steps consecutive IndustrialEnv.step calls per lane in ONE launch.
// State, counter word and running return live in registers for the whole launch; per step a
// lane reads only its action (ring slot k % ring_len) and writes only what the caller asked
// for (reward / flag word / observation of that step).  Lanes are independent, so there is no
// barrier between steps: waves drift apart and the divergent reset path costs its average,
// not its maximum.  The arithmetic, the generator keys (t = t_base + k + 1) and the
// bookkeeping are those of step_kernel: n_steps launches of step_kernel and one launch of
// this kernel leave bit-identical state, counters and tallies.
// This is the loop of the reference's own measurement / data-generation harnesses
// (performance_benchmark.py:106-133; chemical_reactor.py:364-405) with the policy replaced by
// a pre-filled action ring.
struct RolloutArgs {
    StepArgs s;                 // actions = ring base; reward/flags = per-step output bases (optional)
    int n_steps;                // steps [it0, n_steps) of the call are run by this launch
    int it0;
    int ring_len; uint32_t slot_stride;          // elements between ring slots
    uint32_t out_stride;                         // elements between per-step reward/flag rows (0: overwrite)
    float *obs_out; uint32_t ld_obs_out; uint64_t obs_step_stride;   // optional trajectory, [n_steps][S][ld] ...
    int obs_aos;                                                     // ... or row-major transitions [n_steps][B][S]
    uint32_t block0;            // first 256-lane block of this launch (the ragged last block is a launch of its own)
    // Injected draws (nig_rollout_noise; the NOISE kernel variants): s.step_noise = [n_steps][KS][ld_noise] float64, the
    // values the reference's np.random calls inside _dynamics returned for call step k (chemical_reactor.py:149,159,
    // power_grid.py:136-144), s.reset_noise = [n_steps][KR][ld_noise], the draws of _get_initial_state for a lane that
    // finishes its episode in call step k (base.py:133-155) -- nig_step's parity convention, one row set per step.
    uint64_t nz_step_stride, nz_reset_stride;   // elements between the row sets of consecutive steps
};

// OUT: 0 = no per-step outputs, 1 = reward + flag word, 2 = + observation rows [S][ld],
//      3 = + observation row-major [B][S].  Compile-time so that the number of stores per
// iteration is static and the wait for the prefetched action is a counted vmcnt(N), not a
// full drain of the iteration's stores.
// FULL: every lane of every block of the launch exists (the host launches the batch's whole 256-lane blocks with
// FULL = true and a ragged last block on its own with FULL = false).  Without lane predication the loop's loads
// and stores sit in one basic block, so the waits for the prefetched actions stay counted vmcnt(N) instead of
// the vmcnt(0) drains the waitcnt pass has to place behind exec-masked memory operations.
// LDS of one rollout block, carved from ONE buffer the kernel declares (the per-env kernels size it for their own
// env and output mode, the mixed-batch kernel for the largest of its envs): generator table, then the env's
// reset scratch, then the per-wave transpose image of the row-major trajectory.
template <class Env, int OUT, int BLK = 256>
struct RolloutLds {
    static constexpr int BLOCK = BLK;            // shadows the file-wide constant
    static constexpr int NWAVE = BLOCK / 64;
    static constexpr int OFF_PROBIT = 16 * PROBIT_BIAS;     // (nig_detmath.hpp probit_fetch: the piece number's bias rides in the DS offset field)
    // Per-wave scratch: the cooperative reset's image [RESET_ROWS][64] and, for the row-major trajectory, the transpose
    // image [16 S] float4 -- ONE region for both (a wave uses them at different points of its step, and its DS
    // operations execute in order).  Separate regions put RobotAssembly's row-major kernel at 66 KB per block, two
    // blocks per CU instead of the three its registers allow.
    static constexpr int IMG_BYTES = Env::COOP_RESET ? Env::RESET_ROWS * 64 * 4 : 0;
    static constexpr int TR_BYTES = OUT == 3 ? 16 * Env::S * 16 : 0;
    static constexpr bool SHARE_SCRATCH = Env::COOP_RESET && OUT == 3;
    static constexpr int WAVE_SCRATCH = SHARE_SCRATCH ? (IMG_BYTES > TR_BYTES ? IMG_BYTES : TR_BYTES) : IMG_BYTES;   // bytes per wave at OFF_IMG
    static constexpr int OFF_IMG = OFF_PROBIT + 768 * 16;                                          // [NWAVE][WAVE_SCRATCH]
    static constexpr int OFF_WLIST = OFF_IMG + NWAVE * WAVE_SCRATCH;                                // uchar [BLOCK]
    static constexpr int OFF_INIT = OFF_WLIST + (Env::COOP_RESET ? BLOCK : 0);                      // float [S][BLOCK]
    static constexpr int OFF_LIST = OFF_INIT + (Env::COMPACT_RESET ? Env::S * BLOCK * 4 : 0);       // ushort [BLOCK]
    static constexpr int OFF_CNT = OFF_LIST + (Env::COMPACT_RESET ? BLOCK * 2 : 0);                 // int [NWAVE]
    static constexpr int OFF_TR = SHARE_SCRATCH ? OFF_IMG : OFF_CNT + (Env::COMPACT_RESET ? 16 : 0);   // v4f [NWAVE][TR_STRIDE]
    static constexpr int TR_STRIDE = (SHARE_SCRATCH ? WAVE_SCRATCH : TR_BYTES) / 16;                // float4 per wave
    static constexpr int IMG_STRIDE = WAVE_SCRATCH / 4;                                             // floats per wave
    static constexpr int BYTES = SHARE_SCRATCH ? OFF_CNT + (Env::COMPACT_RESET ? 16 : 0) : OFF_TR + NWAVE * TR_BYTES;
};

// NOFREEZE (only with FULL): the host has checked that no lane of the handle can be frozen (auto-reset handle, no lane
// holding NIG_CTR_DONE), so the pre-step state is dead once the dynamics have read it -- with the run-time flag the
// "discard the speculative step" path keeps all S pre-step values alive next to the S new ones through the whole step
// (PowerGrid: 32 of the registers that capped it at two waves per SIMD).
// BLK: threads per block (256; 512 for the wide form of envs with a big per-wave LDS scratch: the 12 KiB generator
// table is then shared by eight waves and two blocks = four waves per SIMD fit a CU).
// NOISE: the reference's recorded draws are injected instead of the generator's (RolloutArgs::nz_*): the step's process
// noise is loaded as the float64 values the dynamics' parity branch takes, and a finishing lane restarts from
// Env::init(recorded draws) -- _get_initial_state itself, per lane, in place of the cooperative / compacted schemes
// (whose work items contain the generator).  Every other instruction of the step is the timed kernel's.
// RING (the paired form of an env with many draws per step, PowerGrid: rollout_pg_pair_kernel<.., REG>): the step's normals
// come from a PRODUCER wave through an LDS ring (nig_pg_lds.hpp pg_pair_producer: [generator block][lane] float4 slots of raw
// normals, two slots, counters at ring_sync) instead of this wave's own generator; the caller has staged the generator's table
// and passed the block barrier.  State, counters and tallies stay in REGISTERS: at the two waves per SIMD of that form the
// register file has room for them, and the step is then one dependent chain of arithmetic instead of a chain of LDS round trips.
template <class Env, int OUT, bool PAIRED, bool FULL, bool NOFREEZE = false, int BLK = 256, bool NOISE = false, bool RING = false>
__device__ __forceinline__ void rollout_body(const RolloutArgs &q, const uint32_t base, unsigned char *smem,
                                             const v4f *ring_slots = nullptr, lds_u32_t *ring_sync = nullptr)
{
    static_assert(!NOFREEZE || FULL, "NOFREEZE is a property of whole-block launches");
    static_assert(!NOISE || !PAIRED, "injected draws: nothing to share between the steps of a pair");
    static_assert(!RING || (!NOISE && !PAIRED && FULL && NOFREEZE && Env::KS > 4 && std::is_same<typename Env::fast_noise_t, float>::value),
                  "ring-fed form: whole blocks of an env with float32 step noise");
    static_assert(BLK == 256 || (Env::COOP_RESET && !Env::COMPACT_RESET), "wide blocks: no block barrier inside the loop");
    constexpr int BLOCK = BLK;                   // shadows the file-wide constant
    constexpr int S = Env::S, A = Env::A, KS = Env::KS, KR = Env::KR;
    constexpr int KSN = KS > 0 ? KS : 1;
    // Envs whose episodes are short (PowerGrid ~6 steps, RobotAssembly: most waves see a reset
    // every step) compact the finishing lanes of the 256-lane block through LDS each step and let
    // ONE wave produce all their initial states at full lane utilisation; the owners read them
    // back from LDS.  Costs two block barriers per step, saves running the whole reset path in
    // every wave for a few active lanes.  ChemicalReactor (0.3 % of lanes per step) keeps the
    // barrier-free divergent form.
    constexpr bool COMPACT = Env::COMPACT_RESET && !NOISE;
    // COOP (PowerGrid: ~11 lanes of every wave finish in every step): each WAVE produces the initial states of
    // its own finishing lanes cooperatively -- work item = (finishing lane, generator block) -> a few state rows,
    // spread over all 64 lanes through a wave-private LDS image.  No block barrier (waves keep drifting), the
    // generator runs at ~70 % lane utilisation instead of one wave carrying the whole block's resets while three
    // wait (53 % of the wave cycles of round 1's kernel were spent at those barriers).
    constexpr bool COOP = Env::COOP_RESET && !NOISE;
    static_assert(!(COMPACT && COOP), "one reset scheme per env");
    constexpr int NWAVE = BLOCK / 64;
    using Lds = RolloutLds<Env, OUT, BLK>;
    float4 *const s_probit = reinterpret_cast<float4 *>(smem + Lds::OFF_PROBIT);
    float *const s_img = reinterpret_cast<float *>(smem + Lds::OFF_IMG);         // per wave: [RESET_ROWS][64] initial states, column = owner lane
    unsigned char *const s_wlist = smem + Lds::OFF_WLIST;                        // per wave: lanes that finished, in lane order
    float *const s_init = reinterpret_cast<float *>(smem + Lds::OFF_INIT);
    unsigned short *const s_list = reinterpret_cast<unsigned short *>(smem + Lds::OFF_LIST);
    int *const s_cnt = reinterpret_cast<int *>(smem + Lds::OFF_CNT);
    v4f *const s_tr = reinterpret_cast<v4f *>(smem + Lds::OFF_TR);               // per wave: [16 S] transpose image of the row-major observation rows (64 x S floats)
    if constexpr (!RING) {                 // (ring-fed form: the kernel staged the table with all its waves)
        for (int i_ = (int)threadIdx.x; i_ < 768; i_ += BLOCK) s_probit[i_] = NIG_PROBIT[i_];
        __syncthreads();                   // every thread of the block passes here before any early exit
    }
    [[maybe_unused]] uint32_t ring_seen = 0u;
    const StepArgs &p = q.s;
    const unsigned tid = threadIdx.x;
    const bool in_range = FULL ? true : (base + tid < p.B);
    if constexpr (FULL) {
    } else if constexpr (COOP) {
        if (base + (tid & ~63u) >= p.B) return;   // whole wave out of range; a partial wave keeps all 64 lanes as workers
    } else if constexpr (!COMPACT) {
        if (!in_range) return;       // compacting blocks keep every thread for the barriers.  (From here on the
    }                                // compiler knows in_range: no exec masking around the loop's loads and stores.)
    const uint32_t t_base = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off;      // step k uses t_base + k + 1
    const uint64_t gi = p.env0 + (uint64_t)(base + tid);
    const bool autoreset = (p.hflags & NIG_F_AUTORESET) != 0;
    const bool tally = p.tally != nullptr;

    uint32_t ctr = in_range ? (p.ctr + base)[tid] : (uint32_t)NIG_CTR_DONE;   // out-of-range lanes idle as "frozen"
    float s[S], a[A], n[S];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = in_range ? (p.state + base + k * p.ld_state)[tid] : 0.0f;
    // running episode return in the precision the reference accumulates it in (float32 for ChemicalReactor:
    // the stored double is exactly that float), widened only when an episode ends
    using ret_t = std::conditional_t<Env::RET_F32, float, double>;
    ret_t ret = (tally && in_range) ? (ret_t)(p.ep_ret + base)[tid] : (ret_t)0;
    LaneTally lt;
    lt.clear();
    // Actions are prefetched TWO steps ahead into two ping-pong register sets (the loop is unrolled
    // by two so no register copy sits between load and use).  vmcnt retires in issue order, so the
    // wait for a prefetched action also waits for every store issued before it; at distance 2 those
    // are the stores of two steps ago, acknowledged long before (a distance-1 prefetch stalled ~20 %
    // of the wave's cycles on the previous step's store acknowledgements).
    //
    // PAIRED (envs that share one Philox block between the two steps of a pair of launch counters
    // 2k-1, 2k: ChemicalReactor; the launch must start on an odd counter, the host peels a misaligned
    // first step into a launch of the unpaired form): process noise is produced one step AHEAD, in the
    // shadow of the current step's stores -- the tail of a pair's second step runs the Philox rounds of
    // the next pair and the normal transform of its first step, the tail of the first step transforms
    // the two words kept for the second.  One block per two steps, LDS table latency off the critical
    // path.
    constexpr bool SHARE = PAIRED;
    static_assert(!PAIRED || (Env::SHARED_STEP_BLOCK && KS > 0 && KS <= 2), "a shared step block holds two steps");
    const float *ring = p.actions + base;
    // DEPTH = steps of slack between an action load and its use = ring of register sets = loop unroll.
    // The wait for a prefetched action is in-order with the stores issued before it; at the headline
    // size a step is ~1.2 us and a streaming store takes longer than two of them to be acknowledged.
    // Four steps for the envs whose step is short enough that four copies stay inside the I-cache.
    constexpr int DEPTH = PAIRED ? 4 : 2;
    float buf[DEPTH][A];
    using nz_t = std::conditional_t<NOISE, double, typename Env::fast_noise_t>;   // injected draws are fp64
    nz_t nzA[KSN], nzB[KSN];
    nzA[0] = (nz_t)0; nzB[0] = (nz_t)0;
    uint32_t kept0 = 0u, kept1 = 0u;          // words 2-3 of the current pair's block
    int slot = 0;
    // Wave-uniform running pointers instead of it * stride products: the per-step 64-bit scalar
    // multiplies and adds of the address arithmetic were ~40 of the step's ~80 SALU issue slots.
    const float *act_next = ring;              // ring slot of the step whose action is fetched next
    float *rew_row = p.reward ? p.reward + base + (size_t)q.it0 * q.out_stride : nullptr;
    uint32_t *fl_row = p.flags ? p.flags + base + (size_t)q.it0 * q.out_stride : nullptr;
    float *obs_row = nullptr;                  // this step's observation block / rows
    // (OUT == 3: the wave's first lane through readfirstlane -- the block pointer is wave-uniform, and only then does the
    // compiler keep it in scalar registers: the KiB stores within the instruction's 4 KiB immediate range are issued as
    // "scalar base + 32-bit lane offset", one 64-bit address computation less per step.  No measurable effect on the
    // launch time, profiles/r03/store_addr.txt and the A/B beside it.)
    if constexpr (OUT == 3)
        obs_row = q.obs_out + (size_t)q.it0 * q.obs_step_stride + (size_t)(base + __builtin_amdgcn_readfirstlane(tid & ~63u)) * S;
    if constexpr (OUT == 2) obs_row = q.obs_out + (size_t)q.it0 * q.obs_step_stride + base;
    // block-uniform: lanes can be frozen (finished and waiting for reset -- also on an auto-reset handle whose lanes
    // were never reset, left out by reset(mask) or marked done by set_state: base.py:159-160 -- or out of range)
    const bool may_freeze = NOFREEZE ? false : (!autoreset || (p.hflags & HF_MAY_HOLD_DONE) != 0 || (!FULL && base + BLOCK > p.B));

    auto one_step = [&](auto pos_tag, float (&abuf)[A], nz_t (&nz)[KSN], const int it) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = abuf[k];
        const bool frozen = may_freeze && (ctr & NIG_CTR_DONE) != 0;   // no auto-reset: base.py:159-160
        const RngKey key = make_key(gi, t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi, s_probit);
        const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
        StepResult<Env> res;
        if constexpr (NOISE) {
            if constexpr (KS > 0) {
                const double *nzr = p.step_noise + (size_t)it * q.nz_step_stride + base;
#pragma unroll
                for (int k = 0; k < KS; ++k) nz[k] = in_range ? (nzr + (size_t)k * p.ld_noise)[tid] : 0.0;
            }
        } else if constexpr (RING) {
            // the producer's slot of this step: raw normals, [generator block][lane]; scaled here exactly as Env::draw_step does
            const int itl = it - q.it0;
            if (ring_seen < (uint32_t)itl + 1u) ring_seen = split_wait(ring_sync + 0, (uint32_t)itl + 1u);
            constexpr int NB = (KS + 3) / 4;
            const v4f *slot = ring_slots + (itl & 1) * (NB * 64) + (tid & 63u);
            float z[4 * NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) { const v4f w = slot[64 * j]; z[4 * j] = w.x; z[4 * j + 1] = w.y; z[4 * j + 2] = w.z; z[4 * j + 3] = w.w; }
            split_post(ring_sync + 1, (uint32_t)itl + 1u, tid & 63u);     // (DS order: the reads above execute before this write)
            Env::scale_step_normals(z, nz);
        } else if constexpr (KS > 0 && !SHARE) draw_one<Env>(key, nz);
        step_core<Env>(s, a, nz, step_pre, p.max_steps, p.dt32, p.dt, p.cmask, n, res);
        const int step = step_pre + 1;
        const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;
        const bool done = (res.terminated || res.truncated) && !frozen;
        uint32_t fl = pack_flags<Env>(res, step);
        float rew = (float)res.reward;
        if (may_freeze) {                          // skipped wholesale (scalar branch) when no lane can be frozen
            if (frozen) {                          // untouched lane: discard the speculative step
                fl = NIG_FLAG_INACTIVE | ((ctr & NIG_CTR_STEP_MASK) << NIG_FLAG_STEP_SHIFT);
                rew = 0.0f;
#pragma unroll
                for (int k = 0; k < S; ++k) n[k] = s[k];
            }
        }
        if (!frozen) {
            ctr = (uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT);
            if (tally) {
                ret = ret + (ret_t)res.reward;
            }
        }
        // Next step's process noise, first half: (second step of a pair) the Philox rounds of the next
        // pair, then the index arithmetic and the LDS table reads of the two draws.  The cubic that
        // consumes them runs after this step's stores: the reads' latency is covered by the store traffic
        // instead of a wait.
        ProbitFetch pf[KSN];
        if constexpr (decltype(pos_tag)::value == 2) {         // next pair: counters t+1, t+2
            const u32x4 x = pair_block<Env>(make_key(gi, t_base + (uint32_t)it + 2u, p.seed_lo, p.seed_hi, s_probit));
            pair_fetch<Env>(x.x, x.y, s_probit, pf);
            kept0 = x.z; kept1 = x.w;
            __builtin_amdgcn_sched_barrier(0);     // keep it here: hipcc would sink it back to its consumer
        } else if constexpr (decltype(pos_tag)::value == 1) {  // this pair's second step
            pair_fetch<Env>(kept0, kept1, s_probit, pf);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (OUT == 3) {                  // stage this lane's row; read back transposed below
            if constexpr (S % 4 == 0) {
                v4f *tr = s_tr + (tid >> 6) * Lds::TR_STRIDE + (tid & 63u) * (S / 4);
#pragma unroll
                for (int k = 0; k < S / 4; ++k) { v4f v = {n[4 * k], n[4 * k + 1], n[4 * k + 2], n[4 * k + 3]}; tr[k] = v; }
            } else {
                float *tr = reinterpret_cast<float *>(s_tr + (tid >> 6) * Lds::TR_STRIDE) + (tid & 63u) * S;
#pragma unroll
                for (int k = 0; k < S; ++k) tr[k] = n[k];
            }
        }
        // Refill this buffer with the action of step it+DEPTH, issued BEFORE this step's stores: the
        // registers of `a` are dead by now (the load lands in place, no rotation of register sets),
        // and the in-order vmcnt wait at the top of step it+DEPTH then only needs the stores of step
        // it-1 and older to have been acknowledged -- DEPTH full steps of slack.
        {
#pragma unroll
            for (int k = 0; k < A; ++k) abuf[k] = in_range ? (act_next + k * p.ld_act)[tid] : 0.0f;
            slot = (slot + 1 == q.ring_len) ? 0 : slot + 1;
            act_next = (slot == 0) ? ring : act_next + q.slot_stride;
        }
        if constexpr (OUT == 3) {
            // row-major transitions [step][lane][S] (the D4RL "observations[N,S]" layout).  A lane's row is
            // 4*S contiguous bytes, but written lane by lane every store instruction would scatter 64
            // 16-byte pieces at a 4*S-byte stride (partial lines: -15 % against the [S][lane] layout, -45 %
            // with streaming stores).  The wave's 64 rows are one contiguous 256*S-byte block, so they go
            // through a wave-private LDS image and leave in lane-contiguous order: S/4 stores of one
            // contiguous KiB each.  (DS operations of one wave execute in order: the reads see the writes
            // issued above without a wait in between.)
            const unsigned lane = tid & 63u, wave_env0 = base + (tid & ~63u);
            const v4f *tr = s_tr + (tid >> 6) * Lds::TR_STRIDE;
            // The reads below are OTHER lanes' writes.  The compiler reasons per thread: a lane's own piece 48 l + 16 can
            // never be the address 16 l + 1024 k it reads, so without this fence it may sink that store out of the loop
            // (it did, in the injected-draw variant -- the only one whose loop holds no other fence).  Wavefront scope:
            // pins the compiler's order, emits no wait (the LDS pipeline executes a wave's operations in order).
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            v4f *oo = reinterpret_cast<v4f *>(obs_row);
            constexpr int NV = (16 * S + 63) / 64;  // float4 pieces per lane: the wave's block is 64*S floats = 16*S float4
            v4f v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = tr[(16 * S % 64 == 0 || lane + 64u * k < 16u * S) ? lane + 64u * k : 0u];
            if (FULL || wave_env0 + 64u <= p.B) {  // wave-uniform: the whole wave exists
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    if (16 * S % 64 == 0 || lane + 64u * k < 16u * S) stream_store(oo + lane + 64u * k, v[k]);
            } else if (in_range) {                 // the batch's last, partial wave (its other lanes may have exited):
                float *row = obs_row + (size_t)lane * S;       // every live lane writes its own row
                if constexpr (S % 4 == 0) {
#pragma unroll
                    for (int k = 0; k < S / 4; ++k) store16(row + 4 * k, n[4 * k], n[4 * k + 1], n[4 * k + 2], n[4 * k + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < S; ++k) row[k] = n[k];
                }
            }
        }
        if (in_range) {
        if constexpr (OUT == 2) {
#pragma unroll
            for (int k = 0; k < S; ++k) stream_store(obs_row + k * q.ld_obs_out + tid, n[k]);
        }
        if constexpr (OUT >= 1) {
            stream_store(rew_row + tid, rew);
            stream_store(fl_row + tid, fl | ((done && autoreset) ? NIG_FLAG_DID_RESET : 0u));
        }
        }   // in_range
        if constexpr (OUT >= 1) { rew_row += q.out_stride; fl_row += q.out_stride; }
        if constexpr (OUT >= 2) obs_row += q.obs_step_stride;
        if constexpr (decltype(pos_tag)::value == 2) {         // second half: the normals themselves
            __builtin_amdgcn_sched_barrier(0);
            pair_eval<Env>(pf, nzA);
        } else if constexpr (decltype(pos_tag)::value == 1) {
            __builtin_amdgcn_sched_barrier(0);
            pair_eval<Env>(pf, nzB);
        }
        if (done) {
            lt.life += (long long)viol_ep;
            if (tally) { lt.episode((double)ret, step, viol_ep, res.ncrit); ret = (ret_t)0; }
            if (!autoreset) ctr |= NIG_CTR_DONE;
        }
        if constexpr (COOP) {
            const unsigned long long m = __ballot(done && autoreset);
            if (m != 0ull) {                       // wave-uniform
                const unsigned lane = tid & 63u, wave = tid >> 6;
                coop_reset<Env>(m, done, lane, s_img + wave * Lds::IMG_STRIDE, s_wlist + wave * 64,
                                p.env0 + (uint64_t)(base + (tid & ~63u)), t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi,
                                s_probit, n);
                if (done) ctr = 0u;
            }
        } else if constexpr (!COMPACT) {
            if (done && autoreset) {               // divergent per-lane reset (base.py:133-155)
                double rn[KR > 0 ? KR : 1];
                if constexpr (NOISE) {             // _get_initial_state on the recorded draws of this step's row set
                    const double *rnr = p.reset_noise + (size_t)it * q.nz_reset_stride + base;
#pragma unroll
                    for (int k = 0; k < KR; ++k) rn[k] = (rnr + (size_t)k * p.ld_noise)[tid];
                } else {
                    Env::draw_init(key, rn);
                }
                Env::init(rn, n);
                ctr = 0u;
            }
        } else if (autoreset) {                    // block-uniform
            const unsigned wave = tid >> 6, lane = tid & 63u;
            const unsigned long long m = __ballot(done);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));   // finishing lanes below this one (v_mbcnt: no per-lane mask register)
            if (lane == 0) s_cnt[wave] = __popcll(m);
            if (done) s_list[wave * 64 + rank] = (unsigned short)tid;
            __syncthreads();
            int cnt[NWAVE], total = 0, mine = rank;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) { cnt[w] = s_cnt[w]; mine += ((unsigned)w < wave) ? cnt[w] : 0; total += cnt[w]; }
            // the worker role rotates over the block's waves so no SIMD carries it every step
            const unsigned widx = (tid + BLOCK - 64u * ((unsigned)it & (NWAVE - 1))) & (BLOCK - 1);
            for (int j = (int)widx; j < total; j += BLOCK) {
                int w = 0, r = j;
#pragma unroll
                for (int qq = 0; qq < NWAVE - 1; ++qq) { const bool nxt = (w == qq) && (r >= cnt[qq]); r = nxt ? r - cnt[qq] : r; w = nxt ? qq + 1 : w; }
                const unsigned tl = s_list[w * 64 + r];
                double rn[KR > 0 ? KR : 1];
                Env::draw_init(make_key(p.env0 + (uint64_t)(base + tl), t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi, s_probit), rn);
                float r0[S];
                Env::init(rn, r0);
#pragma unroll
                for (int k = 0; k < S; ++k) s_init[k * BLOCK + j] = r0[k];
            }
            __syncthreads();
            if (done) {
#pragma unroll
                for (int k = 0; k < S; ++k) n[k] = s_init[k * BLOCK + mine];
                ctr = 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = n[k];
    };

    int it = q.it0;
    slot = it % q.ring_len;
    if constexpr (SHARE) {
        const u32x4 x = pair_block<Env>(make_key(gi, t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi, s_probit));
        pair_noise<Env>(x.x, x.y, s_probit, nzA);
        kept0 = x.z; kept1 = x.w;
    }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {                                  // steps it .. it + DEPTH - 1
        const float *nx = ring + (size_t)slot * q.slot_stride;
#pragma unroll
        for (int k = 0; k < A; ++k) buf[j][k] = in_range ? (nx + k * p.ld_act)[tid] : 0.0f;
        slot = (slot + 1 == q.ring_len) ? 0 : slot + 1;
    }
    act_next = ring + (size_t)slot * q.slot_stride;                   // step it + DEPTH: the first refill
    // Drain the prologue loads HERE (vmcnt(0); expcnt/lgkmcnt untouched).  Otherwise hipcc's waitcnt
    // pass merges "prologue loads still in flight" into the loop header and every iteration inherits
    // waits sized for the first one.
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // no conditional inside the loop: a phi on the action registers would put register copies (and
    // with them the wait for the freshest loads) on the back edge
    using first = std::integral_constant<int, SHARE ? 1 : 0>;      // position in the pair (0: unpaired env)
    using second = std::integral_constant<int, SHARE ? 2 : 0>;
    for (; it + DEPTH <= q.n_steps; it += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; j += 2) {
            one_step(first{}, buf[j], nzA, it + j);
            one_step(second{}, buf[j + 1], nzB, it + j + 1);
        }
    }
    // tail: at most DEPTH - 1 steps (noise drawn past the last step is simply not used)
    static_assert(DEPTH == 2 || DEPTH == 4, "tail written out for these depths");
    if (it < q.n_steps) one_step(first{}, buf[0], nzA, it);
    if constexpr (DEPTH == 4) {
        if (it + 1 < q.n_steps) one_step(second{}, buf[1], nzB, it + 1);
        if (it + 2 < q.n_steps) one_step(first{}, buf[2], nzA, it + 2);
    }
    if (!in_range) return;
#pragma unroll
    for (int k = 0; k < S; ++k) (p.state + base + k * p.ld_state)[tid] = s[k];
    (p.ctr + base)[tid] = ctr;
    if (lt.life != 0) (p.life_viol + base)[tid] += lt.life;
    if (tally) {
        (p.ep_ret + base)[tid] = (double)ret;
        if (lt.episodes > 0) lt.merge(p.tally + base + tid, p.ld, p.n_en);
    }
}

template <class Env, int OUT, bool PAIRED, bool FULL, bool NOISE = false>
__global__ void __launch_bounds__(BLOCK, Env::ROLLOUT_WAVES) rollout_kernel(const RolloutArgs q)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[RolloutLds<Env, OUT>::BYTES];
    rollout_body<Env, OUT, PAIRED, FULL, false, 256, NOISE>(q, (blockIdx.x + q.block0) * BLOCK, smem);
}

// The wide form (envs that declare WIDE_ROLLOUT_BLOCK): whole blocks of BLK lanes of a handle on which no lane can be
// frozen.  q.block0 counts 256-lane blocks.
template <class E, class = void> struct wide_rollout : std::integral_constant<int, 0> {};
template <class E> struct wide_rollout<E, std::void_t<decltype(E::WIDE_ROLLOUT_BLOCK)>> : std::integral_constant<int, E::WIDE_ROLLOUT_BLOCK> {};

// (declared here for nig_pg_lds.hpp's closed-loop form; defined with the policy kernels below)
// The policy's random draws of one step: they depend on the lane's key only, not on the observation, so the
// cooperating-wave forms (nig_split_policy.hpp, nig_pg_lds.hpp) produce them ahead of the step that consumes them.
template <int A>
struct PolicyDraws { float z[A], h[A], ra[A], wmix; };
template <class Env> __device__ __forceinline__ void policy_draws(const nig_policy *__restrict__ P, const RngKey &key, PolicyDraws<Env::A> &d);
template <int A, class PV> __device__ __forceinline__ void policy_switches(const PV *__restrict__ P, bool &any_sigma, bool &any_half, bool &mix);
template <class Env, class PV> __device__ __forceinline__ void policy_affine(const PV *__restrict__ P, const float (&obs)[Env::S], float (&u)[Env::A]);
template <class Env, class PV> __device__ __forceinline__ void policy_finish(const PV *__restrict__ P, const PolicyDraws<Env::A> &d, float (&u)[Env::A]);
template <class Env, class PV> __device__ __forceinline__ void policy_finish_sw(const PV *__restrict__ P, bool any_sigma, bool any_half, bool mix, float lo, float hi,
                                                                                const PolicyDraws<Env::A> &d, float (&u)[Env::A]);

// Register copy of the policy fields a closed-loop form reads every step besides the feedback matrix (same field names as
// nig_policy: policy_finish / policy_switches take either).  Read in place from LDS, every field was an exposed ds_read round
// trip per step on the wave that evaluates the law.
template <int A>
struct PolicyHead {
    int32_t kind; uint32_t colmask;
    float b[A], sigma[A], half_range[A], setpoint[A], p_uniform, uniform_range, clip_lo, clip_hi, kp, ki, kd;
    __device__ __forceinline__ void load(const nig_policy &P)
    {
        kind = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)P.kind);
        colmask = __builtin_amdgcn_readfirstlane(P.colmask);
#pragma unroll
        for (int j = 0; j < A; ++j) { b[j] = P.b[j]; sigma[j] = P.sigma[j]; half_range[j] = P.half_range[j]; setpoint[j] = P.setpoint[j]; }
        p_uniform = P.p_uniform; uniform_range = P.uniform_range; clip_lo = P.clip_lo; clip_hi = P.clip_hi;
        kp = P.kp; ki = P.ki; kd = P.kd;
    }
};

// policy_affine for envs whose feedback matrix does not fit registers (PowerGrid 32 x 8, RobotAssembly 24 x 7): u_j = b_j +
// sum_k Wt[k][j] obs[k], ascending k, zero columns skipped -- the same operations on the same values as policy_affine -- with
// the matrix read from a dense 16-byte-aligned LDS copy [S rounded up to 8][8 actions] EIGHT COLUMNS AHEAD: sixteen
// ds_read_b128 in flight, one wait, then the columns' multiply-adds behind wave-uniform tests of the column mask.  (Read
// column by column inside those tests, every active column cost two exposed LDS round trips: 34 per step for PowerGrid's
// "expert" law -- +1.9 us per step, which made the paired closed loop no faster than the one-wave kernel.)
// (COLS: columns read ahead per batch -- eight where the wave has registers to spare, four on RobotAssembly's integrator)
template <class Env, int COLS = 8, class PV = PolicyHead<Env::A>>
__device__ __forceinline__ void policy_affine_dense(const PV &H, const v4f *__restrict__ wd, const float (&obs)[Env::S], float (&u)[Env::A])
{
    constexpr int S = Env::S, A = Env::A;
    static_assert(A <= 8, "dense copy holds eight actions per column");
#pragma unroll
    for (int j = 0; j < A; ++j) u[j] = H.b[j];
    const uint32_t cm = H.colmask;
#pragma unroll
    for (int q8 = 0; COLS * q8 < S; ++q8) {
        if ((cm >> (COLS * q8)) & ((1u << COLS) - 1u)) {      // wave-uniform: any column of this batch in use?
            v4f c[COLS][2];
#pragma unroll
            for (int k = 0; k < COLS; ++k) {
                if (COLS * q8 + k < S) { c[k][0] = wd[(COLS * q8 + k) * 2]; c[k][1] = wd[(COLS * q8 + k) * 2 + 1]; }
            }
#pragma unroll
            for (int k = 0; k < COLS; ++k) {
                if (COLS * q8 + k < S) {
                    if (cm & (1u << (COLS * q8 + k))) {   // wave-uniform: whole zero columns are skipped (as policy_affine)
                        const float o = obs[COLS * q8 + k];
                        const float w[8] = {c[k][0].x, c[k][0].y, c[k][0].z, c[k][0].w, c[k][1].x, c[k][1].y, c[k][1].z, c[k][1].w};
#pragma unroll
                        for (int j = 0; j < A; ++j) u[j] = u[j] + w[j] * o;
                    }
                }
            }
        }
    }
}

// fills the dense copy (every thread of the block calls it before the block barrier): wd[k][j] = Wt[k][j], j < 8
template <class Env>
__device__ __forceinline__ void policy_stage_dense(const nig_policy *gpol, float *wd, unsigned tid, unsigned nthreads)
{
    constexpr int SP = (Env::S + 7) / 8 * 8;
    for (unsigned i = tid; i < (unsigned)SP * 8u; i += nthreads)
        wd[i] = ((int)(i >> 3) < Env::S && (int)(i & 7u) < Env::A) ? gpol->Wt[i >> 3][i & 7u] : 0.0f;
}


}  // namespace nig
#include "nig_pg_lds.hpp"
namespace nig {
template <class Env, int OUT, int BLK, bool NOISE = false>
struct wide_body {                                // default: the register-resident body without freeze handling
    static constexpr int LDS_BYTES = RolloutLds<Env, OUT, BLK>::BYTES;
    __device__ static __forceinline__ void run(const RolloutArgs &q, uint32_t base, unsigned char *smem)
    {
        rollout_body<Env, OUT, false, true, true, BLK, NOISE>(q, base, smem);
    }
};
template <int OUT, int BLK, bool NOISE>
struct wide_body<PowerGrid, OUT, BLK, NOISE> {    // PowerGrid: state staged in LDS (nig_pg_lds.hpp)
    static constexpr int LDS_BYTES = PgLds<BLK>::BYTES;
    __device__ static __forceinline__ void run(const RolloutArgs &q, uint32_t base, unsigned char *smem)
    {
        pg_lds_rollout_body<OUT, BLK, false, NOISE>(q, base, smem);
    }
};

template <class Env, int OUT, int BLK, bool NOISE = false>
__global__ void __launch_bounds__(BLK, (BLK / 256) * Env::WIDE_ROLLOUT_WAVES) rollout_wide_kernel(const RolloutArgs q)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[wide_body<Env, OUT, BLK, NOISE>::LDS_BYTES];
    wide_body<Env, OUT, BLK, NOISE>::run(q, q.block0 * 256u + blockIdx.x * BLK, smem);
}

}  // namespace nig
#include "nig_split.hpp"
namespace nig {
// Mixed-batch launch (nig_mixed.hip): per-segment rollout arguments + the block -> segment table, in launch order.
constexpr int MIXED_MAX_SEG = NIG_MIXED_MAX_SEGMENTS;
struct MixedArgs {
    RolloutArgs seg[MIXED_MAX_SEG];
    uint32_t blk_end[MIXED_MAX_SEG];     // cumulative block count up to and including segment k
    int env[MIXED_MAX_SEG];
    int n_seg;
};
static_assert(sizeof(MixedArgs) <= 4000, "kernel argument segment is 4 KiB");

// ------------------------------------------------------------------------------------------
// Closed-loop fused rollout: action = on-device policy(observation) -> IndustrialEnv.step, n steps
// per launch, state / counters / tallies / PID memory in registers.  No loads inside the loop
// (the policy struct is staged in LDS), so the optional outputs can stay
// run-time switches.  Spec of the policy arithmetic: include/nig.h "nig-policy-v1".
struct PolicyArgs {
    StepArgs s;
    const nig_policy *pol;      // device copy
    float *pid;                 // PID policies: per-lane controller memory [2*A][ld] (integral rows, then previous-error rows)
    int n_steps;
    uint32_t out_stride;
    float *obs_out; uint64_t obs_step_stride;                        // row-major [B][S] per step, pre-step obs
    float *act_out; uint32_t ld_act_out; uint64_t act_step_stride;   // [A][ld] per step
    uint32_t block0;            // first 256-lane block of this launch (whole blocks and a ragged last block are separate launches)
    int32_t pol_kind;           // host copy of pol->kind (NIG_POLICY_*): selects the kernel form, never read on the device
};


// Register copy of the policy fields an env of this size reads, for a wave that evaluates the feedback law on its
// critical path (the integrator of nig_split_policy.hpp): read in place from LDS, every observation column is one
// exposed ds_read round trip per step.  Same field names as nig_policy: policy_apply takes either.
template <class Env>
struct PolicyRegs {
    static constexpr int S = Env::S, A = Env::A;
    int32_t kind; uint32_t colmask;
    float Wt[S][A], b[A], sigma[A], half_range[A], setpoint[A];
    float p_uniform, uniform_range, clip_lo, clip_hi, kp, ki, kd;
    // (vector registers: as scalars they spill -- 36 weights + 19 other fields against ~100 SGPRs -- and every use of a
    // spilled one costs a v_readlane; the kernel's three waves per SIMD leave each 168 VGPRs)
    __device__ static float sreg(float v) { return v; }
    __device__ __forceinline__ void load(const nig_policy &P)
    {
        kind = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)P.kind);
        colmask = __builtin_amdgcn_readfirstlane(P.colmask);
#pragma unroll
        for (int k = 0; k < S; ++k)
#pragma unroll
            for (int j = 0; j < A; ++j) Wt[k][j] = sreg(P.Wt[k][j]);
#pragma unroll
        for (int j = 0; j < A; ++j) { b[j] = sreg(P.b[j]); sigma[j] = sreg(P.sigma[j]); half_range[j] = sreg(P.half_range[j]); setpoint[j] = sreg(P.setpoint[j]); }
        p_uniform = sreg(P.p_uniform); uniform_range = sreg(P.uniform_range); clip_lo = sreg(P.clip_lo); clip_hi = sreg(P.clip_hi);
        kp = sreg(P.kp); ki = sreg(P.ki); kd = sreg(P.kd);
    }
};

template <int A, class PV>
__device__ __forceinline__ void policy_switches(const PV *__restrict__ P, bool &any_sigma, bool &any_half, bool &mix)
{
    any_sigma = false; any_half = false;
#pragma unroll
    for (int j = 0; j < A; ++j) { any_sigma = any_sigma || (P->sigma[j] != 0.0f); any_half = any_half || (P->half_range[j] != 0.0f); }
    mix = P->p_uniform > 0.0f;
}

template <class Env>
__device__ __forceinline__ void policy_draws(const nig_policy *__restrict__ P, const RngKey &key, PolicyDraws<Env::A> &d)
{
    constexpr int A = Env::A;
    bool any_sigma, any_half, mix;
    policy_switches<A>(P, any_sigma, any_half, mix);
    if (any_sigma) gen_normals<A>(key, STREAM_POLICY + 1u, d.z);
    if (any_half) {
#pragma unroll
        for (int b4 = 0; 4 * b4 < A; ++b4) {
            const u32x4 x = key.block(STREAM_POLICY + 8u + (uint32_t)b4);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * b4 + i < A) d.h[4 * b4 + i] = 2.0f * u01f(w[i]) - 1.0f;
        }
    }
    if (mix) {
        d.wmix = u01f(key.block(STREAM_POLICY).x);
        const float r = P->uniform_range;
#pragma unroll
        for (int b4 = 0; 4 * b4 < A; ++b4) {
            const u32x4 x = key.block(STREAM_POLICY + 16u + (uint32_t)b4);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * b4 + i < A) d.ra[4 * b4 + i] = r * (2.0f * u01f(w[i]) - 1.0f);
        }
    }
}

// feedback law on the observation + the draws + the policy's own clip (include/nig.h "nig-policy-v1"), in its two halves:
// policy_affine = u_j = b_j + sum_k Wt[k][j] obs[k] (ascending k, zero columns skipped), policy_finish = exploration
// noise, epsilon-mix and the policy's clip.  policy_apply composes them (PID: its own law, then policy_finish).
template <class Env, class PV>
__device__ __forceinline__ void policy_affine(const PV *__restrict__ P, const float (&obs)[Env::S], float (&u)[Env::A])
{
    constexpr int S = Env::S, A = Env::A;
#pragma unroll
    for (int j = 0; j < A; ++j) u[j] = P->b[j];
    const uint32_t cm = P->colmask;
#pragma unroll
    for (int k = 0; k < S; ++k) {
        if (cm & (1u << k)) {                  // wave-uniform: whole zero columns are skipped
#pragma unroll
            for (int j = 0; j < A; ++j) u[j] = u[j] + P->Wt[k][j] * obs[k];
        }
    }
}

// (the switches and the clip bounds handed in: a caller that keeps them in registers across its loop spares the wave that
// evaluates the law ~16 LDS reads and a round trip per step)
template <class Env, class PV>
__device__ __forceinline__ void policy_finish_sw(const PV *__restrict__ P, bool any_sigma, bool any_half, bool mix, float lo, float hi,
                                                 const PolicyDraws<Env::A> &d, float (&u)[Env::A])
{
    constexpr int A = Env::A;
    if (any_sigma) {
#pragma unroll
        for (int j = 0; j < A; ++j) u[j] = u[j] + P->sigma[j] * d.z[j];
    }
    if (any_half) {
#pragma unroll
        for (int j = 0; j < A; ++j) u[j] = u[j] + P->half_range[j] * d.h[j];
    }
    if (mix) {
        const bool rnd = d.wmix < P->p_uniform;
#pragma unroll
        for (int j = 0; j < A; ++j) u[j] = rnd ? d.ra[j] : u[j];
    }
#pragma unroll
    for (int j = 0; j < A; ++j) {                  // np.clip == minimum(maximum(x, lo), hi)
        float x = u[j];
        x = (x < lo) ? lo : x;
        x = (x > hi) ? hi : x;
        u[j] = x;
    }
}

template <class Env, class PV>
__device__ __forceinline__ void policy_finish(const PV *__restrict__ P, const PolicyDraws<Env::A> &d, float (&u)[Env::A])
{
    bool any_sigma, any_half, mix;
    policy_switches<Env::A>(P, any_sigma, any_half, mix);
    policy_finish_sw<Env>(P, any_sigma, any_half, mix, P->clip_lo, P->clip_hi, d, u);
}

template <class Env, class PV>
__device__ __forceinline__ void policy_apply(const PV *__restrict__ P, const float (&obs)[Env::S],
                                             const PolicyDraws<Env::A> &d, float (&integ)[Env::A], float (&eprev)[Env::A],
                                             float (&u)[Env::A])
{
    constexpr int A = Env::A;
    if (P->kind == NIG_POLICY_PID) {               // baseline_agents.py:61-80
        const float kp = P->kp, ki = P->ki, kd = P->kd;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float e = P->setpoint[j] - obs[j];
            integ[j] = integ[j] + e;
            u[j] = (kp * e + ki * integ[j]) + kd * (e - eprev[j]);
            eprev[j] = e;
        }
    } else {
        policy_affine<Env>(P, obs, u);
    }
    policy_finish<Env>(P, d, u);
}

// The one-wave kernel's form of the same policy: draws interleaved with their use (shorter live ranges than
// policy_draws + policy_apply; the two forms are pinned against each other by tests/test_gpu_split.py).
template <class Env>
__device__ __forceinline__ void policy_action(const nig_policy *__restrict__ P, const float (&obs)[Env::S],
                                              const RngKey &key, float (&integ)[Env::A], float (&eprev)[Env::A],
                                              float (&u)[Env::A])
{
    constexpr int S = Env::S, A = Env::A;
    if (P->kind == NIG_POLICY_PID) {               // baseline_agents.py:61-80
        const float kp = P->kp, ki = P->ki, kd = P->kd;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float e = P->setpoint[j] - obs[j];
            integ[j] = integ[j] + e;
            u[j] = (kp * e + ki * integ[j]) + kd * (e - eprev[j]);
            eprev[j] = e;
        }
    } else {
#pragma unroll
        for (int j = 0; j < A; ++j) u[j] = P->b[j];
        const uint32_t cm = P->colmask;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            if (cm & (1u << k)) {                  // wave-uniform: whole zero columns are skipped
#pragma unroll
                for (int j = 0; j < A; ++j) u[j] = u[j] + P->Wt[k][j] * obs[k];
            }
        }
    }
    bool any_sigma = false, any_half = false;
#pragma unroll
    for (int j = 0; j < A; ++j) { any_sigma = any_sigma || (P->sigma[j] != 0.0f); any_half = any_half || (P->half_range[j] != 0.0f); }
    if (any_sigma) {
        float z[A];
        gen_normals<A>(key, STREAM_POLICY + 1u, z);
#pragma unroll
        for (int j = 0; j < A; ++j) u[j] = u[j] + P->sigma[j] * z[j];
    }
    if (any_half) {
#pragma unroll
        for (int b4 = 0; 4 * b4 < A; ++b4) {
            const u32x4 x = key.block(STREAM_POLICY + 8u + (uint32_t)b4);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * b4 + i < A) u[4 * b4 + i] = u[4 * b4 + i] + P->half_range[4 * b4 + i] * (2.0f * u01f(w[i]) - 1.0f);
        }
    }
    if (P->p_uniform > 0.0f) {
        const float wmix = u01f(key.block(STREAM_POLICY).x);
        const bool rnd = wmix < P->p_uniform;
        const float r = P->uniform_range;
#pragma unroll
        for (int b4 = 0; 4 * b4 < A; ++b4) {
            const u32x4 x = key.block(STREAM_POLICY + 16u + (uint32_t)b4);
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * b4 + i < A) { const float ra = r * (2.0f * u01f(w[i]) - 1.0f); u[4 * b4 + i] = rnd ? ra : u[4 * b4 + i]; }
        }
    }
    const float lo = P->clip_lo, hi = P->clip_hi;
#pragma unroll
    for (int j = 0; j < A; ++j) {                  // np.clip == minimum(maximum(x, lo), hi)
        float x = u[j];
        x = (x < lo) ? lo : x;
        x = (x > hi) ? hi : x;
        u[j] = x;
    }
}

template <class Env>
__global__ void __launch_bounds__(BLOCK) rollout_policy_kernel(const PolicyArgs q)
{
    constexpr int S = Env::S, A = Env::A, KS = Env::KS, KR = Env::KR;
    constexpr int KSN = KS > 0 ? KS : 1;
    // The policy struct is staged in LDS: read from global memory inside the loop, every field was a
    // vector load followed by a full vmcnt(0) (the loop's stores may alias it, so hipcc neither hoists
    // the loads nor uses the scalar cache) -- ~20 serialised L2 round trips per step, 60 % of the step.
    __shared__ nig_policy s_pol;
    __shared__ v4f s_tr[BLOCK / 64][16 * S];       // per-wave transpose of the row-major observation rows
    // envs with a cooperative reset (PowerGrid: ~11 finishing lanes per wave and step) renew them wave by wave as
    // the open-loop rollout does (coop_reset); the in-place form ran the whole reset path in every wave every step
    constexpr bool COOP = Env::COOP_RESET;
    __shared__ float s_img[COOP ? (BLOCK / 64) * Env::RESET_ROWS * 64 : 1];
    __shared__ unsigned char s_wlist[COOP ? BLOCK : 1];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(q.pol);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&s_pol);
        for (unsigned i = threadIdx.x; i < sizeof(nig_policy) / 4; i += BLOCK) dst[i] = src[i];
    }
    NIG_STAGE_PROBIT(s_probit);                    // (ends with the block barrier that also publishes s_pol)
    const nig_policy *pol = &s_pol;
    const StepArgs &p = q.s;
    const unsigned tid = threadIdx.x;
    const uint32_t base = (blockIdx.x + q.block0) * BLOCK;
    const bool in_range = base + tid < p.B;
    if constexpr (COOP) {
        if (base + (tid & ~63u) >= p.B) return;    // a partial wave keeps all 64 lanes: they are the reset's workers
    } else {
        if (!in_range) return;
    }
    const uint32_t t_base = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off;
    const uint64_t gi = p.env0 + (uint64_t)(base + tid);
    const bool autoreset = (p.hflags & NIG_F_AUTORESET) != 0;
    const bool tally = p.tally != nullptr;

    uint32_t ctr = in_range ? (p.ctr + base)[tid] : (uint32_t)NIG_CTR_DONE;     // out-of-range lanes idle as frozen
    float s[S], a[A], n[S], integ[A], eprev[A];
    typename Env::fast_noise_t nz[KSN];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = in_range ? (p.state + base + k * p.ld_state)[tid] : 0.0f;
    // PID memory lives in the handle (baseline_agents.py:55-80: integral and previous error are the agent's,
    // never reset): loaded here, stored at the end, so launches chain exactly
    const bool pid_mem = q.pid != nullptr && pol->kind == NIG_POLICY_PID && in_range;
#pragma unroll
    for (int j = 0; j < A; ++j) {
        integ[j] = pid_mem ? (q.pid + base + (size_t)j * p.ld)[tid] : 0.0f;
        eprev[j] = pid_mem ? (q.pid + base + (size_t)(A + j) * p.ld)[tid] : 0.0f;
    }
    double ret = (tally && in_range) ? (p.ep_ret + base)[tid] : 0.0;
    LaneTally lt;
    lt.clear();

    for (int it = 0; it < q.n_steps; ++it) {
        const uint32_t orow = (uint32_t)it * q.out_stride;
        bool need_reset = false;
        const bool live = !(ctr & NIG_CTR_DONE);
        if (!live) {                               // frozen lane: base.py:159-160
            if (in_range) {
                if (p.flags) (p.flags + base + orow)[tid] = NIG_FLAG_INACTIVE | ((ctr & NIG_CTR_STEP_MASK) << NIG_FLAG_STEP_SHIFT);
                if (p.reward) (p.reward + base + orow)[tid] = 0.0f;
            }
        } else {
        const RngKey key = make_key(gi, t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi, s_probit);
        policy_action<Env>(pol, s, key, integ, eprev, a);
        if (q.obs_out) {
            // Row-major observations.  When every lane of the wave is live (exists, not frozen) the 64 rows
            // leave through the wave-private LDS image as whole-line streaming stores, as in rollout_kernel;
            // a wave with frozen lanes (their rows stay untouched) or the partial last wave writes row by row.
            // (Only for batches that put several waves on a SIMD: at one wave per SIMD the kernel is
            // issue-bound and the extra LDS round trip costs 5 %, above that it is worth +22 %.)
            if (p.B > 2u * 65536u && __ballot(true) == ~0ull) {
                const unsigned lane = tid & 63u;
                v4f *tr = s_tr[tid >> 6];
                if constexpr (S % 4 == 0) {
#pragma unroll
                    for (int k = 0; k < S / 4; ++k) { v4f v = {s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]}; tr[lane * (S / 4) + k] = v; }
                } else {
                    float *trf = reinterpret_cast<float *>(tr) + lane * S;
#pragma unroll
                    for (int k = 0; k < S; ++k) trf[k] = s[k];
                }
                v4f *oo = reinterpret_cast<v4f *>(q.obs_out + (size_t)it * q.obs_step_stride + (size_t)(base + (tid & ~63u)) * S);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // other lanes' writes are read below: see rollout_body
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                constexpr int NV = (16 * S + 63) / 64;
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    if (16 * S % 64 == 0 || lane + 64u * k < 16u * S) stream_store(oo + lane + 64u * k, tr[lane + 64u * k]);
            } else {
                float *oo = q.obs_out + (size_t)it * q.obs_step_stride + (size_t)(base + tid) * S;
                if constexpr (S % 4 == 0) {
#pragma unroll
                    for (int k = 0; k < S / 4; ++k) store16(oo + 4 * k, s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
                } else {                           // rows that are not a multiple of 16 bytes: dword stores
#pragma unroll
                    for (int k = 0; k < S; ++k) oo[k] = s[k];
                }
            }
        }
        if (q.act_out) {
            float *ao = q.act_out + (size_t)it * q.act_step_stride + base;
#pragma unroll
            for (int j = 0; j < A; ++j) stream_store(ao + j * q.ld_act_out + tid, a[j]);
        }
        if constexpr (KS > 0) Env::draw_step(key, nz); else nz[0] = 0;
        const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
        StepResult<Env> res;
        step_core<Env>(s, a, nz, step_pre, p.max_steps, p.dt32, p.dt, p.cmask, n, res);
        const int step = step_pre + 1;
        const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;
        const bool done = res.terminated || res.truncated;
        uint32_t fl = pack_flags<Env>(res, step) | ((done && autoreset) ? NIG_FLAG_DID_RESET : 0u);
        ctr = (uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT);
        if (tally) {
            if constexpr (Env::RET_F32) ret = (double)((float)ret + res.reward);
            else ret = ret + (double)res.reward;
        }
        if (p.reward) stream_store(p.reward + base + orow + tid, (float)res.reward);
        if (p.flags) stream_store(p.flags + base + orow + tid, fl);
        if (done) {
            lt.life += (long long)viol_ep;
            if (tally) { lt.episode(ret, step, viol_ep, res.ncrit); ret = 0.0; }
            if (autoreset) {
                if constexpr (COOP) {
                    need_reset = true;
                } else {
                    double rn[KR > 0 ? KR : 1];
                    Env::draw_init(key, rn);
                    Env::init(rn, n);
                }
                ctr = 0u;
            } else {
                ctr |= NIG_CTR_DONE;
            }
        }
        }   // live
        if constexpr (COOP) {                      // every lane of the wave arrives here, whatever its own state
            const unsigned long long m = __ballot(need_reset);
            if (m != 0ull)
                coop_reset<Env>(m, need_reset, tid & 63u, s_img + (tid >> 6) * (Env::RESET_ROWS * 64), s_wlist + (tid >> 6) * 64,
                                p.env0 + (uint64_t)(base + (tid & ~63u)), t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi,
                                s_probit, n);
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < S; ++k) s[k] = n[k];
        }
    }
    if (!in_range) return;
#pragma unroll
    for (int k = 0; k < S; ++k) (p.state + base + k * p.ld_state)[tid] = s[k];
    (p.ctr + base)[tid] = ctr;
    if (lt.life != 0) (p.life_viol + base)[tid] += lt.life;
    if (tally) {
        (p.ep_ret + base)[tid] = ret;
        if (lt.episodes > 0) lt.merge(p.tally + base + tid, p.ld, p.n_en);
    }
    if (pid_mem) {
#pragma unroll
        for (int j = 0; j < A; ++j) {
            (q.pid + base + (size_t)j * p.ld)[tid] = integ[j];
            (q.pid + base + (size_t)(A + j) * p.ld)[tid] = eprev[j];
        }
    }
}

}  // namespace nig
#include "nig_split_policy.hpp"
namespace nig {

// ------------------------------------------------------------------------------------------
// Fused MLP actor + env step (the one contraction on this path, so the one place for MFMA).
//
// One wavefront = 32 env instances.  Everything is computed TRANSPOSED, h^T = W^T x^T, so that
//   * the A operand is the weight matrix (one float per lane, streamed from a pre-ordered
//     array: record r = 64 floats = one 256-byte coalesced load),
//   * the B operand has the env on the lane (l & 31) and the k index on the lane half (l >> 5),
//   * the 32x32 result tile has the env on the lane again and the hidden unit in the register,
// which makes an accumulator register of layer n directly usable as a B operand of layer n+1
// (register t of a tile holds hidden rows rho_h(t) = (t&3) + 8(t>>2) + 4h of that tile for lane
// half h; the weight stream is ordered to match).  No LDS, no transposes, no conversion.
// v_mfma_f32_32x32x2_f32 is bit-for-bit fma(a1,b1, fma(a0,b0, c)) (k0 = lane half 0, then k1), so
// the CPU oracle reproduces the actor exactly.  Biases ride along as one extra k-step per tile
// (A = bias on half 0 / 0 on half 1, B = 1 / 0).
// Both lane halves carry the full env state and run the env step redundantly (it is ~3 % of the
// MFMA time); lanes 32-63 never store.
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct MlpArgs {
    StepArgs s;
    const float *wstream;       // MFMA operand stream built by nig_set_mlp_policy
    int n_steps;
    uint32_t out_stride;
    float *obs_out; uint64_t obs_step_stride;
    float *act_out; uint32_t ld_act_out; uint64_t act_step_stride;
};

constexpr int MLP_H = 256, MLP_MT = MLP_H / 32;
// The operand stream is cut into 1 + MLP_MT CHUNKS of MLP_CHREC records (256 bytes each, padded): chunk 0 = layer 1
// (MLP_MT tiles of S/2 weight records + 1 bias record), chunk 1 + m2 = hidden tile m2 of layer 2 with its slice of
// the head (128 + 1 + 16 records; the last chunk also carries the head's bias record).  A chunk is what one fill of
// an LDS buffer holds: MLP_PIECES wave-instructions of 1 KiB (64 lanes x 16 bytes, LDS-DMA).
constexpr int MLP_PER = MLP_MT * 16 + 1 + 16;                 // 145 records per hidden tile
constexpr int MLP_PIECES = (MLP_PER + 1 + 3) / 4;             // 37 KiB pieces per chunk
constexpr int MLP_CHREC = MLP_PIECES * 4;                     // 148 records per chunk slot
constexpr int MLP_CHUNKS = 1 + MLP_MT;
constexpr int MLP_STREAM_FLOATS = MLP_CHUNKS * MLP_CHREC * 64;
typedef __attribute__((address_space(3))) void nig_lds_void;
typedef __attribute__((address_space(1))) const void nig_glb_void;

// Two blocks per CU (round 4): the double-buffered weight image is 74 KiB per block; with the generator's 12 KiB table beside it
// only ONE block fitted a CU's 160 KiB, i.e. one wave per SIMD, and every chunk barrier and ring refill was exposed MFMA idle
// time (duty cycle 0.78, profiles/r04/mlp_cr65536_sq.txt).  An env with a couple of draws per step (KS <= 4) reads its table
// entries from global memory (L2-resident, as step_kernel does) and the kernel is compiled for two waves per SIMD, so a second
// block's waves fill the first's bubbles.
// The head (layer 3) of an env with at most FOUR actions runs on v_mfma_f32_4x4x1_16B_f32 (round 5): sixteen 4 x 4 blocks of
// four lanes, K = 1 -- lane l multiplies ITS OWN h2 value (B) with the four head weights held by the four lanes of its block
// (A: lane 4 b + r holds W3[k][r]) into four accumulators, out[r] += W3[k][r] h2[k]: exactly a head of <= 4 rows, 8 cycles an
// instruction, where the 32 x 32 x 2 tile spent 64 cycles on 32 rows of which 29 were zero padding (11 % of the step's MFMA
// time).  Each lane half accumulates the hidden rows ITS accumulator registers hold (rho_h(t)), the two
// partial sums meet in one v_add_f32 across the halves: a different summation order from the 32 x 32 x 2 head (k0, k1
// interleaved), restated by the oracle (mlp_actor).
// Heads of FIVE TO SIXTEEN actions take the same route on v_mfma_f32_16x16x1_4B_f32: four 16 x 16 blocks of sixteen lanes, K = 1,
// 32 cycles.  Block b = lane >> 4 multiplies its lanes' own h2 values (B: hidden row rho_{b>>1}(t), envs 16 (b & 1) ..) with the
// sixteen head weights its lanes hold (A: lane 16 b + r holds W3[k_b][r]) into registers 4 b .. 4 b + 3 (row r of env column c in
// register 4 b + (r & 3) of lane 16 (r >> 2) + c): blocks 0 / 1 are lane half 0's fma chain for envs 0-15 / 16-31, blocks 2 / 3
// lane half 1's -- the SAME two chains and the same joining add as the 4 x 4 x 1 head, so the oracle has one head for every env.
// The 32 x 32 x 2 head (64 cycles for 32 rows, 24 of them padding with eight actions) is gone.
template <class Env> constexpr bool mlp_head4 = Env::A <= 4;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <class Env> constexpr bool mlp_two_blocks = true;       // (PowerGrid's 23 + 31 table reads per step / reset from L2 as well: they are noise beside 1 217 MFMAs)

template <class Env>
__global__ void __launch_bounds__(BLOCK, mlp_two_blocks<Env> ? 2 : 1) rollout_mlp_kernel(const MlpArgs q)
{
    constexpr int S = Env::S, A = Env::A, KS = Env::KS, KR = Env::KR;
    constexpr int KSN = KS > 0 ? KS : 1;
    static_assert(S % 2 == 0 && A <= 16, "MFMA actor needs an even state dim and at most 16 actions");
    // Weight records are shared by the four waves of the block through LDS: round 1 let every wave stream all
    // ~1 200 records of a step from L2 on its own (the same 256-byte lines requested by every wave of the chip at
    // about the same time: 96 TFLOP/s of 155).  Now the block fills a double-buffered LDS image chunk by chunk with
    // LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no registers), nine fills per step, each wave a
    // quarter of the pieces, the fill of chunk c+1 in flight while chunk c is consumed; an MFMA's A operand is one
    // ds_read_b32.  L2 traffic per block and step: 311 KB instead of 4 x 311 KB.
    __shared__ __attribute__((aligned(16))) float s_w[2][MLP_CHREC * 64];
    __shared__ float4 s_probit_[mlp_two_blocks<Env> ? 1 : 768];
    if constexpr (!mlp_two_blocks<Env>) {
        for (int i_ = (int)threadIdx.x; i_ < 768; i_ += BLOCK) s_probit_[i_] = NIG_PROBIT[i_];
        __syncthreads();
    }
    const float4 *const s_probit = mlp_two_blocks<Env> ? NIG_PROBIT : s_probit_;
    const StepArgs &p = q.s;
    const unsigned tid = threadIdx.x, lane = tid & 63u, half = lane >> 5, e = lane & 31u, wave = tid >> 6;
    const uint32_t lane0 = blockIdx.x * (BLOCK / 2) + (tid >> 6) * 32u;     // first env of this wave
    const uint32_t li = lane0 + e;
    const bool in_range = li < p.B;
    const bool writer = in_range && half == 0;
    const uint32_t t_base = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off;
    const uint64_t gi = p.env0 + (uint64_t)li;
    const bool autoreset = (p.hflags & NIG_F_AUTORESET) != 0;
    const bool tally = p.tally != nullptr;

    uint32_t ctr = in_range ? p.ctr[li] : (uint32_t)NIG_CTR_DONE;
    float s[S], a[A], n[S];
    typename Env::fast_noise_t nz[KSN];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = in_range ? (p.state + k * p.ld_state)[li] : 0.0f;
    double ret = (tally && in_range) ? p.ep_ret[li] : 0.0;
    LaneTally lt;
    lt.clear();
    // fill LDS buffer `buf` with chunk `c` of the operand stream: this wave's quarter of the KiB pieces
    auto fill = [&](int c, int buf, int pieces) __attribute__((always_inline)) {
        const float *src = q.wstream + (size_t)c * (MLP_CHREC * 64) + lane * 4u;
        for (int pc = (int)wave; pc < pieces; pc += BLOCK / 64)
            __builtin_amdgcn_global_load_lds((nig_glb_void *)(src + pc * 256), (nig_lds_void *)(&s_w[buf][pc * 256]), 16, 0, 0);
    };
    constexpr int R1 = S / 2 + 1;                            // records per layer-1 tile
    constexpr int PIECES0 = (MLP_MT * R1 + 3) / 4;           // pieces of chunk 0
    static_assert(MLP_MT * R1 <= MLP_CHREC, "layer 1 must fit one chunk");
    fill(0, 0, PIECES0);
    int gbuf = 0;                                            // buffer that holds (or receives) the chunk consumed next
    for (int it = 0; it < q.n_steps; ++it) {
        // ---------------- actor: 3 layers of f32 MFMA, whole wave (EXEC all ones) ----------------
        // chunk boundary: every wave's share of the fill has landed (the compiler drains vmcnt before the barrier)
        // and every wave is done with the buffer the next fill overwrites
        __syncthreads();
#if defined(NIG_DIAG_MLP_SKIP) && (NIG_DIAG_MLP_SKIP & 2)
        if (it == 0)
#endif
        fill(1, gbuf ^ 1, MLP_PIECES);
        f32x16 h1[MLP_MT];
        {
            const float *wb = &s_w[gbuf][lane];
#pragma unroll
            for (int m = 0; m < MLP_MT; ++m) {
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < S / 2; ++ks) {
                    const float b = half ? s[2 * ks + 1] : s[2 * ks];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[(m * R1 + ks) * 64], b, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[(m * R1 + R1 - 1) * 64], half ? 0.0f : 1.0f, acc, 0, 0, 0);   // + b1
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);                                   // ReLU
                h1[m] = acc;
            }
        }
        gbuf ^= 1;
        f32x16 out = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        [[maybe_unused]] f32x4 out4 = {0, 0, 0, 0};          // mlp_head4: this lane half's partial sums of the (<= 4) head rows
        constexpr int RING = 8;                             // LDS reads in flight ahead of their MFMA (~64 cycles apart)
        for (int m2 = 0; m2 < MLP_MT; ++m2) {               // a real loop: the body is 145 MFMAs of straight-line code
            __syncthreads();                                // chunk 1 + m2 is in s_w[gbuf]; s_w[gbuf ^ 1] is free
#if defined(NIG_DIAG_MLP_SKIP) && (NIG_DIAG_MLP_SKIP & 2)     // (diagnostic, WRONG results: no LDS-DMA traffic after the first step)
            if (it == 0) {
#endif
            if (m2 + 1 < MLP_MT) fill(2 + m2, gbuf ^ 1, MLP_PIECES);
            else if (it + 1 < q.n_steps) fill(0, gbuf ^ 1, PIECES0);   // layer 1 of the NEXT step (the weights do not change)
#if defined(NIG_DIAG_MLP_SKIP) && (NIG_DIAG_MLP_SKIP & 2)
            }
#endif
            const float *wb = &s_w[gbuf][lane];
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            float ring[RING];
#pragma unroll
            for (int j = 0; j < RING; ++j) ring[j] = wb[j * 64];
#pragma unroll
            for (int i = 0; i < MLP_PER; ++i) {
                const float aop = ring[i % RING];
                if (i + RING < MLP_PER + 1) ring[i % RING] = wb[(i + RING) * 64];       // (+1: the head's bias record of the last chunk)
                // pin the source order: hipcc's scheduler otherwise sinks every read to just before its use
                __builtin_amdgcn_sched_barrier(0);
                if (i < MLP_MT * 16) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, h1[i / 16][i % 16], acc, 0, 0, 0);
                } else if (i == MLP_MT * 16) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, half ? 0.0f : 1.0f, acc, 0, 0, 0);      // + b2
                } else if constexpr (mlp_head4<Env>) {  // this h2 tile is consumed at once by the head: 4 x 4 x 1, own value x the block's weights
                    out4 = __builtin_amdgcn_mfma_f32_4x4x1f32(aop, fmaxf(acc[i - MLP_MT * 16 - 1], 0.0f), out4, 0, 0, 0);
                } else {                                // 16 x 16 x 1, four blocks: own value x the block's sixteen weights
                    out = __builtin_amdgcn_mfma_f32_16x16x1f32(aop, fmaxf(acc[i - MLP_MT * 16 - 1], 0.0f), out, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (m2 + 1 == MLP_MT) {                          // record 145 of the last chunk: + b3
                if constexpr (mlp_head4<Env>) out4 = __builtin_amdgcn_mfma_f32_4x4x1f32(ring[MLP_PER % RING], half ? 0.0f : 1.0f, out4, 0, 0, 0);
                else out = __builtin_amdgcn_mfma_f32_16x16x1f32(ring[MLP_PER % RING], half ? 0.0f : 1.0f, out, 0, 0, 0);
            }
            gbuf ^= 1;
        }
        if constexpr (mlp_head4<Env>) {
            // action r = this half's partial sum + the other half's (lane l and l + 32 carry the same env)
#pragma unroll
            for (int r = 0; r < A; ++r) a[r] = det_tanhf(out4[r] + __shfl_xor(out4[r], 32));
        } else {
            // head row r of env column c (env 16 beta + c, beta = block parity): lane 16 (r >> 2) + c, registers 4 beta + (r & 3)
            // (lane half 0's chain) and 8 + 4 beta + (r & 3) (lane half 1's); hand every lane all A rows of its env
            const int beta = (int)(e >> 4);
#pragma unroll
            for (int r = 0; r < A; ++r) {
                const int src = 16 * (r >> 2) + (int)(e & 15u);
                const float v0 = __shfl(out[r & 3] + out[8 + (r & 3)], src);
                const float v1 = __shfl(out[4 + (r & 3)] + out[12 + (r & 3)], src);
                a[r] = det_tanhf(beta ? v1 : v0);
            }
        }

#if defined(NIG_DIAG_MLP_SKIP) && (NIG_DIAG_MLP_SKIP & 1)     // (diagnostic builds only, WRONG results: the actor without the env step)
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = __builtin_fmaf(1e-9f, a[k % A], s[k]);
        continue;
#endif
        // ---------------- IndustrialEnv.step (both lane halves, identical results) ----------------
        const uint32_t orow = (uint32_t)it * q.out_stride;
        const bool frozen = (ctr & NIG_CTR_DONE) != 0;
        if (writer && !frozen) {
            if (q.obs_out) {
                float *oo = q.obs_out + (size_t)it * q.obs_step_stride + (size_t)li * S;
                if constexpr (S % 4 == 0) {
#pragma unroll
                    for (int k = 0; k < S / 4; ++k) store16(oo + 4 * k, s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < S; ++k) oo[k] = s[k];
                }
            }
            if (q.act_out) {
                float *ao = q.act_out + (size_t)it * q.act_step_stride;
#pragma unroll
                for (int j = 0; j < A; ++j) (ao + j * q.ld_act_out)[li] = a[j];
            }
        }
        const RngKey key = make_key(gi, t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi, s_probit);
        if constexpr (KS > 0) Env::draw_step(key, nz); else nz[0] = 0;
        const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
        StepResult<Env> res;
        step_core<Env>(s, a, nz, step_pre, p.max_steps, p.dt32, p.dt, p.cmask, n, res);
        const int step = step_pre + 1;
        const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;
        const bool done = (res.terminated || res.truncated) && !frozen;
        uint32_t fl = pack_flags<Env>(res, step) | ((done && autoreset) ? NIG_FLAG_DID_RESET : 0u);
        float rew = (float)res.reward;
        if (frozen) {
            fl = NIG_FLAG_INACTIVE | ((ctr & NIG_CTR_STEP_MASK) << NIG_FLAG_STEP_SHIFT);
            rew = 0.0f;
#pragma unroll
            for (int k = 0; k < S; ++k) n[k] = s[k];
        } else {
            ctr = (uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT);
            if (tally) {
                if constexpr (Env::RET_F32) ret = (double)((float)ret + res.reward);
                else ret = ret + (double)res.reward;
            }
        }
        if (writer) {
            if (p.reward) (p.reward + orow)[li] = rew;
            if (p.flags) (p.flags + orow)[li] = fl;
        }
        if (done) {
            lt.life += (long long)viol_ep;
            if (tally) { lt.episode(ret, step, viol_ep, res.ncrit); ret = 0.0; }
            if (autoreset) {
                double rn[KR > 0 ? KR : 1];
                Env::draw_init(key, rn);
                Env::init(rn, n);
                ctr = 0u;
            } else {
                ctr |= NIG_CTR_DONE;
            }
        }
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = n[k];
    }
    if (!writer) return;
#pragma unroll
    for (int k = 0; k < S; ++k) (p.state + k * p.ld_state)[li] = s[k];
    p.ctr[li] = ctr;
    if (lt.life != 0) p.life_viol[li] += lt.life;
    if (tally) {
        p.ep_ret[li] = ret;
        if (lt.episodes > 0) lt.merge(p.tally + li, p.ld, p.n_en);
    }
}

struct ResetArgs {
    float *state; uint32_t *ctr; long long *life_viol; double *ep_ret;
    int64_t ld; int64_t B; int64_t ld_state;
    const uint8_t *mask; const double *noise; int64_t ld_noise;
    uint64_t env0; uint32_t seed_lo, seed_hi, t;
};

template <class Env, bool PARITY>
__global__ void __launch_bounds__(BLOCK) reset_kernel(const ResetArgs p)
{
    constexpr int S = Env::S, KR = Env::KR;
    NIG_STAGE_PROBIT(s_probit);
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= p.B) return;
    if (p.mask && !p.mask[i]) return;
    double rn[KR > 0 ? KR : 1];
    if constexpr (PARITY) {
#pragma unroll
        for (int k = 0; k < KR; ++k) rn[k] = p.noise[(int64_t)k * p.ld_noise + i];
    } else {
        Env::draw_init(make_key(p.env0 + (uint64_t)i, p.t, p.seed_lo, p.seed_hi, s_probit), rn);
    }
    float s[S];
    Env::init(rn, s);
#pragma unroll
    for (int k = 0; k < S; ++k) p.state[(int64_t)k * p.ld_state + i] = s[k];
    const uint32_t ctr = p.ctr[i];
    // violations of an abandoned (not finished) episode still belong to total_violations
    if (!(ctr & NIG_CTR_DONE)) p.life_viol[i] += (long long)(ctr >> NIG_CTR_VIOL_SHIFT);
    p.ctr[i] = 0u;                                // base.py:137-139
    if (p.ep_ret) p.ep_ret[i] = 0.0;
}

template <class Env>
__global__ void __launch_bounds__(BLOCK) fill_actions_kernel(float *act, int64_t ld_act, int64_t B, uint64_t env0,
                                                             uint32_t seed_lo, uint32_t seed_hi, uint32_t t)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    const RngKey key = make_key(env0 + (uint64_t)i, t, seed_lo, seed_hi);
    double u[Env::A];
    gen_uniforms<Env::A>(key, STREAM_ACTION, u);
#pragma unroll
    for (int k = 0; k < Env::A; ++k)      // uniform in the env's action Box: low + (high - low) * u
        act[(int64_t)k * ld_act + i] = (float)((double)Env::act_low(k) + ((double)Env::act_high(k) - (double)Env::act_low(k)) * u[k]);
}


}  // namespace nig

// =====================================================================================
// host side: per-environment launch table.  Each env_*.hip instantiates its kernels through
// NIG_DEFINE_ENV_LAUNCH; nig_api.hip reaches them through these function pointers only.
// =====================================================================================
namespace nig {

struct EnvLaunch {
    void (*step)(const StepArgs &, bool parity, unsigned grid, hipStream_t);
    void (*step64)(const StepArgs &, bool parity, unsigned grid, hipStream_t);   // float64 action rows; nullptr: the env takes float32
    void (*rollout)(int out_mode, const RolloutArgs &, uint32_t t0, unsigned grid, hipStream_t);
    void (*policy)(const PolicyArgs &, unsigned grid, hipStream_t);
    void (*mlp)(const MlpArgs &, unsigned grid, hipStream_t);      // nullptr: env shape not supported by the MFMA actor
    void (*reset)(const ResetArgs &, bool parity, unsigned grid, hipStream_t);
    void (*fill)(float *act, int64_t ld_act, int64_t B, uint64_t env0, uint32_t seed_lo, uint32_t seed_hi, uint32_t t,
                 unsigned grid, hipStream_t);
    // does `rollout` read a ROW-MAJOR action ring ([B][A] slots, RolloutArgs.s.ld_act == 0) natively for this request?
    bool (*rows_native)(int out_mode, const RolloutArgs &);
};

template <class Env>
static void launch_reset(const ResetArgs &a, bool parity, unsigned grid, hipStream_t st)
{
    if (parity) hipLaunchKernelGGL((reset_kernel<Env, true>), dim3(grid), dim3(BLOCK), 0, st, a);
    else hipLaunchKernelGGL((reset_kernel<Env, false>), dim3(grid), dim3(BLOCK), 0, st, a);
}

// the MFMA actor exists for even state dims and at most 16 actions (nig_set_mlp_policy refuses the others)
template <class Env>
static void launch_mlp(const MlpArgs &q, unsigned grid, hipStream_t st)
{
    if constexpr (Env::S % 2 == 0 && Env::A <= 16) hipLaunchKernelGGL((rollout_mlp_kernel<Env>), dim3(grid), dim3(BLOCK), 0, st, q);
}

template <class Env>
static void launch_step(const StepArgs &a, bool parity, unsigned grid, hipStream_t st)
{
    constexpr int FB = Env::STEP_BLOCK;
    if (parity) { hipLaunchKernelGGL((step_kernel<Env, true>), dim3(grid), dim3(BLOCK), 0, st, a); return; }
    if constexpr (Env::COOP_RESET) {
        // auto-reset handles whose batch leaves one wave per SIMD (up to nig_tune(NIG_TUNE_SPLIT_BLOCKS) 256-lane blocks, default
        // one per compute unit -- the knob of the three-wave rollout, the same regime): a helper wave per lane wave prepares
        // the restart states beside the step (step_kernel, HELP)
        if ((a.hflags & NIG_F_AUTORESET) != 0 && a.split_blocks != 0 && grid <= a.split_blocks) {
            hipLaunchKernelGGL((step_kernel<Env, false, false, BLOCK, true>), dim3(grid), dim3(2 * BLOCK), 0, st, a);
            return;
        }
    }
    if (FB != BLOCK && a.B > 768u * BLOCK)            // more 256-thread blocks than are resident at once (3 per CU)
        hipLaunchKernelGGL((step_kernel<Env, false, false, FB>), dim3((a.B + FB - 1) / FB), dim3(FB), 0, st, a);
    else hipLaunchKernelGGL((step_kernel<Env, false>), dim3(grid), dim3(BLOCK), 0, st, a);
}

template <class Env>
static void launch_step64(const StepArgs &a, bool parity, unsigned grid, hipStream_t st)
{
    if constexpr (Env::HAS_ACT64) {
        if (parity) hipLaunchKernelGGL((step_kernel<Env, true, true>), dim3(grid), dim3(BLOCK), 0, st, a);
        else hipLaunchKernelGGL((step_kernel<Env, false, true>), dim3(grid), dim3(BLOCK), 0, st, a);
    }
}

// NOISE (here and below): the injected-draw variants of nig_rollout_noise, instantiated for the row-major full-output
// mode only (out_mode 3, what the headline configuration runs) -- same form selection, same launch shapes.
template <class Env, bool PAIRED, bool FULL, bool NOISE = false>
static void launch_rollout_blocks(int out_mode, const RolloutArgs &q, unsigned grid, hipStream_t st)
{
    if constexpr (NOISE) {
        hipLaunchKernelGGL((rollout_kernel<Env, 3, false, FULL, true>), dim3(grid), dim3(BLOCK), 0, st, q);
        return;
    }
    switch (out_mode) {
    case 0: hipLaunchKernelGGL((rollout_kernel<Env, 0, PAIRED, FULL>), dim3(grid), dim3(BLOCK), 0, st, q); break;
    case 1: hipLaunchKernelGGL((rollout_kernel<Env, 1, PAIRED, FULL>), dim3(grid), dim3(BLOCK), 0, st, q); break;
    case 2: hipLaunchKernelGGL((rollout_kernel<Env, 2, PAIRED, FULL>), dim3(grid), dim3(BLOCK), 0, st, q); break;
    default: hipLaunchKernelGGL((rollout_kernel<Env, 3, PAIRED, FULL>), dim3(grid), dim3(BLOCK), 0, st, q); break;
    }
}

// the batch's whole 256-lane blocks in one launch without lane predication, a ragged last block in its own
template <class Env, bool PAIRED, bool NOISE = false>
static void launch_rollout_form(int out_mode, const RolloutArgs &q, unsigned /*grid*/, hipStream_t st)
{
    const unsigned n_full = q.s.B / BLOCK;
    RolloutArgs r = q;
    if constexpr ((PAIRED || Env::KS == 0 || NOISE) && split_rollout<Env>::value) {
        // Up to one 256-lane block per CU the batch leaves a single wave on every SIMD: producer / integrator /
        // recorder wave per 64 lanes instead (nig_split.hpp; one block per CU is resident).  Larger batches run that
        // form in ROUNDS of one block per CU, which beats the one-wave form (lanes filling the SIMDs) by 4-12 % when
        // the rounds come out even -- measured at 2, 3, 4, 8 and 16 rounds, profiles/r02/rounds_probe.txt -- and loses
        // when the last round is mostly empty (1.5 rounds: -8 %): used when the last round is at least 3/4 full.
        // Round 5: ... and only for launches that write an observation trajectory (out_mode >= 2).  A round takes the three-wave
        // pipeline's ~142 us per 250 steps whatever it writes, so with reward + flags or no outputs the rounds LOSE to lanes
        // filling the SIMDs -- 131 072 lanes 280 vs 214-231 us, 262 144 lanes 555 vs 384-409 us, 1 048 576 lanes 2.20 vs
        // 1.31-1.46 ms -- while with the trajectory (HBM-bound either way) they win by 6-15 % (335 vs 378 us, 680 vs 726 us,
        // 3.30 vs 3.72 ms; profiles/r05/cr_rounds_by_output_mode.txt).  Found by the mixed-launch floor table, whose
        // stand-alone ChemicalReactor column was slower than the same body inside the mixed kernel.
        const bool plain = (q.s.hflags & NIG_F_AUTORESET) != 0 && (q.s.hflags & HF_MAY_HOLD_DONE) == 0;
        const unsigned per_round = q.s.split_blocks;       // nig_tune(NIG_TUNE_SPLIT_BLOCKS), default: the device's compute units
        const unsigned last_round = per_round ? n_full % per_round : 0u;
        const bool even_rounds = per_round != 0 && (n_full <= per_round ||
                                                    (split_rounds<Env>::value && (out_mode >= 2 || NOISE) &&
                                                     (last_round == 0 || 4u * last_round >= 3u * per_round)));
        if (plain && n_full > 0 && even_rounds) {
            r.block0 = 0;
            launch_split_blocks<Env, BLOCK / 64, NOISE>(out_mode, r, n_full, st);
            if (q.s.B % BLOCK) { r.block0 = n_full; launch_rollout_blocks<Env, PAIRED, false, NOISE>(out_mode, r, 1u, st); }
            return;
        }
    }
    // (Round 5 tried the three-wave form for the LAST, partial residency round of a RobotAssembly batch -- 262 144 lanes = 768 blocks
    // one-wave + 256 three-wave, two launches -- on the idea that those blocks run one to a compute unit anyway: slower, full outputs
    // 1 857-1 877 vs 1 842-1 863 us, reward + flags 1 590 vs 1 464, none 1 559 vs 1 427 (profiles/r05/ra_tail_round_three_wave_ab.txt):
    // one launch lets the tail's blocks start as compute units free up, two launches drain the chip in between.  Not kept.)
    unsigned first = 0;                            // first 256-lane block the forms below still have to run
    if constexpr (!PAIRED && wide_rollout<Env>::value != 0) {
        // Envs with an LDS-resident rollout body (PowerGrid, nig_pg_lds.hpp), handles on which no lane can be frozen:
        // from nig_tune(NIG_TUNE_WIDE_MIN_BLOCKS) wide blocks up, the batch's whole 512-lane blocks in the wide form
        // (four waves per SIMD); a remaining whole 256-lane block, and every whole block of a smaller batch, in the same
        // body with 256-thread blocks (three blocks per CU by LDS: 262 144 lanes would need 1.33 rounds, which is why
        // big batches take the wide form; small ones spread over more CUs this way and still run ~7 % fewer instructions
        // than the register-resident kernel, without its spills: 65 536 lanes 927 -> 858 us, 98 304 lanes 1.29 -> 1.08 ms
        // per 250 steps, profiles/r03/pg_small.txt).  A knob value of 2^30 or more keeps everything on rollout_kernel.
        constexpr int WB = wide_rollout<Env>::value;
        const bool plain = (q.s.hflags & NIG_F_AUTORESET) != 0 && (q.s.hflags & HF_MAY_HOLD_DONE) == 0;
        if (plain && q.s.wide_min_blocks < (1u << 30)) {
            const unsigned n_wide = q.s.B / WB;
            if (n_wide > 0 && n_wide >= q.s.wide_min_blocks) {
                r.block0 = 0;
                if constexpr (NOISE) hipLaunchKernelGGL((rollout_wide_kernel<Env, 3, WB, true>), dim3(n_wide), dim3(WB), 0, st, r);
                else
                switch (out_mode) {
                case 0: hipLaunchKernelGGL((rollout_wide_kernel<Env, 0, WB>), dim3(n_wide), dim3(WB), 0, st, r); break;
                case 1: hipLaunchKernelGGL((rollout_wide_kernel<Env, 1, WB>), dim3(n_wide), dim3(WB), 0, st, r); break;
                case 2: hipLaunchKernelGGL((rollout_wide_kernel<Env, 2, WB>), dim3(n_wide), dim3(WB), 0, st, r); break;
                default: hipLaunchKernelGGL((rollout_wide_kernel<Env, 3, WB>), dim3(n_wide), dim3(WB), 0, st, r); break;
                }
                first = n_wide * (WB / BLOCK);
            }
            if constexpr (pair_rollout<Env>::value) {
                // a batch that would leave one wave on every SIMD (at most nig_tune(NIG_TUNE_SPLIT_BLOCKS) 256-lane blocks,
                // default one per compute unit): the paired form -- a producer wave draws the step's normals beside every
                // stepping wave (rollout_pg_pair_kernel, nig_pg_lds.hpp)
                if (first == 0 && n_full > 0 && q.s.split_blocks != 0 && n_full <= q.s.split_blocks) {
                    r.block0 = 0;
                    if constexpr (NOISE) hipLaunchKernelGGL((rollout_pg_pair_kernel<3, true>), dim3(n_full), dim3(512), 0, st, r);
                    else
                    switch (out_mode) {
#ifdef NIG_DIAG_PG_PAIR_LDS            // (diagnostic builds only: round 3's LDS-resident stepping waves, for same-box A/Bs)
#define NIG_PG_PAIR_REG false
#else
#define NIG_PG_PAIR_REG true
#endif
                    case 0: hipLaunchKernelGGL((rollout_pg_pair_kernel<0, false, NIG_PG_PAIR_REG>), dim3(n_full), dim3(512), 0, st, r); break;
                    case 1: hipLaunchKernelGGL((rollout_pg_pair_kernel<1, false, NIG_PG_PAIR_REG>), dim3(n_full), dim3(512), 0, st, r); break;
                    // (with an observation trajectory the LDS-resident stepping body stays: its state image IS the transposing
                    // image of the row-major rows; the register body pays an extra LDS round trip for them -- same box, 65 536
                    // lanes x 250 steps: reward + flags 665 -> 582 us, no outputs 643 -> 557 us with registers, but full outputs
                    // 687 -> 774 us: profiles/r04/pg_pair_reg_ab.txt)
                    case 2: hipLaunchKernelGGL((rollout_pg_pair_kernel<2, false, false>), dim3(n_full), dim3(512), 0, st, r); break;
                    default: hipLaunchKernelGGL((rollout_pg_pair_kernel<3, false, false>), dim3(n_full), dim3(512), 0, st, r); break;
                    }
                    first = n_full;
                }
            }
            if (n_full > first) {
                r.block0 = first;
                const unsigned nb = n_full - first;
                if constexpr (NOISE) hipLaunchKernelGGL((rollout_wide_kernel<Env, 3, BLOCK, true>), dim3(nb), dim3(BLOCK), 0, st, r);
                else
                switch (out_mode) {
                case 0: hipLaunchKernelGGL((rollout_wide_kernel<Env, 0, BLOCK>), dim3(nb), dim3(BLOCK), 0, st, r); break;
                case 1: hipLaunchKernelGGL((rollout_wide_kernel<Env, 1, BLOCK>), dim3(nb), dim3(BLOCK), 0, st, r); break;
                case 2: hipLaunchKernelGGL((rollout_wide_kernel<Env, 2, BLOCK>), dim3(nb), dim3(BLOCK), 0, st, r); break;
                default: hipLaunchKernelGGL((rollout_wide_kernel<Env, 3, BLOCK>), dim3(nb), dim3(BLOCK), 0, st, r); break;
                }
                first = n_full;
            }
        }
    }
    if (n_full > first) { r.block0 = first; launch_rollout_blocks<Env, PAIRED, true, NOISE>(out_mode, r, n_full - first, st); }
    if (q.s.B % BLOCK) { r.block0 = n_full; launch_rollout_blocks<Env, PAIRED, false, NOISE>(out_mode, r, 1u, st); }
}

// nig_rollout's row-major action ring (ld_act == 0): true when EVERY kernel launch_rollout_form<Env, false> starts for this
// request reads a lane's actions as contiguous bytes -- the LDS-resident PowerGrid body (nig_pg_lds.hpp: wide 512 / 256 and the
// paired form's LDS stepper).  Mirrors the form selection above, predicate by predicate: whole 256-lane blocks only (a ragged
// block runs rollout_kernel), an auto-reset handle without held lanes, the wide knob on, and not the paired regime's
// register-resident stepper (out_mode 0 / 1 below NIG_TUNE_SPLIT_BLOCKS blocks: rollout_body reads rows).
template <class Env>
static bool rollout_rows_native(int out_mode, const RolloutArgs &q)
{
    if constexpr (wide_rollout<Env>::value == 0 || Env::A != 8) return false;
    else {
        constexpr unsigned WB = (unsigned)wide_rollout<Env>::value;
        const bool plain = (q.s.hflags & NIG_F_AUTORESET) != 0 && (q.s.hflags & HF_MAY_HOLD_DONE) == 0;
        if (!plain || q.s.wide_min_blocks >= (1u << 30) || q.s.B % BLOCK != 0u || q.s.B == 0u) return false;
        const unsigned n_full = q.s.B / BLOCK, n_wide = q.s.B / WB;
        const bool wide = n_wide > 0 && n_wide >= q.s.wide_min_blocks;
        if constexpr (pair_rollout<Env>::value) {
            const bool paired = !wide && q.s.split_blocks != 0 && n_full <= q.s.split_blocks;
            if (paired && out_mode <= 1 && NIG_PG_PAIR_REG) return false;
        }
        return true;
    }
}

// t0 = launch counter of the call's first step (host-known: rollouts are never graph-captured)
// the envs the reference can record draws for (ChemicalReactor, PowerGrid, RobotAssembly): nig_rollout_noise
template <class Env> struct noise_rollout : std::bool_constant<(Env::ID <= 2)> {};

template <class Env>
static void launch_rollout_env(int out_mode, const RolloutArgs &q, uint32_t t0, unsigned grid, hipStream_t st)
{
    if (q.s.step_noise != nullptr || q.s.reset_noise != nullptr) {       // injected draws (nig_rollout_noise has validated the request)
        if constexpr (noise_rollout<Env>::value) launch_rollout_form<Env, false, true>(3, q, grid, st);
        return;
    }
    if constexpr (Env::SHARED_STEP_BLOCK) {
        RolloutArgs r = q;
        if ((t0 & 1u) == 0u) {                    // starts on the second step of a pair: peel it
            r.n_steps = 1;
            launch_rollout_form<Env, false>(out_mode, r, grid, st);
            if (q.n_steps == 1) return;
            r.n_steps = q.n_steps; r.it0 = 1;
        }
        launch_rollout_form<Env, true>(out_mode, r, grid, st);
    } else {
        launch_rollout_form<Env, false>(out_mode, q, grid, st);
    }
}

template <class Env>
static void launch_policy(const PolicyArgs &q, unsigned grid, hipStream_t st)
{
    if constexpr (split_rollout<Env>::value && (Env::SHARED_STEP_BLOCK || Env::KS == 0)) {
        // Producer / integrator / recorder wave per 64 lanes (nig_split_policy.hpp) for the batch's whole 256-lane blocks --
        // up to one block per CU, and (ChemicalReactor) larger batches in rounds under the rule of the open-loop rollout (the last
        // round at least 3/4 full) -- and the one-wave kernel for a ragged last block.  RobotAssembly (round 4: S = 24, the
        // form's BIG layout): a single round only, as in the open loop; the observations of the transition stream ride in its
        // P -> I slots since round 5.
        const bool plain = (q.s.hflags & NIG_F_AUTORESET) != 0 && (q.s.hflags & HF_MAY_HOLD_DONE) == 0;
        const unsigned n_full = q.s.B / BLOCK, per_round = q.s.split_blocks;
        const unsigned last_round = per_round ? n_full % per_round : 0u;
        constexpr bool big = SplitPolicyLds<Env, BLOCK / 64>::BIG;
        // (beyond one round only for calls that write the observation stream, as in the open loop, and only up to TWO rounds:
        // a closed-loop round takes ~1 us per step whatever it writes, lanes filling the SIMDs take 1.75 / 2.5 us per step at
        // 131 072 / 262 144 lanes without the stream and 1.97 / 3.25 with it -- two rounds 1.84, four rounds 3.72:
        // profiles/r05/policy_rounds_cr.txt)
        const bool even_rounds = per_round != 0 && (n_full <= per_round ||
                                                    (!big && split_rounds<Env>::value && q.obs_out != nullptr && n_full <= 2u * per_round &&
                                                     (last_round == 0 || 4u * last_round >= 3u * per_round)));
        if (plain && n_full > 0 && even_rounds) {
            PolicyArgs r = q;
            r.block0 = 0;
            hipLaunchKernelGGL((split_policy_kernel<Env, BLOCK / 64>), dim3(n_full), dim3(192 * (BLOCK / 64)), 0, st, r);
            if (q.s.B % BLOCK) { r.block0 = n_full; hipLaunchKernelGGL((rollout_policy_kernel<Env>), dim3(1), dim3(BLOCK), 0, st, r); }
            return;
        }
    }
    if constexpr (pair_rollout<Env>::value) {
        // PowerGrid, affine policies, batches of at most one 256-lane block per compute unit (the open loop's paired-form
        // regime, nig_tune(NIG_TUNE_SPLIT_BLOCKS)): stepping + producer wave per 64 lanes (rollout_pg_pair_policy_kernel);
        // a ragged last block on the one-wave kernel.  PID policies keep their memory in registers: one-wave kernel.
        const bool plain = (q.s.hflags & NIG_F_AUTORESET) != 0 && (q.s.hflags & HF_MAY_HOLD_DONE) == 0;
        const unsigned n_full = q.s.B / BLOCK;
        if (plain && q.pol_kind == NIG_POLICY_AFFINE && n_full > 0 && q.s.split_blocks != 0 && n_full <= q.s.split_blocks) {
            PolicyArgs r = q;
            r.block0 = 0;
            hipLaunchKernelGGL((rollout_pg_pair_policy_kernel<PolicyArgs>), dim3(n_full), dim3(512), 0, st, r);
            if (q.s.B % BLOCK) { r.block0 = n_full; hipLaunchKernelGGL((rollout_policy_kernel<Env>), dim3(1), dim3(BLOCK), 0, st, r); }
            return;
        }
    }
    hipLaunchKernelGGL((rollout_policy_kernel<Env>), dim3(grid), dim3(BLOCK), 0, st, q);
}

template <class Env>
static void launch_fill(float *act, int64_t ld_act, int64_t B, uint64_t env0, uint32_t seed_lo, uint32_t seed_hi, uint32_t t,
                        unsigned grid, hipStream_t st)
{
    hipLaunchKernelGGL((fill_actions_kernel<Env>), dim3(grid), dim3(BLOCK), 0, st, act, ld_act, B, env0, seed_lo, seed_hi, t);
}

template <class Env>
static const EnvLaunch *env_launch_table()
{
    static const EnvLaunch T = {launch_step<Env>, Env::HAS_ACT64 ? launch_step64<Env> : nullptr, launch_rollout_env<Env>, launch_policy<Env>,
                                (Env::S % 2 == 0 && Env::A <= 16) ? launch_mlp<Env> : nullptr,
                                launch_reset<Env>, launch_fill<Env>, rollout_rows_native<Env>};
    return &T;
}

}  // namespace nig

// nig_mixed.hip
void nig_launch_mixed_rollout(int out_mode, const nig::MixedArgs &m, unsigned grid, hipStream_t st);

#define NIG_DEFINE_ENV_LAUNCH(EnvType, fn_name) \
    const nig::EnvLaunch *fn_name() { return nig::env_launch_table<nig::EnvType>(); }
