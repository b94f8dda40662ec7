// nig_kernels.hip -- HIP kernels (gfx950) and C-ABI of libnig.so.
//
// One wavefront lane per environment instance.  State, actions, noise and outputs are
// structure-of-arrays ([row][lane], row pitch ld) so every global access of a wave is one
// fully coalesced 256-byte row segment.  The step kernel fuses the whole of
// IndustrialEnv.step (environments/base.py:157-213): clip -> constraint checks on the
// pre-state -> dynamics -> reward -> penalties -> counters -> done/truncation -> critical
// shutdown -> (optional) episode tally flush and in-kernel auto-reset.  No MFMA: these are
// elementwise ODE updates (HBM-bound, DESIGN.md "Roofline").
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see __graft_entry__.build).
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
    const double *step_noise; const double *reset_noise; uint32_t ld_noise;
    float *reward; double *reward64; uint32_t *flags; float *final_obs; uint32_t ld_obs;
    // scalars
    uint64_t env0; uint32_t seed_lo, seed_hi;
    const uint32_t *t_ptr; uint32_t t_off;   // launch counter t = (t_ptr ? *t_ptr : 0) + t_off (graph replay keeps t on the device)
    int max_steps; float dt32; double dt; uint32_t hflags; uint32_t cmask;
};

// IndustrialEnv.step for one lane, entirely in registers (base.py:157-213): action clip, constraint
// check on the pre-state and dynamics, then post_core = reward / penalties / termination on the
// finished transition.
template <class Env>
__device__ __forceinline__ void clip_action(float (&a)[Env::A])
{
#pragma unroll
    for (int k = 0; k < Env::A; ++k) {            // base.py:167 np.clip(action, -1, 1) == min(max(x,lo),hi)
        float x = a[k];
        x = (x < -1.0f) ? -1.0f : x;
        x = (x > 1.0f) ? 1.0f : x;
        a[k] = x;
    }
}

template <class Env>
__device__ __forceinline__ void post_core(const float (&n)[Env::S], const float (&a)[Env::A], uint32_t vb,
                                          int step_pre, int max_steps, StepResult<Env> &out)
{
    using R = typename Env::reward_t;
    R r = Env::reward(n, a);                      // base.py:176
#pragma unroll
    for (int k = 0; k < 3; ++k)                   // base.py:179-183, constraint order
        r = (vb & (1u << k)) ? (R)(r + (R)Env::penalty(k)) : r;
    const int nviol = __popc(vb);
    const int ncrit = __popc(vb & Env::CRIT_MASK);
    bool term = Env::done(n);                     // base.py:190
    const bool trunc = (step_pre + 1) >= max_steps;   // base.py:191
    if (ncrit > 0) { term = true; r = r - (R)1000; }  // base.py:195-198
    out.reward = r; out.viol_bits = vb; out.nviol = nviol; out.ncrit = ncrit;
    out.terminated = term; out.truncated = trunc; out.shutdown = ncrit > 0;   // info['critical_shutdown'], base.py:210
}

template <class Env>
__device__ __forceinline__ void step_core(const float (&s)[Env::S], float (&a)[Env::A],
                                          const double (&nz)[Env::KS > 0 ? Env::KS : 1], int step_pre,
                                          int max_steps, float dt32, double dt, uint32_t cmask,
                                          float (&n)[Env::S], StepResult<Env> &out)
{
    if constexpr (Env::CUSTOM_STEP) {             // the Advanced envs override step() wholesale
        Env::custom_step(s, a, step_pre, max_steps, dt32, n, out);
        out.viol_bits &= cmask;
        out.nviol = __popc(out.viol_bits);
        return;
    } else {
        clip_action<Env>(a);
        const uint32_t vb = Env::violated(s, a) & cmask;   // base.py:170 (and again :180, same inputs); cmask: base.py:224-228
        Env::dynamics(s, a, nz, dt32, dt, n);         // base.py:173
        post_core<Env>(n, a, vb, step_pre, max_steps, out);
    }
}

// The per-lane flag word of one step (include/nig.h NIG_FLAG_*).
template <class Env>
__device__ __forceinline__ uint32_t pack_flags(const StepResult<Env> &res, int step)
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
template <class Env>
__device__ __forceinline__ void pair_noise(uint32_t w0, uint32_t w1, const float4 *tab, double (&n)[Env::KS > 0 ? Env::KS : 1])
{
    if constexpr (Env::SHARED_STEP_BLOCK) Env::step_noise(w0, w1, tab, n);
}
template <class Env>
__device__ __forceinline__ void pair_fetch(uint32_t w0, uint32_t w1, const float4 *tab, ProbitFetch (&f)[Env::KS > 0 ? Env::KS : 1])
{
    if constexpr (Env::SHARED_STEP_BLOCK) Env::step_noise_fetch(w0, w1, tab, f);
}
template <class Env>
__device__ __forceinline__ void pair_eval(const ProbitFetch (&f)[Env::KS > 0 ? Env::KS : 1], double (&n)[Env::KS > 0 ? Env::KS : 1])
{
    if constexpr (Env::SHARED_STEP_BLOCK) Env::step_noise_eval(f, n);
}
template <class Env>
__device__ __forceinline__ void draw_one(const RngKey &k, double (&n)[Env::KS > 0 ? Env::KS : 1])
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
__device__ __forceinline__ void flush_tally(double *T, uint32_t ld, double ret, int step, uint32_t viol_ep, int ncrit)
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
#pragma unroll
    for (int r = 0; r < NIG_T_ROWS; ++r) T[(size_t)r * ld] = v[r];
}

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
    __device__ __forceinline__ void merge(double *T, uint32_t ld) const
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
#pragma unroll
        for (int r = 0; r < NIG_T_ROWS; ++r) T[(size_t)r * ld] = v[r];
    }
};

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
template <class Env, bool PARITY>
__global__ void __launch_bounds__(BLOCK, Env::STEP_WAVES) step_kernel(const StepArgs p)
{
    constexpr int S = Env::S, A = Env::A, KS = Env::KS, KR = Env::KR;
    constexpr int KSN = KS > 0 ? KS : 1;
    constexpr int NWAVE = BLOCK / 64;
    __shared__ unsigned short s_list[BLOCK];
    __shared__ int s_cnt[NWAVE];

    const unsigned tid = threadIdx.x;
    const uint32_t base = blockIdx.x * BLOCK;                  // block-uniform
    const bool in_range = base + tid < p.B;
    const uint32_t t_now = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off;

    // ---- one batch of loads -------------------------------------------------------------
    const uint32_t *ctr_row = p.ctr + base;
    const float *st_row = p.state + base;
    const float *act_row = p.actions + base;
    uint32_t ctr = NIG_CTR_DONE;
    float s[S], a[A], n[S];
    double nz[KSN];
    if (in_range) {
        ctr = ctr_row[tid];
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = (st_row + k * p.ld_state)[tid];
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = (act_row + k * p.ld_act)[tid];
        if constexpr (PARITY && KS > 0) {
            const double *nz_row = p.step_noise + base;
#pragma unroll
            for (int k = 0; k < KS; ++k) nz[k] = (nz_row + k * p.ld_noise)[tid];
        }
    } else {
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = 0.0f;
        if constexpr (PARITY && KS > 0) {
#pragma unroll
            for (int k = 0; k < KS; ++k) nz[k] = 0.0;
        }
    }
    // The generator's table: staged in LDS (only now: the state/action loads above are already in flight)
    // when a lane looks up many normals per launch; an env with a couple of draws per step reads its
    // entries straight from the 12 KiB global table (L2-resident) -- staging 12 KiB per block plus a
    // block barrier costs more than two or three 16-byte loads per lane.
    constexpr bool STAGE_TABLE = PARITY ? false : (KS > 4);
    __shared__ float4 s_probit_[STAGE_TABLE ? 768 : 1];
    if constexpr (STAGE_TABLE) {
        for (int i_ = (int)threadIdx.x; i_ < 768; i_ += BLOCK) s_probit_[i_] = NIG_PROBIT[i_];
        __syncthreads();
    }
    const float4 *s_probit = STAGE_TABLE ? s_probit_ : NIG_PROBIT;
    const bool active = in_range && !(ctr & NIG_CTR_DONE);     // base.py:159-160: finished lanes wait for reset

    const RngKey key = make_key(p.env0 + (uint64_t)(base + tid), t_now, p.seed_lo, p.seed_hi, s_probit);
    if constexpr (KS > 0) {
        if constexpr (!PARITY) Env::draw_step(key, nz);
    } else {
        nz[0] = 0.0;
    }

    // ---- IndustrialEnv.step in registers --------------------------------------------------
    const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
    StepResult<Env> res;
    step_core<Env>(s, a, nz, step_pre, p.max_steps, p.dt32, p.dt, p.cmask, n, res);

    const int step = step_pre + 1;
    const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;   // base.py:182
    const bool done = res.terminated || res.truncated;
    uint32_t fl = pack_flags<Env>(res, step);
    uint32_t nctr = (uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT);
    const bool autoreset = (p.hflags & NIG_F_AUTORESET) != 0;
    const bool need_reset = active && done && autoreset;

    if (active) {
        double ret = 0.0;
        if (p.tally) {                            // utils.py:99  episode_return += reward
            const double prev = (p.ep_ret + base)[tid];
            if constexpr (Env::RET_F32) ret = (double)((float)prev + res.reward);   // float32 accumulation (CR)
            else ret = prev + (double)res.reward;
        }
        if (done) {
            (p.life_viol + base)[tid] += (long long)viol_ep;   // base.py:183 total_violations (never reset)
            if (p.tally) { flush_tally(p.tally + base + tid, p.ld, ret, step, viol_ep, res.ncrit); ret = 0.0; }
            if (p.final_obs) {
                float *fo = p.final_obs + base;
#pragma unroll
                for (int k = 0; k < S; ++k) (fo + k * p.ld_obs)[tid] = n[k];
            }
            if (autoreset) { nctr = 0u; fl |= NIG_FLAG_DID_RESET; }
            else nctr |= NIG_CTR_DONE;
        }
        if (!need_reset) {                        // a resetting lane's state is written by the compacted pass below
            float *so = p.state + base;
#pragma unroll
            for (int k = 0; k < S; ++k) (so + k * p.ld_state)[tid] = n[k];
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
};

// OUT: 0 = no per-step outputs, 1 = reward + flag word, 2 = + observation rows [S][ld],
//      3 = + observation row-major [B][S].  Compile-time so that the number of stores per
// iteration is static and the wait for the prefetched action is a counted vmcnt(N), not a
// full drain of the iteration's stores.
template <class Env, int OUT, bool PAIRED>
__global__ void __launch_bounds__(BLOCK, Env::ROLLOUT_WAVES) rollout_kernel(const RolloutArgs q)
{
    constexpr int S = Env::S, A = Env::A, KS = Env::KS, KR = Env::KR;
    constexpr int KSN = KS > 0 ? KS : 1;
    // Envs whose episodes are short (PowerGrid ~6 steps, RobotAssembly: most waves see a reset
    // every step) compact the finishing lanes of the 256-lane block through LDS each step and let
    // ONE wave produce all their initial states at full lane utilisation; the owners read them
    // back from LDS.  Costs two block barriers per step, saves running the whole reset path in
    // every wave for a few active lanes.  ChemicalReactor (0.3 % of lanes per step) keeps the
    // barrier-free divergent form.
    constexpr bool COMPACT = Env::COMPACT_RESET;
    constexpr int NWAVE = BLOCK / 64;
    __shared__ float s_init[COMPACT ? S * BLOCK : 1];
    __shared__ unsigned short s_list[COMPACT ? BLOCK : 1];
    __shared__ int s_cnt[COMPACT ? NWAVE : 1];
    __shared__ v4f s_tr[OUT == 3 ? NWAVE : 1][OUT == 3 ? 16 * S : 1];          // per-wave transpose of the row-major observation rows (64 x S floats)
    NIG_STAGE_PROBIT(s_probit);
    const StepArgs &p = q.s;
    const unsigned tid = threadIdx.x;
    const uint32_t base = blockIdx.x * BLOCK;
    const bool in_range = base + tid < p.B;
    if constexpr (!COMPACT) {
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
    double nzA[KSN], nzB[KSN];
    nzA[0] = 0.0; nzB[0] = 0.0;
    uint32_t kept0 = 0u, kept1 = 0u;          // words 2-3 of the current pair's block
    int slot = 0;
    // Wave-uniform running pointers instead of it * stride products: the per-step 64-bit scalar
    // multiplies and adds of the address arithmetic were ~40 of the step's ~80 SALU issue slots.
    const float *act_next = ring;              // ring slot of the step whose action is fetched next
    float *rew_row = p.reward ? p.reward + base + (size_t)q.it0 * q.out_stride : nullptr;
    uint32_t *fl_row = p.flags ? p.flags + base + (size_t)q.it0 * q.out_stride : nullptr;
    float *obs_row = nullptr;                  // this step's observation block / rows
    if constexpr (OUT == 3) obs_row = q.obs_out + (size_t)q.it0 * q.obs_step_stride + (size_t)(base + (tid & ~63u)) * S;
    if constexpr (OUT == 2) obs_row = q.obs_out + (size_t)q.it0 * q.obs_step_stride + base;
    const bool may_freeze = !autoreset || base + BLOCK > p.B;   // block-uniform: lanes can be frozen (finished, or out of range)

    auto one_step = [&](auto pos_tag, float (&abuf)[A], double (&nz)[KSN], const int it) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = abuf[k];
        const bool frozen = may_freeze && (ctr & NIG_CTR_DONE) != 0;   // no auto-reset: base.py:159-160
        const RngKey key = make_key(gi, t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi, s_probit);
        if constexpr (KS > 0 && !SHARE) draw_one<Env>(key, nz);
        const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
        StepResult<Env> res;
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
                v4f *tr = s_tr[tid >> 6] + (tid & 63u) * (S / 4);
#pragma unroll
                for (int k = 0; k < S / 4; ++k) { v4f v = {n[4 * k], n[4 * k + 1], n[4 * k + 2], n[4 * k + 3]}; tr[k] = v; }
            } else {
                float *tr = reinterpret_cast<float *>(s_tr[tid >> 6]) + (tid & 63u) * S;
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
            const v4f *tr = s_tr[tid >> 6];
            v4f *oo = reinterpret_cast<v4f *>(obs_row);
            constexpr int NV = (16 * S + 63) / 64;  // float4 pieces per lane: the wave's block is 64*S floats = 16*S float4
            v4f v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = tr[(16 * S % 64 == 0 || lane + 64u * k < 16u * S) ? lane + 64u * k : 0u];
            if (wave_env0 + 64u <= p.B) {          // wave-uniform: the whole wave exists
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
        if constexpr (!COMPACT) {
            if (done && autoreset) {               // divergent per-lane reset (base.py:133-155)
                double rn[KR > 0 ? KR : 1];
                Env::draw_init(key, rn);
                Env::init(rn, n);
                ctr = 0u;
            }
        } else if (autoreset) {                    // block-uniform
            const unsigned wave = tid >> 6, lane = tid & 63u;
            const unsigned long long m = __ballot(done);
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
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
        if (lt.episodes > 0) lt.merge(p.tally + base + tid, p.ld);
    }
}

// ------------------------------------------------------------------------------------------
// Closed-loop fused rollout: action = on-device policy(observation) -> IndustrialEnv.step, n steps
// per launch, state / counters / tallies / PID memory in registers.  No loads inside the loop
// (the policy struct is staged in LDS), so the optional outputs can stay
// run-time switches.  Spec of the policy arithmetic: include/nig.h "nig-policy-v1".
struct PolicyArgs {
    StepArgs s;
    const nig_policy *pol;      // device copy
    int n_steps;
    uint32_t out_stride;
    float *obs_out; uint64_t obs_step_stride;                        // row-major [B][S] per step, pre-step obs
    float *act_out; uint32_t ld_act_out; uint64_t act_step_stride;   // [A][ld] per step
};

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
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(q.pol);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&s_pol);
        for (unsigned i = threadIdx.x; i < sizeof(nig_policy) / 4; i += BLOCK) dst[i] = src[i];
    }
    NIG_STAGE_PROBIT(s_probit);                    // (ends with the block barrier that also publishes s_pol)
    const nig_policy *pol = &s_pol;
    const StepArgs &p = q.s;
    const unsigned tid = threadIdx.x;
    const uint32_t base = blockIdx.x * BLOCK;
    if (base + tid >= p.B) return;
    const uint32_t t_base = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off;
    const uint64_t gi = p.env0 + (uint64_t)(base + tid);
    const bool autoreset = (p.hflags & NIG_F_AUTORESET) != 0;
    const bool tally = p.tally != nullptr;

    uint32_t ctr = (p.ctr + base)[tid];
    float s[S], a[A], n[S], integ[A], eprev[A];
    double nz[KSN];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = (p.state + base + k * p.ld_state)[tid];
#pragma unroll
    for (int j = 0; j < A; ++j) { integ[j] = 0.0f; eprev[j] = 0.0f; }
    double ret = tally ? (p.ep_ret + base)[tid] : 0.0;
    LaneTally lt;
    lt.clear();

    for (int it = 0; it < q.n_steps; ++it) {
        const uint32_t orow = (uint32_t)it * q.out_stride;
        if (ctr & NIG_CTR_DONE) {                  // frozen lane: base.py:159-160
            if (p.flags) (p.flags + base + orow)[tid] = NIG_FLAG_INACTIVE | ((ctr & NIG_CTR_STEP_MASK) << NIG_FLAG_STEP_SHIFT);
            if (p.reward) (p.reward + base + orow)[tid] = 0.0f;
            continue;
        }
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
        if constexpr (KS > 0) Env::draw_step(key, nz); else nz[0] = 0.0;
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
#pragma unroll
    for (int k = 0; k < S; ++k) (p.state + base + k * p.ld_state)[tid] = s[k];
    (p.ctr + base)[tid] = ctr;
    if (lt.life != 0) (p.life_viol + base)[tid] += lt.life;
    if (tally) {
        (p.ep_ret + base)[tid] = ret;
        if (lt.episodes > 0) lt.merge(p.tally + base + tid, p.ld);
    }
}

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
__host__ __device__ constexpr int mlp_records(int S) { return MLP_MT * (S / 2 + 1) + MLP_MT * (MLP_MT * 16 + 1 + 16) + 1; }

template <class Env>
__global__ void __launch_bounds__(BLOCK, 1) rollout_mlp_kernel(const MlpArgs q)
{
    constexpr int S = Env::S, A = Env::A, KS = Env::KS, KR = Env::KR;
    constexpr int KSN = KS > 0 ? KS : 1;
    static_assert(S % 2 == 0 && A <= 8, "MFMA actor needs an even state dim and at most 8 actions");
    NIG_STAGE_PROBIT(s_probit);
    const StepArgs &p = q.s;
    const unsigned tid = threadIdx.x, lane = tid & 63u, half = lane >> 5, e = lane & 31u;
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
    double nz[KSN];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = in_range ? (p.state + k * p.ld_state)[li] : 0.0f;
    double ret = (tally && in_range) ? p.ep_ret[li] : 0.0;
    LaneTally lt;
    lt.clear();
    for (int it = 0; it < q.n_steps; ++it) {
        // ---------------- actor: 3 layers of f32 MFMA, whole wave (EXEC all ones) ----------------
        // A operands come from the pre-ordered weight stream (block-uniform base + lane), one
        // 256-byte record per MFMA.  An MFMA issues every 64 cycles and an L2 hit takes ~500-900, so
        // records are fetched RING = 29 MFMAs ahead through a register ring (145 records per
        // hidden-tile iteration = 5 x 29: ring slots are compile-time constants).
        const float *w = q.wstream;                          // uniform; lane offset added per access
        f32x16 h1[MLP_MT];
        {
            constexpr int R1 = S / 2 + 1;                    // records per layer-1 tile
            float cur[R1], nxt[R1];
#pragma unroll
            for (int j = 0; j < R1; ++j) cur[j] = w[j * 64 + lane];
#pragma unroll
            for (int m = 0; m < MLP_MT; ++m) {
                if (m + 1 < MLP_MT) {
#pragma unroll
                    for (int j = 0; j < R1; ++j) nxt[j] = w[(R1 + j) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < S / 2; ++ks) {
                    const float b = half ? s[2 * ks + 1] : s[2 * ks];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[ks], b, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[R1 - 1], half ? 0.0f : 1.0f, acc, 0, 0, 0);   // + b1
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);                                   // ReLU
                h1[m] = acc;
                w += R1 * 64;
#pragma unroll
                for (int j = 0; j < R1; ++j) cur[j] = nxt[j];
            }
        }
        f32x16 out = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        constexpr int RING = 29, PER = MLP_MT * 16 + 1 + 16;     // 145 records per m2 iteration
        static_assert(PER % RING == 0, "ring slots must be static across iterations");
        float ring[RING];
#pragma unroll
        for (int j = 0; j < RING; ++j) ring[j] = w[j * 64 + lane];
        for (int m2 = 0; m2 < MLP_MT; ++m2) {          // a real loop: the body is 145 MFMAs of straight-line code
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const float *wn = w + RING * 64;           // record i + RING (the stream is padded past its end)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const float aop = ring[i % RING];
                ring[i % RING] = wn[i * 64 + lane];
                // pin the source order: hipcc's scheduler otherwise sinks every prefetch to just before
                // its use (one load in flight, MFMA pipe idle ~75 % of the time)
                __builtin_amdgcn_sched_barrier(0);
                if (i < MLP_MT * 16) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, h1[i / 16][i % 16], acc, 0, 0, 0);
                } else if (i == MLP_MT * 16) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, half ? 0.0f : 1.0f, acc, 0, 0, 0);      // + b2
                } else {                               // this h2 tile is consumed at once by the head
                    out = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, fmaxf(acc[i - MLP_MT * 16 - 1], 0.0f), out, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            w += PER * 64;
        }
        out = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[0], half ? 0.0f : 1.0f, out, 0, 0, 0);             // + b3
        // action j sits in register j&3 of lane half j>>2: hand every lane all A of them
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mine = out[r];
            const float other = __shfl_xor(mine, 32);
            if (r < A) a[r] = det_tanhf(half ? other : mine);
            if (r + 4 < A) a[r + 4] = det_tanhf(half ? mine : other);
        }

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
        if constexpr (KS > 0) Env::draw_step(key, nz); else nz[0] = 0.0;
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
        if (lt.episodes > 0) lt.merge(p.tally + li, p.ld);
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

__global__ void __launch_bounds__(BLOCK) init_ws_kernel(uint32_t *ctr, long long *life, double *ep_ret, double *tally,
                                                        int64_t ld, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= ld) return;
    ctr[i] = (i < B) ? NIG_CTR_DONE : NIG_CTR_DONE;   // nothing steps before the first reset
    life[i] = 0;
    if (ep_ret) ep_ret[i] = 0.0;
    if (tally) {
#pragma unroll
        for (int r = 0; r < NIG_T_ROWS; ++r) tally[(int64_t)r * ld + i] = 0.0;
        tally[(int64_t)NIG_T_RET_MIN * ld + i] = __builtin_inf();
        tally[(int64_t)NIG_T_RET_MAX * ld + i] = -__builtin_inf();
    }
}

__global__ void set_u32_kernel(uint32_t *p, uint32_t v) { *p = v; }

__global__ void __launch_bounds__(BLOCK) safety_metrics_kernel(const uint32_t *flags, int32_t *out, int64_t ld_out,
                                                               int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    const uint32_t f = flags[i];
    const int nv = (int)((f >> NIG_FLAG_NVIOL_SHIFT) & 3u), nc = (int)((f >> NIG_FLAG_NCRIT_SHIFT) & 3u);
    out[0 * ld_out + i] = 3 - nv;   // constraints_satisfied   base.py:96-106
    out[1 * ld_out + i] = 3;        // total_constraints       base.py:115
    out[2 * ld_out + i] = nv;       // violation_count
    out[3 * ld_out + i] = nc;       // critical_violations
    out[4 * ld_out + i] = 3 - nv;   // safety_score * total    base.py:116
}

// deterministic two-stage reduction of the tally rows: fixed grid, fixed tree order
__global__ void __launch_bounds__(BLOCK) reduce_tally_stage1(const double *tally, int64_t ld, int64_t B, double *scratch)
{
    __shared__ double sh[BLOCK];
    for (int r = 0; r < NIG_T_ROWS; ++r) {
        const bool is_min = (r == NIG_T_RET_MIN), is_max = (r == NIG_T_RET_MAX);
        double acc = is_min ? __builtin_inf() : (is_max ? -__builtin_inf() : 0.0);
        for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < B; i += (int64_t)gridDim.x * BLOCK) {
            const double v = tally[(int64_t)r * ld + i];
            acc = is_min ? fmin(acc, v) : (is_max ? fmax(acc, v) : acc + v);
        }
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int w = BLOCK / 2; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) {
                const double o = sh[threadIdx.x + w];
                sh[threadIdx.x] = is_min ? fmin(sh[threadIdx.x], o) : (is_max ? fmax(sh[threadIdx.x], o) : sh[threadIdx.x] + o);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) scratch[(int64_t)blockIdx.x * NIG_T_ROWS + r] = sh[0];
        __syncthreads();
    }
}

__global__ void reduce_tally_stage2(const double *scratch, int nblk, double *out)
{
    const int r = threadIdx.x;
    if (r >= NIG_T_ROWS) return;
    const bool is_min = (r == NIG_T_RET_MIN), is_max = (r == NIG_T_RET_MAX);
    double acc = is_min ? __builtin_inf() : (is_max ? -__builtin_inf() : 0.0);
    for (int b = 0; b < nblk; ++b) {
        const double v = scratch[(int64_t)b * NIG_T_ROWS + r];
        acc = is_min ? fmin(acc, v) : (is_max ? fmax(acc, v) : acc + v);
    }
    out[r] = acc;
}

__global__ void __launch_bounds__(BLOCK) copy_rows_kernel(const float *src, int64_t ld_src, float *dst, int64_t ld_dst,
                                                          int rows, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    for (int k = 0; k < rows; ++k) dst[(int64_t)k * ld_dst + i] = src[(int64_t)k * ld_src + i];
}

}  // namespace nig

// =====================================================================================
// host side: C ABI
// =====================================================================================
using namespace nig;

struct nig_handle {
    int env;
    int device;
    int64_t B;
    uint64_t seed, env0;
    int max_steps;
    double dt;
    uint32_t flags;
    uint32_t t;            // RNG launch counter
    uint32_t cmask;        // enabled built-in constraints (bit k)
    nig_layout lay;
    char *ws;
    bool owns_ws;
    double *scratch;       // reduce scratch [REDUCE_BLOCKS][NIG_T_ROWS] (inside workspace tail)
    uint32_t *t_dev;       // device copy of t read by graph-replayed step kernels
    float *state;          // state rows: inside the workspace, or an array bound by the caller
    int64_t ld_state;
    nig_policy *pol_dev;   // device copy of the installed policy
    nig_policy pol_host;   // staging copy (must outlive the async H2D copy)
    bool has_policy;
    float *mlp_stream;     // device copy of the MFMA operand stream of the MLP actor (owned)
    char *hst_pinned;      // host-buffer entry points: pinned staging + its device mirror (owned, lazy)
    char *hst_dev;
    size_t hst_bytes;
};

struct nig_plan {
    nig_handle *h;
    int n_steps;
    hipGraph_t graph;
    hipGraphExec_t exec;
};

static unsigned grid_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) return fail(NIG_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

static const nig_env_spec SPECS[NIG_NUM_ENVS] = {
    {12, 3, 3, 500, 2, 8, 0.1, {-100.0, -50.0, -25.0}, {1, 1, 0}, 1},
    {32, 8, 3, 1000, 23, 31, 0.1, {-50.0, -30.0, -20.0}, {1, 1, 0}, 0},
    {24, 7, 3, 1000, 0, 7, 0.1, {-100.0, -200.0, -50.0}, {1, 1, 0}, 0},
    /* Advanced envs: 4 / 3 safety-metric conditions, no penalties through the base loop, deterministic */
    {20, 6, 4, 1000, 0, 0, 0.1, {0.0, 0.0, 0.0}, {0, 0, 0}, 0},
    {32, 8, 3, 500, 0, 0, 0.1, {0.0, 0.0, 0.0}, {0, 0, 0}, 0},
    /* build-specified plants (spec_plants.py): dims and constraint tables come from the generated data */
#define NIG_SPEC_ROW(K) {SpecPlant<K>::S, SpecPlant<K>::A, 3, SpecPlant<K>::MAX_STEPS, SpecPlant<K>::KS, SpecPlant<K>::KR, 0.1, \
                         {NIG_SPEC_PLANTS[K].pen[0], NIG_SPEC_PLANTS[K].pen[1], NIG_SPEC_PLANTS[K].pen[2]},                    \
                         {NIG_SPEC_PLANTS[K].crit[0], NIG_SPEC_PLANTS[K].crit[1], NIG_SPEC_PLANTS[K].crit[2]}, 1}
    NIG_SPEC_ROW(0), NIG_SPEC_ROW(1), NIG_SPEC_ROW(2), NIG_SPEC_ROW(3),
#undef NIG_SPEC_ROW
};
static const char *NAMES[NIG_NUM_ENVS] = {"ChemicalReactor-v0", "PowerGrid-v0", "RobotAssembly-v0",
                                          "AdvancedChemicalReactor-v0", "AdvancedPowerGrid-v0",
                                          "HVACControl-v0", "WaterTreatment-v0", "SteelAnnealing-v0", "SupplyChain-v0"};

// run `F<Env>(args...)` for the env type behind a run-time id
#define NIG_DISPATCH_ENV(env_id, CALL)                                      \
    switch (env_id) {                                                       \
    case NIG_ENV_CHEMICAL_REACTOR: { using E = ChemicalReactor; CALL; } break;          \
    case NIG_ENV_POWER_GRID: { using E = PowerGrid; CALL; } break;                      \
    case NIG_ENV_ROBOT_ASSEMBLY: { using E = RobotAssembly; CALL; } break;              \
    case NIG_ENV_ADV_CHEMICAL_REACTOR: { using E = AdvancedChemicalReactor; CALL; } break; \
    case NIG_ENV_ADV_POWER_GRID: { using E = AdvancedPowerGrid; CALL; } break;          \
    case NIG_ENV_HVAC_CONTROL: { using E = HVACControl; CALL; } break;                  \
    case NIG_ENV_WATER_TREATMENT: { using E = WaterTreatment; CALL; } break;            \
    case NIG_ENV_STEEL_ANNEALING: { using E = SteelAnnealing; CALL; } break;            \
    default: { using E = SupplyChain; CALL; } break;                        \
    }

template <class Env>
static void launch_reset(const ResetArgs &a, bool parity, hipStream_t st)
{
    if (parity) hipLaunchKernelGGL((reset_kernel<Env, true>), dim3(grid_for(a.B)), dim3(BLOCK), 0, st, a);
    else hipLaunchKernelGGL((reset_kernel<Env, false>), dim3(grid_for(a.B)), dim3(BLOCK), 0, st, a);
}

// the MFMA actor exists for even state dims and at most 8 actions (nig_set_mlp_policy refuses the others,
// so a handle of such an env never gets here with a weight stream installed)
template <class Env>
static void launch_mlp(const MlpArgs &q, unsigned grid, hipStream_t st)
{
    if constexpr (Env::S % 2 == 0 && Env::A <= 8) hipLaunchKernelGGL((rollout_mlp_kernel<Env>), dim3(grid), dim3(BLOCK), 0, st, q);
}

template <class Env>
static void launch_step(const StepArgs &a, bool parity, hipStream_t st)
{
    if (parity) hipLaunchKernelGGL((step_kernel<Env, true>), dim3(grid_for(a.B)), dim3(BLOCK), 0, st, a);
    else hipLaunchKernelGGL((step_kernel<Env, false>), dim3(grid_for(a.B)), dim3(BLOCK), 0, st, a);
}

template <class Env, bool PAIRED>
static void launch_rollout_form(int out_mode, const RolloutArgs &q, unsigned grid, hipStream_t st)
{
    switch (out_mode) {
    case 0: hipLaunchKernelGGL((rollout_kernel<Env, 0, PAIRED>), dim3(grid), dim3(BLOCK), 0, st, q); break;
    case 1: hipLaunchKernelGGL((rollout_kernel<Env, 1, PAIRED>), dim3(grid), dim3(BLOCK), 0, st, q); break;
    case 2: hipLaunchKernelGGL((rollout_kernel<Env, 2, PAIRED>), dim3(grid), dim3(BLOCK), 0, st, q); break;
    default: hipLaunchKernelGGL((rollout_kernel<Env, 3, PAIRED>), dim3(grid), dim3(BLOCK), 0, st, q); break;
    }
}

// t0 = launch counter of the call's first step (host-known: rollouts are never graph-captured)
template <class Env>
static void launch_rollout_env(int out_mode, const RolloutArgs &q, uint32_t t0, unsigned grid, hipStream_t st)
{
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

static void launch_rollout(int env, int out_mode, const RolloutArgs &q, uint32_t t0, unsigned grid, hipStream_t st)
{
    NIG_DISPATCH_ENV(env, launch_rollout_env<E>(out_mode, q, t0, grid, st));
}

static StepArgs base_step_args(const nig_handle *h)
{
    const nig_layout &L = h->lay;
    StepArgs a;
    memset(&a, 0, sizeof a);
    a.state = h->state; a.ld_state = (uint32_t)h->ld_state; a.ctr = (uint32_t *)(h->ws + L.off_ctr);
    a.life_viol = (long long *)(h->ws + L.off_life_viol);
    a.ep_ret = L.off_ep_return >= 0 ? (double *)(h->ws + L.off_ep_return) : nullptr;
    a.tally = L.off_tally >= 0 ? (double *)(h->ws + L.off_tally) : nullptr;
    a.ld = (uint32_t)L.ld; a.B = (uint32_t)h->B;
    a.env0 = h->env0; a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32);
    a.max_steps = h->max_steps; a.dt32 = (float)h->dt; a.dt = h->dt; a.hflags = h->flags; a.cmask = h->cmask;
    return a;
}

static void dispatch_step(const nig_handle *h, const StepArgs &a, bool parity, hipStream_t st)
{
    NIG_DISPATCH_ENV(h->env, launch_step<E>(a, parity, st));
}

extern "C" {

const char *nig_version(void) { return "nig 0.1.0 (gfx950)"; }
const char *nig_last_error(void) { return g_err; }

int nig_env_id(const char *name)
{
    if (!name) return -1;
    for (int i = 0; i < NIG_NUM_ENVS; ++i)
        if (strcmp(name, NAMES[i]) == 0) return i;
    return -1;
}

const char *nig_env_name(int env) { return (env >= 0 && env < NIG_NUM_ENVS) ? NAMES[env] : nullptr; }

int nig_env_spec_get(int env, nig_env_spec *out)
{
    if (env < 0 || env >= NIG_NUM_ENVS || !out) return fail(NIG_ERR_INVALID, "nig_env_spec_get: bad env id%s");
    *out = SPECS[env];
    return NIG_OK;
}

static int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

int nig_layout_query(int env, int64_t batch, uint32_t flags, nig_layout *out)
{
    if (env < 0 || env >= NIG_NUM_ENVS || !out) return fail(NIG_ERR_INVALID, "nig_layout_query: bad env id%s");
    if (batch <= 0) return fail(NIG_ERR_INVALID, "nig_layout_query: batch must be positive%s");
    nig_layout L;
    L.batch = batch;
    L.ld = align_up(batch, 64);
    int64_t off = 0;
    L.off_state = off;      off = align_up(off + (int64_t)SPECS[env].state_dim * L.ld * 4, 256);
    L.off_ctr = off;        off = align_up(off + L.ld * 4, 256);
    L.off_life_viol = off;  off = align_up(off + L.ld * 8, 256);
    if (flags & NIG_F_TALLY) {
        L.off_ep_return = off;  off = align_up(off + L.ld * 8, 256);
        L.off_tally = off;      off = align_up(off + (int64_t)NIG_T_ROWS * L.ld * 8, 256);
    } else {
        L.off_ep_return = -1;
        L.off_tally = -1;
    }
    // tail: reduce scratch + the device-resident launch counter used by plans
    off = align_up(off + (int64_t)REDUCE_BLOCKS * NIG_T_ROWS * 8, 256) + 256 + POLICY_BYTES;
    L.bytes = off;
    *out = L;
    return NIG_OK;
}


int nig_create(int env, int64_t batch, int device, uint64_t seed, uint64_t env_index0, int32_t max_episode_steps,
               double dt, uint32_t flags, void *workspace, nig_handle **out)
{
    if (!out) return fail(NIG_ERR_INVALID, "nig_create: out is NULL%s");
    *out = nullptr;
    if (env < 0 || env >= NIG_NUM_ENVS) return fail(NIG_ERR_INVALID, "nig_create: unknown env id%s");
    if (batch <= 0 || batch > NIG_MAX_BATCH) return fail(NIG_ERR_INVALID, "nig_create: batch outside [1, 2^24]%s");
    if (max_episode_steps < 0 || max_episode_steps > NIG_MAX_EPISODE_STEPS)
        return fail(NIG_ERR_INVALID, "nig_create: max_episode_steps outside [1, 21845]%s");
    if (dt < 0.0 || dt != dt) return fail(NIG_ERR_INVALID, "nig_create: bad dt%s");
    if (SPECS[env].n_constraints > 3 && max_episode_steps > 16383)
        return fail(NIG_ERR_INVALID, "nig_create: max_episode_steps > 16383 for an env with 4 safety conditions%s");
    if (env == NIG_ENV_CHEMICAL_REACTOR && dt != 0.0 && dt != 0.1)
        return fail(NIG_ERR_UNSUPPORTED, "nig_create: ChemicalReactor hard-codes dt=0.1 upstream (chemical_reactor.py:68)%s");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(NIG_ERR_NODEVICE, "nig_create: no HIP device (%s); there is no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= ndev) return fail(NIG_ERR_INVALID, "nig_create: device index out of range%s");
    HIP_TRY(hipSetDevice(device));

    nig_handle *h = new (std::nothrow) nig_handle();
    if (!h) return fail(NIG_ERR_INVALID, "nig_create: out of host memory%s");
    h->env = env; h->device = device; h->B = batch; h->seed = seed; h->env0 = env_index0;
    h->max_steps = max_episode_steps ? max_episode_steps : SPECS[env].max_episode_steps;
    h->dt = (dt != 0.0) ? dt : SPECS[env].dt;
    h->flags = flags; h->t = 0; h->cmask = 0xFu;
    nig_layout_query(env, batch, flags, &h->lay);
    if (workspace) {
        if (((uintptr_t)workspace & 255u) != 0) { delete h; return fail(NIG_ERR_INVALID, "nig_create: workspace not 256-byte aligned%s"); }
        h->ws = (char *)workspace; h->owns_ws = false;
    } else {
        void *p = nullptr;
        hipError_t me = hipMalloc(&p, (size_t)h->lay.bytes);
        if (me != hipSuccess) { delete h; return fail(NIG_ERR_HIP, "hipMalloc workspace: %s", hipGetErrorString(me)); }
        h->ws = (char *)p; h->owns_ws = true;
    }
    h->pol_dev = (nig_policy *)(h->ws + h->lay.bytes - POLICY_BYTES);
    h->t_dev = (uint32_t *)(h->ws + h->lay.bytes - POLICY_BYTES - 256);
    h->scratch = (double *)(h->ws + h->lay.bytes - POLICY_BYTES - 256 - align_up((int64_t)REDUCE_BLOCKS * NIG_T_ROWS * 8, 256));
    h->has_policy = false; h->mlp_stream = nullptr; h->hst_pinned = nullptr; h->hst_dev = nullptr; h->hst_bytes = 0;
    h->state = (float *)(h->ws + h->lay.off_state); h->ld_state = h->lay.ld;
    const nig_layout &L = h->lay;
    hipLaunchKernelGGL(init_ws_kernel, dim3(grid_for(L.ld)), dim3(BLOCK), 0, (hipStream_t)0,
                       (uint32_t *)(h->ws + L.off_ctr), (long long *)(h->ws + L.off_life_viol),
                       L.off_ep_return >= 0 ? (double *)(h->ws + L.off_ep_return) : nullptr,
                       L.off_tally >= 0 ? (double *)(h->ws + L.off_tally) : nullptr, L.ld, L.batch);
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = hipMemsetAsync(h->ws + L.off_state, 0, (size_t)SPECS[env].state_dim * L.ld * 4, (hipStream_t)0);
    if (le == hipSuccess) le = hipStreamSynchronize((hipStream_t)0);
    if (le != hipSuccess) {
        if (h->owns_ws) (void)hipFree(h->ws);
        delete h;
        return fail(NIG_ERR_HIP, "nig_create: workspace init failed: %s", hipGetErrorString(le));
    }
    *out = h;
    return NIG_OK;
}

int nig_destroy(nig_handle *h)
{
    if (!h) return NIG_OK;
    if (h->mlp_stream) (void)hipFree(h->mlp_stream);
    if (h->hst_dev) (void)hipFree(h->hst_dev);
    if (h->hst_pinned) (void)hipHostFree(h->hst_pinned);
    if (h->owns_ws && h->ws) (void)hipFree(h->ws);
    delete h;
    return NIG_OK;
}

int nig_get_layout(const nig_handle *h, nig_layout *out)
{
    if (!h || !out) return fail(NIG_ERR_INVALID, "nig_get_layout: NULL argument%s");
    *out = h->lay;
    return NIG_OK;
}

void *nig_workspace(const nig_handle *h) { return h ? (void *)h->ws : nullptr; }

int nig_get_counter(const nig_handle *h, uint32_t *t)
{
    if (!h || !t) return fail(NIG_ERR_INVALID, "nig_get_counter: NULL argument%s");
    *t = h->t;
    return NIG_OK;
}

int nig_bind_state(nig_handle *h, float *state, int64_t ld)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_bind_state: NULL handle%s");
    if (!state) { h->state = (float *)(h->ws + h->lay.off_state); h->ld_state = h->lay.ld; return NIG_OK; }
    if (ld < h->B || ld > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_bind_state: ld outside [batch, 2^26]%s");
    if (((uintptr_t)state & 3u) != 0) return fail(NIG_ERR_INVALID, "nig_bind_state: unaligned pointer%s");
    h->state = state; h->ld_state = ld;
    return NIG_OK;
}

int nig_set_constraint_mask(nig_handle *h, uint32_t mask)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_set_constraint_mask: NULL handle%s");
    h->cmask = mask & 0xFu;
    return NIG_OK;
}

int nig_set_counter(nig_handle *h, uint32_t t)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_set_counter: NULL handle%s");
    h->t = t;
    return NIG_OK;
}

int nig_reset(nig_handle *h, const uint8_t *mask, const double *init_noise, int64_t ld_noise, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_reset: NULL handle%s");
    if (init_noise && ld_noise < h->B) return fail(NIG_ERR_INVALID, "nig_reset: ld_noise < batch%s");
    const nig_layout &L = h->lay;
    ResetArgs a;
    a.state = h->state; a.ld_state = h->ld_state; a.ctr = (uint32_t *)(h->ws + L.off_ctr);
    a.life_viol = (long long *)(h->ws + L.off_life_viol);
    a.ep_ret = L.off_ep_return >= 0 ? (double *)(h->ws + L.off_ep_return) : nullptr;
    a.ld = L.ld; a.B = h->B; a.mask = mask; a.noise = init_noise; a.ld_noise = ld_noise;
    a.env0 = h->env0; a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32); a.t = h->t;
    hipStream_t st = (hipStream_t)stream;
    NIG_DISPATCH_ENV(h->env, launch_reset<E>(a, init_noise != nullptr, st));
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_step(nig_handle *h, const float *actions, int64_t ld_act, const double *step_noise, const double *reset_noise,
             int64_t ld_noise, float *reward_out, double *reward64_out, uint32_t *flags_out, float *final_obs,
             int64_t ld_obs, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_step: NULL handle%s");
    if (!actions || ld_act < h->B || ld_act > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_step: actions NULL or ld_act outside [batch, 2^26]%s");
    const nig_env_spec &sp = SPECS[h->env];
    const bool autoreset = (h->flags & NIG_F_AUTORESET) != 0;
    // parity mode = the caller supplies every value the reference's RNG would have drawn
    const bool parity = (step_noise != nullptr) || (reset_noise != nullptr);
    if (parity) {
        if (sp.k_step > 0 && !step_noise) return fail(NIG_ERR_INVALID, "nig_step: parity mode needs step_noise%s");
        if (autoreset && !reset_noise) return fail(NIG_ERR_INVALID, "nig_step: parity mode with auto-reset needs reset_noise%s");
        if (ld_noise < h->B || ld_noise > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_step: ld_noise outside [batch, 2^26]%s");
    }
    if (final_obs && (ld_obs < h->B || ld_obs > NIG_MAX_PITCH)) return fail(NIG_ERR_INVALID, "nig_step: ld_obs outside [batch, 2^26]%s");
    h->t += 1;
    StepArgs a = base_step_args(h);
    a.actions = actions; a.ld_act = (uint32_t)ld_act;
    a.step_noise = step_noise; a.reset_noise = reset_noise; a.ld_noise = (uint32_t)ld_noise;
    a.reward = reward_out; a.reward64 = reward64_out; a.flags = flags_out; a.final_obs = final_obs; a.ld_obs = (uint32_t)ld_obs;
    a.t_ptr = nullptr; a.t_off = h->t;
    dispatch_step(h, a, parity, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_rollout(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act, int64_t slot_stride,
                int32_t ring_len, float *reward_out, uint32_t *flags_out, int64_t out_stride,
                float *obs_out, int64_t ld_obs, int64_t obs_step_stride, void *stream)
{
    if (!h || !action_ring || n_steps <= 0 || ring_len <= 0) return fail(NIG_ERR_INVALID, "nig_rollout: bad argument%s");
    if (ld_act < h->B || ld_act > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_rollout: ld_act outside [batch, 2^26]%s");
    if (slot_stride < (int64_t)SPECS[h->env].action_dim * ld_act || slot_stride > 0xffffffffLL)
        return fail(NIG_ERR_INVALID, "nig_rollout: slot_stride smaller than one [A][ld_act] slot (or >= 2^32)%s");
    if (out_stride != 0 && (out_stride < h->B || out_stride > NIG_MAX_PITCH))
        return fail(NIG_ERR_INVALID, "nig_rollout: out_stride outside {0} U [batch, 2^26]%s");
    if ((int64_t)n_steps * out_stride > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout: n_steps*out_stride >= 2^32%s");
    const bool obs_aos = obs_out && ld_obs == 0;
    if (obs_out && !obs_aos && (ld_obs < h->B || ld_obs > NIG_MAX_PITCH || obs_step_stride < (int64_t)SPECS[h->env].state_dim * ld_obs))
        return fail(NIG_ERR_INVALID, "nig_rollout: bad observation trajectory pitch%s");
    if (obs_aos && (obs_step_stride < (int64_t)SPECS[h->env].state_dim * h->B || (obs_step_stride & 3) || ((uintptr_t)obs_out & 15)))
        return fail(NIG_ERR_INVALID, "nig_rollout: row-major trajectory needs 16-byte alignment and obs_step_stride >= S*batch (multiple of 4)%s");
    if ((int64_t)h->t + n_steps > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout: launch counter would wrap%s");
    RolloutArgs q;
    memset(&q, 0, sizeof q);
    q.s = base_step_args(h);
    q.s.actions = action_ring; q.s.ld_act = (uint32_t)ld_act;
    q.s.reward = reward_out; q.s.flags = flags_out;
    q.s.t_ptr = nullptr; q.s.t_off = h->t;
    q.n_steps = n_steps; q.it0 = 0; q.ring_len = ring_len; q.slot_stride = (uint32_t)slot_stride; q.out_stride = (uint32_t)out_stride;
    q.obs_out = obs_out; q.ld_obs_out = (uint32_t)ld_obs; q.obs_step_stride = (uint64_t)obs_step_stride; q.obs_aos = obs_aos ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if ((reward_out == nullptr) != (flags_out == nullptr))
        return fail(NIG_ERR_INVALID, "nig_rollout: reward_out and flags_out go together (both or neither)%s");
    if (obs_out && !reward_out)
        return fail(NIG_ERR_INVALID, "nig_rollout: an observation trajectory needs reward_out and flags_out too%s");
    const int out_mode = !reward_out ? 0 : (!obs_out ? 1 : (obs_aos ? 3 : 2));
    launch_rollout(h->env, out_mode, q, h->t + 1u, grid_for(h->B), st);
    HIP_TRY(hipGetLastError());
    h->t += (uint32_t)n_steps;
    return NIG_OK;
}

int nig_set_policy(nig_handle *h, const nig_policy *policy, void *stream)
{
    if (!h || !policy) return fail(NIG_ERR_INVALID, "nig_set_policy: NULL argument%s");
    static_assert(sizeof(nig_policy) <= POLICY_BYTES, "policy struct outgrew its device slot");
    if (policy->kind != NIG_POLICY_AFFINE && policy->kind != NIG_POLICY_PID)
        return fail(NIG_ERR_INVALID, "nig_set_policy: unknown policy kind%s");
    if (!(policy->clip_lo <= policy->clip_hi)) return fail(NIG_ERR_INVALID, "nig_set_policy: clip_lo > clip_hi%s");
    h->pol_host = *policy;
    // recompute the non-zero column mask here so a caller cannot get it wrong
    uint32_t cm = 0;
    for (int k = 0; k < SPECS[h->env].state_dim; ++k)
        for (int j = 0; j < SPECS[h->env].action_dim; ++j)
            if (policy->Wt[k][j] != 0.0f) cm |= (1u << k);
    h->pol_host.colmask = cm;
    HIP_TRY(hipMemcpyAsync(h->pol_dev, &h->pol_host, sizeof(nig_policy), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));   // pageable source: make the staging copy reusable
    h->has_policy = true;
    return NIG_OK;
}

int nig_rollout_policy(nig_handle *h, int32_t n_steps, float *reward_out, uint32_t *flags_out, int64_t out_stride,
                       float *obs_out, int64_t obs_step_stride, float *act_out, int64_t ld_act,
                       int64_t act_step_stride, void *stream)
{
    if (!h || n_steps <= 0) return fail(NIG_ERR_INVALID, "nig_rollout_policy: bad argument%s");
    if (!h->has_policy) return fail(NIG_ERR_INVALID, "nig_rollout_policy: no policy installed (nig_set_policy)%s");
    if (out_stride != 0 && (out_stride < h->B || out_stride > NIG_MAX_PITCH))
        return fail(NIG_ERR_INVALID, "nig_rollout_policy: out_stride outside {0} U [batch, 2^26]%s");
    if ((int64_t)n_steps * out_stride > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_policy: n_steps*out_stride >= 2^32%s");
    if (obs_out && (obs_step_stride < (int64_t)SPECS[h->env].state_dim * h->B || (obs_step_stride & 3) || ((uintptr_t)obs_out & 15)))
        return fail(NIG_ERR_INVALID, "nig_rollout_policy: obs_out needs 16-byte alignment and obs_step_stride >= S*batch (multiple of 4)%s");
    if (act_out && (ld_act < h->B || ld_act > NIG_MAX_PITCH || act_step_stride < (int64_t)SPECS[h->env].action_dim * ld_act))
        return fail(NIG_ERR_INVALID, "nig_rollout_policy: bad action trajectory pitch%s");
    if ((int64_t)h->t + n_steps > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_policy: launch counter would wrap%s");
    PolicyArgs q;
    memset(&q, 0, sizeof q);
    q.s = base_step_args(h);
    q.s.reward = reward_out; q.s.flags = flags_out;
    q.s.t_ptr = nullptr; q.s.t_off = h->t;
    q.pol = h->pol_dev; q.n_steps = n_steps; q.out_stride = (uint32_t)out_stride;
    q.obs_out = obs_out; q.obs_step_stride = (uint64_t)obs_step_stride;
    q.act_out = act_out; q.ld_act_out = (uint32_t)ld_act; q.act_step_stride = (uint64_t)act_step_stride;
    hipStream_t st = (hipStream_t)stream;
    NIG_DISPATCH_ENV(h->env, hipLaunchKernelGGL((rollout_policy_kernel<E>), dim3(grid_for(h->B)), dim3(BLOCK), 0, st, q));
    HIP_TRY(hipGetLastError());
    h->t += (uint32_t)n_steps;
    return NIG_OK;
}

// Row of a 32x32 MFMA result tile held in register t by lane half hf (MI355X_MICROARCH / guide section 3).
static inline int mfma_row(int t, int hf) { return (t & 3) + 8 * (t >> 2) + 4 * hf; }

int nig_set_mlp_policy(nig_handle *h, int32_t hidden, const float *W1, const float *b1, const float *W2, const float *b2,
                       const float *W3, const float *b3, void *stream)
{
    if (!h || !W1 || !b1 || !W2 || !b2 || !W3 || !b3) return fail(NIG_ERR_INVALID, "nig_set_mlp_policy: NULL argument%s");
    if (hidden != MLP_H) return fail(NIG_ERR_UNSUPPORTED, "nig_set_mlp_policy: hidden must be 256 (agents/networks.py default)%s");
    const int S = SPECS[h->env].state_dim, A = SPECS[h->env].action_dim, H = MLP_H;
    if (S % 2 != 0 || A > 8) return fail(NIG_ERR_UNSUPPORTED, "nig_set_mlp_policy: env shape not supported%s");
    const int nrec = mlp_records(S);
    float *host = (float *)calloc((size_t)nrec * 64, sizeof(float));
    if (!host) return fail(NIG_ERR_INVALID, "nig_set_mlp_policy: out of host memory%s");
    // Build the operand stream in exactly the order rollout_mlp_kernel consumes it.  Record = 64 floats;
    // lane l = (i = l & 31, hf = l >> 5) holds W[k(hf)][32*tile + i].
    int r = 0;
    for (int m = 0; m < MLP_MT; ++m) {                      // layer 1, natural k order: k = 2*ks + hf
        for (int ks = 0; ks < S / 2; ++ks, ++r)
            for (int l = 0; l < 64; ++l) host[(size_t)r * 64 + l] = W1[(size_t)(2 * ks + (l >> 5)) * H + 32 * m + (l & 31)];
        for (int l = 0; l < 32; ++l) host[(size_t)r * 64 + l] = b1[32 * m + l];
        ++r;
    }
    for (int m2 = 0; m2 < MLP_MT; ++m2) {
        for (int kt = 0; kt < MLP_MT; ++kt)                   // layer 2: k follows the accumulator register order of h1
            for (int t = 0; t < 16; ++t, ++r)
                for (int l = 0; l < 64; ++l)
                    host[(size_t)r * 64 + l] = W2[(size_t)(32 * kt + mfma_row(t, l >> 5)) * H + 32 * m2 + (l & 31)];
        for (int l = 0; l < 32; ++l) host[(size_t)r * 64 + l] = b2[32 * m2 + l];
        ++r;
        for (int t = 0; t < 16; ++t, ++r)                     // head: rows i >= A are zero
            for (int l = 0; l < 64; ++l)
                if ((l & 31) < A) host[(size_t)r * 64 + l] = W3[(size_t)(32 * m2 + mfma_row(t, l >> 5)) * A + (l & 31)];
    }
    for (int l = 0; l < A; ++l) host[(size_t)r * 64 + l] = b3[l];
    ++r;
    if (r != nrec) { free(host); return fail(NIG_ERR_INVALID, "nig_set_mlp_policy: internal record count mismatch%s"); }
    hipError_t e = hipSuccess;
    // the kernel prefetches up to 29 records past the end (values unused): keep them inside the allocation
    if (!h->mlp_stream) {
        const size_t bytes = (size_t)(mlp_records(32) + 32) * 64 * sizeof(float);
        e = hipMalloc((void **)&h->mlp_stream, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(h->mlp_stream, 0, bytes, (hipStream_t)stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h->mlp_stream, host, (size_t)nrec * 64 * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    if (e != hipSuccess) return fail(NIG_ERR_HIP, "nig_set_mlp_policy: %s", hipGetErrorString(e));
    return NIG_OK;
}

int nig_rollout_mlp(nig_handle *h, int32_t n_steps, float *reward_out, uint32_t *flags_out, int64_t out_stride,
                    float *obs_out, int64_t obs_step_stride, float *act_out, int64_t ld_act, int64_t act_step_stride,
                    void *stream)
{
    if (!h || n_steps <= 0) return fail(NIG_ERR_INVALID, "nig_rollout_mlp: bad argument%s");
    if (!h->mlp_stream) return fail(NIG_ERR_INVALID, "nig_rollout_mlp: no actor installed (nig_set_mlp_policy)%s");
    if (out_stride != 0 && (out_stride < h->B || out_stride > NIG_MAX_PITCH))
        return fail(NIG_ERR_INVALID, "nig_rollout_mlp: out_stride outside {0} U [batch, 2^26]%s");
    if ((int64_t)n_steps * out_stride > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_mlp: n_steps*out_stride >= 2^32%s");
    if (obs_out && (obs_step_stride < (int64_t)SPECS[h->env].state_dim * h->B || (obs_step_stride & 3) || ((uintptr_t)obs_out & 15)))
        return fail(NIG_ERR_INVALID, "nig_rollout_mlp: obs_out needs 16-byte alignment and obs_step_stride >= S*batch (multiple of 4)%s");
    if (act_out && (ld_act < h->B || ld_act > NIG_MAX_PITCH || act_step_stride < (int64_t)SPECS[h->env].action_dim * ld_act))
        return fail(NIG_ERR_INVALID, "nig_rollout_mlp: bad action trajectory pitch%s");
    if ((int64_t)h->t + n_steps > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_mlp: launch counter would wrap%s");
    MlpArgs q;
    memset(&q, 0, sizeof q);
    q.s = base_step_args(h);
    q.s.reward = reward_out; q.s.flags = flags_out;
    q.s.t_ptr = nullptr; q.s.t_off = h->t;
    q.wstream = h->mlp_stream; q.n_steps = n_steps; q.out_stride = (uint32_t)out_stride;
    q.obs_out = obs_out; q.obs_step_stride = (uint64_t)obs_step_stride;
    q.act_out = act_out; q.ld_act_out = (uint32_t)ld_act; q.act_step_stride = (uint64_t)act_step_stride;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((h->B + BLOCK / 2 - 1) / (BLOCK / 2));     // 32 envs per wave, 128 per block
    NIG_DISPATCH_ENV(h->env, launch_mlp<E>(q, grid, st));
    HIP_TRY(hipGetLastError());
    h->t += (uint32_t)n_steps;
    return NIG_OK;
}

// ---- host-buffer entry points (small batches) -------------------------------------------------
// staging layout (same on host and device): [actions f32 A*B][noise f64 K*B] | [state f32 S*B][reward64 f64 B][flags u32 B]
// (uploads = the first part, downloads = the second part: one memcpy each way)
struct HostStage { size_t off_act, off_noise, off_state, off_rew, off_flags, bytes; };

static HostStage host_stage_layout(const nig_handle *h)
{
    const nig_env_spec &sp = SPECS[h->env];
    const size_t B = (size_t)h->B;
    const size_t K = (size_t)(sp.k_step > sp.k_reset ? sp.k_step : sp.k_reset);
    HostStage L;
    L.off_act = 0;
    L.off_noise = (size_t)align_up((int64_t)(sp.action_dim * B * 4), 256);
    L.off_state = L.off_noise + (size_t)align_up((int64_t)(K * B * 8), 256);
    L.off_rew = L.off_state + (size_t)align_up((int64_t)(sp.state_dim * B * 4), 256);
    L.off_flags = L.off_rew + (size_t)align_up((int64_t)(B * 8), 256);
    L.bytes = L.off_flags + (size_t)align_up((int64_t)(B * 4), 256);
    return L;
}

constexpr int64_t HOST_ZERO_COPY_MAX = 1024;   // lanes up to which the host-buffer entry points skip the staging copies

static int host_stage_ensure(nig_handle *h, const HostStage &L)
{
    if (h->hst_pinned && h->hst_bytes >= L.bytes) return NIG_OK;
    if (h->B > 65536) return fail(NIG_ERR_UNSUPPORTED, "host-buffer entry points are for small batches (<= 65536 lanes)%s");
    if (h->hst_dev) { (void)hipFree(h->hst_dev); h->hst_dev = nullptr; }
    if (h->hst_pinned) { (void)hipHostFree(h->hst_pinned); h->hst_pinned = nullptr; }
    HIP_TRY(hipHostMalloc((void **)&h->hst_pinned, L.bytes, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&h->hst_dev, L.bytes));
    h->hst_bytes = L.bytes;
    return NIG_OK;
}

int nig_reset_host(nig_handle *h, const double *init_noise, float *state_out, void *stream)
{
    if (!h || !state_out) return fail(NIG_ERR_INVALID, "nig_reset_host: NULL argument%s");
    const nig_env_spec &sp = SPECS[h->env];
    const HostStage L = host_stage_layout(h);
    int rc = host_stage_ensure(h, L);
    if (rc != NIG_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t B = (size_t)h->B;
    const double *dn = nullptr;
    if (init_noise && sp.k_reset > 0) {
        memcpy(h->hst_pinned + L.off_noise, init_noise, (size_t)sp.k_reset * B * 8);
        HIP_TRY(hipMemcpyAsync(h->hst_dev + L.off_noise, h->hst_pinned + L.off_noise, (size_t)sp.k_reset * B * 8, hipMemcpyHostToDevice, st));
        dn = (const double *)(h->hst_dev + L.off_noise);
    }
    rc = nig_reset(h, nullptr, dn, (int64_t)B, stream);
    if (rc != NIG_OK) return rc;
    // gather the rows (ld apart on the device) into the contiguous staging image, then ONE download
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, st, (const float *)h->state, h->ld_state,
                       (float *)(h->hst_dev + L.off_state), (int64_t)B, sp.state_dim, h->B);
    HIP_TRY(hipMemcpyAsync(h->hst_pinned + L.off_state, h->hst_dev + L.off_state, (size_t)sp.state_dim * B * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(state_out, h->hst_pinned + L.off_state, (size_t)sp.state_dim * B * 4);
    return NIG_OK;
}

int nig_step_host(nig_handle *h, const float *actions, const double *step_noise, float *state_out, double *reward64_out,
                  uint32_t *flags_out, void *stream)
{
    if (!h || !actions || !state_out || !reward64_out || !flags_out) return fail(NIG_ERR_INVALID, "nig_step_host: NULL argument%s");
    const nig_env_spec &sp = SPECS[h->env];
    const HostStage L = host_stage_layout(h);
    int rc = host_stage_ensure(h, L);
    if (rc != NIG_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t B = (size_t)h->B;
    memcpy(h->hst_pinned + L.off_act, actions, (size_t)sp.action_dim * B * 4);
    size_t up = (size_t)sp.action_dim * B * 4;
    const double *dn = nullptr;
    if (step_noise && sp.k_step > 0) {
        memcpy(h->hst_pinned + L.off_noise, step_noise, (size_t)sp.k_step * B * 8);
        up = L.off_noise + (size_t)sp.k_step * B * 8;          // one contiguous upload covers both
        dn = (const double *)(h->hst_dev + L.off_noise);
    }
    // Small batches (the single-env drop-in class is B = 1): the kernels read the actions / noise from, and
    // write state / reward / flags to, the pinned staging buffer itself -- hipHostMalloc memory is mapped
    // and coherent, a few hundred bytes over PCIe cost less than two copy commands -- so a step is two
    // kernel launches and one stream sync.  Larger batches keep one upload + one download.
    const bool zero_copy = h->B <= HOST_ZERO_COPY_MAX;
    char *io = zero_copy ? h->hst_pinned : h->hst_dev;
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(h->hst_dev, h->hst_pinned, up, hipMemcpyHostToDevice, st));
    if (dn) dn = (const double *)(io + L.off_noise);
    rc = nig_step(h, (const float *)(io + L.off_act), (int64_t)B, dn, nullptr, (int64_t)B, nullptr,
                  (double *)(io + L.off_rew), (uint32_t *)(io + L.off_flags), nullptr, 0, stream);
    if (rc != NIG_OK) return rc;
    // state rows gathered next to reward64 and flags: one download for everything the call returns
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, st, (const float *)h->state, h->ld_state,
                       (float *)(io + L.off_state), (int64_t)B, sp.state_dim, h->B);
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(h->hst_pinned + L.off_state, h->hst_dev + L.off_state, L.bytes - L.off_state, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(state_out, h->hst_pinned + L.off_state, (size_t)sp.state_dim * B * 4);
    memcpy(reward64_out, h->hst_pinned + L.off_rew, B * 8);
    memcpy(flags_out, h->hst_pinned + L.off_flags, B * 4);
    return NIG_OK;
}

int nig_plan_create(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act, int64_t slot_stride,
                    int32_t ring_len, float *reward_out, uint32_t *flags_out, int64_t out_stride, nig_plan **out)
{
    if (!out) return fail(NIG_ERR_INVALID, "nig_plan_create: out is NULL%s");
    *out = nullptr;
    if (!h || !action_ring || n_steps <= 0 || ring_len <= 0 || ld_act < h->B || ld_act > NIG_MAX_PITCH)
        return fail(NIG_ERR_INVALID, "nig_plan_create: bad argument%s");
    if (slot_stride < (int64_t)SPECS[h->env].action_dim * ld_act)
        return fail(NIG_ERR_INVALID, "nig_plan_create: slot_stride smaller than one [A][ld_act] slot%s");
    if (out_stride != 0 && out_stride < h->B) return fail(NIG_ERR_INVALID, "nig_plan_create: out_stride < batch%s");
    nig_plan *p = new (std::nothrow) nig_plan();
    if (!p) return fail(NIG_ERR_INVALID, "nig_plan_create: out of host memory%s");
    p->h = h; p->n_steps = n_steps; p->graph = nullptr; p->exec = nullptr;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t cs = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) { delete p; return fail(NIG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed);
    if (e == hipSuccess) {
        for (int k = 0; k < n_steps; ++k) {
            StepArgs a = base_step_args(h);
            const int slot = k % ring_len;
            a.actions = action_ring + (int64_t)slot * slot_stride; a.ld_act = (uint32_t)ld_act;
            a.reward = reward_out ? reward_out + (int64_t)slot * out_stride : nullptr;
            a.flags = flags_out ? flags_out + (int64_t)slot * out_stride : nullptr;
            a.t_ptr = h->t_dev; a.t_off = (uint32_t)(k + 1);
            dispatch_step(h, a, false, cs);
        }
        e = hipStreamEndCapture(cs, &p->graph);
    }
    if (e == hipSuccess) e = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
    (void)hipStreamDestroy(cs);
    if (e != hipSuccess) {
        if (p->graph) (void)hipGraphDestroy(p->graph);
        delete p;
        return fail(NIG_ERR_HIP, "nig_plan_create: graph capture failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return NIG_OK;
}

int nig_plan_launch(nig_plan *p, void *stream)
{
    if (!p) return fail(NIG_ERR_INVALID, "nig_plan_launch: NULL plan%s");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, p->h->t_dev, p->h->t);
    HIP_TRY(hipGraphLaunch(p->exec, st));
    p->h->t += (uint32_t)p->n_steps;
    return NIG_OK;
}

int nig_plan_destroy(nig_plan *p)
{
    if (!p) return NIG_OK;
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    delete p;
    return NIG_OK;
}

int nig_fill_actions(nig_handle *h, uint32_t t, float *actions, int64_t ld_act, void *stream)
{
    if (!h || !actions || ld_act < h->B) return fail(NIG_ERR_INVALID, "nig_fill_actions: bad argument%s");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t lo = (uint32_t)h->seed, hi = (uint32_t)(h->seed >> 32);
    NIG_DISPATCH_ENV(h->env, hipLaunchKernelGGL((fill_actions_kernel<E>), dim3(grid_for(h->B)), dim3(BLOCK), 0, st, actions, ld_act, h->B, h->env0, lo, hi, t));
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_set_state(nig_handle *h, const float *state, int64_t ld, const uint32_t *ctr, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_set_state: NULL handle%s");
    hipStream_t st = (hipStream_t)stream;
    const nig_layout &L = h->lay;
    if (state) {
        if (ld < h->B) return fail(NIG_ERR_INVALID, "nig_set_state: ld < batch%s");
        hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, st, state, ld,
                           h->state, h->ld_state, SPECS[h->env].state_dim, h->B);
        HIP_TRY(hipGetLastError());
    }
    if (ctr) HIP_TRY(hipMemcpyAsync(h->ws + L.off_ctr, ctr, (size_t)h->B * 4, hipMemcpyDeviceToDevice, st));
    return NIG_OK;
}

int nig_get_state(nig_handle *h, float *state, int64_t ld, uint32_t *ctr, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_get_state: NULL handle%s");
    hipStream_t st = (hipStream_t)stream;
    const nig_layout &L = h->lay;
    if (state) {
        if (ld < h->B) return fail(NIG_ERR_INVALID, "nig_get_state: ld < batch%s");
        hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, st,
                           (const float *)h->state, h->ld_state, state, ld, SPECS[h->env].state_dim, h->B);
        HIP_TRY(hipGetLastError());
    }
    if (ctr) HIP_TRY(hipMemcpyAsync(ctr, h->ws + L.off_ctr, (size_t)h->B * 4, hipMemcpyDeviceToDevice, st));
    return NIG_OK;
}

int nig_get_safety_metrics(nig_handle *h, const uint32_t *flags, int32_t *out, int64_t ld_out, void *stream)
{
    if (!h || !flags || !out || ld_out < h->B) return fail(NIG_ERR_INVALID, "nig_get_safety_metrics: bad argument%s");
    hipLaunchKernelGGL(safety_metrics_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, (hipStream_t)stream, flags, out, ld_out, h->B);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_reduce_tally(nig_handle *h, double *partial_out, void *stream)
{
    if (!h || !partial_out) return fail(NIG_ERR_INVALID, "nig_reduce_tally: NULL argument%s");
    if (h->lay.off_tally < 0) return fail(NIG_ERR_INVALID, "nig_reduce_tally: handle created without NIG_F_TALLY%s");
    hipStream_t st = (hipStream_t)stream;
    int nblk = (int)((h->B + BLOCK - 1) / BLOCK);
    if (nblk > REDUCE_BLOCKS) nblk = REDUCE_BLOCKS;
    hipLaunchKernelGGL(reduce_tally_stage1, dim3(nblk), dim3(BLOCK), 0, st, (const double *)(h->ws + h->lay.off_tally),
                       h->lay.ld, h->B, h->scratch);
    hipLaunchKernelGGL(reduce_tally_stage2, dim3(1), dim3(64), 0, st, (const double *)h->scratch, nblk, partial_out);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

}  // extern "C"
