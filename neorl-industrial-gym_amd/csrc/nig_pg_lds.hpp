// nig_pg_lds.hpp -- PowerGrid-v0's fused rollout with the per-timestep state staged in LDS (included by nig_kernels.hpp).
//
// BASELINE.json configs 3 and 5 (262 144 PowerGrid lanes per GPU).  rollout_kernel<PowerGrid> keeps a lane's 32 state
// values in registers and draws the step's 23 normals into 23 more; with the action prefetch ring and the episode
// tally that is ~185 registers = TWO waves per SIMD, and the SQ counters of round 2 showed the price: 4.9 cycles per
// vector instruction where the instruction mix allows ~3.5 (profiles/r02/pg262144_rollout_full_sq.txt), 22 % of the
// wave cycles waiting.  Here the state lives in a wave-private LDS image between steps:
//
//   image: float4 [8 lanes-of-8][8 groups][8 lanes] per wave (8 KiB) -- float4 group g (state values 4g .. 4g+3) of lane
//   l at float4 index (l >> 3) * 64 + g * 8 + (l & 7): eight consecutive lanes are 128 contiguous bytes, so a lane's
//   own-group ds_read_b128 / ds_write_b128 is conflict-free and the group is an immediate offset (no address math);
//
// and a step streams through it: read the groups a phase needs, update, write back.  PowerGrid's step is made for
// this -- V, load and line-flow rows are independent random walks (power_grid.py:136-144), each touched once; only
// the frequency couples the generation and load sums (:127-133).  Never more than ~28 state values and four normals
// are live, the kernel is compiled for 128 registers, and with 512-thread blocks (the 12 KiB generator table shared by
// eight waves: 12 + 8 x 8 KiB per block) two blocks = FOUR waves per SIMD are resident.
//   * The image is also the transposing image of the row-major trajectory: the wave's 64 x 32 block leaves as eight
//     1 KiB whole-line streaming stores read straight from it (the register kernel wrote the state to LDS every
//     step for that anyway -- as 128-byte-strided ds_write_b128: eight lanes on the same four banks, 8-way conflicts).
//   * The cooperative reset writes the initial values into the finishing lanes' image rows and nothing reads them back.
// Arithmetic: the same expressions in the same order on the same values as PowerGrid::violated / dynamics (float32
// noise branch) / reward / done and post_core, the same generator keys and words as draw_step / draw_init --
// bit-identical to rollout_kernel<PowerGrid> and to the oracle (tests/test_gpu_round3.py, tests/test_gpu_parity.py).
// Whole 512-lane blocks of an auto-reset handle without frozen lanes only; the host keeps every other case on
// rollout_kernel.
#pragma once

namespace nig {

template <int BLK>
struct PgLds {
    static constexpr int NWAVE = BLK / 64;
    // the generator's table at LDS byte 16 x PROBIT_BIAS (3 072), so that probit_fetch's piece number needs no subtraction: the
    // bias rides in the DS instruction's offset field (nig_detmath.hpp); the reset's work lists live in the bytes below it
    static constexpr int OFF_WLIST = 0;                          // uchar [NWAVE][64]
    static constexpr int OFF_PROBIT = 16 * PROBIT_BIAS;
    static_assert(NWAVE * 64 <= OFF_PROBIT, "work lists below the table");
    static constexpr int OFF_IMG = OFF_PROBIT + 768 * 16;        // float4 [NWAVE][512]
    static constexpr int BYTES = OFF_IMG + NWAVE * 8192;
};

// The PAIRED form (rollout_pg_pair_kernel below; batches of at most one 256-lane block per compute unit, where the body
// above would leave a single wave on every SIMD): the block carries a second set of four waves, the PRODUCERS.  Producer w
// draws the step's 23 normals for the 64 lanes of stepping wave w -- six Philox blocks and 23 table transforms, ~510 of the
// step's ~1 090 vector instructions, all a function of the lane's generator key alone -- one or two steps ahead into a
// two-slot ring in LDS; the stepping wave reads them as six float4 per step where it used to compute them.  The two waves
// of a pair share a SIMD (a block's waves go round the four SIMDs), so the SIMD issues from two instruction streams
// instead of one.  Ring protocol, counters and fences: nig_ring.hpp.  The normals pass through LDS as the floats they
// are: same bits as the one-wave form.  65 536 lanes x 250 steps: 822 -> 688 us with full outputs, 810 -> 655 us with
// reward + flags (profiles/r03/pg_pair_ab.txt); the stepping wave is what bounds it then (a chain of LDS round trips
// through its state image: ~10 cycles per instruction) -- TWO producers per stepping wave, three generator blocks each,
// were no faster (711 / 676 us).
// NOISE (nig_rollout_noise: the reference's recorded float64 draws instead of the generator's): a slot holds the step's 23
// draws as doubles, [draw][lane], because power_grid.py:136-144 adds them in float64.
// POLICY (rollout_pg_pair_policy_kernel: the closed loop, nig_rollout_policy): a slot also carries the on-device policy's own
// random draws of the step -- exploration normals z[8], uniform perturbations h[8], the epsilon-mix's uniform action ra[8]
// and its mixing draw: 25 float rows -- and the block keeps a copy of the policy struct.
template <bool NOISE, bool POLICY = false>
struct PgPairLdsT {
    static constexpr int K = 2;                                   // ring slots (steps) per pair
    static constexpr int DRAW_ROWS = 25;                          // POLICY: z[8], h[8], ra[8], wmix
    static constexpr int NZ_V4 = NOISE ? 23 * 64 / 2 : 6 * 64;    // float4 of a slot's process noise: [generator block][lane] (NOISE: 23 x 64 doubles)
    static constexpr int SLOT_V4 = NZ_V4 + (POLICY ? DRAW_ROWS * 16 : 0);
    static constexpr int OFF_POL = (PgLds<256>::BYTES + 15) / 16 * 16;
    // POLICY: behind the policy struct, a dense 16-byte-aligned copy of its feedback matrix, [32 state columns][8 actions]
    // (the struct's rows are 40 bytes apart: no ds_read_b128 there)
    static constexpr int OFF_WD = OFF_POL + (POLICY ? (int)((sizeof(nig_policy) + 15) / 16 * 16) : 0);
    static constexpr int OFF_NZ = OFF_WD + (POLICY ? 32 * 8 * 4 : 0);
    static constexpr int OFF_SYNC = OFF_NZ + 4 * K * SLOT_V4 * 16;     // uint32 [4 pairs][4]: {produced, consumed}
    static constexpr int BYTES = OFF_SYNC + 4 * 16;
    static_assert(BYTES <= 160 * 1024, "LDS of one CU");
};
using PgPairLds = PgPairLdsT<false>;

// producer wave `wave` (0-3) of the block: lanes base + 64 wave .. + 63, local steps [0, n)
template <class QA> __device__ __forceinline__ int arg_it0(const QA &q)
{
    if constexpr (std::is_same<QA, RolloutArgs>::value) return q.it0; else return 0;     // (closed-loop launches start at call step 0)
}

// The paired form with the stepping waves' state in REGISTERS (rollout_body's RING form; round 4): the LDS of a block = the
// register-resident rollout's own layout (generator table, per-wave reset / transpose scratch), then the producers' ring.
template <int OUT>
struct PgPairRegLds {
    static constexpr int K = 2;
    static constexpr int NZ_V4 = 6 * 64, SLOT_V4 = NZ_V4;
    static constexpr int OFF_POL = 0;                                  // (unused: open loop)
    static constexpr int OFF_NZ = (RolloutLds<PowerGrid, OUT, 256>::BYTES + 15) / 16 * 16;
    static constexpr int OFF_SYNC = OFF_NZ + 4 * K * SLOT_V4 * 16;
    static constexpr int BYTES = OFF_SYNC + 4 * 16;
    static_assert(BYTES <= 160 * 1024, "LDS of one CU");
};

template <bool NOISE = false, bool POLICY = false, class QA = RolloutArgs, class LL = PgPairLdsT<NOISE, POLICY>>
__device__ __forceinline__ void pg_pair_producer(const QA &q, const uint32_t base, unsigned char *smem, const unsigned wave, const unsigned lane)
{
    using L = LL;
    static_assert(!(NOISE && POLICY), "recorded draws exist for the open loop only");
    [[maybe_unused]] const nig_policy *const pol = reinterpret_cast<const nig_policy *>(smem + L::OFF_POL);
    const float4 *const s_probit = reinterpret_cast<const float4 *>(smem + PgLds<256>::OFF_PROBIT);
    v4f *const ring = reinterpret_cast<v4f *>(smem + L::OFF_NZ) + wave * (L::K * L::SLOT_V4);
    lds_u32_t *const sync = (lds_u32_t *)(smem + L::OFF_SYNC) + wave * 4;
    const StepArgs &p = q.s;
    const uint32_t t_base = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off + (uint32_t)arg_it0(q);      // local step i uses t_base + i + 1
    const uint64_t gi = p.env0 + (uint64_t)(base + wave * 64u + lane);
    const int n = q.n_steps - arg_it0(q);
    uint32_t freed = 0u;                                           // slots the stepping wave is known to be done with
    [[maybe_unused]] bool any_sigma = false, any_half = false, mix = false;    // wave-uniform switches of the policy
    if constexpr (POLICY) policy_switches<PowerGrid::A>(pol, any_sigma, any_half, mix);
    if constexpr (NOISE) {                         // the step's recorded draws, loaded where the generator would have produced them
        double *const ringd = reinterpret_cast<double *>(ring);
        for (int i = 0; i < n; ++i) {
            const double *nzr = p.step_noise + (size_t)(arg_it0(q) + i) * q.nz_step_stride + base + wave * 64u;
            double z[23];
#pragma unroll
            for (int k = 0; k < 23; ++k) z[k] = (nzr + (size_t)k * p.ld_noise)[lane];
            if (freed + (uint32_t)L::K < (uint32_t)i + 1u) freed = split_wait(sync + 1, (uint32_t)(i + 1 - L::K));
            double *slot = ringd + (i & (L::K - 1)) * (L::SLOT_V4 * 2);
#pragma unroll
            for (int k = 0; k < 23; ++k) slot[64 * k + lane] = z[k];
            NIG_RING_FAULT_GUARD(p.hflags, i) split_post(sync + 0, (uint32_t)i + 1u, lane);
        }
        NIG_RING_REPORT(p.ring_err, sync, lane);
        return;
    }
    for (int i = 0; i < n; ++i) {
        const RngKey key = make_key(gi, t_base + (uint32_t)i + 1u, p.seed_lo, p.seed_hi, s_probit);
        v4f z[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {                              // draw_step: blocks 0-5 of the step stream, words in order
            const u32x4 x = key.block(STREAM_STEP + (uint32_t)j);
            const ProbitFetch f0 = probit_fetch(x.x, s_probit), f1 = probit_fetch(x.y, s_probit), f2 = probit_fetch(x.z, s_probit);
            v4f w = {probit_eval(f0), probit_eval(f1), probit_eval(f2), 0.0f};
            if (j < 5) w.w = probit_eval(probit_fetch(x.w, s_probit));     // z[23] does not exist
            z[j] = w;
        }
        [[maybe_unused]] PolicyDraws<PowerGrid::A> d;
        if constexpr (POLICY) policy_draws<PowerGrid>(pol, key, d);      // the policy's own draws of this step: a function of the key alone
        if (freed + (uint32_t)L::K < (uint32_t)i + 1u) freed = split_wait(sync + 1, (uint32_t)(i + 1 - L::K));   // the slot's previous use
        v4f *slot = ring + (i & (L::K - 1)) * L::SLOT_V4;
#pragma unroll
        for (int j = 0; j < 6; ++j) slot[64 * j + lane] = z[j];
        if constexpr (POLICY) {
            float *dr = reinterpret_cast<float *>(slot + L::NZ_V4) + lane;       // [DRAW_ROWS][64]
            if (any_sigma) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dr[k * 64] = d.z[k];
            }
            if (any_half) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dr[(8 + k) * 64] = d.h[k];
            }
            if (mix) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dr[(16 + k) * 64] = d.ra[k];
                dr[24 * 64] = d.wmix;
            }
        }
        NIG_RING_FAULT_GUARD(p.hflags, i) split_post(sync + 0, (uint32_t)i + 1u, lane);
    }
    NIG_RING_REPORT(p.ring_err, sync, lane);
}

// NOISE: nig_rollout_noise -- the step's 23 draws are the reference's recorded float64 values (loaded from the caller's
// rows, or read from the producer's slot in the paired form) and enter through the float64 adds of power_grid.py:136-144
// (PowerGrid::dynamics' parity branch); a finishing lane's image row is rewritten with PowerGrid::init(recorded draws).
// POLICY (paired form only; QA = PolicyArgs): the closed loop of nig_rollout_policy -- the action is the installed affine
// policy of the lane's observation (policy_affine on the image's pre-step values, ascending state index as everywhere;
// the policy's own draws come from the producer's slot: policy_finish), the optional outputs are those of
// rollout_policy_kernel (the observation acted on, row-major through the image's transposed reads; the policy's action
// rows; reward + flag rows), run-time switches as there.  Bit-identical to rollout_policy_kernel (tests/test_gpu_split.py).
template <int OUT, int BLK, bool PROD = false, bool NOISE = false, bool POLICY = false, class QA = RolloutArgs>
__device__ __forceinline__ void pg_lds_rollout_body(const QA &q, const uint32_t base, unsigned char *smem)
{
    using Env = PowerGrid;
    constexpr int S = Env::S, A = Env::A;
    using Lds = PgLds<BLK>;
    using PgPairLds = PgPairLdsT<NOISE, POLICY>;
    static_assert(!POLICY || (PROD && OUT == 0 && !NOISE), "closed loop: the paired form, outputs as run-time switches");
    [[maybe_unused]] const nig_policy *const pol = reinterpret_cast<const nig_policy *>(smem + PgPairLds::OFF_POL);
    static_assert(!PROD || BLK == 256, "the paired form runs 256-lane blocks");
    float4 *const s_probit = reinterpret_cast<float4 *>(smem + Lds::OFF_PROBIT);
    if constexpr (!PROD) {                         // (paired form: the kernel staged the table with all its waves)
        for (int i_ = (int)threadIdx.x; i_ < 768; i_ += BLK) s_probit[i_] = NIG_PROBIT[i_];
        __syncthreads();
    }
    const StepArgs &p = q.s;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    [[maybe_unused]] const v4f *const nz_ring = reinterpret_cast<const v4f *>(smem + PgPairLds::OFF_NZ) + wave * (PgPairLds::K * PgPairLds::SLOT_V4);
    [[maybe_unused]] lds_u32_t *const nz_sync = (lds_u32_t *)(smem + PgPairLds::OFF_SYNC) + wave * 4;
    [[maybe_unused]] uint32_t nz_seen = 0u;
    v4f *const img = reinterpret_cast<v4f *>(smem + Lds::OFF_IMG) + wave * 512;
    v4f *const mine = img + (lane >> 3) * 64 + (lane & 7u);         // group g of this lane: mine[8 g]
    float *const imgf = reinterpret_cast<float *>(img);
    unsigned char *const wl = smem + Lds::OFF_WLIST + wave * 64;
    const uint32_t t_base = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off;    // step k uses t_base + k + 1
    const uint64_t gi = p.env0 + (uint64_t)(base + tid);
    const uint64_t wave_gi0 = p.env0 + (uint64_t)(base + (tid & ~63u));
    const bool tally = p.tally != nullptr;

    uint32_t ctr = (p.ctr + base)[tid];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float *r = p.state + base + (4 * g) * p.ld_state;
        v4f v = {r[tid], (r + p.ld_state)[tid], (r + 2 * p.ld_state)[tid], (r + 3 * p.ld_state)[tid]};
        mine[8 * g] = v;
    }
    double ret = tally ? (p.ep_ret + base)[tid] : 0.0;
    LaneTally lt;
    lt.clear();

    // the action ring: slot s at p.actions + s * q.slot_stride, each slot [A][ld_act] (one 256-byte row segment per action and
    // wave) -- or, with ld_act == 0 (round 5: the layout agents produce, nig_rollout's row-major form), [B][A]: a lane's eight
    // actions are 32 contiguous bytes, two 16-byte loads, 2 KiB contiguous per wave.  Wave-uniform switch, same values.
    const bool act_aos = p.ld_act == 0u;
    [[maybe_unused]] const float *ring = p.actions + (act_aos ? (size_t)base * A : (size_t)base);
    [[maybe_unused]] auto load_action = [&](const float *slotp, float (&dst)[A]) __attribute__((always_inline)) {
        if (act_aos) {
            const v4f *ap = reinterpret_cast<const v4f *>(slotp) + 2u * tid;
            const v4f w0 = ap[0], w1 = ap[1];
            dst[0] = w0.x; dst[1] = w0.y; dst[2] = w0.z; dst[3] = w0.w; dst[4] = w1.x; dst[5] = w1.y; dst[6] = w1.z; dst[7] = w1.w;
        } else {
#pragma unroll
            for (int k = 0; k < A; ++k) dst[k] = (slotp + k * p.ld_act)[tid];
        }
    };
    static_assert(A == 8, "row-major action slots are read as two float4 per lane");
    // ONE action register set: the action of step it + 1 is loaded into it as soon as step it has consumed its own
    // (clip, generation update, the reward's action term: the first ~80 instructions of a step), i.e. a whole step
    // (~1 200 instructions, four waves sharing the SIMD) before it is used.  The wait for it is in order with the stores
    // issued before it -- those of step it - 1.  Round 5 asked whether that in-order wait is what the action reads cost
    // with full outputs (a build without the loads ran 1.96-1.98 ms per 250 steps against 2.11-2.27): it is NOT.  Two
    // register sets with the copy pinned at the end of the step (the stores a load queues behind are then two steps old)
    // changed nothing (2 270 / 2 274 vs 2 275 / 2 244 us), loads that go out but are never waited for are no faster than
    // the real ones (2 150 vs 2 110 us), and an action ring that fits the 256 MB Infinity Cache gives the no-load time
    // (1 976 us): what costs is READ TRAFFIC REACHING HBM inside a 4.5 TB/s write stream, not the wave's wait
    // (profiles/r05/pg_ab_s2_two_action_sets.txt, pg_ab_s3_action_read_cost.txt).
    constexpr int DEPTH = 1;
    float buf[DEPTH][A];
    [[maybe_unused]] int slot = 0;
    if constexpr (!POLICY) slot = q.it0 % q.ring_len;
    [[maybe_unused]] const float *act_next = nullptr;
    const int it0 = arg_it0(q);
    float *rew_row = p.reward ? p.reward + base + (size_t)it0 * q.out_stride : nullptr;
    uint32_t *fl_row = p.flags ? p.flags + base + (size_t)it0 * q.out_stride : nullptr;
    float *obs_row = nullptr;
    // (the wave's first lane through readfirstlane: the row pointer is wave-uniform, and only then does the compiler keep it
    // in scalar registers; see rollout_body)
    if constexpr (OUT == 3)
        obs_row = q.obs_out + (size_t)q.it0 * q.obs_step_stride + (size_t)(base + __builtin_amdgcn_readfirstlane(tid & ~63u)) * S;
    if constexpr (OUT == 2) obs_row = q.obs_out + (size_t)q.it0 * q.obs_step_stride + base;
    if constexpr (POLICY) {                        // the observation the policy acts on, row-major [n_steps][B][S] (optional)
        if (q.obs_out) obs_row = q.obs_out + (size_t)(base + __builtin_amdgcn_readfirstlane(tid & ~63u)) * S;
    }
    [[maybe_unused]] float *act_row = nullptr;     // POLICY: the policy's action rows [n_steps][A][ld] (optional)
    if constexpr (POLICY) { if (q.act_out) act_row = q.act_out + base; }
    [[maybe_unused]] bool any_sigma = false, any_half = false, mix = false;
    [[maybe_unused]] PolicyHead<A> head;
    [[maybe_unused]] const v4f *const wd = reinterpret_cast<const v4f *>(smem + PgPairLds::OFF_WD);
    if constexpr (POLICY) { policy_switches<A>(pol, any_sigma, any_half, mix); head.load(*pol); }
    const unsigned rd = (lane & 7u) * 8u + (lane >> 3);             // transposed read: image float4 64 j + rd = row-major float4 64 j + lane

    auto one_step = [&](float (&abuf)[A], const int it) __attribute__((always_inline)) {
        float a[A];
        [[maybe_unused]] const int itl0 = it - it0;                  // local step: the producer's slot index
        [[maybe_unused]] float obs[S];                               // POLICY: the pre-step values the policy acts on (and the step then uses)
        if constexpr (POLICY) {
            // ---- action = policy(observation): the feedback law on the image's pre-step values, then the producer's draws
#pragma unroll
            for (int g = 0; g < 8; ++g) { const v4f w = mine[8 * g]; obs[4 * g] = w.x; obs[4 * g + 1] = w.y; obs[4 * g + 2] = w.z; obs[4 * g + 3] = w.w; }
            if (obs_row != nullptr) {              // (wave-uniform) the 64 observation rows in lane-contiguous order
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                v4f *oo = reinterpret_cast<v4f *>(obs_row);
                v4f tv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) tv[j] = img[64 * j + rd];
#pragma unroll
                for (int j = 0; j < 8; ++j) stream_store(oo + lane + 64u * j, tv[j]);
                obs_row += q.obs_step_stride;
            }
            policy_affine_dense<Env>(head, wd, obs, a);
            if (nz_seen < (uint32_t)itl0 + 1u) nz_seen = split_wait(nz_sync + 0, (uint32_t)itl0 + 1u);
            const float *dr = reinterpret_cast<const float *>(nz_ring + (itl0 & (PgPairLds::K - 1)) * PgPairLds::SLOT_V4 + PgPairLds::NZ_V4) + lane;
            PolicyDraws<A> d;
            if (any_sigma) {
#pragma unroll
                for (int k = 0; k < A; ++k) d.z[k] = dr[k * 64];
            }
            if (any_half) {
#pragma unroll
                for (int k = 0; k < A; ++k) d.h[k] = dr[(8 + k) * 64];
            }
            if (mix) {
#pragma unroll
                for (int k = 0; k < A; ++k) d.ra[k] = dr[(16 + k) * 64];
                d.wmix = dr[24 * 64];
            }
            policy_finish<Env>(&head, d, a);
            if (act_row != nullptr) {              // the policy's action, before the env's own clip (as rollout_policy_kernel)
#pragma unroll
                for (int j = 0; j < A; ++j) stream_store(act_row + j * q.ld_act_out + tid, a[j]);
                act_row += q.act_step_stride;
            }
        } else {
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = abuf[k];
        }
        clip_action<Env, float>(a);
        const RngKey key = make_key(gi, t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi, s_probit);
        // ---- generation, load sum, frequency (state values 0, 9 .. 24) --------------------------------------------
        float v[8], ngen7;                         // v: the pre-step voltages, updated in place below
        float n0, fr, ap;
        double er;
        uint32_t vb;
        {
            float s[S];                            // pre-step values, as far as violated() and the dynamics read them
            if constexpr (POLICY) {                // (already read for the policy)
#pragma unroll
                for (int k = 0; k < 25; ++k) s[k] = obs[k];
            } else {
            const v4f g0 = mine[0], g1 = mine[8], g2 = mine[16], g3 = mine[24], g4 = mine[32], g5 = mine[40], g6 = mine[48];
            s[0] = g0.x;
            s[1] = g0.y; s[2] = g0.z; s[3] = g0.w; s[4] = g1.x; s[5] = g1.y; s[6] = g1.z; s[7] = g1.w; s[8] = g2.x;
            s[9] = g2.y; s[10] = g2.z; s[11] = g2.w; s[12] = g3.x; s[13] = g3.y; s[14] = g3.z; s[15] = g3.w; s[16] = g4.x;
            s[17] = g4.y; s[18] = g4.z; s[19] = g4.w; s[20] = g5.x; s[21] = g5.y; s[22] = g5.z; s[23] = g5.w; s[24] = g6.x;
            }
#pragma unroll
            for (int k = 25; k < S; ++k) s[k] = 0.0f;      // line flows: read by nothing before their own update
            vb = Env::violated(s, a) & p.cmask;            // base.py:170 on the pre-state, clipped action
            float ngen[8], load[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float g = s[9 + i] + a[i];                                   // power_grid.py:124 np.clip(gen + a, 0, 100)
                // lower bound: np.maximum(g, 0) keeps a NaN and keeps -0.0 (it returns its first argument on a tie).  The generator-
                // driven path takes v_maximum3_f32 (IEEE-754-2019 maximum: keeps the NaN, returns +0.0 for -0.0) -- one instruction
                // for compare + select; it differs from NumPy only on an input of exactly -0.0, which this arithmetic cannot
                // produce from non-negative generation and a finite action (x + y = -0.0 needs both -0.0) -- only nig_set_state
                // can plant one, and the difference is the sign of a zero in that step's stored value.  Recorded-draw launches
                // (NOISE: the parity path through this body) keep the compare + select.
                if constexpr (NOISE) g = (g < 0.0f) ? 0.0f : g;
                else g = __builtin_elementwise_maximum(g, 0.0f);
                g = __builtin_elementwise_minimum(g, 100.0f);      // np.minimum incl. NaN: one v_minimum3_f32 (nig_envs.hpp dynamics)
                ngen[i] = g;
                load[i] = s[17 + i];
                v[i] = s[1 + i];
            }
            const float imb = sum8(ngen) - sum8(load);                       // :127-129
            const float fd = fdiv_c((-1.0f * s[0]) + imb, 5.0f);             // :132
            n0 = s[0] + fd * p.dt32;                                         // :133
            // the reward's terms that do not wait for the voltages (:162, :169-173): the action and all of the new
            // generation but three values are dead from here on
            fr = Env::reward_freq(n0);
            er = Env::reward_econ(ngen);
            ap = Env::reward_act(a);
            v4f w3 = {ngen[3], ngen[4], ngen[5], ngen[6]};
            mine[24] = w3;
            if constexpr (OUT == 2) {
                stream_store(obs_row + tid, n0);
#pragma unroll
                for (int k = 0; k < 8; ++k) stream_store(obs_row + (9 + k) * q.ld_obs_out + tid, ngen[k]);
            }
            ngen7 = ngen[7];
            // refill this action register set (step it + DEPTH), issued before the step's stores: see rollout_body
            if constexpr (!POLICY) {
#ifdef NIG_DIAG_PG_NOACTLOAD           // (diagnostic builds only, profiles/r05: no global load in the loop -- the action is a cheap hash of lane
            // and step instead; what do the action reads cost?)
#pragma unroll
            for (int k = 0; k < A; ++k)
                abuf[k] = (float)((((uint32_t)gi + 0x9E3779B9u * (uint32_t)(it + k)) * 2654435761u) >> 8) * (1.0f / 8388608.0f) - 1.0f;
#elif defined(NIG_DIAG_PG_ROWMAJOR_LOADS)  // (diagnostic: the slot's [A][ld] block read as if it were [ld][A] -- two 16-byte loads per lane, 2 KiB
            // contiguous per wave instead of eight 256-byte row segments; same bytes, same footprint, and since every entry is an
            // independent uniform draw, the same workload statistically: -1 % with full outputs, profiles/r05/pg_ab_s4_rowmajor_ntload.txt)
            {
                const v4f *ap = reinterpret_cast<const v4f *>((act_next - base) + (size_t)(base + tid) * 8u);
                const v4f w0 = ap[0], w1 = ap[1];
                abuf[0] = w0.x; abuf[1] = w0.y; abuf[2] = w0.z; abuf[3] = w0.w;
                abuf[4] = w1.x; abuf[5] = w1.y; abuf[6] = w1.z; abuf[7] = w1.w;
            }
            slot = (slot + 1 == q.ring_len) ? 0 : slot + 1;
            act_next = (slot == 0) ? ring : act_next + q.slot_stride;
#else
            load_action(act_next, abuf);
            slot = (slot + 1 == q.ring_len) ? 0 : slot + 1;
            act_next = (slot == 0) ? ring : act_next + q.slot_stride;
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- voltages: generator blocks 0 and 1 of the step stream (draw_step: z[0..7], sd 0.005) ---------------
            [[maybe_unused]] const int itl = it - it0;                   // local step: the producer's slot index
            [[maybe_unused]] const v4f *const nzs = nz_ring + (itl & (PgPairLds::K - 1)) * PgPairLds::SLOT_V4;
            u32x4 x = {0u, 0u, 0u, 0u};
            // NOISE: draw k of this step, float64 -- from the producer's slot (paired form) or the caller's rows
            [[maybe_unused]] const double *const nzd_lds = reinterpret_cast<const double *>(nzs) + lane;
            [[maybe_unused]] const double *nzd_glb = nullptr;
            if constexpr (NOISE) nzd_glb = p.step_noise + (size_t)it * q.nz_step_stride + base + tid;
            [[maybe_unused]] auto nzd = [&](int k) __attribute__((always_inline)) -> double {
                if constexpr (PROD) return nzd_lds[64 * k];
                else return (nzd_glb + (size_t)k * p.ld_noise)[0];
            };
            if constexpr (NOISE) {
                if constexpr (PROD) {
                    if (nz_seen < (uint32_t)itl + 1u) nz_seen = split_wait(nz_sync + 0, (uint32_t)itl + 1u);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (float)((double)v[k] + nzd(k));      // :136-137 fp64 add, one rounding
            } else if constexpr (PROD) {
                if (nz_seen < (uint32_t)itl + 1u) nz_seen = split_wait(nz_sync + 0, (uint32_t)itl + 1u);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const v4f zq = nzs[64 * j + lane];
                    v[4 * j + 0] = v[4 * j + 0] + 0.005f * zq.x; v[4 * j + 1] = v[4 * j + 1] + 0.005f * zq.y;   // :136-137
                    v[4 * j + 2] = v[4 * j + 2] + 0.005f * zq.z; v[4 * j + 3] = v[4 * j + 3] + 0.005f * zq.w;
                }
            } else {
            x = key.block(STREAM_STEP);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ProbitFetch f[4];
                f[0] = probit_fetch(x.x, s_probit); f[1] = probit_fetch(x.y, s_probit);
                f[2] = probit_fetch(x.z, s_probit); f[3] = probit_fetch(x.w, s_probit);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[4 * j + c] = v[4 * j + c] + 0.005f * probit_eval(f[c]);   // :136-137
                __builtin_amdgcn_sched_barrier(0);
                // (here the next block's rounds come AFTER the cubics: with the frequency / reward terms still live, rounds
                // running over four table reads in flight cost the six registers that pushed the episode tally into
                // scratch -- and a scratch reload drains vmcnt, i.e. every streaming store of the step.  The other waves
                // of the SIMD cover the table latency; the load / line-flow phase below does overlap.)
                x = key.block(STREAM_STEP + (uint32_t)(j + 1));
            }
            }
            {
                v4f w0 = {n0, v[0], v[1], v[2]}, w1 = {v[3], v[4], v[5], v[6]}, w2 = {v[7], ngen[0], ngen[1], ngen[2]};
                mine[0] = w0; mine[8] = w1; mine[16] = w2;
            }
            if constexpr (OUT == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) stream_store(obs_row + (1 + k) * q.ld_obs_out + tid, v[k]);
            }
            // ---- IndustrialEnv.step after the dynamics (base.py:176-198): reward, penalties, termination ----------
            const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
            StepResult<Env> res;
            post_finish<Env, double>(Env::reward_total(fr, Env::reward_volt(v), er, ap), Env::done_fv(n0, v), vb, step_pre, p.max_steps, res);
            const int step = step_pre + 1;
            const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;
            const bool done = res.terminated || res.truncated;
            ctr = done ? 0u : ((uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT));
            if (tally) ret = ret + res.reward;
            if constexpr (OUT >= 1) {
                stream_store(rew_row + tid, (float)res.reward);
                stream_store(fl_row + tid, pack_flags<Env>(res, step) | (done ? NIG_FLAG_DID_RESET : 0u));
                rew_row += q.out_stride; fl_row += q.out_stride;
            }
            if constexpr (POLICY) {                // run-time switches, as in rollout_policy_kernel (either may be absent)
                if (rew_row) { stream_store(rew_row + tid, (float)res.reward); rew_row += q.out_stride; }
                if (fl_row) { stream_store(fl_row + tid, pack_flags<Env>(res, step) | (done ? NIG_FLAG_DID_RESET : 0u)); fl_row += q.out_stride; }
            }
            if (done) {                            // (lt.viol doubles as the lifetime violation count: two registers less)
                if (tally) { lt.episode(ret, step, viol_ep, res.ncrit); ret = 0.0; }
                else lt.viol += (int)viol_ep;
            }
            // ---- loads: blocks 2 and 3 (z[8..15], sd 1.0), then line flows: blocks 4 and 5 (z[16..22], sd 2.0) ----
            float l7 = 0.0f;
#pragma unroll
            for (int j = 2; j < 6; ++j) {
                ProbitFetch f[4];
                v4f zq = {0.0f, 0.0f, 0.0f, 0.0f};
                [[maybe_unused]] double zd[4] = {0.0, 0.0, 0.0, 0.0};
                if constexpr (NOISE) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (4 * j + c < 23) zd[c] = nzd(4 * j + c);
                    if constexpr (PROD) {
                        if (j == 5) split_post(nz_sync + 1, (uint32_t)itl + 1u, lane);
                    }
                } else if constexpr (PROD) {
                    zq = nzs[64 * j + lane];
                    if (j == 5) split_post(nz_sync + 1, (uint32_t)itl + 1u, lane);     // (DS order: the reads above execute before this write)
                } else {
                f[0] = probit_fetch(x.x, s_probit); f[1] = probit_fetch(x.y, s_probit); f[2] = probit_fetch(x.z, s_probit);
                if (j < 5) f[3] = probit_fetch(x.w, s_probit);               // z[23] does not exist
                if (j < 5) x = key.block(STREAM_STEP + (uint32_t)(j + 1));
                }
                // the step's normals of this phase: transformed here (one-wave form) or read from the producer's slot
                auto zn = [&](int c) __attribute__((always_inline)) -> float {
                    if constexpr (PROD) return c == 0 ? zq.x : c == 1 ? zq.y : c == 2 ? zq.z : zq.w;
                    else return probit_eval(f[c]);
                };
                // one random-walk update: load rows (:140-141, clipped at 0) and line-flow rows (:144, sd 2.0)
                auto walk_load = [&](float h, int c) __attribute__((always_inline)) -> float {
                    if constexpr (NOISE) { double l = (double)h + zd[c]; l = (l < 0.0) ? 0.0 : l; return (float)l; }
                    else return __builtin_elementwise_maximum(h + zn(c), 0.0f);      // (v_maximum3_f32, as the generation clip above: h >= +0, z != 0)
                };
                auto walk_flow = [&](float h, int c) __attribute__((always_inline)) -> float {
                    if constexpr (NOISE) return (float)((double)h + zd[c]);
                    else return __builtin_fmaf(2.0f, zn(c), h);       // == h + 2.0f * z: the product is exact, one rounding either way
                };
                asm volatile("" ::: "memory");                               // the groups are read again HERE, not kept from the top
                __builtin_amdgcn_sched_barrier(0);
                if (j == 2) {                      // load[0..3]: group 4 = {gen7, load0, load1, load2}, load3 = group 5 .x
                    const v4f h4 = mine[32], h5 = mine[40];
                    float l[4] = {walk_load(h4.y, 0), walk_load(h4.z, 1), walk_load(h4.w, 2), walk_load(h5.x, 3)};   // :140-141
                    v4f w4 = {ngen7, l[0], l[1], l[2]};
                    mine[32] = w4;
                    l7 = l[3];                     // load3', parked until its group is complete
                    if constexpr (OUT == 2) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) stream_store(obs_row + (17 + c) * q.ld_obs_out + tid, l[c]);
                    }
                } else if (j == 3) {               // load[4..7]: group 5 = {load3 .. load6}, load7 = group 6 .x
                    const v4f h5 = mine[40], h6 = mine[48];
                    float l[4] = {walk_load(h5.y, 0), walk_load(h5.z, 1), walk_load(h5.w, 2), walk_load(h6.x, 3)};
                    v4f w5 = {l7, l[0], l[1], l[2]};
                    mine[40] = w5;
                    if constexpr (OUT == 2) {
                        stream_store(obs_row + 20 * q.ld_obs_out + tid, l7);
#pragma unroll
                        for (int c = 0; c < 3; ++c) stream_store(obs_row + (21 + c) * q.ld_obs_out + tid, l[c]);
                    }
                    l7 = l[3];                     // load7'
                } else if (j == 4) {               // flows[0..3]: group 6 = {load7, flow0, flow1, flow2}, flow3 = group 7 .x
                    const v4f h6 = mine[48], h7 = mine[56];
                    const float fl[4] = {walk_flow(h6.y, 0), walk_flow(h6.z, 1), walk_flow(h6.w, 2), walk_flow(h7.x, 3)};   // :144
                    v4f w6 = {l7, fl[0], fl[1], fl[2]};
                    mine[48] = w6;
                    if constexpr (OUT == 2) {
                        stream_store(obs_row + 24 * q.ld_obs_out + tid, l7);
#pragma unroll
                        for (int c = 0; c < 3; ++c) stream_store(obs_row + (25 + c) * q.ld_obs_out + tid, fl[c]);
                    }
                    l7 = fl[3];                    // flow3'
                } else {                           // flows[4..6]: group 7 = {flow3 .. flow6}
                    const v4f h7 = mine[56];
                    const float fl[3] = {walk_flow(h7.y, 0), walk_flow(h7.z, 1), walk_flow(h7.w, 2)};
                    v4f w7 = {l7, fl[0], fl[1], fl[2]};
                    mine[56] = w7;
                    if constexpr (OUT == 2) {
                        stream_store(obs_row + 28 * q.ld_obs_out + tid, l7);
#pragma unroll
                        for (int c = 0; c < 3; ++c) stream_store(obs_row + (29 + c) * q.ld_obs_out + tid, fl[c]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (OUT == 3) {
                // the image now holds the wave's 64 post-step rows: row-major float4 64 j + lane sits at 64 j + rd
                v4f *oo = reinterpret_cast<v4f *>(obs_row);
                v4f v[8];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // other lanes' writes are read below (compiler order only: rollout_body)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef NIG_DIAG_PG_NOLDSREAD           // (diagnostic builds only, profiles/r03/pg_store_probe2.sh: what do the transposed reads cost?)
#pragma unroll
                for (int j = 0; j < 8; ++j) { v4f w = {ngen7, l7, (float)j, ngen7}; v[j] = w; }
#else
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = img[64 * j + rd];
#endif
#ifdef NIG_DIAG_PG_NOSTORE             // (diagnostic: the reads without the stores)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(v[j]));
#elif defined(NIG_DIAG_PG_PLAINSTORE)   // (diagnostic: ordinary instead of streaming stores for the trajectory rows)
#pragma unroll
                for (int j = 0; j < 8; ++j) oo[lane + 64u * j] = v[j];
#else
#pragma unroll
                for (int j = 0; j < 8; ++j) stream_store(oo + lane + 64u * j, v[j]);
#endif
            }
            if constexpr (OUT >= 2) obs_row += q.obs_step_stride;
            // ---- IndustrialEnv.reset for the lanes that finished (base.py:133-155), wave-cooperative: work item =
            // (finishing lane, generator block) -> four state values, written into that lane's image row
            if constexpr (NOISE) {
                if (done) {                        // _get_initial_state on the recorded draws of this step's row set, into the lane's own image row
                    double rn[Env::KR];
                    const double *rnr = p.reset_noise + (size_t)it * q.nz_reset_stride + base + tid;
#pragma unroll
                    for (int k = 0; k < Env::KR; ++k) rn[k] = (rnr + (size_t)k * p.ld_noise)[0];
                    float r0[S];
                    Env::init(rn, r0);
#pragma unroll
                    for (int g = 0; g < 8; ++g) { v4f w = {r0[4 * g], r0[4 * g + 1], r0[4 * g + 2], r0[4 * g + 3]}; mine[8 * g] = w; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            const unsigned long long m = NOISE ? 0ull : __ballot(done);
            if (m != 0ull) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));   // finishing lanes below this one (v_mbcnt: no per-lane mask register)
                if (done) wl[rank] = (unsigned char)lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const int total = __popcll(m) * Env::RESET_ITEMS;            // six items per finishing lane
                for (int i = (int)lane; i < total; i += 64) {
                    int ii = i;
                    asm volatile("" : "+v"(ii));   // not a loop invariant of the rollout loop: see coop_reset
                    const unsigned li = ((unsigned)ii * 171u) >> 10;                 // ii / 6, exact for ii < 515
                    const unsigned owner = wl[li];
                    float *row = imgf + ((owner >> 3) * 64u + (owner & 7u)) * 4u;
                    Env::reset_item_to(make_key(wave_gi0 + owner, t_base + (uint32_t)it + 1u, p.seed_lo, p.seed_hi, s_probit),
                                       (unsigned)ii - 6u * li, [row](uint32_t k, float v) { row[(k >> 2) * 32u + (k & 3u)] = v; });
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
    };

    if constexpr (!POLICY) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
        load_action(ring + (size_t)slot * q.slot_stride, buf[j]);
        slot = (slot + 1 == q.ring_len) ? 0 : slot + 1;
    }
    act_next = ring + (size_t)slot * q.slot_stride;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);            // prologue loads drained here, not in the loop header (rollout_body)
#ifdef NIG_DIAG_PG_STAGGER             // (diagnostic builds only: start the four waves of a SIMD a quarter step apart)
    {
        const unsigned ph = ((wave >> 2) + 2u * (blockIdx.x & 1u)) & 3u;
        for (unsigned k = 0; k < ph * NIG_DIAG_PG_STAGGER; ++k) __builtin_amdgcn_s_sleep(64);
    }
#endif
    int it = it0;
    for (; it < q.n_steps; ++it) one_step(buf[0], it);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const v4f v = mine[8 * g];
        float *r = p.state + base + (4 * g) * p.ld_state;
        r[tid] = v.x; (r + p.ld_state)[tid] = v.y; (r + 2 * p.ld_state)[tid] = v.z; (r + 3 * p.ld_state)[tid] = v.w;
    }
    (p.ctr + base)[tid] = ctr;
    if (lt.viol != 0) (p.life_viol + base)[tid] += (long long)lt.viol;    // base.py:183 total_violations of the finished episodes
    if (tally) {
        (p.ep_ret + base)[tid] = ret;
        if (lt.episodes > 0) lt.merge(p.tally + base + tid, p.ld, p.n_en);
    }
    if constexpr (PROD) NIG_RING_REPORT(p.ring_err, nz_sync, lane);
}

// The paired form's kernel: 256 lanes per block, 512 threads -- waves 0-3 step (pg_lds_rollout_body, PROD), waves 4-7
// produce their normals (pg_pair_producer).  q.block0 counts 256-lane blocks.  One block per compute unit is resident.
// REG (round 4, the default of the open loop): the stepping waves run the REGISTER-resident rollout body fed from the
// producers' ring (rollout_body<PowerGrid, ..., RING>): at this form's two waves per SIMD a wave may hold 256 registers, and
// with state, counters and tallies in registers a step is one chain of arithmetic -- the LDS-resident body (REG = false: what
// round 3 ran here, and what the injected-draw variant still runs) walks its state image through ~50 dependent LDS round
// trips per step, ~10 cycles per instruction on the stepping wave.
template <int OUT, bool NOISE = false, bool REG = false>
__global__ void __launch_bounds__(512, 2) rollout_pg_pair_kernel(const RolloutArgs q)
{
    using PL = std::conditional_t<REG, PgPairRegLds<OUT>, PgPairLdsT<NOISE>>;
    static_assert(!(REG && NOISE), "recorded draws: the LDS-resident stepping body");
    __shared__ __attribute__((aligned(16))) unsigned char smem[PL::BYTES];
    float4 *const s_probit = reinterpret_cast<float4 *>(smem + PgLds<256>::OFF_PROBIT);
    for (int i_ = (int)threadIdx.x; i_ < 768; i_ += 512) s_probit[i_] = NIG_PROBIT[i_];
    if (threadIdx.x < 16u) reinterpret_cast<uint32_t *>(smem + PL::OFF_SYNC)[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = (blockIdx.x + q.block0) * 256u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 4u) pg_pair_producer<NOISE, false, RolloutArgs, PL>(q, base, smem, wave - 4u, threadIdx.x & 63u);
    else if constexpr (REG) {
        rollout_body<PowerGrid, OUT, false, true, true, 256, false, true>(
            q, base, smem, reinterpret_cast<const v4f *>(smem + PL::OFF_NZ) + wave * (PL::K * PL::SLOT_V4),
            (lds_u32_t *)(smem + PL::OFF_SYNC) + wave * 4);
        NIG_RING_REPORT(q.s.ring_err, (lds_u32_t *)(smem + PL::OFF_SYNC) + wave * 4, threadIdx.x & 63u);
    } else pg_lds_rollout_body<OUT, 256, true, NOISE>(q, base, smem);
}

// The closed loop's stepping wave with its state in REGISTERS (paired form, calls WITHOUT the observation stream): the loop of
// rollout_policy_kernel -- feedback law, IndustrialEnv.step, bookkeeping, cooperative reset, the same calls on the same values --
// with the step's normals and the policy's draws read from the producer's slot and the feedback matrix from the dense LDS copy.
// At the pair form's two waves per SIMD a wave may hold 256 registers; against the LDS-resident stepper this spares the eight
// ds_read_b128 of the observation and the image's round trips (the open loop's REG form: -12 % without a trajectory).
template <class QA>
__device__ __forceinline__ void pg_policy_reg_body(const QA &q, const uint32_t base, unsigned char *smem)
{
    using Env = PowerGrid;
    using PL = PgPairLdsT<false, true>;
    constexpr int S = Env::S, A = Env::A, KS = Env::KS;
    const float4 *const s_probit = reinterpret_cast<const float4 *>(smem + PgLds<256>::OFF_PROBIT);
    const nig_policy *const pol = reinterpret_cast<const nig_policy *>(smem + PL::OFF_POL);
    const v4f *const wd = reinterpret_cast<const v4f *>(smem + PL::OFF_WD);
    const StepArgs &p = q.s;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    float *const s_img = reinterpret_cast<float *>(smem + PgLds<256>::OFF_IMG) + wave * 2048;      // the cooperative reset's [32][64] image
    unsigned char *const s_wlist = smem + PgLds<256>::OFF_WLIST + wave * 64;
    const v4f *const nz_ring = reinterpret_cast<const v4f *>(smem + PL::OFF_NZ) + wave * (PL::K * PL::SLOT_V4);
    lds_u32_t *const nz_sync = (lds_u32_t *)(smem + PL::OFF_SYNC) + wave * 4;
    const uint32_t t_base = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off;
    const bool tally = p.tally != nullptr;
    uint32_t ctr = (p.ctr + base)[tid];
    float s[S], n[S], a[A];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = (p.state + base + k * p.ld_state)[tid];
    double ret = tally ? (p.ep_ret + base)[tid] : 0.0;
    LaneTally lt;
    lt.clear();
    PolicyHead<A> head;
    head.load(*pol);
    bool any_sigma, any_half, mix;
    policy_switches<A>(&head, any_sigma, any_half, mix);
    float *rew_row = p.reward ? p.reward + base : nullptr;
    uint32_t *fl_row = p.flags ? p.flags + base : nullptr;
    float *act_row = q.act_out ? q.act_out + base : nullptr;
    // the observation the policy acts on, row-major [n_steps][B][S] (optional; round 5: this body writes it too -- through the
    // wave's reset image, which is idle until the step's cooperative reset: the lane's eight float4 groups in the conflict-free
    // layout of pg_lds_rollout_body's state image, read back transposed as eight 1 KiB whole-line streaming stores)
    float *obs_row = q.obs_out ? q.obs_out + (size_t)(base + __builtin_amdgcn_readfirstlane(tid & ~63u)) * S : nullptr;
    v4f *const timg = reinterpret_cast<v4f *>(s_img);
    v4f *const mine = timg + (lane >> 3) * 64 + (lane & 7u);
    const unsigned rd = (lane & 7u) * 8u + (lane >> 3);
    uint32_t seen = 0u;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int it = 0; it < q.n_steps; ++it) {
        if (obs_row != nullptr) {                  // (wave-uniform) the lane's pre-step values into the image; read back transposed below,
#pragma unroll                                     // behind the feedback law, so that the LDS round trip is not on this lone wave's chain
            for (int g = 0; g < 8; ++g) { v4f w = {s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]}; mine[8 * g] = w; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // other lanes' rows are read below (compiler order; DS operations execute in order)
        }
        if (seen < (uint32_t)it + 1u) seen = split_wait(nz_sync + 0, (uint32_t)it + 1u);
        const v4f *slot = nz_ring + (it & (PL::K - 1)) * PL::SLOT_V4 + lane;
        float z[24];
#pragma unroll
        for (int j = 0; j < 6; ++j) { const v4f w = slot[64 * j]; z[4 * j] = w.x; z[4 * j + 1] = w.y; z[4 * j + 2] = w.z; z[4 * j + 3] = w.w; }
        const float *dr = reinterpret_cast<const float *>(nz_ring + (it & (PL::K - 1)) * PL::SLOT_V4 + PL::NZ_V4) + lane;
        PolicyDraws<A> d;
        if (any_sigma) {
#pragma unroll
            for (int k = 0; k < A; ++k) d.z[k] = dr[k * 64];
        }
        if (any_half) {
#pragma unroll
            for (int k = 0; k < A; ++k) d.h[k] = dr[(8 + k) * 64];
        }
        if (mix) {
#pragma unroll
            for (int k = 0; k < A; ++k) d.ra[k] = dr[(16 + k) * 64];
            d.wmix = dr[24 * 64];
        }
        split_post(nz_sync + 1, (uint32_t)it + 1u, lane);          // (DS order: the reads above execute before this write)
        policy_affine_dense<Env, 4>(head, wd, s, a);       // (four columns ahead: this wave also holds the whole state)
        policy_finish_sw<Env>(&head, any_sigma, any_half, mix, head.clip_lo, head.clip_hi, d, a);
        if (act_row) {                             // the policy's action, before the env's own clip
#pragma unroll
            for (int j = 0; j < A; ++j) stream_store(act_row + j * q.ld_act_out + tid, a[j]);
            act_row += q.act_step_stride;
        }
        if (obs_row != nullptr) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            v4f tv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tv[j] = timg[64 * j + rd];
            v4f *oo = reinterpret_cast<v4f *>(obs_row);
#pragma unroll
            for (int j = 0; j < 8; ++j) stream_store(oo + lane + 64u * j, tv[j]);
            obs_row += q.obs_step_stride;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the image is the cooperative reset's scratch later in the step
        }
        float nz[KS];
        Env::scale_step_normals(z, nz);
        const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
        StepResult<Env> res;
        step_core<Env>(s, a, nz, step_pre, p.max_steps, p.dt32, p.dt, p.cmask, n, res);
        const int step = step_pre + 1;
        const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;
        const bool done = res.terminated || res.truncated;
        const uint32_t fl = pack_flags<Env>(res, step) | (done ? NIG_FLAG_DID_RESET : 0u);
        ctr = (uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT);
        if (tally) ret = ret + (double)res.reward;
        if (rew_row) { stream_store(rew_row + tid, (float)res.reward); rew_row += q.out_stride; }
        if (fl_row) { stream_store(fl_row + tid, fl); fl_row += q.out_stride; }
        if (done) {
            lt.life += (long long)viol_ep;
            if (tally) { lt.episode(ret, step, viol_ep, res.ncrit); ret = 0.0; }
            ctr = 0u;
        }
        const unsigned long long m = __ballot(done);
        if (m != 0ull)
            coop_reset<Env>(m, done, lane, s_img, s_wlist, p.env0 + (uint64_t)(base + (tid & ~63u)), t_base + (uint32_t)it + 1u,
                            p.seed_lo, p.seed_hi, s_probit, n);
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = n[k];
    }
#pragma unroll
    for (int k = 0; k < S; ++k) (p.state + base + k * p.ld_state)[tid] = s[k];
    (p.ctr + base)[tid] = ctr;
    if (lt.life != 0) (p.life_viol + base)[tid] += lt.life;
    if (tally) {
        (p.ep_ret + base)[tid] = ret;
        if (lt.episodes > 0) lt.merge(p.tally + base + tid, p.ld, p.n_en);
    }
    NIG_RING_REPORT(p.ring_err, nz_sync, lane);
}

// The paired form of the CLOSED loop (nig_rollout_policy, affine policies): as above, the producers also draw the policy's
// own random numbers of the step, the stepping waves evaluate the feedback law on their image.  QA = PolicyArgs (a template
// parameter only because that type is defined after this header); q.block0 counts 256-lane blocks.
template <class QA>
__global__ void __launch_bounds__(512, 2) rollout_pg_pair_policy_kernel(const QA q)
{
    using PL = PgPairLdsT<false, true>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[PL::BYTES];
    float4 *const s_probit = reinterpret_cast<float4 *>(smem + PgLds<256>::OFF_PROBIT);
    for (int i_ = (int)threadIdx.x; i_ < 768; i_ += 512) s_probit[i_] = NIG_PROBIT[i_];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(q.pol);
        uint32_t *dst = reinterpret_cast<uint32_t *>(smem + PL::OFF_POL);
        for (unsigned i_ = threadIdx.x; i_ < sizeof(nig_policy) / 4; i_ += 512u) dst[i_] = src[i_];
        if (threadIdx.x < 256u)                    // the dense aligned copy of the feedback matrix: [32 columns][8 actions]
            reinterpret_cast<float *>(smem + PL::OFF_WD)[threadIdx.x] = q.pol->Wt[threadIdx.x >> 3][threadIdx.x & 7u];
    }
    if (threadIdx.x < 16u) reinterpret_cast<uint32_t *>(smem + PL::OFF_SYNC)[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = (blockIdx.x + q.block0) * 256u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 4u) pg_pair_producer<false, true, QA>(q, base, smem, wave - 4u, threadIdx.x & 63u);
#ifdef NIG_DIAG_PG_POLICY_LDS          // (diagnostic builds only: the LDS-resident stepper for every call, for same-box A/Bs)
    else pg_lds_rollout_body<0, 256, true, false, true, QA>(q, base, smem);
#elif defined(NIG_DIAG_PG_POLICY_LDS_OBS)   // (diagnostic: round 4's rule -- calls with the observation stream on the LDS-resident stepper)
    else if (q.obs_out != nullptr) pg_lds_rollout_body<0, 256, true, false, true, QA>(q, base, smem);
    else pg_policy_reg_body<QA>(q, base, smem);
#else
    else pg_policy_reg_body<QA>(q, base, smem);    // (round 5: the register-resident stepper writes the observation stream as well)
#endif
}

template <class E, class = void> struct pair_rollout : std::false_type {};
template <class E> struct pair_rollout<E, std::void_t<decltype(E::PAIR_ROLLOUT)>> : std::bool_constant<E::PAIR_ROLLOUT> {};

}  // namespace nig
