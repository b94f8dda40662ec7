// nig_split_policy.hpp -- the closed-loop (device policy) rollout in the three-wave form of nig_split.hpp
// (included by nig_kernels.hpp behind rollout_policy_kernel; same batches: ChemicalReactor, whole 256-lane blocks,
// up to one block per CU, auto-reset handle without frozen lanes).
//
// What changes against the open-loop form: the action is a function of the state, so it cannot be produced ahead.
//   producer   : process noise of the step + the policy's own random draws (exploration normals, uniform
//                perturbation, epsilon-mix draw and its uniform action: policy_draws) -- everything that depends
//                on the lane's key only.  No global memory access at all.
//   integrator : policy_apply on its state (feedback law + the draws + the policy's clip; PID memory lives in
//                its registers), then the step as before.  Leaves the observation it acted on, the post-dynamics
//                state, the violation bits and the policy's action.
//   recorder   : reward / flags / tally as before, plus the transition stream the caller asked for (observation
//                acted on, row-major through the transposing image; the policy's action rows).
// Bit-identical to rollout_policy_kernel (tests/test_gpu_split.py).
#pragma once

namespace nig {

template <class Env, int NP>
struct SplitPolicyLds {
    // BIG (S > 16: RobotAssembly, round 4): two ring slots instead of three, no observation rows in the I -> C slot, and the
    // feedback matrix as a dense LDS copy (policy_affine_dense) instead of registers -- what makes four triples of a 24-value
    // state fit one CU's 160 KiB.  The observation the policy acted on (the transition stream's `observations`, asked for at run
    // time) travels in the P -> I slot instead (round 5): once the integrator has read the step's draws out of it, the slot is
    // dead until the recorder releases it (the producer refills slot j only after the recorder is done with step j - K), so the
    // integrator writes its 64 x S pre-step rows there and the recorder reads them transposed.  The slot has max(draw rows, S)
    // rows for that: +2 rows, 4 KiB per block, 162 320 of 163 840 bytes.
    static constexpr bool BIG = Env::S > 16;
    static constexpr int K = BIG ? 2 : 3;                              // ring slots
    static constexpr int DRAW_ROWS = Env::KS + 3 * Env::A + 1;         // noise, z, h, ra, wmix
    static constexpr int HI_ROWS = (BIG && Env::S > DRAW_ROWS) ? Env::S : DRAW_ROWS;
    static constexpr int HI_SLOT = HI_ROWS * 64;                       // floats
    static constexpr int OBS_ROWS = BIG ? 0 : Env::S;                  // [64][S] observation acted on (for the transition stream)
    static constexpr int IH_SLOT = (OBS_ROWS + Env::S + 1 + Env::A) * 64;   // observation, [64][S] post-dynamics state, violation words, [A][64] action
    static constexpr int OFF_PROBIT = 16 * PROBIT_BIAS;
    static constexpr int OFF_POL = OFF_PROBIT + 768 * 16;
    static constexpr int OFF_WD = OFF_POL + (int)((sizeof(nig_policy) + 15) / 16 * 16);      // BIG: dense [S rounded up to 8][8] feedback matrix
    static constexpr int OFF_IMG = OFF_WD + (BIG ? (Env::S + 7) / 8 * 8 * 8 * 4 : 0);
    static constexpr int OFF_WLIST = OFF_IMG + NP * Env::RESET_ROWS * 64 * 4;
    static constexpr int OFF_SYNC = OFF_WLIST + NP * 64;
    static constexpr int OFF_HI = OFF_SYNC + NP * 16;
    static constexpr int OFF_IH = OFF_HI + NP * K * HI_SLOT * 4;
    static constexpr int BYTES = OFF_IH + NP * K * IH_SLOT * 4;
    static_assert(BYTES <= 160 * 1024, "LDS of one CU");
};

template <class Env, int NP>
__global__ void __launch_bounds__(192 * NP, 1) split_policy_kernel(const PolicyArgs q)
{
    constexpr int S = Env::S, A = Env::A, KS = Env::KS;
    static_assert(Env::COOP_RESET && !Env::CUSTOM_STEP && S % 4 == 0 && ((Env::SHARED_STEP_BLOCK && KS == 2) || KS == 0),
                  "three-wave closed loop: cooperative reset, S a multiple of 4, ChemicalReactor's step noise or none");
    using Lds = SplitPolicyLds<Env, NP>;
    constexpr int K = Lds::K;
    constexpr bool BIG = Lds::BIG;
    constexpr int KN = KS > 0 ? KS : 1;
    constexpr int OBS = Lds::OBS_ROWS;             // rows of the I -> C slot before the post-dynamics state
    constexpr int THREADS = 192 * NP;
    constexpr int ROW_Z = KS, ROW_H = KS + A, ROW_RA = KS + 2 * A, ROW_MIX = KS + 3 * A;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Lds::BYTES];
    float4 *const s_probit = reinterpret_cast<float4 *>(smem + Lds::OFF_PROBIT);
    const nig_policy *const pol = reinterpret_cast<const nig_policy *>(smem + Lds::OFF_POL);
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned pair = wave % NP, role = wave / NP;          // 0 integrator, 1 producer, 2 recorder
    lds_u32_t *const sync = (lds_u32_t *)(smem + Lds::OFF_SYNC) + pair * 4;
    float *const s_hi = reinterpret_cast<float *>(smem + Lds::OFF_HI) + pair * (K * Lds::HI_SLOT);
    float *const s_ih = reinterpret_cast<float *>(smem + Lds::OFF_IH) + pair * (K * Lds::IH_SLOT);
    for (int i_ = (int)tid; i_ < 768; i_ += THREADS) s_probit[i_] = NIG_PROBIT[i_];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(q.pol);
        uint32_t *dst = reinterpret_cast<uint32_t *>(smem + Lds::OFF_POL);
        for (unsigned i_ = tid; i_ < sizeof(nig_policy) / 4; i_ += THREADS) dst[i_] = src[i_];
        if constexpr (BIG) policy_stage_dense<Env>(q.pol, reinterpret_cast<float *>(smem + Lds::OFF_WD), tid, THREADS);
    }
    if (tid < NP * 4) reinterpret_cast<uint32_t *>(smem + Lds::OFF_SYNC)[tid] = 0u;
    __syncthreads();

    const StepArgs &p = q.s;
    const uint32_t base = blockIdx.x * (64u * NP) + pair * 64u;      // the triple's first lane
    const uint32_t t_base = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off;     // step i uses t_base + i + 1
    const int n = q.n_steps;
    bool any_sigma, any_half, mix;                                   // wave-uniform switches of the policy
    policy_switches<A>(pol, any_sigma, any_half, mix);

    if (role == 1) {
        // ------------------------------------------------------------------ producer
        const uint64_t gi = p.env0 + (uint64_t)(base + lane);
        u32x4 blk = {0u, 0u, 0u, 0u};
        int pslot = 0;
        uint32_t freed = 0u;
        for (int j = 0; j < n; ++j) {
            const RngKey key = make_key(gi, t_base + (uint32_t)j + 1u, p.seed_lo, p.seed_hi, s_probit);
            typename Env::fast_noise_t nz[KN];
            if constexpr (KS > 0) {
                const bool second = (key.t & 1u) == 0u;              // launch counters 2k-1, 2k share one Philox block (draw_step)
                if (!second || j == 0) blk = pair_block<Env>(key);
                pair_noise<Env>(second ? blk.z : blk.x, second ? blk.w : blk.y, s_probit, nz);
            }
            PolicyDraws<A> d;
            policy_draws<Env>(pol, key, d);
            if (freed + (uint32_t)K < (uint32_t)j + 1u) freed = split_wait(sync + 2, (uint32_t)(j + 1 - K));
            float *hi = s_hi + pslot * Lds::HI_SLOT;
#pragma unroll
            for (int k = 0; k < KS; ++k) hi[k * 64 + lane] = (float)nz[k];    // exact: the fast-mode noise IS a float
            if (any_sigma) {
#pragma unroll
                for (int k = 0; k < A; ++k) hi[(ROW_Z + k) * 64 + lane] = d.z[k];
            }
            if (any_half) {
#pragma unroll
                for (int k = 0; k < A; ++k) hi[(ROW_H + k) * 64 + lane] = d.h[k];
            }
            if (mix) {
#pragma unroll
                for (int k = 0; k < A; ++k) hi[(ROW_RA + k) * 64 + lane] = d.ra[k];
                hi[ROW_MIX * 64 + lane] = d.wmix;
            }
            NIG_RING_FAULT_GUARD(p.hflags, j) split_post(sync + 0, (uint32_t)j + 1u, lane);
            pslot = (pslot + 1 == K) ? 0 : pslot + 1;
        }
        NIG_RING_REPORT(p.ring_err, sync, lane);
        return;
    }

    if (role == 0) {
        // ------------------------------------------------------------------ integrator
        float *const s_img = reinterpret_cast<float *>(smem + Lds::OFF_IMG) + pair * (Env::RESET_ROWS * 64);
        unsigned char *const s_wlist = smem + Lds::OFF_WLIST + pair * 64;
        int step = (int)((p.ctr + base)[lane] & NIG_CTR_STEP_MASK);
        float s[S], nx[S], integ[A], eprev[A];
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = (p.state + base + k * p.ld_state)[lane];
        // PID memory lives in the handle (baseline_agents.py:55-80): loaded here, stored at the end
        const bool pid_mem = q.pid != nullptr && pol->kind == NIG_POLICY_PID;
#pragma unroll
        for (int j = 0; j < A; ++j) {
            integ[j] = pid_mem ? (q.pid + base + (size_t)j * p.ld)[lane] : 0.0f;
            eprev[j] = pid_mem ? (q.pid + base + (size_t)(A + j) * p.ld)[lane] : 0.0f;
        }
        // the law's coefficients in registers (PolicyRegs); BIG: only offsets and column mask -- RobotAssembly's integrator has
        // no registers to spare (153 of 168 in the open loop), the matrix comes from the dense LDS copy four columns ahead and
        // the rarely used per-action fields (noise scales, PID set-points, clip) are read in place
        struct SmallHead { uint32_t colmask; float b[A], clip_lo, clip_hi; };
        std::conditional_t<BIG, SmallHead, PolicyRegs<Env>> pr;
        if constexpr (BIG) {
            pr.colmask = __builtin_amdgcn_readfirstlane(pol->colmask);
            pr.clip_lo = pol->clip_lo; pr.clip_hi = pol->clip_hi;
#pragma unroll
            for (int j = 0; j < A; ++j) pr.b[j] = pol->b[j];
        } else {
            pr.load(*pol);
        }
        [[maybe_unused]] const v4f *const wd = reinterpret_cast<const v4f *>(smem + Lds::OFF_WD);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_setprio(3);
        int slot = 0;
        uint32_t seen = 0u;
        for (int i = 0; i < n; ++i) {
            if (seen < (uint32_t)i + 1u) seen = split_wait(sync + 0, (uint32_t)i + 1u);
            const float *hi = s_hi + slot * Lds::HI_SLOT;
            double nz[KN];
            nz[0] = 0.0;
#pragma unroll
            for (int k = 0; k < KS; ++k) nz[k] = (double)hi[k * 64 + lane];
            PolicyDraws<A> d;
            if (any_sigma) {
#pragma unroll
                for (int k = 0; k < A; ++k) d.z[k] = hi[(ROW_Z + k) * 64 + lane];
            }
            if (any_half) {
#pragma unroll
                for (int k = 0; k < A; ++k) d.h[k] = hi[(ROW_H + k) * 64 + lane];
            }
            if (mix) {
#pragma unroll
                for (int k = 0; k < A; ++k) d.ra[k] = hi[(ROW_RA + k) * 64 + lane];
                d.wmix = hi[ROW_MIX * 64 + lane];
            }
            float a[A];
            if constexpr (BIG) {                   // the same law, its matrix prefetched from the dense LDS copy
                if (pol->kind == NIG_POLICY_PID) {   // baseline_agents.py:61-80 (policy_apply's branch)
                    const float kp = pol->kp, ki = pol->ki, kd = pol->kd;
#pragma unroll
                    for (int j = 0; j < A; ++j) {
                        const float e = pol->setpoint[j] - s[j];
                        integ[j] = integ[j] + e;
                        a[j] = (kp * e + ki * integ[j]) + kd * (e - eprev[j]);
                        eprev[j] = e;
                    }
                } else {
                    policy_affine_dense<Env, 4>(pr, wd, s, a);
                }
                policy_finish_sw<Env>(pol, any_sigma, any_half, mix, pr.clip_lo, pr.clip_hi, d, a);
            } else {
                policy_apply<Env>(&pr, s, d, integ, eprev, a);
            }
            float *ih = s_ih + slot * Lds::IH_SLOT;
            if constexpr (OBS > 0) {
                v4f *row = reinterpret_cast<v4f *>(ih) + lane * (S / 4);
#pragma unroll
                for (int k = 0; k < S / 4; ++k) { v4f v = {s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]}; row[k] = v; }
            } else if (q.obs_out != nullptr) {     // BIG: the observation rows ride in the P -> I slot, whose draws are in registers by now
                // (DS operations of a wave execute in order: the reads above precede these writes; wave-uniform switch)
                v4f *row = reinterpret_cast<v4f *>(s_hi + slot * Lds::HI_SLOT) + lane * (S / 4);
#pragma unroll
                for (int k = 0; k < S / 4; ++k) { v4f v = {s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]}; row[k] = v; }
            }
#pragma unroll
            for (int k = 0; k < A; ++k) ih[(OBS + S + 1 + k) * 64 + lane] = a[k];        // the policy's action, before the env's clip
            StepResult<Env> res;
            step_core<Env>(s, a, nz, step, p.max_steps, p.dt32, p.dt, p.cmask, nx, res);  // the reward part is dead here
            const bool done = res.terminated || res.truncated;
            v4f *rown = reinterpret_cast<v4f *>(ih + OBS * 64) + lane * (S / 4);
#pragma unroll
            for (int k = 0; k < S / 4; ++k) { v4f v = {nx[4 * k], nx[4 * k + 1], nx[4 * k + 2], nx[4 * k + 3]}; rown[k] = v; }
            reinterpret_cast<uint32_t *>(ih + (OBS + S) * 64)[lane] = res.viol_bits;
            split_post(sync + 1, (uint32_t)i + 1u, lane);
            step = done ? 0 : step + 1;
            const unsigned long long m = __ballot(done);
            if (m != 0ull)
                coop_reset<Env>(m, done, lane, s_img, s_wlist, p.env0 + (uint64_t)base, t_base + (uint32_t)i + 1u,
                                p.seed_lo, p.seed_hi, s_probit, nx);
#pragma unroll
            for (int k = 0; k < S; ++k) s[k] = nx[k];
            slot = (slot + 1 == K) ? 0 : slot + 1;
        }
#pragma unroll
        for (int k = 0; k < S; ++k) (p.state + base + k * p.ld_state)[lane] = s[k];
        if (pid_mem) {
#pragma unroll
            for (int j = 0; j < A; ++j) {
                (q.pid + base + (size_t)j * p.ld)[lane] = integ[j];
                (q.pid + base + (size_t)(A + j) * p.ld)[lane] = eprev[j];
            }
        }
        NIG_RING_REPORT(p.ring_err, sync, lane);
        return;
    }

    // ---------------------------------------------------------------------- recorder
    uint32_t ctr = (p.ctr + base)[lane];
    const bool tally = p.tally != nullptr;
    double ret = tally ? (p.ep_ret + base)[lane] : 0.0;
    LaneTally lt;
    lt.clear();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int cslot = 0;
    uint32_t seen = 0u;
    for (int i = 0; i < n; ++i) {
        if (seen < (uint32_t)i + 1u) seen = split_wait(sync + 1, (uint32_t)i + 1u);
        const float *ih = s_ih + cslot * Lds::IH_SLOT;
        float nx[S], a[A];
        const v4f *rown = reinterpret_cast<const v4f *>(ih + OBS * 64) + lane * (S / 4);
#pragma unroll
        for (int k = 0; k < S / 4; ++k) { const v4f v = rown[k]; nx[4 * k] = v.x; nx[4 * k + 1] = v.y; nx[4 * k + 2] = v.z; nx[4 * k + 3] = v.w; }
        const uint32_t vb = reinterpret_cast<const uint32_t *>(ih + (OBS + S) * 64)[lane];
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = ih[(OBS + S + 1 + k) * 64 + lane];
        [[maybe_unused]] v4f tr[S / 4];
        if (q.obs_out) {                           // the wave's 64 observation rows in lane-contiguous order
            // (written by the integrator before it posted this step: I -> C slot, or -- BIG -- the P -> I slot of the same index)
            const v4f *src = OBS > 0 ? reinterpret_cast<const v4f *>(ih) : reinterpret_cast<const v4f *>(s_hi + cslot * Lds::HI_SLOT);
#pragma unroll
            for (int k = 0; k < S / 4; ++k) tr[k] = src[lane + 64u * k];
        }
        split_post(sync + 2, (uint32_t)i + 1u, lane);
        const uint32_t orow = (uint32_t)i * q.out_stride;
        {
        if (q.obs_out) {
            v4f *oo = reinterpret_cast<v4f *>(q.obs_out + (size_t)i * q.obs_step_stride + (size_t)base * S);
#pragma unroll
            for (int k = 0; k < S / 4; ++k) stream_store(oo + lane + 64u * k, tr[k]);
        }
        }
        if (q.act_out) {
            float *ao = q.act_out + (size_t)i * q.act_step_stride + base;
#pragma unroll
            for (int j = 0; j < A; ++j) stream_store(ao + j * q.ld_act_out + lane, a[j]);
        }
        clip_action<Env, float>(a);                // what step_core did to it on the integrator (base.py:167)
        const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
        StepResult<Env> res;
        post_core<Env, float>(nx, a, vb, step_pre, p.max_steps, res);
        const int step = step_pre + 1;
        const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;
        const bool done = res.terminated || res.truncated;
        const uint32_t fl = pack_flags<Env>(res, step) | (done ? NIG_FLAG_DID_RESET : 0u);
        ctr = (uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT);
        if (tally) {
            if constexpr (Env::RET_F32) ret = (double)((float)ret + res.reward);
            else ret = ret + (double)res.reward;
        }
        if (p.reward) stream_store(p.reward + base + orow + lane, (float)res.reward);
        if (p.flags) stream_store(p.flags + base + orow + lane, fl);
        if (done) {
            lt.life += (long long)viol_ep;
            if (tally) { lt.episode(ret, step, viol_ep, res.ncrit); ret = 0.0; }
            ctr = 0u;
        }
        cslot = (cslot + 1 == K) ? 0 : cslot + 1;
    }
    (p.ctr + base)[lane] = ctr;
    if (lt.life != 0) (p.life_viol + base)[lane] += lt.life;
    if (tally) {
        (p.ep_ret + base)[lane] = ret;
        if (lt.episodes > 0) lt.merge(p.tally + base + lane, p.ld, p.n_en);
    }
    NIG_RING_REPORT(p.ring_err, sync, lane);
}

}  // namespace nig
