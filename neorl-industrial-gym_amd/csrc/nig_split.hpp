// nig_split.hpp -- the fused rollout for batches that leave ONE wave per SIMD (included by nig_kernels.hpp).
//
// At BASELINE.json's headline size (65 536 lanes = 1024 waves on 1024 SIMDs) the rollout kernel is bound by what a
// single wave can issue: a lone wave gets one VALU instruction issued every 5-7 cycles where the SIMD could take
// one every ~2.5 (profiles/ubench/valu_rate.hip), and the lane-count probe (profiles/r02/scale_probe.txt) shows
// the same thing on the kernel itself -- twice the lanes cost 1.25x the time.  More lanes per SIMD are not
// available at that batch size, so this kernel puts THREE waves on every SIMD that work on the SAME 64
// environments, each with a third of IndustrialEnv.step:
//
//   producer   (P): what a step consumes.  Philox + normal transform of the process noise, action load + clip.
//                   Runs ahead of the integrator by up to the ring length.  Loads only, no stores.
//   integrator (I): owns the state.  Constraint check on the pre-state, dynamics, done / truncation, in-kernel
//                   reset.  Touches no global memory inside the loop.  The critical path of the three.
//   recorder   (C): what a step leaves behind.  Reward, penalties, flag word, episode tally, the transposed
//                   trajectory rows and ALL the stores.  Stores only, no loads: it never waits on vmcnt.
//
// They exchange through two rings of K slots in LDS, each guarded by a monotonically increasing counter (slots
// produced so far), plus the recorder's count of slots it is done with.  P -> I slot: the KS noise values and A
// clipped actions of one step, [row][lane].  I -> C slot: the S post-dynamics state values row-major [lane][S]
// (this IS the transposing image of the row-major trajectory) plus the violation bits of the pre-state; C also
// reads the clipped action back from the P -> I slot.  DS operations of one wave execute in order, so "write data,
// then write counter" / "read counter, then read data" needs no wait in between; the wavefront-scope fences only
// pin the compiler's order.  No block barrier inside the loop.  Flow control: P writes slot j only after C is
// done with slot j - K; I needs P's slot j before it writes its own slot j, so it cannot overrun C either.
//
// The arithmetic is the other kernels': same clip, violated, dynamics, post_core, pack_flags, tally and reset
// calls on the same values with the same generator keys -- the results are bit-identical to rollout_kernel's
// (tests/test_gpu_parity.py, tests/test_spec_envs.py: the fused-rollout tests run this form wherever it applies,
// tests/test_gpu_split.py pins it against the one-wave form).
#pragma once

namespace nig {

template <class E, class = void> struct split_rollout : std::false_type {};
template <class E> struct split_rollout<E, std::void_t<decltype(E::SPLIT_ROLLOUT)>> : std::bool_constant<E::SPLIT_ROLLOUT> {};
// batches of more blocks than are resident at once: run the three-wave form in rounds (ChemicalReactor: measured faster,
// profiles/r02/rounds_probe.txt), or leave them to the form that fills the SIMDs with lanes (SPLIT_ROUNDS = false)
template <class E, class = void> struct split_rounds : std::true_type {};
template <class E> struct split_rounds<E, std::void_t<decltype(E::SPLIT_ROUNDS)>> : std::bool_constant<E::SPLIT_ROUNDS> {};

template <class Env, int NP>
struct SplitLds {
    // ring slots: as many as the CU's LDS holds for NP triples (ChemicalReactor, S = 12: six; RobotAssembly, S = 24, whose
    // producer and recorder are light next to the integrator and never need to run far ahead / behind: three)
    static constexpr int K = Env::S > 16 ? 3 : 6;
    static constexpr int HI_ROWS = Env::KS + Env::A;
    static constexpr int HI_SLOT = HI_ROWS * 64;             // floats
    static constexpr int IH_SLOT = (Env::S + 1) * 64;        // floats: [64][S] state rows, then [64] violation words
    static constexpr int OFF_PROBIT = 16 * PROBIT_BIAS;                        // (probit_fetch: the bias rides in the DS offset field)
    static constexpr int OFF_IMG = OFF_PROBIT + 768 * 16;                                   // float [NP][RESET_ROWS][64]
    static constexpr int OFF_WLIST = OFF_IMG + NP * Env::RESET_ROWS * 64 * 4;  // uchar [NP][64]
    static constexpr int OFF_SYNC = OFF_WLIST + NP * 64;                       // uint32 [NP][4]: {P produced, I produced, C done}
    static constexpr int OFF_HI = OFF_SYNC + NP * 16;
    static constexpr int OFF_IH = OFF_HI + NP * K * HI_SLOT * 4;
    static constexpr int BYTES = OFF_IH + NP * K * IH_SLOT * 4;
    static_assert(BYTES <= 160 * 1024, "LDS of one CU");
};

// (ring counters, waits, posts and the ordering assumption they rest on: nig_ring.hpp)

// NP wave triples per block: wave w < NP integrates lanes base + 64 w .. + 63, wave NP + w is their producer and
// wave 2 NP + w their recorder (a block's waves go to the CU's four SIMDs round-robin: with NP = 4 the three
// share one).  Whole 64*NP-lane blocks only, auto-reset handles without frozen lanes only, and for an env with step
// noise a launch starting on an odd counter (PAIRED form): the host keeps every other case on rollout_kernel.
// NOISE (nig_rollout_noise): the reference's recorded draws instead of the generator's -- the producer LOADS the step's
// process noise (float64 rows, handed on as the float the fast-mode ring carries: ChemicalReactor's dynamics round the
// draw to float32 before they use it, chemical_reactor.py:149,159 under NEP 50, so nothing is lost) and a finishing lane
// restarts from Env::init(recorded draws), per lane, in place of the cooperative reset.  Ring protocol, roles, clip,
// constraint check, dynamics, reward, flags, tally and stores are the timed kernel's, instruction for instruction.
template <class Env, int OUT, int NP, bool NOISE = false>
__global__ void __launch_bounds__(192 * NP, 1) split_rollout_kernel(const RolloutArgs q)
{
    constexpr int S = Env::S, A = Env::A, KS = Env::KS;
    // step noise, if any, in ChemicalReactor's shape (two normals per step from one Philox block per two steps); an env
    // without step noise (RobotAssembly, KS = 0) leaves the producer the action load + clip only
    static_assert(Env::COOP_RESET && !Env::CUSTOM_STEP && S % 4 == 0 && (KS == 0 || (Env::SHARED_STEP_BLOCK && KS == 2)),
                  "three-wave form: cooperative reset, S a multiple of 4, no step noise or ChemicalReactor's");
    constexpr int KN = KS > 0 ? KS : 1;
    using Lds = SplitLds<Env, NP>;
    constexpr int K = Lds::K;
    constexpr int THREADS = 192 * NP;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Lds::BYTES];
    float4 *const s_probit = reinterpret_cast<float4 *>(smem + Lds::OFF_PROBIT);
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    // readfirstlane: the wave index (and everything derived from it: role, ring addresses, row pointers) lives in
    // scalar registers -- the role branches are scalar branches and the row pointers advance on the scalar unit
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned pair = wave % NP, role = wave / NP;          // 0 integrator, 1 producer, 2 recorder
    lds_u32_t *const sync = (lds_u32_t *)(smem + Lds::OFF_SYNC) + pair * 4;
    float *const s_hi = reinterpret_cast<float *>(smem + Lds::OFF_HI) + pair * (K * Lds::HI_SLOT);
    float *const s_ih = reinterpret_cast<float *>(smem + Lds::OFF_IH) + pair * (K * Lds::IH_SLOT);
    for (int i_ = (int)tid; i_ < 768; i_ += THREADS) s_probit[i_] = NIG_PROBIT[i_];
    if (tid < NP * 4) reinterpret_cast<uint32_t *>(smem + Lds::OFF_SYNC)[tid] = 0u;
    __syncthreads();

    const StepArgs &p = q.s;
    const uint32_t base = (blockIdx.x + q.block0) * (64u * NP) + pair * 64u;     // the triple's first lane
    const uint32_t t_base = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off + (uint32_t)q.it0;   // local step i uses t_base + i + 1
    const int n = q.n_steps - q.it0;                                             // local steps [0, n)

    if (role == 0) {
        // ------------------------------------------------------------------ integrator
        float *const s_img = reinterpret_cast<float *>(smem + Lds::OFF_IMG) + pair * (Env::RESET_ROWS * 64);
        unsigned char *const s_wlist = smem + Lds::OFF_WLIST + pair * 64;
        int step = (int)((p.ctr + base)[lane] & NIG_CTR_STEP_MASK);     // the recorder keeps (and stores) the whole counter word
        float s[S], nx[S];
#pragma unroll
        for (int k = 0; k < S; ++k) s[k] = (p.state + base + k * p.ld_state)[lane];
        // Inputs are read one step AHEAD: the ring counter and (speculatively) the slot of step i + 1 are read
        // while step i is integrated; the counter is looked at afterwards, and only if the producer had not got
        // that far (it normally is several steps ahead) the wave spins and reads the slot again.  A read issued after
        // the counter read sees at least what the counter promised (DS operations of a wave execute in order).
        // Two input register sets, the loop unrolled by two: no copies between them.
        float in0[KS + A], in1[KS + A];
        auto read_inputs = [&](const int sl, float (&dst)[KS + A]) __attribute__((always_inline)) {
            const float *hi = s_hi + sl * Lds::HI_SLOT;
#pragma unroll
            for (int k = 0; k < KS + A; ++k) dst[k] = hi[k * 64 + lane];
        };
        int slot = 0;
        auto integrate = [&](const float (&in)[KS + A], float (&in_next)[KS + A], const int i) __attribute__((always_inline)) {
            const int nslot = (slot + 1 == K) ? 0 : slot + 1;
            const uint32_t c_next = split_peek(sync + 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            read_inputs(nslot, in_next);
            double nz[KN];
            float a[A];
            nz[0] = 0.0;
#pragma unroll
            for (int k = 0; k < KS; ++k) nz[k] = (double)in[k];
#pragma unroll
            for (int k = 0; k < A; ++k) a[k] = in[KS + k];
            const uint32_t vb = Env::violated(s, a) & p.cmask;
            Env::dynamics(s, a, nz, p.dt32, p.dt, nx);
            StepResult<Env> res;
            post_core<Env, float>(nx, a, vb, step, p.max_steps, res);          // the reward part is dead here
            const bool done = res.terminated || res.truncated;
            float *ih = s_ih + slot * Lds::IH_SLOT;
            v4f *row = reinterpret_cast<v4f *>(ih) + lane * (S / 4);
#pragma unroll
            for (int k = 0; k < S / 4; ++k) { v4f v = {nx[4 * k], nx[4 * k + 1], nx[4 * k + 2], nx[4 * k + 3]}; row[k] = v; }
            reinterpret_cast<uint32_t *>(ih + S * 64)[lane] = vb;
            split_post(sync + 1, (uint32_t)i + 1u, lane);
            step = done ? 0 : step + 1;
            if constexpr (NOISE) {
                if (done) {                        // IndustrialEnv.reset on the recorded draws of this step's row set (base.py:133-155)
                    double rn[Env::KR > 0 ? Env::KR : 1];
                    const double *rnr = p.reset_noise + (size_t)(q.it0 + i) * q.nz_reset_stride + base;
#pragma unroll
                    for (int k = 0; k < Env::KR; ++k) rn[k] = (rnr + (size_t)k * p.ld_noise)[lane];
                    Env::init(rn, nx);
                }
            } else {
            const unsigned long long m = __ballot(done);
            if (m != 0ull)
                coop_reset<Env>(m, done, lane, s_img, s_wlist, p.env0 + (uint64_t)base, t_base + (uint32_t)i + 1u,
                                p.seed_lo, p.seed_hi, s_probit, nx);
            }
#pragma unroll
            for (int k = 0; k < S; ++k) s[k] = nx[k];
            if (i + 1 < n && __builtin_amdgcn_readfirstlane(c_next) < (uint32_t)i + 2u) {   // rare: the producer fell behind
                split_wait(sync + 0, (uint32_t)i + 2u);
                read_inputs(nslot, in_next);
            }
            slot = nslot;
        };
        __builtin_amdgcn_s_waitcnt(0x0F70);       // state loads done: no vmcnt wait is carried into the loop
        __builtin_amdgcn_s_setprio(3);            // the critical path of the three: the SIMD's arbiter serves this wave first
        if (n > 0) { split_wait(sync + 0, 1u); read_inputs(0, in0); }
        int i = 0;
        for (; i + 2 <= n; i += 2) { integrate(in0, in1, i); integrate(in1, in0, i + 1); }
        if (i < n) integrate(in0, in1, i);
#pragma unroll
        for (int k = 0; k < S; ++k) (p.state + base + k * p.ld_state)[lane] = s[k];
        NIG_RING_REPORT(p.ring_err, sync, lane);
        return;
    }

    if (role == 1) {
        // ------------------------------------------------------------------ producer
        const uint64_t gi = p.env0 + (uint64_t)(base + lane);
        const float *ring = p.actions + base;
#ifndef NIG_SPLIT_LA
#define NIG_SPLIT_LA 4
#endif
        constexpr int LA = NIG_SPLIT_LA;                       // actions are loaded LA steps before they are handed on
        float buf[LA][A];
        uint32_t kept0 = 0u, kept1 = 0u;                       // words 2-3 of the current pair's Philox block
        int aslot = q.it0 % q.ring_len;                        // ring slot of the action loaded next
        const float *act_next = ring + (size_t)aslot * q.slot_stride;
        auto load_action = [&](float (&ab)[A]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < A; ++k) ab[k] = (act_next + k * p.ld_act)[lane];
            aslot = (aslot + 1 == q.ring_len) ? 0 : aslot + 1;
            act_next = (aslot == 0) ? ring : act_next + q.slot_stride;
        };
        int pslot = 0;
        uint32_t freed = 0u;                                   // slots the recorder is known to be done with
        // slot of local step j (r = j mod LA, static: action register set and position in the pair of launch counters)
        auto produce = [&](auto r_tag, const int j) __attribute__((always_inline)) {
            constexpr int r = decltype(r_tag)::value;
            float (&ab)[A] = buf[r];
            typename Env::fast_noise_t nz[KN];
            if constexpr (KS > 0 && NOISE) {
                const double *nzr = p.step_noise + (size_t)(q.it0 + j) * q.nz_step_stride + base;
#pragma unroll
                for (int k = 0; k < KS; ++k) nz[k] = (nzr + (size_t)k * p.ld_noise)[lane];
            } else if constexpr (KS > 0) {
                ProbitFetch pf[KN];
                if constexpr ((r & 1) == 0) {     // first step of a pair: the pair's Philox block
                    const u32x4 x = Env::step_block(make_key(gi, t_base + (uint32_t)j + 1u, p.seed_lo, p.seed_hi, s_probit));
                    Env::step_noise_fetch(x.x, x.y, s_probit, pf);
                    kept0 = x.z; kept1 = x.w;
                } else {
                    Env::step_noise_fetch(kept0, kept1, s_probit, pf);
                }
                Env::step_noise_eval(pf, nz);
            }
            float a[A];
#pragma unroll
            for (int k = 0; k < A; ++k) a[k] = ab[k];
            clip_action<Env, float>(a);
            if (freed + (uint32_t)K < (uint32_t)j + 1u) freed = split_wait(sync + 2, (uint32_t)(j + 1 - K));   // the slot's previous use
            float *hi = s_hi + pslot * Lds::HI_SLOT;
#pragma unroll
            for (int k = 0; k < KS; ++k) hi[k * 64 + lane] = (float)nz[k];    // exact: the fast-mode noise IS a float (nig_envs.hpp)
#pragma unroll
            for (int k = 0; k < A; ++k) hi[(KS + k) * 64 + lane] = a[k];
            NIG_RING_FAULT_GUARD(p.hflags, j) split_post(sync + 0, (uint32_t)j + 1u, lane);
            load_action(ab);                      // this register set's next use: local step j + LA
            pslot = (pslot + 1 == K) ? 0 : pslot + 1;
        };
        // the loop is unrolled LA times so that every action register set and the position in the pair of launch counters
        // are compile-time (LA even)
        static_assert(LA % 2 == 0 && LA >= 2 && LA <= 8, "unrolled below for even depths up to 8");
#pragma unroll
        for (int j = 0; j < LA; ++j) load_action(buf[j]);
        auto produce_k = [&](auto k_tag, const int j0) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value;
            if constexpr (k < LA) produce(std::integral_constant<int, k>{}, j0 + k);
        };
        int j = 0;
        for (; j + LA <= n; j += LA) {
            produce_k(std::integral_constant<int, 0>{}, j); produce_k(std::integral_constant<int, 1>{}, j);
            produce_k(std::integral_constant<int, 2>{}, j); produce_k(std::integral_constant<int, 3>{}, j);
            produce_k(std::integral_constant<int, 4>{}, j); produce_k(std::integral_constant<int, 5>{}, j);
            produce_k(std::integral_constant<int, 6>{}, j); produce_k(std::integral_constant<int, 7>{}, j);
        }
        // tail: at most LA - 1 steps
        auto tail_k = [&](auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value;
            if constexpr (k < LA - 1) { if (j + k < n) produce(std::integral_constant<int, k>{}, j + k); }
        };
        tail_k(std::integral_constant<int, 0>{}); tail_k(std::integral_constant<int, 1>{}); tail_k(std::integral_constant<int, 2>{});
        tail_k(std::integral_constant<int, 3>{}); tail_k(std::integral_constant<int, 4>{}); tail_k(std::integral_constant<int, 5>{});
        tail_k(std::integral_constant<int, 6>{});
        NIG_RING_REPORT(p.ring_err, sync, lane);
        return;
    }

    // ---------------------------------------------------------------------- recorder
    // Everything IndustrialEnv.step does after the dynamics (base.py:176-213), the episode bookkeeping and the outputs.
    uint32_t ctr = (p.ctr + base)[lane];                       // mirrors the integrator's counter word
    const bool tally = p.tally != nullptr;
    using ret_t = std::conditional_t<Env::RET_F32, float, double>;
    ret_t ret = tally ? (ret_t)(p.ep_ret + base)[lane] : (ret_t)0;
    LaneTally lt;
    lt.clear();
    float *rew_row = p.reward ? p.reward + base + (size_t)q.it0 * q.out_stride : nullptr;
    uint32_t *fl_row = p.flags ? p.flags + base + (size_t)q.it0 * q.out_stride : nullptr;
    float *obs_row = nullptr;
    if constexpr (OUT == 3) obs_row = q.obs_out + (size_t)q.it0 * q.obs_step_stride + (size_t)base * S;
    if constexpr (OUT == 2) obs_row = q.obs_out + (size_t)q.it0 * q.obs_step_stride + base;
    int cslot = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);           // counter / return loads done: the loop only ever stores
    uint32_t seen = 0u;
    for (int i = 0; i < n; ++i) {
        if (seen < (uint32_t)i + 1u) seen = split_wait(sync + 1, (uint32_t)i + 1u);
        // what the integrator left: post-dynamics state (own row, and the wave's rows in lane-contiguous order),
        // violation bits of the pre-state; and the clipped action the step was handed (its P -> I slot is intact
        // until this wave says so)
        const float *ih = s_ih + cslot * Lds::IH_SLOT;
        const float *hi = s_hi + cslot * Lds::HI_SLOT;
        float nx[S], a[A];
        const v4f *row = reinterpret_cast<const v4f *>(ih) + lane * (S / 4);
#pragma unroll
        for (int k = 0; k < S / 4; ++k) { const v4f v = row[k]; nx[4 * k] = v.x; nx[4 * k + 1] = v.y; nx[4 * k + 2] = v.z; nx[4 * k + 3] = v.w; }
        const uint32_t vb = reinterpret_cast<const uint32_t *>(ih + S * 64)[lane];
#pragma unroll
        for (int k = 0; k < A; ++k) a[k] = hi[(KS + k) * 64 + lane];
        v4f tr[S / 4];
        if constexpr (OUT == 3) {                  // see rollout_body, OUT == 3
#pragma unroll
            for (int k = 0; k < S / 4; ++k) tr[k] = reinterpret_cast<const v4f *>(ih)[lane + 64u * k];
        }
        split_post(sync + 2, (uint32_t)i + 1u, lane);          // (DS order: the reads above execute before this write)
        const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
        StepResult<Env> res;
#ifdef NIG_DIAG_SPLIT_REC_NOCOMPUTE    // (diagnostic builds only, profiles/r05: the recorder as a pure store wave -- no reward, penalties, flags or
        // tally arithmetic, the same LDS reads and the same stores -- the upper bound of what a fourth, store-only wave per triple
        // could give the headline; results are garbage)
        res.reward = nx[0] + a[0]; res.terminated = false; res.truncated = false; res.nviol = (int)(vb & 1u); res.ncrit = 0; res.viol_bits = vb; res.shutdown = false;
#else
        post_core<Env, float>(nx, a, vb, step_pre, p.max_steps, res);
#endif
        const int step = step_pre + 1;
        const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;
        const bool done = res.terminated || res.truncated;
        const uint32_t fl = pack_flags<Env>(res, step) | (done ? NIG_FLAG_DID_RESET : 0u);
        ctr = (uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT);
        if (tally) ret = ret + (ret_t)res.reward;
        if constexpr (OUT == 3) {
            v4f *oo = reinterpret_cast<v4f *>(obs_row);
#pragma unroll
#ifdef NIG_DIAG_SPLIT_PLAINSTORE       // (diagnostic builds only: ordinary instead of streaming stores for the trajectory rows)
            for (int k = 0; k < S / 4; ++k) oo[lane + 64u * k] = tr[k];
#else
            for (int k = 0; k < S / 4; ++k) stream_store(oo + lane + 64u * k, tr[k]);
#endif
        }
        if constexpr (OUT == 2) {
#pragma unroll
            for (int k = 0; k < S; ++k) stream_store(obs_row + k * q.ld_obs_out + lane, nx[k]);
        }
        if constexpr (OUT >= 1) {
            stream_store(rew_row + lane, (float)res.reward);
            stream_store(fl_row + lane, fl);
            rew_row += q.out_stride; fl_row += q.out_stride;
        }
        if constexpr (OUT >= 2) obs_row += q.obs_step_stride;
        if (done) {
            lt.life += (long long)viol_ep;
            if (tally) { lt.episode((double)ret, step, viol_ep, res.ncrit); ret = (ret_t)0; }
            ctr = 0u;
        }
        cslot = (cslot + 1 == K) ? 0 : cslot + 1;
    }
    (p.ctr + base)[lane] = ctr;
    if (lt.life != 0) (p.life_viol + base)[lane] += lt.life;
    if (tally) {
        (p.ep_ret + base)[lane] = (double)ret;
        if (lt.episodes > 0) lt.merge(p.tally + base + lane, p.ld, p.n_en);
    }
    NIG_RING_REPORT(p.ring_err, sync, lane);
}

// whole blocks of 64*NP lanes, PAIRED start; the caller (launch_rollout_form) has checked that the form applies
template <class Env, int NP, bool NOISE = false>
static void launch_split_blocks(int out_mode, const RolloutArgs &q, unsigned grid, hipStream_t st)
{
    if constexpr (NOISE) {                         // injected draws: the row-major full-output variant only (nig_rollout_noise)
        hipLaunchKernelGGL((split_rollout_kernel<Env, 3, NP, true>), dim3(grid), dim3(192 * NP), 0, st, q);
        return;
    }
    switch (out_mode) {
    case 0: hipLaunchKernelGGL((split_rollout_kernel<Env, 0, NP>), dim3(grid), dim3(192 * NP), 0, st, q); break;
    case 1: hipLaunchKernelGGL((split_rollout_kernel<Env, 1, NP>), dim3(grid), dim3(192 * NP), 0, st, q); break;
    case 2: hipLaunchKernelGGL((split_rollout_kernel<Env, 2, NP>), dim3(grid), dim3(192 * NP), 0, st, q); break;
    default: hipLaunchKernelGGL((split_rollout_kernel<Env, 3, NP>), dim3(grid), dim3(192 * NP), 0, st, q); break;
    }
}

}  // namespace nig
