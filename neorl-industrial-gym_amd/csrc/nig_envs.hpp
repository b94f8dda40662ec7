// nig_envs.hpp -- device physics of the three working environments, one env instance
// per wavefront lane, state held in VGPRs.
//
// Each struct restates, for the GPU, the four hooks the reference's step template calls
// (environments/base.py:74-92): _get_initial_state, _dynamics, _compute_reward, _is_done,
// plus the env's safety-constraint check functions.  Paths below are relative to
// /root/reference/src/neorl_industrial/.  Arithmetic follows NumPy 2.x semantics with
// float32 actions: Python scalars are weak (rounded to the array
// dtype), builtin min/max keep their first argument on ties, np.sum over exactly 8
// contiguous elements is the pairwise tree.  The translation unit is compiled with
// -ffp-contract=off so a*b+c is two roundings, as in NumPy.
#pragma once
#include <type_traits>

#include "nig_detmath.hpp"

namespace nig {

// Result of one IndustrialEnv.step for one lane (defined below); R = the type the reward has at that point of the
// reference's arithmetic: Env::reward_t with float32 actions, double with float64 actions.
template <class Env, class R = typename Env::reward_t> struct StepResult;

// Python builtin max(a, b) / min(a, b): second argument only if strictly greater / less.
template <class T> __device__ __forceinline__ T pymax(T a, T b) { return (b > a) ? b : a; }
template <class T> __device__ __forceinline__ T pymin(T a, T b) { return (b < a) ? b : a; }

// np.sum over 8 contiguous elements (NumPy pairwise_sum with n == 8)
template <class T>
__device__ __forceinline__ T sum8(const T (&x)[8])
{
    return ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
}

// =================================================================================
// ChemicalReactor-v0  (environments/chemical_reactor.py), S=12 A=3, all float32
// =================================================================================
struct ChemicalReactor {
    // Only ~0.3 % of lanes finish per step, but that is a finishing lane in ~17 % of the wave-steps, and the
    // divergent in-place reset (2 generator blocks, 8 normals, 8 fp64 initial values: ~300 instructions) then runs
    // for one or two active lanes: ~50 instructions per wave-step on average, 14 % of the step.  Cooperative form:
    // work item = (lane, generator block) -> four state rows through a wave-private LDS image: one short pass.
    static constexpr bool COOP_RESET = true;
    static constexpr int RESET_ITEMS = 2, RESET_ROWS = 8;
    using fast_noise_t = double;
    static constexpr int ID = 0, S = 12, A = 3, KS = 2, KR = 8, MAX_STEPS = 500;
    static constexpr bool COMPACT_RESET = false;
    static constexpr bool CUSTOM_STEP = false, RET_F32 = true;   // episode_return stays np.float32 (utils.py:99 under NEP 50)
    static constexpr int STEP_BLOCK = 256, STEP_WAVES = 6;          // waves per SIMD the step kernel is compiled for (no spills at this cap: 80 VGPRs)
    static constexpr int ROLLOUT_WAVES = 3;       // same, for the fused rollout kernels (four action register sets in flight)
    static constexpr bool SPLIT_ROLLOUT = true;   // small batches: integrator + helper wave per 64 lanes (nig_split.hpp)
    using reward_t = float;   // reward stays np.float32 (0.0 + f32 under NEP 50), :240-269
    __device__ static constexpr float act_low(int) { return -1.0f; }      // base.py:66-71
    __device__ static constexpr float act_high(int) { return 1.0f; }

    // chemical_reactor.py:38-60 (penalty, critical) in list order
    __device__ static constexpr float penalty(int k) { return k == 0 ? -100.0f : (k == 1 ? -50.0f : -25.0f); }
    static constexpr uint32_t CRIT_MASK = 0x3u;

    // _get_initial_state :89-107 -- fp64 "mean + draw", stored float32
    __device__ static void init(const double (&n)[KR], float (&s)[S])
    {
        s[0] = (float)(320.0 + n[0]);     s[1] = (float)(253312.5 + n[1]);
        s[2] = (float)(50.0 + n[2]);      s[3] = (float)(30.0 + n[3]);
        s[4] = (float)(0.5 + n[4]);       s[5] = (float)(95.0 + n[5]);
        s[6] = (float)(295.0 + n[6]);     s[7] = 0.0f; s[8] = 0.0f; s[9] = 0.0f;
        s[10] = (float)(60.0 + n[7]);     s[11] = 0.0f;
    }
    // fast mode: draws in reference call order; since "nig-philox-v2" the draw is the float32 product sd * z (as the
    // step noise), widened to the double init() takes.  The means are float32-representable, so (float)(mean + draw) is
    // the float32 sum mean + draw rounded once: reset_item below computes it with one v_add_f32.
    __device__ static void draw_init(const RngKey &k, double (&n)[KR])
    {
        float z[KR];
        gen_normals<KR>(k, STREAM_RESET, z);
        n[0] = (double)(2.0f * z[0]);         n[1] = (double)(10000.0f * z[1]);
        n[2] = (double)(5.0f * z[2]);         n[3] = (double)(3.0f * z[3]);
        n[4] = (double)(0.1f * z[4]);         n[5] = (double)(2.0f * z[5]);
        n[6] = (double)(1.0f * z[6]);         n[7] = (double)(5.0f * z[7]);
    }
    // One work item of a cooperative reset: generator block `blk` (0 or 1) of the lane with key `k` -> image rows
    // 4 blk .. 4 blk + 3 = initial values of state rows {0,1,2,3} / {4,5,6,10}; same values, operation by operation,
    // as draw_init + init.
    __device__ static void reset_item(const RngKey &k, uint32_t blk, float *img, unsigned owner)
    {
        const u32x4 x = k.block(STREAM_RESET + blk);
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
        const bool hi = blk != 0u;
        const float sd[4] = {hi ? 0.1f : 2.0f, hi ? 2.0f : 10000.0f, hi ? 1.0f : 5.0f, hi ? 5.0f : 3.0f};          // :93-103
        const float mean[4] = {hi ? 0.5f : 320.0f, hi ? 95.0f : 253312.5f, hi ? 295.0f : 50.0f, hi ? 60.0f : 30.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q)     // == (float)((double)mean + (double)(sd * z)): exact in double, rounded once
            img[(4u * blk + (uint32_t)q) * 64u + owner] = mean[q] + sd[q] * probit_normal(w[q], k.tab);
    }
    __device__ static void reset_readback(const float *img, unsigned lane, float (&s)[S])
    {
#pragma unroll
        for (int i = 0; i < 7; ++i) s[i] = img[i * 64 + lane];
        s[7] = 0.0f; s[8] = 0.0f; s[9] = 0.0f;
        s[10] = img[7 * 64 + lane];
        s[11] = 0.0f;
    }
    // Two draws per step: launch counters 2k-1 and 2k share ONE Philox block (counter word t = k, words
    // 0-1 for the odd step, 2-3 for the even one; a fresh env's first step is t = 1) -- a block is ~64
    // of this env's ~400 instructions per step, and half of every block used to be thrown away
    // (fused rollout without per-step outputs: +8 %).
    static constexpr bool SHARED_STEP_BLOCK = true;
    __device__ static u32x4 step_block(const RngKey &k)            // the block of the pair that holds counter k.t
    {
        RngKey kk = k;
        kk.t = (k.t + 1u) >> 1;
        return kk.block(STREAM_STEP);
    }
    __device__ static void step_noise_fetch(uint32_t w0, uint32_t w1, const float4 *tab, ProbitFetch (&f)[KS])
    {
        f[0] = probit_fetch(w0, tab); f[1] = probit_fetch(w1, tab);
    }
    __device__ static void step_noise_eval(const ProbitFetch (&f)[KS], double (&n)[KS])
    {
        // fast-mode step noise is a float32 product (sd * z), handed on as the double the dynamics' noise
        // argument is (parity mode injects the reference's fp64 draws there): one VALU instead of four fp64 ones
        n[0] = (double)(0.1f * probit_eval(f[0]));           // temp_noise_std / 10, :149
        n[1] = (double)(500.0f * probit_eval(f[1]));         // pressure_noise_std / 10, :159
    }
    __device__ static void step_noise(uint32_t w0, uint32_t w1, const float4 *tab, double (&n)[KS])
    {
        ProbitFetch f[KS];
        step_noise_fetch(w0, w1, tab, f);
        step_noise_eval(f, n);
    }
    __device__ static void draw_step(const RngKey &k, double (&n)[KS])
    {
        const u32x4 x = step_block(k);
        const bool second = (k.t & 1u) == 0;
        step_noise(second ? x.z : x.x, second ? x.w : x.y, k.tab, n);
    }

    // constraint checks on the PRE-state, :292-305; bit k set = violated
    __device__ static uint32_t violated(const float (&s)[S], const float (&)[A])
    {
        uint32_t v = 0;
        v |= (s[0] <= 350.0f) ? 0u : 1u;
        v |= (s[1] <= 506625.0f) ? 0u : 2u;
        v |= (20.0f <= s[10] && s[10] <= 90.0f) ? 0u : 4u;
        return v;
    }

    // _dynamics :109-226
    __device__ static void dynamics(const float (&s)[S], const float (&a)[A], const double (&nz)[KS],
                                    float /*dt32*/, double /*dt: hard-coded 0.1 upstream, :68*/, float (&o)[S])
    {
        const float T = s[0], P = s[1], cool = s[2], feed = s[3], conc = s[4], cat = s[5];
        const float hx = s[6], relief = s[7], estop = s[8], alarm = s[9], level = s[10], bt = s[11];
        const bool manual = estop < 0.5f;                               // :126
        const float hp = manual ? a[0] * 50000.0f : -10000.0f;          // :127 / :132
        const float cadj = manual ? a[1] * 0.1f : 0.1f;                 // :128 / :133
        const float fadj = manual ? a[2] * 0.1f : -0.1f;                // :129 / :134
        const float kc = (0.1f * conc) * fdiv_c(cat, 100.0f);                // shared prefix of :137-139 and :175-177
        const float rh = kc * 10000.0f;
        const float ch = ((cool * 100.0f) * (T - hx)) * 0.1f;           // :141
        float dT = fdiv_c((hp + rh) - ch, 418000.0f);                        // :143-146 (4.18e3*1000*0.1 -> f32)
        dT = dT + (float)nz[0];                                         // :149
        const float nT = T + dT * 0.1f;                                 // :151
        float nP = P * (nT / T) + ((conc * 0.1f) * 1000.0f) * 0.1f;     // :155-158
        nP = nP + (float)nz[1];                                         // :159
        const float nrel = pymax(0.0f, pymin(100.0f, relief + (nP - 506625.0f) * 0.001f));  // :162-163
        if (nrel > 0.0f) nP = pymax(101325.0f, nP - (nrel * 0.01f) * 10000.0f);              // :166-168
        const float ncool = pymax(10.0f, pymin(100.0f, cool + cadj));   // :171
        const float nfeed = pymax(5.0f, pymin(50.0f, feed + fadj));     // :172
        const float rr = kc * det_expf(fdiv_c(-(nT - 320.0f), 20.0f));       // :175-178
        const float nconc = pymax(0.0f, conc + (rr - nfeed * 0.001f) * 0.1f);   // :180-182
        const float ncat = pymax(50.0f, cat - ((nT > 340.0f) ? 0.001f : 0.0001f)); // :185-186
        const float nhx = hx + (0.1f * ((290.0f + cool * 0.1f) - hx)) * 0.1f;   // :189-190
        const bool warn = (nT > 345.0f) || (nP > 480000.0f);            // :196
        const bool trip = (nT > 350.0f) || (nP > 506625.0f);            // :199
        const float nalarm = (warn || trip) ? 1.0f : alarm;
        const float nestop = trip ? 1.0f : estop;
        const float nlevel = pymax(0.0f, pymin(100.0f, level + ((nfeed - 20.0f) * 0.1f) * 0.1f)); // :204-205
        o[0] = nT; o[1] = nP; o[2] = ncool; o[3] = nfeed; o[4] = nconc; o[5] = ncat; o[6] = nhx;
        o[7] = nrel; o[8] = nestop; o[9] = nalarm; o[10] = nlevel; o[11] = bt + 0.1f;       // :208
    }

    // ---- float64 actions (what the reference's own callers pass: get_dataset :364-393, baseline agents; base.py:167
    // clips without casting): every expression touching an action element is float64 under NumPy >= 2 and float64
    // spreads until a value is stored into the float32 state (:209-224).  Pinned by tests/golden/cr_g5.npz, cr_g6.npz.
    static constexpr bool HAS_ACT64 = true;
    __device__ static uint32_t violated(const float (&s)[S], const double (&)[A])
    {
        const float none[A] = {0.0f, 0.0f, 0.0f};
        return violated(s, none);
    }
    __device__ static void dynamics(const float (&s)[S], const double (&a)[A], const double (&nz)[KS],
                                    float dt32, double dt, float (&o)[S])
    {
        const float T = s[0], P = s[1], cool = s[2], feed = s[3], conc = s[4], cat = s[5];
        const float hx = s[6], relief = s[7], estop = s[8], alarm = s[9], level = s[10], bt = s[11];
        if (!(estop < 0.5f)) {                                           // emergency branch :131-134: the action is not read
            const float none[A] = {0.0f, 0.0f, 0.0f};
            dynamics(s, none, nz, dt32, dt, o);
            return;
        }
        const double hp = a[0] * 50000.0, cadj = a[1] * 0.1, fadj = a[2] * 0.1;     // :127-129 float64
        const float kc = (0.1f * conc) * fdiv_c(cat, 100.0f);                       // :137-139 float32
        const float rh = kc * 10000.0f;
        const float ch = ((cool * 100.0f) * (T - hx)) * 0.1f;                       // :141 float32
        double dT = ((hp + (double)rh) - (double)ch) / 418000.0;                    // :143-146
        dT = dT + nz[0];                                                            // :149
        const double nT = (double)T + dT * 0.1;                                     // :151
        double nP = (double)P * (nT / (double)T) + (double)(((conc * 0.1f) * 1000.0f) * 0.1f);   // :155-158
        nP = nP + nz[1];                                                            // :159
        const double nrel = pymax(0.0, pymin(100.0, (double)relief + (nP - 506625.0) * 0.001));  // :162-163
        if (nrel > 0.0) nP = pymax(101325.0, nP - (nrel * 0.01) * 10000.0);         // :166-168
        const double ncool = pymax(10.0, pymin(100.0, (double)cool + cadj));        // :171
        const double nfeed = pymax(5.0, pymin(50.0, (double)feed + fadj));          // :172
        const double rr = (double)kc * det_exp((-(nT - 320.0)) / 20.0);             // :175-178 np.exp on a float64 scalar
        const double nconc = pymax(0.0, (double)conc + (rr - nfeed * 0.001) * 0.1); // :180-182
        const float ncat = pymax(50.0f, cat - ((nT > 340.0) ? 0.001f : 0.0001f));   // :185-186 float32
        const float nhx = hx + (0.1f * ((290.0f + cool * 0.1f) - hx)) * 0.1f;       // :189-190 float32
        const bool warn = (nT > 345.0) || (nP > 480000.0);                          // :196
        const bool trip = (nT > 350.0) || (nP > 506625.0);                          // :199
        const double nlevel = pymax(0.0, pymin(100.0, (double)level + ((nfeed - 20.0) * 0.1) * 0.1));   // :204-205
        o[0] = (float)nT; o[1] = (float)nP; o[2] = (float)ncool; o[3] = (float)nfeed; o[4] = (float)nconc;
        o[5] = ncat; o[6] = nhx; o[7] = (float)nrel; o[8] = trip ? 1.0f : estop; o[9] = (warn || trip) ? 1.0f : alarm;
        o[10] = (float)nlevel; o[11] = bt + 0.1f;                                   // :208-224 np.array(..., dtype=float32)
    }
    __device__ static double reward(const float (&n)[S], const double (&a)[A])
    {
        const float none[A] = {0.0f, 0.0f, 0.0f};
        const float r = reward(n, none);                                 // float32 up to the action penalty (- 0 * 0.1 changes nothing)
        const double ap = ((0.0 + fabs(a[0])) + fabs(a[1])) + fabs(a[2]);   // :268 float64
        return (double)r - ap * 0.1;                                     // :269
    }

    // _compute_reward :228-270 on the NEXT state
    __device__ static float reward(const float (&n)[S], const float (&a)[A])
    {
        float r = n[4] * 100.0f;                                        // 0.0 + x == x, :242
        r = r - fabsf(n[0] - 320.0f) * 0.5f;                            // :245-246
        r = r - fdiv_c(fabsf(n[1] - 253312.5f), 1000.0f) * 0.1f;             // :249-250
        r = r + fdiv_c(n[5], 100.0f) * 10.0f;                                // :253
        const bool band = (30.0f <= n[10]) && (n[10] <= 80.0f);         // :256
        r = band ? (r + 5.0f) : (r - fabsf(n[10] - 55.0f) * 0.2f);      // :257-259
        r = (n[9] > 0.5f) ? (r - 50.0f) : r;                            // :262-263
        r = (n[8] > 0.5f) ? (r - 200.0f) : r;                           // :264-265
        const float ap = (fabsf(a[0]) + fabsf(a[1])) + fabsf(a[2]);     // np.sum n<8 sequential, :268
        return r - ap * 0.1f;                                           // :269
    }

    // _is_done :272-290
    __device__ static bool done(const float (&n)[S])
    {
        return (n[8] > 0.5f) || (n[10] < 5.0f) || (n[10] > 95.0f) || (n[11] > 50.0f);
    }
};

// =================================================================================
// PowerGrid-v0  (environments/power_grid.py), S=32 A=8
// =================================================================================
struct PowerGrid {
    static constexpr int ID = 1, S = 32, A = 8, KS = 23, KR = 31, MAX_STEPS = 1000;
    // ~18 % of lanes finish per step (episodes of ~6 steps): ~11 lanes of every wave reset in every step.
    // COOP_RESET: a wave produces the initial states of its finishing lanes cooperatively, one work item =
    // (lane, generator block) -> 4 state rows, through a wave-private LDS image -- no block barrier, and
    // ~70 % lane utilisation instead of a whole wave running the 8-block reset path for a few lanes.
    static constexpr bool COMPACT_RESET = false, COOP_RESET = true;
    static constexpr int RESET_ITEMS = 6, RESET_ROWS = S;   // 6 generator blocks per reset ("nig-philox-v3": the 8 load factors ride in the normals' spare low bytes; v2 drew 2 more blocks for them)
    static constexpr bool SHARED_STEP_BLOCK = false;
    // Fast-mode step noise stays float32 (sd * z is a float32 product): (float)((double)s + (double)n32) is the
    // correctly rounded float32 sum s + n32 for ANY two floats (exact in double when the exponents are within
    // 28 of each other; beyond that both roundings return s), so the reference's fp64 add of :136-144 needs no
    // fp64 instruction when the noise is float32-valued.  Parity mode (injected fp64 draws) keeps the fp64 adds.
    using fast_noise_t = float;
    static constexpr bool CUSTOM_STEP = false, RET_F32 = false;
    static constexpr int STEP_WAVES = 2, STEP_BLOCK = 512;   // big batches: two 512-thread blocks per CU = 4 waves per SIMD (128 VGPRs), LDS 2 x 76 KiB
    static constexpr int ROLLOUT_WAVES = 2;       // same, for the fused rollout kernels
    // Big auto-reset batches run the LDS-resident form (nig_pg_lds.hpp): 512-thread blocks, four waves per SIMD.
    static constexpr int WIDE_ROLLOUT_BLOCK = 512, WIDE_ROLLOUT_WAVES = 2;
    static constexpr bool PAIR_ROLLOUT = true;     // batches of at most one 256-lane block per CU: stepping + producer wave per 64 lanes (nig_pg_lds.hpp)
    static constexpr bool TALLY_ATOMIC = false;   // 18 % of the lanes finish every step: see tally_atomic (nig_kernels.hpp)
    using reward_t = double;  // float(total_reward), :177
    __device__ static constexpr float act_low(int) { return -1.0f; }
    __device__ static constexpr float act_high(int) { return 1.0f; }

    __device__ static constexpr double penalty(int k) { return k == 0 ? -50.0 : (k == 1 ? -30.0 : -20.0); }  // :53-72
    static constexpr uint32_t CRIT_MASK = 0x3u;

    __device__ static constexpr double base_load(int i)   // :82
    {
        return i == 0 ? 50. : i == 1 ? 60. : i == 2 ? 45. : i == 3 ? 55. : i == 4 ? 40. : i == 5 ? 65. : i == 6 ? 35. : 50.;
    }
    __device__ static constexpr double gen_cost(int i)    // :88
    {
        return i == 0 ? 25. : i == 1 ? 30. : i == 2 ? 28. : i == 3 ? 35. : i == 4 ? 32. : i == 5 ? 27. : i == 6 ? 40. : 33.;
    }

    // _get_initial_state :90-110; draws = [8 N(0,.01)] [8 N(0,2)] [8 U(-.2,.2)] [7 N(0,10)]
    __device__ static void init(const double (&n)[KR], float (&s)[S])
    {
        s[0] = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s[1 + i] = (float)(1.0 + n[i]);                              // :98
            s[9 + i] = (float)(base_load(i) + n[8 + i]);                 // :101
            s[17 + i] = (float)(base_load(i) * (1.0 + n[16 + i]));       // :104-105
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) s[25 + i] = (float)n[24 + i];        // :108
    }
    // A load factor's uniform: 16 bits = the low bytes of two words of a reset block (bits 7..0: below the 24 bits a normal
    // takes), as the float32 k / 65536 -- one of the 24-bit uniforms (k << 8) / 2^24, so what was proven for those holds.
    // "nig-philox-v3" (round 4): v2 drew blocks STREAM_RESET + 16 / + 17 for the eight factors -- two of a reset's eight blocks.
    __device__ static float load_factor(uint32_t w_lo, uint32_t w_hi)
    {
        const uint32_t k16 = __builtin_amdgcn_perm(w_hi, w_lo, 0x0c0c0400u);   // byte 0 of w_lo, byte 0 of w_hi, zeros
        return __builtin_fmaf(0.4f / 65536.0f, (float)k16, -0.2f);           // uniform(-0.2, 0.2): low + (high - low) * u, u = k16 / 65536 (the scale is a power of two: same rounding)
    }
    __device__ static void draw_init(const RngKey &k, double (&n)[KR])
    {
        // float32 draws (normals sd * z; the load factor fma(0.4, u, -0.2), exact in float32), widened to the doubles init()
        // takes -- see reset_item_to for what that buys.  Blocks 0-5 of the reset stream, words in order: the 23 normals;
        // block b < 4 also carries load factors 2 b and 2 b + 1 in the low bytes of its words (0, 1) and (2, 3).
        float z[24];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const u32x4 x = k.block(STREAM_RESET + (uint32_t)j);
            z[4 * j] = probit_normal(x.x, k.tab); z[4 * j + 1] = probit_normal(x.y, k.tab); z[4 * j + 2] = probit_normal(x.z, k.tab);
            z[4 * j + 3] = (j < 5) ? probit_normal(x.w, k.tab) : 0.0f;          // z[23] does not exist
            if (j < 4) { n[16 + 2 * j] = (double)load_factor(x.x, x.y); n[17 + 2 * j] = (double)load_factor(x.z, x.w); }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            n[i] = (double)(0.01f * z[i]);
            n[8 + i] = (double)(2.0f * z[8 + i]);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) n[24 + i] = (double)(10.0f * z[16 + i]);
    }
    __device__ static void draw_step(const RngKey &k, float (&n)[KS])
    {
        gen_normals<KS>(k, STREAM_STEP, n);
#pragma unroll
        for (int i = 0; i < 8; ++i) n[i] = 0.005f * n[i];                // :136 (float32 product, see ChemicalReactor); :140 has sd 1.0
#pragma unroll
        for (int i = 0; i < 7; ++i) n[16 + i] = 2.0f * n[16 + i];        // :144
    }
    // the same scaling for normals somebody else generated (the paired form's producer wave): z = the 23 raw normals of
    // gen_normals<KS>(k, STREAM_STEP) in order (a 24th padding entry is ignored)
    __device__ static void scale_step_normals(const float (&z)[24], float (&n)[KS])
    {
#pragma unroll
        for (int i = 0; i < 8; ++i) { n[i] = 0.005f * z[i]; n[8 + i] = z[8 + i]; }
#pragma unroll
        for (int i = 0; i < 7; ++i) n[16 + i] = 2.0f * z[16 + i];
    }
    __device__ static void draw_step(const RngKey &k, double (&n)[KS])
    {
        float z[KS];
        draw_step(k, z);
#pragma unroll
        for (int i = 0; i < KS; ++i) n[i] = (double)z[i];
    }
    // One work item of a cooperative reset: generator block `blk` (0-5) of the lane with key `k` -> the four state rows its
    // words feed (normals z[4 blk + q]: V, gen, line flows) and, for blk < 4, the two load rows whose factors ride in its words'
    // low bytes, stored into column `col` (= img + owner lane) of a [RESET_ROWS][64] image.  Same values, operation by
    // operation, as draw_init + init.  Block 0 also clears row 0 (freq_dev).
    // (A two-phase form with a 16-row image was tried: the rows read back in phase 0 stay live across phase 1's
    // items and pushed the rollout kernel into scratch spills: -13 %.)
    __device__ static void reset_item(const RngKey &k, uint32_t blk, float *img, unsigned owner)
    {
        float *col = img + owner;
        reset_item_to(k, blk, [col](uint32_t row, float v) { col[row * 64u] = v; });
    }
    // the same work item with the destination left to the caller: put(state row, initial value)
    template <class Put>
    __device__ static void reset_item_to(const RngKey &k, uint32_t blk, Put &&put)
    {
        const u32x4 x = k.block(STREAM_RESET + blk);
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
        const bool hi = (blk & 1u) != 0;                                     // second half of an 8-vector
        // init() of draw_init()'s values without a float64 instruction: with a float32 draw d and a float32-representable
        // offset, (float)(off + (double)d) is the exactly computed sum rounded once = off + d in float32, and
        // (float)(b * (1.0 + (double)d)) = the exactly computed b + b d rounded once = fma(b, d, b).
        // V 1.0 + 0.01 z (:98), gen base_load + 2.0 z (:101), flows 10.0 z (:108)
        const float sd = blk < 2u ? 0.01f : (blk < 4u ? 2.0f : 10.0f);
        const uint32_t row0 = (blk < 4u ? 1u : 9u) + 4u * blk;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float off = blk < 2u ? 1.0f : (blk < 4u ? (float)(hi ? base_load(4 + q) : base_load(q)) : 0.0f);
            const float v = off + sd * probit_normal(w[q], k.tab);      // flows: 0.0 + d == d (d is never -0.0)
            if (q < 3 || blk != 5u) put(row0 + (uint32_t)q, v);   // z[23] does not exist
        }
        if (blk == 0u) put(0u, 0.0f);
        if (blk < 4u) {                                                      // loads 2 blk, 2 blk + 1: base * (1 + factor), :104-105
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float nn = load_factor(w[2 * q], w[2 * q + 1]);
                const float b = (float)(blk == 0u ? base_load(q) : blk == 1u ? base_load(2 + q) : blk == 2u ? base_load(4 + q) : base_load(6 + q));
                put(17u + 2u * blk + (uint32_t)q, __builtin_fmaf(b, nn, b));
            }
        }
    }

    __device__ static void reset_readback(const float *img, unsigned lane, float (&n)[S])
    {
#pragma unroll
        for (int k = 0; k < S; ++k) n[k] = img[k * 64 + lane];
    }

    // module-level check functions :10-30 (pre-state, clipped action)
    __device__ static uint32_t violated(const float (&s)[S], const float (&a)[A])
    {
        // all(0.95 <= V <= 1.05) and all(0 <= gen + a <= 100) as "smallest >= lower bound and largest <= upper bound" over
        // minimum / maximum trees of v_minimum3_f32 / v_maximum3_f32 (IEEE 754-2019 minimum / maximum: a NaN anywhere makes
        // both NaN and both compares false -- exactly what the element-wise `all` does with a NaN; eight instructions + two
        // compares per check where sixteen compares and their scalar ands were)
        float vmn = s[1], vmx = s[1];
        float ng0 = s[9] + a[0];                                         // float32 add, :29
        float gmn = ng0, gmx = ng0;
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            vmn = __builtin_elementwise_minimum(vmn, s[1 + i]);           // weak Python floats -> float32
            vmx = __builtin_elementwise_maximum(vmx, s[1 + i]);
            const float ng = s[9 + i] + a[i];
            gmn = __builtin_elementwise_minimum(gmn, ng);
            gmx = __builtin_elementwise_maximum(gmx, ng);
        }
        const bool v_ok = (vmn >= 0.95f) & (vmx <= 1.05f);
        const bool g_ok = (gmn >= 0.0f) & (gmx <= 100.0f);               // vs the fp64 array np.ones(8)*100: (double)x <= 100.0 == x <= 100.0f
        uint32_t v = (fabsf(s[0]) < 0.5f) ? 0u : 1u;                     // :14
        v |= v_ok ? 0u : 2u;
        v |= g_ok ? 0u : 4u;
        return v;
    }

    // _dynamics :112-153
    template <class NZ>
    __device__ static void dynamics(const float (&s)[S], const float (&a)[A], const NZ (&nz)[KS],
                                    float dt, double /*dt64*/, float (&o)[S])
    {
        float ngen[8], load[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float g = s[9 + i] + a[i];                                   // :124 np.clip(gen + a, 0, 100)
            g = (g < 0.0f) ? 0.0f : g;                                   // (compare + select: np.maximum keeps a -0.0, v_maximum3_f32 would not)
            g = __builtin_elementwise_minimum(g, 100.0f);                // np.minimum, NaN handed on: one v_minimum3_f32
            ngen[i] = g;
            load[i] = s[17 + i];
        }
        const float imb = sum8(ngen) - sum8(load);                       // :127-129
        const float fd = fdiv_c((-1.0f * s[0]) + imb, 5.0f);                  // :132
        o[0] = s[0] + fd * dt;                                           // :133
        if constexpr (std::is_same<NZ, float>::value) {                  // float32-valued noise: see fast_noise_t
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                o[1 + i] = s[1 + i] + nz[i];                             // :136-137
                o[9 + i] = ngen[i];
                const float l = s[17 + i] + nz[8 + i];                   // :140-141
                o[17 + i] = (l < 0.0f) ? 0.0f : l;
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) o[25 + i] = s[25 + i] + nz[16 + i];   // :144
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                o[1 + i] = (float)((double)s[1 + i] + nz[i]);            // :136-137 fp64 add, one rounding
                o[9 + i] = ngen[i];
                double l = (double)s[17 + i] + nz[8 + i];                // :140-141
                l = (l < 0.0) ? 0.0 : l;
                o[17 + i] = (float)l;
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) o[25 + i] = (float)((double)s[25 + i] + nz[16 + i]);   // :144
        }
    }

    // ---- float64 actions (power_grid.py:216-233 hands them over; pinned by tests/golden/pg_g5.npz, pg_g6.npz)
    static constexpr bool HAS_ACT64 = true;
    __device__ static uint32_t violated(const float (&s)[S], const double (&a)[A])
    {
        const float none[A] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t v = violated(s, none) & 3u;
        bool g_ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double ng = (double)s[9 + i] + a[i];                   // float64 add, :29
            g_ok = g_ok & (ng >= 0.0) & (ng <= 100.0);
        }
        return v | (g_ok ? 0u : 4u);
    }
    template <class NZ>
    __device__ static void dynamics(const float (&s)[S], const double (&a)[A], const NZ (&nz)[KS],
                                    float dt32, double dt, float (&o)[S])
    {
        const float none[A] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        dynamics(s, none, nz, dt32, dt, o);                              // voltages, loads, line flows never see the action
        double ngen[8];
        float load[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double g = (double)s[9 + i] + a[i];                          // :124 np.clip(gen + a, 0, 100) in float64
            g = (g < 0.0) ? 0.0 : g;
            g = (g > 100.0) ? 100.0 : g;
            ngen[i] = g;
            load[i] = s[17 + i];
        }
        const double imb = sum8(ngen) - (double)sum8(load);              // :127-129
        const double fd = ((double)(-1.0f * s[0]) + imb) / 5.0;          // :132 (-D * f is float32, the sum float64)
        o[0] = (float)((double)s[0] + fd * dt);                          // :133
#pragma unroll
        for (int i = 0; i < 8; ++i) o[9 + i] = (float)ngen[i];
    }
    __device__ static double reward(const float (&n)[S], const double (&a)[A])
    {
        const float none[A] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        double a2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a2[i] = a[i] * a[i];
        return reward(n, none) + (-5.0 * sum8(a2));                      // :173-175 (the float32 form adds -5.0f * 0 = -0.0)
    }

    // _compute_reward :155-177 (float32 terms, fp64 economic term, fp64 total), in the four terms the reference adds up
    // (the LDS-resident rollout, nig_pg_lds.hpp, evaluates them where their inputs are at hand)
    __device__ static float reward_freq(float f) { return -100.0f * (f * f); }          // :162 (scalar ** 2: within 1 ulp of powf)
    __device__ static float reward_volt(const float (&v)[8])                             // :165-166
    {
        float d2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = fabsf(v[i] - 1.0f);
            d2[i] = d * d;
        }
        return -50.0f * sum8(d2);
    }
    __device__ static double reward_econ(const float (&g)[8])                            // :169-170 int64 * float32 -> float64
    {
        // np.sum's pairwise tree over the eight float64 products ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)).  A product
        // cost x generation is EXACT in float64 (a 6-bit integer times a 24-bit significand), so p_even + p_odd is one fused
        // multiply-add with the same single rounding: four v_fma_f64 for four multiplies + four adds (round 5; same bits)
        double pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = __builtin_fma(gen_cost(2 * i + 1), (double)g[2 * i + 1], gen_cost(2 * i) * (double)g[2 * i]);
        return ddiv_y(-((pr[0] + pr[1]) + (pr[2] + pr[3])), 1000.0, 1.0 / 1000.0);
    }
    __device__ static float reward_act(const float (&a)[A])                              // :173
    {
        float a2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a2[i] = a[i] * a[i];
        return -5.0f * sum8(a2);
    }
    __device__ static double reward_total(float fr, float vr, double er, float ap)       // :175
    {
        return ((double)(fr + vr) + er) + (double)ap;
    }
    __device__ static double reward(const float (&n)[S], const float (&a)[A])
    {
        float v[8], g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = n[1 + i]; g[i] = n[9 + i]; }
        return reward_total(reward_freq(n[0]), reward_volt(v), reward_econ(g), reward_act(a));
    }

    // _is_done :179-192
    __device__ static bool done_fv(float f, const float (&v)[8])
    {
        // any(V < 0.9) or any(V > 1.1) as "smallest < 0.9 or largest > 1.1" over v_min3_f32 / v_max3_f32 trees (IEEE minNum /
        // maxNum: a NaN element is passed over, as the element-wise compares pass it over; all NaN -> NaN -> both false)
        float mn = v[0], mx = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) { mn = __builtin_fminf(mn, v[i]); mx = __builtin_fmaxf(mx, v[i]); }
        return (fabsf(f) > 1.0f) | (mn < 0.9f) | (mx > 1.1f);
    }
    __device__ static bool done(const float (&n)[S])
    {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = n[1 + i];
        return done_fv(n[0], v);
    }
};

// =================================================================================
// RobotAssembly-v0  (environments/robot_assembly.py), S=24 A=7, fp64 internals
// =================================================================================
struct RobotAssembly {
    // ~2.4 % of lanes finish per step: ~80 % of the waves (and every block) see a reset in every step, and a reset is
    // heavy (7 fp64 sincos).  Round 1 compacted the finishing lanes across the block behind two block barriers per
    // step and let one wave run them; now each wave renews its own lanes cooperatively (work item = (lane, joint):
    // uniform draw -> joint angle -> sincos -> link * cos / sin terms into a wave-private LDS image, the owner sums
    // them in the reference's order): no block barrier, one short pass instead of a 7-sincos path.
    static constexpr bool COOP_RESET = true;
    static constexpr int RESET_ITEMS = 8;          // 7 joints (+ 1 idle item) per reset
    static constexpr int RESET_ROWS = 29;          // image per wave: 11 rows of 64 doubles (terms) + 7 rows of 64 floats (angles)
    using fast_noise_t = double;
    static constexpr int ID = 2, S = 24, A = 7, KS = 0, KR = 7, MAX_STEPS = 1000;
    static constexpr bool COMPACT_RESET = false;
    static constexpr bool SHARED_STEP_BLOCK = false;
    static constexpr bool CUSTOM_STEP = false, RET_F32 = false;
    static constexpr int STEP_BLOCK = 256, STEP_WAVES = 5;
    static constexpr int ROLLOUT_WAVES = 3;       // same, for the fused rollout kernels (two waves at 209 registers: reward + flags 1.47 -> 1.56 ms, full outputs equal)
    // batches that leave one wave per SIMD: producer / integrator / recorder wave per 64 lanes (nig_split.hpp).  No step
    // noise, so the producer only loads and clips actions; the recorder takes the fp64 reward, the tally and the 24-row
    // stores off the wave that runs the seven fp64 sincos.  Not in rounds: larger batches fill the SIMDs with lanes.
#ifndef NIG_RA_SPLIT_ROUNDS
#define NIG_RA_SPLIT_ROUNDS false
#endif
    static constexpr bool SPLIT_ROLLOUT = true, SPLIT_ROUNDS = NIG_RA_SPLIT_ROUNDS;
    using reward_t = double;
    __device__ static constexpr float act_low(int) { return -1.0f; }
    __device__ static constexpr float act_high(int) { return 1.0f; }

    __device__ static constexpr double penalty(int k) { return k == 0 ? -100.0 : (k == 1 ? -200.0 : -50.0); }  // :56-75
    static constexpr uint32_t CRIT_MASK = 0x3u;
    static constexpr double PI = 3.141592653589793;

    __device__ static constexpr double link(int i)    // :85
    {
        return i == 0 ? 0.3 : i == 1 ? 0.3 : i == 2 ? 0.25 : i == 3 ? 0.25 : i == 4 ? 0.15 : i == 5 ? 0.1 : 0.05;
    }

    // _forward_kinematics :94-111: even joints -> x,z ; odd joints -> y ; sequential fp64
    __device__ static void fk(const double (&q)[7], double &x, double &y, double &z)
    {
        x = 0.0; y = 0.0; z = 0.0;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            double sn, cs;
            det_sincos(q[i], sn, cs);
            if (i % 2 == 0) { x += link(i) * cs; z += link(i) * sn; }
            else { y += link(i) * sn; }
        }
    }

    // _get_initial_state :113-137; the 7 draws ARE the joint angles (uniform(-pi/2, pi/2))
    __device__ static void init(const double (&n)[KR], float (&s)[S])
    {
        double x, y, z;
        fk(n, x, y, z);
        s[0] = (float)x; s[1] = (float)y; s[2] = (float)z;
        s[3] = 0.0f; s[4] = 0.0f; s[5] = 0.0f; s[6] = 1.0f;
#pragma unroll
        for (int i = 0; i < 7; ++i) s[7 + i] = (float)n[i];
#pragma unroll
        for (int i = 14; i < 24; ++i) s[i] = 0.0f;
    }
    __device__ static void draw_init(const RngKey &k, double (&n)[KR])
    {
        double u[KR];
        gen_uniforms<KR>(k, STREAM_RESET, u);
        const double lo = -PI * 0.5, hi = PI * 0.5;                      // :119-120
#pragma unroll
        for (int i = 0; i < KR; ++i) n[i] = lo + (hi - lo) * u[i];
    }
    __device__ static void draw_step(const RngKey &, double (&)[1]) {}

    // One work item of a cooperative reset: joint j of the lane with key `k` (same values, operation by operation, as
    // draw_init + init: u_j = word j & 3 of generator block j >> 2, q_j = lo + (hi - lo) u_j, fk's link * cos / sin).
    // Image rows of 64 doubles: 0-3 = L cos q of joints 0,2,4,6 (x); 4-7 = L sin q of joints 0,2,4,6 (z);
    // 8-10 = L sin q of joints 1,3,5 (y); then rows of 64 floats: the 7 joint angles as stored in the state.
    __device__ static void reset_item(const RngKey &k, uint32_t j, float *img, unsigned owner)
    {
        if (j >= 7u) return;
        const u32x4 x = k.block(STREAM_RESET + (j >> 2));
        const uint32_t sel = j & 3u;
        const uint32_t w = sel == 0u ? x.x : (sel == 1u ? x.y : (sel == 2u ? x.z : x.w));
        const double lo = -PI * 0.5, hi = PI * 0.5;                      // :119-120
        const double q = lo + (hi - lo) * u01(w);
        double sn, cs;
        det_sincos(q, sn, cs);
        const double L = j == 0u ? 0.3 : j == 1u ? 0.3 : j == 2u ? 0.25 : j == 3u ? 0.25 : j == 4u ? 0.15 : j == 5u ? 0.1 : 0.05;   // :85
        double *d = reinterpret_cast<double *>(img);
        const uint32_t h = j >> 1;
        if (j & 1u) {
            d[(8u + h) * 64u + owner] = L * sn;
        } else {
            d[h * 64u + owner] = L * cs;
            d[(4u + h) * 64u + owner] = L * sn;
        }
        img[(22u + j) * 64u + owner] = (float)q;
    }
    __device__ static void reset_readback(const float *img, unsigned lane, float (&n)[S])
    {
        const double *d = reinterpret_cast<const double *>(img);
        double x = 0.0, y = 0.0, z = 0.0;                                // fk :94-111: sequential sums in joint order
#pragma unroll
        for (int h = 0; h < 4; ++h) { x += d[h * 64 + lane]; z += d[(4 + h) * 64 + lane]; }
#pragma unroll
        for (int h = 0; h < 3; ++h) y += d[(8 + h) * 64 + lane];
        n[0] = (float)x; n[1] = (float)y; n[2] = (float)z;
        n[3] = 0.0f; n[4] = 0.0f; n[5] = 0.0f; n[6] = 1.0f;
#pragma unroll
        for (int i = 0; i < 7; ++i) n[7 + i] = img[(22 + i) * 64 + lane];
#pragma unroll
        for (int i = 14; i < 24; ++i) n[i] = 0.0f;
    }

    // module-level check functions :10-32
    __device__ static uint32_t violated(const float (&s)[S], const float (&)[A])
    {
        bool f_ok = true, c_ok = true, v_ok = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) f_ok = f_ok & (fabsf(s[18 + i]) < 50.0f);           // :15-16
        // :24-26 compares a float32 state value with Python floats, i.e. in fp64: for a float x, (double)x <= c holds
        // exactly when x <= the largest float not above c (and >= likewise), so the six compares run in float32 --
        // an fp64 compare costs a conversion and two wait states before its result can be used
        c_ok = (s[0] >= f32_not_below(-0.5)) & (s[0] <= f32_not_above(0.5)) & (s[1] >= f32_not_below(-0.5)) &
               (s[1] <= f32_not_above(0.5)) & (s[2] >= f32_not_below(0.0)) & (s[2] <= f32_not_above(0.8));
#pragma unroll
        for (int i = 0; i < 7; ++i) v_ok = v_ok & (fabsf(s[7 + i]) < 2.0f);             // :30-32
        return (f_ok ? 0u : 1u) | (c_ok ? 0u : 2u) | (v_ok ? 0u : 4u);
    }

    // _dynamics :139-188
    __device__ static void dynamics(const float (&s)[S], const float (&a)[A], const double (&)[1],
                                    float dt32, double dt, float (&o)[S])
    {
        float qf[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) qf[i] = s[7 + i] + a[i] * dt32;              // :148 float32
        double q[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) q[i] = (double)qf[i];
        // fp64 clip to [-pi, pi] :149-153.  Fourteen compares and 28 selects that change nothing unless a joint has
        // left the interval (a random walk of +-0.1 rad steps from +-pi/2 inside an episode of a few dozen steps:
        // almost never): the wave looks at max |q| first -- 0x40490FDA is the largest float not above the double pi,
        // so a float within it is inside the fp64 interval, and a NaN (which the clip leaves alone, as np.clip does)
        // does not raise the maximum -- and skips the clip when no lane needs it.  Same values either way.
        float m;
        asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(qf[0]), "v"(qf[1]), "v"(qf[2]));
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(qf[3]), "v"(qf[4]));
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(qf[5]), "v"(qf[6]));
        if (__builtin_amdgcn_ballot_w64(!(m <= __uint_as_float(0x40490FDAu))) != 0ull) clip_joints(q);
        dynamics_from_joints(s, q, dt, o);
    }
    __device__ static void clip_joints(double (&q)[7])
    {
#pragma unroll
        for (int i = 0; i < 7; ++i) {                                    // fp64 clip :149-153
            double d = q[i];
            d = (d < -PI) ? -PI : d;
            d = (d > PI) ? PI : d;
            q[i] = d;
        }
    }
    // float64 actions (robot_assembly.py:266-290 hands them over; pinned by ra_g5.npz, ra_g6.npz): :148 is float64
    static constexpr bool HAS_ACT64 = true;
    __device__ static uint32_t violated(const float (&s)[S], const double (&)[A])
    {
        const float none[A] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        return violated(s, none);
    }
    __device__ static void dynamics(const float (&s)[S], const double (&a)[A], const double (&)[1],
                                    float /*dt32*/, double dt, float (&o)[S])
    {
        double q[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) q[i] = (double)s[7 + i] + a[i] * dt;
        clip_joints(q);
        dynamics_from_joints(s, q, dt, o);
    }
    __device__ static double reward(const float (&n)[S], const double (&a)[A])
    {
        double ap = 0.0;
#pragma unroll
        for (int i = 0; i < 7; ++i) ap = ap + a[i] * a[i];               // :211 float64, sequential
        return reward_with_penalty(n, -0.1 * ap);
    }
    __device__ static void dynamics_from_joints(const float (&s)[S], double (&q)[7], double dt, float (&o)[S])   // q: clipped
    {
        double x, y, z;
        fk(q, x, y, z);                                                  // :156
        // :159-160 three divisions by dt, the same for every lane: by its reciprocal with the correctly rounded
        // correction of ddiv_y (nig_detmath.hpp) while dt is an ordinary step size, the IEEE sequence otherwise
        double vx, vy, vz;
        if (dt >= 0x1p-100 && dt <= 0x1p100) {
            const double rdt = 1.0 / dt;
            vx = ddiv_y(x - (double)s[0], dt, rdt); vy = ddiv_y(y - (double)s[1], dt, rdt); vz = ddiv_y(z - (double)s[2], dt, rdt);
        } else {
            vx = (x - (double)s[0]) / dt; vy = (y - (double)s[1]) / dt; vz = (z - (double)s[2]) / dt;
        }
        const double dx = x - 0.3, dy = y - 0.0, dz = z - 0.4;           // target :90
        const double dist = sqrt(dx * dx + dy * dy + dz * dz);           // :163
        double fz = 0.0;
        if (dist < 0.01) {                                               // :164-169
            const double nf = pymax(0.0, 0.01 - dist) * 1000.0;
            fz = (nf == 0.0) ? 0.0 : (0.0 - nf);
        }
        const double ae = sqrt(dx * dx + dy * dy);                       // :172
        const double align = pymax(0.0, 1.0 - ddiv_y(ae, 0.005, 1.0 / 0.005));   // :173
        const double ins = pymax(0.0, 0.4 - z);                          // :175
        const double depth = pymin(1.0, ddiv_y(ins, 0.05, 1.0 / 0.05));  // :176
        o[0] = (float)x; o[1] = (float)y; o[2] = (float)z;
        o[3] = 0.0f; o[4] = 0.0f; o[5] = 0.0f; o[6] = 1.0f;              // :182
#pragma unroll
        for (int i = 0; i < 7; ++i) o[7 + i] = (float)q[i];              // :183
        o[14] = (float)vx; o[15] = (float)vy; o[16] = (float)vz; o[17] = 0.0f;   // :184
        o[18] = 0.0f; o[19] = 0.0f; o[20] = (float)fz;                   // :185
        o[21] = (float)align; o[22] = (float)depth; o[23] = (float)(align * depth);   // :178,186
    }

    // _compute_reward :190-222
    __device__ static double reward(const float (&n)[S], const float (&a)[A])
    {
        float ap = 0.0f;
#pragma unroll
        for (int i = 0; i < 7; ++i) ap = ap + a[i] * a[i];               // :211
        return reward_with_penalty(n, (double)(-0.1f * ap));
    }
    __device__ static double reward_with_penalty(const float (&n)[S], const double action_penalty)
    {
        const float cr = 100.0f * n[23];                                 // :197
        const double dx = (double)n[0] - 0.3, dy = (double)n[1] - 0.0, dz = (double)n[2] - 0.4;
        const double dr = -10.0 * sqrt(dx * dx + dy * dy + dz * dz);     // :200-201
        const float fm = sqrtf((n[18] * n[18] + n[19] * n[19]) + n[20] * n[20]);   // :204 float32 norm
        float vp = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) vp = vp + n[14 + i] * n[14 + i];     // :214-215
        double tot = (double)cr + dr;                                    // :217-220, left to right
        tot = tot + ((fm > 30.0f) ? (double)(-50.0f * (fm - 30.0f)) : 0.0);   // :205-208
        tot = tot + action_penalty;
        tot = tot + (double)(-0.5f * vp);
        return tot;
    }

    // _is_done :224-244
    __device__ static bool done(const float (&n)[S])
    {
        bool d = n[23] > 0.95f;                                          // :231
#pragma unroll
        for (int i = 0; i < 3; ++i) d = d | (fabsf(n[18 + i]) > 80.0f);  // :235
        const bool inside = (n[0] >= f32_not_below(-0.6)) & (n[0] <= f32_not_above(0.6)) & (n[1] >= f32_not_below(-0.6)) &
                            (n[1] <= f32_not_above(0.6)) & (n[2] >= f32_not_below(-0.1)) & (n[2] <= f32_not_above(0.9));   // :239-242 (fp64 bounds, see violated)
        return d || !inside;
    }
};


// =================================================================================
// HVACControl-v0 / WaterTreatment-v0 / SteelAnnealing-v0 / SupplyChain-v0 -- BUILD-SPECIFIED.
//
// The reference's README lists these four (name, dims, constraint names: README.md:28-32) and ships
// no implementation, so there is no reference output and parity is undefined; they complete the
// "all 7 envs" mixed batch of BASELINE.json.  One plant family (first-order relaxation + linear
// coupling + linear actuator gains, velocity-form actuators, box constraints), four tables:
// spec_plants.py is the source of the numbers and of the model's description; the table below is
// generated from it (nig_spec_plants.inc; the oracle compiles the same data with its own code).
// They go through the base step template like the three real envs (clip, constraint check on the
// pre-state, penalties, critical shutdown: base.py:157-213).
// =================================================================================
#include "nig_spec_plants.inc"
constexpr int SPEC_MAX_NP = 15, SPEC_MAX_A = 10;
struct spec_plant_t {
    int np, na;
    float y0[SPEC_MAX_NP], sd0[SPEC_MAX_NP], k[SPEC_MAX_NP], amb[SPEC_MAX_NP], cpl[SPEC_MAX_NP];
    int cidx[SPEC_MAX_NP];
    float ymin[SPEC_MAX_NP], ymax[SPEC_MAX_NP], sp[SPEC_MAX_NP], w[SPEC_MAX_NP];
    float G[SPEC_MAX_NP][SPEC_MAX_A];
    float rate[SPEC_MAX_A], ecost[SPEC_MAX_A];
    float nsd[2];
    float we, wu, bonus;
    int cfirst[3], ccount[3];
    float clo[3], chi[3], pen[3];
    int crit[3];
    int d_idx;
    float dlo, dhi;
};
// constexpr: every table access below has a compile-time index once the loops are unrolled, so the
// numbers end up as instruction literals (no table in device memory)
constexpr spec_plant_t NIG_SPEC_PLANTS[4] = {NIG_SPEC_PLANT_ROWS};
constexpr int NIG_SPEC_NP[4] = {NIG_SPEC_NP_LIST}, NIG_SPEC_NA[4] = {NIG_SPEC_NA_LIST};
constexpr int NIG_SPEC_MAXSTEPS[4] = {NIG_SPEC_MAXSTEPS_LIST};

template <int K>
struct SpecPlant {
    static constexpr bool COOP_RESET = false;
    static constexpr bool HAS_ACT64 = false;     // build-specified plants are float32 by design
    static constexpr int RESET_ROWS = 1;
    using fast_noise_t = float;      // step noise is a float32 value (nsd * z): carried as float, widened only where draws are injected
    static constexpr int NP = NIG_SPEC_NP[K], A = NIG_SPEC_NA[K], S = NP + A + 3, ID = 5 + K;
    static constexpr int KS = 2, KR = NP, MAX_STEPS = NIG_SPEC_MAXSTEPS[K];
    static constexpr int ROW_E = NP + A, ROW_ECUM = NP + A + 1, ROW_T = NP + A + 2;
    static constexpr bool COMPACT_RESET = false, SHARED_STEP_BLOCK = true, CUSTOM_STEP = false, RET_F32 = true;
    static constexpr int STEP_BLOCK = 256, STEP_WAVES = 4, ROLLOUT_WAVES = (S > 24) ? 2 : 3;
    using reward_t = float;
    __device__ static constexpr float act_low(int) { return -1.0f; }
    __device__ static constexpr float act_high(int) { return 1.0f; }
    __device__ static constexpr float penalty(int k) { return NIG_SPEC_PLANTS[K].pen[k]; }
    static constexpr uint32_t CRIT_MASK = (NIG_SPEC_PLANTS[K].crit[0] ? 1u : 0u) | (NIG_SPEC_PLANTS[K].crit[1] ? 2u : 0u) |
                                          (NIG_SPEC_PLANTS[K].crit[2] ? 4u : 0u);

    __device__ static void init(const double (&n)[KR], float (&s)[S])
    {
        constexpr spec_plant_t P = NIG_SPEC_PLANTS[K];
#pragma unroll
        for (int i = 0; i < NP; ++i) s[i] = (float)((double)P.y0[i] + n[i]);
#pragma unroll
        for (int j = 0; j < A; ++j) s[NP + j] = 0.5f;
        s[ROW_E] = 0.0f; s[ROW_ECUM] = 0.0f; s[ROW_T] = 0.0f;
    }
    __device__ static void draw_init(const RngKey &k, double (&n)[KR])
    {
        constexpr spec_plant_t P = NIG_SPEC_PLANTS[K];
        float z[KR];
        gen_normals<KR>(k, STREAM_RESET, z);
#pragma unroll
        for (int i = 0; i < NP; ++i) n[i] = 0.0 + (double)P.sd0[i] * (double)z[i];
    }
    // Two draws per step: launch counters 2k-1 and 2k share ONE Philox block (counter word k, words 0-1 for the odd
    // counter, 2-3 for the even one) -- ChemicalReactor's rule, plant model "v2" (0.4.0) on: half of every block used to be thrown
    // away, and a block is 12 of the step's ~45 slow (64-bit multiply) issue slots.
    __device__ static u32x4 step_block(const RngKey &k)
    {
        RngKey kk = k;
        kk.t = (k.t + 1u) >> 1;
        return kk.block(STREAM_STEP);
    }
    __device__ static void step_noise_fetch(uint32_t w0, uint32_t w1, const float4 *tab, ProbitFetch (&f)[KS])
    {
        f[0] = probit_fetch(w0, tab); f[1] = probit_fetch(w1, tab);
    }
    template <class NZ>
    __device__ static void step_noise_eval(const ProbitFetch (&f)[KS], NZ (&n)[KS])
    {
        constexpr spec_plant_t P = NIG_SPEC_PLANTS[K];
        n[0] = (NZ)(P.nsd[0] * probit_eval(f[0]));
        n[1] = (NZ)(P.nsd[1] * probit_eval(f[1]));
    }
    template <class NZ>
    __device__ static void step_noise(uint32_t w0, uint32_t w1, const float4 *tab, NZ (&n)[KS])
    {
        ProbitFetch f[KS];
        step_noise_fetch(w0, w1, tab, f);
        step_noise_eval(f, n);
    }
    template <class NZ>
    __device__ static void draw_step(const RngKey &k, NZ (&n)[KS])
    {
        const u32x4 x = step_block(k);
        const bool second = (k.t & 1u) == 0;
        step_noise(second ? x.z : x.x, second ? x.w : x.y, k.tab, n);
    }

    // The plant model's clip(v, lo, hi) (spec_plants.py): the larger of v and lo, then the smaller of that and hi; a NaN
    // becomes lo, a zero at a zero limit takes the limit's sign.  That is what ONE v_med3_f32 computes (a NaN input
    // makes it return the minimum of the others), where two compare + select pairs are four instructions: the four
    // plants clip 12 to 25 values per step.
    __device__ static float clamp(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

    // box constraint c: every row of [cfirst, cfirst + ccount) inside [clo, chi]; bit set = violated
    __device__ static uint32_t violated(const float (&s)[S], const float (&)[A])
    {
        constexpr spec_plant_t P = NIG_SPEC_PLANTS[K];
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) v |= box_ok(s, c) ? 0u : (1u << c);
        return v;
    }
    // every row of constraint c's run inside [clo, chi].  The run's smallest and largest value decide it: minimum /
    // maximum trees of gfx950's three-input v_minimum3_f32 / v_maximum3_f32, which hand a NaN on -- so a NaN row fails
    // both compares, exactly like the row-by-row "lo <= s && s <= hi" chain (two compares and two scalar ands per row;
    // SupplyChain checks 35 rows per step, two constraints over the same ten).
    __device__ static bool box_ok(const float (&s)[S], int c)
    {
        constexpr spec_plant_t P = NIG_SPEC_PLANTS[K];
        // (the trees are seeded with the run's first row: a constraint over an EMPTY run would test that row against zero
        // limits -- every plant has three non-empty constraints, spec_plants.py write_inc refuses anything else)
        static_assert(P.ccount[0] >= 1 && P.ccount[1] >= 1 && P.ccount[2] >= 1, "SpecPlant: every box constraint covers at least one row");
        float mn = s[P.cfirst[c]], mx = mn;
#pragma unroll
        for (int r = 0; r < S; ++r)
            if (r > P.cfirst[c] && r < P.cfirst[c] + P.ccount[c]) {
                mn = __builtin_elementwise_minimum(mn, s[r]);
                mx = __builtin_elementwise_maximum(mx, s[r]);
            }
        return (P.clo[c] <= mn) & (mx <= P.chi[c]);
    }

    template <class NZ>
    __device__ static void dynamics(const float (&s)[S], const float (&a)[A], const NZ (&nz)[KS],
                                    float dt32, double, float (&o)[S])
    {
        constexpr spec_plant_t P = NIG_SPEC_PLANTS[K];
        // (model arithmetic, spec_plants.py "v2": every multiply-add below is ONE fused operation with one rounding --
        // v_fma_f32 here, fmaf() in the CPU statement)
        float pn[A];
        float e = 0.0f;
#pragma unroll
        for (int j = 0; j < A; ++j) {              // velocity-form actuators, clipped to [0, 1]
            pn[j] = clamp(__builtin_fmaf(P.rate[j] * a[j], dt32, s[NP + j]), 0.0f, 1.0f);
            e = __builtin_fmaf(P.ecost[j], pn[j], e);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float dy = (-P.k[i]) * (s[i] - P.amb[i]);
#pragma unroll
            for (int j = 0; j < A; ++j)
                if (P.G[i][j] != 0.0f) dy = __builtin_fmaf(P.G[i][j], pn[j], dy);
            if (P.cpl[i] != 0.0f) dy = __builtin_fmaf(P.cpl[i], s[P.cidx[i]] - s[i], dy);
            if (i < KS) dy = dy + (float)nz[i];
            o[i] = clamp(__builtin_fmaf(dy, dt32, s[i]), P.ymin[i], P.ymax[i]);
        }
#pragma unroll
        for (int j = 0; j < A; ++j) o[NP + j] = pn[j];
        o[ROW_E] = e;
        o[ROW_ECUM] = __builtin_fmaf(e, dt32, s[ROW_ECUM]);
        o[ROW_T] = s[ROW_T] + dt32;
    }

    __device__ static float reward(const float (&n)[S], const float (&a)[A])
    {
        constexpr spec_plant_t P = NIG_SPEC_PLANTS[K];
        float r = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (P.w[i] != 0.0f) r = __builtin_fmaf(-P.w[i], fabsf(n[i] - P.sp[i]), r);
        r = __builtin_fmaf(-P.we, n[ROW_E], r);
        float ap = 0.0f;
#pragma unroll
        for (int j = 0; j < A; ++j) ap = ap + fabsf(a[j]);
        r = __builtin_fmaf(-P.wu, ap, r);
        return box_ok(n, 0) ? (r + P.bonus) : r;   // bonus while constraint 0 holds on the new state
    }

    __device__ static bool done(const float (&n)[S])
    {
        constexpr spec_plant_t P = NIG_SPEC_PLANTS[K];
        return (n[P.d_idx] < P.dlo) || (n[P.d_idx] > P.dhi);
    }
};
using HVACControl = SpecPlant<0>;
using WaterTreatment = SpecPlant<1>;
using SteelAnnealing = SpecPlant<2>;
using SupplyChain = SpecPlant<3>;

// Result of one IndustrialEnv.step for one lane (filled by step_core or by an env's own step).
template <class Env, class R>
struct StepResult {
    R reward;
    uint32_t viol_bits;        // bit k: constraint k violated (up to 4)
    int nviol, ncrit;
    bool terminated, truncated, shutdown;
};

// =================================================================================
// AdvancedChemicalReactor-v0  (environments/advanced_chemical_reactor.py), S=20 A=6
//
// CANDIDATE ROW: upstream this class cannot be instantiated (abstract hooks
// missing, non-existent SafetyConstraint/SafetyMetrics kwargs, :90-105,445-450) and its step
// reads an attribute that is never set (self.episode_step, :351,364), so there is NO reference
// output to pin.  Restated from the source text as the float32 evaluation JAX would do with
// weak Python scalars (x64 off): Python-only sub-expressions are folded in double first, every
// op that touches an array value is float32, left to right.  exp / tanh / pow are the detmath
// polynomials (XLA's own expansions are not correctly rounded either).  The class overrides
// step(): no action clip, no base constraint loop, no -1000 critical shutdown.
// episode_step is taken as 0 at reset (the evident intent); dt is the BASE default 0.1, because
// IndustrialEnv.__init__ (base.py:44) overwrites the 1.0 assigned at :65 before it.
// =================================================================================
struct AdvancedChemicalReactor {
    static constexpr bool COOP_RESET = false;
    static constexpr bool HAS_ACT64 = false;     // JAX with x64 off: a float64 action becomes float32 on entry
    static constexpr int RESET_ROWS = 1;
    using fast_noise_t = double;
    static constexpr int ID = 3, S = 20, A = 6, KS = 0, KR = 0, MAX_STEPS = 1000;
    static constexpr bool COMPACT_RESET = false, CUSTOM_STEP = true, RET_F32 = false;
    static constexpr bool SHARED_STEP_BLOCK = false;
    static constexpr int STEP_BLOCK = 256, STEP_WAVES = 5;
    static constexpr int ROLLOUT_WAVES = 3;       // same, for the fused rollout kernels
    using reward_t = float;    // float(total_reward) of a float32 scalar, :404
    static constexpr uint32_t CRIT_MASK = 0u;
    __device__ static constexpr float penalty(int) { return 0.0f; }
    __device__ static constexpr float act_low(int j) { return j == 3 ? 273.15f : 0.0f; }                  // :148-155
    __device__ static constexpr float act_high(int j) { return j < 2 ? 0.01f : j == 2 ? 3000.0f : j == 3 ? 473.15f : j == 4 ? 100.0f : 1.0f; }

    __device__ static void init(const double (&)[1], float (&s)[S])       // reset, :158-193
    {
        s[0] = 323.15f; s[1] = 313.15f; s[2] = 2e5f; s[3] = 2.0f; s[4] = 1.5f; s[5] = 0.1f; s[6] = 0.1f;
        s[7] = 0.001f; s[8] = 0.001f; s[9] = 0.005f; s[10] = 300.0f; s[11] = 0.8f;
        s[12] = 323.15f; s[13] = 323.15f; s[14] = 323.15f; s[15] = 323.15f;
        s[16] = 1000.0f; s[17] = 0.05f; s[18] = 50.0f; s[19] = 60.0f;
    }
    __device__ static void draw_init(const RngKey &, double (&)[1]) {}
    __device__ static void draw_step(const RngKey &, double (&)[1]) {}

    // step :195-366, _compute_reward :368-404, _check_termination :406-420, get_safety_metrics :422-450
    __device__ static void custom_step(const float (&s)[S], const float (&a)[A], int step_pre, int max_steps,
                                       float dt, float (&o)[S], StepResult<AdvancedChemicalReactor, float> &out)
    {
        const float T = s[0], Tj = s[1], cA = s[3], cB = s[4], cC = s[5], cD = s[6];
        const float Ff = s[7], Fp = s[8], Fc = s[9], hc = s[10], mix = s[11];
        const bool estop = a[5] > 0.5f;                                  // :219
        const float feed_a = estop ? 0.0f : a[0];
        const float cool_a = estop ? 0.01f : a[1];                       // self.flow_range[1]
        const float rpm = estop ? 0.0f : a[2];
        const float nFf = Ff + 0.1f * (feed_a - Ff);                     // :226
        const float nFc = Fc + 0.2f * (cool_a - Fc);                     // :227
        const float k = 1e8f * det_expf(-83140.0f / (8.314f * T));       // :230
        const float rr = ((k * cA) * cB) * mix;                          // :231
        const float dA = ((nFf * 5.0f - Fp * cA) / 1.0f) - rr;           // :234
        const float dB = ((nFf * 3.0f - Fp * cB) / 1.0f) - rr;           // :235
        const float dC = (((-Fp) * cC) / 1.0f) + rr;                     // :236
        const float dD = (((-Fp) * cD) / 1.0f) + rr;                     // :237
        const float Qgen = (50000.0f * rr) * 1.0f;                       // :240
        const float Qj = (hc * 4.835975862049409f) * (T - Tj);           // :243 jacket_area :78
        float Qw = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) Qw = Qw + 6044.969827561761f * (T - s[12 + i]);   // :246-250
        const float Qf = ((nFf * 1000.0f) * 4180.0f) * (a[3] - T);       // :253
        const float dTr = fdiv_c(((Qgen - Qj) - Qw) + Qf, 4180000.0f);        // :256-259
        const float dTj = fdiv_c(Qj - ((nFc * 1000.0f) * 4180.0f) * (Tj - 293.15f), 418000.0f);   // :262-266
        float nTw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                    // :269-280
            const float w = s[12 + i];
            const float wd = fdiv_c(5000.0f * (T - w) - 10.0f * (w - 293.15f), 25000.0f);
            nTw[i] = w + dt * wd;
        }
        const float moles = (((cA + cB) + cC) + cD) * 1.0f;              // :284
        const float vp = 1000.0f * det_expf(20.0f - 5000.0f / T);        // :287
        float nP = ((((8.314f * T) * moles) / 1.0f) + vp) + 1e5f;        // :290-292
        if (nP > 2400000.0f) nP = nP - fdiv_c(a[4], 100.0f) * (nP - 2400000.0f);   // :295-297 (3e6*0.8)
        const float nmix = det_tanhf(fdiv_c(rpm, 1000.0f)) * 0.9f + 0.1f;       // :300
        const float Re = fdiv_c((rpm * 0.1f) * 1000.0f, 0.001f);              // :301
        const float Nu = 0.023f * det_powf(Re, 0.8f);                    // :302
        const float nhc = fdiv_c(Nu * 0.6f, 0.1f);                            // :303
        const float nFp = 0.001f * (1.0f + 0.5f * ((nP - 1e5f) / 1e5f)); // :306-307
        const float nA = fmaxf(0.0f, cA + dt * dA), nB = fmaxf(0.0f, cB + dt * dB);   // :310-313
        const float nC = fmaxf(0.0f, cC + dt * dC), nD = fmaxf(0.0f, cD + dt * dD);
        const float nT = T + dt * dTr, nTj = Tj + dt * dTj;              // :315-316
        const float tau = 1.0f / fmaxf(nFp, 1e-6f);                      // :319
        const float conv = (2.0f - nA) / 2.0f;                           // :322-323
        const float mT = fdiv_c(673.15f - nT, 673.15f) * 100.0f;            // :326
        const float mP = ((5e6f - nP) / 5e6f) * 100.0f;                  // :327
        o[0] = nT; o[1] = nTj; o[2] = nP; o[3] = nA; o[4] = nB; o[5] = nC; o[6] = nD;
        o[7] = nFf; o[8] = nFp; o[9] = nFc; o[10] = nhc; o[11] = nmix;
        o[12] = nTw[0]; o[13] = nTw[1]; o[14] = nTw[2]; o[15] = nTw[3];
        o[16] = tau; o[17] = conv; o[18] = mT; o[19] = mP;
        // reward on the new state
        const float pr = 100.0f * (fdiv_c(nC, 5.0f) + conv);                    // :379
        const float sr = (mT + mP) / 2.0f;                               // :382
        const float te = 1.0f - fdiv_c(fabsf(nT - 373.15f), 100.0f);            // :385
        const float pe = 1.0f - fabsf(nP - 3e5f) / 1e5f;                 // :386
        const float er = 50.0f * (te + pe);                              // :388
        const float cp = (-((((fabsf(a[0]) + fabsf(a[1])) + fabsf(a[2])) + fabsf(a[3])) + fabsf(a[4]))) * 10.0f;   // :391
        const float ep = estop ? -1000.0f : 0.0f;                        // :394
        out.reward = (((pr + sr) + er) + cp) + ep;                       // :396-402
        out.viol_bits = ((nT > 673.15f) ? 1u : 0u) | ((nP > 5e6f) ? 2u : 0u) |          // :433-443
                        ((mT < 10.0f) ? 4u : 0u) | ((mP < 10.0f) ? 8u : 0u);
        out.nviol = __popc(out.viol_bits); out.ncrit = 0;
        out.terminated = (nT > 673.15f) || (nP > 5e6f) || (nC > 8.0f);   // :412-420
        out.truncated = step_pre >= max_steps;                           // :351 (episode_step before its increment)
        out.shutdown = estop;                                            // info['emergency_shutdown'], :359
    }
};

// =================================================================================
// AdvancedPowerGrid-v0  (environments/advanced_power_grid.py), S=32 A=8
// CANDIDATE ROW: same status as above (not instantiable upstream; :101-121,
// 532-537; self.episode_step :331,345 never set).  State layout as actually built at :217-224.
// =================================================================================
struct AdvancedPowerGrid {
    static constexpr bool COOP_RESET = false;
    static constexpr bool HAS_ACT64 = false;     // JAX with x64 off: a float64 action becomes float32 on entry
    static constexpr int RESET_ROWS = 1;
    using fast_noise_t = double;
    static constexpr int ID = 4, S = 32, A = 8, KS = 0, KR = 0, MAX_STEPS = 500;
    static constexpr bool COMPACT_RESET = false, CUSTOM_STEP = true, RET_F32 = false;
    static constexpr bool SHARED_STEP_BLOCK = false;
    static constexpr int STEP_BLOCK = 256, STEP_WAVES = 5;
    static constexpr int ROLLOUT_WAVES = 2;       // same, for the fused rollout kernels
    using reward_t = float;
    static constexpr uint32_t CRIT_MASK = 0u;
    __device__ static constexpr float penalty(int) { return 0.0f; }
    __device__ static constexpr float H(int i) { return i == 0 ? 5.0f : i == 1 ? 4.0f : i == 2 ? 3.5f : 4.5f; }      // :79-85
    __device__ static constexpr float D(int i) { return i == 0 ? 1.0f : i == 1 ? 0.8f : i == 2 ? 0.9f : 1.1f; }
    __device__ static constexpr float Pmax(int i) { return i == 0 ? 50.0f : i == 1 ? 40.0f : i == 2 ? 35.0f : 45.0f; }
    __device__ static constexpr float Pmin(int i) { return i == 0 ? 10.0f : i == 1 ? 8.0f : i == 2 ? 7.0f : 9.0f; }
    __device__ static constexpr float ramp(int i) { return i == 0 ? 2.0f : i == 1 ? 1.8f : i == 2 ? 1.5f : 2.2f; }
    __device__ static constexpr float base_load(int i) { return i == 0 ? 25.0f : i == 1 ? 20.0f : i == 2 ? 30.0f : 18.0f; }   // :93-97
    __device__ static constexpr float alpha(int i) { return i == 0 ? 1.5f : i == 1 ? 1.2f : i == 2 ? 1.8f : 1.3f; }
    __device__ static constexpr float Kf(int i) { return i == 0 ? 1.0f : i == 1 ? 0.8f : i == 2 ? 1.2f : 0.9f; }
    __device__ static constexpr float act_low(int j) { return j < 4 ? Pmin(j) : j < 6 ? 0.95f : 0.0f; }               // :163-178
    __device__ static constexpr float act_high(int j) { return j < 4 ? Pmax(j) : j < 6 ? 1.05f : j == 6 ? 20.0f : 1.0f; }

    __device__ static void init(const double (&)[1], float (&s)[S])       // reset, :182-226
    {
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 1.0f;
        s[8] = 0.0f; s[9] = -0.1f; s[10] = 0.05f; s[11] = -0.05f; s[12] = 0.02f; s[13] = -0.02f; s[14] = 0.08f; s[15] = -0.08f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { s[16 + i] = 50.0f; s[24 + i] = base_load(i); }
        s[20] = 30.0f; s[21] = 25.0f; s[22] = 20.0f; s[23] = 28.0f;
        s[28] = 15.0f; s[29] = -12.0f; s[30] = 18.0f; s[31] = -14.0f;
    }
    __device__ static void draw_init(const RngKey &, double (&)[1]) {}
    __device__ static void draw_step(const RngKey &, double (&)[1]) {}

    // step :228-354, _solve_power_flow :356-407, _calculate_stability_margin :409-434,
    // _compute_reward :436-482, _check_termination :484-501, get_safety_metrics :503-537
    __device__ static void custom_step(const float (&s)[S], const float (&a)[A], int step_pre, int max_steps,
                                       float dt, float (&o)[S], StepResult<AdvancedPowerGrid, float> &out)
    {
        const bool emerg = a[7] > 0.5f;                                  // :246
        float sp[4], nf[4], nPg[4], nL[4];
        const float shed = emerg ? fminf(a[6] + 10.0f, 30.0f) : a[6];    // :249
        float fsum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sp[i] = emerg ? a[i] * 0.7f : a[i];                          // :248
            const float pm = fdiv_c(sp[i], 100.0f), pe = fdiv_c(s[20 + i], 100.0f);    // :261-262
            const float df = ((pm - pe) - D(i) * (s[16 + i] - 50.0f)) / (2.0f * H(i));   // :264-265
            nf[i] = s[16 + i] + dt * df;                                 // :272
            fsum = fsum + nf[i] * H(i);                                  // :276
        }
        const float fsys = fdiv_c(fsum, 17.0f);                                 // :275-276 (sum of inertias)
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                    // :279-289
            const float mr = ramp(i) * dt;
            float ch = sp[i] - s[20 + i];
            ch = fminf(fmaxf(ch, -mr), mr);
            nPg[i] = fminf(fmaxf(s[20 + i] + ch, Pmin(i)), Pmax(i));
        }
        const float fdev = fdiv_c(fsys - 50.0f, 50.0f);                       // :304
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                    // :293-308
            float bl = base_load(i);
            if (i == 0) bl = fmaxf(bl - shed, 0.0f);
            const float ve = det_powf(s[i] / 1.0f, alpha(i));
            const float fe = 1.0f + Kf(i) * fdev;
            nL[i] = (bl * ve) * fe;
        }
        float nV[8], nTh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {                                    // :368-389
            const float inj = (i < 4) ? fdiv_c(nPg[i], 100.0f) : fdiv_c(-nL[i - 4], 100.0f);
            float v = s[i] + 0.01f * inj;
            if (i == 0) v = a[4];
            if (i == 1) v = a[5];
            nV[i] = fminf(fmaxf(v, 0.8f), 1.2f);
            nTh[i] = s[8 + i] + 0.05f * inj;
        }
        float flow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)                                      // :395-405
            flow[i] = (fdiv_c(nV[i] * nV[i + 4], 0.1f) * det_sinf(nTh[i] - nTh[i + 4])) * 100.0f;
        float vmax = 0.0f, vmean = 0.0f, thmax = nTh[0], thmin = nTh[0], fmax_dev = 0.0f;
        bool vviol = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float dv = fabsf(nV[i] - 1.0f);
            vmax = fmaxf(vmax, dv); vmean = vmean + dv;
            vviol = vviol || (dv > 0.05f);
            thmax = fmaxf(thmax, nTh[i]); thmin = fminf(thmin, nTh[i]);
        }
        vmean = vmean / 8.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) fmax_dev = fmaxf(fmax_dev, fabsf(nf[i] - 50.0f));
        const float stab = fmaxf(fminf(fminf(1.0f - vmax, 1.0f - fdiv_c(thmax - thmin, 3.14159265358979323846f)),
                                       1.0f - fmax_dev / 0.5f), 0.0f);  // :417-434
#pragma unroll
        for (int i = 0; i < 8; ++i) { o[i] = nV[i]; o[8 + i] = nTh[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[16 + i] = nf[i]; o[20 + i] = nPg[i]; o[24 + i] = nL[i]; o[28 + i] = flow[i]; }
        // reward :436-482
        const float ferr = fabsf(fsys - 50.0f);
        const float r_f = 100.0f * det_expf(fdiv_c(-ferr, 0.1f));
        const float r_v = 50.0f * det_expf(fdiv_c(-vmean, 0.05f));
        const float tg = ((nPg[0] + nPg[1]) + nPg[2]) + nPg[3], tl = ((nL[0] + nL[1]) + nL[2]) + nL[3];
        const float r_b = 30.0f * det_expf(fdiv_c(-fabsf(tg - tl), 10.0f));
        const float r_e = -(0.01f * ((((nPg[0] * nPg[0]) + (nPg[1] * nPg[1])) + (nPg[2] * nPg[2])) + (nPg[3] * nPg[3])));
        const float r_c = (-(((((fabsf(a[0]) + fabsf(a[1])) + fabsf(a[2])) + fabsf(a[3])) + fabsf(a[4])) + fabsf(a[5]))) * 1.0f;
        out.reward = (((((r_f + r_v) + r_b) + r_e) + r_c) + (-a[6]) * 50.0f) + (-a[7]) * 200.0f;
        bool glim = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) glim = glim || (nPg[i] < Pmin(i)) || (nPg[i] > Pmax(i));   // :521-523
        out.viol_bits = ((ferr > 0.5f) ? 1u : 0u) | (vviol ? 2u : 0u) | (glim ? 4u : 0u);      // :511-523
        out.nviol = __popc(out.viol_bits); out.ncrit = 0;
        out.terminated = (ferr > 0.5f) || vviol || (stab < 0.1f);        // :492-501
        out.truncated = step_pre >= max_steps;                           // :331
        out.shutdown = emerg;                                            // info['emergency_active'], :341
    }
};

}  // namespace nig
