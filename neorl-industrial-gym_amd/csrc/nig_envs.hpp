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
#include "nig_detmath.hpp"

namespace nig {

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
    static constexpr int ID = 0, S = 12, A = 3, KS = 2, KR = 8, MAX_STEPS = 500;
    static constexpr bool COMPACT_RESET = false;   // ~0.3 % of lanes finish per step: divergent reset is cheaper than barriers
    using reward_t = float;   // reward stays np.float32 (0.0 + f32 under NEP 50), :240-269

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
    // fast mode: draws in reference call order, loc + scale*z in fp64 (np.random.normal)
    __device__ static void draw_init(const RngKey &k, double (&n)[KR])
    {
        float z[KR];
        gen_normals<KR>(k, STREAM_RESET, z);
        n[0] = 0.0 + 2.0 * (double)z[0];      n[1] = 0.0 + 10000.0 * (double)z[1];
        n[2] = 0.0 + 5.0 * (double)z[2];      n[3] = 0.0 + 3.0 * (double)z[3];
        n[4] = 0.0 + 0.1 * (double)z[4];      n[5] = 0.0 + 2.0 * (double)z[5];
        n[6] = 0.0 + 1.0 * (double)z[6];      n[7] = 0.0 + 5.0 * (double)z[7];
    }
    __device__ static void draw_step(const RngKey &k, double (&n)[KS])
    {
        float z[KS];
        gen_normals<KS>(k, STREAM_STEP, z);
        n[0] = 0.0 + 0.1 * (double)z[0];      // temp_noise_std / 10, :149
        n[1] = 0.0 + 500.0 * (double)z[1];    // pressure_noise_std / 10, :159
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
        const float kc = (0.1f * conc) * (cat / 100.0f);                // shared prefix of :137-139 and :175-177
        const float rh = kc * 10000.0f;
        const float ch = ((cool * 100.0f) * (T - hx)) * 0.1f;           // :141
        float dT = ((hp + rh) - ch) / 418000.0f;                        // :143-146 (4.18e3*1000*0.1 -> f32)
        dT = dT + (float)nz[0];                                         // :149
        const float nT = T + dT * 0.1f;                                 // :151
        float nP = P * (nT / T) + ((conc * 0.1f) * 1000.0f) * 0.1f;     // :155-158
        nP = nP + (float)nz[1];                                         // :159
        const float nrel = pymax(0.0f, pymin(100.0f, relief + (nP - 506625.0f) * 0.001f));  // :162-163
        if (nrel > 0.0f) nP = pymax(101325.0f, nP - (nrel * 0.01f) * 10000.0f);              // :166-168
        const float ncool = pymax(10.0f, pymin(100.0f, cool + cadj));   // :171
        const float nfeed = pymax(5.0f, pymin(50.0f, feed + fadj));     // :172
        const float rr = kc * det_expf((-(nT - 320.0f)) / 20.0f);       // :175-178
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

    // _compute_reward :228-270 on the NEXT state
    __device__ static float reward(const float (&n)[S], const float (&a)[A])
    {
        float r = n[4] * 100.0f;                                        // 0.0 + x == x, :242
        r = r - fabsf(n[0] - 320.0f) * 0.5f;                            // :245-246
        r = r - (fabsf(n[1] - 253312.5f) / 1000.0f) * 0.1f;             // :249-250
        r = r + (n[5] / 100.0f) * 10.0f;                                // :253
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
    static constexpr bool COMPACT_RESET = true;    // ~18 % of lanes finish per step (episodes of ~6 steps)
    using reward_t = double;  // float(total_reward), :177

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
    __device__ static void draw_init(const RngKey &k, double (&n)[KR])
    {
        float z[23];
        double u[8];
        gen_normals<23>(k, STREAM_RESET, z);
        gen_uniforms<8>(k, STREAM_RESET + 16u, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            n[i] = 0.0 + 0.01 * (double)z[i];
            n[8 + i] = 0.0 + 2.0 * (double)z[8 + i];
            n[16 + i] = -0.2 + (0.2 - -0.2) * u[i];                      // uniform: low + (high-low)*u
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) n[24 + i] = 0.0 + 10.0 * (double)z[16 + i];
    }
    __device__ static void draw_step(const RngKey &k, double (&n)[KS])
    {
        float z[KS];
        gen_normals<KS>(k, STREAM_STEP, z);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            n[i] = 0.0 + 0.005 * (double)z[i];                           // :136
            n[8 + i] = 0.0 + 1.0 * (double)z[8 + i];                     // :140
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) n[16 + i] = 0.0 + 2.0 * (double)z[16 + i];   // :144
    }

    // module-level check functions :10-30 (pre-state, clipped action)
    __device__ static uint32_t violated(const float (&s)[S], const float (&a)[A])
    {
        bool v_ok = true, g_ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v_ok = v_ok && (s[1 + i] >= 0.95f) && (s[1 + i] <= 1.05f);   // weak Python floats -> float32
            const float ng = s[9 + i] + a[i];                            // float32 add, :29
            g_ok = g_ok && (ng >= 0.0f) && ((double)ng <= 100.0);        // vs fp64 array np.ones(8)*100
        }
        uint32_t v = (fabsf(s[0]) < 0.5f) ? 0u : 1u;                     // :14
        v |= v_ok ? 0u : 2u;
        v |= g_ok ? 0u : 4u;
        return v;
    }

    // _dynamics :112-153
    __device__ static void dynamics(const float (&s)[S], const float (&a)[A], const double (&nz)[KS],
                                    float dt, double /*dt64*/, float (&o)[S])
    {
        float ngen[8], load[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float g = s[9 + i] + a[i];                                   // :124 np.clip(gen + a, 0, 100)
            g = (g < 0.0f) ? 0.0f : g;
            g = (g > 100.0f) ? 100.0f : g;
            ngen[i] = g;
            load[i] = s[17 + i];
        }
        const float imb = sum8(ngen) - sum8(load);                       // :127-129
        const float fd = ((-1.0f * s[0]) + imb) / 5.0f;                  // :132
        o[0] = s[0] + fd * dt;                                           // :133
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o[1 + i] = (float)((double)s[1 + i] + nz[i]);                // :136-137 fp64 add, one rounding
            o[9 + i] = ngen[i];
            double l = (double)s[17 + i] + nz[8 + i];                    // :140-141
            l = (l < 0.0) ? 0.0 : l;
            o[17 + i] = (float)l;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) o[25 + i] = (float)((double)s[25 + i] + nz[16 + i]);   // :144
    }

    // _compute_reward :155-177 (float32 terms, fp64 economic term, fp64 total)
    __device__ static double reward(const float (&n)[S], const float (&a)[A])
    {
        const float fr = -100.0f * (n[0] * n[0]);                        // :162 (scalar ** 2: within 1 ulp of powf)
        float d2[8], a2[8];
        double cg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = fabsf(n[1 + i] - 1.0f);                      // :165
            d2[i] = d * d;
            cg[i] = gen_cost(i) * (double)n[9 + i];                      // :169 int64 * float32 -> float64
            a2[i] = a[i] * a[i];
        }
        const float vr = -50.0f * sum8(d2);                              // :166
        const double er = (-sum8(cg)) / 1000.0;                          // :170
        const float ap = -5.0f * sum8(a2);                               // :173
        return ((double)(fr + vr) + er) + (double)ap;                    // :175
    }

    // _is_done :179-192
    __device__ static bool done(const float (&n)[S])
    {
        bool bad = fabsf(n[0]) > 1.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) bad = bad || (n[1 + i] < 0.9f) || (n[1 + i] > 1.1f);
        return bad;
    }
};

// =================================================================================
// RobotAssembly-v0  (environments/robot_assembly.py), S=24 A=7, fp64 internals
// =================================================================================
struct RobotAssembly {
    static constexpr int ID = 2, S = 24, A = 7, KS = 0, KR = 7, MAX_STEPS = 1000;
    static constexpr bool COMPACT_RESET = true;    // ~2.4 % of lanes per step, i.e. ~80 % of waves see a reset
    using reward_t = double;

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

    // module-level check functions :10-32
    __device__ static uint32_t violated(const float (&s)[S], const float (&)[A])
    {
        bool f_ok = true, c_ok = true, v_ok = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) f_ok = f_ok && (fabsf(s[18 + i]) < 50.0f);          // :15-16
        c_ok = ((double)s[0] >= -0.5) && ((double)s[0] <= 0.5) && ((double)s[1] >= -0.5) &&
               ((double)s[1] <= 0.5) && ((double)s[2] >= 0.0) && ((double)s[2] <= 0.8);   // :24-26 fp64 bounds
#pragma unroll
        for (int i = 0; i < 7; ++i) v_ok = v_ok && (fabsf(s[7 + i]) < 2.0f);            // :30-32
        return (f_ok ? 0u : 1u) | (c_ok ? 0u : 2u) | (v_ok ? 0u : 4u);
    }

    // _dynamics :139-188
    __device__ static void dynamics(const float (&s)[S], const float (&a)[A], const double (&)[1],
                                    float dt32, double dt, float (&o)[S])
    {
        double q[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            double d = (double)(s[7 + i] + a[i] * dt32);                 // :148 float32, then fp64 clip :149-153
            d = (d < -PI) ? -PI : d;
            d = (d > PI) ? PI : d;
            q[i] = d;
        }
        double x, y, z;
        fk(q, x, y, z);                                                  // :156
        const double vx = (x - (double)s[0]) / dt, vy = (y - (double)s[1]) / dt, vz = (z - (double)s[2]) / dt; // :159-160
        const double dx = x - 0.3, dy = y - 0.0, dz = z - 0.4;           // target :90
        const double dist = sqrt(dx * dx + dy * dy + dz * dz);           // :163
        double fz = 0.0;
        if (dist < 0.01) {                                               // :164-169
            const double nf = pymax(0.0, 0.01 - dist) * 1000.0;
            fz = (nf == 0.0) ? 0.0 : (0.0 - nf);
        }
        const double ae = sqrt(dx * dx + dy * dy);                       // :172
        const double align = pymax(0.0, 1.0 - ae / 0.005);               // :173
        const double ins = pymax(0.0, 0.4 - z);                          // :175
        const double depth = pymin(1.0, ins / 0.05);                     // :176
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
        const float cr = 100.0f * n[23];                                 // :197
        const double dx = (double)n[0] - 0.3, dy = (double)n[1] - 0.0, dz = (double)n[2] - 0.4;
        const double dr = -10.0 * sqrt(dx * dx + dy * dy + dz * dz);     // :200-201
        const float fm = sqrtf((n[18] * n[18] + n[19] * n[19]) + n[20] * n[20]);   // :204 float32 norm
        float ap = 0.0f, vp = 0.0f;
#pragma unroll
        for (int i = 0; i < 7; ++i) ap = ap + a[i] * a[i];               // :211
#pragma unroll
        for (int i = 0; i < 4; ++i) vp = vp + n[14 + i] * n[14 + i];     // :214-215
        double tot = (double)cr + dr;                                    // :217-220, left to right
        tot = tot + ((fm > 30.0f) ? (double)(-50.0f * (fm - 30.0f)) : 0.0);   // :205-208
        tot = tot + (double)(-0.1f * ap);
        tot = tot + (double)(-0.5f * vp);
        return tot;
    }

    // _is_done :224-244
    __device__ static bool done(const float (&n)[S])
    {
        bool d = n[23] > 0.95f;                                          // :231
#pragma unroll
        for (int i = 0; i < 3; ++i) d = d || (fabsf(n[18 + i]) > 80.0f); // :235
        const bool inside = ((double)n[0] >= -0.6) && ((double)n[0] <= 0.6) && ((double)n[1] >= -0.6) &&
                            ((double)n[1] <= 0.6) && ((double)n[2] >= -0.1) && ((double)n[2] <= 0.9);   // :239-242
        return d || !inside;
    }
};

}  // namespace nig
