// nig_detmath.hpp -- device-side deterministic math + counter-based RNG ("nig-philox-v2").
//
// Everything here is built from IEEE-754 + - * / only (file is compiled with
// -ffp-contract=off, correctly-rounded division/sqrt), so a host evaluation of the same
// operation sequence gives the same bits.  Specification: DESIGN.md "Deterministic math"
// and "Synthetic input generator".  None of this exists in the reference (it draws from
// NumPy's global MT19937, chemical_reactor.py:93-103,149,159; power_grid.py:98-108,
// 136-144; robot_assembly.py:118-122): it is the workload generator of fast mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nig {

// x / c for a compile-time constant c, correctly rounded like the true division NumPy does, in 4 VALU
// instead of the ~11 of the generic sequence: q0 = RN(x * RN(1/c)), r = x - c*q0 (exact in one fma),
// q = RN(q0 + r * RN(1/c)) (Markstein's correction).  For a given c the result depends only on the
// significand of x, so tests/test_host_logic.py proves each constant used here exhaustively over all
// 2^23 significands (tests/constdiv_check.c); v_div_fixup restores the IEEE special cases (signed
// zero, inf, NaN).  Valid while neither x*RN(1/c) nor the residual leaves the normal range
// (2^-100 < |x| < 2^100 is ample); every call site divides a physical quantity far inside that.
__device__ __forceinline__ float fdiv_c(float x, const float c)
{
    const float rc = 1.0f / c;                    // folded at compile time (c is a literal at every call site)
    const float q0 = x * rc;
    const float r = __builtin_fmaf(-c, q0, x);
    const float q = __builtin_fmaf(r, rc, q0);
    return __builtin_amdgcn_div_fixupf(q, c, x);
}

// a / b in fp64 for a divisor b that is a constant or shared by the wave, y = RN(1 / b) (an IEEE division folded at
// compile time or done once per launch), correctly rounded like NumPy's division in 6 instructions instead of the 13
// of the generic sequence (two v_div_scale, a quarter-rate v_rcp_f64, seven fma, v_div_fmas, v_div_fixup):
//   q0 = RN(a y);  q1 = RN(q0 + RN(a - b q0) y);  q2 = RN(q1 + (a - b q1) y);  v_div_fixup for the special cases.
// q0 is within 1.5 ulp of a / b, so q1 differs from a / b by less than 2^-100 relative before its rounding: a faithful
// quotient.  With a faithful q1 the residual a - b q1 is exact in one fma, and Markstein's theorem (Markstein 1990;
// Muller et al., Handbook of Floating-Point Arithmetic, "division with a correctly rounded reciprocal") gives
// q2 = RN(a / b) for every a when y is the correctly rounded reciprocal.  Needs a y, q b and the residual inside the
// normal range: callers pass quantities of order 1e-20 .. 1e10 and divisors between 2^-100 and 2^100 (a caller with a
// user-chosen divisor checks that and divides the long way otherwise).  tests/ddiv_check.c compares the sequence with
// IEEE division on 10^8 operands, including quotients placed next to rounding boundaries.
__device__ __forceinline__ double ddiv_y(double a, double b, double y)
{
    const double q0 = a * y;
    const double q1 = __builtin_fma(__builtin_fma(-q0, b, a), y, q0);
    const double q2 = __builtin_fma(__builtin_fma(-q1, b, a), y, q1);
    return __builtin_amdgcn_div_fixup(q2, b, a);
}

__device__ __forceinline__ float bits_f32(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f32_bits(float f) { return __float_as_uint(f); }

// e^x, float32, Cephes-style: k = floor(x*log2e + 1/2), two-constant Cody-Waite, degree-5
// polynomial in fused multiply-adds, exact two-step scaling.  <= 1 ulp-ish; stands in for np.exp on float32
// (chemical_reactor.py:177), which itself is only good to ~2 ulp.
__device__ __forceinline__ float det_expf(float x)
{
    float fk = floorf(__builtin_fmaf(x, 1.44269504088896341f, 0.5f));
    float r = __builtin_fmaf(-fk, 0.693359375f, x);
    r = __builtin_fmaf(-fk, -2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    p = __builtin_fmaf(p, z, r);
    p = p + 1.0f;
    // clamp only affects the out-of-range inputs that are overridden below
    fk = fminf(fmaxf(fk, -200.0f), 200.0f);
    const int k = (int)fk;
    const int k1 = k / 2, k2 = k - k1;
    float res = (p * bits_f32((uint32_t)(k1 + 127) << 23)) * bits_f32((uint32_t)(k2 + 127) << 23);
    res = (x < -103.0f) ? 0.0f : res;
    res = (x > 88.72283f) ? __builtin_inff() : res;
    res = (x != x) ? x : res;
    return res;
}

// e^x in double: Cody-Waite by ln2 (hi/lo), fdlibm's degree-5 kernel on r^2, exact two-step scaling; < 1 ulp.
// Stands in for np.exp on a float64 scalar (chemical_reactor.py:177 when the action is float64).  The oracle runs
// the same operation sequence.
__device__ __forceinline__ double det_exp(double x)
{
    const double k = floor(x * 1.44269504088896338700e+00 + 0.5);
    const double hi = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    const double lo = k * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    double c = 4.13813679705723846039e-08;
    c = __builtin_fma(c, t, -1.65339022054652515390e-06);
    c = __builtin_fma(c, t, 6.61375632143793436117e-05);
    c = __builtin_fma(c, t, -2.77777777770155933842e-03);
    c = __builtin_fma(c, t, 1.66666666666666019037e-01);
    c = r - t * c;
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const double kc = fmin(fmax(k, -2000.0), 2000.0);      // only reached by inputs that are overridden below
    const int ki = (int)kc;
    const int k1 = ki / 2, k2 = ki - k1;
    double res = (y * __longlong_as_double((long long)(k1 + 1023) << 52)) * __longlong_as_double((long long)(k2 + 1023) << 52);
    res = (x > 709.78) ? __builtin_inf() : res;
    res = (x < -745.13) ? 0.0 : res;
    res = (x != x) ? x : res;
    return res;
}

// ln(x) for a positive normal float (used by det_powf).
__device__ __forceinline__ float det_logf(float x)
{
    const uint32_t u = f32_bits(x);
    int e = (int)(u >> 23) - 126;
    float m = bits_f32((u & 0x007fffffu) | 0x3f000000u);   // [0.5, 1)
    const bool lo = m < 0.707106781186547524f;
    e = lo ? e - 1 : e;
    m = lo ? (m + m - 1.0f) : (m - 1.0f);
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = y * m + -1.1514610310e-1f;
    y = y * m + 1.1676998740e-1f;
    y = y * m + -1.2420140846e-1f;
    y = y * m + 1.4249322787e-1f;
    y = y * m + -1.6668057665e-1f;
    y = y * m + 2.0000714765e-1f;
    y = y * m + -2.4999993993e-1f;
    y = y * m + 3.3333331174e-1f;
    y = y * m * z;
    const float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    r = r + 0.693359375f * fe;
    return r;
}

// x^y for x >= 0 (x == 0 -> 0): exp(y * ln x) with the polynomials above.  Stands in for jnp's
// float32 power in the two Advanced envs (advanced_chemical_reactor.py:301 Re**0.8,
// advanced_power_grid.py:317 V**alpha); a few ulp, like XLA's own expansion.
__device__ __forceinline__ float det_powf(float x, float y)
{
    const float r = det_expf(y * det_logf(x));
    return (x > 0.0f) ? r : 0.0f;
}

// tanh(x) = (1 - e^-2x) / (1 + e^-2x) for x >= 0 (odd extension).  jnp.tanh, advanced_chemical_reactor.py:299
__device__ __forceinline__ float det_tanhf(float x)
{
    const float ax = fabsf(x);
    const float e = det_expf(-2.0f * ax);
    const float t = (1.0f - e) / (1.0f + e);
    return (x < 0.0f) ? -t : t;
}

// float32 sine, |x| up to a few hundred: Cody-Waite by pi/2 (3 constants), degree-7/8 kernels.
// jnp.sin of a bus-angle difference, advanced_power_grid.py:402
__device__ __forceinline__ float det_sinf(float x)
{
    const float fk = floorf(x * 0.636619772367581343f + 0.5f);
    float r = x - fk * 1.5703125f;
    r = r - fk * 4.837512969970703125e-4f;
    r = r - fk * 7.54978995489188216e-8f;
    const float z = r * r;
    float sp = -1.9515295891e-4f;
    sp = sp * z + 8.3321608736e-3f;
    sp = sp * z + -1.6666654611e-1f;
    sp = sp * z * r + r;
    float cp = 2.443315711809948e-5f;
    cp = cp * z + -1.388731625493765e-3f;
    cp = cp * z + 4.166664568298827e-2f;
    cp = cp * z * z;
    cp = cp + -0.5f * z;
    cp = cp + 1.0f;
    const int q = (int)fk & 3;
    float v = (q & 1) ? cp : sp;
    v = (q & 2) ? -v : v;
    return v;
}

// Thresholds for comparing a float32 value with a double constant in float32: for every float x (NaN and infinities
// included), (double)x <= c  <=>  x <= f32_not_above(c) and (double)x >= c  <=>  x >= f32_not_below(c), because the
// conversion is exact and no float lies strictly between the threshold and c.  (c finite, nonzero or exactly 0.)
constexpr float f32_not_above(double c)
{
    const float f = (float)c;
    if ((double)f <= c) return f;
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return __builtin_bit_cast(float, f > 0.0f ? b - 1u : b + 1u);        // one float down
}
constexpr float f32_not_below(double c)
{
    const float f = (float)c;
    if ((double)f >= c) return f;
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return __builtin_bit_cast(float, f > 0.0f ? b + 1u : b - 1u);        // one float up
}
static_assert(f32_not_above(0.8) == 0.79999995f && f32_not_above(0.5) == 0.5f && f32_not_below(-0.1) == -0.099999994f &&
              f32_not_below(-0.6) == -0.59999996f && f32_not_above(0.9) == 0.9f && (double)f32_not_above(0.6) <= 0.6, "float thresholds");

// double sin/cos for joint angles (|x| small multiples of pi): Cody-Waite with a 33-bit
// pi/2 head (k*head exact), fdlibm kernel polynomials.  Stands in for np.sin/np.cos on
// float64 scalars (robot_assembly.py:103-107).
__device__ __forceinline__ void det_sincos(double x, double &s, double &c)
{
    const double fk = floor(__builtin_fma(x, 0.63661977236758134308, 0.5));
    double r = __builtin_fma(-fk, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-fk, 6.07710050650619224932e-11, r);
    const double z = r * r;
    // the two Horner chains written step by step side by side: independent instructions alternate (a lone wave waits
    // out the latency of every dependent fp64 operation)
    double ps = 1.58969099521155010221e-10, pc = -1.13596475577881948265e-11;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08); pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04); pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01); pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double sn = __builtin_fma(r * z, ps, r);
    const double cs = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    // quadrant q = fk mod 4 from the low mantissa word of fk + 1.5 * 2^52 (two's complement of fk there for
    // |fk| < 2^51; the oracle states it the same way): ONE add where a double -> integer conversion is four
    // instructions.  Quadrant 1, 3: sin and cos trade places (bit select under an all-ones / all-zeros mask);
    // sin is negated in quadrants 2, 3 and cos in 1, 2: bit 1 of q / of q + 1 moved onto the sign bit, one shift and one
    // three-input bit operation each.  RobotAssembly runs seven of these per step and is bound by instruction issue.
    const uint32_t q = (uint32_t)(unsigned long long)__double_as_longlong(fk + 6755399441055744.0);
    const uint32_t swap = (uint32_t)__builtin_amdgcn_sbfe((int32_t)q, 0, 1);      // all ones in quadrants 1, 3
    const unsigned long long bs = (unsigned long long)__double_as_longlong(sn), bc = (unsigned long long)__double_as_longlong(cs);
    const uint32_t s_lo = (uint32_t)bs, s_hi = (uint32_t)(bs >> 32), c_lo = (uint32_t)bc, c_hi = (uint32_t)(bc >> 32);
    // bit select (a & b) | (~a & c): truth table 0xca
    const uint32_t ss_lo = __builtin_amdgcn_bitop3_b32(swap, c_lo, s_lo, 0xca), ss_hi = __builtin_amdgcn_bitop3_b32(swap, c_hi, s_hi, 0xca);
    const uint32_t cc_lo = __builtin_amdgcn_bitop3_b32(swap, s_lo, c_lo, 0xca), cc_hi = __builtin_amdgcn_bitop3_b32(swap, s_hi, c_hi, 0xca);
    const uint32_t ss_h = __builtin_amdgcn_bitop3_b32(q << 30, 0x80000000u, ss_hi, 0x6a);          // (a & b) ^ c
    const uint32_t cc_h = __builtin_amdgcn_bitop3_b32((q + 1u) << 30, 0x80000000u, cc_hi, 0x6a);
    s = __longlong_as_double((long long)(((unsigned long long)ss_h << 32) | ss_lo));
    c = __longlong_as_double((long long)(((unsigned long long)cc_h << 32) | cc_lo));
}

// ---------------------------------------------------------------------------------
// Philox4x32-7 (Salmon et al. 2011), counter = (env_lo, env_hi, t, stream+block),
// key = (seed_lo, seed_hi).  Seven rounds since "nig-philox-v2" (ten in v1): Philox4x32-7 is the variant the
// Random123 authors report as the fewest rounds that pass the whole of BigCrush (ten is their safety-margin default),
// Random123 ships known-answer vectors for it (tests/test_oracle_golden.py), and the generator is this build's
// workload spec, not the reference's (which draws from NumPy's MT19937).  The kernels are bound by vector-instruction
// issue and a round is four instructions: PowerGrid runs 6 blocks per step + 8 per reset.
// ---------------------------------------------------------------------------------
constexpr int PHILOX_ROUNDS = 7;
constexpr uint32_t STREAM_STEP = 0u;
constexpr uint32_t STREAM_RESET = 0x40000000u;
constexpr uint32_t STREAM_ACTION = 0x80000000u;
constexpr uint32_t STREAM_POLICY = 0xC0000000u;   // +0: mixture draw, +1..: normals, +8..: uniform noise, +16..: random action

struct u32x4 { uint32_t x, y, z, w; };

// 32x32 -> 64-bit product in ONE instruction.  hipcc lowers "(uint64_t)a * b" to a v_mul_hi_u32 +
// v_mul_lo_u32 pair (40 instructions per Philox call); v_mad_u64_u32 yields hi and lo together (20).
// (Measured on gfx950: the wide multiply issues like an ordinary VALU instruction -- replacing five of
// PowerGrid's six blocks per step with an add/xor/rotate stream of equal instruction count changed
// nothing beyond the count.)
// Written as "a * b + z" with z an SGPR pair the compiler cannot see through (it holds 0): instruction selection
// then folds multiply and add into v_mad_u64_u32 ITSELF.  Rounds 1-2 had the instruction as inline asm, and hipcc's
// hazard recognizer, blind to what an asm statement is, put an s_nop behind almost every one of them: 70 of
// PowerGrid's ~1 250 issue slots per step.  The asm below has no inputs and no side effects: identical copies are
// merged and hoisted, one v_mov_b64 per kernel.  The zero lives in a VECTOR register pair on purpose: the multiplier is a
// literal / SGPR, and gfx950 lets a VALU instruction read only one scalar source -- with the zero in an SGPR pair hipcc
// re-materialised it into a VGPR pair before almost every multiply (34 v_mov_b64 per PowerGrid step).
__device__ __forceinline__ uint64_t opaque_zero64()
{
    uint64_t z;
    asm("v_mov_b64 %0, 0" : "=v"(z));
    return z;
}
__device__ __forceinline__ void mulhilo32(uint32_t m, uint32_t x, uint32_t &hi, uint32_t &lo)
{
    const uint64_t r = (uint64_t)m * (uint64_t)x + opaque_zero64();
    lo = (uint32_t)r;
    hi = (uint32_t)(r >> 32);
}

__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; ++r) {
        uint32_t h0, l0, h1, l1;
        mulhilo32(0xD2511F53u, c0, h0, l0);
        mulhilo32(0xCD9E8D57u, c2, h1, l1);
        // three-input xor in ONE instruction (gfx950 v_bitop3_b32, truth table 0x96): the kernels are bound by
        // VALU issue and a Philox block was 20 wide multiplies + 40 xors
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(h1, c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(h0, c3, k1, 0x96);
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

struct RngKey {
    uint32_t env_lo, env_hi, t, seed_lo, seed_hi;
    const float4 *tab;        // the probit table staged in LDS by the kernel
    __device__ __forceinline__ u32x4 block(uint32_t stream_block) const
    {
        return philox4x32(env_lo, env_hi, t, stream_block, seed_lo, seed_hi);
    }
};

// One standard normal from one 32-bit word: piecewise-cubic inverse normal CDF.
// Bit 31 is the sign; the next 23 bits m give the tail probability p = (m + 0.5) * 2^-24 in
// (0, 0.5); the piece is picked by the float32 exponent and top 5 mantissa bits of f = m + 0.5
// (24 binades x 32 = 768 pieces, narrower towards the tail), the low 18 mantissa bits are the
// position inside the piece.  Max |error| 4.8e-7 (the float32 grid at z ~ 5); |z| <= 5.42.
// ~13 VALU + one 16-byte LDS read, against ~43 VALU per normal for a polynomial Box-Muller
// (log + sqrt + sincos) -- PowerGrid draws 23 normals per env-step.  The table is generated data
// (gen_probit_table.py); the oracle compiles the same data and runs the same float32 sequence.
__device__ const float4 NIG_PROBIT[768] = {
#include "nig_probit_table.inc"
};

constexpr int PROBIT_BIAS = 192;      // see probit_fetch: an LDS copy of the table belongs at byte 16 x PROBIT_BIAS or above

// In two halves so a kernel can put other work between the LDS read and its use.
struct ProbitFetch { float4 c; float t; uint32_t word; };

__device__ __forceinline__ ProbitFetch probit_fetch(uint32_t word, const float4 *tab)
{
    ProbitFetch f;
    f.word = word;
    // 64 (4 m + 2) = 256 m + 128 = the word with its sign bit and low byte replaced by 0x80: ONE v_and_or_b32 on the word as it
    // stands, then the conversion -- exact, since 2 m + 1 is an odd 24-bit integer (tests/test_host_logic.py checks all 2^23 m).
    // Same mantissa as m + 1/2, exponent field 134 .. 157 for the 24 binades: the piece number (binade x 32 + top five mantissa
    // bits) is bits 27 .. 18 as they stand, plus PROBIT_BIAS = (134 & 31) x 32 = 192 -- a constant that rides in the address
    // (the DS instruction's offset field when the table sits at LDS byte PROBIT_BIAS x 16 = 3 072 or above; the global load's
    // immediate otherwise).  Round 4 extracted m (v_bfe_u32), converted it and formed 4 m + 2 by a fused multiply-add: one
    // instruction more per normal for the same bits.
    // (the mask lives in a VECTOR register: v_and_or_b32 may carry one literal, and 0x80 is not an inline constant -- with both
    // as literals hipcc emits v_and_b32 + v_or_b32, i.e. nothing saved; the asm has no inputs, identical copies are merged and
    // hoisted: one v_mov_b32 per kernel, like opaque_zero64)
#ifdef NIG_DIAG_PROBIT_R04             // (diagnostic builds only: round 4's three-instruction form, for same-box A/Bs)
    const float x = __builtin_fmaf((float)__builtin_amdgcn_ubfe(word, 8, 23), 4.0f, 2.0f);
    const uint32_t b = f32_bits(x);
    tab += PROBIT_BIAS;                    // (its piece numbers start at 0)
#else
    uint32_t keep;
    asm("v_mov_b32 %0, 0x7fffff00" : "=v"(keep));
    const float x = (float)((word & keep) | 0x80u);
    const uint32_t b = f32_bits(x);
#endif
#ifdef NIG_DIAG_PROBIT_NOCONFLICT      // (diagnostic builds only, profiles/r05: what would a conflict-free gather be worth?  Lane l reads the
    // entry of ITS 16-byte slot in the wanted entry's 256-byte bank row -- a neighbouring piece, so the values are off by less than
    // half a binade and the workload's statistics barely move, but no two lanes of a ds_read_b128 group share a bank: same
    // instruction count (v_and_or_b32 for v_and_b32), zero bank conflicts)
    f.c = (tab - PROBIT_BIAS)[((b >> 18) & 0x3F0u) | (__builtin_amdgcn_workitem_id_x() & 15u)];
#else
    f.c = (tab - PROBIT_BIAS)[(b >> 18) & 0x3FFu];
#endif
    f.t = (float)(b & 0x3FFFFu);       // position in the piece as an integer: its 2^-18 is folded into the table's coefficients (bit-identical, tests/probit_scale_check.c)
    return f;
}

__device__ __forceinline__ float probit_eval(const ProbitFetch &f)
{
    float z = __builtin_fmaf(f.c.w, f.t, f.c.z);           // Horner in three fused steps (the generator is this
    z = __builtin_fmaf(z, f.t, f.c.y);                     // build's own spec; the CPU restatement fuses too)
    z = __builtin_fmaf(z, f.t, f.c.x);
    // sign = bit 31 of the word: z ^ (word & 0x80000000) in one v_bitop3_b32 (truth table (a & b) ^ c = 0x6a),
    // the same bits as "bit set ? -z : z"
    return bits_f32(__builtin_amdgcn_bitop3_b32(f.word, 0x80000000u, f32_bits(z), 0x6a));
}

__device__ __forceinline__ float probit_normal(uint32_t word, const float4 *tab)
{
    return probit_eval(probit_fetch(word, tab));
}

__device__ __forceinline__ float u01f(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // exact in float32

__device__ __forceinline__ double u01(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }

// N standard normals of stream `stream` into z[0..N): one word each, four per Philox block.
// Many normals (PowerGrid: 23 per step): software-pipelined by hand -- the table reads of block j+1 are issued
// (behind its ten Philox rounds) before the cubics of block j run, and every finished normal is pinned with an
// empty asm so the evaluation cannot be deferred.  Left alone, hipcc issues the reads block by block but keeps ALL
// coefficient quadruples (4 VGPRs per normal: 92 for PowerGrid) alive until one evaluation burst at the end, which
// pushed the fused PowerGrid rollout into scratch spills.
template <int N>
__device__ __forceinline__ void gen_normals(const RngKey &k, uint32_t stream, float (&z)[N])
{
    constexpr int NB = (N + 3) / 4;
    if constexpr (N > 8) {
        ProbitFetch f[2][4];
        auto fetch = [&](int j, ProbitFetch (&g)[4]) __attribute__((always_inline)) {
            const u32x4 x = k.block(stream + (uint32_t)j);
            g[0] = probit_fetch(x.x, k.tab);
            if (4 * j + 1 < N) g[1] = probit_fetch(x.y, k.tab);
            if (4 * j + 2 < N) g[2] = probit_fetch(x.z, k.tab);
            if (4 * j + 3 < N) g[3] = probit_fetch(x.w, k.tab);
        };
        fetch(0, f[0]);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j + 1 < NB) fetch(j + 1, f[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * j + q < N) {
                    z[4 * j + q] = probit_eval(f[j & 1][q]);
                    asm volatile("" : "+v"(z[4 * j + q]));
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const u32x4 x = k.block(stream + (uint32_t)j);
            if (4 * j + 0 < N) z[4 * j + 0] = probit_normal(x.x, k.tab);
            if (4 * j + 1 < N) z[4 * j + 1] = probit_normal(x.y, k.tab);
            if (4 * j + 2 < N) z[4 * j + 2] = probit_normal(x.z, k.tab);
            if (4 * j + 3 < N) z[4 * j + 3] = probit_normal(x.w, k.tab);
        }
    }
}

template <int N>
__device__ __forceinline__ void gen_uniforms(const RngKey &k, uint32_t stream, double (&u)[N])
{
#pragma unroll
    for (int j = 0; 4 * j < N; ++j) {
        const u32x4 x = k.block(stream + (uint32_t)j);
        if (4 * j + 0 < N) u[4 * j + 0] = u01(x.x);
        if (4 * j + 1 < N) u[4 * j + 1] = u01(x.y);
        if (4 * j + 2 < N) u[4 * j + 2] = u01(x.z);
        if (4 * j + 3 < N) u[4 * j + 3] = u01(x.w);
    }
}

}  // namespace nig
