/* Exhaustive proof, per divisor c, that the device's 4-instruction constant division
 * (fdiv_c in csrc/nig_detmath.hpp: q0 = x*RN(1/c); r = fma(-c, q0, x); q = fma(r, RN(1/c), q0))
 * equals the correctly rounded x / c the reference's NumPy arithmetic performs.  For a fixed c the
 * outcome depends only on the significand of x (scaling x by 2^k scales q0, r and q exactly while
 * nothing leaves the normal range), so all 2^23 significands in a few binades are the whole proof.
 * Usage: constdiv_check c1 c2 ...   exit status = number of divisors with a mismatch.            */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

int main(int argc, char **argv)
{
    int failed = 0;
    for (int a = 1; a < argc; ++a) {
        const float c = strtof(argv[a], 0);
        volatile float rcv = 1.0f / c;
        const float rc = rcv;
        long bad = 0;
        static const int exps[4] = {60, 100, 127, 180};
        for (int e = 0; e < 4; ++e)
            for (uint32_t m = 0; m < (1u << 23); ++m)
                for (uint32_t sign = 0; sign < 2; ++sign) {
                    const float x = from_bits((sign << 31) | ((uint32_t)exps[e] << 23) | m);
                    const float q0 = x * rc;
                    const float r = fmaf(-c, q0, x);
                    const float q = fmaf(r, rc, q0);
                    if (q != x / c) ++bad;
                }
        printf("c=%.9g mismatches=%ld\n", c, bad);
        failed += bad != 0;
    }
    return failed;
}
