/* Exhaustive check that the normal transform's table with the position scale folded into the coefficients
 * ({c0, c1 2^-18, c2 2^-36, c3 2^-54}, position p = the low 18 mantissa bits as an integer-valued float) returns the
 * same bits as the unscaled form (coefficients {c0..c3}, position t = p 2^-18) over ALL 768 pieces x 2^18 positions.
 * Test infrastructure (tests/test_host_logic.py); the table is the generated data the library and the oracle compile. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static const float TAB[768][4] = {
#include "../oracle/nig_probit_table.inc"
};

int main(void)
{
    long bad = 0;
    for (int k = 0; k < 768; k++) {
        const float c0 = TAB[k][0], s1 = TAB[k][1], s2 = TAB[k][2], s3 = TAB[k][3];
        const float c1 = ldexpf(s1, 18), c2 = ldexpf(s2, 36), c3 = ldexpf(s3, 54);      /* the unscaled coefficients, exactly */
        if (ldexpf(c1, -18) != s1 || ldexpf(c2, -36) != s2 || ldexpf(c3, -54) != s3) { printf("piece %d: scaling not exact\n", k); return 2; }
        for (uint32_t p = 0; p < (1u << 18); p++) {
            const float pf = (float)p, t = pf * (1.0f / 262144.0f);
            float za = fmaf(c3, t, c2); za = fmaf(za, t, c1); za = fmaf(za, t, c0);
            float zb = fmaf(s3, pf, s2); zb = fmaf(zb, pf, s1); zb = fmaf(zb, pf, c0);
            if (memcmp(&za, &zb, 4) != 0) { if (bad < 5) printf("piece %d p %u: %a vs %a\n", k, p, za, zb); bad++; }
        }
    }
    printf("checked %ld inputs, mismatches=%ld\n", 768L << 18, bad);
    return bad != 0;
}
