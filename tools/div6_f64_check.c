// exhaustive check of the pair sampler's visit index (sampler.hip, sample_pairs_kernel): for EVERY finite f32 normal z of
// magnitude <= 16 (Box-Muller on a 24-bit uniform gives |z| <= 5.8), x = 3.0 + (double)z,
//     q = x*RN(1/6);  q += fma(-6, q, x)*RN(1/6)     equals     x / 6.0      (the IEEE quotient NumPy computes, DS:15)
// so floor(q * S) is bit for bit np.floor((3 + 1*z)/6*S) on every draw.
// gcc -O2 -fopenmp -ffp-contract=off tools/div6_f64_check.c -lm && ./a.out   ->  "... checked, 0 bad"
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main() {
    const double y = 1.0 / 6.0;
    long bad = 0, n = 0;
    float lim = 16.f;
    uint32_t top;
    memcpy(&top, &lim, 4);
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
    for (int64_t b = 0; b <= (int64_t)top; b++) {
        uint32_t bits = (uint32_t)b;
        float zf;
        memcpy(&zf, &bits, 4);
        for (int sg = 0; sg < 2; sg++) {
            const double x = 3.0 + (double)(sg ? -zf : zf);
            double q = x * y;
            const double r = fma(-6.0, q, x);
            q = fma(r, y, q);
            if (q != x / 6.0) bad++;
            n++;
        }
    }
    printf("%ld checked, %ld bad\n", n, bad);
    return bad != 0;
}
