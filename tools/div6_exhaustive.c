/* div6_exhaustive.c -- every one of the 2^32 float bit patterns s through the smoother's scaling as the
 * reference's expression reads it, (float)((double)s * (1.0/6.0)) (mg_1.cl:42: the literal makes the product
 * double), and through candidate single-precision sequences, counting where the bits differ.
 *   A: q = s*c; r = fma(-6, q, s); q' = fma(r, c, q)          c = 1/6 rounded to nearest (0x3e2aaaab)
 *   C: p = s*c_lo; q = fma(s, c_hi, p)                         c_hi = 1/6 rounded down, c_lo = 1/6 - c_hi
 * and the 2D smoother's (float)((double)s * 0.25) (mg_0.cl:38) against the single product s * 0.25f (column Q:
 * both round the exact quotient once, so no input may differ).
 * Build: gcc -O2 -mfma -ffp-contract=off -fopenmp -o div6 div6_exhaustive.c -lm   (x86 fmaf is the IEEE fused
 * operation, as v_fma_f32 is).  A development aid: the product keeps the double-precision form. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline uint32_t bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float fl(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

int main(void)
{
    const float c = fl(0x3e2aaaabu), chi = fl(0x3e2aaaaau);
    const float clo = (float)(1.0 / 6.0 - (double)chi);
    const char *cls[] = {"zero", "subnormal", "normal, |s| < 2^-100", "normal, other", "inf", "nan"};
    unsigned long long badA[6] = {0}, badC[6] = {0}, badQ[6] = {0}, n[6] = {0};
    float firstA[6], firstC[6];
    int haveA[6] = {0}, haveC[6] = {0};
#pragma omp parallel for schedule(static) reduction(+ : badA[:6], badC[:6], badQ[:6], n[:6])
    for (long long i = 0; i < (1LL << 32); ++i) {
        const uint32_t u = (uint32_t)i;
        volatile float s = fl(u);
        const uint32_t e = (u >> 23) & 0xff, m = u & 0x7fffff;
        const int k = e == 0 ? (m == 0 ? 0 : 1) : e == 255 ? (m == 0 ? 4 : 5) : e < 27 ? 2 : 3;
        const float ref = (float)((double)s * (1.0 / 6.0));
        const float q = s * c, r = fmaf(-6.0f, q, s), a = fmaf(r, c, q);
        const float p = s * clo, cc = fmaf(s, chi, p);
        const float qref = (float)((double)s * 0.25), qf = s * 0.25f;
        if (bits(qf) != bits(qref))
            badQ[k]++;
        n[k]++;
        if (bits(a) != bits(ref)) {
            badA[k]++;
            if (!haveA[k]) haveA[k] = 1, firstA[k] = s;
        }
        if (bits(cc) != bits(ref)) {
            badC[k]++;
            if (!haveC[k]) haveC[k] = 1, firstC[k] = s;
        }
    }
    printf("%-24s %12s %14s %14s %14s\n", "class of s", "inputs", "A differs", "C differs", "Q differs");
    for (int k = 0; k < 6; ++k)
        printf("%-24s %12llu %14llu %14llu %14llu   e.g. A: %a  C: %a\n", cls[k], n[k], badA[k], badC[k], badQ[k],
               haveA[k] ? firstA[k] : 0.0f, haveC[k] ? firstC[k] : 0.0f);
    return 0;
}
