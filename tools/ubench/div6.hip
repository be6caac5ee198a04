// div6.hip -- whether the hardware's sticky floating-point exception bits can guard a single-precision
// replacement of the smoother's x * (1.0/6.0) (mg_1.cl:42: evaluated in double, rounded to float).
//   exact: (float)((double)s * (1.0/6.0))        v_cvt_f64_f32, v_mul_f64, v_cvt_f32_f64
//   fast:  fma(s, c_hi, s * c_lo)                v_mul_f32, v_fma_f32      c_hi + c_lo = 1/6 to 2^-50
// Every one of the 2^32 bit patterns goes through both, a wavefront taking 64 consecutive patterns; after
// the fast form the wavefront reads TRAPSTS.EXCP.  Printed: where the forms differ, and whether every
// wavefront that holds a difference has the underflow or input-denormal bit set.
// Build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o div6.bin div6.hip && ./div6.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

struct Tally {
    unsigned long long differ, waves_differ, waves_unflagged, waves_flagged_same, excp_or_differ, excp_hist[9];
    unsigned first_unflagged;
};

__global__ __launch_bounds__(256) void k(Tally *t, unsigned base, float c_hi, float c_lo)
{
    const unsigned u = base + blockIdx.x * 256u + threadIdx.x;
    const float s = __uint_as_float(u);
    // the fast form first, on a wavefront whose exception bits are still clear
    float p, q;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(s), "v"(c_lo));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q) : "v"(s), "v"(c_hi), "v"(p));
    unsigned excp;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(excp) : "v"(q));
    // the reference form after the bits were read (volatile statements keep their order)
    double d;
    float ref;
    const double sixth = 1.0 / 6.0;
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(s));
    asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(sixth));
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(ref) : "v"(d));
    const bool differ = __float_as_uint(ref) != __float_as_uint(q);
    const unsigned long long m = __ballot(differ);
    if (differ)
        atomicAdd(&t->differ, 1ull);
    if ((threadIdx.x & 63) == 0) {
        const bool flagged = (excp & 0x12u) != 0; // bit 1 input denormal, bit 4 underflow
        if (m) {
            atomicAdd(&t->waves_differ, 1ull);
            atomicOr(&t->excp_or_differ, (unsigned long long)excp);
            if (!flagged) {
                atomicAdd(&t->waves_unflagged, 1ull);
                atomicMin(&t->first_unflagged, u);
            }
        } else if (flagged) {
            atomicAdd(&t->waves_flagged_same, 1ull);
        }
        for (int b = 0; b < 9; ++b)
            if (excp >> b & 1)
                atomicAdd(&t->excp_hist[b], 1ull);
    }
}

int main()
{
    Tally *d, h;
    hipMalloc(&d, sizeof(Tally));
    memset(&h, 0, sizeof(h));
    h.first_unflagged = 0xffffffffu;
    hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
    unsigned hi_bits = 0x3e2aaaaau;
    float c_hi;
    memcpy(&c_hi, &hi_bits, 4);
    const float c_lo = (float)(1.0 / 6.0 - (double)c_hi);
    for (unsigned chunk = 0; chunk < 16; ++chunk) // 2^28 patterns a launch
        k<<<1u << 20, 256>>>(d, chunk << 28, c_hi, c_lo);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("c_hi %a  c_lo %a\n", c_hi, c_lo);
    printf("inputs whose results differ           %llu of 4294967296\n", h.differ);
    printf("wavefronts holding a difference       %llu of 67108864\n", h.waves_differ);
    printf("  of those without underflow/denormal %llu (first input 0x%08x)\n", h.waves_unflagged, h.first_unflagged);
    printf("  exception bits seen on them (or)    0x%llx\n", h.excp_or_differ);
    printf("wavefronts flagged without difference %llu\n", h.waves_flagged_same);
    const char *names[9] = {"invalid", "input denormal", "divide by 0", "overflow", "underflow", "inexact", "int divide by 0", "address watch", "memory violation"};
    for (int b = 0; b < 9; ++b)
        printf("  bit %d %-18s set on %llu wavefronts\n", b, names[b], h.excp_hist[b]);
    return 0;
}
