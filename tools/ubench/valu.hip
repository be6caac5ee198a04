// valu.hip -- issue rate of the vector instructions the smoother is made of, per SIMD, with 1, 2 and 4
// wavefronts per SIMD: cycles per wave-instruction = kernel time * clock / instructions per SIMD.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o valu.bin valu.hip && ./valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP> __global__ void k(float *out, int iters, float seed)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x + i;
    double d0 = seed, d1 = seed + 1;
    const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)iters;
    const unsigned vmask = (threadIdx.x & 1) ? 0xffffffffu : 0u, vsel = (threadIdx.x & 1) ? 0x03020100u : 0x07060504u;
    const float vsixm = -6.0f;
    asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "s"(mask));
            if (OP == 10) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]), "s"(mask));
            if (OP == 11) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]));
            if (OP == 2) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&a[i & 14]) : "v"(d0));
            if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d1) : "v"(a[i]));
            if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d1) : "v"(d0));
            if (OP == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d0));
            if (OP == 7) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            // selects that take their condition from a vector register (all ones / zero per lane) instead of a scalar pair
            if (OP == 12) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(vmask), "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]));
            if (OP == 13) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]), "v"(vsel));
            if (OP == 14) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]) : );
            if (OP == 15) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]), "v"(a[(i + 3) & 15]));
            if (OP == 16) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d1) : "v"(d0));
            if (OP == 17) asm volatile("v_ldexp_f64 %0, %0, -3" : "+v"(d1));
            if (OP == 18) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&a[i & 14]) : "v"(d0));
            if (OP == 19) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&a[i & 14]) : "v"(d0));
            // the mix of an update: adds and selects alternating
            if (OP == 20) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[(i + 8) & 15]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]), "s"(mask));
            }
            if (OP == 21) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a[(i + 8) & 15]) : "v"(vmask), "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]));
            }
            // x/6 three ways: through double as the reference's expression reads, and the two-step correction in single
            if (OP == 22) {
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d1) : "v"(a[i]));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d1) : "v"(d0));
                asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d1));
            }
            if (OP == 23) {
                float q, r;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q) : "v"(a[i]), "v"(seed));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(vsixm), "v"(q), "v"(a[i]));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(r), "v"(seed), "v"(q));
            }
        }
    }
    float s = (float)d1;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678f) out[0] = s;
}

template <int OP> void run(const char *name, float *out)
{
    const int iters = 4000;
    printf("%-20s", name);
    for (int waves : {4, 8, 16}) { // per CU: 1, 2, 4 per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        k<OP><<<256, waves * 64>>>(out, 10, 1.0f);
        hipEventRecord(e0);
        k<OP><<<256, waves * 64>>>(out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)iters * 16 * (waves / 4);
        printf("  %d/SIMD: %6.2f ns per instr", waves / 4, ms * 1e6 / per_simd);
    }
    printf("\n");
}

int main()
{
    float *out;
    hipMalloc(&out, 4);
    run<0>("v_add_f32", out);
    run<9>("v_mul_f32", out);
    run<8>("v_fma_f32", out);
    run<1>("v_cndmask_b32", out);
    run<10>("v_cndmask 3-reg", out);
    run<11>("v_sub_f32 3-reg", out);
    run<2>("v_mov_b32_dpp", out);
    run<7>("v_add_f32_dpp", out);
    run<3>("v_pk_add_f32", out);
    run<4>("v_cvt_f64_f32", out);
    run<5>("v_mul_f64", out);
    run<6>("v_cvt_f32_f64", out);
    run<12>("v_bfi_b32", out);
    run<13>("v_perm_b32", out);
    run<14>("v_cndmask vcc", out);
    run<15>("v_fma_f32 4-reg", out);
    run<16>("v_fma_f64", out);
    run<17>("v_ldexp_f64", out);
    run<18>("v_pk_fma_f32", out);
    run<19>("v_pk_mul_f32", out);
    run<20>("add+cndmask (2)", out);
    run<21>("add+bfi (2)", out);
    run<22>("x/6 via f64 (3)", out);
    run<23>("x/6 mul,fma,fma (3)", out);
    printf("(time per loop slot and SIMD: the rows marked (2) and (3) issue that many instructions per slot; at 2.4 GHz one cycle is 0.417 ns)\n");
    return 0;
}
