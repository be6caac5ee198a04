/*
 * mg_oracle.c -- libmg_oracle.so: the build's own CPU restatement of the reference multigrid
 * hot path (kernels of mg_0.cl / mg_1.cl / common_kernels.cl and the cycle drivers of
 * multigridsolver2D/3D.cpp).  TEST INFRASTRUCTURE ONLY: see mg_oracle.h for who may load it.
 *
 * Parity pinning: checked bit for bit against _ref/libmg_ref.so (the reference's own kernel
 * sources compiled for x86-64) by tests/test_oracle.py whenever that library has been built,
 * and against the golden vectors committed under tests/golden/, which were generated from it.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fopenmp -fPIC -shared (see Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MGO_PREFIX mgo_
#include "mg_oracle.h"

#include "mg_common.inc"

#define R float
#define S _f32
#include "mg_kernels.inc"
#include "mg_cycle.inc"
#undef R
#undef S

#define R double
#define S _f64
#include "mg_kernels.inc"
#include "mg_cycle.inc"
#undef R
#undef S

int mgo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void mgo_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
#else
    (void)n;
#endif
}
