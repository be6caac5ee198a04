/*
 * ref_cycle.c -- host-side half of _ref/libmg_ref.so: the restated cycle drivers and problem
 * setup (mg_cycle.inc, mg_common.inc) running over the reference's own compiled kernels, which
 * ref_driver.cpp wraps.  TEST INFRASTRUCTURE ONLY (see mg_oracle.h).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MGO_PREFIX mgr_
#include "mg_oracle.h"

#include "mg_common.inc"

float mgr_l2sum_tree_f32(const float *x, long n);
double mgr_l2sum_tree_f64(const double *x, long n);

/* auxiliary.cpp:151-153,170-172: blocking read of the tree's single result, then sqrt */
float mgr_l2norm_tree_f32(const float *x, long n) { return (float)sqrt((double)mgr_l2sum_tree_f32(x, n)); }
double mgr_l2norm_tree_f64(const double *x, long n) { return sqrt(mgr_l2sum_tree_f64(x, n)); }

#define R float
#define S _f32
#include "mg_cycle.inc"
#undef R
#undef S

#define R double
#define S _f64
#include "mg_cycle.inc"
#undef R
#undef S

int mgr_num_threads(void) { return 1; }
void mgr_set_num_threads(int n) { (void)n; }
