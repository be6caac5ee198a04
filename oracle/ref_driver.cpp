/*
 * ref_driver.cpp -- serial NDRange driver for the reference's own OpenCL-C kernels, part of
 * _ref/libmg_ref.so.  TEST INFRASTRUCTURE ONLY (see mg_oracle.h).
 *
 * The Makefile compiles /root/reference/mg_0.cl, mg_1.cl and common_kernels.cl, from where they
 * lie, to x86-64 objects (float and double), renaming their kernel symbols per variant with -D.
 * This file supplies the handful of OpenCL built-ins those objects import and runs each kernel
 * over its global range one work-item at a time in the canonical order: id0 fastest, id2 slowest.
 * No reference source is copied here; only the kernels' C signatures are declared.
 *
 * Deliberately free of <cmath>/<math.h>: the built-ins isnan/fabs/min/max are defined below with
 * the C++-mangled names the OpenCL objects reference.
 */
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define MGO_PREFIX mgr_
#include "mg_oracle.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i2 __attribute__((ext_vector_type(2)));

/* ---- work-item state and OpenCL built-ins (C++ linkage on purpose: OpenCL built-ins are
 * overloadable, so the kernel objects import them under their mangled names) ------------------ */
static size_t g_id[3] = {0, 0, 0}, g_size[3] = {1, 1, 1};
size_t get_global_id(unsigned d) { return g_id[d]; }
size_t get_global_size(unsigned d) { return g_size[d]; }
/* dot(): left to right, one rounding per operation; this file is built with -ffp-contract=off */
float dot(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
double dot(d4 a, d4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
int isnan(float x) { return x != x; }
int isnan(double x) { return x != x; }
float fabs(float x) { return x < 0 ? -x : x; }
double fabs(double x) { return x < 0 ? -x : x; }
float max(float a, float b) { return a < b ? b : a; }
double max(double a, double b) { return a < b ? b : a; }
int min(int a, int b) { return b < a ? b : a; }

template <class F> static void ndrange(size_t a, size_t b, size_t c, F f)
{
    g_size[0] = a;
    g_size[1] = b;
    g_size[2] = c;
    for (g_id[2] = 0; g_id[2] < c; ++g_id[2])
        for (g_id[1] = 0; g_id[1] < b; ++g_id[1])
            for (g_id[0] = 0; g_id[0] < a; ++g_id[0])
                f();
    g_id[0] = g_id[1] = g_id[2] = 0;
}

/* ---- cell maps: NaN-encoded normals, the encoding of rectangularborderhandler.cpp:56-58,99-101
 * as decoded by getCellType (mg_1.cl:14-20): (0,0) inner, (nan,nan) outer, (nan,1) Dirichlet,
 * anything else Neumann. */
template <class V, class T> static std::vector<V> encode_cells(const uint8_t *cell, size_t n)
{
    const T nan = (T)__builtin_nan("");
    std::vector<V> out(n);
    for (size_t i = 0; i < n; ++i) {
        V v = (V)(T)0;
        switch (cell[i]) {
        case MGO_INNER: break;
        case MGO_OUTER: v.x = nan; v.y = nan; break;
        case MGO_DIRICHLET: v = (V)(T)1; v.x = nan; break;
        default: v.x = (T)1; break; /* Neumann */
        }
        out[i] = v;
    }
    return out;
}

/* ---- kernel symbols, one set per (dimension, precision): see Makefile for the -D renames ---- */
#define DECL_MG(P, CELL, R, ISZ)                                                                   \
    extern "C" {                                                                                   \
    void P##iteration_kernel(CELL *, R *, R *, R, ISZ, int);                                       \
    void P##residual_kernel(CELL *, R *, R *, R *);                                                \
    void P##reduction_kernel(CELL *, R *, R *, ISZ);                                               \
    void P##residual_correct_kernel(CELL *, R *, R *, R *);                                        \
    void P##prolongation_kernel(CELL *, R *, R *);                                                 \
    void P##zero_out(CELL *, R *);                                                                 \
    }
DECL_MG(r2f_, f2, float, i2)
DECL_MG(r3f_, f4, float, i4)
DECL_MG(r2d_, d2, double, i2)
DECL_MG(r3d_, d4, double, i4)

#define DECL_COMMON(P, R)                                                                          \
    extern "C" {                                                                                   \
    void P##zero_memory(R *);                                                                      \
    void P##L2Norm(R *, R *);                                                                      \
    void P##LInfNorm(R *, R *, int, int);                                                          \
    void P##SumAll(R *, R *, int, int);                                                            \
    void P##Diff(R *, R *, R *);                                                                   \
    void P##Mult(R *, R *, float);                                                                 \
    }
DECL_COMMON(rcf_, float)
DECL_COMMON(rcd_, double)

/* ---- exported wrappers: same launch ranges as the host code of the reference ----------------- */
#define WRAP(R, S, P2, P3, PC, C2, C3)                                                             \
    extern "C" void mgr_smooth_pass##S(int dim, const uint8_t *cell, R *u, const R *f, R omega,    \
                                       int nx, int ny, int nz, int odd, int zcount)                \
    { /* range (ceil(nx/2), ny, zcount): multigridsolver3D.cpp:121, multigridsolver2D.cpp:136 */   \
        if (dim == 3) {                                                                            \
            auto c = encode_cells<C3, R>(cell, (size_t)nx * ny * nz);                              \
            i4 sz = {nx, ny, nz, 1};                                                               \
            ndrange(nx / 2 + nx % 2, ny, zcount,                                                   \
                    [&] { P3##iteration_kernel(c.data(), u, (R *)f, omega, sz, odd); });           \
        } else {                                                                                   \
            auto c = encode_cells<C2, R>(cell, (size_t)nx * ny);                                   \
            i2 sz = {nx, ny};                                                                      \
            ndrange(nx / 2 + nx % 2, ny, 1,                                                        \
                    [&] { P2##iteration_kernel(c.data(), u, (R *)f, omega, sz, odd); });           \
        }                                                                                          \
    }                                                                                              \
    extern "C" void mgr_residual##S(int dim, const uint8_t *cell, R *r, const R *u, const R *f,    \
                                    int nx, int ny, int nz)                                        \
    { /* multigridsolver3D.cpp:134-143 */                                                          \
        if (dim == 3) {                                                                            \
            auto c = encode_cells<C3, R>(cell, (size_t)nx * ny * nz);                              \
            ndrange(nx, ny, nz, [&] { P3##residual_kernel(c.data(), r, (R *)u, (R *)f); });        \
        } else {                                                                                   \
            auto c = encode_cells<C2, R>(cell, (size_t)nx * ny);                                   \
            ndrange(nx, ny, 1, [&] { P2##residual_kernel(c.data(), r, (R *)u, (R *)f); });         \
        }                                                                                          \
    }                                                                                              \
    extern "C" void mgr_restrict##S(int dim, const uint8_t *cell, R *coarse, const R *fine,        \
                                    int nx, int ny, int nz, int cx, int cy, int cz)                \
    { /* multigridsolver3D.cpp:145-157: coarse launch range, fine cell map and size */             \
        if (dim == 3) {                                                                            \
            auto c = encode_cells<C3, R>(cell, (size_t)nx * ny * nz);                              \
            i4 sz = {nx, ny, nz, 1};                                                               \
            ndrange(cx, cy, cz, [&] { P3##reduction_kernel(c.data(), coarse, (R *)fine, sz); });   \
        } else {                                                                                   \
            auto c = encode_cells<C2, R>(cell, (size_t)nx * ny);                                   \
            i2 sz = {nx, ny};                                                                      \
            ndrange(cx, cy, 1, [&] { P2##reduction_kernel(c.data(), coarse, (R *)fine, sz); });    \
        }                                                                                          \
    }                                                                                              \
    extern "C" void mgr_correct##S(int dim, const uint8_t *cell, R *out, const R *u, const R *e,   \
                                   int nx, int ny, int nz)                                         \
    { /* multigridsolver3D.cpp:159-171 */                                                          \
        if (dim == 3) {                                                                            \
            auto c = encode_cells<C3, R>(cell, (size_t)nx * ny * nz);                              \
            ndrange(nx, ny, nz,                                                                    \
                    [&] { P3##residual_correct_kernel(c.data(), out, (R *)u, (R *)e); });          \
        } else {                                                                                   \
            auto c = encode_cells<C2, R>(cell, (size_t)nx * ny);                                   \
            ndrange(nx, ny, 1,                                                                     \
                    [&] { P2##residual_correct_kernel(c.data(), out, (R *)u, (R *)e); });          \
        }                                                                                          \
    }                                                                                              \
    extern "C" void mgr_prolong##S(int dim, const uint8_t *cell, R *out, const R *coarse, int nx,  \
                                   int ny, int nz)                                                 \
    { /* multigridsolver3D.cpp:173-184 */                                                          \
        if (dim == 3) {                                                                            \
            auto c = encode_cells<C3, R>(cell, (size_t)nx * ny * nz);                              \
            ndrange(nx, ny, nz, [&] { P3##prolongation_kernel(c.data(), out, (R *)coarse); });     \
        } else {                                                                                   \
            auto c = encode_cells<C2, R>(cell, (size_t)nx * ny);                                   \
            ndrange(nx, ny, 1, [&] { P2##prolongation_kernel(c.data(), out, (R *)coarse); });      \
        }                                                                                          \
    }                                                                                              \
    extern "C" void mgr_zero_out##S(int dim, const uint8_t *cell, R *u, int nx, int ny, int nz)    \
    { /* multigridsolver3D.cpp:192-198 */                                                          \
        if (dim == 3) {                                                                            \
            auto c = encode_cells<C3, R>(cell, (size_t)nx * ny * nz);                              \
            ndrange(nx, ny, nz, [&] { P3##zero_out(c.data(), u); });                               \
        } else {                                                                                   \
            auto c = encode_cells<C2, R>(cell, (size_t)nx * ny);                                   \
            ndrange(nx, ny, 1, [&] { P2##zero_out(c.data(), u); });                                \
        }                                                                                          \
    }                                                                                              \
    /* performReduction, auxiliary.cpp:119-135: size -> max(1, size/4), chunk 4, fresh buffer */   \
    static std::vector<R> reduce_tree##S(std::vector<R> in, void (*ker)(R *, R *, int, int))       \
    {                                                                                              \
        int size = (int)in.size();                                                                 \
        while (size != 1) {                                                                        \
            const int newsize = size / 4 > 1 ? size / 4 : 1;                                       \
            std::vector<R> tmp(newsize);                                                           \
            ndrange(newsize, 1, 1, [&] { ker(in.data(), tmp.data(), size, 4); });                  \
            in.swap(tmp);                                                                          \
            size = newsize;                                                                        \
        }                                                                                          \
        return in;                                                                                 \
    }                                                                                              \
    extern "C" R mgr_l2sum_tree##S(const R *x, long n)                                             \
    { /* auxiliary.cpp:137-173 without the final host sqrt, which ref_cycle.c applies */           \
        std::vector<R> sq((size_t)n);                                                              \
        ndrange((size_t)n, 1, 1, [&] { PC##L2Norm((R *)x, sq.data()); });                          \
        return reduce_tree##S(sq, PC##SumAll)[0];                                                  \
    }                                                                                              \
    extern "C" R mgr_linfnorm_tree##S(const R *x, long n)                                          \
    { /* auxiliary.h:72-82 */                                                                      \
        return reduce_tree##S(std::vector<R>(x, x + n), PC##LInfNorm)[0];                          \
    }                                                                                              \
    extern "C" void mgr_diff##S(R *out, const R *a, const R *b, long n)                            \
    { /* auxiliary.h:96-109 */                                                                     \
        ndrange((size_t)n, 1, 1, [&] { PC##Diff(out, (R *)a, (R *)b); });                          \
    }                                                                                              \
    extern "C" void mgr_scale##S(R *out, const R *in, float m, long n)                             \
    { /* Mult has no launch site in the reference: one work-item per element, as for Diff */       \
        ndrange((size_t)n, 1, 1, [&] { PC##Mult((R *)in, out, m); });                              \
    }

WRAP(float, _f32, r2f_, r3f_, rcf_, f2, f4)
WRAP(double, _f64, r2d_, r3d_, rcd_, d2, d4)
