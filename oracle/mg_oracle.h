/*
 * mg_oracle.h -- CPU oracle for the multigrid V-cycle hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Two libraries export this same interface (with different symbol prefixes):
 *
 *   libmg_oracle.so   prefix mgo_   the build's own plain-C restatement of the reference
 *                                   algorithm (mg_oracle.c, mg_kernels.inc), each function
 *                                   citing the reference file:line it follows;
 *   _ref/libmg_ref.so prefix mgr_   the reference's own OpenCL-C kernel sources compiled for
 *                                   x86-64 where they lie under /root/reference and driven by a
 *                                   serial NDRange loop (ref_driver.cpp).  Built only when the
 *                                   reference tree is present; never committed.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load either library,
 * and only as the checker.  The product (libmgrt.so and the host classes above it) never links,
 * loads or calls anything declared here.
 *
 * Conventions (reference: multi_array.h:36,75): arrays are dense, x fastest,
 * index = x + y*nx + z*nx*ny.  2D fields have nz == 1.  `cell` holds one byte per grid point with
 * the values of BorderHandler3D::CellType (multigridsolver3D.h:28):
 *   0 inner, 1 outer, 2 Dirichlet, 3 Neumann.
 * The serial execution order is the canonical NDRange order (id0 fastest, id2 slowest).  For the 3D
 * smoother, whose colouring ignores z (mg_1.cl:98), that order is what makes the result
 * well-defined: inside one colour pass every (x,y) column is a first-order recurrence in ascending z.
 *
 * Every function exists for float (_f32) and double (_f64), the two settings of real.cl:2-18.
 */
#ifndef MG_ORACLE_H
#define MG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef MGO_PREFIX
#define MGO_PREFIX mgo_
#endif
#define MGO_CAT2(a, b) a##b
#define MGO_CAT(a, b) MGO_CAT2(a, b)
#define MGO_NAME(n) MGO_CAT(MGO_PREFIX, n)

enum { MGO_INNER = 0, MGO_OUTER = 1, MGO_DIRICHLET = 2, MGO_NEUMANN = 3 };

/* kernel table: one entry per device kernel of mg_0.cl / mg_1.cl, 2D selected by dim == 2 */
#define MGO_DECLARE(R, S)                                                                          \
    /* iteration_kernel: one colour pass, in place.  3D: planes z < zcount only. */               \
    void MGO_NAME(smooth_pass##S)(int dim, const uint8_t *cell, R *u, const R *f, R omega, int nx, \
                                  int ny, int nz, int odd, int zcount);                            \
    /* residual_kernel: r untouched on outer/Neumann cells */                                      \
    void MGO_NAME(residual##S)(int dim, const uint8_t *cell, R *r, const R *u, const R *f, int nx, \
                               int ny, int nz);                                                    \
    /* reduction_kernel: cell map and (nx,ny,nz) are the FINE grid's; (cx,cy,cz) the launch range */ \
    void MGO_NAME(restrict##S)(int dim, const uint8_t *cell, R *coarse, const R *fine, int nx,     \
                               int ny, int nz, int cx, int cy, int cz);                            \
    /* residual_correct_kernel: out = u + 4*P(e); coarse extents derived as n/2+1 */              \
    void MGO_NAME(correct##S)(int dim, const uint8_t *cell, R *out, const R *u, const R *e,        \
                              int nx, int ny, int nz);                                             \
    /* prolongation_kernel: out = P(coarse) */                                                     \
    void MGO_NAME(prolong##S)(int dim, const uint8_t *cell, R *out, const R *coarse, int nx,       \
                              int ny, int nz);                                                     \
    /* zero_out: clear outer cells */                                                              \
    void MGO_NAME(zero_out##S)(int dim, const uint8_t *cell, R *u, int nx, int ny, int nz);        \
    /* common_kernels.cl + auxiliary.cpp:119-173: the reference's /4 reduction tree, which drops  \
       up to 3 tail elements per level and accumulates in R */                                     \
    R MGO_NAME(l2norm_tree##S)(const R *x, long n);                                                \
    R MGO_NAME(linfnorm_tree##S)(const R *x, long n);                                              \
    void MGO_NAME(diff##S)(R *out, const R *a, const R *b, long n);                                \
    void MGO_NAME(scale##S)(R *out, const R *in, float m, long n);                                 \
    /* host drivers restated from multigridsolver3D.cpp:41-132 (2D: multigridsolver2D.cpp:42-147). \
       zfull != 0 launches the 3D smoother over every z plane instead of ceil(nz/2).              \
       u is updated in place (the reference rebinds the caller's handle to the new data). */      \
    void MGO_NAME(smoother_iterate##S)(int dim, R *u, const R *f, R omega, int a, int nx, int ny,  \
                                       int nz, int zfull);                                         \
    void MGO_NAME(iterate##S)(int dim, R *u, const R *f, R omega, int a1, int a2, int v, int nx,   \
                              int ny, int nz, int zfull);                                          \
    void MGO_NAME(mg##S)(int dim, R *u, const R *f, R omega, int a1, int a2, int v, int iters,     \
                         int nx, int ny, int nz, int zfull);                                       \
    void MGO_NAME(fmg##S)(int dim, R *u_out, const R *f, R omega, int a1, int a2, int v,           \
                          int iters, int nx, int ny, int nz, int zfull);

MGO_DECLARE(float, _f32)
MGO_DECLARE(double, _f64)

/* Box cell maps of rectangularborderhandler.cpp:46-68,89-114 (all faces Dirichlet).
 * face[6] = type of the x-lo, x-hi, y-lo, y-hi, z-lo, z-hi faces lets tests build the mixed
 * Dirichlet/Neumann maps of the benchmark configs; later faces overwrite earlier ones on shared
 * edges in the reference's order (z faces, then x faces, then y faces in 3D; y then x in 2D). */
void MGO_NAME(box_cells)(int dim, uint8_t *cell, int nx, int ny, int nz, const int face[6]);

/* Optional user cell map for the host drivers (NULL = all-Dirichlet box per level).  The callback
 * returns the cell type of point (x,y,z) on a level of extents (nx,ny,nz), i.e. the signature of
 * BorderHandler3D::cellType (multigridsolver3D.h:32). */
typedef int (*mgo_celltype_fn)(int x, int y, int z, int nx, int ny, int nz);
void MGO_NAME(set_celltype)(mgo_celltype_fn fn);
/* The same with a rule built into the library: 0 = none (box), 1 = the ball inscribed in the box
 * (inner core, a shell that is Dirichlet on the low-x side and Neumann on the other, outer cells
 * around it).  For cases too large for a callback per point. */
void MGO_NAME(set_builtin_celltype)(int kind);
/* The box with these types on its x-lo, x-hi, y-lo, y-hi, z-lo, z-hi faces on every level, shared edges as
 * box_cells paints them; NULL = back to the all-Dirichlet box. */
void MGO_NAME(set_box_faces)(const int face[6]);
/* copy of the cell map the host drivers use on a level of these extents under the active rule */
void MGO_NAME(copy_level_cells)(int dim, uint8_t *out, int nx, int ny, int nz);

/* Problem setup restated from functionhandler.cpp:25-101 and programstate.cpp:84-122.
 * problem: 0 = 3D tridimF1/tridimS1, 1 = 2D sinfunc2/sinfunc2sol, 2 = 2D sinfunc1/sinfunc1sol.
 * Uses the cell map installed with set_celltype (box when none). */
void MGO_NAME(discretize_f32)(int problem, float *func, float *sol, int nx, int ny, int nz);
void MGO_NAME(discretize_f64)(int problem, double *func, double *sol, int nx, int ny, int nz);

/* exact norms with fp64 accumulation (checker-side helpers, not reference behaviour) */
double MGO_NAME(l2norm_exact_f32)(const float *x, long n);
double MGO_NAME(l2norm_exact_f64)(const double *x, long n);

/* number of OpenMP threads the restatement uses (1 when built without OpenMP) */
int MGO_NAME(num_threads)(void);
void MGO_NAME(set_num_threads)(int n);

#ifdef __cplusplus
}
#endif
#endif /* MG_ORACLE_H */
