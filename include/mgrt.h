/*
 * mgrt.h -- C ABI of libmgrt.so, the MI355X (gfx950) HIP runtime under the multigrid solver.
 *
 * This is the drop-in boundary: the C++ classes that mirror the reference's public interface
 * (Buffer2D/3D, BorderHandler2D/3D, MultigridSolver2D/3D, the free norm functions) call nothing
 * but the functions below, where the reference called OpenCL (cl::Buffer, cl::Kernel::setArg,
 * cl::CommandQueue::enqueueNDRangeKernel).  Plain pointers and sizes only; every function returns
 * 0 on success, a positive hipError_t value when HIP failed, or a negative MGRT_E_* code when an
 * argument was rejected; mgrt_last_error() describes the last failure of the calling thread.
 * Nothing throws across this boundary.  One host thread per stream, as in the reference (one
 * in-order queue per solver, multigridsolver3D.cpp:36).
 *
 * Fields are stored x fastest: element (x,y,z) of a field lives at  x + pitch*y + slab*z
 * (the reference's dense layout, multi_array.h:36,75, is pitch = nx, slab = nx*ny).  2D fields
 * have nz = 1.  Every kernel exists for float (_f32) and double (_f64), the two settings of the
 * reference's real.cl:2-18.  Kernel calls are asynchronous on the given stream.
 *
 * Numerics contract: every kernel evaluates the reference's expressions in the reference's
 * operation order, with the reference's mixed precision (bare literals are double), without
 * fused multiply-add.  The 3D smoother, whose colouring ignores z (mg_1.cl:98), uses the serial
 * NDRange schedule: within one colour pass each (x,y) column is relaxed in ascending z, so
 * u_new[z] sees u_new[z-1] and u_old[z+1].
 */
#ifndef MGRT_H
#define MGRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int mgrt_status;
typedef void *mgrt_stream; /* a hipStream_t; NULL is the default stream */

enum {
    MGRT_OK = 0,
    MGRT_E_INVALID = -1,     /* bad argument (null pointer, extent, range, alignment) */
    MGRT_E_UNSUPPORTED = -2, /* valid request the library does not implement */
    MGRT_E_NOT_INIT = -3,    /* mgrt_init has not been called */
    MGRT_E_NO_DEVICE = -4,   /* no gfx950 device visible */
    MGRT_E_TRANSPORT = -5    /* RCCL or a transport callback reported a failure */
};

/* cell classes: the values of BorderHandler3D::CellType (multigridsolver3D.h:28), which stand
 * for the NaN-encoded normals of the reference's device Cell (mg_1.cl:5-26, mg_0.cl:5-26) */
enum { MGRT_CELL_INNER = 0, MGRT_CELL_OUTER = 1, MGRT_CELL_DIRICHLET = 2, MGRT_CELL_NEUMANN = 3 };

/*
 * One level's geometry and domain description.  Replaces the (Cell buffer, int4 size) pair every
 * reference kernel receives (mg_1.cl:91-96) and the per-extent cell maps the border handler
 * generates and caches (rectangularborderhandler.cpp:34-40,46-68,77-114).
 *
 *  cells != NULL : device byte map, one MGRT_CELL_* value per point, laid out with the same
 *                  pitch and slab (in bytes) as the fields.  Any domain a BorderHandler can
 *                  describe.  Inner cells must not lie on a face of the global grid.
 *  cells == NULL : analytic box, no memory traffic: the six faces carry face[] =
 *                  {x-lo, x-hi, y-lo, y-hi, z-lo, z-hi} types and every other point is inner.
 *                  On shared edges the reference's fill order decides
 *                  (rectangularborderhandler.cpp:59-62,103-108): 3D y faces over x faces over
 *                  z faces, 2D x faces over y faces.
 *
 *  y0, gny describe a y-slab of a larger grid for multi-GPU runs: the arrays hold rows
 *  [y0, y0+ny) of a grid with gny rows.  Colours, cell classes and coarse/fine index relations
 *  are evaluated with the global row y0 + y.  A whole grid is y0 = 0, gny = ny.
 */
typedef struct mgrt_grid {
    int dim;        /* 2 or 3 */
    int nx, ny, nz; /* extents of the arrays (nz = 1 when dim == 2) */
    int pitch;      /* elements between rows, >= nx */
    long long slab; /* elements between z planes, >= pitch*ny */
    int y0, gny;    /* first global row held, global row count */
    const unsigned char *cells;
    int face[6];
    /* What a Neumann cell is.  0: what the reference's kernels make of it -- a cell no kernel ever
     * writes (`case CELL_NEUMANN: break;`, mg_1.cl:76-77,130-131).  1: an EXTENSION, not reference
     * behaviour and not pinned by any reference result: the first-order condition the reference's
     * set-up prepares data for (functionhandler.cpp:40-42,79-81 stores g*h there) and never applies:
     * u_b = u_i + f_b with i the neighbour one step inside along the face's normal; residual
     * f_b - (u_b - u_i); the restriction hands the coarse cell half of that (the spacing doubles).
     * Analytic boxes only (cells == NULL); the single-colour kernels only. */
    int neumann;
} mgrt_grid;

/* ---- runtime: replaces CLContextLoader (clcontextloader.cpp:29-80) and cl::Buffer / queue use
 * in buffer.h:39-45,61-79 and multigridsolver3D.h:68 ------------------------------------------ */
mgrt_status mgrt_init(int device);                  /* select the device of this process */
mgrt_status mgrt_shutdown(void);
const char *mgrt_last_error(void);
mgrt_status mgrt_device_name(char *buf, int len);   /* clcontextloader.cpp:59 prints the device */
mgrt_status mgrt_alloc(void **ptr, size_t bytes);   /* cl::Buffer ctor, buffer.h:39-41 */
mgrt_status mgrt_free(void *ptr);
mgrt_status mgrt_upload(void *dst, const void *src, size_t bytes, mgrt_stream s); /* COPY_HOST_PTR */
mgrt_status mgrt_download(void *dst, const void *src, size_t bytes, mgrt_stream s); /* blocking, buffer.h:77 */
mgrt_status mgrt_stream_create(mgrt_stream *s);     /* cl::CommandQueue ctor, multigridsolver3D.cpp:36 */
mgrt_status mgrt_stream_destroy(mgrt_stream s);
mgrt_status mgrt_sync(mgrt_stream s);               /* queue.finish(), multigridsolver3D.h:68 */
mgrt_status mgrt_zero(void *ptr, size_t bytes, mgrt_stream s); /* zero_memory, common_kernels.cl:6-10 */

/* Fill a host byte map for a box with the given face types (the generator behind cells == NULL),
 * replaces genBuffer (rectangularborderhandler.cpp:46-68,89-114).  Host memory, dense layout. */
mgrt_status mgrt_box_cells_host(int dim, int nx, int ny, int nz, const int face[6], unsigned char *cells);

/* Whether mgrt_correct_smooth_sweep_f32/_f64 (below) has a kernel for this fine grid, element size
 * (4 or 8) and iteration count: 1 or 0.  For callers that plan their launches ahead. */
int mgrt_correct_smooth_sweep_supported(const mgrt_grid *fine, int elem_size, int iters);
/* The same question for mgrt_smooth_sweep_residual_restrict_f32/_f64. */
int mgrt_smooth_sweep_residual_restrict_supported(const mgrt_grid *fine, int elem_size, int iters);

/* Measurement aid, no counterpart in the reference.  mgrt_smooth_sweep_residual_restrict_f32 in 3D is two
 * launches (the sweep, then the restriction of the coarse rows on its tile seams); a harness that times
 * the sweep kernel alone between its own events selects with `mask` which of them the following calls of
 * this process issue: bit 0 the sweep, bit 1 the seam restriction.  Bit 2 makes the fused 3D sweeps launch a
 * plane with more tiles than the device runs at once as one grid instead of band by band (same values,
 * for comparing the two).  Returns the previous mask; the default, 3, is the only setting that computes
 * the documented result the documented way. */
int mgrt_sweep_parts(int mask);
/* Measurement aid, no counterpart in the reference: the number of launches (bands of tile rows) the most
 * recent fused 3D sweep kernel of this process was issued as; 1 unless the plane has more tiles than the
 * device holds at once.  A harness divides an operation's bytes and time by it to get figures per launch. */
int mgrt_last_sweep_launches(void);

#define MGRT_DECLARE_KERNELS(R, S)                                                                 \
    /* One colour pass of the weighted Gauss-Seidel smoother, in place.  Replaces                 \
       iteration_kernel (mg_1.cl:91-104, mg_0.cl:86-98) as launched by smoother_iterate           \
       (multigridsolver3D.cpp:111-132).  Colour = (x + y + odd) % 2.  Relaxes planes z < zcount   \
       (the reference passes ceil(nz/2), multigridsolver3D.cpp:121) and rows ylo <= y < yhi. */   \
    mgrt_status mgrt_smooth_pass##S(const mgrt_grid *g, R *u, const R *f, R omega, int odd,        \
                                    int zcount, int ylo, int yhi, mgrt_stream s);                  \
    /* `iters` full iterations (colour 0 then colour 1 each) with the same results as 2*iters      \
       calls of mgrt_smooth_pass; free to fuse them.  Replaces the loop of smoother_iterate. */   \
    mgrt_status mgrt_smooth##S(const mgrt_grid *g, R *u, const R *f, R omega, int iters,           \
                               int zcount, mgrt_stream s);                                         \
    /* One fused z-march: out = S^iters(in) for iters in {1,2}, every field element read once and  \
       written once; same results as 2*iters calls of mgrt_smooth_pass on a copy of `in`.          \
       Analytic-box grids only (MGRT_E_UNSUPPORTED otherwise: use the pass kernel).  out must not  \
       alias in.  in == NULL stands for a field of zeros (the cleared coarse unknown of              \
       multigridsolver3D.cpp:59).  Replaces the launch loop of smoother_iterate                      \
       (multigridsolver3D.cpp:124-131). */                                                          \
    mgrt_status mgrt_smooth_sweep##S(const mgrt_grid *g, R *out, const R *in, const R *f, R omega, \
                                     int iters, int zcount, mgrt_stream s);                        \
    /* The same sweep, which also leaves r = residual of `out` (as mgrt_residual with               \
       zero_unwritten) behind: smoother_iterate followed by compute_residuals                      \
       (multigridsolver3D.cpp:50,61, multigridsolver2D.cpp:51,62) in one pass over the fields.     \
       Same restrictions. */                                                                       \
    mgrt_status mgrt_smooth_sweep_residual##S(const mgrt_grid *g, R *out, R *r, const R *in,       \
                                              const R *f, R omega, int iters, int zcount,          \
                                              mgrt_stream s);                                      \
    /* The same sweep, which leaves rc = Restrict(residual of `out`) on the coarse grid behind     \
       without a pass over the fine residual: smoother_iterate, compute_residuals and restrict     \
       (multigridsolver2D.cpp:51,62,71, multigridsolver3D.cpp:50,61,64) in one pass over the       \
       fields.  Whole analytic-box grids with odd extents; in 3D only where                        \
       mgrt_smooth_sweep_residual_restrict_supported says so (MGRT_E_UNSUPPORTED otherwise).       \
       iters == 0 (3D, either element size): no smoothing, out is not used (NULL), in must be      \
       given: rc = Restrict(residual of `in`), compute_residuals and restrict alone, in whichever   \
       form is fastest for the level (one march, one launch through LDS, or the two kernels with   \
       the residual passing through r).                                                            \
       r: 2D: not used, may be NULL.  3D: a work array of the fine grid's size; the sweep forms    \
       every coarse row but those on the seams between its rows of tiles, leaves the fine          \
       residual rows around those seams in r (the rest of r is not written) and a restriction of   \
       the seam rows follows on the same stream.  in == NULL: zeros. */                            \
    mgrt_status mgrt_smooth_sweep_residual_restrict##S(const mgrt_grid *fine,                      \
                                                       const mgrt_grid *coarse, R *out, R *rc,     \
                                                       R *r, const R *in, const R *f, R omega,     \
                                                       int iters, int zcount, mgrt_stream s);      \
    /* out = S^iters(u + 4*P(e)): the results of mgrt_correct followed by mgrt_smooth_sweep, the   \
       corrected field never written.  Replaces correct_residual followed by the post-smoothing    \
       smoother_iterate (multigridsolver2D.cpp:77,92, multigridsolver3D.cpp:70,77).  Analytic-box  \
       grids with odd extents; in 3D only iters == 2, an even y0 and at least 100*100 columns,     \
       with 8-byte elements only planes of more tiles than the device has compute units            \
       (MGRT_E_UNSUPPORTED otherwise: call the two kernels; mgrt_correct_smooth_sweep_supported    \
       answers beforehand).  out must not alias an input. */                                       \
    mgrt_status mgrt_correct_smooth_sweep##S(const mgrt_grid *fine, const mgrt_grid *coarse,       \
                                             R *out, const R *u, const R *e, const R *f, R omega,  \
                                             int iters, int zcount, mgrt_stream s);                \
    /* r = f - A u on inner cells, f - u on Dirichlet cells; outer and Neumann cells of r are     \
       left untouched, or set to 0 when zero_unwritten != 0.  Replaces residual_kernel            \
       (mg_1.cl:111-133, mg_0.cl:105-126).  norm2 (device, may be NULL) receives sum(r*r) over    \
       the cells written, accumulated in double. */                                               \
    mgrt_status mgrt_residual##S(const mgrt_grid *g, R *r, const R *u, const R *f,                 \
                                 int zero_unwritten, double *norm2, int ylo, int yhi,              \
                                 mgrt_stream s);                                                   \
    /* Full-weighting restriction with injection next to border cells.  Replaces                  \
       reduction_kernel (mg_1.cl:146-190, mg_0.cl:139-157).  Coarse rows cylo <= j < cyhi. */     \
    mgrt_status mgrt_restrict##S(const mgrt_grid *fine, const mgrt_grid *coarse, R *rc,            \
                                 const R *rf, int cylo, int cyhi, mgrt_stream s);                  \
    /* out = u + 4*P(e), P bi/trilinear, injection on border cells; out may alias u.  Replaces    \
       residual_correct_kernel (mg_1.cl:200-233, mg_0.cl:167-194). */                             \
    mgrt_status mgrt_correct##S(const mgrt_grid *fine, const mgrt_grid *coarse, R *out,            \
                                const R *u, const R *e, int ylo, int yhi, mgrt_stream s);          \
    /* out = P(c).  Replaces prolongation_kernel (mg_1.cl:236-267, mg_0.cl:197-222). */           \
    mgrt_status mgrt_prolong##S(const mgrt_grid *fine, const mgrt_grid *coarse, R *out,            \
                                const R *c, int ylo, int yhi, mgrt_stream s);                      \
    /* clear outer cells.  Replaces zero_out (mg_1.cl:269-275, mg_0.cl:224-230). */               \
    mgrt_status mgrt_zero_outside##S(const mgrt_grid *g, R *u, mgrt_stream s);                     \
    /* out = a - b over the grid.  Replaces Diff (common_kernels.cl:56-62, auxiliary.h:96-109). */ \
    mgrt_status mgrt_diff##S(const mgrt_grid *g, R *out, const R *a, const R *b, mgrt_stream s);   \
    /* out = in * m over the grid, m a float whatever the element type; out may alias in.          \
       Replaces Mult (common_kernels.cl:65-72), which the reference builds and never launches. */  \
    mgrt_status mgrt_scale##S(const mgrt_grid *g, R *out, const R *in, float m, mgrt_stream s);    \
    /* Exact norms over rows ylo..yhi: sum of squares / max |x| accumulated in double into a      \
       device scalar, deterministic (fixed two-stage reduction, no atomics).  These are the       \
       engine's norms; the reference's lossy tree is below. */                                    \
    mgrt_status mgrt_sumsq##S(const mgrt_grid *g, const R *x, double *out, int ylo, int yhi,       \
                              mgrt_stream s);                                                      \
    mgrt_status mgrt_maxabs##S(const mgrt_grid *g, const R *x, double *out, int ylo, int yhi,      \
                               mgrt_stream s);                                                     \
    /* Blocking norms with the reference's exact behaviour: element-wise square, then the /4      \
       reduction tree that accumulates in R and drops up to 3 tail elements per level             \
       (auxiliary.cpp:119-173, auxiliary.h:72-82, common_kernels.cl:13-53); sqrt on the host.     \
       Dense arrays only (pitch == nx, slab == nx*ny). */                                         \
    mgrt_status mgrt_l2norm_reftree##S(const R *x, long long n, R *result, mgrt_stream s);         \
    mgrt_status mgrt_linfnorm_reftree##S(const R *x, long long n, R *result, mgrt_stream s);        \
    /* The same tree over the plain values: the sum that Average divides (auxiliary.h:84-94). */   \
    mgrt_status mgrt_sum_reftree##S(const R *x, long long n, R *result, mgrt_stream s);

MGRT_DECLARE_KERNELS(float, _f32)
MGRT_DECLARE_KERNELS(double, _f64)

/*
 * mgrt_sumsq_f32 for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments
 * and result as mgrt_sumsq_f32 (the sum may differ in the last bits: the partial sums are grouped
 * differently on large grids), but the block partials live in a fixed 512 KiB array inside the
 * library instead of allocated scratch, and no allocating call is made.  Limit: nx * (yhi - ylo) at
 * most 65536 * 256 = 16 777 216 points per plane (MGRT_E_INVALID beyond).  All calls on one GPU share
 * the array: calls that overlap in time on one GPU (different streams) race; order them.
 */
mgrt_status mgrt_sumsq_noalloc_f32(const mgrt_grid *g, const float *x, double *out, int ylo, int yhi,
                                   mgrt_stream s);

/*
 * Pre-planned V-cycle: the level hierarchy of MultigridSolver{2,3}D::iterate
 * (multigridsolver3D.cpp:41-80: extents n -> (n+1)/2 while every extent > 3) with all work
 * arrays allocated once, where the reference allocates three buffers per level per call
 * (multigridsolver3D.cpp:53-55).  The correction is applied in place, which gives the same
 * values as the reference's out-of-place kernel plus handle swap (multigridsolver3D.cpp:70-72).
 * Residual cells the reference never writes (outer/Neumann) and then restricts from
 * uninitialised memory are defined as 0.
 */
typedef struct mgrt_plan mgrt_plan;

typedef struct mgrt_plan_desc {
    int dim;          /* 2 or 3 */
    int nx, ny, nz;   /* finest extents */
    int elem_size;    /* 4 = float, 8 = double */
    int face[6];      /* box face types, used for every level without an uploaded cell map */
    int zfull;        /* 0: reference z range ceil(nz/2) in the 3D smoother; 1: all planes */
    int use_graph;    /* replay each distinct cycle from a captured hipGraph (ignored on the
                         default stream, which cannot be captured) */
    int no_fuse;      /* 1: smooth with one launch per colour pass (the unfused schedule) */
    int no_bottom;    /* 1: one launch per kernel on the coarsest levels too, instead of running them
                         as a single launch with every level held in LDS */
    int neumann;      /* mgrt_grid.neumann for every level: 0 = the reference's frozen Neumann cells,
                         1 = first-order Neumann condition (extension; implies the unfused schedule
                         when a face is Neumann) */
    int separate_restrict; /* 1: 3D levels write the fine residual and restrict it with a launch of its
                         own even where the pre-smoothing sweep could carry the restriction (same
                         values; for timing comparisons) */
} mgrt_plan_desc;

mgrt_status mgrt_plan_create(mgrt_plan **plan, const mgrt_plan_desc *desc, mgrt_stream s);
mgrt_status mgrt_plan_destroy(mgrt_plan *plan);
int mgrt_plan_levels(const mgrt_plan *plan);
/* extents of a level (0 = finest) */
mgrt_status mgrt_plan_level_dims(const mgrt_plan *plan, int level, int *nx, int *ny, int *nz);
/* install a dense host byte map (values MGRT_CELL_*) for one level; replaces the analytic box */
mgrt_status mgrt_plan_set_cells(mgrt_plan *plan, int level, const unsigned char *host_cells);
/* `ncycles` times iterate(u, f, omega, a1, a2, v) on dense device arrays u (in/out) and f.
 * res_hist (host, may be NULL) receives ncycles+1 values: ||f - A u||_2 before the first cycle
 * and after each one (exact norm, double).  Blocking only when res_hist != NULL. */
mgrt_status mgrt_plan_vcycle(mgrt_plan *plan, void *u, const void *f, double omega, int a1, int a2,
                             int v, int ncycles, double *res_hist);
/* fmg(f, ...) of multigridsolver3D.cpp:82-109 into u_out (dense device array) */
mgrt_status mgrt_plan_fmg(mgrt_plan *plan, void *u_out, const void *f, double omega, int a1,
                          int a2, int v, int iters);

/*
 * The same cycle with the grid cut into y-slabs, one slab per process and GPU.  The reference has one
 * device and one in-order queue (clcontextloader.cpp:57, multigridsolver3D.cpp:36); this extends
 * MultigridSolver3D::iterate (multigridsolver3D.cpp:41-80) to several GPUs with bitwise the results
 * of one: the cut is along y only, so every z column, and the z recurrence of a colour pass with it,
 * stays on one rank.  Rank r owns rows [B_r, B_r+1) of level 0, B_r a multiple of 2^(levels cut), and
 * stores four halo rows of each neighbour; levels of at most `crossover` points are gathered and
 * solved whole on every rank.  3D grids with analytic-box domains only: there is no way to hand a slab
 * plan a byte cell map or a 2D grid (mgrt_slab_plan_create returns MGRT_E_UNSUPPORTED for dim != 3 through
 * its extents check); MappedBorderHandler domains and multigrid2D run on one GPU.
 *
 * How halos travel is the transport's business:
 *   MGRT_TRANSPORT_RCCL      ncclSend/ncclRecv with the two neighbours, ncclAllGather, ncclAllReduce on
 *                            the plan's stream; with use_graph a whole cycle, communication included,
 *                            is one hipGraph per rank.  librccl.so.1 is bound at run time.
 *   MGRT_TRANSPORT_CALLBACK  blocking host functions that move device buffers (the stream is
 *                            synchronised before each call); cycles are launched directly.
 *   MGRT_TRANSPORT_NONE      nothing leaves the device: with nranks > 1 the plan is the share of one
 *                            rank of a larger run with copies of the same size in place of the wire,
 *                            for timing on one GPU (the values are not those of the whole problem).
 */
enum { MGRT_TRANSPORT_NONE = 0, MGRT_TRANSPORT_RCCL = 1, MGRT_TRANSPORT_CALLBACK = 2 };

typedef struct mgrt_transport {
    int kind;                   /* MGRT_TRANSPORT_* */
    int rank, nranks;
    unsigned char rccl_id[128]; /* RCCL: the ncclUniqueId of mgrt_rccl_unique_id() on one rank, the same on all */
    /* CALLBACK: each returns 0 on success.  Pointers are device memory, sizes are bytes. */
    void *ctx;
    /* exchange with n peers at once: send[k] goes to peers[k], recv[k] comes from it */
    int (*sendrecv)(void *ctx, int n, const int *peers, void *const *send, void *const *recv, const size_t *bytes);
    /* recv holds nranks parts of `bytes` each, part r from rank r */
    int (*allgather)(void *ctx, const void *send, void *recv, size_t bytes);
    /* sum of one double over all ranks, in place */
    int (*allreduce_sum)(void *ctx, double *value);
} mgrt_transport;

typedef struct mgrt_slab_desc {
    int nx, ny, nz;      /* extents of the whole finest grid */
    int elem_size;       /* 4 = float, 8 = double */
    int face[6];
    int zfull;           /* as mgrt_plan_desc */
    int use_graph;       /* replay each distinct cycle from a captured hipGraph where the transport allows */
    long long crossover; /* levels with at most this many points are not cut */
} mgrt_slab_desc;

typedef struct mgrt_slab_plan mgrt_slab_plan;

/* a fresh ncclUniqueId (ncclGetUniqueId); one rank makes it, every rank passes it in mgrt_transport */
mgrt_status mgrt_rccl_unique_id(unsigned char id[128]);
/* the version code of the RCCL the library bound (ncclGetVersion: major * 10000 + minor * 100 + patch), for
 * the record a benchmark keeps; MGRT_E_UNSUPPORTED where RCCL is absent.  No counterpart in the reference. */
mgrt_status mgrt_rccl_version(int *version);
/* the RCCL operations a slab plan uses, on a communicator of one rank (so that a single GPU can run it):
 * send/receive with itself, all-gather and all-reduce, queued directly and replayed from a captured graph,
 * results checked.  MGRT_E_UNSUPPORTED where RCCL is absent, MGRT_E_TRANSPORT with the finding otherwise.
 * No counterpart in the reference. */
mgrt_status mgrt_rccl_selftest(mgrt_stream s);
/* collective over the transport's ranks; `s` must be a stream of its own (not NULL) */
mgrt_status mgrt_slab_plan_create(mgrt_slab_plan **plan, const mgrt_slab_desc *desc, const mgrt_transport *transport,
                                  mgrt_stream s);
mgrt_status mgrt_slab_plan_destroy(mgrt_slab_plan *plan);
/* number of levels cut into slabs (0: the grid is too small, or one rank: every rank holds all of it) */
int mgrt_slab_plan_distributed_levels(const mgrt_slab_plan *plan);
/* 1 while cycles are replayed from a graph (0 after a refused capture, or by choice of transport) */
int mgrt_slab_plan_uses_graph(const mgrt_slab_plan *plan);
/* this rank stores global rows [lo, hi) of level 0 and owns [own_lo, own_hi) */
mgrt_status mgrt_slab_plan_rows(const mgrt_slab_plan *plan, int *lo, int *hi, int *own_lo, int *own_hi);
/* `ncycles` times iterate(u, f, omega, a1, a2, v) on this rank's rows: u (in/out) and f are dense
 * device arrays of nz x (hi - lo) x nx elements, halo rows included; those of f must be filled, those
 * of u are exchanged here before they are read.  On return only the owned rows [own_lo, own_hi) of u are
 * defined: the halo rows hold whatever the last sweep left there, not the neighbours' values.
 * res_hist as in mgrt_plan_vcycle, norms over the whole grid. */
mgrt_status mgrt_slab_plan_vcycle(mgrt_slab_plan *plan, void *u, const void *f, double omega, int a1, int a2, int v,
                                  int ncycles, double *res_hist);
/* fmg(f, omega, a1, a2, v, iters) of multigridsolver3D.cpp:82-109 on this rank's rows: u_out is written (its
 * owned rows are the result; the halo rows are overwritten with values that are not the neighbours'), f as
 * for mgrt_slab_plan_vcycle.  Coarse levels run the default of 3 cycles as in the reference. */
mgrt_status mgrt_slab_plan_fmg(mgrt_slab_plan *plan, void *u_out, const void *f, double omega, int a1, int a2, int v,
                               int iters);

#ifdef __cplusplus
}
#endif
#endif /* MGRT_H */
