// mgrt_runtime.hip -- device selection, memory, streams, error strings.
//
// Stands where the reference has its OpenCL context singleton and buffer class
// (clcontextloader.cpp:29-80, buffer.h:29-179): one device per process, explicit streams,
// status codes instead of exceptions.
#include "mgrt_internal.h"
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace mgrt {

static thread_local char g_err[512] = "";

mgrt_status fail(mgrt_status code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

Runtime &runtime()
{
    static Runtime rt;
    return rt;
}

namespace {
std::mutex g_partials_mutex;
std::unordered_map<hipStream_t, double *> g_partials;
} // namespace

mgrt_status partials_for(hipStream_t s, double **partials)
{
    std::lock_guard<std::mutex> lock(g_partials_mutex);
    auto it = g_partials.find(s);
    if (it == g_partials.end()) {
        double *p = nullptr;
        MGRT_HIP(hipMalloc((void **)&p, sizeof(double) * kMaxPartials));
        it = g_partials.emplace(s, p).first;
    }
    *partials = it->second;
    return MGRT_OK;
}

void release_partials(hipStream_t s)
{
    std::lock_guard<std::mutex> lock(g_partials_mutex);
    auto it = g_partials.find(s);
    if (it != g_partials.end()) {
        (void)hipFree(it->second);
        g_partials.erase(it);
    }
}

void release_all_partials()
{
    std::lock_guard<std::mutex> lock(g_partials_mutex);
    for (auto &kv : g_partials)
        (void)hipFree(kv.second);
    g_partials.clear();
}

mgrt_status require_init()
{
    if (!runtime().ready)
        return fail(MGRT_E_NOT_INIT, "mgrt_init has not been called");
    return MGRT_OK;
}

mgrt_status check_grid(const mgrt_grid *g, const char *who)
{
    MGRT_REQUIRE(g != nullptr, "%s: grid is null", who);
    MGRT_REQUIRE(g->dim == 2 || g->dim == 3, "%s: dim must be 2 or 3, got %d", who, g->dim);
    MGRT_REQUIRE(g->nx > 0 && g->ny > 0 && g->nz > 0, "%s: extents must be positive (%d,%d,%d)", who,
                 g->nx, g->ny, g->nz);
    MGRT_REQUIRE(g->dim == 3 || g->nz == 1, "%s: a 2D grid must have nz == 1", who);
    MGRT_REQUIRE(g->pitch >= g->nx, "%s: pitch %d < nx %d", who, g->pitch, g->nx);
    MGRT_REQUIRE(g->slab >= (long long)g->pitch * g->ny, "%s: slab %lld < pitch*ny", who, g->slab);
    MGRT_REQUIRE(g->y0 >= 0 && g->y0 + g->ny <= g->gny, "%s: rows [%d,%d) outside the global grid of %d rows",
                 who, g->y0, g->y0 + g->ny, g->gny);
    MGRT_REQUIRE(g->slab * (long long)g->nz < (1LL << 40), "%s: grid too large", who);
    MGRT_REQUIRE(g->neumann == 0 || g->neumann == 1, "%s: neumann must be 0 or 1, got %d", who, g->neumann);
    if (g->neumann && g->cells)
        return fail(MGRT_E_UNSUPPORTED, "%s: the first-order Neumann condition needs the analytic box (a byte map carries "
                                        "no normals)", who);
    if (!g->cells)
        for (int i = 0; i < 6; ++i)
            // an inner cell on a face would reach outside the arrays with its stencil
            MGRT_REQUIRE(g->face[i] >= 1 && g->face[i] <= 3,
                         "%s: face[%d] = %d is not a class a face can have (outer, Dirichlet, Neumann)", who, i,
                         g->face[i]);
    return MGRT_OK;
}

} // namespace mgrt

using namespace mgrt;

extern "C" {

const char *mgrt_last_error(void) { return g_err; }

mgrt_status mgrt_init(int device)
{
    Runtime &rt = runtime();
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(MGRT_E_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    MGRT_REQUIRE(device >= 0 && device < count, "mgrt_init: device %d out of range [0,%d)", device, count);
    if (rt.ready && rt.device == device)
        return MGRT_OK;
    MGRT_REQUIRE(!rt.ready, "mgrt_init: already initialised on device %d", rt.device);
    MGRT_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MGRT_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MGRT_E_NO_DEVICE, "device %d is %s; this library holds gfx950 code only", device,
                    prop.gcnArchName);
    rt.num_cu = prop.multiProcessorCount;
    if (mgrt_status st = static_partials_setup(&rt.static_partials))
        return st;
    rt.device = device;
    rt.ready = true;
    return MGRT_OK;
}

mgrt_status mgrt_shutdown(void)
{
    Runtime &rt = runtime();
    if (!rt.ready)
        return MGRT_OK;
    MGRT_HIP(hipDeviceSynchronize());
    release_all_partials();
    rt = Runtime();
    return MGRT_OK;
}

int mgrt_sweep_parts(int mask)
{
    Runtime &rt = runtime();
    const int before = rt.sweep_parts | (rt.sweep_bands ? 0 : 4);
    rt.sweep_parts = mask & 3;
    rt.sweep_bands = !(mask & 4);
    return before;
}

int mgrt_last_sweep_launches(void) { return runtime().last_sweep_launches; }

mgrt_status mgrt_device_name(char *buf, int len)
{
    if (mgrt_status st = require_init())
        return st;
    MGRT_REQUIRE(buf && len > 0, "mgrt_device_name: bad buffer");
    hipDeviceProp_t prop;
    MGRT_HIP(hipGetDeviceProperties(&prop, runtime().device));
    snprintf(buf, (size_t)len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return MGRT_OK;
}

mgrt_status mgrt_alloc(void **ptr, size_t bytes)
{
    if (mgrt_status st = require_init())
        return st;
    MGRT_REQUIRE(ptr != nullptr, "mgrt_alloc: null result pointer");
    *ptr = nullptr;
    if (bytes == 0)
        return MGRT_OK;
    MGRT_HIP(hipMalloc(ptr, bytes));
    return MGRT_OK;
}

mgrt_status mgrt_free(void *ptr)
{
    if (ptr)
        MGRT_HIP(hipFree(ptr));
    return MGRT_OK;
}

mgrt_status mgrt_upload(void *dst, const void *src, size_t bytes, mgrt_stream s)
{
    if (mgrt_status st = require_init())
        return st;
    MGRT_REQUIRE(bytes == 0 || (dst && src), "mgrt_upload: null pointer");
    if (bytes == 0)
        return MGRT_OK;
    // the source is pageable host memory owned by the caller: complete before returning, as
    // CL_MEM_COPY_HOST_PTR does
    MGRT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    MGRT_HIP(hipStreamSynchronize((hipStream_t)s));
    return MGRT_OK;
}

mgrt_status mgrt_download(void *dst, const void *src, size_t bytes, mgrt_stream s)
{
    if (mgrt_status st = require_init())
        return st;
    MGRT_REQUIRE(bytes == 0 || (dst && src), "mgrt_download: null pointer");
    if (bytes == 0)
        return MGRT_OK;
    MGRT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    MGRT_HIP(hipStreamSynchronize((hipStream_t)s));
    return MGRT_OK;
}

mgrt_status mgrt_stream_create(mgrt_stream *s)
{
    if (mgrt_status st = require_init())
        return st;
    MGRT_REQUIRE(s != nullptr, "mgrt_stream_create: null result pointer");
    hipStream_t hs;
    MGRT_HIP(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
    *s = (mgrt_stream)hs;
    return MGRT_OK;
}

mgrt_status mgrt_stream_destroy(mgrt_stream s)
{
    if (s) {
        MGRT_HIP(hipStreamSynchronize((hipStream_t)s));
        release_partials((hipStream_t)s);
        MGRT_HIP(hipStreamDestroy((hipStream_t)s));
    }
    return MGRT_OK;
}

mgrt_status mgrt_sync(mgrt_stream s)
{
    if (mgrt_status st = require_init())
        return st;
    MGRT_HIP(hipStreamSynchronize((hipStream_t)s));
    return MGRT_OK;
}

mgrt_status mgrt_zero(void *ptr, size_t bytes, mgrt_stream s)
{
    if (mgrt_status st = require_init())
        return st;
    MGRT_REQUIRE(bytes == 0 || ptr, "mgrt_zero: null pointer");
    if (bytes)
        MGRT_HIP(hipMemsetAsync(ptr, 0, bytes, (hipStream_t)s));
    return MGRT_OK;
}

mgrt_status mgrt_box_cells_host(int dim, int nx, int ny, int nz, const int face[6], unsigned char *cells)
{
    MGRT_REQUIRE(dim == 2 || dim == 3, "mgrt_box_cells_host: dim must be 2 or 3");
    if (dim == 2)
        nz = 1;
    MGRT_REQUIRE(nx > 0 && ny > 0 && nz > 0 && face && cells, "mgrt_box_cells_host: bad argument");
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                int t = MGRT_CELL_INNER;
                if (dim == 3) { // y faces over x faces over z faces
                    if (y == 0) t = face[2];
                    else if (y == ny - 1) t = face[3];
                    else if (x == 0) t = face[0];
                    else if (x == nx - 1) t = face[1];
                    else if (z == 0) t = face[4];
                    else if (z == nz - 1) t = face[5];
                } else { // x faces over y faces
                    if (x == 0) t = face[0];
                    else if (x == nx - 1) t = face[1];
                    else if (y == 0) t = face[2];
                    else if (y == ny - 1) t = face[3];
                }
                cells[x + (size_t)nx * y + (size_t)nx * ny * z] = (unsigned char)t;
            }
    return MGRT_OK;
}

} // extern "C"
