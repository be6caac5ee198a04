// mgrt_host.h -- what the host classes share: the `real` type, the error type, the in-order
// stream that stands where the reference has a cl::CommandQueue, the argument block that stands
// where it has a cl::Kernel, and the process-wide device context.
//
// Takes the place of clcontextloader.h (`real`: clcontextloader.h:29-45, the singleton context:
// clcontextloader.h:47-83, clcontextloader.cpp:29-80).  Everything below talks to the GPU only
// through the C entry points of mgrt.h.
#ifndef MGRT_HOST_H
#define MGRT_HOST_H

#include "mgrt.h"
#include <memory>
#include <stdexcept>
#include <string>

#ifdef USE_DOUBLE
typedef double real;
#define MGRT_K(name) mgrt_##name##_f64
#else
typedef float real;
#define MGRT_K(name) mgrt_##name##_f32
#endif

namespace mgrt {

// Thrown for every failing runtime call, as cl::Error is in the reference
// (__CL_ENABLE_EXCEPTIONS, clcontextloader.h:24): what() names the call and the cause, err() is the
// status code.
class Error : public std::runtime_error
{
public:
    Error(const std::string &what, int code) : std::runtime_error(what), m_code(code) {}
    int err() const { return m_code; }

private:
    int m_code;
};

// Turn a status into an exception.  `where` is the failing call as written.
void check(mgrt_status st, const char *where);
#define MGRT_CHECK(call) ::mgrt::check((call), #call)

// The device context, created on first use like CLContextLoader::instance().  Device ordinal from
// the environment variable MGRT_DEVICE (default 0).  Prints "Picking device: <name>"
// (clcontextloader.cpp:59).
class Context
{
public:
    static Context &instance();
    const std::string &deviceName() const { return m_name; }
    int device() const { return m_device; }

private:
    Context();
    ~Context();
    Context(const Context &);
    int m_device;
    std::string m_name;
};

// An in-order queue of GPU work.  Copies share the queue, as cl::CommandQueue copies do.
class Stream
{
public:
    Stream();
    mgrt_stream get() const { return m_s ? m_s->h : 0; }
    void finish() const;
    void enqueueBarrier() const {} // in-order already (multigridsolver3D.cpp:103)

private:
    struct Handle {
        mgrt_stream h;
        ~Handle();
    };
    std::shared_ptr<Handle> m_s;
};

// Extents of a launch range in up to three dimensions: stands where the reference's interface names
// cl::NDRange (auxiliary.h: getBestWorkspaceDim).  Converts to a pointer to its sizes, as that does.
class NDRange
{
public:
    NDRange() : m_dims(0) { m_size[0] = m_size[1] = m_size[2] = 0; }
    NDRange(size_t s0) : m_dims(1) { m_size[0] = s0, m_size[1] = m_size[2] = 0; }
    NDRange(size_t s0, size_t s1) : m_dims(2) { m_size[0] = s0, m_size[1] = s1, m_size[2] = 0; }
    NDRange(size_t s0, size_t s1, size_t s2) : m_dims(3) { m_size[0] = s0, m_size[1] = s1, m_size[2] = s2; }
    size_t dimensions() const { return m_dims; }
    operator const size_t *() const { return m_size; }

private:
    size_t m_size[3];
    size_t m_dims;
};
static const NDRange NullRange;

// Where a BorderHandler binds its cell description for one launch: the handler fills in the box
// face classes or the device pointer of a byte map (one MGRT_CELL_* value per grid point).
struct KernelArgs {
    mgrt_grid grid;
    const unsigned char *host_cells; // host copy of grid.cells, or null
};

// Reference-counted device allocation shared by Buffer copies (cl::Buffer semantics, buffer.h:47-57).
std::shared_ptr<void> deviceAlloc(size_t bytes);

} // namespace mgrt

#endif // MGRT_HOST_H
