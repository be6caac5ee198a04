// mgrt_host.cpp -- device context, stream, allocations and the Buffer plumbing.
#include "buffer.h"
#include <cstdlib>
#include <iostream>

namespace mgrt {

void check(mgrt_status st, const char *where)
{
    if (st == MGRT_OK)
        return;
    const char *why = mgrt_last_error();
    throw Error(std::string(where) + ": " + (why && *why ? why : "failed"), st);
}

Context::Context() : m_device(0)
{
    if (const char *e = std::getenv("MGRT_DEVICE"))
        m_device = std::atoi(e);
    const mgrt_status st = mgrt_init(m_device);
    if (st == MGRT_E_NO_DEVICE)
        throw std::runtime_error("No devices found"); // clcontextloader.cpp:56
    check(st, "mgrt_init");
    char name[256] = "";
    MGRT_CHECK(mgrt_device_name(name, (int)sizeof name));
    m_name = name;
    std::cout << "Picking device: " << m_name << std::endl;
}

Context::~Context() { (void)mgrt_shutdown(); }

Context &Context::instance()
{
    static Context ctx;
    return ctx;
}

Stream::Handle::~Handle()
{
    if (h)
        (void)mgrt_stream_destroy(h);
}

Stream::Stream() : m_s(new Handle())
{
    Context::instance();
    m_s->h = 0;
    MGRT_CHECK(mgrt_stream_create(&m_s->h));
}

void Stream::finish() const { MGRT_CHECK(mgrt_sync(get())); }

std::shared_ptr<void> deviceAlloc(size_t bytes)
{
    Context::instance();
    void *p = 0;
    MGRT_CHECK(mgrt_alloc(&p, bytes ? bytes : 1));
    return std::shared_ptr<void>(p, [](void *q) { (void)mgrt_free(q); });
}

namespace detail {

Field::Field(int w, int h, int d) : m_w(w), m_h(h), m_d(d), m_mem(deviceAlloc(sizeof(real) * count())) {}

Field::Field(int w, int h, int d, const real *host) : m_w(w), m_h(h), m_d(d), m_mem(deviceAlloc(sizeof(real) * count()))
{
    // CL_MEM_COPY_HOST_PTR (buffer.h:43-45): the host array may be released when this returns
    MGRT_CHECK(mgrt_upload(m_mem.get(), host, sizeof(real) * count(), 0));
    MGRT_CHECK(mgrt_sync(0));
}

void Field::clear(const Stream &q) const { MGRT_CHECK(mgrt_zero(m_mem.get(), sizeof(real) * count(), q.get())); }

void Field::download(real *host, const Stream &q) const
{
    MGRT_CHECK(mgrt_download(host, m_mem.get(), sizeof(real) * count(), q.get()));
}

} // namespace detail
} // namespace mgrt
