// buffer.h -- device-resident grid fields.
// Buffer2D / Buffer3D keep the interface of buffer.h:29-99 and :102-179: shallow copies share the
// device allocation, isInitialized(), static empty(), blocking read(), width/height/depth, size(),
// data(), operator()().  The allocation is a dense array, x fastest, `real` elements.
#ifndef BUFFER_H
#define BUFFER_H

#include "mgrt_host.h"
#include "multi_array.h"

namespace mgrt { namespace detail {

struct Int2 { int s[2]; };
struct Int4 { int s[4]; };

// What both buffer shapes are made of.
class Field
{
public:
    // CL_MEM_* access hints of the reference; device memory here is always read-write.
    enum WriteMode { WriteOnly = 2, ReadOnly = 4, ReadWrite = 1 };

    bool isInitialized() const { return m_w != 0 && m_h != 0 && m_d != 0; }
    int width() const { return m_w; }
    int height() const { return m_h; }
    size_t count() const { return (size_t)m_w * m_h * m_d; }
    // the device pointer, where the reference hands out its cl_mem
    real *operator()() const { return static_cast<real *>(m_mem.get()); }
    const std::shared_ptr<void> &data() const { return m_mem; }

protected:
    Field() : m_w(0), m_h(0), m_d(0) {}
    Field(int w, int h, int d);
    Field(int w, int h, int d, const real *host);
    void clear(const Stream &q) const;
    void download(real *host, const Stream &q) const;
    int m_w, m_h, m_d;
    std::shared_ptr<void> m_mem;
};
}} // namespace mgrt::detail

class Buffer2D : public mgrt::detail::Field
{
public:
    Buffer2D() {}
    Buffer2D(int w, int h, WriteMode = ReadWrite) : Field(w, h, 1) {}
    Buffer2D(int w, int h, real *f, WriteMode = ReadWrite) : Field(w, h, 1, f) {}

    static Buffer2D empty(int w, int h, mgrt::Stream &q)
    {
        Buffer2D b(w, h);
        b.clear(q);
        return b;
    }
    BidimArray<real> read(mgrt::Stream &q) const
    {
        BidimArray<real> a(m_w, m_h, mgrt::detail::Uninitialized());
        download(a.data(), q);
        return a;
    }
    mgrt::detail::Int2 size() const
    {
        mgrt::detail::Int2 r = {{m_w, m_h}};
        return r;
    }
};

class Buffer3D : public mgrt::detail::Field
{
public:
    Buffer3D() {}
    Buffer3D(int w, int h, int d, WriteMode = ReadWrite) : Field(w, h, d) {}
    Buffer3D(int w, int h, int d, real *f, WriteMode = ReadWrite) : Field(w, h, d, f) {}

    int depth() const { return m_d; }
    static Buffer3D empty(int w, int h, int d, mgrt::Stream &q)
    {
        Buffer3D b(w, h, d);
        b.clear(q);
        return b;
    }
    TridimArray<real> read(mgrt::Stream &q) const
    {
        TridimArray<real> a(m_w, m_h, m_d, mgrt::detail::Uninitialized());
        download(a.data(), q);
        return a;
    }
    mgrt::detail::Int4 size() const
    {
        mgrt::detail::Int4 r = {{m_w, m_h, m_d, 1}};
        return r;
    }
};

#endif // BUFFER_H
