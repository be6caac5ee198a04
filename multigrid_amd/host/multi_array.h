// multi_array.h -- dense host arrays with x fastest, (x,y[,z]) indexing.
// Same interface as BidimArray / TridimArray (multi_array.h:26-98): data(), operator(), width(),
// height(), depth(), value semantics, swap.
#ifndef MULTI_ARRAY_H
#define MULTI_ARRAY_H

#include <cstddef>
#include <cstdlib>
#include <memory>
#include <new>
#include <sys/mman.h>
#include <utility>
#include <vector>

namespace mgrt { namespace detail {
// std::allocator whose value-less construct() leaves trivially constructible elements alone, so that
// a vector sized for a device read-back is not cleared first (a pass over memory that is about
// to be overwritten: at 513^3 it costs as much as the transfer).
template <class T> struct NoInitAllocator : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAllocator<U> other; };
    NoInitAllocator() {}
    template <class U> NoInitAllocator(const NoInitAllocator<U> &) {}
    // Large arrays sit on 2 MB boundaries and ask for transparent huge pages: a device read-back into
    // fresh memory is otherwise bound by one page fault per 4 KB.
    T *allocate(std::size_t n)
    {
        const std::size_t bytes = n * sizeof(T), huge = (std::size_t)2 << 20;
        if (bytes < 2 * huge)
            return std::allocator<T>::allocate(n);
        void *p = 0;
        if (posix_memalign(&p, huge, (bytes + huge - 1) / huge * huge) != 0)
            throw std::bad_alloc();
        madvise(p, bytes, MADV_HUGEPAGE);
        return (T *)p;
    }
    void deallocate(T *p, std::size_t n)
    {
        if (n * sizeof(T) < ((std::size_t)4 << 20))
            std::allocator<T>::deallocate(p, n);
        else
            free(p);
    }
    template <class U> void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class A> void construct(U *p, const A &a) { ::new ((void *)p) U(a); }
};
struct Uninitialized {};

// storage + extents shared by the two array shapes
template <class T, int DIM> class HostArray
{
public:
    // elements start at T(), as the reference's arrays do
    HostArray(int w, int h, int d) : m_w(w), m_h(h), m_d(d), m_v((size_t)w * h * d, T()) {}
    // elements start undefined: for arrays that are filled whole right away
    HostArray(int w, int h, int d, Uninitialized) : m_w(w), m_h(h), m_d(d), m_v((size_t)w * h * d) {}
    T *data() { return m_v.data(); }
    const T *data() const { return m_v.data(); }
    int width() const { return m_w; }
    int height() const { return m_h; }
    void swap(HostArray &o)
    {
        std::swap(m_w, o.m_w);
        std::swap(m_h, o.m_h);
        std::swap(m_d, o.m_d);
        m_v.swap(o.m_v);
    }

protected:
    size_t at(int x, int y, int z) const { return x + (size_t)m_w * (y + (size_t)m_h * z); }
    int m_w, m_h, m_d;
    std::vector<T, NoInitAllocator<T> > m_v;
};
}} // namespace mgrt::detail

template <class T> class BidimArray : public mgrt::detail::HostArray<T, 2>
{
    typedef mgrt::detail::HostArray<T, 2> Base;

public:
    BidimArray(int w, int h) : Base(w, h, 1) {}
    BidimArray(int w, int h, mgrt::detail::Uninitialized u) : Base(w, h, 1, u) {}
    T &operator()(int x, int y) { return this->m_v[this->at(x, y, 0)]; }
    const T &operator()(int x, int y) const { return this->m_v[this->at(x, y, 0)]; }
};

template <class T> class TridimArray : public mgrt::detail::HostArray<T, 3>
{
    typedef mgrt::detail::HostArray<T, 3> Base;

public:
    TridimArray(int w, int h, int d) : Base(w, h, d) {}
    TridimArray(int w, int h, int d, mgrt::detail::Uninitialized u) : Base(w, h, d, u) {}
    int depth() const { return this->m_d; }
    T &operator()(int x, int y, int z) { return this->m_v[this->at(x, y, z)]; }
    const T &operator()(int x, int y, int z) const { return this->m_v[this->at(x, y, z)]; }
};

namespace std {
template <class T> void swap(BidimArray<T> &a, BidimArray<T> &b) { a.swap(b); }
template <class T> void swap(TridimArray<T> &a, TridimArray<T> &b) { a.swap(b); }
}

#endif // MULTI_ARRAY_H
