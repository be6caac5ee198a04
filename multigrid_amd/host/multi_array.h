// multi_array.h -- dense host arrays with x fastest, (x,y[,z]) indexing.
// Same interface as BidimArray / TridimArray (multi_array.h:26-98): data(), operator(), width(),
// height(), depth(), value semantics, swap.
#ifndef MULTI_ARRAY_H
#define MULTI_ARRAY_H

#include <cstddef>
#include <utility>
#include <vector>

namespace mgrt { namespace detail {
// storage + extents shared by the two array shapes
template <class T, int DIM> class HostArray
{
public:
    HostArray(int w, int h, int d) : m_w(w), m_h(h), m_d(d), m_v((size_t)w * h * d) {}
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
    std::vector<T> m_v;
};
}} // namespace mgrt::detail

template <class T> class BidimArray : public mgrt::detail::HostArray<T, 2>
{
    typedef mgrt::detail::HostArray<T, 2> Base;

public:
    BidimArray(int w, int h) : Base(w, h, 1) {}
    T &operator()(int x, int y) { return this->m_v[this->at(x, y, 0)]; }
    const T &operator()(int x, int y) const { return this->m_v[this->at(x, y, 0)]; }
};

template <class T> class TridimArray : public mgrt::detail::HostArray<T, 3>
{
    typedef mgrt::detail::HostArray<T, 3> Base;

public:
    TridimArray(int w, int h, int d) : Base(w, h, d) {}
    int depth() const { return this->m_d; }
    T &operator()(int x, int y, int z) { return this->m_v[this->at(x, y, z)]; }
    const T &operator()(int x, int y, int z) const { return this->m_v[this->at(x, y, z)]; }
};

namespace std {
template <class T> void swap(BidimArray<T> &a, BidimArray<T> &b) { a.swap(b); }
template <class T> void swap(TridimArray<T> &a, TridimArray<T> &b) { a.swap(b); }
}

#endif // MULTI_ARRAY_H
