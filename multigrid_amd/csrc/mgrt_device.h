// mgrt_device.h -- device-side helpers shared by the kernel files: neighbour-lane reads, plane-relative
// buffer addressing, loads and stores kept out of the compiler's wait-count bookkeeping.
#ifndef MGRT_DEVICE_H
#define MGRT_DEVICE_H

#include <hip/hip_runtime.h>

namespace mgrt {

// Neighbour-lane reads as DPP wave shifts: no memory, no `old` operand to initialise (lanes without
// a source lane, 0 and 63, read 0; their columns are halo).
__device__ __forceinline__ int dpp_lower(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); } // wave_shr:1
__device__ __forceinline__ int dpp_upper(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true); } // wave_shl:1
__device__ __forceinline__ float from_lower_lane(float v) { return __int_as_float(dpp_lower(__float_as_int(v))); }
__device__ __forceinline__ float from_upper_lane(float v) { return __int_as_float(dpp_upper(__float_as_int(v))); }
__device__ __forceinline__ double from_lower_lane(double v)
{
    const long long b = __double_as_longlong(v);
    return __longlong_as_double(((long long)dpp_lower((int)(b >> 32)) << 32) | (unsigned)dpp_lower((int)b));
}
__device__ __forceinline__ double from_upper_lane(double v)
{
    const long long b = __double_as_longlong(v);
    return __longlong_as_double(((long long)dpp_upper((int)(b >> 32)) << 32) | (unsigned)dpp_upper((int)b));
}

// Plane-relative buffer addressing: a descriptor per z plane (base = array + slab*z, all scalar
// registers) plus a per-lane byte offset that never changes during the march.  Offsets at or past
// the plane size are dropped by the hardware range check: loads return 0, stores do nothing, which
// replaces every bounds branch.
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t plane_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void buf_load(float &v, rsrc_t r, unsigned off)
{
    v = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void buf_load(double &v, rsrc_t r, unsigned off)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    v = __longlong_as_double(((long long)t.y << 32) | t.x);
}
// Stores are issued from inline asm so that the compiler's s_waitcnt bookkeeping sees loads only.
// With loads and stores both outstanding it assumes they may complete out of order and waits with
// vmcnt(0) before the first use of any prefetched plane, which drains the whole prefetch ring once
// per unrolled group.  With the stores hidden it emits counted waits; the hardware counts the
// hidden stores as well, so those waits are merely stricter than needed.  Nothing ever reads
// `out` in this kernel, so the stores themselves need no wait.  The descriptor words come from
// scalar ALU results, hence the wait states opening the string.
// a*b + c with one rounding, in the type of the operands (the generic builtin would promote floats to double)
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }

typedef int desc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ desc_t plane_desc(const void *base, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)base;
    desc_t d;
    d.x = (int)(unsigned)a;
    d.y = (int)((unsigned)(a >> 32) & 0xffffu);
    d.z = (int)bytes;
    d.w = 0x00020000;
    return d;
}
// Prefetch loads are hidden from the compiler as well: its own accounting across the unrolled
// loop's back edge waits for all but the last plane requested, i.e. it keeps one plane in flight
// instead of PF.  The loads below are followed, right before the first use of a prefetch slot, by
// wait_slot<N>(), a counted s_waitcnt that names every register of the slot as read-write so that no
// use can be scheduled above it.  N is the number of vector-memory operations (loads and hidden
// stores alike: the hardware retires them in issue order) issued after the slot's loads.
__device__ __forceinline__ void buf_load_hidden(float &v, desc_t r, unsigned off)
{
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(r) : "memory");
}
__device__ __forceinline__ void buf_load_hidden(double &v, desc_t r, unsigned off)
{
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(r) : "memory");
}
// one byte per lane, zero-extended (cell classes)
__device__ __forceinline__ void buf_load_hidden_u8(int &v, desc_t r, unsigned off)
{
    asm volatile("buffer_load_ubyte %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(r) : "memory");
}
template <int N, class R> __device__ __forceinline__ void wait_slot2(R &a, R &b, R &c, R &d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}
template <int N, class R, int SR> __device__ __forceinline__ void wait_slot(R (&u)[SR], R (&f)[SR])
{
    static_assert(SR % 2 == 0 && N >= 0 && N < 64, "vmcnt is a 6-bit counter");
#pragma unroll
    for (int r = 0; r < SR; r += 2)
        wait_slot2<N>(u[r], u[r + 1], f[r], f[r + 1]);
}

// (No padding in front of these: the one hazard of a vector-memory instruction on this part, a descriptor register
// written by a vector-ALU instruction fewer than five wait states earlier, is checked on the generated code by
// tools/check_hidden_loads.py; five idle cycles before every store cost 2.7 % of the 513^3 sweep with restriction.)
__device__ __forceinline__ void buf_store_hidden(float v, desc_t r, unsigned off)
{
    asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(v), "v"(off), "s"(r) : "memory");
}
__device__ __forceinline__ void buf_store_hidden(double v, desc_t r, unsigned off)
{
    asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" ::"v"(v), "v"(off), "s"(r) : "memory");
}


// A plain global store the compiler does not count (same purpose as buf_store_hidden).
__device__ __forceinline__ void store_hidden(float *p, float v)
{
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_hidden(double *p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

} // namespace mgrt

#endif
