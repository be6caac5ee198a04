// Streaming micro-benchmark: what do dword / dwordx2 / dwordx4 accesses reach on this chip, aligned
// and misaligned, for read-only, copy and 2-read-1-write patterns?  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class V> __global__ void k_copy(const V *__restrict__ a, V *__restrict__ o, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) o[i] = a[i];
}
template <class V> __global__ void k_read(const V *__restrict__ a, float *o, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0;
    for (; i < n; i += stride) {
        V v = a[i];
        const float *p = (const float *)&v;
        for (unsigned k = 0; k < sizeof(V) / 4; ++k) acc += p[k];
    }
    if (acc == 12345.678f) o[0] = acc;
}
template <class V> __global__ void k_triad(const V *__restrict__ a, const V *__restrict__ b, V *__restrict__ o, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        V x = a[i], y = b[i];
        float *p = (float *)&x;
        const float *q = (const float *)&y;
        for (unsigned k = 0; k < sizeof(V) / 4; ++k) p[k] -= q[k];
        o[i] = x;
    }
}

template <class F> float time_it(F f, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e-3f;
}

int main()
{
    const size_t N = 513ull * 513 * 513; // floats
    float *a, *b, *o;
    CK(hipMalloc(&a, (N + 64) * 4)); CK(hipMalloc(&b, (N + 64) * 4)); CK(hipMalloc(&o, (N + 64) * 4));
    CK(hipMemset(a, 0, (N + 64) * 4)); CK(hipMemset(b, 0, (N + 64) * 4));
    const int reps = 10;
    for (int grid : {2048, 8192, 65536}) {
        for (int mis = 0; mis < 2; ++mis) {
            const float *A = a + mis, *B = b + mis; float *O = o + mis;
            const size_t n4 = N / 4 - 1, n2 = N / 2 - 1;
            float t;
            t = time_it([&] { k_read<float><<<grid, 256>>>(A, o, N); }, reps);
            printf("grid %6d mis %d read  x1 %7.1f us %6.2f TB/s\n", grid, mis, t * 1e6, N * 4 / t / 1e12);
            t = time_it([&] { k_read<float2><<<grid, 256>>>((const float2 *)A, o, n2); }, reps);
            printf("grid %6d mis %d read  x2 %7.1f us %6.2f TB/s\n", grid, mis, t * 1e6, N * 4 / t / 1e12);
            t = time_it([&] { k_read<float4><<<grid, 256>>>((const float4 *)A, o, n4); }, reps);
            printf("grid %6d mis %d read  x4 %7.1f us %6.2f TB/s\n", grid, mis, t * 1e6, N * 4 / t / 1e12);
            t = time_it([&] { k_copy<float><<<grid, 256>>>(A, O, N); }, reps);
            printf("grid %6d mis %d copy  x1 %7.1f us %6.2f TB/s\n", grid, mis, t * 1e6, N * 8 / t / 1e12);
            t = time_it([&] { k_copy<float4><<<grid, 256>>>((const float4 *)A, (float4 *)O, n4); }, reps);
            printf("grid %6d mis %d copy  x4 %7.1f us %6.2f TB/s\n", grid, mis, t * 1e6, N * 8 / t / 1e12);
            t = time_it([&] { k_triad<float><<<grid, 256>>>(A, B, O, N); }, reps);
            printf("grid %6d mis %d triad x1 %7.1f us %6.2f TB/s\n", grid, mis, t * 1e6, N * 12 / t / 1e12);
            t = time_it([&] { k_triad<float4><<<grid, 256>>>((const float4 *)A, (const float4 *)B, (float4 *)O, n4); }, reps);
            printf("grid %6d mis %d triad x4 %7.1f us %6.2f TB/s\n", grid, mis, t * 1e6, N * 12 / t / 1e12);
        }
    }
    return 0;
}
