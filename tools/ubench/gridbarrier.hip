// gridbarrier.hip -- what a barrier over several workgroups costs on this part: the price a multi-CU kernel
// for the small levels would pay per time step of a march, where the tiled sweeps pay nothing (tiles with a
// recomputed halo never wait for one another).  G resident workgroups of 256 threads meet K times at a counter in
// device memory (one atomic add per workgroup, then an acquire load in a loop); printed: time per barrier.
// Every wait is bounded: a workgroup that does not see the others within 2^20 polls raises a flag that ends
// every workgroup's loop, so the grid drains whatever happens.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o gridbarrier.bin gridbarrier.hip && ./gridbarrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(unsigned *counter, unsigned *gave_up, int nbar, float *sink)
{
    const unsigned g = gridDim.x;
    float acc = threadIdx.x;
    for (int b = 0; b < nbar; ++b) {
        acc = acc * 1.0001f + 1.0f; // a token of work between two barriers
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = g * (unsigned)(b + 1);
            int polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++polls > (1 << 20) || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    atomicAdd(gave_up, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            break;
    }
    if (acc == -1.0f)
        *sink = acc;
}

int main()
{
    unsigned *d;
    float *sink;
    hipMalloc(&d, 2 * sizeof(unsigned));
    hipMalloc(&sink, sizeof(float));
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int nbar = 2000;
    printf("%d CUs; %d barriers a launch, 256 threads a workgroup\n", prop.multiProcessorCount, nbar);
    printf("%12s %16s %10s\n", "workgroups", "us per barrier", "gave up");
    for (int g : {1, 2, 8, 16, 32, 64, 128, 256}) {
        if (g > prop.multiProcessorCount)
            break; // every workgroup must be resident at once
        float best = 1e30f;
        unsigned gave = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d, 0, 2 * sizeof(unsigned));
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            k<<<g, 256>>>(d, d + 1, nbar, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned h[2];
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            gave += h[1];
            if (ms < best)
                best = ms;
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
        printf("%12d %16.3f %10u\n", g, best * 1e3f / nbar, gave);
    }
    return 0;
}
