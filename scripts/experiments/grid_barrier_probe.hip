// What does a device-wide barrier inside ONE kernel cost on MI355X, against the ~6.5 us floor between two dependent launches of a
// batch-1 forward?  A grid of `blocks` x 256 threads runs `rounds` barriers: release fence (agent scope: L2 write-back, the
// XCDs' L2s only meet in memory), one atomic arrival per block, a spin on the generation word, acquire fence.  Every spin is
// bounded: a block that does not see its peers after ~50 ms gives up and the run reports it (a kernel that cannot hang the box).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbp scripts/experiments/grid_barrier_probe.hip && /tmp/gbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe(unsigned *count, unsigned *gen, int rounds, float *data, int touch, int *gave_up)
{
    __shared__ int dead;
    if (threadIdx.x == 0) dead = 0;
    __syncthreads();
    const unsigned nb = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        if (touch) {      // a little cross-block traffic, so that the fences have something to write back / invalidate
            const int i = (int)((blockIdx.x * 256u + threadIdx.x + (unsigned)r * 977u) % (nb * 256u));
            data[i] = data[(i * 7 + 3) % (nb * 256)] + 1.0f;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);          // agent scope: write back what other XCDs will read
            const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned a = __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (a == nb - 1) {
                __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                long long t0 = wall_clock64();
                while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > 5000000LL) { dead = 1; break; }     // 100 MHz clock: 50 ms
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        if (dead) { if (threadIdx.x == 0) atomicAdd(gave_up, 1); return; }
    }
}

int main()
{
    unsigned *sync;
    float *data;
    int *gave_up;
    CHK(hipMalloc(&sync, 256));
    CHK(hipMalloc(&data, 4096 * 256 * 4));
    CHK(hipMalloc(&gave_up, 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int rounds = 200;
    for (int touch = 0; touch < 2; ++touch)
        for (int blocks : {256, 512, 1024}) {
            CHK(hipMemset(sync, 0, 256));
            CHK(hipMemset(data, 0, 4096 * 256 * 4));
            CHK(hipMemset(gave_up, 0, 4));
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, sync, sync + 32, rounds, data, touch, gave_up);
                CHK(hipEventRecord(e1, 0));
                CHK(hipEventSynchronize(e1));
                float ms = 0;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                int g = 0;
                CHK(hipMemcpy(&g, gave_up, 4, hipMemcpyDeviceToHost));
                if (rep == 2 || g) printf("blocks %4d  touch %d: %7.2f us per barrier (%d rounds, %.1f us launch incl.)%s\n", blocks, touch, ms * 1e3 / rounds, rounds, ms * 1e3,
                                          g ? "  -- blocks gave up waiting (not all resident?)" : "");
                if (g) break;
            }
        }
    // for comparison: the same number of dependent empty launches
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, sync, sync + 32, 0, data, 0, gave_up);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%d dependent launches of 256 empty blocks: %.2f us each\n", rounds, ms * 1e3 / rounds);
    return 0;
}
