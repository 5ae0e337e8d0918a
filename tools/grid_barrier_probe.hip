// What does a software grid barrier cost on MI355X (256 CUs, 8 XCDs with private L2s)?  A persistent kernel of G workgroups runs N
// phases; every phase each workgroup writes a line of a buffer, then all meet at a barrier (device-scope release, atomic arrive,
// spin with device-scope acquire), then each reads the line its NEIGHBOUR wrote (a stale read = the barrier does not order memory
// across XCDs).  Compared with N dependent launches of the same one-phase kernel.  Decides whether a persistent "phase interpreter"
// kernel could replace the decoder's ~260 dependent launches per step (7-8 us each on 400 rows).
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o build/grid_barrier_probe && build/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, long long timeout_cycles) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                         // release: this workgroup's writes are visible device-wide
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = clock64();
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (clock64() - t0 > timeout_cycles) break;           // never hang the box
        }
        __threadfence();
    }
    __syncthreads();
    return true;
}

__global__ void persistent(unsigned* counter, int* buf, int nphases, int* errors, unsigned long long* cycles) {
    const int G = gridDim.x, b = blockIdx.x;
    const unsigned long long t0 = clock64();
    int bad = 0;
    for (int p = 0; p < nphases; ++p) {
        buf[(size_t)b * 64 + threadIdx.x % 64] = p * 1000003 + b;          // "the phase's output"
        grid_barrier(counter, (unsigned)(G * (p + 1)), 200000000LL);
        const int nb = (b + 37) % G;
        const int v = __builtin_nontemporal_load(&buf[(size_t)nb * 64 + threadIdx.x % 64]);
        if (v != p * 1000003 + nb) ++bad;
        grid_barrier(counter + 32, (unsigned)(G * (p + 1)), 200000000LL);  // (second barrier: nobody overwrites before all have read)
    }
    if (bad && threadIdx.x == 0) atomicAdd(errors, 1);
    if (b == 0 && threadIdx.x == 0) cycles[0] = clock64() - t0;
}

__global__ void one_phase(int* buf, int p) {
    buf[(size_t)blockIdx.x * 64 + threadIdx.x % 64] = p * 1000003 + blockIdx.x;
}

int main() {
    unsigned* counter; int* buf; int* errors; unsigned long long* cycles;
    hipMalloc(&counter, 256); hipMalloc(&buf, 4096 * 64 * 4); hipMalloc(&errors, 4); hipMalloc(&cycles, 8);
    for (int G : {64, 256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int N = 200;
            hipMemset(counter, 0, 256); hipMemset(errors, 0, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(persistent, dim3(G), dim3(256), 0, 0, counter, buf, N, errors, cycles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int herr; hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
            hipEventRecord(e0);
            for (int p = 0; p < N; ++p) hipLaunchKernelGGL(one_phase, dim3(G), dim3(256), 0, 0, buf, p);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms2; hipEventElapsedTime(&ms2, e0, e1);
            printf("G=%3d workgroups: persistent %7.2f us per phase (2 barriers each; stale reads in %d workgroups)   dependent launches %6.2f us per phase\n",
                   G, ms * 1e3 / N, herr, ms2 * 1e3 / N);
        }
    }
    return 0;
}
