// Grid-barrier probe: what does one device-wide phase boundary cost INSIDE a resident kernel on an MI355X (256 CUs in 8 XCDs, one L2 per
// XCD), against the ~4.8 us floor of a small kernel in a stream (launch + ramp + drain)?  The decoder of the train step is ~340 launches
// of 5-10 us over 400 rows: if a boundary inside a persistent kernel costs well under the launch floor, its phases belong in one kernel.
//   every workgroup: write `bytes` of its own slice (values = the iteration), barrier, read `bytes` of ANOTHER workgroup's slice (an XCD
//   away) and check them -- the check proves cross-XCD visibility, the loop time / iterations is the cost of a phase boundary + that traffic.
// Variants: flat (one counter, every workgroup adds and spins on it), tree (one counter per XCD residue blockIdx % 8, the last arriver of
// each adds to a root, everybody spins on a generation word).
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o build/grid_barrier_probe && build/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define AGENT __HIP_MEMORY_SCOPE_AGENT

struct Bar { unsigned* flat; unsigned* leaf; unsigned* root; unsigned* gen; };

template <int KIND>
__device__ __forceinline__ void grid_barrier(const Bar& b, unsigned it, unsigned G) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if constexpr (KIND == 0) {
            __hip_atomic_fetch_add(b.flat, 1u, __ATOMIC_RELEASE, AGENT);
            const unsigned want = (it + 1) * G;
            while (__hip_atomic_load(b.flat, __ATOMIC_ACQUIRE, AGENT) < want) __builtin_amdgcn_s_sleep(1);
        } else {
            const unsigned x = blockIdx.x & 7, per = (G + 7 - x) / 8;                   // workgroups with this residue
            const unsigned a = __hip_atomic_fetch_add(b.leaf + x * 32, 1u, __ATOMIC_ACQ_REL, AGENT);
            if (a == (it + 1) * per - 1) {
                const unsigned r = __hip_atomic_fetch_add(b.root, 1u, __ATOMIC_ACQ_REL, AGENT);
                if (r == (it + 1) * 8 - 1) __hip_atomic_store(b.gen, it + 1, __ATOMIC_RELEASE, AGENT);
            }
            while (__hip_atomic_load(b.gen, __ATOMIC_ACQUIRE, AGENT) < it + 1) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int KIND>
__global__ __launch_bounds__(256) void probe(Bar b, uint4* __restrict__ buf, int words16, int iters, unsigned* __restrict__ bad) {
    const unsigned G = gridDim.x, me = blockIdx.x, other = (me + 37) % G;
    unsigned errs = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned v = (unsigned)it * 2654435761u + 12345u;
        for (int i = threadIdx.x; i < words16; i += 256) buf[(size_t)me * words16 + i] = uint4{v, v + me, v, v};
        grid_barrier<KIND>(b, 2 * it, G);
        for (int i = threadIdx.x; i < words16; i += 256) {
            const uint4 r = buf[(size_t)other * words16 + i];
            errs += (r.x != v) + (r.y != v + other);
        }
        grid_barrier<KIND>(b, 2 * it + 1, G);          // (the slice is overwritten next iteration: readers first)
    }
    if (errs) atomicAdd(bad, errs);
}

template <int KIND>
static void run(const char* name, int G, int bytes, int iters) {
    unsigned* ctr; uint4* buf; unsigned* bad;
    hipMalloc(&ctr, 4096 * 4); hipMalloc(&buf, (size_t)G * bytes + 64); hipMalloc(&bad, 4);
    hipMemset(ctr, 0, 4096 * 4); hipMemset(bad, 0, 4);
    Bar b{ctr, ctr + 64, ctr + 64 + 8 * 32, ctr + 64 + 8 * 32 + 32};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(G), dim3(256), 0, 0, b, buf, bytes / 16, 10, bad);       // warm
    hipDeviceSynchronize();
    hipMemset(ctr, 0, 4096 * 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(G), dim3(256), 0, 0, b, buf, bytes / 16, iters, bad);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("%-5s G = %4d  %6d B written + read per workgroup and phase: %6.2f us per barrier (2 per iteration), %u stale reads\n", name, G, bytes,
           ms * 1e3 / iters / 2, h);
    hipFree(ctr); hipFree(buf); hipFree(bad);
}

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main() {
    const int iters = 2000;
    for (int G : {256, 512, 1024})
        for (int bytes : {0, 4096, 65536}) {
            if (G == 1024 && bytes == 65536) continue;
            run<0>("flat", G, bytes, iters);
            run<1>("tree", G, bytes, iters);
        }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int G : {256, 1024}) {
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(256), 0, 0, (int*)nullptr);
        hipEventRecord(e0);
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(256), 0, 0, (int*)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel, G = %4d, back to back in a stream: %6.2f us per launch\n", G, ms * 1e3 / 2000);
    }
    return 0;
}
