// Microbenchmark behind the slab-free design of k_mccfr_traverse: how long does it take 256 workgroups to add their LDS-resident
// partial delta table into table(s) in HBM with memory-side float64 atomics?  Findings (MI355X): the cost is set by how many
// workgroups hit the SAME 64-byte line (~20 ns per request and line, lines in parallel): 256 workgroups on one table = +5 us
// whatever the density; G = 8 tables (workgroup b adds into table b % 8) = 32 requests per line = +0.6 us.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o atomic_flush atomic_flush.hip && ./atomic_flush
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kRows = 738, kStride = 8, kCells = kRows * kStride;   // rows padded to one 64-byte line: 5 used cells + 3 unused

// hot != 0: every workgroup touches the SAME subset of cells (what peaked strategies do); else an independent subset per workgroup
__global__ void __launch_bounds__(1024) k_flush(double *__restrict__ g, int n_tables, unsigned density_pm, int hot, int work_iters) {
    __shared__ double s[kCells];
    for (int c = threadIdx.x; c < kCells; c += 1024) {
        const unsigned h = ((unsigned)c * 2654435761u + (hot ? 0u : blockIdx.x * 40503u)) >> 8;
        s[c] = ((c & 7) < 5 && (h % 1000u) < density_pm) ? 1.0 + (double)(c & 3) : 0.0;
    }
    __syncthreads();
    double acc = 0.0;
    for (int i = 0; i < work_iters; i++) acc += s[(threadIdx.x * 7 + i * 13 + blockIdx.x) % kCells];
    if (acc == 12345.678) s[0] = acc;
    __syncthreads();
    double *t = g + (size_t)(blockIdx.x % n_tables) * kCells;
    for (int c = threadIdx.x; c < kCells; c += 1024) {
        const double v = s[c];
        if (v != 0.0) atomicAdd(&t[c], v);
    }
}

static float run(double *g, int n_tables, unsigned density_pm, int hot, int work, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_flush, dim3(256), dim3(1024), 0, 0, g, n_tables, density_pm, hot, work);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_flush, dim3(256), dim3(1024), 0, 0, g, n_tables, density_pm, hot, work);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}

int main() {
    double *g;
    CK(hipMalloc(&g, (size_t)16 * kCells * sizeof(double)));
    CK(hipMemset(g, 0, (size_t)16 * kCells * sizeof(double)));
    const int reps = 2000;
    printf("us per launch (256 WGs x 1024 thr, back-to-back launches: includes the kernel boundary)\n");
    for (int work : {0, 1000}) {
        printf("work_iters=%d   baseline (nothing to add): %.2f\n", work, run(g, 1, 0, 1, work, reps));
        for (int hot : {1, 0})
            for (unsigned d : {34u, 150u, 1000u})
                for (int G : {1, 4, 8, 16})
                    printf("  %s subset, density %4.1f %%, %2d table(s): %.2f\n", hot ? "same " : "indep", d / 10.0, G, run(g, G, d, hot, work, reps));
    }
    return 0;
}
