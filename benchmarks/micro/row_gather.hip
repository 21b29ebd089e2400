// Calibration of the HBM counters (FETCH_SIZE / WRITE_SIZE / TCC_EA0_*) for the access pattern of k_cfr_exact_lanes (scopa_multi.hip): every
// lane touches ONE 64-byte table row at a lane-dependent address -- 32 bytes of it read on entry, the whole row read and written at the
// traverser's visits -- which MI355X_MICROARCH.md's HBM section calls uncalibrated ("other access widths ... calibrate on a known byte
// count in your own access pattern").  Every kernel below moves a KNOWN number of distinct 64-byte rows of a table far larger than the
// 256 MiB Infinity Cache, each row touched exactly once per launch (row = an odd-multiplier bijection of the lane's global index):
//   k_stream16      16 B per lane, coalesced (the pattern the guide calibrated: FETCH_SIZE reads half the bytes)
//   k_gather32      32 B per lane (two 16-byte loads) from the first half of a distinct row
//   k_gather64      64 B per lane (four 16-byte loads), a distinct row
//   k_scatter64     64 B per lane stored (four 16-byte stores), a distinct row
//   k_rmw64         32 B read, then 64 B read + 64 B written to the same row (a traverser visit of the lanes kernel)
// Each prints rows/s and GB/s at 64 B per row from HIP events; under `rocprofv3 --pmc <counter>` the per-dispatch counter values divided by
// the row count give the bytes / requests the counters tally per row.  Physical cross-check: a mode whose rate x 128 B exceeds the 8 TB/s
// peak cannot be moving 128 B per row.
//   hipcc -O3 --offload-arch=gfx950 -o row_gather row_gather.hip && ./row_gather [log2 rows, default 27 = 8 GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ size_t row_of(size_t i, size_t mask) { return (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask; }   // odd multiplier: a bijection on 2^k

__global__ void __launch_bounds__(256) k_stream16(const double2 *__restrict__ t, size_t n16, double *__restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const double2 a = t[i];
    if (a.x == 1.2345e300) sink[0] = a.y;
}
__global__ void __launch_bounds__(256) k_gather32(const double2 *__restrict__ t, size_t n_rows, double *__restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    const double2 *r = t + row_of(i, n_rows - 1) * 4;
    const double2 a = r[0], b = r[1];
    if (a.x + b.y == 1.2345e300) sink[0] = a.y;
}
__global__ void __launch_bounds__(256) k_gather64(const double2 *__restrict__ t, size_t n_rows, double *__restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    const double2 *r = t + row_of(i, n_rows - 1) * 4;
    const double2 a = r[0], b = r[1], c = r[2], d = r[3];
    if (a.x + b.y + c.x + d.y == 1.2345e300) sink[0] = a.y;
}
__global__ void __launch_bounds__(256) k_scatter64(double2 *__restrict__ t, size_t n_rows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    double2 *r = t + row_of(i, n_rows - 1) * 4;
    const double2 v = make_double2((double)i, 1.0);
    r[0] = v; r[1] = v; r[2] = v; r[3] = v;
}
__global__ void __launch_bounds__(256) k_rmw64(double2 *__restrict__ t, size_t n_rows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    double2 *r = t + row_of(i, n_rows - 1) * 4;
    const double2 a0 = r[0], b0 = r[1];                      // entry: the regret half
    double s = a0.x + a0.y + b0.x + b0.y;
    for (int k = 0; k < 64; k++) s = s * 1.0000001 + 1e-9;   // (the recursion below the node stands between the two accesses)
    const double2 a = r[0], b = r[1], c = r[2], d = r[3];    // exit of a traverser visit: whole row read, whole row written
    r[0] = make_double2(a.x + s, a.y); r[1] = make_double2(b.x, b.y + s);
    r[2] = make_double2(c.x + 1.0, c.y); r[3] = make_double2(d.x, d.y + 1.0);
}

template <class F>
static void timed(const char *name, size_t rows, double bytes_per_row, F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("{\"kernel\": \"%s\", \"rows\": %zu, \"launches\": 4, \"best_ms\": %.4f, \"rows_per_s\": %.4g, \"GBps_at_%g_B_per_row\": %.1f}\n", name, rows, best,
           rows / (best * 1e-3), bytes_per_row, rows * bytes_per_row / (best * 1e-3) / 1e9);
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 27;
    const size_t rows = (size_t)1 << lg, bytes = rows * 64;
    double2 *t; double *sink;
    CK(hipMalloc(&t, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(t, 0, bytes));
    const unsigned grid_rows = (unsigned)((rows + 255) / 256);
    const size_t n16 = rows * 4;
    timed("k_stream16", rows, 64.0, [&] { hipLaunchKernelGGL(k_stream16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, t, n16, sink); });
    timed("k_gather32", rows, 64.0, [&] { hipLaunchKernelGGL(k_gather32, dim3(grid_rows), dim3(256), 0, 0, t, rows, sink); });
    timed("k_gather64", rows, 64.0, [&] { hipLaunchKernelGGL(k_gather64, dim3(grid_rows), dim3(256), 0, 0, t, rows, sink); });
    timed("k_scatter64", rows, 64.0, [&] { hipLaunchKernelGGL(k_scatter64, dim3(grid_rows), dim3(256), 0, 0, t, rows); });
    timed("k_rmw64", rows, 128.0, [&] { hipLaunchKernelGGL(k_rmw64, dim3(grid_rows), dim3(256), 0, 0, t, rows); });
    CK(hipFree(t)); CK(hipFree(sink));
    return 0;
}
