// What one wavefront gets out of v_mfma_f32_16x16x4_f32 in the shapes k_sdcfr_traverse's layer 2 uses (gfx950): shader clocks per MFMA
// for 128 MFMAs in NCH independent accumulator chains, with (a) all operands in registers, (b) the B operand produced by a VALU
// instruction just before each group of NCH MFMAs (the relu of the previous layer), (c) the A operands streamed from LDS one step
// ahead, (d) b + c.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_chain mfma_f32_chain.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mfma16(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int NCH, bool VALU_B, bool LDS_A>
__global__ void __launch_bounds__(512) k(unsigned long long *out, const float *in, float *sink, int reps) {
    __shared__ float4 s_w[32 * 64];                          // 32 K-step groups x 64 lanes x 16 bytes = 32 KB
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) s_w[i] = make_float4(in[i & 255], in[(i + 1) & 255], in[(i + 2) & 255], in[(i + 3) & 255]);
    __syncthreads();
    v4f acc[NCH];
    for (int c = 0; c < NCH; c++) acc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
    float hsrc[32];
    for (int i = 0; i < 32; i++) hsrc[i] = in[(lane + i) & 255];
    float4 wr[NCH];                                          // register-resident A operands for the no-LDS variants
    for (int c = 0; c < NCH; c++) wr[c] = s_w[c * 64 + lane];
    const unsigned long long t0 = clock64();
    for (int rep = 0; rep < reps; rep++) {
        constexpr int STEPS = 128 / (4 * NCH);               // steps of 4 * NCH MFMAs each
        float4 cur[NCH], nxt[NCH];
        if (LDS_A) { for (int c = 0; c < NCH; c++) cur[c] = s_w[(c * STEPS + 0) * 64 % (32 * 64) + lane]; }
#pragma unroll
        for (int st = 0; st < STEPS; st++) {
            if (LDS_A) {
#pragma unroll
                for (int c = 0; c < NCH; c++) nxt[c] = s_w[((c * STEPS + (st + 1) % STEPS) * 64) % (32 * 64) + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float b = hsrc[(st * 4 + r) & 31];
                if (VALU_B) b = __builtin_amdgcn_fmed3f(b, 0.0f, __builtin_huge_valf());
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const float4 w = LDS_A ? cur[c] : wr[c];
                    const float a = r == 0 ? w.x : r == 1 ? w.y : r == 2 ? w.z : w.w;
                    acc[c] = mfma16(a, b, acc[c]);
                }
            }
            if (LDS_A) {
#pragma unroll
                for (int c = 0; c < NCH; c++) cur[c] = nxt[c];
            }
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < NCH; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NCH, bool VALU_B, bool LDS_A> static void run(const char *name, int waves) {
    unsigned long long *d, h[16];
    float *in, *sink, hin[256];
    for (int i = 0; i < 256; i++) hin[i] = (float)((i * 37) % 101) / 101.0f - 0.5f;
    (void)hipMalloc(&d, 4096); (void)hipMalloc(&in, 1024); (void)hipMalloc(&sink, 1 << 20);
    (void)hipMemcpy(in, hin, 1024, hipMemcpyHostToDevice);
    const int reps = 64;
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<NCH, VALU_B, LDS_A>), dim3(1), dim3(64 * waves), 0, 0, d, in, sink, reps);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, 8 * waves, hipMemcpyDeviceToHost);
    double mx = 0; for (int w = 0; w < waves; w++) mx = h[w] > mx ? (double)h[w] : mx;
    printf("%-44s %d wave(s)/CU: %6.2f clocks per MFMA per wave; per SIMD %6.2f\n", name, waves, mx / (128.0 * reps), mx / (128.0 * reps) / ((waves + 3) / 4));
    (void)hipFree(d); (void)hipFree(in); (void)hipFree(sink);
}

int main() {
    for (int waves : {1, 4, 8}) {
        run<4, false, false>("4 chains, operands in registers", waves);
        run<8, false, false>("8 chains, operands in registers", waves);
        run<2, false, false>("2 chains, operands in registers", waves);
        run<4, true, false>("4 chains, B through a VALU op per group", waves);
        run<4, false, true>("4 chains, A from LDS one step ahead", waves);
        run<4, true, true>("4 chains, VALU B + LDS A (layer 2's shape)", waves);
        run<8, true, true>("8 chains, VALU B + LDS A", waves);
    }
    return 0;
}
