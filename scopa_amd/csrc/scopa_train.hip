// scopa_train.hip -- one optimiser step of the advantage net (AdvantageNetwork.train, src/algorithms/deep_cfr/deep_cfr.py:77-116) in TWO launches.
//
// The reference's step on a 128-row batch -- gather, 34-128-64-16 MLP forward, MSE(pred * mask, target * mask), backward, clip_grad_norm_(1.0), Adam(5e-4) --
// is about thirty dependent PyTorch kernels of 4-5 us each whatever they compute.  The default stays PyTorch (north_star: "the SDCFR
// advantage MLP trains on PyTorch-ROCm"); this file is the OPT-IN alternative (`DeepCFR(train_backend="hip")`), measured beside it:
//   k_sdcfr_train_grad : a workgroup per group of 16-row tiles; forward and backward of a tile on v_mfma_f32_16x16x4_f32 with the activations in LDS
//                        ([row][unit], odd strides) and the weights staged there once per workgroup from the net's own torch tensors W[out][in] (coalesced
//                        8- / 16-byte loads; both W and W^T operands are read from the one copy); the weight gradients accumulate in MFMA accumulators
//                        across the workgroup's tiles and leave as ONE partial gradient per workgroup (summed later in fixed order: the result does not
//                        depend on scheduling);
//   k_sdcfr_train_adam : one workgroup: sums the partials in order, 2-norm of the whole gradient, clip coefficient min(1, 1 / (norm + 1e-6)), Adam's update
//                        (bias-corrected, eps outside the square root, as torch.optim.Adam) in place on the net's tensors and on the [2][13776] moment buffer.
// MFMA roles (one instruction = a 16 x 16 tile over 4 K values): A lane l = A[l % 16][l / 16], B lane l = B[l / 16][l % 16], D lane l register r = D[4 (l / 16) + r][l % 16].
#include <cmath>

#include "scopa_ctx.h"

using namespace scopa;

namespace {
constexpr int kIn = 34, kH1 = 128, kH2 = 64, kOut = 16;
constexpr int kOffW1 = 0, kOffB1 = kOffW1 + kH1 * kIn, kOffW2 = kOffB1 + kH1, kOffB2 = kOffW2 + kH2 * kH1, kOffW3 = kOffB2 + kH2, kOffB3 = kOffW3 + kOut * kH2;
constexpr int kParams = kOffB3 + kOut;   // 13 776, in net.parameters() order
static_assert(kParams == 13776, "parameter count of the 34-128-64-16 MLP");
constexpr int kMaxPartials = 32;
constexpr int kSX = 37, kS1 = 129, kS2 = 65, kSD = 17;   // LDS row strides (floats), odd: a wavefront reads a tile by rows and by columns
constexpr int kTW1 = 36, kTW2 = 132, kTW3 = 68;          // row strides of the weights' LDS copies (multiples of 2 / 4 floats: they are staged 8 / 16 bytes at a time)

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mfma(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
}  // namespace

__global__ void __launch_bounds__(256)
k_sdcfr_train_grad(const int64_t *__restrict__ g_rows, int n_rows, long long capacity, const float *__restrict__ g_feat, const float *__restrict__ g_regret, const float *__restrict__ g_mask,
                   const float *__restrict__ W1, const float *__restrict__ B1, const float *__restrict__ W2, const float *__restrict__ B2,
                   const float *__restrict__ W3, const float *__restrict__ B3, float *__restrict__ g_partial /* [gridDim.x][kParams + 1] */) {
    __shared__ float s_x[16 * kSX], s_h1[16 * kS1], s_h2[16 * kS2], s_d[16 * kSD], s_dz2[16 * kS2], s_dz1[16 * kS1];
    __shared__ __align__(16) float s_w1[kH1 * kTW1], s_w2[kH2 * kTW2], s_w3[kOut * kTW3];   // the weights, staged once per workgroup with coalesced loads (as MFMA operands straight from
    __shared__ float s_t[16 * kSD], s_m[16 * kSD];   // the tile's target and mask rows, gathered with its features (not on the critical path in front of the loss)
    __shared__ long long s_row[16];                                                          // global memory every lane read 4 bytes of its own cache line, once per K step and phase)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nj = lane & 15, q = lane >> 4;
    const int n_tiles = n_rows / 16;                       // (the host sends whole tiles only)
    const float dscale = 2.0f / (float)(n_rows * kOut);    // d MSE / d pred: 2 (pred m - target m) m / (rows x 16)
    const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
    // weight gradients, accumulated over this workgroup's tiles in MFMA accumulators: wavefront w owns
    //   dW2 rows (unit2) 16 w .. 16 w + 15, all 8 column tiles (unit1);  dW1 rows (unit1) 32 w .. 32 w + 31, 3 column tiles (feature);  dW3 columns (unit2) 16 w .. + 15
    v4f a_w2[8], a_w1[2][3], a_w3 = zero;
#pragma unroll
    for (int i = 0; i < 8; i++) a_w2[i] = zero;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int c = 0; c < 3; c++) a_w1[h][c] = zero;
    float db1 = 0.0f, db2 = 0.0f, db3 = 0.0f, loss = 0.0f;   // thread t: bias gradients of unit t (t < 128 / 64 / 16); loss: wavefront 0
    for (int e = tid; e < kH1 * kIn / 2; e += 256) { const int r = (2 * e) / kIn, c = 2 * e - r * kIn; *reinterpret_cast<float2 *>(&s_w1[r * kTW1 + c]) = reinterpret_cast<const float2 *>(W1)[e]; }
    for (int e = tid; e < kH2 * kH1 / 4; e += 256) { const int r = (4 * e) / kH1, c = 4 * e - r * kH1; *reinterpret_cast<float4 *>(&s_w2[r * kTW2 + c]) = reinterpret_cast<const float4 *>(W2)[e]; }
    for (int e = tid; e < kOut * kH2 / 4; e += 256) { const int r = (4 * e) / kH2, c = 4 * e - r * kH2; *reinterpret_cast<float4 *>(&s_w3[r * kTW3 + c]) = reinterpret_cast<const float4 *>(W3)[e]; }

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                   // the previous tile's LDS is no longer read
        if (tid < 16) { const long long r = g_rows[tile * 16 + tid]; s_row[tid] = r < 0 ? 0 : r >= capacity ? capacity - 1 : r; }   // (a row index outside the ring is clamped, never dereferenced)
        __syncthreads();
        for (int e = tid; e < 16 * kIn; e += 256) { const int r = e / kIn, c = e - r * kIn; s_x[r * kSX + c] = g_feat[(size_t)s_row[r] * kIn + c]; }
        { const int r = tid >> 4, c = tid & 15; s_t[r * kSD + c] = g_regret[(size_t)s_row[r] * kOut + c]; s_m[r * kSD + c] = g_mask ? g_mask[(size_t)s_row[r] * kOut + c] : g_feat[(size_t)s_row[r] * kIn + c]; }   // no mask array: mask = features[0..16) (scopa_sdcfr.hip sd_mask_note)
        __syncthreads();
        {   // ---- layer 1: h1[unit][row] = relu(W1 x + b1); wavefront w: unit tiles 2 w, 2 w + 1
            v4f acc[2];
#pragma unroll
            for (int h = 0; h < 2; h++) { const float *b = B1 + 16 * (2 * w + h) + 4 * q; acc[h] = v4f{b[0], b[1], b[2], b[3]}; }
#pragma unroll
            for (int s = 0; s < 9; s++) {
                const int k = 4 * s + q;
                const bool in = k < kIn;
                const float b = in ? s_x[nj * kSX + k] : 0.0f;
#pragma unroll
                for (int h = 0; h < 2; h++) acc[h] = mfma(in ? s_w1[(16 * (2 * w + h) + nj) * kTW1 + k] : 0.0f, b, acc[h]);
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int r = 0; r < 4; r++) s_h1[nj * kS1 + 16 * (2 * w + h) + 4 * q + r] = fmaxf(acc[h][r], 0.0f);
        }
        __syncthreads();
        {   // ---- layer 2: wavefront w: unit tile w
            const float *b = B2 + 16 * w + 4 * q;
            v4f acc = {b[0], b[1], b[2], b[3]};
#pragma unroll 8
            for (int s = 0; s < 32; s++) { const int k = 4 * s + q; acc = mfma(s_w2[(16 * w + nj) * kTW2 + k], s_h1[nj * kS1 + k], acc); }
#pragma unroll
            for (int r = 0; r < 4; r++) s_h2[nj * kS2 + 16 * w + 4 * q + r] = fmaxf(acc[r], 0.0f);
        }
        __syncthreads();
        if (w == 0) {   // ---- layer 3, the loss and its gradient d[row][out]
            const float *b = B3 + 4 * q;
            v4f acc = {b[0], b[1], b[2], b[3]};
#pragma unroll
            for (int s = 0; s < 16; s++) { const int k = 4 * s + q; acc = mfma(s_w3[nj * kTW3 + k], s_h2[nj * kS2 + k], acc); }
            float l = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = 4 * q + r;
                const float m = s_m[nj * kSD + o], e = (acc[r] - s_t[nj * kSD + o]) * m;   // pred * mask - target * mask with a 0 / 1 mask
                l += e * e;
                s_d[nj * kSD + o] = e * dscale;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) l += __shfl_xor(l, off, 64);
            loss += l;
        }
        __syncthreads();
        {   // ---- dW3[out][unit2] += sum_row d[row][out] h2[row][unit2]   (wavefront w: unit2 tile w);   dz2 = (W3^T d) * (h2 > 0)   (wavefront w: unit2 tile w)
            v4f acc = zero;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int k = 4 * s + q;                                      // row within the tile
                a_w3 = mfma(s_d[k * kSD + nj], s_h2[k * kS2 + 16 * w + nj], a_w3);
                acc = mfma(s_w3[k * kTW3 + 16 * w + nj], s_d[nj * kSD + k], acc);    // here k is an OUTPUT: A = W3^T[unit2][out], B = d[out][row]
            }
#pragma unroll
            for (int r = 0; r < 4; r++) { const int u = 16 * w + 4 * q + r; s_dz2[nj * kS2 + u] = s_h2[nj * kS2 + u] > 0.0f ? acc[r] : 0.0f; }
        }
        __syncthreads();
        {   // ---- dW2[unit2][unit1] += sum_row dz2[row][unit2] h1[row][unit1]   (wavefront w: unit2 tile w);   dz1 = (W2^T dz2) * (h1 > 0)   (wavefront w: unit1 tiles 2 w, 2 w + 1)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int k = 4 * s + q;
                const float a = s_dz2[k * kS2 + 16 * w + nj];
#pragma unroll
                for (int c = 0; c < 8; c++) a_w2[c] = mfma(a, s_h1[k * kS1 + 16 * c + nj], a_w2[c]);
            }
            v4f acc[2] = {zero, zero};
#pragma unroll 4
            for (int s = 0; s < 16; s++) {
                const int k = 4 * s + q;                                      // unit2
                const float b = s_dz2[nj * kS2 + k];
#pragma unroll
                for (int h = 0; h < 2; h++) acc[h] = mfma(s_w2[k * kTW2 + 16 * (2 * w + h) + nj], b, acc[h]);
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int r = 0; r < 4; r++) { const int u = 16 * (2 * w + h) + 4 * q + r; s_dz1[nj * kS1 + u] = s_h1[nj * kS1 + u] > 0.0f ? acc[h][r] : 0.0f; }
        }
        __syncthreads();
        {   // ---- dW1[unit1][feature] += sum_row dz1[row][unit1] x[row][feature]   (wavefront w: unit1 tiles 2 w, 2 w + 1); bias gradients
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int k = 4 * s + q;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int f = 16 * c + nj;
                    const float b = f < kIn ? s_x[k * kSX + f] : 0.0f;
#pragma unroll
                    for (int h = 0; h < 2; h++) a_w1[h][c] = mfma(s_dz1[k * kS1 + 16 * (2 * w + h) + nj], b, a_w1[h][c]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (tid < kH1) db1 += s_dz1[r * kS1 + tid];
                if (tid < kH2) db2 += s_dz2[r * kS2 + tid];
                if (tid < kOut) db3 += s_d[r * kSD + tid];
            }
        }
    }
    // ---- this workgroup's partial gradient (net.parameters() order) and partial loss
    float *out = g_partial + (size_t)blockIdx.x * (kParams + 1);
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) { const int u = 16 * (2 * w + h) + 4 * q + r, f = 16 * c + nj; if (f < kIn) out[kOffW1 + u * kIn + f] = a_w1[h][c][r]; }
#pragma unroll
    for (int c = 0; c < 8; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) out[kOffW2 + (16 * w + 4 * q + r) * kH1 + 16 * c + nj] = a_w2[c][r];
#pragma unroll
    for (int r = 0; r < 4; r++) out[kOffW3 + (4 * q + r) * kH2 + 16 * w + nj] = a_w3[r];
    if (tid < kH1) out[kOffB1 + tid] = db1;
    if (tid < kH2) out[kOffB2 + tid] = db2;
    if (tid < kOut) out[kOffB3 + tid] = db3;
    if (tid == 0) out[kParams] = loss;
}

__global__ void __launch_bounds__(1024)
k_sdcfr_train_adam(const float *__restrict__ g_partial, int n_partials, int n_rows, float *__restrict__ W1, float *__restrict__ B1, float *__restrict__ W2,
                   float *__restrict__ B2, float *__restrict__ W3, float *__restrict__ B3, float *__restrict__ g_state /* [2][kParams] */, float step_size, float bc2_sqrt,
                   float beta1, float beta2, float eps, float *__restrict__ g_loss) {
    __shared__ float s_red[16];
    const int tid = threadIdx.x;
    constexpr int kPer = (kParams + 1023) / 1024;   // 14
    static_assert(kPer == 14, "the partial sums below go in two halves of seven");
    float g[kPer], sq = 0.0f;
    // the partials summed in order, eight per pass: their loads (14 elements x 8 partials per thread) are independent and in flight together -- as a loop over
    // a run-time count the 112 loads of a thread went out one after the other (35 us of a 50 us step)
#pragma unroll
    for (int j = 0; j < kPer; j++) g[j] = 0.0f;
    for (int p0 = 0; p0 < n_partials; p0 += 8) {
#pragma unroll
        for (int jh = 0; jh < kPer; jh += 7) {          // seven elements x eight partials = 56 loads in flight per thread (all fourteen at once spilled registers)
            float t[7][8];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const int i = tid + 1024 * (jh + j);
#pragma unroll
                for (int u = 0; u < 8; u++) t[j][u] = (i < kParams && p0 + u < n_partials) ? g_partial[(size_t)(p0 + u) * (kParams + 1) + i] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 7; j++)
#pragma unroll
                for (int u = 0; u < 8; u++) g[jh + j] += t[j][u];
        }
    }
#pragma unroll
    for (int j = 0; j < kPer; j++) sq += g[j] * g[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = sq;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) tot += s_red[k];
    const float norm = sqrtf(tot);
    const float inv = 1.0f / (norm + 1e-6f), coef = inv < 1.0f ? inv : 1.0f;   // clip_grad_norm_(max_norm = 1.0): clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        const int i = tid + 1024 * j;
        if (i >= kParams) continue;
        float *p = i < kOffB1 ? W1 + (i - kOffW1) : i < kOffW2 ? B1 + (i - kOffB1) : i < kOffB2 ? W2 + (i - kOffW2) : i < kOffW3 ? B2 + (i - kOffB2) : i < kOffB3 ? W3 + (i - kOffW3) : B3 + (i - kOffB3);
        const float gr = g[j] * coef;
        const float m = g_state[i] + (gr - g_state[i]) * (1.0f - beta1);                   // exp_avg.lerp_(grad, 1 - beta1)
        const float v = g_state[kParams + i] * beta2 + (1.0f - beta2) * gr * gr;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        g_state[i] = m;
        g_state[kParams + i] = v;
        *p = *p - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));                          // param.addcdiv_(exp_avg, denom, value = -step_size)
    }
    if (tid == 0) {
        float l = 0.0f;
        for (int p = 0; p < n_partials; p++) l += g_partial[(size_t)p * (kParams + 1) + kParams];
        g_loss[0] += l / (float)(n_rows * kOut);                                          // this step's MSE, added to the caller's running sum
    }
}

extern "C" {

int32_t scopa_sdcfr_train_params(void) { return kParams; }

int32_t scopa_sdcfr_train_steps(scopa_ctx *ctx, const int64_t *d_rows, int32_t n_rows, int32_t n_steps, const float *d_feat, const float *d_regret, const float *d_mask,
                                int64_t capacity, float *d_w1, float *d_b1, float *d_w2, float *d_b2, float *d_w3, float *d_b3, float *d_state, int32_t first_step, float lr,
                                float *d_loss) {
    if (!ctx || !d_rows || !d_feat || !d_regret || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_w3 || !d_b3 || !d_state || !d_loss) return SCOPA_EINVAL;
    SC_REQUIRE(ctx, n_rows >= 16 && n_rows % 16 == 0 && n_rows <= (1 << 20), SCOPA_EINVAL, "scopa_sdcfr_train_steps: the batch must be a multiple of 16 rows (16 .. 2^20)");
    SC_REQUIRE(ctx, capacity >= 1 && first_step >= 1 && n_steps >= 0 && n_steps <= 4096 && lr > 0.0f, SCOPA_EINVAL,
               "scopa_sdcfr_train_steps: capacity, first_step (1-based) and lr must be positive, n_steps in 0 .. 4096");
    SC_REQUIRE(ctx, ((uintptr_t)d_w1 & 7) == 0 && ((uintptr_t)d_w2 & 15) == 0 && ((uintptr_t)d_w3 & 15) == 0, SCOPA_EINVAL,
               "scopa_sdcfr_train_steps: the weight tensors are staged 8 / 16 bytes at a time: d_w1 must be 8-byte, d_w2 / d_w3 16-byte aligned");
    SC_HIP(ctx, hipSetDevice(ctx->device));
    scopa::Range range_("scopa sdcfr train");
    if (!ctx->d_train_partial) SC_HIP(ctx, hipMalloc(&ctx->d_train_partial, sizeof(float) * (size_t)kMaxPartials * (kParams + 1)));
    const int n_tiles = n_rows / 16, grid = n_tiles < kMaxPartials ? n_tiles : kMaxPartials;
    for (int e = 0; e < n_steps; e++) {
        hipLaunchKernelGGL(k_sdcfr_train_grad, dim3(grid), dim3(256), 0, ctx->stream, d_rows + (size_t)e * n_rows, (int)n_rows, (long long)capacity, d_feat, d_regret, d_mask, (const float *)d_w1,
                           (const float *)d_b1, (const float *)d_w2, (const float *)d_b2, (const float *)d_w3, (const float *)d_b3, (float *)ctx->d_train_partial);
        // torch.optim.Adam's scalars (_single_tensor_adam): bias_correction1 = 1 - beta1 ** step, step_size = lr / bias_correction1, bias_correction2_sqrt = sqrt(1 - beta2 ** step)
        // in Python floats (float64), rounded to float32 only where they meet the tensors -- computed the same way here (in-kernel powf on float32 betas was ~1e-5 off at small steps)
        const double step = (double)(first_step + e), bc1 = 1.0 - std::pow(0.9, step), bc2 = 1.0 - std::pow(0.999, step);
        hipLaunchKernelGGL(k_sdcfr_train_adam, dim3(1), dim3(1024), 0, ctx->stream, (const float *)ctx->d_train_partial, grid, (int)n_rows, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3,
                           d_state, (float)((double)lr / bc1), (float)std::sqrt(bc2), 0.9f, 0.999f, 1e-8f, d_loss);
    }
    SC_HIP(ctx, hipGetLastError());
    return SCOPA_OK;
}

int32_t scopa_sdcfr_train_step(scopa_ctx *ctx, const int64_t *d_rows, int32_t n_rows, const float *d_feat, const float *d_regret, const float *d_mask, int64_t capacity,
                               float *d_w1, float *d_b1, float *d_w2, float *d_b2, float *d_w3, float *d_b3, float *d_state, int32_t step, float lr, float *d_loss) {
    return scopa_sdcfr_train_steps(ctx, d_rows, n_rows, 1, d_feat, d_regret, d_mask, capacity, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, d_state, step, lr, d_loss);
}

}  // extern "C"
