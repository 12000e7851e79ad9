// LayerNorm (+residual, +ReLU, +dropout) forward/backward for the DIN trunk and head (HBM-bound):
//   nl_emb_1 (infer_model.py:185)  rows = B*T*N, len = 1024, affine [1024]
//   point_ln (:192), dpi_nl (:214), hier_LN (dynamic_infer_module.py:493)  rows = B, len = T*N*C, affine [T,N,C]
// One workgroup per row; the row is streamed with float4 loads.  Two-pass mean/variance (matches ATen's numerics to
// fp32 rounding).  Dropout uses a counter-based hash so the backward regenerates the same keep-mask from (seed, index).
#include "din_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}

constexpr int LN_THREADS = 512;
constexpr int LN_WAVES = LN_THREADS / 64;

__global__ __launch_bounds__(LN_THREADS) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, float* __restrict__ y, float* __restrict__ stats, int64_t len, int relu, float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_off) {
    __shared__ float red[LN_WAVES];
    seed = fold_seed(seed, seed_off);
    const int64_t row = blockIdx.x;
    const float* xr = x + row * len;
    const float* rr = res ? res + row * len : nullptr;
    // rows are streamed as float4 when len % 4 == 0 (every DIN shape); element order inside the sums differs from the scalar loop only
    // by fp32 rounding
    const bool vec = (len & 3) == 0;
    const int64_t n4 = len >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(xr);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(rr);
    float s = 0.f;
    if (vec) {
        for (int64_t i = threadIdx.x; i < n4; i += LN_THREADS) {
            f32x4 a = x4[i];
            if (rr) a += r4[i];
            s += (a[0] + a[1]) + (a[2] + a[3]);
        }
    } else {
        for (int64_t i = threadIdx.x; i < len; i += LN_THREADS) s += xr[i] + (rr ? rr[i] : 0.f);
    }
    const float mean = block_sum<LN_WAVES>(s, red) / (float)len;
    float q = 0.f;
    if (vec) {
        for (int64_t i = threadIdx.x; i < n4; i += LN_THREADS) {
            f32x4 a = x4[i];
            if (rr) a += r4[i];
            a -= mean;
            q += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
        }
    } else {
        for (int64_t i = threadIdx.x; i < len; i += LN_THREADS) {
            float d = xr[i] + (rr ? rr[i] : 0.f) - mean;
            q += d * d;
        }
    }
    const float var = block_sum<LN_WAVES>(q, red) / (float)len;
    const float rstd = 1.f / sqrtf(var + eps);
    if (threadIdx.x == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    float* yr = y + row * len;
    if (vec) {
        const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
        const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
        f32x4* y4 = reinterpret_cast<f32x4*>(yr);
        for (int64_t i = threadIdx.x; i < n4; i += LN_THREADS) {
            f32x4 a = x4[i];
            if (rr) a += r4[i];
            const f32x4 g = g4[i], bb = b4[i];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (a[e] - mean) * rstd * g[e] + bb[e];
                if (relu) t = fmaxf(t, 0.f);
                t *= keep_scale(seed, row * len + i * 4 + e, drop_p);
                v[e] = t;
            }
            y4[i] = v;
        }
    } else {
        for (int64_t i = threadIdx.x; i < len; i += LN_THREADS) {
            float v = (xr[i] + (rr ? rr[i] : 0.f) - mean) * rstd * gamma[i] + beta[i];
            if (relu) v = fmaxf(v, 0.f);
            v *= keep_scale(seed, row * len + i, drop_p);
            yr[i] = v;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat*mean(g*xhat)),  g = dy * mask * gamma ; dgamma += dy*mask*xhat ; dbeta += dy*mask
__global__ __launch_bounds__(LN_THREADS) void layernorm_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ y, const float* __restrict__ stats, float* __restrict__ dx, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int64_t len, int relu, float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_off, int atomic_params) {
    seed = fold_seed(seed, seed_off);
    __shared__ float red[LN_WAVES];
    const int64_t row = blockIdx.x;
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    const float* xr = x + row * len;
    const float* rr = res ? res + row * len : nullptr;
    const float* dyr = dy + row * len;
    const float* yr = y + row * len;
    const bool vec = (len & 3) == 0;
    const int64_t n4 = len >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(xr);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(rr);
    const f32x4* dy4 = reinterpret_cast<const f32x4*>(dyr);
    const f32x4* y4 = reinterpret_cast<const f32x4*>(yr);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    float s1 = 0.f, s2 = 0.f;
    if (vec) {
        for (int64_t i = threadIdx.x; i < n4; i += LN_THREADS) {
            f32x4 a = x4[i];
            if (rr) a += r4[i];
            const f32x4 d = dy4[i], yv = y4[i], gm = g4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (a[e] - mean) * rstd;
                float g = d[e] * keep_scale(seed, row * len + i * 4 + e, drop_p);
                if (relu && !(yv[e] > 0.f)) g = 0.f;
                const float gg = g * gm[e];
                s1 += gg;
                s2 += gg * xh;
                if (atomic_params) { atomicAdd(dgamma + i * 4 + e, g * xh); atomicAdd(dbeta + i * 4 + e, g); }
            }
        }
    } else {
        for (int64_t i = threadIdx.x; i < len; i += LN_THREADS) {
            float xh = (xr[i] + (rr ? rr[i] : 0.f) - mean) * rstd;
            float g = dyr[i] * keep_scale(seed, row * len + i, drop_p);
            if (relu && !(yr[i] > 0.f)) {
                // y == 0 either because ReLU clipped or because dropout zeroed a positive value; in the latter case the
                // keep-scale is already 0, so masking by (y > 0) is exact for both
                g = 0.f;
            }
            float gg = g * gamma[i];
            s1 += gg;
            s2 += gg * xh;
            if (atomic_params) { atomicAdd(dgamma + i, g * xh); atomicAdd(dbeta + i, g); }
        }
    }
    const float m1 = block_sum<LN_WAVES>(s1, red) / (float)len;
    const float m2 = block_sum<LN_WAVES>(s2, red) / (float)len;
    float* dxr = dx + row * len;
    if (vec) {
        f32x4* dx4 = reinterpret_cast<f32x4*>(dxr);
        for (int64_t i = threadIdx.x; i < n4; i += LN_THREADS) {
            f32x4 a = x4[i];
            if (rr) a += r4[i];
            const f32x4 d = dy4[i], yv = y4[i], gm = g4[i];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (a[e] - mean) * rstd;
                float g = d[e] * keep_scale(seed, row * len + i * 4 + e, drop_p);
                if (relu && !(yv[e] > 0.f)) g = 0.f;
                o[e] = rstd * (g * gm[e] - m1 - xh * m2);
            }
            dx4[i] = o;
        }
    } else {
        for (int64_t i = threadIdx.x; i < len; i += LN_THREADS) {
            float xh = (xr[i] + (rr ? rr[i] : 0.f) - mean) * rstd;
            float g = dyr[i] * keep_scale(seed, row * len + i, drop_p);
            if (relu && !(yr[i] > 0.f)) g = 0.f;
            dxr[i] = rstd * (g * gamma[i] - m1 - xh * m2);
        }
    }
}

// many-rows case (nl_emb_1): column reduction of dgamma/dbeta without len*rows atomics -- one thread per column chunk
__global__ void layernorm_param_grad_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ res,
                                            const float* __restrict__ y, const float* __restrict__ stats, float* __restrict__ dgamma,
                                            float* __restrict__ dbeta, int64_t rows, int64_t len, int relu, float drop_p,
                                            uint64_t seed, const uint64_t* __restrict__ seed_off, int64_t rows_per_block) {
    seed = fold_seed(seed, seed_off);
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= len) return;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float a = 0.f, b = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        int64_t i = r * len + col;
        float xh = (x[i] + (res ? res[i] : 0.f) - stats[r * 2]) * stats[r * 2 + 1];
        float g = dy[i] * keep_scale(seed, i, drop_p);
        if (relu && !(y[i] > 0.f)) g = 0.f;
        a += g * xh;
        b += g;
    }
    atomicAdd(dgamma + col, a);
    atomicAdd(dbeta + col, b);
}

}  // namespace

extern "C" {

int din_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps, float* y,
                      float* stats, int64_t rows, int64_t len, int relu, float drop_p, uint64_t seed, const uint64_t* seed_offset, void* stream) {
    DIN_REQUIRE(x && gamma && beta && y && stats, "layernorm_fwd: null pointer");
    DIN_REQUIRE(rows >= 0 && len > 0 && drop_p >= 0.f && drop_p < 1.f, "layernorm_fwd: bad argument");
    if (rows == 0) return DIN_OK;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)rows), dim3(LN_THREADS), 0, as_stream(stream), x, res, gamma, beta, eps, y,
                       stats, len, relu, drop_p, seed, seed_offset);
    DIN_CHECK_LAUNCH("layernorm_fwd");
    return DIN_OK;
}

int din_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma, const float* y,
                      const float* stats, float* dx, float* dgamma, float* dbeta, int64_t rows, int64_t len, int relu,
                      float drop_p, uint64_t seed, const uint64_t* seed_offset, void* stream) {
    DIN_REQUIRE(dy && x && gamma && y && stats && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
    DIN_REQUIRE(rows >= 0 && len > 0, "layernorm_bwd: bad argument");
    if (rows == 0) return DIN_OK;
    // few long rows (per-clip LN): per-element atomics from the row kernel are cheap (rows adds per address);
    // many short rows (nl_emb_1): dedicated column-reduction kernel
    const int atomic_params = rows <= 8 ? 1 : 0;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)rows), dim3(LN_THREADS), 0, as_stream(stream), dy, x, res, gamma, y, stats, dx,
                       dgamma, dbeta, len, relu, drop_p, seed, seed_offset, atomic_params);
    DIN_CHECK_LAUNCH("layernorm_bwd");
    if (!atomic_params) {
        int64_t rpb = 32;
        dim3 grid((unsigned)ceil_div64(len, 256), (unsigned)ceil_div64(rows, rpb));
        hipLaunchKernelGGL(layernorm_param_grad_kernel, grid, dim3(256), 0, as_stream(stream), dy, x, res, y, stats, dgamma, dbeta, rows, len,
                           relu, drop_p, seed, seed_offset, rpb);
        DIN_CHECK_LAUNCH("layernorm_param_grad");
    }
    return DIN_OK;
}

}  // extern "C"
