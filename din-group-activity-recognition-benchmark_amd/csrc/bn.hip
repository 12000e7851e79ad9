// Batch-statistics BatchNorm for the NHWC conv graph (gfx950).
//
// The reference trains the Inception-v3 backbone with model.train() and cfg.set_bn_eval = False (train_net_dynamic.py:98-100,170-172,
// config.py:80): every BasicConv2d normalises with the statistics of the current B*T frames and updates running_mean / running_var.
// Folding BatchNorm into the filters (din_bn_fold*) only covers the running-statistics mode; these kernels are the other one:
//   forward : raw conv output y [M][C]  ->  per-channel sum / sum of squares (din_bn_stats)  ->  mean, rstd, running statistics,
//             a = gamma * rstd, b = beta - mean * a (din_bn_finalize)  ->  z = relu(a * y + b) written into the consumer's view (din_bn_apply)
//   backward: masked gradient gz and y  ->  s1 = sum gz, s2 = sum gz * yhat (din_bn_bwd_stats)  ->
//             dy = gamma * rstd * (gz - s1 / M - yhat * s2 / M), dgamma = s2, dbeta = s1 (din_bn_bwd_apply)
// All four passes are HBM-bound streams over [M][C] views (pixel stride ld, channel offset coff): 16-byte lanes along the channels,
// fp32 partial sums per thread, one fp64 atomic per channel per workgroup (native global_atomic_add_f64), so the statistics do not
// depend on M in precision.  torch.nn.functional.batch_norm is the arithmetic being replaced (reached from torchvision BasicConv2d).
#include "din_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BN_ROWS_PER_BLOCK = 2048;
// Rows per workgroup of the statistics kernels: 2048, but at most BN_MAX_BLOCKS workgroups (every workgroup ends with 2 C fp64 atomics on the
// same few cache lines, which L2 serialises: tools/colsum_probe.py).  Measured on the batch-statistics step (bench.py --bn-mode batch, one box):
// four rows per trip instead of one 93.1 -> 88.7 ms (the loads of a trip are independent: more bytes in flight), the cap on top of it
// 88.7 -> 87.9-88.5 ms (1024; 256 .. 2048 within noise).  DIN_BN_MAX_BLOCKS: tuning aid (0 = no cap).
static int bn_rows_per_block(int64_t rows) {
    static const int cap = getenv("DIN_BN_MAX_BLOCKS") ? atoi(getenv("DIN_BN_MAX_BLOCKS")) : 1024;
    int64_t rpb = BN_ROWS_PER_BLOCK;
    if (cap > 0 && (rows + rpb - 1) / rpb > cap) rpb = (rows + cap - 1) / cap;
    return (int)rpb;
}

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int V = 4;
    __device__ static void load(const float* p, float (&v)[4]) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p);
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    }
    __device__ static void store(float* p, const float (&v)[4]) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct Vec<bf16_t> {
    static constexpr int V = 8;
    __device__ static void load(const bf16_t* p, float (&v)[8]) {
        const u32x4 x = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(x[e] << 16); v[2 * e + 1] = __uint_as_float(x[e] & 0xffff0000u); }
    }
    __device__ static void store(bf16_t* p, const float (&v)[8]) {
        *reinterpret_cast<u32x4*>(p) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
};

// thread t of a 256-thread workgroup owns channel chunk t % cv and walks rows t / cv, + rpp, ... of the workgroup's row range
// (cv = C / V chunks per row, rpp = 256 / cv rows per pass; threads beyond rpp * cv idle)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int ldx, int cxoff, const T* __restrict__ g, int ldg, int cgoff,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int64_t M, int C,
                                                       double* __restrict__ sums, int rows_per_block) {
    constexpr int V = Vec<T>::V;
    extern __shared__ float red[];                                  // [2][C]
    const int cv = C / V, rpp = 256 / cv;
    const int tid = threadIdx.x, ch = tid % cv, rl = tid / cv;
    for (int i = tid; i < 2 * C; i += 256) red[i] = 0.f;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s1[V], s2[V], mu[V], rs[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s1[e] = 0.f; s2[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f; }
    if (rl < rpp) {
        if (BWD) {
#pragma unroll
            for (int e = 0; e < V; ++e) { mu[e] = mean[ch * V + e]; rs[e] = rstd[ch * V + e]; }
        } else if (mean) {                                          // forward: sums of (x - shift), shift = the running mean before its update:
#pragma unroll                                                      // E[(x-s)^2] - E[x-s]^2 does not cancel when |mean| >> std
            for (int e = 0; e < V; ++e) mu[e] = mean[ch * V + e];
        }
        auto fold = [&](const float (&xv)[V], const float (&gv)[V]) {
            if (BWD) {
#pragma unroll
                for (int e = 0; e < V; ++e) { s1[e] += gv[e]; s2[e] += gv[e] * ((xv[e] - mu[e]) * rs[e]); }
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) { const float dv = xv[e] - mu[e]; s1[e] += dv; s2[e] += dv * dv; }
            }
        };
        int64_t r = r0 + rl;
        // four rows per trip: their loads are independent and stay in flight together (a workgroup now walks up to M / 1024 rows)
        for (; r + 3 * (int64_t)rpp < r1; r += 4 * (int64_t)rpp) {
            float xv[4][V], gv[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Vec<T>::load(x + (r + u * (int64_t)rpp) * ldx + cxoff + ch * V, xv[u]);
                if (BWD) Vec<T>::load(g + (r + u * (int64_t)rpp) * ldg + cgoff + ch * V, gv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) fold(xv[u], gv[u]);
        }
        for (; r < r1; r += rpp) {
            float xv[V], gv[V];
            Vec<T>::load(x + r * ldx + cxoff + ch * V, xv);
            if (BWD) Vec<T>::load(g + r * ldg + cgoff + ch * V, gv);
            fold(xv, gv);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) { atomicAdd(&red[ch * V + e], s1[e]); atomicAdd(&red[C + ch * V + e], s2[e]); }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) atomicAdd(&sums[i], (double)red[i]);
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, int64_t M, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ a, float* __restrict__ b, float* __restrict__ mean,
                                   float* __restrict__ rstd, const float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double ms = sums[c] / (double)M;                          // mean of (x - shift)
    const double mu = ms + (shift ? (double)shift[c] : 0.0);
    double var = sums[C + c] / (double)M - ms * ms;                 // biased (normalisation) variance
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)mu; rstd[c] = r;
    const float av = gamma[c] * r;
    a[c] = av; b[c] = beta[c] - (float)mu * av;
    if (running_mean) {                                             // torch: running = (1 - momentum) * running + momentum * batch (unbiased var)
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, int ldx, int cxoff, const float* __restrict__ a,
                                                       const float* __restrict__ b, int relu, T* __restrict__ y, int ldy, int cyoff,
                                                       int64_t M, int C) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V;
    const int64_t total = M * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cv;
        const int ch = (int)(i - r * cv);
        float v[V];
        Vec<T>::load(x + r * ldx + cxoff + ch * V, v);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float z = v[e] * a[ch * V + e] + b[ch * V + e];
            v[e] = relu ? fmaxf(z, 0.f) : z;
        }
        Vec<T>::store(y + r * ldy + cyoff + ch * V, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ g, int ldg, int cgoff, const T* __restrict__ x, int ldx, int cxoff,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const double* __restrict__ sums,
                                                           T* __restrict__ dy, int ldy, int cyoff, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int64_t M, int C) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V;
    const int64_t total = M * cv;
    const float invM = 1.f / (float)M;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) { dbeta[c] = (float)sums[c]; dgamma[c] = (float)sums[C + c]; }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cv;
        const int ch = (int)(i - r * cv);
        float gv[V], xv[V];
        Vec<T>::load(g + r * ldg + cgoff + ch * V, gv);
        Vec<T>::load(x + r * ldx + cxoff + ch * V, xv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int c = ch * V + e;
            const float xh = (xv[e] - mean[c]) * rstd[c];
            gv[e] = gamma[c] * rstd[c] * (gv[e] - (float)sums[c] * invM - xh * ((float)sums[C + c] * invM));
        }
        Vec<T>::store(dy + r * ldy + cyoff + ch * V, gv);
    }
}

// Row-walk forms of the two apply kernels (default; DIN_BN_APPLY_ROWS=0 restores the element-per-iteration kernels above).  Thread t of a workgroup
// owns channel chunk t % cv for the whole launch, so the per-channel constants (fp64 sums converted, gamma * rstd, ...) are formed ONCE instead of
// per element, there is no 64-bit division per element, and four rows' loads are in flight per trip.  The arithmetic per element is the expression
// of the kernels above with the same operands in the same order: same bits.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_rows_kernel(const T* __restrict__ x, int ldx, int cxoff, const float* __restrict__ a,
                                                            const float* __restrict__ b, int relu, T* __restrict__ y, int ldy, int cyoff,
                                                            int64_t M, int C, int rows_per_block) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V, rpp = 256 / cv;
    const int ch = threadIdx.x % cv, rl = threadIdx.x / cv;
    if (rl >= rpp) return;
    float av[V], bv[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { av[e] = a[ch * V + e]; bv[e] = b[ch * V + e]; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    const T* xs = x + cxoff + ch * V;
    T* ys = y + cyoff + ch * V;
    auto one = [&](float (&v)[V]) {
#pragma unroll
        for (int e = 0; e < V; ++e) { const float z = v[e] * av[e] + bv[e]; v[e] = relu ? fmaxf(z, 0.f) : z; }
    };
    int64_t r = r0 + rl;
    for (; r + 3 * (int64_t)rpp < r1; r += 4 * (int64_t)rpp) {
        float v[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) Vec<T>::load(xs + (r + u * (int64_t)rpp) * ldx, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) { one(v[u]); Vec<T>::store(ys + (r + u * (int64_t)rpp) * ldy, v[u]); }
    }
    for (; r < r1; r += rpp) {
        float v[V];
        Vec<T>::load(xs + r * ldx, v);
        one(v);
        Vec<T>::store(ys + r * ldy, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(const T* __restrict__ g, int ldg, int cgoff, const T* __restrict__ x, int ldx, int cxoff,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const double* __restrict__ sums,
                                                                T* __restrict__ dy, int ldy, int cyoff, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int64_t M, int C, int rows_per_block) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V, rpp = 256 / cv;
    const float invM = 1.f / (float)M;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) { dbeta[c] = (float)sums[c]; dgamma[c] = (float)sums[C + c]; }
    const int ch = threadIdx.x % cv, rl = threadIdx.x / cv;
    if (rl >= rpp) return;
    float mu[V], rs[V], gr[V], k1[V], k2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int c = ch * V + e;
        mu[e] = mean[c]; rs[e] = rstd[c]; gr[e] = gamma[c] * rstd[c];
        k1[e] = (float)sums[c] * invM; k2[e] = (float)sums[C + c] * invM;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    const T* gs = g + cgoff + ch * V;
    const T* xs = x + cxoff + ch * V;
    T* ds = dy + cyoff + ch * V;
    auto one = [&](float (&gv)[V], const float (&xv)[V]) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float xh = (xv[e] - mu[e]) * rs[e];
            gv[e] = gr[e] * (gv[e] - k1[e] - xh * k2[e]);
        }
    };
    int64_t r = r0 + rl;
    for (; r + 3 * (int64_t)rpp < r1; r += 4 * (int64_t)rpp) {
        float gv[4][V], xv[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Vec<T>::load(gs + (r + u * (int64_t)rpp) * ldg, gv[u]);
            Vec<T>::load(xs + (r + u * (int64_t)rpp) * ldx, xv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { one(gv[u], xv[u]); Vec<T>::store(ds + (r + u * (int64_t)rpp) * ldy, gv[u]); }
    }
    for (; r < r1; r += rpp) {
        float gv[V], xv[V];
        Vec<T>::load(gs + r * ldg, gv);
        Vec<T>::load(xs + r * ldx, xv);
        one(gv, xv);
        Vec<T>::store(ds + r * ldy, gv);
    }
}

static bool bn_apply_rows() { static const bool on = !(getenv("DIN_BN_APPLY_ROWS") && atoi(getenv("DIN_BN_APPLY_ROWS")) == 0); return on; }
// rows per workgroup of the row-walk apply kernels: ~2048 workgroups, at least four passes of the workgroup's 256 / (C / V) rows each
static int bn_apply_rpb(int64_t rows, int c, int v) {
    const int rpp = 256 / (c / v);
    int64_t rpb = (rows + 2047) / 2048;
    if (rpb < 4 * rpp) rpb = 4 * rpp;
    if (rpb > 8192) rpb = 8192;
    return (int)rpb;
}

int check_view(int dtype, int64_t rows, int c, int ld, int coff, const char* what) {
    DIN_REQUIRE(dtype == DIN_F32 || dtype == DIN_BF16, "%s: bad dtype", what);
    const int v = dtype == DIN_F32 ? 4 : 8;
    DIN_REQUIRE(rows > 0 && c > 0 && c % v == 0 && c / v <= 256, "%s: channel count %d must be a multiple of %d (at most %d)", what, c, v, 256 * v);
    DIN_REQUIRE(ld % v == 0 && coff % v == 0 && coff >= 0 && ld >= coff + c, "%s: pixel stride / channel offset must be multiples of %d", what, v);
    return DIN_OK;
}

}  // namespace

extern "C" {

int din_bn_stats(const void* x, int dtype, int64_t rows, int c, int ld, int coff, const float* shift, double* sums, void* stream) {
    DIN_REQUIRE(x && sums, "bn_stats: null pointer");
    if (int e = check_view(dtype, rows, c, ld, coff, "bn_stats")) return e;
    const int rpb = bn_rows_per_block(rows);
    const int blocks = (int)ceil_div64(rows, rpb);
    const size_t lds = 2 * (size_t)c * sizeof(float);
    if (dtype == DIN_F32)
        hipLaunchKernelGGL((bn_stats_kernel<float, false>), dim3(blocks), dim3(256), lds, as_stream(stream), (const float*)x, ld, coff,
                           (const float*)nullptr, 0, 0, shift, (const float*)nullptr, rows, c, sums, rpb);
    else
        hipLaunchKernelGGL((bn_stats_kernel<bf16_t, false>), dim3(blocks), dim3(256), lds, as_stream(stream), (const bf16_t*)x, ld, coff,
                           (const bf16_t*)nullptr, 0, 0, shift, (const float*)nullptr, rows, c, sums, rpb);
    DIN_CHECK_LAUNCH("bn_stats");
    return DIN_OK;
}

int din_bn_finalize(const double* sums, int64_t rows, int c, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, float* a, float* b, float* mean, float* rstd, const float* shift, void* stream) {
    DIN_REQUIRE(sums && gamma && beta && a && b && mean && rstd && rows > 0 && c > 0, "bn_finalize: bad argument");
    DIN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running_mean and running_var go together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), sums, rows, c, gamma, beta, eps, momentum,
                       running_mean, running_var, a, b, mean, rstd, shift);
    DIN_CHECK_LAUNCH("bn_finalize");
    return DIN_OK;
}

int din_bn_apply(const void* x, int dtype, int64_t rows, int c, int ldx, int cxoff, const float* a, const float* b, int relu, void* y,
                 int ldy, int cyoff, void* stream) {
    DIN_REQUIRE(x && y && a && b, "bn_apply: null pointer");
    if (int e = check_view(dtype, rows, c, ldx, cxoff, "bn_apply(x)")) return e;
    if (int e = check_view(dtype, rows, c, ldy, cyoff, "bn_apply(y)")) return e;
    const int v = dtype == DIN_F32 ? 4 : 8;
    if (bn_apply_rows()) {
        const int rpb = bn_apply_rpb(rows, c, v), nblk = (int)ceil_div64(rows, rpb);
        if (dtype == DIN_F32)
            hipLaunchKernelGGL(bn_apply_rows_kernel<float>, dim3(nblk), dim3(256), 0, as_stream(stream), (const float*)x, ldx, cxoff, a, b, relu,
                               (float*)y, ldy, cyoff, rows, c, rpb);
        else
            hipLaunchKernelGGL(bn_apply_rows_kernel<bf16_t>, dim3(nblk), dim3(256), 0, as_stream(stream), (const bf16_t*)x, ldx, cxoff, a, b, relu,
                               (bf16_t*)y, ldy, cyoff, rows, c, rpb);
        DIN_CHECK_LAUNCH("bn_apply");
        return DIN_OK;
    }
    const int blocks = grid_1d(rows * (c / v), 256, 256 * 16);
    if (dtype == DIN_F32)
        hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)x, ldx, cxoff, a, b, relu,
                           (float*)y, ldy, cyoff, rows, c);
    else
        hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)x, ldx, cxoff, a, b, relu,
                           (bf16_t*)y, ldy, cyoff, rows, c);
    DIN_CHECK_LAUNCH("bn_apply");
    return DIN_OK;
}

int din_bn_bwd_stats(const void* gz, int ldg, int cgoff, const void* x, int ldx, int cxoff, int dtype, int64_t rows, int c,
                     const float* mean, const float* rstd, double* sums, void* stream) {
    DIN_REQUIRE(gz && x && mean && rstd && sums, "bn_bwd_stats: null pointer");
    if (int e = check_view(dtype, rows, c, ldg, cgoff, "bn_bwd_stats(gz)")) return e;
    if (int e = check_view(dtype, rows, c, ldx, cxoff, "bn_bwd_stats(x)")) return e;
    const int rpb = bn_rows_per_block(rows);
    const int blocks = (int)ceil_div64(rows, rpb);
    const size_t lds = 2 * (size_t)c * sizeof(float);
    if (dtype == DIN_F32)
        hipLaunchKernelGGL((bn_stats_kernel<float, true>), dim3(blocks), dim3(256), lds, as_stream(stream), (const float*)x, ldx, cxoff,
                           (const float*)gz, ldg, cgoff, mean, rstd, rows, c, sums, rpb);
    else
        hipLaunchKernelGGL((bn_stats_kernel<bf16_t, true>), dim3(blocks), dim3(256), lds, as_stream(stream), (const bf16_t*)x, ldx, cxoff,
                           (const bf16_t*)gz, ldg, cgoff, mean, rstd, rows, c, sums, rpb);
    DIN_CHECK_LAUNCH("bn_bwd_stats");
    return DIN_OK;
}

int din_bn_bwd_apply(const void* gz, int ldg, int cgoff, const void* x, int ldx, int cxoff, int dtype, int64_t rows, int c,
                     const float* gamma, const float* mean, const float* rstd, const double* sums, void* dy, int ldy, int cyoff,
                     float* dgamma, float* dbeta, void* stream) {
    DIN_REQUIRE(gz && x && gamma && mean && rstd && sums && dy && dgamma && dbeta, "bn_bwd_apply: null pointer");
    if (int e = check_view(dtype, rows, c, ldg, cgoff, "bn_bwd_apply(gz)")) return e;
    if (int e = check_view(dtype, rows, c, ldx, cxoff, "bn_bwd_apply(x)")) return e;
    if (int e = check_view(dtype, rows, c, ldy, cyoff, "bn_bwd_apply(dy)")) return e;
    const int v = dtype == DIN_F32 ? 4 : 8;
    if (bn_apply_rows()) {
        const int rpb = bn_apply_rpb(rows, c, v), nblk = (int)ceil_div64(rows, rpb);
        if (dtype == DIN_F32)
            hipLaunchKernelGGL(bn_bwd_apply_rows_kernel<float>, dim3(nblk), dim3(256), 0, as_stream(stream), (const float*)gz, ldg, cgoff,
                               (const float*)x, ldx, cxoff, gamma, mean, rstd, sums, (float*)dy, ldy, cyoff, dgamma, dbeta, rows, c, rpb);
        else
            hipLaunchKernelGGL(bn_bwd_apply_rows_kernel<bf16_t>, dim3(nblk), dim3(256), 0, as_stream(stream), (const bf16_t*)gz, ldg, cgoff,
                               (const bf16_t*)x, ldx, cxoff, gamma, mean, rstd, sums, (bf16_t*)dy, ldy, cyoff, dgamma, dbeta, rows, c, rpb);
        DIN_CHECK_LAUNCH("bn_bwd_apply");
        return DIN_OK;
    }
    const int blocks = grid_1d(rows * (c / v), 256, 256 * 16);
    if (dtype == DIN_F32)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)gz, ldg, cgoff,
                           (const float*)x, ldx, cxoff, gamma, mean, rstd, sums, (float*)dy, ldy, cyoff, dgamma, dbeta, rows, c);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16_t*)gz, ldg, cgoff,
                           (const bf16_t*)x, ldx, cxoff, gamma, mean, rstd, sums, (bf16_t*)dy, ldy, cyoff, dgamma, dbeta, rows, c);
    DIN_CHECK_LAUNCH("bn_bwd_apply");
    return DIN_OK;
}

}  // extern "C"
