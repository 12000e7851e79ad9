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
// fp64 partial sums per thread, a fixed-order LDS sum per workgroup, one slab per workgroup summed in slab order: the statistics do not
// depend on M in precision and are bit-reproducible (no atomics anywhere).  torch.nn.functional.batch_norm is the arithmetic being replaced (reached from torchvision BasicConv2d).
#include "din_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

// Rows per workgroup of the statistics kernels.  The decomposition is a function of (`rows`, direction) ALONE (no environment switch in the
// shipped build, no device query): together with the fixed summation order below it makes the statistics bit-reproducible run to run and
// box to box.  Targets measured with tools/bn_bench.py on the trunk's views (profiles/r04_bn_bench.txt): the forward statistics want
// ~1024 workgroups (4.8 -> 5.9 TB/s on the stem layers, 2.6 -> 5.3 TB/s on the 43 x 78 maps, which 2048-row workgroups left at 158
// workgroups on 256 CUs); the backward statistics (two streams per row, twice the slab) are fastest at ~512.
constexpr int BN_PARTS_FWD = 1024, BN_PARTS_BWD = 512, BN_MIN_ROWS = 256;
static int bn_rows_per_block(int64_t rows, bool bwd) {
    int64_t maxp = bwd ? BN_PARTS_BWD : BN_PARTS_FWD, rpb = BN_MIN_ROWS;
#ifdef DIN_EXPERIMENTS
    if (const char* e = DIN_OPT("DIN_BN_RPB")) rpb = atoi(e);
    if (const char* e = bwd ? DIN_OPT("DIN_BN_PARTS_BWD") : DIN_OPT("DIN_BN_PARTS")) maxp = atoi(e);
#endif
    if ((rows + rpb - 1) / rpb > maxp) rpb = (rows + maxp - 1) / maxp;
    return (int)rpb;
}
static int bn_parts(int64_t rows, bool bwd = false) { return (int)ceil_div64(rows, bn_rows_per_block(rows, bwd)); }

// Non-temporal loads for the statistics kernels (the raw conv output / gradient maps are hundreds of MB, read once per pass): measured
// +10 % on the stem-sized views (5.9 -> 6.7 TB/s forward, 6.1 -> 6.9 backward; tools/bn_bench.py, profiles/r04_bn_bench.txt).  Non-temporal
// STORES lose on the narrow channel views of the concatenated block outputs (96-byte rows on a 576-byte pitch: 65 -> 125 us), so the apply
// kernels keep plain stores; their LOADS are non-temporal too (the strided gradient views of the concatenated block outputs gain 10-25 %:
// Mixed_5 apply 62 -> 50 / 132 -> 95 us, backward apply 103 -> 93 / 167 -> 153 us; the contiguous stem views are unchanged).
// DIN_BN_NT (build flag, experiments): 0 none, 1 statistics loads only, 2 statistics + apply-kernel loads (shipped).
#ifndef DIN_BN_NT
#define DIN_BN_NT 2
#endif
#define BN_LD(T, p) (*reinterpret_cast<const T*>(p))
#define BN_LD_NT(T, p) __builtin_nontemporal_load(reinterpret_cast<const T*>(p))
#define BN_ST(T, p, v) (*reinterpret_cast<T*>(p) = (v))
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int V = 4;
    template <bool NT = (DIN_BN_NT >= 2)>
    __device__ static void load(const float* p, float (&v)[4]) {
        const f32x4 x = NT ? BN_LD_NT(f32x4, p) : BN_LD(f32x4, p);
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    }
    __device__ static void store(float* p, const float (&v)[4]) { BN_ST(f32x4, p, (f32x4{v[0], v[1], v[2], v[3]})); }
};
template <> struct Vec<bf16_t> {
    static constexpr int V = 8;
    template <bool NT = (DIN_BN_NT >= 2)>
    __device__ static void load(const bf16_t* p, float (&v)[8]) {
        const u32x4 x = NT ? BN_LD_NT(u32x4, p) : BN_LD(u32x4, p);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(x[e] << 16); v[2 * e + 1] = __uint_as_float(x[e] & 0xffff0000u); }
    }
    __device__ static void store(bf16_t* p, const float (&v)[8]) {
        BN_ST(u32x4, p, (u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])}));
    }
};

// Statistics, DETERMINISTIC form (round 4; the round-3 kernel combined fp32 per-thread partial sums with LDS float atomics and fp64 global
// atomics: order-dependent bits, and a whole-model test with a zero-margin bound went red on the driver's box).
//   thread t of a 256-thread workgroup owns channel chunk t % cv and walks rows t / cv, + rpp, ... of the workgroup's row range
//   (cv = C / V chunks per row, rpp = 256 / cv rows per pass; threads beyond rpp * cv idle);
//   per-thread accumulators are FLOAT64: the products x * x of fp32 / bf16 inputs are exact in fp64 (24 + 24 significand bits), so
//   E[x^2] - mean^2 loses nothing to cancellation whatever |mean| / std is and no shift heuristic is needed (an optional shift is still
//   subtracted, exactly, for callers that pass one);
//   the workgroup combines its rpp row-lanes through LDS in lane order 0, 1, 2, ... (a fixed-order sum, no atomics) and writes ONE slab
//   part[blockIdx][2 C]; bn_reduce / bn_finalize add the slabs in block order.  Same input -> same bits, every run.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int ldx, int cxoff, const T* __restrict__ g, int ldg, int cgoff,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int64_t M, int C,
                                                       double* __restrict__ part, int rows_per_block) {
    constexpr int V = Vec<T>::V;
    extern __shared__ double red[];                                 // [rpp][2 C]
    const int cv = C / V, rpp = 256 / cv;
    const int tid = threadIdx.x, ch = tid % cv, rl = tid / cv;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    if (rl < rpp) {
        double s1[V], s2[V];
        float mu[V], rs[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { s1[e] = 0.0; s2[e] = 0.0; mu[e] = 0.f; rs[e] = 1.f; }
        if (BWD) {
#pragma unroll
            for (int e = 0; e < V; ++e) { mu[e] = mean[ch * V + e]; rs[e] = rstd[ch * V + e]; }
        } else if (mean) {
#pragma unroll
            for (int e = 0; e < V; ++e) mu[e] = mean[ch * V + e];
        }
        auto fold = [&](const float (&xv)[V], const float (&gv)[V]) {
            if (BWD) {                                              // xhat is the fp32 expression of bn_bwd_apply; its product with gz is exact in fp64
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const double gd = (double)gv[e];
                    s1[e] += gd;
                    s2[e] = fma(gd, (double)((xv[e] - mu[e]) * rs[e]), s2[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) { const double dv = (double)xv[e] - (double)mu[e]; s1[e] += dv; s2[e] = fma(dv, dv, s2[e]); }
            }
        };
        int64_t r = r0 + rl;
        // four rows per trip: their loads are independent and stay in flight together
        for (; r + 3 * (int64_t)rpp < r1; r += 4 * (int64_t)rpp) {
            float xv[4][V], gv[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Vec<T>::template load<(DIN_BN_NT >= 1)>(x + (r + u * (int64_t)rpp) * ldx + cxoff + ch * V, xv[u]);
                if (BWD) Vec<T>::template load<(DIN_BN_NT >= 1)>(g + (r + u * (int64_t)rpp) * ldg + cgoff + ch * V, gv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) fold(xv[u], gv[u]);
        }
        for (; r < r1; r += rpp) {
            float xv[V], gv[V];
            Vec<T>::template load<(DIN_BN_NT >= 1)>(x + r * ldx + cxoff + ch * V, xv);
            if (BWD) Vec<T>::template load<(DIN_BN_NT >= 1)>(g + r * ldg + cgoff + ch * V, gv);
            fold(xv, gv);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) { red[rl * 2 * C + ch * V + e] = s1[e]; red[rl * 2 * C + C + ch * V + e] = s2[e]; }
    }
    __syncthreads();
    double* out = part + (int64_t)blockIdx.x * 2 * C;
    for (int i = tid; i < 2 * C; i += 256) {
        double s = red[i];
        for (int l = 1; l < rpp; ++l) s += red[l * 2 * C + i];      // lane order: fixed
        out[i] = s;
    }
}

// Fixed-order column sums of a slab stack part[nparts][n] -> out[n].  A workgroup owns 16 columns; its 16 segments each add a contiguous run
// of slabs in slab order, then segment 0 adds the 16 segment sums in segment order.  (nparts, n) -> one summation tree, whatever the grid.
__device__ __forceinline__ double bn_column_sum(const double* __restrict__ part, int nparts, int n, int col, int seg, double* lds /*[16][16]*/) {
    const int per = (nparts + 15) / 16;
    int p0 = seg * per, p1 = p0 + per;
    if (p1 > nparts) p1 = nparts;
    double s = 0.0;
    if (col < n)
        for (int p = p0; p < p1; ++p) s += part[(int64_t)p * n + col];
    lds[seg * 16 + (threadIdx.x & 15)] = s;
    __syncthreads();
    double t = 0.0;
    if (seg == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) t += lds[k * 16 + (threadIdx.x & 15)];
    }
    __syncthreads();
    return t;                                                       // valid in segment 0
}

__global__ __launch_bounds__(256) void bn_reduce_kernel(const double* __restrict__ part, int nparts, int n, double* __restrict__ out) {
    __shared__ double lds[256];
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), seg = threadIdx.x >> 4;
    const double t = bn_column_sum(part, nparts, n, col, seg, lds);
    if (seg == 0 && col < n) out[col] = t;
}

// ws = [2 C reduced sums][nparts slabs of 2 C]: reduce (nparts > 0) in fixed order, then the per-channel constants
__global__ __launch_bounds__(256) void bn_finalize_kernel(double* __restrict__ ws, int nparts, int64_t M, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ a, float* __restrict__ b, float* __restrict__ mean,
                                   float* __restrict__ rstd, const float* __restrict__ shift) {
    __shared__ double lds[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), seg = threadIdx.x >> 4;
    double q1, q2;
    if (nparts > 0) {
        q1 = bn_column_sum(ws + 2 * C, nparts, 2 * C, c < C ? c : 2 * C, seg, lds);
        q2 = bn_column_sum(ws + 2 * C, nparts, 2 * C, c < C ? C + c : 2 * C, seg, lds);
        if (seg == 0 && c < C) { ws[c] = q1; ws[C + c] = q2; }
    } else if (c < C) {
        q1 = ws[c]; q2 = ws[C + c];
    }
    if (seg != 0 || c >= C) return;
    const double ms = q1 / (double)M;                               // mean of (x - shift)
    const double mu = ms + (shift ? (double)shift[c] : 0.0);
    double var = q2 / (double)M - ms * ms;                          // biased (normalisation) variance
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

static bool bn_apply_rows() { const char* e = DIN_OPT("DIN_BN_APPLY_ROWS"); return !(e && atoi(e) == 0); }     // read per launch, like every option
// rows per workgroup of the row-walk apply kernels: ~2048 workgroups, at least four passes of the workgroup's 256 / (C / V) rows each
static int bn_apply_rpb(int64_t rows, int c, int v) {
    const int rpp = 256 / (c / v);
    int64_t wgs = 2048;
#ifdef DIN_EXPERIMENTS
    if (const char* e = DIN_OPT("DIN_BN_APPLY_WGS")) wgs = atoi(e);
#endif
    int64_t rpb = (rows + wgs - 1) / wgs;
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

int din_bn_parts(int64_t rows) { return rows > 0 ? bn_parts(rows) : 0; }

int64_t din_bn_workspace(int64_t rows, int c) {
    if (rows <= 0 || c <= 0) return 0;
    const int parts = bn_parts(rows, false) > bn_parts(rows, true) ? bn_parts(rows, false) : bn_parts(rows, true);
    return (int64_t)(parts + 1) * 2 * c * (int64_t)sizeof(double);
}

int din_bn_stats(const void* x, int dtype, int64_t rows, int c, int ld, int coff, const float* shift, double* ws, void* stream) {
    DIN_REQUIRE(x && ws, "bn_stats: null pointer");
    if (int e = check_view(dtype, rows, c, ld, coff, "bn_stats")) return e;
    const int rpb = bn_rows_per_block(rows, false);
    const int blocks = (int)ceil_div64(rows, rpb);
    const int v = dtype == DIN_F32 ? 4 : 8;
    const size_t lds = (size_t)(256 / (c / v)) * 2 * c * sizeof(double);
    double* part = ws + 2 * c;
    if (dtype == DIN_F32)
        hipLaunchKernelGGL((bn_stats_kernel<float, false>), dim3(blocks), dim3(256), lds, as_stream(stream), (const float*)x, ld, coff,
                           (const float*)nullptr, 0, 0, shift, (const float*)nullptr, rows, c, part, rpb);
    else
        hipLaunchKernelGGL((bn_stats_kernel<bf16_t, false>), dim3(blocks), dim3(256), lds, as_stream(stream), (const bf16_t*)x, ld, coff,
                           (const bf16_t*)nullptr, 0, 0, shift, (const float*)nullptr, rows, c, part, rpb);
    DIN_CHECK_LAUNCH("bn_stats");
    return DIN_OK;
}

int din_bn_reduce(double* ws, int nparts, int c, void* stream) {
    DIN_REQUIRE(ws && nparts > 0 && c > 0, "bn_reduce: bad argument");
    hipLaunchKernelGGL(bn_reduce_kernel, dim3((2 * c + 15) / 16), dim3(256), 0, as_stream(stream), (const double*)(ws + 2 * c), nparts, 2 * c, ws);
    DIN_CHECK_LAUNCH("bn_reduce");
    return DIN_OK;
}

int din_bn_finalize(double* ws, int nparts, int64_t rows, int c, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, float* a, float* b, float* mean, float* rstd, const float* shift, void* stream) {
    DIN_REQUIRE(ws && gamma && beta && a && b && mean && rstd && rows > 0 && c > 0 && nparts >= 0, "bn_finalize: bad argument");
    DIN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running_mean and running_var go together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 15) / 16), dim3(256), 0, as_stream(stream), ws, nparts, rows, c, gamma, beta, eps, momentum,
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
                     const float* mean, const float* rstd, double* ws, void* stream) {
    DIN_REQUIRE(gz && x && mean && rstd && ws, "bn_bwd_stats: null pointer");
    if (int e = check_view(dtype, rows, c, ldg, cgoff, "bn_bwd_stats(gz)")) return e;
    if (int e = check_view(dtype, rows, c, ldx, cxoff, "bn_bwd_stats(x)")) return e;
    const int rpb = bn_rows_per_block(rows, true);
    const int blocks = (int)ceil_div64(rows, rpb);
    const int v = dtype == DIN_F32 ? 4 : 8;
    const size_t lds = (size_t)(256 / (c / v)) * 2 * c * sizeof(double);
    double* part = ws + 2 * c;
    if (dtype == DIN_F32)
        hipLaunchKernelGGL((bn_stats_kernel<float, true>), dim3(blocks), dim3(256), lds, as_stream(stream), (const float*)x, ldx, cxoff,
                           (const float*)gz, ldg, cgoff, mean, rstd, rows, c, part, rpb);
    else
        hipLaunchKernelGGL((bn_stats_kernel<bf16_t, true>), dim3(blocks), dim3(256), lds, as_stream(stream), (const bf16_t*)x, ldx, cxoff,
                           (const bf16_t*)gz, ldg, cgoff, mean, rstd, rows, c, part, rpb);
    DIN_CHECK_LAUNCH("bn_bwd_stats");
    return din_bn_reduce(ws, blocks, c, stream);
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
