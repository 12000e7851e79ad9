// Context-encoding transformer of Dynamic_TCE_volleyball (reference infer_model.py:404-410 and
// infer_module/TCE_STBiP_module.py:252-286) for gfx950: every box embedding attends over the pixels of its own frame's context map.
//
//   keys = values = kf[bt][p][h*C + c]   (1x1 conv 512 -> heads*C of the context map + position embedding; produced by the conv kernels)
//   q[bt][n][h*C + c]                    (Linear NFB -> heads*C of the box embeddings)
//   S[bt][h][n][p] = <q[bt][n][h], kf[bt][p][h]>;  A = softmax_p(S);  ctx[bt][n][h*C + c] = sum_p A[bt][h][n][p] kf[bt][p][h*C + c]
//
// N is 12 (asserted by the reference, TCE_STBiP_module.py:263), C = 128, P = OH*OW (3600 at 720x1280): 2*N*P*C flops per (frame, head)
// against P*C*4 bytes of keys -- a few FLOP per byte, i.e. HBM-bound streaming of kf (0.7 GB at BT = 96).  So: no MFMA; each kernel reads
// kf exactly once with coalesced or whole-line accesses, the N queries / probabilities of the workgroup sit in LDS (broadcast reads).
// fp32 throughout.  Backward:
//   dA = <dctx, kf> (the scores kernel again), dS = A * (dA - sum_p A dA) (row kernel), dq = sum_p dS kf (the apply kernel again),
//   dkf[bt][p][h*C + c] = sum_n A[n][p] dctx[n][c] + dS[n][p] q[n][c]   (keys_grad kernel, one coalesced write per element)
#include "din_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAXN = 16;                                              // queries (boxes) per frame

// ---- S[bt][h][n][p] = sum_c q[bt][n][h*C+c] * kf[bt][p][h*C+c]: thread = pixel (reads its 4*C-byte key row), q in LDS ----------------
__global__ __launch_bounds__(256) void ctx_scores_kernel(const float* __restrict__ q, const float* __restrict__ kf, float* __restrict__ s,
                                                         int n, int p, int heads, int c) {
    extern __shared__ float qs[];                                      // [n][c]
    const int h = blockIdx.y, bt = blockIdx.z;
    const int hc = heads * c;
    for (int i = threadIdx.x; i < n * c; i += blockDim.x) qs[i] = q[((int64_t)bt * n + i / c) * hc + h * c + (i % c)];
    __syncthreads();
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= p) return;
    float acc[MAXN];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) acc[i] = 0.f;
    const float* kr = kf + ((int64_t)bt * p + pix) * hc + h * c;
    for (int cc = 0; cc < c; cc += 4) {
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(kr + cc);
#pragma unroll
        for (int i = 0; i < MAXN; ++i) {
            if (i < n) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(qs + i * c + cc);
                acc[i] += q4[0] * k4[0] + q4[1] * k4[1] + q4[2] * k4[2] + q4[3] * k4[3];
            }
        }
    }
    float* dst = s + (((int64_t)bt * heads + h) * n) * p + pix;
#pragma unroll
    for (int i = 0; i < MAXN; ++i)
        if (i < n) dst[(int64_t)i * p] = acc[i];
}

// ---- row softmax in place (rows of length len), and its backward ds = a * (da - sum(a * da)) in place on da ------------------------
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int len) {
    __shared__ float red[4];
    float* row = s + (int64_t)blockIdx.x * len;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < len; i += blockDim.x) m = fmaxf(m, row[i]);
    m = block_reduce(m, red, true);
    float z = 0.f;
    for (int i = threadIdx.x; i < len; i += blockDim.x) { const float e = expf(row[i] - m); row[i] = e; z += e; }
    z = block_reduce(z, red, false);
    const float inv = 1.f / z;
    for (int i = threadIdx.x; i < len; i += blockDim.x) row[i] *= inv;
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ a, float* __restrict__ da, int len) {
    __shared__ float red[4];
    const float* ar = a + (int64_t)blockIdx.x * len;
    float* dr = da + (int64_t)blockIdx.x * len;
    float dot = 0.f;
    for (int i = threadIdx.x; i < len; i += blockDim.x) dot += ar[i] * dr[i];
    dot = block_reduce(dot, red, false);
    for (int i = threadIdx.x; i < len; i += blockDim.x) dr[i] = ar[i] * (dr[i] - dot);
}

// ---- out[bt][n][h*C+c] = sum_p a[bt][h][n][p] * kf[bt][p][h*C+c]: workgroup = (frame, head); thread = (channel, pixel group) ------------
// 64-pixel chunks of the n probability rows go through LDS (broadcast reads); a group's partial sums meet in LDS at the end.
__global__ __launch_bounds__(256) void ctx_apply_kernel(const float* __restrict__ a, const float* __restrict__ kf, float* __restrict__ out,
                                                        int n, int p, int heads, int c) {
    extern __shared__ float sm[];                                      // [n][64] chunk of a, then [groups][n][c] partials
    const int h = blockIdx.x, bt = blockIdx.y;
    const int hc = heads * c, groups = blockDim.x / c;
    const int cc = threadIdx.x % c, g = threadIdx.x / c;
    const float* ab = a + (((int64_t)bt * heads + h) * n) * p;
    const float* kb = kf + (int64_t)bt * p * hc + h * c + cc;
    float acc[MAXN];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) acc[i] = 0.f;
    for (int p0 = 0; p0 < p; p0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < n * 64; i += blockDim.x) {
            const int r = i >> 6, j = i & 63;
            sm[i] = p0 + j < p ? ab[(int64_t)r * p + p0 + j] : 0.f;
        }
        __syncthreads();
        const int jend = min(64, p - p0);
        for (int j = g; j < jend; j += groups) {
            const float kv = kb[(int64_t)(p0 + j) * hc];
#pragma unroll
            for (int i = 0; i < MAXN; ++i)
                if (i < n) acc[i] += sm[i * 64 + j] * kv;
        }
    }
    __syncthreads();
    float* part = sm;                                                  // [groups][n][c]
#pragma unroll
    for (int i = 0; i < MAXN; ++i)
        if (i < n) part[(g * n + i) * c + cc] = acc[i];
    __syncthreads();
    for (int i = threadIdx.x; i < n * c; i += blockDim.x) {
        float v = 0.f;
        for (int gg = 0; gg < groups; ++gg) v += part[gg * n * c + i];
        out[((int64_t)bt * n + i / c) * hc + h * c + (i % c)] = v;
    }
}

// ---- dkf[bt][p][h*C+c] = sum_n a[n][p] * dctx[n][c] + ds[n][p] * q[n][c]: workgroup = (32-pixel chunk, head, frame) -------------------
__global__ __launch_bounds__(256) void ctx_keys_grad_kernel(const float* __restrict__ a, const float* __restrict__ ds,
                                                            const float* __restrict__ dctx, const float* __restrict__ q,
                                                            float* __restrict__ dkf, int n, int p, int heads, int c) {
    __shared__ float as_[MAXN * 32], dss[MAXN * 32];
    const int h = blockIdx.y, bt = blockIdx.z, p0 = blockIdx.x * 32;
    const int hc = heads * c, groups = blockDim.x / c;
    const int cc = threadIdx.x % c, g = threadIdx.x / c;
    const int64_t rowb = ((int64_t)bt * heads + h) * n;
    for (int i = threadIdx.x; i < n * 32; i += blockDim.x) {
        const int r = i >> 5, j = i & 31;
        const bool ok = p0 + j < p;
        as_[i] = ok ? a[(rowb + r) * p + p0 + j] : 0.f;
        dss[i] = ok ? ds[(rowb + r) * p + p0 + j] : 0.f;
    }
    float dc[MAXN], qq[MAXN];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) {
        dc[i] = i < n ? dctx[((int64_t)bt * n + i) * hc + h * c + cc] : 0.f;
        qq[i] = i < n ? q[((int64_t)bt * n + i) * hc + h * c + cc] : 0.f;
    }
    __syncthreads();
    const int jend = min(32, p - p0);
    for (int j = g; j < jend; j += groups) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < MAXN; ++i)
            if (i < n) v += as_[i * 32 + j] * dc[i] + dss[i * 32 + j] * qq[i];
        dkf[((int64_t)bt * p + p0 + j) * hc + h * c + cc] = v;
    }
}

// ---- context + position embedding (infer_model.py:404-406): y[f][i] = float(x[f][i]) + pos[i], x in the backbone's storage type ---------
__global__ void add_position_kernel(const void* __restrict__ x, int dtype, const float* __restrict__ pos, float* __restrict__ y,
                                    int64_t frames, int64_t per_frame) {
    const int64_t total = frames * per_frame / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4;
        f32x4 v;
        if (dtype == DIN_F32) v = *reinterpret_cast<const f32x4*>((const float*)x + e);
        else {
            const uint2 r = *reinterpret_cast<const uint2*>((const bf16_t*)x + e);
            v = f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
        }
        const f32x4 pv = *reinterpret_cast<const f32x4*>(pos + e % per_frame);
        *reinterpret_cast<f32x4*>(y + e) = v + pv;
    }
}
// backward: gx = cast(gy) (* (x > 0): the ReLU that produced the context map -- gradients handed to the backbone graph are pre-masked)
__global__ void add_position_bwd_kernel(const float* __restrict__ gy, const void* __restrict__ x, int dtype, void* __restrict__ gx,
                                        int64_t total4, int use_mask) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4;
        f32x4 g = *reinterpret_cast<const f32x4*>(gy + e);
        if (dtype == DIN_F32) {
            if (use_mask) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>((const float*)x + e);
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = xv[k] > 0.f ? g[k] : 0.f;
            }
            *reinterpret_cast<f32x4*>((float*)gx + e) = g;
        } else {
            if (use_mask) {
                const uint2 r = *reinterpret_cast<const uint2*>((const bf16_t*)x + e);
                const float xv[4] = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                                     __uint_as_float(r.y & 0xffff0000u)};
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = xv[k] > 0.f ? g[k] : 0.f;
            }
            *reinterpret_cast<uint2*>((bf16_t*)gx + e) = uint2{pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3])};
        }
    }
}

// ---- y = dropout(relu?(x)) and its backward (nn.ReLU + nn.Dropout inside the transformer's FFN, nn.Dropout on the attended context) --
__global__ void act_dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int relu, float p, uint64_t seed,
                                   const uint64_t* __restrict__ seed_off) {
    seed = fold_seed(seed, seed_off);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i];
        if (relu) v = fmaxf(v, 0.f);
        y[i] = v * keep_scale(seed, i, p);
    }
}
__global__ void act_dropout_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ gx, int64_t n, int relu,
                                       float p, uint64_t seed, const uint64_t* __restrict__ seed_off) {
    seed = fold_seed(seed, seed_off);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float g = gy[i] * keep_scale(seed, i, p);
        if (relu && !(x[i] > 0.f)) g = 0.f;
        gx[i] = g;
    }
}

inline bool attention_shape_ok(int bt, int n, int p, int heads, int c) {
    return bt > 0 && n > 0 && n <= MAXN && p > 0 && heads > 0 && c >= 4 && c <= 256 && (c & (c - 1)) == 0;
}

}  // namespace

extern "C" {

int din_ctx_scores(const float* q, const float* kf, float* s, int bt, int n, int p, int heads, int c, void* stream) {
    DIN_REQUIRE(q && kf && s, "ctx_scores: null pointer");
    DIN_REQUIRE(attention_shape_ok(bt, n, p, heads, c), "ctx_scores: need 1 <= n <= %d boxes and a power-of-two feature width 4..256 (n %d, c %d)", MAXN, n, c);
    hipLaunchKernelGGL(ctx_scores_kernel, dim3((p + 255) / 256, heads, bt), dim3(256), (size_t)n * c * sizeof(float), as_stream(stream), q, kf,
                       s, n, p, heads, c);
    DIN_CHECK_LAUNCH("ctx_scores");
    return DIN_OK;
}

int din_softmax_rows(float* s, int64_t rows, int len, void* stream) {
    DIN_REQUIRE(s && rows >= 0 && len > 0 && rows < (1ll << 31), "softmax_rows: bad argument");
    if (rows == 0) return DIN_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), s, len);
    DIN_CHECK_LAUNCH("softmax_rows");
    return DIN_OK;
}

int din_softmax_rows_bwd(const float* a, float* da, int64_t rows, int len, void* stream) {
    DIN_REQUIRE(a && da && rows >= 0 && len > 0 && rows < (1ll << 31), "softmax_rows_bwd: bad argument");
    if (rows == 0) return DIN_OK;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), a, da, len);
    DIN_CHECK_LAUNCH("softmax_rows_bwd");
    return DIN_OK;
}

int din_ctx_apply(const float* a, const float* kf, float* out, int bt, int n, int p, int heads, int c, void* stream) {
    DIN_REQUIRE(a && kf && out, "ctx_apply: null pointer");
    DIN_REQUIRE(attention_shape_ok(bt, n, p, heads, c), "ctx_apply: need 1 <= n <= %d boxes and a power-of-two feature width 4..256 (n %d, c %d)", MAXN, n, c);
    const int groups = 256 / c;
    const size_t lds = sizeof(float) * (size_t)((n * 64 > groups * n * c) ? n * 64 : groups * n * c);
    hipLaunchKernelGGL(ctx_apply_kernel, dim3(heads, bt), dim3(256), lds, as_stream(stream), a, kf, out, n, p, heads, c);
    DIN_CHECK_LAUNCH("ctx_apply");
    return DIN_OK;
}

int din_ctx_keys_grad(const float* a, const float* ds, const float* dctx, const float* q, float* dkf, int bt, int n, int p, int heads, int c,
                      void* stream) {
    DIN_REQUIRE(a && ds && dctx && q && dkf, "ctx_keys_grad: null pointer");
    DIN_REQUIRE(attention_shape_ok(bt, n, p, heads, c), "ctx_keys_grad: need 1 <= n <= %d boxes and a power-of-two feature width 4..256 (n %d, c %d)", MAXN, n, c);
    hipLaunchKernelGGL(ctx_keys_grad_kernel, dim3((p + 31) / 32, heads, bt), dim3(256), 0, as_stream(stream), a, ds, dctx, q, dkf, n, p, heads, c);
    DIN_CHECK_LAUNCH("ctx_keys_grad");
    return DIN_OK;
}

int din_add_position(const void* x, int dtype, const float* pos, float* y, int64_t frames, int64_t per_frame, void* stream) {
    DIN_REQUIRE(x && pos && y, "add_position: null pointer");
    DIN_REQUIRE((dtype == DIN_F32 || dtype == DIN_BF16) && frames >= 0 && per_frame > 0 && per_frame % 4 == 0, "add_position: bad argument");
    if (frames == 0) return DIN_OK;
    hipLaunchKernelGGL(add_position_kernel, dim3(grid_1d(frames * per_frame / 4, 256)), dim3(256), 0, as_stream(stream), x, dtype, pos, y, frames,
                       per_frame);
    DIN_CHECK_LAUNCH("add_position");
    return DIN_OK;
}

int din_add_position_bwd(const float* gy, const void* x, int dtype, void* gx, int64_t elems, int use_mask, void* stream) {
    DIN_REQUIRE(gy && gx && (x || !use_mask), "add_position_bwd: null pointer");
    DIN_REQUIRE((dtype == DIN_F32 || dtype == DIN_BF16) && elems >= 0 && elems % 4 == 0, "add_position_bwd: bad argument");
    if (elems == 0) return DIN_OK;
    hipLaunchKernelGGL(add_position_bwd_kernel, dim3(grid_1d(elems / 4, 256)), dim3(256), 0, as_stream(stream), gy, x, dtype, gx, elems / 4,
                       use_mask);
    DIN_CHECK_LAUNCH("add_position_bwd");
    return DIN_OK;
}

int din_act_dropout_fwd(const float* x, float* y, int64_t n, int relu, float drop_p, uint64_t seed, const uint64_t* seed_offset, void* stream) {
    DIN_REQUIRE(x && y && n >= 0 && drop_p >= 0.f && drop_p < 1.f, "act_dropout_fwd: bad argument");
    if (n == 0) return DIN_OK;
    hipLaunchKernelGGL(act_dropout_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, as_stream(stream), x, y, n, relu, drop_p, seed, seed_offset);
    DIN_CHECK_LAUNCH("act_dropout_fwd");
    return DIN_OK;
}

int din_act_dropout_bwd(const float* gy, const float* x, float* gx, int64_t n, int relu, float drop_p, uint64_t seed,
                        const uint64_t* seed_offset, void* stream) {
    DIN_REQUIRE(gy && x && gx && n >= 0 && drop_p >= 0.f && drop_p < 1.f, "act_dropout_bwd: bad argument");
    if (n == 0) return DIN_OK;
    hipLaunchKernelGGL(act_dropout_bwd_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, as_stream(stream), gy, x, gx, n, relu, drop_p, seed, seed_offset);
    DIN_CHECK_LAUNCH("act_dropout_bwd");
    return DIN_OK;
}

}  // extern "C"
