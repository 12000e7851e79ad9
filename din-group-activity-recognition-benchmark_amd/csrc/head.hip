// Row H: activity head (infer_model.py:224-232): max over actors -> fc_activities -> mean over frames.
// s [b,t,n,c] -> scores [b,a].  One workgroup per (clip, frame); lanes stream channels (coalesced), the arg-max actor per
// channel is saved for the backward scatter.  Variable actor counts (Collective, infer_model.py:1288-1314) through
// n_per_clip.  Tiny and HBM/latency-bound.
#include "din_common.h"

namespace {

constexpr int HEAD_THREADS = 256;
constexpr int MAX_ACT = 16;

__global__ __launch_bounds__(HEAD_THREADS) void head_fwd_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const int32_t* __restrict__ n_per_clip,
                                                                int b, int t, int n, int c, int a, float* __restrict__ frame_scores,
                                                                int32_t* __restrict__ argmax) {
    __shared__ float red[HEAD_THREADS / 64][MAX_ACT];
    const int bi = blockIdx.x / t, ti = blockIdx.x % t;
    const int nv = n_per_clip ? n_per_clip[bi] : n;
    const float* base = s + ((int64_t)(bi * t + ti) * n) * c;
    float acc[MAX_ACT];
#pragma unroll
    for (int j = 0; j < MAX_ACT; ++j) acc[j] = 0.f;
    for (int ch = threadIdx.x; ch < c; ch += HEAD_THREADS) {
        float m = -INFINITY; int am = 0;
        for (int i = 0; i < nv; ++i) {
            float v = base[(int64_t)i * c + ch];
            if (v > m) { m = v; am = i; }           // first maximum wins (torch.max tie rule on CPU)
        }
        argmax[(int64_t)(bi * t + ti) * c + ch] = am;
#pragma unroll
        for (int j = 0; j < MAX_ACT; ++j) if (j < a) acc[j] += m * w[(int64_t)j * c + ch];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < MAX_ACT; ++j) {
        if (j < a) {
            float v = wave_sum(acc[j]);
            if (lane == 0) red[wv][j] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < a) {
        float v = bias[threadIdx.x];
        for (int i = 0; i < HEAD_THREADS / 64; ++i) v += red[i][threadIdx.x];
        frame_scores[(int64_t)(bi * t + ti) * a + threadIdx.x] = v;
    }
}

__global__ void head_mean_kernel(const float* __restrict__ frame_scores, float* __restrict__ scores, int b, int t, int a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * a) return;
    int bi = i / a, j = i - bi * a;
    float v = 0.f;
    for (int ti = 0; ti < t; ++ti) v += frame_scores[(int64_t)(bi * t + ti) * a + j];
    scores[i] = v / (float)t;
}

// ds[b,t,argmax,c] = sum_j dscores[b,j]/t * w[j,c];  dw[j,c] += dscores[b,j]/t * pooled[b,t,c];  dbias[j] += dscores[b,j]/t
__global__ __launch_bounds__(HEAD_THREADS) void head_bwd_kernel(const float* __restrict__ dscores, const float* __restrict__ s,
                                                                const float* __restrict__ w, const int32_t* __restrict__ argmax,
                                                                int b, int t, int n, int c, int a, float* __restrict__ ds,
                                                                float* __restrict__ dw, float* __restrict__ dbias) {
    const int bi = blockIdx.x / t, ti = blockIdx.x % t;
    const float inv_t = 1.f / (float)t;
    float* dsb = ds + ((int64_t)(bi * t + ti) * n) * c;
    const float* sb = s + ((int64_t)(bi * t + ti) * n) * c;
    for (int ch = threadIdx.x; ch < c; ch += HEAD_THREADS) {
        int am = argmax[(int64_t)(bi * t + ti) * c + ch];
        float pooled = sb[(int64_t)am * c + ch];
        float g = 0.f;
        for (int j = 0; j < a; ++j) {
            float dsc = dscores[bi * a + j] * inv_t;
            g += dsc * w[(int64_t)j * c + ch];
            atomicAdd(dw + (int64_t)j * c + ch, dsc * pooled);
        }
        for (int i = 0; i < n; ++i) dsb[(int64_t)i * c + ch] = (i == am) ? g : 0.f;
    }
    if (threadIdx.x < a) atomicAdd(dbias + threadIdx.x, dscores[bi * a + threadIdx.x] * inv_t);
}

}  // namespace

extern "C" {

// workspace-free: frame scores live in the tail of `argmax`'s sibling buffer provided by the caller?  No -- keep the ABI
// simple: scores must have room for b*a floats and the caller passes a scratch of b*t*a floats through `scores + b*a`.
int din_head_fwd(const float* s, const float* w, const float* bias, const int32_t* n_per_clip, int b, int t, int n, int c,
                 int a, float* scores, int32_t* argmax, void* stream) {
    DIN_REQUIRE(s && w && bias && scores && argmax, "head_fwd: null pointer");
    DIN_REQUIRE(b > 0 && t > 0 && n > 0 && c > 0 && a > 0 && a <= MAX_ACT, "head_fwd: bad shape (a <= %d)", MAX_ACT);
    float* frame_scores = scores + (int64_t)b * a;       // caller allocates b*a + b*t*a floats
    hipLaunchKernelGGL(head_fwd_kernel, dim3(b * t), dim3(HEAD_THREADS), 0, as_stream(stream), s, w, bias, n_per_clip, b, t, n, c, a,
                       frame_scores, argmax);
    DIN_CHECK_LAUNCH("head_fwd");
    hipLaunchKernelGGL(head_mean_kernel, dim3((b * a + 63) / 64), dim3(64), 0, as_stream(stream), frame_scores, scores, b, t, a);
    DIN_CHECK_LAUNCH("head_mean");
    return DIN_OK;
}

int din_head_bwd(const float* dscores, const float* s, const float* w, const int32_t* argmax, int b, int t, int n, int c,
                 int a, float* ds, float* dw, float* dbias, void* stream) {
    DIN_REQUIRE(dscores && s && w && argmax && ds && dw && dbias, "head_bwd: null pointer");
    DIN_REQUIRE(b > 0 && t > 0 && n > 0 && c > 0 && a > 0 && a <= MAX_ACT, "head_bwd: bad shape");
    hipLaunchKernelGGL(head_bwd_kernel, dim3(b * t), dim3(HEAD_THREADS), 0, as_stream(stream), dscores, s, w, argmax, b, t, n, c, a, ds, dw, dbias);
    DIN_CHECK_LAUNCH("head_bwd");
    return DIN_OK;
}

}  // extern "C"
