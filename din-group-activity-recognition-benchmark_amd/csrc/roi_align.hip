// Row R: RoIAlign(K,K) = TF crop_and_resize with transform_fpcoor=True (third-party longcw/RoIAlign.pytorch,
// un-vendored by the reference; call site infer_model.py:178-180).  HBM/L2-bound gather: one wave per (box, ky, kx)
// sample streams the channel axis of the NHWC feature map (coalesced 256-byte rows), output is written in the
// reference's flatten order [m][c][ky][kx] (infer_model.py:181) so fc_emb_1's weight layout is unchanged.
//
// The sample coordinates follow the exact fp32 operation order of the algorithm so that the integer decisions
// (floor/ceil indices, out-of-range tests) are bit-exact with the CPU oracle.
#include "din_common.h"

namespace {

struct Sample { float in; int lo, hi; float l; bool oob; };

// y-axis (or x-axis) sample `i` of a box side [c1,c2] on a feature axis of `extent` cells, crop size k
__device__ __forceinline__ Sample roi_sample(float c1, float c2, int extent, int k, int i) {
    // spacing = (c2-c1)/k ; n0 = (c1 + spacing/2 - 0.5)/(extent-1) ; nlen = spacing*(k-1)/(extent-1)
    const float kf = (float)k, em1 = (float)(extent - 1);
    float sp = __fdiv_rn(__fsub_rn(c2, c1), kf);
    float n0 = __fdiv_rn(__fsub_rn(__fadd_rn(c1, __fdiv_rn(sp, 2.0f)), 0.5f), em1);
    float nl = __fdiv_rn(__fmul_rn(sp, (float)(k - 1)), em1);
    float n1 = __fadd_rn(n0, nl);
    float in;
    if (k > 1) {
        float step = __fdiv_rn(__fmul_rn(__fsub_rn(n1, n0), em1), (float)(k - 1));
        in = __fadd_rn(__fmul_rn(n0, em1), __fmul_rn((float)i, step));
    } else {
        in = __fmul_rn(__fmul_rn(0.5f, __fadd_rn(n0, n1)), em1);
    }
    Sample s;
    s.in = in;
    s.oob = (in < 0.f) || (in > em1);
    float fl = floorf(in), ce = ceilf(in);
    s.l = __fsub_rn(in, fl);
    int lo = (int)fl, hi = (int)ce;
    s.lo = lo < 0 ? 0 : (lo > extent - 1 ? extent - 1 : lo);
    s.hi = hi < 0 ? 0 : (hi > extent - 1 ? extent - 1 : hi);
    return s;
}

// grid: one workgroup per (box, ky); wave w handles kx = w, w+4, ...; lanes stream channels
__global__ void roi_align_fwd_kernel(const void* __restrict__ fm, int fm_dtype, int nb, int hf, int wf, int c, int ldf,
                                     const float* __restrict__ boxes, const int32_t* __restrict__ box_ind, int m, int k,
                                     float* __restrict__ out, int32_t* __restrict__ idx_out) {
    const int b = blockIdx.x / k, ky = blockIdx.x % k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const float x1 = boxes[b * 4 + 0], y1 = boxes[b * 4 + 1], x2 = boxes[b * 4 + 2], y2 = boxes[b * 4 + 3];
    const int n = box_ind[b];
    const Sample sy = roi_sample(y1, y2, hf, k, ky);
    if (idx_out && threadIdx.x < k) {
        // record the x-axis decisions of sample kx = threadIdx.x together with this row's y decisions
        Sample sx = roi_sample(x1, x2, wf, k, threadIdx.x);
        int32_t* o = idx_out + ((int64_t)(b * k + ky) * k + threadIdx.x) * 6;
        o[0] = sy.lo; o[1] = sy.hi; o[2] = sx.lo; o[3] = sx.hi; o[4] = sy.oob; o[5] = sx.oob;
    }
    for (int kx = wave; kx < k; kx += nwaves) {
        const Sample sx = roi_sample(x1, x2, wf, k, kx);
        const bool dead = sy.oob || sx.oob || n < 0 || n >= nb;
        const int64_t r_tl = ((int64_t)(n * hf + sy.lo) * wf + sx.lo) * ldf, r_tr = ((int64_t)(n * hf + sy.lo) * wf + sx.hi) * ldf;
        const int64_t r_bl = ((int64_t)(n * hf + sy.hi) * wf + sx.lo) * ldf, r_br = ((int64_t)(n * hf + sy.hi) * wf + sx.hi) * ldf;
        float* dst = out + (int64_t)b * c * k * k + ky * k + kx;
        for (int ch = lane; ch < c; ch += 64) {
            float v = 0.f;
            if (!dead) {
                float tl = load_as_f32(fm, fm_dtype, r_tl + ch), tr = load_as_f32(fm, fm_dtype, r_tr + ch);
                float bl = load_as_f32(fm, fm_dtype, r_bl + ch), br = load_as_f32(fm, fm_dtype, r_br + ch);
                float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), sx.l));
                float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), sx.l));
                v = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), sy.l));
            }
            dst[(int64_t)ch * k * k] = v;
        }
    }
}

// backward: dfm[n, corner, ch] += w_corner * dout[b, ch, ky, kx]   (fp32 atomics, coalesced along channels)
__global__ void roi_align_bwd_kernel(const float* __restrict__ dout, int nb, int hf, int wf, int c,
                                     const float* __restrict__ boxes, const int32_t* __restrict__ box_ind, int m, int k,
                                     float* __restrict__ dfm) {
    const int b = blockIdx.x / k, ky = blockIdx.x % k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const float x1 = boxes[b * 4 + 0], y1 = boxes[b * 4 + 1], x2 = boxes[b * 4 + 2], y2 = boxes[b * 4 + 3];
    const int n = box_ind[b];
    const Sample sy = roi_sample(y1, y2, hf, k, ky);
    if (sy.oob || n < 0 || n >= nb) return;
    for (int kx = wave; kx < k; kx += nwaves) {
        const Sample sx = roi_sample(x1, x2, wf, k, kx);
        if (sx.oob) continue;
        float* p_tl = dfm + ((int64_t)(n * hf + sy.lo) * wf + sx.lo) * c;
        float* p_tr = dfm + ((int64_t)(n * hf + sy.lo) * wf + sx.hi) * c;
        float* p_bl = dfm + ((int64_t)(n * hf + sy.hi) * wf + sx.lo) * c;
        float* p_br = dfm + ((int64_t)(n * hf + sy.hi) * wf + sx.hi) * c;
        const float* src = dout + (int64_t)b * c * k * k + ky * k + kx;
        for (int ch = lane; ch < c; ch += 64) {
            float g = src[(int64_t)ch * k * k];
            // out = top + (bot-top)*ly ; top = tl + (tr-tl)*lx
            float gt = g * (1.f - sy.l), gb = g * sy.l;
            atomicAdd(p_tl + ch, gt * (1.f - sx.l));
            atomicAdd(p_tr + ch, gt * sx.l);
            atomicAdd(p_bl + ch, gb * (1.f - sx.l));
            atomicAdd(p_br + ch, gb * sx.l);
        }
    }
}

}  // namespace

extern "C" {

int din_roi_align_fwd(const void* fm, int fm_dtype, int nb, int hf, int wf, int c, int ldf, const float* boxes,
                      const int32_t* box_ind, int m, int k, float* out, int32_t* idx_out, void* stream) {
    DIN_REQUIRE(fm && boxes && box_ind && out, "roi_align_fwd: null pointer");
    DIN_REQUIRE(nb > 0 && hf > 1 && wf > 1 && c > 0 && k > 0 && m >= 0 && ldf >= c, "roi_align_fwd: bad shape");
    DIN_REQUIRE(k <= 64, "roi_align_fwd: crop size > 64 unsupported");
    if (m == 0) return DIN_OK;
    hipLaunchKernelGGL(roi_align_fwd_kernel, dim3(m * k), dim3(256), 0, as_stream(stream), fm, fm_dtype, nb, hf, wf, c, ldf,
                       boxes, box_ind, m, k, out, idx_out);
    DIN_CHECK_LAUNCH("roi_align_fwd");
    return DIN_OK;
}

int din_roi_align_bwd(const float* dout, int nb, int hf, int wf, int c, const float* boxes, const int32_t* box_ind, int m,
                      int k, float* dfm, void* stream) {
    DIN_REQUIRE(dout && boxes && box_ind && dfm, "roi_align_bwd: null pointer");
    DIN_REQUIRE(nb > 0 && hf > 1 && wf > 1 && c > 0 && k > 0 && m >= 0, "roi_align_bwd: bad shape");
    if (m == 0) return DIN_OK;
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(m * k), dim3(256), 0, as_stream(stream), dout, nb, hf, wf, c, boxes, box_ind, m, k, dfm);
    DIN_CHECK_LAUNCH("roi_align_bwd");
    return DIN_OK;
}

}  // extern "C"
