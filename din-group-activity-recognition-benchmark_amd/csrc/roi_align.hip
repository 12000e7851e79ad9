// Row R: RoIAlign(K,K) = TF crop_and_resize with transform_fpcoor=True (third-party longcw/RoIAlign.pytorch,
// un-vendored by the reference; call site infer_model.py:178-180).  HBM/L2-bound gather: one wave per (box, ky, kx)
// sample streams the channel axis of the NHWC feature map (coalesced 256-byte rows), output is written in the
// reference's flatten order [m][c][ky][kx] (infer_model.py:181) so fc_emb_1's weight layout is unchanged.
//
// The sample coordinates follow the exact fp32 operation order of the algorithm so that the integer decisions
// (floor/ceil indices, out-of-range tests) are bit-exact with the CPU oracle.
#include "din_common.h"

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

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

// ---- composed sampling: the map the boxes live on is a VIRTUAL align_corners bilinear resize (infer_model.py:165-170, F.interpolate) of a
// smaller stored map.  Both operations are separable and linear, so a sample's value is a 3 x 3 weighted sum of stored pixels: per axis
// the sample's two grid cells (weights 1-l, l) each read two stored cells, and because the stored axis is not longer than the grid axis
// the (at most) four cells are stored cells r0, r0+1, r0+2.  With extent == grid the taps reduce to {lo: 1-l, hi: l} -- the plain RoIAlign.
struct Tap3 { int r0; float w[3]; bool oob; };

// grid cell g -> stored coordinate g * (extent-1)/(grid-1): same fp32 operations as the resize kernel (pool.hip bil_coord)
__device__ __forceinline__ void resize_coord(int g, int extent, int grid, int& i0, int& i1, float& l) {
    const float sc = grid > 1 ? (float)(extent - 1) / (float)(grid - 1) : 0.f;
    const float src = sc * (float)g;
    i0 = (int)src;
    if (i0 > extent - 1) i0 = extent - 1;
    i1 = i0 + 1 < extent ? i0 + 1 : extent - 1;
    l = src - (float)i0;
}

__device__ __forceinline__ Tap3 axis_taps(float c1, float c2, int grid, int extent, int k, int i) {
    const Sample s = roi_sample(c1, c2, grid, k, i);
    Tap3 t;
    t.oob = s.oob;
    t.w[0] = t.w[1] = t.w[2] = 0.f;
    if (extent == grid) {
        t.r0 = s.lo;
        t.w[0] = 1.f - s.l;
        if (s.hi == s.lo) t.w[0] += s.l; else t.w[1] = s.l;
        return t;
    }
    int a0, a1, b0, b1; float la, lb;
    resize_coord(s.lo, extent, grid, a0, a1, la);
    resize_coord(s.hi, extent, grid, b0, b1, lb);
    t.r0 = a0;
    const float wl = 1.f - s.l, wh = s.l;
    // (indices relative to r0 are 0..2: b0 is a0 or a0 + 1 because one grid step is at most one stored step)
    const int ia1 = a1 - a0, ib0 = min(b0 - a0, 2), ib1 = min(b1 - a0, 2);
    t.w[0] += wl * (1.f - la);
    if (ia1 == 0) t.w[0] += wl * la; else t.w[1] += wl * la;
    if (ib0 == 0) t.w[0] += wh * (1.f - lb); else if (ib0 == 1) t.w[1] += wh * (1.f - lb); else t.w[2] += wh * (1.f - lb);
    if (ib1 == 0) t.w[0] += wh * lb; else if (ib1 == 1) t.w[1] += wh * lb; else t.w[2] += wh * lb;
    return t;
}

// Forward through a virtually resized map: one workgroup per (box, ky), wave w handles kx = w, w+4, ..., lanes stream 16-byte channel
// chunks of the 3 x 3 stored pixels.  Output as roi_align_fwd_kernel: out[b][out_coff + ch][ky][kx] of an [m][out_c][k][k] crop tensor.
template <int V>
__global__ void roi_align_fwd_composed_kernel(const void* __restrict__ fm, int fm_dtype, int nb, int hf, int wf, int c, int ldf, int gh, int gw,
                                              const float* __restrict__ boxes, const int32_t* __restrict__ box_ind, int m, int k,
                                              float* __restrict__ out, int out_c, int out_coff) {
    extern __shared__ float row[];                                     // [c][k]
    const int b = blockIdx.x / k, ky = blockIdx.x % k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const float x1 = boxes[b * 4 + 0], y1 = boxes[b * 4 + 1], x2 = boxes[b * 4 + 2], y2 = boxes[b * 4 + 3];
    const int n = box_ind[b];
    const Tap3 ty = axis_taps(y1, y2, gh, hf, k, ky);
    for (int kx = wave; kx < k; kx += nwaves) {
        const Tap3 tx = axis_taps(x1, x2, gw, wf, k, kx);
        const bool dead = ty.oob || tx.oob || n < 0 || n >= nb;
        for (int ch = lane * V; ch < c; ch += 64 * V) {
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
            if (!dead) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int yy = min(ty.r0 + i, hf - 1);
                    float rowacc[V];
#pragma unroll
                    for (int e = 0; e < V; ++e) rowacc[e] = 0.f;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int xx = min(tx.r0 + j, wf - 1);
                        const int64_t off = ((int64_t)(n * hf + yy) * wf + xx) * ldf + ch;
                        const float wj = tx.w[j];
                        if (fm_dtype == DIN_F32) {
                            const f32x4_t q = *reinterpret_cast<const f32x4_t*>((const float*)fm + off);
#pragma unroll
                            for (int e = 0; e < V; ++e) rowacc[e] += wj * q[e & 3];
                        } else {
                            const u32x4_t q = *reinterpret_cast<const u32x4_t*>((const bf16_t*)fm + off);
#pragma unroll
                            for (int e = 0; e < V; ++e)
                                rowacc[e] += wj * ((e & 1) ? __uint_as_float(q[(e >> 1) & 3] & 0xffff0000u) : __uint_as_float(q[(e >> 1) & 3] << 16));
                        }
                    }
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += ty.w[i] * rowacc[e];
                }
            }
#pragma unroll
            for (int e = 0; e < V; ++e) row[(ch + e) * k + kx] = acc[e];
        }
    }
    __syncthreads();
    float* dst = out + ((int64_t)b * out_c + out_coff) * k * k + ky * k;
    for (int i = threadIdx.x; i < c * k; i += blockDim.x) {
        const int ch = i / k, kx = i - ch * k;
        dst[(int64_t)ch * k * k + kx] = row[i];
    }
}

// grid: one workgroup per (box, ky); wave w handles kx = w, w+4, ...; lanes stream channels.
// The crop goes out in the reference's flatten order [m][c][k*k] (the column order of fc_emb_1, infer_model.py:181-184), i.e. a lane's
// channels are k*k floats apart.  With `row` (dynamic LDS, c * k floats) the workgroup first assembles its k samples channel-by-channel
// in LDS and then writes [c][k] runs of k contiguous floats; the corner loads take V channels (16 bytes) per lane.
template <int V>
__global__ void roi_align_fwd_kernel(const void* __restrict__ fm, int fm_dtype, int nb, int hf, int wf, int c, int ldf,
                                     const float* __restrict__ boxes, const int32_t* __restrict__ box_ind, int m, int k,
                                     float* __restrict__ out, int32_t* __restrict__ idx_out, int out_c, int out_coff) {
    extern __shared__ float row[];                                     // [c][k] (V > 1 only)
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
        if constexpr (V == 1) {
            float* dst = out + ((int64_t)b * out_c + out_coff) * k * k + ky * k + kx;
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
        } else {
            // V = 8 bf16 / 4 fp32 channels per lane: one 16-byte load per corner
            for (int ch = lane * V; ch < c; ch += 64 * V) {
                float tl[V], tr[V], bl[V], br[V];
                if (!dead) {
                    if (fm_dtype == DIN_F32) {
                        const f32x4_t a = *reinterpret_cast<const f32x4_t*>((const float*)fm + r_tl + ch), bq = *reinterpret_cast<const f32x4_t*>((const float*)fm + r_tr + ch);
                        const f32x4_t cq = *reinterpret_cast<const f32x4_t*>((const float*)fm + r_bl + ch), dq = *reinterpret_cast<const f32x4_t*>((const float*)fm + r_br + ch);
#pragma unroll
                        for (int e = 0; e < V; ++e) { tl[e] = a[e & 3]; tr[e] = bq[e & 3]; bl[e] = cq[e & 3]; br[e] = dq[e & 3]; }
                    } else {
                        const u32x4_t a = *reinterpret_cast<const u32x4_t*>((const bf16_t*)fm + r_tl + ch), bq = *reinterpret_cast<const u32x4_t*>((const bf16_t*)fm + r_tr + ch);
                        const u32x4_t cq = *reinterpret_cast<const u32x4_t*>((const bf16_t*)fm + r_bl + ch), dq = *reinterpret_cast<const u32x4_t*>((const bf16_t*)fm + r_br + ch);
                        auto up = [](const u32x4_t& q, int e) { return (e & 1) ? __uint_as_float(q[(e >> 1) & 3] & 0xffff0000u) : __uint_as_float(q[(e >> 1) & 3] << 16); };
#pragma unroll
                        for (int e = 0; e < V; ++e) { tl[e] = up(a, e); tr[e] = up(bq, e); bl[e] = up(cq, e); br[e] = up(dq, e); }
                    }
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    float v = 0.f;
                    if (!dead) {
                        float top = __fadd_rn(tl[e], __fmul_rn(__fsub_rn(tr[e], tl[e]), sx.l));
                        float bot = __fadd_rn(bl[e], __fmul_rn(__fsub_rn(br[e], bl[e]), sx.l));
                        v = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), sy.l));
                    }
                    row[(ch + e) * k + kx] = v;
                }
            }
        }
    }
    if constexpr (V > 1) {
        __syncthreads();
        // out[b][out_coff + ch][ky][0..k): k contiguous floats per channel
        float* dst = out + ((int64_t)b * out_c + out_coff) * k * k + ky * k;
        for (int i = threadIdx.x; i < c * k; i += blockDim.x) {
            const int ch = i / k, kx = i - ch * k;
            dst[(int64_t)ch * k * k + kx] = row[i];
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


// Gather-form backward, written straight into the feature map's gradient tensor (storage type T, NHWC view, channels [0, c)):
//   gfm[n, y, x, ch] = (mask: fm > 0) * sum over boxes b of frame n, samples (ky, kx):  wy(b,ky,y) * wx(b,kx,x) * dout[b, ch, ky, kx]
// with wy = (lo == y ? 1-l : 0) + (hi == y ? l : 0) -- the same four corner weights the scatter form adds, but no atomics, no fp32
// staging tensor, no zero-fill and no separate cast/mask pass: every element of the view is written exactly once (zeros outside
// the boxes' footprints, ~90 % of the map).  Deterministic: contributions are summed in (box, ky, kx) order.
// One workgroup per (frame, row).  LDS: the frame's boxes (ascending), their k x-samples and this row's k y-weights.  Wave w owns
// pixels x = w, w+4, ...: the (wave-uniform) scan over the row's active boxes finds the samples touching x, lanes stream 16-byte
// channel chunks.  `cap` boxes of a frame are handled per batch; later batches (frames with more than `cap` boxes) re-read the
// partial result, which is then rounded to T once per batch.
template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_gather_kernel(const float* __restrict__ dout, int nb, int hf, int wf, int c,
                                                                   const float* __restrict__ boxes, const int32_t* __restrict__ box_ind,
                                                                   int m, int k, const T* __restrict__ fm, int ldf, T* __restrict__ gfm,
                                                                   int ldg, int cap, int prezeroed, int transposed, int gh, int gw,
                                                                   int dout_c, int dout_coff) {
    // (gh, gw): the grid the boxes live on; the stored map hf x wf is its virtual align_corners resize source (axis_taps) -- equal extents
    // give the plain RoIAlign.  dout holds dout_c channels per crop, this map's c channels start at dout_coff.
    // transposed: dout is the channel-contiguous copy [m][k*k][c] made by roi_transpose_kernel (a lane's V channels are one 16/32-byte
    // load and a wave reads 2 KiB contiguous); otherwise the reference layout [m][c][k*k] (4-byte loads 100 B apart)
    constexpr int V = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* boxid = reinterpret_cast<int*>(smem);                         // [cap] boxes of this frame (batch), ascending
    int* xc0 = boxid + cap;                                            // [cap * k] first stored column of x-sample q's three taps (-4: out of range)
    float* xw = reinterpret_cast<float*>(xc0 + cap * k);               // [cap * k * 3] their weights
    float* wyv = xw + 3 * cap * k;                                     // [cap * k] y-weight of sample ky on this row (0: none)
    int* xmin = reinterpret_cast<int*>(wyv + cap * k);                 // [cap] x extent of the box's samples
    int* xmax = xmin + cap;
    int* act = xmax + cap;                                             // [cap] slots of the boxes that touch this row
    __shared__ int wtot[4];
    __shared__ int nact_s, nfound_s, rx0_s, rx1_s;
    const int n = blockIdx.x / hf, y = blockIdx.x - n * hf;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = k * k;
    int scan_from = 0;                                                 // next box id to look at (uniform)
    for (int batch = 0;; ++batch) {
        // ---- the next <= cap boxes of frame n, in ascending id order (ballot compaction, 256 ids per round) ----
        int found = 0;
        while (scan_from < m && found < cap) {
            const int b = scan_from + tid;
            const bool hit = b < m && box_ind[b] == n;
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wtot[wave] = __popcll(bal);
            __syncthreads();
            int off = found;
            for (int w = 0; w < wave; ++w) off += wtot[w];
            const int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            const int slot = off + __popcll(bal & ((1ull << lane) - 1ull));
            // a round that would overflow the batch is cut at the first id that does not fit; the scan resumes there
            __shared__ int cut_s;
            if (tid == 0) cut_s = scan_from + 256;
            __syncthreads();
            if (hit && slot == cap) cut_s = b;                         // exactly one thread
            if (hit && slot < cap) boxid[slot] = b;
            __syncthreads();
            found = min(found + tot, cap);
            scan_from = cut_s;
            __syncthreads();
        }
        if (batch > 0 && found == 0) break;
        // ---- per box: x samples, this row's y weights, x extent ----
        for (int t = tid; t < found * k; t += 256) {
            const int j = t / k, q = t - j * k;
            const int b = boxid[j];
            const Tap3 tx = axis_taps(boxes[b * 4 + 0], boxes[b * 4 + 2], gw, wf, k, q);
            xc0[t] = tx.oob ? -4 : tx.r0;
            xw[3 * t] = tx.w[0]; xw[3 * t + 1] = tx.w[1]; xw[3 * t + 2] = tx.w[2];
            const Tap3 ty = axis_taps(boxes[b * 4 + 1], boxes[b * 4 + 3], gh, hf, k, q);
            const int dy = y - ty.r0;
            wyv[t] = (ty.oob || dy < 0 || dy > 2) ? 0.f : ty.w[dy];
        }
        __syncthreads();
        if (tid == 0) {
            int na = 0, rx0 = 1 << 30, rx1 = -1;
            for (int j = 0; j < found; ++j) {
                bool any = false; int lo = 1 << 30, hi = -1;
                for (int q = 0; q < k; ++q) {
                    any = any || wyv[j * k + q] != 0.f;
                    if (xc0[j * k + q] >= 0) { lo = min(lo, xc0[j * k + q]); hi = max(hi, min(xc0[j * k + q] + 2, wf - 1)); }
                }
                xmin[j] = lo; xmax[j] = hi;
                if (any && hi >= 0) { act[na++] = j; rx0 = min(rx0, lo); rx1 = max(rx1, hi); }
            }
            nact_s = na; nfound_s = found; rx0_s = rx0; rx1_s = rx1;
        }
        __syncthreads();
        const int nact = nact_s;
        // ---- pixels of the row ----
        const int nchunk = c / V;
        // prezeroed (the host cleared the whole view with one memset, ~90 % of it stays zero): only the columns the row's active boxes
        // span are visited, and pixels without a contribution are left alone
        const int x_begin = prezeroed ? max(rx0_s, 0) : 0, x_end = prezeroed ? min(rx1_s, wf - 1) : wf - 1;
        // per pixel the (wave-uniform) scan over boxes x samples runs ONCE and leaves the contributing samples in a small per-wave LDS
        // list (source offset, weight); the channel passes then only walk that list
        constexpr int CL = 48;
        __shared__ int cl_off[4][CL];
        __shared__ float cl_w[4][CL];
        for (int x = x_begin + wave; x <= x_end; x += 4) {
            const int64_t pix = ((int64_t)n * hf + y) * wf + x;
            int ncl = 0;
            bool overflow = false;
            for (int a = 0; a < nact; ++a) {
                const int j = act[a];
                if (x < xmin[j] || x > xmax[j]) continue;
                const int b = boxid[j];
                for (int qy = 0; qy < k; ++qy) {
                    const float wy = wyv[j * k + qy];
                    if (wy == 0.f) continue;
                    for (int qx = 0; qx < k; ++qx) {
                        const int dx = x - xc0[j * k + qx];
                        if (dx < 0 || dx > 2) continue;                 // (out-of-range samples: c0 = -4)
                        const float wx = xw[3 * (j * k + qx) + dx];
                        if (wx == 0.f) continue;
                        if (ncl < CL) { cl_off[wave][ncl] = b * kk + qy * k + qx; cl_w[wave][ncl] = wy * wx; }   // same value from every lane
                        else overflow = true;
                        ++ncl;
                    }
                }
            }
            const bool any = ncl > 0;
            if (prezeroed && !any) continue;
            for (int ch0 = lane; ch0 - lane < nchunk; ch0 += 64) {
                const bool lane_ok = ch0 < nchunk;
                float acc[V];
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = 0.f;
                if (lane_ok && !overflow) {
                    for (int t = 0; t < ncl; ++t) {
                        const int off = cl_off[wave][t];
                        const float w = cl_w[wave][t];
                        if (transposed) {                                   // (box * kk + sample) * c + ch: contiguous channels
                            const float* src = dout + (int64_t)off * dout_c + dout_coff + ch0 * V;
#pragma unroll
                            for (int e4 = 0; e4 < V / 4; ++e4) {
                                const f32x4_t v4 = *reinterpret_cast<const f32x4_t*>(src + 4 * e4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[4 * e4 + e] += w * v4[e];
                            }
                        } else {
                            const int bq = off / kk;                        // box * kk + sample -> ((box * c + ch) * kk + sample)
                            const float* src = dout + ((int64_t)bq * dout_c + dout_coff + ch0 * V) * kk + (off - bq * kk);
#pragma unroll
                            for (int e = 0; e < V; ++e) acc[e] += w * src[(int64_t)e * kk];
                        }
                    }
                } else if (lane_ok) {                                       // more than CL samples touch this pixel: rescan (rare)
                    for (int a = 0; a < nact; ++a) {
                        const int j = act[a];
                        if (x < xmin[j] || x > xmax[j]) continue;
                        const int b = boxid[j];
                        for (int qy = 0; qy < k; ++qy) {
                            const float wy = wyv[j * k + qy];
                            if (wy == 0.f) continue;
                            for (int qx = 0; qx < k; ++qx) {
                                const int dx = x - xc0[j * k + qx];
                                if (dx < 0 || dx > 2) continue;
                                const float wx = xw[3 * (j * k + qx) + dx];
                                if (wx == 0.f) continue;
                                const float w = wy * wx;
                                const float* src = transposed ? dout + ((int64_t)b * kk + qy * k + qx) * dout_c + dout_coff + ch0 * V
                                                              : dout + ((int64_t)b * dout_c + dout_coff + ch0 * V) * kk + qy * k + qx;
                                const int64_t es = transposed ? 1 : kk;
#pragma unroll
                                for (int e = 0; e < V; ++e) acc[e] += w * src[(int64_t)e * es];
                            }
                        }
                    }
                }
                if (!lane_ok) continue;
                if (prezeroed && !any) continue;
                T* dst = gfm + pix * ldg + ch0 * V;
                if (any && fm != nullptr) {
                    const u32x4_t mv = *reinterpret_cast<const u32x4_t*>(fm + pix * ldf + ch0 * V);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        float yv;
                        if constexpr (sizeof(T) == 4) yv = __uint_as_float(mv[e]);
                        else yv = (e & 1) ? __uint_as_float(mv[e >> 1] & 0xffff0000u) : __uint_as_float(mv[e >> 1] << 16);
                        acc[e] = yv > 0.f ? acc[e] : 0.f;
                    }
                }
                if (batch > 0) {
                    if (!any) continue;                                // nothing to add to the earlier batches' result
                    const u32x4_t ov = *reinterpret_cast<const u32x4_t*>(dst);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        if constexpr (sizeof(T) == 4) acc[e] += __uint_as_float(ov[e]);
                        else acc[e] += (e & 1) ? __uint_as_float(ov[e >> 1] & 0xffff0000u) : __uint_as_float(ov[e >> 1] << 16);
                    }
                }
                u32x4_t o;
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(acc[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(acc[2 * e], acc[2 * e + 1]);
                }
                *reinterpret_cast<u32x4_t*>(dst) = o;
            }
        }
        if (nfound_s < cap || scan_from >= m) break;                   // the frame had no more boxes than this batch
        __syncthreads();
    }
}

// [m][c][kk] -> [m][kk][c]: the crop gradient arrives in the reference's flatten order (channel-major, infer_model.py:181) because
// that is the column order of fc_emb_1; the gather backward wants channels contiguous.  One workgroup per (box, 64-channel block).
__global__ __launch_bounds__(256) void roi_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int c, int kk) {
    extern __shared__ float tile[];                                   // [64][kk + 1]
    const int b = blockIdx.x, c0 = blockIdx.y * 64;
    const int nch = min(64, c - c0);
    const float* src = in + ((int64_t)b * c + c0) * kk;
    for (int i = threadIdx.x; i < nch * kk; i += 256) tile[(i / kk) * (kk + 1) + (i % kk)] = src[i];
    __syncthreads();
    float* dst = out + (int64_t)b * kk * c + c0;
    for (int i = threadIdx.x; i < kk * 64; i += 256) {
        const int s = i >> 6, ch = i & 63;
        if (ch < nch) dst[(int64_t)s * c + ch] = tile[ch * (kk + 1) + s];
    }
}

}  // namespace

extern "C" {

int din_roi_align_fwd(const void* fm, int fm_dtype, int nb, int hf, int wf, int c, int ldf, int gh, int gw, const float* boxes,
                      const int32_t* box_ind, int m, int k, float* out, int out_c, int out_coff, int32_t* idx_out, void* stream) {
    DIN_REQUIRE(fm && boxes && box_ind && out, "roi_align_fwd: null pointer");
    DIN_REQUIRE(nb > 0 && hf > 1 && wf > 1 && c > 0 && k > 0 && m >= 0 && ldf >= c, "roi_align_fwd: bad shape");
    DIN_REQUIRE(k <= 64, "roi_align_fwd: crop size > 64 unsupported");
    DIN_REQUIRE(out_coff >= 0 && out_c >= out_coff + c, "roi_align_fwd: channels [%d, %d) do not fit a %d-channel crop", out_coff, out_coff + c, out_c);
    DIN_REQUIRE(gh >= hf && gw >= wf, "roi_align_fwd: the box grid %dx%d must not be smaller than the stored map %dx%d", gh, gw, hf, wf);
    if (m == 0) return DIN_OK;
    const int v = fm_dtype == DIN_F32 ? 4 : 8;
    const size_t lds = (size_t)c * k * sizeof(float);
    if (gh != hf || gw != wf) {
        // boxes on a virtually resized map (multi-scale fuse, infer_model.py:165-172): 3 x 3 stored taps per sample
        DIN_REQUIRE(!idx_out, "roi_align_fwd: the index record describes the plain (unresized) sampling only");
        DIN_REQUIRE(c % v == 0 && ldf % v == 0 && lds <= 64 * 1024, "roi_align_fwd: resized sampling needs channels / stride in multiples of %d", v);
        if (v == 4) hipLaunchKernelGGL(roi_align_fwd_composed_kernel<4>, dim3(m * k), dim3(256), lds, as_stream(stream), fm, fm_dtype, nb, hf, wf, c,
                                       ldf, gh, gw, boxes, box_ind, m, k, out, out_c, out_coff);
        else hipLaunchKernelGGL(roi_align_fwd_composed_kernel<8>, dim3(m * k), dim3(256), lds, as_stream(stream), fm, fm_dtype, nb, hf, wf, c,
                                ldf, gh, gw, boxes, box_ind, m, k, out, out_c, out_coff);
        DIN_CHECK_LAUNCH("roi_align_fwd(resized)");
        return DIN_OK;
    }
    if (c % v == 0 && ldf % v == 0 && lds <= 64 * 1024) {
        if (v == 4) hipLaunchKernelGGL(roi_align_fwd_kernel<4>, dim3(m * k), dim3(256), lds, as_stream(stream), fm, fm_dtype, nb, hf, wf, c, ldf,
                                       boxes, box_ind, m, k, out, idx_out, out_c, out_coff);
        else hipLaunchKernelGGL(roi_align_fwd_kernel<8>, dim3(m * k), dim3(256), lds, as_stream(stream), fm, fm_dtype, nb, hf, wf, c, ldf,
                                boxes, box_ind, m, k, out, idx_out, out_c, out_coff);
    } else
        hipLaunchKernelGGL(roi_align_fwd_kernel<1>, dim3(m * k), dim3(256), 0, as_stream(stream), fm, fm_dtype, nb, hf, wf, c, ldf,
                           boxes, box_ind, m, k, out, idx_out, out_c, out_coff);
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

int din_roi_crop_grad_transpose(const float* dout, int m, int c, int k, float* out, void* stream) {
    DIN_REQUIRE(dout && out, "roi_crop_grad_transpose: null pointer");
    DIN_REQUIRE(m >= 0 && c > 0 && k > 0 && k <= 64, "roi_crop_grad_transpose: bad shape");
    if (m == 0) return DIN_OK;
    const int kk = k * k;
    hipLaunchKernelGGL(roi_transpose_kernel, dim3(m, (c + 63) / 64), dim3(256), (size_t)64 * (kk + 1) * sizeof(float), as_stream(stream), dout, out, c, kk);
    DIN_CHECK_LAUNCH("roi_crop_grad_transpose");
    return DIN_OK;
}

int din_roi_align_bwd_nhwc(const float* dout, int dout_c, int dout_coff, int transposed, int nb, int hf, int wf, int c, int gh, int gw,
                           const float* boxes, const int32_t* box_ind, int m, int k, const void* fm_mask, int dtype, int ldf, void* gfm,
                           int ldg, void* stream) {
    DIN_REQUIRE(dout && boxes && box_ind && gfm, "roi_align_bwd_nhwc: null pointer");
    DIN_REQUIRE(nb > 0 && hf > 1 && wf > 1 && c > 0 && k > 0 && k <= 64 && m >= 0, "roi_align_bwd_nhwc: bad shape");
    DIN_REQUIRE(dtype == DIN_F32 || dtype == DIN_BF16, "roi_align_bwd_nhwc: bad dtype");
    DIN_REQUIRE(dout_coff >= 0 && dout_c >= dout_coff + c, "roi_align_bwd_nhwc: channels [%d, %d) do not fit a %d-channel crop", dout_coff, dout_coff + c, dout_c);
    DIN_REQUIRE(gh >= hf && gw >= wf, "roi_align_bwd_nhwc: the box grid %dx%d must not be smaller than the stored map %dx%d", gh, gw, hf, wf);
    const int v = dtype == DIN_F32 ? 4 : 8;
    DIN_REQUIRE(c % v == 0 && ldg % v == 0 && ldg >= c && (!fm_mask || (ldf % v == 0 && ldf >= c)),
                "roi_align_bwd_nhwc: channels / pixel strides must be multiples of %d", v);
    DIN_REQUIRE(!transposed || (dout_c % 4 == 0 && dout_coff % 4 == 0), "roi_align_bwd_nhwc: transposed crop gradient needs 16-byte channel groups");
    // boxes of one frame handled per batch: LDS = cap * (16 + 20 k) bytes, at most 48 KiB
    int cap = m < 1 ? 1 : m;
    const int cap_max = (48 * 1024) / (16 + 20 * k);
    if (cap > cap_max) cap = cap_max;
    if (cap > 256) cap = 256;
    const size_t lds = (size_t)cap * (16 + 20 * k);
    // a dense view is cleared with one memset (runs at the HBM write rate) and the kernel only touches the boxes' footprints
    const int prezeroed = ldg == c ? 1 : 0;
    if (prezeroed && hipMemsetAsync(gfm, 0, (size_t)nb * hf * wf * ldg * (dtype == DIN_F32 ? 4 : 2), as_stream(stream)) != hipSuccess)
        DIN_FAIL(DIN_E_LAUNCH, "roi_align_bwd_nhwc: memset");
    if (dtype == DIN_F32)
        hipLaunchKernelGGL(roi_align_bwd_gather_kernel<float>, dim3(nb * hf), dim3(256), lds, as_stream(stream), dout, nb, hf, wf, c, boxes,
                           box_ind, m, k, (const float*)fm_mask, ldf, (float*)gfm, ldg, cap, prezeroed, transposed, gh, gw, dout_c, dout_coff);
    else
        hipLaunchKernelGGL(roi_align_bwd_gather_kernel<bf16_t>, dim3(nb * hf), dim3(256), lds, as_stream(stream), dout, nb, hf, wf, c, boxes,
                           box_ind, m, k, (const bf16_t*)fm_mask, ldf, (bf16_t*)gfm, ldg, cap, prezeroed, transposed, gh, gw, dout_c, dout_coff);
    DIN_CHECK_LAUNCH("roi_align_bwd_nhwc");
    return DIN_OK;
}

}  // extern "C"
