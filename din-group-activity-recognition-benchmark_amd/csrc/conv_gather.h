// Shared between conv_igemm.hip and conv_gather_pipe.hip: the forward / dgrad kernel argument block and the second half of the staged
// epilogue (LDS tile -> 16-byte coalesced NHWC stores with the fused ReLU-backward mask / accumulate / second destination).
#pragma once
// Timing knock-outs (DIN_GATHER_KNOCK: drop the pixel fetches / filter fetches / MFMAs of the gather kernels -- results are WRONG) and the
// other experiment switches of rounds 2-3 (DIN_CONV_W16, DIN_CONV_RING) exist only in builds made with -DDIN_EXPERIMENTS
// (tools/ab_gather.sh builds one into knock_build/); the shipped library never reads those variables and the kernel-side tests fold to
// `false` at compile time (a stray environment variable must not be able to corrupt every conv).
#ifdef DIN_EXPERIMENTS
#define DIN_KNOCK(flags, bit) (((flags) & (bit)) != 0)
#else
#define DIN_KNOCK(flags, bit) false
#endif
#include "din_common.h"

typedef uint32_t din_u32x4 __attribute__((ext_vector_type(4)));

namespace din_gather {

struct ConvK {
    const void* in; const void* w; void* out; const float* bias; const void* mask; float* partial;
    int NB, H, W, Cin, ldi, cioff;
    int OH, OW, Cout, ldo, cooff;
    int kh, kw;
    int ay, by, cy, divy;       // ty = oy*ay + by + r*cy ; needs ty % divy == 0 ; iy = ty / divy
    int ax, bx, cx, divx;
    int cpt, Q, nk, M, wld;     // chunks per tap, total chunks, k-steps, pixels, packed row length (chunks)
    int flags, ldm, moff;
    int splitk, ks_per_split, n_co_tiles;
    long long in_bytes, w_bytes;    // extents for the buffer resources
    // output pixel of tile row m=(n,a,b): ((n*out_H + a*out_sy + out_y0)*out_W + b*out_sx + out_x0); identity when out_sy==0
    int out_sy, out_sx, out_y0, out_x0, out_H, out_W;
    int korder;                     // 1: reduction runs channel-chunk outer / tap inner (uniform taps only): consecutive k-steps
                                    //    re-read nearly the same pixels (shifted by one tap) -> L1 hits instead of L2 traffic
    int remap;                      // 1: filter tap t of this launch is tap wtap[t] of the packed bank (tap subsets)
    unsigned char wtap[32];
    // multi-source 1x1 gather (fused dgrad of several 1x1 convs that read the same tensor): the reduction runs over the
    // concatenation of the sources' channels; source b = its own tensor (pixel stride, channel offset) + its own filter bank
    int nsrc;
    struct Src { const void* in; const void* w; long long in_bytes, w_bytes; int cpt, ld, coff, wld; } src[4];
    // extra 1x1 source of a single-source launch (nsrc == 0, xsteps > 0): after the launch's own reduction, xsteps more k-steps read src[0]
    // AT THE OUTPUT PIXEL (a 1x1 / stride-1 conv over the same input view: its dgrad lands on the same pixels) -- din_conv_dgrad_x folds the
    // block-entry 1x1 of Mixed_6a into the parity-class launches of the strided 3x3 beside it (one pass over dX instead of two)
    int xsteps;
    // second destination (fused sibling convs that read one tensor): produced channels >= csplit go to out2 (pixel stride ldo2, channel
    // offset cooff2 + (channel - csplit)); csplit == 0: single destination.  Staged (aligned) epilogue only, no mask / accumulate.
    void* out2; int ldo2, cooff2, csplit;
    int craw;                       // > 0: produced channels >= craw get neither bias nor ReLU (a sibling whose epilogue runs later, after its pool)
    const unsigned char* u8;        // image layer only: raw uint8 frames [NB][3][H][W], normalised on load (din_conv_desc::in_u8)
};

// host entries of conv_stream.hip: persistent streaming kernel for 1x1 convolutions with a short reduction (ring kept full across tiles)
bool conv1x1_stream_eligible(const ConvK& k, int dtype);
int launch_conv1x1_stream(ConvK k, hipStream_t st);
int conv1x1_stream_tile(int cout);                 // filter tile of a launch producing cout channels (64 | 96 | 192)
// host entries of conv_regw.hip: 1x1 convolutions with a 640..768-channel reduction over a large map, filters resident in registers
bool conv1x1_regw_eligible(const ConvK& k, int dtype);
int launch_conv1x1_regw(const ConvK& k, hipStream_t st);

__device__ __forceinline__ int64_t out_pixel(const ConvK& p, int m) {
    if (p.out_sy == 0) return m;
    int n = m / (p.OH * p.OW);
    int rem = m - n * (p.OH * p.OW);
    int a = rem / p.OW, b = rem - a * p.OW;
    return ((int64_t)n * p.out_H + a * p.out_sy + p.out_y0) * p.out_W + b * p.out_sx + p.out_x0;
}


// The workgroup's output tile sits in LDS as BM rows of BN elements, row pitch BN * sizeof(T) + 16 bytes (written by the caller, NOT yet
// synchronised).  Every thread stores 16-byte chunks of whole rows: plain stores for forward launches, and for gradient launches the
// ReLU-backward mask / accumulate inputs of ALL the thread's rows are requested before the staged tile is read back (one memory latency
// per tile instead of one per row).  Channels >= p.csplit go to the second destination (fused sibling convs).
template <typename T, int BM, int BN, int NT>
__device__ __forceinline__ void staged_tile_store(const ConvK& p, unsigned char* smem_raw, int tid, int co_tile, int m_first) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CPITCH = BN * (int)sizeof(T) + 16;
    typedef din_u32x4 u32x4;
        constexpr int CPR = BN * (int)sizeof(T) / 16;                // 16-byte chunks per tile row
        constexpr int RPP = NT / CPR;                                // rows per pass (threads beyond RPP*CPR idle when CPR !| NT)
        constexpr int NROW = (BM + RPP - 1) / RPP;                   // rows per thread
        const int c = tid % CPR, rr = tid / CPR;
        const int co = co_tile * BN + c * EPC;
        const bool act = co < p.Cout && rr < RPP;
        T* __restrict__ outp = reinterpret_cast<T*>(p.out);
        const T* __restrict__ maskp = reinterpret_cast<const T*>(p.mask);
        if (!(p.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM))) {
            // plain stores (every forward launch): one pass over the thread's rows.  (Routing these through the batched form below cost
            // the short-K, store-bound layers 15-20 %: Conv2d_3b forward 558 -> 635 us.)
            __syncthreads();
            if (act) {
                const bool second = p.csplit > 0 && co >= p.csplit;
                T* __restrict__ dstp = second ? reinterpret_cast<T*>(p.out2) : outp;
                const int ldd = second ? p.ldo2 : p.ldo, offd = second ? p.cooff2 + (co - p.csplit) : p.cooff + co;
                for (int row = rr; row < BM; row += RPP) {
                    const int m = m_first + row;
                    if (m >= p.M) break;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(smem_raw + row * CPITCH + c * 16);
                    *reinterpret_cast<u32x4*>(dstp + out_pixel(p, m) * ldd + offd) = v;
                }
            }
            return;
        }
        if (p.flags & 0x100) {                                       // tuning aid (DIN_CONV_EPI_BATCH=0): per-row load -> combine -> store
            __syncthreads();
            if (act) {
                for (int row = rr; row < BM; row += RPP) {
                    const int m = m_first + row;
                    if (m >= p.M) break;
                    u32x4 v = *reinterpret_cast<const u32x4*>(smem_raw + row * CPITCH + c * 16);
                    const int64_t px = out_pixel(p, m);
                    const int64_t o = px * p.ldo + p.cooff + co;
                    u32x4 mk = {0u, 0u, 0u, 0u}, old = {0u, 0u, 0u, 0u};
                    if (p.flags & DIN_CONV_MASK) mk = *reinterpret_cast<const u32x4*>(maskp + px * p.ldm + p.moff + co);
                    if (p.flags & DIN_CONV_ACCUM) old = *reinterpret_cast<const u32x4*>(outp + o);
                    if constexpr (sizeof(T) == 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = __uint_as_float(v[e]);
                            if ((p.flags & DIN_CONV_MASK) && !(__uint_as_float(mk[e]) > 0.f)) x = 0.f;
                            if (p.flags & DIN_CONV_ACCUM) x += __uint_as_float(old[e]);
                            v[e] = __float_as_uint(x);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                            if (p.flags & DIN_CONV_MASK) {
                                if (!(__uint_as_float(mk[e] << 16) > 0.f)) lo = 0.f;
                                if (!(__uint_as_float(mk[e] & 0xffff0000u) > 0.f)) hi = 0.f;
                            }
                            if (p.flags & DIN_CONV_ACCUM) { lo += __uint_as_float(old[e] << 16); hi += __uint_as_float(old[e] & 0xffff0000u); }
                            v[e] = pack_bf16x2(lo, hi);
                        }
                    }
                    *reinterpret_cast<u32x4*>(outp + o) = v;
                }
            }
            return;
        }
        // ReLU-backward mask / accumulate inputs of ALL this thread's rows are requested before the staged tile is read back: one
        // memory latency per tile instead of one per row (the per-row load -> wait -> store chain cost 60-80 us per launch on the
        // 288-channel dgrads; profiles/r01_stream_probe.txt)
        u32x4 mkv[NROW], oldv[NROW];
        int opx[NROW];
#pragma unroll
        for (int q = 0; q < NROW; ++q) {
            mkv[q] = u32x4{0u, 0u, 0u, 0u}; oldv[q] = u32x4{0u, 0u, 0u, 0u}; opx[q] = -1;
            const int row = rr + q * RPP, m = m_first + row;
            if (act && row < BM && m < p.M) {
                const int64_t px = out_pixel(p, m);
                opx[q] = (int)px;
                if (p.flags & DIN_CONV_MASK) mkv[q] = *reinterpret_cast<const u32x4*>(maskp + px * p.ldm + p.moff + co);
                if (p.flags & DIN_CONV_ACCUM) oldv[q] = *reinterpret_cast<const u32x4*>(outp + px * p.ldo + p.cooff + co);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NROW; ++q) {
            if (opx[q] < 0) continue;
            const int row = rr + q * RPP;
            u32x4 v = *reinterpret_cast<const u32x4*>(smem_raw + row * CPITCH + c * 16);
            const bool second = p.csplit > 0 && co >= p.csplit;
            if (second) outp = reinterpret_cast<T*>(p.out2);
            const int64_t o = second ? (int64_t)opx[q] * p.ldo2 + p.cooff2 + (co - p.csplit) : (int64_t)opx[q] * p.ldo + p.cooff + co;
            if (p.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM)) {
                const u32x4 mk = mkv[q], old = oldv[q];
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = __uint_as_float(v[e]);
                        if ((p.flags & DIN_CONV_MASK) && !(__uint_as_float(mk[e]) > 0.f)) x = 0.f;
                        if (p.flags & DIN_CONV_ACCUM) x += __uint_as_float(old[e]);
                        v[e] = __float_as_uint(x);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                        if (p.flags & DIN_CONV_MASK) {
                            if (!(__uint_as_float(mk[e] << 16) > 0.f)) lo = 0.f;
                            if (!(__uint_as_float(mk[e] & 0xffff0000u) > 0.f)) hi = 0.f;
                        }
                        if (p.flags & DIN_CONV_ACCUM) { lo += __uint_as_float(old[e] << 16); hi += __uint_as_float(old[e] & 0xffff0000u); }
                        v[e] = pack_bf16x2(lo, hi);
                    }
                }
            }
            *reinterpret_cast<u32x4*>(outp + o) = v;
        }
}

// host entries of conv_gather_pipe.hip
bool gather_pipe_tile_ok(int bn);
int launch_gather_pipe(const ConvK& k, int bn, int n_px_tiles, hipStream_t st);

}  // namespace din_gather
