// THIRD GENERATION of the tap-line experiment (round 5): the loader / consumer split INSIDE one barrier domain.  Same data layout as
// conv_line64.hip (pixel line of a 256-position tile resident in LDS across the taps, 64-channel filter stages, 32 x 32 x 16 MFMA fragments,
// epilogue straight from the accumulators) -- but no LDS flags at all: every wave of the workgroup meets at ONE s_barrier per stage.
//   * loader waves: before barrier j they wait (counted vmcnt) until their share of stage j + G - 1 has landed; after it -- every consumer is done
//     with stage j - 1 -- they issue stage j + R - 1 into the slot that just became free (and, in front of a channel block's first stage, its
//     pixel line).  A stage has R - G stage-times to land.
//   * consumer waves: fragment reads and MFMAs only -- no transfer issue, no vmcnt wait, no polling, no atomics.  With G = 2 the first fragments of
//     stage j + 1 are requested before barrier j + 1 (they were guaranteed at barrier j), so the matrix pipe does not drain at the barrier.
// R = filter ring slots (what LDS is left: 5 at 128 filters, 4 at 160, 3 at 192), G = guarantee distance.  Not part of libdin_hip.so.
#include "../../din-group-activity-recognition-benchmark_amd/csrc/din_common.h"
#include "../../din-group-activity-recognition-benchmark_amd/csrc/conv_wgrad.h"
#include "../../din-group-activity-recognition-benchmark_amd/csrc/conv_gather.h"
#include <type_traits>


namespace din_lineb {
using din_gather::ConvK;
using din_wgrad::lds_dma16;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 256, HALO = 4, PROWS = 272, PBYTES = PROWS * 128, NPB = 2;
constexpr uint32_t OOB = 0x80000000u;

struct LineK {
    const void* in; const void* w; void* out; const float* bias; const void* mask;
    int* err;                              // optional: set to 1 when a hand-off wait ran into its bound (diagnostics)
    uint32_t* prof;                        // PROF builds: [workgroup][wave][8] cycle counts (tools/probes/line_probe.hip)
    int L, OUTER, HW;                      // walk: inner length, outer count per image, pixels per image
    int strideA, strideB;                  // pixel index of walk position (n, a, b) = n * HW + a * strideA + b * strideB
    int Q;                                 // walk positions = pixels of the launch
    int ldi, cioff, ldo, cooff, ldm, moff;
    int Cout, cpt, ncb;                    // produced channels, 16-byte chunks per tap (even), 64-channel blocks
    int taps, shift0, dshift;              // tap t reads walk position q + shift0 + t * dshift
    int wld;                               // packed filter row length in chunks
    int flags;
    int ntiles, n_co_tiles;
    long long in_bytes, w_bytes;
};

template <int BN, int R> constexpr int lds_bytes() { return NPB * PBYTES + R * BN * 128; }

__device__ __forceinline__ int walk_pixel(const LineK& p, int q) {
    const int t2 = q / p.L, b = q - t2 * p.L;
    const int n = t2 / p.OUTER, a = t2 - n * p.OUTER;
    return n * p.HW + a * p.strideA + b * p.strideB;
}
__device__ __forceinline__ void wait_vmcnt(int n) {
#define DIN_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        DIN_VM(0) DIN_VM(1) DIN_VM(2) DIN_VM(3) DIN_VM(4) DIN_VM(5) DIN_VM(6) DIN_VM(7) DIN_VM(8) DIN_VM(9) DIN_VM(10) DIN_VM(11) DIN_VM(12) DIN_VM(13) DIN_VM(14) DIN_VM(15)
        DIN_VM(16) DIN_VM(17) DIN_VM(18) DIN_VM(19) DIN_VM(20) DIN_VM(21) DIN_VM(22) DIN_VM(23) DIN_VM(24) DIN_VM(25) DIN_VM(26) DIN_VM(27) DIN_VM(28) DIN_VM(29) DIN_VM(30) DIN_VM(31)
        DIN_VM(32) DIN_VM(33) DIN_VM(34) DIN_VM(35) DIN_VM(36) DIN_VM(37) DIN_VM(38) DIN_VM(39) DIN_VM(40) DIN_VM(41) DIN_VM(42) DIN_VM(43) DIN_VM(44) DIN_VM(45) DIN_VM(46) DIN_VM(47)
        DIN_VM(48) DIN_VM(49) DIN_VM(50) DIN_VM(51) DIN_VM(52) DIN_VM(53) DIN_VM(54) DIN_VM(55) DIN_VM(56) DIN_VM(57) DIN_VM(58) DIN_VM(59) DIN_VM(60) DIN_VM(61) DIN_VM(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
#undef DIN_VM
}

// ---- loader waves ----------------------------------------------------------------------------------------------------------------------------
template <int BN, int NC, int NL, int R, int G, bool PROF>
__device__ __forceinline__ void loader(const LineK& p, unsigned char* smem, int lw, int lane) {
    constexpr int NFP = BN / 8 / NL, NPP = PROWS / 8 / NL;
    static_assert((BN / 8) % NL == 0 && (PROWS / 8) % NL == 0 && G >= 1 && G < R && (R - 1) * NFP + NPP <= 63, "transfers");
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    const int lrow = lane >> 3, slot = lane & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    int voffF[NFP];
    unsigned voffP[NPP];
    const int nst = p.ncb * p.taps;                             // stages per tile
    int my_tiles = 0;
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) ++my_tiles;
    const int total = my_tiles * nst;                           // stages of this workgroup
    // issue cursor
    int is = 0, i_tile = blockIdx.x, i_cb = 0, i_t = 0, i_line = 0, i_cotile = -1;
    uint32_t issued = 0;                                        // transfers issued so far by this wave
    uint64_t endq = 0;                                          // (issued & 255) right after stage s went out, at byte s % 8

    auto set_filter_rows = [&](int co_tile) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NFP; ++i) {
            const int rho = (lw + NL * i) * 8 + lrow, rr = rho & 31;
            const int q4 = rr >> 3, h = (rr >> 2) & 1, e = rr & 3;
            const int chan = co_tile * BN + (rho & ~31) + (q4 >> 1) * 16 + h * 8 + (q4 & 1) * 4 + e;
            voffF[i] = chan < p.Cout ? (chan * p.wld + (slot ^ ((rho >> 1) & 7))) * 16 : (int)OOB;
        }
    };
    auto set_line = [&](int q0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int r = (lw + NL * i) * 8 + lrow, q = q0 - HALO + r;
            unsigned vo = OOB;
            if (q >= 0 && q < p.Q) vo = (unsigned)(walk_pixel(p, q) * p.ldi * 2 + p.cioff * 2 + ((slot ^ ((r >> 1) & 7)) << 4));
            voffP[i] = vo;
        }
    };
    auto issue_next = [&]() __attribute__((always_inline)) {    // stage `is` of this workgroup (+ its channel block's pixel line in front of tap 0)
        if (is >= total) return;
        const int px_tile = i_tile / p.n_co_tiles, co_tile = i_tile - px_tile * p.n_co_tiles;
        if (i_t == 0) {
            if (i_cb == 0) { set_line(px_tile * TP); if (co_tile != i_cotile) { set_filter_rows(co_tile); i_cotile = co_tile; } }
            const uint32_t dstP = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((i_line & 1) * PBYTES));
            const int soffP = __builtin_amdgcn_readfirstlane(i_cb * 128);
#pragma unroll
            for (int i = 0; i < NPP; ++i) lds_dma16(dstP + (uint32_t)((lw + NL * i) * 1024), rsA, (int)voffP[i], soffP);
            issued += NPP;
            i_line = __builtin_amdgcn_readfirstlane(i_line + 1);
        }
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(NPB * PBYTES + (is % R) * BN * 128));
        const int soff = __builtin_amdgcn_readfirstlane((i_t * p.cpt + i_cb * 8) * 16);
#pragma unroll
        for (int i = 0; i < NFP; ++i) lds_dma16(dst + (uint32_t)((lw + NL * i) * 1024), rsB, voffF[i], soff);
        issued += NFP;
        const int sh = (is & 7) * 8;
        endq = (endq & ~(0xffull << sh)) | ((uint64_t)(issued & 255u) << sh);
        is = __builtin_amdgcn_readfirstlane(is + 1);
        if (++i_t == p.taps) { i_t = 0; if (++i_cb == p.ncb) { i_cb = 0; i_tile += gridDim.x; } }
        i_t = __builtin_amdgcn_readfirstlane(i_t); i_cb = __builtin_amdgcn_readfirstlane(i_cb); i_tile = __builtin_amdgcn_readfirstlane(i_tile);
    };
    [[maybe_unused]] uint32_t t_land = 0, t_bar = 0, t_issue = 0;
    [[maybe_unused]] const uint64_t T0 = PROF ? __builtin_readcyclecounter() : 0;
    for (int s = 0; s < R - 1; ++s) issue_next();
    for (int j = 0; j < total; ++j) {
        int need = j + G - 1;                                   // this stage must have landed before the barrier that opens stage j
        if (need > total - 1) need = total - 1;
        const int younger = (int)((issued - (uint32_t)((endq >> ((need & 7) * 8)) & 0xff)) & 255u);
        [[maybe_unused]] const uint64_t c0 = PROF ? __builtin_readcyclecounter() : 0;
        wait_vmcnt(younger);
        [[maybe_unused]] const uint64_t c1 = PROF ? __builtin_readcyclecounter() : 0;
        __builtin_amdgcn_s_barrier();
        [[maybe_unused]] const uint64_t c2 = PROF ? __builtin_readcyclecounter() : 0;
        issue_next();                                           // stage j + R - 1 into the slot stage j - 1 just left
        if constexpr (PROF) { t_land += (uint32_t)(c1 - c0); t_bar += (uint32_t)(c2 - c1); t_issue += (uint32_t)(__builtin_readcyclecounter() - c2); }
    }
    if constexpr (PROF) {
        if (p.prof && lane == 0) {
            uint32_t* o = p.prof + ((size_t)blockIdx.x * (NC + NL) + NC + lw) * 8;
            o[0] = (uint32_t)(__builtin_readcyclecounter() - T0); o[1] = t_land; o[2] = t_bar; o[3] = t_issue; o[4] = (uint32_t)total;
        }
    }
}

// ---- consumer waves --------------------------------------------------------------------------------------------------------------------------
template <int BN, int WM, int WN, int NL, int R, int G, bool PROF, int STAG>
__device__ __forceinline__ void consumer(const LineK& p, unsigned char* smem, int wid, int lane) {
    constexpr int TI = BN / WN / 32, TJ = TP / WM / 32, BNW = BN / WN, PXW = TP / WM;
    static_assert(BN % (32 * WN) == 0 && TP % (32 * WM) == 0 && TJ * 8 <= 32, "wave tile = whole 32 x 32 MFMA tiles");
    [[maybe_unused]] uint32_t t_bar = 0, t_epi = 0, t_first = 0;
    [[maybe_unused]] const uint64_t T0 = PROF ? __builtin_readcyclecounter() : 0;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int col = lane & 31, hh = lane >> 5;
    const int frow = wn * BNW + col;
    const uint32_t fbase = (uint32_t)(frow * 128) + (uint32_t)((hh ^ ((frow >> 1) & 7)) << 4);   // chunk (2u + hh) ^ sw = (hh ^ sw) ^ 2u
    const int xrow0 = wm * PXW + col + HALO;
    uint32_t gs = 0, gl = 0;                                    // stages / pixel lines consumed so far (ring position = gs % R, line buffer = gl & 1)
    const int nst = p.ncb * p.taps;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int px_tile = tile / p.n_co_tiles, co_tile = tile - px_tile * p.n_co_tiles;
        const int q0 = px_tile * TP;
        // tap validity of this lane's pixels: bit j * 8 + t
        uint32_t vbits = 0u;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int q = q0 + wm * PXW + j * 32 + col;
            const int b = q - (q / p.L) * p.L;
            for (int t = 0; t < p.taps; ++t) {
                const int s = b + p.shift0 + t * p.dshift;
                vbits |= (s >= 0 && s < p.L) ? (1u << (j * 8 + t)) : 0u;
            }
        }
        // accumulators start at the bias: element v of tile i is channel cbase + i * 32 + (v >> 3) * 16 + hh * 8 + (v & 7)
        f32x16 acc[TI][TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            f32x16 b16;
#pragma unroll
            for (int e = 0; e < 16; ++e) b16[e] = 0.f;
            if (p.flags & DIN_CONV_BIAS) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int c = co_tile * BN + wn * BNW + i * 32 + half * 16 + hh * 8;
                    if (c < p.Cout) {
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + c), b1 = *reinterpret_cast<const f32x4*>(p.bias + c + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { b16[half * 8 + e] = b0[e]; b16[half * 8 + 4 + e] = b1[e]; }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = b16;
        }


        uint32_t pa[TJ];
        u32x4 wf[2][TI], xf[2][TJ];
        const unsigned char* Pb = smem;
        const unsigned char* Fs = smem;
        auto set_stage = [&](int t_, uint32_t line, uint32_t stage) __attribute__((always_inline)) {
            Pb = smem + __builtin_amdgcn_readfirstlane((int)(line & 1) * PBYTES);
            Fs = smem + __builtin_amdgcn_readfirstlane(NPB * PBYTES + (int)(stage % R) * (BN * 128));
            const int shift = p.shift0 + t_ * p.dshift;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int Rw = xrow0 + j * 32 + shift;
                pa[j] = (uint32_t)(Rw * 128) + (uint32_t)((hh ^ ((Rw >> 1) & 7)) << 4);
            }
        };
        auto rdw = [&](int set, int u) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TI; ++i) wf[set][i] = *reinterpret_cast<const u32x4*>(Fs + i * 4096 + (fbase ^ (uint32_t)(u << 5)));
#pragma unroll
            for (int j = 0; j < TJ; ++j) xf[set][j] = *reinterpret_cast<const u32x4*>(Pb + (pa[j] ^ (uint32_t)(u << 5)));
        };
        int cb = 0, t = 0;
        bool have = false;                                     // set 0 already holds (or is receiving) sub-step 0 of the stage about to start
        for (int g = 0; g < nst; ++g) {
            [[maybe_unused]] const uint64_t b0 = PROF ? __builtin_readcyclecounter() : 0;
            __builtin_amdgcn_s_barrier();                      // stage gs (and, G = 2, gs + 1) has landed; everyone is done with stage gs - 1
            asm volatile("" ::: "memory");
            if constexpr (PROF) { const uint32_t d = (uint32_t)(__builtin_readcyclecounter() - b0); t_bar += d; if (g == 0) t_first += d; }
            // the two consumer waves of a SIMD (w and w + 4) leave the barrier in lock-step and then alternate MFMAs fairly: both do their fragment
            // reads / masking at the same time and the matrix pipe idles.  A one-off delay of the second wave puts them half a sub-step apart.
            if constexpr (STAG > 0) { if (wid >= 4) __builtin_amdgcn_s_sleep(STAG); }
            if (!have) { set_stage(t, gl, gs); rdw(0, 0); }
            const bool last_tap = t + 1 == p.taps;
            const bool exists = g + 1 < nst;
            const int tn = last_tap ? 0 : t + 1;
            uint32_t mk[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) mk[j] = (uint32_t)__builtin_amdgcn_sbfe((int)vbits, j * 8 + t, 1);
            const int nh = (p.cpt - cb * 8 >= 8) ? 2 : 1;      // host: cpt % 4 == 0
            auto mma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xf[set][j][e] &= mk[j];
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[set][i]), __builtin_bit_cast(bf16x8, xf[set][j]), acc[i][j], 0, 0, 0);
            };
            have = false;
            for (int h = 0; h < nh; ++h) {
                const bool last_half = h + 1 == nh;
                rdw(1, 2 * h + 1);
                mma(0);
                int un = 2 * h + 2;
                if (last_half) {
                    un = 0;
                    if (G >= 2 && exists) { have = true; set_stage(tn, last_tap ? gl + 1 : gl, gs + 1); }   // next stage: guaranteed since this stage's barrier
                    asm volatile("" ::: "memory");
                }
                rdw(0, un);                                    // next half / next stage (G = 1 or the tile's last stage: a harmless dummy read)
                mma(1);
            }
            gs += 1; if (last_tap) { gl += 1; cb += 1; }
            t = tn;
        }

        [[maybe_unused]] const uint64_t E0 = PROF ? __builtin_readcyclecounter() : 0;
        // ---- epilogue: lane holds, per (i, j), output channels cbase + half * 16 + hh * 8 + [0, 8) of pixel column `col` -------------------
        int pix[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int q = q0 + wm * PXW + j * 32 + col;
            pix[j] = q < p.Q ? walk_pixel(p, q) : -1;
        }
        bf16_t* __restrict__ outp = reinterpret_cast<bf16_t*>(p.out);
        const bf16_t* __restrict__ maskp = reinterpret_cast<const bf16_t*>(p.mask);
        const int cw = co_tile * BN + wn * BNW + hh * 8;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = cw + i * 32 + half * 16;
                if (c >= p.Cout) continue;
                // gradient launches: the mask / accumulate operands of both pixel columns are requested together
                u32x4 mk[TJ], old[TJ];
                if (p.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM)) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        mk[j] = u32x4{0u, 0u, 0u, 0u}; old[j] = u32x4{0u, 0u, 0u, 0u};
                        if (pix[j] >= 0) {
                            if (p.flags & DIN_CONV_MASK) mk[j] = *reinterpret_cast<const u32x4*>(maskp + (int64_t)pix[j] * p.ldm + p.moff + c);
                            if (p.flags & DIN_CONV_ACCUM) old[j] = *reinterpret_cast<const u32x4*>(outp + (int64_t)pix[j] * p.ldo + p.cooff + c);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    if (pix[j] < 0) continue;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[i][j][half * 8 + e];
                    if (p.flags & DIN_CONV_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                    if (p.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM)) {
                        // same arithmetic as staged_tile_store (conv_gather.h): the bf16-rounded value is masked, the old value added in fp32
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo = __uint_as_float(o[e] << 16), hi = __uint_as_float(o[e] & 0xffff0000u);
                            if (p.flags & DIN_CONV_MASK) {
                                if (!(__uint_as_float(mk[j][e] << 16) > 0.f)) lo = 0.f;
                                if (!(__uint_as_float(mk[j][e] & 0xffff0000u) > 0.f)) hi = 0.f;
                            }
                            if (p.flags & DIN_CONV_ACCUM) { lo += __uint_as_float(old[j][e] << 16); hi += __uint_as_float(old[j][e] & 0xffff0000u); }
                            o[e] = pack_bf16x2(lo, hi);
                        }
                    }
                    *reinterpret_cast<u32x4*>(outp + (int64_t)pix[j] * p.ldo + p.cooff + c) = o;
                }
            }
        }
        if constexpr (PROF) t_epi += (uint32_t)(__builtin_readcyclecounter() - E0);
    }
    if constexpr (PROF) {
        if (p.prof && lane == 0) {
            uint32_t* o = p.prof + ((size_t)blockIdx.x * (WM * WN + NL) + wid) * 8;
            o[0] = (uint32_t)(__builtin_readcyclecounter() - T0); o[1] = t_bar; o[2] = t_epi; o[3] = t_first; o[4] = 0;
        }
    }
}

template <int BN, int WM, int WN, int NL, int R, int G, bool PROF = false, int STAG = 0>
__global__ __launch_bounds__(64 * (WM * WN + NL), 1) void conv_lineb_kernel(LineK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wid >= WM * WN) loader<BN, WM * WN, NL, R, G, PROF>(p, smem, wid - WM * WN, lane);
    else consumer<BN, WM, WN, NL, R, G, PROF, STAG>(p, smem, wid, lane);
#endif
}

int launch_lineb(LineK lk, int bn, int ncu, hipStream_t st, int variant = 0) {
    lk.n_co_tiles = (lk.Cout + bn - 1) / bn;
    lk.ntiles = (lk.Q + TP - 1) / TP * lk.n_co_tiles;
    lk.ncb = (lk.cpt + 7) / 8;
    const int grid = lk.ntiles < ncu ? lk.ntiles : ncu;
    auto go = [&](auto kern, size_t lds, int threads) __attribute__((always_inline)) {
        din_raise_lds(reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, lk);
    };
    if (lk.taps < 5) return -1;                                  // (a line must outlive R - 1 stages: 7-tap layers only)
    if (bn == 192 && lk.prof) go(conv_lineb_kernel<192, 4, 2, 2, 3, 1, true>, lds_bytes<192, 3>(), 640);
    else if (bn == 128 && lk.prof) go(conv_lineb_kernel<128, 4, 2, 2, 5, 2, true>, lds_bytes<128, 5>(), 640);
    else if (bn == 192 && variant == 0) go(conv_lineb_kernel<192, 4, 2, 2, 3, 1>, lds_bytes<192, 3>(), 640);
    else if (bn == 192 && variant == 1) go(conv_lineb_kernel<192, 4, 2, 2, 3, 2>, lds_bytes<192, 3>(), 640);
    else if (bn == 192 && variant == 2) go(conv_lineb_kernel<192, 4, 2, 2, 3, 1, false, 1>, lds_bytes<192, 3>(), 640);
    else if (bn == 192 && variant == 3) go(conv_lineb_kernel<192, 4, 2, 2, 3, 1, false, 2>, lds_bytes<192, 3>(), 640);
    else if (bn == 192 && variant == 4) go(conv_lineb_kernel<192, 4, 2, 2, 3, 1, false, 3>, lds_bytes<192, 3>(), 640);
    else if (bn == 128 && variant == 2) go(conv_lineb_kernel<128, 4, 2, 2, 5, 2, false, 1>, lds_bytes<128, 5>(), 640);
    else if (bn == 128 && variant == 3) go(conv_lineb_kernel<128, 4, 2, 2, 5, 2, false, 2>, lds_bytes<128, 5>(), 640);
    else if (bn == 160 && variant == 0) go(conv_lineb_kernel<160, 8, 1, 2, 4, 2>, lds_bytes<160, 4>(), 640);
    else if (bn == 160 && variant == 1) go(conv_lineb_kernel<160, 8, 1, 2, 4, 1>, lds_bytes<160, 4>(), 640);
    else if (bn == 128 && variant == 0) go(conv_lineb_kernel<128, 4, 2, 2, 5, 2>, lds_bytes<128, 5>(), 640);
    else if (bn == 128 && variant == 1) go(conv_lineb_kernel<128, 4, 2, 2, 4, 1>, lds_bytes<128, 4>(), 640);
    else return -1;
    return 0;
}

}  // namespace din_lineb
