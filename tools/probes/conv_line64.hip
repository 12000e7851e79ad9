// FIRST GENERATION of the tap-line experiment (64-channel stages, 3-slot ring, one stage in flight per loader wave; the faster of the two:
// profiles/r05_line_kernel.txt).  Kept for the record next to conv_line32.hip; not part of libdin_hip.so.
// "Tap-line" convolution for gfx950: the 1 x k / k x 1 (and 1 x 1) stride-1 convolutions of Inception's Mixed_6 blocks, forward and data
// gradient, bf16 operands, fp32 accumulation -- wave-specialised (loader / consumer) persistent workgroups, no workgroup barrier in the loop.
//
//   D[co][q] = sum_{t, ci} Wpk[co][t][ci] * X[q + shift_t][ci]       q = position along the WALK (pixels ordered so that the conv's taps are
//                                                                   +-1 steps: row-major for 1 x k, column-major for k x 1)
//   (torch.nn.Conv2d reached from the reference at backbone/backbone.py:67-74 -- InceptionC branch7x7_2/_3, branch7x7dbl_2.._5 -- and its
//    autograd for the data gradient)
//
// Why another kernel (DESIGN 6c / profiles/r04_gather_knockout_1x1.txt): the 128 x 192 gather tile stages 40 KB per 64-deep k-step (a
// fresh copy of the pixel tile for EVERY tap, the filter slab once per 128 pixels), every wave both issues the transfers and consumes
// them, and a workgroup barrier + vmcnt(0) closes every k-step -- issue skeleton, fetch and MFMA time add up.  Here
//   * the pixel LINE of a 256-position tile (+ halo) for one 64-channel block sits in LDS ONCE and serves all taps: tap t reads the same
//     rows shifted by shift_t (the (row >> 1) & 7 chunk swizzle is conflict-free for ANY row shift with the 32-row fragments of
//     v_mfma_f32_32x32x16_bf16: the 16 lanes of a ds_read_b128 group always cover every row residue mod 16 exactly once);
//     staged bytes per 256 pixels and channel block: 34 KB of pixels + 7 x 24 KB of filters = 202 KB instead of 560 KB;
//   * positions whose tap falls outside the image row (walk neighbours that belong to the next row / image) are masked in registers
//     (one v_bfe + four v_and per pixel fragment) -- no padded walk, no wasted MFMA columns;
//   * NL loader waves own every LDS-DMA issue, the address walk and the ring bookkeeping; NC = 8 consumer waves (4 along the pixels x 2
//     along the filters: 64 x 96 per wave = 2 x 3 MFMA tiles, 5 fragment reads per 6 MFMAs) do fragment reads and MFMAs only;
//   * hand-off through monotonic counters in LDS (full / free per ring slot): a consumer never waits on vmcnt or on a barrier, the
//     loader runs up to NFS - 1 filter stages and one pixel line ahead -- across tile boundaries, so the next tile's first stages are
//     in flight while the consumers store the current tile (persistent workgroups, one per CU);
//   * epilogue straight from the accumulators: the filter rows of a stage are permuted on the loader side so that a lane ends up with
//     8 consecutive output channels per accumulator half (16-byte stores, 64-byte runs per pixel) -- no LDS staging, no barrier.
// Every spin is bounded (a lost hand-off traps instead of hanging the GPU).
#include "../../din-group-activity-recognition-benchmark_amd/csrc/din_common.h"
#include "../../din-group-activity-recognition-benchmark_amd/csrc/conv_wgrad.h"
#include "../../din-group-activity-recognition-benchmark_amd/csrc/conv_gather.h"
#include <type_traits>

namespace din_line64 {
using din_gather::ConvK;
using din_wgrad::lds_dma16;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 256;                    // walk positions per tile
constexpr int HALO = 4;                    // LDS row of walk position q0 + x is x + HALO (taps reach +-HALO)
// rows of a pixel line in LDS: positions q0 - 4 .. q0 + 259 are needed (264 rows = 33 wave-level transfers of 8 rows), rounded up so that the
// NL loader waves issue the same number of transfers each (272 rows for NL = 2, 288 for NL = 4)
template <int NL> constexpr int prows() { return (33 + NL - 1) / NL * NL * 8; }
template <int NL> constexpr int pbytes() { return prows<NL>() * 128; }
constexpr int NPB = 2;                     // pixel line buffers (NFS, the filter ring depth, is a template parameter: what LDS is left)
constexpr uint32_t OOB = 0x80000000u;
constexpr int SPIN_LIMIT = 1 << 22;
// flag words (uint32) behind the buffers
constexpr int FL_FULLF = 0, FL_FREEF = 16, FL_FULLP = 32, FL_FREEP = 34, FL_ABORT = 36, FL_WORDS = 40;   // (up to 16 ring slots)

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

template <int BN, int NL, int NFS> constexpr int lds_bytes() { return NPB * pbytes<NL>() + NFS * BN * 128 + FL_WORDS * 4; }

// Flag words are read / bumped with explicit DS instructions on 32-bit LDS byte addresses (a generic pointer would turn the polls into
// flat loads that wait on vmcnt -- in the loader that drains the LDS-DMA queue).
__device__ __forceinline__ uint32_t lds_peek(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ bool wait_ge(uint32_t fl, int idx, uint32_t target) {
    int spins = 0;
    while ((int)(lds_peek(fl + idx * 4) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT || ((spins & 255) == 0 && lds_peek(fl + FL_ABORT * 4))) return false;
    }
    return true;
}
__device__ __forceinline__ void bump(uint32_t fl, int idx, int lane) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fl + idx * 4), "v"(1u) : "memory");
}
__device__ __forceinline__ void give_up(uint32_t fl, int* err, int lane) {          // a hand-off never came: tell the other waves, end this wave
    if (lane == 0) {
        asm volatile("ds_write_b32 %0, %1" ::"v"(fl + FL_ABORT * 4), "v"(1u) : "memory");
        if (err) *err = 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_endpgm" ::: "memory");
    __builtin_unreachable();
}
__device__ __forceinline__ int walk_pixel(const LineK& p, int q) {
    const int t2 = q / p.L, b = q - t2 * p.L;
    const int n = t2 / p.OUTER, a = t2 - n * p.OUTER;
    return n * p.HW + a * p.strideA + b * p.strideB;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// loader waves: every LDS-DMA of the workgroup.  Items in consumption order: P(tile, cb) = the pixel line of a 64-channel block, F(tap)
// = one filter stage (BN rows x 64 channels of one tap); the line of the NEXT block / tile is requested after the third filter stage
// of the current block.  After issuing item j the wave waits until item j - 1 has landed (VMEM completes in order: vmcnt = pieces of
// item j) and publishes it.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int BN, int NC, int NL, int NFS, bool PROF, int KN>
__device__ __forceinline__ void loader(const LineK& p, unsigned char* smem, uint32_t fl, int lw, int lane) {
    constexpr int PROWS = prows<NL>(), PBYTES = pbytes<NL>();
    constexpr int NFP = BN / 8 / NL, NPP = PROWS / 8 / NL;
    static_assert((BN / 8) % NL == 0 && (PROWS / 8) % NL == 0, "pieces divide among the loader waves");
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    const int lrow = lane >> 3, slot = lane & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    int voffF[NFP];
    unsigned voffP[NPP];
    uint32_t nF = 0, nP = 0;
    int prev = -1;                                                 // flag word of the item issued last (not yet published)
    [[maybe_unused]] uint32_t t_free = 0, t_land = 0;
    [[maybe_unused]] const uint64_t T0 = PROF ? __builtin_readcyclecounter() : 0;
    auto timed_wait = [&](int idx, uint32_t target) __attribute__((always_inline)) {
        if constexpr (PROF) {
            const uint64_t c0 = __builtin_readcyclecounter();
            const bool r = wait_ge(fl, idx, target);
            t_free += (uint32_t)(__builtin_readcyclecounter() - c0);
            return r;
        } else return wait_ge(fl, idx, target);
    };

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
    auto publish_prev = [&](auto npieces) __attribute__((always_inline)) {
        if (prev >= 0) {
            [[maybe_unused]] const uint64_t c0 = PROF ? __builtin_readcyclecounter() : 0;
            if constexpr (KN & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(npieces)::value) : "memory");
            if constexpr (PROF) t_land += (uint32_t)(__builtin_readcyclecounter() - c0);
            bump(fl, prev, lane);
        }
    };
    auto issue_line = [&](int cb) __attribute__((always_inline)) {
        const int pb = __builtin_amdgcn_readfirstlane(nP % NPB);
        const uint32_t round = nP / NPB;
        if (round > 0 && !timed_wait(FL_FREEP + pb, (uint32_t)NC * round)) give_up(fl, p.err, lane);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(pb * PBYTES));
        const int soffP = __builtin_amdgcn_readfirstlane(cb * 128);
#pragma unroll
        for (int i = 0; i < NPP; ++i) { if constexpr (!(KN & 4)) lds_dma16(dst + (uint32_t)((lw + NL * i) * 1024), rsA, (int)voffP[i], soffP); }
        publish_prev(std::integral_constant<int, NPP>{});
        prev = FL_FULLP + pb;
        nP = __builtin_amdgcn_readfirstlane(nP + 1);
    };
    auto issue_filter = [&](int tap, int cb) __attribute__((always_inline)) {
        const int fs = __builtin_amdgcn_readfirstlane(nF % NFS);
        const uint32_t round = nF / NFS;
        if (round > 0 && !timed_wait(FL_FREEF + fs, (uint32_t)NC * round)) give_up(fl, p.err, lane);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(NPB * PBYTES + fs * BN * 128));
        const int soff = __builtin_amdgcn_readfirstlane((tap * p.cpt + cb * 8) * 16);
#pragma unroll
        for (int i = 0; i < NFP; ++i) { if constexpr (!(KN & 4)) lds_dma16(dst + (uint32_t)((lw + NL * i) * 1024), rsB, voffF[i], soff); }
        // a filter stage is published the moment it has landed (the ring leaves room for ONE stage in flight: nothing else could be issued
        // meanwhile anyway); a pixel line -- needed several stages later -- is published after the next item went out
        publish_prev(std::integral_constant<int, NFP>{});
        prev = FL_FULLF + fs;
        publish_prev(std::integral_constant<int, 0>{});
        prev = -1;
        nF = __builtin_amdgcn_readfirstlane(nF + 1);
    };

    const int nins = p.taps < 3 ? p.taps : 3;
    bool first = true;
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int px_tile = tile / p.n_co_tiles, co_tile = tile - px_tile * p.n_co_tiles;
        set_filter_rows(co_tile);
        if (first) { set_line(px_tile * TP); issue_line(0); first = false; }
        for (int cb = 0; cb < p.ncb; ++cb) {
            for (int t = 0; t < p.taps; ++t) {
                issue_filter(t, cb);
                if (t == nins - 1) {
                    if (cb + 1 < p.ncb) issue_line(cb + 1);
                    else {
                        const int nxt = tile + gridDim.x;
                        if (nxt < p.ntiles) { set_line((nxt / p.n_co_tiles) * TP); issue_line(0); }
                    }
                }
            }
        }
    }
    if (prev >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bump(fl, prev, lane);
    }
    if constexpr (PROF) {
        if (p.prof && lane == 0) {
            uint32_t* o = p.prof + ((size_t)blockIdx.x * (NC + NL) + NC + lw) * 8;
            o[0] = (uint32_t)(__builtin_readcyclecounter() - T0); o[1] = t_free; o[2] = t_land; o[3] = nF; o[4] = nP;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// consumer waves
// ------------------------------------------------------------------------------------------------------------------------------------
template <int BN, int WM, int WN, int NL, int NFS, bool PROF, int KN>
__device__ __forceinline__ void consumer(const LineK& p, unsigned char* smem, uint32_t fl, int wid, int lane) {
    constexpr int NC = WM * WN, TI = BN / WN / 32, TJ = TP / WM / 32, BNW = BN / WN, PXW = TP / WM, PBYTES = pbytes<NL>();
    static_assert(BN % (32 * WN) == 0 && TP % (32 * WM) == 0 && TJ * 8 <= 32, "wave tile = whole 32 x 32 MFMA tiles");
    typedef volatile __attribute__((address_space(3))) uint32_t* lds_flag_t;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int col = lane & 31, hh = lane >> 5;
    const int frow = wn * BNW + col;
    const uint32_t fbase = (uint32_t)(frow * 128) + (uint32_t)((hh ^ ((frow >> 1) & 7)) << 4);   // chunk (2u + hh) ^ sw = (hh ^ sw) ^ 2u
    const int xrow0 = wm * PXW + col + HALO;
    uint32_t nF = 0, nP = 0;
    [[maybe_unused]] uint32_t t_wait = 0, t_epi = 0, t_all = 0, n_pref = 0, n_block = 0;
    [[maybe_unused]] const uint64_t T0 = PROF ? __builtin_readcyclecounter() : 0;
    auto timed_wait = [&](int idx, uint32_t target) __attribute__((always_inline)) {
        if constexpr (PROF) {
            const uint64_t c0 = __builtin_readcyclecounter();
            const bool r = wait_ge(fl, idx, target);
            t_wait += (uint32_t)(__builtin_readcyclecounter() - c0);
            return r;
        } else return wait_ge(fl, idx, target);
    };

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

        // ---- main loop: one continuous software pipeline over the 16-deep sub-steps of every (channel block, tap) stage ------------------
        // While the MFMAs of a stage's last sub-step run, the first fragments of the NEXT stage are already in flight (its flags were read
        // one sub-step earlier, without waiting); only a stage that is not there yet costs a blocking wait.
        uint32_t pa[TJ];
        u32x4 wf[2][TI], xf[2][TJ];
        if constexpr (KN & 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < TI; ++i) wf[q][i] = u32x4{(uint32_t)lane, 1u, 2u, 3u};
#pragma unroll
                for (int j = 0; j < TJ; ++j) xf[q][j] = u32x4{(uint32_t)lane, 5u, 6u, 7u};
            }
        }
        const unsigned char* Pb = smem;
        const unsigned char* Fs = smem;
        auto set_stage = [&](int t_, uint32_t nP_, uint32_t nF_) __attribute__((always_inline)) {
            Pb = smem + __builtin_amdgcn_readfirstlane((int)(nP_ % NPB) * PBYTES);
            Fs = smem + __builtin_amdgcn_readfirstlane(NPB * PBYTES + (int)(nF_ % NFS) * (BN * 128));
            const int shift = p.shift0 + t_ * p.dshift;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int R = xrow0 + j * 32 + shift;
                pa[j] = (uint32_t)(R * 128) + (uint32_t)((hh ^ ((R >> 1) & 7)) << 4);
            }
        };
        // fragment addresses: sub-step u flips bits 5-6 of the (swizzled) chunk field -- XOR, not add
        auto rdw = [&](int set, int u) __attribute__((always_inline)) {
            if constexpr (KN & 2) return;
#pragma unroll
            for (int i = 0; i < TI; ++i) wf[set][i] = *reinterpret_cast<const u32x4*>(Fs + i * 4096 + (fbase ^ (uint32_t)(u << 5)));
#pragma unroll
            for (int j = 0; j < TJ; ++j) xf[set][j] = *reinterpret_cast<const u32x4*>(Pb + (pa[j] ^ (uint32_t)(u << 5)));
        };
        auto blocking_fetch = [&](int t_, bool need_line) __attribute__((always_inline)) {      // wait for the stage (nP, nF), request its first fragments
            if constexpr (PROF) ++n_block;
            if (need_line && !timed_wait(FL_FULLP + nP % NPB, (uint32_t)NL * (nP / NPB + 1))) give_up(fl, p.err, lane);
            if (!timed_wait(FL_FULLF + nF % NFS, (uint32_t)NL * (nF / NFS + 1))) give_up(fl, p.err, lane);
            set_stage(t_, nP, nF);
            rdw(0, 0);
        };
        const int nst = p.ncb * p.taps;
        int cb = 0, t = 0;
        bool have = false;                                     // set 0 holds (or is receiving) sub-step 0 of the upcoming stage
        blocking_fetch(0, true);
        for (int g = 0; g < nst; ++g) {
            const bool last_tap = t + 1 == p.taps;
            const bool exists = g + 1 < nst;
            const int tn = last_tap ? 0 : t + 1;
            const uint32_t nFn = nF + 1, nPn = last_tap ? nP + 1 : nP;
            const int fs = __builtin_amdgcn_readfirstlane((int)(nF % NFS)), pb = __builtin_amdgcn_readfirstlane((int)(nP % NPB));
            uint32_t mk[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) mk[j] = (uint32_t)__builtin_amdgcn_sbfe((int)vbits, j * 8 + t, 1);
            // a stage is one or two HALVES of two 16-deep sub-steps (fragment sets 0 / 1): one code path for whole and half channel blocks
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
                    for (int j = 0; j < TJ; ++j) {
                        if constexpr (KN & 1) asm volatile("" ::"v"(wf[set][i]), "v"(xf[set][j]));
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[set][i]), __builtin_bit_cast(bf16x8, xf[set][j]), acc[i][j], 0, 0, 0);
                    }
            };
            // peek address: the next stage's flags (this stage's own, already satisfied, when there is no next stage)
            const uint32_t pkFa = fl + (FL_FULLF + (exists ? nFn : nF) % NFS) * 4, pkPa = fl + (FL_FULLP + (exists ? nPn : nP) % NPB) * 4;
            for (int h = 0; h < nh; ++h) {
                const bool last_half = h + 1 == nh;
                // straight-line issue order (no fragment read sits in a branch: the compiler then counts its lgkmcnt waits exactly):
                //   reads(set 1) + flag peeks -> MFMAs(set 0) -> [stage end: release, decide] -> reads(set 0 of what comes next) -> MFMAs(set 1)
                rdw(1, 2 * h + 1);
                const uint32_t pkF = *(lds_flag_t)pkFa, pkP = *(lds_flag_t)pkPa;
                mma(0);
                int un = 2 * h + 2;
                if (last_half) {
                    // every fragment read of this stage has been issued: once they are back the slot (and, after the last tap, the line) is free
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    bump(fl, FL_FREEF + fs, lane);
                    if (last_tap) bump(fl, FL_FREEP + pb, lane);
                    have = exists && (int)(__builtin_amdgcn_readfirstlane(pkF) - (uint32_t)NL * (nFn / NFS + 1)) >= 0 &&
                           (int)(__builtin_amdgcn_readfirstlane(pkP) - (uint32_t)NL * (nPn / NPB + 1)) >= 0;
                    if (have) {
                        if constexpr (PROF) ++n_pref;
                        set_stage(tn, nPn, nFn);
                    }
                    un = 0;
                    asm volatile("" ::: "memory");
                }
                rdw(0, un);                                    // next half / next stage (a stage that is not there yet: a harmless dummy read)
                mma(1);
            }
            nF = __builtin_amdgcn_readfirstlane(nFn); nP = __builtin_amdgcn_readfirstlane(nPn);
            t = tn; cb += last_tap ? 1 : 0;
            if (exists && !have) blocking_fetch(t, t == 0);
        }

        // ---- epilogue: lane holds, per (i, j), output channels cbase + half * 16 + hh * 8 + [0, 8) of pixel column `col` -------------------
        [[maybe_unused]] const uint64_t E0 = PROF ? __builtin_readcyclecounter() : 0;
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
        t_all = (uint32_t)(__builtin_readcyclecounter() - T0);
        if (p.prof && lane == 0) {
            uint32_t* o = p.prof + ((size_t)blockIdx.x * (NC + NL) + wid) * 8;
            o[0] = t_all; o[1] = t_wait; o[2] = t_epi; o[3] = n_pref; o[4] = n_block;
        }
    }
}

// KN (timing knock-outs, tools/probes/line_probe.hip only; results are wrong): 1 = no MFMA, 2 = no fragment reads, 4 = no LDS-DMA
template <int BN, int WM, int WN, int NL, int NFS = 3, bool PROF = false, int KN = 0>
__global__ __launch_bounds__(64 * (WM * WN + NL), 1) void conv_line_kernel(LineK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* flp = reinterpret_cast<uint32_t*>(smem + NPB * pbytes<NL>() + NFS * BN * 128);
    const uint32_t fl = (uint32_t)(uintptr_t)flp;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < FL_WORDS) flp[threadIdx.x] = 0u;
    __syncthreads();
    if (wid >= WM * WN) loader<BN, WM * WN, NL, NFS, PROF, KN>(p, smem, fl, wid - WM * WN, lane);
    else consumer<BN, WM, WN, NL, NFS, PROF, KN>(p, smem, fl, wid, lane);
#endif
}

// ---- host --------------------------------------------------------------------------------------------------------------------------------
// Does the launch described by k (as run_gather sees it) fit this kernel?  Fills lk.
bool plan_line(const ConvK& k, int dtype, int cpt, LineK& lk) {
    if (dtype != DIN_BF16 || k.remap || k.nsrc != 0 || k.xsteps != 0 || k.csplit != 0 || k.u8 || k.craw > 0) return false;
    if (k.divy != 1 || k.divx != 1 || k.ay != 1 || k.ax != 1 || k.out_sy != 0 || k.OH != k.H || k.OW != k.W) return false;
    if (!(k.kh == 1 || k.kw == 1)) return false;
    const int taps = k.kh * k.kw;
    if (taps < 1 || taps > 8 || (cpt & 3)) return false;
    const bool along_x = k.kh == 1;
    if (taps > 1 && (along_x ? k.by != 0 : k.bx != 0)) return false;
    if (taps == 1 && (k.by != 0 || k.bx != 0)) return false;
    const int shift0 = along_x ? k.bx : k.by, dshift = along_x ? k.cx : k.cy;
    for (int t = 0; t < taps; ++t) { const int s = shift0 + t * dshift; if (s < -HALO || s > HALO) return false; }
    if (k.Cout % 8 || k.ldo % 8 || k.cooff % 8 || k.ldi % 8 || k.cioff % 8) return false;
    if ((k.flags & DIN_CONV_MASK) && (k.ldm % 8 || k.moff % 8)) return false;
    if (k.in_bytes >= 0x7fffffffll || k.w_bytes >= 0x7fffffffll || (long long)k.M * k.ldo * 2 >= 0x7fffffffll) return false;
    lk = LineK{};
    lk.in = k.in; lk.w = k.w; lk.out = k.out; lk.bias = k.bias; lk.mask = k.mask; lk.err = nullptr;
    lk.L = along_x ? k.W : k.H; lk.OUTER = along_x ? k.H : k.W; lk.HW = k.H * k.W;
    lk.strideA = along_x ? k.W : 1; lk.strideB = along_x ? 1 : k.W;
    lk.Q = k.M;
    lk.ldi = k.ldi; lk.cioff = k.cioff; lk.ldo = k.ldo; lk.cooff = k.cooff; lk.ldm = k.ldm; lk.moff = k.moff;
    lk.Cout = k.Cout; lk.cpt = cpt; lk.ncb = (cpt + 7) / 8;
    lk.taps = taps; lk.shift0 = shift0; lk.dshift = dshift;
    lk.wld = k.wld; lk.flags = k.flags;
    lk.in_bytes = k.in_bytes; lk.w_bytes = k.w_bytes;
    return true;
}

int launch_line(LineK lk, int bn, int ncu, hipStream_t st, int variant = 0) {
    lk.n_co_tiles = (lk.Cout + bn - 1) / bn;
    lk.ntiles = (lk.Q + TP - 1) / TP * lk.n_co_tiles;
    const int grid = lk.ntiles < ncu ? lk.ntiles : ncu;
    auto go = [&](auto kern, size_t lds, int threads) __attribute__((always_inline)) {
        din_raise_lds(reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, lk);
    };
    // 192 filters: 8 consumer waves (4 x 2: 64 pixels x 96 filters each) + 2 loader waves; 128 filters: the LDS left over buys a 5-slot ring
    if (bn == 192 && lk.prof) go(conv_line_kernel<192, 4, 2, 2, 3, true>, lds_bytes<192, 2, 3>(), 640);
    else if (bn == 192 && variant == 1) go(conv_line_kernel<192, 4, 3, 4, 3>, lds_bytes<192, 4, 3>(), 1024);
    else if (bn == 192) go(conv_line_kernel<192, 4, 2, 2, 3>, lds_bytes<192, 2, 3>(), 640);
    else if (bn == 128 && variant == 1) go(conv_line_kernel<128, 4, 2, 2, 3>, lds_bytes<128, 2, 3>(), 640);
    else if (bn == 128 && variant == 2) go(conv_line_kernel<128, 4, 2, 2, 4>, lds_bytes<128, 2, 4>(), 640);
    else if (bn == 128) go(conv_line_kernel<128, 4, 2, 2, 5>, lds_bytes<128, 2, 5>(), 640);
#ifdef DIN_LINE_KNOCK
    else if (bn == 192 && variant >= 16 && variant < 24) {
        switch (variant - 16) {
            case 1: go(conv_line_kernel<192, 4, 2, 2, 3, false, 1>, lds_bytes<192, 2, 3>(), 640); break;
            case 2: go(conv_line_kernel<192, 4, 2, 2, 3, false, 2>, lds_bytes<192, 2, 3>(), 640); break;
            case 4: go(conv_line_kernel<192, 4, 2, 2, 3, false, 4>, lds_bytes<192, 2, 3>(), 640); break;
            case 7: go(conv_line_kernel<192, 4, 2, 2, 3, false, 7>, lds_bytes<192, 2, 3>(), 640); break;
            default: return -1;
        }
    }
#endif
    else return -1;
    return 0;
}

}  // namespace din_line64
