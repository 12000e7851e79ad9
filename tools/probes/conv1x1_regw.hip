// 1x1 convolution with a long reduction and <= 192 filters (the 768-channel block entries of Inception's Mixed_6: reference backbone/backbone.py:67-74,
// InceptionC branch1x1 / branch7x7_1 / branch7x7dbl_1 / branch_pool) with the FILTER OPERAND RESIDENT IN REGISTERS: experiment, not part of libdin_hip.so.
//   D[co][pix] = sum_k W[co][k] * X[pix][k]
// profiles/r05_slab_stream_probe.txt: the pixel stream of these launches alone runs at 6.1 TB/s; the 192-filter tile's filter transfers (every
// workgroup re-reads the same 24 KB slab from L2 per k-step) take it to 4.5-4.9.  Here a persistent workgroup of FOUR waves (one per SIMD, up to 512
// registers each) loads its 48 filters x K once -- 288 VGPRs per lane at K = 768, as ready-made A fragments of v_mfma_f32_16x16x32_bf16 -- and then
// only pixels move: 128-pixel x 64-channel stages through a 9-slot LDS ring (144 KB, eight stages in flight), filled by LDS-DMA with counted
// vmcnt + one s_barrier per stage; every wave multiplies the whole pixel stage with its own filters (16 fragment reads per 48 MFMAs).
// The filter rows are assigned to MFMA rows so that a lane holds 12 consecutive channels; the epilogue goes through 4 KB of wave-private LDS.
#include "../../din-group-activity-recognition-benchmark_amd/csrc/din_common.h"
#include "../../din-group-activity-recognition-benchmark_amd/csrc/conv_wgrad.h"

namespace din_regw {
using din_wgrad::lds_dma16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int TPX = 128, STAGE = TPX * 128;
constexpr uint32_t OOB = 0x80000000u;

struct RegwK {
    const void* in; const void* w; void* out; const float* bias;
    int M, ldi, cioff, ldo, cooff, Cout, wld, flags, ntiles, co_base;      // wld: packed filter row length in 16-byte chunks; co_base: first filter of this launch
    long long in_bytes, w_bytes;
    uint32_t* prof;                                                           // optional: [workgroup][wave][8] cycle counts
    int ncls;                                                                 // > 1: filter classes of 16 RT x 4 filters; the classes of one TEAM run on CUs of one XCD (shared L2) over the same tiles
};

template <int NKS, int RT, int NS, bool PROF = false, int D = 4, int KNOCK = 0>   // KNOCK (timing experiments, wrong results): 1 no MFMAs, 2 no transfers in the loop, 3 no fragment reads, 4 no barriers, 5 no epilogue      // 32-deep k-steps (K / 32), 16-row filter tiles per wave, ring slots
__global__ __launch_bounds__(256, 1) void conv1x1_regw_kernel(RegwK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(NKS % 2 == 0 && NS >= 4 && (NS - 2) * 4 <= 63, "whole 64-channel stages");
    constexpr int NST = NKS / 2;                                   // stages per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int frow = lane & 15, g = lane >> 4;
    // workgroup b runs on XCD b & 7 (round-robin dispatch); its slot there, b >> 3, is (team, class): the classes of a team walk the same tile
    // sequence at the same pace, so the pixel stages of all but the first to arrive come out of that XCD's L2
    const int ncls = p.ncls > 1 ? p.ncls : 1, xslot = (int)(blockIdx.x >> 3), cls = ncls > 1 ? xslot % ncls : 0;
    const int team = ncls > 1 ? (xslot / ncls) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x;
    const int nteams = ncls > 1 ? ((int)(gridDim.x >> 3) / ncls) * 8 : (int)gridDim.x;
    if (team >= nteams) return;
    const int wbase = p.co_base + cls * 64 * RT + wid * 16 * RT;
    [[maybe_unused]] uint32_t t_w = 0, t_vm = 0, t_bar = 0, t_epi = 0;
    [[maybe_unused]] const uint64_t T0 = PROF ? __builtin_readcyclecounter() : 0;
    // ---- the wave's filters, as A fragments: MFMA row r = 4 g' + e' of row tile rt is channel wbase + g' * 4 RT + rt * 4 + e'
    u32x4 A[RT][NKS];
    {
        __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int chan = wbase + (frow >> 2) * 4 * RT + rt * 4 + (frow & 3);
            const int rowoff = chan < p.Cout ? chan * p.wld * 16 + g * 16 : (int)OOB;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) A[rt][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsW, rowoff, ks * 64, 0);
        }
    }
    if constexpr (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_w = (uint32_t)(__builtin_readcyclecounter() - T0); }
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // ---- transfer generator: stage s of this workgroup = (tile, 64-channel block); this wave moves rows wid * 32 + t * 8 + (lane >> 3), t < 4
    const int gstep = __builtin_amdgcn_readfirstlane(nteams);
    int i_tile = team, i_blk = 0, i_slot = 0;
    unsigned voff[4];
    auto set_rows = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = wid * 32 + t * 8 + (lane >> 3), m = tile * TPX + r;
            voff[t] = (tile < p.ntiles && m < p.M) ? (unsigned)(m * p.ldi * 2 + p.cioff * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4)) : OOB;
        }
    };
    set_rows(i_tile);
    auto issue = [&]() __attribute__((always_inline)) {            // always four transfers (past the last stage: out of range, into the free slot)
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(i_slot * STAGE + wid * 4096));
        const int soff = __builtin_amdgcn_readfirstlane(i_blk * 128);
#pragma unroll
        for (int t = 0; t < 4; ++t) lds_dma16(dst + (uint32_t)(t * 1024), rsX, (int)voff[t], soff);
        i_slot = i_slot + 1 == NS ? 0 : i_slot + 1;
        if (++i_blk == NST) { i_blk = 0; i_tile += gstep; set_rows(i_tile); }
        i_slot = __builtin_amdgcn_readfirstlane(i_slot); i_blk = __builtin_amdgcn_readfirstlane(i_blk); i_tile = __builtin_amdgcn_readfirstlane(i_tile);
    };
#pragma unroll 1
    for (int s = 0; s < NS - 1; ++s) issue();

    // fragment f of a stage: kk = f >> 3 (32-deep k-step), pixel fragment j = f & 7: row j * 16 + frow, chunk (4 kk + g) ^ swizzle(row); rows
    // j * 16 + frow share (row >> 1) & 7 with frow.  Fragment reads are inline asm with hand-counted lgkmcnt: a rolling window of D fragments runs
    // ahead of the MFMAs ACROSS stage and tile boundaries (the compiler, left alone, collapses any prefetch to two fragments in flight: the
    // register file is full).  Before barrier j every wave has waited for its share of stage j + 1, so after it stages <= j + 1 are complete.
    const uint32_t vb0 = lds0 + (uint32_t)(frow * 128) + (uint32_t)((g ^ ((frow >> 1) & 7)) << 4), vb1 = vb0 ^ 64u;
    u32x4 xf[D];
    auto rdf = [&](u32x4& dst, uint32_t addr, int j) __attribute__((always_inline)) {
        switch (j) {
#define DIN_RDF(J) case J: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(J * 2048) : "memory"); break;
            DIN_RDF(0) DIN_RDF(1) DIN_RDF(2) DIN_RDF(3) DIN_RDF(4) DIN_RDF(5) DIN_RDF(6) DIN_RDF(7)
#undef DIN_RDF
        }
    };
    int slot = 0;
    bf16_t* __restrict__ outp = reinterpret_cast<bf16_t*>(p.out);
    if (lane < 16 * RT) {
        const int c = wbase + lane;
        *reinterpret_cast<float*>(smem + NS * STAGE + wid * 4096 + 3584 + lane * 4) = ((p.flags & DIN_CONV_BIAS) && c < p.Cout) ? p.bias[c] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 4) : "memory");        // stage 0 of the first tile
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < D; ++f) rdf(xf[f], vb0, f);
    for (int tile = team; tile < p.ntiles; tile += gstep) {
        f32x4 acc[RT][8];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(smem + NS * STAGE + wid * 4096 + 3584 + (g * 4 * RT + rt * 4) * 4);   // (a global load here
            // would make the compiler wait vmcnt(0): every transfer in flight, once per tile)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[rt][j] = b;
        }
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            [[maybe_unused]] const uint64_t c0 = PROF ? __builtin_readcyclecounter() : 0;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * 4) : "memory");      // this wave's share of the NEXT stage has landed ...
            [[maybe_unused]] const uint64_t c1 = PROF ? __builtin_readcyclecounter() : 0;
            if constexpr (KNOCK != 4) __builtin_amdgcn_s_barrier();                  // ... and everyone's; the slot of the previous stage is free
            asm volatile("" ::: "memory");
            if constexpr (PROF) { t_vm += (uint32_t)(c1 - c0); t_bar += (uint32_t)(__builtin_readcyclecounter() - c1); }
            if constexpr (KNOCK != 2) issue();
            const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(slot * STAGE);
            slot = slot + 1 == NS ? 0 : slot + 1;
            const uint32_t sn = (uint32_t)__builtin_amdgcn_readfirstlane(slot * STAGE);
            const uint32_t a0 = vb0 + so, a1 = vb1 + so, n0 = vb0 + sn;
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xf[f % D]) : "n"(D - 1));
                if constexpr (KNOCK == 1) { asm volatile("" ::"v"(xf[f % D])); }
                else
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][f & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[rt][st * 2 + (f >> 3)]), __builtin_bit_cast(bf16x8, xf[f % D]),
                                                                              acc[rt][f & 7], 0, 0, 0);
                const int nf = f + D;                                                 // (past the very last stage: a harmless read of a free slot)
                if constexpr (KNOCK != 3) { if (nf < 16) rdf(xf[f % D], (nf >> 3) ? a1 : a0, nf & 7); else rdf(xf[f % D], n0, nf - 16); }
            }
        }
        [[maybe_unused]] const uint64_t E0 = PROF ? __builtin_readcyclecounter() : 0;
        // ---- epilogue.  Lane holds channels wbase + g * 4 RT + rt * 4 + [0, 4) of pixel tile * 128 + j * 16 + frow: 24 contiguous bytes of the
        // wave's 96 per pixel.  Two pixel fragments at a time go through the wave's private staging and leave as 16-byte stores of whole 96-byte runs
        // (8-byte stores straight from the accumulators: 24 % of the kernel, the write path sees 64 scattered pieces per instruction).
        unsigned char* stg = smem + NS * STAGE + wid * 4096;                         // the wave's own 4 KB: two pixel fragments x 96 B, row pitch 112
#pragma unroll
        for (int ps = 0; ps < (KNOCK == 5 ? 1 : 4); ++ps) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 v = acc[rt][ps * 2 + jj];
                    if (p.flags & DIN_CONV_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    *reinterpret_cast<u32x2*>(stg + (jj * 16 + frow) * 112 + g * 24 + rt * 8) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const unsigned fi = (unsigned)(lane + 64 * i), px = fi / 6u, pc = fi - px * 6u;      // 32 pixels x 6 sixteen-byte pieces
                const u32x4 v = *reinterpret_cast<const u32x4*>(stg + px * 112 + pc * 16);
                const int m = tile * TPX + ps * 32 + (int)px, c = wbase + (int)pc * 8;
                if (m < p.M && c < p.Cout) *reinterpret_cast<u32x4*>(outp + (int64_t)m * p.ldo + p.cooff + c) = v;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (PROF) t_epi += (uint32_t)(__builtin_readcyclecounter() - E0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (PROF) {
        if (p.prof && lane == 0) { uint32_t* o = p.prof + ((size_t)blockIdx.x * 4 + wid) * 8; o[0] = (uint32_t)(__builtin_readcyclecounter() - T0); o[1] = t_w; o[2] = t_vm; o[3] = t_bar; o[4] = t_epi; }
    }
#endif
}

template <int D, int KNOCK = 0>
inline int launch_regw_d(RegwK k, int ncu, hipStream_t st) {          // K = 768, <= 192 filters starting at k.co_base
    k.ntiles = (k.M + TPX - 1) / TPX;
    const int grid = k.ncls > 1 ? ncu : (k.ntiles < ncu ? k.ntiles : ncu);
    constexpr int NS = 9;
    if (k.prof) {
        auto kp = conv1x1_regw_kernel<24, 3, NS, true, D>;
        din_raise_lds(reinterpret_cast<const void*>(kp), NS * STAGE + 16384);
        hipLaunchKernelGGL(kp, dim3(grid), dim3(256), NS * STAGE + 16384, st, k);
        return 0;
    }
    auto kern = conv1x1_regw_kernel<24, 3, NS, false, D, KNOCK>;
    din_raise_lds(reinterpret_cast<const void*>(kern), NS * STAGE + 16384);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), NS * STAGE + 16384, st, k);
    return 0;
}
inline int launch_regw(RegwK k, int ncu, hipStream_t st, int d = 4, int knock = 0) {
    switch (knock) {
        case 1: return launch_regw_d<4, 1>(k, ncu, st);
        case 2: return launch_regw_d<4, 2>(k, ncu, st);
        case 3: return launch_regw_d<4, 3>(k, ncu, st);
        case 4: return launch_regw_d<4, 4>(k, ncu, st);
        case 5: return launch_regw_d<4, 5>(k, ncu, st);
    }
    return d == 8 ? launch_regw_d<8>(k, ncu, st) : launch_regw_d<4>(k, ncu, st);
}
}  // namespace din_regw
