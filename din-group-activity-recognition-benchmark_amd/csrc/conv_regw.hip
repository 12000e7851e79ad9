// conv1x1_regw_kernel: 1x1 / stride-1 convolutions with a reduction of 640..768 channels over a large map -- the block ENTRIES of Inception's
// Mixed_6b..6e (reference backbone/backbone.py:67-74 runs torchvision's InceptionC: branch1x1, branch7x7_1, branch7x7dbl_1 and the branch_pool
// conv all read the 768-channel block input; their data gradients land on it again) and, in the short-reduction form at the end of this comment,
// of Mixed_5b..5d (InceptionA, 192 / 256 / 288 channels).  Forward: the sibling group 768 -> 192 + c7 + c7 (+ 192: the commuted branch_pool conv
// as a raw-stored fourth sibling) as one launch with two destinations.  Backward: din_conv1x1_dgrad_multi, 2..4 gradient sources -> 768 channels
// with the fused ReLU-backward mask.
//
// Why another kernel (profiles/r05_slab_stream_probe.txt, profiles/r05_conv1x1_regw.txt): the 128 x 192 tile kernels re-read a 24 KB filter slab
// from L2 per k-step and workgroup; that traffic, not HBM, holds these launches at 4.0 TB/s / 690-770 TF.  Here the FILTERS ARE RESIDENT IN
// REGISTERS: a persistent workgroup of FOUR waves (one per SIMD, 512 registers each) owns 192 filters -- 48 per wave x K <= 768 = up to 288 VGPRs
// per lane, loaded once as ready-made A fragments of v_mfma_f32_16x16x32_bf16 -- and only pixels move: 128-pixel x 64-channel stages through a
// 9-slot LDS ring (144 KB, eight stages in flight), filled by LDS-DMA with counted vmcnt and ONE s_barrier per stage; every wave multiplies the
// whole pixel stage with its own filters (16 fragment reads per 48 MFMAs, a rolling window of four fragments ahead of the MFMAs, issued as inline
// asm with hand-counted lgkmcnt: with the register file full the compiler collapses any prefetch it is given).
// More than 192 filters: CLASSES of 192.  The classes of one TEAM sit on CUs of one XCD (workgroup b -> XCD b & 7) and walk the same tile sequence
// at the same pace, so the pixel stages of all but the first to arrive come out of that XCD's L2 -- HBM sees the pixels once.
//   Same box, 96 frames (profiles/r05_conv1x1_regw.txt): 768 -> 192 161 -> 140 us, 768 -> 576 386 -> 314 us, masked 768 -> 768 dgrad 571 -> 457 us.
//   Tried and dropped there: the four transfers of a step spread among its MFMAs instead of in front of them (+-1 %).
// Layout of a pixel stage: [128 pixels][128 B], 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7): conflict-free for the 16-row
// ds_read_b128 fragments; every transfer moves whole 128-byte lines (a first version with 64-byte half rows -- sources switching at k-step
// granularity -- ran 30 % slower: twice the line requests).  The sources of a multi-source launch switch at STAGE granularity; a source whose
// channels end inside its last stage (the 160-channel gradients of Mixed_6c / 6d) is padded with zeros on both operands.
// The ReLU-backward mask of a gradient launch travels through the SAME ring: after the pixel stages of a tile come four mask stages
// ([32 pixels][192 channels] each), consumed by the four passes of the epilogue -- prefetched eight stages ahead like everything else, no
// second counter, no exposed HBM latency in the epilogue.
// The epilogue goes through 4 KB of wave-private LDS (the MFMA result layout gives a lane 24 contiguous bytes; stored directly, the write path
// sees 64 scattered 8-byte pieces per instruction: 24 % of the kernel) and leaves as 16-byte stores of whole 96-byte runs.
// Short reductions (<= 10 k-steps: Mixed_5): classes of 128 filters (RT = 2) need <= 80 filter + 64 accumulator registers, so TWO workgroups of
// 80 KB LDS (4-slot ring) share a CU and one's epilogue overlaps the other's MFMAs -- which the one-wave-per-SIMD long form cannot have.
// Counters against the tile kernel (profiles/r05_pmc_regw.txt): pixels fetched from HBM once per team, 3.0 instead of 7.5 instructions per MFMA.
#include "conv_gather.h"
#include <atomic>

namespace din_gather {
namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int RW_TPX = 128;                    // pixels per tile
constexpr int RW_STAGE = 16384;                // bytes per ring slot
constexpr int RW_D = 4;                        // fragment reads in flight ahead of the MFMAs
constexpr uint32_t RW_OOB = 0x80000000u;
constexpr size_t rw_lds(int ns) { return (size_t)ns * RW_STAGE + 16384; }     // ring + 4 KB of staging per wave

struct RegwSrc { const void* in; const void* w; unsigned in_bytes, w_bytes; int ld, coff, wld, st0, cpt; };   // st0: first 64-channel stage of the source; cpt: its 16-byte chunks per pixel
struct RegwK {
    RegwSrc src[4];
    int nsrc;
    void* out; void* out2; const float* bias; const void* mask;
    unsigned mask_bytes;
    int M, ldo, cooff, ldo2, cooff2, csplit, Cout, flags, ldm, moff, ntiles, ncls;
    int craw;                        // > 0: produced channels >= craw get neither bias nor ReLU (ConvK::craw)
};

__device__ __forceinline__ void rw_dma16(uint32_t lds_addr, __amdgpu_buffer_rsrc_t rs, uint32_t voff, int soff) {   // see din_wgrad::lds_dma16
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

// NKS: 32-channel k-steps (filter registers: 12 NKS per lane); NS: ring slots; OCC: workgroups per CU -- 1 with the 640..768-channel reductions (the
// filters fill the register file), 2 with the 192..288-channel reductions of Mixed_5 (<= 120 filter registers: two workgroups of 80 KB LDS, one
// in its epilogue while the other multiplies)
// RT: 16-row filter tiles per wave (4 waves x 16 RT = 192 | 128 filters per workgroup class)
template <int NKS, bool MASKED, int NS, int OCC, int RT>
__global__ __launch_bounds__(256, OCC) void conv1x1_regw_kernel(RegwK p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(NKS % 2 == 0 && NKS <= 24, "whole 64-channel stages, <= 288 filter registers");
    static_assert(NS >= 4 && (NS - 2) * 4 <= 63, "ring depth");
    constexpr int D = RW_D, STAGE = RW_STAGE;
    constexpr int PITCH = 32 * RT + 16, BIAS_OFF = 32 * PITCH;     // staging: 32 pixels x 32 RT bytes per wave (+ 16: ds_write_b64 without 4-way conflicts), bias behind it
    static_assert(RT == 2 || RT == 3, "pieces per pixel");
    constexpr int NST = NKS / 2, STEPS = NST + (MASKED ? 4 : 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int frow = lane & 15, g = lane >> 4;
    // ---- (team, class) of this workgroup: it runs on XCD b & 7; its slot there is b >> 3
    const int ncls = p.ncls, xcd = (int)(blockIdx.x & 7), xslot = (int)(blockIdx.x >> 3), per_xcd = (int)(gridDim.x >> 3);
    const int full = per_xcd / ncls, nteams = full * 8 + ((per_xcd - full * ncls) * 8) / ncls;
    int cls, team;
    if (xslot < full * ncls) { cls = xslot % ncls; team = (xslot / ncls) * 8 + xcd; }
    else { const int q = (xslot - full * ncls) * 8 + xcd; cls = q % ncls; team = full * 8 + q / ncls; }   // left-over CUs: teams across XCDs
    if (team >= nteams || team >= p.ntiles) return;
    const int gstep = __builtin_amdgcn_readfirstlane(nteams);
    const int cbase = cls * 64 * RT, wbase = cbase + wid * 16 * RT;               // first filter of the class / of the wave

    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsM =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(MASKED ? p.mask : p.src[0].in), 0, (int)(MASKED ? p.mask_bytes : 0u), 0x00020000);
    // sources change at 64-channel STAGE granularity (128-byte pixel rows: whole cache lines per transfer; 64-byte rows cost 30 % -- profiles/
    // r05_conv1x1_regw.txt); a source whose channels end in the middle of its last stage is padded with zeros on BOTH operands
    const int sb1 = p.nsrc > 1 ? p.src[1].st0 : NKS, sb2 = p.nsrc > 2 ? p.src[2].st0 : NKS, sb3 = p.nsrc > 3 ? p.src[3].st0 : NKS;

    // ---- the wave's filters as A fragments: MFMA row 4 g' + e' of row tile rt is channel wbase + g' * 4 RT + rt * 4 + e' (a lane of the result
    // then holds 4 RT consecutive channels of one pixel)
    u32x4 A[RT][NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int st = ks >> 1, s = (st >= sb1) + (st >= sb2) + (st >= sb3);
        // (the source is picked with scalar selects on pointer / extent and the descriptor rebuilt: a ternary on descriptors becomes control flow)
        const void* wp = s == 0 ? p.src[0].w : s == 1 ? p.src[1].w : s == 2 ? p.src[2].w : p.src[3].w;
        const unsigned wb = s == 0 ? p.src[0].w_bytes : s == 1 ? p.src[1].w_bytes : s == 2 ? p.src[2].w_bytes : p.src[3].w_bytes;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wp), 0, (int)wb, 0x00020000);
        const int wld = s == 0 ? p.src[0].wld : s == 1 ? p.src[1].wld : s == 2 ? p.src[2].wld : p.src[3].wld;
        const int cpt = s == 0 ? p.src[0].cpt : s == 1 ? p.src[1].cpt : s == 2 ? p.src[2].cpt : p.src[3].cpt;
        const int ksl = ks - 2 * (s == 0 ? 0 : s == 1 ? sb1 : s == 2 ? sb2 : sb3);          // k-step within the source
        const bool real = ksl * 4 + g < cpt;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int chan = wbase + (frow >> 2) * 4 * RT + rt * 4 + (frow & 3);
            const int rowoff = (chan < p.Cout && real) ? chan * wld * 16 + g * 16 : (int)RW_OOB;
            A[rt][ks] = __builtin_amdgcn_raw_buffer_load_b128(rw, rowoff, ksl * 64, 0);
        }
    }

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // ---- transfers.  The stage issued at the top of step j of a tile is stage j + NS - 1 of the workgroup's sequence: step (j + NS - 1) % STEPS of
    // tile + ((j + NS - 1) / STEPS) * gstep -- both known at compile time in the unrolled step loop.  Always four transfers per wave and stage.
    //   pixel stage: [128 pixels][128 B], chunk c of row r at chunk position c ^ ((r >> 1) & 7); this wave moves rows wid * 32 + t * 8 + (lane >> 3)
    //   mask stage q: [32 pixels][384 B] of the class's 192 channels; this wave moves chunks (3 wid + t) * 64 + lane, t < 3 (+ one empty transfer)
    // stage table: lane j < NST holds pixel stage j's source (pointer, extent, row pitch in bytes, byte offset of the stage within a row,
    // chunks of the source left from this stage on): six VGPRs instead of ~40 scalar loads / selects per stage in the wave that issues the MFMAs
    uint32_t tab_plo, tab_phi, tab_bytes, tab_ld2, tab_soff, tab_left;
    {
        const int j = lane < NST ? lane : NST - 1, s = (j >= sb1) + (j >= sb2) + (j >= sb3);
        const RegwSrc& sc = p.src[s];
        const int stl = j - sc.st0;
        tab_plo = (uint32_t)(uintptr_t)sc.in; tab_phi = (uint32_t)((uintptr_t)sc.in >> 32); tab_bytes = sc.in_bytes;
        tab_ld2 = (uint32_t)(sc.ld * 2); tab_soff = (uint32_t)(stl * 128 + sc.coff * 2); tab_left = (uint32_t)(sc.cpt - stl * 8);
    }
    int i_slot = 0;
    int vm[4] = {-1, -1, -1, -1};
    const int cl16_even = ((lane & 7) ^ (lane >> 4)) << 4, cl16_odd = ((lane & 7) ^ (4 | (lane >> 4))) << 4;
    auto issue = [&](int tile, int step) __attribute__((always_inline)) {
        asm volatile("" : "+s"(tile));              // opaque: otherwise every (step, row group) address becomes an induction VGPR of the tile loop (~50 registers)
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(i_slot * STAGE));
        const bool live = tile < p.ntiles;
        if (step < NST) {
            // this stage's source: lane `step` of the stage table (v_readlane with a compile-time lane: no kernel-argument loads, no selects)
            const uint32_t plo = __builtin_amdgcn_readlane(tab_plo, step), phi = __builtin_amdgcn_readlane(tab_phi, step);
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)phi << 32) | plo), 0,
                                                                                 (int)__builtin_amdgcn_readlane(tab_bytes, step), 0x00020000);
            const uint32_t ld2 = __builtin_amdgcn_readlane(tab_ld2, step);
            const int soff = (int)__builtin_amdgcn_readlane(tab_soff, step), left = (int)__builtin_amdgcn_readlane(tab_left, step);
            if (step == 0) {                                                              // first stage of a tile: its pixel rows (-1: none)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = tile * RW_TPX + wid * 32 + t * 8 + (lane >> 3);
                    vm[t] = (live && m < p.M) ? m : -1;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cl16 = (t & 1) ? cl16_odd : cl16_even;                          // 16 x logical chunk of this lane: (lane & 7) ^ (r >> 1) & 7 of row r = wid * 32 + t * 8 + (lane >> 3)
                const uint32_t voff = (vm[t] >= 0 && cl16 < left * 16) ? __umul24((uint32_t)vm[t], ld2) + (uint32_t)cl16 : RW_OOB;   // (m, ld2 < 2^24)
                rw_dma16(dst + (uint32_t)(wid * 4096 + t * 1024), rx, voff, soff);
            }
        } else if constexpr (MASKED) {
            const int q = step - NST;
            const int soff = __builtin_amdgcn_readfirstlane((p.moff + cbase) * 2);
            int ln = lane;
            asm volatile("" : "+v"(ln));            // recomputed per use: hoisted out of the tile loop, these addresses cost registers the kernel does not have
#pragma unroll
            for (int t = 0; t < 4; ++t) {                                                 // RT real transfers ([32 pixels][8 RT chunks] = 64 RT lanes' worth), the rest empty
                if (t < RT) {
                    const unsigned ci = (unsigned)((RT * wid + t) * 64 + ln), px = ci / (unsigned)(8 * RT), c = ci - px * (unsigned)(8 * RT);
                    const int m = tile * RW_TPX + q * 32 + (int)px;
                    // (the channel test: the SGPR soffset is outside the descriptor's range check, so a last class narrower than 64 RT channels would read past the mask tensor's end)
                    const uint32_t voff = (live && m < p.M && cbase + (int)c * 8 < p.Cout) ? (uint32_t)(m * p.ldm * 2) + c * 16u : RW_OOB;
                    rw_dma16(dst + (uint32_t)((RT * wid + t) * 1024), rsM, voff, soff);
                } else {
                    rw_dma16(dst + (uint32_t)(4096 * RT + ((4 - RT) * wid + t - RT) * 1024), rsM, RW_OOB, 0);   // keeps the count at four (zeros into the slot's unused tail)
                }
            }
        }
        i_slot = i_slot + 1 == NS ? 0 : i_slot + 1;
        i_slot = __builtin_amdgcn_readfirstlane(i_slot);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(team + (s / STEPS) * gstep, s % STEPS);

    // fragment f of a pixel stage: k-step kk = f >> 3, pixels (f & 7) * 16 + frow: byte ((f & 7) * 16 + frow) * 128 + (((4 kk + g) ^ ((frow >> 1) & 7)) << 4)
    // of the slot (rows j * 16 + frow share (row >> 1) & 7 with frow; conflict-free for the four 16-lane groups of ds_read_b128)
    const uint32_t vb = lds0 + (uint32_t)(frow * 128) + (uint32_t)((g ^ ((frow >> 1) & 7)) << 4);
    u32x4 xf[D];
    auto rdf = [&](u32x4& dst, uint32_t slot_off, int f) __attribute__((always_inline)) {
        const uint32_t addr = ((f >> 3) ? (vb ^ 64u) : vb) + slot_off;
        switch (f & 7) {
#define DIN_RDF(J) case J: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(J * 2048) : "memory"); break;
            DIN_RDF(0) DIN_RDF(1) DIN_RDF(2) DIN_RDF(3) DIN_RDF(4) DIN_RDF(5) DIN_RDF(6) DIN_RDF(7)
#undef DIN_RDF
        }
    };
    auto wait_lgkm = [&](u32x4& x, int n) __attribute__((always_inline)) {
        switch (n) {
            case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x)); break;
            case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(x)); break;
            case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x)); break;
            default: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(x)); break;
        }
    };
    static_assert(D == 4, "wait_lgkm covers a window of four");
    int slot = 0;
    bf16_t* __restrict__ outp = reinterpret_cast<bf16_t*>(p.out);
    bf16_t* __restrict__ outp2 = reinterpret_cast<bf16_t*>(p.out2);
    unsigned char* stg = smem + NS * STAGE + wid * 4096;                             // the wave's own 4 KB: two pixel fragments x 96 B, row pitch 112
    const bool relu = !MASKED && (p.flags & DIN_CONV_RELU) != 0;                 // (masked launches are gradient launches: no bias, no ReLU, one destination)

    // the wave's 48 bias values sit behind its staging rows: read from global memory inside the tile loop they would make the compiler wait for
    // vmcnt(0) -- every transfer in flight -- once per tile (measured: 768 -> 576 filters 311 -> 285 us)
    if (lane < 16 * RT) {
        const int c = wbase + lane;
        *reinterpret_cast<float*>(stg + BIAS_OFF + lane * 4) = (!MASKED && (p.flags & DIN_CONV_BIAS) && c < p.Cout && (p.craw == 0 || c < p.craw)) ? p.bias[c] : 0.f;
    }
    f32x4 acc[RT][8];
    // one pass of the epilogue: pixel fragments 2 ps and 2 ps + 1 (32 pixels) of the tile; mslot: the mask stage [32][384 B] of these pixels
    auto epilogue_pass = [&](int tile, int ps, [[maybe_unused]] const unsigned char* mslot) __attribute__((always_inline)) {
        int relu_o = relu ? 1 : 0, csplit = MASKED ? 0 : p.csplit, craw = MASKED ? 0 : p.craw;
        asm volatile("" : "+s"(relu_o), "+s"(csplit), "+s"(craw));          // (opaque: no loop unswitching on launch flags)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                asm volatile("" : "+a"(acc[rt][ps * 2 + jj]));        // pinned in its accumulation registers until HERE: the scheduler otherwise copies the
                f32x4 v = acc[rt][ps * 2 + jj];                        // accumulators of all four passes into VGPRs early (+80 live registers -> filter spills)
                if (!MASKED && relu_o) {
                    const float lo = (craw > 0 && wbase + g * 4 * RT + rt * 4 >= craw) ? -__builtin_inff() : 0.f;      // (raw-stored sibling: no clamp)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo);
                }
                *reinterpret_cast<u32x2*>(stg + (jj * 16 + frow) * PITCH + g * 8 * RT + rt * 8) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                asm volatile("" ::: "memory");        // one piece at a time: scheduled freely, the epilogue's temporaries push filters out of the register file
            }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        int ln = lane;
        asm volatile("" : "+v"(ln));                // (as in issue(): keep the piece addresses out of the tile loop's live registers)
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const unsigned fi = (unsigned)(ln + 64 * i), px = fi / (unsigned)(2 * RT), pc = fi - px * (unsigned)(2 * RT);   // 32 pixels x 2 RT sixteen-byte pieces
            u32x4 v = *reinterpret_cast<const u32x4*>(stg + px * PITCH + pc * 16);
            if constexpr (MASKED) {                                                                  // ReLU backward: keep where the forward output was > 0
                const u32x4 mk = *reinterpret_cast<const u32x4*>(mslot + px * (128 * RT) + (wid * 2 * RT + (int)pc) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)                                                          // bf16 > 0: sign clear and not zero, per half
                    v[e] &= ((int32_t)mk[e] >= 0x00010000 ? 0xffff0000u : 0u) | ((int32_t)(mk[e] << 16) >= 0x00010000 ? 0x0000ffffu : 0u);
            }
            const int m = tile * RW_TPX + ps * 32 + (int)px, c = wbase + (int)pc * 8;
            if (m < p.M && c < p.Cout) {
                if (!MASKED && csplit > 0 && c >= csplit) *reinterpret_cast<u32x4*>(outp2 + (int64_t)m * p.ldo2 + p.cooff2 + (c - csplit)) = v;
                else *reinterpret_cast<u32x4*>(outp + (int64_t)m * p.ldo + p.cooff + c) = v;
            }
            asm volatile("" ::: "memory");
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };

    if constexpr (!MASKED) {                                                         // the fragment window runs across tile boundaries: prime it once
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 4) : "memory");         // stage 0 of the first tile
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int f = 0; f < D; ++f) rdf(xf[f], 0u, f);
    }
    for (int tile = team; tile < p.ntiles; tile += gstep) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(stg + BIAS_OFF + (g * 4 * RT + rt * 4) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[rt][j] = b;
        }
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            // before barrier j every wave has waited for its share of stage j + 1: after it, stages <= j + 1 are complete and the slot of stage
            // j - 1 is free.  (Stores and filter loads are older or younger entries of the same counter: they only make the wait conservative.)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * 4) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(tile + ((st + NS - 1) / STEPS) * gstep, (st + NS - 1) % STEPS);
            const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(slot * STAGE);
            slot = slot + 1 == NS ? 0 : slot + 1;
            const uint32_t sn = (uint32_t)__builtin_amdgcn_readfirstlane(slot * STAGE);
            if (st < NST) {
                if constexpr (MASKED) {
                    if (st == 0) {                                                   // the window restarts per tile (mask stages lie between)
#pragma unroll
                        for (int f = 0; f < D; ++f) rdf(xf[f], so, f);
                    }
                }
#pragma unroll
                for (int f = 0; f < 16; ++f) {
                    const bool last = MASKED && st == NST - 1;                       // no reads beyond the last pixel stage of the tile
                    wait_lgkm(xf[f % D], last && 16 - f < D ? 16 - f - 1 : D - 1);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        acc[rt][f & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[rt][st * 2 + (f >> 3)]), __builtin_bit_cast(bf16x8, xf[f % D]),
                                                                                  acc[rt][f & 7], 0, 0, 0);
                    const int nf = f + D;                                             // (past the very last stage: a harmless read of a landed or free slot)
                    if (nf < 16) rdf(xf[f % D], so, nf);
                    else if (!last) rdf(xf[f % D], sn, nf - 16);
                }
            } else {
                epilogue_pass(tile, st - NST, smem + so);
            }
        }
        if constexpr (!MASKED) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) epilogue_pass(tile, ps, nullptr);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
}

int regw_mode() { const char* e = DIN_OPT("DIN_CONV_REGW"); return e ? atoi(e) : 1; }       // 0: never, 1: where measured to win, 2: every eligible launch

template <int NKS, int NS, int OCC, int RT>
void launch_regw_nks(const RegwK& r, bool masked, int cus, hipStream_t st) {
    if (masked) {
        auto kern = conv1x1_regw_kernel<NKS, true, NS, OCC, RT>;
        din_raise_lds(reinterpret_cast<const void*>(kern), rw_lds(NS));
        hipLaunchKernelGGL(kern, dim3(cus * OCC), dim3(256), rw_lds(NS), st, r);
    } else {
        auto kern = conv1x1_regw_kernel<NKS, false, NS, OCC, RT>;
        din_raise_lds(reinterpret_cast<const void*>(kern), rw_lds(NS));
        hipLaunchKernelGGL(kern, dim3(cus * OCC), dim3(256), rw_lds(NS), st, r);
    }
}

int regw_ksteps(const ConvK& k) {                         // 32-channel k-steps of the launch (every source padded to whole 64-channel stages), 0: not a shape this kernel takes
    int nks = 0;
    const int ns = k.nsrc > 0 ? k.nsrc : 1;
    for (int s = 0; s < ns; ++s) {
        const int cpt = k.nsrc > 0 ? k.src[s].cpt : k.cpt;
        if (cpt <= 0) return 0;
        nks += (cpt + 7) / 8 * 2;
    }
    return (nks == 6 || nks == 8 || nks == 10 || nks == 20 || nks == 24) ? nks : 0;
}
}  // namespace

bool conv1x1_regw_eligible(const ConvK& k, int dtype) {
    const int mode = regw_mode();
    if (!mode || dtype != DIN_BF16 || k.kh != 1 || k.kw != 1 || k.ay != 1 || k.ax != 1 || k.by != 0 || k.bx != 0 || k.divy != 1 || k.divx != 1 ||
        k.remap || k.out_sy != 0 || k.u8 || k.xsteps != 0 || k.craw % 4 != 0 || ((k.flags & (DIN_CONV_MASK | DIN_CONV_ACCUM)) && k.craw != 0) || k.M <= 0 || k.M >= (1 << 24) || (k.splitk > 1) || k.nsrc > 4)
        return false;
    if (k.flags & DIN_CONV_ACCUM) return false;
    if (k.Cout <= 0 || k.Cout > 4 * 64 * (regw_ksteps(k) <= 10 ? 2 : 3) || k.Cout % 8 != 0 || k.ldo % 8 != 0 || k.cooff % 8 != 0) return false;
    if ((k.flags & DIN_CONV_MASK) && (!k.mask || k.ldm % 8 != 0 || k.moff % 8 != 0 || (long long)k.M * k.ldm * 2 >= 0x7fffffffll || k.csplit > 0)) return false;
    if (k.csplit > 0 && (k.csplit % 8 != 0 || k.ldo2 % 8 != 0 || k.cooff2 % 8 != 0 || !k.out2)) return false;
    if (!regw_ksteps(k)) return false;
    const int ns = k.nsrc > 0 ? k.nsrc : 1;
    for (int s = 0; s < ns; ++s) {
        const int ld = k.nsrc > 0 ? k.src[s].ld : k.ldi, coff = k.nsrc > 0 ? k.src[s].coff : k.cioff;
        const long long ib = k.nsrc > 0 ? k.src[s].in_bytes : k.in_bytes, wb = k.nsrc > 0 ? k.src[s].w_bytes : k.w_bytes;
        if (ld % 8 != 0 || coff % 8 != 0 || ib <= 0 || wb <= 0 || ib >= 0x7fffffffll || wb >= 0x7fffffffll || (long long)k.M * ld * 2 >= 0x7fffffffll || ld * 2 >= (1 << 24)) return false;
    }
    if (mode == 2) return true;
    // measured window (profiles/r05_conv1x1_regw.txt): every launch pays ~9 us for loading its filters into the registers of all CUs
    const char* mp = DIN_OPT("DIN_CONV_REGW_MINPIX");
    if (regw_ksteps(k) <= 10) {                                                      // Mixed_5: against the streaming kernel (conv_stream.hip) / the 128-pixel tiles
        const char* sv = DIN_OPT("DIN_CONV_REGW_SHORT");
        const int sm = sv ? atoi(sv) : 1;                                            // 0: never, 1: launches of more than 96 filters, 2: all
        return sm && (long long)k.M >= (mp ? atoll(mp) : 128 * 1024) && (k.Cout > 96 || sm == 2);      // (4 clips, 164 K pixels: 9.60 -> 9.52 ms per step)   // (<= 96 filters: conv1x1_stream_kernel is faster, 154 vs 169 us)
    }
    return (long long)k.M >= (mp ? atoll(mp) : 64 * 1024);                           // (8 clips, 80 K pixels: 15.40 -> 15.29 ms; 40 K pixels: neutral)
}

int launch_conv1x1_regw(const ConvK& k, hipStream_t st) {
    RegwK r{};
    const int ns = k.nsrc > 0 ? k.nsrc : 1;
    int ks = 0;
    for (int s = 0; s < ns; ++s) {
        RegwSrc& o = r.src[s];
        if (k.nsrc > 0) { o.in = k.src[s].in; o.w = k.src[s].w; o.in_bytes = (unsigned)k.src[s].in_bytes; o.w_bytes = (unsigned)k.src[s].w_bytes; o.ld = k.src[s].ld; o.coff = k.src[s].coff; o.wld = k.src[s].wld; }
        else { o.in = k.in; o.w = k.w; o.in_bytes = (unsigned)k.in_bytes; o.w_bytes = (unsigned)k.w_bytes; o.ld = k.ldi; o.coff = k.cioff; o.wld = k.wld; }
        o.cpt = k.nsrc > 0 ? k.src[s].cpt : k.cpt;
        o.st0 = ks / 2;
        ks += (o.cpt + 7) / 8 * 2;
    }
    r.nsrc = ns;
    r.out = k.out; r.out2 = k.out2; r.bias = k.bias; r.mask = k.mask;
    r.mask_bytes = (k.flags & DIN_CONV_MASK) ? (unsigned)((long long)k.M * k.ldm * 2) : 0u;
    r.M = k.M; r.ldo = k.ldo; r.cooff = k.cooff; r.ldo2 = k.ldo2; r.cooff2 = k.cooff2; r.csplit = k.csplit; r.Cout = k.Cout;
    r.flags = k.flags; r.ldm = k.ldm; r.moff = k.moff; r.craw = k.craw;
    r.ntiles = (k.M + RW_TPX - 1) / RW_TPX;
    const int cw = ks <= 10 ? 128 : 192;                      // filters per class (RT = 2 | 3)
    r.ncls = (k.Cout + cw - 1) / cw;
    static std::atomic<int> cus_of[64];                          // CUs per device (a multiple of 8: the team layout counts workgroups per XCD), cached
    int dev = 0;
    (void)hipGetDevice(&dev);
    int cus = dev >= 0 && dev < 64 ? cus_of[dev].load(std::memory_order_relaxed) : 0;
    if (!cus) {
        int n = 0;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cus = n >= 8 ? n / 8 * 8 : 8;
        if (dev >= 0 && dev < 64) cus_of[dev].store(cus, std::memory_order_relaxed);
    }
    const bool masked = (k.flags & DIN_CONV_MASK) != 0;
    switch (ks) {
        case 6: launch_regw_nks<6, 4, 2, 2>(r, masked, cus, st); break;
        case 8: launch_regw_nks<8, 4, 2, 2>(r, masked, cus, st); break;
        case 10: launch_regw_nks<10, 4, 2, 2>(r, masked, cus, st); break;
        case 20: launch_regw_nks<20, 9, 1, 3>(r, masked, cus, st); break;
        case 24: launch_regw_nks<24, 9, 1, 3>(r, masked, cus, st); break;
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace din_gather
