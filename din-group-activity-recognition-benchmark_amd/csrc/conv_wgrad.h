// Shared between conv_igemm.hip and conv_wgrad_pipe.hip: the weight-gradient kernel argument block and the wave-level LDS-DMA helper.
#pragma once
#include "din_common.h"

namespace din_wgrad {

// wgrad:  dW[co][(r,s,ci)] = sum_pix G[pix][co] * im2col(X)[pix][(r,s,ci)]
struct WgradK {
    const void* in; const void* g; float* partial; float* dbias;
    int NB, H, W, Cin, ldi, cioff;
    int OH, OW, Cout, ldo, cooff;
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int cin_pad, kcols, kcols_pad, cout_pad;   // kcols = kh*kw*cin_pad
    int M, n_co_tiles, n_k_tiles, slices, m_per_slice;
    int probe;      // diagnostics (env DIN_WGRAD_PROBE): 1 = stream only (no transpose reads / MFMA), 2 = compute only (one DMA stage)
    const unsigned char* u8;   // image-layer small kernel only: raw uint8 frames [NB][3][H][W], normalised on load (din_conv_desc::in_u8)
    int atomic;     // pipe kernel: 1 = every workgroup ADDS its tile into slice 0 of `partial` (fp32 atomics, buffer zeroed by the host)
    int pace_base;  // this launch's tag in the progress words: a word counts only inside (pace_base, pace_base + 2^20] (no per-launch memset)
    float* direct;  // pipe kernel, optional: a launch with ONE slice of a 1x1 layer without scale writes its tiles straight into dW [Cout][Cin]
                    // (the partial buffer and the reduce launch exist to add slices and un-permute taps: nothing to do here -- the
                    // 1024 x 26400 embedding layer, 108 MB of fp32 per step, was written, read and written again: 28 + 76 us per step)
    int* pace;      // pipe kernel, optional: [slices * n_co_tiles][n_k_tiles] progress words (zeroed by the host).  The k-tile workgroups of
                    // one (filter tile, pixel slice) stream the same dY rows; each publishes the stage it is at and a workgroup that is
                    // more than PACE_SLACK stages ahead of the slowest sibling naps (bounded), so the siblings stay inside the window an
                    // XCD's L2 holds and dY comes from HBM once.  Speed only: no sibling is ever waited for indefinitely.
};

// conv_wgrad_pipe.hip, grouped launch: the weight gradients of up to WGRAD_GROUP_MAX LAYERS (16 x 184 B of arguments: inside the 4 KB kernel-argument segment) in one launch (din_conv_wgrad_group).  Every launch of
// the pipe kernel fills the chip with one workgroup per CU, and every workgroup writes a full fp32 partial tile: slices x |dW| = ~50 MB per
// layer whatever the batch, read back by the reduce launch.  Layers that share a launch share the 256 workgroups: a layer of a group of six
// is cut into 7 instead of 42 pixel slices and leaves 8 MB of partials.  Logical block l (after the XCD remap) belongs to item g with
// first[g] <= l < first[g + 1]; inside the item it is (tile, slice) exactly as in the single-layer launch.
constexpr int WGRAD_GROUP_MAX = 16;
struct WgradGroupK { WgradK k[WGRAD_GROUP_MAX]; int first[WGRAD_GROUP_MAX + 1]; int n; };

// conv_wgrad_1x1.hip: weight gradients of up to four 1x1 convs that read one tensor, in one launch (dW stationary in registers)
struct Wg1x1K {
    const void* x; float* partial;
    int M, Cin, ldi, cioff;                     // pixels, input channels (192 | 256 | 288), pixel stride, channel offset
    int nsrc, rows_pad;                         // sources; rows of a slab = sum of their couts
    struct Src { const void* g; float* dbias; int cout, ld, coff, row0; } src[4];   // gradient operand [M][ld] at channel coff; first slab row
    int cls_src[2][2];                          // the (<= 2) sources of each workgroup class, -1 = none
    int cls_gbytes[2], gbytes_max;              // bytes of a stage's G image per class
};

// One wave-level LDS-DMA: 64 lanes x 16 B land at LDS byte address `lds_addr` + lane*16 (lane-linear; out-of-range lanes write
// zeros -- measured, profiles/r01_probe_lds_dma.txt).  Issued through inline asm on purpose: with the builtin, hipcc tracks the LDS
// write, cannot tell the ring stages apart and drains vmcnt(0) before the next barrier/ds_read, which serialises the pipeline.
// Here the compiler does not see the transfer at all; completion is counted by hand (s_waitcnt vmcnt(N) + s_barrier in the loop).
// M0 carries the LDS destination; it is declared clobbered (3 instructions per transfer instead of 5 with a save / restore pair).
__device__ __forceinline__ void lds_dma16(uint32_t lds_addr, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

// host entry of conv_wgrad_pipe.hip: launches the software-pipelined 32x32x16 ring kernel for a plan with pipe != 0
int launch_wgrad_pipe(const WgradK& k, int bco, int bk, dim3 grid, hipStream_t st);
int launch_wgrad_pipe_group(const WgradGroupK& g, int bco, bool wide, hipStream_t st);     // 16-wave instantiations only; grid = g.first[g.n] blocks
size_t wgrad_pipe_lds_bytes(int bco, int bk);
// host entries of conv_wgrad_halo.hip: the halo-tiled weight gradient of the narrow mid-network layers (dW block stationary in registers)
bool wgrad_halo_shape(int cin, int cout, int kh, int kw, int* bnt);
int launch_wgrad_halo(const WgradK& k, int nwg, hipStream_t st);
// host entries of conv_wgrad_1x1.hip
bool plan_wgrad_1x1_multi(int nsrc, const int* couts, int cin, Wg1x1K* k);
int launch_wgrad_1x1_multi(const Wg1x1K& k, int nwg, hipStream_t st);

}  // namespace din_wgrad
