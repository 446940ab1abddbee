// conv.hip -- fused convolution blocks for the SuperPoint / VGG / SiLK encoders on gfx950.
//
// One kernel = Conv2d(3x3 pad 1 | 1x1) + bias -> [ReLU] -> [BN(eval) affine] -> [MaxPool 2x2],
// as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
//   M = output channels (A operand = weights), N = pixels of a TH x TW spatial tile
//   (B operand = im2col patches read straight out of an LDS halo tile), K = (ci>>1, tap, ci&1).
// K is walked in that single fixed order with ONE accumulator chain per output, so every output
// is bit-for-bit the k-ordered fmaf chain of oracle/einx_oracle.c::orc_conv_block.
//
// Why this shape (see DESIGN.md "K1"): fp32 MFMA issues every 64 cycles per SIMD, so one
// ds_read_b32 per operand per MFMA is far below the LDS roof; pixels sit on the MFMA column
// (= lane) index so NCHW stores are 128-byte coalesced; the next chunk's global loads are
// issued into registers before the current chunk's 144 MFMAs (issue-early / write-late).
//
// Replaces (reference file:line): core/modules/net/vgg.py:34-38, net/backbone.py:105-128,
// net/detector_head.py:42-48, net/descriptor_head.py:40-43,
// image_extractors/superpoint_extractor.py:388-406, silk/backbones/superpoint/vgg.py:284-290,
// utils/util.py:17-32 (replicate pad folded into the first layer's addressing).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "einx_common.h"

namespace {

constexpr int kCoutTile = 64;  // output channels per workgroup (2 MFMA M-tiles)

struct ConvArgs {
  const float* in;
  const float* w;  // native [K][CoutPad]
  const float* bias;
  const float* scale;
  const float* shift;
  float* out;
  int B, Cin, Cout, CoutPad;
  int Hs, Ws, h0, w0;  // source tensor geometry (replicate-pad fold)
  int H, W;            // logical input size == conv output size
  int tilesX, tilesY;
  int relu;
};

// Smallest LDS row pitch >= tw + 2 for which every 32-aligned run of a th x tw tile's pixel slots (slot q = row q / tw,
// column q % tw: what one half-wave reads with one ds_read_b32 per K-step, at any tap shift) lands on 32 distinct banks.
// tw == 32: any pitch (a run is one row).  Otherwise the rows of a run must tile the 32 banks: pitch == tw (mod 32) always
// works, smaller pitches exist when 32 % tw == 0 (tw 8: 24).  With the plain pitch tw + 2 the rows of a run overlap by two
// banks: a 2-way conflict on every B-operand read (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.29-0.38 on the 12x16, 11x22,
// 22x8 and 11x11 tiles in round 2, 0 on 8x32).
constexpr bool pitch_conflict_free(int th, int tw, int pitch) {
  const int npix = th * tw;
  for (int q0 = 0; q0 < npix; q0 += 32) {
    unsigned used = 0;
    for (int q = q0; q < q0 + 32 && q < npix; ++q) {
      const unsigned bit = 1u << (((q / tw) * pitch + q % tw) & 31);
      if (used & bit) return false;
      used |= bit;
    }
  }
  return true;
}
constexpr int conv_lds_pitch(int th, int tw) {
  for (int p = tw + 2; p < tw + 2 + 32; ++p)
    if (pitch_conflict_free(th, tw, p)) return p;
  return tw + 2;
}

// lane ^ X exchange without an LDS round trip where the ISA has one: DPP quad_perm (X = 1), DPP row_ror:8 inside rows of 16
// lanes (X = 8), ds_swizzle SWAP16 (X = 16: crossbar only, no LDS access, no address register)
template <int X>
__device__ __forceinline__ float lane_xor(float v) {
  const int i = __builtin_bit_cast(int, v);
  if constexpr (X == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xF, 0xF, true));
  else if constexpr (X == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x128, 0xF, 0xF, true));
  else if constexpr (X == 16) return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(i, 0x401F));
  else return __shfl_xor(v, X, 64);
}

// max(v, v of lane ^ X) in ONE vector instruction for X = 1 / 8 (v_max_f32 with a DPP operand; the compiler's own sequence is
// v_mov_dpp + a canonicalising v_max + v_max).  `s_nop 1`: a DPP read of a register needs two wait states after the VALU
// write of it, and the hazard recogniser does not look inside asm statements.
template <int X>
__device__ __forceinline__ float max_lane_xor(float v) {
  if constexpr (X == 1) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
    return r;
  } else if constexpr (X == 8) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
    return r;
  } else {
    return fmaxf(v, lane_xor<X>(v));
  }
}

// raw buffer descriptor over `bytes` bytes at p (wave-uniform arguments only: the four words live in SGPRs).  Accesses whose
// byte offset is >= bytes are dropped by the hardware: loads return 0, stores write nothing.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// channel groups the staging threads split a chunk into: the most groups of PLANE_E threads that fit the workgroup and divide CK
constexpr int conv_stage_groups(int nthr, int plane_e, int ck) {
  int ng = nthr / plane_e;
  while (ng > 1 && ck % ng != 0) --ng;
  return ng < 1 ? 1 : ng;
}

// KS: 1|3.  TH x TW: spatial tile (KS==1: flat run of TH*TW pixels).  Waves are arranged
// WM (output-channel groups) x WN (pixel groups); each wave owns MT x NT MFMA tiles of 32x32, with
// WM*MT == 2 (64 output channels per workgroup).  CK: input channels staged per LDS round.
// POOL: fuse MaxPool2d(2,2).
// XTRA (1x1 layers with 64 n + 1 output channels: the detector's 65): the last channel is NOT given a channel tile of its own
// (63 of 64 rows padding: half of the 256 -> 65 layer's matrix-core time); the workgroups of the last whole tile
// accumulate it beside their MFMAs, one pixel per lane, as the same k-ordered fmaf chain from +0 (v_fma_f32 on the LDS tile).
//
// Round 5 -- every vector instruction outside the MFMA loop is paid in matrix-pipe time.  fp32 MFMA and fp32 VALU share the
// SIMD's datapath (tools/coexec_ceiling.hip: one wave of each reaches 65.6 + 65.6 TFLOP/s, not 144 + 118), and across the
// 8-wave variants of round 4 the idle share of the matrix pipe fits ~8 cycles per VALU instruction (conv1b 1.4 VALU per MFMA ->
// 0.84 busy, event-side first layer 10.9 -> 0.42, 1x1 heads 8.8 -> 0.46; profiles/r03_pmc_conv_tiles.json).  So:
//   * staging is position-major: a thread owns ONE halo position (one division by the halo width, one bounds / replicate
//     clamp) and brings in that position of every channel of the chunk.  Channel planes are stepped in the wave-uniform
//     buffer descriptor (scalar ALU), not in per-element vector address arithmetic: one offset register, no 64-bit adds.
//   * zero padding costs nothing per chunk: positions outside the image are zeroed in LDS once and their threads sit out the
//     loads and commits (EXEC mask); channels past Cin fall outside the descriptor and load as 0.  The generic and the
//     "exact" reload paths of rounds 1-4 are one path now (half the instantiations).
//   * epilogue: ReLU / affine flags are hoisted into four straight-line bodies; the per-channel constants come as three
//     ds_read_b128 per 4 rows; the 2x2 pool exchanges lanes by DPP; stores go through a descriptor re-based per output
//     channel in the scalar ALU, pixels outside the image carry an out-of-range offset (no branches, no 64-bit multiplies).
// WPS: waves per SIMD the register allocation must allow.  6 = three 8-wave workgroups per CU (<= 80 registers): the thin first
// layers (their load -> MFMA -> store phases only overlap across workgroups) and launches of many rounds (conv1b at B=32:
// 15.1 rounds of 768 instead of 22.7 of 512, -2.4 %; launches of few rounds lose more to the coarser last round than they gain).
template <int KS, int TH, int TW, int WM, int WN, int MT, int NT, int CK, bool POOL, bool XTRA = false, int WPS = (CK < 8 ? 6 : 1)>
__global__ __launch_bounds__(WM * WN * 64, WPS) void conv_block_kernel(const ConvArgs a) {
  constexpr int kMT = MT, kNT = NT;
  constexpr int NW = WM * WN;
  constexpr int TAPS = KS * KS;
  constexpr int HALO = KS / 2;
  constexpr int PW = TW + 2 * HALO;
  constexpr int PH = TH + 2 * HALO;
  constexpr int PITCH = KS == 1 ? PW : conv_lds_pitch(TH, TW);  // LDS row pitch (bank-conflict-free B reads)
  constexpr int PLANE = PH * PITCH;                             // LDS floats per staged channel
  constexpr int PLANE_E = PH * PW;                              // elements staged per channel
  constexpr int NTHR = NW * 64;
  constexpr int NPIX = TH * TW;
  static_assert(WM * MT * 32 == kCoutTile, "a workgroup covers 64 output channels");
  static_assert(WN * NT * 32 >= NPIX, "tile does not fit the workgroup's pixel slots");
  static_assert(KS == 1 || NPIX >= 32, "unused slots alias the slot one run earlier");
  static_assert(CK % 2 == 0, "channels are consumed in pairs");
  static_assert(PLANE_E <= NTHR, "one staging thread per halo position");
  constexpr int NG = conv_stage_groups(NTHR, PLANE_E, CK);  // channel groups among the staging threads
  constexpr int CPT = CK / NG;                              // channels a staging thread brings in per chunk
  constexpr int IN_LDS = CK * PLANE;
  static_assert(IN_LDS % 4 == 0, "the weight tile behind the input tile is float4 aligned");
  constexpr int W_ROWS = CK * TAPS;
  constexpr int W_F4 = W_ROWS * kCoutTile / 4;
  constexpr int W_PER_THR = (W_F4 + NTHR - 1) / NTHR;
  constexpr int LDS_FLOATS = IN_LDS + W_ROWS * kCoutTile;

  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_bias[kCoutTile], s_scale[kCoutTile], s_shift[kCoutTile];  // epilogue constants
  float* in_tile = lds;
  float* w_tile = lds + IN_LDS;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and provably so for the compiler (scalar registers)
  const int wm = wave / WN, wn = wave % WN;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int j = lane & 31;

  // work item = (pixel tile, output-channel tile), channel tile fastest: the workgroups that read the same input tile and
  // the tiles that share halo rows are neighbours in the XCD-contiguous order (conv1b: FETCH_SIZE 1285 -> 370 MiB per launch)
  const int ncot = (int)gridDim.y;
  int item = xcd_contiguous((int)(blockIdx.x + blockIdx.y * gridDim.x), (int)gridDim.x * ncot);
  const int co0 = (item % ncot) * kCoutTile;
  int bid = item / ncot;
  const int tx_i = bid % a.tilesX;
  bid /= a.tilesX;
  const int ty_i = bid % a.tilesY;
  const int b = bid / a.tilesY;

  if (tid < kCoutTile) {  // visible to everyone after the main loop's first barrier
    const int co = co0 + tid;
    const bool cv = co < a.Cout;
    s_bias[tid] = (cv && a.bias) ? a.bias[co] : 0.0f;
    s_scale[tid] = (cv && a.scale) ? a.scale[co] : 1.0f;
    s_shift[tid] = (cv && a.scale) ? a.shift[co] : 0.0f;
  }
  const int HW = a.H * a.W;
  int y0, x0, p0;
  if (KS == 1) {
    p0 = tx_i * NPIX;  // flat pixel run
    y0 = x0 = 0;
  } else {
    y0 = ty_i * TH;
    x0 = tx_i * TW;
    p0 = 0;
  }

  // ---- per-lane pixel bookkeeping -------------------------------------------------------
  int bBase[kNT];
  int opix[kNT];  // output offset within a channel plane, or -1
  int qidx[kNT];
#pragma unroll
  for (int nt = 0; nt < kNT; ++nt) {
    const int q = (wn * kNT + nt) * 32 + j;
    const bool vq = q < NPIX;
    qidx[nt] = vq ? q : -1;
    if (KS == 1) {
      bBase[nt] = half * PLANE + (vq ? q : 0);
      const int p = p0 + q;
      opix[nt] = (vq && p < HW) ? p : -1;
    } else {
      // an unused slot reads the address of the valid slot one or more whole runs before it: same bank as its own
      // position would have, so it never collides with the valid lanes of its run
      const int qa = vq ? q : q - 32 * ((q - NPIX) / 32 + 1);
      const int ty = qa / TW, tx = qa % TW;
      bBase[nt] = half * PLANE + ty * PITCH + tx;
      const int y = y0 + ty, x = x0 + tx;
      opix[nt] = (vq && y < a.H && x < a.W) ? y * a.W + x : -1;
    }
  }
  const int aBase = half * kCoutTile + wm * MT * 32 + j;

  // ---- staging plan: this thread's halo position, the same in every channel of every chunk -----
  const unsigned src_plane = (unsigned)(a.Hs * a.Ws);
  const float* in_b = a.in + (size_t)b * a.Cin * src_plane;
  const int pos = NG == 1 ? tid : tid % PLANE_E;
  const int grp = NG == 1 ? 0 : tid / PLANE_E;
  const bool s_active = NG * PLANE_E == NTHR || tid < NG * PLANE_E;
  int src_off = -1, lds_pos;
  if (KS == 1) {
    const int p = p0 + pos;
    if (p < HW) src_off = p;  // 1x1 layers never carry the replicate fold
    lds_pos = grp * CPT * PLANE + pos;
  } else {
    const int py = pos / PW, px = pos % PW;
    const int y = y0 - HALO + py, x = x0 - HALO + px;
    if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
      int sy = y - a.h0, sx = x - a.w0;
      sy = sy < 0 ? 0 : (sy > a.Hs - 1 ? a.Hs - 1 : sy);
      sx = sx < 0 ? 0 : (sx > a.Ws - 1 ? a.Ws - 1 : sx);
      src_off = sy * a.Ws + sx;
    }
    lds_pos = grp * CPT * PLANE + py * PITCH + px;
  }
  const bool s_ok = s_active && src_off >= 0;
  // byte offset of this thread's element of channel i (+ CPT grp) of a chunk, from the chunk's first channel plane: the chunk
  // itself is stepped in the descriptor's base (scalar ALU), so these registers never change
  unsigned in_voff[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) in_voff[i] = ((unsigned)(src_off >= 0 ? src_off : 0) + (unsigned)(grp * CPT + i) * src_plane) * 4u;
  if (s_active && src_off < 0) {  // padding: zero once, never written again
#pragma unroll
    for (int i = 0; i < CPT; ++i) in_tile[lds_pos + i * PLANE] = 0.0f;
  }
  unsigned woff[W_PER_THR];  // byte offset of this thread's weight float4 inside a chunk's rows
#pragma unroll
  for (int i = 0; i < W_PER_THR; ++i) {
    const int f = tid + i * NTHR;
    const int r = f / (kCoutTile / 4), c4 = f % (kCoutTile / 4);
    woff[i] = (unsigned)((f < W_F4 ? r : 0) * a.CoutPad + co0 + c4 * 4) * 4u;
  }

  f32x16 acc[kMT][kNT];
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  const int pairs_total = (a.Cin + 1) / 2;
  const int nchunks = (pairs_total * 2 + CK - 1) / CK;
  static_assert(!XTRA || (KS == 1 && NPIX <= NTHR && !POOL), "the extra channel is a 1x1 feature: one pixel per thread");
  float xacc = 0.0f;                                                      // XTRA: the extra channel of pixel p0 + tid
  const bool xtra_wg = XTRA && co0 + kCoutTile == a.CoutPad - kCoutTile;  // the last whole channel tile carries it

  // Register image of one chunk in flight (global -> registers -> LDS): the loads of chunk c + 1 are issued when chunk c has
  // been committed to LDS and land under its MFMAs (issue early, write late).
  struct StageRegs {
    float in[CPT];
    f32x4 w[W_PER_THR];
  };
  StageRegs sA;
  auto issue_loads = [&](int c) {
    const int ci0 = c * CK;
    if (s_ok) {
      // the descriptor spans channels ci0 .. Cin - 1 of this image: channels past Cin lie outside it and load as 0
      const __amdgpu_buffer_rsrc_t rs = conv_rsrc(in_b + (size_t)ci0 * src_plane, (unsigned)(a.Cin - ci0) * src_plane * 4u);
#pragma unroll
      for (int i = 0; i < CPT; ++i) sA.in[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, in_voff[i], 0, 0));
    }
    // the native weight image is zero-padded to whole 32-channel row groups (einx_conv_repack): every row a chunk names exists
    const __amdgpu_buffer_rsrc_t rw = conv_rsrc(a.w + (size_t)c * W_ROWS * a.CoutPad, (unsigned)(W_ROWS * a.CoutPad) * 4u);
#pragma unroll
    for (int i = 0; i < W_PER_THR; ++i)
      if ((i + 1) * NTHR <= W_F4 || tid + i * NTHR < W_F4)
        sA.w[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff[i], 0, 0));
  };
  auto commit_loads = [&]() {
    if (s_ok) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) in_tile[lds_pos + i * PLANE] = sA.in[i];
    }
#pragma unroll
    for (int i = 0; i < W_PER_THR; ++i)
      if ((i + 1) * NTHR <= W_F4 || tid + i * NTHR < W_F4) *reinterpret_cast<f32x4*>(w_tile + (tid + i * NTHR) * 4) = sA.w[i];
  };
  // 36 K-steps (CK/2 channel pairs x taps), software pipelined: the LDS fragments of step t+1
  // are requested before the MFMAs of step t so that no MFMA group waits on a fresh ds_read.
  auto mfma_chunk = [&]() {
    constexpr int STEPS = (CK / 2) * TAPS;
    constexpr int PF = 2;  // fragment prefetch distance in K-steps (LDS latency under 8-16 waves/CU > 1 step)
    float av[PF + 1][kMT], bv[PF + 1][kNT];
    auto load_frag = [&](int st, int buf) {
      const int kp = st / TAPS, tap = st % TAPS;
      const int ky = tap / KS, kx = tap % KS;
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) av[buf][mt] = w_tile[aBase + (kp * TAPS + tap) * 2 * kCoutTile + mt * 32];
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt) bv[buf][nt] = in_tile[bBase[nt] + kp * 2 * PLANE + ky * PITCH + kx];
    };
#pragma unroll
    for (int st = 0; st < PF && st < STEPS; ++st) load_frag(st, st % (PF + 1));
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if (st + PF < STEPS) load_frag(st + PF, (st + PF) % (PF + 1));
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this step's MFMAs (hipcc sinks it otherwise)
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st % (PF + 1)][mt], bv[st % (PF + 1)][nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  issue_loads(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();  // previous round's LDS reads are done
    commit_loads();
    __syncthreads();
    if (c + 1 < nchunks) issue_loads(c + 1);  // in flight under this chunk's MFMAs
    mfma_chunk();
    if (XTRA && xtra_wg && tid < NPIX) {  // rows of a 1x1 layer's native weight image are the input channels in order
      const float* wx = a.w + (size_t)c * W_ROWS * a.CoutPad + (a.CoutPad - kCoutTile);
#pragma unroll
      for (int r = 0; r < CK; ++r) xacc = fmaf(wx[(size_t)r * a.CoutPad], in_tile[r * PLANE + tid], xacc);
    }
  }

  // ---- epilogue: bias -> ReLU -> BN affine -> (pool) -> NCHW store -------------------------
  const int Ho = POOL ? a.H / 2 : a.H, Wo = POOL ? a.W / 2 : a.W;
  const unsigned plane_o = (unsigned)(Ho * Wo);
  float* out_b = a.out + (size_t)b * a.Cout * plane_o;
  if (XTRA && xtra_wg && tid < NPIX && p0 + tid < HW) {
    const int co = a.CoutPad - kCoutTile;  // == Cout - 1
    float v = xacc + (a.bias ? a.bias[co] : 0.0f);
    if (a.relu) v = v > 0.0f ? v : 0.0f;
    if (a.scale) v = fmaf(v, a.scale[co], a.shift[co]);
    out_b[(size_t)co * HW + p0 + tid] = v;
  }
  // Pooling happens in registers: a 2x2 window is {lane j, lane j^1} horizontally and, vertically,
  // either the wave's neighbouring N-tile (TW == 32: tiles nt, nt+1 are rows 2k, 2k+1) or lane
  // j^TW inside one N-tile (TW in {8,16}: an N-tile holds 32/TW whole rows).
  static_assert(!POOL || TW == 32 || TW == 16 || TW == 8, "in-register pooling needs TW in {8,16,32}");
  static_assert(!POOL || TW != 32 || (NT % 2 == 0), "TW == 32 pools across N-tile pairs");
  // Byte offset of each output this lane stores, inside the descriptor of one accumulator row: the row's channel for lanes
  // 0-31 starts at byte 0, the channel of lanes 32-63 lies four planes up; all ones = nothing to store (dropped by the range check)
  constexpr int NO = (POOL && TW == 32) ? kNT / 2 : kNT;
  unsigned ovoff[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    int pix;
    if (POOL) {
      pix = -1;
      const int q = qidx[(TW == 32) ? 2 * o : o];
      if (q >= 0) {
        const int ty = q / TW, tx = q % TW;
        const int yo = (y0 + ty) >> 1, xo = (x0 + tx) >> 1;
        if (!(ty & 1) && !(tx & 1) && yo < Ho && xo < Wo) pix = yo * Wo + xo;  // the window's top-left lane stores
      }
    } else {
      pix = opix[o];
    }
    ovoff[o] = pix >= 0 ? ((unsigned)pix + (unsigned)(4 * half) * plane_o) * 4u : 0xFFFFFFFFu;
  }
  auto epilogue = [&](auto relu_c, auto aff_c) {
    constexpr bool RELU = decltype(relu_c)::value, AFF = decltype(aff_c)::value;
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // accumulator rows 4g .. 4g+3 are channels cl0 + 4 half + (0..3) of the workgroup's 64
        const int cl0 = (wm * MT + mt) * 32 + 8 * g;
        const f32x4 bi = *reinterpret_cast<const f32x4*>(&s_bias[cl0 + 4 * half]);
        f32x4 sc, sh;
        if (AFF) {
          sc = *reinterpret_cast<const f32x4*>(&s_scale[cl0 + 4 * half]);
          sh = *reinterpret_cast<const f32x4*>(&s_shift[cl0 + 4 * half]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * g + i;
          const int co_r = co0 + cl0 + i;  // channel of lanes 0-31; lanes 32-63 hold channel co_r + 4
          const int nch = a.Cout - co_r > 0 ? a.Cout - co_r : 0;  // channels co_r .. Cout - 1 (none: every store of the row is dropped)
          const __amdgpu_buffer_rsrc_t ro = conv_rsrc(out_b + (size_t)co_r * plane_o, (unsigned)nch * plane_o * 4u);
          float pv[kNT];
#pragma unroll
          for (int nt = 0; nt < kNT; ++nt) {
            float v = acc[mt][nt][r] + bi[i];
            if (RELU) v = fmaxf(v, 0.0f);  // v_max_f32: max(-0, +0) = +0 and max(NaN, 0) = 0, as `v > 0 ? v : 0`
            if (AFF) v = fmaf(v, sc[i], sh[i]);
            pv[nt] = v;
          }
          if (!POOL) {
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv[nt]), ro, ovoff[nt], 0, 0);
          } else if (TW == 32) {
#pragma unroll
            for (int o = 0; o < NO; ++o) {
              float m = fmaxf(pv[2 * o], pv[2 * o + 1]);  // rows 2k, 2k+1
              m = max_lane_xor<1>(m);                     // columns 2c, 2c+1
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), ro, ovoff[o], 0, 0);
            }
          } else {
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
              float m = max_lane_xor<TW>(pv[nt]);  // rows 2k, 2k+1
              m = max_lane_xor<1>(m);              // columns 2c, 2c+1
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), ro, ovoff[nt], 0, 0);
            }
          }
        }
      }
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  if (a.relu) {
    if (a.scale) epilogue(T_{}, T_{});
    else epilogue(T_{}, F_{});
  } else {
    if (a.scale) epilogue(F_{}, T_{});
    else epilogue(F_{}, F_{});
  }
}

// ------------------------------------------------------------------------------------------
// Round 6: the first two layers of a 1-channel network (SuperPointv1: conv1a 1 -> 64, conv1b 64 -> 64 + pool) as ONE launch, so that
// the 761 MB conv1a output (B = 32) is never written to or read from HBM.  The second layer is conv_block_kernel<3,8,32,2,4,1,2,8>
// (same tile, wave layout, K order, epilogue); what changes is where its input planes come from: instead of staging 8 channels
// of the first layer's output per chunk from global memory, the workgroup recomputes 16 of them (two chunks) at a time on the
// 10 x 34 halo of its tile, on the matrix cores:
//   v_mfma_f32_16x16x4_f32: M = 16 first-layer channels, N = 16 consecutive halo positions, K = the 9 taps padded to 12
//   (three instructions; taps in order = the first layer's k-ordered chain, the padding rows of A are zero: fma(0, x, acc) = acc);
//   A = first-layer weights (its native image: row 2 tap), B = the raw 12 x 36 input tile in LDS (loaded once, replicate fold and
//   zero padding applied), C + bias -> ReLU -> (affine) -> the LDS planes the second layer's B operand is read from; positions
//   outside the image stay zero (the second layer's zero padding).  22 N-tiles per 16 channels, dealt to the 8 waves.
// Every output is the same chain of operations as the two launches: bit-identical (tests/test_conv_gpu.py).
// Cost against the two launches: +2.9 % matrix time, ~+12 vector instructions per wave and chunk, 43 KB of LDS (three per CU).
// ------------------------------------------------------------------------------------------
struct Conv1abArgs {
  ConvArgs c;          // the SECOND layer (in = the raw input of the first one; Hs / Ws / h0 / w0: the first layer's replicate fold)
  const float* w0;     // first layer, native image [rows][cout0pad]: row 2 tap holds tap's weights of the 64 channels
  const float* bias0;  // [64] or null
  const float* scale0; // [64] or null (BatchNorm affine after the ReLU)
  const float* shift0;
  int relu0, cout0pad;
};

// CIN0: input channels of the first layer (1: SuperPointv1's gray image; 5: the event voxel grid of BASELINE.json's bench shape).
// K of the first layer = its 9 CIN0 real products in the oracle's order (channel pair, tap, parity; the zero slots of an odd
// channel count are no-ops of the chain and are left out), padded to a multiple of four.
template <int CIN0, bool POOL, int WPS>
__global__ __launch_bounds__(512, WPS) void conv1ab_kernel(const Conv1abArgs fa) {
  const ConvArgs& a = fa.c;
  constexpr int KS = 3, TH = 8, TW = 32, WN = 4, MT = 1, NT = 2, CK = 8;
  constexpr int kMT = MT, kNT = NT;
  constexpr int TAPS = 9, PW = TW + 2, PH = TH + 2;
  constexpr int PITCH = conv_lds_pitch(TH, TW);
  static_assert(PITCH == PW, "the halo's flat index is its LDS offset");
  constexpr int PLANE = PH * PITCH;     // 340
  constexpr int NTHR = 512;
  constexpr int C0 = 64;                // first-layer channels = second-layer input channels
  constexpr int PAIR = 16;              // first-layer channels recomputed per round (two chunks of the second layer)
  constexpr int IN_LDS = PAIR * PLANE;  // 5440 floats
  constexpr int W_ROWS = CK * TAPS, W_F4 = W_ROWS * kCoutTile / 4, W_PER_THR = (W_F4 + NTHR - 1) / NTHR;
  constexpr int RH = PH + 2, RP = PW + 2;  // raw tile: halo 2 (12 x 36)
  constexpr int NT16 = (PLANE + 15) / 16;  // 22 N-tiles of 16 halo positions
  constexpr int SLOTS = (NT16 + 7) / 8;    // N-tiles per wave (3)
  constexpr int RAWP = RH * RP;            // raw-tile floats per input channel (432)
  constexpr int NK = TAPS * CIN0;          // real products of a first-layer output
  constexpr int NI = (NK + 3) / 4;         // 16x16x4 instructions per N-tile (1 channel: 3; 5 channels: 12)
  static_assert(NI <= 4 || NI % 4 == 0, "operands are requested four instructions at a time");
  __shared__ __attribute__((aligned(16))) float in_tile[IN_LDS];
  __shared__ __attribute__((aligned(16))) float w_tile[W_ROWS * kCoutTile];
  __shared__ __attribute__((aligned(16))) float raw[CIN0 * RAWP];
  __shared__ __attribute__((aligned(16))) float w0s[NI * 4 * PAIR], c0s[3 * 64];  // first layer: weights [k][16 channels of the current pair], bias | scale | shift
  __shared__ int kofs[NI * 4];             // raw-tile offset of product k (its channel plane + tap shift); padding ks: 0
  __shared__ __attribute__((aligned(16))) float s_bias[kCoutTile], s_scale[kCoutTile], s_shift[kCoutTile];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int ncot = (int)gridDim.y;
  int item = xcd_contiguous((int)(blockIdx.x + blockIdx.y * gridDim.x), (int)gridDim.x * ncot);
  const int co0 = (item % ncot) * kCoutTile;
  int bid = item / ncot;
  const int tx_i = bid % a.tilesX;
  bid /= a.tilesX;
  const int ty_i = bid % a.tilesY;
  const int b = bid / a.tilesY;
  if (tid < kCoutTile) {
    const int co = co0 + tid;
    const bool cv = co < a.Cout;
    s_bias[tid] = (cv && a.bias) ? a.bias[co] : 0.0f;
    s_scale[tid] = (cv && a.scale) ? a.scale[co] : 1.0f;
    s_shift[tid] = (cv && a.scale) ? a.shift[co] : 0.0f;
  }
  const int y0 = ty_i * TH, x0 = tx_i * TW;
  // ---- raw input tile (CIN0 channels): logical (y, x) inside [0,H) x [0,W) = source clamped (replicate fold), outside = 0
  for (int i = tid; i < CIN0 * RAWP; i += NTHR) {
    const int ci = i / RAWP, r_ = i % RAWP;
    const int ry = r_ / RP, rx = r_ % RP;
    const int y = y0 - 2 + ry, x = x0 - 2 + rx;
    float v = 0.0f;
    if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
      int sy = y - a.h0, sx = x - a.w0;
      sy = sy < 0 ? 0 : (sy > a.Hs - 1 ? a.Hs - 1 : sy);
      sx = sx < 0 ? 0 : (sx > a.Ws - 1 ? a.Ws - 1 : sx);
      v = a.in[((size_t)b * CIN0 + ci) * a.Hs * a.Ws + (size_t)sy * a.Ws + sx];
    }
    raw[i] = v;
  }
  // product k of a first-layer output, in the oracle's order (channel pair, tap, parity) without the zero slots -> (channel, tap)
  auto k_to_ci_tap = [](int k, int* ci, int* tap) {
    constexpr int FULL = (CIN0 / 2) * 2 * TAPS;  // products of the complete channel pairs
    if (k < FULL) {
      *ci = 2 * (k / (2 * TAPS)) + (k & 1);
      *tap = (k % (2 * TAPS)) >> 1;
    } else {  // the last, single channel of an odd count
      *ci = CIN0 - 1;
      *tap = k - FULL;
    }
  };
  if (tid < NI * 4) {
    int ci, tap;
    k_to_ci_tap(tid < NK ? tid : 0, &ci, &tap);
    kofs[tid] = tid < NK ? ci * RAWP + (tap / 3) * RP + tap % 3 : 0;
  }
  // halo positions outside the image: zero in all 16 planes, once (never written again)
  if (tid < PLANE) {
    const int py = tid / PW, px = tid % PW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    if (!(y >= 0 && y < a.H && x >= 0 && x < a.W)) {
#pragma unroll
      for (int c = 0; c < PAIR; ++c) in_tile[c * PLANE + tid] = 0.0f;
    }
  }
  // ---- second layer: per-lane pixel bookkeeping (as conv_block_kernel)
  int bBase[kNT], qidx[kNT];
#pragma unroll
  for (int nt = 0; nt < kNT; ++nt) {
    const int q = (wn * kNT + nt) * 32 + j;
    qidx[nt] = q;  // NPIX == 256 slots: every slot is a pixel
    bBase[nt] = half * PLANE + (q / TW) * PITCH + q % TW;
  }
  const int aBase = half * kCoutTile + wm * MT * 32 + j;
  unsigned woff[W_PER_THR];
#pragma unroll
  for (int i = 0; i < W_PER_THR; ++i) {
    const int f = tid + i * NTHR;
    const int r = f / (kCoutTile / 4), c4 = f % (kCoutTile / 4);
    woff[i] = (unsigned)((f < W_F4 ? r : 0) * a.CoutPad + co0 + c4 * 4) * 4u;
  }
  // ---- first layer on the matrix cores: this wave's N-tiles (16 halo positions each).  Its weights (the current pair's 16
  // channels, NI x 4 K rows, rows past the last product zero) and epilogue constants sit in LDS: nothing of it is held in
  // registers across the second layer's chunks (the kernel has 80 registers for three workgroups per CU).
  auto load_w0s = [&](int cp) {  // K rows of first-layer channels 16 cp .. 16 cp + 15 (read by the NEXT first_layer_pair, two barriers later)
    for (int i = tid; i < NI * 4 * PAIR; i += NTHR) {
      const int k = i / PAIR, c = i % PAIR;
      int ci, tap;
      k_to_ci_tap(k < NK ? k : 0, &ci, &tap);
      const int row = (ci >> 1) * 2 * TAPS + tap * 2 + (ci & 1);  // the native weight image's K order
      w0s[i] = k < NK ? fa.w0[(size_t)row * fa.cout0pad + PAIR * cp + c] : 0.0f;
    }
  };
  load_w0s(0);
  if (tid < C0) {
    c0s[tid] = fa.bias0 ? fa.bias0[tid] : 0.0f;
    c0s[C0 + tid] = fa.scale0 ? fa.scale0[tid] : 1.0f;
    c0s[2 * C0 + tid] = fa.scale0 ? fa.shift0[tid] : 0.0f;
  }
  const int q16 = lane >> 4, n16 = lane & 15;
  int rawBase[SLOTS];                                 // raw-tile offset of the 3x3 patch of slot s's position (channel 0)
  const int wr0 = 4 * q16 * PLANE + 16 * wave + n16;  // LDS offset of slot 0's position in channel 4 q16 of the pair; slot s: + 128 s
  const int aBase0 = q16 * PAIR + n16;                // A: k = 4 g + q16 -> + 64 g
  unsigned live = 0;                                  // bit s: slot s holds a position inside the image (its outputs are written)
#pragma unroll
  for (int s_ = 0; s_ < SLOTS; ++s_) {
    const int t = wave + 8 * s_;
    const int p = 16 * t + n16;
    const bool pv = t < NT16 && p < PLANE;
    const int pc = pv ? p : 0;
    const int py = pc / PW, px = pc % PW;
    rawBase[s_] = py * RP + px;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    if (pv && y >= 0 && y < a.H && x >= 0 && x < a.W) live |= 1u << s_;
  }
  typedef float f32x4v_ __attribute__((ext_vector_type(4)));
  auto first_layer_pair = [&](int cp) {  // first-layer channels 16 cp .. 16 cp + 15 on the 340 halo positions -> in_tile
    const float* cst = c0s + PAIR * cp + 4 * q16;  // this lane's four channels: bias at +i, scale at +64 + i, shift at +128 + i
#pragma unroll
    for (int s_ = 0; s_ < SLOTS; ++s_) {
      if (wave + 8 * s_ >= NT16) break;  // wave-uniform
      f32x4v_ c4 = {0.0f, 0.0f, 0.0f, 0.0f};
      constexpr int GRP = NI < 4 ? NI : 4;  // instructions whose operands are requested together (register budget)
#pragma unroll
      for (int g0 = 0; g0 < NI; g0 += GRP) {
        float fA[GRP], fB[GRP];
#pragma unroll
        for (int g = 0; g < GRP; ++g) {
          fA[g] = w0s[aBase0 + 64 * (g0 + g)];
          fB[g] = raw[rawBase[s_] + kofs[4 * (g0 + g) + q16]];
        }
#pragma unroll
        for (int g = 0; g < GRP; ++g) c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fA[g], fB[g], c4, 0, 0, 0);
        if (NI > 4) __builtin_amdgcn_sched_barrier(0);
      }
      // (interleaving the wave's three N-tiles as independent chains measured the same on the image side and does not rescue the
      // 5-channel form: profiles/r06_notes.md 2)
      if ((live >> s_) & 1u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = c4[i] + cst[i];
          if (fa.relu0) v = fmaxf(v, 0.0f);
          if (fa.scale0) v = fmaf(v, cst[C0 + i], cst[2 * C0 + i]);
          in_tile[wr0 + 128 * s_ + i * PLANE] = v;
        }
      }
    }
  };

  f32x16 acc[kMT][kNT];
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
  f32x4 sW[W_PER_THR];
  auto issue_w = [&](int c) {
    const __amdgpu_buffer_rsrc_t rw = conv_rsrc(a.w + (size_t)c * W_ROWS * a.CoutPad, (unsigned)(W_ROWS * a.CoutPad) * 4u);
#pragma unroll
    for (int i = 0; i < W_PER_THR; ++i)
      if ((i + 1) * NTHR <= W_F4 || tid + i * NTHR < W_F4) sW[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff[i], 0, 0));
  };
  auto commit_w = [&]() {
#pragma unroll
    for (int i = 0; i < W_PER_THR; ++i)
      if ((i + 1) * NTHR <= W_F4 || tid + i * NTHR < W_F4) *reinterpret_cast<f32x4*>(w_tile + (tid + i * NTHR) * 4) = sW[i];
  };
  auto mfma_chunk = [&](int plane0) {  // the second layer's 36 K-steps on planes plane0 .. plane0 + 7
    constexpr int STEPS = (CK / 2) * TAPS;
    constexpr int PF = 2;
    float av[PF + 1][kMT], bv[PF + 1][kNT];
    const float* it = in_tile + plane0 * PLANE;
    auto load_frag = [&](int st, int buf) {
      const int kp = st / TAPS, tap = st % TAPS;
      const int ky = tap / KS, kx = tap % KS;
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) av[buf][mt] = w_tile[aBase + (kp * TAPS + tap) * 2 * kCoutTile + mt * 32];
#pragma unroll
      for (int nt = 0; nt < kNT; ++nt) bv[buf][nt] = it[bBase[nt] + kp * 2 * PLANE + ky * PITCH + kx];
    };
#pragma unroll
    for (int st = 0; st < PF && st < STEPS; ++st) load_frag(st, st % (PF + 1));
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if (st + PF < STEPS) load_frag(st + PF, (st + PF) % (PF + 1));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st % (PF + 1)][mt], bv[st % (PF + 1)][nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  issue_w(0);
  constexpr int NPAIRS = C0 / PAIR;  // 4
  for (int cp = 0; cp < NPAIRS; ++cp) {
    __syncthreads();  // raw tile / first-layer constants / zeroed cells visible (first round); the previous round's LDS reads are done
    first_layer_pair(cp);
    commit_w();
    __syncthreads();
    issue_w(2 * cp + 1);
    if (cp + 1 < NPAIRS) load_w0s(cp + 1);  // read after the next round's first barrier
    mfma_chunk(0);
    __syncthreads();
    commit_w();
    __syncthreads();
    if (2 * cp + 2 < 2 * NPAIRS) issue_w(2 * cp + 2);
    mfma_chunk(CK);
  }

  // ---- epilogue of the second layer: bias -> ReLU -> BN affine -> (pool) -> NCHW store (as conv_block_kernel)
  const int Ho = POOL ? a.H / 2 : a.H, Wo = POOL ? a.W / 2 : a.W;
  const unsigned plane_o = (unsigned)(Ho * Wo);
  float* out_b = a.out + (size_t)b * a.Cout * plane_o;
  constexpr int NO = POOL ? kNT / 2 : kNT;
  unsigned ovoff[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    int pix = -1;
    const int q = qidx[POOL ? 2 * o : o];
    const int ty = q / TW, tx = q % TW;
    if (POOL) {
      const int yo = (y0 + ty) >> 1, xo = (x0 + tx) >> 1;
      if (!(ty & 1) && !(tx & 1) && yo < Ho && xo < Wo) pix = yo * Wo + xo;
    } else {
      const int y = y0 + ty, x = x0 + tx;
      if (y < a.H && x < a.W) pix = y * a.W + x;
    }
    ovoff[o] = pix >= 0 ? ((unsigned)pix + (unsigned)(4 * half) * plane_o) * 4u : 0xFFFFFFFFu;
  }
  auto epilogue = [&](auto relu_c, auto aff_c) {
    constexpr bool RELU = decltype(relu_c)::value, AFF = decltype(aff_c)::value;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cl0 = wm * 32 + 8 * g;
      const f32x4 bi = *reinterpret_cast<const f32x4*>(&s_bias[cl0 + 4 * half]);
      f32x4 sc, sh;
      if (AFF) {
        sc = *reinterpret_cast<const f32x4*>(&s_scale[cl0 + 4 * half]);
        sh = *reinterpret_cast<const f32x4*>(&s_shift[cl0 + 4 * half]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        const int co_r = co0 + cl0 + i;
        const int nch = a.Cout - co_r > 0 ? a.Cout - co_r : 0;
        const __amdgpu_buffer_rsrc_t ro = conv_rsrc(out_b + (size_t)co_r * plane_o, (unsigned)nch * plane_o * 4u);
        float pv[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
          float v = acc[0][nt][r] + bi[i];
          if (RELU) v = fmaxf(v, 0.0f);
          if (AFF) v = fmaf(v, sc[i], sh[i]);
          pv[nt] = v;
        }
        if (!POOL) {
#pragma unroll
          for (int nt = 0; nt < kNT; ++nt) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv[nt]), ro, ovoff[nt], 0, 0);
        } else {
#pragma unroll
          for (int o = 0; o < NO; ++o) {
            float m = fmaxf(pv[2 * o], pv[2 * o + 1]);
            m = max_lane_xor<1>(m);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), ro, ovoff[o], 0, 0);
          }
        }
      }
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  if (a.relu) {
    if (a.scale) epilogue(T_{}, T_{});
    else epilogue(T_{}, F_{});
  } else {
    if (a.scale) epilogue(F_{}, T_{});
    else epilogue(F_{}, F_{});
  }
}

// ------------------------------------------------------------------------------------------
// Small grids (single pairs: the reference's own call pattern): conv3x3 on v_mfma_f32_16x16x4_f32.
// A launch that cannot fill the chip is bound by the LATENCY of one workgroup = chunks x K-steps x MFMA latency; the
// k-ordered chain forbids splitting K.  The 16x16x4 instruction consumes four K per 32-cycle issue (40 dependent) instead
// of two per 64, and a wave that owns ONE 16x16 accumulator (16 output channels x 16 pixels) walks a chunk of 72 K in
// 18 x 40 = 720 cycles instead of 36 x 64 = 2304; the work spreads over 4x as many waves.  Its four products per
// instruction accumulate as the sequential chain k, k+1, k+2, k+3 (tools/mfma16_order.hip: bit-equal to the fmaf chain, no
// other order matches), and the K order is the one of conv_block_kernel, so results are bit-identical to it and to the oracle.
//   workgroup = 64 output channels x one 2 x 8 pixel tile, 4 waves = 4 M-tiles of 16 channels (one wave per SIMD)
//   lane l: A = w[k = 4g + l/16][channel l%16], B = patch[k = 4g + l/16][pixel l%16], C rows 4 (l/16) + i, column l%16
//   k -> (channel pair k / 18, tap (k % 18) / 2, parity k % 2); 4g is even, so the parity is the lane group's
// Cin must be a multiple of 8, no replicate-pad fold (never the first layer).  POOL: 2x2 max over lanes j^1 (x), j^8 (y).
// ------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
// CK input channels per LDS round.  Measured at B=1 on the 33x44 128->128 layer (tools/experiments/r3_exp12.sh): 27.8 us with the finest
// 32x32x2 tile (11x5, one accumulator per wave), 14.5 us here with CK = 8 (2,170 cycles per chunk for 720 of MFMAs: every
// workgroup streams the layer's whole weight set, 18 KB per chunk, through L2 and LDS for 16 pixels); CK = 16 (half the
// barriers) 14.6-15.1 us, three chunks of registers in flight 14.5-15.1 us, no LDS at all (every lane loads its own
// operands from L1 / L2) 21 us.  Forward at B=1: 0.80 -> 0.70 ms on the device (0.86 -> 0.77 ms wall).
// NPW: 16-pixel N-tiles per wave (accumulators sharing one A fragment); the workgroup's tile is 2 x 8 NPW pixels.  More
// pixels per workgroup = less weight traffic per pixel (the kernel is L2-bound on the larger maps) at NPW x the chain length.
template <bool POOL, int CK, int NPW>
__global__ __launch_bounds__(256) void conv16_kernel(const ConvArgs a) {
  constexpr int TAPS = 9, KCH = CK * TAPS, G = KCH / 4;  // CK = 8: 72 K per chunk = 18 instructions
  constexpr int TH = 2, TW = 8 * NPW, PH = TH + 2, PW = TW + 2;
  constexpr int PITCH = TW + 8, PLANE = PH * PITCH + 8;
  constexpr int IN_LDS = CK * PLANE;
  constexpr int W_F4 = KCH * kCoutTile / 4;       // float4 of weights per chunk (CK = 8: 1152)
  constexpr int W_PER_THR = (W_F4 + 255) / 256;
  constexpr int IN_ELEMS = CK * PH * PW;
  constexpr int IN_PER_THR = (IN_ELEMS + 255) / 256;
  __shared__ __attribute__((aligned(16))) float in_tile[IN_LDS];
  __shared__ __attribute__((aligned(16))) float w_tile[KCH * kCoutTile];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, j = lane & 15;
  const int ncot = (int)gridDim.y;
  int item = xcd_contiguous((int)(blockIdx.x + blockIdx.y * gridDim.x), (int)gridDim.x * ncot);
  const int co0 = (item % ncot) * kCoutTile;
  int bid = item / ncot;
  const int tx_i = bid % a.tilesX;
  bid /= a.tilesX;
  const int ty_i = bid % a.tilesY;
  const int b = bid / a.tilesY;
  const int y0 = ty_i * TH, x0 = tx_i * TW;
  const int HW = a.H * a.W;
  // ---- staging plan
  const float* in_b = a.in + (size_t)b * a.Cin * HW;
  unsigned goff[IN_PER_THR], loff[IN_PER_THR], okmask = 0;
#pragma unroll
  for (int i = 0; i < IN_PER_THR; ++i) {
    const int e = tid + i * 256;
    int off = -1, cil = 0, r = 0;
    if (e < IN_ELEMS) {
      cil = e / (PH * PW);
      r = e % (PH * PW);
      const int y = y0 - 1 + r / PW, x = x0 - 1 + r % PW;
      if (y >= 0 && y < a.H && x >= 0 && x < a.W) off = y * a.W + x;
    }
    goff[i] = (unsigned)cil * (unsigned)HW + (unsigned)(off >= 0 ? off : 0);
    loff[i] = (unsigned)(cil * PLANE + (r / PW) * PITCH + r % PW);
    okmask |= (off >= 0 ? 1u : 0u) << i;
  }
  unsigned woff[W_PER_THR];
#pragma unroll
  for (int i = 0; i < W_PER_THR; ++i) {
    const int f = tid + i * 256;
    const int r = f / (kCoutTile / 4), c4 = f % (kCoutTile / 4);
    woff[i] = (unsigned)((f < W_F4 ? r : 0) * a.CoutPad + co0 + c4 * 4);
  }
  // ---- per-lane operand offsets: koff[g] = LDS offset of patch element k = 4g + q relative to the pixel's top-left tap
  int koff[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k = 4 * g + q;
    const int pair = k / 18, tap = (k % 18) >> 1;
    koff[g] = (2 * pair + (k & 1)) * PLANE + (tap / 3) * PITCH + tap % 3;
  }
  const int bpix = (j >> 3) * PITCH + (j & 7);     // pixel (j / 8, j % 8) of the tile, halo origin at (0, 0)
  const int aoff = q * kCoutTile + wave * 16 + j;  // A: k = 4g + q rows of 64 channels; this wave's M-tile
  float r_in[IN_PER_THR];
  f32x4 r_w[W_PER_THR];
  auto issue_loads = [&](int c) {
    const float* ib = in_b + (size_t)c * CK * HW;
#pragma unroll
    for (int i = 0; i < IN_PER_THR; ++i) r_in[i] = ib[goff[i]];
    const float* wb = a.w + (size_t)c * KCH * a.CoutPad;
#pragma unroll
    for (int i = 0; i < W_PER_THR; ++i) r_w[i] = *reinterpret_cast<const f32x4*>(wb + woff[i]);
  };
  f32x4v acc[NPW];
#pragma unroll
  for (int n = 0; n < NPW; ++n) acc[n] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
  const int nchunks = a.Cin / CK;
  issue_loads(0);
  // the epilogue's per-channel constants are requested here, behind the first chunk's loads: fetched in the epilogue they are
  // twelve guarded loads that hipcc serialises (a memory round trip each) at the end of a latency-bound workgroup
  float e_bi[4], e_sc[4], e_sh[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int coc = min(co0 + wave * 16 + 4 * q + i, a.Cout - 1);
    e_bi[i] = a.bias ? a.bias[coc] : 0.0f;
    e_sc[i] = a.scale ? a.scale[coc] : 1.0f;
    e_sh[i] = a.scale ? a.shift[coc] : 0.0f;
  }
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IN_PER_THR; ++i)
      if (tid + i * 256 < IN_ELEMS) in_tile[loff[i]] = ((okmask >> i) & 1u) ? r_in[i] : 0.0f;
#pragma unroll
    for (int i = 0; i < W_PER_THR; ++i)
      if (tid + i * 256 < W_F4) *reinterpret_cast<f32x4*>(w_tile + (tid + i * 256) * 4) = r_w[i];
    __syncthreads();
    if (c + 1 < nchunks) issue_loads(c + 1);
    constexpr int PF = 3;  // (6 / 9 / 17 measured the same at single pairs, round 5)
    float av[PF + 1], bv[PF + 1][NPW];
#pragma unroll
    for (int g = 0; g < PF; ++g) {
      av[g] = w_tile[aoff + g * 4 * kCoutTile];
#pragma unroll
      for (int n = 0; n < NPW; ++n) bv[g][n] = in_tile[bpix + 8 * n + koff[g]];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g + PF < G) {
        av[(g + PF) % (PF + 1)] = w_tile[aoff + (g + PF) * 4 * kCoutTile];
#pragma unroll
        for (int n = 0; n < NPW; ++n) bv[(g + PF) % (PF + 1)][n] = in_tile[bpix + 8 * n + koff[g + PF]];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NPW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g % (PF + 1)], bv[g % (PF + 1)][n], acc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue: bias -> ReLU -> BN affine -> (pool) -> NCHW store; C row 4q + i = output channel, column j = pixel
  const int ty = j >> 3, tx = j & 7;
  const int Ho = POOL ? a.H / 2 : a.H, Wo = POOL ? a.W / 2 : a.W;
  float* out_b = a.out + (size_t)b * a.Cout * Ho * Wo;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + wave * 16 + 4 * q + i;
    const bool cv = co < a.Cout;
    const float bi = e_bi[i], sc = e_sc[i], sh = e_sh[i];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
      const int y = y0 + ty, x = x0 + 8 * n + tx;
      float v = acc[n][i] + bi;
      if (a.relu) v = v > 0.0f ? v : 0.0f;
      if (a.scale) v = fmaf(v, sc, sh);
      if (POOL) {
        float m = fmaxf(v, __shfl_xor(v, 8, 64));  // rows 2k, 2k+1
        m = fmaxf(m, __shfl_xor(m, 1, 64));        // columns 2c, 2c+1
        if (cv && ty == 0 && !(tx & 1) && (y >> 1) < Ho && (x >> 1) < Wo) out_b[(size_t)co * Ho * Wo + (size_t)(y >> 1) * Wo + (x >> 1)] = m;
      } else {
        if (cv && y < a.H && x < a.W) out_b[(size_t)co * HW + (size_t)y * a.W + x] = v;
      }
    }
  }
}

// 1x1 layers of small grids (the heads' second layers at single pairs) on the same instruction: workgroup = 64 output
// channels x 16 NPW consecutive pixels of the flattened map, 4 waves = 4 M-tiles of 16 channels; K = the input channels in
// their natural order (the 1x1 case of the K order above), 32 per LDS round = 8 instructions.  The 128-pixel tiles of
// conv_block_kernel<1,...> give a 33x44 map 12 workgroups per 64 channels, each walking 8 rounds of 32 dependent 64-cycle
// instructions per wave (23.6 us per launch at B=1); here 91 workgroups per 64 channels walk 8 x 8 40-cycle instructions.
// Cin must be a multiple of 32.  Weight rows sit 80 floats apart in LDS (k rows q = 0..3 of one instruction on banks
// 0 / 16 / 0 / 16 + channel: conflict-free per half-wave); pixel columns past the map read a clamped address and are never stored.
template <int NPW>
__global__ __launch_bounds__(256) void conv16_1x1_kernel(const ConvArgs a) {
  constexpr int CK = 32, G = CK / 4, NPX = 16 * NPW, WP = kCoutTile + 16;
  constexpr int IN_PER_THR = CK * NPX / 256, W_PER_THR = CK * (kCoutTile / 4) / 256;
  __shared__ __attribute__((aligned(16))) float in_tile[CK * NPX];
  __shared__ __attribute__((aligned(16))) float w_tile[CK * WP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, j = lane & 15;
  const int HW = a.H * a.W;
  const int tile = (int)blockIdx.x % a.tilesX, b = (int)blockIdx.x / a.tilesX;
  const int co0 = (int)blockIdx.y * kCoutTile;
  const int p0 = tile * NPX;
  const float* in_b = a.in + (size_t)b * a.Cin * HW;
  unsigned goff[IN_PER_THR], loff[IN_PER_THR];
#pragma unroll
  for (int i = 0; i < IN_PER_THR; ++i) {
    const int e = tid + i * 256;
    const int ch = e / NPX, px = e % NPX;
    goff[i] = (unsigned)ch * (unsigned)HW + (unsigned)min(p0 + px, HW - 1);
    loff[i] = (unsigned)e;
  }
  unsigned woff[W_PER_THR], wl[W_PER_THR];
#pragma unroll
  for (int i = 0; i < W_PER_THR; ++i) {
    const int f = tid + i * 256;
    const int r = f / (kCoutTile / 4), c4 = f % (kCoutTile / 4);
    woff[i] = (unsigned)(r * a.CoutPad + co0 + c4 * 4);
    wl[i] = (unsigned)(r * WP + c4 * 4);
  }
  float r_in[IN_PER_THR];
  f32x4 r_w[W_PER_THR];
  auto issue_loads = [&](int c) {
    const float* ib = in_b + (size_t)c * CK * HW;
#pragma unroll
    for (int i = 0; i < IN_PER_THR; ++i) r_in[i] = ib[goff[i]];
    const float* wb = a.w + (size_t)c * CK * a.CoutPad;
#pragma unroll
    for (int i = 0; i < W_PER_THR; ++i) r_w[i] = *reinterpret_cast<const f32x4*>(wb + woff[i]);
  };
  f32x4v acc[NPW];
#pragma unroll
  for (int n = 0; n < NPW; ++n) acc[n] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
  const int nchunks = a.Cin / CK;
  issue_loads(0);
  float e_bi[4], e_sc[4], e_sh[4];  // epilogue constants, requested behind the first round's loads (see conv16_kernel)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int coc = min(co0 + wave * 16 + 4 * q + i, a.Cout - 1);
    e_bi[i] = a.bias ? a.bias[coc] : 0.0f;
    e_sc[i] = a.scale ? a.scale[coc] : 1.0f;
    e_sh[i] = a.scale ? a.shift[coc] : 0.0f;
  }
  const int aoff = q * WP + wave * 16 + j, boff = q * NPX + j;
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IN_PER_THR; ++i) in_tile[loff[i]] = r_in[i];
#pragma unroll
    for (int i = 0; i < W_PER_THR; ++i) *reinterpret_cast<f32x4*>(w_tile + wl[i]) = r_w[i];
    __syncthreads();
    if (c + 1 < nchunks) issue_loads(c + 1);
    float av[G], bv[G][NPW];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      av[g] = w_tile[aoff + g * 4 * WP];
#pragma unroll
      for (int n = 0; n < NPW; ++n) bv[g][n] = in_tile[boff + g * 4 * NPX + 16 * n];
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int n = 0; n < NPW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g], bv[g][n], acc[n], 0, 0, 0);
  }
  // ---- epilogue: bias -> ReLU -> BN affine -> NCHW store; C row 4q + i = output channel, column j = pixel
  float* out_b = a.out + (size_t)b * a.Cout * HW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + wave * 16 + 4 * q + i;
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
      const int p = p0 + 16 * n + j;
      float v = acc[n][i] + e_bi[i];
      if (a.relu) v = v > 0.0f ? v : 0.0f;
      if (a.scale) v = fmaf(v, e_sc[i], e_sh[i]);
      if (co < a.Cout && p < HW) out_b[(size_t)co * HW + p] = v;
    }
  }
}

// OIHW -> native [K][CoutPad], K = (ci>>1)*2*taps + tap*2 + (ci&1); zero padded.
__global__ void conv_repack_kernel(const float* w, int cin, int cout, int taps, int coutPad, int krows, float* out) {
  const size_t n = (size_t)krows * coutPad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % coutPad);
    const int k = (int)(i / coutPad);
    const int h = k & 1;
    const int tap = (k >> 1) % taps;
    const int cp = (k >> 1) / taps;
    const int ci = 2 * cp + h;
    float v = 0.0f;
    if (ci < cin && co < cout) v = w[((size_t)co * cin + ci) * taps + tap];
    out[i] = v;
  }
}

// BatchNorm2d(eval) -> per-channel affine.  IEEE sqrt/div (hipcc's default correctly rounded
// forms), same op order as oracle.bn_fold: scale = g / sqrt(var + eps); shift = b - mean * scale.
__global__ void bn_fold_kernel(const float* g, const float* b, const float* mean, const float* var, float eps, int n, float* scale,
                               float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = g[i] / sqrtf(var[i] + eps);
  scale[i] = s;
  shift[i] = b[i] - mean[i] * s;
}

// rows of the native weight image: K = (ci>>1, tap, ci&1), zero-padded to whole groups of 32 input
// channels (the largest chunk any kernel variant stages) so that chunk loads never need bounds
int native_krows(int cin, int taps) { return einx_cdiv(cin, 32) * 32 * taps; }

thread_local const char* g_last_conv_kernel = "";

struct TileCfg {
  int th, tw, slots;  // slots = pixel slots a workgroup launches for this tile
};

template <int KS, int TH, int TW, int WM, int WN, int MT, int NT, int CK, bool POOL, bool DENSE3 = false>
void launch(const ConvArgs& a, int B, hipStream_t s) {
  dim3 grid((unsigned)(a.tilesX * a.tilesY * B), (unsigned)(a.CoutPad / kCoutTile));
  // DENSE3: launches of at least eight rounds of three workgroups per CU take the instantiation that fits three per CU
  const bool three = DENSE3 && (long)grid.x * grid.y >= 8L * 768;
  {
    // name of the instantiation this call launches (einx_conv_last_kernel: measurement provenance)
    static char nm[2][96] = {{0}, {0}};
    char* n = nm[three ? 1 : 0];
    if (!n[0]) snprintf(n, sizeof nm[0], "conv_block_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%s>%s", KS, TH, TW, WM, WN, MT, NT, CK, POOL ? "true" : "false", three ? " (3 per CU)" : "");
    g_last_conv_kernel = n;
  }
  EINX_PROF(KS == 1 ? "conv_block_kernel 1x1" : (CK < 8 ? "conv_block_kernel 3x3 first layer" : "conv_block_kernel 3x3"), s);
  if constexpr (DENSE3) {
    if (three) {
      hipLaunchKernelGGL((conv_block_kernel<KS, TH, TW, WM, WN, MT, NT, CK, POOL, false, 6>), grid, dim3(WM * WN * 64), 0, s, a);
      return;
    }
  }
  hipLaunchKernelGGL((conv_block_kernel<KS, TH, TW, WM, WN, MT, NT, CK, POOL>), grid, dim3(WM * WN * 64), 0, s, a);
}

// waste = slots launched / pixels useful, for picking a tile shape per layer
double tile_waste(int H, int W, const TileCfg& c) {
  const double tiles = (double)einx_cdiv(H, c.th) * einx_cdiv(W, c.tw);
  return tiles * c.slots / ((double)H * W);
}

}  // namespace

EINX_EXPORT const char* einx_conv_last_kernel(void) { return g_last_conv_kernel; }

EINX_EXPORT size_t einx_conv_weight_elems(int cin, int cout, int ks) {
  const int taps = ks * ks;
  const int coutPad = einx_cdiv(cout, kCoutTile) * kCoutTile;
  return (size_t)native_krows(cin, taps) * coutPad;
}

EINX_EXPORT int einx_conv_repack(const float* w_oihw, int cin, int cout, int ks, float* w_native, void* stream) {
  EINX_CHECK_ARG(w_oihw && w_native, "null pointer");
  EINX_CHECK_ARG(ks == 1 || ks == 3, "kernel size must be 1 or 3");
  EINX_CHECK_ARG(cin > 0 && cout > 0, "bad channel count");
  const int taps = ks * ks;
  const int coutPad = einx_cdiv(cout, kCoutTile) * kCoutTile;
  const int krows = native_krows(cin, taps);
  const size_t n = (size_t)krows * coutPad;
  const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(conv_repack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, cin, cout, taps, coutPad, krows,
                     w_native);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int n, float* scale,
                             float* shift, void* stream) {
  EINX_CHECK_ARG(gamma && beta && mean && var && scale && shift, "null pointer");
  EINX_CHECK_ARG(n > 0, "bad channel count");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(einx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, n, scale,
                     shift);
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

// first two layers of a 1-channel network as one launch (conv1ab_kernel); -1 from the dispatcher check = not applicable, the
// caller runs the two layers one after the other
EINX_EXPORT int einx_conv_first_two_fused_ok(const einx_conv_desc* d0, const einx_conv_desc* d1, int B, int H, int W) {
  if (!d0 || !d1) return 0;
  // 1: the dispatcher's choice (1-channel first layers: +1 % on the step).  2: covered by the kernel, but slower than the two launches
  // (5 input channels: 12 instead of 3 first-layer instructions per N-tile, 8.34 -> 8.62 ms per step; kept callable and tested).
  const bool shapes = (d0->cin == 1 || d0->cin == 5) && d0->cout == 64 && d0->ks == 3 && !d0->pool && d1->cin == 64 && d1->cout == 64 && d1->ks == 3;
  if (!shapes || (d1->pool && ((H & 1) || (W & 1)))) return 0;
  // the large-grid regime of conv_block_kernel's three-per-CU instantiation (what the fused kernel replaces); smaller launches keep
  // the two latency-tuned launches
  const long grid = (long)einx_cdiv(W, 32) * einx_cdiv(H, 8) * B;
  if (grid < 8L * 768) return 0;
  return d0->cin == 1 ? 1 : 2;
}

EINX_EXPORT int einx_conv_first_two_fused(const float* in, int B, int Hs, int Ws, int h0, int w0, int H, int W, const einx_conv_desc* d0,
                                          const einx_conv_desc* d1, float* out, void* stream) {
  EINX_CHECK_ARG(in && out && d0 && d1 && d0->w_native && d1->w_native, "null pointer");
  EINX_CHECK_ARG(einx_conv_first_two_fused_ok(d0, d1, B, H, W), "layers / launch size outside what the fused first-two-layers kernel covers");
  EINX_CHECK_ARG((d0->scale == nullptr) == (d0->shift == nullptr) && (d1->scale == nullptr) == (d1->shift == nullptr), "scale and shift go together");
  EINX_CHECK_ARG((size_t)d0->cin * Hs * Ws < (1u << 30) && (size_t)64 * H * W < (1u << 30), "image too large");
  Conv1abArgs fa;
  ConvArgs& a = fa.c;
  a.in = in;
  a.w = d1->w_native;
  a.bias = d1->bias;
  a.scale = d1->scale;
  a.shift = d1->shift;
  a.out = out;
  a.B = B;
  a.Cin = 64;
  a.Cout = 64;
  a.CoutPad = 64;
  a.Hs = Hs;
  a.Ws = Ws;
  a.h0 = h0;
  a.w0 = w0;
  a.H = H;
  a.W = W;
  a.relu = d1->relu;
  a.tilesX = einx_cdiv(W, 32);
  a.tilesY = einx_cdiv(H, 8);
  fa.w0 = d0->w_native;
  fa.bias0 = d0->bias;
  fa.scale0 = d0->scale;
  fa.shift0 = d0->shift;
  fa.relu0 = d0->relu;
  fa.cout0pad = 64;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(a.tilesX * a.tilesY * B), 1);
  static thread_local char nm[96];
  snprintf(nm, sizeof nm, "conv1ab_kernel<%d,%s> (first two layers fused, 3 per CU)", d0->cin, d1->pool ? "true" : "false");
  g_last_conv_kernel = nm;
  EINX_PROF("conv1ab_kernel (first two layers)", s);
  if (d0->cin == 1) {
    if (d1->pool) hipLaunchKernelGGL((conv1ab_kernel<1, true, 6>), grid, dim3(512), 0, s, fa);
    else hipLaunchKernelGGL((conv1ab_kernel<1, false, 6>), grid, dim3(512), 0, s, fa);
  } else {
    if (d1->pool) hipLaunchKernelGGL((conv1ab_kernel<5, true, 6>), grid, dim3(512), 0, s, fa);
    else hipLaunchKernelGGL((conv1ab_kernel<5, false, 6>), grid, dim3(512), 0, s, fa);
  }
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}

EINX_EXPORT int einx_conv_block(const float* in, int B, int Hs, int Ws, int h0, int w0, int H, int W, const einx_conv_desc* d,
                                float* out, void* stream) {
  EINX_CHECK_ARG(in && out && d && d->w_native, "null pointer");
  EINX_CHECK_ARG(d->ks == 1 || d->ks == 3, "kernel size must be 1 or 3");
  EINX_CHECK_ARG(B > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && d->cin > 0 && d->cout > 0, "bad shape");
  EINX_CHECK_ARG((d->scale == nullptr) == (d->shift == nullptr), "scale and shift go together");
  EINX_CHECK_ARG(!d->pool || (H % 2 == 0 && W % 2 == 0), "pooling needs even H and W");
  EINX_CHECK_ARG(d->ks == 3 || (Hs == H && Ws == W && h0 == 0 && w0 == 0), "1x1 layers take no padding fold");
  EINX_CHECK_ARG(d->ks == 3 || !d->pool, "pooled 1x1 not supported");
  // byte offsets inside one image's [C,H,W] tensor are 32-bit (buffer descriptors per image)
  EINX_CHECK_ARG((size_t)d->cin * Hs * Ws < (1u << 30) && (size_t)d->cout * H * W < (1u << 30) && (size_t)d->cin * H * W < (1u << 30),
                 "image too large (2^30 elements per image and tensor)");
  hipStream_t s = (hipStream_t)stream;
  ConvArgs a;
  a.in = in;
  a.w = d->w_native;
  a.bias = d->bias;
  a.scale = d->scale;
  a.shift = d->shift;
  a.out = out;
  a.B = B;
  a.Cin = d->cin;
  a.Cout = d->cout;
  a.CoutPad = einx_cdiv(d->cout, kCoutTile) * kCoutTile;
  a.Hs = Hs;
  a.Ws = Ws;
  a.h0 = h0;
  a.w0 = w0;
  a.H = H;
  a.W = W;
  a.relu = d->relu;
  if (d->ks == 1) {
    // small maps (the 33x44 heads): 128-pixel runs double the workgroup count so that the 256 CUs hold
    // enough waves to hide the staging latency (768 -> 1536 workgroups for 256 output channels at B=32)
    const long blocks256 = (long)einx_cdiv(H * W, 256) * B * (a.CoutPad / kCoutTile);
    a.tilesY = 1;
    {  // small grids: one 16x16 accumulator per wave (conv16_1x1_kernel), the widest pixel run that still gives 512 workgroups
      const long max_wg128 = 512;
      const long blocks128 = (long)einx_cdiv(H * W, 128) * B * (a.CoutPad / kCoutTile);
      if (blocks128 < max_wg128 && d->cin % 32 == 0) {
        int npw = 1;
        for (int cand = 4; cand > 1; cand >>= 1)
          if ((long)einx_cdiv(H * W, 16 * cand) * B * (a.CoutPad / kCoutTile) >= 512) {
            npw = cand;
            break;
          }
        a.tilesX = einx_cdiv(H * W, 16 * npw);
        dim3 grid((unsigned)(a.tilesX * B), (unsigned)(a.CoutPad / kCoutTile));
        static thread_local char nm[64];
        snprintf(nm, sizeof nm, "conv16_1x1_kernel<%d>", npw);
        g_last_conv_kernel = nm;
        EINX_PROF("conv16_1x1_kernel (small grid)", s);
        if (npw == 4) hipLaunchKernelGGL(conv16_1x1_kernel<4>, grid, dim3(256), 0, s, a);
        else if (npw == 2) hipLaunchKernelGGL(conv16_1x1_kernel<2>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv16_1x1_kernel<1>, grid, dim3(256), 0, s, a);
        EINX_CHECK_LAUNCH();
        return EINX_OK;
      }
    }
    const long min256 = 1024;
    if (blocks256 < min256) {
      a.tilesX = einx_cdiv(H * W, 128);
      if (d->cout == a.CoutPad - kCoutTile + 1 && d->cout > kCoutTile && d->cin % 32 == 0) {
        // 64 n + 1 output channels (the detector's 65): n channel tiles, the last one also carries channel 64 n (see XTRA)
        dim3 grid((unsigned)(a.tilesX * B), (unsigned)(a.CoutPad / kCoutTile - 1));
        g_last_conv_kernel = "conv_block_kernel<1,1,128,1,4,2,1,32,false,xtra>";
        EINX_PROF("conv_block_kernel 1x1", s);
        hipLaunchKernelGGL((conv_block_kernel<1, 1, 128, 1, 4, 2, 1, 32, false, true>), grid, dim3(256), 0, s, a);
      } else {
        launch<1, 1, 128, 1, 4, 2, 1, 32, false>(a, B, s);
      }
    } else {
      a.tilesX = einx_cdiv(H * W, 256);
      launch<1, 1, 256, 1, 4, 2, 2, 32, false>(a, B, s);
    }
    EINX_CHECK_LAUNCH();
    return EINX_OK;
  }
  // candidate tile shapes: (8,32) and (11,22) with 256 pixel slots (8 waves), (12,16) and (22,8) with 192
  // (4 waves), (11,11) with 128 for small maps.  Pooled layers need even tile dims so that every 2x2 window
  // lives inside one tile.  First choice: the least pixel-slot waste; then the two corrections below.
  static const TileCfg cfgs[5] = {{8, 32, 256}, {12, 16, 192}, {22, 8, 192}, {11, 22, 256}, {11, 11, 128}};
  int best = 0;
  double bw = 1e30;
  for (int i = 0; i < 4; ++i) {
    if (d->pool && ((cfgs[i].th & 1) || (cfgs[i].tw & 1))) continue;
    const double wst = tile_waste(H, W, cfgs[i]);
    if (wst < bw - 1e-9) {
      bw = wst;
      best = i;
    }
  }
  // small maps: when the launch would not even give every CU two workgroups, halve the tile
  // (11x11, 128 pixel slots) so the 256 CUs are loaded evenly (33x44 map, 128 channels: 384 -> 768)
  if (!d->pool) {
    const long blocks = (long)einx_cdiv(H, cfgs[best].th) * einx_cdiv(W, cfgs[best].tw) * B * (a.CoutPad / kCoutTile);
    if (blocks < 640 && tile_waste(H, W, cfgs[4]) <= bw * 1.05 + 1e-9) best = 4;
  }
  // (rounds 2-4 preferred the 8-wave 11x22 tile over the 4-wave 192-slot tiles at up to 6 % more pixel slots; with the
  // round-5 kernel the exact 12x16 tiling of the 132x176 maps is 10 % faster: 455 -> 412 us at B=32)
  {
    // Small grids (single images: the reference's own call pattern): the launch does not fill the chip, so what counts is the
    // LATENCY of one workgroup = (waves it puts on a SIMD) x (accumulator tiles per wave) x K-steps x 64 cycles -- the k-ordered
    // accumulation forbids splitting K -- times the rounds the grid needs on 256 CUs.  Finer tiles with one single-tile wave
    // per SIMD cut it up to 4x; bit-identical results (same kernel, other template arguments).  Thin first layers stay on the
    // generic path (they are store-bound).
    const long blocks_best = (long)einx_cdiv(H, cfgs[best].th) * einx_cdiv(W, cfgs[best].tw) * B * (a.CoutPad / kCoutTile);
    // the finest grain: one 16x16 accumulator per wave on the 16x16x4 instruction, when even that leaves SIMDs to spare
    // (MFMA tiles = pixels / 16 x channels / 16 <= 8192) -- see conv16_kernel
    {
      const long max_tiles = 8192;
      const long t16 = (long)einx_cdiv(H, 2) * einx_cdiv(W, 8) * B * (a.CoutPad / 16);
      if (blocks_best < 512 && d->cin % 8 == 0 && Hs == H && Ws == W && h0 == 0 && w0 == 0 && t16 <= max_tiles && (!d->pool || (H % 2 == 0 && W % 2 == 0))) {
        // pixels per workgroup: the widest tile that still leaves every CU two workgroups (weights are re-streamed per
        // workgroup; measured at B=1: 132x176 layers 28 -> 24.5 us with two N-tiles per wave, 29 with one or four)
        const long min_wg = 512;
        int npw = 1;
        for (int cand = 4; cand > 1; cand >>= 1)
          if ((long)einx_cdiv(H, 2) * einx_cdiv(W, 8 * cand) * B * (a.CoutPad / kCoutTile) >= min_wg) {
            npw = cand;
            break;
          }
        a.tilesX = einx_cdiv(W, 8 * npw);
        a.tilesY = einx_cdiv(H, 2);
        dim3 grid((unsigned)(a.tilesX * a.tilesY * B), (unsigned)(a.CoutPad / kCoutTile));
        static thread_local char nm[64];
        snprintf(nm, sizeof nm, "conv16_kernel<%s,8,%d>", d->pool ? "true" : "false", npw);
        g_last_conv_kernel = nm;
        EINX_PROF("conv16_kernel 3x3 (small grid)", s);
#define EINX_C16(P, N) hipLaunchKernelGGL((conv16_kernel<P, 8, N>), grid, dim3(256), 0, s, a)
        if (d->pool) {
          if (npw == 4) EINX_C16(true, 4);
          else if (npw == 2) EINX_C16(true, 2);
          else EINX_C16(true, 1);
        } else {
          if (npw == 4) EINX_C16(false, 4);
          else if (npw == 2) EINX_C16(false, 2);
          else EINX_C16(false, 1);
        }
#undef EINX_C16
        EINX_CHECK_LAUNCH();
        return EINX_OK;
      }
    }
    if (blocks_best < 512 && d->cin > 6) {
      struct Lat {
        int th, tw, unit;  // unit = waves per SIMD x accumulator tiles per wave
      };
      static const Lat pooled[4] = {{8, 32, 4}, {12, 16, 3}, {8, 16, 2}, {4, 16, 1}};
      static const Lat plain[5] = {{8, 32, 4}, {11, 22, 4}, {12, 16, 3}, {11, 11, 2}, {11, 5, 1}};
      const Lat* cand = d->pool ? pooled : plain;
      const int nc = d->pool ? 4 : 5;
      int pick = -1;
      double best_est = 1e30;
      for (int i = 0; i < nc; ++i) {
        const long blocks = (long)einx_cdiv(H, cand[i].th) * einx_cdiv(W, cand[i].tw) * B * (a.CoutPad / kCoutTile);
        const double est = (double)((blocks + 255) / 256) * cand[i].unit;
        if (est < best_est - 1e-9) {
          best_est = est;
          pick = i;
        }
      }
      a.tilesX = einx_cdiv(W, cand[pick].tw);
      a.tilesY = einx_cdiv(H, cand[pick].th);
      if (d->pool) {
        switch (pick) {
          case 0: launch<3, 8, 32, 2, 4, 1, 2, 8, true>(a, B, s); break;
          case 1: launch<3, 12, 16, 2, 2, 1, 3, 8, true>(a, B, s); break;
          case 2: launch<3, 8, 16, 2, 2, 1, 2, 8, true>(a, B, s); break;
          default: launch<3, 4, 16, 2, 2, 1, 1, 8, true>(a, B, s); break;
        }
      } else {
        switch (pick) {
          case 0: launch<3, 8, 32, 2, 4, 1, 2, 8, false>(a, B, s); break;
          case 1: launch<3, 11, 22, 2, 4, 1, 2, 8, false>(a, B, s); break;
          case 2: launch<3, 12, 16, 2, 2, 1, 3, 8, false>(a, B, s); break;
          case 3: launch<3, 11, 11, 2, 2, 1, 2, 8, false>(a, B, s); break;
          default: launch<3, 11, 5, 2, 2, 1, 1, 8, false>(a, B, s); break;
        }
      }
      EINX_CHECK_LAUNCH();
      return EINX_OK;
    }
  }
  a.tilesX = einx_cdiv(W, cfgs[best].tw);
  a.tilesY = einx_cdiv(H, cfgs[best].th);
  // Wave layouts (measured per layer, bench.py --layer-table): 8 waves (2 channel groups x 4 pixel
  // groups, one 32x64 accumulator block each) for the 256-slot tiles -- two waves per SIMD from one
  // workgroup hide each other's LDS/epilogue latency at half the accumulator registers per wave;
  // 4 waves of 1x3 tiles for the 192-slot tiles (6 waves load the 4 SIMDs unevenly: -25 %) and
  // 4 waves for the 1x1 heads.
  if (d->pool) {
    switch (best) {
      case 0: launch<3, 8, 32, 2, 4, 1, 2, 8, true, true>(a, B, s); break;
      case 1: launch<3, 12, 16, 2, 2, 1, 3, 8, true>(a, B, s); break;
      default: launch<3, 22, 8, 2, 2, 1, 3, 8, true>(a, B, s); break;
    }
  } else {
    switch (best) {
      case 0:
        // thin first layers (1 / 5 input channels): stage only the channel pairs that exist
        if (d->cin <= 2) launch<3, 8, 32, 2, 4, 1, 2, 2, false>(a, B, s);
        else if (d->cin <= 6) launch<3, 8, 32, 2, 4, 1, 2, 6, false>(a, B, s);
        else launch<3, 8, 32, 2, 4, 1, 2, 8, false, true>(a, B, s);
        break;
      case 1: launch<3, 12, 16, 2, 2, 1, 3, 8, false>(a, B, s); break;
      case 2: launch<3, 22, 8, 2, 2, 1, 3, 8, false>(a, B, s); break;
      case 3: launch<3, 11, 22, 2, 4, 1, 2, 8, false>(a, B, s); break;
      default: launch<3, 11, 11, 2, 2, 1, 2, 8, false>(a, B, s); break;
    }
  }
  EINX_CHECK_LAUNCH();
  return EINX_OK;
}
