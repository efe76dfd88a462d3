// conv_rows.hip — the 32 -> 32 channel 3x3 stride-1 convs of the first residual stage (resnet.py:29-55: layer1, 48 x 86 maps at
// 341 x 192) as a ROW-STREAMING kernel on the float16 matrix cores, gfx950 only.
//
// What round 5 measured about conv_x3_kernel / conv_x3p_kernel on these layers (270-540 MB of float32 activations per conv at 256
// pairs, 20 GFLOP): they are bound by neither HBM (3.1-3.4 TB/s of 6.3 achievable; a batch that fits the 256 MB Infinity Cache runs
// no faster) nor the matrix pipe (0.35 busy) but by INSTRUCTION ISSUE: ~2000 VALU / LDS / memory / scalar instructions per 16 x 16
// tile and wave — run-time tile geometry, a pixel table, 64-bit addresses, predicated stores, an 18 x 18 patch whose halo is a
// quarter of the loads, a cold memory round trip per tile.  This kernel removes the tile:
//
//   * a workgroup (8 waves, one per CU, two per SIMD) owns a BAND OF WHOLE ROWS of one sample (all 48 at 256 pairs) and walks it as a
//     1-D stream of "flat" positions f = R * P + C over the zero-padded image (P = W + 2): a 3x3 tap is a constant shift of the
//     stream, (kh - 1) * P + (kw - 1), so an M-tile is ANY 32 consecutive positions — no ragged tiles, no pixel table in the K loop,
//     every input pixel loaded exactly once (2 pad columns of 88 are the only waste);
//   * the stream is staged 128 positions at a time into an LDS ring of 512 positions (two float16 planes interleaved, 144 B per
//     position: conflict-free ds_read_b128) by one half of the waves while the other half multiplies (see "Half-steps" below): loads
//     are in flight two half-steps ahead of their conversion — one barrier per 128 positions, a constant number of bytes in flight;
//   * the layer's whole B operand (9 taps x 2 k-chunks x 2 pieces = 36 fragments, 144 registers) is resident in every wave;
//   * zero padding comes from buffer loads with an out-of-range offset, invalid outputs leave through out-of-range buffer stores: no
//     predicates, 32-bit offsets;
//   * GroupNorm partial sums stay in registers for the whole band: ONE slot per band, and with one band per sample the kernel
//     finalises the sample's GroupNorm itself (gn_finalize_lane: bit-identical to gn_finalize_kernel) — no finalisation launch.
//
// Arithmetic = conv_x3_kernel NP = 2 (two float16 pieces per operand, the terms a1 w0 + a0 w1 + a0 w0 per step, steps tap-major /
// k-chunk inner, the power-of-two weight scale undone on the accumulators) with the even and the odd steps in two accumulators that
// are added at the end (a chain of 54 MFMAs on one accumulator is paced by the MFMA latency): float32-grade equal to conv_x3_kernel
// (measured 3e-6 of the output range end to end), deterministic; the statistics are summed in another fixed order.
// Stager modes as in conv_x3.hip: 0 final activations, 1 relu(x * scale + shift), 2 the previous block's tail relu(x * scale + shift +
// skip) with the block output written for the band's own rows (the skip branch is loaded into registers at the top of a stage role and
// used at its end; one accumulator), 3 pooled stem keys (decode, |scale|, shift, ReLU; pooled activations written): all four convs of
// the first stage.
//
// Measured at 256 pairs (profiles/r5_rows_*): the GroupNorm-input convs 86 -> 72-80 us, pooled keys 119 -> 90, block tail 160 -> 138, and
// no GroupNorm finalisation launch behind any of them (-7 us each): the first stage 0.52 -> 0.41 ms with its finalisations.  Role
// cycles per half-step: compute 2.7 k (54 MFMAs = 1.7 k), stage 3.3 k (~330 instructions), barrier 0.5 k — the stage role is the pole:
// a transposed product (positions as columns: dwordx4 stores, one table entry per lane) would cut its epilogue by 55 instructions but
// needs 32 registers of per-lane partial sums that the 256-register budget (144 of them weights) does not have.
#include <cstdio>
#include <cstdlib>

#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Developer ablations (WRONG RESULTS, timing only; build with -DPNVO_ROWS_ABL=1, then env PNVO_ROWS_DBG = bits: 1 no loads, 2 no stores,
// 4 no MFMAs, 8 no conversion).  Compiled out of the product: a run-time test around every store costs more than the store.
#ifndef PNVO_ROWS_ABL
#define PNVO_ROWS_ABL 0
#endif
#if PNVO_ROWS_ABL
#define RS_DBG(p) ((p).rs_dbg)
#else
#define RS_DBG(p) 0
#endif

namespace {
constexpr int RS_THREADS = 512;
constexpr int RS_HG = 128;                        // stream positions per half-group (= 4 M-tiles: one per wave of a wave set)
constexpr int RS_RING = 512;                      // ring of 4 half-groups ...
constexpr int RS_TAIL = 48;                       // ... + a mirror of its first positions behind its end: a tile's window of 32 + 2 positions per
                                                  // kernel row never wraps, so a tap is an IMMEDIATE offset from the row's (wrapped) base
constexpr int RS_PITCH = 144;                     // bytes per position: 64 B hi pieces | 64 B lo pieces | 16 B pad (36 dwords: conflict-free)
constexpr int RS_RAW = (RS_RING + RS_TAIL) * RS_PITCH;    // [2 sets][2 slots][4 waves][4 loads][64 lanes][16 B]: raw float32 input by LDS-DMA
constexpr int RS_OTAB = RS_RAW + 4 * 16384;       // [2 sets][2][128] output byte offset of a position inside the sample's plane (bit 31: none)
constexpr int RS_MTAB = RS_OTAB + 4 * RS_HG * 4;  // [2 sets][2][128] oscale for a position that exists, 0 otherwise
constexpr int RS_SS = RS_MTAB + 4 * RS_HG * 4;    // [2][32] the sample's GroupNorm scale | shift (MODE 1)
constexpr int RS_RED = RS_SS + 64 * 4;            // [8 waves][32 channels][2]
constexpr int RS_DOF = RS_RED + 8 * 32 * 2 * 4;   // [512 threads][4] MODE >= 2: byte offsets of the thread's four pixels in flight (thread-private:
                                                  // parked here instead of in four registers across the compute role)
constexpr int RS_LDS = RS_DOF + 512 * 16;         // 160 768 B

// Workgroup barrier for LDS traffic only: __syncthreads() also waits for every global load and store in flight (vmcnt(0)) — here
// loads issued two half-steps ahead and the raw-output stores must stay in flight across it.
__device__ __forceinline__ void rs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned rs_pack2h(float a, float b) {
  const f16x2 r = __builtin_convertvector(f32x2{a, b}, f16x2);    // v_cvt_pk_f16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, r);
}
}  // namespace

// Half-steps.  The band's output stream is cut into half-groups of 128 positions (4 tiles).  The eight waves form two SETS (waves
// 0-3 and 4-7: wave w and wave w + 4 share a SIMD) that alternate roles every half-step h, one workgroup barrier between half-steps:
//   compute set (h & 1):      the K loop of its four tiles of output half-group h — 54 MFMAs per wave and nothing else;
//   stage set (the other):    the epilogue of the tiles it multiplied in half-step h - 1 (un-scale, mask, raw stores, partial sums), then
//                             input half-group h + 4 from registers (loaded two half-steps ago) -> GroupNorm + ReLU -> float16 pieces ->
//                             ring, the output tables of half-group h + 1, and the loads of input half-group h + 6.
// So on every SIMD one wave multiplies while the other converts and stores — by construction, not by chance.  Output half-group h
// reads input half-groups h + 1 .. h + 3 (P + 1 <= 128: a tile's taps reach P + 1 positions back and ahead); h + 4 is being written:
// a ring of four half-groups.  The loop starts at h = -3 (the first three input half-groups) and ends at h = NH (the last epilogue).
template <int MODE>
__global__ __launch_bounds__(RS_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_rows32_kernel(const ConvX3Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = wave >> 2, sw = wave & 3, stid = tid & 255;
  const int H = p.H, W = p.W, P = W + 2;
  const int nbands = p.rs_bands, brows = p.rs_rows;
  const int nitems = p.B * nbands;
  const int dR = (2 * RS_HG) / P, dC = 2 * RS_HG - dR * P;   // a thread's next duty is 256 positions down the stream: R += dR, C += dC (carry below)

  // ---- the layer's B operand, resident: [step = tap * 2 + k-chunk][piece]
  u32x4 bres[18][2];
#pragma unroll
  for (int st = 0; st < 18; ++st)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc)
      bres[st][pc] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.wpk) + (size_t)st * 2048u + pc * 1024 + lane * 16);
  const float os = p.oscale_ptr != nullptr ? *p.oscale_ptr : p.oscale;

  // staging role of this thread inside its set: channel quad q (4 channels), positions jb + 32 k (k = 0..3) of a half-group
  const int q = stid & 7, jb = stid >> 3;
  int q_ = q, jb_ = jb, lane_ = lane;                              // the same, re-materialised per role (see the stage role)
  unsigned *otab = reinterpret_cast<unsigned *>(lds + RS_OTAB) + set * 2 * RS_HG;
  float *mtab = reinterpret_cast<float *>(lds + RS_MTAB) + set * 2 * RS_HG;
  float *red = reinterpret_cast<float *>(lds + RS_RED);
  float *sstab = reinterpret_cast<float *>(lds + RS_SS);
  // raw input by LDS-DMA: this wave's 4 KB of slot 0 / 1 of its set; a lane reads back the 16 bytes it asked for
  const unsigned rawbase = (unsigned)(RS_RAW + set * 32768 + sw * 4096);

  // one descriptor per tensor for the whole launch; a pixel's offset carries its sample's base
  const unsigned tbytes = (unsigned)p.B * (unsigned)(H * W * 128);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void *)(MODE == 2 ? p.res : p.x), 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxo = __builtin_amdgcn_make_buffer_rsrc((void *)(MODE >= 2 ? p.xout : p.y), 0, tbytes, 0x00020000);

  for (int item = (int)blockIdx.x; item < nitems; item += (int)gridDim.x) {
    const int n = item / nbands, band = item - n * nbands;
    const int rb = band * brows, re = min(H, rb + brows);
    const int O0 = (rb + 1) * P;                            // first output position: padded row rb + 1, padded column 0
    const int nout = (re - rb) * P;
    const int NH = (nout + RS_HG - 1) / RS_HG;              // output half-groups; input half-groups 1 .. NH + 2 are read
    const int nbase = n * H * W * 128;                      // the sample's plane inside the tensors (bytes; B H W 128 < 2^31: conv_rows32_plan)
    if (MODE >= 1 && tid < 64) sstab[tid] = tid < 32 ? p.in_scale[(size_t)n * 32 + tid] : p.in_shift[(size_t)n * 32 + tid - 32];
    const int r_lo = max(rb - 1, 0), r_hi = min(re, H - 1);        // image rows the band reads (halo included)
    // input stream: half-group g starts at flat position O0 - 256 + 128 g.  This thread's first duty: g = 1 (set 0) or 2 (set 1).
    // Tracked for the thread's pixel k = 0: padded row / column and the byte offset of its channel quad in the sample's plane (valid
    // or not); pixels k = 1..3 are 32 k positions further: (32 k) / P rows and (32 k) % P columns (scalars), one possible carry.
    int iR, iC, iD;
    {
      const int f = O0 - 2 * RS_HG + RS_HG * (1 + set) + jb + 256 * P;    // (+ 256 P keeps the dividend positive)
      iR = f / P - 256;
      iC = f - (iR + 256) * P;
      iD = (((iR - 1) * W + iC - 1) * 32 + 4 * q) * 4 + nbase;
    }
    // output tables: thread stid < 128 of a set owns position stid of the set's half-groups (set 0: 0, 2, ..; set 1: 1, 3, ..)
    int oR = 0, oC = 0, oD = 0;
    if (stid < RS_HG) {
      const int f = O0 + RS_HG * set + stid;
      oR = f / P;
      oC = f - oR * P;
      oD = ((oR - 1) * W + oC - 1) * 128 + nbase;
    }

    unsigned vok = 0, vokn = 0;                                    // bit k: pixel k of the duty being converted / of the loads in flight exists;
                                                                   // bit 4 + k (MODE >= 2): ... and lies in the band's own rows (this band writes it to xout)
    u32x4 *const dofp = reinterpret_cast<u32x4 *>(lds + RS_DOF) + tid;   // MODE >= 2: the pixels' byte offsets in the tensors (bit 31: absent)
    auto issue = [&](int slot) {                                   // LDS-DMA of the thread's four pixels of its next input half-group; the tracker advances
      vokn = 0;
      unsigned off[4];
      u32x4 dn = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r32 = (32 * k) / P, c32 = 32 * k - r32 * P;      // (scalar)
        const int cw = iC + c32 >= P ? 1 : 0;
        const int R = iR + r32 + cw, Cc = iC + c32 - cw * P;
        const bool ok = (unsigned)(R - 1 - r_lo) <= (unsigned)(r_hi - r_lo) && (unsigned)(Cc - 1) < (unsigned)W;
        // (the instruction offset k * 1024 moves the LDS destination AND the memory address: pre-decremented; absent pixels: out of
        //  range -> zeros)
        const unsigned dense = (unsigned)(iD + (32 * k - 2 * (r32 + cw)) * 128), absent = ok ? 0u : 0x80000000u;
        // (a select, not `| absent`: the top-right padding corner of SAMPLE 0 has dense index 0, so dense - 1024 k is "negative", bit 31
        //  is set either way and the instruction offset wraps it back INTO range — the pad pixel then read pixel (0, 0); masked by the
        //  input transform in modes 1-3, visible in mode 0: found in round 6 by a tap at 300 pairs)
        off[k] = ok ? dense - 1024u * k : 0x80000000u;
        vokn |= ok ? (1u << k) : 0u;
        if (MODE >= 2) {
          dn[k] = dense | absent;
          vokn |= (ok && (unsigned)(R - 1 - rb) < (unsigned)(re - rb)) ? (16u << k) : 0u;
        }
      }
      if (MODE >= 2) *dofp = dn;
      if (!(RS_DBG(p) & 1)) {
        const unsigned m0v = rawbase + (unsigned)slot * 16384u;
        asm volatile(
            "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
            "buffer_load_dwordx4 %0, %4, 0 offen lds\n\t"
            "buffer_load_dwordx4 %1, %4, 0 offen offset:1024 lds\n\t"
            "buffer_load_dwordx4 %2, %4, 0 offen offset:2048 lds\n\t"
            "buffer_load_dwordx4 %3, %4, 0 offen offset:3072 lds"
            ::"v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(rx), "s"(m0v)
            : "memory");
      }
      const int wrap = iC + dC >= P ? 1 : 0;
      iC += dC - wrap * P;
      iR += dR + wrap;
      iD += (2 * RS_HG - 2 * (dR + wrap)) * 128;                   // dense index = flat - 2 R - W - 1
    };
    f32x4 w[4];                                                    // MODE 2: the skip branch's pixels of this duty (register loads at the top of the stage role,
                                                                   // used at its end — the epilogue in between covers most of their latency; kept across the
                                                                   // compute role instead they would cost 16 registers where the budget is tightest)
    u32x4 dof = {0, 0, 0, 0};                                      // this duty's offsets (read back from the thread's LDS slot where they are needed)
    auto load_skip = [&]() {
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, dof[k], 0, 0));
      asm volatile("" ::: "memory");
    };
    auto convert_store = [&](int g, int slot) {                    // the landed pixels -> GroupNorm + ReLU -> float16 pieces -> ring slot of input half-group g
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (MODE >= 1) {
        sc = *reinterpret_cast<const f32x4 *>(sstab + 4 * q_);
        sh = *reinterpret_cast<const f32x4 *>(sstab + 32 + 4 * q_);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 vk = *reinterpret_cast<const f32x4 *>(lds + rawbase + (unsigned)slot * 16384u + 1024 * k + 16 * lane_);
        float f[4];
        const float vmk = (float)((vok >> k) & 1u);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = vk[e];
          // (zero padding AFTER the input transform; the float operations of conv_x3_kernel's stager / residual_kernel, in their order)
          if (MODE == 1) x = fmaxf(__builtin_fmaf(x, sc[e], sh[e]), 0.f) * vmk;     // the producer's GroupNorm + ReLU
          if (MODE == 2) x = fmaxf(__builtin_fmaf(x, sc[e], sh[e]) + w[k][e], 0.f) * vmk;   // block tail: relu(GN2(conv2) + skip), resnet.py:47-55
          if (MODE == 3) {                                                          // pooled stem keys: decode, |scale|, shift, ReLU
            int key = __builtin_bit_cast(int, x);
            key = key >= 0 ? key : key ^ 0x7fffffff;
            x = fmaxf(__builtin_fmaf(__builtin_bit_cast(float, key), __builtin_fabsf(sc[e]), sh[e]), 0.f) * vmk;
          }
          f[e] = x;
        }
        if (MODE >= 2 && !(RS_DBG(p) & 2))                         // the block output / pooled activations the next skip branch reads: own rows only
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{f[0], f[1], f[2], f[3]}), rxo,
                                                 dof[k] | (((vok >> (4 + k)) & 1u) ? 0u : 0x80000000u), 0, 0);
        u32x2 hi, lo;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float a = f[2 * e], b = f[2 * e + 1];
          const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
          hi[e] = __builtin_bit_cast(unsigned, h);
          lo[e] = rs_pack2h(a - (float)h[0], b - (float)h[1]);
        }
        const unsigned pos = (unsigned)((g & 3) * RS_HG + jb_ + 32 * k);
        *reinterpret_cast<u32x2 *>(lds + pos * RS_PITCH + 8 * q_) = hi;
        *reinterpret_cast<u32x2 *>(lds + pos * RS_PITCH + 64 + 8 * q_) = lo;
        if (k < 2 && (g & 3) == 0 && pos < (unsigned)RS_TAIL) {   // the ring's first positions once more behind its end (the mirror)
          *reinterpret_cast<u32x2 *>(lds + (pos + RS_RING) * RS_PITCH + 8 * q_) = hi;
          *reinterpret_cast<u32x2 *>(lds + (pos + RS_RING) * RS_PITCH + 64 + 8 * q_) = lo;
        }
      }
    };
    auto tables = [&](int hg) {                                    // output offsets / masks of the set's output half-group hg (threads 0..127 of the set)
      if (stid < RS_HG) {
        const bool ok = (unsigned)(oR - 1 - rb) < (unsigned)(re - rb) && (unsigned)(oC - 1) < (unsigned)W;
        otab[((hg >> 1) & 1) * RS_HG + stid] = ok ? (unsigned)oD : 0x80000000u;
        mtab[((hg >> 1) & 1) * RS_HG + stid] = ok ? os : 0.f;
        const int wrap = oC + dC >= P ? 1 : 0;
        oC += dC - wrap * P;
        oR += dR + wrap;
        oD += (2 * RS_HG - 2 * (dR + wrap)) * 128;
      }
    };

    float t1 = 0.f, t2 = 0.f;                                     // GroupNorm partial sums of channel (lane & 31), this wave's tiles
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // un-scale + mask, raw output, partial sums of the tile of output half-group hg held in `acc`
    constexpr bool EPI_IN_COMPUTE = MODE >= 2;
    auto epilogue = [&](int hg) {
      const int rr16 = lane_ >> 5;
      const unsigned tb = (unsigned)(((hg >> 1) & 1) * RS_HG + 32 * sw + 4 * rr16);
      const unsigned ch4 = (unsigned)((lane_ & 31) * 4);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const u32x4 ent = *reinterpret_cast<const u32x4 *>(otab + tb + 8 * g4);
        const f32x4 msk = *reinterpret_cast<const f32x4 *>(mtab + tb + 8 * g4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = acc[4 * g4 + e] * msk[e];
          if (!(RS_DBG(p) & 2)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), ry, ent[e] + ch4, 0, 0);   // bit 31: out of range, dropped
          t1 += x;
          t2 = __builtin_fmaf(x, x, t2);
        }
      }
    };
    __syncthreads();                                              // (the scale / shift table)
    issue(0);                                                     // the set's first input half-group (1 or 2): duty 0, raw slot 0
    int duty = 0;

#if PNVO_ROWS_ABL
    unsigned long long pc_compute = 0, pc_stage = 0, pc_bar = 0, pc_wait = 0;
#endif
    for (int h = -3; h <= NH; ++h) {
#if PNVO_ROWS_ABL
      const unsigned long long pt0 = __builtin_readcyclecounter();
      unsigned long long pt1 = pt0;
#endif
      if ((h & 1) == set) {
        // ---- compute role: the K loop of this wave's tile, output positions O0 + 128 h + 32 sw + (0..31)
        if (h >= 0 && h < NH && !(RS_DBG(p) & 4)) {
          // ring position of the window start of kernel row kh: centre - 1 + (kh - 1) P, wrapped; the window (32 + 2 positions)
          // continues into the mirror, so tap kw and k-chunk kc are immediates: kw * 144 + kc * 32 (+ 64: the lo pieces)
          const unsigned base = (unsigned)(RS_HG * h + 2 * RS_HG + 32 * sw - 1);
          asm volatile("" : "+v"(lane_));
          unsigned arow[3];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
            arow[kh] = (((base + (unsigned)((kh - 1) * P)) & (RS_RING - 1)) + (unsigned)(lane_ & 31)) * RS_PITCH + (unsigned)((lane_ >> 5) * 16);
          // two accumulators (even / odd steps): a chain of 54 MFMAs on ONE accumulator is paced by the MFMA's latency, not its
          // issue rate; summed once at the end (a fixed order: deterministic, float32-grade equal to conv_x3_kernel's single chain)
          // (MODE 2 keeps the skip branch's sixteen registers across this role instead: one chain)
          constexpr bool TWO_ACC = MODE != 2;
          f32x16 acc1;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
          u32x4 a[2], an[2];
          a[0] = *reinterpret_cast<const u32x4 *>(lds + arow[0]);
          a[1] = *reinterpret_cast<const u32x4 *>(lds + arow[0] + 64);
#pragma unroll
          for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) {
              const int nt = (st + 1) >> 1, nk = (st + 1) & 1;
              an[0] = *reinterpret_cast<const u32x4 *>(lds + arow[nt / 3] + (nt % 3) * RS_PITCH + nk * 32);
              an[1] = *reinterpret_cast<const u32x4 *>(lds + arow[nt / 3] + (nt % 3) * RS_PITCH + nk * 32 + 64);
            }
            f32x16 &c = (TWO_ACC && (st & 1)) ? acc1 : acc;
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, bres[st][0]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, bres[st][1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, bres[st][0]), c, 0, 0, 0);
            a[0] = an[0];
            a[1] = an[1];
          }
          if (TWO_ACC) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
          }
          if (EPI_IN_COMPUTE) epilogue(h);
        }
      } else {
        // ---- stage role.  Everything this wave has in flight — the DMA of this duty's pixels (issued at the top of its previous
        // duty, two half-steps ago) and the raw-output stores of that duty — is waited for in one go: loads and stores share a
        // counter and may retire out of order with respect to each other, so counting through the stores would not be safe.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if PNVO_ROWS_ABL
        pt1 = __builtin_readcyclecounter();
        pc_wait += pt1 - pt0;
#endif
        // (the lane geometry through an empty asm: LICM would hoist every address derived from it out of the half-step loop — a
        //  dozen registers that then spill, and a spill reload waits for everything this wave has in flight)
        asm volatile("" : "+v"(q_), "+v"(jb_), "+v"(lane_));
        vok = vokn;
        if (MODE >= 2) dof = *dofp;                                // (before issue() parks the next duty's there)
        if (MODE == 2) load_skip();
        if (h + 6 <= NH + 2) issue((duty + 1) & 1);                // the next duty's pixels: in flight for two half-steps
        // epilogue of the tile multiplied in half-step h - 1 (modes 0 / 1; the fuller stage roles of modes 2 / 3 leave it to the compute role) ...
        if (!EPI_IN_COMPUTE && h >= 1) epilogue(h - 1);
        // ... then input half-group h + 4 into the ring, the tables of output half-group h + 1
        if (h + 4 <= NH + 2 && !(RS_DBG(p) & 8)) convert_store(h + 4, duty & 1);

        if (h + 1 >= 0 && h + 1 < NH) tables(h + 1);
        ++duty;
      }
#if PNVO_ROWS_ABL
      const unsigned long long pt2 = __builtin_readcyclecounter();
      if ((h & 1) == set) pc_compute += pt2 - pt0; else pc_stage += pt2 - pt1;
#endif
      rs_barrier();
#if PNVO_ROWS_ABL
      pc_bar += __builtin_readcyclecounter() - pt2;
#endif
    }
#if PNVO_ROWS_ABL
    if (p.prof != nullptr && item == 7 && lane == 0) {
      unsigned long long *d = p.prof + wave * 8;
      d[0] = pc_compute; d[1] = pc_stage; d[2] = pc_bar; d[3] = pc_wait; d[4] = (unsigned long long)(NH + 4);
    }
#endif

    // ---- the band's GroupNorm partial sums: the two half-waves, then the eight waves in a fixed order -> one slot
    if (p.stats != nullptr) {
      t1 += __shfl_xor(t1, 32);
      t2 += __shfl_xor(t2, 32);
      if (lane < 32) *reinterpret_cast<f32x2 *>(red + (wave * 32 + lane) * 2) = f32x2{t1, t2};
      __syncthreads();
      if (wave == 0 && lane < 32) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const f32x2 o = *reinterpret_cast<const f32x2 *>(red + (w * 32 + lane) * 2);
          s1 += o[0];
          s2 += o[1];
        }
        float *dst = p.stats + (((size_t)n * p.slots + band) * 32 + lane) * 2;
        dst[0] = s1;
        dst[1] = s2;
        if (p.gn_scale != nullptr) {                              // one band per sample: finalise here (no launch)
          const int c = lane;
          const bool first = p.gn_mu != nullptr && c % p.gn_cpg == 0;
          const long gi = (long)n * (32 / p.gn_cpg) + c / p.gn_cpg;
          gn_finalize_lane(s1, s2, p.gn_cpg, p.gn_P, p.gn_eps, p.gn_gamma[c], p.gn_beta[c], p.gn_scale + (size_t)n * 32 + c,
                           p.gn_shift + (size_t)n * 32 + c, first ? p.gn_mu + gi : nullptr, first ? p.gn_rstd + gi : nullptr);
        }
      }
    }
    __syncthreads();                                              // ring, tables and scratch are free for the next band
  }
}

// Does the row-streaming kernel take this launch?  32 -> 32 channels, two float16 pieces, rows short enough for the ring
// (a tile's taps reach P + 1 positions back and ahead: P + 1 <= 128), and enough bands to occupy the chip.
bool conv_rows32_plan(ConvX3Args &a, int ks, int stride, int mode, int num_cus) {
  if (ks != 3 || stride != 1 || a.np != 2 || a.CIN != 32 || a.COUTP != 32 || a.in_absmax != nullptr) return false;
  if (mode < 0 || mode > 3 || (mode == 2 && a.res_scale != nullptr)) return false;      // (a downsample skip branch: 32 -> 32 stride-1 blocks have none)
  if (a.Ho != a.H || a.Wo != a.W || a.W + 3 > RS_HG || a.W < 8 || a.H < 8 || (long)a.B * a.H * a.W * 128 >= (1L << 31)) return false;
  // bands: whole samples from one per CU on (one wave of items); below that 2 or 4 bands of >= 12 rows (a band re-reads two halo rows:
  // narrower ones measured slower than the tile kernels); fewer items than CUs: the tile kernels spread better
  int bands = 1;
  while ((long)a.B * bands < num_cus && bands < 4 && a.H / (bands * 2) >= 12) bands *= 2;
  if ((long)a.B * bands < num_cus) return false;
  if (const char *e = std::getenv("PNVO_ROWS_DBG")) a.rs_dbg = std::atoi(e);
  a.rs_rows = (a.H + bands - 1) / bands;
  a.rs_bands = (a.H + a.rs_rows - 1) / a.rs_rows;
  a.slots = a.rs_bands;
  return true;
}

hipError_t launch_conv_rows32(const ConvX3Args &a, int mode, int num_cus, hipStream_t s) {
  static std::mutex attr_mu;
  static unsigned long long attr_seen = 0;
  if (pnvo_first_launch_on_device(attr_mu, attr_seen)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_rows32_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_rows32_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_rows32_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_rows32_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    if (e != hipSuccess) return e;
  }
  const long items = (long)a.B * a.rs_bands;
  const unsigned gx = (unsigned)(items < num_cus ? items : num_cus);
  ConvX3Args q = a;
#if PNVO_ROWS_ABL
  static unsigned long long *prof = nullptr;             // PNVO_ROWS_PROF=1: role cycles of the waves of item 7, printed per launch (syncs)
  if (std::getenv("PNVO_ROWS_PROF") != nullptr) {
    if (!prof && hipMalloc((void **)&prof, 8 * 8 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(prof, 0, 8 * 8 * 8, s);
    q.prof = prof;
  }
#endif
  if (mode == 0)
    hipLaunchKernelGGL((conv_rows32_kernel<0>), dim3(gx), dim3(RS_THREADS), RS_LDS, s, q);
  else if (mode == 1)
    hipLaunchKernelGGL((conv_rows32_kernel<1>), dim3(gx), dim3(RS_THREADS), RS_LDS, s, q);
  else if (mode == 2)
    hipLaunchKernelGGL((conv_rows32_kernel<2>), dim3(gx), dim3(RS_THREADS), RS_LDS, s, q);
  else if (mode == 3)
    hipLaunchKernelGGL((conv_rows32_kernel<3>), dim3(gx), dim3(RS_THREADS), RS_LDS, s, q);
  else
    return hipErrorInvalidValue;
#if PNVO_ROWS_ABL
  if (q.prof != nullptr) {
    unsigned long long h[64];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; w += 3)
      std::fprintf(stderr, "[pnvo] conv_rows32 wave %d: per half-step (cycles): compute role %.0f  stage role %.0f (+ wait for its loads/stores %.0f)  barrier %.0f  (%llu half-steps)\n",
                   w, 2.0 * h[8 * w] / h[8 * w + 4], 2.0 * h[8 * w + 1] / h[8 * w + 4], 2.0 * h[8 * w + 3] / h[8 * w + 4], (double)h[8 * w + 2] / h[8 * w + 4], h[8 * w + 4]);
  }
#endif
  return hipGetLastError();
}

}  // namespace pnvo
