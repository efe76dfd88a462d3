// stem_mx.hip — the fused 7x7 stride-2 stem on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), gfx950 only.
//
// Replaces, for every registered model, conv1 of the reference's backbone together with the input assembly and the
// RunningMeanAndVar whitening in front of it (vo_cnn.py:110-176, running_mean_and_var.py:62-63, resnet.py:156-163).
// Two numerical modes share the kernel:
//
//   PIECES = 3  ("f32x": float32 results from the bf16 pipe).  The A operand holds the RAW observation values, which the
//       reference's own contract makes exact in bf16: rgb is uint8-valued (0..255, 8 significant bits), discretised
//       depth is one-hot {0,1}, the "inside the image" indicator is {0,1}.  The B operand is the float32 weight with
//       1/(255 std) or 1/std folded in, split into THREE bf16 pieces hi + mid + lo == w exactly (3 x 8 = 24 significand
//       bits).  Every product a*piece is then exact in float32 and only the float32 summation order differs from an fp32
//       FMA chain.  The two float modalities (depth, top-down view: <= 4 channels) are split the same way on the A side
//       (x = x_hi + x_mid + x_lo); their nine cross terms are cut to the six that are not below float32 resolution
//       (x_hi w_hi, x_hi w_mid, x_hi w_lo, x_mid w_hi, x_mid w_mid, x_lo w_hi): the dropped ones are < 2^-24 of the
//       product.  Cost per tap and 32 pixels x 32 output channels: 7 MFMAs of 32x32x16 (224 cycles) against 48 fp32-MFMA
//       K-steps (768 cycles) for the dense fp32 stem.  A value that is NOT exact in bf16 where the contract says it is
//       (rgb that is not an integer, a soft depth code) raises the host-visible flag (pnvo_check_inputs) — such callers
//       select PNVO_STEM=dense.
//   PIECES = 2  ("f32h": float32-grade results from the float16 pipe, the inference default).  The same raw values are exact in
//       float16 too (integers up to 2048, {0,1}); the weight is split into TWO float16 pieces (22 significant bits, after a
//       power-of-two scale that keeps the second piece a normal number; rgb slots carry 2^-8 on the A side and 2^8 on the B
//       side for the same reason), the float modalities into two pieces on the A side: x0 w0 + x0 w1 + x1 w0, each product
//       within 3 * 2^-22.  5 MFMAs per tap instead of 7, the scale is undone exactly on the accumulators.
//   PIECES = 1  (native bf16, BASELINE config 3): one bf16 weight piece, bf16-rounded float modalities, 2 MFMAs per tap;
//       the indicator weight keeps two pieces (slots 30 and 31) because it carries -sum_c W mean_c/std_c, a large
//       cancelling term.
//
// Whitening never touches the activations: (x/div - mean)/std * W = x * (W/(div std)) - W mean/std, and the second term is
// the weight of an indicator channel that is 1 inside the image and 0 in the conv's zero padding (padding is applied
// AFTER whitening, vo_cnn.py:176-177), so the borders are exact.
//
// Work decomposition.  Workgroup = 4 waves, output tile = 8 rows x 16 columns = four 32-pixel M-tiles (2 rows x 16
// columns each), patch = 21 x 37 input pixels in LDS as 80 B per pixel: 32 bf16 K-slots (16 two-channel "units" in the
// order of the observation tensors; unit 15 = indicator) + 16 B of float-modality remainders [x_mid(4) | x_lo(4)].
// Patch columns are de-interleaved by parity (a stride-2 conv reads every other column) and the pixel pitch is an odd
// number of 16-byte units, so the 16 lanes a ds_read_b128 serves per cycle hit all 64 banks once.  The 49 taps are
// split over the 4 waves (K split): every wave keeps 4 x NT accumulators (all M-tiles), streams its taps' B fragments
// from L2 (one 1 KiB fragment feeds 4 MFMAs) and the partial sums meet in LDS in a fixed order (deterministic).
// 67 KB LDS -> 2 workgroups per CU: one stages (HBM -> registers -> bf16 -> LDS) while the other computes.
#include <cmath>
#include <cstring>

#include "stem_tile.h"

namespace pnvo {

namespace {
inline unsigned short host_bf16(float f) {        // round to nearest even
  unsigned u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
inline float host_bf16_to_float(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
}  // namespace


// RAW: the stager reads the SENSOR frames — uint8 rgb [B][2][H][W][3], float32 depth [B][2][H][W] (prev frame, cur frame) —
// and the 2-channel top-down view instead of the float32 observation-pair tensors: the pair concatenation, the uint8 -> float
// cast and _discretize_depth_func (base_trainer_with_vo.py:135-167,196-229) happen while staging, the 7.86 MB per pair of
// observation tensors are never written or read (0.92 MB of frames + 0.52 MB of top-down view instead).  The one-hot depth
// becomes two 2-byte LDS writes into a zeroed row: bin = the reference's (e_i <= d < e_{i+1}, last bin closed) from an
// 11-entry edge table in LDS, bit-identical to discretize_depth_kernel.
template <int PIECES, int NT, bool BF16OUT, bool POOL = false, bool RAW = false>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_mx_kernel(const StemMXArgs p) {
  constexpr bool EXTRA = PIECES >= 2;                      // float32(-grade) results: float modalities split on the A side too
  constexpr bool H = PIECES == 2;                          // float16 pieces (else bf16)
  constexpr int NFT = PIECES * 2 + (EXTRA ? 1 : 0);        // B fragments per (tap, N-tile)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float *red = reinterpret_cast<float *>(lds + RED_OFF);       // [4 waves][NT*32][2]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntg = (int)gridDim.y * NT;                     // N-tiles of the whole launch
  const int gy = blockIdx.y;

  // tile of this workgroup: consecutive ids of one XCD (id % 8) walk neighbouring tiles, so halos meet in that L2
  const int ntiles = p.B * p.tiles_x * p.tiles_y;
  const int chunk = (ntiles + 7) >> 3;
  int bid = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (bid >= ntiles) return;
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  // grouped forward (pnvo_forward_grouped_raw): the sample's action model picks the weights (one model: both ends INT_MAX)
  const int g_mdl = (p.grp_end0 > 0 && n >= p.grp_end0) + (p.grp_end1 > 0 && n >= p.grp_end1);   // (0: no such model)
  const unsigned short *g_wpk = g_mdl == 0 ? p.wpk : p.wpk_g[g_mdl - 1];
  const int ho0 = ty * TH, wo0 = tx * TW;
  const int hi_base = 2 * ho0 - 3, wi_base = 2 * wo0 - 3;

  const bool prof = p.prof != nullptr && (blockIdx.x % 61) == 0;   // sampled: the atomics below perturb the memory pipe                  // PNVO_STEM_DBG=9: per-phase cycles of wave 0 (s_memtime)
  const unsigned long long tp0 = prof ? __builtin_readcyclecounter() : 0;
  // ---------------------------------------------------------------- staging: observation tensors -> bf16 patch in LDS
  // One thread per patch pixel (777 pixels: three full rounds + nine pixels for wave 0): the per-pixel work — index
  // arithmetic, bounds, four tensor addresses — is paid once per 120 B of input instead of once per 8 B.  What bounds
  // this phase is the INSTRUCTION COUNT (it shares the SIMDs with the other workgroup's MFMAs: measured ~4.5 cycles per
  // instruction of either wave), not bytes: 16-byte loads, one v_cvt_pk_bf16_f32 per channel pair, 16-byte LDS writes.
  // Fixed K-slot layout (absent modalities stay zero, their weights are zero):
  //   slots 0-19 discretised depth | 20-25 rgb | 26-27 depth | 28-29 top-down view | 30-31 indicator
  // Out-of-image pixels read a page of zeros (zero padding AFTER whitening); a value that must be exact in bf16 and is not
  // (low 16 bits set) raises the host-visible flag.
  if (RAW) {
    float *etab = reinterpret_cast<float *>(lds + RED_OFF + 4 * NT * 32 * 2 * 4);   // bin edges e_0 .. e_10 (+ 1 pad)
    if (threadIdx.x < 12) etab[threadIdx.x] = p.edges[threadIdx.x];
    __syncthreads();
    const float *zp = p.zero_page;
    const unsigned char *zpb = reinterpret_cast<const unsigned char *>(zp);
    const long fpix = (long)p.H * p.W;
    const bool use_rgb = p.raw_rgb != nullptr, use_d = (p.raw_flags & 1) != 0, use_dd = (p.raw_flags & 2) != 0;
    const float *b_t = p.src[3] ? p.src[3] + (long)n * fpix * 2 : nullptr;
    bool bad_depth = false;
#pragma unroll
    for (int r0 = 0; r0 < 4; r0 += 2) {
      unsigned rgbw[2][2];
      float dv[2][2];
      f32x2 vt[2];
      bool inb[2];
      unsigned ldsoff[2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int pix = (r0 + rr) * NTHREADS + (int)threadIdx.x;
        const int py = pix / PW, px = pix - py * PW;
        const int hi = hi_base + py, wi = wi_base + px;
        const bool in = pix < NPIX && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
        const int e = in ? hi * p.W + wi : 0;
        inb[rr] = in;
        ldsoff[rr] = (unsigned)(py * ROW + (px & 1) * PAR + (px >> 1) * PITCH);
        if (r0 + rr == 3 && wave != 0) continue;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const long gi = ((long)n * 2 + f) * fpix + e;                  // pixel index over all frames
          // three bytes at 3 gi, fetched as ONE (unaligned) dword that starts a byte early except at the very first pixel of
          // the tensor: never before its start, never past its end
          const long boff = 3 * gi - (gi > 0 ? 1 : 0);
          const unsigned char *a_rgb = (in && use_rgb) ? p.raw_rgb + boff : zpb;
          unsigned wv;
          __builtin_memcpy(&wv, a_rgb, 4);
          rgbw[rr][f] = (in && use_rgb && gi > 0) ? (wv >> 8) : wv;
          const float *a_d = in ? p.raw_depth + gi : zp;
          dv[rr][f] = *a_d;
        }
        const float *a_t = (in && b_t) ? b_t + (long)e * 2 : zp;
        vt[rr] = *reinterpret_cast<const f32x2 *>(a_t);
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        if (r0 + rr == 3 && wave != 0) continue;
        const int pix = (r0 + rr) * NTHREADS + (int)threadIdx.x;
        if (pix >= NPIX) continue;
        // bins of the two frames (valid only inside the image and for depth in [0, 1])
        int bidx[2];
        bool bok[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const float d = dv[rr][f];
          int g = (int)(d * 10.0f);
          g = min(max(g, 0), 9);
          const float lo = etab[g], hi = etab[g + 1];
          bidx[f] = g - (d < lo ? 1 : 0) + ((d >= hi && g < 9) ? 1 : 0);
          bok[f] = d >= 0.f && d <= 1.f;
          bad_depth = bad_depth || (inb[rr] && !bok[f]);
        }
        unsigned w[16];
#pragma unroll
        for (int c = 0; c < 10; ++c) w[c] = 0u;
        {
          const unsigned a0 = rgbw[rr][0], a1 = rgbw[rr][1];
          const float pr = (float)(a0 & 0xffu), pg = (float)((a0 >> 8) & 0xffu), pb = (float)((a0 >> 16) & 0xffu);
          const float cr = (float)(a1 & 0xffu), cg = (float)((a1 >> 8) & 0xffu), cb = (float)((a1 >> 16) & 0xffu);
          if (H) {
            w[10] = pack_f16(pr * 0.00390625f, pg * 0.00390625f);
            w[11] = pack_f16(pb * 0.00390625f, cr * 0.00390625f);
            w[12] = pack_f16(cg * 0.00390625f, cb * 0.00390625f);
          } else {
            w[10] = pack_bf16(pr, pg);
            w[11] = pack_bf16(pb, cr);
            w[12] = pack_bf16(cg, cb);
          }
        }
        const float d0 = use_d ? dv[rr][0] : 0.f, d1 = use_d ? dv[rr][1] : 0.f;
        w[13] = pack_pair<H>(d0, d1);
        w[14] = pack_pair<H>(vt[rr][0], vt[rr][1]);
        w[15] = inb[rr] ? (H ? 0x3c003c00u : 0x3f803f80u) : 0u;
        unsigned char *dst = lds + ldsoff[rr];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<u32x4 *>(dst + 16 * q) = u32x4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
        if (use_dd && inb[rr]) {                                       // the one-hot depth: one 1.0 per frame
          const unsigned short one = H ? 0x3c00 : 0x3f80;
          if (bok[0]) *reinterpret_cast<unsigned short *>(dst + 2 * bidx[0]) = one;
          if (bok[1]) *reinterpret_cast<unsigned short *>(dst + 20 + 2 * bidx[1]) = one;
        }
        if (H) {
          const f16x2 hd = __builtin_bit_cast(f16x2, w[13]), ht = __builtin_bit_cast(f16x2, w[14]);
          const unsigned md = pack_f16(d0 - (float)hd[0], d1 - (float)hd[1]);
          const unsigned mt = pack_f16(vt[rr][0] - (float)ht[0], vt[rr][1] - (float)ht[1]);
          *reinterpret_cast<u32x4 *>(dst + 64) = u32x4{md, mt, 0u, 0u};
        } else if (EXTRA) {
          const float e0 = d0 - bf16_lo(w[13]), e1 = d1 - bf16_hi(w[13]);
          const float t0 = vt[rr][0] - bf16_lo(w[14]), t1 = vt[rr][1] - bf16_hi(w[14]);
          const unsigned md = pack_bf16(e0, e1), mt = pack_bf16(t0, t1);
          const unsigned ld = pack_bf16(e0 - bf16_lo(md), e1 - bf16_hi(md));
          const unsigned lt = pack_bf16(t0 - bf16_lo(mt), t1 - bf16_hi(mt));
          *reinterpret_cast<u32x4 *>(dst + 64) = u32x4{md, mt, ld, lt};
        }
      }
    }
    if (bad_depth && p.raw_err != nullptr) *p.raw_err = 1;              // the reference asserts depth in [0, 1] (:136-137)
  } else {
    const float *zp = p.zero_page;
    const float *b_rgb = p.src[0] ? p.src[0] + (long)n * p.H * p.W * 6 : nullptr;
    const float *b_d = p.src[1] ? p.src[1] + (long)n * p.H * p.W * 2 : nullptr;
    const float *b_dd = p.src[2] ? p.src[2] + (long)n * p.H * p.W * 20 : nullptr;
    const float *b_t = p.src[3] ? p.src[3] + (long)n * p.H * p.W * 2 : nullptr;
    unsigned lowbits = 0;
#pragma unroll
    for (int r0 = 0; r0 < 4; r0 += 2) {
      f32x4 vdd[2][5], vr4[2];
      f32x2 vr2[2], vd[2], vt[2];
      bool inb[2];
      unsigned ldsoff[2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int pix = (r0 + rr) * NTHREADS + (int)threadIdx.x;
        const int py = pix / PW, px = pix - py * PW;
        const int hi = hi_base + py, wi = wi_base + px;
        const bool in = pix < NPIX && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
        const int e = in ? hi * p.W + wi : 0;                          // pixel index inside the image
        inb[rr] = in;
        ldsoff[rr] = (unsigned)(py * ROW + (px & 1) * PAR + (px >> 1) * PITCH);
        if (r0 + rr == 3 && wave != 0) continue;                       // the last nine pixels belong to wave 0
        // absent modalities and out-of-image pixels read the zero page: no branches, no register initialisation
        const float *a_dd = (in && b_dd) ? b_dd + (long)e * 20 : zp;
        const float *a_rgb = (in && b_rgb) ? b_rgb + (long)e * 6 : zp;   // 24 B per pixel: 8-byte aligned
        const float *a_d = (in && b_d) ? b_d + (long)e * 2 : zp;
        const float *a_t = (in && b_t) ? b_t + (long)e * 2 : zp;
#pragma unroll
        for (int c = 0; c < 5; ++c) vdd[rr][c] = *reinterpret_cast<const f32x4 *>(a_dd + 4 * c);
        {
          const f32x2 q0 = *reinterpret_cast<const f32x2 *>(a_rgb), q1 = *reinterpret_cast<const f32x2 *>(a_rgb + 2);
          vr4[rr] = f32x4{q0[0], q0[1], q1[0], q1[1]};
          vr2[rr] = *reinterpret_cast<const f32x2 *>(a_rgb + 4);
        }
        vd[rr] = *reinterpret_cast<const f32x2 *>(a_d);
        vt[rr] = *reinterpret_cast<const f32x2 *>(a_t);
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        if (r0 + rr == 3 && wave != 0) continue;
        const int pix = (r0 + rr) * NTHREADS + (int)threadIdx.x;
        if (pix >= NPIX) continue;
        unsigned w[16];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          w[2 * c] = pack_pair<H>(vdd[rr][c][0], vdd[rr][c][1]);
          w[2 * c + 1] = pack_pair<H>(vdd[rr][c][2], vdd[rr][c][3]);
          if (EXTRA) {
            const float f0 = vdd[rr][c][0], f1 = vdd[rr][c][1], f2 = vdd[rr][c][2], f3 = vdd[rr][c][3];
            lowbits |= __builtin_bit_cast(unsigned, f0) | __builtin_bit_cast(unsigned, f1);
            lowbits |= __builtin_bit_cast(unsigned, f2) | __builtin_bit_cast(unsigned, f3);
          }
        }
        if (H) {                                                       // rgb * 2^-8 here, 2^8 in the packed weights (exact)
          w[10] = pack_f16(vr4[rr][0] * 0.00390625f, vr4[rr][1] * 0.00390625f);
          w[11] = pack_f16(vr4[rr][2] * 0.00390625f, vr4[rr][3] * 0.00390625f);
          w[12] = pack_f16(vr2[rr][0] * 0.00390625f, vr2[rr][1] * 0.00390625f);
        } else {
          w[10] = pack_bf16(vr4[rr][0], vr4[rr][1]);
          w[11] = pack_bf16(vr4[rr][2], vr4[rr][3]);
          w[12] = pack_bf16(vr2[rr][0], vr2[rr][1]);
        }
        if (EXTRA) {
          const float f0 = vr4[rr][0], f1 = vr4[rr][1], f2 = vr4[rr][2], f3 = vr4[rr][3], f4 = vr2[rr][0], f5 = vr2[rr][1];
          lowbits |= __builtin_bit_cast(unsigned, f0) | __builtin_bit_cast(unsigned, f1);
          lowbits |= __builtin_bit_cast(unsigned, f2) | __builtin_bit_cast(unsigned, f3);
          lowbits |= __builtin_bit_cast(unsigned, f4) | __builtin_bit_cast(unsigned, f5);
        }
        w[13] = pack_pair<H>(vd[rr][0], vd[rr][1]);
        w[14] = pack_pair<H>(vt[rr][0], vt[rr][1]);
        w[15] = inb[rr] ? (H ? 0x3c003c00u : 0x3f803f80u) : 0u;        // indicator (two slots): 1.0 in float16 / bf16
        unsigned char *dst = lds + ldsoff[rr];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<u32x4 *>(dst + 16 * q) = u32x4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
        if (H) {                                                       // float modalities: x = x0 + x1 to 22 bits
          const f16x2 hd = __builtin_bit_cast(f16x2, w[13]), ht = __builtin_bit_cast(f16x2, w[14]);
          const unsigned md = pack_f16(vd[rr][0] - (float)hd[0], vd[rr][1] - (float)hd[1]);
          const unsigned mt = pack_f16(vt[rr][0] - (float)ht[0], vt[rr][1] - (float)ht[1]);
          *reinterpret_cast<u32x4 *>(dst + 64) = u32x4{md, mt, 0u, 0u};
        } else if (EXTRA) {                                            // float modalities: x - hi = mid + lo (exact)
          const float d0 = vd[rr][0] - bf16_lo(w[13]), d1 = vd[rr][1] - bf16_hi(w[13]);
          const float t0 = vt[rr][0] - bf16_lo(w[14]), t1 = vt[rr][1] - bf16_hi(w[14]);
          const unsigned md = pack_bf16(d0, d1), mt = pack_bf16(t0, t1);
          const unsigned ld = pack_bf16(d0 - bf16_lo(md), d1 - bf16_hi(md));
          const unsigned lt = pack_bf16(t0 - bf16_lo(mt), t1 - bf16_hi(mt));
          *reinterpret_cast<u32x4 *>(dst + 64) = u32x4{md, mt, ld, lt};
        }
      }
    }
    // exact in bf16: low 16 mantissa bits clear; exact in float16 (normal range — the contract's values are integers <= 255
    // and {0, 1}): low 13 bits clear
    if (EXTRA && (lowbits & (H ? 0x1fffu : 0xffffu)) != 0 && p.bad_input != nullptr) *p.bad_input = 1;
  }
  const unsigned long long tp1 = prof ? __builtin_readcyclecounter() : 0;
  __syncthreads();
  const unsigned long long tp2 = prof ? __builtin_readcyclecounter() : 0;

  // ---------------------------------------------------------------- K loop: this wave's taps over all four M-tiles
  f32x16 acc[4][NT];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.f;
  {
    const int rr = (lane & 31) >> 4, c = lane & 15, h = lane >> 5;
    const unsigned baseA = (unsigned)(2 * rr * ROW + c * PITCH + h * 16);
    const unsigned baseX = (unsigned)(2 * rr * ROW + c * PITCH + 64);
    const int nfrag = NFT * ntg;                          // fragments per tap in the packed array
    auto loadB = [&](int tap, u32x4 *b) {                 // uniform base (SGPRs) + lane offset + immediates
      const u32x4 *wt = reinterpret_cast<const u32x4 *>(g_wpk) + ((long)tap * nfrag + gy * NT) * 64;
#pragma unroll
      for (int f = 0; f < NFT; ++f)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int q = (f * ntg + nt) * 64;                  // fragment offset (16-byte units), uniform
          const u32x4 *base = wt + (q & ~255);                //  split so that the remainder fits the load's immediate (< 4 KiB)
          b[f * NT + nt] = base[(q & 255) + lane];
        }
    };
    auto tapoff = [&](int tap) { return tap_lds_offset(tap); };
    auto loadA0 = [&](int tap, u32x4 *a) {                 // K-slots 0..15 of the four M-tiles
      const unsigned toff = tapoff(tap);
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const u32x4 *>(lds + baseA + toff + m * 4 * ROW);
    };
    auto loadA1 = [&](int tap, u32x4 *a, u32x4 *ax) {      // K-slots 16..31 and the float-modality remainders
      const unsigned toff = tapoff(tap);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        a[m] = *reinterpret_cast<const u32x4 *>(lds + baseA + toff + m * 4 * ROW + 32);
        if (EXTRA) ax[m] = *reinterpret_cast<const u32x4 *>(lds + baseX + toff + m * 4 * ROW);
      }
    };
    // Software pipeline.  B fragments come from L2 (~1 us under load): fetched TWO taps ahead into three register sets in
    // rotation.  A chunks come from LDS: chunk 1 / the remainders of a tap at its start (first needed 12 / 24 MFMAs
    // later), chunk 0 of the NEXT tap as soon as this tap's chunk-0 MFMAs have issued (16 MFMAs before its first use) —
    // one register set each.  sched_barriers keep the fetches where they are (the scheduler would sink them).
    // Taps of this wave: wave, wave+4, ... wave+44; tap 48 goes to wave 3 (wave 0 staged the nine extra pixels).
    u32x4 b0[NFT * NT], b1[NFT * NT], b2[NFT * NT], a0[4], a1[4], ax[4];
    auto mfmas_q = [&](int q, const u32x4 *aq, const u32x4 *b) {
#pragma unroll
      for (int pc = 0; pc < PIECES; ++pc)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[m][nt] = H ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aq[m]),
                                                                    __builtin_bit_cast(f16x8, b[(pc * 2 + q) * NT + nt]), acc[m][nt], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq[m]),
                                                                     __builtin_bit_cast(bf16x8, b[(pc * 2 + q) * NT + nt]), acc[m][nt], 0, 0, 0);
    };
    auto mfmas_x = [&](const u32x4 *b) {
      if (EXTRA) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[m][nt] = H ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ax[m]),
                                                                    __builtin_bit_cast(f16x8, b[(PIECES * 2) * NT + nt]), acc[m][nt], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ax[m]),
                                                                     __builtin_bit_cast(bf16x8, b[(PIECES * 2) * NT + nt]), acc[m][nt], 0, 0, 0);
      }
    };
    auto tapof = [&](int i) { return i < 12 ? wave + 4 * i : (wave == 3 ? 48 : wave + 44); };   // (past the end: harmless repeats)
    // One scheduling region per tap with the fetches spread between the MFMAs (round 4): an issue slot is ~4 cycles, an MFMA 32 —
    // a burst of ~25 fetch / address instructions in front of the MFMAs drains the matrix pipe, <= 5 per gap are free.
    constexpr int TM = (PIECES * 2 + (EXTRA ? 1 : 0)) * NT * 4, NBF = NFT * NT, XM = TM - 4 - NBF - 8;
    auto step = [&](int i, const u32x4 *bcur, u32x4 *bnext2) {
      if (XM < 0) {
        loadA1(tapof(i), a1, ax);
        loadB(tapof(i + 2), bnext2);
        __builtin_amdgcn_sched_barrier(0);
        mfmas_q(0, a0, bcur);
        __builtin_amdgcn_sched_barrier(0);
        loadA0(tapof(i + 1), a0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas_q(1, a1, bcur);
        mfmas_x(bcur);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        __builtin_amdgcn_sched_barrier(0);
        loadA1(tapof(i), a1, ax);
        loadB(tapof(i + 2), bnext2);
        mfmas_q(0, a0, bcur);
        u32x4 a0n[4];
        loadA0(tapof(i + 1), a0n);
        mfmas_q(1, a1, bcur);
        mfmas_x(bcur);
#pragma unroll
        for (int m = 0; m < 4; ++m) a0[m] = a0n[m];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                         // the A1 / AX fragment reads of this tap
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, EXTRA ? 2 : 1, 0);
        }
#pragma unroll
        for (int k = 0; k < NBF; ++k) {                       // the weight-fragment loads of a later tap
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if (XM > 0) __builtin_amdgcn_sched_group_barrier(0x008, XM, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                         // the next tap's A0 fragments
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    loadB(tapof(0), b0);
    loadB(tapof(1), b1);
    loadA0(tapof(0), a0);
#pragma unroll 1
    for (int i = 0; i < 12; i += 3) {
      step(i, b0, b2);
      step(i + 1, b1, b0);
      step(i + 2, b2, b1);
    }
    if (wave == 3) {
      loadA1(48, a1, ax);
      __builtin_amdgcn_sched_barrier(0);
      mfmas_q(0, a0, b0);
      mfmas_q(1, a1, b0);
      mfmas_x(b0);
    }
  }

  const unsigned long long tp3 = prof ? __builtin_readcyclecounter() : 0;
  // ---------------------------------------------------------------- K-split reduction through LDS (fixed order) + epilogue
  // Every wave publishes its partial sums of all four M-tiles (64 KB per N-tile: the patch is dead by now) and reduces
  // M-tile `wave` in wave order 0,1,2,3 — no data-dependent register selection, bit-reproducible.
  const int rr16 = lane >> 5;                             // accumulator row = (r & 3) + 8 (r >> 2) + 4 rr16
  const bool full = ho0 + TH <= p.Ho && wo0 + TW <= p.Wo; // the whole tile is inside the output (wave-uniform)
  const float oscale = (H && p.oscale_ptr != nullptr) ? *p.oscale_ptr : (g_mdl == 0 ? p.oscale : p.oscale_g[g_mdl - 1]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    __syncthreads();                                      // patch (or the previous N-tile's exchange) no longer read
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<f32x4 *>(lds + (((m * 4 + wave) * 4 + rq) * 64 + lane) * 16) =
            f32x4{acc[m][nt][4 * rq], acc[m][nt][4 * rq + 1], acc[m][nt][4 * rq + 2], acc[m][nt][4 * rq + 3]};
    __syncthreads();
    f32x16 tot;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(lds + (((wave * 4 + s4) * 4 + rq) * 64 + lane) * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[4 * rq + e] = s4 == 0 ? t[e] : tot[4 * rq + e] + t[e];
      }
    if (H) {                                              // undo the weights' power-of-two scale (exact)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[r] *= oscale;
    }
    // epilogue of M-tile `wave`: rows 2*wave, 2*wave+1 of the tile; lane = output channel, registers = pixels
    const int g = gy * NT + nt;                           // N-tile of the launch
    const int co = p.y_coff[g] + (lane & 31);
    const long rowpix = ((long)n * p.Ho + ho0 + 2 * wave) * p.Wo + wo0 + 4 * rr16;   // pixel of register 0
    float s1 = 0.f, s2 = 0.f;
    if (POOL) {
      // MaxPool2d(3, 2, 1) of the stem (resnet.py:168) without a pass of its own.  relu(x*scale+shift) is monotone in x, rising
      // when the GroupNorm weight is >= 0 and falling otherwise, so the window maximum of the activation is the activation of
      // the window maximum of sgn(gamma)*x: the tile pools sgn*x through LDS and writes ORDER-PRESERVING INTEGER KEYS of the
      // maxima — plain stores for windows inside the tile, integer atomic max (exact, order-free) for windows that straddle
      // tiles; the consumer (conv_x3 MODE 3) decodes, applies |scale|, shift and ReLU: the same bits as pooling afterwards.
      // The raw stem output is never written.  GroupNorm partial sums come from the full-resolution values as always.
      const float sgn = (g_mdl == 0 ? p.pool_gamma : p.pool_gamma_g[g_mdl - 1])[co] < 0.f ? -1.f : 1.f;
      __syncthreads();                                    // every wave has its `tot`: the exchange area is free
      float *pb = reinterpret_cast<float *>(lds);        // [8 rows][16 cols][33: 32 channels + 1 pad]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * rr16;
        const int row = 2 * wave + (i >> 4), col = i & 15;
        const bool ok = ho0 + row < p.Ho && wo0 + col < p.Wo;
        const float v = ok ? tot[r] : 0.f;
        pb[(row * 16 + col) * 33 + (lane & 31)] = ok ? sgn * tot[r] : -__builtin_inff();
        s1 += v;
        s2 = __builtin_fmaf(v, v, s2);
      }
      __syncthreads();
      // 5 x 9 pooled pixels touch the 8 x 16 tile.  Thread = (pooled column pj 0..7, channel): the three-column maxima of the
      // eight tile rows first, then the five pooled rows from them; the ninth column (tile column 15 only) afterwards.
      const int ch = threadIdx.x & 31, pj = threadIdx.x >> 5;
      const int Ib = ho0 >> 1, Jb = wo0 >> 1;
      int *const pool0 = p.pool + (long)n * p.Hp * p.Wp * p.y_cstride + p.y_coff[g] + ch;
      auto emit = [&](int I, int J, float mx, bool inside) {
        if (I >= p.Hp || J >= p.Wp) return;
        int key = __builtin_bit_cast(int, mx);
        key = key >= 0 ? key : key ^ 0x7fffffff;
        int *dst = pool0 + (I * p.Wp + J) * p.y_cstride;
        if (inside)
          *dst = key;
        else
          atomicMax(dst, key);
      };
      {
        const int c0 = pj > 0 ? 2 * pj - 1 : 0, c1 = 2 * pj, c2 = 2 * pj + 1;   // (pj = 0: column -1 belongs to the left tile)
        float cm[8];
#pragma unroll
        for (int lr = 0; lr < 8; ++lr)
          cm[lr] = fmaxf(fmaxf(pb[(lr * 16 + c0) * 33 + ch], pb[(lr * 16 + c1) * 33 + ch]), pb[(lr * 16 + c2) * 33 + ch]);
        const bool colin = pj >= 1 || wo0 == 0;           // the window's columns are all this tile's (or padding)
        emit(Ib + 0, Jb + pj, fmaxf(cm[0], cm[1]), colin && ho0 == 0);
        emit(Ib + 1, Jb + pj, fmaxf(fmaxf(cm[1], cm[2]), cm[3]), colin);
        emit(Ib + 2, Jb + pj, fmaxf(fmaxf(cm[3], cm[4]), cm[5]), colin);
        emit(Ib + 3, Jb + pj, fmaxf(fmaxf(cm[5], cm[6]), cm[7]), colin);
        emit(Ib + 4, Jb + pj, cm[7], false);
      }
      if (threadIdx.x < 160) {                            // ninth pooled column: tile column 15, always shared with the right tile
        const int pi = threadIdx.x >> 5;
        float mx = -__builtin_inff();
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr) {
          const int lr = 2 * pi + dr;
          if (lr >= 0 && lr < 8) mx = fmaxf(mx, pb[(lr * 16 + 15) * 33 + ch]);
        }
        emit(Ib + pi, Jb + 8, mx, false);
      }
    } else if (full) {
      if (BF16OUT) {
        __bf16 *y0 = reinterpret_cast<__bf16 *>(p.y[g]) + rowpix * p.y_cstride + co;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dpix = (r >> 3) * p.Wo + (r & 3) + 8 * ((r >> 2) & 1);
          y0[(long)dpix * p.y_cstride] = (__bf16)tot[r];
          s1 += tot[r];
          s2 = __builtin_fmaf(tot[r], tot[r], s2);
        }
      } else {
        float *y0 = reinterpret_cast<float *>(p.y[g]) + rowpix * p.y_cstride + co;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dpix = (r >> 3) * p.Wo + (r & 3) + 8 * ((r >> 2) & 1);
          y0[(long)dpix * p.y_cstride] = tot[r];
          s1 += tot[r];
          s2 = __builtin_fmaf(tot[r], tot[r], s2);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * rr16;
        const int ho = ho0 + 2 * wave + (i >> 4), wo = wo0 + (i & 15);
        const bool ok = ho < p.Ho && wo < p.Wo;
        const float v = ok ? tot[r] : 0.f;
        if (ok) {
          const long off = (((long)n * p.Ho + ho) * p.Wo + wo) * p.y_cstride + co;
          if (BF16OUT)
            reinterpret_cast<__bf16 *>(p.y[g])[off] = (__bf16)v;
          else
            reinterpret_cast<float *>(p.y[g])[off] = v;
        }
        s1 += v;
        s2 = __builtin_fmaf(v, v, s2);
      }
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lane < 32) {
      red[((wave * NT + nt) * 32 + lane) * 2] = s1;
      red[((wave * NT + nt) * 32 + lane) * 2 + 1] = s2;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < NT * 32) {
    const int nt = threadIdx.x >> 5, c = threadIdx.x & 31;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s1 += red[((w * NT + nt) * 32 + c) * 2];
      s2 += red[((w * NT + nt) * 32 + c) * 2 + 1];
    }
    const int g = gy * NT + nt;
    const int slot = ty * p.tiles_x + tx;
    float *dst = p.stats[g] + (((long)n * p.slots + slot) * p.stats_cstride + p.y_coff[g] + c) * 2;
    dst[0] = s1;
    dst[1] = s2;
  }
  if (prof && (threadIdx.x & 63) == 0) {
    const unsigned long long tp4 = __builtin_readcyclecounter();
    unsigned long long *q = p.prof + 8 * wave;
    atomicAdd(q + 0, tp1 - tp0);     // staging (loads + convert + LDS writes)
    atomicAdd(q + 1, tp2 - tp1);     // barrier wait
    atomicAdd(q + 2, tp3 - tp2);     // K loop
    atomicAdd(q + 3, tp4 - tp3);     // reduction + epilogue
    atomicAdd(q + 4, 1ull);
  }
}


int stem_mx_slots(int Ho, int Wo) { return ((Ho + TH - 1) / TH) * ((Wo + TW - 1) / TW); }

size_t stem_mx_packed_u16(int pieces, int ntiles) { return (size_t)49 * (pieces * 2 + (pieces >= 2 ? 1 : 0)) * ntiles * 64 * 8; }

// pieces = 2: float16 pieces of scale * wk (rgb slots 20..25 times 2^8: the stager hands rgb * 2^-8 over); the extra fragment
// pairs the A bytes [x1(4) | 0] with [w0(x0..x3) | 0] in the first K half.  Returns 1 / scale (StemMXArgs::oscale).
float pack_stem_mx_weight_h(const float *wk, int cout, const int *xslot, unsigned short *out) {
  const int ntl = cout / 32, nft = 5;
  auto slotw = [&](int co, int slot, int tap) {
    const float v = wk[((size_t)co * 32 + slot) * 49 + tap];
    return (slot >= 20 && slot <= 25) ? v * 256.f : v;
  };
  float mx = 0.f;
  for (int co = 0; co < cout; ++co)
    for (int slot = 0; slot < 31; ++slot)
      for (int tap = 0; tap < 49; ++tap) mx = std::fmax(mx, std::fabs(slotw(co, slot, tap)));
  int e = 0;
  if (mx > 0.f) std::frexp(mx, &e);
  const float scale = std::ldexp(1.0f, 12 - e), inv = std::ldexp(1.0f, e - 12);
  auto h16 = [](float f) {
    const _Float16 h = (_Float16)f;
    unsigned short u;
    std::memcpy(&u, &h, 2);
    return u;
  };
  auto f32 = [](unsigned short u) {
    _Float16 h;
    std::memcpy(&h, &u, 2);
    return (float)h;
  };
  auto piece = [&](float v, int pc) {
    const unsigned short a = h16(v);
    return pc == 0 ? a : h16(v - f32(a));
  };
  for (int tap = 0; tap < 49; ++tap)
    for (int f = 0; f < nft; ++f)
      for (int nt = 0; nt < ntl; ++nt)
        for (int ln = 0; ln < 64; ++ln)
          for (int j = 0; j < 8; ++j) {
            const int kh = ln >> 5, co = nt * 32 + (ln & 31);
            unsigned short val = 0;
            if (f < 4) {
              const int pc = f / 2, q = f % 2, slot = 16 * q + 8 * kh + j;
              if (slot < 31) val = piece(slotw(co, slot, tap) * scale, pc);
            } else if (kh == 0 && j < 4 && xslot[j] >= 0) {
              val = piece(slotw(co, xslot[j], tap) * scale, 0);
            }
            out[((((size_t)tap * nft + f) * ntl + nt) * 64 + ln) * 8 + j] = val;
          }
  return inv;
}

// B operand of the stem:  out[tap][fragment][N-tile][lane = kh*32 + n][8 bf16]  (kh = K half of the lane)
//   fragment (piece pc, chunk q):  value = piece pc of  wk[co][slot 16q + 8kh + j][tap]
//   extra fragment (PIECES = 3):   kh = 0: [w_hi(x0..x3) | w_hi(x0..x3)]  (pairs with the A bytes [x_mid | x_lo])
//                                  kh = 1: [w_mid(x0..x3) | 0]
// wk [cout][32 slots][49]: float32 weights with the whitening scale folded in (slot 30 = indicator weight; 31 = 0);
// xslot[4]: K-slot of float-modality channel x0..x3 or -1.  pieces = 1: slot 31 receives the second piece of the
// indicator weight (the A operand carries the indicator in slots 30 AND 31).
void pack_stem_mx_weight(const float *wk, int cout, int pieces, const int *xslot, unsigned short *out) {
  const int ntl = cout / 32, nft = pieces * 2 + (pieces == 3 ? 1 : 0);
  auto piece = [&](float v, int pc) {
    unsigned short hs = host_bf16(v);
    if (pc == 0) return hs;
    const float r1 = v - host_bf16_to_float(hs);
    const unsigned short ms = host_bf16(r1);
    if (pc == 1) return ms;
    return host_bf16(r1 - host_bf16_to_float(ms));
  };
  for (int tap = 0; tap < 49; ++tap)
    for (int f = 0; f < nft; ++f)
      for (int nt = 0; nt < ntl; ++nt)
        for (int ln = 0; ln < 64; ++ln)
          for (int j = 0; j < 8; ++j) {
            const int kh = ln >> 5, co = nt * 32 + (ln & 31);
            unsigned short val = 0;
            if (f < pieces * 2) {
              const int pc = f / 2, q = f % 2, slot = 16 * q + 8 * kh + j;
              if (pieces == 1 && slot == 31)
                val = piece(wk[((size_t)co * 32 + 30) * 49 + tap], 1);
              else
                val = piece(wk[((size_t)co * 32 + slot) * 49 + tap], pc);
            } else {
              const int x = j & 3;
              if (xslot[x] >= 0) {
                const float w = wk[((size_t)co * 32 + xslot[x]) * 49 + tap];
                if (kh == 0) val = piece(w, 0);
                else if (j < 4) val = piece(w, 1);
              }
            }
            out[((((size_t)tap * nft + f) * ntl + nt) * 64 + ln) * 8 + j] = val;
          }
}

// Training: rebuild the three-piece B operand ON THE DEVICE from the current OIHW stem weight (inside the flat parameter
// buffer) and the current whitening tables sc/sh of the stem's "new" channel order (sc = 1/(div*std), sh = -mean/std) —
// both change every optimisation step.  Same arithmetic as pnvo_load_weights + pack_stem_mx_weight (double products,
// rounded to float once, then split exactly).  One thread per bf16 element of the packed array (32 output channels).
__global__ __launch_bounds__(256) void stem_mx_repack_kernel(const float *w, int cin, const float *sc_new, const float *sh_new,
                                                           const int *slot_ref, const int *slot_new, const int *xslot,
                                                           unsigned short *out, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = e & 7, ln = (e >> 3) & 63;
  int r = e >> 9;
  const int f = r % 7, tap = r / 7;                  // (one N-tile: cout = 32)
  const int kh = ln >> 5, co = ln & 31;
  auto wk = [&](int slot) -> float {
    if (slot == 30) {                                // "inside the image" indicator: sum_c W * (-mean_c / std_c)
      double ind = 0.0;
      for (int k = 0; k < 30; ++k)
        if (slot_ref[k] >= 0) ind += (double)w[((long)co * cin + slot_ref[k]) * 49 + tap] * (double)sh_new[slot_new[k]];
      return (float)ind;
    }
    if (slot > 30 || slot_ref[slot] < 0) return 0.f;
    return (float)((double)w[((long)co * cin + slot_ref[slot]) * 49 + tap] * (double)sc_new[slot_new[slot]]);
  };
  auto piece = [&](float v, int pc) -> unsigned short {
    const unsigned hs = pack_bf16(v, 0.f) & 0xffffu;
    if (pc == 0) return (unsigned short)hs;
    const float r1 = v - bf16_lo(hs);
    const unsigned ms = pack_bf16(r1, 0.f) & 0xffffu;
    if (pc == 1) return (unsigned short)ms;
    return (unsigned short)(pack_bf16(r1 - bf16_lo(ms), 0.f) & 0xffffu);
  };
  unsigned short val = 0;
  if (f < 6) {
    const int pc = f / 2, q = f % 2;
    val = piece(wk(16 * q + 8 * kh + j), pc);
  } else {
    const int x = j & 3;
    if (xslot[x] >= 0) {
      const float v = wk(xslot[x]);
      if (kh == 0) val = piece(v, 0);
      else if (j < 4) val = piece(v, 1);
    }
  }
  out[e] = val;
}

// The two-piece float16 operand of the same weights (pack_stem_mx_weight_h on the device), for the training forward.
// Folded weight of (co, slot, tap) exactly as above (double products rounded to float once); rgb slots times 2^8.
__device__ __forceinline__ float stem_folded_weight(const float *w, int cin, const float *sc_new, const float *sh_new, const int *slot_ref,
                                                    const int *slot_new, int co, int slot, int tap) {
  float v;
  if (slot == 30) {
    double ind = 0.0;
    for (int k = 0; k < 30; ++k)
      if (slot_ref[k] >= 0) ind += (double)w[((long)co * cin + slot_ref[k]) * 49 + tap] * (double)sh_new[slot_new[k]];
    v = (float)ind;
  } else if (slot > 30 || slot_ref[slot] < 0) {
    v = 0.f;
  } else {
    v = (float)((double)w[((long)co * cin + slot_ref[slot]) * 49 + tap] * (double)sc_new[slot_new[slot]]);
  }
  return (slot >= 20 && slot <= 25) ? v * 256.f : v;
}

// max |folded weight| -> scale2[0] = 2^(12 - e), scale2[1] = 1 / scale.  One folded weight per thread, the maximum as an integer
// maximum of float bits in scale2[2]; a one-thread kernel derives the pair and clears the maximum.  (One 1024-thread block walked
// the 48 608 folded weights alone — the indicator slot sums 30 channels — in 55 us, twice per training step.)
__global__ __launch_bounds__(256) void stem_mx_absmax_kernel(const float *w, int cin, const float *sc_new, const float *sh_new,
                                                            const int *slot_ref, const int *slot_new, float *scale2) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  float mx = 0.f;
  if (e < 32 * 31 * 49) {
    const int tap = e % 49, slot = (e / 49) % 31, co = e / (49 * 31);
    mx = fabsf(stem_folded_weight(w, cin, sc_new, sh_new, slot_ref, slot_new, co, slot, tap));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(reinterpret_cast<unsigned *>(scale2) + 2, __builtin_bit_cast(unsigned, mx));
}
__global__ void stem_mx_scale_finish_kernel(float *scale2) {
  unsigned *bits = reinterpret_cast<unsigned *>(scale2) + 2;
  const unsigned mb = *bits;
  *bits = 0u;
  int e = 0;
  if (mb != 0u) e = (int)((mb >> 23) & 0xffu) - 126;
  scale2[0] = __builtin_bit_cast(float, (unsigned)(12 - e + 127) << 23);
  scale2[1] = __builtin_bit_cast(float, (unsigned)(e - 12 + 127) << 23);
}

__global__ __launch_bounds__(256) void stem_mx_repack_h_kernel(const float *w, int cin, const float *sc_new, const float *sh_new,
                                                             const int *slot_ref, const int *slot_new, const int *xslot,
                                                             const float *scale2, unsigned short *out, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = e & 7, ln = (e >> 3) & 63;
  const int r = e >> 9;
  const int f = r % 5, tap = r / 5;                  // (one N-tile: cout = 32)
  const int kh = ln >> 5, co = ln & 31;
  const float sc = scale2[0];
  auto piece = [&](float v, int pc) -> unsigned short {
    const _Float16 a = (_Float16)v;
    if (pc == 0) return __builtin_bit_cast(unsigned short, a);
    return __builtin_bit_cast(unsigned short, (_Float16)(v - (float)a));
  };
  unsigned short val = 0;
  if (f < 4) {
    const int pc = f / 2, q = f % 2, slot = 16 * q + 8 * kh + j;
    if (slot < 31) val = piece(stem_folded_weight(w, cin, sc_new, sh_new, slot_ref, slot_new, co, slot, tap) * sc, pc);
  } else if (kh == 0 && j < 4 && xslot[j] >= 0) {
    val = piece(stem_folded_weight(w, cin, sc_new, sh_new, slot_ref, slot_new, co, xslot[j], tap) * sc, 0);
  }
  out[e] = val;
}

hipError_t launch_stem_mx_repack_h(const float *w_oihw, int cin, const float *sc_new, const float *sh_new, const int *slot_ref,
                                   const int *slot_new, const int *xslot, float *scale2, unsigned short *wpk2, hipStream_t s) {
  hipLaunchKernelGGL(stem_mx_absmax_kernel, dim3((32 * 31 * 49 + 255) / 256), dim3(256), 0, s, w_oihw, cin, sc_new, sh_new, slot_ref, slot_new,
                     scale2);
  hipLaunchKernelGGL(stem_mx_scale_finish_kernel, dim3(1), dim3(1), 0, s, scale2);
  const int total = 49 * 5 * 64 * 8;
  hipLaunchKernelGGL(stem_mx_repack_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oihw, cin, sc_new, sh_new,
                     slot_ref, slot_new, xslot, scale2, wpk2, total);
  return hipGetLastError();
}

hipError_t launch_stem_mx_repack(const float *w_oihw, int cin, const float *sc_new, const float *sh_new, const int *slot_ref,
                                 const int *slot_new, const int *xslot, unsigned short *wpk3, hipStream_t s) {
  const int total = 49 * 7 * 64 * 8;
  hipLaunchKernelGGL(stem_mx_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oihw, cin, sc_new, sh_new,
                     slot_ref, slot_new, xslot, wpk3, total);
  return hipGetLastError();
}

hipError_t launch_stem_mx(const StemMXArgs &a, int pieces, int ntiles_n, bool bf16_out, hipStream_t s) {
  StemMXArgs p = a;
  p.tiles_x = (a.Wo + TW - 1) / TW;
  p.tiles_y = (a.Ho + TH - 1) / TH;
  const long ntiles = (long)a.B * p.tiles_x * p.tiles_y;
  const unsigned gx = (unsigned)(((ntiles + 7) / 8) * 8);
  if (a.raw_depth != nullptr) {                          // sensor frames in (pnvo_forward_raw): the RAW stager (+ 64 B: bin edges)
    if (pieces == 2 && !bf16_out) {
      const size_t ldsb = RED_OFF + 4 * 1 * 32 * 2 * sizeof(float) + 64;
      if (a.pool != nullptr)
        hipLaunchKernelGGL((stem_mx_kernel<2, 1, false, true, true>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
      else
        hipLaunchKernelGGL((stem_mx_kernel<2, 1, false, false, true>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
    } else if (pieces == 1 && bf16_out && ntiles_n % 2 == 0) {
      const size_t ldsb = RED_OFF + 4 * 2 * 32 * 2 * sizeof(float) + 64;
      hipLaunchKernelGGL((stem_mx_kernel<1, 2, true, false, true>), dim3(gx, (unsigned)(ntiles_n / 2)), dim3(NTHREADS), ldsb, s, p);
    } else if (pieces == 1 && bf16_out) {
      const size_t ldsb = RED_OFF + 4 * 1 * 32 * 2 * sizeof(float) + 64;
      hipLaunchKernelGGL((stem_mx_kernel<1, 1, true, false, true>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
    } else {
      return hipErrorInvalidValue;
    }
  } else if (pieces == 2 && !bf16_out) {                 // float16 pieces (inference default), pooled keys or raw output
    const size_t ldsb = RED_OFF + 4 * 1 * 32 * 2 * sizeof(float);
    if (a.pool != nullptr)
      hipLaunchKernelGGL((stem_mx_kernel<2, 1, false, true>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
    else
      hipLaunchKernelGGL((stem_mx_kernel<2, 1, false>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
  } else if (pieces == 3 && !bf16_out && a.pool != nullptr) {   // pooled keys instead of the raw output (inference)
    const size_t ldsb = RED_OFF + 4 * 1 * 32 * 2 * sizeof(float);
    hipLaunchKernelGGL((stem_mx_kernel<3, 1, false, true>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
  } else if (pieces == 3 && !bf16_out) {
    const size_t ldsb = RED_OFF + 4 * 1 * 32 * 2 * sizeof(float);
    hipLaunchKernelGGL((stem_mx_kernel<3, 1, false>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
  } else if (pieces == 3 && bf16_out) {                  // float32-exact stem feeding the bf16 stages (accuracy experiments)
    const size_t ldsb = RED_OFF + 4 * 1 * 32 * 2 * sizeof(float);
    hipLaunchKernelGGL((stem_mx_kernel<3, 1, true>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
  } else if (pieces == 1 && bf16_out && ntiles_n % 2 == 0) {
    const size_t ldsb = RED_OFF + 4 * 2 * 32 * 2 * sizeof(float);
    hipLaunchKernelGGL((stem_mx_kernel<1, 2, true>), dim3(gx, (unsigned)(ntiles_n / 2)), dim3(NTHREADS), ldsb, s, p);
  } else if (pieces == 1 && bf16_out) {
    const size_t ldsb = RED_OFF + 4 * 1 * 32 * 2 * sizeof(float);
    hipLaunchKernelGGL((stem_mx_kernel<1, 1, true>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace pnvo
