// wgrad_stem_mx.hip — weight gradient of the 7x7 stride-2 stem on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16), gfx950.
//
//   dW[co][c][kh][kw] = sum_{n,oy,ox} dY[n,oy,ox,co] * xw_c[n, 2 oy + kh - 3, 2 ox + kw - 3]
// with xw the assembled + whitened input of the reference (vo_cnn.py:110-176, running_mean_and_var.py:62-63; zero outside
// the image).  It is the training step's largest kernel: 198 GFLOP at 128 pairs, 2.1 ms on the fp32
// matrix cores.  Same idea as the forward stem (stem_mx.hip): whitening is affine, xw_c = sc_c raw_c + sh_c inside the
// image, so
//   dW[co][c][tap] = sc_c * G[tap][c][co] + sh_c * G[tap][ind][co],   G[tap][r][co] = sum_pix dY[pix][co] * A_r[pix + tap]
// where the rows A_r are RAW observation values that are exact in bf16 (one-hot depth, uint8-valued rgb minus 128, the 0/1
// "inside the image" indicator) or split exactly into three bf16 pieces (depth, top-down view: hi + mid + lo); dY is split
// into three bf16 pieces on the other side.  Every product is exact in float32; the sums run in the MFMA's float32
// accumulators and, across workgroups, in fp64 in a fixed order (bit-reproducible).
//
// The contraction index of a weight gradient is the PIXEL, and a bf16 MFMA lane holds 8 consecutive k: both operands must be
// channel-major in LDS.  Layout: the input patch is de-interleaved by column parity (stride-2 conv), Xt[row][parity][slot]
// [jj] with jj = (ox - ox0) + (kw >> 1) the contraction index, so the A fragment of ANY tap is an aligned 16-byte read; the
// tap's column shift lands on the dY side, dYt[row][piece][co][u], read as five dwords and funnel-shifted when odd.
// Workgroup = 8 waves, one persistent workgroup per CU.  Waves 0-6 multiply (wave = kw: seven kernel rows x 3 M-tiles x 2
// N-tiles of 16x16 accumulators = 168 registers; one dY fragment set feeds 126 MFMAs).  Wave 7 meanwhile fetches the NEXT
// tile's raw float32 data (observation tensors + dY, 75 KB) into a raw LDS buffer by LDS-DMA (global_load_lds_dwordx4: no
// registers, many instructions in flight); after the barrier all eight waves convert / split / transpose it into the operand
// buffer and the next multiply starts.  Measured per tile (PNVO_WSM_PROF=1): DMA 14.5 k cycles (one 1-KiB instruction per
// ~165 cycles = 6 B/cycle/CU), multiply 9.5-11.5 k, convert 5 k: 1.12 ms for 128 pairs against 2.07 ms on the fp32 pipe.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TH = 6, TW = 13;                     // output tile (rows x columns); a K-chunk = 2 rows x 16 contraction indices
constexpr int KCH = TH / 2;
constexpr int PR = 2 * TH + 5;                     // patch rows (17); patch columns 2*TW+5 = 31 -> 16 / 15 per parity
constexpr int NROW = 48;                           // M rows: 0-19 one-hot depth | 20-25 rgb-128 | 26-29 float hi | 30 indicator |
                                                   //         31 zero | 32-35 float mid | 36-39 float lo | 40-47 zero
constexpr int XS_SLOT = 32, XS_PAR = NROW * XS_SLOT + 32, XS_ROW = 2 * XS_PAR;   // bytes: 16 jj x bf16 per slot; the pads put the
                                                   // converter's 16 lanes (jj half, parity, 4 rows) on 16 different 16-byte bank groups
constexpr int X_BYTES = PR * XS_ROW;               // 53312
constexpr int DU = 24;                             // dYt elements per (row, piece, co): u = (ox - ox0) + 3, zero padded
constexpr int DS_CO = DU * 2, DS_PIECE = 32 * DS_CO, DS_ROW = 3 * DS_PIECE + 16;   // (+16: rows on different banks)
constexpr int D_BYTES = TH * DS_ROW;               // 27744
constexpr int BUF_BYTES = X_BYTES + D_BYTES;       // 79968: the bf16 operands of the current tile
// raw float32 tile as DMA'd: per patch row the 31-pixel segment of each tensor as FLAT bytes in 16-byte pieces,
// [row][dd 155 | rgb 47 | depth 16 | tdv 16 | pad 22] = four wave instructions (one M0), then dY [tile row][pixel][8 | pad] = two
// per tile row.  A wave instruction moves 64 consecutive pieces = 1 KiB of LDS from (mostly) contiguous memory.  Measured
// on the way here: 4-byte pieces quarter the rate; 64 lanes gathering 16 B from 64 different pixels retire one instruction
// per ~340 cycles; every write of M0 waits for the wave's outstanding DMA; and the issuing wave's address arithmetic
// competes with the MFMAs of the multiplier on its SIMD — hence row-shaped groups whose lane constants are computed once.
// dd pieces are whole-pixel (80 B = 5 pieces): outside the image they read a page of zeros.  rgb / depth / tdv pieces (24 /
// 8 / 8 B per pixel) straddle pixels: the converter masks by pixel.
constexpr int PCOLS = 2 * TW + 5;                  // 31
constexpr int FROW = 155 + 47 + 16 + 16;           // used pieces per patch row (234 of 256)
constexpr int F_RGB = 155 * 16, F_DEP = F_RGB + 47 * 16, F_TDV = F_DEP + 16 * 16;   // byte offsets inside a row
constexpr int RAW_ROW = 4096 + 16;                 // (+16: the rows on different banks for the converter)
constexpr int RAW_OFF = BUF_BYTES;                 // raw buffer behind the operands
constexpr int DY_OFF = RAW_OFF + PR * RAW_ROW, DY_ROW = 2048;
constexpr int LDS_BYTES = DY_OFF + TH * DY_ROW;    // 163 248
constexpr int NPOS = PR * 4;                       // (patch row, parity, jj group of 8) positions of the patch
constexpr int NTHREADS = 512;

__device__ __forceinline__ unsigned pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));   // v_cvt_pk_bf16_f32, RNE
}
__device__ __forceinline__ float lo16(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi16(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
}  // namespace

__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_stem_mx_kernel(const WgradStemMXArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = p.B * p.tiles_y * p.tiles_x;
  const int t0 = blockIdx.x * p.tiles_per_wg, t1 = min(ntiles, t0 + p.tiles_per_wg);
  const long npix = (long)p.H * p.W;

  // rows 31 and 40-47 and the dYt pads are never written by the converter: zero the operand buffer once
  for (int e = threadIdx.x; e < BUF_BYTES / 16; e += NTHREADS) reinterpret_cast<u32x4 *>(lds)[e] = u32x4{0u, 0u, 0u, 0u};

  struct Tile {
    int n, oy0, ox0, hi0, wi0;
  };
  auto tile_of = [&](int t) {
    Tile r;
    int q = t;
    const int tx = q % p.tiles_x;
    q /= p.tiles_x;
    const int ty = q % p.tiles_y;
    r.n = q / p.tiles_y;
    r.oy0 = ty * TH;
    r.ox0 = tx * TW;
    r.hi0 = 2 * r.oy0 - 3;
    r.wi0 = 2 * r.ox0 - 3;
    return r;
  };

  // ---------------------------------------------------------------------------------- raw tile -> LDS by DMA (wave 7)
  // Issued by the eighth wave alone: a wave that issues a tile's worth of loads stalls on the memory queue for about as long
  // as the tile takes to arrive, which must not happen to a multiplying wave.  Per tile the lane's nine patch pixels get one
  // validity test and one pixel offset; the 15 regions reuse them (address = tensor + pixel * stride + piece).
  // Four instructions share one M0: the instruction offset moves BOTH the LDS destination and the memory address, so the
  // lane's address is pre-decremented.  (Writing M0 waits for this wave's outstanding LDS-DMA: ~750 cycles each otherwise.)
  auto set_m0 = [&](unsigned lc) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lc) : "memory"); };
  auto dma16_0 = [&](const char *ga) { asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(ga) : "memory"); };
  auto dma16_1 = [&](const char *ga) { asm volatile("global_load_lds_dwordx4 %0, off offset:1024" ::"v"(ga - 1024) : "memory"); };
  auto dma16_2 = [&](const char *ga) { asm volatile("global_load_lds_dwordx4 %0, off offset:2048" ::"v"(ga - 2048) : "memory"); };
  auto dma16_3 = [&](const char *ga) { asm volatile("global_load_lds_dwordx4 %0, off offset:3072" ::"v"(ga - 3072) : "memory"); };
  // wave 7 issues the whole tile (73 instructions): a wave that issues DMA stalls on every instruction for about as long as
  // the CU takes to retire one, which must not happen to a multiplying wave
  // lane constants of the four instructions of a patch row (piece r = 64 i + lane): byte offset from the row's first patch
  // pixel in its tensor | tensor << 16 | patch column << 20 (dd only) | valid << 28; and of the two of a dY row
  int fc[4], dc[2];
  auto lane_consts = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 64 * i + lane;
      int c, ten, pc = 0;
      if (r < 155) {
        pc = r / 5;
        c = pc * 80 + 16 * (r - 5 * pc);
        ten = 2;
      } else if (r < 202) {
        c = 16 * (r - 155);
        ten = 0;
      } else if (r < 218) {
        c = 16 * (r - 202);
        ten = 1;
      } else {
        c = 16 * (r - 218);
        ten = 3;
      }
      fc[i] = c | (ten << 16) | (pc << 20) | ((r < FROW ? 1 : 0) << 28);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 64 * i + lane;                       // piece (pixel r >> 3, 16-byte eighth r & 7) of a tile row
      dc[i] = ((r >> 3) * 128 + 16 * (r & 7)) | ((r >> 3) << 16) | ((r < TW * 8 ? 1 : 0) << 28);
    }
  };
  auto issue_dma7 = [&](int t) {
    const Tile T = tile_of(t);
    const char *zp = reinterpret_cast<const char *>(p.zero_page);
    const long tot_rgb = (long)p.B * npix * 24, tot_f = (long)p.B * npix * 8;
#pragma unroll 1
    for (int row = 0; row < PR; ++row) {
      const int hi = T.hi0 + row;
      const bool rowok = hi >= 0 && hi < p.H;
      const long pix0 = (long)T.n * npix + (long)hi * p.W + T.wi0;          // first patch pixel of the row (may be outside)
      const char *g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = fc[i] & 0xffff, ten = (fc[i] >> 16) & 3, pc = (fc[i] >> 20) & 63;
        const bool valid = rowok && (fc[i] >> 28) != 0;
        const long pstride = ten == 2 ? 80 : (ten == 0 ? 24 : 8);
        const long off = pix0 * pstride + c;
        const float *base = p.src[ten];
        bool ok;
        if (ten == 2) {
          const int wi = T.wi0 + pc;
          ok = wi >= 0 && wi < p.W;
        } else {
          ok = off >= 0 && off + 16 <= (ten == 0 ? tot_rgb : tot_f);
        }
        g[i] = valid && ok && base != nullptr ? reinterpret_cast<const char *>(base) + off : zp;
      }
      set_m0((unsigned)(RAW_OFF + row * RAW_ROW));
      dma16_0(g[0]);
      dma16_1(g[1]);
      dma16_2(g[2]);
      dma16_3(g[3]);
    }
#pragma unroll 1
    for (int row = 0; row < TH; row += 2) {              // two tile rows = four instructions per M0
      const char *g[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = row + (i >> 1), oy = T.oy0 + rr;
        const int d = dc[i & 1], ox = T.ox0 + ((d >> 16) & 63);
        const bool ok = (d >> 28) != 0 && oy < p.Ho && ox < p.Wo;
        g[i] = ok ? reinterpret_cast<const char *>(p.dy + (((long)T.n * p.Ho + oy) * p.Wo + T.ox0) * 32) + (d & 0xffff) : zp;
      }
      set_m0((unsigned)(DY_OFF + row * DY_ROW));
      dma16_0(g[0]);
      dma16_1(g[1]);
      dma16_2(g[2]);
      dma16_3(g[3]);
    }
  };

  // ---------------------------------------------------------------------------------- raw -> bf16 operands (all waves)
  // unit = (kind, patch position): kind 0-9 the channel pairs of the one-hot depth, 10-14 rgb 0-1 / 2-3 / 4-5, depth, tdv,
  // 15 the indicator.  Then the dY units (tile row, output channel).
  auto convert = [&](int t) {
    const Tile T = tile_of(t);
    for (int u = threadIdx.x; u < 16 * NPOS + TH * 32; u += NTHREADS) {
      if (u < 16 * NPOS) {
        const int kind = u / NPOS, pos = u - kind * NPOS;  // kind-major: a wave runs one code path; lanes = consecutive positions
        const int pr = pos >> 2, par = (pos >> 1) & 1, g = pos & 1;
        const int hi = T.hi0 + pr;
        const bool rowok = hi >= 0 && hi < p.H;
        unsigned char *dst = lds + pr * XS_ROW + par * XS_PAR;
        if (kind < 10) {
          // the one-hot depth: channels 2 kind, 2 kind + 1 (slots 0-19)
          f32x2 v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int pc = 2 * (8 * g + e) + par;
            v[e] = pc < PCOLS ? *reinterpret_cast<const f32x2 *>(lds + RAW_OFF + pr * RAW_ROW + pc * 80 + 8 * kind) : f32x2{0.f, 0.f};
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int slot = 2 * kind + j;
            const u32x4 w = {pk(v[0][j], v[1][j]), pk(v[2][j], v[3][j]), pk(v[4][j], v[5][j]), pk(v[6][j], v[7][j])};
            *reinterpret_cast<u32x4 *>(dst + slot * XS_SLOT + ((g ^ ((slot >> 3) & 1)) << 4)) = w;
          }
        } else if (kind < 15) {
          // rgb 0-1, 2-3, 4-5 (slots 20-25, minus 128), depth (26-27), top-down view (28-29); the float modalities also
          // write their second and third bf16 pieces (rows 32-35, 36-39)
          const int type = kind - 10;
          const bool present = (type < 3 ? p.src[0] : (type == 3 ? p.src[1] : p.src[3])) != nullptr;
          const float *base = type < 3 ? p.src[0] : (type == 3 ? p.src[1] : p.src[3]);
          const int pstride = type < 3 ? 24 : 8;               // bytes per pixel
          const int seg = type < 3 ? F_RGB : (type == 3 ? F_DEP : F_TDV);   // the tensor's pieces inside the row's flat pieces
          const int chb = type < 3 ? 8 * type : 0;             // byte of the channel pair inside the pixel
          const float cen = type < 3 ? 128.f : 0.f;
          // pieces that would cross the first / last byte of the tensor were not fetched (zeros): the pixels they hold are
          // read from global memory here (only the first patch rows of sample 0 / the last of the last sample)
          const long rowoff = ((long)T.n * npix + (long)hi * p.W + T.wi0) * pstride, total = (long)p.B * npix * pstride;
          const bool edge = rowoff < 0 || rowoff + 47 * 16 > total;
          f32x2 v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int pc = 2 * (8 * g + e) + par, wi = T.wi0 + pc;
            const bool in = present && rowok && pc < PCOLS && wi >= 0 && wi < p.W;   // (rgb is centred: outside stays 0, not -128)
            const int bo = min(pc, PCOLS - 1) * pstride + chb;                        // byte inside the tensor's row segment
            f32x2 x = *reinterpret_cast<const f32x2 *>(lds + RAW_OFF + pr * RAW_ROW + seg + bo);
            if (edge && in) {
              const long ps = rowoff + (bo & ~15);
              if (ps < 0 || ps + 16 > total) x = *reinterpret_cast<const f32x2 *>(reinterpret_cast<const char *>(base) + rowoff + bo);
            }
            v[e] = in ? f32x2{x[0] - cen, x[1] - cen} : f32x2{0.f, 0.f};
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float a = v[2 * k][j], b = v[2 * k + 1][j];
              h[k] = pk(a, b);
              const float ra = a - lo16(h[k]), rb = b - hi16(h[k]);
              m[k] = pk(ra, rb);
              l[k] = pk(ra - lo16(m[k]), rb - hi16(m[k]));
            }
            const int slot = 20 + 2 * type + j;
            *reinterpret_cast<u32x4 *>(dst + slot * XS_SLOT + ((g ^ ((slot >> 3) & 1)) << 4)) = u32x4{h[0], h[1], h[2], h[3]};
            if (type >= 3) {
              const int sm = 32 + 2 * (type - 3) + j, sl = 36 + 2 * (type - 3) + j;
              *reinterpret_cast<u32x4 *>(dst + sm * XS_SLOT + ((g ^ ((sm >> 3) & 1)) << 4)) = u32x4{m[0], m[1], m[2], m[3]};
              *reinterpret_cast<u32x4 *>(dst + sl * XS_SLOT + ((g ^ ((sl >> 3) & 1)) << 4)) = u32x4{l[0], l[1], l[2], l[3]};
            }
          }
        } else {
          // the "inside the image" indicator (row 30)
          unsigned w[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int wa = T.wi0 + 2 * (8 * g + 2 * k) + par, wb = wa + 2;
            w[k] = ((rowok && wa >= 0 && wa < p.W) ? 0x3f80u : 0u) | ((rowok && wb >= 0 && wb < p.W) ? 0x3f800000u : 0u);
          }
          *reinterpret_cast<u32x4 *>(dst + 30 * XS_SLOT + ((g ^ 1) << 4)) = u32x4{w[0], w[1], w[2], w[3]};
        }
      } else {
        // dY: unit = (tile row, output channel); thirteen pixels, three pieces, element u = (ox - ox0) + 3
        const int ud = u - 16 * NPOS;
        const int row = ud >> 5, co = ud & 31;
        float v[TW + 1];
#pragma unroll
        for (int e = 0; e < TW; ++e) v[e] = *reinterpret_cast<const float *>(lds + DY_OFF + row * DY_ROW + e * 128 + 4 * co);
        v[TW] = 0.f;
        // pairs (u even, u + 1): dword 1 + k = (v[2k-1], v[2k]), v[-1] = v[13] = 0; dword 0 and 9..11 stay zero
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float a = k == 0 ? 0.f : v[2 * k - 1], b = k == 7 ? 0.f : v[2 * k];
          h[k] = pk(a, b);
          const float ra = a - lo16(h[k]), rb = b - hi16(h[k]);
          m[k] = pk(ra, rb);
          l[k] = pk(ra - lo16(m[k]), rb - hi16(m[k]));
        }
        unsigned char *dst = lds + X_BYTES + row * DS_ROW + co * DS_CO;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          const unsigned *sp = pc == 0 ? h : (pc == 1 ? m : l);
          unsigned char *d2 = dst + pc * DS_PIECE;
          *reinterpret_cast<u32x4 *>(d2) = u32x4{0u, sp[0], sp[1], sp[2]};
          *reinterpret_cast<u32x4 *>(d2 + 16) = u32x4{sp[3], sp[4], sp[5], sp[6]};
          *reinterpret_cast<u32x4 *>(d2 + 32) = u32x4{sp[7], 0u, 0u, 0u};
        }
      }
    }
  };

  // ---------------------------------------------------------------------------------- wave 7: the DMA issuer
  // (a code path of its own: the 168 accumulator registers of the multipliers must not be live in it; both paths execute the
  //  same barriers)
  if (wave == 7) {
    lane_consts();
    if (t0 < t1) issue_dma7(t0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // raw tile t0 landed; operand buffer zeroed
    if (t0 < t1) convert(t0);
    __syncthreads();
    unsigned long long c_issue = 0, c_wait = 0, c_bar = 0, c_conv = 0;      // PNVO_WSM_PROF: phase cycles of this wave
    for (int t = t0; t < t1; ++t) {
      const unsigned long long a0 = __builtin_readcyclecounter();
      if (t + 1 < t1) issue_dma7(t + 1);                 // lands in the raw buffer while the seven waves multiply
      const unsigned long long a1 = __builtin_readcyclecounter();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long a2 = __builtin_readcyclecounter();
      __syncthreads();                                   // tile t + 1 landed; the operands of tile t are no longer read
      const unsigned long long a3 = __builtin_readcyclecounter();
      if (t + 1 < t1) convert(t + 1);
      __syncthreads();
      const unsigned long long a4 = __builtin_readcyclecounter();
      c_issue += a1 - a0;
      c_wait += a2 - a1;
      c_bar += a3 - a2;
      c_conv += a4 - a3;
    }
    if (p.prof && lane == 0 && blockIdx.x == 5) {
      p.prof[0] = c_issue;
      p.prof[1] = c_wait;
      p.prof[2] = c_bar;
      p.prof[3] = c_conv;
      p.prof[4] = t1 - t0;
    }
    return;
  }
  // ---------------------------------------------------------------------------------- waves 0-6: the multipliers
  f32x4 acc[7][3][2];
#pragma unroll
  for (int a = 0; a < 7; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int i16 = lane & 15, q = lane >> 4;
  const int kw = wave, sft = kw >> 1, par = kw & 1;
  // A: Xt[pr = 4 c + 2 (q >> 1) + kh][par][slot = 16 mt + i16][8 (q & 1) ..]   (half swizzled by bit 3 of the slot)
  const unsigned abase = (unsigned)((2 * (q >> 1)) * XS_ROW + par * XS_PAR + i16 * XS_SLOT + (((q & 1) ^ (i16 >> 3)) << 4));
  // B: dYt[row = 2 c + (q >> 1)][piece][co = 16 nt + i16][u0 ..], u0 = 8 (q & 1) + 3 - sft
  const int u0 = 8 * (q & 1) + 3 - sft;
  const unsigned bbase = (unsigned)(X_BYTES + (q >> 1) * DS_ROW + i16 * DS_CO + (u0 >> 1) * 4);
  const bool odd = (u0 & 1) != 0;                        // (wave-uniform: depends on kw only)
  __syncthreads();
  if (t0 < t1) convert(t0);
  __syncthreads();
  unsigned long long c_mm = 0, c_wait = 0, c_bar = 0, c_conv = 0;
  for (int t = t0; t < t1; ++t) {
    const unsigned long long a0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int c = 0; c < KCH; ++c) {
      u32x4 bf[3][2];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const unsigned *src = reinterpret_cast<const unsigned *>(lds + bbase + 2 * c * DS_ROW + pc * DS_PIECE + nt * 16 * DS_CO);
          const unsigned d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4];
          bf[pc][nt] = odd ? u32x4{__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
                                   __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(d4, d3, 16)}
                           : u32x4{d0, d1, d2, d3};
        }
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
          const u32x4 a = *reinterpret_cast<const u32x4 *>(lds + abase + (4 * c + kh) * XS_ROW + mt * 16 * XS_SLOT);
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[kh][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bf[pc][nt]),
                                                                        acc[kh][mt][nt], 0, 0, 0);
        }
      }
    }
    const unsigned long long a1 = __builtin_readcyclecounter();
    const unsigned long long a2 = __builtin_readcyclecounter();
    __syncthreads();                                     // ... everybody's; the operands of tile t are no longer read
    const unsigned long long a3 = __builtin_readcyclecounter();
    if (t + 1 < t1) convert(t + 1);
    __syncthreads();
    const unsigned long long a4 = __builtin_readcyclecounter();
    c_mm += a1 - a0;
    c_wait += a2 - a1;
    c_bar += a3 - a2;
    c_conv += a4 - a3;
  }
  if (p.prof && lane == 0 && blockIdx.x == 5 && (wave == 0 || wave == 3)) {
    p.prof[8 + wave] = c_mm;
    p.prof[16 + wave] = c_wait;
    p.prof[24 + wave] = c_bar;
    p.prof[32 + wave] = c_conv;
  }
  // partial[wg][tap][row 48][co 32]; C/D layout of 16x16: col = lane & 15, row = 4 (lane >> 4) + r
  {
    float *dst = p.partial + (long)blockIdx.x * 49 * NROW * 32;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dst[((long)(kh * 7 + wave) * NROW + mt * 16 + 4 * q + r) * 32 + nt * 16 + i16] = acc[kh][mt][nt][r];
  }
}

// G[tap][row][co] = sum over workgroups (fixed order, fp64) of the partials.  One block per (tap, row): 32 channels x 8 lanes.
__global__ __launch_bounds__(256) void wgrad_stem_mx_sum_kernel(const float *partial, int nwg, float *G) {
  __shared__ double red[8][32];
  const int tr = blockIdx.x;                             // tap * 48 + row
  const int co = threadIdx.x & 31, cl = threadIdx.x >> 5;
  double acc = 0.0;
  for (int w = cl; w < nwg; w += 8) acc += (double)partial[((long)w * 49 * NROW + tr) * 32 + co];
  red[cl][co] = acc;
  __syncthreads();
  if (cl == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][co];
    G[(long)tr * 32 + co] = (float)s;
  }
}

// dW[co][c][tap] = sc_c * Graw_c + sh_c * G_ind with Graw_c the sum of the channel's rows (+ 128 G_ind for the centred rgb).
// K-slot k (0-29, the forward stem's order): reference channel slot_ref[k], whitening tables at slot_new[k].
__global__ __launch_bounds__(256) void wgrad_stem_mx_fold_kernel(const float *G, const float *sc_new, const float *sh_new,
                                                                const int *slot_ref, const int *slot_new, int cin, float *grad) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 49 * 30 * 32) return;
  const int co = e & 31, k = (e >> 5) % 30, tap = (e >> 5) / 30;
  const int c = slot_ref[k];
  if (c < 0 || c >= cin) return;
  const float *g = G + (long)tap * NROW * 32 + co;
  const double gind = (double)g[30 * 32];
  double graw = (double)g[k * 32];
  if (k >= 20 && k < 26) graw += 128.0 * gind;
  if (k >= 26) graw += (double)g[(32 + k - 26) * 32] + (double)g[(36 + k - 26) * 32];
  const int kn = slot_new[k];
  grad[((long)co * cin + c) * 49 + tap] = (float)((double)sc_new[kn] * graw + (double)sh_new[kn] * gind);
}

void wgrad_stem_mx_plan(WgradStemMXArgs &a) {
  a.tiles_x = (a.Wo + TW - 1) / TW;
  a.tiles_y = (a.Ho + TH - 1) / TH;
  const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
  long nwg = ntiles < 256 ? ntiles : 256;                 // one persistent workgroup per CU
  a.tiles_per_wg = (int)((ntiles + nwg - 1) / nwg);
  a.nwg = (int)((ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg);
}

size_t wgrad_stem_mx_scratch_floats(const WgradStemMXArgs &a) { return (size_t)(a.nwg + 1) * 49 * NROW * 32; }

// `scratch`: wgrad_stem_mx_scratch_floats(a) floats (partials of every workgroup + the summed G)
hipError_t launch_wgrad_stem_mx(const WgradStemMXArgs &a0, float *scratch, const float *sc_new, const float *sh_new, const int *slot_ref,
                                const int *slot_new, int cin, float *grad, hipStream_t s) {
  WgradStemMXArgs a = a0;
  a.partial = scratch;
  static unsigned long long *prof = nullptr;             // PNVO_WSM_PROF=1: per-phase cycles of workgroup 5, printed once
  if (std::getenv("PNVO_WSM_PROF") && !prof) {
    (void)hipMalloc((void **)&prof, 512);
    (void)hipMemset(prof, 0, 512);
  }
  a.prof = prof;
  float *G = scratch + (size_t)a.nwg * 49 * NROW * 32;
  hipLaunchKernelGGL(wgrad_stem_mx_kernel, dim3((unsigned)a.nwg), dim3(NTHREADS), (size_t)LDS_BYTES, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wgrad_stem_mx_sum_kernel, dim3(49 * NROW), dim3(256), 0, s, a.partial, a.nwg, G);
  hipLaunchKernelGGL(wgrad_stem_mx_fold_kernel, dim3((49 * 30 * 32 + 255) / 256), dim3(256), 0, s, G, sc_new, sh_new, slot_ref, slot_new,
                     cin, grad);
  if (prof) {
    static int calls = 0;
    if (++calls == 8) {
      unsigned long long h[64];
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h, prof, 512, hipMemcpyDeviceToHost);
      const unsigned long long n = h[4] ? h[4] : 1;
      std::fprintf(stderr, "[pnvo] wgrad_stem_mx cycles per tile (%llu tiles): wave 7 issue %llu wait %llu barrier %llu convert %llu | wave 0 "
                           "mfma+dma %llu wait %llu barrier %llu convert %llu | wave 3 mfma+dma %llu wait %llu barrier %llu convert %llu\n",
                   n, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[8] / n, h[16] / n, h[24] / n, h[32] / n, h[11] / n, h[19] / n, h[27] / n, h[35] / n);
    }
  }
  return hipGetLastError();
}

}  // namespace pnvo
