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
#include <cstring>

#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TH = 8, TW = 16;
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5;   // 21 x 37
constexpr int NPIX = PH * PW;                     // 777
constexpr int PITCH = 80;                         // bytes per patch pixel: 64 (K-slots) + 16 (remainders)
constexpr int PAR = 19 * PITCH;                   // odd-column plane of a patch row
constexpr int ROW = 3072;                         // patch row pitch (2 x 19 x 80 = 3040, padded: 2 rows = 0 mod 256 B)
constexpr int PATCH_BYTES = PH * ROW;             // 64512
constexpr int NTHREADS = 256;
constexpr int RPR = 11;                           // staging: patch rows per round

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 r = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

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

template <int PIECES, int NT, bool BF16OUT>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_mx_kernel(const StemMXArgs p) {
  constexpr bool EXTRA = PIECES == 3;
  constexpr int NFT = PIECES * 2 + (EXTRA ? 1 : 0);        // B fragments per (tap, N-tile)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float *red = reinterpret_cast<float *>(lds + PATCH_BYTES);   // [4 waves][NT*32][2]

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
  const int ho0 = ty * TH, wo0 = tx * TW;
  const int hi_base = 2 * ho0 - 3, wi_base = 2 * wo0 - 3;

  const bool prof = p.prof != nullptr && (blockIdx.x % 61) == 0;   // sampled: the atomics below perturb the memory pipe                  // PNVO_STEM_DBG=9: per-phase cycles of wave 0 (s_memtime)
  const unsigned long long tp0 = prof ? __builtin_readcyclecounter() : 0;
  // ---------------------------------------------------------------- staging: observation tensors -> bf16 patch in LDS
  // Pass 1, thread = (unit u, pixel lane pl): patch columns pl, pl+16, pl+32 of all 21 rows, 63 x 8 B per thread in flight
  // at once.  Instruction count is what bounds this phase (it shares the SIMDs with the other workgroup's MFMAs), so:
  // no bounds branches (out-of-image addresses are redirected to a page of zeros; the indicator unit reads a page of
  // ones), a pair is packed by one v_cvt_pk_bf16_f32 (exact by contract: any set low bit raises the flag), and the LDS
  // address is one register per column group + an immediate per row.
  {
    const int u = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const StemMXUnit ud = p.units[u];
    const float *tb = ud.tensor == 0 ? p.src[0] : ud.tensor == 1 ? p.src[1] : ud.tensor == 2 ? p.src[2] : p.src[3];
    const float *zp = p.zero_page;
    const float *img = ud.kind == 2 ? zp + 32 : ud.kind == 3 ? zp : tb + (long)n * p.H * p.W * ud.nch + ud.choff;
    const int rowstep = ud.kind < 2 ? p.W * ud.nch : 0;                    // floats per image row of that tensor
    const int colstep = ud.kind < 2 ? ud.nch : 0;
    int coff[3];
    unsigned loff[3];
    bool cok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int px = pl + 16 * j, wi = wi_base + px;
      cok[j] = px < PW && wi >= 0 && wi < p.W;
      coff[j] = wi * colstep;
      loff[j] = (unsigned)((px & 1) * PAR + (px >> 1) * PITCH + 4 * u);
    }
    unsigned lowbits = 0;
    int roff = hi_base * rowstep;
#pragma unroll
    for (int r0 = 0; r0 < PH; r0 += RPR) {                                 // two rounds (11 + 10 rows): <= 33 loads in flight
      f32x2 v[RPR][3];
#pragma unroll
      for (int k = 0; k < RPR; ++k) {
        if (r0 + k >= PH) continue;
        const int hi = hi_base + r0 + k;                                   // wave-uniform
        const bool rok = hi >= 0 && hi < p.H;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float *ad = (rok && cok[j] && !(p.dbg & 1)) ? img + (roff + coff[j]) : zp;
          v[k][j] = *reinterpret_cast<const f32x2 *>(ad);
        }
        roff += rowstep;
      }
#pragma unroll
      for (int k = 0; k < RPR; ++k) {
        if (r0 + k >= PH) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (j == 2 && pl + 32 >= PW) continue;
          const float f0 = v[k][j][0], f1 = v[k][j][1];      // (bit_cast straight from a vector element misreads it)
          lowbits |= __builtin_bit_cast(unsigned, f0) | __builtin_bit_cast(unsigned, f1);
          const unsigned w0 = pack_bf16(v[k][j][0], v[k][j][1]);   // one v_cvt_pk_bf16_f32 (exact for contract inputs)
          *reinterpret_cast<unsigned *>(lds + (r0 + k) * ROW + loff[j]) = w0;
        }
      }
    }
    // kind 0 (rgb, one-hot depth, by contract exact in bf16): a dropped low bit would be a silently rounded input.
    // kind 1 (float modalities): the rounded-off bits are carried by the remainder pieces of pass 2 (PIECES = 3) or are
    //         the bf16 rounding of the native bf16 mode.
    if (EXTRA && ud.kind == 0 && (lowbits & 0xffffu) != 0 && p.bad_input != nullptr) *p.bad_input = 1;
  }
  if (EXTRA) {
    // Pass 2: remainders of the float modalities, x - hi = mid + lo (exact: <= 16 significant bits are left).
    // thread = (s = which float unit, column, row parity); 11 iterations of 2 patch rows.
    const int sx = threadIdx.x & 1, col = (threadIdx.x >> 1) & 63, rsub = threadIdx.x >> 7;
    const StemMXUnit ud = p.units[p.xunit[sx] >= 0 ? p.xunit[sx] : 15];
    const bool have = p.xunit[sx] >= 0;
    const float *tb = ud.tensor == 1 ? p.src[1] : p.src[3];
    const float *zp = p.zero_page;
    const float *img = have ? tb + (long)n * p.H * p.W * ud.nch + ud.choff : zp;
    const int wi = wi_base + col;
    const bool cok = col < PW && wi >= 0 && wi < p.W && have;
    const unsigned lo = (unsigned)((col & 1) * PAR + (col >> 1) * PITCH + 64 + 4 * sx);
    f32x2 v[11];
#pragma unroll
    for (int it = 0; it < 11; ++it) {
      const int k = 2 * it + rsub, hi = hi_base + k;
      const bool ok = cok && k < PH && hi >= 0 && hi < p.H;
      const float *ad = ok ? img + ((long)hi * p.W + wi) * ud.nch : zp;
      v[it] = *reinterpret_cast<const f32x2 *>(ad);
    }
#pragma unroll
    for (int it = 0; it < 11; ++it) {
      const int k = 2 * it + rsub;
      const unsigned w0 = pack_bf16(v[it][0], v[it][1]);           // the piece pass 1 stored
      const float q0 = v[it][0] - bf16_lo(w0), q1 = v[it][1] - bf16_hi(w0);
      const unsigned w1 = pack_bf16(q0, q1);
      const unsigned w2 = pack_bf16(q0 - bf16_lo(w1), q1 - bf16_hi(w1));
      if (col < PW && k < PH) {
        *reinterpret_cast<unsigned *>(lds + k * ROW + lo) = w1;
        *reinterpret_cast<unsigned *>(lds + k * ROW + lo + 8) = w2;
      }
    }
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
    const u32x4 *wl = reinterpret_cast<const u32x4 *>(p.wpk) + lane;
    const int nfrag = NFT * ntg;                          // fragments per tap in the packed array
    auto loadB = [&](int tap, u32x4 *b) {
      const u32x4 *wt = wl + (long)tap * nfrag * 64;
#pragma unroll
      for (int f = 0; f < NFT; ++f)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[f * NT + nt] = wt[(f * ntg + gy * NT + nt) * 64];
    };
    auto tapoff = [&](int tap) {
      const int kh = tap / 7, kw = tap - 7 * kh;
      return (unsigned)(kh * ROW + (kw & 1) * PAR + (kw >> 1) * PITCH);
    };
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
    // Software pipeline: the next tap's B fragments (L2) and first A chunk (LDS) are fetched during this tap's MFMAs,
    // this tap's second A chunk / remainders at its start (first needed 12 / 24 MFMAs later); two register sets used
    // alternately (no copies); sched_barriers keep the fetches ahead of the MFMAs (the scheduler would sink them).
    u32x4 b0[NFT * NT], b1[NFT * NT], a00[4], a01[4], a1[4], ax[4];
    auto mfmas = [&](const u32x4 *aq0, const u32x4 *b) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int m = 0; m < 4; ++m)
              acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, q == 0 ? aq0[m] : a1[m]),
                                                                   __builtin_bit_cast(bf16x8, b[(pc * 2 + q) * NT + nt]),
                                                                   acc[m][nt], 0, 0, 0);
      if (EXTRA) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ax[m]),
                                                                 __builtin_bit_cast(bf16x8, b[(PIECES * 2) * NT + nt]),
                                                                 acc[m][nt], 0, 0, 0);
      }
    };
    loadB(wave, b0);
    loadA0(wave, a00);
    for (int tap = wave; tap < 49; tap += 8) {
      const int t1 = tap + 4 < 49 ? tap + 4 : tap;
      loadA1(tap, a1, ax);
      loadB(t1, b1);
      loadA0(t1, a01);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(a00, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 4 >= 49) break;
      const int t2 = tap + 8 < 49 ? tap + 8 : tap;
      loadA1(tap + 4, a1, ax);
      loadB(t2, b0);
      loadA0(t2, a00);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(a01, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  const unsigned long long tp3 = prof ? __builtin_readcyclecounter() : 0;
  // ---------------------------------------------------------------- K-split reduction through LDS (fixed order) + epilogue
  const int rr16 = lane >> 5;                             // accumulator row = (r & 3) + 8 (r >> 2) + 4 rr16
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    __syncthreads();                                      // patch (or the previous N-tile's exchange) no longer read
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m == wave) continue;
      const int sp = wave - (wave > m ? 1 : 0);
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<f32x4 *>(lds + (((m * 3 + sp) * 4 + rq) * 64 + lane) * 16) =
            f32x4{acc[m][nt][4 * rq], acc[m][nt][4 * rq + 1], acc[m][nt][4 * rq + 2], acc[m][nt][4 * rq + 3]};
    }
    __syncthreads();
    f32x16 tot;
    {
      // own tile: sum the four waves' partials in wave order 0,1,2,3
      f32x16 own = acc[0][nt];
#pragma unroll
      for (int m = 1; m < 4; ++m)
        if (m == wave) own = acc[m][nt];
      bool first = true;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f32x16 part;
        if (s == wave) {
          part = own;
        } else {
          const int sp = s - (s > wave ? 1 : 0);
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(lds + (((wave * 3 + sp) * 4 + rq) * 64 + lane) * 16);
            part[4 * rq] = t[0];
            part[4 * rq + 1] = t[1];
            part[4 * rq + 2] = t[2];
            part[4 * rq + 3] = t[3];
          }
        }
        if (first) {
          tot = part;
          first = false;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[r] += part[r];
        }
      }
    }
    // epilogue of M-tile `wave`: rows 2*wave, 2*wave+1 of the tile; lane = output channel, registers = pixels
    const int g = gy * NT + nt;                           // N-tile of the launch
    const int co = p.y_coff[g] + (lane & 31);
    float s1 = 0.f, s2 = 0.f;
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

size_t stem_mx_packed_u16(int pieces, int ntiles) { return (size_t)49 * (pieces * 2 + (pieces == 3 ? 1 : 0)) * ntiles * 64 * 8; }

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

hipError_t launch_stem_mx(const StemMXArgs &a, int pieces, int ntiles_n, bool bf16_out, hipStream_t s) {
  StemMXArgs p = a;
  p.tiles_x = (a.Wo + TW - 1) / TW;
  p.tiles_y = (a.Ho + TH - 1) / TH;
  const long ntiles = (long)a.B * p.tiles_x * p.tiles_y;
  const unsigned gx = (unsigned)(((ntiles + 7) / 8) * 8);
  if (pieces == 3 && !bf16_out) {
    const size_t ldsb = PATCH_BYTES + 4 * 1 * 32 * 2 * sizeof(float);
    hipLaunchKernelGGL((stem_mx_kernel<3, 1, false>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
  } else if (pieces == 1 && bf16_out && ntiles_n % 2 == 0) {
    const size_t ldsb = PATCH_BYTES + 4 * 2 * 32 * 2 * sizeof(float);
    hipLaunchKernelGGL((stem_mx_kernel<1, 2, true>), dim3(gx, (unsigned)(ntiles_n / 2)), dim3(NTHREADS), ldsb, s, p);
  } else if (pieces == 1 && bf16_out) {
    const size_t ldsb = PATCH_BYTES + 4 * 1 * 32 * 2 * sizeof(float);
    hipLaunchKernelGGL((stem_mx_kernel<1, 1, true>), dim3(gx, (unsigned)ntiles_n), dim3(NTHREADS), ldsb, s, p);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace pnvo
