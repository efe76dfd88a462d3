// conv_bf16.hip — the residual stages of the native-bf16 forward (BASELINE config 3), gfx950 only.
//
// Every conv after the stem (3x3 stride 1 / 2, 1x1 stride 2, the compression conv; resnet.py:29-55,189-212, vo_cnn.py:
// 76-101) as an implicit GEMM on v_mfma_f32_32x32x16_bf16: bf16 activations in HBM (NHWC, channels a multiple of 32),
// bf16 weights, fp32 accumulation, GroupNorm statistics taken from the fp32 accumulators and kept in fp32.  The whole
// path is HBM/issue-bound at bf16 rates (the 3x3 blocks are 0.29 TFLOP per 256 pairs), so the kernel is built around
// touching every activation once: a workgroup stages the input patch of its output tile in LDS, applying the producer's
// GroupNorm + ReLU (MODE 1: relu(x*scale+shift), per-sample scale/shift from gn_finalize) ONCE per element while it
// converts, and zero padding after that transform as the reference does.  Pixel pitch in LDS = channels*2 + 16 bytes
// (an odd number of 16-byte units: the 16 lanes a ds_read_b128 serves per cycle hit all banks once).
//
// Workgroup = 4 waves, output tile = TR x TC pixels = MT 32-pixel M-tiles (row-major pixel order; a table in LDS maps a
// tile pixel to its patch offset and output position, so ragged tiles such as 4x22 or 6x11 cost nothing per MFMA).
// The waves form a WM x WN grid: wave = (M-tile group, N-tile group), MW x NW accumulators of 32x32 each; the B
// fragments stream from L2 (uniform base + lane), A fragments from LDS, both double-buffered one (tap, k-chunk) step
// ahead.  Input channels beyond CK are processed in chunks with the accumulators kept in registers.
// blockIdx.z selects one of up to two models (the geometric-invariance dual forward runs both in every launch).
#include <cstdlib>
#include <algorithm>
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
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2 r = __builtin_convertvector(f32x2{a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
}  // namespace

// BRES (round 4; 32 input channels: the first stage, the first conv of the second, its 1x1 downsample conv): the workgroup is
// PERSISTENT — it walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... — and the B fragments of the wave's N-tile(s), 9 taps x 2
// k-chunks = 18 fragments (72 registers), are loaded once and stay in registers.  What bounds these layers is the CU's vector-memory
// pipe (~20 B/clk), and re-fetching the same 18 KB of weights for every tile, by each of the four waves, was most of its traffic
// (conv_x3.hip conv_x3p_kernel: the float32-grade path's measurement).  Same MFMA order: bit-identical outputs.
// DSF (round 6; KS = 3, STRIDE = 2: the first conv of a stride-2 block): the block's 1x1 stride-2 downsample conv rides on the launch — it
// reads this conv's centre-tap pixels, so behind the nine taps of a chunk the centre tap's A fragments are read once more and multiplied
// with the downsample conv's weights (p.ds_wpk) into a second accumulator set; second epilogue (p.ds_y, p.ds_stats, own GroupNorm).  Same
// MFMAs in the same order as the separate 1x1 launch: bit-identical raw output.  As conv_x3_kernel's DSF.
template <int KS, int STRIDE, int MODE, bool F32OUT, int MW, int NW, bool BRES = false, bool DSF = false>
// (the resident-weight form is built for TWO waves per SIMD — its persistent grid is sized by the occupancy: 224 + 32 registers fit exactly,
//  two more and the compiler falls to one wave per SIMD, half the workgroups and 194 -> 309 us on the first stage's block tail)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BRES && MW <= 2 ? 2 : 1))) void conv_bf16_kernel(const ConvBArgs p) {
  static_assert(!DSF || (KS == 3 && STRIDE == 2 && !F32OUT && !BRES), "the downsample conv rides on a streaming 3x3 stride-2 conv");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PSTEP = KS == 1 ? STRIDE : 1;      // input pixels per patch pixel (a 1x1 conv stages only what it reads)
  constexpr int CS = KS == 1 ? 1 : STRIDE;         // patch pixels per output pixel
  constexpr int PAD = KS / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int z = blockIdx.z;
  const unsigned short *x = p.x[z];
  const u32x4 *wpk = reinterpret_cast<const u32x4 *>(p.wpk[z]);

  const int ntiles = p.B * p.tiles_r * p.tiles_c;
  const int chunk = (ntiles + 7) >> 3;
  constexpr int RB = BRES ? KS * KS * 2 : 1;                             // resident B: [step = tap * 2 + k-chunk][N-tile of the wave]
  u32x4 bres[RB][NW];
  if (BRES) {
    const int wn_ = p.wn, ntt_ = p.COUTP >> 5, kct_ = p.CIN >> 4;
    const int wave_n_ = (wave & (wn_ - 1)) + (int)blockIdx.y * (8 / NW);
#pragma unroll
    for (int st = 0; st < RB; ++st)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int nt = min(wave_n_ * NW + j, ntt_ - 1);
        bres[st][j] = wpk[(((st >> 1) * kct_ + (st & 1)) * ntt_ + nt) * 64 + lane];
      }
  }
  for (int vb = (int)blockIdx.x;; vb += (int)gridDim.x) {                // (one pass unless BRES)
  int bid = (vb & 7) * chunk + (vb >> 3);                               // consecutive tiles of an XCD are neighbours
  if (vb >= 8 * chunk || bid >= ntiles) return;
  const int tci = bid % p.tiles_c;
  bid /= p.tiles_c;
  const int tri = bid % p.tiles_r;
  const int n = bid / p.tiles_r;
  const int r0 = tri * p.TR, c0 = tci * p.TC;
  const int hi0 = r0 * STRIDE - PAD, wi0 = c0 * STRIDE - PAD;
  const int PR = p.PR, PC = p.PC, CK = p.CK;
  const int pitch = CK * 2 + 16;
  const int npix = p.TR * p.TC;
  // pixel table, two planes of MT*32 words: [0] patch byte offset of tile pixel q's top-left tap; [1] element offset of its
  // output pixel inside the sample's output plane ((tr*Wo + tc)*COUTP) with bit 31 set when the pixel does not exist
  // (beyond the tile's pixel count or outside the output)
  unsigned *qtab = reinterpret_cast<unsigned *>(lds + PR * PC * pitch);
  unsigned *otab = qtab + p.MT * 32;
  if ((int)threadIdx.x < p.MT * 32) {
    const int q = min((int)threadIdx.x, npix - 1);
    const int tr = q / p.TC, tc = q - tr * p.TC;
    qtab[threadIdx.x] = (unsigned)(((tr * CS) * PC + tc * CS) * pitch);
    const bool ok = (int)threadIdx.x < npix && r0 + tr < p.Ho && c0 + tc < p.Wo;
    otab[threadIdx.x] = (unsigned)((tr * p.Wo + tc) * p.COUTP) | (ok ? 0u : 0x80000000u);
  }

  const int wn = p.wn;                                                   // wave grid: (4 / wn) x wn
  const int wave_m = wave / wn;
  const int wave_n = (wave & (wn - 1)) + (int)blockIdx.y * (8 / NW);      // layers wider than 8 N-tiles: blockIdx.y = N group
  const int ntt = p.COUTP >> 5, kct = p.CIN >> 4;                        // N-tiles, 16-channel k-chunks of the layer

  f32x16 acc[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x16 accd[DSF ? MW : 1][DSF ? NW : 1];
  if (DSF) {
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accd[DSF ? i : 0][DSF ? j : 0][r] = 0.f;
  }

  // staging role: thread -> (pixel lane, 8-channel group); G groups per pixel, 256 / G pixels per step
  const int G = CK >> 3;
  const int cg = threadIdx.x & (G - 1), pl = threadIdx.x / G, PS = 256 / G;
  const int dr = PS / PC, dc = PS - dr * PC;
  const int nppix = PR * PC;

  for (int ck0 = 0; ck0 < p.CIN; ck0 += CK) {
    if (ck0 > 0) __syncthreads();                                        // the previous chunk's patch is no longer read
    {
      f32x4 sc0, sc1, sh0, sh1, rc0, rc1, rh0, rh1;
      if (MODE >= 1) {
        const float *ps = p.in_scale[z] + (long)n * p.CIN + ck0 + 8 * cg;
        const float *pt = p.in_shift[z] + (long)n * p.CIN + ck0 + 8 * cg;
        sc0 = *reinterpret_cast<const f32x4 *>(ps);
        sc1 = *reinterpret_cast<const f32x4 *>(ps + 4);
        sh0 = *reinterpret_cast<const f32x4 *>(pt);
        sh1 = *reinterpret_cast<const f32x4 *>(pt + 4);
      }
      const bool r_affine = MODE == 2 && p.in_scale2[z] != nullptr;       // the skip branch is a raw conv output (downsample)
      if (MODE == 2) {
        rc0 = rc1 = f32x4{1.f, 1.f, 1.f, 1.f};
        rh0 = rh1 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r_affine) {
          const float *ps = p.in_scale2[z] + (long)n * p.CIN + ck0 + 8 * cg;
          const float *pt = p.in_shift2[z] + (long)n * p.CIN + ck0 + 8 * cg;
          rc0 = *reinterpret_cast<const f32x4 *>(ps);
          rc1 = *reinterpret_cast<const f32x4 *>(ps + 4);
          rh0 = *reinterpret_cast<const f32x4 *>(pt);
          rh1 = *reinterpret_cast<const f32x4 *>(pt + 4);
        }
      }
      int pr = pl / PC, pc = pl - pr * PC;
      const long xoff = ((long)n * p.H * p.W) * p.CIN + ck0 + 8 * cg;
      const unsigned short *xb = x + xoff;
      const unsigned short *xb2 = MODE == 2 ? p.x2[z] + xoff : nullptr;
      unsigned short *xo = (MODE == 2 && p.xout[z] != nullptr) ? p.xout[z] + xoff : nullptr;
      for (int pix = pl; pix < nppix; pix += 4 * PS) {
        u32x4 v[4], v2[4];
        unsigned off[4], inmask = 0, ownmask = 0;
        long eoff[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int hi = hi0 + pr * PSTEP, wi = wi0 + pc * PSTEP;
          ok[k] = pix + k * PS < nppix;
          const bool in = ok[k] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
          off[k] = (unsigned)((pr * PC + pc) * pitch + 16 * cg);
          eoff[k] = ((long)hi * p.W + wi) * p.CIN;
          v[k] = u32x4{0u, 0u, 0u, 0u};
          if (in) v[k] = *reinterpret_cast<const u32x4 *>(xb + eoff[k]);
          if (MODE == 2) {
            v2[k] = u32x4{0u, 0u, 0u, 0u};
            if (in) v2[k] = *reinterpret_cast<const u32x4 *>(xb2 + eoff[k]);
            // every input pixel is materialised by exactly one tile: the one whose output region it falls into
            const int qr = hi / STRIDE - r0, qc = wi / STRIDE - c0;
            ownmask |= (unsigned)(in && qr >= 0 && qr < p.TR && qc >= 0 && qc < p.TC) << k;
          }
          inmask |= (unsigned)in << k;                                   // outside the image: zero AFTER the transform
          pr += dr;
          pc += dc;
          if (pc >= PC) {
            pc -= PC;
            pr += 1;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!ok[k]) continue;
          u32x4 o = v[k];
          if (MODE >= 1) {
            if (!((inmask >> k) & 1u)) {
              o = u32x4{0u, 0u, 0u, 0u};
            } else {
              float f[8];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const f32x4 &sc = e < 2 ? sc0 : sc1, &sh = e < 2 ? sh0 : sh1;
                f[2 * e] = __builtin_fmaf(lo_f(v[k][e]), sc[(2 * e) & 3], sh[(2 * e) & 3]);
                f[2 * e + 1] = __builtin_fmaf(hi_f(v[k][e]), sc[(2 * e + 1) & 3], sh[(2 * e + 1) & 3]);
              }
              if (MODE == 2) {                                           // BasicBlock tail: + skip branch, then the ReLU
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const f32x4 &sc = e < 2 ? rc0 : rc1, &sh = e < 2 ? rh0 : rh1;
                  f[2 * e] += __builtin_fmaf(lo_f(v2[k][e]), sc[(2 * e) & 3], sh[(2 * e) & 3]);
                  f[2 * e + 1] += __builtin_fmaf(hi_f(v2[k][e]), sc[(2 * e + 1) & 3], sh[(2 * e + 1) & 3]);
                }
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = pack2(fmaxf(f[2 * e], 0.f), fmaxf(f[2 * e + 1], 0.f));
            }
          }
          *reinterpret_cast<u32x4 *>(lds + off[k]) = o;
          if (MODE == 2 && xo != nullptr && ((ownmask >> k) & 1u)) *reinterpret_cast<u32x4 *>(xo + eoff[k]) = o;
        }
      }
    }
    __syncthreads();

    // ---- compute: steps s = (tap, 16-channel chunk); fragments of step s+1 are fetched during the MFMAs of step s
    unsigned aoff[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int mt = min(wave_m * MW + i, p.MT - 1);
      aoff[i] = qtab[mt * 32 + (lane & 31)] + (unsigned)((lane >> 5) * 16);
    }
    const int kcc = CK >> 4;                                             // k-chunks per staged chunk
    const int nsteps = KS * KS * kcc;
    // The step being fetched, kept as running offsets (a division per step costs ~50 scalar instructions between two small
    // MFMA groups — these kernels are bound by their instruction count, not by the matrix pipe).
    const unsigned kstep = (unsigned)ntt * 1024u;                        // bytes of one k-chunk of B (all N-tiles)
    const char *wb_n = reinterpret_cast<const char *>(wpk) + (long)(ck0 >> 4) * kstep;
    unsigned toff_n = 0;                                                 // patch byte offset of the step's (tap, k-chunk)
    int kc_n = 0, kw_n = 0;
    auto advance = [&]() {
      ++kc_n;
      toff_n += 32;
      wb_n += kstep;
      if (kc_n == kcc) {
        kc_n = 0;
        toff_n += (unsigned)(pitch - kcc * 32);
        wb_n += (long)(kct - kcc) * kstep;
        if (++kw_n == KS) {
          kw_n = 0;
          toff_n += (unsigned)((PC - KS) * pitch);
        }
      }
    };
    unsigned voff[NW];                                                   // lane's byte offset inside a k-chunk of B
#pragma unroll
    for (int j = 0; j < NW; ++j) voff[j] = (unsigned)min(wave_n * NW + j, ntt - 1) * 1024u + (unsigned)lane * 16u;   // (N-tiles past the layer's repeat the last one)
    auto loadAB = [&](u32x4 *a, u32x4 *b) {
#pragma unroll
      for (int i = 0; i < MW; ++i) a[i] = *reinterpret_cast<const u32x4 *>(lds + aoff[i] + toff_n);
#pragma unroll
      for (int j = 0; j < NW; ++j) b[j] = *reinterpret_cast<const u32x4 *>(wb_n + (size_t)voff[j]);
    };
    auto mfmas = [&](const u32x4 *a, const u32x4 *b) {
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int i = 0; i < MW; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                              acc[i][j], 0, 0, 0);
    };
    u32x4 a0[MW], b0[NW], a1[MW], b1[NW];
    if (BRES) {
      // steps in the streaming loop's order (tap-major, two k-chunks inner); A of step st + 1 is fetched behind the MFMAs of step st
      auto loadA = [&](int st, u32x4 *a) {
        const int tap = st >> 1;
        const unsigned toff = (unsigned)(((tap / KS) * PC + (tap % KS)) * pitch + (st & 1) * 32);
#pragma unroll
        for (int i = 0; i < MW; ++i) a[i] = *reinterpret_cast<const u32x4 *>(lds + aoff[i] + toff);
      };
      loadA(0, a0);
#pragma unroll
      for (int st = 0; st < RB; st += 2) {
        loadA(st + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a0, bres[BRES ? st : 0]);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < RB) loadA(st + 2, a0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(a1, bres[BRES ? st + 1 : 0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    loadAB(a0, b0);
    advance();                                                           // -> step 1
#pragma unroll 1
    for (int s = 0; s < nsteps; s += 2) {                                // nsteps is even for every supported shape
      const bool more = s + 2 < nsteps;                                  // (the last iteration re-fetches step s + 1: unused)
      loadAB(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) advance();
      loadAB(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      if (more) advance();
    }
    }
    if constexpr (DSF) {
      // the riding downsample conv: this conv's centre tap (patch offset (PC + 1) pixels) on its own weights, k-chunks ascending
      const unsigned tc0 = (unsigned)((PC + 1) * pitch);
      const char *wd = reinterpret_cast<const char *>(p.ds_wpk[z]) + (long)(ck0 >> 4) * kstep;
      auto dstep = [&](int kc, const u32x4 *bd) {
        u32x4 ad[MW];
#pragma unroll
        for (int i = 0; i < MW; ++i) ad[i] = *reinterpret_cast<const u32x4 *>(lds + aoff[i] + tc0 + (unsigned)kc * 32u);
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
          for (int i = 0; i < MW; ++i)
            accd[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ad[i]), __builtin_bit_cast(bf16x8, bd[j]), accd[i][j], 0,
                                                                 0, 0);
      };
#pragma unroll 1
      for (int kc = 0; kc < kcc; ++kc) {
        u32x4 bd[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) bd[j] = *reinterpret_cast<const u32x4 *>(wd + (long)kc * kstep + (size_t)voff[j]);
        dstep(kc, bd);
      }
    }
  }

  // ---- epilogue: raw output + per-(sample, slot, channel) GroupNorm partial sums (one writer per slot).
  // lane = output channel, registers = 16 pixels of the M-tile (rows 4 q4 + e + 8 ... of the accumulator layout); the
  // pixel's output offset comes from the table (four 16-byte LDS reads per M-tile), the store is base + 32-bit offset.
  const int rr16 = lane >> 5;
  const long ybase = (((long)n * p.Ho + r0) * p.Wo + c0) * p.COUTP;
  // (runs once for the conv and, DSF, once more for the downsample conv that rode on it)
  auto emit = [&](f32x16 (*ac)[NW], void *yout, float *stats, const float *ggamma, const float *gbeta, float *gscale, float *gshift, bool is_ds) {
    const bool f32o = F32OUT && !is_ds;                                  // (compile-time false for every bf16-output variant: the riding conv's output is bf16)
    float t1[NW], t2[NW];
  #pragma unroll
    for (int j = 0; j < NW; ++j) t1[j] = t2[j] = 0.f;
  #pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int mt = wave_m * MW + i;
      if (mt >= p.MT) continue;
      u32x4 ent[4];
  #pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) ent[g4] = *reinterpret_cast<const u32x4 *>(otab + mt * 32 + 8 * g4 + 4 * rr16);
  #pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int nt = wave_n * NW + j;
        if (nt >= ntt) continue;
        const int co = nt * 32 + (lane & 31);
        float s1 = 0.f, s2 = 0.f;
  #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned e = ent[r >> 2][r & 3];
          const bool ok = (int)e >= 0;
          const float v = ok ? ac[i][j][r] : 0.f;
          if (ok) {
            if (f32o)
              (reinterpret_cast<float *>(yout) + ybase + co)[e] = v;
            else
              (reinterpret_cast<__bf16 *>(yout) + ybase + co)[e] = (__bf16)v;
          }
          s1 += v;
          s2 = __builtin_fmaf(v, v, s2);
        }
        t1[j] += s1 + __shfl_xor(s1, 32);                                  // M-tiles of this wave, in order
        t2[j] += s2 + __shfl_xor(s2, 32);
      }
    }
    if (stats != nullptr) {                                           // one slot per tile: the waves along M meet in LDS
      const int rows = 4 / wn;
      float *red = reinterpret_cast<float *>(lds);                        // [wave][NW][32 channels][2]
      if (rows > 1) {
        __syncthreads();                                                   // the patch is no longer read
        if (lane < 32)
  #pragma unroll
          for (int j = 0; j < NW; ++j) *reinterpret_cast<f32x2 *>(red + ((wave * NW + j) * 32 + lane) * 2) = f32x2{t1[j], t2[j]};
        __syncthreads();
      }
      if (wave_m == 0 && lane < 32) {
  #pragma unroll
        for (int j = 0; j < NW; ++j) {
          const int nt = wave_n * NW + j;
          if (nt >= ntt) continue;
          float s1 = t1[j], s2 = t2[j];
          for (int w = 1; w < rows; ++w) {                                 // fixed order: bit-reproducible
            const f32x2 o = *reinterpret_cast<const f32x2 *>(red + (((w * wn + (wave & (wn - 1))) * NW + j) * 32 + lane) * 2);
            s1 += o[0];
            s2 += o[1];
          }
          float *dst = stats + (((long)n * p.slots + tri * p.tiles_c + tci) * p.COUTP + nt * 32 + lane) * 2;
          dst[0] = s1;
          dst[1] = s2;
          if (gscale != nullptr) {                                  // the sample's only slot: finalise here (no launch), as conv_x3_kernel
            const int c = nt * 32 + lane;
            gn_finalize_lane(s1, s2, p.gn_cpg, p.gn_P, p.gn_eps, ggamma[c], gbeta[c], gscale + (long)n * p.COUTP + c,
                             gshift + (long)n * p.COUTP + c);
          }
        }
      }
    }
  };
  emit(acc, p.y[z], p.stats[z], p.gn_gamma[z], p.gn_beta[z], p.gn_scale[z], p.gn_shift[z], false);
  if constexpr (DSF) emit(accd, p.ds_y[z], p.ds_stats[z], p.ds_gamma[z], p.ds_beta[z], p.ds_scale[z], p.ds_shift[z], true);
  if (!BRES) return;
  __syncthreads();                                                       // patch, tables and reduction scratch are free again
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side: tile / wave-grid plan of one layer and the launch.
namespace {
template <int KS, int STRIDE>
hipError_t launch_ks(const ConvBArgs &a, int mode, bool f32out, int mw, int nw, dim3 grid, size_t ldsb, hipStream_t s) {
#define PNVO_CB(MODE_, F32_, MW_, NW_)                                                                          \
  if (mode == MODE_ && f32out == F32_ && mw == MW_ && nw == NW_) {                                             \
    hipLaunchKernelGGL((conv_bf16_kernel<KS, STRIDE, MODE_, F32_, MW_, NW_>), grid, dim3(256), ldsb, s, a);    \
    return hipGetLastError();                                                                                  \
  }
  if constexpr (KS == 3 && STRIDE == 2) {
    if (a.ds_wpk[0] != nullptr) {                  // the block's downsample conv rides on this launch
#define PNVO_CBD(MODE_, MW_, NW_)                                                                                        \
  if (mode == MODE_ && !f32out && mw == MW_ && nw == NW_) {                                                             \
    hipLaunchKernelGGL((conv_bf16_kernel<KS, STRIDE, MODE_, false, MW_, NW_, false, true>), grid, dim3(256), ldsb, s, a); \
    return hipGetLastError();                                                                                           \
  }
      PNVO_CBD(0, 1, 1) PNVO_CBD(2, 1, 1) PNVO_CBD(0, 2, 1) PNVO_CBD(2, 2, 1) PNVO_CBD(0, 3, 1) PNVO_CBD(2, 3, 1)
      PNVO_CBD(0, 4, 1) PNVO_CBD(2, 4, 1)          // (two N-tiles per wave: 2 x 128 accumulator registers, not built; the host keeps the launch)
#undef PNVO_CBD
      return hipErrorInvalidValue;
    }
  }
  PNVO_CB(0, false, 1, 1) PNVO_CB(1, false, 1, 1) PNVO_CB(0, false, 2, 1) PNVO_CB(1, false, 2, 1)
  PNVO_CB(0, false, 4, 1) PNVO_CB(1, false, 4, 1) PNVO_CB(0, false, 4, 2) PNVO_CB(1, false, 4, 2)
  PNVO_CB(0, false, 3, 1) PNVO_CB(1, false, 3, 1) PNVO_CB(0, false, 3, 2) PNVO_CB(1, false, 3, 2) PNVO_CB(0, true, 3, 2)
  PNVO_CB(0, true, 3, 1)
  PNVO_CB(0, true, 1, 1) PNVO_CB(0, true, 2, 1) PNVO_CB(0, true, 4, 1) PNVO_CB(0, true, 4, 2)
  if (KS == 3) {                                   // fused BasicBlock tail (MODE 2): conv1 of a block / the compression conv
    PNVO_CB(2, false, 1, 1) PNVO_CB(2, false, 2, 1) PNVO_CB(2, false, 4, 1) PNVO_CB(2, false, 4, 2)
    PNVO_CB(2, false, 3, 1) PNVO_CB(2, false, 3, 2)
    PNVO_CB(2, true, 1, 1) PNVO_CB(2, true, 2, 1) PNVO_CB(2, true, 3, 1) PNVO_CB(2, true, 3, 2) PNVO_CB(2, true, 4, 1) PNVO_CB(2, true, 4, 2)
  }
#undef PNVO_CB
  return hipErrorInvalidValue;
}
}  // namespace

// Fills the plan fields of `a` (TR, TC, tiles, PR, PC, CK, MT, wn, slots) from its shape fields; returns false when the
// layer is outside what the kernel covers (the caller then refuses the bf16 mode for that model).
bool conv_bf16_plan(ConvBArgs &a, int ks, int stride, int *mw, int *nw, size_t *lds_bytes) {
  if (a.CIN % 32 || a.COUTP % 32 || a.COUTP > 1024 || a.CIN > 256) return false;
  if (!((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 2))) return false;
  const int ntt = a.COUTP / 32;
  // output tile: <= 128 pixels; 8 x 16 for the wide stages, whole-width strips for the narrow ones
  int TR, TC;
  if (a.Wo >= 32) {
    // 16 x 16 tiles (two / four M-tiles per wave: every B fragment feeds more MFMAs, the per-tile overheads halve) where
    // the layer is narrow enough for the accumulators and the patch still fits; measured on layer 1: 0.155 -> 0.116 ms
    TR = 8;
    TC = 16;
    if (ntt == 1 && a.Ho % 16 == 0 && !std::getenv("PNVO_BF16_T8")) {   // (layer 2 at 24 x 43 wastes a third of 16-row tiles: slower)
      const int cs_ = ks == 1 ? 1 : stride;
      const size_t patch = (size_t)(15 * cs_ + ks) * (15 * cs_ + ks) * (a.CIN * 2 + 16);
      if (patch <= (size_t)56 * 1024) TR = 16;
    }
  } else {
    TC = a.Wo;
    TR = 128 / TC;
    if (TR > a.Ho) TR = a.Ho;
    // balance the strips over the rows
    const int nstr = (a.Ho + TR - 1) / TR;
    TR = (a.Ho + nstr - 1) / nstr;
  }
  a.TR = TR;
  a.TC = TC;
  a.tiles_r = (a.Ho + TR - 1) / TR;
  a.tiles_c = (a.Wo + TC - 1) / TC;
  a.MT = (TR * TC + 31) / 32;
  const int cs = ks == 1 ? 1 : stride;
  a.PR = (TR - 1) * cs + ks;
  a.PC = (TC - 1) * cs + ks;
  // wave grid and accumulators per wave
  if (ntt == 1) {
    a.wn = 1;
    *mw = TR * TC > 128 ? 2 : 1;
    *nw = 1;
  } else if (ntt == 2) {
    a.wn = 2;
    *mw = TR * TC > 128 ? 4 : 2;
    *nw = 1;
  } else if (ntt == 4) {
    a.wn = 4;
    *mw = 4;
    *nw = 1;
  } else {
    a.wn = 4;
    *mw = 4;
    *nw = 2;
  }
  if (a.wn == 4 && a.MT < *mw) *mw = a.MT == 3 ? 3 : *mw;               // ragged strips: three M-tiles, fewer accumulators
  if ((4 / a.wn) * *mw < a.MT) return false;
  // channel chunk: the largest power of two (<= CIN) whose patch fits 56 KB
  int ck = a.CIN;
  while (ck > 32 && (size_t)a.PR * a.PC * (ck * 2 + 16) > (size_t)56 * 1024) ck /= 2;
  if ((size_t)a.PR * a.PC * (ck * 2 + 16) > (size_t)60 * 1024) return false;
  if (a.CIN % ck) return false;
  a.CK = ck;
  a.slots = a.tiles_r * a.tiles_c;                               // one GroupNorm partial per tile
  *lds_bytes = (size_t)a.PR * a.PC * (ck * 2 + 16) + (size_t)a.MT * 32 * 4 * 2;
  return true;
}

hipError_t launch_conv_bf16(const ConvBArgs &a, int ks, int stride, int mode, bool f32out, int mw, int nw, size_t lds_bytes,
                            int nmodels, hipStream_t s) {
  const long ntiles = (long)a.B * a.tiles_r * a.tiles_c;
  dim3 grid((unsigned)(((ntiles + 7) / 8) * 8), (unsigned)((a.COUTP + 255) / 256)   /* groups of 8 N-tiles */, (unsigned)nmodels);
  // persistent workgroups with the weights resident in registers (BRES): 32 input channels in one staged chunk, one N-tile per wave
  if (a.persist_wgs >= 8 && a.CIN == 32 && a.CK == 32 && nw == 1 && !f32out && a.ds_wpk[0] == nullptr && ntiles >= 2L * a.persist_wgs) {
    const int cus = a.persist_wgs / 3;
    // as many workgroups as are RESIDENT at once (registers / LDS of the variant decide: 2-4 per CU): a persistent grid larger than
    // that would run its surplus workgroups as a second, unbalanced round
#define PNVO_CBP(KS_, ST_, MODE_, MW_)                                                                                  \
  if (ks == KS_ && stride == ST_ && mode == MODE_ && mw == MW_) {                                                       \
    auto kfn = conv_bf16_kernel<KS_, ST_, MODE_, false, MW_, 1, true>;                                                  \
    int occ = 0;                                                                                                        \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 256, lds_bytes) != hipSuccess || occ < 1) occ = 1;      \
    const int per_cu = std::max(1, occ / (int)(grid.y * grid.z));                                                       \
    dim3 pg((unsigned)((per_cu * cus) & ~7), grid.y, grid.z);                                                           \
    if (pg.x >= 8 && ntiles >= 2L * pg.x) {                                                                             \
      hipLaunchKernelGGL(kfn, pg, dim3(256), lds_bytes, s, a);                                                          \
      return hipGetLastError();                                                                                         \
    }                                                                                                                   \
  }
    // (measured at 256 pairs x 2 models: the block-tail convs 0.240 -> 0.199 and 0.167 -> 0.146 ms, the 1x1 downsample conv 0.047 ->
    //  0.039; the plain and GroupNorm-input convs LOSE — 0.11 -> 0.13 ms: their 60-register streaming form keeps five workgroups
    //  per CU in flight, the 72 resident registers leave three — and stay on the streaming form)
    PNVO_CBP(3, 1, 2, 2) PNVO_CBP(3, 1, 2, 4) PNVO_CBP(3, 2, 2, 2) PNVO_CBP(1, 2, 0, 2) PNVO_CBP(3, 2, 2, 1) PNVO_CBP(1, 2, 0, 1)
#undef PNVO_CBP
  }
  if (ks == 3 && stride == 1) return launch_ks<3, 1>(a, mode, f32out, mw, nw, grid, lds_bytes, s);
  if (ks == 3 && stride == 2) return launch_ks<3, 2>(a, mode, f32out, mw, nw, grid, lds_bytes, s);
  if (ks == 1 && stride == 2) return launch_ks<1, 2>(a, mode, f32out, mw, nw, grid, lds_bytes, s);
  return hipErrorInvalidValue;
}

// B operand:  out[tap][k-chunk (cin/16)][N-tile (coutp/32)][lane = kh*32 + n][8 bf16] = W[N-tile*32 + n][16 kc + 8 kh + j][tap]
void pack_conv_bf16_weight(const float *oihw, int cout, int cin, int cinp, int coutp, int kh_, int kw_, unsigned short *out) {
  const int T = kh_ * kw_, kct = cinp / 16, ntt = coutp / 32;
  auto bf = [](float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  };
  for (int tap = 0; tap < T; ++tap)
    for (int kc = 0; kc < kct; ++kc)
      for (int nt = 0; nt < ntt; ++nt)
        for (int ln = 0; ln < 64; ++ln)
          for (int j = 0; j < 8; ++j) {
            const int co = nt * 32 + (ln & 31), ci = 16 * kc + 8 * (ln >> 5) + j;
            float v = 0.f;
            if (co < cout && ci < cin) v = oihw[((size_t)co * cin + ci) * T + tap];
            out[((((size_t)tap * kct + kc) * ntt + nt) * 64 + ln) * 8 + j] = bf(v);
          }
}

// ---------------------------------------------------------------------------------------------------------------------
// relu(gn(x)) then MaxPool 3x3 s2 p1 (resnet.py:165-168), bf16 in / bf16 out, 8 channels per thread.
__global__ __launch_bounds__(256) void gn_relu_maxpool_bf16_kernel(const unsigned short *x0, const unsigned short *x1,
                                                                 const float *sc0, const float *sc1, const float *sh0,
                                                                 const float *sh1, int B, int H, int W, int C, int Ho, int Wo,
                                                                 unsigned short *o0, unsigned short *o1) {
  const int z = blockIdx.z;
  const unsigned short *x = z ? x1 : x0;
  const float *scale = z ? sc1 : sc0, *shift = z ? sh1 : sh0;
  unsigned short *out = z ? o1 : o0;
  const int Q = C >> 3;
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * Ho * Wo * Q;
  if (g >= total) return;
  const int q = (int)(g % Q);
  long r = g / Q;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const int n = (int)(r / Ho);
  float sc[8], sh[8], m[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    sc[t] = scale[(long)n * C + 8 * q + t];
    sh[t] = shift[(long)n * C + 8 * q + t];
    m[t] = 0.f;                                // every window holds >= 1 real pixel and relu(.) >= 0
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = 2 * ho - 1 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = 2 * wo - 1 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(x + (((long)n * H + hi) * W + wi) * C + 8 * q);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        m[2 * t] = fmaxf(m[2 * t], __builtin_fmaf(lo_f(v[t]), sc[2 * t], sh[2 * t]));
        m[2 * t + 1] = fmaxf(m[2 * t + 1], __builtin_fmaf(hi_f(v[t]), sc[2 * t + 1], sh[2 * t + 1]));
      }
    }
  }
  reinterpret_cast<u32x4 *>(out)[g] = u32x4{pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7])};
}

hipError_t launch_gn_relu_maxpool_bf16(const unsigned short *const *x, const float *const *scale, const float *const *shift,
                                       int B, int H, int W, int C, unsigned short *const *out, int nmodels, hipStream_t s) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(gn_relu_maxpool_bf16_kernel, dim3((unsigned)((total + 255) / 256), 1, (unsigned)nmodels), dim3(256), 0, s,
                     x[0], x[nmodels - 1], scale[0], scale[nmodels - 1], shift[0], shift[nmodels - 1], B, H, W, C, Ho, Wo,
                     out[0], out[nmodels - 1]);
  return hipGetLastError();
}

// BasicBlock tail (resnet.py:47-55): y = relu(GN2(conv2) + residual), residual = x (final activations) or GN_d(conv1x1(x)).
struct ResidualBArgs {
  const unsigned short *a[2], *b[2];
  const float *sa[2], *ta[2], *sb[2], *tb[2];
  unsigned short *y[2];
  long PC;
  int C;
  long total8;
};
__global__ __launch_bounds__(256) void residual_bf16_kernel(const ResidualBArgs p) {
  const int z = blockIdx.z;
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  if (g >= p.total8) return;
  const long e = g * 8;
  const int n = (int)(e / p.PC);
  const int c = (int)(e % p.C);
  const u32x4 va = reinterpret_cast<const u32x4 *>(p.a[z])[g];
  const u32x4 vb = reinterpret_cast<const u32x4 *>(p.b[z])[g];
  const float *sa = p.sa[z] + (long)n * p.C + c, *ta = p.ta[z] + (long)n * p.C + c;
  float r[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    r[2 * t] = lo_f(vb[t]);
    r[2 * t + 1] = hi_f(vb[t]);
  }
  if (p.sb[z] != nullptr) {
    const float *sb = p.sb[z] + (long)n * p.C + c, *tb = p.tb[z] + (long)n * p.C + c;
#pragma unroll
    for (int t = 0; t < 8; ++t) r[t] = __builtin_fmaf(r[t], sb[t], tb[t]);
  }
  float o[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    o[2 * t] = fmaxf(__builtin_fmaf(lo_f(va[t]), sa[2 * t], ta[2 * t]) + r[2 * t], 0.f);
    o[2 * t + 1] = fmaxf(__builtin_fmaf(hi_f(va[t]), sa[2 * t + 1], ta[2 * t + 1]) + r[2 * t + 1], 0.f);
  }
  reinterpret_cast<u32x4 *>(p.y[z])[g] = u32x4{pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
}

hipError_t launch_residual_bf16(const unsigned short *const *a, const float *const *sa, const float *const *ta,
                                const unsigned short *const *b, const float *const *sb, const float *const *tb, int B, long P,
                                int C, unsigned short *const *y, int nmodels, hipStream_t s) {
  ResidualBArgs p;
  for (int z = 0; z < 2; ++z) {
    const int k = z < nmodels ? z : 0;
    p.a[z] = a[k];
    p.b[z] = b[k];
    p.sa[z] = sa[k];
    p.ta[z] = ta[k];
    p.sb[z] = sb ? sb[k] : nullptr;
    p.tb[z] = tb ? tb[k] : nullptr;
    p.y[z] = y[k];
  }
  p.PC = P * C;
  p.C = C;
  p.total8 = (long)B * P * C / 8;
  hipLaunchKernelGGL(residual_bf16_kernel, dim3((unsigned)((p.total8 + 255) / 256), 1, (unsigned)nmodels), dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace pnvo
