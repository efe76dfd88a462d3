// stem_rs.hip — the 7x7 stride-2 stem on the 16-bit matrix cores with its WEIGHTS RESIDENT IN REGISTERS (gfx950 only).
//
// The operation, operand layout and tap split of stem_mx_kernel (stem_mx.hip: conv1 of resnet.py:156-163 with the input assembly and
// whitening of vo_cnn.py:110-176 in front and, POOL, the max-pool of resnet.py:168 behind).  What changes is where the B operand
// lives.  Three uses:
//   stem_rs_kernel<2, POOL, RAW, FAST>   float32-grade results from two float16 weight pieces (the inference stem; !POOL: the raw-output
//                                        stem of pool=separate and of the training forward).  FAST = false: stem_mx_kernel's summation
//                                        order, bit-identical results; FAST = true (default): 15 % fewer MFMAs, see the kernel.
//   stem_rs_kernel<1, false, RAW>        the bf16 dual stem (two models' 32 channels: stem_mx_kernel<1, 2, true>), bit-identical.
//
// Round 4 measured that the tile-per-workgroup stem and its role-specialised persistent form are bound by the CU's vector-memory
// pipe: every 128-pixel tile re-fetches all 245 KB of weight fragments next to 93 KB of patch, ~16 k cycles per tile against 7.8 k
// cycles of MFMA issue.  245 KB is half of a CU's 512 KB register file.  Here ONE workgroup of FOUR waves owns a CU for the whole
// launch, one wave per SIMD with the full 512-register budget (256 VGPRs + 256 AGPRs), and each wave keeps the fragments of its
// twelve or thirteen taps — pinned to AGPRs, where the MFMAs read them in place — for all of its ~130 tiles.  What is left on the
// vector-memory pipe is the patch.
//
// With one wave per SIMD nothing overlaps by itself, and the wave issues in order.  Measured in this kernel: about 3.6 instruction
// slots hide behind one 32x32x16 MFMA; beyond that every instruction costs its ~4 cycles.  So:
//   - the body is compiled once per wave (a switch over the wave index): taps, patch rows and M-tile are constants, every LDS address
//     of the K loop is one register plus an immediate;
//   - the K loop is fully unrolled (it has to be: the resident fragments are indexed statically) in groups of four MFMAs — one B
//     fragment x the four M-tiles — each with its share of the other work in source order: fragment reads, the loads of the NEXT
//     tile's patch (buffer loads: an out-of-range offset for absent lanes instead of an address select, which the compiler turns into
//     a branch — and a branch, or a short-circuit &&, ends the scheduling region), their conversion and LDS writes three taps later,
//     and the epilogue of the PREVIOUS tile cut into pieces that read LDS in one group and compute in the next;
//   - per tile three workgroup barriers around the K-split exchange (through the patch buffer just consumed) and one over LDS only
//     inside the K loop (the previous tile's partial sums / pooling scratch).
//
// LDS: two 64 KB buffers (patch of tile i, then exchange of tile i | patch of tile i+1) + pooling scratch + partial sums: 147 KB.
// Evidence: profiles/r4_stem_rs_phases.txt (cycles per tile), profiles/r4_stem_rs_ablations.txt (what each part costs).
#include <type_traits>

#include "stem_tile.h"

namespace pnvo {

namespace {
constexpr int RS_BUF = XCHG_BYTES;                          // 65536 >= PATCH_BYTES
constexpr int RS_PB_OFF = 2 * RS_BUF;                       // pooling scratch [8][16][33] floats
constexpr int RS_RED_OFF = RS_PB_OFF + 8 * 16 * 33 * 4;     // [4 waves][N-tiles <= 2][32][2] floats
constexpr int RS_ETAB_OFF = RS_RED_OFF + 4 * 2 * 32 * 2 * 4;   // RAW: bin edges (12 floats)
constexpr int RS_TRASH_OFF = RS_ETAB_OFF + 64;              // target of the writes of absent lanes
constexpr int RS_LDS = RS_TRASH_OFF + 64;
static_assert(RS_BUF >= PATCH_BYTES, "patch must fit its buffer");
static_assert(RS_LDS <= 160 * 1024, "LDS budget");
#define PNVO_INL __attribute__((always_inline))
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): an unrolled loop whose index is a constant BEFORE the optimiser
// runs (the tap loop holds every epilogue piece behind `if (tap == ...)`: as a #pragma unroll loop its body is over the unroller's
// size limit until those conditions fold)
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}
// compile-time tunables (each one measured, profiles/r4_stem_rs_ablations.txt; the defaults are what ships)
#ifndef PNVO_RS_LAGD
#define PNVO_RS_LAGD 3   // taps between the load of a one-hot-depth granule round and its conversion + LDS write (2..5: no difference)
#endif
#ifndef PNVO_RS_LAGP
#define PNVO_RS_LAGP 3   // the same for the pixel rounds
#endif
#ifndef PNVO_RS_GDS
#define PNVO_RS_GDS 2    // scheduling pattern per MFMA of a region: at most this many LDS reads, ...
#endif
#ifndef PNVO_RS_GVM
#define PNVO_RS_GVM 1    // ... vector-memory instructions ...
#endif
#ifndef PNVO_RS_GVA
#define PNVO_RS_GVA 4    // ... and VALU instructions (plus one LDS write); 3..8: no difference
#endif
#ifndef PNVO_RS_A1AHEAD
#define PNVO_RS_A1AHEAD 0   // chunk-1 fragments fetched a whole tap ahead as well (no difference: not LDS latency)
#endif
#ifndef PNVO_RS_MERGE
#define PNVO_RS_MERGE 1  // bit 0: groups 0+1 and 2+3 of a tap are one scheduling region (K loop -4 %); bits 1, 2: larger unions (slower)
#endif
#ifndef PNVO_RS_MIN_TILES
#define PNVO_RS_MIN_TILES 4   // tiles per workgroup from which the resident kernel takes the launch (8 pairs of 341x192; measured break-even ~6)
#endif
#ifndef PNVO_RS_NPIN
#define PNVO_RS_NPIN 10  // taps whose fragments are pinned to AGPRs (8..13: no difference once most are)
#endif
#ifndef PNVO_RS_ABL
#define PNVO_RS_ABL 0   // developer: compile-time ablations for register-pressure studies (1 no granules, 2 no pixel rounds, 4 no epilogue pieces in the K loop)
#endif
}  // namespace

// PIECES = 2: float32-grade results from two float16 weight pieces, one N-tile of 32 channels, float32 output (pooled keys or raw).
// PIECES = 1: the bf16 dual stem (stem_mx_kernel<1, 2, true>): one bf16 weight piece, TWO N-tiles (two models' 32 channels), bf16
//             raw output; 192 fragment registers + 128 accumulator registers per wave, four regions of four MFMAs per tap.
// FAST (float16 pieces only): two changes of the SUMMATION ORDER, so the results are float32-grade equal to the tile kernel's instead of
//   bit-identical (still deterministic):
//   - the remainder MFMAs (w0 of the four float-valued channels x the inputs' low float16 halves: 4 of their 16 K-slots used) of FOUR
//     consecutive taps of a wave share ONE K = 16 chunk — A gathered from the four taps' remainders (two 8-byte LDS reads per lane),
//     B built once per launch: 12 instead of 48 such MFMAs per wave and tile, 8 instead of 16 fragment reads per four taps;
//   - tap 48, which the tile kernel gives to wave 3 for all four M-tiles, is split by M-tile: wave w multiplies it for M-tile w
//     (five MFMAs): every wave issues 209 MFMAs per tile instead of 240 / 240 / 240 / 260.
template <int PIECES, bool POOL, bool RAW, bool FAST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void stem_rs_kernel(const StemMXArgs p) {
  static_assert(PIECES == 2 || (PIECES == 1 && !POOL && !FAST), "float16 pieces, or the bf16 dual stem with raw output");
  constexpr bool H = PIECES == 2;                           // float16 pieces (else bf16)
  constexpr int NTL = H ? 1 : 2;                            // N-tiles (32 output channels each)
  constexpr int NFT = H ? 5 : 4;                            // B fragments per tap
  constexpr bool A1AHEAD = H && PNVO_RS_A1AHEAD;            // chunk-1 fragments fetched a tap ahead too
  constexpr int LAGP = PNVO_RS_LAGP, LAGD = PNVO_RS_LAGD;                         // taps between the loads of a staging piece (pixels / granules) and its LDS writes
  constexpr int GROW = PW * 5;                              // 16-byte granules of one-hot depth per patch row (185)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float *red = reinterpret_cast<float *>(lds + RS_RED_OFF);
  float *pb = reinterpret_cast<float *>(lds + RS_PB_OFF);
  float *etab = reinterpret_cast<float *>(lds + RS_ETAB_OFF);

  // tiles of this workgroup: the workgroups of one XCD (id % 8) walk neighbouring tiles at the same time, so halos meet in that L2
  const int ntiles = p.B * p.tiles_x * p.tiles_y;
  const int chunk = (ntiles + 7) >> 3;
  const int per = (int)gridDim.x >> 3;                      // workgroups per XCD
  const int t_first = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  const int t_end = min(((int)(blockIdx.x & 7) + 1) * chunk, ntiles);
  if (t_first >= t_end) return;
  const int nit = (t_end - t_first + per - 1) / per;
  const bool prof = p.prof != nullptr;
  unsigned long long pc[4] = {0, 0, 0, 0};
  auto now = [&]() PNVO_INL -> unsigned long long { return prof ? __builtin_readcyclecounter() : 0ull; };
  if (RAW && tid < 12) etab[tid] = p.edges[tid];

  // ---------------------------------------------------------------- staging: tile-independent lane geometry
  // pixel rounds: patch pixel r * 256 + tid (three full rounds; the nine pixels of round 3 belong to wave 0)
  unsigned pmeta[4];                                        // LDS offset | px << 16 | py << 22
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int pix = min(r * 256 + tid, NPIX - 1);
    const int py = pix / PW, px = pix - py * PW;
    pmeta[r] = (unsigned)(py * ROW + (px & 1) * PAR + (px >> 1) * PITCH) | ((unsigned)px << 16) | ((unsigned)py << 22);
  }
  const bool haslast = tid < NPIX - 3 * 256;
  // One-hot depth of the observation tensors: fetched by 16-BYTE GRANULE of a patch row (37 pixels x 80 B contiguous = 185
  // granules; a wave-instruction covers 1 KiB of consecutive bytes, see stem_ps_kernel).  Waves 0..2 take patch rows w, w + 3,
  // ... (seven each; wave 3 has a tap more instead), three 64-lane sub-rounds per row: the lane geometry is three registers
  // (sub-round s: granule g = 64 s + lane -> pixel g / 5, chunk g % 5), the row is a compile-time constant.  Round q = 3 j + s is
  // row w + 3 j; one row per tap.
  unsigned gm[RAW ? 1 : 3];                                 // LDS offset inside the row | px << 16 | (granule exists) << 31
  if (!RAW) {
#pragma unroll
    for (int sr = 0; sr < 3; ++sr) {
      const int g = sr * 64 + lane;
      const bool ex = g < GROW;
      const int gg = ex ? g : 0;
      const int px = gg / 5, c = gg - px * 5;
      gm[sr] = (unsigned)((px & 1) * PAR + (px >> 1) * PITCH + c * 8) | ((unsigned)px << 16) | (ex ? 0x80000000u : 0u);
    }
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  const float *zp = p.zero_page;
  const long fpix = (long)p.H * p.W;

  // registers of the patch pieces in flight (nothing is computed on a loaded value before its store_* — a wave that touches it
  // waits for the load, and there is no second wave on the SIMD to fill the gap)
  f32x4 vr4[4], gdd[21];
  f32x2 vr2[4], vd[4], vt[4];
  unsigned rgbw[4][2];
  float dv[4][2];
  unsigned pflag[4];                                        // bit 0: inside the image, bit 1: not the first pixel of the sample
  unsigned lowbits = 0, bad_depth = 0;
  // Every staging load is a BUFFER load of the staged sample's tensors: out-of-image and absent lanes get an offset beyond the
  // descriptor's range and read 0 (the zero padding after whitening), absent modalities a descriptor of zero records — no address
  // selects on 64-bit pointers, which the compiler turns into branches (a branch ends the tap's scheduling region).
  constexpr unsigned OOB = 0x80000000u;
  int shi = 0, swi = 0;                                     // patch origin of the tile being staged (wave-uniform)
  int st_n = 0, st_ty = 0, st_tx = 0;                       //   ... and its coordinates
  __amdgpu_buffer_rsrc_t r_rgb, r_d, r_dd, r_t;
  auto rsrc = [&](const void *base, long off, long bytes) PNVO_INL {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base ? reinterpret_cast<const char *>(base) + off : reinterpret_cast<const char *>(zp)),
                                             0, base ? (unsigned)bytes : 0u, 0x00020000);
  };
  // coordinates of the tile `per` tiles further on, without dividing again (the walk's stride is fixed: one carry per digit)
  const int per_n = per / (p.tiles_x * p.tiles_y), per_r = per % (p.tiles_x * p.tiles_y), per_y = per_r / p.tiles_x, per_x = per_r % p.tiles_x;
  auto set_stage_coords = [&](int sn_i, int ty, int tx) PNVO_INL {
    const long sn = sn_i;
    shi = 2 * ty * TH - 3;
    swi = 2 * tx * TW - 3;
    st_n = sn_i;
    st_ty = ty;
    st_tx = tx;
    if (RAW) {
      r_rgb = rsrc(p.raw_rgb, sn * fpix * 6, fpix * 6);     // both frames of the sample: [2][H][W][3] uint8
      r_d = rsrc(p.raw_depth, sn * fpix * 8, fpix * 8);     //                            [2][H][W] float32
      r_dd = rsrc(nullptr, 0, 0);
    } else {
      r_rgb = rsrc(p.src[0], sn * fpix * 24, fpix * 24);
      r_d = rsrc(p.src[1], sn * fpix * 8, fpix * 8);
      r_dd = rsrc(p.src[2], sn * fpix * 80, fpix * 80);
    }
    r_t = rsrc(p.src[3], sn * fpix * 8, fpix * 8);
  };
  auto stage_next_tile = [&](bool advance) PNVO_INL {       // advance = false: the same tile again (last iteration)
    int tx = st_tx + (advance ? per_x : 0);
    const int cx = tx >= p.tiles_x ? 1 : 0;
    tx -= cx * p.tiles_x;
    int ty = st_ty + (advance ? per_y + cx : 0);
    const int cy = ty >= p.tiles_y ? 1 : 0;
    ty -= cy * p.tiles_y;
    set_stage_coords(st_n + (advance ? per_n + cy : 0), ty, tx);
  };
  auto set_stage_tile = [&](int t) PNVO_INL {
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const long sn = t / p.tiles_y;
    shi = 2 * ty * TH - 3;
    swi = 2 * tx * TW - 3;
    st_n = (int)sn;
    st_ty = ty;
    st_tx = tx;
    if (RAW) {
      r_rgb = rsrc(p.raw_rgb, sn * fpix * 6, fpix * 6);     // both frames of the sample: [2][H][W][3] uint8
      r_d = rsrc(p.raw_depth, sn * fpix * 8, fpix * 8);     //                            [2][H][W] float32
      r_dd = rsrc(nullptr, 0, 0);
    } else {
      r_rgb = rsrc(p.src[0], sn * fpix * 24, fpix * 24);
      r_d = rsrc(p.src[1], sn * fpix * 8, fpix * 8);
      r_dd = rsrc(p.src[2], sn * fpix * 80, fpix * 80);
    }
    r_t = rsrc(p.src[3], sn * fpix * 8, fpix * 8);
  };
  // (the packed lane geometry is read through an empty asm: the decode stays where it is used — hoisted out of the tile loop, as the
  //  optimiser would, it costs three more live registers per staging piece)
  auto opaque = [&](unsigned v) PNVO_INL {
    asm volatile("" : "+v"(v));
    return v;
  };
  auto load_px = [&](int r, bool exists) PNVO_INL {
    const unsigned pm = opaque(pmeta[r]);
    const int hi = shi + (int)(pm >> 22), wi = swi + (int)((pm >> 16) & 0x3fu);
    // (bitwise: a short-circuit && becomes an exec-mask region, which ends the scheduling region like a branch)
    const bool in = (int)exists & (int)((unsigned)hi < (unsigned)p.H) & (int)((unsigned)wi < (unsigned)p.W);
    const unsigned e = (unsigned)(hi * p.W + wi);
    pflag[r] = (in ? 1u : 0u) | (e > 0u ? 2u : 0u);
    if (RAW) {
      // three bytes at 3 li as ONE unaligned dword that starts a byte early except at the sample's first pixel (never outside)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const unsigned li = (unsigned)f * (unsigned)fpix + e;
        const unsigned boff = 3u * li - ((f > 0 || e > 0u) ? 1u : 0u);
        rgbw[r][f] = __builtin_amdgcn_raw_buffer_load_b32(r_rgb, in ? boff : OOB, 0, 0);
        dv[r][f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_d, in ? li * 4u : OOB, 0, 0));
      }
      vt[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_t, in ? e * 8u : OOB, 0, 0));
    } else {
      const unsigned o24 = in ? e * 24u : OOB, o8 = in ? e * 8u : OOB;
      const f32x2 q0 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_rgb, o24, 0, 0));
      const f32x2 q1 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_rgb, o24 + 8u, 0, 0));
      vr4[r] = f32x4{q0[0], q0[1], q1[0], q1[1]};
      vr2[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_rgb, o24 + 16u, 0, 0));
      vd[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_d, o8, 0, 0));
      vt[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_t, o8, 0, 0));
    }
  };
  // part 1: the rgb K-slots (kept in registers), part 2: depth, top-down view, indicator, remainders and the LDS writes; 0: both
  // (two halves of about twenty instructions: a whole pixel round overflows a scheduling region of four MFMAs)
  unsigned wrgb[3] = {0u, 0u, 0u};
  constexpr float RGBS = H ? 0.00390625f : 1.f;             // float16: rgb * 2^-8 here, 2^8 in the packed weights (exact)
  constexpr unsigned ONE2 = H ? 0x3c003c00u : 0x3f803f80u;  // 1.0 twice in float16 / bf16 (the indicator's two slots)
  constexpr unsigned short ONE1 = H ? 0x3c00 : 0x3f80;
  auto store_px = [&](int r, unsigned buf, bool exists, int part) PNVO_INL {
    if (part != 2) {
      if (RAW) {
        const unsigned x0 = (pflag[r] & 2u) ? rgbw[r][0] >> 8 : rgbw[r][0], x1 = rgbw[r][1] >> 8;
        const float pr = (float)(x0 & 0xffu), pg = (float)((x0 >> 8) & 0xffu), pbl = (float)((x0 >> 16) & 0xffu);
        const float cr = (float)(x1 & 0xffu), cg = (float)((x1 >> 8) & 0xffu), cb = (float)((x1 >> 16) & 0xffu);
        wrgb[0] = pack_pair<H>(pr * RGBS, pg * RGBS);
        wrgb[1] = pack_pair<H>(pbl * RGBS, cr * RGBS);
        wrgb[2] = pack_pair<H>(cg * RGBS, cb * RGBS);
      } else {
        wrgb[0] = pack_pair<H>(vr4[r][0] * RGBS, vr4[r][1] * RGBS);
        wrgb[1] = pack_pair<H>(vr4[r][2] * RGBS, vr4[r][3] * RGBS);
        wrgb[2] = pack_pair<H>(vr2[r][0] * RGBS, vr2[r][1] * RGBS);
        const float f0 = vr4[r][0], f1 = vr4[r][1], f2 = vr4[r][2], f3 = vr4[r][3], f4 = vr2[r][0], f5 = vr2[r][1];
        lowbits |= __builtin_bit_cast(unsigned, f0) | __builtin_bit_cast(unsigned, f1);
        lowbits |= __builtin_bit_cast(unsigned, f2) | __builtin_bit_cast(unsigned, f3);
        lowbits |= __builtin_bit_cast(unsigned, f4) | __builtin_bit_cast(unsigned, f5);
      }
    }
    if (part == 1) return;
    float d0, d1;
    int bidx[2] = {0, 0};
    unsigned bok[2] = {0u, 0u};
    const bool in = (pflag[r] & 1u) != 0u;
    if (RAW) {
      const bool use_d = (p.raw_flags & 1) != 0;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float d = dv[r][f];
        int g = (int)(d * 10.0f);
        g = min(max(g, 0), 9);
        const float lo = etab[g], hi = etab[g + 1];
        bidx[f] = g - (d < lo ? 1 : 0) + (((d >= hi) & (g < 9)) ? 1 : 0);
        bok[f] = ((d >= 0.f) & (d <= 1.f)) ? 1u : 0u;
        bad_depth |= (pflag[r] & 1u) & (bok[f] ^ 1u);
      }
      d0 = use_d ? dv[r][0] : 0.f;
      d1 = use_d ? dv[r][1] : 0.f;
    } else {
      d0 = vd[r][0];
      d1 = vd[r][1];
    }
    const unsigned w13 = pack_pair<H>(d0, d1), w14 = pack_pair<H>(vt[r][0], vt[r][1]), w15 = in ? ONE2 : 0u;
    const unsigned base = buf + (opaque(pmeta[r]) & 0xffffu);
    constexpr unsigned TR = (unsigned)RS_TRASH_OFF;
    if (RAW) {
      *reinterpret_cast<u32x4 *>(lds + (exists ? base : TR)) = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4 *>(lds + (exists ? base + 16u : TR)) = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4 *>(lds + (exists ? base + 32u : TR)) = u32x4{0u, 0u, wrgb[0], wrgb[1]};
      *reinterpret_cast<u32x4 *>(lds + (exists ? base + 48u : TR)) = u32x4{wrgb[2], w13, w14, w15};
      const bool dd = (p.raw_flags & 2) != 0 && in && exists;
      *reinterpret_cast<unsigned short *>(lds + ((dd && bok[0]) ? base + 2u * (unsigned)bidx[0] : TR)) = ONE1;
      *reinterpret_cast<unsigned short *>(lds + ((dd && bok[1]) ? base + 20u + 2u * (unsigned)bidx[1] : TR)) = ONE1;
    } else {                                                 // K-slots 20..31 (rgb, depth, top-down view, indicator): bytes 40..63
      *reinterpret_cast<u32x2 *>(lds + (exists ? base + 40u : TR)) = u32x2{wrgb[0], wrgb[1]};
      *reinterpret_cast<u32x4 *>(lds + (exists ? base + 48u : TR)) = u32x4{wrgb[2], w13, w14, w15};
    }
    if (H) {                                                 // float-valued channels: x = x0 + x1 to 22 bits
      const f16x2 hd = __builtin_bit_cast(f16x2, w13), ht = __builtin_bit_cast(f16x2, w14);
      const unsigned md = pack_f16(d0 - (float)hd[0], d1 - (float)hd[1]);
      const unsigned mt = pack_f16(vt[r][0] - (float)ht[0], vt[r][1] - (float)ht[1]);
      *reinterpret_cast<u32x4 *>(lds + (exists ? base + 64u : TR)) = u32x4{md, mt, 0u, 0u};
    }
  };
  // Granules of the one-hot depth, patch row `row` (compile-time), sub-round sr.  Per tile and sub-round: the vector offset of the
  // lane's granule in patch row 0 (or OOB for a column outside the image / a lane past the row's 185 granules) and its LDS address
  // in row 0 (absent lanes: the 32 bytes of padding behind a patch row); a row adds a scalar to the first and an immediate to the
  // second.  Rows outside the image are out of the descriptor's range by themselves (or land in a neighbouring row's columns that
  // are forced out of range); OOB + a row offset stays out of range.
  unsigned ddv[3], ddl[3];
  auto set_dd_tile = [&](unsigned buf) PNVO_INL {
    if (RAW) return;
#pragma unroll
    for (int sr = 0; sr < 3; ++sr) {
      const unsigned g = opaque(gm[sr]);
      const int wi = swi + (int)((g >> 16) & 0x3fu);
      const bool ok = (int)((int)g < 0) & (int)((unsigned)wi < (unsigned)p.W);
      // (everything in the vector offset: the range check does not see a scalar offset)
      ddv[sr] = ok ? (unsigned)((shi * p.W + swi) * 80 + sr * 1024) + lane16 : OOB;
      ddl[sr] = buf + ((int)g < 0 ? (g & 0xffffu) : (unsigned)(2 * PAR));
    }
  };
  auto load_dd = [&](int slot, int row, int sr) PNVO_INL {
    gdd[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_dd, ddv[sr] + (unsigned)(row * p.W * 80), 0, 0));
  };
  auto store_dd = [&](int slot, int row, int sr) PNVO_INL {
    const f32x4 v = gdd[slot];
    lowbits |= __builtin_bit_cast(unsigned, v[0]) | __builtin_bit_cast(unsigned, v[1]);
    lowbits |= __builtin_bit_cast(unsigned, v[2]) | __builtin_bit_cast(unsigned, v[3]);
    *reinterpret_cast<u32x2 *>(lds + ddl[sr] + row * ROW) = u32x2{pack_pair<H>(v[0], v[1]), pack_pair<H>(v[2], v[3])};
  };

  // lane geometry of the A fragments and of the epilogue
  const int arr = (lane & 31) >> 4, ac = lane & 15, ah = lane >> 5;
  const unsigned baseA0 = (unsigned)(2 * arr * ROW + ac * PITCH + ah * 16);
  const unsigned baseX0 = (unsigned)(2 * arr * ROW + ac * PITCH + 64);
  const int rr16 = lane >> 5;
  const float oscale = !H ? 1.f : p.oscale_ptr != nullptr ? *p.oscale_ptr : p.oscale;
  const int co = p.y_coff[0] + (lane & 31);
  const float sgn = POOL ? (p.pool_gamma[co] < 0.f ? -1.f : 1.f) : 1.f;

  // Everything below is compiled once per wave of the workgroup (WV = 0..3): the wave's taps, patch rows and M-tile are then
  // compile-time constants — every LDS address of the K loop is one register plus an immediate, no scalar arithmetic per tap.
  auto body = [&](auto wv_c) PNVO_INL {
    constexpr int WV = decltype(wv_c)::value;
    // patch rows of the granule fetch: wave 3 has a tap more -> rows WV, WV + 3, ... on waves 0..2; FAST (equal MFMA counts): rows WV,
    // WV + 4, ... on all four, the twenty-first row on wave 1 (wave 0 has the nine left-over pixels)
    constexpr int NROWS = FAST ? (WV == 1 ? 6 : 5) : WV == 3 ? 0 : 7;
    // staging placement: granule rounds per tap (loads from tap 0 on, stores LAGD taps later), taps of the three pixel rounds' loads
    // (stores LAGP taps later).  FAST spreads both over the whole K loop: two rounds per tap, pixel rounds at taps 1, 4, 7.
    constexpr int DPT = FAST ? 2 : 3, PX0 = FAST ? 1 : 0, PXS = FAST ? 3 : 2;
    auto px_store_tap = [](int i) constexpr { return i >= LAGP + PX0 && (i - LAGP - PX0) % PXS == 0 && (i - LAGP - PX0) / PXS < 3; };
    auto dd_row = [](int j) constexpr { return FAST ? (j < 5 ? WV + 4 * j : 20) : WV + 3 * j; };
    constexpr int RD = RAW ? 0 : 3 * NROWS;                 // granule rounds of this wave
    constexpr int NT = (WV == 3 && !FAST) ? 13 : 12;        // taps: WV, WV + 4, ..., WV + 44 (+ tap 48 on wave 3; FAST: by M-tile)
    auto tap_of = [](int i) constexpr { return i < 12 ? WV + 4 * i : 48; };
    auto tap_off = [](int t) constexpr { return (t / 7) * ROW + ((t % 7) & 1) * PAR + ((t % 7) >> 1) * PITCH; };

    // ---- prologue: the first patch, all at once
    __syncthreads();                                        // (the edge table)
    set_stage_tile(t_first);
    set_dd_tile(0u);
#pragma unroll
    for (int r = 0; r < 3; ++r) load_px(r, true);
    if (WV == 0) load_px(3, haslast);
#pragma unroll
    for (int q = 0; q < RD; ++q) load_dd(q, dd_row(q / 3), q % 3);
#pragma unroll
    for (int r = 0; r < 3; ++r) store_px(r, 0u, true, 0);
    if (WV == 0) store_px(3, 0u, haslast, 0);
#pragma unroll
    for (int q = 0; q < RD; ++q) store_dd(q, dd_row(q / 3), q % 3);
    __syncthreads();

    // ---- the resident B operand (fetched AFTER the first patch is staged: the all-at-once prologue needs ~150 registers of its
    // own).  The fifth fragment of a tap — w0 of the four float-valued channels against the remainders — has only its first 8
    // bytes per lane non-zero by construction (pack_stem_mx_weight_h): two registers instead of four.
    u32x4 bres[NT][4];                                      // float16: (w0, w1) x (chunk 0, 1); bf16: (chunk 0, 1) x (N-tile 0, 1)
    u32x2 bxr[(H && !FAST) ? NT : 1];
    u32x4 bxs[FAST ? 3 : 1], b48[FAST ? 4 : 1];             // FAST: remainder fragments of taps 4 g .. 4 g + 3; tap 48's fragments
    u32x2 b48x = {0u, 0u};
    {
      const u32x4 *wp = reinterpret_cast<const u32x4 *>(p.wpk) + lane;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int f = 0; f < 4; ++f) bres[i][f] = wp[(tap_of(i) * NFT + f) * 64];
        if (H && !FAST) bxr[i] = *reinterpret_cast<const u32x2 *>(wp + (tap_of(i) * NFT + 4) * 64);
      }
      if (FAST) {
        // K-slots 0-3 / 4-7 of the shared chunk (lanes 0-31) = the remainder weights of taps 4 g / 4 g + 1, slots 8-11 / 12-15 (lanes
        // 32-63) = taps 4 g + 2 / 4 g + 3: the first 8 bytes of those taps' fifth fragment at lane (n = lane & 31)
        const u32x4 *wq = reinterpret_cast<const u32x4 *>(p.wpk) + (lane & 31);
        const bool kh1 = lane >= 32;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const u32x2 lo = *reinterpret_cast<const u32x2 *>(wq + ((kh1 ? tap_of(4 * g + 2) : tap_of(4 * g)) * NFT + 4) * 64);
          const u32x2 hi = *reinterpret_cast<const u32x2 *>(wq + ((kh1 ? tap_of(4 * g + 3) : tap_of(4 * g + 1)) * NFT + 4) * 64);
          bxs[g] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) b48[f] = wp[(48 * NFT + f) * 64];
        b48x = *reinterpret_cast<const u32x2 *>(wp + (48 * NFT + 4) * 64);
      }
      // Register classes: a wave's 512 registers are 256 VGPRs + 256 AGPRs, and only matrix instructions, loads and stores reach
      // the second half.  Pinned there, a fragment is read by its MFMAs in place; left to the allocator it is parked there and
      // copied back (four v_accvgpr_read per fragment and tap, ~240 issue slots per tile).  Accumulators (64) + PNVO_RS_NPIN taps fit.
      constexpr int NPIN = H ? PNVO_RS_NPIN : 7;              // (bf16 dual stem: 128 accumulator registers leave room for seven taps)
#pragma unroll
      for (int i = 0; i < (NT < NPIN ? NT : NPIN); ++i) {
#pragma unroll
        for (int f = 0; f < 4; ++f) asm volatile("" : "+a"(bres[i][f]));
        if (H && !FAST) asm volatile("" : "+a"(bxr[i]));
      }
      if (FAST) {
#pragma unroll
        for (int g = 0; g < 3; ++g) asm volatile("" : "+a"(bxs[g]));
      }
    }

    // ---- the epilogue of a tile in pieces (POOL: they run between the MFMAs of the NEXT tile, see `epi` below): K-split sum in wave
    // order, un-scale, GroupNorm partial sums, pooled keys.  e_*: the tile they work on; ebuf: the buffer its exchange is in.
    int e_n = 0, e_ho0 = 0, e_wo0 = 0, e_slot = 0;
    bool e_valid = false;
    unsigned ebuf = 0u;
    // Every piece that reads LDS is split into its reads and, one region later, the arithmetic on them: with one wave per SIMD a
    // value used right behind its ds_read costs the read's whole latency.
    f32x4 xq[4], tq4;
    float s1 = 0.f, s2 = 0.f, cm[8], pl[6], rs9[3];
    f32x2 rsv[4];
    int rr16v = 0, lc = 0, colv = 0;
    auto epi_begin = [&]() PNVO_INL {                       // per-tile lane values (opaque: no hoisted per-pixel addresses)
      rr16v = (int)opaque((unsigned)rr16);
      lc = (int)opaque((unsigned)(lane & 31));
      colv = p.Wo - e_wo0 - 4 * rr16v;                      // columns of this lane's pixel group that exist
      s1 = 0.f;
      s2 = 0.f;
    };
    auto XAl = [&](int rq) PNVO_INL {                       // accumulator quad rq of M-tile WV: the four waves' partials
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        xq[s4] = *reinterpret_cast<const f32x4 *>(lds + ebuf + (((WV * 4 + s4) * 4 + rq) * 64 + lane) * 16);
    };
    auto XAs = [&]() PNVO_INL {                             //   ... summed in wave order, power-of-two scale undone (exact)
#pragma unroll
      for (int e = 0; e < 4; ++e) tq4[e] = (((xq[0][e] + xq[1][e]) + xq[2][e]) + xq[3][e]) * oscale;
    };
    auto XB = [&](int rq) PNVO_INL {                        // its four pixels: pooling scratch (sgn x, -inf outside) + partial sums
      const bool rowok = e_ho0 + 2 * WV + (rq >> 1) < p.Ho;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int colc = e + 8 * (rq & 1);                  // pixel (row 2 WV + (rq >> 1), column colc + 4 rr16) of the tile
        const bool ok = (int)rowok & (int)(colc < colv);
        const float v = ok ? tq4[e] : 0.f;
        pb[((2 * WV + (rq >> 1)) * 16 + colc + 4 * rr16v) * 33 + lc] = ok ? sgn * tq4[e] : -__builtin_inff();
        s1 += v;
        s2 = __builtin_fmaf(v, v, s2);
      }
    };
    auto XC = [&]() PNVO_INL {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      // (both halves of the wave hold the sums now and write the same two words)
      *reinterpret_cast<f32x2 *>(&red[(WV * 32 + lc) * 2]) = f32x2{s1, s2};
    };
    auto ebar = [&]() PNVO_INL {                            // workgroup barrier over LDS only (global loads stay in flight)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // MaxPool2d(3, 2, 1) on order-preserving integer keys of sgn(gamma) * x (see stem_mx_kernel): lane = (pooled column pj, channel);
    // every key goes out as an integer atomic max (exact, order-free), absent ones as the identity at a clamped address — no branch
    auto PAl = [&](int j) PNVO_INL {                        // tile rows 2 j, 2 j + 1: the three columns of the lane's window
      const int pj = 2 * WV + rr16v;
      const int c0 = pj > 0 ? 2 * pj - 1 : 0, c1 = 2 * pj;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        pl[3 * d] = pb[((2 * j + d) * 16 + c0) * 33 + lc];
        pl[3 * d + 1] = pb[((2 * j + d) * 16 + c1) * 33 + lc];
        pl[3 * d + 2] = pb[((2 * j + d) * 16 + c1 + 1) * 33 + lc];
      }
    };
    auto PAm = [&](int j) PNVO_INL {
      cm[2 * j] = fmaxf(fmaxf(pl[0], pl[1]), pl[2]);
      cm[2 * j + 1] = fmaxf(fmaxf(pl[3], pl[4]), pl[5]);
    };
    // (per tile: the sample's key plane, the lane's column offset and validity; per key a row offset)
    int *e_pool = p.pool;
    unsigned e_joff = 0u, e_j9off = 0u;
    bool e_jok = false, e_j9ok = false;
    auto emit_begin = [&]() PNVO_INL {
      const int pj = 2 * WV + rr16v, J = (e_wo0 >> 1) + pj, J9 = (e_wo0 >> 1) + 8;
      e_pool = p.pool + (long)e_n * p.Hp * p.Wp * p.y_cstride + p.y_coff[0];   // wave-uniform
      e_joff = (unsigned)(min(J, p.Wp - 1) * p.y_cstride + lc);
      e_j9off = (unsigned)(min(J9, p.Wp - 1) * p.y_cstride + lc);
      e_jok = (int)e_valid & (int)(J < p.Wp);
      e_j9ok = (int)e_valid & (int)(J9 < p.Wp) & (int)(pj < 5);
    };
    auto emit = [&](int I, unsigned joff, float mx, bool valid) PNVO_INL {
      int key = __builtin_bit_cast(int, mx);
      key = key >= 0 ? key : key ^ 0x7fffffff;
      const bool ok = (int)valid & (int)(I < p.Hp);
      atomicMax(e_pool + (long)(min(I, p.Hp - 1) * p.Wp * p.y_cstride) + joff, ok ? key : STEM_POOL_INIT);
    };
    auto PE = [&](int k) PNVO_INL {                         // pooled row k of the 5 x 9 pooled pixels the tile touches
      const float mx = k == 0 ? fmaxf(cm[0], cm[1]) : k == 4 ? cm[7] : fmaxf(fmaxf(cm[2 * k - 1], cm[2 * k]), cm[2 * k + 1]);
      emit((e_ho0 >> 1) + k, e_joff, mx, e_jok);
    };
    auto P9l = [&]() PNVO_INL {                             // ninth pooled column (tile column 15): lanes with pj < 5 take pooled row pj
      const int pi = 2 * WV + rr16v;
#pragma unroll
      for (int dr = -1; dr <= 1; ++dr) rs9[dr + 1] = pb[(min(max(2 * pi + dr, 0), 7) * 16 + 15) * 33 + lc];
    };
    auto P9e = [&]() PNVO_INL {
      const int pi = 2 * WV + rr16v;
      float mx = -__builtin_inff();
#pragma unroll
      for (int dr = -1; dr <= 1; ++dr) {
        const int lr = 2 * pi + dr;
        mx = fmaxf(mx, ((int)(lr >= 0) & (int)(lr < 8)) ? rs9[dr + 1] : -__builtin_inff());
      }
      emit((e_ho0 >> 1) + pi, e_j9off, mx, e_j9ok);
    };
    auto PSl = [&]() PNVO_INL {                             // GroupNorm partial sums of the tile, wave order (as stem_mx_kernel)
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) rsv[w4] = *reinterpret_cast<const f32x2 *>(&red[(w4 * 32 + lc) * 2]);
    };
    auto PSs = [&]() PNVO_INL {
      float a1s = 0.f, a2s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        a1s += rsv[w4][0];
        a2s += rsv[w4][1];
      }
      float *dst = p.stats[0] + (((long)e_n * p.slots + e_slot) * p.stats_cstride + p.y_coff[0]) * 2;   // wave-uniform
      *reinterpret_cast<f32x2 *>(dst + 2 * lc) = f32x2{a1s, a2s};   // (both halves of the wave: the same two words)
    };
    // ---- raw-output form (!POOL): the K-split sum is taken in the serial section behind the K loop (the exchange buffer is free
    // again before the next K loop starts); the pieces store the pixels — buffer stores, an out-of-range offset for pixels outside
    // the output — and add up the GroupNorm partial sums.
    constexpr int ESZ = H ? 4 : 2;                          // bytes per output element (float32 / bf16)
    f32x16 totv[NTL];
    float es1[NTL], es2[NTL];
    __amdgpu_buffer_rsrc_t r_y[NTL];
    unsigned e_yoff = 0u;                                   // pixel (row 2 WV, column 4 rr16) of the tile, channel lc: byte offset in the sample
    auto out_begin = [&]() PNVO_INL {
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) {
        const long plane = (long)p.Ho * p.Wo * p.y_cstride * ESZ;
        r_y[nt] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.y[nt]) + (long)e_n * plane + (long)p.y_coff[nt] * ESZ, 0,
                                                    (unsigned)plane, 0x00020000);
        es1[nt] = 0.f;
        es2[nt] = 0.f;
      }
      e_yoff = (unsigned)((((e_ho0 + 2 * WV) * p.Wo + e_wo0 + 4 * rr16v) * p.y_cstride + lc) * ESZ);
    };
    auto XS = [&](int nt, int rq) PNVO_INL {                // accumulator quad rq: four pixels of row 2 WV + (rq >> 1)
      const bool rowok = (int)e_valid & (int)(e_ho0 + 2 * WV + (rq >> 1) < p.Ho);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int colc = e + 8 * (rq & 1);
        const bool ok = (int)rowok & (int)(colc < colv);
        const float v = ok ? totv[nt][4 * rq + e] : 0.f;
        const unsigned off = e_yoff + (unsigned)(((rq >> 1) * p.Wo + colc) * p.y_cstride * ESZ);
        if (H)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_y[nt], ok ? off : OOB, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v), r_y[nt], ok ? off : OOB, 0, 0);
        es1[nt] += v;
        es2[nt] = __builtin_fmaf(v, v, es2[nt]);
      }
    };
    auto XCn = [&](int nt) PNVO_INL {
      const float a = es1[nt] + __shfl_xor(es1[nt], 32), b = es2[nt] + __shfl_xor(es2[nt], 32);
      *reinterpret_cast<f32x2 *>(&red[((WV * NTL + nt) * 32 + lc) * 2]) = f32x2{a, b};   // (both halves of the wave: the same words)
    };
    f32x2 rsn[NTL][4];
    auto PSnl = [&](int nt) PNVO_INL {
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) rsn[nt][w4] = *reinterpret_cast<const f32x2 *>(&red[((w4 * NTL + nt) * 32 + lc) * 2]);
    };
    auto PSns = [&](int nt) PNVO_INL {
      float a1s = 0.f, a2s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        a1s += rsn[nt][w4][0];
        a2s += rsn[nt][w4][1];
      }
      float *dst = p.stats[nt] + (((long)e_n * p.slots + e_slot) * p.stats_cstride + p.y_coff[nt]) * 2;   // wave-uniform
      *reinterpret_cast<f32x2 *>(dst + 2 * lc) = f32x2{a1s, a2s};
    };
    // which piece runs in which region of the next tile's K loop (tap i, region rg: 0 and 2 carry only fragment reads of their
    // own, 3 the granule stores, 4 the pixel stores of taps 3, 5, 7).  Taps 0-2: the exchange is read before the barrier behind
    // tap 2 — from tap 3 on the stager overwrites that buffer.
    auto epi = [&](int i, int rg) PNVO_INL {
      if (PNVO_RS_ABL & 4) return;
      const int k = 5 * i + rg;
      if (!H) {                                             // four regions per tap: k = 4 i + rg (rg 1 carries the loads)
        const int k4 = 4 * i + rg;
        if (k4 == 0) { epi_begin(); out_begin(); }
        if (k4 == 2) XS(0, 0);
        if (k4 == 3) XS(0, 1);
        if (k4 == 4) XS(0, 2);
        if (k4 == 6) XS(0, 3);
        if (k4 == 7) XS(1, 0);
        if (k4 == 8) XS(1, 1);
        if (k4 == 10) XS(1, 2);
        if (k4 == 11) XS(1, 3);
        if (k4 == 12) { XCn(0); XCn(1); }
        if (k4 == 16 && WV == 1) PSnl(0);
        if (k4 == 16 && WV == 2) PSnl(1);
        if (k4 == 18 && WV == 1) PSns(0);
        if (k4 == 18 && WV == 2) PSns(1);
        return;
      }
      if (!POOL) {
        if (k == 0) { epi_begin(); out_begin(); }
        if (k == 2) XS(0, 0);
        if (k == 3) XS(0, 1);
        if (k == 4) XS(0, 2);
        if (k == 5) XS(0, 3);
        if (k == 7) XCn(0);
        if (k == 15 && WV == 1) PSnl(0);
        if (k == 17 && WV == 1) PSns(0);
        return;
      }
      if (k == 0) epi_begin();
      if ((PNVO_RS_ABL & 8) && k < 15) return;            // (ablation: no exchange sum / scratch pieces)
      if ((PNVO_RS_ABL & 16) && k >= 15) return;          // (ablation: no pooling / key pieces)
      if (k == 0) XAl(0);
      if (k == 2) { XAs(); XAl(1); }
      if (k == 3) XB(0);
      if (k == 4) { XAs(); XAl(2); }
      if (k == 5) XB(1);
      if (k == 7) { XAs(); XAl(3); }
      if (k == 8) XB(2);
      if (k == 9) XAs();
      if (k == 10) XB(3);
      if (k == 12) XC();
      if (k == 15) PAl(0);
      if (k == 17) { PAm(0); PAl(1); }
      if (k == 20) { PAm(1); PAl(2); }
      if (k == 22) { PAm(2); PAl(3); }
      if (k == 24) { PAm(3); emit_begin(); }
      if (k == 25) PE(0);
      if (k == 27) PE(1);
      if (k == 30) PE(2);
      if (k == 32) PE(3);
      if (k == 35) { PE(4); P9l(); }
      if (k == 37) P9e();
      if (k == 40 && WV == 1) PSl();
      if (k == 42 && WV == 1) PSs();
    };
    auto epi_serial = [&]() PNVO_INL {                      // the same pieces one after the other (after the last tile)
      epi_begin();
      if (!POOL) {
        out_begin();
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) XS(nt, rq);
          XCn(nt);
        }
        ebar();
        if (WV == 1) {
          PSnl(0);
          PSns(0);
        }
        if (NTL > 1 && WV == 2) {
          PSnl(NTL - 1);
          PSns(NTL - 1);
        }
        return;
      }
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        XAl(rq);
        XAs();
        XB(rq);
      }
      XC();
      ebar();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        PAl(j);
        PAm(j);
      }
      emit_begin();
#pragma unroll
      for (int k = 0; k < 5; ++k) PE(k);
      P9l();
      P9e();
      if (WV == 1) {
        PSl();
        PSs();
      }
    };

    int c_n = st_n, c_ty = st_ty, c_tx = st_tx;             // the current tile (staged by the prologue)
    // (first tile: the pieces run on nothing — keys off, the partial sums land in this tile's own slot and are overwritten by its
    //  real epilogue one iteration later)
    e_n = c_n;
    e_ho0 = c_ty * TH;
    e_wo0 = c_tx * TW;
    e_slot = c_ty * p.tiles_x + c_tx;
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
      const unsigned buf = (unsigned)(it & 1) * RS_BUF, obuf = buf ^ (unsigned)RS_BUF;
      // the tile staged during this K loop: the next one (the last iteration re-stages its own tile into the idle buffer)
      stage_next_tile(it + 1 < nit);
      set_dd_tile(obuf);
      ebuf = obuf;                                          // POOL: the previous tile's exchange (first tile: nothing valid, keys off)
      const unsigned long long t0 = now();

      // ---------------------------------------------------------- K loop of tile it, staging of the next tile between its MFMAs
      f32x16 acc[NTL][4];                                   // (never zeroed: the first MFMA of a tile takes the constant 0 as its C operand)
      // (opaque: hoisted out of the tile loop, the two buffers' bases would be separate live registers)
      const unsigned baseA = opaque(baseA0 + buf), baseX = opaque(baseX0 + buf);
      // A fragments: chunk 0 of a tap is fetched during the previous tap (two register sets in turn), chunk 1 and the remainders
      // in the tap's first region, eight MFMAs before their first use — with one wave per SIMD an LDS read waited for right before
      // its MFMA is ~100 idle cycles of the matrix pipe.
      u32x4 a0[2][4], a1s[2][4], ax[4];
      u32x4 t48[3];                                         // FAST: tap 48's fragments of M-tile WV
      const bool xkh = lane >= 32;
      auto tofs = [&](int i, int m) constexpr { return tap_off(tap_of(i)) + m * 4 * ROW; };
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        a0[0][m] = *reinterpret_cast<const u32x4 *>(lds + baseA + tofs(0, m));
        if (A1AHEAD) a1s[0][m] = *reinterpret_cast<const u32x4 *>(lds + baseA + tofs(0, m) + 32);
      }
      // One tap = five groups of four MFMAs (one B fragment x the four M-tiles), each with its share of the other work in source
      // order; groups 0+1 and 2+3 form one scheduling region each (measured: a region per group 10.1 k cycles per tile, pairs 9.7 k,
      // the whole tap 10.7 k); inside a region every MFMA is followed by at most eight other instructions.
      auto region_end = [&](bool hard = true) PNVO_INL {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, PNVO_RS_GDS, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, PNVO_RS_GVM, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, PNVO_RS_GVA, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        if (hard) __builtin_amdgcn_sched_barrier(0);
      };
      auto mfma4 = [&](const u32x4 *aq, const u32x4 bq, bool first = false, int nt = 0) PNVO_INL {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[nt][m] = H ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aq[m]), __builtin_bit_cast(f16x8, bq),
                                                                  first ? zero : acc[nt][m], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq[m]), __builtin_bit_cast(bf16x8, bq),
                                                                   first ? zero : acc[nt][m], 0, 0, 0);
      };
      __builtin_amdgcn_sched_barrier(0);
      static_for<NT>([&](auto ic) PNVO_INL {
        constexpr int i = decltype(ic)::value;
        const bool st = !(PNVO_RS_ABL & 2), sd = !RAW && !(PNVO_RS_ABL & 1);
        if (H) {
          // the MFMA order of stem_mx_kernel: chunk 0 x {w0, w1}, chunk 1 x {w0, w1}, remainders x w0; M-tiles innermost
          // -- region 0: chunk 0 x w0 | this tap's chunk-1 and remainder fragments
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            if (!A1AHEAD) a1s[i & 1][m] = *reinterpret_cast<const u32x4 *>(lds + baseA + tofs(i, m) + 32);
            if (!FAST) ax[m] = *reinterpret_cast<const u32x4 *>(lds + baseX + tofs(i, m));
          }
          if (FAST && i > 0 && (i - 1) % 4 != 3) {            // (the previous tap had no fifth region: its pieces ride here)
            epi(i - 1, 4);
            if (st && px_store_tap(i - 1)) store_px((i - 1 - LAGP - PX0) / PXS, obuf, true, 2);
          }
          epi(i, 0);
          mfma4(a0[i & 1], bres[i][0], i == 0);
          region_end(!(PNVO_RS_MERGE & 1));
          // -- region 1: chunk 0 x w1 | loads of the next patch: two granule rounds, a pixel round on taps 0, 2, 4
          if (sd) {
#pragma unroll
            for (int q = DPT * i; q < DPT * i + DPT; ++q)
              if (q < RD) load_dd(q, dd_row(q / 3), q % 3);
          }
          if (st && i >= PX0 && (i - PX0) % PXS == 0 && (i - PX0) / PXS < 3) load_px((i - PX0) / PXS, true);
          if (st && WV == 0 && i == 8) load_px(3, haslast);
          mfma4(a0[i & 1], bres[i][2]);
          region_end(!(PNVO_RS_MERGE & 2));
          // -- region 2: chunk 1 x w0 | the next tap's chunk-0 fragments
          if (i + 1 < NT) {
#pragma unroll
            for (int m = 0; m < 4; ++m) a0[(i + 1) & 1][m] = *reinterpret_cast<const u32x4 *>(lds + baseA + tofs(i + 1, m));
          }
          if (FAST && i % 4 == 3) {
            // the shared remainder chunk of taps i - 3 .. i: lanes 0-31 read the remainders of taps i - 3 | i - 2, lanes 32-63 of i - 1 | i
            const unsigned xa = baseX + (unsigned)(xkh ? tofs(i - 1, 0) : tofs(i - 3, 0)), xb = baseX + (unsigned)(xkh ? tofs(i, 0) : tofs(i - 2, 0));
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const u32x2 lo = *reinterpret_cast<const u32x2 *>(lds + xa + m * 4 * ROW), hi = *reinterpret_cast<const u32x2 *>(lds + xb + m * 4 * ROW);
              ax[m] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
          }
          if (FAST && i == 11) {                              // tap 48, M-tile WV
            t48[0] = *reinterpret_cast<const u32x4 *>(lds + baseA + tap_off(48) + WV * 4 * ROW);
            t48[1] = *reinterpret_cast<const u32x4 *>(lds + baseA + tap_off(48) + WV * 4 * ROW + 32);
            t48[2] = *reinterpret_cast<const u32x4 *>(lds + baseX + tap_off(48) + WV * 4 * ROW);
          }
          epi(i, 2);
          mfma4(a1s[i & 1], bres[i][1]);
          region_end(!(PNVO_RS_MERGE & 1));
          // -- region 3: chunk 1 x w1 | conversion + LDS writes of the granule rounds loaded LAGD taps ago
          if (sd && i >= LAGD) {
#pragma unroll
            for (int q = DPT * (i - LAGD); q < DPT * (i - LAGD) + DPT; ++q)
              if (q < RD) store_dd(q, dd_row(q / 3), q % 3);
          }
          if (st && px_store_tap(i)) store_px((i - LAGP - PX0) / PXS, obuf, true, 1);
          if (A1AHEAD && i + 1 < NT) {
#pragma unroll
            for (int m = 0; m < 4; ++m) a1s[(i + 1) & 1][m] = *reinterpret_cast<const u32x4 *>(lds + baseA + tofs(i + 1, m) + 32);
          }
          epi(i, 3);
          mfma4(a1s[i & 1], bres[i][3]);
          region_end(!(PNVO_RS_MERGE & 4));
          // -- region 4: remainders x w0 | conversion + LDS writes of the pixel round loaded LAGP taps ago
          //    (FAST: only behind every fourth tap, for the four of them)
          if (!FAST || i % 4 == 3) {
            if (st && px_store_tap(i)) store_px((i - LAGP - PX0) / PXS, obuf, true, 2);
            if (st && WV == 0 && i == 11) store_px(3, obuf, haslast, 0);
            epi(i, 4);
            mfma4(ax, FAST ? bxs[FAST ? i / 4 : 0] : u32x4{bxr[FAST ? 0 : i][0], bxr[FAST ? 0 : i][1], 0u, 0u});
            region_end();
          }
          if (FAST && i == 11) {
            // tap 48 for M-tile WV: five MFMAs in the tile kernel's order (chunk 0 x {w0, w1}, chunk 1 x {w0, w1}, remainders x w0)
            const u32x4 bq5[5] = {b48[0], b48[2], b48[1], b48[3], u32x4{b48x[0], b48x[1], 0u, 0u}};
#pragma unroll
            for (int g5 = 0; g5 < 5; ++g5)
              acc[0][WV] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, t48[g5 < 2 ? 0 : g5 < 4 ? 1 : 2]),
                                                                  __builtin_bit_cast(f16x8, bq5[g5]), acc[0][WV], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          // bf16 dual stem — the MFMA order of stem_mx_kernel<1, 2>: chunk 0 x {N-tile 0, 1}, chunk 1 x {N-tile 0, 1}; M-tiles innermost
          // -- region 0: chunk 0, N-tile 0 | this tap's chunk-1 fragments
#pragma unroll
          for (int m = 0; m < 4; ++m) a1s[i & 1][m] = *reinterpret_cast<const u32x4 *>(lds + baseA + tofs(i, m) + 32);
          epi(i, 0);
          mfma4(a0[i & 1], bres[i][0], i == 0, 0);
          region_end(!(PNVO_RS_MERGE & 1));
          // -- region 1: chunk 0, N-tile 1 | loads of the next patch
          if (sd) {
#pragma unroll
            for (int q = 3 * i; q < 3 * i + 3; ++q)
              if (q < RD) load_dd(q, dd_row(q / 3), q % 3);
          }
          if (st && i % 2 == 0 && i < 6) load_px(i / 2, true);
          if (st && WV == 0 && i == 8) load_px(3, haslast);
          mfma4(a0[i & 1], bres[i][1], i == 0, 1);
          region_end();
          // -- region 2: chunk 1, N-tile 0 | the next tap's chunk-0 fragments, second half of a pixel round's store
          if (i + 1 < NT) {
#pragma unroll
            for (int m = 0; m < 4; ++m) a0[(i + 1) & 1][m] = *reinterpret_cast<const u32x4 *>(lds + baseA + tofs(i + 1, m));
          }
          if (st && i >= LAGP + 1 && (i - LAGP - 1) % 2 == 0 && i - LAGP - 1 < 6) store_px((i - LAGP - 1) / 2, obuf, true, 2);
          if (st && WV == 0 && i == 11) store_px(3, obuf, haslast, 0);
          epi(i, 2);
          mfma4(a1s[i & 1], bres[i][2], false, 0);
          region_end(!(PNVO_RS_MERGE & 1));
          // -- region 3: chunk 1, N-tile 1 | granule stores, first half of a pixel round's store
          if (sd && i >= LAGD) {
#pragma unroll
            for (int q = 3 * (i - LAGD); q < 3 * (i - LAGD) + 3; ++q)
              if (q < RD) store_dd(q, dd_row(q / 3), q % 3);
          }
          if (st && i >= LAGP && (i - LAGP) % 2 == 0 && i - LAGP < 6) store_px((i - LAGP) / 2, obuf, true, 1);
          epi(i, 3);
          mfma4(a1s[i & 1], bres[i][3], false, 1);
          region_end();
        }
        if (i == (H ? 2 : 3) && !(PNVO_RS_ABL & 32)) ebar();                          // scratch + partial sums of the previous tile complete; its exchange is read
      });
      const unsigned long long t1 = now();
      __syncthreads();                                      // every wave has left patch(it)
      const unsigned long long t2 = now();
      const int n = c_n, ty = c_ty, tx = c_tx;
      if (POOL) {
        // K-split exchange through the buffer just consumed; the rest of this tile's epilogue rides in the next tile's K loop
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<f32x4 *>(lds + buf + (((m * 4 + WV) * 4 + rq) * 64 + lane) * 16) =
                f32x4{acc[0][m][4 * rq], acc[0][m][4 * rq + 1], acc[0][m][4 * rq + 2], acc[0][m][4 * rq + 3]};
        __syncthreads();                                    // exchange(it) complete; patch(it + 1) complete
        e_n = n;
        e_ho0 = ty * TH;
        e_wo0 = tx * TW;
        e_slot = ty * p.tiles_x + tx;
        e_valid = true;
      } else {
        // raw-output form: exchange and K-split sum here (fixed order), stores and partial sums ride in the next tile's K loop
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
          if (nt > 0) __syncthreads();                      // the previous N-tile's partials have been read
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
              *reinterpret_cast<f32x4 *>(lds + buf + (((m * 4 + WV) * 4 + rq) * 64 + lane) * 16) =
                  f32x4{acc[nt][m][4 * rq], acc[nt][m][4 * rq + 1], acc[nt][m][4 * rq + 2], acc[nt][m][4 * rq + 3]};
          __syncthreads();                                  // exchange(it) complete; patch(it + 1) complete
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
              const f32x4 tq = *reinterpret_cast<const f32x4 *>(lds + buf + (((WV * 4 + s4) * 4 + rq) * 64 + lane) * 16);
#pragma unroll
              for (int e = 0; e < 4; ++e) totv[nt][4 * rq + e] = s4 == 0 ? tq[e] : totv[nt][4 * rq + e] + tq[e];
            }
          if (H) {
#pragma unroll
            for (int r = 0; r < 16; ++r) totv[nt][r] *= oscale;   // undo the weights' power-of-two scale (exact)
          }
        }
        e_n = n;
        e_ho0 = ty * TH;
        e_wo0 = tx * TW;
        e_slot = ty * p.tiles_x + tx;
        e_valid = true;
      }
      c_n = st_n;
      c_ty = st_ty;
      c_tx = st_tx;
      if (prof) {
        pc[0] += t1 - t0;                                   // K loop (with the next patch's staging and, POOL, the previous epilogue inside)
        pc[1] += t2 - t1;                                   // wait for the other waves
        pc[2] += now() - t2;                                // exchange (+ the serial epilogue of the raw-output form)
      }
    }
    ebuf = (unsigned)((nit - 1) & 1) * RS_BUF;
    epi_serial();
  };
  switch (wave) {
    case 0: body(std::integral_constant<int, 0>{}); break;
    case 1: body(std::integral_constant<int, 1>{}); break;
    case 2: body(std::integral_constant<int, 2>{}); break;
    default: body(std::integral_constant<int, 3>{}); break;
  }
  if (H && !RAW && (lowbits & 0x1fffu) != 0 && p.bad_input != nullptr) *p.bad_input = 1;
  if (RAW && bad_depth != 0u && p.raw_err != nullptr) *p.raw_err = 1;
  if (prof && lane == 0 && (blockIdx.x % 16) == 0) {
    unsigned long long *q = p.prof + 16 * wave;
#pragma unroll
    for (int k = 0; k < 3; ++k) atomicAdd(q + k, pc[k]);
    atomicAdd(q + 5, (unsigned long long)nit);
  }
}

// Takes the launch when the tiles keep every workgroup busy for many rounds (the resident fragments cost ~one tile-time to load):
// the float16-piece stem with one N-tile and float32 output, or the bf16 dual stem (one piece, two N-tiles, bf16 raw output).
bool stem_rs_takes(const StemMXArgs &a, int pieces, int ntiles_n, bool bf16_out, int wgs) {
  const long ntiles = (long)a.B * ((a.Wo + TW - 1) / TW) * ((a.Ho + TH - 1) / TH);
  const bool f16 = pieces == 2 && ntiles_n == 1 && !bf16_out, dual = pieces == 1 && ntiles_n == 2 && bf16_out && a.pool == nullptr;
  return (f16 || dual) && wgs >= 8 && ntiles >= PNVO_RS_MIN_TILES * (long)wgs;
}

hipError_t launch_stem_rs(const StemMXArgs &a, int pieces, bool fast, int wgs, hipStream_t s) {
  static std::mutex attr_mu;
  static unsigned long long attr_seen = 0;            // one bit per device: the attribute is per device, not per process
  if (pnvo_first_launch_on_device(attr_mu, attr_seen)) {
    hipError_t e = hipSuccess;
    auto set = [&](const void *f) {
      if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    };
    set(reinterpret_cast<const void *>(stem_rs_kernel<2, true, true>));
    set(reinterpret_cast<const void *>(stem_rs_kernel<2, true, false>));
    set(reinterpret_cast<const void *>(stem_rs_kernel<2, false, true>));
    set(reinterpret_cast<const void *>(stem_rs_kernel<2, false, false>));
    set(reinterpret_cast<const void *>(stem_rs_kernel<2, true, true, true>));
    set(reinterpret_cast<const void *>(stem_rs_kernel<2, true, false, true>));
    set(reinterpret_cast<const void *>(stem_rs_kernel<2, false, true, true>));
    set(reinterpret_cast<const void *>(stem_rs_kernel<2, false, false, true>));
    set(reinterpret_cast<const void *>(stem_rs_kernel<1, false, true>));
    set(reinterpret_cast<const void *>(stem_rs_kernel<1, false, false>));
    if (e != hipSuccess) return e;
  }
  StemMXArgs p = a;
  p.tiles_x = (a.Wo + TW - 1) / TW;
  p.tiles_y = (a.Ho + TH - 1) / TH;
  const dim3 g((unsigned)(wgs & ~7)), b(256);
  const bool raw = p.raw_depth != nullptr, pool = p.pool != nullptr;
#define PNVO_RS_LAUNCH(...) hipLaunchKernelGGL((stem_rs_kernel<__VA_ARGS__>), g, b, RS_LDS, s, p)
  if (pieces == 1) {
    if (raw) PNVO_RS_LAUNCH(1, false, true);
    else PNVO_RS_LAUNCH(1, false, false);
  } else if (fast) {
    if (pool && raw) PNVO_RS_LAUNCH(2, true, true, true);
    else if (pool) PNVO_RS_LAUNCH(2, true, false, true);
    else if (raw) PNVO_RS_LAUNCH(2, false, true, true);
    else PNVO_RS_LAUNCH(2, false, false, true);
  } else {
    if (pool && raw) PNVO_RS_LAUNCH(2, true, true);
    else if (pool) PNVO_RS_LAUNCH(2, true, false);
    else if (raw) PNVO_RS_LAUNCH(2, false, true);
    else PNVO_RS_LAUNCH(2, false, false);
  }
#undef PNVO_RS_LAUNCH
  return hipGetLastError();
}

}  // namespace pnvo
