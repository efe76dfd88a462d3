// stem_tile.h — tile geometry, LDS patch layout and packing helpers shared by the stem kernels on the 16-bit matrix cores
// (stem_mx.hip: one tile per workgroup and the role-specialised persistent form; stem_rs.hip: weights resident in registers).
#pragma once
#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int TH = 8, TW = 16;
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5;   // 21 x 37
constexpr int NPIX = PH * PW;                     // 777
constexpr int PITCH = 80;                         // bytes per patch pixel: 64 (K-slots) + 16 (remainders)
constexpr int PAR = 19 * PITCH;                   // odd-column plane of a patch row
constexpr int ROW = 3072;                         // patch row pitch (2 x 19 x 80 = 3040, padded: 2 rows = 0 mod 256 B)
constexpr int PATCH_BYTES = PH * ROW;             // 64512
constexpr int XCHG_BYTES = 4 * 4 * 4096;          // K-split exchange: [M-tile][wave][4 KB], reuses the patch area
constexpr int RED_OFF = XCHG_BYTES > PATCH_BYTES ? XCHG_BYTES : PATCH_BYTES;
constexpr int NTHREADS = 256;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 r = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  const f16x2 r = __builtin_convertvector(f32x2{a, b}, f16x2);     // v_cvt_pk_f16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, r);
}
template <bool H>
__device__ __forceinline__ unsigned pack_pair(float a, float b) {
  return H ? pack_f16(a, b) : pack_bf16(a, b);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

}  // namespace

// LDS offset of tap t = (kh, kw) inside the patch: kh * ROW + (kw & 1) * PAR + (kw >> 1) * PITCH.  Scalar arithmetic only (t is
// wave-uniform; t / 7 == (37 t) >> 8 for t < 56): a constant-memory table cost an s_load + s_waitcnt lgkmcnt(0) per tap INSIDE the K
// loop — the wait also drains the A-fragment ds_reads in flight, ~300 exposed cycles per tap (round 4, from the ISA).
__device__ __forceinline__ unsigned tap_lds_offset(int t) {
  const int kh = (t * 37) >> 8, kw = t - 7 * kh;
  return (unsigned)(kh * ROW + (kw & 1) * PAR + (kw >> 1) * PITCH);
}
}  // namespace pnvo
