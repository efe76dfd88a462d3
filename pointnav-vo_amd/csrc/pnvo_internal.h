// pnvo_internal.h — declarations shared by the HIP translation units of libpnvo.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pnvo {

// A 2-channel slice of one observation tensor feeding the fused stem (MODE 2 of conv_mfma_kernel).
struct SrcPiece {
  const float *base;   // tensor base (nullptr: zero padding piece)
  int nch;             // channels per pixel of that tensor
  int choff;           // first channel of the slice
};

// One convolution / linear layer as an implicit GEMM:  M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.
struct ConvArgs {
  const float *x;        // [B,H,W,CIN] NHWC, CIN % 8 == 0 (channel-padded)
  const float *wpk;      // packed weights, see pack_conv_weight()
  float *y;              // [M, y_cstride] raw output
  const float *in_scale; // [B,CIN] per-(sample,channel) scale of the fused input transform, or nullptr
  const float *in_shift; // [B,CIN] shift; transform = relu(x*scale+shift) (GroupNorm+ReLU of the producer)
  float *stats;          // [B,slots,COUTP,2] per-(sample,slot,channel) partial (sum, sumsq) or nullptr
  const float *bias;     // [bias_rows,COUT] epilogue bias or nullptr (linear layers)
  const int64_t *bias_row; // [B] row of `bias` per sample (act-embed variants) or nullptr (row 0)
  int B, H, W, CIN;
  int Ho, Wo, COUT, COUTP; // COUTP = COUT rounded up to 32
  int KH, KW, stride, pad;
  int y_cstride;         // channel stride of y (COUT or COUTP)
  int relu_out;          // epilogue ReLU (after bias)
  int slots;             // stats slots per sample
  int lds_floats;        // dynamic LDS available for staging the input transform tables
  int MT, NT;            // wave tile: MT*32 pixels x NT*32 output channels
  int src_mode;          // 1: A is gathered from the observation tensors through `pieces` (x unused)
  const float *zero_page;            // >= 16 B of zeros (target of masked gathers)
  SrcPiece pieces[8][2][2];          // [j][lane half h][q]: channels 8j+4h+2q, +1 of the stem's K order
};

// The LDS-staged fused stem (stem_lds.hip).
struct StemArgs {
  SrcPiece pieces[8][2][2];   // [j][h][q] -> channels 8j+4h+2q, +1 of the stem's K order (slot g = 2j+h)
  const float *sc, *sh;       // [CPL] whitening x*sc+sh in the stem's channel order (0 for pad channels)
  const float *wpk;           // pack_stem_weight()
  const float *zero_page;     // >= 16 B of zeros (target of masked gathers)
  float *y;                   // [B,Ho,Wo,COUT] raw conv output
  float *stats;               // [B,slots,COUT,2]
  int B, H, W, Ho, Wo, CPL, slots, tiles_x, tiles_y;
  int dbg, lds_pad;           // experiment knobs (PNVO_STEM_DBG="<flags>,<lds_pad_bytes>"): 1 skip staging, 2 skip epilogue
};
int stem_tiles_x(int Wo);
int stem_tiles_y(int Ho);
hipError_t launch_stem_lds(const StemArgs &a, int cout, hipStream_t s);
void pack_stem_weight(const float *oihw_new, int cout, int cinp, int cpl, float *out);

int conv_slots(int P, int MT);                       // stats slots per sample for a given wave tile
void choose_tile(long M, int COUTP, int *MT, int *NT);
hipError_t launch_conv(const ConvArgs &a, hipStream_t s);

size_t packed_conv_floats(int cout, int cin, int kh, int kw);
void pack_conv_weight(const float *oihw, int cout, int cin, int kh, int kw, float *out);

// GroupNorm statistics -> per-(sample,channel) scale/shift.
// fixed_ns > 0: every sample has exactly fixed_ns slots (stem tiles); else slots follow the flattened wave tiles.
hipError_t launch_gn_finalize(const float *stats, int B, int slots, int CP, int C, int G, long P, int WM,
                              const float *gamma, const float *beta, float eps, float *scale, float *shift,
                              hipStream_t s, int fixed_ns = 0);

// Input assembly + whitening (vo_cnn.py:110-176) into channel-padded NHWC.
struct AssembleArgs {
  const float *src[4];   // rgb, depth, dd, tdv (nullptr if absent)
  int nsrc[4];           // pair channel counts
  const float *mean;     // [C] or nullptr
  const float *stdev;    // [C]
  int C, CP;
  long npix;             // B*H*W
  float *out;            // [npix, CP]
};
hipError_t launch_assemble(const AssembleArgs &a, hipStream_t s);

// relu(gn(x)) then MaxPool 3x3 s2 p1.
hipError_t launch_gn_relu_maxpool(const float *x, const float *scale, const float *shift, int B, int H, int W,
                                  int C, float *out, hipStream_t s);

// y = relu(a*sa+ta + r)  with r = b (plain) or b*sb+tb.  a,b,y: [B,P,C].
hipError_t launch_residual(const float *a, const float *sa, const float *ta, const float *b, const float *sb,
                           const float *tb, int B, long P, int C, float *y, hipStream_t s);

// y = relu(x*s+t)  (materialise a normalised activation; used for taps only)
hipError_t launch_apply_ss_relu(const float *x, const float *s, const float *t, int B, long P, int C, float *y,
                                hipStream_t st);

hipError_t launch_discretize_depth(const float *depth, int64_t n, int64_t in_stride, int bins, float *onehot,
                                   int64_t out_stride, int32_t *err_flag, hipStream_t s);
size_t topdown_workspace_bytes(int N, int H, int W);
hipError_t launch_topdown(const float *depth, int N, int H, int W, int64_t in_fstride, int64_t in_pstride,
                          const float *consts_host, int rows_around_center, float *out, int64_t out_fstride,
                          int64_t out_pstride, void *work, hipStream_t s);

}  // namespace pnvo
