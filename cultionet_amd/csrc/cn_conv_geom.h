// Geometry of one implicit-GEMM launch (passed by value as a kernel argument).
#pragma once
#define CN_MAX_TAPS 9
#define CN_MAX_CLASSES 16
#define CN_MAX_GROUPS 4

// One parity class of the output grid (a plain convolution has exactly one).
//   input coord = g*is + d[t],  output coord = g*os + o0
struct CnConvClass {
  int Hg, Wg;        // logical pixel grid of this class (per image)
  int oy0, ox0;
  int ntaps;
  int dy[CN_MAX_TAPS], dx[CN_MAX_TAPS], wt[CN_MAX_TAPS];  // input offsets; tap index in packed weights
  int min_dy, min_dx;
  int pitch, plane;  // LDS halo plane of this class (plane = rows * pitch <= NI*256)
  int vplane;        // 16-byte path: floats of the flattened-row image (multiple of 4)
  int rows;          // input rows staged per tile
  int tiles_per_img;
  int block_begin;   // first blockIdx.x of this class
  int grp;           // which (input, weights, bias, output) set this class works on
};

struct CnConvGeom {
  // gathered tensor [B, Cin, Hin, Win] and written tensor [B, Cout, Hout, Wout]
  int B, Cin, Hin, Win;
  int Cout, Hout, Wout;
  long xbs, ybs;  // batch strides in elements (channel stride is H*W)
  int is, os;
  int Kpad, Npad;           // packed weights [T][Kpad][Npad]
  int w_lds_off;            // float offset of the weight tile in LDS (after KC * max plane)
  int tap_lds_off;          // float offset of the per-class tap table in LDS
  int chunks_per_split;     // K-chunks (of 8 channels) per grid.z slice
  int atomic_out;           // split-K: accumulate with atomics into a pre-initialised output
  int accumulate, has_bias;
  // groups: G independent (input, weights, bias, output) sets in one launch; each class names its group
  // (CnConvClass::grp), so groups may differ in taps (dilation). shared_y: all groups sum into one output.
  int G, splits, shared_y;
  int odd_planes;      // H*W % 4 == 1: the 16-byte staging kernel runs its odd-plane variant (RP == 3)
  int want_interleave; // the launch asks for it (strided scatter); cn_plan grants it when the classes' tile counts agree
  int interleave;      // != 0: logical block l = tile * ncls + class (every class has the same number of tiles), classes in
                       // order of DESCENDING taps: the blocks of one cell tile sit next to each other (shared halo in L2) and
                       // every XCD / dispatch round gets the same mix of heavy and light parity classes
  float* part;         // split-K partial slices [(grp * splits + split)][B][Cout][Hout][Wout] (slice_stride != 0)
  long slice_stride;
  int grid_x, grid_y;  // logical grid (pixel tiles, N tiles); z = splits. Launched 1-D in XCD-aware order
  const float* gx[CN_MAX_GROUPS];
  const float* gwp[CN_MAX_GROUPS];
  const float* gbias[CN_MAX_GROUPS];
  float* gy[CN_MAX_GROUPS];
  int ncls;
  CnConvClass cls[CN_MAX_CLASSES];
};
