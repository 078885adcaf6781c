// Geometry of one implicit-GEMM launch (passed by value as a kernel argument).
#pragma once
#define CN_MAX_TAPS 9

struct CnConvGeom {
  // gathered tensor [B, Cin, Hin, Win] and written tensor [B, Cout, Hout, Wout]
  int B, Cin, Hin, Win;
  int Cout, Hout, Wout;
  long xbs, ybs;  // batch strides in elements (channel stride is H*W)
  // logical pixel grid of this launch (per image): input coord = g*is + d[t], output coord = g*os + o0
  int Hg, Wg;
  int is;
  int os, oy0, ox0;
  // taps: input offsets and the index of each tap in the packed weights [T][Kpad][Npad]
  int ntaps;
  int dy[CN_MAX_TAPS], dx[CN_MAX_TAPS], wt[CN_MAX_TAPS];
  int min_dy, min_dx;
  // LDS staging geometry (filled in by the launcher)
  int rows_cap, pitch, plane, w_lds_off;
  int Kpad, Npad;
  int tiles_per_img;
  int accumulate, has_bias;
};
