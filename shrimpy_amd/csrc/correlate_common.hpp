// Argument block shared by the correlation kernels (correlate.hip, correlate_sep.hip).
#pragma once

#include "common.hpp"

namespace lsr {

struct CorrArgs {
  const float* in;
  float* out;
  const float* aux;
  int64_t Z, Y, X;
  const float* wz;  // separable factors (device), pz / py / px taps
  const float* wy;
  const float* wx;
  const float* w;   // dense taps (device), C-order (pz, py, px)
  int pz, py, px;
  int epilogue;
  float eps;
  const float* nz;  // separable norm factors (Z, Y, X floats)
  const float* ny;
  const float* nx;
  const double* norm_table;  // dense: (pz+1)(py+1)(px+1) prefix sums
  int64_t tiles_x, tiles_y;
  int64_t z_chunk;  // output planes per workgroup along z
};

// correlate_sep.hip: compile-time-tap specialisations of the separable kernel.
bool sep_fast_supported(int pz, int py, int px, int* PZ, int* PYX);
int launch_sep_fast(const CorrArgs& p, int PZ, int PYX, hipStream_t s);

}  // namespace lsr
