// Shared by ewald.hip and pme.hip: the cell's inverse (fractional coordinates) in double.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace aimnet {

// m = the row-vector cell in double, E.inv = its inverse (fractional coordinate a = sum_c x_c inv[c * 3 + a]); returns det(cell)
__device__ __forceinline__ double ewald_cell_geometry(const float* __restrict__ c, double m[9], EwaldSystem& E) {
  for (int k = 0; k < 9; ++k) m[k] = c[k];
  const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  const double id = 1.0 / det;
  E.inv[0] = (m[4] * m[8] - m[5] * m[7]) * id;
  E.inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  E.inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  E.inv[3] = (m[5] * m[6] - m[3] * m[8]) * id;
  E.inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  E.inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  E.inv[6] = (m[3] * m[7] - m[4] * m[6]) * id;
  E.inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  E.inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return det;
}

}  // namespace aimnet
