// map_kernels.h -- traction-map preprocessing on the device (gfx950).
//
// The step BEFORE the hot path: what the reference does in numpy on
// the host every time a map arrives (/root/reference/mppi_numba/terrain.py):
//   terrain.py:408-452  use_det_dynamics: all mass into the first bin whose value is >= the
//                       CVaR_alpha traction of the cell (mean of the worst alpha fraction)
//   terrain.py:470-491  use_nom_dynamics_with_speed_map: nominal PMF (last bin) + int8 risk
//                       traction map 100*(CVaR - lo)/(hi - lo), truncated
//   terrain.py:511-583  crop to max_map_dim, ring of zero-traction cells (mass in bin 0),
//                       zero ring around the masks and the risk map
// One thread per padded cell; per cell the arithmetic is the reference's, operation for
// operation in float64 (numpy: int64 cumsum of the int8 masses, 0.01*cum, sequential
// cumsum of (0.01*p)*bin_value), so the outputs are bit-identical to the host path
// (tests/test_gpu_maps.py).  Bins are planes: every load and store is coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mppi {

enum PrepKind { PREP_TDM = 0, PREP_DET = 1, PREP_SPEED = 2 };

struct PrepJob {
  const int8_t* raw_pmf;  // [bins][src_rows][src_cols]
  const int8_t* raw_obs;  // [src_rows][src_cols] or nullptr (zeros)
  const int8_t* raw_unk;
  const float* bin_values;  // [bins] float32 (terrain.py:393)
  int bins, src_rows, src_cols;
  int valid_rows, valid_cols, pad;  // padded size = valid + 2*pad
  float lo, span;                   // bin_values_bounds[0], float32(hi - lo)
  double alpha;
  int kind;
  int8_t* pmf;  // [bins][Rp][Cp]
  int8_t* obs;  // [Rp][Cp]
  int8_t* unk;
  int8_t* risk;
  int* flags;  // [0] raw columns not summing to 100, [1] output columns that are not one-hot,
               // [2] mask values outside {0, 1}
};

__global__ __launch_bounds__(256) void k_prepare_maps(PrepJob J) {
  const int rp = J.valid_rows + 2 * J.pad, cp = J.valid_cols + 2 * J.pad;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rp * cp) return;
  const int r = i / cp, c = i - r * cp;
  const size_t plane = (size_t)rp * cp;
  const int rr = r - J.pad, cc = c - J.pad;
  if (rr < 0 || rr >= J.valid_rows || cc < 0 || cc >= J.valid_cols) {
    // the ring: zero traction for sure, no obstacle / unknown flag, zero risk speed
    for (int b = 0; b < J.bins; ++b) J.pmf[(size_t)b * plane + i] = (b == 0) ? (int8_t)100 : (int8_t)0;
    J.obs[i] = 0;
    J.unk[i] = 0;
    J.risk[i] = 0;
    return;
  }
  const size_t src_plane = (size_t)J.src_rows * J.src_cols;
  const size_t src = (size_t)rr * J.src_cols + cc;
  const int8_t ob = J.raw_obs ? J.raw_obs[src] : (int8_t)0;
  const int8_t un = J.raw_unk ? J.raw_unk[src] : (int8_t)0;
  J.obs[i] = ob;
  J.unk[i] = un;
  if ((ob != 0 && ob != 1) || (un != 0 && un != 1)) atomicAdd(&J.flags[2], 1);

  long isum = 0;                    // numpy: cumsum of an int8 array accumulates in int64
  double wc = 0.0;                  // sequential cumsum of (0.01 * p) * bin_value, float64
  double target = 0.0;              // CVaR_alpha traction of the cell
  bool found = false;
  double first_cum = 0.0, first_wc = 0.0;
  int hundred = 0, other = 0;
  for (int b = 0; b < J.bins; ++b) {
    const int p = (int)J.raw_pmf[(size_t)b * src_plane + src];
    if (J.kind == PREP_TDM) {
      J.pmf[(size_t)b * plane + i] = (int8_t)p;
      if (p == 100) ++hundred;
      else if (p != 0) ++other;
    }
    isum += p;
    const double term = (0.01 * (double)p) * (double)J.bin_values[b];
    wc = (b == 0) ? term : wc + term;
    const double cum = 0.01 * (double)isum;
    if (b == 0) { first_cum = cum; first_wc = wc; }
    if (!found && cum >= J.alpha) {  // np.argmax(cum >= alpha): the first such bin
      found = true;
      target = wc / (cum + 1e-6);
    }
  }
  if (isum != 100) atomicAdd(&J.flags[0], 1);
  if (J.kind == PREP_TDM) {
    if (!(hundred == 1 && other == 0)) atomicAdd(&J.flags[1], 1);
    J.risk[i] = 0;
    return;
  }
  if (J.alpha == 1.0) target = wc;                            // plain mean (terrain.py:426-431)
  else if (!found) target = first_wc / (first_cum + 1e-6);    // argmax of all-False is 0
  if (J.kind == PREP_DET) {
    int which = 0;  // first bin with target <= value; argmax of all-False is 0
    for (int b = 0; b < J.bins; ++b)
      if (target <= (double)J.bin_values[b]) { which = b; break; }
    for (int b = 0; b < J.bins; ++b) J.pmf[(size_t)b * plane + i] = (b == which) ? (int8_t)100 : (int8_t)0;
    J.risk[i] = 0;
  } else {  // PREP_SPEED
    for (int b = 0; b < J.bins; ++b) J.pmf[(size_t)b * plane + i] = (b == J.bins - 1) ? (int8_t)100 : (int8_t)0;
    // the mean is scaled as (100*(mean - lo))/range (terrain.py:476-478), the CVaR as
    // 100*((cvar - lo)/range) (terrain.py:488-490): after truncation the two can differ by one
    const double scaled = (J.alpha == 1.0) ? (100.0 * (target - (double)J.lo)) / (double)J.span
                                           : 100.0 * ((target - (double)J.lo) / (double)J.span);
    J.risk[i] = (int8_t)(int)scaled;  // astype(np.int8) of a value in [0, 100]: truncation
  }
}

}  // namespace mppi
