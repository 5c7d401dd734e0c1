// dpp_device.h -- sums over quads and over rows of sixteen lanes by DPP (data-parallel primitives of the VALU).
// __shfl_xor compiles to ds_bpermute_b32, one LDS-pipe instruction per 32-bit half and step; where such a sum sits
// on a kernel's critical path (the diagonal lanes of kernels_gather.hip: 72 of them per wave, queued behind the
// other waves' LDS traffic) the DPP form is the one to use.  Every lane of the group ends with the group's sum.
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double x)
{
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// lanes 4k .. 4k+3
__device__ __forceinline__ double dpp_quad_sum(double v)
{
  v += dpp_move_f64<0xB1>(v);            // quad_perm [1,0,3,2]
  v += dpp_move_f64<0x4E>(v);            // quad_perm [2,3,0,1]
  return v;
}
// lanes 16k .. 16k+15: four rotations within the row
__device__ __forceinline__ double dpp_row16_sum(double v)
{
  v += dpp_move_f64<0x128>(v);           // row_ror:8
  v += dpp_move_f64<0x124>(v);           // row_ror:4
  v += dpp_move_f64<0x122>(v);           // row_ror:2
  v += dpp_move_f64<0x121>(v);           // row_ror:1
  return v;
}
