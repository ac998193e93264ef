// wave_ops.h — wave64 cross-lane primitives for gfx950 built on DPP row operations and the
// CDNA4 v_permlane16_swap / v_permlane32_swap instructions (all VALU, no LDS crossbar traffic;
// ds_bpermute-based __shfl costs ~10x more per step on this chip).
//
// Lane groups: a "row" is 16 lanes (DPP scope); a "half" is 32 lanes; the wave is 64.
// All functions require every lane of the participating group to be active.
#pragma once
#include <hip/hip_runtime.h>

namespace liodom_dev {

// DPP controls
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // i <-> 7-i within 8 lanes
constexpr int DPP_MIRROR = 0x140;       // i <-> 15-i within a row
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const int lo = dpp_i32<CTRL>((int)(unsigned int)v);
  const int hi = dpp_i32<CTRL>((int)(unsigned int)(v >> 32));
  return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  return __longlong_as_double((long long)dpp_u64<CTRL>((unsigned long long)__double_as_longlong(v)));
}

// value of the lane 16 away (same position in the neighbouring row: rows 0<->1, 2<->3)
__device__ __forceinline__ int xor16_i32(int v) {
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned int)v, (unsigned int)v, false, false);
  // one result holds this lane's own row, the other the neighbouring row (rows (0,0,2,2) and
  // (1,1,3,3) of v); pick whichever differs from the own value (robust to the result order)
  const int a = (int)r[0], b = (int)r[1];
  return (a != v) ? a : b;
}
__device__ __forceinline__ int xor32_i32(int v) {
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned int)v, (unsigned int)v, false, false);
  const int a = (int)r[0], b = (int)r[1];   // (lo, lo) and (hi, hi) halves of v
  return (a != v) ? a : b;
}
__device__ __forceinline__ unsigned long long xor16_u64(unsigned long long v) {
  const unsigned int lo = (unsigned int)xor16_i32((int)(unsigned int)v);
  const unsigned int hi = (unsigned int)xor16_i32((int)(unsigned int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long xor32_u64(unsigned long long v) {
  const unsigned int lo = (unsigned int)xor32_i32((int)(unsigned int)v);
  const unsigned int hi = (unsigned int)xor32_i32((int)(unsigned int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// ---- max / min of u64 -------------------------------------------------------------------
__device__ __forceinline__ unsigned long long row_max_u64(unsigned long long v) {
  unsigned long long o;
  o = dpp_u64<DPP_XOR1>(v); v = o > v ? o : v;
  o = dpp_u64<DPP_XOR2>(v); v = o > v ? o : v;
  o = dpp_u64<DPP_HALF_MIRROR>(v); v = o > v ? o : v;
  o = dpp_u64<DPP_MIRROR>(v); v = o > v ? o : v;
  return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = row_max_u64(v);
  unsigned long long o = xor16_u64(v); v = o > v ? o : v;
  o = xor32_u64(v); v = o > v ? o : v;
  return v;   // uniform over the wave
}
__device__ __forceinline__ unsigned long long half_min_u64(unsigned long long v) {
  unsigned long long o;
  o = dpp_u64<DPP_XOR1>(v); v = o < v ? o : v;
  o = dpp_u64<DPP_XOR2>(v); v = o < v ? o : v;
  o = dpp_u64<DPP_HALF_MIRROR>(v); v = o < v ? o : v;
  o = dpp_u64<DPP_MIRROR>(v); v = o < v ? o : v;
  o = xor16_u64(v); v = o < v ? o : v;
  return v;   // uniform over each 32-lane half
}
__device__ __forceinline__ unsigned int half_min_u32(unsigned int v) {
  unsigned int o;
  o = (unsigned int)dpp_i32<DPP_XOR1>((int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<DPP_XOR2>((int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<DPP_HALF_MIRROR>((int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<DPP_MIRROR>((int)v); v = o < v ? o : v;
  o = (unsigned int)xor16_i32((int)v); v = o < v ? o : v;
  return v;   // uniform over each 32-lane half
}
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
  unsigned int o;
  o = (unsigned int)dpp_i32<DPP_XOR1>((int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<DPP_XOR2>((int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<DPP_HALF_MIRROR>((int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<DPP_MIRROR>((int)v); v = o > v ? o : v;
  o = (unsigned int)xor16_i32((int)v); v = o > v ? o : v;
  o = (unsigned int)xor32_i32((int)v); v = o > v ? o : v;
  return v;   // uniform over the wave
}
__device__ __forceinline__ int wave_min_i32(int v) {
  int o;
  o = dpp_i32<DPP_XOR1>(v); v = o < v ? o : v;
  o = dpp_i32<DPP_XOR2>(v); v = o < v ? o : v;
  o = dpp_i32<DPP_HALF_MIRROR>(v); v = o < v ? o : v;
  o = dpp_i32<DPP_MIRROR>(v); v = o < v ? o : v;
  o = xor16_i32(v); v = o < v ? o : v;
  o = xor32_i32(v); v = o < v ? o : v;
  return v;
}

// ---- sums -------------------------------------------------------------------------------
// Sum over each 16-lane row, fixed butterfly order; every lane of the row holds the result.
__device__ __forceinline__ double row_sum_f64(double x) {
  x += dpp_f64<DPP_XOR1>(x);
  x += dpp_f64<DPP_XOR2>(x);
  x += dpp_f64<DPP_HALF_MIRROR>(x);
  x += dpp_f64<DPP_MIRROR>(x);
  return x;
}

// ---- scans ------------------------------------------------------------------------------
// Inclusive prefix sum over the wave (64 lanes).
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR1, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR2, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR4, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR8, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15, 0xA, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST31, 0xC, 0xF, true);
  return v;
}
// Inclusive prefix sum over each 32-lane half.
__device__ __forceinline__ int half_incl_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR1, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR2, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR4, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR8, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15, 0xA, 0xF, true);
  return v;
}

__device__ __forceinline__ int readlane_i32(int v, int lane /*wave-uniform*/) {
  return __builtin_amdgcn_readlane(v, lane);
}

}  // namespace liodom_dev
