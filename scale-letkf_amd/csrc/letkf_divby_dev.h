// letkf_divby_dev.h -- a / b for MANY numerators over ONE denominator, bit-identical to the IEEE double division hipcc
// emits (v_div_scale x2, v_rcp, two Newton steps on the reciprocal, q0 = a y, r = a - b q0, v_div_fmas, v_div_fixup):
// the denominator's part of that sequence -- the refined reciprocal y -- is computed once, and a quotient then costs the
// three instructions that depend on the numerator.  Used by the limited column search for |v_obs - v_ref| / vert_loc
// (scale/letkf/letkf_tools.f90:1853-1865), whose quotient enters the selection key and has to be the reference's to the
// last bit.
//
// Why it is the same number: for a denominator in [2^-100, 2^100] and a numerator a = 0 or 2^-900 <= a < 2^600 neither
// v_div_scale rescales its operand, v_div_fmas is a plain fma and v_div_fixup passes its input through, so the full
// sequence computes exactly the operations below on exactly the same values.  Outside that range of a:
//   a >= 2^600, +inf or NaN: the caller gets a itself -- like the true quotient it is > any cut-off, +inf or NaN;
//   0 < a < 2^-900: both quotients are < 2^-800, where the search only asks "> cut-off?" (no) and squares it (0).
// tests/test_gpu_divby.py runs the two side by side on the device over random and boundary operands.
#pragma once
#include <hip/hip_runtime.h>

namespace letkf {
namespace divby {

__device__ __forceinline__ bool in_range(const double b) { return b >= 0x1p-100 && b <= 0x1p100; }

__device__ __forceinline__ double reciprocal(const double b) {
#pragma clang fp contract(off)
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  return __builtin_fma(y, e, y);
}

// a >= 0 (or NaN); y = reciprocal(b), in_range(b)
__device__ __forceinline__ double quotient(const double a, const double b, const double y) {
#pragma clang fp contract(off)
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  const double q = __builtin_fma(r, y, q0);
  return a < 0x1p600 ? q : a;
}

}  // namespace divby
}  // namespace letkf
