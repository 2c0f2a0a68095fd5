// Device check of scale-letkf_amd/csrc/letkf_divby_dev.h: the reciprocal-based quotient against the division hipcc emits,
// over random operands of the search's ranges and over the boundary cases the header lists.  Prints "mismatches N".
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "letkf_divby_dev.h"

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ double from_bits(uint64_t b) { return __longlong_as_double((long long)b); }
__device__ __forceinline__ uint64_t bits(double d) { return (uint64_t)__double_as_longlong(d); }

// mode 0: a, b uniform in their exponents over the ranges of the identity claim; mode 1: "ordinary" magnitudes
// (a in [0, 1e5), b in [1e-3, 1e4)); mode 2: tiny / huge / special numerators
__global__ void check(const int mode, const uint64_t seed, unsigned long long* bad, unsigned long long* bad_first) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t r1 = mix(seed + 2 * i), r2 = mix(seed + 2 * i + 1);
  double a, b;
  if (mode == 0) {
    const uint64_t eb = 1023 - 100 + (r2 >> 52) % 200;                 // 2^-100 .. 2^99
    b = from_bits((eb << 52) | (r2 & 0xFFFFFFFFFFFFFull));
    const uint64_t ea = 1023 - 900 + (r1 >> 52) % 1499;                // 2^-900 .. 2^598
    a = from_bits((ea << 52) | (r1 & 0xFFFFFFFFFFFFFull));
    if ((i & 1023) == 0) a = 0.0;
    if ((i & 1023) == 1) b = 0x1p-100;
    if ((i & 1023) == 2) b = 0x1p100;
  } else if (mode == 1) {
    a = (double)(r1 >> 11) * 0x1p-53 * 1e5;
    b = 1e-3 + (double)(r2 >> 11) * 0x1p-53 * 1e4;
    if ((i & 7) == 0) a = fabs(log((double)(r1 >> 40)) - log(1.0 + (double)(r2 >> 40)));   // differences of logarithms
  } else {
    b = 1e-3 + (double)(r2 >> 11) * 0x1p-53 * 1e4;
    const int k = (int)(i % 6);
    if (k == 0) a = from_bits(r1 & 0x0FFFFFFFFFFFFFFFull);             // anything below 2^-767, denormals included
    else if (k == 1) a = from_bits(0x7FF0000000000000ull);             // +inf
    else if (k == 2) a = from_bits(0x7FF8000000000000ull);             // NaN
    else if (k == 3) a = from_bits(((uint64_t)(1023 + 600 + (r1 >> 52) % 400) << 52) | (r1 & 0xFFFFFFFFFFFFFull));
    else if (k == 4) a = from_bits((uint64_t)(r1 % 4096));             // smallest denormals
    else a = 0x1p600;
  }
  if (!letkf::divby::in_range(b)) return;
  const double y = letkf::divby::reciprocal(b);
  const double q = letkf::divby::quotient(a, b, y);
  const double t = a / b;
  bool ok;
  const double cut = 3.651483717;
  if (a == 0.0 || (a >= 0x1p-900 && a < 0x1p600)) ok = bits(q) == bits(t);
  else if (a != a) ok = q != q && t != t;
  else if (a >= 0x1p600) ok = (q > cut) && (t > cut);
  else ok = !(q > cut) && !(t > cut) && q * q == 0.0 && t * t == 0.0;
  if (!ok) {
    if (atomicAdd(bad, 1ull) == 0ull) {
      bad_first[0] = bits(a);
      bad_first[1] = bits(b);
      bad_first[2] = bits(q);
      bad_first[3] = bits(t);
    }
  }
}

int main() {
  unsigned long long *bad, *first;
  if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first, 32) != hipSuccess) return 2;
  unsigned long long total = 0;
  for (int mode = 0; mode < 3; ++mode) {
    (void)hipMemset(bad, 0, 8);
    (void)hipMemset(first, 0, 32);
    for (int rep = 0; rep < 8; ++rep) hipLaunchKernelGGL(check, dim3(16384), dim3(256), 0, 0, mode, 0x1234567ull * (rep + 1) + mode, bad, first);
    unsigned long long h = 0, f[4];
    if (hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    (void)hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
    printf("mode %d: mismatches %llu of %llu", mode, h, 8ull * 16384 * 256);
    if (h) printf("  first: a=%016llx b=%016llx q=%016llx a/b=%016llx", f[0], f[1], f[2], f[3]);
    printf("\n");
    total += h;
  }
  printf("mismatches %llu\n", total);
  return total ? 1 : 0;
}
