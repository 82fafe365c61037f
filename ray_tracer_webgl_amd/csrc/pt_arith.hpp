// pt_arith.hpp — PT-SPEC arithmetic of the path tracer, device side.
//
// Every function here restates one piece of static/shader.frag in the fp32 operation order the CPU
// oracle (oracle/pt_oracle.c) states independently; the two share no source.  The translation unit is
// compiled with -ffp-contract=off: every fused multiply-add is an explicit __builtin_fmaf; / and sqrtf
// are IEEE correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); sin / cos / cbrt
// are the PT-SPEC polynomial forms, not v_sin / v_cos / v_exp / v_log.
//
// EXACTNESS ARGUMENT carried by this file: the unscaled division and square-root sequences
// (div_core, sqrt_core, hit_root, inv_sqrt_rn) return the SAME BITS as the plain operators for
// operands in the ranges stated at each function, and every caller guards the range wave-uniformly
// (plain operators for the lanes outside).  Checked on the device against the operators on 2 x 10^6
// operand triples per form: tests/test_gpu_parity.py::test_probe_unscaled_div_sqrt_match_the_operators.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_MAX_T 1e5f   // static/shader.frag:5
#define PT_MIN_T 0.001f // static/shader.frag:6
#define PT_TWO_PI 6.2831855f

// hip's __ballot takes an int: the bool -> int -> "!= 0" round trip costs two VALU ops per use
#define pt_ballot(cond) __builtin_amdgcn_ballot_w64(cond)

namespace ptd {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }

// static/shader.frag:15-19
__device__ __forceinline__ uint32_t base_hash(uint32_t px, uint32_t py) {
  uint32_t qx = 1103515245u * ((px >> 1) ^ py);
  uint32_t qy = 1103515245u * ((py >> 1) ^ px);
  uint32_t h32 = 1103515245u * (qx ^ (qy >> 3));
  return h32 ^ (h32 >> 16);
}

// `vec2(seed += .1, seed += .1)` of static/shader.frag:22,27,33: two rounded fp32 adds
__device__ __forceinline__ uint32_t seed_step_hash(float& seed) {
  float s1 = seed + 0.1f;
  float s2 = s1 + 0.1f;
  seed = s2;
  return base_hash(f2u(s1), f2u(s2));
}

// static/shader.frag:21-24 — float(0xffffffffU) == 2^32
__device__ __forceinline__ float hash1(float& seed) {
  uint32_t n = seed_step_hash(seed);
  return (float)n * (1.0f / 4294967296.0f);
}

// static/shader.frag:26-30 — float(0x7fffffff) == 2^31
__device__ __forceinline__ void hash2(float& seed, float& a, float& b) {
  uint32_t n = seed_step_hash(seed);
  a = (float)(n & 0x7fffffffu) * (1.0f / 2147483648.0f);
  b = (float)((n * 48271u) & 0x7fffffffu) * (1.0f / 2147483648.0f);
}

// static/shader.frag:32-36
__device__ __forceinline__ void hash3(float& seed, float& a, float& b, float& c) {
  uint32_t n = seed_step_hash(seed);
  a = (float)(n & 0x7fffffffu) * (1.0f / 2147483648.0f);
  b = (float)((n * 16807u) & 0x7fffffffu) * (1.0f / 2147483648.0f);
  c = (float)((n * 48271u) & 0x7fffffffu) * (1.0f / 2147483648.0f);
}

// PT-SPEC sin(2*pi*u), cos(2*pi*u), u >= 0
__device__ __forceinline__ void sincos2pi(float u, float& s_out, float& c_out) {
  float q = __builtin_rintf(u * 4.0f);
  float f = u - q * 0.25f;
  float x = f * PT_TWO_PI;
  float x2 = x * x;
  float ps = fma_(fma_(-1.9515295891e-4f, x2, 8.3321608736e-3f), x2, -1.6666654611e-1f);
  float s = fma_(x * x2, ps, x);
  float pc = fma_(fma_(2.443315711809948e-5f, x2, -1.388731625493765e-3f), x2, 4.166664568298827e-2f);
  float c = fma_(x2 * x2, pc, fma_(-0.5f, x2, 1.0f));
  int qi = ((int)q) & 3;
  float ss = (qi & 1) ? c : s;
  float cc = (qi & 1) ? s : c;
  if (qi == 1 || qi == 2) cc = -cc;
  if (qi >= 2) ss = -ss;
  s_out = ss;
  c_out = cc;
}

// PT-SPEC cbrt, x >= 0 finite
__device__ __forceinline__ float cbrt_(float x) {
  float y = u2f(0x54a2fa8cu - f2u(x) / 3u);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float t = x * y;
    t = t * y;
    t = t * y;
    y = y * fma_(t, -0.33333334f, 1.3333334f);
  }
  float r = (x * y) * y;
  return (x == 0.0f) ? 0.0f : r;
}


// --------------------------------------------------------------------------------------------
// Correctly rounded sqrt and division without the range scaling.
//
// `__builtin_sqrtf(x)` and `n / b` compile to correctly rounded fp32 results (PT-SPEC relies on
// that).  The compiler's expansions are, for the division n / b:
//     b' = v_div_scale(b)  n' = v_div_scale(n)          power-of-two scaling for extreme exponents
//     y0 = v_rcp(b')  y = fma(fma(-b', y0, 1), y0, y0)
//     q0 = n' y   q1 = fma(fma(-b', q0, n'), y, q0)   q = v_div_fmas(fma(-b', q1, n'), y, q1)
//     v_div_fixup(q, b, n)                               zeros, infinities, NaNs, the sign of 0
// and for the square root: scale by 2^32 below 2^-96, s = v_sqrt, pick s-1ulp / s / s+1ulp by the
// signs of the two residuals fma(-(s -+ 1ulp), s, x), unscale, pass 0 / inf through.
// v_div_scale is the identity (and v_div_fmas a plain fma, v_div_fixup the identity) when
//     b normal, |b| < 2^126, n != 0, |n| >= 2^-103, exponent(n) - exponent(b) < 96, n / b normal,
// so for such operands div_core() below IS the compiler's sequence, operation for operation, and
// returns the same correctly rounded quotient — with y computed once per denominator instead of
// once per division.  Likewise sqrt_core() is the compiler's sequence for x >= 2^-96 (it also
// returns 0 for 0 and inf for inf: both residual tests are then false).  Callers guard the
// operand ranges and fall back to the plain operators, wave-uniformly, when any lane is outside
// (practically never); the guards are stated at each call site.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float sqrt_core(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = u2f(f2u(s) - 1u), s_up = u2f(f2u(s) + 1u);
  const float r_dn = fma_(-s_dn, s, x), r_up = fma_(-s_up, s, x);
  float r = (0.0f >= r_dn) ? s_dn : s;
  r = (0.0f < r_up) ? s_up : r;
  return r;
}
__device__ __forceinline__ float rcp_newton(float b) {
  const float y0 = __builtin_amdgcn_rcpf(b);
  return fma_(fma_(-b, y0, 1.0f), y0, y0);
}
__device__ __forceinline__ float div_core(float n, float b, float y) {
  const float q0 = n * y;
  const float q1 = fma_(fma_(-b, q0, n), y, q0);
  return fma_(fma_(-b, q1, n), y, q1);
}
// x in [lo, hi) for positive floats lo < hi, false for negative x and NaN: one subtract and one
// unsigned compare on the bit patterns (which order like the values for positive floats)
__device__ __forceinline__ bool in_range_bits(float x, float lo, float hi) {
  return f2u(x) - f2u(lo) < f2u(hi) - f2u(lo);
}
// denominators for which 1/b and the exponent-difference conditions hold for every |n| < 2^76
__device__ __forceinline__ bool div_den_ok(float b) {
  return in_range_bits(__builtin_fabsf(b), 0x1p-20f, 0x1p20f);
}
// per-ray guard word for hit_root: the width of the accepted discriminant range [2^-96, 2^127),
// or 0 (nothing accepted) when the ray's |d|^2 is no denominator for the fast form
__device__ __forceinline__ uint32_t hit_root_guard(float a) {
  return div_den_ok(a) ? f2u(0x1p127f) - f2u(0x1p-96f) : 0u;
}
// correctly rounded sqrt for any x (same bits as __builtin_sqrtf)
__device__ __forceinline__ float sqrt_rn(float x) {
  float r = sqrt_core(x);
  const bool odd = !(x >= 0x1p-96f); // tiny, negative, NaN
  if (__builtin_expect(pt_ballot(odd) != 0ull, 0)) { // (rare)
    if (odd) r = __builtin_sqrtf(x);
  }
  return r;
}

// 1.0f / sqrtf(x), both roundings as written (normalize(), background()): for x in [2^-40, 2^40) the
// square root s is in [2^-20, 2^20) and the numerator is 1, so sqrt_core and div_core apply (with
// n = 1 the first product of div_core is the reciprocal itself)
__device__ __forceinline__ float inv_sqrt_rn(float x) {
  const float s = sqrt_core(x);
  const float y = rcp_newton(s);
  const float q1 = fma_(fma_(-s, y, 1.0f), y, y);
  float r = fma_(fma_(-s, q1, 1.0f), y, q1);
  const bool odd = !in_range_bits(x, 0x1p-40f, 0x1p40f);
  if (__builtin_expect(pt_ballot(odd) != 0ull, 0)) { // (rare)
    if (odd) r = 1.0f / __builtin_sqrtf(x);
  }
  return r;
}

// The exact part of hit_sphere, static/shader.frag:156-161, for a candidate with discriminant
// disc = fma(-a, c, half_b * half_b) >= 0 (or NaN): the root `v` the shader would test first,
// replaced by the far root when the near one is below MIN_T.  ya = rcp_newton(a) and
// guard = hit_root_guard(a), both per ray.
// Fast form when every lane that is in here has a in [2^-20, 2^20) and 2^-96 <= disc < 2^127.
// A finite disc means half_b * half_b did not overflow: |half_b| < 2^64, and sqrt(disc) is in
// [2^-48, 2^64), so both numerators n = -half_b -+ sqrt(disc) have |n| < 2^65; a numerator is
// either exactly 0 or at least one ulp of a number >= 2^-48 (>= 2^-71 > 2^-103), so every
// condition above holds for a non-zero n and div_core returns the correctly rounded root.
// For n == 0 div_core returns a zero, as the division does (its sign is v_div_fixup's business
// and is never looked at: a root below MIN_T is only compared with MIN_T — a near root is
// replaced by the far root, a far root rejected).
__device__ __forceinline__ float hit_root(float half_b, float disc, float a, float ya, uint32_t guard) {
  // straight-line fast form (both roots: the far one is needed by some lane in most evaluations, and
  // five multiply-adds cost less than the divergent region around them) ...
  const float sqrtd = sqrt_core(disc);
  const float v_near = div_core(-half_b - sqrtd, a, ya);
  const float v_far = div_core(-half_b + sqrtd, a, ya);
  float v = v_near < PT_MIN_T ? v_far : v_near;
  // ... and, if any lane's operands are outside the guarded range, the plain operators for those lanes
  const bool odd = f2u(disc) - f2u(0x1p-96f) >= guard;
  if (__builtin_expect(pt_ballot(odd) != 0ull, 0)) { // (rare)
    if (odd) {
      const float s = __builtin_sqrtf(disc);
      v = (-half_b - s) / a;             // :158
      if (v < PT_MIN_T) v = (-half_b + s) / a; // :159-160
    }
  }
  return v;
}

// static/shader.frag:114-121, from the hash of its seed step (n = seed_step_hash(seed)): hash3's three numbers :32-36
__device__ __forceinline__ V3 random_in_unit_sphere_from(uint32_t n) {
  const float h0 = (float)(n & 0x7fffffffu) * (1.0f / 2147483648.0f);
  const float h1 = (float)((n * 16807u) & 0x7fffffffu) * (1.0f / 2147483648.0f);
  const float h2 = (float)((n * 48271u) & 0x7fffffffu) * (1.0f / 2147483648.0f);
  float hx = fma_(h0, 2.0f, -1.0f);
  float sp, cp;
  sincos2pi(h1, sp, cp);
  float r = cbrt_(h2);
  float sq = sqrt_rn(fma_(-hx, hx, 1.0f));
  return mk(r * (sq * sp), r * (sq * cp), r * hx);
}
__device__ __forceinline__ V3 random_in_unit_sphere(float& seed) {
  float h0, h1, h2;
  hash3(seed, h0, h1, h2);
  float hx = fma_(h0, 2.0f, -1.0f);
  float sp, cp;
  sincos2pi(h1, sp, cp);
  float r = cbrt_(h2);
  float sq = sqrt_rn(fma_(-hx, hx, 1.0f));
  return mk(r * (sq * sp), r * (sq * cp), r * hx);
}

__device__ __forceinline__ V3 normalize3(V3 a) {
  float inv = inv_sqrt_rn(dot3(a, a));
  return mk(a.x * inv, a.y * inv, a.z * inv);
}

// GLSL reflect: I - 2*dot(N,I)*N
__device__ __forceinline__ V3 reflect3(V3 I, V3 N) {
  float k = 2.0f * dot3(N, I);
  return mk(fma_(-k, N.x, I.x), fma_(-k, N.y, I.y), fma_(-k, N.z, I.z));
}

// static/shader.frag:204-207
__device__ __forceinline__ float reflectance(float cosine, float ri) {
  float q = (1.0f - ri) / (1.0f + ri);
  float r0 = q * q;
  float x = 1.0f - cosine;
  float x2 = x * x;
  float x5 = (x2 * x2) * x;
  return fma_(1.0f - r0, x5, r0);
}

} // namespace ptd
