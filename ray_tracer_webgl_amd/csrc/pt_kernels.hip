// pt_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the path tracer.
//
// The hot path of the reference is one fragment-shader invocation per pixel
// (static/shader.frag:406-413): seed -> for each sample {camera ray -> bounce loop {scan the
// sphere list -> scatter}}.  Here it is ONE persistent kernel:
//
//   * work item = (pixel, pass): the serial fp32 seed chain of static/shader.frag:11,21-36 ties
//     all samples of one fragment invocation together, so a (pixel, pass) stream is the finest
//     unit that can run independently.  Items are dealt from a global queue in 8x8-pixel-tile
//     order, one wave-level atomicAdd per refill.
//   * one lane owns one item at a time and keeps its whole path state in VGPRs; when its path
//     ends it starts its own next sample, when its item ends it pulls the next item — so every
//     lane of the wave enters the sphere loop with a live ray (wave-level culling of finished
//     paths by regeneration instead of idling), and a wave leaves only when the queue is dry.
//   * the sphere list's geometry (cx,cy,cz,r^2: 16 B) is staged into LDS once per workgroup and
//     the intersection loop walks it with wave-uniform ds_read_b128 broadcasts; shading data
//     (32 B/sphere) stays in global memory / L2 and is read once per segment for the closest hit.
//   * each item's radiance sum is written once, as one 16-byte store, into a per-pass slab; a
//     second tiny kernel folds the slabs into the accumulation buffer in pass order, so the
//     fp32 sum is bit-identical however the queue was scheduled.
//
// ARITHMETIC: this file implements PT-SPEC — the same contract the CPU oracle
// states independently in oracle/pt_oracle.c.  It is compiled with -ffp-contract=off; every
// fused multiply-add below is an explicit __builtin_fmaf; / and sqrtf are IEEE correctly rounded
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); sin/cos/cbrt are the PT-SPEC
// polynomial forms, not v_sin/v_cos/v_exp/v_log.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_kernel_args.h"

#define PT_MAX_T 1e5f   // static/shader.frag:5
#define PT_MIN_T 0.001f // static/shader.frag:6
#define PT_TWO_PI 6.2831855f
#ifndef PT_PARKING
#define PT_PARKING 1 // walk kernels: park the path state in LDS during the walk (fewer VGPRs -> more waves)
#endif
#define PT_COOP_MAX_LIVE 16 // tail mode when at most this many lanes of a wave hold a ray

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
#ifndef PT_FAST_EXACT
#define PT_FAST_EXACT 1 // hit_root: unscaled forms under a guard (0: the plain operators)
#endif
#ifndef PT_FAST_SQRT
#define PT_FAST_SQRT 1
#endif
#ifndef PT_FAST_NORMAL
#define PT_FAST_NORMAL 1
#endif
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
#if PT_FAST_SQRT
  float r = sqrt_core(x);
  const bool odd = !(x >= 0x1p-96f); // tiny, negative, NaN
  if (__builtin_expect(pt_ballot(odd) != 0ull, 0)) { // (rare)
    if (odd) r = __builtin_sqrtf(x);
  }
  return r;
#else
  return __builtin_sqrtf(x);
#endif
}

// 1.0f / sqrtf(x), both roundings as written (normalize(), background()): for x in [2^-40, 2^40) the
// square root s is in [2^-20, 2^20) and the numerator is 1, so sqrt_core and div_core apply (with
// n = 1 the first product of div_core is the reciprocal itself)
__device__ __forceinline__ float inv_sqrt_rn(float x) {
#if PT_FAST_SQRT
  const float s = sqrt_core(x);
  const float y = rcp_newton(s);
  const float q1 = fma_(fma_(-s, y, 1.0f), y, y);
  float r = fma_(fma_(-s, q1, 1.0f), y, q1);
  const bool odd = !in_range_bits(x, 0x1p-40f, 0x1p40f);
  if (__builtin_expect(pt_ballot(odd) != 0ull, 0)) { // (rare)
    if (odd) r = 1.0f / __builtin_sqrtf(x);
  }
  return r;
#else
  return 1.0f / __builtin_sqrtf(x);
#endif
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
#if PT_FAST_EXACT
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
#else
  const float sqrtd = __builtin_sqrtf(disc);
  float v = (-half_b - sqrtd) / a;             // :158
  if (v < PT_MIN_T) v = (-half_b + sqrtd) / a; // :159-160
  return v;
#endif
}

// static/shader.frag:114-121
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

using namespace ptd;

// --------------------------------------------------------------------------------------------
// The path-tracing kernel body.
// --------------------------------------------------------------------------------------------
// SCAN_LDS   : the scan (PHASE 1) walks the LDS copy with wave-uniform ds_read_b128 broadcasts.
// !SCAN_LDS  : the scan walks the padded global copy with wave-uniform SCALAR loads (constant
//              address space -> s_load_dwordx16 per four spheres via the scalar cache / L2);
//              sphere data reaches the VALU as SGPR operands, no LDS traffic in the scan, 12
//              fewer VGPRs.  Same arithmetic, bit-identical images.  Faster on dense mid-size
//              scenes (config 2: -11 %), slower once the list outgrows the scalar cache;
//              PT_GEOM_AUTO measures both per scene.
// HAVE_LDS   : an LDS copy of the list exists (n <= 10 232) and serves every PER-LANE indexed
//              read (exact phase, tail mode, shading) whichever way the scan reads; without it
//              (lists beyond the 160 KiB LDS) those gathers go to global memory.
// WALK       : 0 = PHASE 1 scans the whole list.  Otherwise PHASE 1 walks a culling structure
//              that only decides which spheres are LOOKED AT (see the notes at the walks):
//                1..3  the hierarchy of pt_bvh.hpp: 1 = nodes and slots staged in LDS, 2 = nodes
//                      in LDS, slots in global memory / L2, 3 = both in global memory;
//                4..6  the uniform grid of pt_grid.hpp: 4 = cells and entries staged in LDS,
//                      5 = cells in LDS, entries in global memory / L2, 6 = both global.
//              List-order reads (tail mode, PHASE 3, shading) go to the global copy.
// COUNT      : tally what the walk executes (iterations and active lanes per phase) into
//              A.counters — the measuring twin of a kernel, never the one that is timed.
template <bool SCAN_LDS, bool HAVE_LDS, int WALK = 0, bool COUNT = false>
__device__ __forceinline__ void pt_trace_body(const PtKernelArgs& A) {
  constexpr bool BVH = WALK >= 1 && WALK <= 3;
  constexpr bool GRID = WALK >= 4;
  constexpr bool TREE = BVH || GRID;  // a culling structure: hits are (slot, value) pairs
  constexpr int BVH_MODE = BVH ? WALK : 0;
  constexpr bool NODES_LDS = WALK == 1 || WALK == 2;
  constexpr bool CELLS_LDS = WALK == 4 || WALK == 5;
  constexpr bool SLOTS_LDS = WALK == 1 || WALK == 4;
  static_assert(HAVE_LDS || !SCAN_LDS, "an LDS scan needs the LDS copy");
  static_assert(!TREE || (!SCAN_LDS && !HAVE_LDS), "the walk kernels read list-order data from global memory");
  extern __shared__ float4 s_geom[];
  const float4* __restrict__ g_geom = reinterpret_cast<const float4*>(A.geom);
  const uint4* __restrict__ g_nodes = reinterpret_cast<const uint4*>(A.bvh_nodes);
  const float4* __restrict__ g_nodes32 = reinterpret_cast<const float4*>(A.bvh_nodes32);
  const float4* __restrict__ g_slots = reinterpret_cast<const float4*>(A.bvh_slots);

  // ---- copy the (already padded, {cx,cy,cz,r*r}) geometry into LDS once per workgroup --------
  if constexpr (HAVE_LDS && !TREE) {
    const uint32_t n_padded = PT_LDS_ENTRIES(A.n_spheres);
    for (uint32_t i = threadIdx.x; i < n_padded; i += blockDim.x) s_geom[i] = g_geom[i];
    __syncthreads();
  }
  if constexpr (WALK == 1) { // [2 * (n_nodes + 1) halves of fp32 nodes][n_slots slots]
    const uint32_t n_a = 2u * (A.n_nodes + 1u);
    typedef float4 __attribute__((address_space(3))) lds_f4s;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_f4s*)s_geom;
    for (uint32_t i = threadIdx.x; i < n_a; i += blockDim.x) {
      float4 v = g_nodes32[i];
      if ((i & 1u) == 0u) v.w = u2f(f2u(v.w) + lds_base); // skip link: byte offset -> LDS address
      s_geom[i] = v;
    }
    for (uint32_t i = threadIdx.x; i < A.n_slots; i += blockDim.x) s_geom[n_a + i] = g_slots[i];
    __syncthreads();
  }
  if constexpr (WALK == 2) { // [n_nodes + 1 packed nodes]
    const uint32_t n_a = A.n_nodes + 1u;
    uint4* s_nodes = reinterpret_cast<uint4*>(s_geom);
    for (uint32_t i = threadIdx.x; i < n_a; i += blockDim.x) s_nodes[i] = g_nodes[i];
    __syncthreads();
  }
  if constexpr (CELLS_LDS) { // [n_cells cell records, padded to 16 B][mode 4: n_slots entries]
    const uint32_t n_c4 = (A.n_cells + 3u) >> 2;
    const uint4* g_c4 = reinterpret_cast<const uint4*>(A.grid_cells); // the array is padded to 16 B
    uint4* s_c4 = reinterpret_cast<uint4*>(s_geom);
    for (uint32_t i = threadIdx.x; i < n_c4; i += blockDim.x) s_c4[i] = g_c4[i];
    if constexpr (WALK == 4)
      for (uint32_t i = threadIdx.x; i < A.n_slots; i += blockDim.x) s_geom[n_c4 + i] = g_slots[i];
    __syncthreads();
  }
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef const f4v __attribute__((address_space(4))) const_f4v;
  const_f4v* c_geom = (const_f4v*)A.geom;
  // wave-uniform index (the scan)
  auto geom_scan = [&](uint32_t i) -> float4 {
    if constexpr (SCAN_LDS) {
      return s_geom[i];
    } else {
      const f4v v = c_geom[i];
      return make_float4(v.x, v.y, v.z, v.w);
    }
  };
  // per-lane index (exact phase, tail mode, shading)
  auto geom_at = [&](uint32_t i) -> float4 {
    if constexpr (HAVE_LDS && !TREE) {
      return s_geom[i];
    } else {
      return g_geom[i];
    }
  };
  // culling-structure reads (per-lane index)
  auto node_at = [&](uint32_t i) -> uint4 { // packed nodes (modes 2, 3)
    if constexpr (NODES_LDS) return reinterpret_cast<const uint4*>(s_geom)[i];
    else return g_nodes[i];
  };
  auto slot_at = [&](uint32_t i) -> float4 {
    if constexpr (WALK == 1) return s_geom[2u * (A.n_nodes + 1u) + i];
    else if constexpr (WALK == 4) return s_geom[((A.n_cells + 3u) >> 2) + i];
    else return g_slots[i];
  };
  auto cell_at = [&](uint32_t i) -> uint32_t {
    if constexpr (CELLS_LDS) return reinterpret_cast<const uint32_t*>(s_geom)[i];
    else return A.grid_cells[i];
  };
  (void)node_at; (void)cell_at; (void)SLOTS_LDS;
  const_f4v* c_slots = (const_f4v*)A.bvh_slots;
  // LDS behind the staged scene: PT_PARK_STRIDE dwords per lane for the parked path state.  The
  // stride is odd, so the 32 lanes of a half-wave hit 32 different banks at any fixed field, and
  // every field is an immediate offset from the lane's base address.
  typedef volatile uint32_t __attribute__((address_space(3))) lds_u32;
  lds_u32* park = (lds_u32*)reinterpret_cast<uint32_t*>(s_geom) + (A.lds_scene_bytes >> 2) + PT_PARK_STRIDE * threadIdx.x;

  // (recomputed where needed rather than kept in a VGPR for the kernel's lifetime)
#define lane (threadIdx.x & 63u)
  auto div_ = [](uint32_t n, uint32_t m, uint32_t s1, uint32_t s2) -> uint32_t {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> s1)) >> s2;
  };
  const uint32_t n_spheres = A.n_spheres;
  // K: the same argument block, read from the kernarg segment AT THE POINT OF USE (scalar loads
  // through the scalar cache).  The once-per-wave-step sections (item decode, camera ray, walk
  // set-up) use it so that their ~70 uniforms do not sit in SGPRs (or spill to VGPR lanes)
  // across the walk and the shading code.
  typedef const PtKernelArgs __attribute__((address_space(4))) karg_t;
  karg_t& K = *(karg_t*)__builtin_amdgcn_kernarg_segment_ptr();

  // ---- per-lane path state ---------------------------------------------------------------------
  bool alive = false;     // lane holds a live ray
  bool exhausted = false; // queue returned "no more items" to this lane
  bool new_path = false;  // lane must generate its next camera ray before the next scan
  uint32_t slab_index = 0;
  uint32_t item_tile = 0xffffffffu, item_segs = 0; // cost feedback for the next launch's tile order
  int sample = 0, depth = 0;
  float seed = 0.f, st_s = 0.f, st_t = 0.f;
  V3 o = mk(0, 0, 0), d = mk(0, 0, 0);
  float a = 0.f; // dot(d,d), hoisted out of the sphere loop (static/shader.frag:147)
  V3 col = mk(1, 1, 1), sum = mk(0, 0, 0);

  uint32_t seg_count = 0; // wave-uniform tally (samples are derived on the host: pixels * spp * passes)
  uint32_t pool_next = 0, pool_end = 0;     // wave-uniform: this wave's reserved queue items
  uint32_t refill_waited = 0;               // wave-uniform: steps the waiting lanes have been put off
  uint32_t pool_tp0 = 0, pool_split = 0, pool_tile0 = 0, pool_tile1 = 0; // wave-uniform: the reservation's tile(s)

  // ---- walk state that survives a wave step (walk kernels) ------------------------------------
  // The 64 walks of a wave step differ in length, and every loop runs for its longest lane:
  // after the bulk has finished, a handful of stragglers (config 2: ~5 lanes for the last third
  // of the node iterations) would keep the whole wave walking.  Instead, once fewer than
  // A.carry_lanes lanes (and less than half of the wave's live lanes) are still walking, the
  // wave moves on: the finished lanes are shaded and get their next ray, the stragglers are
  // CARRIED — they keep their walk state in registers, skip shading, and continue their walk
  // in the next wave step beside the fresh walks.  Results cannot change (each lane performs
  // the same operations on the same ray, only later); segments are counted when shaded.
  bool carried = false;
  float closest_w = PT_MAX_T;
  uint32_t hit_pos = 0xffffffffu; // the slot of the closest hit (its sphere index is looked up once, at the end)
  // hierarchy: cursor, queued leaves (8 x 16 bit), queued candidates (8 x 16 bit)
  uint32_t cur = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0, l_cnt = 0;
  uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, q_cnt = 0;
  // grid: boundary-crossing times, linear cell index, steps left per axis (3 x 10 bit, +1),
  // the cell being tested (first untested entry | entries left << 24), its exit time
  float tmx = 0.f, tmy = 0.f, tmz = 0.f, t_exit = 0.f;
  uint32_t cell = 0, rem = 0, pend = 0;
  bool gactive = false;

  // executed-work tallies of the COUNT twin (wave-uniform)
  unsigned long long t_wave_start = 0, t_wave_dry = 0;
  uint32_t tb_bin = 0xffffffffu, tb_acc = 0;
  if constexpr (COUNT) t_wave_start = __builtin_amdgcn_s_memrealtime();
  uint32_t n_walk_it = 0, n_walk_ln = 0, n_leaf_it = 0, n_leaf_ln = 0, n_exact_it = 0, n_exact_ln = 0, n_steps = 0,
           n_carried = 0;
#define PT_COUNT(IT, LN, MASK) do { if constexpr (COUNT) { IT++; LN += (uint32_t)__popcll(MASK); } } while (0)
  // phase clock of the COUNT twins (shader cycles, s_memtime): where a wave's time goes
  //   0 refill  1 camera ray  2 set-up + always-tested  3 advance / node loops  4 leaf + exact  5 literal + rest  6 shade
  unsigned long long ph_t[7] = {0, 0, 0, 0, 0, 0, 0}, ph_mark = 0;
  if constexpr (COUNT) ph_mark = __builtin_amdgcn_s_memtime();
#define PT_PHASE(k) do { if constexpr (COUNT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph_t[k] += now_ - ph_mark; ph_mark = now_; } } while (0)

  // start the next camera path of this lane's item: static/shader.frag:365-370 + :342-351
  // the pixel jitter is divided by the image size (:367-368): uniform denominators, numerators that
  // are 0 or >= 2^-31 and < 1, so div_core applies whenever width and height are in [2^-20, 2^20)
  // (a +0 numerator gives +0 either way: positive operands)
  const bool wh_ok = div_den_ok(K.fw) && div_den_ok(K.fh); // wave-uniform
  const float y_fw = u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(rcp_newton(K.fw)))); // uniform: kept in SGPRs
  const float y_fh = u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(rcp_newton(K.fh))));
  auto start_sample = [&]() {
    float r0, r1;
    hash2(seed, r0, r1);
    float jx, jy;
    if (wh_ok) { jx = div_core(r0, K.fw, y_fw); jy = div_core(r1, K.fh, y_fh); }
    else { jx = r0 / K.fw; jy = r1 / K.fh; }
    float s = st_s + jx;
    float t = st_t + jy;
    float ua = hash1(seed); // random_in_unit_circle :123-129, consumed even if lens_radius == 0
    float sa, ca;
    sincos2pi(ua, sa, ca);
    float rr = sqrt_rn(hash1(seed));
    float rdx = K.lens_radius * (rr * ca);
    float rdy = K.lens_radius * (rr * sa);
    V3 off = mk(fma_(K.cam_v[0], rdy, K.cam_u[0] * rdx), fma_(K.cam_v[1], rdy, K.cam_u[1] * rdx),
                fma_(K.cam_v[2], rdy, K.cam_u[2] * rdx));
    V3 dd = mk(fma_(t, K.vertical[0], fma_(s, K.horizontal[0], K.llc[0])),
               fma_(t, K.vertical[1], fma_(s, K.horizontal[1], K.llc[1])),
               fma_(t, K.vertical[2], fma_(s, K.horizontal[2], K.llc[2])));
    const V3 cam_o = mk(K.origin[0], K.origin[1], K.origin[2]);
    d = mk((dd.x - cam_o.x) - off.x, (dd.y - cam_o.y) - off.y, (dd.z - cam_o.z) - off.z);
    o = mk(cam_o.x + off.x, cam_o.y + off.y, cam_o.z + off.z);
    a = dot3(d, d);
    col = mk(1.0f, 1.0f, 1.0f);
    depth = 0;
  };

  for (;;) {
    // ---- refill: lanes without a ray pull work items ---------------------------------------------
    // The wave reserves A.queue_chunk consecutive items from the global queue with ONE atomic
    // (a memory-side atomic moves 64 B, so per-item atomics would dominate the kernel's HBM
    // traffic) and deals them to its lanes from a wave-uniform local pool.
    // Lanes that finish an item wait until a few of them can be refilled together: the item decode
    // below costs ~115 VALU instructions for the whole wave whether one lane needs it or sixty,
    // and with 16-spp items about one lane per wave step does.  A wave that is mostly idle (the
    // drain of the launch, or its start) refills at once.
    for (;;) {
      bool need = !alive && !exhausted;
      unsigned long long mask = pt_ballot(need);
      if (mask == 0ull) break;
      if ((uint32_t)__popcll(mask) < K.refill_min && (uint32_t)__popcll(pt_ballot(alive)) >= 32u && refill_waited < 8u) {
        refill_waited++; // ... but not for long: with long items the next lane may be hundreds of steps away
        break;
      }
      refill_waited = 0;
      if (pool_next == pool_end) { // wave-uniform
        unsigned long long base = 0;
        if (lane == 0u) base = atomicAdd(&A.counters[PT_CTR_HEAD], (unsigned long long)A.queue_chunk);
        // wave-uniform, and known to the compiler as such (readfirstlane): the pool bookkeeping derived
        // from it then lives in SGPRs instead of occupying VGPRs for the kernel's lifetime
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        if (base >= (unsigned long long)A.n_items) { // queue dry: these lanes are done
          if constexpr (COUNT) { if (t_wave_dry == 0) t_wave_dry = __builtin_amdgcn_s_memrealtime(); }
          if (need) exhausted = true;
          continue;
        }
        pool_next = (uint32_t)base;
        unsigned long long end = base + A.queue_chunk;
        pool_end = end < (unsigned long long)A.n_items ? (uint32_t)end : A.n_items;
        // a reservation no longer than one tile's items touches at most two tiles: look their numbers
        // up once, here, instead of one dependent global load per lane in every refill
        pool_tp0 = div_(pool_next, K.div_per_tile.m, K.div_per_tile.s1, K.div_per_tile.s2);
        pool_split = (pool_tp0 + 1u) * (64u * K.n_passes);
        const uint32_t n_tiles_w = K.tiles_x * K.tiles_y;
        pool_tile0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)K.tile_order[pool_tp0]);
        pool_tile1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)K.tile_order[pool_tp0 + 1u < n_tiles_w ? pool_tp0 + 1u : pool_tp0]);
      }
      const uint32_t avail = pool_end - pool_next;
      const uint32_t cnt = (uint32_t)__popcll(mask);
      const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      const uint32_t pool_base = pool_next;
      pool_next += cnt < avail ? cnt : avail;
      if (need && rank < avail) {
        uint32_t item = pool_base + rank;
        uint32_t per_tile = 64u * K.n_passes;
        uint32_t tile_pos, tile; // heaviest tiles are dealt first (tile_order)
        if (K.queue_chunk <= per_tile) { // wave-uniform
          const bool second = item >= pool_split;
          tile_pos = pool_tp0 + (second ? 1u : 0u);
          tile = second ? pool_tile1 : pool_tile0;
        } else {
          tile_pos = div_(item, K.div_per_tile.m, K.div_per_tile.s1, K.div_per_tile.s2);
          tile = K.tile_order[tile_pos];
        }
        uint32_t rem_i = item - tile_pos * per_tile;
        uint32_t pass = rem_i >> 6, l = rem_i & 63u;
        uint32_t ty = div_(tile, K.div_tiles_x.m, K.div_tiles_x.s1, K.div_tiles_x.s2), tx = tile - ty * K.tiles_x;
        uint32_t px = tx * 8u + (l & 7u), ly = ty * 8u + (l >> 3);
        if (px < K.width && ly < K.local_rows) {
          uint32_t y = ly;
          if (K.band_count > 1u) {
            uint32_t b = div_(ly, K.div_band_rows.m, K.div_band_rows.s1, K.div_band_rows.s2), r = ly - b * K.band_rows;
            y = (b * K.band_count + K.band_index) * K.band_rows + r;
          }
          // static/shader.vert:8 + rasteriser: v_position at the pixel centre
          const float fx2 = (float)(2u * px + 1u), fy2 = (float)(2u * y + 1u); // odd integers >= 1
          float vx, vy;
          if (wh_ok) { vx = div_core(fx2, K.fw, y_fw) - 1.0f; vy = div_core(fy2, K.fh, y_fh) - 1.0f; }
          else { vx = fx2 / K.fw - 1.0f; vy = fy2 / K.fh - 1.0f; }
          float u_time = K.time0 + (float)(K.first_pass + pass) * K.time_step;
          // init_global_seed, static/shader.frag:354-357
          seed = (float)base_hash(f2u(vx), f2u(vy)) * (1.0f / 4294967296.0f) + u_time;
          st_s = (vx + 1.0f) * 0.5f; // :410
          st_t = (vy + 1.0f) * 0.5f;
          slab_index = (pass * K.local_rows + ly) * K.width + px;
          // only the launch's first pass reports its cost (atomicMax per pixel: the tile's
          // heaviest item): plenty for ordering tiles, and a memory-side atomic moves 64 B
          item_tile = pass == 0u ? tile : 0xffffffffu;
          item_segs = 0;
          sum = mk(0.f, 0.f, 0.f);
          sample = 0;
          new_path = true;
          alive = true;
        }
        // an item that falls outside the image (edge tile) is simply dropped
      }
    }
    PT_PHASE(0);
    // one copy of the camera-ray code per step serves both kinds of lanes: those that just
    // pulled an item and those whose previous path ended in the last step
    if (alive && new_path) {
      start_sample();
      new_path = false;
    }
    PT_PHASE(1);
    unsigned long long live = pt_ballot(alive);
    if (live == 0ull) break; // every lane is exhausted: the queue is dry
    if constexpr (COUNT) n_steps++;

    // ---- hit_world: static/shader.frag:175-196 over the LDS list -------------------------------
    //
    // PHASE 1, scan.  Every sphere gets the cheap part of hit_sphere (:146-153: oc, half_b, c,
    // discriminant), four spheres per trip: the next group's four ds_read_b128 are issued before
    // the current group's arithmetic (two register sets ping-pong), the four discriminants are
    // independent, and one wave-uniform branch guards the rare "discriminant not < 0" case.
    // A sphere that survives is only NOTED in a small per-lane queue (16-bit indices in three
    // VGPRs); no sqrt or division happens inside the scan.
    //
    // PHASE 2, exact.  Each lane pops its own candidates and runs the rest of hit_sphere
    // (:157-164) on them with IEEE sqrt and division.  All lanes do this in lockstep, so the
    // wave executes max-over-lanes(candidates) ~ 2-4 exact evaluations per segment instead of
    // one per distinct (lane, sphere) pair.
    //
    // Why the result is the shader's, bit for bit.  For a REGULAR ray (finite, 0 < |d|^2 < 1e6,
    // |o| < 1e15, in a scene whose spheres are finite and < 1e15 — so nothing overflows and no
    // NaN can arise) each sphere i has a candidate value v_i that does not depend on the scan
    // state: v_i = near root if near >= MIN_T else far root (far >= near because rounding is
    // monotone), and the shader accepts it iff MIN_T <= v_i <= closest-so-far.  Its loop
    // therefore returns min v_i with ties going to the largest index (:159 rejects only
    // `t_max < root`).  Any processing order over any superset of the possible winners gives
    // the same pair, provided ties are resolved the same way; phase 2 pops in DESCENDING index
    // order and accepts on strict `<` (or `<=` for the very first hit, for v == MAX_T).
    // A sphere is left out of the queue only when the shader would reject it too:
    //   - discriminant < 0 (:153), or
    //   - c > 0 and half_b >= 0: the origin is outside and the sphere is behind; then
    //     disc <= fl(half_b^2), sqrtd <= |half_b|, both numerators are <= 0 and both roots
    //     are <= 0 < MIN_T.
    // An IRREGULAR ray (NaN/Inf/zero direction, e.g. after refract() returned vec3(0)), a lane
    // whose queue overflows, or an irregular scene falls back to PHASE 3: the shader's loop
    // verbatim, in ascending order, from the first sphere the queue does not cover.
    float closest = PT_MAX_T;
    if constexpr (TREE) closest = carried ? closest_w : PT_MAX_T;
    if (!carried) hit_pos = 0xffffffffu;
    int hit = -1;
    const bool fast = A.scene_regular && (a > 1e-12f) && (a < 1e6f) &&
                      (__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)),
                                       __builtin_fabsf(o.z)) < 1e15f);
    // TAIL MODE.  When the queue is dry and only a few lanes of the wave still hold a ray, the
    // scan above would spend a whole wave on them.  Instead the wave turns around: for each
    // live ray in turn, its origin/direction are broadcast (v_readlane) and the 64 lanes test
    // 64 DIFFERENT spheres per round, run the exact part on their own candidates, and a
    // butterfly reduction picks min v with ties to the largest index — the same pair the
    // shader's loop returns (see the note above; regular rays only).  ~n/64 rounds per ray
    // instead of n tests: the heaviest items no longer set the launch's drain time.
    const int n_live = (int)__popcll(live);
    // A launch cannot end before its longest (pixel, pass) stream has run its serial course,
    // so waves carrying a long-running item get issue priority: their iterations complete
    // sooner at no cost in total throughput (the SIMD arbitrates by priority, then age).
    if (pt_ballot(alive && item_segs > A.long_item_segments) != 0ull) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
    // The walks are latency-bound (per-lane LDS gathers, short dependent loops), so they
    // want waves, i.e. few VGPRs: the part of the path state that the walk does not touch is
    // parked in LDS while it runs (14 dwords per lane, conflict-free, see `park`) and
    // fetched back for shading.  volatile: the values must not be forwarded in registers.
    if constexpr (TREE && PT_PARKING) {
      lds_u32* ps = park;
      ps[0] = f2u(sum.x); ps[1] = f2u(sum.y); ps[2] = f2u(sum.z);
      ps[3] = f2u(col.x); ps[4] = f2u(col.y); ps[5] = f2u(col.z);
      ps[6] = f2u(seed); ps[7] = f2u(st_s); ps[8] = f2u(st_t);
      ps[9] = slab_index; ps[10] = item_tile; ps[11] = item_segs;
      ps[12] = (uint32_t)sample; ps[13] = (uint32_t)depth;
    }
    const bool coop = (n_live <= (int)A.coop_max_live) && (pt_ballot(alive && !fast) == 0ull) &&
                      (pt_ballot(carried) == 0ull);
    if (coop) {
      unsigned long long todo = live;
      const uint32_t last_entry = PT_LDS_ENTRIES(n_spheres) - 1u;
      while (todo != 0ull) {
        const int L = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const float rox = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(o.x), L));
        const float roy = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(o.y), L));
        const float roz = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(o.z), L));
        const float rdx = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(d.x), L));
        const float rdy = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(d.y), L));
        const float rdz = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(d.z), L));
        const float ra = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(a), L));
        float best = PT_MAX_T;
        int best_idx = -1;
        for (uint32_t base = 0; base < n_spheres; base += 64u) {
          const uint32_t idx = base + lane;
          const float4 g = geom_at(idx < last_entry ? idx : last_entry);
          // hit_sphere :146-150 with the broadcast ray (same operation order as PT_TEST)
          const V3 oc = mk(rox - g.x, roy - g.y, roz - g.z);
          const V3 rd = mk(rdx, rdy, rdz);
          const float half_b = dot3(oc, rd);
          const float c = fma_(oc.z, oc.z, fma_(oc.y, oc.y, fma_(oc.x, oc.x, -g.w)));
          const float disc = fma_(-ra, c, half_b * half_b);
          if (idx < n_spheres && !(disc < 0.0f) && !(c > 0.0f && half_b >= 0.0f)) {
            const float sqrtd = __builtin_sqrtf(disc);
            float v = (-half_b - sqrtd) / ra;
            if (v < PT_MIN_T) v = (-half_b + sqrtd) / ra;
            if (!(v < PT_MIN_T) && v <= best) { // ascending within a lane: ties -> later sphere
              best = v;
              best_idx = (int)idx;
            }
          }
        }
        // lexicographic min of (v, ~idx) over the wave; v >= MIN_T > 0, so float bits order as uints
        uint32_t k_hi = best_idx >= 0 ? f2u(best) : 0xffffffffu;
        uint32_t k_lo = best_idx >= 0 ? 0xffffffffu - (uint32_t)best_idx : 0xffffffffu;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const uint32_t o_hi = (uint32_t)__shfl_xor((int)k_hi, off);
          const uint32_t o_lo = (uint32_t)__shfl_xor((int)k_lo, off);
          const bool take = (o_hi < k_hi) || (o_hi == k_hi && o_lo < k_lo);
          k_hi = take ? o_hi : k_hi;
          k_lo = take ? o_lo : k_lo;
        }
        if ((int)lane == L && k_hi != 0xffffffffu) {
          closest = u2f(k_hi);
          hit = (int)(0xffffffffu - k_lo);
        }
      }
    } else {
    uint32_t lit_from = fast ? 0xffffffffu : 0u; // first sphere index PHASE 3 must take over
    const bool scan_lane = alive && fast;

#define PT_TEST(G, HB, CC, DISC)                                  \
  float HB, CC, DISC;                                             \
  {                                                               \
    V3 oc = mk(o.x - G.x, o.y - G.y, o.z - G.z);                  \
    HB = dot3(oc, d);                                             \
    CC = fma_(oc.z, oc.z, fma_(oc.y, oc.y, fma_(oc.x, oc.x, -G.w))); \
    DISC = fma_(-a, CC, HB * HB);                                 \
  }

    if constexpr (BVH) {
      // PHASE 1 (hierarchy).  hit_world's result for a regular ray is the lexicographic minimum
      // of (v_i, -i) over the spheres that pass hit_sphere (note above), so the ORDER in which
      // spheres are looked at is free and spheres that cannot pass need not be looked at.  A
      // sphere can pass only if its fp32 discriminant is >= 0, and then the ray's half-line
      // comes within |r| + sqrt(E) of the centre, E = u (18 |o-C|^2 + 7 r^2), u = 2^-24 (forward
      // error of PT_TEST; the `behind` rule only removes spheres).  Every box of the tree
      // (pt_bvh.hpp, rounded outward) is therefore inflated by a per-ray margin
      //     m = 1.25e-3 (|o - c0|_1 + s0) + 1e-6     >= sqrt(E) + slab rounding
      // (|o-C| <= |o-c0| + |C-c0|, |C-c0| + |r| <= s0; sqrt(18 u) = 1.04e-3, sqrt(7 u) = 6.5e-4;
      // the 20 % on top cover the roundings of o - c0, (o - c0 +- m) / d and of the fused slab
      // form, each of relative size u, i.e. < 4u (|o - c0|_1 + s0) in space; 1e-6 keeps m
      // positive for degenerate scenes) and tested with a plain slab test.  Boxes live in the
      // frame x - c0, rounded outward (pt_bvh.hpp): fp32 for scenes whose nodes and slots fit
      // the LDS together, otherwise packed to binary16 of (x - c0) * k, which enter the fused
      // multiply-add directly (v_fma_mix_f32, half rate).  Reciprocal directions are
      // clamped to +-1e18: a component that small moves the ray by < 1e-13 over t <= MAX_T, far
      // inside m, and the clamp keeps every product finite (no 0 * inf).  A box that fails the
      // inflated test contains no sphere that could pass; a leaf that survives runs the LITERAL
      // test on its four slots.  Far-out giants (ground spheres) are not in the tree: every ray
      // tests them first, through scalar loads.
      //
      // Each lane walks the tree on its own (depth-first order with skip links: next = hit ?
      // i + 1 : skip[i]); leaves are queued (8 x 16 bit) and processed in a second lockstep
      // loop so that node steps and leaf steps do not serialise against each other.
      const uint32_t n_nodes = A.n_nodes;
      const bool fresh = scan_lane && !carried;
      const float ya = rcp_newton(a); // per-ray reciprocal for hit_root
      const uint32_t a_guard = hit_root_guard(a);

      auto eval_slot = [&](uint32_t pos) {
        const float4 g = slot_at(pos);
        PT_TEST(g, half_b, c, disc)
        (void)c;
        const float v = hit_root(half_b, disc, a, ya, a_guard); // :156-161
        // order-free form of the shader's acceptance: smaller root wins, equal roots go to the
        // LATER sphere of the list (no hit yet loses to everything, so v == MAX_T is accepted as
        // in :159).  Sphere indices are only looked up for the rare exact tie.
        bool wins = v < closest;
        if (v == closest)
          wins = hit_pos == 0xffffffffu || A.bvh_slot_index[pos] > A.bvh_slot_index[hit_pos];
        if (!(v < PT_MIN_T) && wins) {
          closest = v;
          hit_pos = pos;
        }
      };
      // pops and evaluates queued candidates while more than `keep` are queued (lockstep)
      auto drain_to = [&](uint32_t keep) {
        for (;;) {
          const unsigned long long m_q = pt_ballot(q_cnt > keep);
          if (m_q == 0ull) break;
          PT_COUNT(n_exact_it, n_exact_ln, m_q);
          if (q_cnt > keep) {
            const uint32_t pp = q0 & 0xffffu;
            q0 = __builtin_amdgcn_alignbit(q1, q0, 16);
            q1 = __builtin_amdgcn_alignbit(q2, q1, 16);
            q2 = __builtin_amdgcn_alignbit(q3, q2, 16);
            q3 >>= 16;
            q_cnt--;
            eval_slot(pp);
          }
        }
      };
      auto note_slot = [&](uint32_t pos, float half_b, float c) {
        if (c > 0.0f && half_b >= 0.0f) return; // behind the ray: both roots <= 0
        q3 = __builtin_amdgcn_alignbit(q3, q2, 16);
        q2 = __builtin_amdgcn_alignbit(q2, q1, 16);
        q1 = __builtin_amdgcn_alignbit(q1, q0, 16);
        q0 = (q0 << 16) | pos;
        q_cnt++;
      };
#define PT_SLOT_PAIR(C0, C1, BASE, ACTIVE)                                          \
  {                                                                               \
    PT_TEST(C0, hb0, cc0, ds0)                                                    \
    PT_TEST(C1, hb1, cc1, ds1)                                                    \
    if ((ACTIVE) && __builtin_fmaxf(ds0, ds1) >= 0.0f) {                          \
      if (!(ds0 < 0.0f)) note_slot((BASE) + 0u, hb0, cc0);                        \
      if (!(ds1 < 0.0f)) note_slot((BASE) + 1u, hb1, cc1);                        \
    }                                                                             \
  }
      // the outliers: wave-uniform walk (scalar loads), as the list kernels do for every sphere
      // (one at a time: there is usually exactly one, the ground); carried lanes have done this
      for (uint32_t i = A.n_tree_slots; i < A.n_tree_slots + A.n_outliers; i++) {
        if (((i - A.n_tree_slots) & 3u) == 0u && i != A.n_tree_slots) drain_to(4u); // room for four more
        const f4v e0 = c_slots[i];
        PT_TEST(e0, hb0, cc0, ds0)
        if (fresh && !(ds0 < 0.0f)) note_slot(i, hb0, cc0);
      }

      const float px = o.x - A.bvh_c0[0], py = o.y - A.bvh_c0[1], pz = o.z - A.bvh_c0[2];
      const float mrg = fma_(1.25e-3f, ((__builtin_fabsf(px) + __builtin_fabsf(py)) + __builtin_fabsf(pz)) + A.bvh_s0, 1e-6f);
      const float ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e18f, 1e18f);
      const float iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e18f, 1e18f);
      const float iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e18f, 1e18f);
      // plane parameters t = (lo / k) * i - (p + m) * i  and  t = (hi / k) * i - (p - m) * i
      // (mode 1 keeps fp32 boxes, k = 1; the packed binary16 boxes of modes 2 and 3 enter the
      // fused multiply-add directly, v_fma_mix_f32, which issues at half rate)
      const float kinv = BVH_MODE == 1 ? 1.0f : A.bvh_kinv;
      const float kx = ix * kinv, ky = iy * kinv, kz = iz * kinv;
      const float ahx = -((px + mrg) * ix), alx = -((px - mrg) * ix);
      const float ahy = -((py + mrg) * iy), aly = -((py - mrg) * iy);
      const float ahz = -((pz + mrg) * iz), alz = -((pz - mrg) * iz);
      typedef _Float16 h2v __attribute__((ext_vector_type(2)));

      // fp32 nodes: the cursor is the node's LDS address (skip links are stored as byte offsets
      // and rebased to LDS addresses when the nodes are staged); packed nodes: the node index
      constexpr uint32_t walk_step = BVH_MODE == 1 ? 32u : 1u;
      typedef float4 __attribute__((address_space(3))) lds_f4w;
      const uint32_t walk_base = BVH_MODE == 1 ? (uint32_t)(uintptr_t)(lds_f4w*)s_geom : 0u;
      const uint32_t walk_end = walk_base + n_nodes * walk_step;
      if (!carried) cur = scan_lane ? walk_base : walk_end;
      uint32_t walk_iters = 0;
      bool stop = false; // wave-uniform: the stragglers are carried into the next wave step
      for (;;) {
        // Loop-carried state changes through selects only; the one real branch is the push.  A
        // lane whose walk is over rests on the spare node behind the tree (it links to itself,
        // and `through` is masked); the loop pauses for the leaf phase as soon as ANY lane's leaf
        // queue is full, so no lane ever has to stall on its own.  With fp32 nodes `cur` is
        // the node's byte offset (skip links are stored scaled): no address arithmetic.
        for (;;) {
          const bool on = cur < walk_end;
          const unsigned long long m_on = pt_ballot(on);
          if (m_on == 0ull) break;
          if (pt_ballot(l_cnt == 8u) != 0ull) break;
          {
            const uint32_t n_on = (uint32_t)__popcll(m_on);
            if (walk_iters >= 4u && n_on < A.carry_lanes && 2u * n_on < (uint32_t)n_live) { stop = true; break; }
          }
          walk_iters++;
          PT_COUNT(n_walk_it, n_walk_ln, m_on);
          float t1x, t2x, t1y, t2y, t1z, t2z;
          uint32_t skip, leaf;
          if constexpr (BVH_MODE == 1) {
            typedef const f4v __attribute__((address_space(3))) lds_f4;
            lds_f4* np = (lds_f4*)(uintptr_t)cur; // `cur` is the node's LDS address itself
            const f4v na = np[0], nb = np[1];     // lo.xyz skip | hi.xyz leaf
            t1x = fma_(na.x, kx, ahx); t2x = fma_(nb.x, kx, alx);
            t1y = fma_(na.y, ky, ahy); t2y = fma_(nb.y, ky, aly);
            t1z = fma_(na.z, kz, ahz); t2z = fma_(nb.z, kz, alz);
            skip = f2u(na.w);
            leaf = f2u(nb.w);
          } else {
            const uint4 nd = node_at(cur);
            const h2v b0 = __builtin_bit_cast(h2v, nd.x), b1 = __builtin_bit_cast(h2v, nd.y),
                      b2 = __builtin_bit_cast(h2v, nd.z); // lo.x lo.y | lo.z hi.x | hi.y hi.z
            t1x = fma_((float)b0.x, kx, ahx); t2x = fma_((float)b1.y, kx, alx);
            t1y = fma_((float)b0.y, ky, ahy); t2y = fma_((float)b2.x, ky, aly);
            t1z = fma_((float)b1.x, kz, ahz); t2z = fma_((float)b2.y, kz, alz);
            skip = nd.w & 0xffffu;
            leaf = nd.w >> 16;
          }
          const float tn = __builtin_fmaxf(
              __builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)),
              __builtin_fmaxf(__builtin_fminf(t1z, t2z), 0.0f));
          const float tf = __builtin_fminf(
              __builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)),
              __builtin_fmaxf(t1z, t2z));
          // no relative slack on the comparison: the slab arithmetic's rounding, <= 4u (|p|_1 + s0)
          // in space, is a thousandth of the 20 % the margin carries beyond sqrt(E)
          const bool through = on && tn <= tf;
          if (through && leaf != 0xffffu) {
            l3 = __builtin_amdgcn_alignbit(l3, l2, 16);
            l2 = __builtin_amdgcn_alignbit(l2, l1, 16);
            l1 = __builtin_amdgcn_alignbit(l1, l0, 16);
            l0 = (l0 << 16) | leaf;
            l_cnt++;
          }
          cur = through ? cur + walk_step : skip;
        }
        // leaf phase: a lane takes a leaf only while its candidate queue (eight entries) has room
        // for the four a leaf can add; when the only leaves left belong to lanes with fuller
        // queues, those are drained and the loop resumes.  Pops are branch-free (a variable
        // shift; an empty queue is all zeros and stays so).
        for (;;) {
          for (;;) {
            const bool busy = (l_cnt != 0u) & (q_cnt <= 4u);
            const unsigned long long m_busy = pt_ballot(busy);
            if (m_busy == 0ull) break;
            PT_COUNT(n_leaf_it, n_leaf_ln, m_busy);
            const uint32_t base = (l0 & 0xffffu) << 2;
            const uint32_t sh = busy ? 16u : 0u;
            l0 = __builtin_amdgcn_alignbit(l1, l0, sh);
            l1 = __builtin_amdgcn_alignbit(l2, l1, sh);
            l2 = __builtin_amdgcn_alignbit(l3, l2, sh);
            l3 >>= sh;
            l_cnt -= busy ? 1u : 0u;
            // two slots at a time: the leaf phase is where register pressure peaks
            {
              const float4 g0 = slot_at(base), g1 = slot_at(base + 1u);
              PT_SLOT_PAIR(g0, g1, base, busy)
            }
            {
              const float4 g2 = slot_at(base + 2u), g3 = slot_at(base + 3u);
              PT_SLOT_PAIR(g2, g3, base + 2u, busy)
            }
          }
          if (pt_ballot(l_cnt != 0u) == 0ull) break;
          drain_to(4u);
        }
        if (stop || pt_ballot(cur < walk_end) == 0ull) break;
      }
#undef PT_SLOT_PAIR

      // PHASE 2: exact evaluation of whatever is still queued
      drain_to(0u);
      carried = cur < walk_end;
      if constexpr (COUNT) n_carried += (uint32_t)__popcll(pt_ballot(carried));
      if (hit_pos != 0xffffffffu) hit = 0; // a hit; shading reads the slot's own copies (index not needed)
    } else if constexpr (GRID) {
      // PHASE 1 (uniform grid, pt_grid.hpp).  The ORDER in which spheres are looked at is free and
      // spheres that cannot pass need not be looked at (note at PHASE 1 above).  A sphere can be
      // hit only at a point within delta of its surface (error analysis in pt_grid.hpp), hence
      // inside its bounding box inflated by delta; the grid registers every sphere in all cells
      // that box touches (inflation delta_g: the bound for rays that start within d_near of the
      // scene's middle, plus the rounding of this walk), so a ray only has to look at the
      // entries of the cells it passes through — in order, which lets it stop as soon as the
      // closest accepted root lies before the exit of the cell just finished (whatever is
      // registered only in later cells has a later root).  Every entry that is looked at runs
      // the LITERAL test; far-out giants and spheres much larger than a cell are not gridded:
      // every ray tests them first, through scalar loads.
      const uint32_t n_cell_entries = A.n_tree_slots;
      const bool fresh = scan_lane && !carried;
      const float ya = rcp_newton(a); // per-ray reciprocal for hit_root
      const uint32_t a_guard = hit_root_guard(a);

      // exact part of hit_sphere for the candidates of ONE group of four entries (4-bit mask),
      // all lanes in lockstep: max-over-lanes(popcount) ~ 1-2 evaluations per group
#define PT_EXACT_GROUP(BASE, MASK)                                                              \
  for (;;) {                                                                                    \
    const unsigned long long m_x = pt_ballot((MASK) != 0u);                                     \
    if (m_x == 0ull) break;                                                                     \
    PT_COUNT(n_exact_it, n_exact_ln, m_x);                                                      \
    if ((MASK) != 0u) {                                                                         \
      const uint32_t k = (uint32_t)__builtin_ctz(MASK);                                         \
      MASK &= MASK - 1u;                                                                        \
      const float half_b = k == 0u ? hb0 : (k == 1u ? hb1 : (k == 2u ? hb2 : hb3));             \
      const float disc = k == 0u ? ds0 : (k == 1u ? ds1 : (k == 2u ? ds2 : ds3));               \
      const float v = hit_root(half_b, disc, a, ya, a_guard); /* :156-161 */                       \
      const uint32_t pos = (BASE) + k;                                                          \
      /* order-free acceptance: smaller root wins, equal roots go to the LATER sphere of the   \
         list; indices are only looked up for the rare exact tie (a sphere registered in two   \
         cells meets ITSELF again: same index, no change) */                                    \
      bool wins = v < closest;                                                                  \
      if (v == closest)                                                                         \
        wins = hit_pos == 0xffffffffu || A.bvh_slot_index[pos] > A.bvh_slot_index[hit_pos];     \
      if (!(v < PT_MIN_T) && wins) {                                                            \
        closest = v;                                                                            \
        hit_pos = pos;                                                                          \
      }                                                                                         \
    }                                                                                           \
  }
#define PT_PASSES(HB, CC, DS) (!((DS) < 0.0f) && !((CC) > 0.0f && (HB) >= 0.0f))

      // the always-tested spheres, four at a time (wave-uniform scalar loads; the last group is
      // padded with entries that never pass); carried lanes have done this
      {
        const uint32_t n_grp = (A.n_outliers + 3u) >> 2;
        for (uint32_t gi = 0; gi < n_grp; gi++) {
          const uint32_t base = n_cell_entries + 4u * gi;
          // wave-uniform index: an LDS broadcast where the entries are staged, scalar loads otherwise
          float4 e0, e1, e2, e3;
          if constexpr (WALK == 4) {
            e0 = slot_at(base); e1 = slot_at(base + 1u); e2 = slot_at(base + 2u); e3 = slot_at(base + 3u);
          } else {
            const f4v s0 = c_slots[base], s1 = c_slots[base + 1u], s2 = c_slots[base + 2u], s3 = c_slots[base + 3u];
            e0 = make_float4(s0.x, s0.y, s0.z, s0.w); e1 = make_float4(s1.x, s1.y, s1.z, s1.w);
            e2 = make_float4(s2.x, s2.y, s2.z, s2.w); e3 = make_float4(s3.x, s3.y, s3.z, s3.w);
          }
          PT_TEST(e0, hb0, cc0, ds0)
          PT_TEST(e1, hb1, cc1, ds1)
          PT_TEST(e2, hb2, cc2, ds2)
          PT_TEST(e3, hb3, cc3, ds3)
          uint32_t mask = 0u;
          if (fresh)
            mask = (PT_PASSES(hb0, cc0, ds0) ? 1u : 0u) | (PT_PASSES(hb1, cc1, ds1) ? 2u : 0u) |
                   (PT_PASSES(hb2, cc2, ds2) ? 4u : 0u) | (PT_PASSES(hb3, cc3, ds3) ? 8u : 0u);
          PT_EXACT_GROUP(base, mask)
        }
      }

      // per-ray constants of the walk (recomputed for carried lanes: cheaper than keeping them)
      const float ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e18f, 1e18f);
      const float iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e18f, 1e18f);
      const float iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e18f, 1e18f);
      const bool posx = ix > 0.0f, posy = iy > 0.0f, posz = iz > 0.0f;
      const float tdx = K.grid_h[0] * __builtin_fabsf(ix), tdy = K.grid_h[1] * __builtin_fabsf(iy),
                  tdz = K.grid_h[2] * __builtin_fabsf(iz);
      const int gnx = (int)K.grid_n[0], gny = (int)K.grid_n[1], gnz = (int)K.grid_n[2];
      const int sdx = posx ? 1 : -1;
      const int sdy = posy ? gnx : -gnx;
      const int sdz = posz ? gnx * gny : -(gnx * gny);

      // entry: where does the half-line meet the grid's box?
      if (!carried) { gactive = false; pend = 0u; }
      if (pt_ballot(fresh) != 0ull) {
        // near rays (|o - c0| + s0 <= d_near, tested on squares: grid_r2_near = (0.9999 d_near - s0)^2):
        // every registered box lies inside [lo, hi] (delta_g is part of it); the host has widened
        // grid_lo_n / grid_hi_n by 1e-6 d_near for the rounding of this slab arithmetic.  Far
        // rays test the box inflated by their own delta(D) <= sqrt(40 u) D + 16 u rmax < 1.7e-3 D;
        // if they enter they take the literal loop (PHASE 3) over the whole list.
        const float px = o.x - K.bvh_c0[0], py = o.y - K.bvh_c0[1], pz = o.z - K.bvh_c0[2];
        const float r2 = fma_(pz, pz, fma_(py, py, px * px));
        const bool near = r2 <= K.grid_r2_near;
        float mm = 0.0f;
        if (pt_ballot(fresh && !near) != 0ull) // (rare) v_sqrt_f32 is good to 1 ulp, the factor carries 10 % slack
          mm = near ? 0.0f : 1.7e-3f * (__builtin_amdgcn_sqrtf(r2) + K.bvh_s0);
        const float oix = o.x * ix, oiy = o.y * iy, oiz = o.z * iz;
        const float t1x = fma_(K.grid_lo_n[0] - mm, ix, -oix), t2x = fma_(K.grid_hi_n[0] + mm, ix, -oix);
        const float t1y = fma_(K.grid_lo_n[1] - mm, iy, -oiy), t2y = fma_(K.grid_hi_n[1] + mm, iy, -oiy);
        const float t1z = fma_(K.grid_lo_n[2] - mm, iz, -oiz), t2z = fma_(K.grid_hi_n[2] + mm, iz, -oiz);
        const float tn = __builtin_fmaxf(
            __builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)),
            __builtin_fmaxf(__builtin_fminf(t1z, t2z), 0.0f));
        const float tf = __builtin_fminf(
            __builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)),
            __builtin_fmaxf(t1z, t2z));
        bool enter = fresh && tn <= __builtin_fminf(tf, closest);
        if (enter && !near) { // (rare) a ray from far away that does reach the grid
          lit_from = 0u;
          closest = PT_MAX_T;
          hit_pos = 0xffffffffu;
          enter = false;
        }
        if (enter) {
          gactive = true;
          // the cell that holds the entry point (clamped: rounding may put it a hair outside)
          const float glx = K.grid_lo[0], gly = K.grid_lo[1], glz = K.grid_lo[2];
          const float ghx = K.grid_h[0], ghy = K.grid_h[1], ghz = K.grid_h[2];
          const float fx = (fma_(d.x, tn, o.x) - glx) * K.grid_inv_h[0];
          const float fy = (fma_(d.y, tn, o.y) - gly) * K.grid_inv_h[1];
          const float fz = (fma_(d.z, tn, o.z) - glz) * K.grid_inv_h[2];
          const int nx1 = gnx - 1, ny1 = gny - 1, nz1 = gnz - 1;
          int cx = (int)__builtin_floorf(fx), cy = (int)__builtin_floorf(fy), cz = (int)__builtin_floorf(fz);
          cx = cx < 0 ? 0 : (cx > nx1 ? nx1 : cx);
          cy = cy < 0 ? 0 : (cy > ny1 ? ny1 : cy);
          cz = cz < 0 ? 0 : (cz > nz1 ? nz1 : cz);
          // times at which the ray crosses the cell's far planes (the side follows the sign of
          // the CLAMPED reciprocal, so a zero component gets a plane it never reaches: +-1e18 times
          // a non-negative distance) — never before the entry time
          const float bx = fma_((float)(cx + (posx ? 1 : 0)), ghx, glx);
          const float by = fma_((float)(cy + (posy ? 1 : 0)), ghy, gly);
          const float bz = fma_((float)(cz + (posz ? 1 : 0)), ghz, glz);
          tmx = __builtin_fmaxf(fma_(bx, ix, -oix), tn);
          tmy = __builtin_fmaxf(fma_(by, iy, -oiy), tn);
          tmz = __builtin_fmaxf(fma_(bz, iz, -oiz), tn);
          // steps left before the walk leaves the grid, + 1, three 10-bit fields
          rem = (uint32_t)((posx ? nx1 - cx : cx) + 1) | ((uint32_t)((posy ? ny1 - cy : cy) + 1) << 10) |
                ((uint32_t)((posz ? nz1 - cz : cz) + 1) << 20);
          cell = ((uint32_t)cz * (uint32_t)gny + (uint32_t)cy) * (uint32_t)gnx + (uint32_t)cx;
        }
      }

      PT_PHASE(2);
      uint32_t walk_iters = 0;
      for (;;) {
        // advance: a lane without a cell under test looks at the cell it stands in, notes its
        // exit time, and steps on; it leaves this loop with a non-empty cell or with its walk over
        for (;;) {
          const bool mv = gactive && (pend >> 24) == 0u;
          const unsigned long long m_mv = pt_ballot(mv);
          if (m_mv == 0ull) break;
          PT_COUNT(n_walk_it, n_walk_ln, m_mv);
          if (mv) {
            const uint32_t rec = cell_at(cell);
            const float tmin = __builtin_fminf(__builtin_fminf(tmx, tmy), tmz);
            const bool isx = tmx == tmin;
            const bool isy = !isx && tmy == tmin;
            const bool isz = !isx && !isy;
            t_exit = tmin;
            pend = rec;
            tmx += isx ? tdx : 0.0f;
            tmy += isy ? tdy : 0.0f;
            tmz += isz ? tdz : 0.0f;
            const uint32_t dec = isx ? 1u : (isy ? 1024u : 1048576u);
            rem -= dec;
            const bool out = (rem & (dec * 1023u)) == 0u;
            cell += (uint32_t)(isx ? sdx : (isy ? sdy : sdz));
            // the walk is over when it leaves the grid — or, on an empty cell, when the closest
            // root so far lies strictly before this cell's exit (a non-empty cell asks again
            // after its entries have been tested)
            if (out || ((rec >> 24) == 0u && closest < tmin)) gactive = false;
          }
        }
        PT_PHASE(3);
        const bool has = (pend >> 24) != 0u;
        const unsigned long long m_has = pt_ballot(has);
        if (m_has == 0ull) break; // no cell under test and nobody can move: every walk is over
        PT_COUNT(n_leaf_it, n_leaf_ln, m_has);
        {
          // four consecutive entries of the cell under test (those beyond its count belong to the
          // next cell or to the slack behind the array: tested, then masked)
          const uint32_t base = pend & 0xffffffu;
          const uint32_t left = pend >> 24;
          const float4 g0 = slot_at(base), g1 = slot_at(base + 1u), g2 = slot_at(base + 2u), g3 = slot_at(base + 3u);
          PT_TEST(g0, hb0, cc0, ds0)
          PT_TEST(g1, hb1, cc1, ds1)
          PT_TEST(g2, hb2, cc2, ds2)
          PT_TEST(g3, hb3, cc3, ds3)
          uint32_t mask = 0u;
          if (has) {
            mask = (PT_PASSES(hb0, cc0, ds0) ? 1u : 0u) | (PT_PASSES(hb1, cc1, ds1) ? 2u : 0u) |
                   (PT_PASSES(hb2, cc2, ds2) ? 4u : 0u) | (PT_PASSES(hb3, cc3, ds3) ? 8u : 0u);
            mask &= left >= 4u ? 0xfu : ((1u << left) - 1u);
            pend = left > 4u ? (base + 4u) | ((left - 4u) << 24) : 0u;
          }
          PT_EXACT_GROUP(base, mask)
          // the cell is done: can anything registered only in later cells still win?
          if (has && (pend >> 24) == 0u && closest < t_exit) gactive = false;
        }
        PT_PHASE(4);
        walk_iters++;
        const unsigned long long m_on = pt_ballot(gactive || (pend >> 24) != 0u);
        if (m_on == 0ull) break;
        const uint32_t n_on = (uint32_t)__popcll(m_on);
        // carry the stragglers: the longer this step's walk has run, the more lanes may be left behind
        // (a long walk means a scene of long walks, where waiting for the last quarter of the lanes costs
        // more than shading at three quarters; short walks never get past the base threshold)
        if (walk_iters >= 2u && A.carry_lanes != 0u && n_on < A.carry_lanes + 4u * (walk_iters - 2u) &&
            2u * n_on < (uint32_t)n_live) break;
      }
#undef PT_EXACT_GROUP
#undef PT_PASSES
      carried = gactive || (pend >> 24) != 0u;
      if constexpr (COUNT) n_carried += (uint32_t)__popcll(pt_ballot(carried));
      if (hit_pos != 0xffffffffu) hit = 0; // a hit; shading reads the slot's own copies (index not needed)
    } else {
    uint32_t q_cnt = 0, q0 = 0, q1 = 0, q2 = 0; // candidate queue, newest in the low half of q0
    auto note_candidate = [&](uint32_t idx, float half_b, float c) {
      if (c > 0.0f && half_b >= 0.0f) return; // behind the ray: both roots <= 0
      if (q_cnt < 6u) {
        q2 = __builtin_amdgcn_alignbit(q2, q1, 16);
        q1 = __builtin_amdgcn_alignbit(q1, q0, 16);
        q0 = (q0 << 16) | idx;
        q_cnt++;
      } else {
        // queue full (it stays full, so nothing is pushed after this): the literal loop
        // continues from the FIRST sphere that did not fit
        lit_from = idx < lit_from ? idx : lit_from;
      }
    };

#define PT_GROUP(C0, C1, C2, C3, BASE)                                            \
  {                                                                               \
    PT_TEST(C0, hb0, cc0, ds0)                                                    \
    PT_TEST(C1, hb1, cc1, ds1)                                                    \
    PT_TEST(C2, hb2, cc2, ds2)                                                    \
    PT_TEST(C3, hb3, cc3, ds3)                                                    \
    /* :153 `if (discriminant < 0.) return false;`  One compare per group: only regular   \
       lanes use the scan (lit_from == 0 sends the others to PHASE 3), and a regular ray's \
       discriminant is never NaN, so max(ds0..ds3) >= 0 <=> some ds_k is not < 0. */         \
    const float dsmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(ds0, ds1), ds2), ds3); \
    if (scan_lane && dsmax >= 0.0f) {                                             \
      const bool m0 = !(ds0 < 0.0f), m1 = !(ds1 < 0.0f), m2 = !(ds2 < 0.0f), m3 = !(ds3 < 0.0f); \
      /* padding entries (index >= n_spheres) are never candidates */             \
      if (m0 && (BASE) + 0u < n_spheres) note_candidate((BASE) + 0u, hb0, cc0);   \
      if (m1 && (BASE) + 1u < n_spheres) note_candidate((BASE) + 1u, hb1, cc1);   \
      if (m2 && (BASE) + 2u < n_spheres) note_candidate((BASE) + 2u, hb2, cc2);   \
      if (m3 && (BASE) + 3u < n_spheres) note_candidate((BASE) + 3u, hb3, cc3);   \
    }                                                                             \
  }

    {
      // pairs of groups (two ping-pong register sets), then at most one trailing group of four
      const uint32_t n_groups4 = (n_spheres + 3u) & ~3u;
      float4 a0 = geom_scan(0), a1 = geom_scan(1), a2 = geom_scan(2), a3 = geom_scan(3);
      uint32_t i = 0;
      for (; i + 8u <= n_groups4; i += 8) {
        float4 b0 = geom_scan(i + 4), b1 = geom_scan(i + 5), b2 = geom_scan(i + 6), b3 = geom_scan(i + 7);
        PT_GROUP(a0, a1, a2, a3, i)
        a0 = geom_scan(i + 8); // the list is padded by one extra group, so this stays in bounds
        a1 = geom_scan(i + 9);
        a2 = geom_scan(i + 10);
        a3 = geom_scan(i + 11);
        PT_GROUP(b0, b1, b2, b3, i + 4u)
      }
      if (i < n_groups4) PT_GROUP(a0, a1, a2, a3, i)
    }
#undef PT_GROUP

    // PHASE 2: exact evaluation of the queued candidates, newest (largest index) first
    const float ya = rcp_newton(a); // per-ray reciprocal for hit_root
    const uint32_t a_guard = hit_root_guard(a);
    while (pt_ballot(q_cnt != 0u) != 0ull) {
      if (q_cnt != 0u) {
        const uint32_t idx = q0 & 0xffffu;
        q0 = __builtin_amdgcn_alignbit(q1, q0, 16);
        q1 = __builtin_amdgcn_alignbit(q2, q1, 16);
        q2 >>= 16;
        q_cnt--;
        const float4 g = geom_at(idx);
        PT_TEST(g, half_b, c, disc) // bit-identical to the scan's values
        (void)c;
        const float v = hit_root(half_b, disc, a, ya, a_guard); // :156-161 (see the note above)
        const bool in_range = !(v < PT_MIN_T) && (v < closest || (hit < 0 && v <= closest));
        if (in_range) {
          closest = v;
          hit = (int)idx;
        }
      }
    }
    } // list scan

    // PHASE 3: the shader's loop verbatim for whatever the queue does not cover (rare)
    {
      const bool lit = alive && lit_from < n_spheres;
      unsigned long long lit_mask = pt_ballot(lit);
      if (lit_mask != 0ull) {
        // wave-uniform start: the smallest lit_from of any lane
        uint32_t start = lit ? lit_from : 0xffffffffu;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          uint32_t other = (uint32_t)__shfl_xor((int)start, off);
          start = other < start ? other : start;
        }
        start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
        for (uint32_t i = start; i < n_spheres; i++) {
          const float4 g = geom_scan(i);
          PT_TEST(g, half_b, c, disc)
          (void)c;
          if (lit && i >= lit_from && !(disc < 0.0f)) { // :153 (NaN falls through)
            const float sqrtd = __builtin_sqrtf(disc);
            float root = (-half_b - sqrtd) / a;
            bool ok = true;
            if (root < PT_MIN_T || closest < root) { // :159
              root = (-half_b + sqrtd) / a;
              if (root < PT_MIN_T || closest < root) ok = false; // :161
            }
            if (ok) {
              closest = root;
              hit = (int)i;
            }
          }
        }
      }
    }
    } // !coop
#undef PT_TEST
    if constexpr (TREE) {
      if (coop) carried = false;
      closest_w = closest;
    }

    if constexpr (TREE && PT_PARKING) {
      lds_u32* ps = park;
      sum = mk(u2f(ps[0]), u2f(ps[1]), u2f(ps[2]));
      col = mk(u2f(ps[3]), u2f(ps[4]), u2f(ps[5]));
      seed = u2f(ps[6]); st_s = u2f(ps[7]); st_t = u2f(ps[8]);
      slab_index = ps[9]; item_tile = ps[10]; item_segs = ps[11];
      sample = (int)ps[12]; depth = (int)ps[13];
    }

    PT_PHASE(5);
    // ---- shade: static/shader.frag:304-335 (carried lanes are not there yet) --------------------
    const bool shade = alive && !carried;
    seg_count += (uint32_t)__popcll(pt_ballot(shade));
    if constexpr (COUNT) { // segments per time bin (dev diagnostics: where in a launch does the rate sag?)
      const uint32_t bin = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> 16) & 63u;
      if (bin != tb_bin) {
        if (lane == 0 && tb_acc) atomicAdd(&A.counters[PT_CTR_TIMEBINS + tb_bin], (unsigned long long)tb_acc);
        tb_bin = bin; tb_acc = 0;
      }
      tb_acc += (uint32_t)__popcll(pt_ballot(shade));
    }
    if (shade) {
      item_segs++;
      bool finished = false; // this camera path is over
      if (hit < 0) {
        if (A.background_mode == 0) { // background(), :289-294
          float inv = inv_sqrt_rn(a);
          float uy = d.y * inv;
          float t = 0.5f * (uy + 1.0f);
          float omt = 1.0f - t;
          sum.x += col.x * fma_(0.5f, t, omt);
          sum.y += col.y * fma_(0.7f, t, omt);
          sum.z += col.z * fma_(1.0f, t, omt);
        }
        finished = true;
      } else {
        float4 g;
        if constexpr (TREE) { // the walk's hits come with their slot (same four floats as the list entry)
          if (hit_pos != 0xffffffffu) g = slot_at(hit_pos);
          else g = geom_at((uint32_t)hit);
        } else {
          g = geom_at((uint32_t)hit);
        }
        const float4* mp = reinterpret_cast<const float4*>(A.mat + hit);
        if constexpr (TREE) { // one load instead of index -> material (two dependent memory round trips)
          if (hit_pos != 0xffffffffu) mp = reinterpret_cast<const float4*>(A.slot_mat + hit_pos);
        }
        float4 m0 = mp[0]; // albedo.xyz, fuzz
        float4 m1 = mp[1]; // refraction_index, type, radius, uuid
        int mtype = __float_as_int(m1.y);
        float radius = m1.z;
        // hit record, :166-171
        V3 p = mk(fma_(d.x, closest, o.x), fma_(d.y, closest, o.y), fma_(d.z, closest, o.z));
        // outward normal (p - centre) / radius, :168: three divisions by one denominator.  Fast form
        // when |radius| is in [2^-20, 2^20) and every numerator has 2^-103 <= |n| < 2^76 (a zero
        // numerator takes the plain operator: its quotient's sign of zero comes from v_div_fixup)
        const float nx = p.x - g.x, ny = p.y - g.y, nz = p.z - g.z;
        V3 on;
#if PT_FAST_NORMAL
        const float n_lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(nx), __builtin_fabsf(ny)), __builtin_fabsf(nz));
        const float n_hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(nx), __builtin_fabsf(ny)), __builtin_fabsf(nz));
        const uint32_t r_guard = div_den_ok(radius) ? f2u(0x1p76f) - f2u(0x1p-103f) : 0u;
        const bool n_odd = f2u(n_lo) - f2u(0x1p-103f) >= r_guard || f2u(n_hi) - f2u(0x1p-103f) >= r_guard;
        const float yr = rcp_newton(radius);
        on = mk(div_core(nx, radius, yr), div_core(ny, radius, yr), div_core(nz, radius, yr));
        if (__builtin_expect(pt_ballot(n_odd) != 0ull, 0)) { // (rare)
          if (n_odd) on = mk(nx / radius, ny / radius, nz / radius);
        }
#else
        on = mk(nx / radius, ny / radius, nz / radius);
#endif
        bool front = dot3(d, on) < 0.0f; // :137
        V3 n = front ? on : mk(-on.x, -on.y, -on.z);
        V3 alb = mk(m0.x, m0.y, m0.z);

        if (mtype == 0 || mtype == 1) {
          V3 rs = random_in_unit_sphere(seed); // both DIFFUSE (:217) and METAL (:240) draw one
          V3 nd;
          bool ok = true;
          if (mtype == 0) { // DIFFUSE :212-229
            V3 ruv = normalize3(rs);
            nd = mk(n.x + ruv.x, n.y + ruv.y, n.z + ruv.z);
          } else { // METAL :232-247
            V3 refl = reflect3(d, n);
            float fuzz = m0.w;
            nd = mk(fma_(fuzz, rs.x, refl.x), fma_(fuzz, rs.y, refl.y), fma_(fuzz, rs.z, refl.z));
            ok = dot3(n, nd) > 0.0f;
          }
          if (ok) {
            o = p; d = nd;
            col.x *= alb.x; col.y *= alb.y; col.z *= alb.z;
          } else {
            finished = true; // absorbed: return vec3(0.) :327-329
          }
        } else if (mtype == 2) { // GLASS :250-282
          float ri = m1.x;
          float ratio = front ? (1.0f / ri) : ri;
          float inv = inv_sqrt_rn(a);
          V3 ud = mk(d.x * inv, d.y * inv, d.z * inv);
          float cdot = dot3(mk(-ud.x, -ud.y, -ud.z), n);
          float cos_theta = (1.0f < cdot) ? 1.0f : cdot; // min(cdot, 1.0)
          float sin_theta = sqrt_rn(fma_(-cos_theta, cos_theta, 1.0f));
          bool cannot_refract = ratio * sin_theta > 1.0f;
          float refl_amount = reflectance(cos_theta, ratio);
          float rnd = hash1(seed);
          V3 nd;
          if (cannot_refract || refl_amount > rnd) {
            nd = reflect3(ud, n);
          } else { // GLSL refract
            float dni = dot3(n, ud);
            float k = fma_(-(ratio * ratio), fma_(-dni, dni, 1.0f), 1.0f);
            if (k < 0.0f) {
              nd = mk(0.f, 0.f, 0.f);
            } else {
              float t = fma_(ratio, dni, sqrt_rn(k));
              nd = mk(fma_(-t, n.x, ratio * ud.x), fma_(-t, n.y, ratio * ud.y),
                      fma_(-t, n.z, ratio * ud.z));
            }
          }
          o = p; d = nd;
          col.x *= alb.x; col.y *= alb.y; col.z *= alb.z;
        } else if (mtype == 3) { // EMISSIVE (extension): radiance = throughput * emission
          sum.x += col.x * alb.x; sum.y += col.y * alb.y; sum.z += col.z * alb.z;
          finished = true;
        } else {
          finished = true; // unrecognised material absorbs, :284-285
        }

        if (!finished) {
          a = dot3(d, d);
          depth++;
          if (depth >= A.max_depth) { // loop bound :300 exhausted -> return color :338
            sum.x += col.x; sum.y += col.y; sum.z += col.z;
            finished = true;
          }
        }
      }

      if (finished) {
        sample++;
        if (sample >= A.spp) {
          float4 outv = make_float4(sum.x, sum.y, sum.z, (float)A.spp);
          reinterpret_cast<float4*>(A.slab)[slab_index] = outv;
          if (item_tile != 0xffffffffu) atomicMax(&A.tile_cost[item_tile], item_segs);
          alive = false;
        } else {
          new_path = true;
        }
      }
    }
    PT_PHASE(6);
  }

  if (lane == 0) atomicAdd(&A.counters[PT_CTR_SEGMENTS], (unsigned long long)seg_count);
  if constexpr (COUNT) {
    if (lane == 0) {
      atomicAdd(&A.counters[PT_CTR_WORK + 0], (unsigned long long)n_walk_it);
      atomicAdd(&A.counters[PT_CTR_WORK + 1], (unsigned long long)n_walk_ln);
      atomicAdd(&A.counters[PT_CTR_WORK + 2], (unsigned long long)n_leaf_it);
      atomicAdd(&A.counters[PT_CTR_WORK + 3], (unsigned long long)n_leaf_ln);
      atomicAdd(&A.counters[PT_CTR_WORK + 4], (unsigned long long)n_exact_it);
      atomicAdd(&A.counters[PT_CTR_WORK + 5], (unsigned long long)n_exact_ln);
      atomicAdd(&A.counters[PT_CTR_WORK + 6], (unsigned long long)n_steps);
      atomicAdd(&A.counters[PT_CTR_WORK + 7], (unsigned long long)n_carried);
      if (tb_acc) atomicAdd(&A.counters[PT_CTR_TIMEBINS + tb_bin], (unsigned long long)tb_acc);
      for (int k = 0; k < 7; k++) atomicAdd(&A.counters[PT_CTR_PHASES + k], ph_t[k]);
      if (A.wave_log) {
        unsigned long long* wl = A.wave_log + 3ull * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        wl[0] = t_wave_start; wl[1] = t_wave_dry; wl[2] = __builtin_amdgcn_s_memrealtime();
      }
    }
  }
  (void)n_walk_it; (void)n_walk_ln; (void)n_leaf_it; (void)n_leaf_ln; (void)n_exact_it; (void)n_exact_ln;
  (void)n_steps; (void)n_carried; (void)t_wave_start; (void)t_wave_dry; (void)tb_bin; (void)tb_acc; (void)ph_t; (void)ph_mark; (void)tmx; (void)tmy; (void)tmz; (void)t_exit; (void)cell; (void)rem;
  (void)pend; (void)gactive; (void)cur; (void)l0; (void)l1; (void)l2; (void)l3; (void)l_cnt; (void)q0; (void)q1;
  (void)q2; (void)q3; (void)q_cnt; (void)closest_w;
#undef PT_COUNT
#undef PT_PHASE
#undef lane
}

// blockDim.x is a multiple of 64 (256 normally, 1024 when the staged list is large and only one
// workgroup fits per CU); dynamic LDS = PT_LDS_ENTRIES(n_spheres) * 16 bytes.
extern "C" __global__ __launch_bounds__(1024) void pt_trace_kernel(const PtKernelArgs A) {
  pt_trace_body<true, true>(A);
}

// the scalar-load walk (PT_GEOM_SCALAR) with the LDS copy kept for the per-lane gathers
extern "C" __global__ __launch_bounds__(1024) void pt_trace_kernel_scalar(const PtKernelArgs A) {
  pt_trace_body<false, true>(A);
}

// lists beyond the LDS (10 232 < n <= 65 528): scalar-load walk, gathers from global memory
extern "C" __global__ __launch_bounds__(1024) void pt_trace_kernel_scalar_nolds(const PtKernelArgs A) {
  pt_trace_body<false, false>(A);
}

// The walk kernels are latency-bound, not issue-bound: for scenes small enough that LDS
// leaves room for them, six waves per SIMD (80 VGPRs) beat five with no spills (config 2: -5 %).
// The kernels for larger scenes are held to four waves by their LDS footprint and keep their
// registers.
#ifndef PT_BVH_WAVES
#define PT_BVH_WAVES __attribute__((amdgpu_waves_per_eu(6, 6)))
#endif
// the hierarchy walk (PT_GEOM_BVH): nodes + slots staged in LDS (dynamic LDS =
// PT_BVH_LDS_BYTES32(n_nodes, n_slots) + parking), or read from global memory / L2 when they do not fit
extern "C" __global__ __launch_bounds__(1024) PT_BVH_WAVES void pt_trace_kernel_bvh(const PtKernelArgs A) {
  pt_trace_body<false, false, 1>(A);
}
// nodes staged (dynamic LDS = (n_nodes + 1) * 16 bytes + parking), slots read from global memory
extern "C" __global__ __launch_bounds__(1024) void pt_trace_kernel_bvh_nodes(const PtKernelArgs A) {
  pt_trace_body<false, false, 2>(A);
}
extern "C" __global__ __launch_bounds__(1024) void pt_trace_kernel_bvh_gmem(const PtKernelArgs A) {
  pt_trace_body<false, false, 3>(A);
}
// the grid walk (PT_GEOM_GRID): cells + entries staged in LDS, cells only, or nothing
extern "C" __global__ __launch_bounds__(1024) PT_BVH_WAVES void pt_trace_kernel_grid(const PtKernelArgs A) {
  pt_trace_body<false, false, 4>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BVH_WAVES void pt_trace_kernel_grid_cells(const PtKernelArgs A) {
  pt_trace_body<false, false, 5>(A);
}
extern "C" __global__ __launch_bounds__(1024) void pt_trace_kernel_grid_gmem(const PtKernelArgs A) {
  pt_trace_body<false, false, 6>(A);
}
// measuring twins (PT_OPT_COUNT_WORK): the same walks with the executed-work tallies
extern "C" __global__ __launch_bounds__(1024) PT_BVH_WAVES void pt_trace_kernel_bvh_count(const PtKernelArgs A) {
  pt_trace_body<false, false, 1, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BVH_WAVES void pt_trace_kernel_grid_count(const PtKernelArgs A) {
  pt_trace_body<false, false, 4, true>(A);
}
extern "C" __global__ __launch_bounds__(1024) void pt_trace_kernel_grid_cells_count(const PtKernelArgs A) {
  pt_trace_body<false, false, 5, true>(A);
}

// --------------------------------------------------------------------------------------------
// Work-queue order for the NEXT launch: tiles sorted by the segment count of their HEAVIEST
// item in the previous launch (pass 0), largest first.  A short launch cannot end before its
// longest (pixel, pass) stream has run its serial course, so those streams must start first;
// keyed on the tile's maximum instead of its sum, a two-pass launch is 7 % shorter (the sum
// lets a tile with one very long pixel among cheap ones start late).  One 1024-thread workgroup: max ->
// 1024-bucket histogram in LDS -> scan -> scatter; then the costs are cleared.  The order only
// affects scheduling, never results (each item writes its own slab slot).
// --------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(1024) void pt_tile_order_kernel(uint32_t* cost, uint32_t* order,
                                                                        uint32_t n_tiles) {
  __shared__ uint32_t s_hist[1024];
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_max;
  const uint32_t t = threadIdx.x;
  if (t == 0) s_max = 0;
  s_hist[t] = 0;
  __syncthreads();
  uint32_t m = 0;
  for (uint32_t i = t; i < n_tiles; i += 1024) m = cost[i] > m ? cost[i] : m;
  atomicMax(&s_max, m);
  __syncthreads();
  const uint32_t mx = s_max;
  if (mx == 0) { // no feedback yet: identity order
    for (uint32_t i = t; i < n_tiles; i += 1024) order[i] = i;
    return;
  }
  const float scale = 1023.0f / (float)mx;
  for (uint32_t i = t; i < n_tiles; i += 1024) {
    uint32_t b = 1023u - (uint32_t)((float)cost[i] * scale); // heavy -> low bucket
    atomicAdd(&s_hist[b > 1023u ? 0u : b], 1u);
  }
  __syncthreads();
  // inclusive scan (Hillis-Steele), then shift to exclusive
  s_scan[t] = s_hist[t];
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t v = t >= off ? s_scan[t - off] : 0u;
    __syncthreads();
    s_scan[t] += v;
    __syncthreads();
  }
  s_hist[t] = s_scan[t] - s_hist[t]; // exclusive start of bucket t
  __syncthreads();
  for (uint32_t i = t; i < n_tiles; i += 1024) {
    uint32_t b = 1023u - (uint32_t)((float)cost[i] * scale);
    uint32_t pos = atomicAdd(&s_hist[b > 1023u ? 0u : b], 1u);
    order[pos] = i;
  }
  __syncthreads();
  for (uint32_t i = t; i < n_tiles; i += 1024) cost[i] = 0;
}

// --------------------------------------------------------------------------------------------
// accum[i] += slab[0][i] + slab[1][i] + ... in pass order (sequential fp32 adds, like n_passes
// separate pt_render calls would perform them).  16 B per lane, coalesced.
// --------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void pt_accumulate_kernel(float4* accum,
                                                                       const float4* slab,
                                                                       uint32_t n_pix,
                                                                       uint32_t n_passes) {
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    float4 acc = accum[i];
    for (uint32_t p = 0; p < n_passes; p++) {
      float4 s = slab[(size_t)p * n_pix + i];
      acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
    }
    accum[i] = acc;
  }
}

__device__ __forceinline__ uint32_t unorm8(float v) {
  if (!(v > 0.0f)) return 0u;
  if (v >= 1.0f) return 255u;
  return (uint32_t)(v * 255.0f + 0.5f);
}

// read-out, static/shader.frag:376-380 on the accumulated sum.  The divisor is the pixel's own
// sample count: accum.w carries the sum of float(spp) over the passes folded so far (an exact
// integer below 2^24), so the scale is the same fp32 value as 1/float(total spp) and it stays
// right when a captured launch is replayed by a hipGraph behind the host's back.  A pixel that
// has received nothing reads as 0.
__device__ __forceinline__ float pixel_scale(float w) { return w > 0.0f ? 1.0f / w : 0.0f; }

extern "C" __global__ __launch_bounds__(256) void pt_resolve_kernel(const float4* accum, float4* out,
                                                                    uint32_t n_pix, int gamma) {
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    float4 v = accum[i];
    const float scale = pixel_scale(v.w);
    float r = v.x * scale, g = v.y * scale, b = v.z * scale;
    if (gamma) { r = __builtin_sqrtf(r); g = __builtin_sqrtf(g); b = __builtin_sqrtf(b); }
    out[i] = make_float4(r, g, b, 1.0f);
  }
}

extern "C" __global__ __launch_bounds__(256) void pt_resolve_rgba8_kernel(const float4* accum,
                                                                          uint32_t* out, uint32_t n_pix,
                                                                          int gamma) {
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    float4 v = accum[i];
    const float scale = pixel_scale(v.w);
    float r = v.x * scale, g = v.y * scale, b = v.z * scale;
    if (gamma) { r = __builtin_sqrtf(r); g = __builtin_sqrtf(g); b = __builtin_sqrtf(b); }
    out[i] = unorm8(r) | (unorm8(g) << 8) | (unorm8(b) << 16) | (255u << 24);
  }
}

// temporal running mean of the reference, static/shader.frag:387-404 (RGBA8 ping-pong textures)
extern "C" __global__ __launch_bounds__(256) void pt_blend_rgba8_kernel(
    const float4* accum, const uint32_t* prev, uint32_t* out, uint32_t n_pix,
    int render_count, int should_average, float last_frame_weight) {
  uint32_t stride = gridDim.x * blockDim.x;
  float rc = (float)render_count;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    float4 v = accum[i];
    const float scale = pixel_scale(v.w);
    float px[3] = {__builtin_sqrtf(v.x * scale), __builtin_sqrtf(v.y * scale),
                   __builtin_sqrtf(v.z * scale)};
    uint32_t pv = prev[i];
    float pa = (float)(pv >> 24) / 255.0f;
    uint32_t o = 255u << 24;
    if (should_average && !(pa == 0.0f || render_count <= 1)) {
      float total = rc + last_frame_weight;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float pr = (float)((pv >> (8 * c)) & 255u) / 255.0f;
        float merged = fma_(px[c], last_frame_weight, pr * rc) / total;
        o |= unorm8(merged) << (8 * c);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; c++) o |= unorm8(px[c]) << (8 * c);
    }
    out[i] = o;
  }
}

// --------------------------------------------------------------------------------------------
// pt_probe: evaluate single PT-SPEC functions on the device (parity tests of SURVEY §8a rows).
// --------------------------------------------------------------------------------------------
extern "C" __global__ void pt_probe_kernel(int kind, const float* in, float* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (kind) {
    case PT_PROBE_HASH: {
      float seed = in[i];
      float* o = out + 9 * (size_t)i;
      float h1 = hash1(seed);
      o[0] = seed; o[1] = h1;
      float a2, b2;
      hash2(seed, a2, b2);
      o[2] = seed; o[3] = a2; o[4] = b2;
      float a3, b3, c3;
      hash3(seed, a3, b3, c3);
      o[5] = seed; o[6] = a3; o[7] = b3; o[8] = c3;
      break;
    }
    case PT_PROBE_SINCOS: {
      float s, c;
      sincos2pi(in[i], s, c);
      out[2 * (size_t)i] = s; out[2 * (size_t)i + 1] = c;
      break;
    }
    case PT_PROBE_CBRT: out[i] = cbrt_(in[i]); break;
    case PT_PROBE_UNIT_SPHERE: {
      float seed = in[i];
      V3 r = random_in_unit_sphere(seed);
      float* o = out + 4 * (size_t)i;
      o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = seed;
      break;
    }
    case PT_PROBE_DIVSQRT: {
      float x = in[2 * (size_t)i], y = in[2 * (size_t)i + 1];
      float* o = out + 3 * (size_t)i;
      o[0] = x / y; o[1] = __builtin_sqrtf(__builtin_fabsf(x)); o[2] = fma_(x, y, x);
      break;
    }
    case PT_PROBE_FAST_ARITH: { // the unscaled sqrt / division forms beside the plain operators
      const float x = in[3 * (size_t)i], y = in[3 * (size_t)i + 1], z = in[3 * (size_t)i + 2];
      float* o = out + 10 * (size_t)i;
      o[0] = x / y;
      o[1] = div_core(x, y, rcp_newton(y));
      o[2] = __builtin_sqrtf(x);
      o[3] = sqrt_core(x);
      o[4] = sqrt_rn(x);
      // hit_root(half_b = x, disc = y, a = z) and static/shader.frag:156-161 written out
      o[5] = hit_root(x, y, z, rcp_newton(z), hit_root_guard(z));
      const float sqrtd = __builtin_sqrtf(y);
      float v = (-x - sqrtd) / z;
      if (v < PT_MIN_T) v = (-x + sqrtd) / z;
      o[6] = v;
      o[7] = div_den_ok(y) ? 1.0f : 0.0f;
      o[8] = inv_sqrt_rn(x);
      o[9] = 1.0f / __builtin_sqrtf(x);
      break;
    }
    case PT_PROBE_BASE_HASH: {
      uint32_t h = base_hash(f2u(in[2 * (size_t)i]), f2u(in[2 * (size_t)i + 1]));
      out[i] = u2f(h);
      break;
    }
    default: break;
  }
}
