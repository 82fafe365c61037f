// pt_list.hpp — hit_world (static/shader.frag:175-196) over the sphere LIST: the scan + exact
// phases of the list kernels, the tail mode, and the literal loop every kernel falls back to.
//
// EXACTNESS ARGUMENT (the one the walk kernels build on, pt_bvh_walk.hpp / pt_grid_walk.hpp).
// For a REGULAR ray (finite, 1e-12 < |d|^2 < 1e6, |o| < 1e15, in a scene whose spheres are finite
// and < 1e15 — so nothing overflows and no NaN can arise) each sphere i has a candidate value v_i
// that does not depend on the scan state: v_i = near root if near >= MIN_T else far root (far >=
// near because rounding is monotone), and the shader accepts it iff MIN_T <= v_i <= closest-so-far.
// Its loop therefore returns min v_i with ties going to the largest index (:159 rejects only
// `t_max < root`).  Any processing order over any superset of the possible winners gives the same
// pair, provided ties are resolved the same way; the exact phase pops in DESCENDING index order and
// accepts on strict `<` (or `<=` for the very first hit, for v == MAX_T).
// A sphere is left out of the candidate queue only when the shader would reject it too:
//   - discriminant < 0 (:153), or
//   - c > 0 and half_b >= 0: the origin is outside and the sphere is behind; then
//     disc <= fl(half_b^2), sqrtd <= |half_b|, both numerators are <= 0 and both roots
//     are <= 0 < MIN_T.  (The small-list and grid kernels take this test as sign-bit arithmetic, which
//     also drops c == +0: the same three lines with `<=` for `<`.)
// An IRREGULAR ray (NaN/Inf/zero direction, e.g. after refract() returned vec3(0)), a lane whose
// queue overflows, or an irregular scene falls back to the literal loop: the shader's loop verbatim,
// in ascending order, from the first sphere the queue does not cover.
#pragma once
#include "pt_scene.hpp"

namespace ptk {

// a ray the scan / the walks may handle (see the argument above); everything else takes the literal loop
__device__ __forceinline__ bool regular_ray(const PtKernelArgs& A, const Path& p) {
  return A.scene_regular && (p.a > 1e-12f) && (p.a < 1e6f) &&
         (__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p.o.x), __builtin_fabsf(p.o.y)),
                          __builtin_fabsf(p.o.z)) < 1e15f);
}

// TAIL MODE.  When the queue is dry and only a few lanes of the wave still hold a ray, the scan
// would spend a whole wave on them.  Instead the wave turns around: for each live ray in turn, its
// origin/direction are broadcast (v_readlane) and the 64 lanes test 64 DIFFERENT spheres per
// round, run the exact part on their own candidates, and a butterfly reduction picks min v with
// ties to the largest index — the same pair the shader's loop returns (see the note above; regular
// rays only).  ~n/64 rounds per ray instead of n tests: the heaviest items no longer set the
// launch's drain time.
template <typename S>
__device__ __forceinline__ void tail_mode(const PtKernelArgs& A, const Path& p, unsigned long long live, Hit& h) {
  // (local copies, written back at the end: see pt_grid_walk.hpp)
  const V3 o = p.o; const V3 d = p.d; const float a = p.a;
  float closest = h.closest; int hit = h.hit;
  const uint32_t n_spheres = A.n_spheres;
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
      const uint32_t idx = base + lane_id();
      const float4 g = S::geom_at(A, idx < last_entry ? idx : last_entry);
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
    if ((int)lane_id() == L && k_hi != 0xffffffffu) {
      closest = u2f(k_hi);
      hit = (int)(0xffffffffu - k_lo);
    }
  }
  h.closest = closest; h.hit = hit;
}

// PHASE 1, scan.  Every sphere gets the cheap part of hit_sphere (:146-153: oc, half_b, c,
// discriminant), four spheres per trip: the next group's four reads are issued before the current
// group's arithmetic (two register sets ping-pong), the four discriminants are independent, and one
// wave-uniform branch guards the rare "discriminant not < 0" case.  A sphere that survives is only
// NOTED in a small per-lane queue (16-bit indices in three VGPRs); no sqrt or division happens
// inside the scan.
//
// PHASE 2, exact.  Each lane pops its own candidates and runs the rest of hit_sphere (:157-164) on
// them with IEEE sqrt and division.  All lanes do this in lockstep, so the wave executes
// max-over-lanes(candidates) ~ 2-4 exact evaluations per segment instead of one per distinct
// (lane, sphere) pair.
template <typename S>
__device__ __forceinline__ void list_scan(const PtKernelArgs& A, const Path& p, bool scan_lane, Hit& h) {
  const V3 o = p.o; const V3 d = p.d; const float a = p.a;
  float closest = h.closest; int hit = h.hit; uint32_t lit_from = h.lit_from;
  const uint32_t n_spheres = A.n_spheres;
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
  // four spheres of the list, base index `base`
  auto group = [&](const float4& c0, const float4& c1, const float4& c2, const float4& c3, uint32_t base) {
    float hb0, cc0, ds0; sphere_test(o, d, a, c0, hb0, cc0, ds0);
    float hb1, cc1, ds1; sphere_test(o, d, a, c1, hb1, cc1, ds1);
    float hb2, cc2, ds2; sphere_test(o, d, a, c2, hb2, cc2, ds2);
    float hb3, cc3, ds3; sphere_test(o, d, a, c3, hb3, cc3, ds3);
    // :153 `if (discriminant < 0.) return false;`  One compare per group: only regular lanes use
    // the scan (lit_from == 0 sends the others to PHASE 3), and a regular ray's discriminant is
    // never NaN, so max(ds0..ds3) >= 0 <=> some ds_k is not < 0.
    const float dsmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(ds0, ds1), ds2), ds3);
    if (scan_lane && dsmax >= 0.0f) {
      const bool m0 = !(ds0 < 0.0f), m1 = !(ds1 < 0.0f), m2 = !(ds2 < 0.0f), m3 = !(ds3 < 0.0f);
      // padding entries (index >= n_spheres) are never candidates
      if (m0 && base + 0u < n_spheres) note_candidate(base + 0u, hb0, cc0);
      if (m1 && base + 1u < n_spheres) note_candidate(base + 1u, hb1, cc1);
      if (m2 && base + 2u < n_spheres) note_candidate(base + 2u, hb2, cc2);
      if (m3 && base + 3u < n_spheres) note_candidate(base + 3u, hb3, cc3);
    }
  };
  {
    // pairs of groups (two ping-pong register sets), then at most one trailing group of four
    const uint32_t n_groups4 = (n_spheres + 3u) & ~3u;
    float4 a0 = S::geom_scan(A, 0), a1 = S::geom_scan(A, 1), a2 = S::geom_scan(A, 2), a3 = S::geom_scan(A, 3);
    uint32_t i = 0;
    for (; i + 8u <= n_groups4; i += 8) {
      float4 b0 = S::geom_scan(A, i + 4), b1 = S::geom_scan(A, i + 5), b2 = S::geom_scan(A, i + 6), b3 = S::geom_scan(A, i + 7);
      group(a0, a1, a2, a3, i);
      a0 = S::geom_scan(A, i + 8); // the list is padded by one extra group, so this stays in bounds
      a1 = S::geom_scan(A, i + 9);
      a2 = S::geom_scan(A, i + 10);
      a3 = S::geom_scan(A, i + 11);
      group(b0, b1, b2, b3, i + 4u);
    }
    if (i < n_groups4) group(a0, a1, a2, a3, i);
  }

  // PHASE 2: exact evaluation of the queued candidates, newest (largest index) first
  const float ya = rcp_newton(a); // per-ray reciprocal for hit_root
  const uint32_t a_guard = hit_root_guard(a);
  // One evaluation: the lane's newest queued candidate (bit-identical re-test, hit_root :156-161, order-free acceptance);
  // the first one stands in front of the loop (a loop's first trip costs its wave three taken branches, small_scan below).
#define PT_LIST_EXACT_STEP \
    if (q_cnt != 0u) { \
      const uint32_t idx = q0 & 0xffffu; \
      q0 = __builtin_amdgcn_alignbit(q1, q0, 16); \
      q1 = __builtin_amdgcn_alignbit(q2, q1, 16); \
      q2 >>= 16; \
      q_cnt--; \
      const float4 g = S::geom_at(A, idx); \
      float half_b, c, disc; sphere_test(o, d, a, g, half_b, c, disc); \
      (void)c; \
      const float v = hit_root(half_b, disc, a, ya, a_guard); \
      const bool in_range = !(v < PT_MIN_T) && (v < closest || (hit < 0 && v <= closest)); \
      if (in_range) { \
        closest = v; \
        hit = (int)idx; \
      } \
    }
  if (pt_ballot(q_cnt != 0u) != 0ull) {
    PT_LIST_EXACT_STEP
    while (pt_ballot(q_cnt != 0u) != 0ull) {
      PT_LIST_EXACT_STEP
    }
  }
#undef PT_LIST_EXACT_STEP
  h.closest = closest; h.hit = hit; h.lit_from = lit_from;
}

// SMALL LISTS (at most 16 spheres: the reference's own regime, `uniform Sphere[15] u_sphere_list`,
// static/shader.frag:103).  No LDS traffic in the scan, no candidate queue: the list is read four spheres at a time
// with ONE wave-uniform s_load_dwordx16 through the scalar cache (the data reach the VALU as SGPR
// operands), every group runs the literal cheap half (sphere_test) on all four, and the lanes
// finish THEIR candidates of the group at once, in lockstep, from the values still in registers
// (no re-gather, no re-test) — in ascending list order, so the acceptance is the shader's own
// sequential rule (:159-161): accept iff MIN_T <= v <= closest-so-far, which hands ties to the
// LATER sphere exactly as the shader's loop does.  A candidate is skipped only when the shader
// would reject it too (discriminant < 0, or both roots <= 0: see the note at the top).  The builds for a
// known list remainder (S::SMALL_TAIL = n mod 4, pt_kernels_small.hip) test exactly the n spheres; the build for
// any length (SMALL_TAIL < 0: the opt-in twin) tests the last group's padding entries and masks them.  Regular rays
// only; the others take the literal loop.
template <typename S>
__device__ __forceinline__ void small_scan(const PtKernelArgs& A, const Path& p, bool scan_lane, Hit& h) {
  const V3 o = p.o; const V3 d = p.d; const float a = p.a;
  float closest = h.closest; int hit = h.hit;
  const uint32_t n_spheres = A.n_spheres;
  const float ya = rcp_newton(a); // per-ray reciprocal for hit_root
  const uint32_t a_guard = hit_root_guard(a);
  const_f4v* c_geom = (const_f4v*)A.geom;
  // the candidate test as ONE bit of sign arithmetic (pt_grid_walk.hpp: `ds + 0` has a clear sign bit iff
  // !(ds < 0); a set sign bit in c or half_b means "not provably behind"; regular rays only)
  auto pass_bit = [](float hb, float cc, float ds) -> uint32_t { return ((f2u(cc) | f2u(hb)) & ~f2u(ds + 0.0f)) >> 31; };
  // the candidates of one group (mask: one bit per sphere of the group), finished in lockstep from the values still in
  // registers, in ascending list order
  auto finish = [&](uint32_t base, uint32_t mask, float hb0, float hb1, float hb2, float hb3, float ds0, float ds1, float ds2, float ds3) {
    // (the first evaluation in front of the loop: a TAKEN branch is a ~30-tick bubble in its wave, and a group with one
    // candidate per lane — the common case — now runs through without one: config 4 -2.3 %, State::default -0.9 %)
    // One evaluation: the lane's first remaining candidate (ascending list order), hit_root :156-161, accepted as the
    // shader does sequentially (:159-161: a later sphere at the same root wins).
#define PT_FINISH_STEP \
      if (mask != 0u) { \
        const uint32_t k = first_candidate(mask); \
        mask &= mask - 1u; \
        const float half_b = k == 0u ? hb0 : (k == 1u ? hb1 : (k == 2u ? hb2 : hb3)); \
        const float disc = k == 0u ? ds0 : (k == 1u ? ds1 : (k == 2u ? ds2 : ds3)); \
        const float v = hit_root(half_b, disc, a, ya, a_guard); \
        if (!(v < PT_MIN_T) && v <= closest) { \
          closest = v; \
          hit = (int)(base + k); \
        } \
      }
    if (pt_ballot(mask != 0u) != 0ull) {
      PT_FINISH_STEP
      while (pt_ballot(mask != 0u) != 0ull) {
        PT_FINISH_STEP
      }
    }
#undef PT_FINISH_STEP
  };
  if constexpr (S::SMALL_TAIL < 0) {
    // any list length: every group tests four entries (the last one's padding is masked)
#pragma unroll
    for (uint32_t g = 0; g < 4u; g++) {
      const uint32_t base = 4u * g;
      if (base < n_spheres) { // wave-uniform
        const f4v e0 = c_geom[base], e1 = c_geom[base + 1u], e2 = c_geom[base + 2u], e3 = c_geom[base + 3u];
        float hb0, cc0, ds0; sphere_test(o, d, a, e0, hb0, cc0, ds0);
        float hb1, cc1, ds1; sphere_test(o, d, a, e1, hb1, cc1, ds1);
        float hb2, cc2, ds2; sphere_test(o, d, a, e2, hb2, cc2, ds2);
        float hb3, cc3, ds3; sphere_test(o, d, a, e3, hb3, cc3, ds3);
        uint32_t mask = 0u;
        if (scan_lane) {
          mask = pass_bit(hb0, cc0, ds0) | (pass_bit(hb1, cc1, ds1) << 1) | (pass_bit(hb2, cc2, ds2) << 2) | (pass_bit(hb3, cc3, ds3) << 3);
          const uint32_t left = n_spheres - base; // >= 1, wave-uniform
          mask &= left >= 4u ? 0xfu : ((1u << left) - 1u);
        }
        finish(base, mask, hb0, hb1, hb2, hb3, ds0, ds1, ds2, ds3);
      }
    }
  } else {
    // a list of 4 q + SMALL_TAIL spheres (the build for this remainder): q full groups, nothing to mask, then a
    // last group of exactly SMALL_TAIL tests — the padding entries of a nine-sphere list (the reference's
    // State::default, config 4) cost three tests per segment otherwise
    const uint32_t n_full = n_spheres >> 2;  // wave-uniform
#pragma unroll
    for (uint32_t g = 0; g < 4u; g++) {
      if (g < n_full) { // wave-uniform
        const uint32_t base = 4u * g;
        const f4v e0 = c_geom[base], e1 = c_geom[base + 1u], e2 = c_geom[base + 2u], e3 = c_geom[base + 3u];
        float hb0, cc0, ds0; sphere_test(o, d, a, e0, hb0, cc0, ds0);
        float hb1, cc1, ds1; sphere_test(o, d, a, e1, hb1, cc1, ds1);
        float hb2, cc2, ds2; sphere_test(o, d, a, e2, hb2, cc2, ds2);
        float hb3, cc3, ds3; sphere_test(o, d, a, e3, hb3, cc3, ds3);
        uint32_t mask = pass_bit(hb0, cc0, ds0) | (pass_bit(hb1, cc1, ds1) << 1) | (pass_bit(hb2, cc2, ds2) << 2) | (pass_bit(hb3, cc3, ds3) << 3);
        mask = scan_lane ? mask : 0u;
        finish(base, mask, hb0, hb1, hb2, hb3, ds0, ds1, ds2, ds3);
      }
    }
    if constexpr (S::SMALL_TAIL > 0) {
      const uint32_t base = 4u * n_full;
      float hb0 = 0.f, cc0 = 0.f, ds0 = -1.f, hb1 = 0.f, cc1 = 0.f, ds1 = -1.f, hb2 = 0.f, cc2 = 0.f, ds2 = -1.f;
      { const f4v e0 = c_geom[base]; sphere_test(o, d, a, e0, hb0, cc0, ds0); }
      uint32_t mask = pass_bit(hb0, cc0, ds0);
      if constexpr (S::SMALL_TAIL > 1) { const f4v e1 = c_geom[base + 1u]; sphere_test(o, d, a, e1, hb1, cc1, ds1); mask |= pass_bit(hb1, cc1, ds1) << 1; }
      if constexpr (S::SMALL_TAIL > 2) { const f4v e2 = c_geom[base + 2u]; sphere_test(o, d, a, e2, hb2, cc2, ds2); mask |= pass_bit(hb2, cc2, ds2) << 2; }
      mask = scan_lane ? mask : 0u;
      finish(base, mask, hb0, hb1, hb2, hb2, ds0, ds1, ds2, ds2);
    }
  }
  h.closest = closest; h.hit = hit;
}

// PHASE 3: the shader's loop verbatim for whatever the queue / the walk does not cover (rare)
template <typename S>
__device__ __forceinline__ void literal_loop(const PtKernelArgs& A, const Path& p, Hit& h) {
  const V3 o = p.o; const V3 d = p.d; const float a = p.a; const bool alive = p.alive;
  float closest = h.closest; int hit = h.hit; const uint32_t lit_from = h.lit_from;
  const uint32_t n_spheres = A.n_spheres;
  bool lit = alive && lit_from < n_spheres;
  unsigned long long lit_mask = pt_ballot(lit);
  if (lit_mask != 0ull) {
    // A ray with a NaN in its origin or direction needs no loop: for EVERY sphere oc or half_b is NaN,
    // so c or half_b^2 and with it the discriminant are NaN, :153 does not return, both roots are NaN,
    // and NaN fails all four rejections of :159/:161 whatever closest_so_far holds — each sphere of the
    // list is accepted in turn.  The loop ends on the LAST sphere with a NaN root.  (Once a path has
    // left the real numbers — refract() returned vec3(0), :273, and the next hit point is 0 * NaN — it
    // stays there for the rest of its max_depth segments; in a scene of 10^4 spheres those segments
    // cost the whole wave 0.3 ms each through the loop below.)
    // The same holds for a direction of exactly zero (what refract() leaves behind): half_b is +-0 or
    // NaN, |d|^2 is 0, the discriminant +-0 or NaN, and (+-0 -+ 0) / 0 is NaN.
    const bool nan_ray = lit && (o.x != o.x || o.y != o.y || o.z != o.z || d.x != d.x || d.y != d.y || d.z != d.z ||
                                 (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f));
    if (nan_ray) {
      closest = __builtin_nanf("");
      hit = (int)n_spheres - 1;
      lit = false;
    }
    lit_mask = pt_ballot(lit);
  }
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
      const float4 g = S::geom_scan(A, i);
      float half_b, c, disc; sphere_test(o, d, a, g, half_b, c, disc);
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
  h.closest = closest; h.hit = hit;
}

} // namespace ptk
