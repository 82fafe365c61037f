// pt_grid_walk.hpp — PHASE 1 of hit_world through the uniform grid of pt_grid.hpp (PT_GEOM_GRID):
// which spheres a ray LOOKS AT.  Every entry that is looked at runs the LITERAL test (sphere_test +
// hit_root), so only the skipping needs an argument.
//
// EXACTNESS ARGUMENT.  The ORDER in which spheres are looked at is free and spheres that cannot
// pass need not be looked at (pt_list.hpp).  A sphere can be hit only at a point within delta of its
// surface (error analysis in pt_grid.hpp), hence inside its bounding box inflated by delta; the
// grid registers every sphere in all cells that box touches (inflation delta_g: the bound for rays
// that start within d_near of the scene's middle, plus the rounding of this walk), so a ray only
// has to look at the entries of the cells it passes through — in order, which lets it stop as soon
// as the closest accepted root lies STRICTLY before the exit of the cell just finished (whatever
// is registered only in later cells has a later root; equal roots are resolved by list index
// whatever the entry order, and a sphere met again in a second cell meets itself).  Far-out
// giants and spheres much larger than a cell are not gridded: every ray tests them first, through
// scalar loads / LDS broadcasts.  The rare ray from farther away than d_near tests the grid's box
// inflated by its own delta(D); if it enters, the ray is handed over whole: to the literal loop over the
// list, or (kernels whose entries do not fit the LDS: long lists) to the wave, 64 spheres at a time
// (pt_kernels.hip).
// Host-side check of the claim (registration invariant, the bound, a numpy emulation of this walk
// against brute force): tests/test_grid.py.
#pragma once
#include "pt_scene.hpp"

namespace ptk {

template <typename S, bool COUNT>
__device__ __forceinline__ void grid_walk(const PtKernelArgs& A, const Path& p, bool scan_lane, int n_live, Carry& cw,
                                          GridWalk& w, Hit& h, Tally<COUNT>& tally) {
  karg_t& K = *kargs();
  // the state is worked on in LOCAL copies and written back at the end: through the reference
  // parameters it would be memory to every pass that runs before this function is inlined, and the
  // loops below would be shaped (rotated, merged, made divergent) for memory operands instead of registers
  const V3 o = p.o; const V3 d = p.d; const float a = p.a;
  float closest = h.closest; int hit = h.hit; uint32_t lit_from = h.lit_from;
  bool carried = cw.carried; uint32_t hit_pos = cw.hit_pos;
  float tmx = w.tmx, tmy = w.tmy, tmz = w.tmz, t_exit = w.t_exit;
  uint32_t cell = w.cell, rem = w.rem, pend = w.pend; // rem == 0: this lane's walk is over (or has not begun)
  const uint32_t n_cell_entries = A.n_tree_slots;
  const bool fresh = scan_lane && !carried;
  const float ya = rcp_newton(a); // per-ray reciprocal for hit_root
  const uint32_t a_guard = hit_root_guard(a);

  // exact part of hit_sphere for the candidates of ONE group of four entries (4-bit mask),
  // all lanes in lockstep: max-over-lanes(popcount) ~ 1-2 evaluations per group
  auto exact_group = [&](uint32_t base, uint32_t& mask, float hb0, float hb1, float hb2, float hb3, float ds0, float ds1,
                         float ds2, float ds3) {
    // (loops on a ballot are written with the ballot as the loop condition: a wave-uniform branch on SCC)
    for (unsigned long long m_x = pt_ballot(mask != 0u); m_x != 0ull; m_x = pt_ballot(mask != 0u)) {
      tally.exact(m_x);
      if (mask != 0u) {
        const uint32_t k = first_candidate(mask);
        mask &= mask - 1u;
        const float half_b = k == 0u ? hb0 : (k == 1u ? hb1 : (k == 2u ? hb2 : hb3));
        const float disc = k == 0u ? ds0 : (k == 1u ? ds1 : (k == 2u ? ds2 : ds3));
        const float v = hit_root(half_b, disc, a, ya, a_guard); // :156-161
        const uint32_t pos = base + k;
        // order-free acceptance: smaller root wins, equal roots go to the LATER sphere of the
        // list; indices are only looked up for the rare exact tie (a sphere registered in two
        // cells meets ITSELF again: same index, no change)
        bool wins = v < closest;
        if (v == closest)
          wins = hit_pos == 0xffffffffu || A.bvh_slot_index[pos] > A.bvh_slot_index[hit_pos];
        if (!(v < PT_MIN_T) && wins) {
          closest = v;
          hit_pos = pos;
        }
      }
    }
  };
  // the literal test passes (:153) and the sphere is not behind the ray
  // ... as ONE bit: sign-bit arithmetic instead of three compares.  `ds + 0` has its sign bit clear iff
  // !(ds < 0) (a -0 from underflow becomes +0); a set sign bit in c or half_b means "not provably behind"
  // (c >= +0 and half_b >= +0 is: discriminant <= half_b^2, both roots <= 0 < MIN_T).  Regular rays only
  // (no NaN).  A -0 in c or half_b merely keeps a candidate the float compares would have dropped; c == +0
  // with half_b >= +0 is dropped here and kept there — rightly: then disc = fl(half_b^2), sqrt(disc) = |half_b|,
  // and the far root is exactly 0.  The padding entries (r^2 = -inf) have disc = -inf: sign bit set.
  auto pass_bit = [](float hb, float cc, float ds) -> uint32_t { return ((f2u(cc) | f2u(hb)) & ~f2u(ds + 0.0f)) >> 31; };

  // the always-tested spheres, four at a time (wave-uniform scalar loads; the last group is
  // padded with entries that never pass); carried lanes have done this
  tally.always_group(true);
  // A scene with ONE always-tested sphere (a ground under a field: config 5) tests it alone, not beside three padding
  // entries: -2 % there.  Only in the builds whose entries are not staged in the LDS (large scenes): in
  // pt_trace_kernel_grid the wave-uniform branch alone cost config 2 +0.8 % (four always-tested spheres: a full group).
  if (S::WALK != 4 && A.n_outliers == 1u) {
    const uint32_t base = n_cell_entries;
    float4 e0;
    if constexpr (S::WALK == 4) { e0 = S::slot_at(A, base); }
    else { const f4v s0 = S::c_slots(A)[base]; e0 = make_float4(s0.x, s0.y, s0.z, s0.w); }
    float hb0, cc0, ds0; sphere_test(o, d, a, e0, hb0, cc0, ds0);
    const bool cand = fresh && pass_bit(hb0, cc0, ds0) != 0u;
    if (pt_ballot(cand) != 0ull) {
      tally.exact(pt_ballot(cand));
      if (cand) {
        const float v = hit_root(hb0, ds0, a, ya, a_guard);
        if (!(v < PT_MIN_T) && v <= closest) {  // (the first candidate of a fresh walk: closest is still MAX_T, nothing to tie with)
          closest = v;
          hit_pos = base;
        }
      }
    }
  } else {
    const uint32_t n_grp = (A.n_outliers + 3u) >> 2;
    for (uint32_t gi = 0; gi < n_grp; gi++) {
      const uint32_t base = n_cell_entries + 4u * gi;
      // wave-uniform index: an LDS broadcast where the entries are staged, scalar loads otherwise
      float4 e0, e1, e2, e3;
      if constexpr (S::WALK == 4) {
        e0 = S::slot_at(A, base); e1 = S::slot_at(A, base + 1u); e2 = S::slot_at(A, base + 2u); e3 = S::slot_at(A, base + 3u);
      } else {
        const f4v s0 = S::c_slots(A)[base], s1 = S::c_slots(A)[base + 1u], s2 = S::c_slots(A)[base + 2u], s3 = S::c_slots(A)[base + 3u];
        e0 = make_float4(s0.x, s0.y, s0.z, s0.w); e1 = make_float4(s1.x, s1.y, s1.z, s1.w);
        e2 = make_float4(s2.x, s2.y, s2.z, s2.w); e3 = make_float4(s3.x, s3.y, s3.z, s3.w);
      }
      float hb0, cc0, ds0; sphere_test(o, d, a, e0, hb0, cc0, ds0);
      float hb1, cc1, ds1; sphere_test(o, d, a, e1, hb1, cc1, ds1);
      float hb2, cc2, ds2; sphere_test(o, d, a, e2, hb2, cc2, ds2);
      float hb3, cc3, ds3; sphere_test(o, d, a, e3, hb3, cc3, ds3);
      uint32_t mask = pass_bit(hb0, cc0, ds0) | (pass_bit(hb1, cc1, ds1) << 1) | (pass_bit(hb2, cc2, ds2) << 2) | (pass_bit(hb3, cc3, ds3) << 3);
      mask = fresh ? mask : 0u;
      exact_group(base, mask, hb0, hb1, hb2, hb3, ds0, ds1, ds2, ds3);
    }
  }
  tally.always_group(false);
  tally.phase(2);
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
  if (!carried) { rem = 0u; pend = 0u; }
  if (pt_ballot(fresh) != 0ull) {
    if (fresh) tally.flag(PT_REG_WALK_ENTRY);
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
      tally.flag(PT_REG_WALK_FAR_RAY);
      // PtStats.far_rays: how the host learns that the grid no longer fits its camera (one atomic per wave step that has such
      // lanes — the compiler folds the lanes of a uniform address into one; ~2e-5 of the rays on a fitted grid, and a ray
      // that comes here is about to be tested against the whole list)
      atomicAdd(&A.counters[PT_CTR_FAR_RAYS], 1ull);  // (priced against a build without it: config 2 -0.1 %, config 5 +0.3 %, bands +-0.1 %: noise)
      lit_from = 0u;
      closest = PT_MAX_T;
      hit_pos = 0xffffffffu;
      enter = false;
    }
    if (enter) {
      tally.flag(PT_REG_WALK_ENTER_CELL);
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

  tally.phase(3);
  uint32_t walk_iters = 0;
  const uint32_t carry_base = A.carry_lanes, carry_slope = A.carry_lanes != 0u ? 4u : 0u; // (0: nobody is left behind)
  const uint32_t half_live = ((uint32_t)n_live + 1u) >> 1;
  for (;;) {
    // advance: a lane without a cell under test looks at the cell it stands in, notes its
    // exit time, and steps on; it leaves this loop with a non-empty cell or with its walk over
    // (ballots of single compares, joined as masks: a ballot of `a && b` goes through a VGPR)
    unsigned long long m_mv = pt_ballot(rem != 0u) & pt_ballot(pend < 0x1000000u);
    // The FIRST cell step is written out in front of the loop: one step per leaf round is the common case, and a TAKEN
    // branch is a ~30-tick bubble in its wave (tools/micro/valu_chain.hip) — the loop as the compiler lays it out takes
    // three (entry, back-edge, exit) for its one trip, this form none: config 2 -0.8 %, a band of eight -1.4 %, config 5
    // -1.2 % (profiles/r04_ab_runs.txt).
#define PT_CELL_STEP \
        tally.walk(m_mv); \
        if (rem != 0u && pend < 0x1000000u) { \
          const uint32_t rec = S::cell_at(A, cell); \
          const float tmin = __builtin_fminf(__builtin_fminf(tmx, tmy), tmz); \
          const bool isx = tmx == tmin; \
          const bool isy = !isx && tmy == tmin; \
          const bool isz = !isx && !isy; \
          t_exit = tmin; \
          pend = rec; \
          tmx += isx ? tdx : 0.0f; \
          tmy += isy ? tdy : 0.0f; \
          tmz += isz ? tdz : 0.0f; \
          const uint32_t dec = isx ? 1u : (isy ? 1024u : 1048576u); \
          rem -= dec; \
          const bool out = (rem & (dec * 1023u)) == 0u; \
          cell += (uint32_t)(isx ? sdx : (isy ? sdy : sdz)); \
          rem = (out || ((rec >> 24) == 0u && closest < tmin)) ? 0u : rem; \
        }
    if (m_mv != 0ull) {
      tally.walk_first();
      PT_CELL_STEP
      m_mv = pt_ballot(rem != 0u) & pt_ballot(pend < 0x1000000u);
      while (__builtin_expect(m_mv != 0ull, 0)) {
        PT_CELL_STEP
        m_mv = pt_ballot(rem != 0u) & pt_ballot(pend < 0x1000000u);
      }
    }
#undef PT_CELL_STEP
    tally.phase(4);
    const bool has = (pend >> 24) != 0u;
    const unsigned long long m_has = pt_ballot(has);
    if (m_has == 0ull) break; // no cell under test and nobody can move: every walk is over
    tally.leaf(m_has);
    {
      // G = 4 consecutive entries of the cell under test (those beyond its count belong to the next cell or to the
      // slack behind the array: tested, then masked).  Where every lane GATHERS its entries from global memory (scenes of
      // thousands of spheres) a round costs its gather instructions — ~30 cycles of the CU's vector-memory pipe each, the
      // pipe 92 % busy on config 5 — and a cell of 3.7 entries on average would be read with fewer of them three or two
      // at a time (1.58 rounds x 3 = 4.75 or 2.12 x 2 = 4.24 gathers per visited cell instead of 1.33 x 4 = 5.31).
      // Measured (PT_LEAF_GROUP_GMEM = 3 / 2, profiles/r05_ab_runs.txt): config 5 +0.6 % / +9 %: what the rounds save in
      // gathers they cost in trips of this loop, each a dependent round trip to the L1.  Four it stays.
      constexpr uint32_t G = S::WALK == 4 ? 4u : (uint32_t)PT_LEAF_GROUP_GMEM;
      static_assert(G >= 2u && G <= 4u, "a leaf round tests two, three or four entries");
      const uint32_t base = pend & 0xffffffu;
      const uint32_t left = pend >> 24;
      tally.leaf_cells(A, has, base, A.n_slots);
      float4 g[4];
#pragma unroll
      for (uint32_t k = 0; k < G; k++) g[k] = S::slot_at(A, base + k);
      float hb[4], cc[4], ds[4];
#pragma unroll
      for (uint32_t k = 0; k < G; k++) sphere_test(o, d, a, g[k], hb[k], cc[k], ds[k]);
#pragma unroll
      for (uint32_t k = G; k < 4u; k++) { hb[k] = hb[G - 1u]; ds[k] = ds[G - 1u]; }  // (never selected: the mask has G bits)
      // (no `if (has)` around this: a lane without a cell under test has left == 0, so its mask is empty and
      // its pend becomes 0 — which is all that `pend < 2^24` meant for it)
      uint32_t mask = 0u;
#pragma unroll
      for (uint32_t k = 0; k < G; k++) mask |= pass_bit(hb[k], cc[k], ds[k]) << k;
      mask &= left >= G ? ((1u << G) - 1u) : ((1u << left) - 1u);
      pend = left > G ? (base + G) | ((left - G) << 24) : 0u;
      exact_group(base, mask, hb[0], hb[1], hb[2], hb[3], ds[0], ds[1], ds[2], ds[3]);
      // the cell is done: can anything registered only in later cells still win?
      rem = (has && pend < 0x1000000u && closest < t_exit) ? 0u : rem;
    }
    tally.phase(5);
    walk_iters++;
    const uint32_t n_on = (uint32_t)__popcll(pt_ballot(rem != 0u) | pt_ballot(pend >= 0x1000000u));
    // The loop ends when nobody walks any more — or with a few stragglers left, which are carried: the
    // longer this step's walk has run, the more lanes may be left behind (a long walk means a scene of
    // long walks, where waiting for the last quarter of the lanes costs more than shading at three
    // quarters; short walks never get past the base threshold).  ONE scalar compare: `n_on < lim` with
    // lim = max(1, min(carry_lanes + 4 (trips - 2), ceil(n_live / 2))) from the second trip on, 1 before
    // (conditions joined with && would be materialised as lane masks, a dozen scalar instructions per trip).
    const int t = (int)walk_iters - 2;
    const uint32_t thr = t >= 0 ? carry_base + carry_slope * (uint32_t)t : 0u;
    const uint32_t lim = thr < half_live ? thr : half_live;
    if (n_on < (lim > 1u ? lim : 1u)) break;
  }
  carried = rem != 0u || pend >= 0x1000000u;
  tally.carried(carried);
  if (hit_pos != 0xffffffffu) hit = 0; // a hit; shading reads the slot's own copies (index not needed)
  h.closest = closest; h.hit = hit; h.lit_from = lit_from;
  cw.carried = carried; cw.hit_pos = hit_pos;
  w.tmx = tmx; w.tmy = tmy; w.tmz = tmz; w.t_exit = t_exit;
  w.cell = cell; w.rem = rem; w.pend = pend;
}

} // namespace ptk
