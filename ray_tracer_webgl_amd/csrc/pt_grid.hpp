// pt_grid.hpp — host-side construction of the uniform grid used by the PT_GEOM_GRID trace
// kernels (pt_kernels.hip).  Header-only; included by pt_api.hip and pt_host.cpp.
//
// Like the hierarchy of pt_bvh.hpp this structure only decides WHICH spheres a ray looks at;
// every sphere that is looked at runs the literal fp32 test of static/shader.frag:145-173, and
// hit_world's result (:175-196) is order-free: the lexicographic minimum of (root, -index) over
// the spheres whose test accepts (proof sketch in pt_kernels.hip).  A grid is the better
// culling structure for what BASELINE's scenes are — many small spheres of similar size — because
// a bounce ray starts INSIDE the scene: a tree spends ~2 box tests per level (17 of the ~18 nodes
// a config-2 ray visits) just descending to the leaf around the ray's origin, a grid starts in
// the right cell.
//
// What may be skipped.  For a regular ray and sphere, with o' = o - C, forward
// error analysis of the literal test (oc, half_b, c, discriminant in fp32 with the fused forms of
// PT-SPEC, u = 2^-24) gives |disc_fp32 - disc_exact| <= |d|^2 E, E = u (18 |o'|^2 + 7 r^2), and
// for the root v the shader would accept (either sign of the square root, fp32 sqrt and divide)
//
//     |P(v) - C|^2  <=  r^2 + E',     E' = u (32 |o'|^2 + 20 r^2),     P(t) = o + t d
//
// (|P(v)-C|^2 - r^2 = |d|^2 (v - t_ca)^2 - disc_exact / |d|^2; the first term is disc_fp32 / |d|^2
// up to the rounding of half_b, |d|^2, sqrt and the division, each of relative size a few u,
// which the gap between 18/7 and 32/20 covers).  Hence a sphere can only be hit at a point within
//
//     delta(D) = sqrt(rmin^2 + 32 u D^2) - rmin + 10 u rmax,        D >= |o - C|
//
// of its surface (sqrt(r^2 + x) - r decreases with r), i.e. inside its bounding box inflated by
// delta.  With D = |o - c0| + s0 (s0 = max |C - c0| + |r| over the gridded spheres) that is a
// per-RAY bound; the grid registers every sphere in all cells its box inflated by
//
//     delta_g = delta(d_near) + eps_dda
//
// touches (each sphere with the delta of its OWN radius and distance bound in the first term: delta_i <= delta_g), and rays with |o - c0| + s0 <= d_near (= 3 s0: every ray that starts within 2 s0 of
// the scene's middle) walk the cells.  The rare ray from farther away first tests the grid's
// bounding box inflated by its own delta(D); if it misses, no gridded sphere can be hit; if it
// enters, the lane falls back to the literal loop over the whole list (measured on config 2 and
// config 5: 0 and 2e-5 of the rays).  eps_dda covers the walk's own rounding: the 3D-DDA's
// boundary-crossing times are sums of up to n per-axis increments, each rounded, so the cell
// the walk stands in at its time t is within (n + 8) u |d| t of the cell that contains the exact
// P(t); the slab test of the entry point and the evaluation of the cell boundaries are of the
// same size.  To first order, per axis k, in POSITION along the axis (time error x |d_k|), with
// B = |c0|_inf + diag >= |plane|, L = d_near + diag >= |plane - o_k| and the ray's extent in the grid:
//   first plane time fma(b, i, -fl(o i)), b = fma(cell + 1, h, lo) (u B), i = v_rcp_f32 (1 ulp = 2 u), the product
//   (u |o_k|) and the fma's own rounding (u L):                                          <= u (3 L + B + |o_k|)
//   m <= n_k running sums t += td, each rounded (u t), td = fl(h |i|) good to 3 u:         <= u (m L + 3 diag)
// together <= u (n_k + 8) (d_near + diag + |c0|_inf); the builder takes n_sum = n_x + n_y + n_z for n_k and
// EIGHT times the result (rounds 2-4: 32 times; measured by tests/test_grid.py
// test_the_walks_boundary_times_stay_within_the_rounding_budget_of_the_registration: the fp32 walk against float64
// planes, reciprocals perturbed by an ulp, scenes at the origin and 3 000 units away — the worst crossing lies
// 0.03 ... 0.065 of the bound WITHOUT the factor off its plane).  The walk ends when it leaves the grid or when the closest accepted root lies
// strictly before the current cell's exit time: a sphere not tested yet is registered only in
// cells the walk reaches later, so its root is not smaller.
//
// Spheres that would be registered in very many cells — far-out giants (the ground), and spheres
// much larger than a cell — stay out of the cells and are tested for every ray through scalar
// loads, as the list kernels test every sphere (at most kMaxAlways).
//
// Layout produced:
//   cells   : n[0]*n[1]*n[2] records, x fastest: first entry | n_entries << 24 (a leaf round of the
//             kernel tests four consecutive entries and masks those beyond the cell's count)
//   entries : {cx, cy, cz, r*r} copies, cell after cell without padding (config 2: 794 entries
//             for 480 gridded spheres — 13 KB of LDS, which is what lets six waves per SIMD
//             fit), then the always-tested spheres, padded to four with entries that can never
//             pass (r*r = -inf)
//   entry_index : original sphere index per entry (0xffffffff = padding)
//
// morton_runs() re-lays the entries out for the kernels that GATHER them from global memory / L2 (scenes whose
// entries do not fit the LDS: thousands of spheres): the cells' runs follow each other in MORTON order of their
// cells instead of x-fastest (the cell record is `first | count`, so placement is free and the walk's linear cell
// index stays), so the rays of neighbouring lanes, which stand in neighbouring cells, share cache lines in all
// three axes: config 5 -0.7 % (118.2-118.6 against 119.3 ms).  Measured beside it and NOT used: every run padded
// to four entries on a 64-byte boundary, so that a leaf round reads one aligned piece of one line per lane — +2.5 %
// (122.0-122.6 ms): the padding grows the array by 60 % and with it the misses of the 32 KB vector L1, which costs
// more than the straddled lines did (tools/sweep_knobs.py PT_PAD_RUNS, docs/HISTORY.md round 4).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace ptgrid {

constexpr uint32_t kMaxAlways = 32;   // spheres tested for every ray
constexpr uint32_t kMaxCellEntries = 255;  // entries per cell (8-bit field)
constexpr uint32_t kMaxAxis = 1023;   // cells per axis (10-bit fields in the walk's step counter)

struct Grid {
  uint32_t n[3] = {1, 1, 1};
  float lo[3] = {0, 0, 0}, h[3] = {1, 1, 1}, inv_h[3] = {1, 1, 1};
  float hi[3] = {0, 0, 0};  // lo + n * h, rounded up
  std::vector<uint32_t> cells;
  std::vector<float> entries;
  std::vector<uint32_t> entry_index;
  uint32_t n_cell_entries = 0, n_always = 0, n_entries = 0;
  float c0[3] = {0, 0, 0};
  float s0 = 0, rmin = 0, rmax = 0;
  float d_near = 0;   // rays with |o - c0|_2 + s0 <= d_near may walk the cells
  float near_factor = 3.0f;  // d_near / s0 as asked for
  float delta_g = 0;  // registration inflation
  // what the kernel's entry test uses (made here so that the kernel launch and the host emulation
  // of tests/test_grid.py read the same numbers): near rays are those with |o - c0|^2 <= r2_near
  // = ((0.9999 d_near - s0)^2, rounded down); they test the slab [lo_n, hi_n] = [lo, hi] widened by
  // 1e-6 (d_near + |c0|_inf) for the rounding of the slab arithmetic, which runs in ABSOLUTE
  // coordinates (fma(plane, 1/d, -(o * 1/d)): position error ~ 2u (|c0| + d_near))
  float r2_near = 0, lo_n[3] = {0, 0, 0}, hi_n[3] = {0, 0, 0};
  uint32_t max_cell_entries = 0, nonempty = 0;
};

inline float round_up(double v) {
  float f = (float)v;
  return (double)f < v ? std::nextafterf(f, std::numeric_limits<float>::infinity()) : f;
}
inline float round_down(double v) {
  float f = (float)v;
  return (double)f > v ? std::nextafterf(f, -std::numeric_limits<float>::infinity()) : f;
}

// Cell edge relative to the heuristic below (occupied volume / (n/2))^(1/d), i.e. about two spheres
// per cell of the occupied region.  Measured (grid walk, carry-over on): config 2 149.5 / 145.0 /
// 138.8 / 136.4 / 146.0 ms and config 5 - / 267.9 / 243.5 / 232.4 / 237.3 ms at 0.6 / 0.7 / 0.85 / 1.0 /
// 1.2 — fewer, fuller cells: fewer steps and leaf rounds, a second round only in the fullest.
// (A/B builds can override it.)
inline double edge_scale() {
#ifdef PT_DEV_KNOBS
  if (const char* e = std::getenv("PT_GRID_EDGE")) { const double v = std::atof(e); if (v > 0.05 && v < 20.0) return v; }
#endif
  return 1.0;
}

// delta(D) as the kernel evaluates it must not be smaller than this (the kernel adds slack)
inline double delta_of(double rmin, double rmax, double D) {
  const double u = 5.9604644775390625e-08;
  return std::sqrt(rmin * rmin + 32.0 * u * D * D) - rmin + 10.0 * u * rmax;
}

// geom: n x {cx, cy, cz, r*r} as the list kernels read it; radius: n signed radii; all finite
// (the caller only builds for scenes it classified as regular).  Returns false when a grid
// would be useless or cannot be represented; the other paths are used then.
// near_factor: rays that start within (near_factor - 1) s0 of the scene's middle walk the cells (d_near = near_factor s0).
// A matter of speed only: rays from farther away are handled exactly, slowly (pt_grid_walk.hpp), and the margin's first
// term grows with d_near^2.  3 unless the caller knows where the rays come from (pt_api.hip fit_grid_to_view).
inline bool build(const float* geom, const float* radius, uint32_t n, Grid* out, double near_factor = 3.0) {
  *out = Grid();
  if (n < 16) return false;
  // ---- far-out giants (same rule as pt_bvh.hpp) ---------------------------------------------
  double med[3];
  {
    std::vector<float> tmp(n);
    for (int k = 0; k < 3; k++) {
      for (uint32_t i = 0; i < n; i++) tmp[i] = geom[4 * i + k];
      std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
      med[k] = tmp[n / 2];
    }
  }
  std::vector<double> reach(n);
  for (uint32_t i = 0; i < n; i++) {
    double dx = geom[4 * i] - med[0], dy = geom[4 * i + 1] - med[1], dz = geom[4 * i + 2] - med[2];
    reach[i] = std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)radius[i]);
  }
  std::vector<uint32_t> by_reach(n);
  for (uint32_t i = 0; i < n; i++) by_reach[i] = i;
  std::sort(by_reach.begin(), by_reach.end(), [&](uint32_t a, uint32_t b) { return reach[a] < reach[b]; });
  const double ref = reach[by_reach[(size_t)(0.9 * (n - 1))]];
  std::vector<uint8_t> always(n, 0);
  uint32_t n_always = 0;
  for (uint32_t k = n; k-- > 0 && n_always < kMaxAlways;) {
    if (reach[by_reach[k]] > 8.0 * ref) { always[by_reach[k]] = 1; n_always++; } else break;
  }
  if (n - n_always < 8) return false;

  // ---- cell edge: about two spheres per cell of the occupied volume; flat axes get one layer --
  std::vector<float> rs;
  double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
  for (uint32_t i = 0; i < n; i++) {
    if (always[i]) continue;
    rs.push_back(std::fabs(radius[i]));
    for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], (double)geom[4 * i + k]); chi[k] = std::max(chi[k], (double)geom[4 * i + k]); }
  }
  std::nth_element(rs.begin(), rs.begin() + rs.size() / 2, rs.end());
  const double rmed = std::max((double)rs[rs.size() / 2], 1e-30);
  const double cnt = (double)rs.size();
  double ext[3];
  for (int k = 0; k < 3; k++) ext[k] = std::max(chi[k] - clo[k], 2.0 * rmed);
  double edge = std::cbrt(ext[0] * ext[1] * ext[2] / (cnt / 2.0));
  for (int it = 0; it < 3; it++) {
    double v = 1.0; int free_axes = 0;
    for (int k = 0; k < 3; k++) if (ext[k] > 1.5 * edge) { v *= ext[k]; free_axes++; }
    if (free_axes) edge = std::pow(v / (cnt / 2.0), 1.0 / free_axes);
  }
  edge *= edge_scale();
  if (!(edge > 0.0) || !std::isfinite(edge)) return false;

  for (int attempt = 0; attempt < 4; attempt++, edge *= 0.6) {
    // spheres much larger than a cell would be copied into many cells: test them for every ray
    std::vector<uint8_t> alw = always;
    uint32_t n_alw = n_always;
    {
      std::vector<uint32_t> big;
      for (uint32_t i = 0; i < n; i++) if (!alw[i] && std::fabs((double)radius[i]) > 0.5 * edge) big.push_back(i);
      std::sort(big.begin(), big.end(), [&](uint32_t a, uint32_t b) { return std::fabs(radius[a]) > std::fabs(radius[b]); });
      for (uint32_t i : big) {
        if (n_alw >= kMaxAlways) break;  // the rest is gridded, at the price of copies
        if (std::fabs((double)radius[i]) > 1.0 * edge || big.size() <= kMaxAlways - n_always) { alw[i] = 1; n_alw++; }
      }
    }
    if (n - n_alw < 8) return false;
    // ---- margin constants over the gridded spheres ----------------------------------------
    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
    double rmin = 1e300, rmax = 0.0;
    clo[0] = clo[1] = clo[2] = 1e300; chi[0] = chi[1] = chi[2] = -1e300;
    for (uint32_t i = 0; i < n; i++) {
      if (alw[i]) continue;
      const double r = std::fabs((double)radius[i]);
      rmin = std::min(rmin, r); rmax = std::max(rmax, r);
      for (int k = 0; k < 3; k++) {
        clo[k] = std::min(clo[k], (double)geom[4 * i + k]); chi[k] = std::max(chi[k], (double)geom[4 * i + k]);
        blo[k] = std::min(blo[k], (double)geom[4 * i + k] - r); bhi[k] = std::max(bhi[k], (double)geom[4 * i + k] + r);
      }
    }
    Grid g;
    for (int k = 0; k < 3; k++) g.c0[k] = (float)(0.5 * (clo[k] + chi[k]));
    double s0 = 0.0;
    for (uint32_t i = 0; i < n; i++) {
      if (alw[i]) continue;
      double dx = geom[4 * i] - (double)g.c0[0], dy = geom[4 * i + 1] - (double)g.c0[1], dz = geom[4 * i + 2] - (double)g.c0[2];
      s0 = std::max(s0, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)radius[i]));
    }
    g.s0 = round_up(s0 * (1.0 + 1e-6));
    g.rmin = round_down(rmin); g.rmax = round_up(rmax);
#ifdef PT_DEV_KNOBS
    if (const char* e = std::getenv("PT_GRID_DNEAR")) { const double v = std::atof(e); if (v >= 2.0 && v <= 6.0) near_factor = v; }
#endif
    g.d_near = round_up(near_factor * (double)g.s0);
    g.near_factor = (float)near_factor;
    // ---- resolution --------------------------------------------------------------------------
    double diag = 0.0;
    uint32_t n_sum = 0;
    double dg = 0.0;
    bool ok = true;
    // the bounds are built with an inflation dg that must cover what it needs itself (it depends,
    // weakly, on the resolution and the extent): iterate to a fixed point
    for (int pass = 0; pass < 16; pass++) {
      diag = 0.0; n_sum = 0; ok = true;
      const double dgr = (double)round_up(dg);  // the inflation as it is stored (delta_g): boxes registered with it stay inside [lo, hi]
      for (int k = 0; k < 3; k++) {
        const double a = blo[k] - dgr, b = bhi[k] + dgr;
        double cells = std::floor((b - a) / edge + 0.5);
        if (cells < 1.0) cells = 1.0;
        if (cells > (double)kMaxAxis) { ok = false; cells = kMaxAxis; }
        g.n[k] = (uint32_t)cells;
        g.lo[k] = round_down(a);
        g.h[k] = round_up((b - (double)g.lo[k]) / cells * (1.0 + 1e-6));
        g.inv_h[k] = (float)(1.0 / (double)g.h[k]);
        g.hi[k] = round_up((double)g.lo[k] + cells * (double)g.h[k]);
        diag += ((double)g.hi[k] - g.lo[k]) * ((double)g.hi[k] - g.lo[k]);
        n_sum += g.n[k];
      }
      diag = std::sqrt(diag);
      const double u = 5.9604644775390625e-08;
      // the walk's plane times are evaluated in absolute coordinates, so their position error scales
      // with |c0| + d_near (a scene centred far from the origin), not with d_near alone
      const double c0abs = std::max(std::fabs((double)g.c0[0]), std::max(std::fabs((double)g.c0[1]), std::fabs((double)g.c0[2])));
      const double eps_dda = 8.0 * (n_sum + 8.0) * u * ((double)g.d_near + diag + c0abs);
      // the kernel's per-ray delta carries 25 % slack on E' and is compared against this value
      const double need = (std::sqrt(rmin * rmin + 32.0 * 1.25 * u * (double)g.d_near * g.d_near) - rmin + 16.0 * u * rmax) + eps_dda;
      if (need <= dg) break;
      dg = need * 1.02;
      ok = false;  // not settled yet
    }
    if (!ok) return false;
    g.delta_g = round_up(dg);
    const size_t n_cells = (size_t)g.n[0] * g.n[1] * g.n[2];
    if (n_cells > (1u << 22)) return false;
    // ---- registration --------------------------------------------------------------------------
    std::vector<uint32_t> count(n_cells, 0);
    // (the inflation of sphere i: delta with ITS radius and ITS distance bound in the first term — sqrt(r^2 + x) - r
    // decreases with r, a sphere of five times rmin needs less than half of delta_g's first term; and a walking ray starts
    // within d_near - s0 of c0 (the kernel's near test), hence within D_i = d_near - s0 + |C_i - c0| <= d_near of C_i —
    // plus what delta_g carries beyond that term for the walk's rounding; never more than delta_g, which stays the bound
    // of every registered box.  Config 5, radii 0.1 ... 0.5 in a scene of extent 150: 24 127 -> 21 423 (own radius) ->
    // 20 504 (own distance) entries in the cells, 1.51 -> 1.40 -> 1.37 leaf rounds per visited cell)
    const double u40 = 40.0 * 5.9604644775390625e-08;
    const double dg_walk = (double)g.delta_g - (std::sqrt(rmin * rmin + u40 * (double)g.d_near * (double)g.d_near) - rmin);
    std::vector<double> infl(n, 0.0);
    for (uint32_t i = 0; i < n; i++) {
      if (alw[i]) continue;
      const double ri = std::fabs((double)radius[i]);
      const double dx = geom[4 * i] - (double)g.c0[0], dy = geom[4 * i + 1] - (double)g.c0[1], dz = geom[4 * i + 2] - (double)g.c0[2];
      const double Di = std::min((double)g.d_near, ((double)g.d_near - (double)g.s0 + std::sqrt(dx * dx + dy * dy + dz * dz)) * (1.0 + 1e-6));
      infl[i] = std::min((double)g.delta_g, (std::sqrt(ri * ri + u40 * Di * Di) - ri) * (1.0 + 1e-9) + dg_walk);
    }
    auto range = [&](uint32_t i, int k, uint32_t& a, uint32_t& b) {
      const double r = std::fabs((double)radius[i]) + infl[i];
      double fa = std::floor(((double)geom[4 * i + k] - r - (double)g.lo[k]) / (double)g.h[k]);
      double fb = std::floor(((double)geom[4 * i + k] + r - (double)g.lo[k]) / (double)g.h[k]);
      fa = std::min(std::max(fa, 0.0), (double)g.n[k] - 1.0);
      fb = std::min(std::max(fb, 0.0), (double)g.n[k] - 1.0);
      a = (uint32_t)fa; b = (uint32_t)fb;
    };
    for (int fill = 0; fill < 2; fill++) {
      std::vector<uint32_t> cursor;
      if (fill) {
        g.cells.assign(n_cells, 0);
        uint32_t first = 0;
        g.max_cell_entries = 0; g.nonempty = 0;
        for (size_t c = 0; c < n_cells; c++) {
          if (count[c] > kMaxCellEntries) { ok = false; break; }
          g.cells[c] = first | (count[c] << 24);
          first += count[c];
          g.max_cell_entries = std::max(g.max_cell_entries, count[c]);
          g.nonempty += count[c] ? 1u : 0u;
          if (first >= (1u << 24) - 8u) { ok = false; break; }
        }
        if (!ok) break;
        g.n_cell_entries = first;
        g.entries.assign((size_t)g.n_cell_entries * 4, 0.f);
        g.entry_index.assign(g.n_cell_entries, 0xffffffffu);
        cursor.assign(n_cells, 0);
      }
      for (uint32_t i = 0; i < n; i++) {  // ascending index: deterministic layout
        if (alw[i]) continue;
        uint32_t a[3], b[3];
        for (int k = 0; k < 3; k++) range(i, k, a[k], b[k]);
        for (uint32_t z = a[2]; z <= b[2]; z++)
          for (uint32_t y = a[1]; y <= b[1]; y++)
            for (uint32_t x = a[0]; x <= b[0]; x++) {
              const size_t c = ((size_t)z * g.n[1] + y) * g.n[0] + x;
              if (!fill) { count[c]++; continue; }
              const uint32_t e = (g.cells[c] & 0xffffffu) + cursor[c]++;
              std::memcpy(&g.entries[4 * (size_t)e], geom + 4 * (size_t)i, 16);
              g.entry_index[e] = i;
            }
      }
    }
    if (!ok) continue;  // a cell overflowed: finer cells
    // ---- the always-tested spheres, ascending index, padded to four -----------------------------
    for (uint32_t i = 0; i < n; i++) {
      if (!alw[i]) continue;
      g.entries.insert(g.entries.end(), geom + 4 * (size_t)i, geom + 4 * (size_t)i + 4);
      g.entry_index.push_back(i);
    }
    g.n_always = n_alw;
    while (((g.entry_index.size() - g.n_cell_entries) & 3u) != 0) {
      const float pad[4] = {0.f, 0.f, 0.f, -std::numeric_limits<float>::infinity()};
      g.entries.insert(g.entries.end(), pad, pad + 4);
      g.entry_index.push_back(0xffffffffu);
    }
    g.n_entries = (uint32_t)g.entry_index.size();
    if (g.n_entries > (1u << 24)) return false;
    {
      const double c0abs = std::max(std::fabs((double)g.c0[0]), std::max(std::fabs((double)g.c0[1]), std::fabs((double)g.c0[2])));
      const double rn = 0.9999 * (double)g.d_near - (double)g.s0;
      g.r2_near = round_down(rn * rn * (1.0 - 1e-6));
      const double widen = 1e-6 * ((double)g.d_near + c0abs) + 1e-30;
      for (int k = 0; k < 3; k++) {
        g.lo_n[k] = round_down((double)g.lo[k] - widen);
        g.hi_n[k] = round_up((double)g.hi[k] + widen);
      }
    }
    *out = std::move(g);
    return true;
  }
  return false;
}

// interleave the low 10 bits of x, y, z (cells per axis <= kMaxAxis = 1023)
inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  auto spread = [](uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
  };
  return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

// The layout for entries gathered from global memory (see the header comment): the cells' runs in Morton order
// of their cells (`pad`: also padded to four entries — measured, not used); the always-tested group follows as
// before.  Which spheres a cell holds, and in which order, does not change — only where its run lies.  Returns
// false (and leaves the grid as it was) when the array would not fit the 24-bit entry field.
inline bool morton_runs(Grid* g, bool pad = false, bool morton = true) {
  const size_t n_cells = g->cells.size();
  std::vector<uint32_t> order;
  order.reserve(g->nonempty);
  size_t padded = 0;
  for (size_t c = 0; c < n_cells; c++) {
    const uint32_t cnt = g->cells[c] >> 24;
    if (cnt) { order.push_back((uint32_t)c); padded += pad ? ((cnt + 3u) & ~3u) : cnt; }
  }
  const size_t n_tail = (size_t)g->n_entries - g->n_cell_entries;  // the always-tested group, padded to four already
  if (padded + n_tail >= (1u << 24) - 8u) return false;
  const uint32_t nx = g->n[0], ny = g->n[1];
  auto key = [&](uint32_t c) { return morton3(c % nx, (c / nx) % ny, c / (nx * ny)); };
  if (morton) std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
  std::vector<float> entries((padded + n_tail) * 4);
  std::vector<uint32_t> index(padded + n_tail, 0xffffffffu);
  const float never[4] = {0.f, 0.f, 0.f, -std::numeric_limits<float>::infinity()};
  for (size_t e = 0; e < padded; e++) std::memcpy(&entries[4 * e], never, 16);
  uint32_t first = 0;
  for (uint32_t c : order) {
    const uint32_t old = g->cells[c] & 0xffffffu, cnt = g->cells[c] >> 24;
    std::memcpy(&entries[4 * (size_t)first], &g->entries[4 * (size_t)old], (size_t)cnt * 16);
    std::memcpy(&index[first], &g->entry_index[old], (size_t)cnt * 4);
    g->cells[c] = first | (cnt << 24);
    first += pad ? ((cnt + 3u) & ~3u) : cnt;
  }
  std::memcpy(&entries[4 * padded], &g->entries[4 * (size_t)g->n_cell_entries], n_tail * 16);
  std::memcpy(&index[padded], &g->entry_index[g->n_cell_entries], n_tail * 4);
  g->entries.swap(entries);
  g->entry_index.swap(index);
  g->n_cell_entries = (uint32_t)padded;
  g->n_entries = (uint32_t)(padded + n_tail);
  return true;
}

}  // namespace ptgrid
