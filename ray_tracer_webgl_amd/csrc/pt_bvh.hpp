// pt_bvh.hpp — host-side construction of the culling hierarchy used by the PT_GEOM_BVH trace
// kernels (pt_kernels.hip).  Header-only; included by pt_api.hip.
//
// The reference's hit_world (static/shader.frag:175-196) tests every sphere for every ray.  Its
// RESULT is the lexicographic minimum of (root, -index) over the spheres whose hit_sphere
// accepts (see the proof sketch in pt_kernels.hip), so any procedure that evaluates the exact
// fp32 test on a SUPERSET of the spheres that can pass returns the same bits.  The hierarchy
// below only decides which spheres are looked at; every sphere that is looked at still runs
// the literal arithmetic.  What makes that safe is a conservative bound:
//
//   literal fp32 discriminant >= 0   ==>   the ray's supporting half-line passes within
//       |r| + sqrt(E),  E = u (18 |o-C|^2 + 7 r^2),  u = 2^-24,
//   of the centre C (forward error analysis of PT_TEST under -ffp-contract=off),
//
// so the kernel inflates every box by a per-ray margin m >= sqrt(E) (plus the rounding of its own
// slab arithmetic) before testing it.  m grows with the distance between the ray origin and the
// spheres, hence a few far-out giants (a ground sphere of radius 1000) would inflate everything:
// such OUTLIERS are kept out of the tree and are tested for every ray, like the reference does.
//
// Layout produced:
//   nodes : n_nodes x 8 floats  {lo.x, lo.y, lo.z, bits(skip), hi.x, hi.y, hi.z, bits(leaf)}
//           in depth-first order: the left child of node i is i+1; `skip` is the node that
//           follows i's subtree (n_nodes ends the walk); leaf = first slot of a leaf (a multiple
//           of 4) or 0xffffffff for an inner node.  (Host form: tests, pt_build_bvh.)
//   nodes32 : what the kernel for small scenes reads — {lo - c0, bits(32 * skip), hi - c0,
//           bits(leaf_number)}, fp32 rounded outward, plus the spare node described below.
//   nodes16 : what the kernels for larger scenes read — the same nodes packed into 16 bytes: six binary16 box
//           coordinates {lo.x|lo.y, lo.z|hi.x, hi.y|hi.z} in the frame (x - c0) * kscale
//           (kscale a power of two), each rounded OUTWARD, then skip | leaf_number << 16
//           (leaf_number = first slot / 4, 0xffff for an inner node); one spare node at the
//           end, which lanes that have finished their walk keep reading.  Coarser boxes only
//           cost a few extra visits: at the rim of the scene a binary16 step is s0 / 2048,
//           below the per-ray margin of 1.25e-3 s0.
//   slots : n_slots x {cx, cy, cz, r*r}; four slots per leaf, unused ones hold a sphere that can
//           never pass (r*r = -inf -> discriminant = -inf); the outliers follow the leaves, also
//           in groups of four.  slots[n_tree_slots .. n_slots) is the brute-force part.
//   slot_index : original sphere index of each slot (0xffffffff for padding).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace ptbvh {

constexpr uint32_t kLeafSize = 4;      // spheres per leaf == slots per leaf
constexpr uint32_t kMaxOutliers = 32;  // tested for every ray
constexpr uint32_t kInner = 0xffffffffu;

struct Bvh {
  std::vector<float> nodes;        // 8 per node
  std::vector<uint32_t> nodes16;   // 4 per node, n_nodes + 1 records
  std::vector<float> nodes32;      // 8 per node, n_nodes + 1 records: fp32 boxes in the frame x - c0
  float kscale = 1.0f, kinv = 1.0f;  // box frame: (x - c0) * kscale, kinv = 1 / kscale
  std::vector<float> slots;        // 4 per slot
  std::vector<uint32_t> slot_index;
  uint32_t n_nodes = 0, n_slots = 0, n_tree_slots = 0, n_outliers = 0, depth = 0;
  float c0[3] = {0, 0, 0};  // reference point of the per-ray margin
  float s0 = 0.0f;          // max over tree spheres of |C - c0| + |r|, rounded up
};

struct Box {
  float lo[3], hi[3];
  void clear() {
    for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; }
  }
  void grow(const Box& b) {
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
  }
  double half_area() const {
    double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
};

inline float round_down(double v) {
  float f = (float)v;
  return (double)f > v ? std::nextafterf(f, -std::numeric_limits<float>::infinity()) : f;
}
inline float round_up(double v) {
  float f = (float)v;
  return (double)f < v ? std::nextafterf(f, std::numeric_limits<float>::infinity()) : f;
}
inline uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// ---- binary16 with directed rounding (no _Float16 on the host side) ---------------------------
inline float half_to_float(uint16_t h) {
  const uint32_t sgn = (h >> 15) & 1u, e = (h >> 10) & 31u, m = h & 1023u;
  float v;
  if (e == 0) v = std::ldexp((float)m, -24);
  else if (e == 31) v = m ? std::numeric_limits<float>::quiet_NaN() : std::numeric_limits<float>::infinity();
  else v = std::ldexp((float)(m + 1024u), (int)e - 25);
  return sgn ? -v : v;
}
inline uint16_t half_next_up(uint16_t h) {    // next value towards +inf
  if (h == 0x8000u) h = 0;
  return (h & 0x8000u) ? (uint16_t)(h - 1) : (uint16_t)(h + 1);
}
inline uint16_t half_next_down(uint16_t h) {  // next value towards -inf
  if (h == 0) h = 0x8000u;
  return (h & 0x8000u) ? (uint16_t)(h + 1) : (uint16_t)(h - 1);
}
inline uint16_t half_nearest(double v) {
  if (v == 0.0 || v != v) return 0;
  const uint16_t sgn = v < 0 ? 0x8000u : 0;
  double a = std::fabs(v);
  if (a >= 65504.0) return (uint16_t)(sgn | 0x7bffu);
  int e;
  const double f = std::frexp(a, &e);  // a = f * 2^e, f in [0.5, 1)
  int E = e - 1;
  uint32_t bits16;
  if (E < -14) {
    bits16 = (uint32_t)std::llround(std::ldexp(a, 24));
  } else {
    uint32_t m = (uint32_t)std::llround((2.0 * f - 1.0) * 1024.0);
    if (m == 1024u) { m = 0; E++; }
    bits16 = ((uint32_t)(E + 15) << 10) | m;
  }
  return (uint16_t)(sgn | bits16);
}
inline uint16_t half_round_down(double v) {
  uint16_t h = half_nearest(v);
  while ((double)half_to_float(h) > v) h = half_next_down(h);
  return h;
}
inline uint16_t half_round_up(double v) {
  uint16_t h = half_nearest(v);
  while ((double)half_to_float(h) < v) h = half_next_up(h);
  return h;
}

struct Builder {
  const float* geom;  // n x {cx, cy, cz, r*r} as the brute-force kernels read it
  const float* radius;
  std::vector<Box> box;        // per sphere
  std::vector<uint32_t> order; // tree spheres, permuted in place
  Bvh* out;

  void emit_leaf(const Box& b, uint32_t first, uint32_t count, uint32_t depth) {
    const uint32_t slot0 = out->n_tree_slots;
    float rec[8] = {b.lo[0], b.lo[1], b.lo[2], 0.f, b.hi[0], b.hi[1], b.hi[2], from_bits(slot0)};
    out->nodes.insert(out->nodes.end(), rec, rec + 8);
    for (uint32_t k = 0; k < kLeafSize; k++) {
      if (k < count) {
        const uint32_t s = order[first + k];
        out->slots.insert(out->slots.end(), geom + 4 * s, geom + 4 * s + 4);
        out->slot_index.push_back(s);
      } else {
        const float pad[4] = {0.f, 0.f, 0.f, -std::numeric_limits<float>::infinity()};
        out->slots.insert(out->slots.end(), pad, pad + 4);
        out->slot_index.push_back(0xffffffffu);
      }
    }
    out->n_tree_slots += kLeafSize;
    out->depth = std::max(out->depth, depth);
  }

  // builds the subtree over order[first, first+count); returns nothing, nodes are appended in
  // depth-first order and each node's skip link is patched once its subtree is complete
  void build(uint32_t first, uint32_t count, uint32_t depth) {
    Box b; b.clear();
    Box cb; cb.clear();
    for (uint32_t i = first; i < first + count; i++) {
      const Box& s = box[order[i]];
      b.grow(s);
      for (int k = 0; k < 3; k++) {
        float c = geom[4 * order[i] + k];
        cb.lo[k] = std::min(cb.lo[k], c);
        cb.hi[k] = std::max(cb.hi[k], c);
      }
    }
    const uint32_t me = (uint32_t)(out->nodes.size() / 8);
    if (count <= kLeafSize) {
      emit_leaf(b, first, count, depth);
      out->nodes[8 * me + 3] = from_bits(me + 1);
      return;
    }
    // binned surface-area heuristic over the centroid bounds
    constexpr int kBins = 16;
    int best_axis = -1, best_bin = -1;
    double best_cost = std::numeric_limits<double>::infinity();
    if (depth < 40) {
      for (int ax = 0; ax < 3; ax++) {
        const double lo = cb.lo[ax], ext = (double)cb.hi[ax] - lo;
        if (!(ext > 0)) continue;
        Box bb[kBins]; uint32_t cnt[kBins];
        for (int k = 0; k < kBins; k++) { bb[k].clear(); cnt[k] = 0; }
        for (uint32_t i = first; i < first + count; i++) {
          int k = (int)(((double)geom[4 * order[i] + ax] - lo) / ext * kBins);
          k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
          bb[k].grow(box[order[i]]); cnt[k]++;
        }
        double right_area[kBins]; uint32_t right_cnt[kBins];
        Box acc; acc.clear(); uint32_t n = 0;
        for (int k = kBins - 1; k > 0; k--) {
          if (cnt[k]) acc.grow(bb[k]);
          n += cnt[k];
          right_area[k] = n ? acc.half_area() : 0.0; right_cnt[k] = n;
        }
        acc.clear(); n = 0;
        for (int k = 0; k < kBins - 1; k++) {
          if (cnt[k]) acc.grow(bb[k]);
          n += cnt[k];
          if (n == 0 || right_cnt[k + 1] == 0) continue;
          // leaves hold kLeafSize spheres whether full or not: cost in units of leaves
          double cost = acc.half_area() * std::ceil(n / (double)kLeafSize) +
                        right_area[k + 1] * std::ceil(right_cnt[k + 1] / (double)kLeafSize);
          if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = k; }
        }
      }
    }
    uint32_t mid;
    if (best_axis >= 0) {
      const double lo = cb.lo[best_axis], ext = (double)cb.hi[best_axis] - lo;
      auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t s) {
        int k = (int)(((double)geom[4 * s + best_axis] - lo) / ext * kBins);
        k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
        return k <= best_bin;
      });
      mid = (uint32_t)(it - order.begin());
    } else {
      // identical centroids, or a tree that refuses to get shallower: split down the middle
      // along the widest axis
      int ax = 0;
      for (int k = 1; k < 3; k++)
        if ((double)cb.hi[k] - cb.lo[k] > (double)cb.hi[ax] - cb.lo[ax]) ax = k;
      mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                       [&](uint32_t a, uint32_t c) { return geom[4 * a + ax] < geom[4 * c + ax]; });
    }
    if (mid == first || mid == first + count) mid = first + count / 2;
    float rec[8] = {b.lo[0], b.lo[1], b.lo[2], 0.f, b.hi[0], b.hi[1], b.hi[2], from_bits(kInner)};
    out->nodes.insert(out->nodes.end(), rec, rec + 8);
    build(first, mid - first, depth + 1);
    build(mid, first + count - mid, depth + 1);
    out->nodes[8 * me + 3] = from_bits((uint32_t)(out->nodes.size() / 8));
  }
};

// geom: n x {cx, cy, cz, r*r}; radius: n signed radii.  Every value must be finite (the caller
// only builds for scenes it classified as regular).  Returns false when a tree would be useless
// (too few spheres); the brute-force kernels are used then.
inline bool build(const float* geom, const float* radius, uint32_t n, Bvh* out) {
  *out = Bvh();
  if (n < 16) return false;
  // ---- outliers: far-out or huge spheres that would blow up the per-ray margin ----------------
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
  std::vector<uint8_t> is_outlier(n, 0);
  uint32_t n_out = 0;
  for (uint32_t k = n; k-- > 0 && n_out < kMaxOutliers;) {
    if (reach[by_reach[k]] > 8.0 * ref) { is_outlier[by_reach[k]] = 1; n_out++; } else break;
  }
  if (n - n_out < 8) return false;

  Builder B;
  B.geom = geom; B.radius = radius; B.out = out;
  B.box.resize(n);
  Box all; all.clear();
  for (uint32_t i = 0; i < n; i++) {
    const double r = std::fabs((double)radius[i]);
    for (int k = 0; k < 3; k++) {
      B.box[i].lo[k] = round_down((double)geom[4 * i + k] - r);
      B.box[i].hi[k] = round_up((double)geom[4 * i + k] + r);
    }
    if (!is_outlier[i]) {
      B.order.push_back(i);
      for (int k = 0; k < 3; k++) {
        all.lo[k] = std::min(all.lo[k], geom[4 * i + k]);
        all.hi[k] = std::max(all.hi[k], geom[4 * i + k]);
      }
    }
  }
  // ascending index order inside the tree keeps the layout deterministic
  double c0[3];
  for (int k = 0; k < 3; k++) { c0[k] = 0.5 * ((double)all.lo[k] + all.hi[k]); out->c0[k] = (float)c0[k]; }
  double s0 = 0.0;
  for (uint32_t s : B.order) {
    double dx = geom[4 * s] - (double)out->c0[0], dy = geom[4 * s + 1] - (double)out->c0[1],
           dz = geom[4 * s + 2] - (double)out->c0[2];
    s0 = std::max(s0, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)radius[s]));
  }
  out->s0 = round_up(s0 * (1.0 + 1e-6));
  out->nodes.reserve((size_t)n * 8);
  out->slots.reserve((size_t)n * 8);
  B.build(0, (uint32_t)B.order.size(), 1);
  out->n_nodes = (uint32_t)(out->nodes.size() / 8);
  if (out->n_nodes > 0xfffeu) return false;  // 16-bit skip links
  // ---- the packed form the kernels read --------------------------------------------------------
  {
    double ext = 0.0;  // largest |box coordinate - c0|
    for (uint32_t i = 0; i < out->n_nodes; i++)
      for (int k = 0; k < 3; k++) {
        ext = std::max(ext, std::fabs((double)out->nodes[8 * i + k] - (double)out->c0[k]));
        ext = std::max(ext, std::fabs((double)out->nodes[8 * i + 4 + k] - (double)out->c0[k]));
      }
    int e = 0;
    if (ext > 0) (void)std::frexp(ext, &e);  // ext < 2^e
    const int shift = 10 - e;                // scaled coordinates stay below 2^10 (binary16: 65504)
    out->kscale = std::ldexp(1.0f, shift);
    out->kinv = std::ldexp(1.0f, -shift);
    out->nodes16.resize((size_t)(out->n_nodes + 1) * 4);
    for (uint32_t i = 0; i < out->n_nodes; i++) {
      uint16_t h[6];
      for (int k = 0; k < 3; k++) {
        h[k] = half_round_down(((double)out->nodes[8 * i + k] - (double)out->c0[k]) * (double)out->kscale);
        h[3 + k] = half_round_up(((double)out->nodes[8 * i + 4 + k] - (double)out->c0[k]) * (double)out->kscale);
      }
      const uint32_t skip = bits(out->nodes[8 * i + 3]), leaf = bits(out->nodes[8 * i + 7]);
      const uint32_t leaf16 = leaf == kInner ? 0xffffu : leaf / kLeafSize;
      out->nodes16[4 * i + 0] = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
      out->nodes16[4 * i + 1] = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
      out->nodes16[4 * i + 2] = (uint32_t)h[4] | ((uint32_t)h[5] << 16);
      out->nodes16[4 * i + 3] = skip | (leaf16 << 16);
    }
    // the fp32 device form (small scenes): {lo - c0, bits(skip), hi - c0, bits(leaf number)}
    out->nodes32.resize((size_t)(out->n_nodes + 1) * 8);
    for (uint32_t i = 0; i < out->n_nodes; i++) {
      for (int k = 0; k < 3; k++) {
        out->nodes32[8 * i + k] = round_down((double)out->nodes[8 * i + k] - (double)out->c0[k]);
        out->nodes32[8 * i + 4 + k] = round_up((double)out->nodes[8 * i + 4 + k] - (double)out->c0[k]);
      }
      const uint32_t leaf = bits(out->nodes[8 * i + 7]);
      out->nodes32[8 * i + 3] = from_bits(bits(out->nodes[8 * i + 3]) * 32u);  // skip link as a byte offset
      out->nodes32[8 * i + 7] = from_bits(leaf == kInner ? 0xffffu : leaf / kLeafSize);
    }
    for (int k = 0; k < 8; k++) out->nodes32[8 * out->n_nodes + k] = 0.f;
    out->nodes32[8 * out->n_nodes + 3] = from_bits(out->n_nodes * 32u);
    out->nodes32[8 * out->n_nodes + 7] = from_bits(0xffffu);
    // the spare node: an inner node that links to the end of the walk
    out->nodes16[4 * out->n_nodes + 0] = 0; out->nodes16[4 * out->n_nodes + 1] = 0;
    out->nodes16[4 * out->n_nodes + 2] = 0; out->nodes16[4 * out->n_nodes + 3] = out->n_nodes | 0xffff0000u;
  }
  // ---- the brute-force tail: outliers in ascending index order, groups of four ------------------
  for (uint32_t i = 0; i < n; i++) {
    if (!is_outlier[i]) continue;
    out->slots.insert(out->slots.end(), geom + 4 * i, geom + 4 * i + 4);
    out->slot_index.push_back(i);
  }
  out->n_outliers = n_out;
  while ((out->slot_index.size() & 3u) != 0) {
    const float pad[4] = {0.f, 0.f, 0.f, -std::numeric_limits<float>::infinity()};
    out->slots.insert(out->slots.end(), pad, pad + 4);
    out->slot_index.push_back(0xffffffffu);
  }
  out->n_slots = (uint32_t)out->slot_index.size();
  if (out->n_slots > 0xffffu) return false; // candidate and leaf queues hold 16-bit slot numbers
  return true;
}

}  // namespace ptbvh
