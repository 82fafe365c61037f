// pt_bvh_walk.hpp — PHASE 1 of hit_world through the bounding-box hierarchy of pt_bvh.hpp
// (PT_GEOM_BVH): which spheres a ray LOOKS AT.  Every sphere that is looked at runs the literal
// test (sphere_test + hit_root), so only the skipping needs an argument.
//
// EXACTNESS ARGUMENT.  hit_world's result for a regular ray is the lexicographic minimum of
// (v_i, -i) over the spheres that pass hit_sphere (pt_list.hpp), so the ORDER in which spheres
// are looked at is free and spheres that cannot pass need not be looked at.  A sphere can pass
// only if its fp32 discriminant is >= 0, and then the ray's half-line comes within
// |r| + sqrt(E) of the centre, E = u (18 |o-C|^2 + 7 r^2), u = 2^-24 (forward error of
// sphere_test; the `behind` rule only removes spheres).  Every box of the tree (pt_bvh.hpp,
// rounded outward) is therefore inflated by a per-ray margin
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
// tests them first, through scalar loads.  Host-side check of the claim: tests/test_bvh.py.
//
// Each lane walks the tree on its own (depth-first order with skip links: next = hit ?
// i + 1 : skip[i]); leaves are queued (8 x 16 bit) and processed in a second lockstep
// loop so that node steps and leaf steps do not serialise against each other.
#pragma once
#include "pt_scene.hpp"

namespace ptk {

template <typename S, bool COUNT>
__device__ __forceinline__ void bvh_walk(const PtKernelArgs& A, const Path& p, bool scan_lane, int n_live, Carry& cw,
                                         BvhWalk& w, Hit& h, Tally<COUNT>& tally) {
  // the state is worked on in LOCAL copies and written back at the end: through the reference
  // parameters it would be memory to every pass that runs before this function is inlined, and the
  // loops below would be shaped (rotated, merged, made divergent) for memory operands instead of registers
  const V3 o = p.o; const V3 d = p.d; const float a = p.a;
  float closest = h.closest; int hit = h.hit;
  bool carried = cw.carried; uint32_t hit_pos = cw.hit_pos;
  uint32_t cur = w.cur, l0 = w.l0, l1 = w.l1, l2 = w.l2, l3 = w.l3, l_cnt = w.l_cnt;
  uint32_t q0 = w.q0, q1 = w.q1, q2 = w.q2, q3 = w.q3, q_cnt = w.q_cnt;
  const uint32_t n_nodes = A.n_nodes;
  const bool fresh = scan_lane && !carried;
  const float ya = rcp_newton(a); // per-ray reciprocal for hit_root
  const uint32_t a_guard = hit_root_guard(a);

  auto eval_slot = [&](uint32_t pos) {
    const float4 g = S::slot_at(A, pos);
    float half_b, c, disc; sphere_test(o, d, a, g, half_b, c, disc);
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
    // (the ballot is the loop condition: a wave-uniform branch on SCC, see pt_grid_walk.hpp)
    for (unsigned long long m_q = pt_ballot(q_cnt > keep); m_q != 0ull; m_q = pt_ballot(q_cnt > keep)) {
      tally.exact(m_q);
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
  // two slots of a leaf, first slot `base`
  auto slot_pair = [&](const float4& c0, const float4& c1, uint32_t base, bool active) {
    float hb0, cc0, ds0; sphere_test(o, d, a, c0, hb0, cc0, ds0);
    float hb1, cc1, ds1; sphere_test(o, d, a, c1, hb1, cc1, ds1);
    if (active && __builtin_fmaxf(ds0, ds1) >= 0.0f) {
      if (!(ds0 < 0.0f)) note_slot(base + 0u, hb0, cc0);
      if (!(ds1 < 0.0f)) note_slot(base + 1u, hb1, cc1);
    }
  };
  // the outliers: wave-uniform walk (scalar loads), as the list kernels do for every sphere
  // (one at a time: there is usually exactly one, the ground); carried lanes have done this
  for (uint32_t i = A.n_tree_slots; i < A.n_tree_slots + A.n_outliers; i++) {
    if (((i - A.n_tree_slots) & 3u) == 0u && i != A.n_tree_slots) drain_to(4u); // room for four more
    const f4v e0 = S::c_slots(A)[i];
    float hb0, cc0, ds0; sphere_test(o, d, a, e0, hb0, cc0, ds0);
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
  const float kinv = S::BVH_MODE == 1 ? 1.0f : A.bvh_kinv;
  const float kx = ix * kinv, ky = iy * kinv, kz = iz * kinv;
  const float ahx = -((px + mrg) * ix), alx = -((px - mrg) * ix);
  const float ahy = -((py + mrg) * iy), aly = -((py - mrg) * iy);
  const float ahz = -((pz + mrg) * iz), alz = -((pz - mrg) * iz);
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));

  // fp32 nodes: the cursor is the node's LDS address (skip links are stored as byte offsets
  // and rebased to LDS addresses when the nodes are staged); packed nodes: the node index
  constexpr uint32_t walk_step = S::BVH_MODE == 1 ? 32u : 1u;
  typedef float4 __attribute__((address_space(3))) lds_f4w;
  const uint32_t walk_base = S::BVH_MODE == 1 ? (uint32_t)(uintptr_t)(lds_f4w*)pt_lds : 0u;
  const uint32_t walk_end = walk_base + n_nodes * walk_step;
  if (!carried) cur = scan_lane ? walk_base : walk_end;
  uint32_t walk_iters = 0;
  bool stop = false; // wave-uniform: the stragglers are carried into the next wave step
  const uint32_t half_live = ((uint32_t)n_live + 1u) >> 1;
  for (;;) {
    // Loop-carried state changes through selects only; the one real branch is the push.  A
    // lane whose walk is over rests on the spare node behind the tree (it links to itself,
    // and `through` is masked); the loop pauses for the leaf phase as soon as ANY lane's leaf
    // queue is full, so no lane ever has to stall on its own.  With fp32 nodes `cur` is
    // the node's byte offset (skip links are stored scaled): no address arithmetic.
    // The node loop runs while somebody walks, nobody's leaf queue is full, and the walkers are not
    // few enough to be carried — ONE scalar compare per trip: n_on >= need, with need = max(1, the carry
    // limit from the fourth trip on) + 64 while a queue is full (conditions joined with && before a
    // `break` are materialised as lane masks: a dozen scalar instructions per trip, pt_grid_walk.hpp).
    const uint32_t carry_lim = A.carry_lanes < half_live ? A.carry_lanes : half_live;
    uint32_t n_on = 0, n_full = 0;
    for (;;) {
      const bool on = cur < walk_end;
      const unsigned long long m_on = pt_ballot(cur < walk_end);
      n_on = (uint32_t)__popcll(m_on);
      n_full = (uint32_t)__popcll(pt_ballot(l_cnt == 8u));
      const uint32_t lim = walk_iters >= 4u ? carry_lim : 0u;
      const uint32_t need = (lim > 1u ? lim : 1u) + ((n_full < 1u ? n_full : 1u) << 6);
      if (n_on < need) break;
      walk_iters++;
      tally.walk(m_on);
      float t1x, t2x, t1y, t2y, t1z, t2z;
      uint32_t skip, leaf;
      if constexpr (S::BVH_MODE == 1) {
        typedef const f4v __attribute__((address_space(3))) lds_f4;
        lds_f4* np = (lds_f4*)(uintptr_t)cur; // `cur` is the node's LDS address itself
        const f4v na = np[0], nb = np[1];     // lo.xyz skip | hi.xyz leaf
        t1x = fma_(na.x, kx, ahx); t2x = fma_(nb.x, kx, alx);
        t1y = fma_(na.y, ky, ahy); t2y = fma_(nb.y, ky, aly);
        t1z = fma_(na.z, kz, ahz); t2z = fma_(nb.z, kz, alz);
        skip = f2u(na.w);
        leaf = f2u(nb.w);
      } else {
        const uint4 nd = S::node_at(A, cur);
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
      for (unsigned long long m_busy = pt_ballot(l_cnt != 0u) & pt_ballot(q_cnt <= 4u); m_busy != 0ull;
           m_busy = pt_ballot(l_cnt != 0u) & pt_ballot(q_cnt <= 4u)) {
        const bool busy = (l_cnt != 0u) & (q_cnt <= 4u);
        tally.leaf(m_busy);
        const uint32_t base = (l0 & 0xffffu) << 2;
        const uint32_t sh = busy ? 16u : 0u;
        l0 = __builtin_amdgcn_alignbit(l1, l0, sh);
        l1 = __builtin_amdgcn_alignbit(l2, l1, sh);
        l2 = __builtin_amdgcn_alignbit(l3, l2, sh);
        l3 >>= sh;
        l_cnt -= busy ? 1u : 0u;
        // two slots at a time: the leaf phase is where register pressure peaks
        {
          const float4 g0 = S::slot_at(A, base), g1 = S::slot_at(A, base + 1u);
          slot_pair(g0, g1, base, busy);
        }
        {
          const float4 g2 = S::slot_at(A, base + 2u), g3 = S::slot_at(A, base + 3u);
          slot_pair(g2, g3, base + 2u, busy);
        }
      }
      if (pt_ballot(l_cnt != 0u) == 0ull) break;
      drain_to(4u);
    }
    // (the node loop ended for the leaf phase: go on; it ended with walkers left and no full queue: those
    // are the stragglers, carried; or nobody walks any more)
    stop = n_full == 0u;
    if (stop || pt_ballot(cur < walk_end) == 0ull) break;
  }

  // PHASE 2: exact evaluation of whatever is still queued
  drain_to(0u);
  carried = cur < walk_end;
  tally.carried(carried);
  if (hit_pos != 0xffffffffu) hit = 0; // a hit; shading reads the slot's own copies (index not needed)
  h.closest = closest; h.hit = hit;
  cw.carried = carried; cw.hit_pos = hit_pos;
  w.cur = cur; w.l0 = l0; w.l1 = l1; w.l2 = l2; w.l3 = l3; w.l_cnt = l_cnt;
  w.q0 = q0; w.q1 = q1; w.q2 = q2; w.q3 = q3; w.q_cnt = q_cnt;
}

} // namespace ptk
