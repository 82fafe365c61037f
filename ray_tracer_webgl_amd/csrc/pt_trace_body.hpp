// pt_trace_body.hpp — the path-tracing kernel body shared by the three device translation units (each a gfx950
// code object of its own, loaded when one of its kernels is first asked for): pt_kernels.hip (list and walk
// kernels), pt_kernels_small.hip (the small-list kernels, one per list length modulo four) and
// pt_kernels_extra.hip (the opt-in builds: the Russian-roulette kernels and the measuring twins).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_kernel_args.h"
#include "pt_arith.hpp"
#include "pt_scene.hpp"
#include "pt_refill.hpp"
#include "pt_list.hpp"
#include "pt_bvh_walk.hpp"
#include "pt_grid_walk.hpp"
#include "pt_shade.hpp"

using namespace ptd;

// --------------------------------------------------------------------------------------------
// The path-tracing kernel body: one persistent wave working through (pixel, pass) items.
// Template parameters: pt_scene.hpp `Scene`; COUNT = the measuring twin (tallies live); RR = the
// opt-in Russian-roulette build (pt_shade.hpp); TAIL = the small-list builds' list remainder (pt_scene.hpp SMALL_TAIL).
// --------------------------------------------------------------------------------------------
template <bool SCAN_LDS, bool HAVE_LDS, int WALK = 0, bool COUNT = false, bool RR = false, int TAIL = -1>
__device__ __forceinline__ void pt_trace_body(const PtKernelArgs& A) {
  using namespace ptk;
  using S = Scene<SCAN_LDS, HAVE_LDS, WALK, TAIL>;
  S::stage(A);

  Path p;           // per lane
  Queue q;          // wave-uniform
  Carry cw;         // walk kernels: what survives a wave step
  BvhWalk bw;
  GridWalk gw;
  Tally<COUNT> tally;
  const PixelDiv pd = pixel_div();
  uint32_t seg_count = 0; // wave-uniform tally (samples are derived on the host: pixels * spp * passes)
  tally.start();

  for (;;) {
    refill<COUNT>(A, p, q, pd, tally);
    tally.phase(0);
    // one copy of the camera-ray code per step serves both kinds of lanes: those that just
    // pulled an item and those whose previous path ended in the last step
    if (p.alive && p.new_path) {
      tally.flag(PT_REG_CAMERA_RAY);
      start_sample(p, pd);
      p.new_path = false;
    }
    tally.phase(1);
    const unsigned long long live = pt_ballot(p.alive);
    if (live == 0ull) break; // every lane is exhausted: the queue is dry
    tally.step();

    // ---- hit_world: static/shader.frag:175-196 -------------------------------------------------
    Hit h;
    if constexpr (S::TREE) h.closest = cw.carried ? cw.closest_w : PT_MAX_T;
    if (!cw.carried) cw.hit_pos = 0xffffffffu;
    const bool fast = regular_ray(A, p);
    const int n_live = (int)__popcll(live);
    if constexpr (S::TREE) park_store(A, p);
    // TAIL MODE (pt_list.hpp) is the list kernels' way out of a launch's drain: the last few rays of a wave are looked at
    // by all 64 lanes.  The walk kernels do not have it: a walk costs a wave ~1 500 issue slots whatever its lane count
    // and a turn-around n / 64 x 45 + 70 per ray, so it only ever served waves of <= 3 live rays there (config 2), while
    // its test (three ballots per step) and its code cost every step: without it config 2 -1.3 %, the hierarchy -1 %, one
    // rank's band of eight -0.9 ... -1.4 % (profiles/r04_ab_runs.txt).  Nor do the small-list kernels: a turn-around per ray
    // costs what a whole step over nine spheres costs — without it config 4 -2 %, the reference's 1-spp frame 0.132 -> 0.114 ms.
    constexpr bool kTail = !S::TREE && !S::SMALL;
    const bool coop = kTail && (n_live <= (int)A.coop_max_live) && (pt_ballot(p.alive && !fast) == 0ull) &&
                      (pt_ballot(cw.carried) == 0ull);
    if (coop) {
      tail_mode<S>(A, p, live, h);
    } else {
      h.lit_from = fast ? 0xffffffffu : 0u;
      const bool scan_lane = p.alive && fast;
      if constexpr (S::BVH) bvh_walk<S, COUNT>(A, p, scan_lane, n_live, cw, bw, h, tally);
      else if constexpr (S::GRID) grid_walk<S, COUNT>(A, p, scan_lane, n_live, cw, gw, h, tally);
      else if constexpr (S::SMALL) small_scan<S>(A, p, scan_lane, h);
      else list_scan<S>(A, p, scan_lane, h);
      tally.literal(A, p.alive && !fast, scan_lane && h.lit_from < A.n_spheres, p.slab_index);
      // a REGULAR ray the grid walk hands over whole (it starts far outside the scene and reaches the
      // grid: pt_grid_walk.hpp) is looked at by the whole wave, 64 spheres at a time, like the rays of
      // tail mode, where the list is long (the kernels whose entries do not fit the LDS: thousands of
      // spheres, 0.3 ms per ray through the literal loop; config 5: 2.3e-5 of the rays, 1.5 % of the time)
      // (not in the build whose entries are staged in the LDS — scenes of hundreds of spheres: measured in round 6, the hand-over
      // there takes a far ray of a 1 500-sphere scene from ~18 000 to ~1 500 issue slots, but its code costs config 2, which has
      // no such ray, +0.5 ... +0.8 %; those scenes are served by a grid class that keeps far rays rare instead — pt_tune measures
      // the classes, pt_refit_grid follows the camera: profiles/r06_ab_runs.txt)
      if constexpr (S::GRID && S::WALK != 4) {
        const bool handed_over = scan_lane && h.lit_from == 0u;
        const unsigned long long m_h = pt_ballot(handed_over);
        if (m_h != 0ull) {
          tail_mode<S>(A, p, m_h, h);
          if (handed_over) h.lit_from = 0xffffffffu;
        }
      }
      literal_loop<S>(A, p, h); // (irregular rays; list kernels: candidates that did not fit the queue)
    }
    if constexpr (S::TREE) {
      if (coop) cw.carried = false;
      cw.closest_w = h.closest;
      park_load(A, p);
    }
    tally.phase(6);

    // ---- shade: static/shader.frag:304-335 (carried lanes are not there yet) --------------------
    const bool shade = p.alive && !cw.carried;
    seg_count += (uint32_t)__popcll(pt_ballot(shade));
    tally.timebin(A, shade);
    if (shade) shade_segment<S, RR>(A, p, h, cw, tally);
    tally.collect();
    tally.phase(7);
  }

  if (lane_id() == 0) atomicAdd(&A.counters[PT_CTR_SEGMENTS], (unsigned long long)seg_count);
  tally.flush(A);
}
