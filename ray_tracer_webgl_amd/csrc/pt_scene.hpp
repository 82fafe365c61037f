// pt_scene.hpp — what every phase of the trace kernel shares: the per-lane path state, the
// wave-uniform queue state, the view of the scene data (LDS / scalar cache / global memory,
// chosen at compile time), the parking area, and the tallies of the measuring twins.
//
// Nothing in here computes a result of static/shader.frag; it only says WHERE the data a phase
// reads lives.  Every accessor returns the same bits whichever memory it reads from (the LDS
// copies are byte copies of the global arrays made by Scene::stage), which is why the geometry
// paths give bit-identical images (tests/test_gpu_parity.py::test_geometry_paths_are_bit_identical).
#pragma once
#include "pt_arith.hpp"
#include "pt_kernel_args.h"

namespace ptk {
using namespace ptd;

// dynamic LDS of the trace kernels: [staged scene][parked path state, PT_PARK_STRIDE dwords per lane]
extern __shared__ float4 pt_lds[];

typedef float f4v __attribute__((ext_vector_type(4)));
typedef const f4v __attribute__((address_space(4))) const_f4v;  // constant address space: s_load through the scalar cache
typedef const PtKernelArgs __attribute__((address_space(4))) karg_t;

// The argument block again, read from the kernarg segment AT THE POINT OF USE (scalar loads through
// the scalar cache): `karg_t& K = *kargs();` at the top of a phase.  The once-per-wave-step sections
// (item decode, camera ray, walk set-up) use it so that their ~70 uniforms do not sit in SGPRs (or
// spill to VGPR lanes) across the walk and the shading code.
// Returned as a POINTER on purpose: a function returning a C++ reference is annotated
// `dereferenceable`, which makes every load through it speculatable — LICM then hoists all of them
// out of the wave-step loop, which is exactly what this view exists to avoid (measured: 87 SGPR
// spills and +24 % instructions in pt_trace_kernel_grid).
__device__ __forceinline__ karg_t* kargs() { return (karg_t*)__builtin_amdgcn_kernarg_segment_ptr(); }

// the lane's index in its wave (recomputed where needed rather than kept in a VGPR for the kernel's lifetime)
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// n / d for a launch-invariant d, constants made on the host (pt_kernel_args.h PtDiv)
__device__ __forceinline__ uint32_t div_by(uint32_t n, uint32_t m, uint32_t s1, uint32_t s2) {
  const uint32_t t = __umulhi(m, n);
  return (t + ((n - t) >> s1)) >> s2;
}

// ---- per-lane path state: one (pixel, pass) stream of static/shader.frag:360-383 ----------------
struct Path {
  bool alive = false;     // lane holds a live ray
  bool new_path = false;  // lane must generate its next camera ray before the next scan
  uint32_t slab_index = 0;
  uint32_t item_tile = 0xffffffffu, item_segs = 0; // cost feedback for the next launch's tile order
  int sample = 0, depth = 0;
  float seed = 0.f, st_s = 0.f, st_t = 0.f;
  V3 o = mk(0, 0, 0), d = mk(0, 0, 0);
  float a = 0.f; // dot(d,d), hoisted out of the sphere loop (static/shader.frag:147)
  V3 col = mk(1, 1, 1), sum = mk(0, 0, 0);
};

// ---- wave-uniform work-queue state (pt_refill.hpp) ------------------------------------------------
struct Queue {
  uint32_t pool_next = 0, pool_end = 0;  // this wave's reserved queue items
  uint32_t refill_waited = 0;            // steps the waiting lanes have been put off
  uint32_t pool_tp0 = 0, pool_split = 0, pool_tile0 = 0, pool_tile1 = 0;  // the reservation's tile(s)
  uint32_t round = 0;                    // static dealing (A.queue_static): reservations this wave has taken
  bool dry = false;                      // a reservation came back empty: this wave gets no more items
};

// ---- the closest hit of the current segment (hit_world's HitRecord, reduced to what shading needs) --
struct Hit {
  float closest = PT_MAX_T;          // closest_so_far, static/shader.frag:177
  int hit = -1;                      // list index of the closest sphere; walk kernels: 0 = "a hit, see hit_pos"
  uint32_t lit_from = 0xffffffffu;   // first sphere index the literal loop (PHASE 3) must take over
};

// ---- walk state that survives a wave step (walk kernels; DESIGN.md §4.6 "stragglers are carried") --
// The 64 walks of a wave step differ in length, and every loop runs for its longest lane: after the
// bulk has finished, a handful of stragglers would keep the whole wave walking.  Instead, once fewer
// than A.carry_lanes lanes (and less than half of the wave's live lanes) are still walking, the wave
// moves on: the finished lanes are shaded and get their next ray, the stragglers are CARRIED — they
// keep their walk state in registers, skip shading, and continue their walk in the next wave step
// beside the fresh walks.  Results cannot change (each lane performs the same operations on the same
// ray, only later); segments are counted when shaded.
struct Carry {
  bool carried = false;
  float closest_w = PT_MAX_T;
  uint32_t hit_pos = 0xffffffffu;  // the slot of the closest hit (its sphere index is looked up once, at the end)
};
struct BvhWalk {  // cursor, queued leaves (8 x 16 bit), queued candidates (8 x 16 bit)
  uint32_t cur = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0, l_cnt = 0;
  uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, q_cnt = 0;
};
struct GridWalk {  // boundary-crossing times, linear cell index, steps left per axis (3 x 10 bit, each + 1),
                   // the cell being tested (first untested entry | entries left << 24), its exit time
  float tmx = 0.f, tmy = 0.f, tmz = 0.f, t_exit = 0.f;
  uint32_t cell = 0, rem = 0, pend = 0;
};

// ---- scene data: which memory each kind of read goes to ------------------------------------------
// SCAN_LDS   : the scan (PHASE 1) walks the LDS copy with wave-uniform ds_read_b128 broadcasts.
// !SCAN_LDS  : the scan walks the padded global copy with wave-uniform SCALAR loads (constant
//              address space -> s_load_dwordx16 per four spheres via the scalar cache / L2);
//              sphere data reaches the VALU as SGPR operands, no LDS traffic in the scan, 12
//              fewer VGPRs.  Same arithmetic, bit-identical images.
// HAVE_LDS   : an LDS copy of the list exists (n <= 10 232) and serves every PER-LANE indexed
//              read (exact phase, tail mode, shading) whichever way the scan reads; without it
//              (lists beyond the 160 KiB LDS) those gathers go to global memory.
// WALK       : 0 = PHASE 1 scans the whole list.  Otherwise PHASE 1 walks a culling structure
//              that only decides which spheres are LOOKED AT:
//                1..3  the hierarchy of pt_bvh.hpp: 1 = nodes and slots staged in LDS, 2 = nodes
//                      in LDS, slots in global memory / L2, 3 = both in global memory;
//                4..6  the uniform grid of pt_grid.hpp: 4 = cells and entries staged in LDS,
//                      5 = cells in LDS, entries in global memory / L2, 6 = both global;
//                7     no structure, the LDS copy only for per-lane gathers: a list of at most 16 spheres (the reference's
//                      u_sphere_list[15], static/shader.frag:103) tested group by group from SGPRs.
//              List-order reads (tail mode, PHASE 3, shading) go to the global copy.
template <bool SCAN_LDS_, bool HAVE_LDS_, int WALK_, int TAIL_ = -1>
struct Scene {
  static constexpr bool SCAN_LDS = SCAN_LDS_, HAVE_LDS = HAVE_LDS_;
  static constexpr int WALK = WALK_;
  static constexpr bool BVH = WALK >= 1 && WALK <= 3;
  static constexpr bool GRID = WALK >= 4 && WALK <= 6;
  static constexpr bool SMALL = WALK == 7;   // at most 16 spheres, tested straight from SGPRs (pt_list.hpp small_scan)
  // small-list builds: the list holds 4 q + SMALL_TAIL spheres (0 .. 3: the build for that remainder, pt_kernels_small.hip);
  // -1: any length, the last group's padding tested and masked (the opt-in builds of pt_kernels_extra.hip)
  static constexpr int SMALL_TAIL = TAIL_;
  static_assert(TAIL_ >= -1 && TAIL_ <= 3 && (TAIL_ < 0 || WALK_ == 7), "a list remainder belongs to the small-list kernel");
  static constexpr bool TREE = BVH || GRID;  // a culling structure: hits are (slot, value) pairs
  static constexpr int BVH_MODE = BVH ? WALK : 0;
  static constexpr bool NODES_LDS = WALK == 1 || WALK == 2;
  static constexpr bool CELLS_LDS = WALK == 4 || WALK == 5;
  static_assert(HAVE_LDS || !SCAN_LDS, "an LDS scan needs the LDS copy");
  static_assert(!TREE || (!SCAN_LDS && !HAVE_LDS), "the walk kernels read list-order data from global memory");

  // copy the (already padded, {cx,cy,cz,r*r}) scene data into LDS once per workgroup
  static __device__ __forceinline__ void stage(const PtKernelArgs& A) {
    const float4* __restrict__ g_geom = reinterpret_cast<const float4*>(A.geom);
    const uint4* __restrict__ g_nodes = reinterpret_cast<const uint4*>(A.bvh_nodes);
    const float4* __restrict__ g_nodes32 = reinterpret_cast<const float4*>(A.bvh_nodes32);
    const float4* __restrict__ g_slots = reinterpret_cast<const float4*>(A.bvh_slots);
    if constexpr (HAVE_LDS && !TREE) {
      const uint32_t n_padded = PT_LDS_ENTRIES(A.n_spheres);
      for (uint32_t i = threadIdx.x; i < n_padded; i += blockDim.x) pt_lds[i] = g_geom[i];
      __syncthreads();
    }
    if constexpr (WALK == 1) { // [2 * (n_nodes + 1) halves of fp32 nodes][n_slots slots]
      const uint32_t n_a = 2u * (A.n_nodes + 1u);
      typedef float4 __attribute__((address_space(3))) lds_f4s;
      const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_f4s*)pt_lds;
      for (uint32_t i = threadIdx.x; i < n_a; i += blockDim.x) {
        float4 v = g_nodes32[i];
        if ((i & 1u) == 0u) v.w = u2f(f2u(v.w) + lds_base); // skip link: byte offset -> LDS address
        pt_lds[i] = v;
      }
      for (uint32_t i = threadIdx.x; i < A.n_slots; i += blockDim.x) pt_lds[n_a + i] = g_slots[i];
      __syncthreads();
    }
    if constexpr (WALK == 2) { // [n_nodes + 1 packed nodes]
      const uint32_t n_a = A.n_nodes + 1u;
      uint4* s_nodes = reinterpret_cast<uint4*>(pt_lds);
      for (uint32_t i = threadIdx.x; i < n_a; i += blockDim.x) s_nodes[i] = g_nodes[i];
      __syncthreads();
    }
    if constexpr (CELLS_LDS) { // [n_cells cell records, padded to 16 B][mode 4: n_slots entries]
      const uint32_t n_c4 = (A.n_cells + 3u) >> 2;
      const uint4* g_c4 = reinterpret_cast<const uint4*>(A.grid_cells); // the array is padded to 16 B
      uint4* s_c4 = reinterpret_cast<uint4*>(pt_lds);
      for (uint32_t i = threadIdx.x; i < n_c4; i += blockDim.x) s_c4[i] = g_c4[i];
      if constexpr (WALK == 4)
        for (uint32_t i = threadIdx.x; i < A.n_slots; i += blockDim.x) pt_lds[n_c4 + i] = g_slots[i];
      __syncthreads();
    }
  }

  // list order, wave-uniform index (the scan, PHASE 3)
  static __device__ __forceinline__ float4 geom_scan(const PtKernelArgs& A, uint32_t i) {
    if constexpr (SCAN_LDS) {
      return pt_lds[i];
    } else {
      const f4v v = ((const_f4v*)A.geom)[i];
      return make_float4(v.x, v.y, v.z, v.w);
    }
  }
  // list order, per-lane index (exact phase, tail mode, shading)
  static __device__ __forceinline__ float4 geom_at(const PtKernelArgs& A, uint32_t i) {
    if constexpr (HAVE_LDS && !TREE) {
      return pt_lds[i];
    } else {
      return reinterpret_cast<const float4*>(A.geom)[i];
    }
  }
  // culling-structure reads (per-lane index)
  static __device__ __forceinline__ uint4 node_at(const PtKernelArgs& A, uint32_t i) { // packed nodes (modes 2, 3)
    if constexpr (NODES_LDS) return reinterpret_cast<const uint4*>(pt_lds)[i];
    else return reinterpret_cast<const uint4*>(A.bvh_nodes)[i];
  }
  static __device__ __forceinline__ float4 slot_at(const PtKernelArgs& A, uint32_t i) {
    if constexpr (WALK == 1) return pt_lds[2u * (A.n_nodes + 1u) + i];
    else if constexpr (WALK == 4) return pt_lds[((A.n_cells + 3u) >> 2) + i];
    else return reinterpret_cast<const float4*>(A.bvh_slots)[i];
  }
  static __device__ __forceinline__ uint32_t cell_at(const PtKernelArgs& A, uint32_t i) {
    if constexpr (CELLS_LDS) return reinterpret_cast<const uint32_t*>(pt_lds)[i];
    else return A.grid_cells[i];
  }
  // slots through the scalar cache (wave-uniform index: the always-tested spheres)
  static __device__ __forceinline__ const_f4v* c_slots(const PtKernelArgs& A) { return (const_f4v*)A.bvh_slots; }
};

// ---- parking (walk kernels) -----------------------------------------------------------------------
// The walks are latency-bound (per-lane LDS gathers, short dependent loops), so they want waves,
// i.e. few VGPRs: the part of the path state that the walk does not touch is parked in LDS while
// it runs (14 dwords per lane) and fetched back for shading.  LDS behind the staged scene:
// PT_PARK_STRIDE dwords per lane.  The stride is odd, so the 32 lanes of a half-wave hit 32
// different banks at any fixed field, and every field is an immediate offset from the lane's base
// address.  volatile: the values must not be forwarded in registers.
typedef volatile uint32_t __attribute__((address_space(3))) lds_u32;
__device__ __forceinline__ lds_u32* park_of(const PtKernelArgs& A) {
  return (lds_u32*)reinterpret_cast<uint32_t*>(pt_lds) + (A.lds_scene_bytes >> 2) + PT_PARK_STRIDE * threadIdx.x;
}
__device__ __forceinline__ void park_store(const PtKernelArgs& A, const Path& p) {
  lds_u32* ps = park_of(A);
  ps[0] = f2u(p.sum.x); ps[1] = f2u(p.sum.y); ps[2] = f2u(p.sum.z);
  ps[3] = f2u(p.col.x); ps[4] = f2u(p.col.y); ps[5] = f2u(p.col.z);
  ps[6] = f2u(p.seed);
  ps[7] = f2u(p.st_s); ps[8] = f2u(p.st_t);
  ps[12] = (uint32_t)p.sample; ps[13] = (uint32_t)p.depth;
  ps[9] = p.slab_index; ps[10] = p.item_tile; ps[11] = p.item_segs;
}
__device__ __forceinline__ void park_load(const PtKernelArgs& A, Path& p) {
  lds_u32* ps = park_of(A);
  p.sum = mk(u2f(ps[0]), u2f(ps[1]), u2f(ps[2]));
  p.col = mk(u2f(ps[3]), u2f(ps[4]), u2f(ps[5]));
  p.seed = u2f(ps[6]);
  p.st_s = u2f(ps[7]); p.st_t = u2f(ps[8]);
  p.sample = (int)ps[12]; p.depth = (int)ps[13];
  p.slab_index = ps[9]; p.item_tile = ps[10]; p.item_segs = ps[11];
}

// ---- executed-work tallies of the COUNT twins (wave-uniform; compiled away in the timed kernels) ----
// The measuring twin of a kernel is the same body with these counters live: iterations and active
// lanes per phase into A.counters — never the kernel that is timed.
template <bool COUNT>
struct Tally {
  unsigned long long t_wave_start = 0, t_wave_dry = 0;
  // which REGIONS of a wave step ran, and for how many lanes (tools/instruction_mix.py weights the kernel's ISA with these):
  // a lane notes the regions it enters as bits of lane_flags (inside divergent code), collect() turns them into wave-level
  // counts in uniform control flow at the end of the step
  // (the 2 x PT_N_REGIONS counters live in ONE VGPR: lane 2 k holds region k's wave-step count, lane 2 k + 1 its lane count —
  // as wave-uniform scalars they were 32 more SGPRs, spilled through VGPR lanes, and pushed the twins into scratch)
  uint32_t lane_flags = 0, reg_acc = 0;
  uint32_t hw_id = 0, xcc_id = 0;  // where the wave runs: HW_ID (wave slot, SIMD, CU, SH, SE) and XCC_ID, read once at its start
  uint32_t tb_bin = 0xffffffffu, tb_acc = 0;
  uint32_t n_walk_it = 0, n_walk_ln = 0, n_leaf_it = 0, n_leaf_ln = 0, n_exact_it = 0, n_exact_ln = 0, n_steps = 0,
           n_carried = 0;
  uint32_t n_walk_first = 0;  // cell steps taken through the copy written out in front of the loop (pt_grid_walk.hpp)
  uint32_t n_exact_always_it = 0, n_exact_always_ln = 0;  // (the share of the exact evaluations that serves the always-tested group)
  bool in_always = false;
  // phase clock (shader cycles, s_memtime): where a wave's time goes
  //   0 refill  1 camera ray  2 always-tested spheres (hierarchy: set-up + outliers)  3 per-ray constants + grid entry
  //   4 advance / node loops  5 leaf + exact  6 literal + parking + the rest  7 shade
  unsigned long long ph_t[PT_N_PHASES] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_mark = 0;

  __device__ __forceinline__ void start() {
    if constexpr (COUNT) {
      t_wave_start = __builtin_amdgcn_s_memrealtime();
      ph_mark = __builtin_amdgcn_s_memtime();
      // s_getreg_b32 hwreg(HW_REG_HW_ID) [31:0] and hwreg(HW_REG_XCC_ID) [3:0]; simm16 = (size - 1) << 11 | offset << 6 | id
      hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
      xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    }
  }
  __device__ __forceinline__ void walk(unsigned long long mask) { if constexpr (COUNT) { n_walk_it++; n_walk_ln += (uint32_t)__popcll(mask); } }
  __device__ __forceinline__ void walk_first() { if constexpr (COUNT) n_walk_first++; }
  // config 5's question: WHICH entry runs do the leaf rounds gather, and how many different ones per round?
  // Every PT_HIST_WAVE_STRIDE-th wave counts, per leaf round, its lanes per run start (global histogram) and the number of distinct
  // runs among them (how coherent is one wave-level gather).
  __device__ __forceinline__ void leaf_cells(const PtKernelArgs& A, bool has, uint32_t base, uint32_t n_slots) {
    if constexpr (COUNT) {
      if (A.cell_hist == nullptr) return;
      const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
      if (wave % PT_HIST_WAVE_STRIDE != 0u) return;
      if (has && base < n_slots) atomicAdd(&A.cell_hist[base], 1u);
      unsigned long long left = pt_ballot(has);
      const uint32_t lanes = (uint32_t)__popcll(left);
      uint32_t distinct = 0;
      while (left != 0ull) {
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(left));
        left &= ~pt_ballot(has && base == b);
        distinct++;
      }
      if (lane_id() == 0u) {
        atomicAdd(&A.cell_hist[n_slots + (distinct < 64u ? distinct : 64u)], 1u);
        atomicAdd(&A.cell_hist[n_slots + 65u], lanes);
      }
    }
  }
  __device__ __forceinline__ void leaf(unsigned long long mask) { if constexpr (COUNT) { n_leaf_it++; n_leaf_ln += (uint32_t)__popcll(mask); } }
  __device__ __forceinline__ void exact(unsigned long long mask) {
    if constexpr (COUNT) {
      n_exact_it++; n_exact_ln += (uint32_t)__popcll(mask);
      if (in_always) { n_exact_always_it++; n_exact_always_ln += (uint32_t)__popcll(mask); }
    }
  }
  __device__ __forceinline__ void always_group(bool on) { if constexpr (COUNT) in_always = on; }
  __device__ __forceinline__ void step() { if constexpr (COUNT) n_steps++; }
  __device__ __forceinline__ void flag(int region) { if constexpr (COUNT) lane_flags |= 1u << region; }
  __device__ __forceinline__ void collect() {
    if constexpr (COUNT) {
      const unsigned long long t_in = __builtin_amdgcn_s_memtime();  // (its own ~100 instructions stay out of the phase clock)
      const uint32_t me = lane_id();
      for (int k = 0; k < PT_N_REGIONS; k++) {
        const unsigned long long m = pt_ballot(((lane_flags >> k) & 1u) != 0u);
        const uint32_t add = (me & 1u) ? (uint32_t)__popcll(m) : (m != 0ull ? 1u : 0u);
        reg_acc += (me >> 1) == (uint32_t)k ? add : 0u;
      }
      lane_flags = 0u;
      ph_mark += __builtin_amdgcn_s_memtime() - t_in;
    }
  }
  __device__ __forceinline__ void carried(bool c) { if constexpr (COUNT) n_carried += (uint32_t)__popcll(pt_ballot(c)); }
  // who takes PHASE 3 (the literal loop over the whole list): irregular rays, and regular ones the walk hands over
  __device__ __forceinline__ void literal(const PtKernelArgs& A, bool irregular, bool handed_over, uint32_t slab_index) {
    if constexpr (COUNT) {
      const unsigned long long mi = pt_ballot(irregular), mh = pt_ballot(handed_over);
      if ((mi | mh) != 0ull && lane_id() == 0) {
        atomicAdd(&A.counters[PT_CTR_LITERAL + 0], (unsigned long long)__popcll(mi));
        atomicAdd(&A.counters[PT_CTR_LITERAL + 1], (unsigned long long)__popcll(mh));
        atomicAdd(&A.counters[PT_CTR_LITERAL + 2], 1ull); // wave steps that ran the loop
      }
      if (irregular) A.counters[PT_CTR_LITERAL + 3] = 1ull + slab_index; // (any one of them: where to look)
      else if (handed_over) A.counters[PT_CTR_LITERAL + 4] = 1ull + slab_index;
    }
  }
  __device__ __forceinline__ void queue_dry() { if constexpr (COUNT) { if (t_wave_dry == 0) t_wave_dry = __builtin_amdgcn_s_memrealtime(); } }
  __device__ __forceinline__ void phase(int k) {
    if constexpr (COUNT) {
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();
      ph_t[k] += now_ - ph_mark;
      ph_mark = now_;
    }
  }
  // segments per time bin (dev diagnostics: where in a launch does the rate sag?)
  __device__ __forceinline__ void timebin(const PtKernelArgs& A, bool shade) {
    if constexpr (COUNT) {
      const uint32_t bin = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> 16) & 63u;
      if (bin != tb_bin) {
        if (lane_id() == 0 && tb_acc) atomicAdd(&A.counters[PT_CTR_TIMEBINS + tb_bin], (unsigned long long)tb_acc);
        tb_bin = bin; tb_acc = 0;
      }
      tb_acc += (uint32_t)__popcll(pt_ballot(shade));
    }
  }
  __device__ __forceinline__ void flush(const PtKernelArgs& A) {
    if constexpr (COUNT) {
      if (lane_id() < 2u * PT_N_REGIONS && reg_acc != 0u) atomicAdd(&A.counters[PT_CTR_REGIONS + lane_id()], (unsigned long long)reg_acc);
      if (lane_id() == 0) {
        atomicAdd(&A.counters[PT_CTR_WORK + 0], (unsigned long long)n_walk_it);
        atomicAdd(&A.counters[PT_CTR_WORK + 1], (unsigned long long)n_walk_ln);
        atomicAdd(&A.counters[PT_CTR_WORK + 2], (unsigned long long)n_leaf_it);
        atomicAdd(&A.counters[PT_CTR_WORK + 3], (unsigned long long)n_leaf_ln);
        atomicAdd(&A.counters[PT_CTR_WORK + 4], (unsigned long long)n_exact_it);
        atomicAdd(&A.counters[PT_CTR_WORK + 5], (unsigned long long)n_exact_ln);
        atomicAdd(&A.counters[PT_CTR_WORK + 6], (unsigned long long)n_steps);
        atomicAdd(&A.counters[PT_CTR_WORK + 7], (unsigned long long)n_carried);
        atomicAdd(&A.counters[PT_CTR_LITERAL + 5], (unsigned long long)n_exact_always_it);
        atomicAdd(&A.counters[PT_CTR_LITERAL + 6], (unsigned long long)n_exact_always_ln);
        atomicAdd(&A.counters[PT_CTR_LITERAL + 7], (unsigned long long)n_walk_first);
        if (tb_acc) atomicAdd(&A.counters[PT_CTR_TIMEBINS + tb_bin], (unsigned long long)tb_acc);
        for (int k = 0; k < PT_N_PHASES; k++) atomicAdd(&A.counters[PT_CTR_PHASES + k], ph_t[k]);

        if (A.wave_log) {
          unsigned long long* wl = A.wave_log + (unsigned long long)PT_WAVE_LOG_WORDS * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
          wl[0] = t_wave_start; wl[1] = t_wave_dry; wl[2] = __builtin_amdgcn_s_memrealtime();
          wl[3] = (unsigned long long)hw_id | ((unsigned long long)xcc_id << 32);
        }
      }
    }
  }
};

// The lowest set bit of a 4-bit candidate mask (the exact loops of pt_grid_walk.hpp and pt_list.hpp small_scan).
// TOTAL on purpose: __builtin_ctz(0) is undefined, and what gfx950 makes of it (v_ffbl_b32 -> 0xffffffff) turned a
// harmless-looking refactoring of the grid walk's exact loop into a wild read in round 3: with the loop's
// `if (mask != 0u)` region removed, an idle lane computed the slot `base + 0xffffffff`, and lanes without a cell
// under test (base 0) indexed bvh_slot_index[] 16 GiB out of bounds on their first tie.  With the
// fifth bit set the answer for an empty mask is 4 — a slot inside the four entries of slack every entry array
// carries — at the price of one v_or_b32 per evaluation (A/B on one device: +0.2 % on config 2,
// inside the noise on configs 4 and 5).
__device__ __forceinline__ uint32_t first_candidate(uint32_t mask) {
  return (uint32_t)__builtin_ctz(mask | 16u);
}

// The literal cheap half of hit_sphere, static/shader.frag:146-152: oc, half_b, c, discriminant, in the
// PT-SPEC operation order (the one expression every phase that looks at a sphere evaluates, so a
// sphere's (half_b, c, disc) are the same bits wherever they are computed).  g = {cx, cy, cz, r*r}.
template <typename G>
__device__ __forceinline__ void sphere_test(const V3& o, const V3& d, float a, const G& g, float& half_b, float& c, float& disc) {
  V3 oc = mk(o.x - g.x, o.y - g.y, o.z - g.z);
  half_b = dot3(oc, d);
  c = fma_(oc.z, oc.z, fma_(oc.y, oc.y, fma_(oc.x, oc.x, -g.w)));
  disc = fma_(-a, c, half_b * half_b);
}

} // namespace ptk
