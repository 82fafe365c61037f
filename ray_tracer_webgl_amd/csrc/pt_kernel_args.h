// pt_kernel_args.h — launch-argument block shared by the host API (pt_api.hip) and the
// kernels (pt_kernels.hip).  Passed by value; everything in it is wave-uniform (SGPRs).
#pragma once
#include <stdint.h>

// Per-sphere shading record, fetched from global memory / L2 only for the closest hit of a
// segment (the intersection loop itself reads just the 16-byte LDS geometry record).
struct PtMatRec {
  float albedo[3];
  float fuzz;
  float refraction_index;
  int32_t type;
  float radius;  // signed: the outward normal divides by it (static/shader.frag:170)
  float inv_ri;  // 1.0f / refraction_index, the IEEE quotient made once on the host: GLASS divides by the index whenever the
                 // ray enters (static/shader.frag:252), and a correctly rounded fp32 division is the same bits wherever it is done
};
static_assert(sizeof(PtMatRec) == 32, "PtMatRec must be 32 bytes");
// Exact division of a 32-bit n by a launch-invariant d without a divide (Granlund & Montgomery):
// q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(m, n).  The constants are made on the host so
// that they arrive in SGPRs; a wave-uniform division done in the kernel would be hoisted out
// of the loop by the compiler and pin VGPRs for the kernel's lifetime.
struct PtDiv {
  uint32_t m, s1, s2;
};
static inline PtDiv pt_div_make(uint32_t d) {
  PtDiv r;
  uint32_t l = 0;
  while (l < 32 && (1ull << l) < (unsigned long long)d) l++;
  r.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1ull);
  r.s1 = l < 1 ? l : 1;
  r.s2 = l > 0 ? l - 1 : 0;
  return r;
}

struct PtKernelArgs {
  // uniform block, static/shader.frag:79-99 (see include/ptrace.h PtParams)
  float origin[3], horizontal[3], vertical[3], llc[3], cam_u[3], cam_v[3];
  float lens_radius;
  float time0, time_step;  // pass p renders with u_time = time0 + float(first_pass + p) * time_step
  uint32_t first_pass;
  int32_t spp;
  int32_t max_depth;
  int32_t background_mode;
  uint32_t width, height;  // full image
  uint32_t local_rows;     // rows owned by this context
  uint32_t band_rows, band_index, band_count;
  uint32_t n_passes;
  uint32_t n_spheres;
  uint32_t scene_regular;  // every sphere finite with |centre|, |radius| < 1e15 (host-checked)
  uint32_t tiles_x, tiles_y;  // 8x8 pixel tiles over width x local_rows
  uint32_t n_items;           // tiles_x * tiles_y * n_passes * 64 work items
  const float* geom;          // PT_LDS_ENTRIES(n_spheres) * {cx, cy, cz, r*r}, padded with unreachable spheres
  const PtMatRec* mat;        // n_spheres
  float* slab;                // n_passes * local_rows * width * float4 (rgb sum, spp)
  unsigned long long* counters;  // [0] work-queue head, [1] segments; [8..15] executed-work tallies of the COUNT twins
  const uint32_t* tile_order;    // n_tiles: queue position -> tile (heaviest tiles first)
  uint32_t* tile_cost;           // n_tiles: longest item (segments) per tile, feeds the next launch's order
  // culling structure of the walk kernels (pt_bvh.hpp / pt_grid.hpp); unused (NULL / 0) by the others
  const uint32_t* bvh_nodes;       // (n_nodes + 1) x 16 B: six binary16 box coordinates, skip | leaf << 16
  const float* bvh_nodes32;        // (n_nodes + 1) x 32 B: {lo - c0, skip, hi - c0, leaf}, fp32 (small scenes)
  const float* bvh_slots;          // n_slots x {cx, cy, cz, r*r}: leaves / cell groups (4 slots each), then the
                                   // spheres tested for every ray (padded to 4)
  const uint32_t* bvh_slot_index;  // n_slots: original sphere index of a slot
  const PtMatRec* slot_mat;        // n_slots: the slot's material record (a copy: shading needs no index look-up)
  uint32_t n_nodes, n_tree_slots, n_slots, n_outliers;
  float bvh_c0[3], bvh_s0;         // hierarchy: per-ray margin = 1.25e-3 (|o - c0|_1 + s0) + 1e-6; grid: D = |o - c0| + s0
  // uniform grid (PT_GEOM_GRID)
  const uint32_t* grid_cells;      // n_cells records (padded to 16 B): first entry | entries << 24
  uint32_t n_cells;
  uint32_t grid_n[3];
  float grid_lo[3], grid_hi[3], grid_h[3], grid_inv_h[3];
  float grid_r2_near;              // rays with |o - c0|^2 <= this ((0.9999 d_near - s0)^2) walk the cells
  float grid_lo_n[3], grid_hi_n[3];  // [lo, hi] widened by 1e-6 d_near: the entry slab test of those rays
  uint32_t lds_scene_bytes;        // dynamic LDS taken by the staged scene; the parked path state follows
  unsigned long long* wave_log;    // COUNT twins only (NULL otherwise): per wave PT_WAVE_LOG_WORDS x u64 {start, queue dry, end in 100 MHz ticks; HW_ID | XCC_ID << 32}
  uint32_t refill_min;             // lanes that must be waiting for an item before a busy wave runs the refill code
  uint32_t carry_lanes;            // the walk moves on when fewer lanes than this (and less than half) still walk
  float bvh_kinv;                  // boxes are stored in the frame (x - c0) / kinv
  uint32_t block_threads;          // blockDim.x of the launch
  uint32_t queue_chunk;            // work items a wave reserves per global-queue atomic
  float fw, fh;                    // float(width), float(height)
  PtDiv div_per_tile, div_tiles_x, div_band_rows;  // by 64 * n_passes, tiles_x, band_rows
  uint32_t coop_max_live;          // tail mode when at most this many lanes of a wave hold a ray
  int32_t rr_min_depth;            // Russian roulette after this many bounces (0 = never: the reference's estimator)
  uint32_t lens_off;               // 1: lens_radius is 0, u and v are finite, no component of the origin is -0 (pt_refill.hpp start_sample)
  uint32_t queue_groups;           // queue_static == 2: the number of wave groups G (a power of two), each with a head of its own
  uint32_t queue_static;           // 2: GROUPED queue — wave w belongs to group w % G and takes the group's reservations g, g + G, g + 2 G, ...
                                   // in the order its group's head hands them out (one atomic per reservation on one of G addresses);
                                   // 1: no queue atomics — wave w takes the reservations w, w + n_waves, w + 2 n_waves, ...
                                   // (launches of a few items per lane: every reservation of the shared queue is an atomic
                                   // on ONE address, ~25 ns each in turn; 28 000 of them ARE the 1-spp frame's 0.78 ms)
  uint32_t n_waves;                // waves of the launch (grid x workgroup / 64)
  uint32_t cost_feedback;          // 1: pass 0's items report their length (one atomicMax each) for the next launch's tile order
  uint32_t* cell_hist;             // COUNT twins of the grid kernels only (NULL otherwise): [n_slots] leaf-round lanes per entry-run start
                                   // (sampled: every 8th wave), then PT_COH_BINS bins of "distinct runs among a leaf round's lanes"
  const uint32_t* frame_ctr;       // device cell added to the pass number in u_time (pt_render_frames: frames replayed from a
                                   // hipGraph advance it on the device); points at a zero cell otherwise.  Never NULL.
  const float* mat_r0;             // n_spheres x {r0 of reflectance() for the ratio 1 / ri, for the ratio ri}: ((1 - ratio) / (1 + ratio))^2 made on the host (small-list kernels)
                                   // (last: the other kernels' argument offsets — and with them their scalar loads and spills — stay as they were)
};

// ---- waves per SIMD each trace kernel is BUILT FOR -------------------------------------------------------------
// Every trace kernel carries amdgpu_waves_per_eu(N, N): the compiler keeps it inside the VGPR and SGPR budget of N
// waves per SIMD, and its SGPR budget counts the 16 SGPRs per wave this platform's trap handler takes
// (800 / N - 16, rounded down to the allocation granule of 16; AMDGPUBaseInfo getMaxNumSGPRs).  The HIP occupancy
// query does NOT count them: for a kernel of 106 SGPRs it answers 800 / 112 = 7 waves, the SIMD holds
// 800 / (112 + 16) = 6.  Rounds 1-4 launched "7 workgroups of 256 threads per CU" for the list and small-list
// kernels on that answer; the wave log (round 5: HW_ID per wave, profiles/r05_residency.txt) shows wave slots
// 0-5 in use on every SIMD and the seventh workgroup of each CU starting only when another one has ended.  The host
// therefore takes min(occupancy query, N) — pt_api.hip resident_blocks() — with N from this table.
#ifndef PT_WAVES_WALK
#define PT_WAVES_WALK 6       // hierarchy and grid walks: latency-bound, 80 VGPRs (pt_kernels.hip)
#endif
#ifndef PT_WAVES_LIST
#define PT_WAVES_LIST 7       // scalar-load list walks: 94 SGPRs with 8-16 of them spilled, 58-61 VGPRs (config 4 through it -2.4 %, the
#endif                        // reference's scene -2.0 %, config 2 -0.4 % against six; profiles/r05_ab_runs.txt)
#define PT_WAVES_LIST_LDS 6   // the LDS list walk: 79 VGPRs (at seven it spills vector registers)
#ifndef PT_WAVES_SMALL
#define PT_WAVES_SMALL 7      // small-list kernels: 94 SGPRs (+16 = 112: seven really fit), 59 VGPRs
#endif
#define PT_WAVES_TWIN_CELLS 4 // the measuring twin of the cells-only grid kernel (98 VGPRs with its tallies live)
#define PT_BUILT_FOR(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#ifndef PT_LEAF_GROUP_GMEM
#define PT_LEAF_GROUP_GMEM 4  // entries a leaf round of the grid walk tests where they are gathered from global memory (pt_grid_walk.hpp)
#endif

enum { PT_WAVE_LOG_WORDS = 4 };
enum { PT_COH_BINS = 66, PT_HIST_WAVE_STRIDE = 8 };  // cell_hist: bins 0..64 = distinct entry runs in a sampled leaf round, bin 65 = sampled rounds' lanes  // u64 per wave in PtKernelArgs.wave_log
// regions of a wave step the measuring twins count (Tally::flag / collect): [PT_CTR_REGIONS + 2 k] wave steps in which any lane
// ran region k, [+ 2 k + 1] lanes that did
enum { PT_REG_REFILL_DECODE = 0, PT_REG_REFILL_RESERVE, PT_REG_CAMERA_RAY, PT_REG_SHADE_HIT_RECORD, PT_REG_SHADE_SKY, PT_REG_SHADE_DIFFUSE,
       PT_REG_SHADE_METAL, PT_REG_SHADE_GLASS, PT_REG_SHADE_GLASS_REFRACT, PT_REG_SHADE_CONTINUES, PT_REG_SHADE_FINISHED, PT_REG_SHADE_ITEM_STORE,
       PT_REG_WALK_ENTRY, PT_REG_WALK_ENTER_CELL, PT_REG_WALK_FAR_RAY, PT_REG_SHADE_ANY, PT_N_REGIONS };
enum { PT_CTR_HEAD = 0, PT_CTR_SEGMENTS = 1, PT_CTR_SAMPLES = 2, PT_CTR_FAR_RAYS = 3 /* grid walk: segments handed to the whole list because they reached the grid from outside its near region */, PT_CTR_SCRATCH = 7 /* host-side save / restore of a counter around a probe launch */, PT_CTR_WORK = 8, PT_CTR_LITERAL = 16, PT_CTR_PHASES = 24, PT_N_PHASES = 8, PT_CTR_REGIONS = 32, PT_CTR_TIMEBINS = 64, PT_CTR_COUNT = 128,
       PT_QUEUE_GROUPS_MAX = 256, PT_CTR_GROUP_HEADS = 128 /* then PT_QUEUE_GROUPS_MAX heads, 8 u64 (one 64-byte line) apart */,
       PT_CTR_ALLOC = PT_CTR_GROUP_HEADS + 8 * PT_QUEUE_GROUPS_MAX };  // [64..127]: COUNT twins, segments shaded per 0.655 ms bin of s_memrealtime

// Largest sphere list one workgroup can stage: 160 KiB LDS / 16 B (MI355X_MICROARCH.md §LDS);
// the staged list is padded to a multiple of 8 (two ping-pong groups of 4) plus one prefetch group.
#define PT_LDS_ENTRIES(n) ((((n) + 7u) & ~7u) + 4u)
#define PT_MAX_SPHERES_SMALL 16u   // PT_GEOM_SMALL: the whole list reaches the VALU from SGPRs, four spheres per s_load_dwordx16
#define PT_MAX_SPHERES_LDS 10232u  // PT_LDS_ENTRIES(10232) * 16 B = 163 776 B <= 160 KiB
#define PT_PARK_DWORDS 14u  // per-lane path state parked in LDS during the walks
#ifndef PT_PARK_STRIDE
#define PT_PARK_STRIDE 15u  // dwords per lane in the parking area (odd: conflict-free columns)
#endif
#define PT_BVH_LDS_BYTES32(n_nodes, n_slots) ((((size_t)(n_nodes) + 1u) * 2u + (size_t)(n_slots)) * 16u)
#define PT_BVH_LDS_BYTES16(n_nodes) (((size_t)(n_nodes) + 1u) * 16u)
#define PT_GRID_LDS_CELLS(n_cells) ((((size_t)(n_cells) + 3u) / 4u) * 16u)
#define PT_MAX_SPHERES 65528u      // candidate queues hold 16-bit indices; beyond the LDS list the
                                   // scan reads the padded global copy (pt_trace_kernel_gmem)

// pt_probe kinds (device-side evaluation of single PT-SPEC functions, for parity tests)
enum {
  PT_PROBE_HASH = 0,       // in: seed                 out: seed', h1 | seed'', h2x,h2y | seed''', h3x,h3y,h3z (9 floats)
  PT_PROBE_SINCOS = 1,     // in: u                    out: sin, cos
  PT_PROBE_CBRT = 2,       // in: x                    out: cbrt
  PT_PROBE_UNIT_SPHERE = 3,// in: seed                 out: x,y,z,seed'
  PT_PROBE_DIVSQRT = 4,    // in: a,b                  out: a/b, sqrt(|a|), fma(a,b,a)
  PT_PROBE_BASE_HASH = 5,  // in: bits x, bits y       out: hash (as float bits)
  PT_PROBE_FAST_ARITH = 6, // in: x,y,z                out: x/y, div_core, sqrt(x), sqrt_core, sqrt_rn, hit_root(x,y,z), plain roots, div_den_ok(y), inv_sqrt_rn(x), 1/sqrt(x) (10 floats)
};
