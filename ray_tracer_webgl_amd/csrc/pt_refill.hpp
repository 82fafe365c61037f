// pt_refill.hpp — where a lane's work comes from: the global work queue, the decode of a work item
// into a fragment (static/shader.vert:8, static/shader.frag:354-357, :410), and the camera ray of
// the fragment's next sample (static/shader.frag:365-370 + :342-351).
//
// EXACTNESS ARGUMENT carried by this file: (i) scheduling never changes a result — an item is one
// (pixel, pass) stream whose seed is a function of (v_position, u_time) only (:354-357), its radiance
// sum goes to its own slab slot, so WHICH lane runs it and WHEN is free; (ii) the pixel-centre and
// jitter divisions by the image size use div_core under the wave-uniform guard `wh_ok` (width and
// height in [2^-20, 2^20): numerators are odd integers >= 1 or values that are 0 or >= 2^-31 and < 1,
// inside div_core's range; a +0 numerator gives +0 either way) and the plain operators otherwise.
#pragma once
#include "pt_scene.hpp"

namespace ptk {

// uniform divisors of the image size: kept in SGPRs for the kernel's lifetime
struct PixelDiv {
  bool wh_ok;        // wave-uniform: div_core applies to divisions by width and height
  float y_fw, y_fh;  // rcp_newton of float(width), float(height)
};
__device__ __forceinline__ PixelDiv pixel_div() {
  karg_t& K = *kargs();
  PixelDiv pd;
  pd.wh_ok = div_den_ok(K.fw) && div_den_ok(K.fh);
  pd.y_fw = u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(rcp_newton(K.fw))));
  pd.y_fh = u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(rcp_newton(K.fh))));
  return pd;
}

// start the next camera path of this lane's item: static/shader.frag:365-370 + :342-351
// the pixel jitter is divided by the image size (:367-368)
__device__ __forceinline__ void start_sample(Path& p, const PixelDiv& pd) {
  karg_t& K = *kargs();
  float seed = p.seed;  // (local copies, written back at the end: see pt_grid_walk.hpp)
  const float st_s = p.st_s, st_t = p.st_t;
  const bool wh_ok = pd.wh_ok;
  const float y_fw = pd.y_fw, y_fh = pd.y_fh;
  V3 o, d; float a; V3 col; int depth;
  float r0, r1;
  hash2(seed, r0, r1);
  float jx, jy;
  if (wh_ok) { jx = div_core(r0, K.fw, y_fw); jy = div_core(r1, K.fh, y_fh); }
  else { jx = r0 / K.fw; jy = r1 / K.fh; }
  float s = st_s + jx;
  float t = st_t + jy;
  V3 dd = mk(fma_(t, K.vertical[0], fma_(s, K.horizontal[0], K.llc[0])),
             fma_(t, K.vertical[1], fma_(s, K.horizontal[1], K.llc[1])),
             fma_(t, K.vertical[2], fma_(s, K.horizontal[2], K.llc[2])));
  const V3 cam_o = mk(K.origin[0], K.origin[1], K.origin[2]);
  const V3 tt = mk(dd.x - cam_o.x, dd.y - cam_o.y, dd.z - cam_o.z);
  // random_in_unit_circle :123-129 draws twice whatever the lens: the seed takes its four steps here,
  // the draws themselves are made below from a copy, where their values are needed
  float seed_l = seed;
  seed = (seed + 0.1f) + 0.1f;
  seed = (seed + 0.1f) + 0.1f;
  // LENS OFF (aperture 0: State::default, src/state.rs:102,123).  Then rd = 0 * (...) and the offset
  // off = u * rd.x + v * rd.y is a vector of zeros of either sign: origin + off is the origin bit for
  // bit (its components are not -0: host-checked, as is that u and v are finite), and tt - off is tt
  // unless a component of tt is exactly -0, where the result takes its sign from off — if that
  // happens anywhere in the wave (it practically never does) the wave runs the full form below.
  bool lens = K.lens_off == 0u; // wave-uniform
  if (!lens) {
    const uint32_t nz = 0x80000000u;
    lens = pt_ballot(f2u(tt.x) == nz || f2u(tt.y) == nz || f2u(tt.z) == nz) != 0ull;
  }
  if (lens) {
    float ua = hash1(seed_l);
    float sa, ca;
    sincos2pi(ua, sa, ca);
    float rr = sqrt_rn(hash1(seed_l));
    float rdx = K.lens_radius * (rr * ca);
    float rdy = K.lens_radius * (rr * sa);
    V3 off = mk(fma_(K.cam_v[0], rdy, K.cam_u[0] * rdx), fma_(K.cam_v[1], rdy, K.cam_u[1] * rdx),
                fma_(K.cam_v[2], rdy, K.cam_u[2] * rdx));
    d = mk(tt.x - off.x, tt.y - off.y, tt.z - off.z);
    o = mk(cam_o.x + off.x, cam_o.y + off.y, cam_o.z + off.z);
  } else {
    d = tt;
    o = cam_o;
  }
  a = dot3(d, d);
  col = mk(1.0f, 1.0f, 1.0f);
  depth = 0;
  p.seed = seed; p.o = o; p.d = d; p.a = a; p.col = col; p.depth = depth;
}

// ---- refill: lanes without a ray pull work items -------------------------------------------------
// The wave reserves A.queue_chunk consecutive items from the global queue with ONE atomic
// (a memory-side atomic moves 64 B, so per-item atomics would dominate the kernel's HBM
// traffic) and deals them to its lanes from a wave-uniform local pool.
// Lanes that finish an item wait until a few of them can be refilled together: the item decode
// below costs ~115 VALU instructions for the whole wave whether one lane needs it or sixty,
// and with 16-spp items about one lane per wave step does.  A wave that is mostly idle (the
// drain of the launch, or its start) refills at once.
// Every wave reaches the "queue dry" exit: the head counter is monotone.
template <bool COUNT>
__device__ __forceinline__ void refill(const PtKernelArgs& A, Path& p, Queue& q, const PixelDiv& pd, Tally<COUNT>& tally) {
  karg_t& K = *kargs();
  // (local copies, written back at the end: see pt_grid_walk.hpp)
  bool alive = p.alive, new_path = p.new_path;
  bool dry = q.dry; // wave-uniform: a reservation came back empty, so no lane of this wave will get another item
  uint32_t slab_index = p.slab_index, item_tile = p.item_tile, item_segs = p.item_segs;
  int sample = p.sample; float seed = p.seed, st_s = p.st_s, st_t = p.st_t; V3 sum = p.sum;
  uint32_t pool_next = q.pool_next, pool_end = q.pool_end, refill_waited = q.refill_waited;
  uint32_t pool_tp0 = q.pool_tp0, pool_split = q.pool_split, pool_tile0 = q.pool_tile0, pool_tile1 = q.pool_tile1;
  uint32_t q_round = q.round;
  const bool wh_ok = pd.wh_ok;
  const float y_fw = pd.y_fw, y_fh = pd.y_fh;
  // (`waited` is reset after the loop, not inside it: a loop-carried value that becomes invariant after
  // the first trip makes the compiler peel that trip off, i.e. emit the whole item decode twice)
  bool dealt = false, put_off = false;
  for (;;) {
    bool need = !alive && !dry;
    unsigned long long mask = pt_ballot(need);
    if (mask == 0ull) break;
    // ... but not for long: with long items the next lane may be hundreds of steps away
    if ((uint32_t)__popcll(mask) < K.refill_min && (uint32_t)__popcll(pt_ballot(alive)) >= 32u &&
        (dealt ? 0u : refill_waited) < 8u) {
      put_off = true;
      break;
    }
    dealt = true;
    if (pool_next == pool_end) { // wave-uniform
      tally.flag(PT_REG_REFILL_RESERVE);
      unsigned long long base = 0;
      if (K.queue_static == 2u) {
        // GROUPED queue (short launches): the waves form G groups, group g owns the reservations g, g + G, g + 2 G, ... of the
        // tile-major list and hands them to its waves in the order they ask — the balance of a queue among a group's ~28 waves
        // with one atomic per reservation on one of G addresses (the single head of the shared queue takes a reservation
        // every ~13 ns: too slow for items of a few segments; no atomics at all — static dealing — leaves every wave
        // alone with the cost of the ~31 tiles it happens to get)
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        const uint32_t group = wave & (K.queue_groups - 1u);
        unsigned long long r = 0;
        if (lane_id() == 0u) r = atomicAdd(&A.counters[PT_CTR_GROUP_HEADS + 8u * group], 1ull);
        const uint32_t r_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r);
        base = ((unsigned long long)r_lo * K.queue_groups + group) * (unsigned long long)A.queue_chunk;
      } else if (K.queue_static != 0u) {
        // reservations dealt round-robin, no atomics (a wave's number is the same in all its lanes)
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        base = ((unsigned long long)q_round * K.n_waves + wave) * (unsigned long long)A.queue_chunk;
        q_round++;
      } else {
        if (lane_id() == 0u) base = atomicAdd(&A.counters[PT_CTR_HEAD], (unsigned long long)A.queue_chunk);
      }
      // wave-uniform, and known to the compiler as such (readfirstlane): the pool bookkeeping derived
      // from it then lives in SGPRs instead of occupying VGPRs for the kernel's lifetime
      base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
             (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
      if (base >= (unsigned long long)A.n_items) { // queue dry: these lanes are done
        tally.queue_dry();
        dry = true;
        need = false;  // nothing to deal in this trip; the loop ends at its next test
      } else {
        pool_next = (uint32_t)base;
        unsigned long long end = base + A.queue_chunk;
        pool_end = end < (unsigned long long)A.n_items ? (uint32_t)end : A.n_items;
        // a reservation no longer than one tile's items touches at most two tiles: look their numbers
        // up once, here, instead of one dependent global load per lane in every refill
        pool_tp0 = div_by(pool_next, K.div_per_tile.m, K.div_per_tile.s1, K.div_per_tile.s2);
        pool_split = (pool_tp0 + 1u) * (64u * K.n_passes);
        const uint32_t n_tiles_w = K.tiles_x * K.tiles_y;
        pool_tile0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)K.tile_order[pool_tp0]);
        pool_tile1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)K.tile_order[pool_tp0 + 1u < n_tiles_w ? pool_tp0 + 1u : pool_tp0]);
      }
    }
    const uint32_t avail = pool_end - pool_next;
    const uint32_t cnt = (uint32_t)__popcll(mask);
    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
    const uint32_t pool_base = pool_next;
    pool_next += cnt < avail ? cnt : avail;
    if (need && rank < avail) {
      tally.flag(PT_REG_REFILL_DECODE);
      uint32_t item = pool_base + rank;
      uint32_t per_tile = 64u * K.n_passes;
      uint32_t tile_pos, tile; // heaviest tiles are dealt first (tile_order)
      if (K.queue_chunk <= per_tile) { // wave-uniform
        const bool second = item >= pool_split;
        tile_pos = pool_tp0 + (second ? 1u : 0u);
        const uint32_t t0 = pool_tile0, t1 = pool_tile1;  // values, not a choice between two lvalues
        tile = second ? t1 : t0;
      } else {
        tile_pos = div_by(item, K.div_per_tile.m, K.div_per_tile.s1, K.div_per_tile.s2);
        tile = K.tile_order[tile_pos];
      }
      uint32_t rem_i = item - tile_pos * per_tile;
      uint32_t pass = rem_i >> 6, l = rem_i & 63u;
      uint32_t ty = div_by(tile, K.div_tiles_x.m, K.div_tiles_x.s1, K.div_tiles_x.s2), tx = tile - ty * K.tiles_x;
      uint32_t px = tx * 8u + (l & 7u), ly = ty * 8u + (l >> 3);
      if (px < K.width && ly < K.local_rows) {
        uint32_t y = ly;
        if (K.band_count > 1u) {
          uint32_t b = div_by(ly, K.div_band_rows.m, K.div_band_rows.s1, K.div_band_rows.s2), r = ly - b * K.band_rows;
          y = (b * K.band_count + K.band_index) * K.band_rows + r;
        }
        // static/shader.vert:8 + rasteriser: v_position at the pixel centre
        const float fx2 = (float)(2u * px + 1u), fy2 = (float)(2u * y + 1u); // odd integers >= 1
        float vx, vy;
        if (wh_ok) { vx = div_core(fx2, K.fw, y_fw) - 1.0f; vy = div_core(fy2, K.fh, y_fh) - 1.0f; }
        else { vx = fx2 / K.fw - 1.0f; vy = fy2 / K.fh - 1.0f; }
        // frames replayed from a hipGraph (pt_render_frames) count on the device: scalar load, constant within a launch
        const uint32_t frame_k = *(const uint32_t __attribute__((address_space(4)))*)K.frame_ctr;
        float u_time = K.time0 + (float)(K.first_pass + pass + frame_k) * K.time_step;
        // init_global_seed, static/shader.frag:354-357
        seed = (float)base_hash(f2u(vx), f2u(vy)) * (1.0f / 4294967296.0f) + u_time;
        st_s = (vx + 1.0f) * 0.5f; // :410
        st_t = (vy + 1.0f) * 0.5f;
        slab_index = (pass * K.local_rows + ly) * K.width + px;
        // only the launch's first pass reports its cost (atomicMax per pixel: the tile's
        // heaviest item): plenty for ordering tiles, and a memory-side atomic moves 64 B.  Launches of
        // one short pass (the reference's 1-spp frame) report nothing: there EVERY item would, the 64
        // lanes of a tile on one address, and — loads, stores and atomics complete in order — the next
        // material load of each wave would wait for them (measured on the 1280x702 1-spp frame:
        // 85 % of the wave time in s_waitcnt).
        item_tile = (pass == 0u && K.cost_feedback != 0u) ? tile : 0xffffffffu;
        item_segs = 0;
        sum = mk(0.f, 0.f, 0.f);
        sample = 0;
        new_path = true;
        alive = true;
      }
      // an item that falls outside the image (edge tile) is simply dropped
    }
  }
  refill_waited = (dealt ? 0u : refill_waited) + (put_off ? 1u : 0u);
  p.alive = alive; p.new_path = new_path;
  p.slab_index = slab_index; p.item_tile = item_tile; p.item_segs = item_segs;
  p.sample = sample; p.seed = seed; p.st_s = st_s; p.st_t = st_t; p.sum = sum;
  q.pool_next = pool_next; q.pool_end = pool_end; q.refill_waited = refill_waited;
  q.pool_tp0 = pool_tp0; q.pool_split = pool_split; q.pool_tile0 = pool_tile0; q.pool_tile1 = pool_tile1;
  q.round = q_round; q.dry = dry;
}

} // namespace ptk
