// tools/wavesim.cpp — DEV TOOL (never shipped, never linked into libptrace): a wave-level cost
// model of the hierarchy kernel's control flow, run on the host.
//
// The hierarchy kernel (pt_trace_kernel_bvh) is bound by VALU issue and by lane occupancy: 64
// lanes walk 64 different rays in lockstep, so every loop runs for the LONGEST lane.  Whether a
// change of control structure pays (dealing the phases of a wave step out by population instead
// of running them in a fixed order; ordered traversal; pruning by the closest hit; another
// margin) is a question about iteration counts and lane occupancy, which do not need a GPU to be
// counted.  This program builds the product's own tree (pt_bvh.hpp), path-traces the scene with
// its own RNG (statistics only — nothing here is bit-exact or used by any test of results), and
// replays the kernel's control flow for groups of 64 lanes exactly as the kernel would run it:
// per-lane leaf / candidate queues, queue limits, regeneration, work items in tile order.
//
//   g++ -O2 -std=c++17 -o /tmp/wavesim tools/wavesim.cpp
//   python tools/wavesim_scene.py config2 480 270 /tmp/c2.scene
//   /tmp/wavesim /tmp/c2.scene [key=value ...]
//
// Cost unit: VALU wave-instructions (the per-phase constants are the measured ones of
// DESIGN.md §4.5); `per 64 seg` is directly comparable with the kernel's ≈2150.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../ray_tracer_webgl_amd/csrc/pt_bvh.hpp"

struct Sph { float c[3], r; int type; float alb[3], fuzz, ri; int uuid, pad; };
static_assert(sizeof(Sph) == 48, "PtSphere");

struct Scene {
  uint32_t n, W, H, depth;
  float cam[19];
  int bg;
  std::vector<Sph> s;
};

struct Opt {
  int mode = 1;          // 0 = lockstep (the round-1 kernel), 1 = phase scheduler
  int ordered = 0;       // near-to-far child order (per-node split axis + direction sign)
  int prune = 0;         // slab test against the closest exact hit so far
  int margin = 0;        // 0 = 1.25e-3 (|p|_1 + s0), 1 = sqrt(rmin^2 + 54u D^2) - rmin
  int leaf_fifo = 0;     // leaves leave the queue oldest first
  int spp = 16, passes = 2, waves = 64, chunk = 128;
  int t_cam = 24, t_node = 32, t_node_exit = 24, t_leaf = 24, t_leaf_exit = 16, t_exact = 24, t_exact_exit = 16,
      t_shade = 24;
  int node_burst = 8;    // node iterations per scheduler trip at most
  int eager = 0;         // mode 0: leaf + exact rounds (to empty queues) after every `eager` node iterations
  int block = 0;         // mode 2: a lane with candidates waits (no cell steps, no leaf rounds) until `block` lanes wait or nobody else can move; 0 = evaluate after every leaf round
  int carry = 0;         // mode 0: stop walking when fewer than `carry` lanes walk; they resume in the next wave step
  int outlier_exact = 0;
  int cell_x100 = 100;   // mode 2: grid cell edge = cell_x100 / 100 x the heuristic edge
  int big_x100 = 150;    // mode 2: spheres with |r| > big_x100/100 x cell edge are tested for every ray // evaluate the always-tested spheres exactly at set-up (closest known before the walk)
};

// cost constants (VALU wave-instructions per execution)
static const double C_NODE = 21, C_LEAF = 75, C_EXACT = 65, C_SHADE = 380, C_CAM = 150, C_REFILL = 60, C_SETUP = 70,
                    C_PARK = 30, C_MISC = 100, C_SCHED = 30;

struct XRng {
  uint64_t s;
  float next() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (float)((s >> 40) * (1.0 / 16777216.0));
  }
};

struct Lane {
  bool alive = false, exhausted = false, new_path = false;
  int samples_left = 0, depth = 0;
  XRng rng{1};
  float px = 0, py = 0; // pixel
  float o[3], d[3], a = 0;
  // walk
  uint32_t cur = 0;
  uint32_t stack[64]; int sp = 0;
  float inv[3], mrg = 0, p[3];
  uint16_t lq[8]; int l_cnt = 0;
  uint16_t cq[8]; int q_cnt = 0;
  float closest = 1e5f; int hit_pos = -1;
  uint32_t nodes_this_seg = 0;
};

struct Phase { double iters = 0, lanes = 0; };

struct Sim {
  Scene sc; Opt op; ptbvh::Bvh T;
  std::vector<int> left, right, axis; // ordered traversal
  float rmin = 0, rmax = 0;
  uint32_t tiles_x, tiles_y;
  uint64_t n_items, head = 0;
  // stats
  Phase ph_node, ph_leaf, ph_exact, ph_shade, ph_cam;
  double cost = 0, segs = 0, sched_trips = 0, forced = 0, wave_steps = 0;
  double node_visits = 0, leaf_visits = 0, exact_evals = 0;
  double hist_hit[16] = {0}, hist_miss[16] = {0}, hist_ground[16] = {0};

  void build() {
    std::vector<float> geom(sc.n * 4), rad(sc.n);
    for (uint32_t i = 0; i < sc.n; i++) {
      for (int k = 0; k < 3; k++) geom[4 * i + k] = sc.s[i].c[k];
      geom[4 * i + 3] = sc.s[i].r * sc.s[i].r; rad[i] = sc.s[i].r;
    }
    if (!ptbvh::build(geom.data(), rad.data(), sc.n, &T)) { fprintf(stderr, "no tree\n"); exit(1); }
    left.assign(T.n_nodes, -1); right.assign(T.n_nodes, -1); axis.assign(T.n_nodes, 0);
    for (uint32_t i = 0; i < T.n_nodes; i++) {
      if (ptbvh::bits(T.nodes[8 * i + 7]) != ptbvh::kInner) continue;
      int l = i + 1, r = (int)ptbvh::bits(T.nodes[8 * l + 3]);
      left[i] = l; right[i] = r;
      float best = -1; int ax = 0;
      for (int k = 0; k < 3; k++) {
        float cl = T.nodes[8 * l + k] + T.nodes[8 * l + 4 + k], cr = T.nodes[8 * r + k] + T.nodes[8 * r + 4 + k];
        if (std::fabs(cl - cr) > best) { best = std::fabs(cl - cr); ax = k; }
      }
      float cl = T.nodes[8 * l + ax] + T.nodes[8 * l + 4 + ax], cr = T.nodes[8 * r + ax] + T.nodes[8 * r + 4 + ax];
      axis[i] = ax | (cl <= cr ? 0 : 4); // bit 2: left child is the HIGH one along the axis
    }
    rmin = 1e30f; rmax = 0;
    for (uint32_t k = 0; k < T.n_tree_slots; k++)
      if (T.slot_index[k] != 0xffffffffu) {
        float r = std::fabs(sc.s[T.slot_index[k]].r);
        rmin = std::min(rmin, r); rmax = std::max(rmax, r);
      }
    tiles_x = (sc.W + 7) / 8; tiles_y = (sc.H + 7) / 8;
    n_items = (uint64_t)tiles_x * tiles_y * op.passes * 64;
    fprintf(stderr, "tree: %u nodes, %u slots, %u outliers, depth %u, s0 %.2f, rmin %.3f\n", T.n_nodes, T.n_slots,
            T.n_outliers, T.depth, T.s0, rmin);
  }

  // ---------------------------------------------------------------- the walk
  void setup(Lane& L) {
    for (int k = 0; k < 3; k++) {
      L.p[k] = L.o[k] - T.c0[k];
      float r = 1.0f / L.d[k];
      L.inv[k] = std::min(std::max(r, -1e18f), 1e18f);
    }
    L.a = L.d[0] * L.d[0] + L.d[1] * L.d[1] + L.d[2] * L.d[2];
    if (op.margin == 0) {
      L.mrg = 1.25e-3f * (std::fabs(L.p[0]) + std::fabs(L.p[1]) + std::fabs(L.p[2]) + T.s0) + 1e-6f;
    } else {
      const float D = std::sqrt(L.p[0] * L.p[0] + L.p[1] * L.p[1] + L.p[2] * L.p[2]) + T.s0;
      const float u = 5.9604645e-8f;
      L.mrg = (std::sqrt(rmin * rmin + 54.0f * u * 1.2f * D * D) - rmin) + 1e-6f * rmax + 8.0f * u * D * 1.8f + 1e-6f;
    }
    L.closest = 1e5f; L.hit_pos = -1; L.l_cnt = 0; L.q_cnt = 0; L.nodes_this_seg = 0;
    L.cur = 0; L.sp = 0;
    if (op.ordered) L.stack[L.sp++] = 0;
    for (uint32_t i = T.n_tree_slots; i < T.n_tree_slots + T.n_outliers; i++) {
      if (test_slot(L, i)) {
        if (op.outlier_exact) exact(L, (uint16_t)i);
        else if (L.q_cnt < 8) L.cq[L.q_cnt++] = (uint16_t)i;
      }
    }
  }
  bool walking(const Lane& L) const { return L.alive && !L.new_path && (op.ordered ? L.sp > 0 : L.cur < T.n_nodes); }
  bool box(const Lane& L, uint32_t node) const {
    float tn = 0.f, tf = 3e38f;
    for (int k = 0; k < 3; k++) {
      float lo = T.nodes[8 * node + k] - T.c0[k], hi = T.nodes[8 * node + 4 + k] - T.c0[k];
      float t1 = (lo - (L.p[k] + L.mrg)) * L.inv[k], t2 = (hi - (L.p[k] - L.mrg)) * L.inv[k];
      tn = std::max(tn, std::min(t1, t2)); tf = std::min(tf, std::max(t1, t2));
    }
    if (op.prune) tf = std::min(tf, L.closest);
    return tn <= tf;
  }
  void node_step(Lane& L) {
    L.nodes_this_seg++; node_visits++;
    uint32_t node;
    if (op.ordered) node = L.stack[--L.sp]; else node = L.cur;
    const bool through = box(L, node);
    const uint32_t leaf = ptbvh::bits(T.nodes[8 * node + 7]);
    if (through && leaf != ptbvh::kInner) L.lq[L.l_cnt++] = (uint16_t)(leaf / 4);
    if (op.ordered) {
      if (through && leaf == ptbvh::kInner) {
        const int ax = axis[node] & 3; const bool left_high = (axis[node] & 4) != 0;
        // direction positive along the axis: the LOW child is near
        const bool near_is_left = (L.d[ax] > 0) != left_high;
        const int nr = near_is_left ? left[node] : right[node], fr = near_is_left ? right[node] : left[node];
        L.stack[L.sp++] = fr; L.stack[L.sp++] = nr;
      }
    } else {
      L.cur = through ? node + 1 : ptbvh::bits(T.nodes[8 * node + 3]);
    }
  }
  // PT_TEST: is the slot a candidate?
  bool test_slot(const Lane& L, uint32_t pos) const {
    const float* g = &T.slots[4 * pos];
    float oc[3] = {L.o[0] - g[0], L.o[1] - g[1], L.o[2] - g[2]};
    float hb = oc[0] * L.d[0] + oc[1] * L.d[1] + oc[2] * L.d[2];
    float cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - g[3];
    float disc = hb * hb - L.a * cc;
    if (disc < 0) return false;
    if (cc > 0 && hb >= 0) return false;
    return true;
  }
  void leaf_step(Lane& L) {
    leaf_visits++;
    uint16_t lf;
    if (op.leaf_fifo) { lf = L.lq[0]; for (int k = 1; k < L.l_cnt; k++) L.lq[k - 1] = L.lq[k]; L.l_cnt--; }
    else lf = L.lq[--L.l_cnt];
    for (int k = 0; k < 4; k++)
      if (test_slot(L, lf * 4u + k)) L.cq[L.q_cnt++] = (uint16_t)(lf * 4u + k);
  }
  void exact(Lane& L, uint16_t pos) {
    exact_evals++;
    const float* g = &T.slots[4 * pos];
    float oc[3] = {L.o[0] - g[0], L.o[1] - g[1], L.o[2] - g[2]};
    float hb = oc[0] * L.d[0] + oc[1] * L.d[1] + oc[2] * L.d[2];
    float cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - g[3];
    float disc = hb * hb - L.a * cc;
    float sq = std::sqrt(std::max(disc, 0.f));
    float v = (-hb - sq) / L.a;
    if (v < 0.001f) v = (-hb + sq) / L.a;
    if (!(v < 0.001f) && v < L.closest) { L.closest = v; L.hit_pos = pos; }
  }
  void exact_step(Lane& L) { exact(L, L.cq[--L.q_cnt]); }

  // ---------------------------------------------------------------- items, camera, shading
  bool refill(Lane& L, uint64_t& pool_next, uint64_t& pool_end) {
    for (;;) {
      if (pool_next == pool_end) {
        if (head >= n_items) return false;
        pool_next = head; head += op.chunk; pool_end = std::min<uint64_t>(head, n_items);
      }
      uint64_t item = pool_next++;
      uint32_t per_tile = 64u * op.passes;
      uint32_t tile = (uint32_t)(item / per_tile), rem = (uint32_t)(item % per_tile);
      uint32_t pass = rem >> 6, l = rem & 63;
      uint32_t ty = tile / tiles_x, tx = tile % tiles_x;
      uint32_t x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
      if (x >= sc.W || y >= sc.H) continue;
      L.px = (float)x; L.py = (float)y;
      L.rng.s = 0x9E3779B97F4A7C15ull * (item + 1) + pass;
      L.rng.next();
      L.samples_left = op.spp;
      L.alive = true; L.new_path = true;
      return true;
    }
  }
  void camera_ray(Lane& L) {
    const float* c = sc.cam;
    float s = (L.px + 0.5f + L.rng.next()) / sc.W, t = (L.py + 0.5f + L.rng.next()) / sc.H;
    float ang = 6.2831853f * L.rng.next(), rr = std::sqrt(L.rng.next());
    float rdx = c[18] * rr * std::cos(ang), rdy = c[18] * rr * std::sin(ang);
    for (int k = 0; k < 3; k++) {
      float off = c[12 + k] * rdx + c[15 + k] * rdy;
      L.d[k] = c[9 + k] + s * c[3 + k] + t * c[6 + k] - c[k] - off;
      L.o[k] = c[k] + off;
    }
    L.depth = 0; L.new_path = false;
  }
  void unit_sphere(XRng& r, float* v) {
    float hx = 2 * r.next() - 1, ph = 6.2831853f * r.next(), rr = std::cbrt(r.next());
    float sq = std::sqrt(std::max(0.f, 1 - hx * hx));
    v[0] = rr * sq * std::sin(ph); v[1] = rr * sq * std::cos(ph); v[2] = rr * hx;
  }
  // returns true when the path continues (L.o / L.d hold the next ray)
  bool shade(Lane& L) {
    bool finished = false;
    { int b = std::min(15u, L.nodes_this_seg / 5u); if (L.hit_pos < 0) hist_miss[b]++; else if ((uint32_t)L.hit_pos >= T.n_tree_slots) hist_ground[b]++; else hist_hit[b]++; }
    if (L.hit_pos < 0) finished = true;
    else {
      const uint32_t idx = T.slot_index[L.hit_pos];
      const Sph& S = sc.s[idx];
      float p[3], n[3];
      for (int k = 0; k < 3; k++) p[k] = L.o[k] + L.d[k] * L.closest;
      for (int k = 0; k < 3; k++) n[k] = (p[k] - S.c[k]) / S.r;
      float dn = L.d[0] * n[0] + L.d[1] * n[1] + L.d[2] * n[2];
      bool front = dn < 0;
      if (!front) for (int k = 0; k < 3; k++) n[k] = -n[k];
      float nd[3];
      if (S.type == 0) {
        float v[3]; unit_sphere(L.rng, v);
        float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-20f;
        for (int k = 0; k < 3; k++) nd[k] = n[k] + v[k] / l;
      } else if (S.type == 1) {
        float v[3]; unit_sphere(L.rng, v);
        float k2 = 2 * (L.d[0] * n[0] + L.d[1] * n[1] + L.d[2] * n[2]);
        for (int k = 0; k < 3; k++) nd[k] = L.d[k] - k2 * n[k] + S.fuzz * v[k];
        if (!(nd[0] * n[0] + nd[1] * n[1] + nd[2] * n[2] > 0)) finished = true;
      } else if (S.type == 2) {
        float ratio = front ? 1.0f / S.ri : S.ri;
        float il = 1.0f / std::sqrt(L.a);
        float ud[3] = {L.d[0] * il, L.d[1] * il, L.d[2] * il};
        float ct = std::min(1.0f, -(ud[0] * n[0] + ud[1] * n[1] + ud[2] * n[2]));
        float st = std::sqrt(std::max(0.f, 1 - ct * ct));
        float r0 = (1 - ratio) / (1 + ratio); r0 *= r0;
        float refl = r0 + (1 - r0) * std::pow(1 - ct, 5.f);
        if (ratio * st > 1 || refl > L.rng.next()) {
          float k2 = 2 * (ud[0] * n[0] + ud[1] * n[1] + ud[2] * n[2]);
          for (int k = 0; k < 3; k++) nd[k] = ud[k] - k2 * n[k];
        } else {
          float dni = n[0] * ud[0] + n[1] * ud[1] + n[2] * ud[2];
          float kk = 1 - ratio * ratio * (1 - dni * dni);
          float tt = ratio * dni + std::sqrt(std::max(kk, 0.f));
          for (int k = 0; k < 3; k++) nd[k] = ratio * ud[k] - tt * n[k];
        }
      } else finished = true;
      if (!finished) {
        for (int k = 0; k < 3; k++) { L.o[k] = p[k]; L.d[k] = nd[k]; }
        L.depth++;
        if (L.depth >= (int)sc.depth) finished = true;
      }
    }
    if (finished) {
      L.samples_left--;
      if (L.samples_left <= 0) L.alive = false; else L.new_path = true;
      return false;
    }
    return true;
  }

  // ---------------------------------------------------------------- mode 0: the round-1 kernel
  void run_wave_lockstep() {
    Lane L[64];
    bool carried[64] = {false};
    uint64_t pn = 0, pe = 0;
    for (;;) {
      bool any_cam = false, any_refill = false;
      for (auto& l : L) if (!l.alive && !l.exhausted) { any_refill = true; if (!refill(l, pn, pe)) l.exhausted = true; }
      int n_cam = 0;
      for (auto& l : L) if (l.alive && l.new_path) { camera_ray(l); any_cam = true; n_cam++; }
      int live = 0; for (auto& l : L) live += l.alive;
      if (!live) break;
      wave_steps++;
      segs += live;
      if (any_refill) cost += C_REFILL;
      if (any_cam) { cost += C_CAM; ph_cam.iters++; ph_cam.lanes += n_cam; }
      cost += C_SETUP + 2 * C_PARK + C_MISC;
      for (int i = 0; i < 64; i++) if (L[i].alive && !carried[i]) setup(L[i]);
      for (int i = 0; i < 64; i++) carried[i] = false;
      bool stop = false;
      for (;;) {
        int burst = 0;
        for (;;) {
          int on = 0; bool full = false;
          for (auto& l : L) { if (walking(l)) on++; if (l.l_cnt == 8) full = true; }
          if (!on || full) break;
          if (op.carry && on < op.carry && on < live / 2) { stop = true; break; }
          if (op.eager && burst >= op.eager) break;
          for (auto& l : L) if (walking(l)) node_step(l);
          ph_node.iters++; ph_node.lanes += on; cost += C_NODE + op.prune;
          burst++;
        }
        for (;;) {
          for (;;) {
            int busy = 0;
            for (auto& l : L) if (l.l_cnt > 0 && l.q_cnt <= 4) busy++;
            if (!busy) break;
            for (auto& l : L) if (l.l_cnt > 0 && l.q_cnt <= 4) leaf_step(l);
            ph_leaf.iters++; ph_leaf.lanes += busy; cost += C_LEAF;
          }
          bool any = false; for (auto& l : L) if (l.l_cnt) any = true;
          if (!any) break;
          drain(L, 4);
        }
        if (op.eager) drain(L, 0);
        bool any = false; for (auto& l : L) if (walking(l)) any = true;
        if (!any || stop) break;
      }
      drain(L, 0);
      int n = 0;
      for (int i = 0; i < 64; i++) if (L[i].alive) { if (walking(L[i])) { carried[i] = true; segs--; continue; } shade(L[i]); n++; }
      ph_shade.iters++; ph_shade.lanes += n; cost += C_SHADE;
    }
  }
  void drain(Lane* L, int keep) {
    for (;;) {
      int n = 0; for (int i = 0; i < 64; i++) if (L[i].q_cnt > keep) n++;
      if (!n) break;
      for (int i = 0; i < 64; i++) if (L[i].q_cnt > keep) exact_step(L[i]);
      ph_exact.iters++; ph_exact.lanes += n; cost += C_EXACT;
    }
  }

  // ---------------------------------------------------------------- mode 1: phases by population
  void run_wave_sched() {
    Lane L[64];
    bool need_setup[64] = {false};
    uint64_t pn = 0, pe = 0;
    bool force = false;
    for (;;) {
      sched_trips++;
      cost += C_SCHED;
      bool did = false;
      auto thr = [&](int t) { return force ? 1 : t; };
      // REFILL + CAMERA (+ walk set-up of the fresh rays)
      {
        int n = 0; for (auto& l : L) if ((!l.alive && !l.exhausted) || (l.alive && l.new_path)) n++;
        if (n >= thr(op.t_cam)) {
          bool any_refill = false; int n_cam = 0;
          for (auto& l : L) if (!l.alive && !l.exhausted) { any_refill = true; if (!refill(l, pn, pe)) l.exhausted = true; }
          for (auto& l : L) if (l.alive && l.new_path) { camera_ray(l); setup(l); segs++; n_cam++; }
          if (any_refill) cost += C_REFILL;
          if (n_cam) { cost += C_CAM + C_SETUP + 2 * C_PARK; ph_cam.iters++; ph_cam.lanes += n_cam; }
          did = true;
        }
      }
      // NODE
      {
        auto can = [&](const Lane& l) { return walking(l) && l.l_cnt < 8; };
        int n = 0; for (auto& l : L) if (can(l)) n++;
        if (n >= thr(op.t_node)) {
          for (int it = 0; it < op.node_burst; it++) {
            n = 0; for (auto& l : L) if (can(l)) n++;
            if (n < (force ? 1 : op.t_node_exit)) break;
            for (auto& l : L) if (can(l)) node_step(l);
            ph_node.iters++; ph_node.lanes += n; cost += C_NODE + op.prune + 2;
          }
          did = true;
        }
      }
      // LEAF
      {
        auto can = [&](const Lane& l) { return l.l_cnt > 0 && l.q_cnt <= 4; };
        int n = 0; for (auto& l : L) if (can(l)) n++;
        if (n >= thr(op.t_leaf)) {
          for (;;) {
            n = 0; for (auto& l : L) if (can(l)) n++;
            if (n < (force ? 1 : op.t_leaf_exit)) break;
            for (auto& l : L) if (can(l)) leaf_step(l);
            ph_leaf.iters++; ph_leaf.lanes += n; cost += C_LEAF;
            if (force) break;
          }
          did = true;
        }
      }
      // EXACT
      {
        int n = 0; for (auto& l : L) if (l.q_cnt > 0) n++;
        if (n >= thr(op.t_exact)) {
          for (;;) {
            n = 0; for (auto& l : L) if (l.q_cnt > 0) n++;
            if (n < (force ? 1 : op.t_exact_exit)) break;
            for (auto& l : L) if (l.q_cnt > 0) exact_step(l);
            ph_exact.iters++; ph_exact.lanes += n; cost += C_EXACT;
            if (force) break;
          }
          did = true;
        }
      }
      // SHADE (+ walk set-up of the continuing rays)
      {
        auto ready = [&](const Lane& l) { return l.alive && !l.new_path && !walking(l) && l.l_cnt == 0 && l.q_cnt == 0; };
        int n = 0; for (auto& l : L) if (ready(l)) n++;
        if (n >= thr(op.t_shade)) {
          for (auto& l : L) if (ready(l)) { if (shade(l)) { setup(l); segs++; } }
          ph_shade.iters++; ph_shade.lanes += n; cost += C_SHADE + C_SETUP + 2 * C_PARK;
          did = true;
        }
      }
      (void)need_setup;
      bool all_done = true; for (auto& l : L) if (!l.exhausted) all_done = false;
      if (all_done) break;
      if (!did && force) { fprintf(stderr, "stuck\n"); exit(2); }
      if (!did) forced++;
      force = !did;
    }
  }


  // ---------------------------------------------------------------- mode 2: uniform grid (3D-DDA)
  struct Grid {
    float lo[3], hi[3], h[3]; int n[3];
    std::vector<uint32_t> start, count;     // per cell
    std::vector<float> slots;               // 4 per entry (copies)
    std::vector<uint32_t> slot_sphere;
    std::vector<uint32_t> big;              // spheres tested for every ray
  } G;
  Phase ph_dda;
  double cells_visited = 0, cells_nonempty = 0, fallback_rays = 0, far_all[4] = {0}, far_enter[4] = {0};
  static constexpr double C_DDA = 15, C_SETUP_G = 120;

  void build_grid() {
    // bounds over the small spheres
    std::vector<uint32_t> small;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    // first pass: heuristic cell edge from all spheres that are not far-out giants (use the tree's outlier rule)
    std::vector<char> is_out(sc.n, 0);
    for (uint32_t k = T.n_tree_slots; k < T.n_slots; k++) if (T.slot_index[k] != 0xffffffffu) is_out[T.slot_index[k]] = 1;
    double vol_lo[3] = {1e30, 1e30, 1e30}, vol_hi[3] = {-1e30, -1e30, -1e30}; uint32_t cnt = 0;
    for (uint32_t i = 0; i < sc.n; i++) if (!is_out[i]) {
      for (int k = 0; k < 3; k++) { vol_lo[k] = std::min<double>(vol_lo[k], sc.s[i].c[k]); vol_hi[k] = std::max<double>(vol_hi[k], sc.s[i].c[k]); }
      cnt++;
    }
    std::vector<float> rs; for (uint32_t i = 0; i < sc.n; i++) if (!is_out[i]) rs.push_back(std::fabs(sc.s[i].r));
    std::nth_element(rs.begin(), rs.begin() + rs.size() / 2, rs.end());
    const double rmed = rs[rs.size() / 2];
    double ext[3]; for (int k = 0; k < 3; k++) ext[k] = std::max(vol_hi[k] - vol_lo[k], 2 * rmed);
    // cells ~ n / 2, flat axes collapse to one layer
    double vol = ext[0] * ext[1] * ext[2];
    double edge = std::cbrt(vol / (cnt / 2.0));
    for (int it = 0; it < 3; it++) { // re-solve with collapsed axes
      double v = 1; int free_axes = 0;
      for (int k = 0; k < 3; k++) if (ext[k] > 1.5 * edge) { v *= ext[k]; free_axes++; }
      if (free_axes) edge = std::pow(v / (cnt / 2.0), 1.0 / free_axes);
    }
    edge *= op.cell_x100 / 100.0;
    const double big_r = op.big_x100 / 100.0 * edge;
    for (uint32_t i = 0; i < sc.n; i++) {
      if (is_out[i] || std::fabs(sc.s[i].r) > big_r) { G.big.push_back(i); continue; }
      small.push_back(i);
      for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], sc.s[i].c[k] - std::fabs(sc.s[i].r)); hi[k] = std::max(hi[k], sc.s[i].c[k] + std::fabs(sc.s[i].r)); }
    }
    const float infl = 0.04f; // registration inflation (per-ray margin bound)
    for (int k = 0; k < 3; k++) {
      G.lo[k] = lo[k] - infl; G.hi[k] = hi[k] + infl;
      G.n[k] = std::max(1, (int)std::floor((G.hi[k] - G.lo[k]) / edge + 0.5));
      G.h[k] = (G.hi[k] - G.lo[k]) / G.n[k];
    }
    const size_t nc = (size_t)G.n[0] * G.n[1] * G.n[2];
    std::vector<std::vector<uint32_t>> lists(nc);
    for (uint32_t i : small) {
      int a[3], b[3];
      for (int k = 0; k < 3; k++) {
        a[k] = std::max(0, std::min(G.n[k] - 1, (int)std::floor((sc.s[i].c[k] - std::fabs(sc.s[i].r) - infl - G.lo[k]) / G.h[k])));
        b[k] = std::max(0, std::min(G.n[k] - 1, (int)std::floor((sc.s[i].c[k] + std::fabs(sc.s[i].r) + infl - G.lo[k]) / G.h[k])));
      }
      for (int z = a[2]; z <= b[2]; z++) for (int y = a[1]; y <= b[1]; y++) for (int x = a[0]; x <= b[0]; x++)
        lists[((size_t)z * G.n[1] + y) * G.n[0] + x].push_back(i);
    }
    G.start.resize(nc); G.count.resize(nc);
    size_t entries = 0, nonempty = 0, maxc = 0;
    for (size_t c = 0; c < nc; c++) {
      G.start[c] = (uint32_t)G.slot_sphere.size(); G.count[c] = (uint32_t)lists[c].size();
      for (uint32_t i : lists[c]) {
        G.slot_sphere.push_back(i);
        G.slots.push_back(sc.s[i].c[0]); G.slots.push_back(sc.s[i].c[1]); G.slots.push_back(sc.s[i].c[2]); G.slots.push_back(sc.s[i].r * sc.s[i].r);
      }
      entries += lists[c].size(); nonempty += !lists[c].empty(); maxc = std::max(maxc, lists[c].size());
    }
    fprintf(stderr, "grid: %d x %d x %d cells of %.2f x %.2f x %.2f, %zu entries for %zu spheres (%zu non-empty cells, max %zu per cell), %zu always-tested\n",
            G.n[0], G.n[1], G.n[2], G.h[0], G.h[1], G.h[2], entries, small.size(), nonempty, maxc, G.big.size());
  }

  struct GLane { // DDA state beside Lane
    bool active = false; int cell[3], step[3]; float tmax[3], tdelta[3]; float t_exit = 0;
    uint32_t pend_start = 0, pend_count = 0; // entries of the current cell not tested yet
  };
  bool test_g(const Lane& L, const float* g) const {
    float oc[3] = {L.o[0] - g[0], L.o[1] - g[1], L.o[2] - g[2]};
    float hb = oc[0] * L.d[0] + oc[1] * L.d[1] + oc[2] * L.d[2];
    float cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - g[3];
    float disc = hb * hb - L.a * cc;
    return !(disc < 0) && !(cc > 0 && hb >= 0);
  }
  void exact_g(Lane& L, const float* g, int id) {
    exact_evals++;
    float oc[3] = {L.o[0] - g[0], L.o[1] - g[1], L.o[2] - g[2]};
    float hb = oc[0] * L.d[0] + oc[1] * L.d[1] + oc[2] * L.d[2];
    float cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - g[3];
    float disc = hb * hb - L.a * cc;
    float sq = std::sqrt(std::max(disc, 0.f));
    float v = (-hb - sq) / L.a;
    if (v < 0.001f) v = (-hb + sq) / L.a;
    if (!(v < 0.001f) && v < L.closest) { L.closest = v; L.hit_pos = id; }
  }
  // hit_pos encoding in mode 2: sphere index directly
  void setup_grid(Lane& L, GLane& g, std::vector<uint32_t>& cand) {
    L.a = L.d[0] * L.d[0] + L.d[1] * L.d[1] + L.d[2] * L.d[2];
    L.closest = 1e5f; L.hit_pos = -1; L.nodes_this_seg = 0;
    cand.clear();
    for (uint32_t i : G.big) {
      float gg[4] = {sc.s[i].c[0], sc.s[i].c[1], sc.s[i].c[2], sc.s[i].r * sc.s[i].r};
      if (test_g(L, gg)) cand.push_back(0x80000000u | i);
    }
    float t0 = 0.f, t1 = 3e38f;
    for (int k = 0; k < 3; k++) {
      float inv = 1.0f / L.d[k];
      float ta = (G.lo[k] - L.o[k]) * inv, tb = (G.hi[k] - L.o[k]) * inv;
      t0 = std::max(t0, std::min(ta, tb)); t1 = std::min(t1, std::max(ta, tb));
    }
    g.active = t0 <= t1; g.pend_count = 0;
    {
      float pp[3] = {L.o[0] - T.c0[0], L.o[1] - T.c0[1], L.o[2] - T.c0[2]};
      float D = std::sqrt(pp[0] * pp[0] + pp[1] * pp[1] + pp[2] * pp[2]);
      int b = D < 2 * T.s0 ? 0 : (D < 4 * T.s0 ? 1 : (D < 8 * T.s0 ? 2 : 3));
      far_all[b]++; if (g.active && L.closest >= t0) far_enter[b]++;
    }
    if (!g.active) return;
    for (int k = 0; k < 3; k++) {
      float p = L.o[k] + L.d[k] * t0;
      int c = (int)std::floor((p - G.lo[k]) / G.h[k]);
      c = std::max(0, std::min(G.n[k] - 1, c));
      g.cell[k] = c; g.step[k] = L.d[k] > 0 ? 1 : -1;
      if (L.d[k] != 0) {
        float nb = G.lo[k] + (c + (L.d[k] > 0 ? 1 : 0)) * G.h[k];
        g.tmax[k] = (nb - L.o[k]) / L.d[k]; g.tdelta[k] = G.h[k] / std::fabs(L.d[k]);
      } else { g.tmax[k] = 3e38f; g.tdelta[k] = 3e38f; }
    }
  }
  void run_wave_grid() {
    Lane L[64]; GLane Gs[64]; std::vector<uint32_t> cand[64];
    bool carried[64] = {false};
    uint64_t pn = 0, pe = 0;
    for (;;) {
      bool any_cam = false, any_refill = false; int n_cam = 0;
      for (auto& l : L) if (!l.alive && !l.exhausted) { any_refill = true; if (!refill(l, pn, pe)) l.exhausted = true; }
      for (auto& l : L) if (l.alive && l.new_path) { camera_ray(l); any_cam = true; n_cam++; }
      int live = 0; for (auto& l : L) live += l.alive;
      if (!live) break;
      wave_steps++; segs += live;
      if (any_refill) cost += C_REFILL;
      if (any_cam) { cost += C_CAM; ph_cam.iters++; ph_cam.lanes += n_cam; }
      cost += C_SETUP_G + 2 * C_PARK + C_MISC;
      for (int i = 0; i < 64; i++) { if (L[i].alive) { if (!carried[i]) setup_grid(L[i], Gs[i], cand[i]); } else Gs[i].active = false; carried[i] = false; }
      // the always-tested spheres first: closest is known before the walk
      auto drain_all = [&]() {
        for (;;) {
          int n = 0; for (int i = 0; i < 64; i++) if (!cand[i].empty()) n++;
          if (!n) break;
          for (int i = 0; i < 64; i++) if (!cand[i].empty()) {
            uint32_t c = cand[i].back(); cand[i].pop_back();
            if (c & 0x80000000u) { uint32_t s = c & 0x7fffffffu; float gg[4] = {sc.s[s].c[0], sc.s[s].c[1], sc.s[s].c[2], sc.s[s].r * sc.s[s].r}; exact_g(L[i], gg, (int)s); }
            else exact_g(L[i], &G.slots[4 * c], (int)G.slot_sphere[c]);
          }
          ph_exact.iters++; ph_exact.lanes += n; cost += C_EXACT;
        }
      };
      drain_all();
      for (;;) {
        if (op.carry) { int on = 0; for (int i = 0; i < 64; i++) if (Gs[i].active || Gs[i].pend_count) on++; if (on && on < op.carry && on < live / 2) break; }
        // advance every lane to its next non-empty cell (or out)
        for (;;) {
          int mv = 0;
          for (int i = 0; i < 64; i++) {
            GLane& g = Gs[i];
            if (!g.active || g.pend_count || (op.block && !cand[i].empty())) continue;
            mv++;
          }
          if (!mv) break;
          for (int i = 0; i < 64; i++) {
            GLane& g = Gs[i];
            if (!g.active || g.pend_count || (op.block && !cand[i].empty())) continue;
            // stand on cell: look at it, then step
            const size_t c = ((size_t)g.cell[2] * G.n[1] + g.cell[1]) * G.n[0] + g.cell[0];
            cells_visited++; L[i].nodes_this_seg++;
            int ax = 0; if (g.tmax[1] < g.tmax[ax]) ax = 1; if (g.tmax[2] < g.tmax[ax]) ax = 2;
            g.t_exit = g.tmax[ax];
            if (G.count[c]) { g.pend_start = G.start[c]; g.pend_count = G.count[c]; cells_nonempty++; }
            // step (applied now; termination is checked against t_exit after the cell's tests)
            g.cell[ax] += g.step[ax]; g.tmax[ax] += g.tdelta[ax];
            if (g.cell[ax] < 0 || g.cell[ax] >= G.n[ax]) g.active = false;
            if (!G.count[c] && L[i].closest <= g.t_exit) g.active = false;
          }
          ph_dda.iters++; ph_dda.lanes += mv; cost += C_DDA;
        }
        int busy = 0; for (int i = 0; i < 64; i++) if (Gs[i].pend_count && !(op.block && !cand[i].empty())) busy++;
        if (!busy) {
          int nb = 0; for (int i = 0; i < 64; i++) if (!cand[i].empty()) nb++;
          if (!nb) break;
          drain_all(); // nobody else can move: the waiting lanes are served
          for (int i = 0; i < 64; i++) if (Gs[i].active && !Gs[i].pend_count && L[i].closest <= Gs[i].t_exit) Gs[i].active = false;
          continue;
        }
        // one leaf round: up to four entries of the pending cell
        for (int i = 0; i < 64; i++) {
          GLane& g = Gs[i];
          if (!g.pend_count || (op.block && !cand[i].empty())) continue;
          leaf_visits++;
          uint32_t k = std::min(4u, g.pend_count);
          for (uint32_t j = 0; j < k; j++) if (test_g(L[i], &G.slots[4 * (g.pend_start + j)])) cand[i].push_back(g.pend_start + j);
          g.pend_start += k; g.pend_count -= k;
        }
        ph_leaf.iters++; ph_leaf.lanes += busy; cost += C_LEAF;
        if (op.block) { int nb = 0; for (int i = 0; i < 64; i++) if (!cand[i].empty()) nb++; if (nb >= op.block) drain_all(); }
        else drain_all();
        for (int i = 0; i < 64; i++) if (cand[i].empty() && Gs[i].active && !Gs[i].pend_count && L[i].closest <= Gs[i].t_exit) Gs[i].active = false;
      }
      int n = 0;
      for (int i = 0; i < 64; i++) if (L[i].alive) {
        if (Gs[i].active || Gs[i].pend_count) { carried[i] = true; segs--; continue; }
        // shade() looks the sphere up through the tree's slot table: translate
        Lane& l = L[i];
        int sphere = l.hit_pos; int pos = -1;
        if (sphere >= 0) { for (uint32_t k = 0; k < T.n_slots; k++) if (T.slot_index[k] == (uint32_t)sphere) { pos = (int)k; break; } }
        l.hit_pos = pos;
        shade(l); n++;
      }
      ph_shade.iters++; ph_shade.lanes += n; cost += C_SHADE;
    }
  }

  void run() {
    for (int w = 0; head < n_items; w++) {
      if (op.mode == 0) run_wave_lockstep(); else if (op.mode == 2) run_wave_grid(); else run_wave_sched();
    }
    auto pr = [&](const char* nm, const Phase& p, double c) {
      printf("  %-6s %8.2f iters/64seg  x %5.1f lanes  = %7.1f VALU/64seg\n", nm, p.iters / segs * 64,
             p.iters ? p.lanes / p.iters : 0.0, p.iters * c / segs * 64);
    };
    printf("segments %.0f  cost/64seg %.1f  nodes/seg %.2f leaves/seg %.2f exact/seg %.2f  sched trips/64seg %.2f forced %.2f\n",
           segs, cost / segs * 64, node_visits / segs, leaf_visits / segs, exact_evals / segs, sched_trips / segs * 64,
           forced / segs * 64);
    printf("  walk length (nodes, buckets of 5):\n");
    for (int b = 0; b < 16; b++) printf("   %3d+: miss %.4f ground %.4f sphere %.4f\n", b * 5, hist_miss[b] / segs, hist_ground[b] / segs, hist_hit[b] / segs);
    if (op.mode == 2) { for (int b = 0; b < 4; b++) printf("  origin distance class %d (<2,<4,<8,>=8 x s0): %.5f of rays, %.5f enter the grid\n", b, far_all[b] / segs, far_enter[b] / segs);
      printf("  cells/seg %.2f non-empty %.2f\n", cells_visited / segs, cells_nonempty / segs); pr("dda", ph_dda, C_DDA); }
    pr("node", ph_node, C_NODE); pr("leaf", ph_leaf, C_LEAF); pr("exact", ph_exact, C_EXACT);
    pr("shade", ph_shade, C_SHADE); pr("cam", ph_cam, C_CAM);
  }
};

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: wavesim scene [key=value ...]\n"); return 1; }
  Sim S;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("scene"); return 1; }
  uint32_t hd[4];
  if (fread(hd, 4, 4, f) != 4) return 1;
  S.sc.n = hd[0]; S.sc.W = hd[1]; S.sc.H = hd[2]; S.sc.depth = hd[3];
  if (fread(S.sc.cam, 4, 19, f) != 19) return 1;
  if (fread(&S.sc.bg, 4, 1, f) != 1) return 1;
  S.sc.s.resize(S.sc.n);
  if (fread(S.sc.s.data(), 48, S.sc.n, f) != S.sc.n) return 1;
  fclose(f);
  std::map<std::string, int*> keys = {
      {"mode", &S.op.mode}, {"ordered", &S.op.ordered}, {"prune", &S.op.prune}, {"margin", &S.op.margin},
      {"fifo", &S.op.leaf_fifo}, {"spp", &S.op.spp}, {"passes", &S.op.passes}, {"chunk", &S.op.chunk},
      {"t_cam", &S.op.t_cam}, {"t_node", &S.op.t_node}, {"t_node_exit", &S.op.t_node_exit}, {"t_leaf", &S.op.t_leaf},
      {"t_leaf_exit", &S.op.t_leaf_exit}, {"t_exact", &S.op.t_exact}, {"t_exact_exit", &S.op.t_exact_exit},
      {"t_shade", &S.op.t_shade}, {"burst", &S.op.node_burst}, {"outlier_exact", &S.op.outlier_exact}, {"eager", &S.op.eager}, {"cell", &S.op.cell_x100}, {"big", &S.op.big_x100}, {"carry", &S.op.carry}, {"block", &S.op.block}};
  for (int i = 2; i < argc; i++) {
    char* eq = strchr(argv[i], '=');
    if (!eq) continue;
    std::string k(argv[i], eq - argv[i]);
    if (k == "t") { int v = atoi(eq + 1); S.op.t_cam = S.op.t_node = S.op.t_leaf = S.op.t_exact = S.op.t_shade = v; continue; }
    if (!keys.count(k)) { fprintf(stderr, "unknown key %s\n", k.c_str()); return 1; }
    *keys[k] = atoi(eq + 1);
  }
  S.build();
  if (S.op.mode == 2) S.build_grid();
  S.run();
  return 0;
}
