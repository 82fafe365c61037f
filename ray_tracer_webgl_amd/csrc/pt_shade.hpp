// pt_shade.hpp — what happens at the end of a segment: static/shader.frag:304-338 (miss ->
// background :289-294, hit -> hit record :166-171, front face :136-143, scatter :210-286,
// depth bookkeeping :300,:338) and the end-of-item store.
//
// EXACTNESS ARGUMENT carried by this file: the outward normal's three divisions by the radius use
// div_core under a per-lane range guard (|radius| in [2^-20, 2^20), every numerator with
// 2^-103 <= |n| < 2^76; a zero numerator takes the plain operator because its quotient's sign of
// zero comes from v_div_fixup); everything else is the oracle's operation order, statement by
// statement.  RNG draws happen exactly where the shader draws them (one random_in_unit_sphere for
// DIFFUSE :217 and METAL :240 alike — also when fuzz is 0 —, one hash1 for GLASS :267).
#pragma once
#include "pt_scene.hpp"

namespace ptk {

// shade the lanes whose walk / scan has finished (`p.alive && !carried`): the closest hit is
// (h.closest, h.hit) — walk kernels: the slot cw.hit_pos, whose copy of the sphere and of its
// material record are read instead of going through the list index
// RR: the opt-in Russian-roulette build of a kernel (PT_OPT_RUSSIAN_ROULETTE); the kernels without it
// carry none of its code or registers
template <typename S, bool RR, bool COUNT>
__device__ __forceinline__ void shade_segment(const PtKernelArgs& A, Path& p, const Hit& h, const Carry& cw, Tally<COUNT>& tally) {
  tally.flag(PT_REG_SHADE_ANY);
  // local copies, written back at the end (see pt_grid_walk.hpp: references would be memory to the
  // passes that run before inlining)
  bool alive = p.alive, new_path = p.new_path;
  const uint32_t slab_index = p.slab_index; const uint32_t item_tile = p.item_tile; uint32_t item_segs = p.item_segs;
  int sample = p.sample, depth = p.depth; float seed = p.seed;
  V3 o = p.o, d = p.d; float a = p.a; V3 col = p.col, sum = p.sum;
  const float closest = h.closest; const int hit = h.hit; const uint32_t hit_pos = cw.hit_pos;
  (void)hit_pos;
  item_segs++;
  bool finished = false; // this camera path is over
  // ONE copy of what several outcomes share.  The lanes of a wave end their segments differently (sky, DIFFUSE,
  // METAL, GLASS ...) and a wave runs every branch some lane takes: the sky and GLASS both want 1 / sqrt(|d|^2), and
  // DIFFUSE, METAL and GLASS all begin their draw with the same seed step + hash.  Both are computed once, in straight
  // line for every lane (no lane masks to set up and restore: a region costs more scalar work than it saves vector work;
  // the seed moves only for the lanes that draw): State::default -1.4 %, config 5 -0.6 %, configs 2 and 4 +-0.1 %
  // (profiles/r04_ab_runs.txt).
  int mtype = -1;
  V3 hp = mk(0.f, 0.f, 0.f), n = hp, alb = hp; bool front = false; float fuzz = 0.f, ri = 1.f, inv_ri = 1.f;
  if (hit >= 0) {
    tally.flag(PT_REG_SHADE_HIT_RECORD);
    float4 g;
    if constexpr (S::TREE) { // the walk's hits come with their slot (same four floats as the list entry)
      if (hit_pos != 0xffffffffu) g = S::slot_at(A, hit_pos);
      else g = S::geom_at(A, (uint32_t)hit);
    } else {
      g = S::geom_at(A, (uint32_t)hit);
    }
    const float4* mp = reinterpret_cast<const float4*>(A.mat + hit);
    if constexpr (S::TREE) { // one load instead of index -> material (two dependent memory round trips)
      if (hit_pos != 0xffffffffu) mp = reinterpret_cast<const float4*>(A.slot_mat + hit_pos);
    }
    float4 m0 = mp[0]; // albedo.xyz, fuzz
    float4 m1 = mp[1]; // refraction_index, type, radius, 1 / refraction_index
    mtype = __float_as_int(m1.y);
    float radius = m1.z;
    fuzz = m0.w; ri = m1.x; inv_ri = m1.w;
    hp = mk(fma_(d.x, closest, o.x), fma_(d.y, closest, o.y), fma_(d.z, closest, o.z)); // hit record, :166-171
    const float nx = hp.x - g.x, ny = hp.y - g.y, nz = hp.z - g.z; // outward normal (p - centre) / radius, :168 (see above)
    V3 on;
    const float n_lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(nx), __builtin_fabsf(ny)), __builtin_fabsf(nz));
    const float n_hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(nx), __builtin_fabsf(ny)), __builtin_fabsf(nz));
    const uint32_t r_guard = div_den_ok(radius) ? f2u(0x1p76f) - f2u(0x1p-103f) : 0u;
    const bool n_odd = f2u(n_lo) - f2u(0x1p-103f) >= r_guard || f2u(n_hi) - f2u(0x1p-103f) >= r_guard;
    const float yr = rcp_newton(radius);
    on = mk(div_core(nx, radius, yr), div_core(ny, radius, yr), div_core(nz, radius, yr));
    if (__builtin_expect(pt_ballot(n_odd) != 0ull, 0)) { // (rare)
      if (n_odd) on = mk(nx / radius, ny / radius, nz / radius);
    }
    front = dot3(d, on) < 0.0f; // :137
    n = front ? on : mk(-on.x, -on.y, -on.z);
    alb = mk(m0.x, m0.y, m0.z);
  }
  const bool sky = hit < 0 && A.background_mode == 0;
  const float inv = inv_sqrt_rn(a); // background() :290, GLASS :253
  float seed_drawn = seed;
  const uint32_t draw = seed_step_hash(seed_drawn); // DIFFUSE :217, METAL :240 (hash3), GLASS :267 (hash1)
  seed = (uint32_t)mtype <= 2u ? seed_drawn : seed;
  if (hit < 0) {
    if (sky) { // background(), :289-294
      tally.flag(PT_REG_SHADE_SKY);
      float uy = d.y * inv;
      float t = 0.5f * (uy + 1.0f);
      float omt = 1.0f - t;
      sum.x += col.x * fma_(0.5f, t, omt);
      sum.y += col.y * fma_(0.7f, t, omt);
      sum.z += col.z * fma_(1.0f, t, omt);
    }
    finished = true;
  } else {
    const V3 p = hp;
    if (mtype == 0 || mtype == 1) {
      V3 rs = random_in_unit_sphere_from(draw); // both DIFFUSE (:217) and METAL (:240) draw one
      V3 nd;
      bool ok = true;
      if (mtype == 0) { // DIFFUSE :212-229
        tally.flag(PT_REG_SHADE_DIFFUSE);
        V3 ruv = normalize3(rs);
        nd = mk(n.x + ruv.x, n.y + ruv.y, n.z + ruv.z);
      } else { // METAL :232-247
        tally.flag(PT_REG_SHADE_METAL);
        V3 refl = reflect3(d, n);
        nd = mk(fma_(fuzz, rs.x, refl.x), fma_(fuzz, rs.y, refl.y), fma_(fuzz, rs.z, refl.z));
        ok = dot3(n, nd) > 0.0f;
      }
      if (ok) {
        o = p; d = nd;
        col.x *= alb.x; col.y *= alb.y; col.z *= alb.z;
      } else {
        finished = true; // absorbed: return vec3(0.) :327-329
      }
    } else if (mtype == 2) { // GLASS :250-282
      tally.flag(PT_REG_SHADE_GLASS);
      float ratio = front ? inv_ri : ri;  // :252 `1.0 / ri`: the host's quotient, same bits as the division here (pt_kernel_args.h)
      V3 ud = mk(d.x * inv, d.y * inv, d.z * inv);
      float cdot = dot3(mk(-ud.x, -ud.y, -ud.z), n);
      float cos_theta = (1.0f < cdot) ? 1.0f : cdot; // min(cdot, 1.0)
      float sin_theta = sqrt_rn(fma_(-cos_theta, cos_theta, 1.0f));
      bool cannot_refract = ratio * sin_theta > 1.0f;
      float refl_amount;
      if constexpr (S::SMALL) {  // r0 from the host: the same bits as reflectance()'s own (pt_api.hip pt_set_spheres)
        const float r0 = A.mat_r0[2u * (uint32_t)hit + (front ? 0u : 1u)];
        const float x = 1.0f - cos_theta, x2 = x * x, x5 = (x2 * x2) * x;
        refl_amount = fma_(1.0f - r0, x5, r0);
      } else {
        refl_amount = reflectance(cos_theta, ratio);
      }
      float rnd = (float)draw * (1.0f / 4294967296.0f); // hash1 :21-24
      V3 nd;
      if (cannot_refract || refl_amount > rnd) {
        nd = reflect3(ud, n);
      } else { // GLSL refract
        tally.flag(PT_REG_SHADE_GLASS_REFRACT);
        float dni = dot3(n, ud);
        float k = fma_(-(ratio * ratio), fma_(-dni, dni, 1.0f), 1.0f);
        if (k < 0.0f) {
          nd = mk(0.f, 0.f, 0.f);
        } else {
          float t = fma_(ratio, dni, sqrt_rn(k));
          nd = mk(fma_(-t, n.x, ratio * ud.x), fma_(-t, n.y, ratio * ud.y),
                  fma_(-t, n.z, ratio * ud.z));
        }
      }
      o = p; d = nd;
      col.x *= alb.x; col.y *= alb.y; col.z *= alb.z;
    } else if (mtype == 3) { // EMISSIVE (extension): radiance = throughput * emission
      sum.x += col.x * alb.x; sum.y += col.y * alb.y; sum.z += col.z * alb.z;
      finished = true;
    } else {
      finished = true; // unrecognised material absorbs, :284-285
    }
    if (!finished) {
      tally.flag(PT_REG_SHADE_CONTINUES);
      a = dot3(d, d);
      depth++;
      if (depth >= A.max_depth) { // loop bound :300 exhausted -> return color :338
        sum.x += col.x; sum.y += col.y; sum.z += col.z;
        finished = true;
      }
      if constexpr (RR) if (!finished && depth >= A.rr_min_depth) {
        // RUSSIAN ROULETTE — opt-in (PT_OPT_RUSSIAN_ROULETTE), NOT the reference's estimator sample for
        // sample: static/shader.frag:297-339 never ends a path early.  After rr_min_depth bounces a path
        // survives with probability q = min(max(throughput), 1) and, if it does, carries throughput / q:
        // E[throughput'] = q (throughput / q) = throughput, so every pixel's expectation — including the
        // `return color` term of depth exhaustion — is the reference's; only the variance and the work
        // change.  One extra hash1 draw per decision, which moves this stream's later random numbers:
        // images of this mode are compared with the oracle statistically (tests/test_gpu_roulette.py),
        // never bit for bit.
        const float q = __builtin_fminf(__builtin_fmaxf(__builtin_fmaxf(col.x, col.y), col.z), 1.0f);
        const float xi = hash1(seed);
        if (!(xi < q)) { // also ends paths whose throughput is 0 or NaN
          finished = true;
        } else {
          const float inv = 1.0f / q;
          col.x *= inv; col.y *= inv; col.z *= inv;
        }
      }
    }
  }

  if (finished) {
    tally.flag(PT_REG_SHADE_FINISHED);
    sample++;
    if (sample >= A.spp) {
      tally.flag(PT_REG_SHADE_ITEM_STORE);
      // (a plain store: the non-temporal form — global_store_dwordx4 ... nt — measured no time change and MORE written traffic,
      // config 5 WRITE_SIZE 0.87 -> 1.37 GiB per 16-pass launch against 0.53 GB of slab: profiles/r06_ab_runs.txt)
      float4 outv = make_float4(sum.x, sum.y, sum.z, (float)A.spp);
      reinterpret_cast<float4*>(A.slab)[slab_index] = outv;
      if (item_tile != 0xffffffffu) atomicMax(&A.tile_cost[item_tile], item_segs);
      alive = false;
    } else {
      new_path = true;
    }
  }
  p.alive = alive; p.new_path = new_path; p.item_segs = item_segs;
  p.sample = sample; p.depth = depth; p.seed = seed;
  p.o = o; p.d = d; p.a = a; p.col = col; p.sum = sum;
}

} // namespace ptk
