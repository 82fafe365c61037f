// pt_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the path tracer.
//
// The hot path of the reference is one fragment-shader invocation per pixel
// (static/shader.frag:406-413): seed -> for each sample {camera ray -> bounce loop {scan the
// sphere list -> scatter}}.  Here it is ONE persistent kernel:
//
//   * work item = (pixel, pass): the serial fp32 seed chain of static/shader.frag:11,21-36 ties
//     all samples of one fragment invocation together, so a (pixel, pass) stream is the finest
//     unit that can run independently.  Items are dealt from a global queue in 8x8-pixel-tile
//     order, one wave-level atomicAdd per reservation.
//   * one lane owns one item at a time and keeps its whole path state in VGPRs; when its path
//     ends it starts its own next sample, when its item ends it pulls the next item — so every
//     lane of the wave enters the sphere loop with a live ray (wave-level culling of finished
//     paths by regeneration instead of idling), and a wave leaves only when the queue is dry.
//   * the sphere list's geometry (cx,cy,cz,r^2: 16 B) is staged into LDS once per workgroup and
//     the intersection loop walks it with wave-uniform ds_read_b128 broadcasts; shading data
//     (32 B/sphere) stays in global memory / L2 and is read once per segment for the closest hit.
//   * each item's radiance sum is written once, as one 16-byte store, into a per-pass slab; a
//     second tiny kernel folds the slabs into the accumulation buffer in pass order, so the
//     fp32 sum is bit-identical however the queue was scheduled.
//
// One wave step = the phases below, each in its own header with its part of the exactness argument:
//
//   pt_refill.hpp     work queue, item decode (shader.vert:8, shader.frag:354-357, :410), camera ray (:342-351, :365-370)
//   pt_list.hpp       hit_world over the LIST (:175-196): scan + exact phase, tail mode, the literal loop
//   pt_bvh_walk.hpp   hit_world through the hierarchy of pt_bvh.hpp  (which spheres are looked at)
//   pt_grid_walk.hpp  hit_world through the uniform grid of pt_grid.hpp (which spheres are looked at)
//   pt_shade.hpp      miss / hit record / scatter / depth bookkeeping (:289-294, :136-143, :166-171, :210-286, :300, :338)
//   pt_scene.hpp      path state, scene accessors (LDS / scalar cache / global), parking, tallies
//   pt_arith.hpp      PT-SPEC arithmetic: hash, sin/cos, cbrt, unscaled sqrt / division
//
// ARITHMETIC: PT-SPEC — the same contract the CPU oracle states independently in
// oracle/pt_oracle.c.  Compiled with -ffp-contract=off; every fused multiply-add is an explicit
// __builtin_fmaf.
#include "pt_trace_body.hpp"

// blockDim.x is a multiple of 64 (256 normally, 1024 when the staged list is large and only one
// workgroup fits per CU); dynamic LDS = PT_LDS_ENTRIES(n_spheres) * 16 bytes.
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_LIST_LDS) void pt_trace_kernel(const PtKernelArgs A) {
  pt_trace_body<true, true>(A);
}

// the scalar-load walk (PT_GEOM_SCALAR) with the LDS copy kept for the per-lane gathers
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_LIST) void pt_trace_kernel_scalar(const PtKernelArgs A) {
  pt_trace_body<false, true>(A);
}

// lists beyond the LDS (10 232 < n <= 65 528): scalar-load walk, gathers from global memory
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_LIST) void pt_trace_kernel_scalar_nolds(const PtKernelArgs A) {
  pt_trace_body<false, false>(A);
}

// The walk kernels are latency-bound, not issue-bound: for scenes small enough that LDS
// leaves room for them, six waves per SIMD (80 VGPRs) beat five with no spills (config 2: -5 %).
// The kernels for larger scenes are held to four waves by their LDS footprint and keep their
// registers.
// (PT_BUILT_FOR / PT_WAVES_*: pt_kernel_args.h — what a kernel is built for is what the host launches)
// the hierarchy walk (PT_GEOM_BVH): nodes + slots staged in LDS (dynamic LDS =
// PT_BVH_LDS_BYTES32(n_nodes, n_slots) + parking), or read from global memory / L2 when they do not fit
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_bvh(const PtKernelArgs A) {
  pt_trace_body<false, false, 1>(A);
}
// nodes staged (dynamic LDS = (n_nodes + 1) * 16 bytes + parking), slots read from global memory
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_bvh_nodes(const PtKernelArgs A) {
  pt_trace_body<false, false, 2>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_bvh_gmem(const PtKernelArgs A) {
  pt_trace_body<false, false, 3>(A);
}
// the grid walk (PT_GEOM_GRID): cells + entries staged in LDS, cells only, or nothing
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_grid(const PtKernelArgs A) {
  pt_trace_body<false, false, 4>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_grid_cells(const PtKernelArgs A) {
  pt_trace_body<false, false, 5>(A);
}
extern "C" __global__ __launch_bounds__(1024) PT_BUILT_FOR(PT_WAVES_WALK) void pt_trace_kernel_grid_gmem(const PtKernelArgs A) {
  pt_trace_body<false, false, 6>(A);
}
// --------------------------------------------------------------------------------------------
// Work-queue order for the NEXT launch: tiles sorted by the segment count of their HEAVIEST
// item in the previous launch (pass 0), largest first.  A short launch cannot end before its
// longest (pixel, pass) stream has run its serial course, so those streams must start first;
// keyed on the tile's maximum instead of its sum, a two-pass launch is 7 % shorter (the sum
// lets a tile with one very long pixel among cheap ones start late).  One 1024-thread workgroup: max ->
// 1024-bucket histogram in LDS -> scan -> scatter; then the costs are cleared.  The order only
// affects scheduling, never results (each item writes its own slab slot).
// --------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(1024) void pt_tile_order_kernel(uint32_t* cost, uint32_t* order,
                                                                        uint32_t n_tiles) {
  __shared__ uint32_t s_hist[1024];
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_max;
  const uint32_t t = threadIdx.x;
  if (t == 0) s_max = 0;
  s_hist[t] = 0;
  __syncthreads();
  uint32_t m = 0;
  for (uint32_t i = t; i < n_tiles; i += 1024) m = cost[i] > m ? cost[i] : m;
  atomicMax(&s_max, m);
  __syncthreads();
  const uint32_t mx = s_max;
  if (mx == 0) { // no feedback yet: identity order
    for (uint32_t i = t; i < n_tiles; i += 1024) order[i] = i;
    return;
  }
  const float scale = 1023.0f / (float)mx;
  for (uint32_t i = t; i < n_tiles; i += 1024) {
    uint32_t b = 1023u - (uint32_t)((float)cost[i] * scale); // heavy -> low bucket
    atomicAdd(&s_hist[b > 1023u ? 0u : b], 1u);
  }
  __syncthreads();
  // inclusive scan (Hillis-Steele), then shift to exclusive
  s_scan[t] = s_hist[t];
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t v = t >= off ? s_scan[t - off] : 0u;
    __syncthreads();
    s_scan[t] += v;
    __syncthreads();
  }
  s_hist[t] = s_scan[t] - s_hist[t]; // exclusive start of bucket t
  __syncthreads();
  for (uint32_t i = t; i < n_tiles; i += 1024) {
    uint32_t b = 1023u - (uint32_t)((float)cost[i] * scale);
    uint32_t pos = atomicAdd(&s_hist[b > 1023u ? 0u : b], 1u);
    order[pos] = i;
  }
  __syncthreads();
  for (uint32_t i = t; i < n_tiles; i += 1024) cost[i] = 0;
}

// --------------------------------------------------------------------------------------------
// accum[i] += slab[0][i] + slab[1][i] + ... in pass order (sequential fp32 adds, like n_passes
// separate pt_render calls would perform them).  16 B per lane, coalesced.
// --------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void pt_accumulate_kernel(float4* accum,
                                                                       const float4* slab,
                                                                       uint32_t n_pix,
                                                                       uint32_t n_passes) {
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    float4 acc = accum[i];
    for (uint32_t p = 0; p < n_passes; p++) {
      float4 s = slab[(size_t)p * n_pix + i];
      acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
    }
    accum[i] = acc;
  }
}

__device__ __forceinline__ uint32_t unorm8(float v) {
  if (!(v > 0.0f)) return 0u;
  if (v >= 1.0f) return 255u;
  return (uint32_t)(v * 255.0f + 0.5f);
}

// read-out, static/shader.frag:376-380 on the accumulated sum.  The divisor is the pixel's own
// sample count: accum.w carries the sum of float(spp) over the passes folded so far (an exact
// integer below 2^24), so the scale is the same fp32 value as 1/float(total spp) and it stays
// right when a captured launch is replayed by a hipGraph behind the host's back.  A pixel that
// has received nothing reads as 0.
__device__ __forceinline__ float pixel_scale(float w) { return w > 0.0f ? 1.0f / w : 0.0f; }

// One texel of the shader's render() (static/shader.frag:387-404): the frame's colour sqrt(sum * scale), blended into the
// previous frame's RGBA8 texel `pv` with the running-mean rule when averaging — as the statements read
//     px = sqrt(v.xyz * scale);  pa = float(pv.a) / 255;  pr = float(pv.c) / 255;
//     merged = (px * last_frame_weight + pr * rc) / (rc + last_frame_weight);  o.c = unorm8(merged)
// with the same correctly rounded results from cheaper sequences (pt_arith.hpp: sqrt_core and div_core ARE the compiler's
// expansions of sqrtf and `/` without their range scaling, valid where the scaling is the identity):
//   * sqrt_core for an operand that is 0 or in [2^-96, +inf] (it returns 0 for 0, inf for inf);
//   * byte / 255 through div_core with ONE Newton reciprocal of 255 per thread: 255 is a normal denominator, the numerator
//     0 or >= 1 (div_core returns +0 for +0);
//   * merged / total through div_core with one reciprocal per frame when total is in [2^-20, 2^20) and the numerator is 0
//     or in [2^-103, 2^76);
//   * `pa == 0.0` is `pv.a == 0` (a byte over 255 is zero for the zero byte only): no division at all.
// A lane whose operands lie outside those ranges (a colour below 2^-96, a NaN, a weight of 1e-30 ...) takes the plain
// operators, wave by wave and practically never.  Per texel 12 + 3 x 14 + 12 + 3 x 12 + 3 x 12 vector instructions of
// division and square root become 12 + 3 x 9 + 0 + 3 x 5 + 3 x 5 (+ 3 per frame, + 3 per thread).
struct BlendRule {
  float rc, lfw, total, y_total, y255;
  bool averaging;  // should_average && render_count > 1 (wave-uniform)
  bool total_ok;   // total is a denominator for div_core (wave-uniform)
};
__device__ __forceinline__ BlendRule blend_rule(int render_count, int should_average, float last_frame_weight) {
  BlendRule B;
  B.rc = (float)render_count;
  B.lfw = last_frame_weight;
  B.total = B.rc + last_frame_weight;
  B.total_ok = ptk::div_den_ok(B.total);
  B.y_total = ptk::rcp_newton(B.total_ok ? B.total : 1.0f);
  B.y255 = ptk::rcp_newton(255.0f);
  B.averaging = should_average && render_count > 1;
  return B;
}
__device__ __forceinline__ uint32_t blend_texel(const float4 v, const uint32_t pv, const BlendRule& B) {
  using namespace ptk;
  const float scale = pixel_scale(v.w);
  const float x0 = v.x * scale, x1 = v.y * scale, x2 = v.z * scale;
  float px[3] = {sqrt_core(x0), sqrt_core(x1), sqrt_core(x2)};
  // operands sqrt_core does not cover: 0 < x < 2^-96, negative, NaN  (bit patterns: not 0 and not in [2^-96, +inf])
  const uint32_t lo = f2u(0x1p-96f), span = 0x7f800000u - f2u(0x1p-96f);
  bool odd = (f2u(x0) != 0u && f2u(x0) - lo > span) || (f2u(x1) != 0u && f2u(x1) - lo > span) || (f2u(x2) != 0u && f2u(x2) - lo > span);
  uint32_t o = 255u << 24;
  const bool merge = B.averaging && (pv >> 24) != 0u;
  float m[3] = {px[0], px[1], px[2]};
  if (B.averaging) {  // wave-uniform
    const float n_lo = 0x1p-103f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float pr = div_core((float)((pv >> (8 * c)) & 255u), 255.0f, B.y255);
      const float num = fma_(px[c], B.lfw, pr * B.rc);
      const float q = div_core(num, B.total, B.y_total);
      odd = odd || (merge && !(B.total_ok && (f2u(num) == 0u || f2u(num) - f2u(n_lo) < f2u(0x1p76f) - f2u(n_lo))));
      m[c] = merge ? q : px[c];
    }
  }
  if (__builtin_expect(pt_ballot(odd) != 0ull, 0)) {  // (rare) the statements as they read
    if (odd) {
      m[0] = px[0] = __builtin_sqrtf(x0); m[1] = px[1] = __builtin_sqrtf(x1); m[2] = px[2] = __builtin_sqrtf(x2);
      if (merge) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float pr = (float)((pv >> (8 * c)) & 255u) / 255.0f;
          m[c] = fma_(px[c], B.lfw, pr * B.rc) / B.total;
        }
      }
    }
  }
  o |= unorm8(m[0]) | (unorm8(m[1]) << 8) | (unorm8(m[2]) << 16);
  return o;
}

extern "C" __global__ __launch_bounds__(256) void pt_resolve_kernel(const float4* accum, float4* out,
                                                                    uint32_t n_pix, int gamma) {
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    float4 v = accum[i];
    const float scale = pixel_scale(v.w);
    float r = v.x * scale, g = v.y * scale, b = v.z * scale;
    if (gamma) { r = __builtin_sqrtf(r); g = __builtin_sqrtf(g); b = __builtin_sqrtf(b); }
    out[i] = make_float4(r, g, b, 1.0f);
  }
}

extern "C" __global__ __launch_bounds__(256) void pt_resolve_rgba8_kernel(const float4* accum,
                                                                          uint32_t* out, uint32_t n_pix,
                                                                          int gamma) {
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    float4 v = accum[i];
    const float scale = pixel_scale(v.w);
    float r = v.x * scale, g = v.y * scale, b = v.z * scale;
    if (gamma) { r = __builtin_sqrtf(r); g = __builtin_sqrtf(g); b = __builtin_sqrtf(b); }
    out[i] = unorm8(r) | (unorm8(g) << 8) | (unorm8(b) << 16) | (255u << 24);
  }
}

// temporal running mean of the reference, static/shader.frag:387-404 (RGBA8 ping-pong textures)
extern "C" __global__ __launch_bounds__(256) void pt_blend_rgba8_kernel(
    const float4* accum, const uint32_t* prev, uint32_t* out, uint32_t n_pix,
    int render_count, int should_average, float last_frame_weight) {
  uint32_t stride = gridDim.x * blockDim.x;
  const BlendRule B = blend_rule(render_count, should_average, last_frame_weight);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    const uint32_t o = blend_texel(accum[i], prev[i], B);
    out[i] = o;
  }
}

// --------------------------------------------------------------------------------------------
// The reference's FRAME on device-resident textures: webgl::render (src/webgl.rs:180-205) +
// update_render_globals (src/state.rs:443-450) with the per-frame state on the device, so that a
// frame can be replayed from a hipGraph without the host in the loop (pt_render_frames).
// `slab` holds the frame's one pass ({sum r, g, b, spp} per pixel, straight from the trace kernel: no
// accumulation buffer in between); ctr[0] = k, the number of frames drawn since the series began:
//     render_count = min(render_count0 + k, max_render_count)     src/state.rs:449
//     even_odd     = even_odd0 + k                                src/state.rs:448
//     previous frame = texture[(even_odd + 1) % 2]                src/webgl.rs:186-190
//     draw to the canvas; if should_average also to texture[even_odd % 2]   :193-204
// The blend itself is static/shader.frag:387-404, operation for operation as pt_blend_rgba8_kernel.
// --------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void pt_frame_blend_kernel(
    const float4* slab, uint32_t* tex0, uint32_t* tex1, uint32_t* canvas, uint32_t n_pix, const uint32_t* ctr, uint32_t k_off,
    int render_count0, uint32_t even_odd0, int max_render_count, int should_average, float last_frame_weight) {
  const uint32_t k = ctr[0] + k_off;  // (k_off: this frame's place among the frames one launch has traced, pt_render_frames)
  const long long rc_ll = (long long)render_count0 + (long long)k;
  const int render_count = rc_ll < (long long)max_render_count ? (int)rc_ll : max_render_count;
  const uint32_t even_odd = even_odd0 + k;
  const uint32_t* prev = ((even_odd + 1u) & 1u) ? tex1 : tex0;
  uint32_t* out_tex = (even_odd & 1u) ? tex1 : tex0;
  uint32_t stride = gridDim.x * blockDim.x;
  const BlendRule B = blend_rule(render_count, should_average, last_frame_weight);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    const uint32_t o = blend_texel(slab[i], prev[i], B);
    canvas[i] = o;
    if (should_average) out_tex[i] = o;
  }
}

// The blends of a GROUP of frames in one pass over the pixels (pt_render_frames: the group's frames are the passes
// of one trace launch, slab f = frame k + f).  A pixel's chain of blends touches no other pixel, so each thread runs
// its pixel's n_frames blends one after the other — pt_frame_blend_kernel's rule, operation for operation and
// quantised to RGBA8 after every frame exactly as the separate launches would leave it in the texture — and keeps
// the previous frame's texel in a register instead of writing it and reading it back: with averaging on, frame
// f reads the texture frame f - 1 wrote (the ping-pong pair has period two), so only the group's last two frames
// reach memory, and the canvas shows the last.  Without averaging no frame writes a texture and every frame
// blends against the same stored one.
extern "C" __global__ __launch_bounds__(256) void pt_frames_blend_kernel(
    const float4* slab, uint32_t n_frames, uint32_t* tex0, uint32_t* tex1, uint32_t* canvas, uint32_t n_pix, const uint32_t* ctr,
    int render_count0, uint32_t even_odd0, int max_render_count, int should_average, float last_frame_weight) {
  const uint32_t k0 = ctr[0];
  uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    uint32_t pv = 0u, o = 0u;
    // one frame of the chain (its slab value already in a register)
    auto frame = [&](uint32_t f, const float4 v) {
      const uint32_t k = k0 + f;
      const long long rc_ll = (long long)render_count0 + (long long)k;
      const int render_count = rc_ll < (long long)max_render_count ? (int)rc_ll : max_render_count;
      const BlendRule B = blend_rule(render_count, should_average, last_frame_weight);
      const uint32_t even_odd = even_odd0 + k;
      // the previous frame's texel: from memory for the group's first frame (and always when nothing is written
      // back), from the register afterwards
      if (f == 0u || !should_average) pv = (((even_odd + 1u) & 1u) ? tex1 : tex0)[i];
      else pv = o;
      o = blend_texel(v, pv, B);
      // what reaches memory: the last two frames' textures (earlier ones are overwritten by them), the last canvas
      if (should_average && f + 2u >= n_frames) ((even_odd & 1u) ? tex1 : tex0)[i] = o;
    };
    // A pixel's blends are a chain, its slab loads are not: EIGHT frames' values are requested before the first of them is
    // blended (the chain alone — one load, then ~100 instructions, 64 times over — left the kernel waiting on memory at
    // 3.3 TB/s with every wave slot of the chip taken; 0.245 -> 0.220 ms per group of 64 frames; one pixel per thread — 3 510
    // workgroups instead of 2 048 — 0.232: profiles/r05_ab_runs.txt)
    uint32_t f = 0;
    for (; f + 8u <= n_frames; f += 8u) {
      float4 v[8];
#pragma unroll
      for (uint32_t j = 0; j < 8u; j++) v[j] = slab[(size_t)(f + j) * n_pix + i];
#pragma unroll
      for (uint32_t j = 0; j < 8u; j++) frame(f + j, v[j]);
    }
    for (; f < n_frames; f++) frame(f, slab[(size_t)f * n_pix + i]);
    canvas[i] = o;
  }
}

// end of a replay of n frames: the next one starts at frame k + n, and its work queue at item 0
extern "C" __global__ void pt_frame_advance_kernel(uint32_t* ctr, unsigned long long* counters, uint32_t n) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctr[0] += n;
    counters[PT_CTR_HEAD] = 0ull;
  }
  // (the grouped queue's heads: launched with PT_QUEUE_GROUPS_MAX threads)
  if (blockIdx.x == 0 && threadIdx.x < PT_QUEUE_GROUPS_MAX) counters[PT_CTR_GROUP_HEADS + 8u * threadIdx.x] = 0ull;
}

// --------------------------------------------------------------------------------------------
// pt_probe: evaluate single PT-SPEC functions on the device (parity tests of SURVEY §8a rows).
// --------------------------------------------------------------------------------------------
extern "C" __global__ void pt_probe_kernel(int kind, const float* in, float* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (kind) {
    case PT_PROBE_HASH: {
      float seed = in[i];
      float* o = out + 9 * (size_t)i;
      float h1 = hash1(seed);
      o[0] = seed; o[1] = h1;
      float a2, b2;
      hash2(seed, a2, b2);
      o[2] = seed; o[3] = a2; o[4] = b2;
      float a3, b3, c3;
      hash3(seed, a3, b3, c3);
      o[5] = seed; o[6] = a3; o[7] = b3; o[8] = c3;
      break;
    }
    case PT_PROBE_SINCOS: {
      float s, c;
      sincos2pi(in[i], s, c);
      out[2 * (size_t)i] = s; out[2 * (size_t)i + 1] = c;
      break;
    }
    case PT_PROBE_CBRT: out[i] = cbrt_(in[i]); break;
    case PT_PROBE_UNIT_SPHERE: {
      float seed = in[i];
      V3 r = random_in_unit_sphere(seed);
      float* o = out + 4 * (size_t)i;
      o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = seed;
      break;
    }
    case PT_PROBE_DIVSQRT: {
      float x = in[2 * (size_t)i], y = in[2 * (size_t)i + 1];
      float* o = out + 3 * (size_t)i;
      o[0] = x / y; o[1] = __builtin_sqrtf(__builtin_fabsf(x)); o[2] = fma_(x, y, x);
      break;
    }
    case PT_PROBE_FAST_ARITH: { // the unscaled sqrt / division forms beside the plain operators
      const float x = in[3 * (size_t)i], y = in[3 * (size_t)i + 1], z = in[3 * (size_t)i + 2];
      float* o = out + 10 * (size_t)i;
      o[0] = x / y;
      o[1] = div_core(x, y, rcp_newton(y));
      o[2] = __builtin_sqrtf(x);
      o[3] = sqrt_core(x);
      o[4] = sqrt_rn(x);
      // hit_root(half_b = x, disc = y, a = z) and static/shader.frag:156-161 written out
      o[5] = hit_root(x, y, z, rcp_newton(z), hit_root_guard(z));
      const float sqrtd = __builtin_sqrtf(y);
      float v = (-x - sqrtd) / z;
      if (v < PT_MIN_T) v = (-x + sqrtd) / z;
      o[6] = v;
      o[7] = div_den_ok(y) ? 1.0f : 0.0f;
      o[8] = inv_sqrt_rn(x);
      o[9] = 1.0f / __builtin_sqrtf(x);
      break;
    }
    case PT_PROBE_BASE_HASH: {
      uint32_t h = base_hash(f2u(in[2 * (size_t)i]), f2u(in[2 * (size_t)i + 1]));
      out[i] = u2f(h);
      break;
    }
    default: break;
  }
}
