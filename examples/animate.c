/*
 * examples/animate.c — the reference's animation loop (src/lib.rs:65-104) from plain C: State::default,
 * unpaused (1 spp per tick, src/state.rs:127), `n` ticks at a constant frame interval, each blended into the
 * RGBA8 ping-pong textures by the shader's render() rule (static/shader.frag:387-404, src/webgl.rs:180-205).
 * The ticks are ONE pt_render_frames call: the uniforms of the first tick go up once, libptrace replays one
 * captured hipGraph n times with u_time / render_count / the texture roles counted on the device.  The
 * host's State follows with update_render_globals, as it would after n rAF callbacks.  (Valid because nothing
 * but the clock changes between these ticks: should_average is on and no movement key is held; otherwise the
 * ticks are issued one by one with pt_render_frame, as ray_tracer_webgl_amd/app.py FrameLoop.frames does.)
 *
 *   make -C examples && examples/animate out.ppm 640 351 320
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptrace.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ < 0) {                                                                   \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, pt_last_error(ctx));       \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

int main(int argc, char** argv) {
  const char* out = argc > 1 ? argv[1] : "animate.ppm";
  uint32_t w = argc > 2 ? (uint32_t)atoi(argv[2]) : 640, h = argc > 3 ? (uint32_t)atoi(argv[3]) : 351;
  uint32_t n = argc > 4 ? (uint32_t)atoi(argv[4]) : 320;
  const double now0 = 3000.0, interval = 16.5; /* ms; both exact in fp32 */
  pt_ctx* ctx = NULL;
  pt_state* st = NULL;

  if (pt_state_create(&st, w, h) != PT_OK) { fprintf(stderr, "pt_state_create failed\n"); return 1; }
  if (pt_create(&ctx, 0, w, h) != PT_OK) { fprintf(stderr, "pt_create: %s\n", pt_last_error(NULL)); return 1; }
  if (pt_state_set_flags(st, /*is_paused*/ 0, /*should_average*/ 1, 1.0f) != PT_OK) return 1;

  PtSphere spheres[16];
  int n_sph = pt_state_spheres(st, spheres, 16);      /* webgl::set_geometry, once */
  CHECK(pt_set_spheres(ctx, spheres, (uint32_t)n_sph));
  CHECK(pt_clear_textures(ctx));                      /* create_texture x2: alpha 0 = "no data" */

  /* the first tick: update_position (no input: nothing moves), update_render_globals, run_setters */
  if (pt_state_update_position(st, now0) != PT_OK || pt_state_update_render_globals(st) != PT_OK) return 1;
  PtStateView v;
  PtParams p;
  memset(&p, 0, sizeof p);
  if (pt_state_get(st, &v) != PT_OK || pt_state_to_params(st, now0, &p) != PT_OK) return 1;
  p.band_rows = 8; p.band_index = 0; p.band_count = 1;
  p.time_step = (float)interval;                      /* tick k renders at u_time = now0 + k * interval */
  p.first_pass = 0;
  CHECK(pt_set_params(ctx, &p));
  /* the camera may have moved since the grid of a large scene was fitted (State::update_position, src/state.rs:411-441):
   * host arithmetic, 0 for State::default's nine spheres (no grid) — INTEGRATION.md §1 */
  if (pt_grid_fit(ctx) == 1) CHECK(pt_refit_grid(ctx, 0));
  CHECK(pt_render_frames(ctx, v.even_odd_count, v.max_render_count, n));  /* webgl::render, n times */
  for (uint32_t k = 1; k < n; k++) pt_state_update_render_globals(st);    /* the host's counters follow */

  unsigned char* rgba = (unsigned char*)malloc((size_t)w * h * 4);
  CHECK(pt_read_canvas(ctx, rgba));                   /* what the canvas shows after the last tick */
  PtStats stats;
  CHECK(pt_get_stats(ctx, &stats));
  pt_state_get(st, &v);

  FILE* f = fopen(out, "wb");
  if (!f) { perror(out); return 1; }
  fprintf(f, "P6\n%u %u\n255\n", w, h);
  for (uint32_t y = 0; y < h; y++) {                  /* row 0 of the buffer is the BOTTOM row */
    const unsigned char* row = rgba + (size_t)(h - 1 - y) * w * 4;
    for (uint32_t x = 0; x < w; x++) fwrite(row + 4 * x, 1, 3, f);
  }
  fclose(f);
  printf("%s: %ux%u, %d spheres, %u frames of %d spp, render_count %u, %llu segments, %.3f ms on the device (%.4f ms per frame)\n",
         out, w, h, n_sph, n, p.samples_per_pixel, v.render_count, (unsigned long long)stats.segments, stats.render_kernel_ms,
         stats.render_kernel_ms / n);
  free(rgba);
  pt_destroy(ctx);
  pt_state_destroy(st);
  return 0;
}
