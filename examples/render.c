/*
 * examples/render.c — the C ABI on its own (no Python, no torch): the reference's State::default
 * scene through pt_state_* / pt_create / pt_set_spheres / pt_set_params / pt_render_passes /
 * pt_resolve_rgba8, written as a binary PPM.  This is the call sequence of src/lib.rs:30-104
 * (set_geometry once, then run_setters + render per frame) against libptrace.so.
 *
 *   gcc -O2 -Iinclude examples/render.c -o examples/render -Lray_tracer_webgl_amd -lptrace \
 *       -Wl,-rpath,'$ORIGIN/../ray_tracer_webgl_amd'
 *   examples/render out.ppm 640 351 16
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
  const char* out = argc > 1 ? argv[1] : "render.ppm";
  uint32_t w = argc > 2 ? (uint32_t)atoi(argv[2]) : 640, h = argc > 3 ? (uint32_t)atoi(argv[3]) : 351;
  uint32_t frames = argc > 4 ? (uint32_t)atoi(argv[4]) : 16;
  pt_ctx* ctx = NULL;
  pt_state* st = NULL;

  if (pt_state_create(&st, w, h) != PT_OK) { fprintf(stderr, "pt_state_create failed\n"); return 1; }
  if (pt_create(&ctx, 0, w, h) != PT_OK) { fprintf(stderr, "pt_create: %s\n", pt_last_error(NULL)); return 1; }

  PtSphere spheres[16];
  int n = pt_state_spheres(st, spheres, 16);          /* webgl::set_geometry, once */
  CHECK(pt_set_spheres(ctx, spheres, (uint32_t)n));
  CHECK(pt_reserve_passes(ctx, frames));

  PtParams p;
  memset(&p, 0, sizeof p);
  CHECK(pt_state_to_params(st, 0.0, &p));             /* Uniforms::run_setters: paused -> 25 spp */
  p.background_mode = PT_BG_SKY;
  p.band_rows = 8; p.band_index = 0; p.band_count = 1;
  CHECK(pt_set_params(ctx, &p));
  CHECK(pt_render_passes(ctx, frames));               /* `frames` passes, u_time = 0, 1, 2, ... */

  unsigned char* rgba = (unsigned char*)malloc((size_t)w * h * 4);
  CHECK(pt_resolve_rgba8(ctx, rgba, 1));
  PtStats stats;
  CHECK(pt_get_stats(ctx, &stats));

  FILE* f = fopen(out, "wb");
  if (!f) { perror(out); return 1; }
  fprintf(f, "P6\n%u %u\n255\n", w, h);
  for (uint32_t y = 0; y < h; y++) {                  /* row 0 of the buffer is the BOTTOM row */
    const unsigned char* row = rgba + (size_t)(h - 1 - y) * w * 4;
    for (uint32_t x = 0; x < w; x++) fwrite(row + 4 * x, 1, 3, f);
  }
  fclose(f);
  printf("%s: %ux%u, %d spheres, %u spp, %llu segments, %.3f ms of kernel time\n", out, w, h, n,
         stats.total_spp, (unsigned long long)stats.segments, stats.render_kernel_ms);
  free(rgba);
  pt_destroy(ctx);
  pt_state_destroy(st);
  return 0;
}
