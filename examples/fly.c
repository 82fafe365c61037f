/*
 * examples/fly.c — the reference's rAF loop WITH A MOVEMENT KEY HELD, on a scene of hundreds of spheres, from plain C:
 * what src/lib.rs:65-104 does every tick — State::update_position (src/state.rs:411-441: the camera moves),
 * update_render_globals, run_setters, webgl::render — with the GPU boundary replaced by libptrace, plus the two lines a
 * host of a LARGE scene adds: the grid the walk kernels use is fitted to the region rays start in, the camera leaves
 * that region as it flies, and the boundary says so (pt_grid_fit: host arithmetic, no synchronisation); pt_refit_grid
 * rebuilds the grid for the class the camera now needs before the tick's frame is traced.
 * Without them every frame is still the same bits, several times slower (PtStats.far_rays counts why).
 *
 *   make -C examples fly
 *   examples/fly out.ppm scene.bin [width height ticks [refit]]        refit 0: never refit (to see the difference)
 *
 * scene.bin: "PTSC", u32 n_spheres, u32 sizeof(PtSphere), u32 sizeof(PtParams), u32 n_passes, PtParams, PtSphere[n]
 * (tools/write_scene_bin.py; only the spheres are used here — the camera is the State's).  The camera starts inside the
 * scene, 's' is held (State moves it backwards along its view direction, MOVEMENT_SPEED * dt * fov per tick), one 1-spp
 * frame per tick is blended into the RGBA8 ping-pong textures on the device; out.ppm is the canvas after the last tick.
 */
#include <math.h>
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
  if (argc < 3) { fprintf(stderr, "usage: fly out.ppm scene.bin [width height ticks [refit]]\n"); return 2; }
  const char* out = argv[1];
  const char* scene_path = argv[2];
  const uint32_t w = argc > 3 ? (uint32_t)atoi(argv[3]) : 320, h = argc > 4 ? (uint32_t)atoi(argv[4]) : 180;
  const uint32_t ticks = argc > 5 ? (uint32_t)atoi(argv[5]) : 30;
  const int refit = argc > 6 ? atoi(argv[6]) : 1;
  const double dt = 11500.0; /* ms per tick (exact in fp32, and so are its multiples): ~12 units of flight per tick */
  pt_ctx* ctx = NULL;
  pt_state* st = NULL;

  /* ---- the scene: f32 records from the file -> the State's f64 spheres (src/glsl.rs:27-40) ---- */
  FILE* f = fopen(scene_path, "rb");
  uint32_t hdr[5];
  PtParams file_params;
  if (!f || fread(hdr, 4, 5, f) != 5 || memcmp(hdr, "PTSC", 4) != 0 || hdr[2] != sizeof(PtSphere) || hdr[3] != sizeof(PtParams) ||
      fread(&file_params, sizeof file_params, 1, f) != 1) {
    fprintf(stderr, "%s: not a scene file of this ABI\n", scene_path);
    return 1;
  }
  const uint32_t n = hdr[1];
  PtSphere* rec = (PtSphere*)malloc((size_t)n * sizeof(PtSphere));
  PtHostSphere* host = (PtHostSphere*)malloc((size_t)n * sizeof(PtHostSphere));
  if (!rec || !host || fread(rec, sizeof(PtSphere), n, f) != n) { fprintf(stderr, "%s: short file\n", scene_path); return 1; }
  fclose(f);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (uint32_t i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) {
      host[i].center[k] = rec[i].center[k];
      host[i].albedo[k] = rec[i].albedo[k];
      if (fabs((double)rec[i].radius) < 100.0) { /* (not the ground) */
        if (rec[i].center[k] < lo[k]) lo[k] = rec[i].center[k];
        if (rec[i].center[k] > hi[k]) hi[k] = rec[i].center[k];
      }
    }
    host[i].radius = rec[i].radius;
    host[i].type = rec[i].type;
    host[i].uuid = (int32_t)i;
    host[i].fuzz = rec[i].fuzz;
    host[i].refraction_index = rec[i].refraction_index;
  }
  if (pt_state_create(&st, w, h) != PT_OK || pt_state_set_spheres(st, host, n) != PT_OK) { fprintf(stderr, "pt_state: failed\n"); return 1; }
  if (pt_create(&ctx, 0, w, h) != PT_OK) { fprintf(stderr, "pt_create: %s\n", pt_last_error(NULL)); return 1; }
  pt_state_set_flags(st, /*is_paused*/ 0, /*should_average*/ 1, 1.0f);
  CHECK(pt_state_spheres(st, rec, n));                /* webgl::set_geometry's narrowing, once */
  CHECK(pt_set_spheres(ctx, rec, n));
  CHECK(pt_clear_textures(ctx));

  /* the camera: a little off the scene's middle, looking back at it (yaw / pitch as State keeps them, src/state.rs:354-358);
   * 's' held — it flies backwards along its view direction, out of the scene */
  const double origin[3] = {0.5 * (lo[0] + hi[0]) + 6.5, 0.5 * (lo[1] + hi[1]) + 2.0, 0.5 * (lo[2] + hi[2]) + 7.25};
  pt_state_set_camera_origin(st, origin);
  pt_state_set_camera_angles(st, -132.0, -11.5);
  pt_state_set_keys(st, 4u);                          /* KeydownMap.s */

  uint32_t refits = 0;
  for (uint32_t k = 0; k < ticks; k++) {
    const double now = dt * (double)(k + 1);
    if (pt_state_update_position(st, dt) != PT_OK) return 1;           /* the camera moves (src/lib.rs:73) */
    if (!pt_state_should_render(st, 0)) continue;                       /* :77-82 */
    pt_state_update_render_globals(st);                                 /* :93 */
    PtStateView v;
    PtParams p;
    memset(&p, 0, sizeof p);
    if (pt_state_get(st, &v) != PT_OK || pt_state_to_params(st, now, &p) != PT_OK) return 1;
    CHECK(pt_set_params(ctx, &p));                                      /* uniforms.run_setters, :96 */
    if (refit && pt_grid_fit(ctx) == 1) {                               /* the camera has left the region the grid serves */
      CHECK(pt_refit_grid(ctx, 0));
      refits++;
    }
    CHECK(pt_render_frame(ctx, v.even_odd_count));                      /* webgl::render */
  }

  unsigned char* rgba = (unsigned char*)malloc((size_t)w * h * 4);
  CHECK(pt_read_canvas(ctx, rgba));
  PtStats stats;
  CHECK(pt_get_stats(ctx, &stats));
  f = fopen(out, "wb");
  if (!f) { perror(out); return 1; }
  fprintf(f, "P6\n%u %u\n255\n", w, h);
  for (uint32_t y = 0; y < h; y++) {                  /* row 0 of the buffer is the BOTTOM row */
    const unsigned char* row = rgba + (size_t)(h - 1 - y) * w * 4;
    for (uint32_t x = 0; x < w; x++) fwrite(row + 4 * x, 1, 3, f);
  }
  fclose(f);
  printf("%s: %ux%u, %u spheres, %u ticks with 's' held, %u refits, grid for %.1f scene radii (the camera needs %.1f, fit flag %u, kernel build %u), "
         "%llu segments, %llu of them on the far path (%.2e)\n",
         out, w, h, n, ticks, refits, stats.grid_near_factor, stats.grid_need_factor, stats.grid_fit_stale, stats.grid_kernel_build,
         (unsigned long long)stats.segments, (unsigned long long)stats.far_rays,
         stats.segments ? (double)stats.far_rays / (double)stats.segments : 0.0);
  free(rgba); free(rec); free(host);
  pt_destroy(ctx);
  pt_state_destroy(st);
  return 0;
}
