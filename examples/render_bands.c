/*
 * examples/render_bands.c — multi-GPU rendering through the C ABI alone, from ONE host process, the way the
 * reference's own host (Rust, src/lib.rs) would drive it: one pt_ctx per GPU, each rendering its interleaved row
 * bands of the same frame (PtParams.band_*; a pixel's random stream depends only on its position and u_time,
 * static/shader.frag:354-357, so any partition reproduces the single-GPU frame bit for bit), then ONE RCCL
 * all-gather of the per-rank fp32 radiance buffers over xGMI (ncclAllGather on equal-sized padded buffers) and the
 * de-interleave into image order with pt_band_row.  No Python, no torch: gcc + libptrace + libamdhip64 + librccl.
 *
 *   make -C examples render_bands
 *   examples/render_bands out.f32 [ranks [band_rows [scene.bin]]]
 *
 * ranks <= GPUs in the node: rank i runs on device i and the gather is RCCL (ncclCommInitAll: one communicator per
 * device in this process).  ranks > GPUs (a one-GPU box): REHEARSAL — the ranks share the devices round-robin and the
 * gather is done with device-to-device copies instead (RCCL refuses two ranks on one device); everything else —
 * partition, padding, de-interleave — is the same code.  The program says which of the two it did.
 *
 * out.f32 receives the gathered frame: height x width x {sum r, sum g, sum b, spp} fp32, row 0 = bottom row, i.e. the
 * accumulation buffer a single context would hold (tests/test_gpu_parity.py compares it bit for bit).
 *
 * scene.bin (written by the test / any host): "PTSC", u32 n_spheres, u32 sizeof(PtSphere), u32 sizeof(PtParams),
 * u32 n_passes, PtParams, PtSphere[n].  Without it: the reference's State::default, 640 x 352, 4 passes of 4 spp.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ptrace.h"

#define MAX_RANKS 64

#define PT(call, ctx)                                                                      \
  do {                                                                                     \
    int rc_ = (call);                                                                      \
    if (rc_ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, pt_last_error(ctx)); return 1; } \
  } while (0)
#define HIP(call)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_)); return 1; } \
  } while (0)
#define NCCL(call)                                                                         \
  do {                                                                                     \
    ncclResult_t r_ = (call);                                                              \
    if (r_ != ncclSuccess) { fprintf(stderr, "%s failed: %s\n", #call, ncclGetErrorString(r_)); return 1; } \
  } while (0)

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char** argv) {
  const char* out = argc > 1 ? argv[1] : "bands.f32";
  int ranks = argc > 2 ? atoi(argv[2]) : 0;
  uint32_t band_rows = argc > 3 ? (uint32_t)atoi(argv[3]) : 4u;
  const char* scene_path = argc > 4 ? argv[4] : NULL;
  const int n_dev = pt_device_count();
  if (n_dev < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
  if (ranks < 1) ranks = n_dev;
  if (ranks > MAX_RANKS || band_rows < 1) { fprintf(stderr, "ranks <= %d, band_rows >= 1\n", MAX_RANKS); return 1; }
  const int use_rccl = ranks <= n_dev;

  /* ---- scene + uniforms (the same bytes for every rank) ---- */
  PtParams p;
  PtSphere* spheres = NULL;
  uint32_t n_spheres = 0, n_passes = 4;
  memset(&p, 0, sizeof p);
  if (scene_path) {
    FILE* f = fopen(scene_path, "rb");
    uint32_t hdr[5];
    if (!f || fread(hdr, 4, 5, f) != 5 || memcmp(hdr, "PTSC", 4) != 0 || hdr[2] != sizeof(PtSphere) || hdr[3] != sizeof(PtParams)) {
      fprintf(stderr, "%s: not a scene file of this ABI (PtSphere %zu B, PtParams %zu B)\n", scene_path, sizeof(PtSphere), sizeof(PtParams));
      return 1;
    }
    n_spheres = hdr[1]; n_passes = hdr[4];
    spheres = (PtSphere*)malloc((size_t)n_spheres * sizeof(PtSphere));
    if (fread(&p, sizeof p, 1, f) != 1 || fread(spheres, sizeof(PtSphere), n_spheres, f) != n_spheres) { fprintf(stderr, "%s: short file\n", scene_path); return 1; }
    fclose(f);
  } else {
    pt_state* st = NULL;
    if (pt_state_create(&st, 640, 352) != PT_OK) return 1;
    pt_state_set_quality(st, 4, 8);
    pt_state_set_flags(st, 0, 1, 1.0f);
    spheres = (PtSphere*)malloc(16 * sizeof(PtSphere));
    n_spheres = (uint32_t)pt_state_spheres(st, spheres, 16);   /* webgl::set_geometry narrowing */
    pt_state_to_params(st, 0.0, &p);                           /* Uniforms::run_setters */
    pt_state_destroy(st);
  }
  const uint32_t W = p.width, H = p.height;

  /* ---- one context per rank ---- */
  pt_ctx* ctx[MAX_RANKS];
  hipStream_t stream[MAX_RANKS];
  float* send[MAX_RANKS];
  float* recv[MAX_RANKS];
  int dev_of[MAX_RANKS];
  uint32_t rows_of[MAX_RANKS], pad_rows = 0;
  for (int r = 0; r < ranks; r++) {
    rows_of[r] = pt_local_rows(H, band_rows, (uint32_t)r, (uint32_t)ranks);
    if (rows_of[r] > pad_rows) pad_rows = rows_of[r];
  }
  const size_t row_floats = (size_t)W * 4, slot_floats = (size_t)pad_rows * row_floats;
  for (int r = 0; r < ranks; r++) {
    dev_of[r] = r % n_dev;
    HIP(hipSetDevice(dev_of[r]));
    HIP(hipStreamCreateWithFlags(&stream[r], hipStreamNonBlocking));
    ctx[r] = NULL;
    if (pt_create(&ctx[r], dev_of[r], W, H) != PT_OK) { fprintf(stderr, "pt_create(device %d): %s\n", dev_of[r], pt_last_error(NULL)); return 1; }
    PT(pt_set_stream(ctx[r], stream[r]), ctx[r]);
    PtParams q = p;
    q.band_rows = band_rows; q.band_index = (uint32_t)r; q.band_count = (uint32_t)ranks;
    PT(pt_set_spheres(ctx[r], spheres, n_spheres), ctx[r]);    /* every rank uploads the same scene: no broadcast needed */
    PT(pt_set_params(ctx[r], &q), ctx[r]);
    PT(pt_reserve_passes(ctx[r], n_passes), ctx[r]);
    PT(pt_tune(ctx[r], n_passes < 8u ? n_passes : 8u), ctx[r]);  /* set-up: fits the grid to the camera, settles PT_GEOM_AUTO (synchronous; the same choice on every rank: same scene, same camera) */
    HIP(hipMalloc((void**)&send[r], slot_floats * sizeof(float)));
    HIP(hipMemsetAsync(send[r], 0, slot_floats * sizeof(float), stream[r]));   /* rows beyond rows_of[r] stay 0 */
    HIP(hipMalloc((void**)&recv[r], (size_t)ranks * slot_floats * sizeof(float)));
  }
  ncclComm_t comm[MAX_RANKS];
  if (use_rccl) NCCL(ncclCommInitAll(comm, ranks, dev_of));
  for (int r = 0; r < ranks; r++) { HIP(hipSetDevice(dev_of[r])); HIP(hipStreamSynchronize(stream[r])); }

  /* ---- the frame: every rank renders its rows (no collective while rendering), then ONE gather ---- */
  const double t0 = now_s();
  for (int r = 0; r < ranks; r++) {
    HIP(hipSetDevice(dev_of[r]));
    PT(pt_render_passes(ctx[r], n_passes), ctx[r]);                             /* asynchronous on stream[r] */
    void* acc = NULL; size_t bytes = 0;
    PT(pt_accum_ptr(ctx[r], &acc, &bytes), ctx[r]);
    if (rows_of[r]) HIP(hipMemcpyAsync(send[r], acc, (size_t)rows_of[r] * row_floats * sizeof(float), hipMemcpyDeviceToDevice, stream[r]));
  }
  if (use_rccl) {
    NCCL(ncclGroupStart());
    for (int r = 0; r < ranks; r++) NCCL(ncclAllGather(send[r], recv[r], slot_floats, ncclFloat, comm[r], stream[r]));
    NCCL(ncclGroupEnd());
  } else {  /* rehearsal on fewer devices than ranks: the same data movement with copies */
    for (int r = 0; r < ranks; r++) { HIP(hipSetDevice(dev_of[r])); HIP(hipStreamSynchronize(stream[r])); }
    for (int r = 0; r < ranks; r++)
      for (int s = 0; s < ranks; s++)
        HIP(hipMemcpyAsync(recv[r] + (size_t)s * slot_floats, send[s], slot_floats * sizeof(float), hipMemcpyDeviceToDevice, stream[r]));
  }
  for (int r = 0; r < ranks; r++) { HIP(hipSetDevice(dev_of[r])); HIP(hipStreamSynchronize(stream[r])); }
  const double t1 = now_s();

  /* ---- rank 0's gathered buffer back into image order: row l of rank r is image row pt_band_row(band_rows, r, ranks, l) ---- */
  float* host = (float*)malloc((size_t)ranks * slot_floats * sizeof(float));
  float* full = (float*)malloc((size_t)H * row_floats * sizeof(float));
  float* other = ranks > 1 ? (float*)malloc((size_t)ranks * slot_floats * sizeof(float)) : NULL;
  HIP(hipSetDevice(dev_of[0]));
  HIP(hipMemcpy(host, recv[0], (size_t)ranks * slot_floats * sizeof(float), hipMemcpyDeviceToHost));
  for (int r = 0; r < ranks; r++)
    for (uint32_t l = 0; l < rows_of[r]; l++)
      memcpy(full + (size_t)pt_band_row(band_rows, (uint32_t)r, (uint32_t)ranks, l) * row_floats,
             host + (size_t)r * slot_floats + (size_t)l * row_floats, row_floats * sizeof(float));
  int same = 1;  /* an all-gather leaves the same bytes on every rank */
  for (int r = 1; r < ranks; r++) {
    HIP(hipSetDevice(dev_of[r]));
    HIP(hipMemcpy(other, recv[r], (size_t)ranks * slot_floats * sizeof(float), hipMemcpyDeviceToHost));
    if (memcmp(other, host, (size_t)ranks * slot_floats * sizeof(float)) != 0) same = 0;
  }
  FILE* f = fopen(out, "wb");
  if (!f || fwrite(full, sizeof(float), (size_t)H * row_floats, f) != (size_t)H * row_floats) { perror(out); return 1; }
  fclose(f);

  unsigned long long segments = 0;
  double kernel_ms_max = 0.0;
  for (int r = 0; r < ranks; r++) {
    PtStats s;
    PT(pt_get_stats(ctx[r], &s), ctx[r]);
    segments += s.segments;
    if (s.render_kernel_ms > kernel_ms_max) kernel_ms_max = s.render_kernel_ms;
  }
  printf("%s: %ux%u, %u spheres, %u passes of %d spp, %d rank(s) on %d device(s), %u-row bands, gather: %s; %llu segments, "
         "render + gather %.3f ms (longest rank's kernel %.3f ms), every rank holds the same gathered frame: %s\n",
         out, W, H, n_spheres, n_passes, p.samples_per_pixel, ranks, n_dev < ranks ? n_dev : ranks, band_rows,
         use_rccl ? "ncclAllGather (RCCL)" : "device-to-device copies (REHEARSAL: more ranks than devices)",
         segments, (t1 - t0) * 1e3, kernel_ms_max, same ? "yes" : "NO");
  for (int r = 0; r < ranks; r++) {
    HIP(hipSetDevice(dev_of[r]));
    if (use_rccl) ncclCommDestroy(comm[r]);
    pt_destroy(ctx[r]);
    hipFree(send[r]); hipFree(recv[r]);
    hipStreamDestroy(stream[r]);
  }
  free(host); free(full); free(other); free(spheres);
  return same ? 0 : 2;
}
