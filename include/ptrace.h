/*
 * ptrace.h — C ABI of libptrace.so, the MI355X-native replacement for the GPU boundary of
 * austintheriot/ray-tracer-webgl (the WebGL2 surface used by src/webgl.rs + static/shader.frag).
 *
 * Everything here is plain C: POD structs, plain pointers, sizes, int error codes.  No torch /
 * C++ types cross this boundary, nothing throws or aborts across it.  A Rust `extern "C"` block
 *  binds these symbols unchanged.
 *
 * Reference citations are `path:line` relative to the reference repository root.
 *
 * Threading contract (same as the reference, src/lib.rs:66-69): one host thread drives a
 * pt_ctx at a time; separate contexts (one per GPU) are independent.
 *
 * There is NO CPU backend: pt_create fails with PT_ERR_NO_DEVICE when no HIP device exists.
 */
#ifndef PTRACE_H
#define PTRACE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 5 /* 5: PtStats grew (grid_fit_stale, grid_near_factor, far_rays); pt_refit_grid, pt_create_on_stream;
                            pt_tune refits the grid; the dev header's pt_debug_wave_log writes four words per wave */

/* ---- error codes (returned by every int function; 0 = success) ------------------------------ */
enum {
  PT_OK = 0,
  PT_ERR_INVALID = -1,    /* bad argument (NULL, zero size, spp < 1, max_depth < 1 ...)        */
  PT_ERR_NO_DEVICE = -2,  /* no HIP device / device index out of range                         */
  PT_ERR_HIP = -3,        /* a HIP runtime call failed; pt_last_error() has the text           */
  PT_ERR_NOT_READY = -4,  /* render before pt_set_spheres / pt_set_params                      */
  PT_ERR_CAPACITY = -5,   /* scene or pass count exceeds what was reserved                     */
};

/* ---- material types: static/shader.frag:45-47, src/glsl.rs:10-24 ---------------------------- */
enum {
  PT_DIFFUSE = 0,
  PT_METAL = 1,
  PT_GLASS = 2,
  PT_EMISSIVE = 3, /* build extension (BASELINE config 4): emits `albedo`, absorbs the path     */
};

/* ---- how the intersection loop looks at the sphere list (pt_set_option PT_OPT_GEOMETRY_PATH) --
 * All of them run the same fp32 arithmetic on every sphere that can possibly be hit and give
 * bit-identical images.
 *   LDS    the whole list, staged into LDS once per workgroup, ds_read_b128 broadcasts
 *          (n <= 10 232)
 *   SCALAR the whole list through wave-uniform scalar loads (s_load_dwordx16) via the scalar
 *          cache / L2; sphere data reaches the VALU as SGPR operands (any n up to 65 528)
 *   BVH    a bounding-box hierarchy built by pt_set_spheres decides which spheres a ray looks
 *          at: boxes are inflated by a per-ray margin that covers the rounding of the literal
 *          test, so only spheres that cannot pass it are skipped (regular scenes of >= 16
 *          spheres; a scene without a hierarchy falls back to SCALAR)
 *   GRID   a uniform grid built by pt_set_spheres (cells of about two spheres each, 3D-DDA walk,
 *          entries registered with a margin that covers the rounding of the literal test and of
 *          the walk): a ray looks at the entries of the cells it passes through, in order, and
 *          stops once its closest root lies before the current cell's exit.  The better
 *          structure when most rays start inside the scene (bounce rays): a hierarchy spends two
 *          box tests per level just locating the ray's origin.  Same preconditions as BVH; a
 *          scene without a grid falls back to BVH, then SCALAR
 *   AUTO   (default) after pt_set_spheres the first launch runs cold (unmeasured), the next
 *          ones measure one usable path each, and the fastest (time per camera sample) is
 *          used from then on.  Paths whose outcome is not open are not measured: the list
 *          walks beside a structure on more than 64 spheres, the hierarchy beside an even
 *          grid (no cell with more than 16 entries, at most 8 always-tested spheres)       */
enum {
  PT_GEOM_AUTO = 0,
  PT_GEOM_LDS = 1,
  PT_GEOM_SCALAR = 2,
  PT_GEOM_BVH = 3,
  PT_GEOM_GRID = 4,
  PT_GEOM_SMALL = 5, /* lists of at most 16 spheres (the reference's `uniform Sphere[15] u_sphere_list`,
                        static/shader.frag:103): no LDS traffic in the scan, no candidate queue — four spheres per
                        s_load_dwordx16 reach the VALU as SGPR operands, candidates are finished group by
                        group in list order with the shader's own sequential acceptance.  A longer list
                        falls back to SCALAR.  PT_GEOM_AUTO tries it first on such scenes. */
};
enum {
  PT_OPT_GEOMETRY_PATH = 1,
  PT_OPT_COUNT_WORK = 2,  /* 1: the walk kernels' measuring twins fill PtStats.work (slower; never time them) */
  PT_OPT_CARRY_LANES = 3, /* walk kernels: move on to shading when fewer lanes than this (and less than half
                             of the wave) are still walking; the stragglers continue in the next wave step.
                             The grid walk raises the threshold by 4 per iteration a step's walk has already
                             run (scenes of long walks).  Scheduling only — images do not depend on it.
                             0 = lockstep.  Default 12. */
  PT_OPT_REFILL_MIN = 4,  /* lanes of a busy wave that wait for a new work item before the (wave-wide) item
                             decode runs for them.  Scheduling only.  1 = refill at once.  Default 4. */
  PT_OPT_RUSSIAN_ROULETTE = 5, /* value k > 0: after k bounces a path survives each further bounce with
                             probability q = min(max(throughput), 1) and carries throughput / q (unbiased:
                             every pixel's EXPECTATION is the reference's, depth-exhaustion term included).
                             NOT the reference's estimator sample for sample — static/shader.frag:297-339
                             never ends a path early — so images of this mode match the oracle only
                             statistically; it exists for deep-bounce scenes (BASELINE config 4: mean path
                             length 40 segments).  0 = off (default): bit-exact against the oracle. */
  PT_OPT_GRID_FIT = 6,    /* how pt_tune chooses the grid's margin class.  0 (default): it measures the candidates (see
                             pt_tune).  1: the smallest class that covers the camera, no launches.  Speed only. */
};

/* ---- background modes ----------------------------------------------------------------------- */
enum {
  PT_BG_SKY = 0,   /* static/shader.frag:289-294 white→(0.5,0.7,1.0) gradient                   */
  PT_BG_BLACK = 1, /* build extension for enclosed scenes                                      */
};

/*
 * One sphere, exactly the fields webgl::set_geometry uploads per element of `u_sphere_list`
 * (src/webgl.rs:225-274; struct Sphere static/shader.frag:55-61).  `is_active` is implied by
 * the count passed to pt_set_spheres.  `radius` may be negative (src/state.rs:200,213): the
 * outward normal flips, r*r is unchanged.  48 bytes.
 */
typedef struct PtSphere {
  float center[3];
  float radius;
  int32_t type; /* PT_DIFFUSE / PT_METAL / PT_GLASS / PT_EMISSIVE; anything else absorbs      */
  float albedo[3];
  float fuzz;
  float refraction_index;
  int32_t uuid;
  int32_t _pad;
} PtSphere;

/*
 * The per-frame uniform block: what Uniforms::run_setters uploads (src/webgl.rs:279-593,
 * :629-633; declarations static/shader.frag:79-102), minus the uniforms no code path reads
 * (u_aspect_ratio, u_viewport_*, u_focal_length, u_w) and the debug overlay (:100-102, dead:
 * enable_debugging is 0, src/state.rs:259).
 */
typedef struct PtParams {
  uint32_t width;  /* u_width  */
  uint32_t height; /* u_height */
  float time;      /* u_time: per-pass seed offset (static/shader.frag:356)                    */
  int32_t samples_per_pixel; /* u_samples_per_pixel (per pass)                                 */
  int32_t max_depth;         /* u_max_depth                                                    */
  float camera_origin[3];     /* u_camera_origin     */
  float horizontal[3];        /* u_horizontal        */
  float vertical[3];          /* u_vertical          */
  float lower_left_corner[3]; /* u_lower_left_corner */
  float u[3];                 /* u_u                 */
  float v[3];                 /* u_v                 */
  float lens_radius;          /* u_lens_radius       */
  int32_t render_count;       /* u_render_count      (temporal blend only) */
  int32_t should_average;     /* u_should_average    (temporal blend only) */
  float last_frame_weight;    /* u_last_frame_weight (temporal blend only) */
  /* ---- build extensions ---- */
  int32_t background_mode; /* PT_BG_* */
  /*
   * Row partition for multi-GPU rendering.  This context owns the image rows y (0 = bottom,
   * static/shader.frag:410) with (y / band_rows) % band_count == band_index; its buffers hold
   * only those rows, compacted in increasing y.  band_count <= 1 means "all rows".
   */
  uint32_t band_rows;
  uint32_t band_index;
  uint32_t band_count;
  /*
   * pt_render_passes: pass p of the call renders with
   *     u_time = time + float(first_pass + p) * time_step      (fp32 multiply, then add; step 0 means 1)
   * so a frame rendered as several calls (first_pass = passes done so far) has the bits of one
   * call, whatever the step.  Whole-number steps make consecutive passes REUSE each other's random
   * numbers (the seed advances by .1 per draw, static/shader.frag:22, so pass p + 1 walks the
   * seeds pass p reaches ten draws later): measured +27 % variance of the accumulated frame.
   * The reference's own u_time is performance.now(), which never lines up like that; a step such
   * as PT_TIME_STEP_DECORRELATED keeps the passes independent (tests/test_reference_pins.py).
   */
  float time_step;
  uint32_t first_pass;
} PtParams;
#define PT_TIME_STEP_DECORRELATED 0.3618034f /* 0.1 x (3 + 1/golden ratio): multiples never near a multiple of .1 */

/*
 * Inputs of State::update_pipeline (src/state.rs:319-347) / State::default (:98-125): the
 * camera-derived members of `State`.  All double, like the reference's Vec3 (src/math.rs:17).
 */
typedef struct PtCameraIn {
  uint32_t width, height;
  double camera_origin[3];
  double yaw_degrees;   /* State.yaw   */
  double pitch_degrees; /* State.pitch */
  double vup[3];
  double fov_radians;    /* State.camera_field_of_view */
  double focus_distance; /* State.focus_distance       */
  double aperture;       /* State.aperture; lens_radius = aperture / 2 (src/state.rs:102)      */
} PtCameraIn;

/* Look-at form (Shirley-style), for scenes the yaw/pitch form cannot express conveniently. */
typedef struct PtLookAtIn {
  uint32_t width, height;
  double look_from[3], look_at[3], vup[3];
  double vfov_radians;
  double focus_distance;
  double aperture;
} PtLookAtIn;

typedef struct PtStats {
  uint64_t segments;        /* ray segments (hit_world invocations) since the last reset        */
  uint64_t samples;         /* camera paths started                                             */
  uint64_t sphere_tests;    /* segments * n_spheres                                             */
  double render_kernel_ms;  /* sum of the path-tracing kernel's durations (HIP events)          */
  uint32_t render_launches; /* number of path-tracing kernel launches in that sum               */
  uint32_t total_spp;       /* samples per pixel accumulated                                    */
  uint32_t n_spheres;
  uint32_t local_rows;      /* rows held by this context (row partition)                        */
  uint32_t geometry_path;   /* PT_GEOM_LDS / _SCALAR / _BVH used by the most recent launch          */
  uint32_t geometry_tuned;  /* 1 once PT_GEOM_AUTO has measured the usable paths for this scene     */
  uint32_t bvh_nodes;       /* hierarchy of the current scene: nodes (0 = none), slots (4 per leaf   */
  uint32_t bvh_slots;       /* + the spheres tested for every ray), how many of those, depth        */
  uint32_t bvh_outliers;
  uint32_t bvh_depth;
  uint32_t grid_cells[3];   /* uniform grid of the current scene (0 = none): cells per axis,          */
  uint32_t grid_entries;    /* entries (copies of a sphere, one per cell it is registered in + padding) */
  uint32_t grid_always;     /* spheres tested for every ray                                            */
  uint32_t grid_fit_stale;  /* does the grid still fit the camera of the last pt_set_params?  0: yes (or no grid / not the
                               path in use).  1: NO — the camera stands outside the region whose rays walk the cells, so every
                               primary ray takes the far path (exact, but tested against the WHOLE list): call pt_refit_grid
                               (or pt_tune).  2: looser than needed — a smaller margin class (not below the default, 3) would
                               cover the camera (a few per cent of speed, never a cliff).  The reference moves its camera every tick
                               (src/state.rs:411-441); pt_set_params only sets this flag, it never rebuilds.       */
  /* executed work of the walk kernels since the last reset, filled when PT_OPT_COUNT_WORK is on:
   * [0] walk iterations (node steps / cell steps, wave-level)   [1] lanes active in them (sum)
   * [2] leaf rounds (four literal tests per lane)               [3] lanes active in them
   * [4] exact evaluations (sqrt + division), wave-level         [5] lanes active in them
   * [6] wave steps                                              [7] lanes carried over (sum)   */
  uint64_t work[8];
  float grid_near_factor;   /* d_near / s0 the grid in place was built for (3 from pt_set_spheres; pt_tune / pt_refit_grid
                               choose among 2.5 / 3 / 4 / 5.5 / 8 / 12 / 16); 0 = no grid                              */
  float grid_need_factor;   /* the smallest of those classes that covers the current camera and its lens           */
  uint64_t far_rays;        /* ray segments since the last reset that reached the grid from OUTSIDE its near region and
                               were therefore tested against the whole list (grid walk only; ~0 on a fitted grid)     */
  uint32_t grid_kernel_build; /* which build of the grid kernel the next launch gets: 1 cell records AND entries staged in the LDS
                               (pt_trace_kernel_grid), 2 the cell records staged, the entries gathered from L2 (…_grid_cells),
                               3 nothing staged (…_grid_gmem), 0 no grid.  What fits is staged — except while grid_fit_stale is 1
                               (the builds that gather hand a far ray to the whole wave: a stale view costs 4 x, not 16 x) and
                               where pt_tune timed the gathering build faster.  Scheduling only.                       */
  uint32_t _pad2;
} PtStats;

typedef struct pt_ctx pt_ctx;

/* ---- lifetime: replaces setup_program + create_texture x2 + create_framebuffer x2 ------------
 * (src/webgl.rs:66-80, :82-123, :153-167).  Allocates the fp32 linear accumulation buffer for
 * a w*h image on HIP device `device` and a stream.  Caller frees with pt_destroy. */
int pt_create(pt_ctx** out, int device, uint32_t width, uint32_t height);
/* Same, for a host that brings its own hipStream_t (a torch stream, an engine's render queue): the context runs on it
 * from the start and never creates a stream — hence an HSA queue, ~80-150 ms when it is the process's first — of its own.
 * `hip_stream` as in pt_set_stream (PT_STREAM_LEGACY names the default stream); NULL behaves like pt_create.  A later
 * pt_set_stream(ctx, NULL) on such a context creates the own stream then. */
int pt_create_on_stream(pt_ctx** out, int device, uint32_t width, uint32_t height, void* hip_stream);
int pt_destroy(pt_ctx* ctx);
/* resize path, src/state.rs:364-398: reallocates buffers and clears the accumulation */
int pt_resize(pt_ctx* ctx, uint32_t width, uint32_t height);

/* ---- scene + uniforms ------------------------------------------------------------------------
 * pt_set_spheres replaces webgl::set_geometry (src/webgl.rs:225-274); n is not capped at 15:
 * up to 10 232 spheres are walked from LDS, up to 65 528 from global memory (PT_ERR_CAPACITY beyond).
 * pt_set_params replaces Uniforms::run_setters (src/webgl.rs:629-633). Both copy. */
int pt_set_spheres(pt_ctx* ctx, const PtSphere* spheres, uint32_t n);
int pt_set_params(pt_ctx* ctx, const PtParams* params);

/* ---- rendering: replaces webgl::render -> draw (src/webgl.rs:180-205, :169-178) ---------------
 * pt_render enqueues ONE pass (`samples_per_pixel` samples at seed offset `time`) that adds each
 * pixel's linear radiance sum into the accumulation buffer.  Asynchronous on the context's
 * stream; no host synchronisation, no allocation (after pt_reserve_passes).
 * pt_render_passes enqueues n_passes passes in one launch; pass p uses u_time = time + p. The
 * result is bit-identical to n_passes pt_render calls with those times. */
int pt_render(pt_ctx* ctx);
int pt_render_passes(pt_ctx* ctx, uint32_t n_passes);
/* Pre-sizes the per-pass workspace so pt_render_passes(n <= max_passes) never allocates. */
int pt_reserve_passes(pt_ctx* ctx, uint32_t max_passes);
/* render_count = 0 (src/state.rs:343-346): clears accumulation, spp counter and statistics */
int pt_reset_accum(pt_ctx* ctx);
int pt_synchronize(pt_ctx* ctx);

/* ---- read-out: replaces the canvas read (src/dom.rs:126-143) -----------------------------------
 * Writes local_rows*width RGBA fp32 texels (row 0 = lowest owned row): rgb = accum / total_spp,
 * then sqrt when gamma != 0 (static/shader.frag:376-380); a = 1.  `out` may be a host or a
 * device pointer.  Synchronises the stream.  The divisor is read per pixel from the device-side
 * sample count (the buffer's .a), so it is also right after hipGraph replays of a captured
 * pt_render_passes, which the host-side counters cannot see. */
int pt_resolve(pt_ctx* ctx, float* rgba_out, int gamma);
/* Same, clamped and quantised to RGBA8 like the reference framebuffer (src/webgl.rs:109-119). */
int pt_resolve_rgba8(pt_ctx* ctx, uint8_t* rgba_out, int gamma);
/* Raw accumulation buffer (local_rows*width float4: r,g,b sums, a = spp): device pointer. */
int pt_accum_ptr(pt_ctx* ctx, void** dev_ptr, size_t* bytes);
/* Checkpoint / resume (the reference's accumulation state is its ping-pong textures +
 * render_count, src/state.rs:443-450): copy the accumulation buffer out / back in.  The sample
 * count travels inside the buffer (the .a of every pixel), so rendering k passes, pt_read_accum,
 * a new context with the same scene / uniforms / row partition, pt_load_accum and k more passes
 * give the bits of 2k uninterrupted passes.  `dst` / `src`: host or device pointers to
 * local_rows*width*16 bytes.  Both synchronise the stream. */
int pt_read_accum(pt_ctx* ctx, float* dst, size_t bytes);
int pt_load_accum(pt_ctx* ctx, const float* src, size_t bytes);
/* Render into caller-owned device memory (e.g. a torch tensor) instead; NULL restores. */
int pt_bind_accum(pt_ctx* ctx, void* dev_ptr, size_t bytes);
/* Use a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the own stream.  The
 * device's default stream IS the NULL handle: name it as hipStreamLegacy, PT_STREAM_LEGACY. */
int pt_set_stream(pt_ctx* ctx, void* hip_stream);
#define PT_STREAM_LEGACY ((void*)1) /* = hipStreamLegacy */

/* ---- temporal blend of the reference, static/shader.frag:387-404 + src/webgl.rs:186-204 --------
 * Blends the current resolved, gamma-encoded frame with `prev_rgba8` (the ping-pong texture)
 * using params.render_count / should_average / last_frame_weight, writes RGBA8 to `out_rgba8`.
 * Both are device or host pointers to local_rows*width*4 bytes. */
int pt_blend_rgba8(pt_ctx* ctx, const uint8_t* prev_rgba8, uint8_t* out_rgba8);

/* ---- the reference's FRAME on device-resident textures: webgl::render (src/webgl.rs:180-205) ------
 * The context owns the two RGBA8 ping-pong textures of src/webgl.rs:82-123 (local_rows*width texels
 * each, cleared to 0: alpha 0 = "no data", static/shader.frag:391) and a canvas, all in HBM.
 * pt_render_frame is ONE animation tick of src/lib.rs:92-102 with the current uniforms: trace one
 * pass (samples_per_pixel samples at u_time = time + float(first_pass) * time_step), blend it with
 * texture[(even_odd_count + 1) % 2] by the shader's render() rule using params.render_count /
 * should_average / last_frame_weight, draw the result to the canvas and, when should_average, to
 * texture[even_odd_count % 2].  Asynchronous on the context's stream; nothing crosses PCIe.
 * pt_render_frames replays n_frames ticks at a constant frame interval from captured hipGraphs —
 * groups of 64 (while their slabs stay below 1 GiB), 16 and 4 frames (ONE trace launch renders a group's frames as its passes into slabs of
 * their own, allocated by the first call that needs them; their blends follow in order; advance) and
 * single frames for the remainder (trace + blend + advance) — with the per-frame
 * state of State::update_render_globals (src/state.rs:443-450) kept on the device: frame k of the
 * call renders at u_time = time + float(first_pass + k) * time_step with
 * render_count = min(params.render_count + k, max_render_count) and even_odd_count + k — the bits
 * of n_frames pt_render_frame calls made with those uniforms (u_time is fp32(time) + float(k) * fp32(time_step),
 * computed in fp32 on the device: it equals fp32(time + k * time_step) rounded from the host's f64 — what a tick-by-tick
 * host uploads — for every k only when time, time_step and their products are exact in fp32, e.g. whole or half
 * milliseconds; otherwise some frames get a neighbouring seed: equally valid samples, not the same bits).  That equals n_frames ticks of the
 * reference's loop only while nothing but the clock changes between them: with should_average off
 * just the first tick draws (update_render_globals clears should_render, src/state.rs:443-447), with a
 * movement key held every tick has its own camera — issue such ticks one by one (pt_render_frame).
 * The graph is re-captured only when something it bakes in has changed (uniforms, scene size, launch
 * shape, stream).  NOT on a context bound to the legacy default stream (PT_STREAM_LEGACY): HIP does not
 * capture that stream, the call returns PT_ERR_INVALID there.  The accumulation buffer of
 * pt_render / pt_resolve is not involved.  pt_read_canvas / pt_read_texture copy RGBA8 texels out
 * (host or device pointer; they synchronise), pt_write_texture copies a texture in. */
int pt_clear_textures(pt_ctx* ctx);
int pt_render_frame(pt_ctx* ctx, uint32_t even_odd_count);
int pt_render_frames(pt_ctx* ctx, uint32_t even_odd_count, uint32_t max_render_count, uint32_t n_frames);
int pt_read_canvas(pt_ctx* ctx, uint8_t* rgba_out);
int pt_read_texture(pt_ctx* ctx, int index, uint8_t* rgba_out);
int pt_write_texture(pt_ctx* ctx, int index, const uint8_t* rgba_in);

/* ---- diagnostics ------------------------------------------------------------------------------ */
int pt_get_stats(pt_ctx* ctx, PtStats* out);
int pt_set_option(pt_ctx* ctx, int key, int value);
/* Fits the context to scene AND uniforms.  (i) The uniform grid pt_set_spheres built for rays that start within twice the
 * scene's radius of its middle is rebuilt for the margin class that renders THIS view fastest: the smallest class that
 * covers the camera set by pt_set_params is a lower bound (a camera outside it sends every primary ray down the far path), and
 * whether a class at or above it wins depends on where bounce rays start, so the candidates are measured — one timed launch of
 * a few passes each (as many as make ~4 ms, at most four and at most n_passes): the class the camera needs, the default class when that is smaller, and up to two classes wider while
 * the launch's far-ray tally says such rays matter and the wider class keeps winning; then, on scenes whose staged entries
 * take more than 16 KB of the LDS, the build that gathers its entries from L2 against the LDS-staged one (PtStats.grid_kernel_build)
 * (pt_set_option PT_OPT_GRID_FIT 1: no launches, the class the camera needs).  Speed only, the image does not depend on it; skipped once a launch has been captured
 * into a caller's hipGraph.  (ii) Settles PT_GEOM_AUTO now
 * instead of lazily: renders n_passes passes with the current scene
 * and uniforms once cold and once per usable path, keeps the fastest path.  Whenever it has launched anything it clears the
 * accumulation and statistics again.  Synchronous; a set-up call like pt_reserve_passes (which must have reserved n_passes).
 * Call it before capturing pt_render* into a hipGraph: a captured launch keeps the path it was
 * captured with (no measuring happens while a stream is capturing). */
int pt_tune(pt_ctx* ctx, uint32_t n_passes);
/* The frame loop's half of (i): rebuild the grid when the CURRENT camera has left the region it serves — for the smallest
 * margin class that covers the camera, never below the default class (3 scene radii: whether a tighter class pays is a
 * measurement, pt_tune's) — or, with only_if_stale == 0, also when the grid is looser than that.  Nothing else: no launches,
 * accumulation, textures and statistics untouched.  What a frame loop calls when PtStats.grid_fit_stale (or pt_grid_fit(ctx))
 * says so: the reference's camera moves every tick (State::update_position, src/state.rs:411-441), and a camera that has left
 * the region the grid was fitted to sends every primary ray down the far path.  Synchronises the stream when it rebuilds
 * (~2 ms of host work for 10 000 spheres); a no-op (PT_OK) when the grid fits, when there is none, when another geometry path
 * is forced or settled, and once a launch has been captured into a caller's hipGraph (its arguments hold the old grid's
 * numbers; pt_render_frames' own graphs are re-captured by themselves). */
int pt_refit_grid(pt_ctx* ctx, int only_if_stale);
/* PtStats.grid_fit_stale without the synchronisation pt_get_stats implies: 0 / 1 / 2 as there, < 0 on error.  Host arithmetic only. */
int pt_grid_fit(pt_ctx* ctx);
/* The hierarchy pt_set_spheres builds for PT_GEOM_BVH, on the host (no device needed; tests
 * check its invariants): nodes = 8 floats each {lo.xyz, bits(skip), hi.xyz, bits(first slot of
 * the leaf | 0xffffffff)}, slots = 4 floats each {cx, cy, cz, r*r}, slot_index = original sphere
 * index per slot (0xffffffff = padding), margin4 = {c0.xyz, s0}, counts5 = {n_nodes, n_slots,
 * n_tree_slots, n_outliers, depth}; nodes16 = the packed form the kernels read, 4 words per node
 * for n_nodes + 1 nodes {lo.x|lo.y, lo.z|hi.x, hi.y|hi.z as binary16 of (x - c0) * kscale rounded
 * outward, skip | leaf_number << 16}; nodes32 = the fp32 form small scenes use, 8 floats per
 * node for n_nodes + 1 nodes {lo - c0, bits(32 * skip), hi - c0, bits(leaf_number)}.  Array pointers
 * may be NULL (sizes only); capacities are in elements.  Returns PT_ERR_NOT_READY when the scene gets no hierarchy (fewer than 16
 * spheres, non-finite values), PT_ERR_CAPACITY when an array is too small. */
int pt_build_bvh(const PtSphere* spheres, uint32_t n, float* nodes, size_t node_floats, float* slots,
                 size_t slot_floats, uint32_t* slot_index, size_t n_index, float* margin4,
                 uint32_t* counts5, uint32_t* nodes16, size_t n_words16, float* kscale,
                 float* nodes32, size_t n_floats32);
/* The grid pt_set_spheres builds for PT_GEOM_GRID, on the host (no device needed; tests check the
 * registration invariant): dims9 = {n[3] as floats' bit patterns are NOT used: see counts}, i.e.
 * counts8 = {n.x, n.y, n.z, n_cell_entries, n_always, n_entries, max entries per cell, non-empty
 * cells}; geom12 = {lo.xyz, h.xyz, hi.xyz, c0.xyz}; margin4 = {s0, rmin, rmax, d_near};
 * delta_g = registration inflation; cells = n.x*n.y*n.z records (first entry | entries << 24),
 * entries = 4 floats each {cx, cy, cz, r*r}, entry_index = original sphere
 * index per entry (0xffffffff = padding).  Array pointers may be NULL (sizes only); capacities in
 * elements.  PT_ERR_NOT_READY when the scene gets no grid, PT_ERR_CAPACITY when an array is too small. */
int pt_build_grid(const PtSphere* spheres, uint32_t n, uint32_t* counts8, float* geom12, float* margin4,
                  float* delta_g, uint32_t* cells, size_t n_cells, float* entries, size_t entry_floats,
                  uint32_t* entry_index, size_t n_index);
/* What the grid kernels' entry test and cell look-up read beside geom12 / margin4 (tests emulate the
 * kernel's walk on the host with exactly these): out10 = {r2_near: rays with |o - c0|^2 <= it walk the
 * cells; lo_n.xyz, hi_n.xyz: the grid's box widened for the rounding of the slab test; inv_h.xyz}. */
int pt_grid_walk_constants(const PtSphere* spheres, uint32_t n, float* out10);
const char* pt_last_error(pt_ctx* ctx); /* ctx may be NULL: last create-time error */
int pt_abi_version(void);
int pt_device_count(void);
/* Device-side evaluation of single arithmetic building blocks (hash, sin/cos, cbrt, ...) for
 * parity tests; kinds are listed in ray_tracer_webgl_amd/csrc/pt_kernel_args.h.  in/out are
 * host pointers, n_in/n_out their lengths in floats, n the number of work items. */
int pt_probe(pt_ctx* ctx, int kind, const float* in, size_t n_in, float* out, size_t n_out,
             uint32_t n);
uint32_t pt_local_rows(uint32_t height, uint32_t band_rows, uint32_t band_index, uint32_t band_count);
/* The image row (0 = bottom) that local row `local_row` of band `band_index` is: rank r of n owns the rows y with
 * (y / band_rows) % n == r, in ascending order (PtParams.band_*).  What a host needs to put gathered per-rank buffers
 * back into image order (examples/render_bands.c): row l of rank r goes to image row
 * pt_band_row(band_rows, r, n, l), for l < pt_local_rows(height, band_rows, r, n).  Pure arithmetic, no device. */
uint32_t pt_band_row(uint32_t band_rows, uint32_t band_index, uint32_t band_count, uint32_t local_row);

/* ---- host-side camera derivation: State::update_pipeline (src/state.rs:319-347) ----------------
 * Double precision throughout, narrowed to float at the end like Vec3::to_array
 * (src/math.rs:107-109).  Fills the camera members + width/height + lens_radius of *out and
 * leaves the rest of *out untouched. */
int pt_camera_from_state(const PtCameraIn* in, PtParams* out);
int pt_camera_look_at(const PtLookAtIn* in, PtParams* out);

/* ---- host-side scene types: src/glsl.rs:27-40 ---------------------------------------------------
 * The reference keeps spheres in f64 on the host (Vec3 is 3 x f64, src/math.rs:17; radius f64;
 * fuzz / refraction_index f32) and narrows to f32 only at upload (src/webgl.rs:232-262). */
typedef struct PtHostSphere {
  double center[3];
  double radius;
  int32_t type;
  int32_t uuid;
  double albedo[3];
  float fuzz;
  float refraction_index;
} PtHostSphere;

/* The narrowing webgl::set_geometry performs (src/webgl.rs:225-274, Vec3::to_array
 * src/math.rs:107-109, `sphere.radius as f32`). */
int pt_narrow_spheres(const PtHostSphere* in, uint32_t n, PtSphere* out);
/* glsl::set_sphere_uuids (src/glsl.rs:84-88): uuid = index. */
int pt_set_sphere_uuids(PtHostSphere* spheres, uint32_t n);

/* ---- the reference's built-in scene: State::default sphere_list (src/state.rs:148-257) ---------
 * Writes up to cap spheres, returns the scene's sphere count (9). */
int pt_default_scene(PtHostSphere* out, uint32_t cap);
/* State::default camera block (src/state.rs:98-125) for a w*h canvas. */
int pt_default_camera(uint32_t width, uint32_t height, PtCameraIn* out);

/* ---- CPU pick ray: glsl::get_center_hit (src/glsl.rs:213-239) + Sphere::hit (:42-82), f64,
 * t_min = 0, t_max = inf.  Returns 1 on hit (fills t, hit point, normal, front_face, uuid),
 * 0 on miss, <0 on error. */
typedef struct PtCenterHit {
  double t;
  double hit_point[3];
  double normal[3];
  int32_t front_face;
  int32_t uuid;
} PtCenterHit;
int pt_center_hit(const PtHostSphere* spheres, uint32_t n, const PtCameraIn* cam,
                  PtCenterHit* out);

/* ---- the reference's State object (src/state.rs:31-94) behind an opaque handle -------------------
 * The caller side of the boundary: camera + render bookkeeping exactly as src/state.rs keeps
 * them (f64), so an interactive / turntable harness can drive the tracer the way src/lib.rs:65-104
 * drives WebGL.  Pure host code, no GPU needed. */
typedef struct pt_state pt_state;

typedef struct PtStateView {
  uint32_t width, height;
  uint32_t samples_per_pixel, max_depth;
  double aspect_ratio;
  double camera_origin[3], camera_front[3], vup[3];
  double yaw, pitch, camera_field_of_view;
  double u[3], v[3], w[3];
  double aperture, lens_radius, focus_distance;
  double viewport_height, viewport_width;
  double horizontal[3], vertical[3], lower_left_corner[3];
  double cursor_point[3];
  int32_t selected_object;
  int32_t is_paused, should_average, should_render;
  uint32_t even_odd_count, render_count, max_render_count;
  float last_frame_weight;
  uint32_t n_spheres;
} PtStateView;

int pt_state_create(pt_state** out, uint32_t width, uint32_t height); /* State::default :96-315 */
int pt_state_destroy(pt_state* s);
int pt_state_get(const pt_state* s, PtStateView* out);
int pt_state_set_fov(pt_state* s, double fov_radians);                 /* :349-352 */
int pt_state_set_camera_angles(pt_state* s, double yaw, double pitch); /* :354-358 */
int pt_state_set_camera_origin(pt_state* s, const double origin[3]);
int pt_state_set_lens(pt_state* s, double aperture, double focus_distance); /* lens_radius = aperture/2 */
int pt_state_set_quality(pt_state* s, uint32_t samples_per_pixel, uint32_t max_depth);
int pt_state_set_flags(pt_state* s, int is_paused, int should_average, float last_frame_weight);
/* KeydownMap :14-28 as a bit mask: 1 w, 2 a, 4 s, 8 d, 16 space, 32 shift */
int pt_state_set_keys(pt_state* s, uint32_t key_mask);
int pt_state_update_position(pt_state* s, double dt_ms);      /* :411-441 (+ autofocus :453-471) */
int pt_state_update_render_globals(pt_state* s);              /* :443-450 */
int pt_state_resize(pt_state* s, uint32_t width, uint32_t height); /* :364-398, State half */
int pt_state_should_render(const pt_state* s, int should_save);    /* src/lib.rs:77-82 */
int pt_state_set_spheres(pt_state* s, const PtHostSphere* spheres, uint32_t n); /* uuids reassigned */
int pt_state_spheres(const pt_state* s, PtSphere* out, uint32_t cap);  /* set_geometry narrowing */
int pt_state_to_params(const pt_state* s, double now_ms, PtParams* out); /* run_setters :279-593 */
/* dom::get_adjusted_screen_dimensions (src/dom.rs:277-291) */
int pt_adjusted_screen_dimensions(double raw_width, double raw_height, uint32_t* w, uint32_t* h);

#ifdef __cplusplus
}
#endif
#endif /* PTRACE_H */
