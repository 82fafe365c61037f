/*
 * pt_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the hot path of austintheriot/ray-tracer-webgl: the per-fragment
 * program of static/shader.frag (whole file) with the pixel->v_position mapping of
 * static/shader.vert:8, plus the host-side camera derivation of src/state.rs:319-347.
 * Every function cites the reference lines it follows (paths relative to the reference root).
 *
 * Who may use this file: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — as
 * the CHECKER (and as a timed CPU baseline), never as something shipped.  Nothing under
 * ray_tracer_webgl_amd/ includes, links or calls it, and the product library has no CPU path.
 *
 * PARITY UNPINNED.  The reference holds no test, fixture or golden vector for this path
 * (tests/web.rs:10-13 is `assert_eq!(1 + 1, 2)`), its only implementation is GLSL ES 3.00,
 * which nothing in the build image can compile or execute, and its seeds are wall-clock
 * (src/webgl.rs:320-331).  This oracle is therefore pinned only by (a) an independent numpy
 * restatement of the integer hash (tests/test_oracle_kat.py), (b) hand-derived known-answer
 * cases for intersection / scatter, (c) the f64 Sphere::hit of src/glsl.rs:42-82 restated in
 * ora_center_hit_f64, (d) the committed fixtures under tests/golden/, which this oracle
 * itself generated (tests/golden/make_golden.py), and (e) for the deterministic part of the path
 * (camera, pixel mapping, background, gamma) 43 sky pixels of the reference's own published
 * screenshot of State::default, matched to +-1.5/255 (tests/golden/reference_sky_pixels.json),
 * and that screenshot's sky / not-sky silhouettes, matched on every pixel outside a 2-pixel
 * antialiasing band (tests/golden/reference_sky_mask.npz) -> scene geometry and camera, and 170
 * mirror pixels of its fuzz-0 metal spheres (+-2/255, reference_mirror_pixels.json) -> hit point,
 * normal orientation for either sign of the radius, reflect(), metal scatter.
 * The Monte-Carlo part (RNG use, scatter, accumulation) remains unpinned.
 *
 * ARITHMETIC CONTRACT ("PT-SPEC").  GLSL leaves operation order, fusion and
 * built-in precision to the driver, so a bit-reproducible restatement has to pin them:
 *   - IEEE-754 binary32, round-to-nearest-even, subnormals kept, for + - * / sqrt and
 *     int<->float conversion.  Build with -ffp-contract=off and never -ffast-math.
 *   - fused multiply-add ONLY where written as fmaf() below:
 *       dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *       o + d*t    = fma(d, t, o)                                  (ray_at, shader.frag:106)
 *       mix(1,b,t) = fma(b, t, 1 - t)                              (background, :292)
 *       |oc|^2 - r^2 = fma(oc.z,oc.z, fma(oc.y,oc.y, fma(oc.x,oc.x, -(r*r))))   (:149)
 *     and the handful of places marked "PT-SPEC fma" in the code.
 *   - pow(x,2.) = x*x (shader.frag:149,150,205; also defined for the reference's negative
 *     radii, src/state.rs:200,213); length_squared(v) = dot(v,v) instead of
 *     pow(length(v),2.) (:110-112); pow(x,5.) = (x*x)*(x*x)*x (:206).
 *   - normalize(v) = v * (1.0f / sqrtf(dot(v,v))).
 *   - sin/cos of (2*pi*u) — the only way the shader ever calls them (:117-120, :124-127) —
 *     are ora_sincos2pi(u): exact quadrant reduction of u, then Cephes-style minimax
 *     polynomials; pow(x, 1./3.) (:119) is ora_cbrt(x): integer seed + 3 Newton steps on
 *     x^(-1/3).  Both are plain fp32 + - * fma sequences, so host libm and the GPU's
 *     approximate v_sin/v_cos/v_exp/v_log cannot diverge.
 *   - min(x,y) = (y < x) ? y : x (GLSL ES 3.00 §8.3); comparisons exactly as written in the
 *     shader, so NaNs take the same branches.
 *   - reflect / refract as defined in GLSL ES 3.00 §8.5, with the fma placement shown below.
 *
 * BUILD EXTENSIONS beyond the shader: unbounded sphere count
 * (the shader caps at 15, :103), material type 3 = emissive, background_mode black, and the
 * row-band partition of PtParams.  The per-pass OUTPUT is the linear radiance SUM of the
 * pass's samples (not the gamma-encoded mean of :376-380): passes add up in an fp32
 * accumulation buffer and ora_resolve applies the /spp and sqrt at read-out.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ptrace.h"

#define ORA_API __attribute__((visibility("default")))

/* static/shader.frag:4-6 */
#define ORA_MAX_T 1e5f
#define ORA_MIN_T 0.001f
#define ORA_TWO_PI 6.2831855f /* fp32(2*PI), PI = 3.141592653589793 (:4) */

typedef struct { float x, y, z; } v3;
typedef struct { v3 origin, direction; } ray_t; /* shader.frag:39-42 */

typedef struct { /* shader.frag:63-70 */
  v3 hit_point;
  float hit_t;
  v3 normal;
  int front_face;
  int mat_type;
  v3 albedo;
  float fuzz;
  float refraction_index;
  int uuid;
} hit_record_t;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* PT-SPEC dot */
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
/* PT-SPEC normalize */
static inline v3 normalize3(v3 a) {
  float inv = 1.0f / sqrtf(dot3(a, a));
  return vscale(a, inv);
}
/* GLSL min, ES 3.00 §8.3 */
static inline float glsl_min(float x, float y) { return (y < x) ? y : x; }

/* ============================ RNG: shader.frag:8-36 ======================================= */

/* shader.frag:15-19 */
ORA_API uint32_t ora_base_hash(uint32_t px, uint32_t py) {
  uint32_t qx = 1103515245u * ((px >> 1) ^ py);
  uint32_t qy = 1103515245u * ((py >> 1) ^ px);
  uint32_t h32 = 1103515245u * (qx ^ (qy >> 3));
  return h32 ^ (h32 >> 16);
}

/* `vec2(seed += .1, seed += .1)`: GLSL ES 3.00 evaluates arguments left to right, so the two
 * hashed floats are seed+0.1 and (seed+0.1)+0.1, each a rounded fp32 add (shader.frag:22,27,33) */
static inline uint32_t seed_step_hash(float* seed) {
  float s1 = *seed + 0.1f;
  float s2 = s1 + 0.1f;
  *seed = s2;
  return ora_base_hash(f2u(s1), f2u(s2));
}

/* shader.frag:21-24.  float(0xffffffffU) rounds to 2^32, so the scale is exactly 2^-32 and the
 * range is [0,1] inclusive (float(n) rounds up to 2^32 for n >= 0xffffff80). */
ORA_API float ora_hash1(float* seed) {
  uint32_t n = seed_step_hash(seed);
  return (float)n * (1.0f / 4294967296.0f);
}

/* shader.frag:26-30.  float(0x7fffffff) rounds to 2^31; dividing by it is exact scaling. */
ORA_API void ora_hash2(float* seed, float out[2]) {
  uint32_t n = seed_step_hash(seed);
  out[0] = (float)(n & 0x7fffffffu) / 2147483648.0f;
  out[1] = (float)((n * 48271u) & 0x7fffffffu) / 2147483648.0f;
}

/* shader.frag:32-36: components (n, n*16807, n*48271) */
ORA_API void ora_hash3(float* seed, float out[3]) {
  uint32_t n = seed_step_hash(seed);
  out[0] = (float)(n & 0x7fffffffu) / 2147483648.0f;
  out[1] = (float)((n * 16807u) & 0x7fffffffu) / 2147483648.0f;
  out[2] = (float)((n * 48271u) & 0x7fffffffu) / 2147483648.0f;
}

/* ============================ PT-SPEC transcendental replacements ========================== */

/* sin(2*pi*u), cos(2*pi*u) for u >= 0 (the shader only ever forms sin/cos of hash*2*PI,
 * shader.frag:117-120 and :124-127).  q = nearest quarter turn (exact), f = u - q/4 (exact),
 * x = f * fp32(2*pi) in [-pi/4, pi/4], Cephes sinf/cosf minimax polynomials, quadrant swap. */
ORA_API void ora_sincos2pi(float u, float* s_out, float* c_out) {
  float q = rintf(u * 4.0f); /* round-half-even, like v_rndne_f32 */
  float f = u - q * 0.25f;
  float x = f * ORA_TWO_PI;
  float x2 = x * x;
  float ps = fmaf(fmaf(-1.9515295891e-4f, x2, 8.3321608736e-3f), x2, -1.6666654611e-1f);
  float s = fmaf(x * x2, ps, x);
  float pc = fmaf(fmaf(2.443315711809948e-5f, x2, -1.388731625493765e-3f), x2,
                  4.166664568298827e-2f);
  float c = fmaf(x2 * x2, pc, fmaf(-0.5f, x2, 1.0f));
  int qi = ((int)q) & 3;
  float ss = (qi & 1) ? c : s;
  float cc = (qi & 1) ? s : c;
  if (qi == 1 || qi == 2) cc = -cc;
  if (qi >= 2) ss = -ss;
  *s_out = ss;
  *c_out = cc;
}

/* x^(1/3) for finite x >= 0: y ~ x^(-1/3) from an integer seed, three Newton steps
 * y <- y*(4/3 - (x*y^3)/3), result (x*y)*y.  cbrt(0) = 0 exactly. */
ORA_API float ora_cbrt(float x) {
  if (x == 0.0f) return 0.0f;
  float y = u2f(0x54a2fa8cu - f2u(x) / 3u);
  for (int i = 0; i < 3; i++) {
    float t = x * y;
    t = t * y;
    t = t * y;
    y = y * fmaf(t, -0.33333334f, 1.3333334f);
  }
  return (x * y) * y;
}

/* ============================ sampling helpers: shader.frag:114-133 ======================= */

/* shader.frag:114-121 */
static v3 random_in_unit_sphere(float* seed) {
  float h[3];
  ora_hash3(seed, h);
  float hx = fmaf(h[0], 2.0f, -1.0f); /* h.x*2 - 1: 2*h.x is exact, one rounding */
  /* phi = h.y * 2*PI is only ever used as sin(phi), cos(phi) -> PT-SPEC sincos2pi(h.y) */
  float sp, cp;
  ora_sincos2pi(h[1], &sp, &cp);
  float r = ora_cbrt(h[2]); /* pow(h.z, 1./3.) */
  float sq = sqrtf(fmaf(-hx, hx, 1.0f)); /* sqrt(1. - h.x*h.x), PT-SPEC fma */
  return V(r * (sq * sp), r * (sq * cp), r * hx);
}

/* shader.frag:123-129: angle first, then radius */
static void random_in_unit_circle(float* seed, float* x, float* y) {
  float ua = ora_hash1(seed);
  float sa, ca;
  ora_sincos2pi(ua, &sa, &ca); /* a = hash1*2*PI, used only as cos(a), sin(a) */
  float r = sqrtf(ora_hash1(seed));
  *x = r * ca;
  *y = r * sa;
}

/* shader.frag:131-133 */
static v3 random_unit_vec(float* seed) { return normalize3(random_in_unit_sphere(seed)); }

/* ============================ intersection: shader.frag:136-196 =========================== */

/* shader.frag:136-143 */
static void set_hit_record_front_face(hit_record_t* h, const ray_t* r, v3 outward_normal) {
  h->front_face = dot3(r->direction, outward_normal) < 0.0f;
  if (h->front_face) h->normal = outward_normal;
  else h->normal = vneg(outward_normal);
}

/* shader.frag:145-173 */
static int hit_sphere(const PtSphere* sp, const ray_t* r, float t_min, float t_max,
                      hit_record_t* h) {
  v3 center = V(sp->center[0], sp->center[1], sp->center[2]);
  v3 oc = vsub(r->origin, center);
  float a = dot3(r->direction, r->direction);
  float half_b = dot3(oc, r->direction);
  /* length_squared(oc) - pow(radius,2.) as ONE fma chain seeded with -r*r (PT-SPEC fma) */
  float r2 = sp->radius * sp->radius;
  float c = fmaf(oc.z, oc.z, fmaf(oc.y, oc.y, fmaf(oc.x, oc.x, -r2)));
  float discriminant = fmaf(-a, c, half_b * half_b); /* pow(half_b,2.) - a*c, PT-SPEC fma */

  if (discriminant < 0.0f) return 0;

  float sqrtd = sqrtf(discriminant);
  float root = (-half_b - sqrtd) / a;
  if (root < t_min || t_max < root) {
    root = (-half_b + sqrtd) / a;
    if (root < t_min || t_max < root) return 0;
  }

  h->mat_type = sp->type;
  h->albedo = V(sp->albedo[0], sp->albedo[1], sp->albedo[2]);
  h->fuzz = sp->fuzz;
  h->refraction_index = sp->refraction_index;
  h->hit_t = root;
  /* ray_at, shader.frag:106-108 */
  h->hit_point = V(fmaf(r->direction.x, root, r->origin.x), fmaf(r->direction.y, root, r->origin.y),
                   fmaf(r->direction.z, root, r->origin.z));
  h->uuid = sp->uuid;
  v3 d = vsub(h->hit_point, center);
  v3 outward_normal = V(d.x / sp->radius, d.y / sp->radius, d.z / sp->radius);
  set_hit_record_front_face(h, r, outward_normal);
  return 1;
}

/* shader.frag:175-196.  `is_active` == (i < n): set_geometry marks every uploaded sphere
 * active (src/webgl.rs:264-266) and the rest of the uniform array stays 0. */
static int hit_world(const PtSphere* spheres, uint32_t n, const ray_t* r, float t_min, float t_max,
                     hit_record_t* hit_record) {
  int hit_anything = 0;
  float closest_so_far = t_max;
  hit_record_t temp;
  for (uint32_t i = 0; i < n; i++) {
    if (hit_sphere(&spheres[i], r, t_min, closest_so_far, &temp)) {
      hit_anything = 1;
      closest_so_far = temp.hit_t;
      *hit_record = temp;
    }
  }
  return hit_anything;
}

/* ============================ materials: shader.frag:203-286 ============================== */

/* shader.frag:204-207 */
static float reflectance(float cosine, float reflection_index) {
  float q = (1.0f - reflection_index) / (1.0f + reflection_index);
  float r0 = q * q;
  float x = 1.0f - cosine;
  float x2 = x * x;
  float x5 = (x2 * x2) * x;
  return fmaf(1.0f - r0, x5, r0); /* r0 + (1-r0)*x^5, PT-SPEC fma */
}

/* GLSL reflect: I - 2*dot(N,I)*N */
static v3 glsl_reflect(v3 I, v3 N) {
  float k = 2.0f * dot3(N, I);
  return V(fmaf(-k, N.x, I.x), fmaf(-k, N.y, I.y), fmaf(-k, N.z, I.z));
}

/* GLSL refract: k = 1 - eta^2 (1 - dot(N,I)^2); k<0 ? 0 : eta*I - (eta*dot(N,I)+sqrt(k))*N */
static v3 glsl_refract(v3 I, v3 N, float eta) {
  float dni = dot3(N, I);
  float k = fmaf(-(eta * eta), fmaf(-dni, dni, 1.0f), 1.0f);
  if (k < 0.0f) return V(0.0f, 0.0f, 0.0f);
  float t = fmaf(eta, dni, sqrtf(k));
  return V(fmaf(-t, N.x, eta * I.x), fmaf(-t, N.y, eta * I.y), fmaf(-t, N.z, eta * I.z));
}

/* shader.frag:210-286.  Returns did_scatter. */
static int scatter(const ray_t* r, const hit_record_t* h, v3* attenuation, ray_t* scattered,
                   float* seed) {
  if (h->mat_type == PT_DIFFUSE) { /* :212-229 */
    *attenuation = h->albedo;
    v3 scatter_direction = vadd(h->normal, random_unit_vec(seed));
    scattered->origin = h->hit_point;
    scattered->direction = scatter_direction;
    return 1;
  }
  if (h->mat_type == PT_METAL) { /* :232-247 — direction NOT normalised, RNG always consumed */
    *attenuation = h->albedo;
    v3 reflected = glsl_reflect(r->direction, h->normal);
    v3 rs = random_in_unit_sphere(seed);
    v3 fuzzed = V(fmaf(h->fuzz, rs.x, reflected.x), fmaf(h->fuzz, rs.y, reflected.y),
                  fmaf(h->fuzz, rs.z, reflected.z));
    scattered->origin = h->hit_point;
    scattered->direction = fuzzed;
    return dot3(h->normal, fuzzed) > 0.0f;
  }
  if (h->mat_type == PT_GLASS) { /* :250-282 */
    *attenuation = h->albedo;
    float refraction_ratio = h->front_face ? (1.0f / h->refraction_index) : h->refraction_index;
    v3 unit_direction = normalize3(r->direction);
    float cos_theta = glsl_min(dot3(vneg(unit_direction), h->normal), 1.0f);
    float sin_theta = sqrtf(fmaf(-cos_theta, cos_theta, 1.0f));
    int cannot_refract = refraction_ratio * sin_theta > 1.0f;
    float reflectance_amount = reflectance(cos_theta, refraction_ratio);
    float random_float = ora_hash1(seed);
    v3 direction;
    if (cannot_refract || reflectance_amount > random_float)
      direction = glsl_reflect(unit_direction, h->normal);
    else
      direction = glsl_refract(unit_direction, h->normal, refraction_ratio);
    scattered->origin = h->hit_point;
    scattered->direction = direction;
    return 1;
  }
  return 0; /* :284-285 unrecognised material absorbs */
}

/* shader.frag:289-294 */
static v3 background(const ray_t* r) {
  float inv = 1.0f / sqrtf(dot3(r->direction, r->direction));
  float uy = r->direction.y * inv;
  float t = 0.5f * (uy + 1.0f);
  float omt = 1.0f - t;
  return V(fmaf(0.5f, t, omt), fmaf(0.7f, t, omt), fmaf(1.0f, t, omt));
}

/* shader.frag:297-339 (debug overlay :307-318 is dead: enable_debugging == 0, src/state.rs:259).
 * Extensions: PT_EMISSIVE ends the path with color*albedo; background_mode black. */
static v3 ray_color(const PtSphere* spheres, uint32_t n, const PtParams* p, ray_t r, float* seed,
                    uint64_t* segments) {
  v3 color = V(1.0f, 1.0f, 1.0f);
  for (int i = 0; i < p->max_depth; i++) {
    hit_record_t h;
    (*segments)++;
    if (hit_world(spheres, n, &r, ORA_MIN_T, ORA_MAX_T, &h)) {
      if (h.mat_type == PT_EMISSIVE) return vmul(color, h.albedo);
      v3 attenuation;
      ray_t scattered;
      int did_scatter = scatter(&r, &h, &attenuation, &scattered, seed);
      if (did_scatter) {
        r = scattered;
        color = vmul(color, attenuation);
      } else {
        return V(0.0f, 0.0f, 0.0f);
      }
    } else {
      if (p->background_mode == PT_BG_BLACK) return V(0.0f, 0.0f, 0.0f);
      return vmul(color, background(&r));
    }
  }
  return color; /* :338 depth exhausted -> throughput, not black */
}

/* shader.frag:342-351 */
static ray_t get_ray_from_camera(const PtParams* p, float s, float t, float* seed) {
  float cx, cy;
  random_in_unit_circle(seed, &cx, &cy); /* consumed even when lens_radius == 0 */
  float rdx = p->lens_radius * cx, rdy = p->lens_radius * cy;
  v3 off = V(fmaf(p->v[0], rdy, p->u[0] * rdx), fmaf(p->v[1], rdy, p->u[1] * rdx),
             fmaf(p->v[2], rdy, p->u[2] * rdx));
  v3 o = V(p->camera_origin[0], p->camera_origin[1], p->camera_origin[2]);
  /* ((llc + s*horizontal) + t*vertical) - origin - offset */
  v3 d = V(fmaf(t, p->vertical[0], fmaf(s, p->horizontal[0], p->lower_left_corner[0])),
           fmaf(t, p->vertical[1], fmaf(s, p->horizontal[1], p->lower_left_corner[1])),
           fmaf(t, p->vertical[2], fmaf(s, p->horizontal[2], p->lower_left_corner[2])));
  d = vsub(vsub(d, o), off);
  ray_t r;
  r.origin = vadd(o, off);
  r.direction = d;
  return r;
}

/* static/shader.vert:8 + rasteriser: v_position at the centre of pixel (px,py), y up.
 * PT-SPEC: v = float(2*p+1)/float(W) - 1. */
static inline float v_position_of(uint32_t p, uint32_t extent) {
  return (float)(2u * p + 1u) / (float)extent - 1.0f;
}

/* shader.frag:354-357 */
static inline float init_global_seed(float vx, float vy, float u_time) {
  return (float)ora_base_hash(f2u(vx), f2u(vy)) / 4294967296.0f + u_time;
}

/* One fragment, shader.frag:406-413 + :360-373 up to (not including) the /spp and sqrt:
 * returns the linear radiance SUM of the pass's samples. */
static v3 pixel_pass_sum(const PtSphere* spheres, uint32_t n, const PtParams* p, uint32_t px,
                         uint32_t py, float u_time, uint64_t* segments) {
  float vx = v_position_of(px, p->width), vy = v_position_of(py, p->height);
  float seed = init_global_seed(vx, vy, u_time);
  float st_s = (vx + 1.0f) * 0.5f, st_t = (vy + 1.0f) * 0.5f; /* :410 */
  float fw = (float)p->width, fh = (float)p->height;
  v3 color = V(0.0f, 0.0f, 0.0f);
  for (int i = 0; i < p->samples_per_pixel; i++) {
    float rnd[2];
    ora_hash2(&seed, rnd);
    float s = st_s + rnd[0] / fw; /* :366-369 jitter added to the pixel centre */
    float t = st_t + rnd[1] / fh;
    ray_t r = get_ray_from_camera(p, s, t, &seed);
    color = vadd(color, ray_color(spheres, n, p, r, &seed, segments));
  }
  return color;
}

/* ============================ exported single-function probes (KATs) ======================= */

ORA_API float ora_v_position(uint32_t p, uint32_t extent) { return v_position_of(p, extent); }
ORA_API float ora_init_seed(float vx, float vy, float t) { return init_global_seed(vx, vy, t); }

ORA_API void ora_random_in_unit_sphere(float* seed, float out[3]) {
  v3 r = random_in_unit_sphere(seed);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
ORA_API void ora_random_in_unit_circle(float* seed, float out[2]) {
  random_in_unit_circle(seed, &out[0], &out[1]);
}

typedef struct OraHit {
  int32_t hit;
  int32_t index; /* uuid */
  float t;
  float point[3];
  float normal[3];
  int32_t front_face;
} OraHit;

ORA_API int ora_hit_sphere(const PtSphere* sp, const float origin[3], const float dir[3],
                           float t_min, float t_max, OraHit* out) {
  ray_t r = {V(origin[0], origin[1], origin[2]), V(dir[0], dir[1], dir[2])};
  hit_record_t h;
  memset(&h, 0, sizeof h);
  int ok = hit_sphere(sp, &r, t_min, t_max, &h);
  out->hit = ok;
  if (ok) {
    out->index = h.uuid; out->t = h.hit_t; out->front_face = h.front_face;
    out->point[0] = h.hit_point.x; out->point[1] = h.hit_point.y; out->point[2] = h.hit_point.z;
    out->normal[0] = h.normal.x; out->normal[1] = h.normal.y; out->normal[2] = h.normal.z;
  }
  return ok;
}

ORA_API int ora_hit_world(const PtSphere* spheres, uint32_t n, const float origin[3],
                          const float dir[3], OraHit* out) {
  ray_t r = {V(origin[0], origin[1], origin[2]), V(dir[0], dir[1], dir[2])};
  hit_record_t h;
  memset(&h, 0, sizeof h);
  int ok = hit_world(spheres, n, &r, ORA_MIN_T, ORA_MAX_T, &h);
  out->hit = ok;
  if (ok) {
    out->index = h.uuid; out->t = h.hit_t; out->front_face = h.front_face;
    out->point[0] = h.hit_point.x; out->point[1] = h.hit_point.y; out->point[2] = h.hit_point.z;
    out->normal[0] = h.normal.x; out->normal[1] = h.normal.y; out->normal[2] = h.normal.z;
  }
  return ok;
}

typedef struct OraScatter {
  int32_t did_scatter;
  float attenuation[3];
  float origin[3];
  float direction[3];
  float seed_after;
} OraScatter;

/* hit_world + scatter for one ray (shader.frag:304-329) */
ORA_API int ora_scatter(const PtSphere* spheres, uint32_t n, const float origin[3],
                        const float dir[3], float seed, OraScatter* out) {
  ray_t r = {V(origin[0], origin[1], origin[2]), V(dir[0], dir[1], dir[2])};
  hit_record_t h;
  memset(out, 0, sizeof *out);
  if (!hit_world(spheres, n, &r, ORA_MIN_T, ORA_MAX_T, &h)) return -1;
  v3 att = V(0, 0, 0);
  ray_t sc = r;
  out->did_scatter = scatter(&r, &h, &att, &sc, &seed);
  out->attenuation[0] = att.x; out->attenuation[1] = att.y; out->attenuation[2] = att.z;
  out->origin[0] = sc.origin.x; out->origin[1] = sc.origin.y; out->origin[2] = sc.origin.z;
  out->direction[0] = sc.direction.x; out->direction[1] = sc.direction.y;
  out->direction[2] = sc.direction.z;
  out->seed_after = seed;
  return out->did_scatter;
}

ORA_API void ora_ray_color(const PtSphere* spheres, uint32_t n, const PtParams* p,
                           const float origin[3], const float dir[3], float* seed, float out[3],
                           uint64_t* segments) {
  ray_t r = {V(origin[0], origin[1], origin[2]), V(dir[0], dir[1], dir[2])};
  uint64_t seg = 0;
  v3 c = ray_color(spheres, n, p, r, seed, &seg);
  out[0] = c.x; out[1] = c.y; out[2] = c.z;
  if (segments) *segments = seg;
}

ORA_API void ora_camera_ray(const PtParams* p, float s, float t, float* seed, float origin[3],
                            float dir[3]) {
  ray_t r = get_ray_from_camera(p, s, t, seed);
  origin[0] = r.origin.x; origin[1] = r.origin.y; origin[2] = r.origin.z;
  dir[0] = r.direction.x; dir[1] = r.direction.y; dir[2] = r.direction.z;
}


/* First-hit map through pixel CENTRES (no jitter, no lens): uuid of the nearest sphere or -1.
 * Used to compare silhouettes with the reference's screenshot (tests/test_oracle_kat.py). */
ORA_API void ora_first_hit_map(const PtSphere* spheres, uint32_t n, const PtParams* p, int32_t* out) {
  v3 o = V(p->camera_origin[0], p->camera_origin[1], p->camera_origin[2]);
  for (uint32_t y = 0; y < p->height; y++)
    for (uint32_t x = 0; x < p->width; x++) {
      float vx = v_position_of(x, p->width), vy = v_position_of(y, p->height);
      float s = (vx + 1.0f) * 0.5f, t = (vy + 1.0f) * 0.5f;
      ray_t r;
      r.origin = o;
      r.direction = vsub(V(fmaf(t, p->vertical[0], fmaf(s, p->horizontal[0], p->lower_left_corner[0])),
                           fmaf(t, p->vertical[1], fmaf(s, p->horizontal[1], p->lower_left_corner[1])),
                           fmaf(t, p->vertical[2], fmaf(s, p->horizontal[2], p->lower_left_corner[2]))), o);
      hit_record_t h;
      out[(size_t)y * p->width + x] = hit_world(spheres, n, &r, ORA_MIN_T, ORA_MAX_T, &h) ? h.uuid : -1;
    }
}

/* ============================ frame drivers ============================================== */

static inline int row_owned(const PtParams* p, uint32_t y) {
  if (p->band_count <= 1 || p->band_rows == 0) return 1;
  return (y / p->band_rows) % p->band_count == p->band_index;
}

ORA_API uint32_t ora_local_rows(const PtParams* p) {
  uint32_t n = 0;
  for (uint32_t y = 0; y < p->height; y++) n += row_owned(p, y);
  return n;
}

typedef struct {
  const PtSphere* spheres; uint32_t n; const PtParams* p; float u_time;
  float* slab; /* local_rows*width*4 */
  const uint32_t* rows; uint32_t n_rows; /* owned global rows, ascending */
  uint32_t x0, x1;                      /* column window [x0,x1) actually computed */
  uint32_t tid, nthreads; uint64_t segments;
} job_t;

static void* pass_worker(void* arg) {
  job_t* j = (job_t*)arg;
  uint64_t seg = 0;
  for (uint32_t ly = j->tid; ly < j->n_rows; ly += j->nthreads) {
    uint32_t y = j->rows[ly];
    for (uint32_t x = j->x0; x < j->x1; x++) {
      v3 c = pixel_pass_sum(j->spheres, j->n, j->p, x, y, j->u_time, &seg);
      float* o = j->slab + 4 * ((size_t)ly * j->p->width + x);
      o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = (float)j->p->samples_per_pixel;
    }
  }
  j->segments = seg;
  return NULL;
}

/*
 * One pass (one "draw" of the reference, src/webgl.rs:169-178) restricted to the owned rows and
 * to the pixel window [x0,x1) x [y0,y1) (global coordinates; pass 0,width,0,height for all).
 * ADDS each pixel's radiance sum into accum (local_rows*width float4; .a accumulates spp).
 * Pixels outside the window are untouched.  Returns the number of ray segments traced.
 */
ORA_API uint64_t ora_render_pass(const PtSphere* spheres, uint32_t n, const PtParams* p,
                                 float u_time, float* accum, uint32_t x0, uint32_t x1,
                                 uint32_t y0, uint32_t y1, uint32_t nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  if (x1 > p->width) x1 = p->width;
  if (y1 > p->height) y1 = p->height;
  uint32_t* rows = (uint32_t*)malloc(sizeof(uint32_t) * (p->height + 1));
  uint32_t* lrow = (uint32_t*)malloc(sizeof(uint32_t) * (p->height + 1));
  uint32_t n_rows = 0, local = 0;
  for (uint32_t y = 0; y < p->height; y++) {
    if (!row_owned(p, y)) continue;
    if (y >= y0 && y < y1) { rows[n_rows] = y; lrow[n_rows] = local; n_rows++; }
    local++;
  }
  size_t slab_elems = (size_t)n_rows * p->width * 4;
  float* slab = (float*)calloc(slab_elems ? slab_elems : 1, sizeof(float));
  job_t jobs[256];
  pthread_t th[256];
  for (uint32_t t = 0; t < nthreads; t++) {
    job_t j = {spheres, n, p, u_time, slab, rows, n_rows, x0, x1, t, nthreads, 0};
    jobs[t] = j;
  }
  if (nthreads == 1) pass_worker(&jobs[0]);
  else {
    for (uint32_t t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, pass_worker, &jobs[t]);
    for (uint32_t t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  }
  uint64_t seg = 0;
  for (uint32_t t = 0; t < nthreads; t++) seg += jobs[t].segments;
  for (uint32_t i = 0; i < n_rows; i++)
    for (uint32_t x = x0; x < x1; x++) {
      float* a = accum + 4 * ((size_t)lrow[i] * p->width + x);
      const float* s = slab + 4 * ((size_t)i * p->width + x);
      a[0] += s[0]; a[1] += s[1]; a[2] += s[2]; a[3] += s[3];
    }
  free(slab); free(rows); free(lrow);
  return seg;
}

/* n_passes passes with u_time = p->time + float(first_pass + pass) * time_step (fp32 multiply, then add; a
 * step of 0 means 1), accumulated in pass order. */
ORA_API uint64_t ora_render_passes(const PtSphere* spheres, uint32_t n, const PtParams* p,
                                   uint32_t n_passes, float* accum, uint32_t x0, uint32_t x1,
                                   uint32_t y0, uint32_t y1, uint32_t nthreads) {
  uint64_t seg = 0;
  for (uint32_t k = 0; k < n_passes; k++)
    seg += ora_render_pass(spheres, n, p, p->time + (float)(p->first_pass + k) * (p->time_step != 0.0f ? p->time_step : 1.0f), accum, x0,
                           x1, y0, y1, nthreads);
  return seg;
}

/* Read-out: shader.frag:376-380 applied to the accumulated sum: scale = 1/float(total_spp);
 * color *= scale; optional sqrt; alpha = 1. */
ORA_API void ora_resolve(const float* accum, size_t n_pixels, uint32_t total_spp, int gamma,
                         float* rgba_out) {
  float scale = 1.0f / (float)total_spp;
  for (size_t i = 0; i < n_pixels; i++) {
    for (int c = 0; c < 3; c++) {
      float v = accum[4 * i + c] * scale;
      rgba_out[4 * i + c] = gamma ? sqrtf(v) : v;
    }
    rgba_out[4 * i + 3] = 1.0f;
  }
}

/* Framebuffer write of an RGBA8 target (src/webgl.rs:109-119): clamp to [0,1], round(x*255).
 * NaN clamps to 0. */
static inline uint8_t unorm8(float v) {
  if (!(v > 0.0f)) return 0;
  if (v >= 1.0f) return 255;
  return (uint8_t)(v * 255.0f + 0.5f);
}

ORA_API void ora_resolve_rgba8(const float* accum, size_t n_pixels, uint32_t total_spp, int gamma,
                               uint8_t* rgba_out) {
  float scale = 1.0f / (float)total_spp;
  for (size_t i = 0; i < n_pixels; i++) {
    for (int c = 0; c < 3; c++) {
      float v = accum[4 * i + c] * scale;
      rgba_out[4 * i + c] = unorm8(gamma ? sqrtf(v) : v);
    }
    rgba_out[4 * i + 3] = 255;
  }
}

/* shader.frag:387-404: the temporal running mean the reference keeps in its RGBA8 ping-pong
 * textures (src/webgl.rs:186-204).  pixel = this frame's gamma-encoded colour. */
ORA_API void ora_blend_rgba8(const float* accum, size_t n_pixels, uint32_t total_spp,
                             const PtParams* p, const uint8_t* prev, uint8_t* out) {
  float scale = 1.0f / (float)total_spp;
  float render_count = (float)p->render_count;
  for (size_t i = 0; i < n_pixels; i++) {
    float px[3];
    for (int c = 0; c < 3; c++) px[c] = sqrtf(accum[4 * i + c] * scale);
    float pa = (float)prev[4 * i + 3] / 255.0f;
    if (p->should_average && !(pa == 0.0f || p->render_count <= 1)) {
      float total_frames = render_count + p->last_frame_weight;
      for (int c = 0; c < 3; c++) {
        float pr = (float)prev[4 * i + c] / 255.0f;
        float merged = fmaf(px[c], p->last_frame_weight, pr * render_count) / total_frames;
        out[4 * i + c] = unorm8(merged);
      }
    } else {
      for (int c = 0; c < 3; c++) out[4 * i + c] = unorm8(px[c]);
    }
    out[4 * i + 3] = 255;
  }
}

/* ============================ host camera: src/state.rs:319-347 =========================== */

typedef struct { double x, y, z; } d3;
static inline d3 D(double x, double y, double z) { d3 r = {x, y, z}; return r; }
static inline d3 dsub(d3 a, d3 b) { return D(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline d3 dadd(d3 a, d3 b) { return D(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline d3 dscale(d3 a, double s) { return D(a.x * s, a.y * s, a.z * s); }
static inline d3 ddiv(d3 a, double s) { return D(a.x / s, a.y / s, a.z / s); }
/* src/math.rs:56-66 */
static inline double ddot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline d3 dcross(d3 a, d3 b) {
  return D(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* src/math.rs:44-54, :68-70: length = sqrt(x^2+y^2+z^2); normalize = v / length */
static inline d3 dnormalize(d3 a) { return ddiv(a, sqrt(a.x * a.x + a.y * a.y + a.z * a.z)); }

static void basis_to_params(uint32_t w, uint32_t h, d3 origin, d3 u, d3 v, d3 wv, double fov,
                            double focus, double aperture, PtParams* out) {
  double aspect = (double)w / (double)h;            /* :323 */
  double camera_h = tan(fov / 2.0);                  /* :324 */
  double viewport_height = 2.0 * camera_h;           /* :334 */
  double viewport_width = viewport_height * aspect;  /* :335 */
  d3 horizontal = dscale(u, focus * viewport_width); /* :336 focus*vw*u */
  d3 vertical = dscale(v, focus * viewport_height);  /* :337 */
  d3 llc = dsub(dsub(dsub(origin, ddiv(horizontal, 2.0)), ddiv(vertical, 2.0)),
                dscale(wv, focus));                  /* :338-341 */
  out->width = w; out->height = h;
  out->camera_origin[0] = (float)origin.x; out->camera_origin[1] = (float)origin.y;
  out->camera_origin[2] = (float)origin.z;
  out->horizontal[0] = (float)horizontal.x; out->horizontal[1] = (float)horizontal.y;
  out->horizontal[2] = (float)horizontal.z;
  out->vertical[0] = (float)vertical.x; out->vertical[1] = (float)vertical.y;
  out->vertical[2] = (float)vertical.z;
  out->lower_left_corner[0] = (float)llc.x; out->lower_left_corner[1] = (float)llc.y;
  out->lower_left_corner[2] = (float)llc.z;
  out->u[0] = (float)u.x; out->u[1] = (float)u.y; out->u[2] = (float)u.z;
  out->v[0] = (float)v.x; out->v[1] = (float)v.y; out->v[2] = (float)v.z;
  out->lens_radius = (float)(aperture / 2.0); /* src/state.rs:102 */
}

ORA_API int ora_camera_from_state(const PtCameraIn* in, PtParams* out) {
  /* src/math.rs:375-377 degrees_to_radians: (degrees * PI) / 180. */
  double yaw = in->yaw_degrees * 3.14159265358979323846 / 180.0;
  double pitch = in->pitch_degrees * 3.14159265358979323846 / 180.0;
  d3 origin = D(in->camera_origin[0], in->camera_origin[1], in->camera_origin[2]);
  d3 front = D(cos(yaw) * cos(pitch), sin(pitch), sin(yaw) * cos(pitch)); /* :325-329 */
  d3 look_at = dadd(origin, front);                                       /* :330 */
  d3 wv = dnormalize(dsub(origin, look_at));                              /* :331 */
  d3 vup = D(in->vup[0], in->vup[1], in->vup[2]);
  d3 u = dnormalize(dcross(vup, wv));                                     /* :332 */
  d3 v = dcross(wv, u);                                                   /* :333 */
  basis_to_params(in->width, in->height, origin, u, v, wv, in->fov_radians, in->focus_distance,
                  in->aperture, out);
  return 0;
}

ORA_API int ora_camera_look_at(const PtLookAtIn* in, PtParams* out) {
  d3 origin = D(in->look_from[0], in->look_from[1], in->look_from[2]);
  d3 look_at = D(in->look_at[0], in->look_at[1], in->look_at[2]);
  d3 wv = dnormalize(dsub(origin, look_at));
  d3 vup = D(in->vup[0], in->vup[1], in->vup[2]);
  d3 u = dnormalize(dcross(vup, wv));
  d3 v = dcross(wv, u);
  basis_to_params(in->width, in->height, origin, u, v, wv, in->vfov_radians, in->focus_distance,
                  in->aperture, out);
  return 0;
}

/* ============================ f64 pick ray: src/glsl.rs:42-82, :213-239 =================== */

ORA_API int ora_center_hit_f64(const PtHostSphere* spheres, uint32_t n, const PtCameraIn* cam,
                               PtCenterHit* out) {
  /* the ray of :216-220 needs the f64 pipeline values, so redo update_pipeline in double */
  double yaw = cam->yaw_degrees * 3.14159265358979323846 / 180.0;
  double pitch = cam->pitch_degrees * 3.14159265358979323846 / 180.0;
  d3 origin = D(cam->camera_origin[0], cam->camera_origin[1], cam->camera_origin[2]);
  d3 front = D(cos(yaw) * cos(pitch), sin(pitch), sin(yaw) * cos(pitch));
  d3 wv = dnormalize(dsub(origin, dadd(origin, front)));
  d3 u = dnormalize(dcross(D(cam->vup[0], cam->vup[1], cam->vup[2]), wv));
  d3 v = dcross(wv, u);
  double vh = 2.0 * tan(cam->fov_radians / 2.0);
  double vw = vh * ((double)cam->width / (double)cam->height);
  d3 horizontal = dscale(u, cam->focus_distance * vw);
  d3 vertical = dscale(v, cam->focus_distance * vh);
  d3 llc = dsub(dsub(dsub(origin, ddiv(horizontal, 2.0)), ddiv(vertical, 2.0)),
                dscale(wv, cam->focus_distance));
  d3 dir = dsub(dadd(dadd(llc, ddiv(horizontal, 2.0)), ddiv(vertical, 2.0)), origin);

  int hit = 0;
  double closest = INFINITY;
  for (uint32_t i = 0; i < n; i++) {
    const PtHostSphere* s = &spheres[i];
    d3 c = D(s->center[0], s->center[1], s->center[2]);
    d3 oc = dsub(origin, c);
    double a = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
    double half_b = ddot(oc, dir);
    double cc = (oc.x * oc.x + oc.y * oc.y + oc.z * oc.z) - s->radius * s->radius;
    double disc = half_b * half_b - a * cc;
    if (disc < 0.0) continue;
    double sq = sqrt(disc);
    double root = (-half_b - sq) / a;
    if (root < 0.0 || closest < root) {
      root = (-half_b + sq) / a;
      if (root < 0.0 || closest < root) continue;
    }
    d3 p = dadd(origin, dscale(dir, root));
    d3 on = ddiv(dsub(p, c), s->radius);
    int front_face = ddot(dir, on) < 0.0;
    if (!front_face) on = D(-on.x, -on.y, -on.z);
    hit = 1; closest = root;
    out->t = root; out->uuid = s->uuid; out->front_face = front_face;
    out->hit_point[0] = p.x; out->hit_point[1] = p.y; out->hit_point[2] = p.z;
    out->normal[0] = on.x; out->normal[1] = on.y; out->normal[2] = on.z;
  }
  return hit;
}
