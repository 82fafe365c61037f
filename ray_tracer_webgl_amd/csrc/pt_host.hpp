// pt_host.hpp — host-side mirror of the reference's scene / camera types, in C++ because the
// reference's host code is compiled Rust and no Rust toolchain exists in the build image.
// Same names, same f64 arithmetic, same update rules as the reference, so scene setup written
// against src/state.rs / src/glsl.rs / src/math.rs / src/ray.rs ports line for line.
//
//   pt::Vec3          src/math.rs:17 (+ ops :133-371, dot :56, cross :60, normalize :68)
//   pt::Ray           src/ray.rs:3-12
//   pt::MaterialType  src/glsl.rs:10-24        pt::Material  src/glsl.rs:27-32
//   pt::Sphere        src/glsl.rs:35-40, hit() :42-82
//   pt::State         src/state.rs:31-94 (camera + render members), Default :96-315,
//                     update_pipeline :319-347, set_fov :349, set_camera_angles :354,
//                     update_position :411-441, update_render_globals :443-450,
//                     update_cursor_position_in_world :453-471
//   pt::get_center_hit   src/glsl.rs:213-239
//   pt::to_params        Uniforms::run_setters, src/webgl.rs:279-593
//   pt::narrow           webgl::set_geometry,   src/webgl.rs:225-274
//
// Built with -ffp-contract=off: Rust never fuses a*b+c, so neither may this file.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/ptrace.h"

namespace pt {

constexpr double PI = 3.14159265358979323846; // std::f64::consts::PI

struct Vec3 {
  double x = 0, y = 0, z = 0;
  constexpr Vec3() = default;
  constexpr Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  double length_squared() const { return x * x + y * y + z * z; } // math.rs:48-50
  double length() const { return std::sqrt(length_squared()); }   // math.rs:44-46
};
using Point = Vec3;

inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(const Vec3& a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(const Vec3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(double s, const Vec3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator*(const Vec3& a, const Vec3& b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator/(const Vec3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3& operator+=(Vec3& a, const Vec3& b) { a = a + b; return a; }
inline Vec3& operator-=(Vec3& a, const Vec3& b) { a = a - b; return a; }
inline bool operator==(const Vec3& a, const Vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(const Vec3& a, const Vec3& b) { return !(a == b); }
inline double dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(const Vec3& a, const Vec3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline Vec3 normalize(const Vec3& a) { return a / a.length(); }
inline double degrees_to_radians(double d) { return d * PI / 180.0; } // math.rs:375-377

struct Ray {
  Point origin;
  Vec3 direction;
  Point at(double t) const { return origin + direction * t; } // ray.rs:9-11
};

enum class MaterialType : int32_t { Diffuse = 0, Metal = 1, Glass = 2, Emissive = 3 /*ext*/ };

struct Material {
  MaterialType material_type = MaterialType::Diffuse;
  Vec3 albedo;
  float fuzz = 0.f;
  float refraction_index = 0.f;
};

struct HitResultData { // glsl.rs:96-103
  Point hit_point;
  Vec3 normal;
  double t = 0;
  bool front_face = false;
  int32_t uuid = 0;
};

struct Sphere {
  Vec3 center;
  double radius = 0;
  Material material;
  int32_t uuid = 0;

  // glsl.rs:42-82.  Returns true and fills `out` on a hit in [t_min, t_max].
  bool hit(const Ray& ray, double t_min, double t_max, HitResultData& out) const {
    Vec3 oc = ray.origin - center;
    double a = ray.direction.length_squared();
    double half_b = dot(oc, ray.direction);
    double c = oc.length_squared() - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.) return false;
    double sqrt_discriminant = std::sqrt(discriminant);
    double root = (-half_b - sqrt_discriminant) / a;
    if (root < t_min || t_max < root) {
      root = (-half_b + sqrt_discriminant) / a;
      if (root < t_min || t_max < root) return false;
    }
    Point hit_point = ray.at(root);
    Vec3 outward_normal = (hit_point - center) / radius;
    out.t = root;
    out.hit_point = hit_point;
    out.front_face = dot(ray.direction, outward_normal) < 0.; // glsl.rs:137-145
    out.normal = out.front_face ? outward_normal : -outward_normal;
    out.uuid = uuid;
    return true;
  }
};

inline void set_sphere_uuids(std::vector<Sphere>& spheres) { // glsl.rs:84-88
  for (size_t i = 0; i < spheres.size(); i++) spheres[i].uuid = (int32_t)i;
}

constexpr double MOVEMENT_SPEED = 0.001;     // state.rs:9
constexpr int32_t NO_SELECTED_OBJECT_ID = 1000; // state.rs:12
constexpr uint32_t MAX_CANVAS_SIZE = 1280;   // dom.rs:13

struct KeydownMap { // state.rs:14-28
  bool w = false, a = false, s = false, d = false, space = false, shift = false;
  bool all_false() const { return !w && !a && !s && !d && !space && !shift; }
};

struct State {
  uint32_t width = 0, height = 0;
  double aspect_ratio = 1;
  uint32_t samples_per_pixel = 1;
  uint32_t max_depth = 8;
  double focal_length = 1;
  Point camera_origin;
  double pitch = 0, yaw = -90;
  Point camera_front;
  Vec3 vup{0, 1, 0};
  double camera_field_of_view = PI / 3.;
  Vec3 u, v, w;
  double aperture = 0, lens_radius = 0, focus_distance = 0.75;
  double viewport_height = 0, viewport_width = 0;
  Vec3 horizontal, vertical;
  Point lower_left_corner;
  std::vector<Sphere> sphere_list;

  bool is_paused = true;
  bool should_average = true;
  bool should_render = true;
  uint32_t even_odd_count = 0;
  uint32_t render_count = 0;
  float last_frame_weight = 1.f;
  uint32_t max_render_count = 100000;

  KeydownMap keydown_map;
  int32_t enable_debugging = 0;
  Point cursor_point;
  int32_t selected_object = NO_SELECTED_OBJECT_ID;

  // State::default, state.rs:96-315 — (width,height) come from
  // dom::get_adjusted_screen_dimensions in the reference; here the caller passes them.
  static State default_for(uint32_t width, uint32_t height) {
    State s;
    s.width = width;
    s.height = height;
    s.aperture = 0.;
    s.focus_distance = 0.75;
    s.lens_radius = s.aperture / 2.0; // :102 — set here only, never by update_pipeline
    s.camera_field_of_view = PI / 3.;
    s.camera_origin = Point(0., 0., 1.);
    s.pitch = 0.;
    s.yaw = -90.;
    s.vup = Vec3(0., 1., 0.);
    s.samples_per_pixel = 1;
    s.max_depth = 8;
    s.sphere_list = default_spheres();
    set_sphere_uuids(s.sphere_list);
    s.recompute();
    return s;
  }

  static std::vector<Sphere> default_spheres() { // state.rs:148-257
    auto mk = [](Vec3 c, double r, MaterialType t, Vec3 alb, float fuzz, float ri) {
      Sphere s;
      s.center = c; s.radius = r;
      s.material.material_type = t; s.material.albedo = alb;
      s.material.fuzz = fuzz; s.material.refraction_index = ri;
      return s;
    };
    using MT = MaterialType;
    return {
        mk({0., -100.5, -1.}, 100., MT::Diffuse, {0.75, 0.6, 0.5}, 0.f, 0.f),   // ground
        mk({0., 0., -1.}, 0.5, MT::Diffuse, {0.3, 0.3, 0.4}, 0.f, 0.f),        // centre
        mk({-1.1, 0., -1.}, 0.5, MT::Metal, {1.0, 1.0, 1.0}, 0.f, 0.f),        // left
        mk({1.1, 0., -1.}, 0.5, MT::Glass, {1.0, 1.0, 1.0}, 0.f, 1.5f),        // right
        mk({-0.5, -0.35, -0.55}, -0.15, MT::Metal, {1.0, 1.0, 1.0}, 0.f, 0.f), // back left
        mk({-0.75, -0.4, -0.35}, -0.1, MT::Metal, {1.0, 1.0, 1.0}, 0.f, 0.f),  // front left
        mk({0., 1.2, 4.}, 2., MT::Diffuse, {1.0, 0.8, 0.8}, 0.f, 0.f),         // behind
        mk({150., 20., -500.}, 100., MT::Diffuse, {0.95, 0.95, 1.0}, 0.f, 0.f), // moon
        mk({170., -20., -350.}, 30., MT::Diffuse, {1.0, 1.0, 1.0}, 0.f, 0.f),  // moon's moon
    };
  }

  // the arithmetic of update_pipeline, state.rs:323-341 (also State::default :99-125)
  void recompute() {
    aspect_ratio = (double)width / (double)height;
    double camera_h = std::tan(camera_field_of_view / 2.);
    camera_front = Point(std::cos(degrees_to_radians(yaw)) * std::cos(degrees_to_radians(pitch)),
                         std::sin(degrees_to_radians(pitch)),
                         std::sin(degrees_to_radians(yaw)) * std::cos(degrees_to_radians(pitch)));
    Point look_at = camera_origin + camera_front;
    w = normalize(camera_origin - look_at);
    u = normalize(cross(vup, w));
    v = cross(w, u);
    viewport_height = 2. * camera_h;
    viewport_width = viewport_height * aspect_ratio;
    horizontal = focus_distance * viewport_width * u;
    vertical = focus_distance * viewport_height * v;
    lower_left_corner = camera_origin - horizontal / 2. - vertical / 2. - focus_distance * w;
  }

  // state.rs:319-347: recompute; any change restarts accumulation
  void update_pipeline() {
    Vec3 pu = u, pv = v, pw = w, ph = horizontal, pve = vertical, pl = lower_left_corner;
    double pa = aspect_ratio;
    recompute();
    if (pu != u || pv != v || pw != w || ph != horizontal || pve != vertical ||
        pl != lower_left_corner || pa != aspect_ratio || dirty) {
      render_count = 0;
      should_render = true;
      dirty = false;
    }
  }

  void set_fov(double new_fov_radians) { // state.rs:349-352
    double f = new_fov_radians < 0.0001 ? 0.0001 : (new_fov_radians > PI * 0.75 ? PI * 0.75 : new_fov_radians);
    if (f != camera_field_of_view) dirty = true;
    camera_field_of_view = f;
    update_pipeline();
  }

  void set_camera_angles(double new_yaw, double new_pitch) { // state.rs:354-358
    double p = new_pitch < -89. ? -89. : (new_pitch > 89. ? 89. : new_pitch);
    if (new_yaw != yaw || p != pitch) dirty = true;
    yaw = new_yaw;
    pitch = p;
    update_pipeline();
  }

  void update_render_globals() { // state.rs:443-450
    if (!should_average) should_render = false;
    even_odd_count += 1;
    render_count = render_count + 1 < max_render_count ? render_count + 1 : max_render_count;
  }

  // src/lib.rs:77-82
  bool frame_should_render(bool should_save = false) const {
    return (should_render && !is_paused) || (should_render && is_paused && should_save) ||
           (should_render && is_paused && !should_save && render_count == 0);
  }
  // state.rs:364-398 (the State half of update_render_dimensions_to_match_window)
  void resize(uint32_t w, uint32_t h) {
    if (w != width || h != height) dirty = true;
    width = w; height = h;
    update_pipeline();
  }
  void update_position(double dt); // state.rs:411-441, defined below
  void update_cursor_position_in_world(); // state.rs:453-471

  bool dirty = false; // stands in for the `self != &prev_state` whole-struct compare (:343)
};

// glsl.rs:213-239
inline bool get_center_hit(const State& state, HitResultData& out) {
  Ray ray{state.camera_origin, state.lower_left_corner + state.horizontal / 2. +
                                   state.vertical / 2. - state.camera_origin};
  bool any = false;
  double closest_so_far = std::numeric_limits<double>::infinity();
  for (const Sphere& s : state.sphere_list) {
    HitResultData h;
    if (s.hit(ray, 0., closest_so_far, h)) {
      closest_so_far = h.t;
      out = h;
      any = true;
    }
  }
  return any;
}

inline void State::update_cursor_position_in_world() {
  HitResultData data;
  if (get_center_hit(*this, data)) {
    double distance = (data.hit_point - camera_origin).length();
    if (aperture > 0.) { if (focus_distance != distance) dirty = true; focus_distance = distance; }
    cursor_point = data.hit_point;
    selected_object = data.uuid;
  } else {
    if (aperture > 0.) { if (focus_distance != 10.) dirty = true; focus_distance = 10.; }
    cursor_point = Point(0., 0., 0.);
    selected_object = NO_SELECTED_OBJECT_ID;
  }
  update_pipeline();
}

inline void State::update_position(double dt) {
  if (keydown_map.all_false()) return;
  Point before = camera_origin;
  Vec3 front = camera_front, up = vup;
  double fov = camera_field_of_view;
  if (keydown_map.w) camera_origin += front * MOVEMENT_SPEED * dt * fov;
  if (keydown_map.a) camera_origin -= cross(front, up) * MOVEMENT_SPEED * dt * fov;
  if (keydown_map.s) camera_origin -= front * MOVEMENT_SPEED * dt * fov;
  if (keydown_map.d) camera_origin += cross(front, up) * MOVEMENT_SPEED * dt * fov;
  if (keydown_map.space) camera_origin += up * MOVEMENT_SPEED * dt * fov;
  if (keydown_map.shift) camera_origin -= up * MOVEMENT_SPEED * dt * fov;
  if (before != camera_origin) dirty = true;
  update_cursor_position_in_world();
  update_pipeline();
}

// dom.rs:277-291 get_adjusted_screen_dimensions, literally (including its quirk: the portrait
// branch clamps the raw WIDTH, not the height)
inline void adjusted_screen_dimensions(double raw_w, double raw_h, uint32_t& w, uint32_t& h) {
  double aspect_ratio = raw_w / raw_h;
  if (raw_w > raw_h) {
    double adjusted_width = raw_w < (double)MAX_CANVAS_SIZE ? raw_w : (double)MAX_CANVAS_SIZE;
    double adjusted_height = adjusted_width / aspect_ratio;
    w = (uint32_t)adjusted_width; h = (uint32_t)adjusted_height;
  } else {
    double adjusted_height = raw_w < (double)MAX_CANVAS_SIZE ? raw_w : (double)MAX_CANVAS_SIZE;
    double adjusted_width = adjusted_height * aspect_ratio;
    w = (uint32_t)adjusted_width; h = (uint32_t)adjusted_height;
  }
}

inline void put3(float dst[3], const Vec3& v) { // Vec3::to_array, math.rs:107-109
  dst[0] = (float)v.x; dst[1] = (float)v.y; dst[2] = (float)v.z;
}

// Uniforms::run_setters (webgl.rs:279-593): State -> the uniform block.  `now_ms` is u_time.
inline void to_params(const State& s, double now_ms, PtParams& p) {
  p.width = s.width;
  p.height = s.height;
  p.time = (float)now_ms; // webgl.rs:320-331
  uint32_t spp = s.is_paused ? (s.samples_per_pixel > 25 ? s.samples_per_pixel : 25) : s.samples_per_pixel;
  p.samples_per_pixel = (int32_t)spp; // webgl.rs:342-346
  p.max_depth = (int32_t)s.max_depth;
  put3(p.camera_origin, s.camera_origin);
  put3(p.horizontal, s.horizontal);
  put3(p.vertical, s.vertical);
  put3(p.lower_left_corner, s.lower_left_corner);
  put3(p.u, s.u);
  put3(p.v, s.v);
  p.lens_radius = (float)s.lens_radius;
  p.render_count = (int32_t)s.render_count;
  p.should_average = s.should_average ? 1 : 0;
  p.last_frame_weight = s.last_frame_weight;
  // build extensions the reference has no counterpart for: sky background, all rows, whole-number pass times
  p.background_mode = PT_BG_SKY;
  p.band_rows = 0; p.band_index = 0; p.band_count = 1;
  p.time_step = 0.0f; p.first_pass = 0;
}

// webgl::set_geometry narrowing (webgl.rs:232-272)
inline PtSphere narrow(const Sphere& s) {
  PtSphere o{};
  put3(o.center, s.center);
  o.radius = (float)s.radius;
  o.type = (int32_t)s.material.material_type;
  put3(o.albedo, s.material.albedo);
  o.fuzz = s.material.fuzz;
  o.refraction_index = s.material.refraction_index;
  o.uuid = s.uuid;
  return o;
}

} // namespace pt
