// pt_host.cpp — the host-only entry points of include/ptrace.h (no HIP calls in this file):
// camera derivation, the reference's built-in scene, the f32 narrowing, the f64 pick ray.
#include "pt_host.hpp"

#include "pt_bvh.hpp"
#include "pt_grid.hpp"

#include <new>

#define PT_API extern "C" __attribute__((visibility("default")))

namespace {

pt::State state_from_camera_in(const PtCameraIn& in) {
  pt::State s;
  s.width = in.width;
  s.height = in.height;
  s.camera_origin = pt::Point(in.camera_origin[0], in.camera_origin[1], in.camera_origin[2]);
  s.yaw = in.yaw_degrees;
  s.pitch = in.pitch_degrees;
  s.vup = pt::Vec3(in.vup[0], in.vup[1], in.vup[2]);
  s.camera_field_of_view = in.fov_radians;
  s.focus_distance = in.focus_distance;
  s.aperture = in.aperture;
  s.lens_radius = in.aperture / 2.0; // src/state.rs:102
  s.recompute();
  return s;
}

void camera_members_to_params(const pt::State& s, PtParams* out) {
  out->width = s.width;
  out->height = s.height;
  pt::put3(out->camera_origin, s.camera_origin);
  pt::put3(out->horizontal, s.horizontal);
  pt::put3(out->vertical, s.vertical);
  pt::put3(out->lower_left_corner, s.lower_left_corner);
  pt::put3(out->u, s.u);
  pt::put3(out->v, s.v);
  out->lens_radius = (float)s.lens_radius;
}

} // namespace

PT_API int pt_camera_from_state(const PtCameraIn* in, PtParams* out) {
  if (!in || !out || in->width == 0 || in->height == 0) return PT_ERR_INVALID;
  pt::State s = state_from_camera_in(*in);
  camera_members_to_params(s, out);
  return PT_OK;
}

// Same pipeline with (look_from, look_at) instead of (origin, yaw, pitch): w = normalize(origin
// - look_at) is what src/state.rs:330-331 computes from origin + camera_front.
PT_API int pt_camera_look_at(const PtLookAtIn* in, PtParams* out) {
  if (!in || !out || in->width == 0 || in->height == 0) return PT_ERR_INVALID;
  pt::State s;
  s.width = in->width;
  s.height = in->height;
  s.camera_origin = pt::Point(in->look_from[0], in->look_from[1], in->look_from[2]);
  pt::Point look_at(in->look_at[0], in->look_at[1], in->look_at[2]);
  s.vup = pt::Vec3(in->vup[0], in->vup[1], in->vup[2]);
  s.camera_field_of_view = in->vfov_radians;
  s.focus_distance = in->focus_distance;
  s.aperture = in->aperture;
  s.lens_radius = in->aperture / 2.0;
  s.aspect_ratio = (double)s.width / (double)s.height;
  double camera_h = std::tan(s.camera_field_of_view / 2.);
  s.w = pt::normalize(s.camera_origin - look_at);
  s.u = pt::normalize(pt::cross(s.vup, s.w));
  s.v = pt::cross(s.w, s.u);
  s.viewport_height = 2. * camera_h;
  s.viewport_width = s.viewport_height * s.aspect_ratio;
  s.horizontal = s.focus_distance * s.viewport_width * s.u;
  s.vertical = s.focus_distance * s.viewport_height * s.v;
  s.lower_left_corner =
      s.camera_origin - s.horizontal / 2. - s.vertical / 2. - s.focus_distance * s.w;
  camera_members_to_params(s, out);
  return PT_OK;
}

PT_API int pt_default_camera(uint32_t width, uint32_t height, PtCameraIn* out) {
  if (!out || width == 0 || height == 0) return PT_ERR_INVALID;
  pt::State s = pt::State::default_for(width, height);
  out->width = width;
  out->height = height;
  out->camera_origin[0] = s.camera_origin.x;
  out->camera_origin[1] = s.camera_origin.y;
  out->camera_origin[2] = s.camera_origin.z;
  out->yaw_degrees = s.yaw;
  out->pitch_degrees = s.pitch;
  out->vup[0] = s.vup.x; out->vup[1] = s.vup.y; out->vup[2] = s.vup.z;
  out->fov_radians = s.camera_field_of_view;
  out->focus_distance = s.focus_distance;
  out->aperture = s.aperture;
  return PT_OK;
}

PT_API int pt_default_scene(PtHostSphere* out, uint32_t cap) {
  std::vector<pt::Sphere> v = pt::State::default_spheres();
  pt::set_sphere_uuids(v);
  if (out) {
    for (uint32_t i = 0; i < v.size() && i < cap; i++) {
      const pt::Sphere& s = v[i];
      PtHostSphere& o = out[i];
      o.center[0] = s.center.x; o.center[1] = s.center.y; o.center[2] = s.center.z;
      o.radius = s.radius;
      o.type = (int32_t)s.material.material_type;
      o.uuid = s.uuid;
      o.albedo[0] = s.material.albedo.x; o.albedo[1] = s.material.albedo.y;
      o.albedo[2] = s.material.albedo.z;
      o.fuzz = s.material.fuzz;
      o.refraction_index = s.material.refraction_index;
    }
  }
  return (int)v.size();
}

static pt::Sphere from_host(const PtHostSphere& h) {
  pt::Sphere s;
  s.center = pt::Vec3(h.center[0], h.center[1], h.center[2]);
  s.radius = h.radius;
  s.material.material_type = (pt::MaterialType)h.type;
  s.material.albedo = pt::Vec3(h.albedo[0], h.albedo[1], h.albedo[2]);
  s.material.fuzz = h.fuzz;
  s.material.refraction_index = h.refraction_index;
  s.uuid = h.uuid;
  return s;
}

// the hierarchy of PT_GEOM_BVH exactly as pt_set_spheres (pt_api.hip) builds and uploads it
static int build_grid_export(bool runs, const PtSphere* s, uint32_t n, uint32_t* counts8, float* geom12, float* margin4,
                         float* delta_g, uint32_t* cells, size_t n_cells, float* entries, size_t entry_floats,
                         uint32_t* entry_index, size_t n_index) {
  if (!s && n) return PT_ERR_INVALID;
  std::vector<float> geom((size_t)n * 4), radii(n);
  bool regular = true;
  for (uint32_t i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) {
      regular = regular && (std::fabs(s[i].center[k]) < 1e15f);
      geom[4 * (size_t)i + k] = s[i].center[k];
    }
    regular = regular && (std::fabs(s[i].radius) < 1e15f);
    geom[4 * (size_t)i + 3] = s[i].radius * s[i].radius;
    radii[i] = s[i].radius;
  }
  ptgrid::Grid g;
  if (!regular || !ptgrid::build(geom.data(), radii.data(), n, &g)) return PT_ERR_NOT_READY;
  if (runs && !ptgrid::morton_runs(&g)) return PT_ERR_CAPACITY;
  if (counts8) {
    counts8[0] = g.n[0]; counts8[1] = g.n[1]; counts8[2] = g.n[2]; counts8[3] = g.n_cell_entries;
    counts8[4] = g.n_always; counts8[5] = g.n_entries; counts8[6] = g.max_cell_entries; counts8[7] = g.nonempty;
  }
  if (geom12)
    for (int k = 0; k < 3; k++) { geom12[k] = g.lo[k]; geom12[3 + k] = g.h[k]; geom12[6 + k] = g.hi[k]; geom12[9 + k] = g.c0[k]; }
  if (margin4) { margin4[0] = g.s0; margin4[1] = g.rmin; margin4[2] = g.rmax; margin4[3] = g.d_near; }
  if (delta_g) *delta_g = g.delta_g;
  if ((cells && n_cells < g.cells.size()) || (entries && entry_floats < g.entries.size()) ||
      (entry_index && n_index < g.entry_index.size()))
    return PT_ERR_CAPACITY;
  if (cells) std::copy(g.cells.begin(), g.cells.end(), cells);
  if (entries) std::copy(g.entries.begin(), g.entries.end(), entries);
  if (entry_index) std::copy(g.entry_index.begin(), g.entry_index.end(), entry_index);
  return PT_OK;
}

PT_API int pt_build_grid(const PtSphere* s, uint32_t n, uint32_t* counts8, float* geom12, float* margin4,
                         float* delta_g, uint32_t* cells, size_t n_cells, float* entries, size_t entry_floats,
                         uint32_t* entry_index, size_t n_index) {
  return build_grid_export(false, s, n, counts8, geom12, margin4, delta_g, cells, n_cells, entries, entry_floats, entry_index, n_index);
}
// include/ptrace_dev.h: the same grid in the layout the kernels use when the entries are gathered from
// global memory (ptgrid::morton_runs: the cells' runs in Morton order of their cells)
PT_API int pt_build_grid_runs(const PtSphere* s, uint32_t n, uint32_t* counts8, float* geom12, float* margin4,
                              float* delta_g, uint32_t* cells, size_t n_cells, float* entries, size_t entry_floats,
                              uint32_t* entry_index, size_t n_index) {
  return build_grid_export(true, s, n, counts8, geom12, margin4, delta_g, cells, n_cells, entries, entry_floats, entry_index, n_index);
}

// The numbers the grid kernels' ENTRY test and cell look-up use beside geom12 / margin4 of
// pt_build_grid (made by ptgrid::build, copied into the launch arguments by pt_render_passes):
// out10 = {r2_near, lo_n.xyz, hi_n.xyz, inv_h.xyz}.
PT_API int pt_grid_walk_constants(const PtSphere* s, uint32_t n, float* out10) {
  if ((!s && n) || !out10) return PT_ERR_INVALID;
  std::vector<float> geom((size_t)n * 4), radii(n);
  bool regular = true;
  for (uint32_t i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) {
      regular = regular && (std::fabs(s[i].center[k]) < 1e15f);
      geom[4 * (size_t)i + k] = s[i].center[k];
    }
    regular = regular && (std::fabs(s[i].radius) < 1e15f);
    geom[4 * (size_t)i + 3] = s[i].radius * s[i].radius;
    radii[i] = s[i].radius;
  }
  ptgrid::Grid g;
  if (!regular || !ptgrid::build(geom.data(), radii.data(), n, &g)) return PT_ERR_NOT_READY;
  out10[0] = g.r2_near;
  for (int k = 0; k < 3; k++) { out10[1 + k] = g.lo_n[k]; out10[4 + k] = g.hi_n[k]; out10[7 + k] = g.inv_h[k]; }
  return PT_OK;
}

PT_API int pt_build_bvh(const PtSphere* s, uint32_t n, float* nodes, size_t node_floats, float* slots,
                        size_t slot_floats, uint32_t* slot_index, size_t n_index, float* margin4,
                        uint32_t* counts5, uint32_t* nodes16, size_t n_words16, float* kscale,
                        float* nodes32, size_t n_floats32) {
  if (!s && n) return PT_ERR_INVALID;
  std::vector<float> geom((size_t)n * 4), radii(n);
  bool regular = true;
  for (uint32_t i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) {
      regular = regular && (std::fabs(s[i].center[k]) < 1e15f);
      geom[4 * (size_t)i + k] = s[i].center[k];
    }
    regular = regular && (std::fabs(s[i].radius) < 1e15f);
    geom[4 * (size_t)i + 3] = s[i].radius * s[i].radius;
    radii[i] = s[i].radius;
  }
  ptbvh::Bvh b;
  if (!regular || !ptbvh::build(geom.data(), radii.data(), n, &b)) return PT_ERR_NOT_READY;
  if (counts5) {
    counts5[0] = b.n_nodes; counts5[1] = b.n_slots; counts5[2] = b.n_tree_slots;
    counts5[3] = b.n_outliers; counts5[4] = b.depth;
  }
  if (margin4) { for (int k = 0; k < 3; k++) margin4[k] = b.c0[k]; margin4[3] = b.s0; }
  if ((nodes && node_floats < b.nodes.size()) || (slots && slot_floats < b.slots.size()) ||
      (slot_index && n_index < b.slot_index.size()))
    return PT_ERR_CAPACITY;
  if (nodes16 && n_words16 < b.nodes16.size()) return PT_ERR_CAPACITY;
  if (nodes32 && n_floats32 < b.nodes32.size()) return PT_ERR_CAPACITY;
  if (nodes32) std::copy(b.nodes32.begin(), b.nodes32.end(), nodes32);
  if (nodes16) std::copy(b.nodes16.begin(), b.nodes16.end(), nodes16);
  if (kscale) *kscale = b.kscale;
  if (nodes) std::copy(b.nodes.begin(), b.nodes.end(), nodes);
  if (slots) std::copy(b.slots.begin(), b.slots.end(), slots);
  if (slot_index) std::copy(b.slot_index.begin(), b.slot_index.end(), slot_index);
  return PT_OK;
}

PT_API int pt_narrow_spheres(const PtHostSphere* in, uint32_t n, PtSphere* out) {
  if ((!in || !out) && n) return PT_ERR_INVALID;
  for (uint32_t i = 0; i < n; i++) out[i] = pt::narrow(from_host(in[i]));
  return PT_OK;
}

PT_API int pt_set_sphere_uuids(PtHostSphere* spheres, uint32_t n) {
  if (!spheres && n) return PT_ERR_INVALID;
  for (uint32_t i = 0; i < n; i++) spheres[i].uuid = (int32_t)i;
  return PT_OK;
}

PT_API int pt_center_hit(const PtHostSphere* spheres, uint32_t n, const PtCameraIn* cam,
                         PtCenterHit* out) {
  if (!cam || !out || (!spheres && n) || cam->width == 0 || cam->height == 0)
    return PT_ERR_INVALID;
  pt::State s = state_from_camera_in(*cam);
  s.sphere_list.reserve(n);
  for (uint32_t i = 0; i < n; i++) s.sphere_list.push_back(from_host(spheres[i]));
  pt::HitResultData h;
  if (!pt::get_center_hit(s, h)) return 0;
  out->t = h.t;
  out->hit_point[0] = h.hit_point.x; out->hit_point[1] = h.hit_point.y;
  out->hit_point[2] = h.hit_point.z;
  out->normal[0] = h.normal.x; out->normal[1] = h.normal.y; out->normal[2] = h.normal.z;
  out->front_face = h.front_face ? 1 : 0;
  out->uuid = h.uuid;
  return 1;
}

PT_API uint32_t pt_local_rows(uint32_t height, uint32_t band_rows, uint32_t band_index,
                              uint32_t band_count) {
  if (band_count <= 1 || band_rows == 0) return height;
  uint32_t n = 0;
  for (uint32_t y = 0; y < height; y++) n += ((y / band_rows) % band_count == band_index);
  return n;
}

PT_API uint32_t pt_band_row(uint32_t band_rows, uint32_t band_index, uint32_t band_count, uint32_t local_row) {
  if (band_count <= 1 || band_rows == 0) return local_row;
  // local rows come in runs of band_rows: run k of this band is image band k * band_count + band_index
  return (local_row / band_rows * band_count + band_index) * band_rows + local_row % band_rows;
}

PT_API int pt_abi_version(void) { return PT_ABI_VERSION; }

// ---- pt_state: the reference's State behind an opaque handle ----------------------------------
struct pt_state { pt::State s; };

PT_API int pt_state_create(pt_state** out, uint32_t width, uint32_t height) {
  if (!out || width == 0 || height == 0) return PT_ERR_INVALID;
  pt_state* p = new (std::nothrow) pt_state();
  if (!p) return PT_ERR_INVALID;
  p->s = pt::State::default_for(width, height);
  *out = p;
  return PT_OK;
}

PT_API int pt_state_destroy(pt_state* s) {
  if (!s) return PT_ERR_INVALID;
  delete s;
  return PT_OK;
}

static void put3d(double dst[3], const pt::Vec3& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

PT_API int pt_state_get(const pt_state* h, PtStateView* o) {
  if (!h || !o) return PT_ERR_INVALID;
  const pt::State& s = h->s;
  o->width = s.width; o->height = s.height;
  o->samples_per_pixel = s.samples_per_pixel; o->max_depth = s.max_depth;
  o->aspect_ratio = s.aspect_ratio;
  put3d(o->camera_origin, s.camera_origin); put3d(o->camera_front, s.camera_front); put3d(o->vup, s.vup);
  o->yaw = s.yaw; o->pitch = s.pitch; o->camera_field_of_view = s.camera_field_of_view;
  put3d(o->u, s.u); put3d(o->v, s.v); put3d(o->w, s.w);
  o->aperture = s.aperture; o->lens_radius = s.lens_radius; o->focus_distance = s.focus_distance;
  o->viewport_height = s.viewport_height; o->viewport_width = s.viewport_width;
  put3d(o->horizontal, s.horizontal); put3d(o->vertical, s.vertical);
  put3d(o->lower_left_corner, s.lower_left_corner);
  put3d(o->cursor_point, s.cursor_point);
  o->selected_object = s.selected_object;
  o->is_paused = s.is_paused; o->should_average = s.should_average; o->should_render = s.should_render;
  o->even_odd_count = s.even_odd_count; o->render_count = s.render_count;
  o->max_render_count = s.max_render_count;
  o->last_frame_weight = s.last_frame_weight;
  o->n_spheres = (uint32_t)s.sphere_list.size();
  return PT_OK;
}

PT_API int pt_state_set_fov(pt_state* h, double fov) {
  if (!h) return PT_ERR_INVALID;
  h->s.set_fov(fov);
  return PT_OK;
}

PT_API int pt_state_set_camera_angles(pt_state* h, double yaw, double pitch) {
  if (!h) return PT_ERR_INVALID;
  h->s.set_camera_angles(yaw, pitch);
  return PT_OK;
}

PT_API int pt_state_set_camera_origin(pt_state* h, const double origin[3]) {
  if (!h || !origin) return PT_ERR_INVALID;
  pt::Point p(origin[0], origin[1], origin[2]);
  if (p != h->s.camera_origin) h->s.dirty = true;
  h->s.camera_origin = p;
  h->s.update_pipeline();
  return PT_OK;
}

PT_API int pt_state_set_lens(pt_state* h, double aperture, double focus_distance) {
  if (!h) return PT_ERR_INVALID;
  if (aperture != h->s.aperture || focus_distance != h->s.focus_distance) h->s.dirty = true;
  h->s.aperture = aperture;
  h->s.lens_radius = aperture / 2.0; // src/state.rs:102
  h->s.focus_distance = focus_distance;
  h->s.update_pipeline();
  return PT_OK;
}

PT_API int pt_state_set_quality(pt_state* h, uint32_t spp, uint32_t max_depth) {
  if (!h || spp == 0 || max_depth == 0) return PT_ERR_INVALID;
  if (spp != h->s.samples_per_pixel || max_depth != h->s.max_depth) h->s.dirty = true;
  h->s.samples_per_pixel = spp;
  h->s.max_depth = max_depth;
  h->s.update_pipeline();
  return PT_OK;
}

PT_API int pt_state_set_flags(pt_state* h, int is_paused, int should_average, float last_frame_weight) {
  if (!h) return PT_ERR_INVALID;
  h->s.is_paused = is_paused != 0;
  h->s.should_average = should_average != 0;
  h->s.last_frame_weight = last_frame_weight;
  return PT_OK;
}

PT_API int pt_state_set_keys(pt_state* h, uint32_t m) {
  if (!h) return PT_ERR_INVALID;
  pt::KeydownMap& k = h->s.keydown_map;
  k.w = m & 1u; k.a = m & 2u; k.s = m & 4u; k.d = m & 8u; k.space = m & 16u; k.shift = m & 32u;
  return PT_OK;
}

PT_API int pt_state_update_position(pt_state* h, double dt_ms) {
  if (!h) return PT_ERR_INVALID;
  h->s.update_position(dt_ms);
  return PT_OK;
}

PT_API int pt_state_update_render_globals(pt_state* h) {
  if (!h) return PT_ERR_INVALID;
  h->s.update_render_globals();
  return PT_OK;
}

PT_API int pt_state_resize(pt_state* h, uint32_t w, uint32_t hh) {
  if (!h || w == 0 || hh == 0) return PT_ERR_INVALID;
  h->s.resize(w, hh);
  return PT_OK;
}

PT_API int pt_state_should_render(const pt_state* h, int should_save) {
  if (!h) return PT_ERR_INVALID;
  return h->s.frame_should_render(should_save != 0) ? 1 : 0;
}

PT_API int pt_state_set_spheres(pt_state* h, const PtHostSphere* spheres, uint32_t n) {
  if (!h || (!spheres && n)) return PT_ERR_INVALID;
  h->s.sphere_list.clear();
  for (uint32_t i = 0; i < n; i++) h->s.sphere_list.push_back(from_host(spheres[i]));
  pt::set_sphere_uuids(h->s.sphere_list);
  h->s.dirty = true;
  h->s.update_pipeline();
  return PT_OK;
}

PT_API int pt_state_spheres(const pt_state* h, PtSphere* out, uint32_t cap) {
  if (!h) return PT_ERR_INVALID;
  uint32_t n = (uint32_t)h->s.sphere_list.size();
  if (out)
    for (uint32_t i = 0; i < n && i < cap; i++) out[i] = pt::narrow(h->s.sphere_list[i]);
  return (int)n;
}

PT_API int pt_state_to_params(const pt_state* h, double now_ms, PtParams* out) {
  if (!h || !out) return PT_ERR_INVALID;
  pt::to_params(h->s, now_ms, *out);
  return PT_OK;
}

PT_API int pt_adjusted_screen_dimensions(double raw_w, double raw_h, uint32_t* w, uint32_t* hh) {
  if (!w || !hh || !(raw_w > 0) || !(raw_h > 0)) return PT_ERR_INVALID;
  pt::adjusted_screen_dimensions(raw_w, raw_h, *w, *hh);
  return PT_OK;
}
