// bindings/rust/ptrace_sys.rs — SOURCE-ONLY Rust FFI for libptrace.so (include/ptrace.h).
//
// The reference's host code is Rust (src/*.rs); this file is what replaces its `mod webgl`.
// The build image has no Rust toolchain (no cargo/rustc, no network), so this file has never
// been compiled here; the same ABI is exercised by the ctypes binding
// (ray_tracer_webgl_amd/_lib.py) and by every `-m gpu` test.  Keep in sync with the header:
// tests/test_abi.py::test_rust_binding_lists_the_render_abi checks the symbol list.
// src/ptrace_sys.rs — FFI for libptrace.so (include/ptrace.h).  Replaces `mod webgl`.
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct PtSphere {            // == u_sphere_list[i], static/shader.frag:55-61
    pub center: [f32; 3], pub radius: f32,
    pub type_: i32, pub albedo: [f32; 3],
    pub fuzz: f32, pub refraction_index: f32,
    pub uuid: i32, pub _pad: i32,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct PtParams {            // == uniform block, static/shader.frag:79-99
    pub width: u32, pub height: u32,
    pub time: f32, pub samples_per_pixel: i32, pub max_depth: i32,
    pub camera_origin: [f32; 3], pub horizontal: [f32; 3], pub vertical: [f32; 3],
    pub lower_left_corner: [f32; 3], pub u: [f32; 3], pub v: [f32; 3],
    pub lens_radius: f32,
    pub render_count: i32, pub should_average: i32, pub last_frame_weight: f32,
    pub background_mode: i32,
    pub band_rows: u32, pub band_index: u32, pub band_count: u32,
    pub time_step: f32, pub first_pass: u32,
}

#[repr(C)] pub struct PtCtx { _private: [u8; 0] }

#[link(name = "ptrace")]
extern "C" {
    pub fn pt_create(out: *mut *mut PtCtx, device: c_int, width: u32, height: u32) -> c_int;
    // for a host that brings its own hipStream_t: the context never creates a stream (an HSA queue) of its own
    pub fn pt_create_on_stream(out: *mut *mut PtCtx, device: c_int, width: u32, height: u32, hip_stream: *mut c_void) -> c_int;
    pub fn pt_destroy(ctx: *mut PtCtx) -> c_int;
    pub fn pt_resize(ctx: *mut PtCtx, width: u32, height: u32) -> c_int;
    pub fn pt_set_spheres(ctx: *mut PtCtx, spheres: *const PtSphere, n: u32) -> c_int;
    pub fn pt_set_params(ctx: *mut PtCtx, params: *const PtParams) -> c_int;
    pub fn pt_render(ctx: *mut PtCtx) -> c_int;
    pub fn pt_render_passes(ctx: *mut PtCtx, n_passes: u32) -> c_int;
    pub fn pt_reserve_passes(ctx: *mut PtCtx, max_passes: u32) -> c_int;
    pub fn pt_reset_accum(ctx: *mut PtCtx) -> c_int;
    pub fn pt_synchronize(ctx: *mut PtCtx) -> c_int;
    pub fn pt_resolve(ctx: *mut PtCtx, rgba_out: *mut f32, gamma: c_int) -> c_int;
    pub fn pt_resolve_rgba8(ctx: *mut PtCtx, rgba_out: *mut u8, gamma: c_int) -> c_int;
    pub fn pt_blend_rgba8(ctx: *mut PtCtx, prev: *const u8, out: *mut u8) -> c_int;
    pub fn pt_accum_ptr(ctx: *mut PtCtx, dev_ptr: *mut *mut c_void, bytes: *mut usize) -> c_int;
    pub fn pt_bind_accum(ctx: *mut PtCtx, dev_ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn pt_read_accum(ctx: *mut PtCtx, dst: *mut f32, bytes: usize) -> c_int;
    pub fn pt_load_accum(ctx: *mut PtCtx, src: *const f32, bytes: usize) -> c_int;
    pub fn pt_set_stream(ctx: *mut PtCtx, hip_stream: *mut c_void) -> c_int;
    pub fn pt_set_option(ctx: *mut PtCtx, key: c_int, value: c_int) -> c_int;
    pub fn pt_tune(ctx: *mut PtCtx, n_passes: u32) -> c_int;
    // the camera moves every tick (State::update_position, src/state.rs:411-441): does the grid of a large scene still fit
    // it (0 yes / 1 no: refit now / 2 looser than needed), and the rebuild for the margin class the camera needs
    pub fn pt_grid_fit(ctx: *mut PtCtx) -> c_int;
    pub fn pt_refit_grid(ctx: *mut PtCtx, only_if_stale: c_int) -> c_int;
    // the reference's frame on device-resident textures: webgl::render, src/webgl.rs:180-205
    pub fn pt_clear_textures(ctx: *mut PtCtx) -> c_int;
    pub fn pt_render_frame(ctx: *mut PtCtx, even_odd_count: u32) -> c_int;
    pub fn pt_render_frames(ctx: *mut PtCtx, even_odd_count: u32, max_render_count: u32, n_frames: u32) -> c_int;
    pub fn pt_read_canvas(ctx: *mut PtCtx, rgba_out: *mut u8) -> c_int;
    pub fn pt_read_texture(ctx: *mut PtCtx, index: c_int, rgba_out: *mut u8) -> c_int;
    pub fn pt_write_texture(ctx: *mut PtCtx, index: c_int, rgba_in: *const u8) -> c_int;
    pub fn pt_last_error(ctx: *mut PtCtx) -> *const c_char;
    pub fn pt_abi_version() -> c_int;
    pub fn pt_device_count() -> c_int;
    // multi-GPU from one host process: the row arithmetic of the band partition
    pub fn pt_local_rows(height: u32, band_rows: u32, band_index: u32, band_count: u32) -> u32;
    pub fn pt_band_row(band_rows: u32, band_index: u32, band_count: u32, local_row: u32) -> u32;
}

pub const PT_OPT_GEOMETRY_PATH: c_int = 1;      // PT_GEOM_AUTO 0 / LDS 1 / SCALAR 2 / BVH 3 / GRID 4 / SMALL 5
pub const PT_OPT_RUSSIAN_ROULETTE: c_int = 5;   // opt-in, 0 = off: the reference's estimator has none (shader.frag:297-339)

// The rAF closure of src/lib.rs:65-104 with webgl::render replaced (one tick):
//     state::update_render_globals(&mut state);
//     let p = PtParams::from_state(&state, now);            // uniforms.run_setters(now)
//     pt_set_params(ctx, &p);
//     if pt_grid_fit(ctx) == 1 { pt_refit_grid(ctx, 0); }   // scenes of hundreds of spheres: the camera has left the region the
//                                                           // grid was fitted to (host arithmetic; 0 for the reference's 9 spheres)
//     pt_render_frame(ctx, state.even_odd_count);           // webgl::render: trace + blend into the ping-pong textures
//     if should_save { pt_read_canvas(ctx, pixels.as_mut_ptr()); }
// A run of ticks with nothing else happening (no input: no key held; should_average on — without it only the
// first tick draws, src/state.rs:443-447) is one call:
//     p.time_step = frame_interval_ms; pt_set_params(ctx, &p);
//     pt_render_frames(ctx, state.even_odd_count, state.max_render_count, n);   // one hipGraph replayed n times
//     for _ in 1..n { state::update_render_globals(&mut state); }

// State -> uniforms: what Uniforms::run_setters uploads (src/webgl.rs:279-593)
impl PtParams {
    pub fn from_state(s: &crate::state::State, now_ms: f64) -> Self {
        let spp = if s.is_paused { s.samples_per_pixel.max(25) } else { s.samples_per_pixel };
        PtParams {
            width: s.width, height: s.height, time: now_ms as f32,
            samples_per_pixel: spp as i32, max_depth: s.max_depth as i32,
            camera_origin: s.camera_origin.to_array(), horizontal: s.horizontal.to_array(),
            vertical: s.vertical.to_array(), lower_left_corner: s.lower_left_corner.to_array(),
            u: s.u.to_array(), v: s.v.to_array(), lens_radius: s.lens_radius as f32,
            render_count: s.render_count as i32, should_average: s.should_average as i32,
            last_frame_weight: s.last_frame_weight, background_mode: 0,
            band_rows: 8, band_index: 0, band_count: 1,
            time_step: 0.0, first_pass: 0,   // one pass per frame at u_time = now, as the reference renders
        }
    }
}

// Sphere -> u_sphere_list[i]: what webgl::set_geometry uploads (src/webgl.rs:225-274)
impl From<&crate::glsl::Sphere> for PtSphere {
    fn from(s: &crate::glsl::Sphere) -> Self {
        PtSphere {
            center: s.center.to_array(), radius: s.radius as f32,
            type_: s.material.material_type.value(), albedo: s.material.albedo.to_array(),
            fuzz: s.material.fuzz, refraction_index: s.material.refraction_index,
            uuid: s.uuid, _pad: 0,
        }
    }
}
