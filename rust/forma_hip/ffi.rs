//! `forma/src/hip/ffi.rs` — the C ABI of `libforma_hip.so`, declaration for declaration the same as
//! `include/forma_hip.h` (struct layouts are checked against that header by `tests/test_abi_and_host.py` through the
//! ctypes mirror `forma_amd/_lib.py`; the Rust `#[repr(C)]` structs below have the same field order and widths).
//!
//! Not compiled in this repository (no Rust toolchain in the build image); see `rust/forma_hip/README.md`.

#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

pub const FORMA_NONE: u32 = 0xFFFF_FFFF;

pub const FORMA_OK: c_int = 0;
pub const FORMA_E_ARG: c_int = -1;
pub const FORMA_E_HIP: c_int = -2;
pub const FORMA_E_NO_DEVICE: c_int = -3;
pub const FORMA_E_CAPACITY: c_int = -4;
pub const FORMA_E_STATE: c_int = -5;
pub const FORMA_E_INTERNAL: c_int = -6;
pub const FORMA_E_COMM: c_int = -7;

pub const FORMA_GEOM_HAS_XF: u32 = 1;

pub const FORMA_FILL_SOLID: u32 = 0;
pub const FORMA_FILL_LINEAR: u32 = 1;
pub const FORMA_FILL_RADIAL: u32 = 2;
pub const FORMA_FILL_TEXTURE: u32 = 3;

pub const FORMA_CH_RED: u8 = 0;
pub const FORMA_CH_GREEN: u8 = 1;
pub const FORMA_CH_BLUE: u8 = 2;
pub const FORMA_CH_ALPHA: u8 = 3;
pub const FORMA_CH_ZERO: u8 = 4;
pub const FORMA_CH_ONE: u8 = 5;

/// Opaque `forma_hip_ctx`.
#[repr(C)]
pub struct forma_hip_ctx {
    _private: [u8; 0],
}

/// `forma_geom_t`: one entry per geometry slot (dense stand-in of a `GeomId`).
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct forma_geom_t {
    pub order: u32,
    pub flags: u32,
    pub xf: [f32; 6],
}

impl forma_geom_t {
    pub const HIDDEN: Self = Self {
        order: FORMA_NONE,
        flags: 0,
        xf: [0.0; 6],
    };
}

/// `forma_image_t`.
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct forma_image_t {
    pub texel_offset: u64,
    pub width: u32,
    pub height: u32,
}

/// `forma_rect_t`, in pixels.
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct forma_rect_t {
    pub x0: u32,
    pub x1: u32,
    pub y0: u32,
    pub y1: u32,
}

/// `forma_timings_t`.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct forma_timings_t {
    pub prepare_us: f32,
    pub rasterize_us: f32,
    pub sort_us: f32,
    pub sort_pass_us: f32,
    pub carry_us: f32,
    pub paint_us: f32,
    pub total_us: f32,
    pub d2h_us: f32,
    pub n_lines: u32,
    pub n_segments: u32,
    pub n_sort_passes: u32,
    pub n_runs: u32,
    pub n_tile_entries: u32,
    pub n_tiles_written: u32,
    pub exchange_us: f32,
}

/// `forma_flatten_tables_t`.
#[repr(C)]
pub struct forma_flatten_tables_t {
    pub point_commands: *const u32,
    pub point_indices: *const u32,
    pub quad_indices: *const u32,
    pub n_points: usize,
    pub qx: *const f32,
    pub qy: *const f32,
    pub qw: *const f32,
    pub x0: *const f32,
    pub dx_recip: *const f32,
    pub k0: *const f32,
    pub dk: *const f32,
    pub curvatures_recip: *const f32,
    pub partial_spline: *const u32,
    pub partial_curv: *const f32,
    pub n_quads: usize,
    pub sp0x: *const f32,
    pub sp0y: *const f32,
    pub sp2x: *const f32,
    pub sp2y: *const f32,
    pub n_splines: usize,
}

pub const FORMA_MAX_DEVICES: usize = 8;
pub const FORMA_TRANSPORT_NONE: u32 = 0;
pub const FORMA_TRANSPORT_RCCL: u32 = 1;
pub const FORMA_TRANSPORT_COPY: u32 = 2;
pub const FORMA_LAYOUT_AUTO: c_int = 0;
pub const FORMA_LAYOUT_EXCHANGE: c_int = 1;
pub const FORMA_LAYOUT_BANDS: c_int = 2;

/// `forma_context_info_t` (`include/forma_hip.h`): devices, frame slots and the exchange transport of a context.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct forma_context_info_t {
    pub n_devices: u32,
    pub frames_in_flight: u32,
    pub transport: u32,
    pub layout: u32,
    pub devices: [i32; FORMA_MAX_DEVICES],
}

/// `forma_kernel_time_t` (`include/forma_hip.h`): one kernel of the last timed frame, timed by its own launch events.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct forma_kernel_time_t {
    pub name: [c_char; 48],
    pub start_us: f32,
    pub us: f32,
    pub stage: u32,
    pub reserved: u32,
}

pub const FORMA_SORT_MAX_PASSES: usize = 12;

/// `forma_sort_plan_t` (`include/forma_hip.h`): the digit plan of a frame's segment sort (introspection, host logic only).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct forma_sort_plan_t {
    pub n_passes: u32,
    pub biased: u32,
    pub shift: [u32; FORMA_SORT_MAX_PASSES],
    pub mask: [u32; FORMA_SORT_MAX_PASSES],
    pub bias: [u32; FORMA_SORT_MAX_PASSES],
}

#[link(name = "forma_hip")]
extern "C" {
    // lifetime
    pub fn forma_hip_create(out: *mut *mut forma_hip_ctx, device: c_int) -> c_int;
    pub fn forma_hip_create_multi(out: *mut *mut forma_hip_ctx, devices: *const c_int, n: c_int) -> c_int;
    pub fn forma_hip_destroy(ctx: *mut forma_hip_ctx);
    pub fn forma_hip_last_error(ctx: *const forma_hip_ctx) -> *const c_char;
    pub fn forma_hip_version() -> *const c_char;

    // scene upload
    pub fn forma_hip_set_geometry(
        ctx: *mut forma_hip_ctx,
        x: *const f32,
        y: *const f32,
        line_slot: *const u32,
        n_points: usize,
    ) -> c_int;
    pub fn forma_hip_set_geoms(ctx: *mut forma_hip_ctx, geoms: *const forma_geom_t, n_geoms: usize) -> c_int;
    pub fn forma_hip_set_styles(
        ctx: *mut forma_hip_ctx,
        style_offsets: *const u32,
        n_orders: usize,
        style_words: *const u32,
        n_words: usize,
        unchanged: *const u8,
    ) -> c_int;
    pub fn forma_hip_set_images(
        ctx: *mut forma_hip_ctx,
        images: *const forma_image_t,
        n_images: usize,
        texels: *const u16,
        n_texels: usize,
    ) -> c_int;

    // stage 1
    pub fn forma_hip_flatten(
        ctx: *mut forma_hip_ctx,
        t: *const forma_flatten_tables_t,
        out_x: *mut f32,
        out_y: *mut f32,
    ) -> c_int;

    // stage entry points (parity tests)
    pub fn forma_hip_prepare_lines(
        ctx: *mut forma_hip_ctx,
        width: u32,
        height: u32,
        orders: *mut u32,
        x0: *mut f32,
        y0: *mut f32,
        dx: *mut f32,
        dy: *mut f32,
        a: *mut f32,
        b: *mut f32,
        c: *mut f32,
        d: *mut f32,
        lengths: *mut u32,
    ) -> c_int;
    pub fn forma_hip_rasterize(
        ctx: *mut forma_hip_ctx,
        n_lines: usize,
        orders: *const u32,
        x0: *const f32,
        y0: *const f32,
        dx: *const f32,
        dy: *const f32,
        a: *const f32,
        b: *const f32,
        c: *const f32,
        d: *const f32,
        lengths: *const u32,
        out_segments: *mut u64,
        capacity: usize,
        out_n: *mut usize,
    ) -> c_int;
    pub fn forma_hip_sort(ctx: *mut forma_hip_ctx, segments: *mut u64, n: usize, digit_bits: c_int) -> c_int;
    pub fn forma_hip_paint(
        ctx: *mut forma_hip_ctx,
        sorted_segments: *const u64,
        n: usize,
        dst: *mut u8,
        width: u32,
        height: u32,
        stride_bytes: usize,
        channels: *const u8,
        clear_color: *const f32,
        crop_or_null: *const forma_rect_t,
    ) -> c_int;

    // the frame
    pub fn forma_hip_render(
        ctx: *mut forma_hip_ctx,
        dst: *mut u8,
        width: u32,
        height: u32,
        stride_bytes: usize,
        channels: *const u8,
        clear_color: *const f32,
        crop_or_null: *const forma_rect_t,
        cache_id: c_int,
        timings: *mut forma_timings_t,
    ) -> c_int;
    pub fn forma_hip_cache_clear(ctx: *mut forma_hip_ctx, cache_id: c_int) -> c_int;
    pub fn forma_hip_set_frames_in_flight(ctx: *mut forma_hip_ctx, n: c_int) -> c_int;
    pub fn forma_hip_multi_layout(ctx: *mut forma_hip_ctx, layout: c_int) -> c_int;
    pub fn forma_hip_sync(ctx: *mut forma_hip_ctx) -> c_int;
    pub fn forma_hip_context_info(ctx: *mut forma_hip_ctx, out: *mut forma_context_info_t) -> c_int;
    pub fn forma_hip_kernel_times(ctx: *mut forma_hip_ctx, out: *mut forma_kernel_time_t, capacity: usize, out_n: *mut usize) -> c_int;
    pub fn forma_hip_sort_plan(
        live_key_bits: u64,
        layer_sorted: c_int,
        digit_bits: c_int,
        field_range: *const u32,
        out: *mut forma_sort_plan_t,
    ) -> c_int;
    pub fn forma_hip_render_enqueue(
        ctx: *mut forma_hip_ctx,
        dst: *mut u8,
        width: u32,
        height: u32,
        stride_bytes: usize,
        channels: *const u8,
        clear_color: *const f32,
        crop_or_null: *const forma_rect_t,
    ) -> c_int;
    pub fn forma_hip_register_buffer(ctx: *mut forma_hip_ctx, ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn forma_hip_unregister_buffer(ctx: *mut forma_hip_ctx, ptr: *mut c_void) -> c_int;
    pub fn forma_hip_trim(ctx: *mut forma_hip_ctx) -> c_int;

    // inspection
    pub fn forma_hip_read_segments(
        ctx: *mut forma_hip_ctx,
        which: c_int,
        out: *mut u64,
        capacity: usize,
        out_n: *mut usize,
    ) -> c_int;
    pub fn forma_hip_read_image(ctx: *mut forma_hip_ctx, dst: *mut u8, stride_bytes: usize) -> c_int;
    pub fn forma_hip_tiles_written(ctx: *mut forma_hip_ctx, flags: *mut u8, n_tiles: usize) -> c_int;

    // multi-GPU: band ownership
    pub fn forma_hip_set_band(ctx: *mut forma_hip_ctx, row0: u32, row1: u32) -> c_int;
    pub fn forma_hip_segments_device(
        ctx: *mut forma_hip_ctx,
        which: c_int,
        dev_ptr: *mut *mut u64,
        n: *mut usize,
    ) -> c_int;
    pub fn forma_hip_rasterize_frame(
        ctx: *mut forma_hip_ctx,
        width: u32,
        height: u32,
        timings: *mut forma_timings_t,
    ) -> c_int;
    pub fn forma_hip_reserve_segments(ctx: *mut forma_hip_ctx, n: usize, dev_ptr: *mut *mut u64) -> c_int;
    pub fn forma_hip_sort_paint_frame(
        ctx: *mut forma_hip_ctx,
        n: usize,
        dst: *mut u8,
        width: u32,
        height: u32,
        stride_bytes: usize,
        channels: *const u8,
        clear_color: *const f32,
        crop_or_null: *const forma_rect_t,
        timings: *mut forma_timings_t,
    ) -> c_int;

    // multi-GPU: exchange layout
    pub fn forma_hip_stream(ctx: *mut forma_hip_ctx, hip_stream: *mut *mut c_void) -> c_int;
    pub fn forma_hip_exchange_plan(
        ctx: *mut forma_hip_ctx,
        row_edges: *const u32,
        n_ranks: u32,
        pair_capacity: u32,
    ) -> c_int;
    pub fn forma_hip_exchange_buffers(
        ctx: *mut forma_hip_ctx,
        send: *mut *mut u64,
        recv: *mut *mut u64,
        words_per_pair: *mut usize,
    ) -> c_int;
    pub fn forma_hip_rasterize_bucket_frame(
        ctx: *mut forma_hip_ctx,
        width: u32,
        height: u32,
        timings: *mut forma_timings_t,
    ) -> c_int;
    pub fn forma_hip_gather_sort_paint_frame(
        ctx: *mut forma_hip_ctx,
        dst: *mut u8,
        width: u32,
        height: u32,
        stride_bytes: usize,
        channels: *const u8,
        clear_color: *const f32,
        crop_or_null: *const forma_rect_t,
        timings: *mut forma_timings_t,
    ) -> c_int;
}
