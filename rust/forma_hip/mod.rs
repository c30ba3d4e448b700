//! `forma/src/hip/mod.rs` — a third backend next to `forma::cpu` and `forma::gpu`: `forma::hip::Renderer` has the same
//! three methods as `cpu::Renderer` (reference `forma/src/cpu/renderer.rs:61-224`) and hands the frame to
//! `libforma_hip.so` (MI355X / gfx950) through the C ABI of `include/forma_hip.h`.
//!
//! Lives INSIDE the forma crate (it reads `pub(crate)` fields: `Composition::{layers, shared_state}`
//! `composition/mod.rs:52-56`, `Layer::{inner, props}` + `is_unchanged/set_is_unchanged` `composition/layer.rs:62-69,
//! 179-189`, `Buffer::{buffer, layout, layer_cache, flusher}` `cpu/buffer/mod.rs:43-49`, `BufferLayerCache::{id,
//! cache}` `:167-170`, `Gradient`/`Image` internals `styling.rs:141-147,272-284`, `Rect::{hor, vert}`
//! `cpu/renderer.rs:37-41`).  Three small hooks in reference files are needed, listed in `README.md` next to this
//! file: `SegmentBuffer::{generation, raw_parts}` (segment.rs has no change counter), `Layout::linear_stride`, and
//! the `pub mod hip` line.
//!
//! Not compiled in this repository (no `rustc`/`cargo` in the image); the same ABI calls in the same order are what
//! `forma_amd/api.py::Renderer` makes through ctypes, and that path is what the GPU tests exercise.

#![allow(clippy::too_many_arguments)]

pub mod ffi;

use std::{cell::RefCell, ffi::CStr, ptr, rc::Rc};

use rustc_hash::FxHashMap;

use crate::{
    consts,
    cpu::{
        buffer::{
            layout::{Flusher, Layout, TileFill},
            Buffer, BufferLayerCache,
        },
        Channel, Rect,
    },
    styling::{Color, Fill, FillRule, Func, GradientType, ImageId, Props},
    utils::SmallBitSet,
    Composition, GeomId,
};

use self::ffi::{forma_geom_t, forma_image_t, forma_rect_t, forma_timings_t};

const TILE_WIDTH: usize = consts::cpu::TILE_WIDTH;
const TILE_HEIGHT: usize = consts::cpu::TILE_HEIGHT;

/// Image table of the last upload: the ids in slot order and the texel pool built from them.
#[derive(Debug, Default)]
struct ImageTable {
    ids: Vec<ImageId>,
    entries: Vec<forma_image_t>,
    texels: Vec<u16>,
}

/// Host-side copies of what is resident on the device, so that a frame in which nothing changed uploads nothing.
#[derive(Debug, Default)]
struct Resident {
    /// `SegmentBuffer::generation()` of the geometry on the device; `None` before the first upload.
    geometry_generation: Option<u64>,
    /// Which composition's store that was (`Rc::as_ptr` of the shared state), to tell two compositions apart.
    geometry_owner: usize,
    /// Dense slot per `GeomId` present in the uploaded store, first appearance order.  `GeomId`s grow without bound
    /// (`segment.rs:100-131`); slots are re-assigned from 0 at every geometry upload, so the per-slot table is never
    /// larger than the number of distinct ids in the store (garbage included until `compact_geom` drops it).
    slot_of: FxHashMap<GeomId, u32>,
    geoms: Vec<forma_geom_t>,
    style_offsets: Vec<u32>,
    style_words: Vec<u32>,
    unchanged: Vec<u8>,
    images: ImageTable,
}

/// MI355X renderer with the interface of `cpu::Renderer`.
#[derive(Debug)]
pub struct Renderer {
    ctx: *mut ffi::forma_hip_ctx,
    buffers_with_caches: Rc<RefCell<SmallBitSet>>,
    resident: Resident,
    /// Scratch for layouts that are not linear: the frame as a tightly packed RGBA8 image.
    linear_scratch: Vec<u8>,
    tile_flags: Vec<u8>,
    timings: forma_timings_t,
}

impl Default for Renderer {
    fn default() -> Self {
        Self::new()
    }
}

impl Drop for Renderer {
    fn drop(&mut self) {
        // SAFETY: `ctx` came from `forma_hip_create` and is destroyed exactly once.
        unsafe { ffi::forma_hip_destroy(self.ctx) }
    }
}

impl Renderer {
    /// Creates a renderer on HIP device 0.  Panics when `libforma_hip.so` finds no gfx950 device — there is no CPU
    /// fallback behind this type; use `cpu::Renderer` for that.
    #[inline]
    pub fn new() -> Self {
        Self::with_device(0)
    }

    /// One process per GPU: pass the local rank.
    pub fn with_device(device: i32) -> Self {
        let mut ctx = ptr::null_mut();
        // SAFETY: plain out-pointer call.
        let rc = unsafe { ffi::forma_hip_create(&mut ctx, device) };
        if rc != ffi::FORMA_OK {
            panic!("forma_hip_create(device = {}) failed with {}", device, rc);
        }

        Self {
            ctx,
            buffers_with_caches: Rc::default(),
            resident: Resident::default(),
            linear_scratch: Vec::new(),
            tile_flags: Vec::new(),
            timings: forma_timings_t::default(),
        }
    }

    /// ONE renderer over several GPUs of this process (`forma_hip_create_multi`): `render` keeps its signature and its
    /// contract — the buffer is fully written when it returns — and the multi-GPU frame happens inside the library:
    /// every device rasterizes its share of the lines, one RCCL all-to-all over xGMI moves the pixel segments to the
    /// device that owns their tile row, every device sorts and paints its band and copies its rows into the caller's
    /// buffer.  `demo` and `e2e-tests` need no change beyond this constructor.
    pub fn with_devices(devices: &[i32]) -> Self {
        let mut ctx = ptr::null_mut();
        // SAFETY: `devices` outlives the call; plain out-pointer.
        let rc = unsafe { ffi::forma_hip_create_multi(&mut ctx, devices.as_ptr(), devices.len() as i32) };
        if rc != ffi::FORMA_OK {
            panic!("forma_hip_create_multi({:?}) failed with {}", devices, rc);
        }

        Self {
            ctx,
            buffers_with_caches: Rc::default(),
            resident: Resident::default(),
            linear_scratch: Vec::new(),
            tile_flags: Vec::new(),
            timings: forma_timings_t::default(),
        }
    }

    /// Frames in flight inside this renderer (`forma_hip_set_frames_in_flight`): relevant to callers that keep the image
    /// on the device (`ffi::forma_hip_render` with a null `dst`, e.g. a presenter that blits from device memory);
    /// `render` into a `Buffer` stays synchronous — the reference's contract (`cpu/buffer/mod.rs:43-49`).
    pub fn set_frames_in_flight(&mut self, n: i32) {
        // SAFETY: `ctx` is live for the lifetime of `self`.
        let rc = unsafe { ffi::forma_hip_set_frames_in_flight(self.ctx, n) };
        self.check(rc, "forma_hip_set_frames_in_flight");
    }

    /// How a renderer over several devices splits a frame (`forma_hip_multi_layout`): `ffi::FORMA_LAYOUT_EXCHANGE` (line
    /// shares + one all-to-all of pixel segments), `ffi::FORMA_LAYOUT_BANDS` (no exchange: every device culls the scene to its
    /// band of tile rows) or `ffi::FORMA_LAYOUT_AUTO` (the default).  Same image either way.
    pub fn set_multi_layout(&mut self, layout: i32) {
        // SAFETY: `ctx` is live for the lifetime of `self`.
        let rc = unsafe { ffi::forma_hip_multi_layout(self.ctx, layout) };
        self.check(rc, "forma_hip_multi_layout");
    }

    /// Waits for every enqueued frame (`forma_hip_sync`); panics with the first error one of them produced.
    pub fn sync(&mut self) {
        // SAFETY: as above.
        let rc = unsafe { ffi::forma_hip_sync(self.ctx) };
        self.check(rc, "forma_hip_sync");
    }

    /// What this renderer is made of (`forma_hip_context_info`): devices, frame slots, how pixel segments travel.
    pub fn info(&self) -> ffi::forma_context_info_t {
        let mut out = ffi::forma_context_info_t::default();
        // SAFETY: `ctx` is live; `out` is a plain `#[repr(C)]` value of the layout the header declares.
        let rc = unsafe { ffi::forma_hip_context_info(self.ctx, &mut out) };
        self.check(rc, "forma_hip_context_info");
        out
    }

    /// The kernels of the last `render` call that asked for timings, in launch order: (name, stage, start µs, µs), each timed
    /// by its own launch (`forma_hip_kernel_times`) — what replaces the reference's host-side `duration!` timers
    /// (`cpu/renderer.rs:106-223`) when the stages run on the device.
    pub fn kernel_times(&self) -> Vec<(String, u32, f32, f32)> {
        // SAFETY: `forma_kernel_time_t` is plain `#[repr(C)]` data; the library writes at most `capacity` entries.
        let mut raw: Vec<ffi::forma_kernel_time_t> = vec![unsafe { std::mem::zeroed() }; 96];
        let mut n = 0usize;
        let rc = unsafe { ffi::forma_hip_kernel_times(self.ctx, raw.as_mut_ptr(), raw.len(), &mut n) };
        self.check(rc, "forma_hip_kernel_times");
        raw.truncate(n.min(96));
        raw.iter()
            .map(|k| {
                let name: Vec<u8> = k.name.iter().take_while(|&&c| c != 0).map(|&c| c as u8).collect();
                (String::from_utf8_lossy(&name).into_owned(), k.stage, k.start_us, k.us)
            })
            .collect()
    }

    /// The frame AND its copy into `buffer` are enqueued on the next frame slot (`forma_hip_render_enqueue`); `buffer` must be one
    /// the caller registered with [`Renderer::register`] and must not be read before `sync()` — or before `frames in flight`
    /// further frames have been enqueued.  This is the one call that offers more than `cpu::Renderer`: a presenter that rotates
    /// three window buffers gets the 33 MB PCIe copy of frame k under the kernels of frames k + 1 and k + 2 (1 650 frames/s at 4K
    /// against 900 for `render`, which stays synchronous — `cpu/buffer/mod.rs:43-49`).  Same scene walk as `render`.
    ///
    /// # Safety
    /// The library keeps `pixels.as_mut_ptr()` and DMA-writes through it AFTER this call returns, until `sync()` (or until
    /// `frames in flight` further frames were enqueued).  The caller must keep the allocation alive, registered and untouched
    /// for that long: dropping, reallocating or reading the buffer earlier is a use-after-free / data race the borrow checker
    /// cannot see, which is why this is not a safe fn.
    pub unsafe fn render_enqueue(
        &mut self,
        composition: &mut Composition,
        pixels: &mut [u8],
        layout: &crate::buffer::layout::LinearLayout,
        channels: [Channel; 4],
        clear_color: Color,
        crop: Option<Rect>,
    ) {
        // cpu/renderer.rs:113-118, then the same scene walk as `render`
        composition.compact_geom();
        composition.shared_state.borrow_mut().props_interner.compact();
        self.upload_geometry(composition);
        self.upload_tables(composition, None);
        let stride = layout.linear_stride().expect("LinearLayout has a stride");
        assert!(pixels.len() >= stride * layout.height());
        let ch = channel_codes(channels, clear_color);
        let clear = [clear_color.r, clear_color.g, clear_color.b, clear_color.a];
        let rect = pixel_rect(crop.as_ref(), layout.width(), layout.height());
        // SAFETY: `pixels` is registered (page-locked, kept alive by the caller until `unregister`), `ch` / `clear` / `rect` live
        // for the call; the library copies what it needs to keep (the arguments of a deferred frame) before returning.
        let rc = unsafe {
            ffi::forma_hip_render_enqueue(
                self.ctx,
                pixels.as_mut_ptr(),
                layout.width() as u32,
                layout.height() as u32,
                stride,
                ch.as_ptr(),
                clear.as_ptr(),
                rect.as_ref().map_or(std::ptr::null(), |r| r as *const _),
            )
        };
        self.check(rc, "forma_hip_render_enqueue");
    }

    /// Page-locks a pixel buffer the renderer writes often (`forma_hip_register_buffer`); undo with [`Renderer::unregister`]
    /// before the memory is freed.
    ///
    /// # Safety
    /// `hipHostRegister` pins the pages behind `pixels` beyond the borrow: the allocation must stay alive and must not move
    /// (no `Vec` growth) until `unregister` was called with the same slice.
    pub unsafe fn register(&mut self, pixels: &mut [u8]) {
        // SAFETY: the slice is valid for its length; the caller keeps it alive until `unregister`.
        let rc = unsafe { ffi::forma_hip_register_buffer(self.ctx, pixels.as_mut_ptr().cast(), pixels.len()) };
        self.check(rc, "forma_hip_register_buffer");
    }
    /// # Safety
    /// `pixels` must be the slice given to [`Renderer::register`]; no frame enqueued into it may be read before this returns.
    pub unsafe fn unregister(&mut self, pixels: &mut [u8]) {
        // SAFETY: as above; the call waits for frames in flight first.
        let rc = unsafe { ffi::forma_hip_unregister_buffer(self.ctx, pixels.as_mut_ptr().cast()) };
        self.check(rc, "forma_hip_unregister_buffer");
    }

    /// Gives the per-frame device memory back (`forma_hip_trim`): the scene and the caches stay, the next frame allocates again.
    pub fn trim(&mut self) {
        // SAFETY: as above.
        let rc = unsafe { ffi::forma_hip_trim(self.ctx) };
        self.check(rc, "forma_hip_trim");
    }

    /// Same contract as `cpu::Renderer::create_buffer_layer_cache` (`cpu/renderer.rs:68-73`): at most 32 live caches.
    #[inline]
    pub fn create_buffer_layer_cache(&mut self) -> Option<BufferLayerCache> {
        self.buffers_with_caches
            .borrow_mut()
            .first_empty_slot()
            .map(|id| BufferLayerCache::new(id, Rc::downgrade(&self.buffers_with_caches)))
    }

    /// Per-stage device times of the last `render` call.
    #[inline]
    pub fn last_timings(&self) -> &forma_timings_t {
        &self.timings
    }

    fn check(&self, rc: i32, what: &str) {
        if rc != ffi::FORMA_OK {
            // SAFETY: the library returns a NUL-terminated string owned by the context.
            let msg = unsafe { CStr::from_ptr(ffi::forma_hip_last_error(self.ctx)) };
            // The reference panics on invariant violations; nothing unwinds across the ABI itself.
            panic!("{} failed with {}: {}", what, rc, msg.to_string_lossy());
        }
    }

    /// Same signature and observable behaviour as `cpu::Renderer::render` (`cpu/renderer.rs:75-224`).
    pub fn render<L>(
        &mut self,
        composition: &mut Composition,
        buffer: &mut Buffer<'_, '_, L>,
        mut channels: [Channel; 4],
        clear_color: Color,
        crop: Option<Rect>,
    ) where
        L: Layout,
    {
        // renderer.rs:87-92 (the library does the same upgrade; doing it here keeps the two call sites alike).
        if clear_color.a == 1.0 {
            channels = channels.map(|c| match c {
                Channel::Alpha => Channel::One,
                c => c,
            });
        }

        let width = buffer.layout.width();
        let height = buffer.layout.height();

        // renderer.rs:94-111.  The per-tile `CachedTile` state lives on the device (keyed by cache id); the host-side
        // `CacheInner` keeps what the reference keeps there that users can observe or reset: the size and the clear
        // colour.  `clear_color == None` means "never rendered, resized, or `BufferLayerCache::clear()` was called"
        // (`cpu/buffer/mod.rs:189-196`) — in all three cases the device state is dropped as well.
        let cache_id = match buffer.layer_cache.as_ref() {
            Some(layer_cache) => {
                let mut cache = layer_cache.cache.borrow_mut();

                if cache.width != Some(width) || cache.height != Some(height) {
                    cache.width = Some(width);
                    cache.height = Some(height);
                    cache.clear_color = None;
                }

                if cache.clear_color.is_none() {
                    // SAFETY: valid context, id < 32.
                    let rc = unsafe { ffi::forma_hip_cache_clear(self.ctx, i32::from(layer_cache.id)) };
                    self.check(rc, "forma_hip_cache_clear");
                }

                Some(layer_cache.id)
            }
            None => None,
        };

        // renderer.rs:113-118.
        composition.compact_geom();
        composition.shared_state.borrow_mut().props_interner.compact();

        self.upload_geometry(composition);
        self.upload_tables(composition, cache_id);

        let channels = channel_codes(channels, clear_color);
        let clear = [clear_color.r, clear_color.g, clear_color.b, clear_color.a];
        let rect = pixel_rect(crop.as_ref(), width, height);
        let rect_ptr = rect.as_ref().map_or(ptr::null(), |rect| rect as *const forma_rect_t);
        let cache_arg = cache_id.map_or(-1, i32::from);

        let width_in_tiles = buffer.layout.width_in_tiles();
        let height_in_tiles = buffer.layout.height_in_tiles();
        let flusher = buffer.flusher.as_deref();

        match buffer.layout.linear_stride() {
            // LinearLayout: the library copies the written tiles straight into the caller's rows.
            Some(width_stride) => {
                assert!(
                    height * width_stride <= buffer.buffer.len(),
                    "height * width_stride exceeds buffer length: {} > {}",
                    height * width_stride,
                    buffer.buffer.len(),
                );

                // SAFETY: `dst` covers `height` rows of `width_stride` bytes (asserted above); all other pointers
                // outlive the call.
                let rc = unsafe {
                    ffi::forma_hip_render(
                        self.ctx,
                        buffer.buffer.as_mut_ptr(),
                        width as u32,
                        height as u32,
                        width_stride,
                        channels.as_ptr(),
                        clear.as_ptr(),
                        rect_ptr,
                        cache_arg,
                        &mut self.timings,
                    )
                };
                self.check(rc, "forma_hip_render");

                // `Flusher::flush` on every row slice of every tile that was written, exactly the slices
                // `LinearLayout::write` flushes (`layout/mod.rs:283-294`, `painter/mod.rs:537-548`).
                if let Some(flusher) = flusher {
                    self.read_tile_flags(width_in_tiles * height_in_tiles);

                    for (tile, _) in self.tile_flags.iter().enumerate().filter(|(_, &f)| f != 0) {
                        let tile_x = tile % width_in_tiles;
                        let tile_y = tile / width_in_tiles;
                        let len = ((width - tile_x * TILE_WIDTH) * 4).min(TILE_WIDTH * 4);

                        for y in tile_y * TILE_HEIGHT..((tile_y + 1) * TILE_HEIGHT).min(height) {
                            let start = y * width_stride + tile_x * TILE_WIDTH * 4;
                            flusher.flush(&mut buffer.buffer[start..start + len]);
                        }
                    }
                }
            }
            // Any other `Layout`: render device-side, fetch the packed image once, and hand every written tile to
            // `L::write` as `TileFill::Full` (column-major, `layout/mod.rs:37-44,108-126`).  Not the hot path.
            None => {
                // SAFETY: dst == NULL keeps the image on the device.
                let rc = unsafe {
                    ffi::forma_hip_render(
                        self.ctx,
                        ptr::null_mut(),
                        width as u32,
                        height as u32,
                        width * 4,
                        channels.as_ptr(),
                        clear.as_ptr(),
                        rect_ptr,
                        cache_arg,
                        &mut self.timings,
                    )
                };
                self.check(rc, "forma_hip_render");

                self.linear_scratch.resize(width * height * 4, 0);
                // SAFETY: the scratch holds `height` rows of `width * 4` bytes.
                let rc = unsafe { ffi::forma_hip_read_image(self.ctx, self.linear_scratch.as_mut_ptr(), width * 4) };
                self.check(rc, "forma_hip_read_image");
                self.read_tile_flags(width_in_tiles * height_in_tiles);

                let slices_per_tile = buffer.layout.slices_per_tile();
                let mut slices = buffer.layout.slices(buffer.buffer);
                let mut colors = [[0u8; 4]; TILE_WIDTH * TILE_HEIGHT];

                for (tile, _) in self.tile_flags.iter().enumerate().filter(|(_, &f)| f != 0) {
                    let tile_x = tile % width_in_tiles;
                    let tile_y = tile / width_in_tiles;

                    for x in 0..TILE_WIDTH.min(width - tile_x * TILE_WIDTH) {
                        for y in 0..TILE_HEIGHT.min(height - tile_y * TILE_HEIGHT) {
                            let src = ((tile_y * TILE_HEIGHT + y) * width + tile_x * TILE_WIDTH + x) * 4;
                            colors[x * TILE_HEIGHT + y].copy_from_slice(&self.linear_scratch[src..src + 4]);
                        }
                    }

                    L::write(
                        &mut slices[tile * slices_per_tile..(tile + 1) * slices_per_tile],
                        flusher,
                        TileFill::Full(&colors),
                    );
                }
            }
        }

        // renderer.rs:216-223.
        if let Some(layer_cache) = &buffer.layer_cache {
            layer_cache.cache.borrow_mut().clear_color = Some(clear_color);

            for layer in composition.layers.values_mut() {
                layer.set_is_unchanged(layer_cache.id, layer.inner.is_enabled);
            }
        }
    }

    fn read_tile_flags(&mut self, tiles_len: usize) {
        self.tile_flags.resize(tiles_len, 0);
        // SAFETY: `tile_flags` holds `tiles_len` bytes.
        let rc = unsafe { ffi::forma_hip_tiles_written(self.ctx, self.tile_flags.as_mut_ptr(), tiles_len) };
        self.check(rc, "forma_hip_tiles_written");
    }

    /// Geometry store -> device, only when it changed since the last upload (insert / clear+insert / compaction).
    /// `x`, `y`, `ids` are `SegmentBufferView`'s first three fields (`segment.rs:529-534`); `ids[i]` names the geometry
    /// of the line point i -> point i + 1, `None` between two polygonal chains.
    fn upload_geometry(&mut self, composition: &Composition) {
        let owner = Rc::as_ptr(&composition.shared_state) as usize;
        let state = composition.shared_state.borrow();
        let segment_buffer = state.segment_buffer.as_ref().expect("segment_buffer should not be None");
        let generation = segment_buffer.generation();

        if self.resident.geometry_generation == Some(generation) && self.resident.geometry_owner == owner {
            return;
        }

        let (x, y, ids) = segment_buffer.raw_parts();
        let lines_len = x.len().saturating_sub(1);

        let slot_of = &mut self.resident.slot_of;
        slot_of.clear();

        let line_slot: Vec<u32> = ids[..lines_len.min(ids.len())]
            .iter()
            .map(|id| match id {
                Some(id) => {
                    let next = slot_of.len() as u32;
                    *slot_of.entry(*id).or_insert(next)
                }
                None => ffi::FORMA_NONE,
            })
            .collect();
        debug_assert_eq!(line_slot.len(), lines_len);

        // SAFETY: x and y hold `x.len()` floats, `line_slot` holds `x.len() - 1` words (0 for an empty store).
        let rc = unsafe { ffi::forma_hip_set_geometry(self.ctx, x.as_ptr(), y.as_ptr(), line_slot.as_ptr(), x.len()) };
        self.check(rc, "forma_hip_set_geometry");

        self.resident.geometry_generation = Some(generation);
        self.resident.geometry_owner = owner;
        // The slot numbering changed: force the per-slot table out.
        self.resident.geoms.clear();
    }

    /// Per-frame tables: slot -> (order, transform) = `geom_id_to_order` + `InnerLayer` exactly as
    /// `SegmentBuffer::fill_cpu_view` resolves them per line (`segment.rs:141-149,309-329`); order -> style words +
    /// the `is_unchanged(cache_id)` bit the painter's optimizer passes read (`cpu/renderer.rs:126-157`).
    fn upload_tables(&mut self, composition: &Composition, cache_id: Option<u8>) {
        let state = composition.shared_state.borrow();
        let layers = &composition.layers;

        let mut geoms = vec![forma_geom_t::HIDDEN; self.resident.slot_of.len().max(1)];
        for (geom_id, &slot) in &self.resident.slot_of {
            let inner = state
                .geom_id_to_order
                .get(geom_id)
                .copied()
                .flatten()
                .and_then(|order| layers.get(&order))
                .map(|layer| &layer.inner);

            if let Some(inner) = inner {
                if let (true, Some(order)) = (inner.is_enabled, inner.order) {
                    geoms[slot as usize] = match inner.affine_transform.as_ref() {
                        // the ABI wants AffineTransform::to_array's order (ux, uy, vx, vy, tx, ty; transform.rs:54-56), not
                        // GeomPresTransform::to_array's (ux, vx, uy, vy, tx, ty; :194-198): take the inner transform
                        Some(transform) => forma_geom_t {
                            order: order.as_u32(),
                            flags: ffi::FORMA_GEOM_HAS_XF,
                            xf: transform.0.to_array(),
                        },
                        None => forma_geom_t {
                            order: order.as_u32(),
                            flags: 0,
                            xf: [0.0; 6],
                        },
                    };
                }
            }
        }

        let orders_len = layers.keys().map(|order| order.as_u32() as usize + 1).max().unwrap_or(0);
        let mut style_offsets = vec![ffi::FORMA_NONE; orders_len];
        let mut unchanged = vec![0u8; orders_len];
        let mut style_words = Vec::new();
        let mut image_ids: Vec<ImageId> = Vec::new();
        let mut new_images: Vec<&crate::styling::Image> = Vec::new();
        let mut offset_of: FxHashMap<&Props, u32> = FxHashMap::default();

        for (order, layer) in layers {
            let props: &Props = &layer.props;
            let offset = *offset_of.entry(props).or_insert_with(|| {
                let offset = style_words.len() as u32;
                encode_props(props, &mut style_words, &mut image_ids, &mut new_images);
                offset
            });

            style_offsets[order.as_u32() as usize] = offset;
            unchanged[order.as_u32() as usize] = cache_id.map_or(false, |id| layer.is_unchanged(id)) as u8;
        }

        if geoms != self.resident.geoms {
            // SAFETY: `geoms` holds `geoms.len()` entries.
            let rc = unsafe { ffi::forma_hip_set_geoms(self.ctx, geoms.as_ptr(), geoms.len()) };
            self.check(rc, "forma_hip_set_geoms");
            self.resident.geoms = geoms;
        }

        if image_ids != self.resident.images.ids {
            let mut entries = Vec::with_capacity(new_images.len());
            let mut texels: Vec<u16> = Vec::new();

            for image in &new_images {
                entries.push(forma_image_t {
                    texel_offset: (texels.len() / 4) as u64,
                    width: image.width(),
                    height: image.height(),
                });

                let data = image.data();
                // SAFETY: `styling::f16` is `#[repr(C)] struct f16(u16)` (`styling.rs:224-228`): `[[f16; 4]]` is a
                // plain array of u16 with 4 * len elements.
                texels.extend_from_slice(unsafe { std::slice::from_raw_parts(data.as_ptr().cast::<u16>(), data.len() * 4) });
            }

            // SAFETY: lengths match the vectors.
            let rc = unsafe {
                ffi::forma_hip_set_images(self.ctx, entries.as_ptr(), entries.len(), texels.as_ptr(), texels.len() / 4)
            };
            self.check(rc, "forma_hip_set_images");
            self.resident.images = ImageTable {
                ids: image_ids,
                entries,
                texels,
            };
        }

        if style_offsets != self.resident.style_offsets
            || style_words != self.resident.style_words
            || unchanged != self.resident.unchanged
        {
            // SAFETY: lengths match the vectors; `unchanged` has one byte per order.
            let rc = unsafe {
                ffi::forma_hip_set_styles(
                    self.ctx,
                    style_offsets.as_ptr(),
                    style_offsets.len(),
                    style_words.as_ptr(),
                    style_words.len(),
                    unchanged.as_ptr(),
                )
            };
            self.check(rc, "forma_hip_set_styles");
            self.resident.style_offsets = style_offsets;
            self.resident.style_words = style_words;
            self.resident.unchanged = unchanged;
        }
    }
}

/// `Props` -> style words, the layout documented in `include/forma_hip.h` ("Style table").
fn encode_props<'i>(
    props: &'i Props,
    words: &mut Vec<u32>,
    image_ids: &mut Vec<ImageId>,
    images: &mut Vec<&'i crate::styling::Image>,
) {
    let mut header = match props.fill_rule {
        FillRule::NonZero => 0u32,
        FillRule::EvenOdd => 1 << 6,
    };

    let style = match &props.func {
        Func::Clip(layers) => {
            words.push(header | 1 << 8);
            words.push(*layers as u32);
            return;
        }
        Func::Draw(style) => style,
    };

    // Ordinal of `BlendMode` in declaration order (`styling.rs:390-408`).
    header |= style.blend_mode as u32;
    if style.is_clipped {
        header |= 1 << 7;
    }

    match &style.fill {
        Fill::Solid(color) => {
            words.push(header | ffi::FORMA_FILL_SOLID << 4);
            words.push(0);
            words.extend([color.r, color.g, color.b, color.a].map(f32::to_bits));
        }
        Fill::Gradient(gradient) => {
            let fill = match gradient.r#type() {
                GradientType::Linear => ffi::FORMA_FILL_LINEAR,
                GradientType::Radial => ffi::FORMA_FILL_RADIAL,
            };
            let stops = gradient.colors_with_stops();

            words.push(header | fill << 4 | (stops.len() as u32) << 16);
            words.push(0);
            words.extend([gradient.start().x, gradient.start().y, gradient.end().x, gradient.end().y].map(f32::to_bits));
            for (color, stop) in stops {
                words.extend([color.r, color.g, color.b, color.a, *stop].map(f32::to_bits));
            }
        }
        Fill::Texture(texture) => {
            let id = texture.image.id();
            let index = match image_ids.iter().position(|&other| other == id) {
                Some(index) => index,
                None => {
                    image_ids.push(id);
                    images.push(&texture.image);
                    image_ids.len() - 1
                }
            };

            words.push(header | ffi::FORMA_FILL_TEXTURE << 4);
            words.push(0);
            words.extend(texture.transform.to_array().map(f32::to_bits));
            words.push(index as u32);
        }
    }
}

#[cfg(test)]
mod tests {
    //! The reference's own renderer tests, pointed at this backend: every test of `composition/mod.rs:419-948` that
    //! builds `cpu::Renderer::new()` runs unchanged with `hip::Renderer::new()`.  `tests/test_composition_vectors.py`
    //! in this repository is that list, replayed through the ctypes binding.
    use super::*;

    use crate::{
        cpu::{buffer::{layout::LinearLayout, BufferBuilder}, RGBA},
        math::Point,
        styling::Style,
        Order, PathBuilder,
    };

    #[test]
    fn triangle_matches_cpu_backend() {
        let mut composition = Composition::new();
        let mut builder = PathBuilder::new();
        builder.move_to(Point::new(2.0, 2.0));
        builder.line_to(Point::new(30.0, 2.0));
        builder.line_to(Point::new(2.0, 30.0));
        builder.line_to(Point::new(2.0, 2.0));

        composition
            .get_mut_or_insert_default(Order::new(0).unwrap())
            .insert(&builder.build())
            .set_props(Props {
                func: Func::Draw(Style {
                    fill: Fill::Solid(Color { r: 1.0, g: 0.0, b: 0.0, a: 1.0 }),
                    ..Default::default()
                }),
                ..Default::default()
            });

        let clear = Color { r: 1.0, g: 1.0, b: 1.0, a: 1.0 };
        let mut want = vec![0u8; 32 * 32 * 4];
        let mut got = vec![0u8; 32 * 32 * 4];

        crate::cpu::Renderer::new().render(
            &mut composition,
            &mut BufferBuilder::new(&mut want, &mut LinearLayout::new(32, 32 * 4, 32)).build(),
            RGBA,
            clear,
            None,
        );
        Renderer::new().render(
            &mut composition,
            &mut BufferBuilder::new(&mut got, &mut LinearLayout::new(32, 32 * 4, 32)).build(),
            RGBA,
            clear,
            None,
        );

        assert_eq!(want, got);
    }
}

/// `Channel` -> the ABI's selector bytes, after the alpha upgrade of `cpu/renderer.rs:87-92`.
fn channel_codes(channels: [Channel; 4], clear_color: Color) -> [u8; 4] {
    channels.map(|c| match c {
        Channel::Alpha if clear_color.a == 1.0 => ffi::FORMA_CH_ONE,
        Channel::Red => ffi::FORMA_CH_RED,
        Channel::Green => ffi::FORMA_CH_GREEN,
        Channel::Blue => ffi::FORMA_CH_BLUE,
        Channel::Alpha => ffi::FORMA_CH_ALPHA,
        Channel::Zero => ffi::FORMA_CH_ZERO,
        Channel::One => ffi::FORMA_CH_ONE,
    })
}

/// `Rect` holds TILE ranges (`cpu/renderer.rs:43-52`); the ABI takes pixels and rounds out to the same grid.
fn pixel_rect(crop: Option<&Rect>, width: usize, height: usize) -> Option<forma_rect_t> {
    crop.map(|rect| forma_rect_t {
        x0: (rect.hor.start * TILE_WIDTH).min(width) as u32,
        x1: (rect.hor.end * TILE_WIDTH).min(width) as u32,
        y0: (rect.vert.start * TILE_HEIGHT).min(height) as u32,
        y1: (rect.vert.end * TILE_HEIGHT).min(height) as u32,
    })
}

// ---- hook 4: stage 1 behind the drop-in ------------------------------------------------------------------------------------
// The reference flattens a `Path` once, inside `PathData::segments` (`path.rs:617-654`), by `Primitives::into_segments`
// (`:473-538`): `populate_buffers` (`:400-445`, sequential) writes one work item per output point into thread-local
// `ScratchBuffers`, then a Rayon map evaluates the points.  That map is `k_flatten` (`forma_hip_flatten`).  The hook below is
// what `into_segments` calls INSTEAD of the `par_iter` map when the path is large enough to pay for a launch: it hands the
// scratch buffers and the per-quad arrays to the device and fills `Segments::{x, y}`; `start_new_contour` comes from the End
// commands on the host (bit 22, `path.rs:137-168`).  Everything up to and including `populate_buffers` stays the reference's
// own code, so the points are bit-identical (tests/test_parity_contract.py drives `forma_hip_flatten` with tables produced by an
// independent restatement of `populate_buffers`, not by this library's host code).
//
// In `forma/src/path.rs` (new lines only; `Primitives`' fields are private to that file, so the hook lives next to them):
//
//     #[cfg(feature = "hip")]
//     if buffers.point_commands.len() >= crate::hip::FLATTEN_ON_DEVICE_MIN_POINTS {
//         if let Some((x, y)) = crate::hip::flatten_on_device(&crate::hip::FlattenTables {
//             point_commands: &buffers.point_commands, point_indices: &buffers.point_indices, quad_indices: &buffers.quad_indices,
//             qx: &self.x, qy: &self.y, qw: &self.weight, x0: &self.x0, dx_recip: &self.dx_recip, k0: &self.k0, dk: &self.dk,
//             curvatures_recip: &self.curvatures_recip, partial_curvatures: &self.partial_curvatures, splines: &self.splines,
//         }) {
//             segments.start_new_contour = buffers.point_commands.iter().map(|&c| c & 0xFFC0_0000 == 0xFFC0_0000).collect();
//             segments.x = x; segments.y = y;
//             return segments;
//         }
//     }
//
// (`point_indices` / `quad_indices` are `Vec<usize>` in the reference; the hook narrows them to `u32` — a path has far fewer
// than 2^32 points.)

/// Below this many output points a path is flattened by the reference's own Rayon map: a launch plus two copies cost ~30 us.
pub const FLATTEN_ON_DEVICE_MIN_POINTS: usize = 16_384;

/// Borrowed view of `Primitives` + `ScratchBuffers` (`path.rs:123-127, 190-203`) for [`flatten_on_device`].
pub struct FlattenTables<'a> {
    pub point_commands: &'a [u32],
    pub point_indices: &'a [usize],
    pub quad_indices: &'a [usize],
    pub qx: &'a [f32],
    pub qy: &'a [f32],
    pub qw: &'a [f32],
    pub x0: &'a [f32],
    pub dx_recip: &'a [f32],
    pub k0: &'a [f32],
    pub dk: &'a [f32],
    pub curvatures_recip: &'a [f32],
    pub partial_curvatures: &'a [(u32, f32)],
    pub splines: &'a [crate::path::Spline],
}

thread_local! {
    /// `Path`s are built and flattened on whatever thread the application uses, before any `Renderer` exists: stage 1 has its
    /// own small context per thread (device 0), created on first use.  `None` = no device / no library: the caller falls back
    /// to the reference's map (flattening is not part of `Renderer::render`, so this is the one place a fallback is right).
    static FLATTEN_CTX: std::cell::RefCell<Option<Option<FlattenCtx>>> = std::cell::RefCell::new(None);
}

/// Owns a thread's stage-1 context: destroyed when the thread ends (thread-local destructors run `Drop`), so paths built on a
/// thread pool do not leak a stream, events and pinned memory per worker.
struct FlattenCtx(*mut ffi::forma_hip_ctx);
impl Drop for FlattenCtx {
    fn drop(&mut self) {
        // SAFETY: the pointer came from `forma_hip_create` and is destroyed exactly once.
        unsafe { ffi::forma_hip_destroy(self.0) }
    }
}

/// Device stage 1 runs on: `FORMA_HIP_FLATTEN_DEVICE` (default 0) — a `Path` is flattened before any `Renderer` exists, so the
/// host, not a renderer, has to say which GPU of the machine does it.
fn flatten_device() -> i32 {
    std::env::var("FORMA_HIP_FLATTEN_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0)
}

pub fn flatten_on_device(t: &FlattenTables<'_>) -> Option<(Vec<f32>, Vec<f32>)> {
    let ctx = FLATTEN_CTX.with(|c| {
        c.borrow_mut()
            .get_or_insert_with(|| {
                let mut ctx = std::ptr::null_mut();
                // SAFETY: plain out-pointer call.
                (unsafe { ffi::forma_hip_create(&mut ctx, flatten_device()) } == 0).then(|| FlattenCtx(ctx))
            })
            .as_ref()
            .map(|f| f.0)
    })?;
    let n = t.point_commands.len();
    let narrow = |v: &[usize]| v.iter().map(|&i| i as u32).collect::<Vec<u32>>();
    let (pi, qi) = (narrow(t.point_indices), narrow(t.quad_indices));
    let (ps, pc): (Vec<u32>, Vec<f32>) = t.partial_curvatures.iter().copied().unzip();
    let col = |f: fn(&crate::path::Spline) -> f32| t.splines.iter().map(f).collect::<Vec<f32>>();
    let (sp0x, sp0y, sp2x, sp2y) = (col(|s| s.p0.x), col(|s| s.p0.y), col(|s| s.p2.x), col(|s| s.p2.y));
    let tables = ffi::forma_flatten_tables_t {
        point_commands: t.point_commands.as_ptr(), point_indices: pi.as_ptr(), quad_indices: qi.as_ptr(), n_points: n,
        qx: t.qx.as_ptr(), qy: t.qy.as_ptr(), qw: t.qw.as_ptr(), x0: t.x0.as_ptr(), dx_recip: t.dx_recip.as_ptr(),
        k0: t.k0.as_ptr(), dk: t.dk.as_ptr(), curvatures_recip: t.curvatures_recip.as_ptr(),
        partial_spline: ps.as_ptr(), partial_curv: pc.as_ptr(), n_quads: t.x0.len(),
        sp0x: sp0x.as_ptr(), sp0y: sp0y.as_ptr(), sp2x: sp2x.as_ptr(), sp2y: sp2y.as_ptr(), n_splines: t.splines.len(),
    };
    let (mut x, mut y) = (vec![0.0f32; n], vec![0.0f32; n]);
    // SAFETY: every pointer of `tables` borrows a slice that outlives the call; `x` / `y` hold `n` floats each.
    let rc = unsafe { ffi::forma_hip_flatten(ctx, &tables, x.as_mut_ptr(), y.as_mut_ptr()) };
    (rc == 0).then_some((x, y))
}

