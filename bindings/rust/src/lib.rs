//! Raw FFI to `libwafer_hip.so` (`include/wafer_hip.h`, ABI version 1) plus a thin safe wrapper
//! with the call shapes of Wafer's `grid.rs`.
//!
//! UNTESTED: no Rust toolchain exists in the image this engine was built in.  Struct layouts and
//! signatures mirror the C header field for field; `tests/test_abi.py` checks the same layout from
//! the Python side.
#![allow(non_camel_case_types)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct wafer_params {
    pub struct_size: u32,
    pub nx: u32,
    pub ny: u32,
    pub nz: u32,
    pub central_difference: i32, // CentralDifference::ext(): 1 | 2 | 3
    pub dtype: i32,              // 0 = f64, 1 = f32 storage
    pub dn: f64,
    pub dt: f64,
    pub mass: f64,
    pub sig: f64,
    pub max_states: u32,
    pub device: i32,
    pub z_begin: u32,
    pub z_count: u32,
    pub halo_depth: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct wafer_observables_t {
    pub energy: f64,
    pub norm2: f64,
    pub v_infinity: f64,
    pub r2: f64,
}

/// One row of the convergence table (grid.rs:126-221, output.rs:497-521).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct wafer_block_record {
    pub step: u64,
    pub tau: f64,
    pub obs: wafer_observables_t,
    pub diff: f64,
}

/// output.rs:32-45, 540-547
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct wafer_observables_output {
    pub state: u32,
    pub energy: f64,
    pub binding_energy: f64,
    pub r: f64,
    pub l_r: f64,
}

/// wafer_div_plan_t: how the step kernels divide by the run's stencil denominator (wafer_div_plan, wafer_get_div_plan)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct wafer_div_plan_t {
    pub den: f64,
    pub zh: f64,
    pub zl: f64,
    pub checked: i32,
    pub n_candidates: i32,
    pub zl_shift: i32,
    pub reserved: i32,
}

/// wafer_div_plan_f32_t: the same plan in fp32 (WAFER_F32_FAST contexts)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct wafer_div_plan_f32_t {
    pub den: f32,
    pub zh: f32,
    pub zl: f32,
    pub checked: i32,
    pub zl_shift: i32,
}

/// wafer_params.flags
pub const WAFER_FLAG_SKIP_DT_CHECK: u32 = 1;
pub const WAFER_FLAG_UNPLANNED_DIV: u32 = 2;

/// wafer_peer_info (peer stores, wafer_set_overlap mode 3): what a rank publishes to its z-neighbours
#[repr(C)]
#[derive(Clone, Copy)]
pub struct wafer_peer_info {
    pub struct_size: u32,
    pub z_begin: u32,
    pub z_count: u32,
    pub halo_depth: u32,
    pub pid: u64,
    pub phi_addr: [u64; 2],
    pub flags_addr: u64,
    pub phi_alloc_offset: [u64; 2],
    pub phi_ipc: [[u8; 64]; 2],
    pub flags_ipc: [u8; 64],
    pub process_nonce: u64,
    pub device: i32,
    pub reserved: u32,
    pub device_uuid: [u8; 16],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct wafer_slab_info {
    pub z_begin: u32,
    pub z_count: u32,
    pub halo_depth: u32,
    pub ext: u32,
    pub plane_elems: u64,
    pub elem_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct wafer_device_info {
    pub name: [c_char; 64],
    pub arch: [c_char; 64],
    pub compute_units: u32,
    pub memory_clock_khz: u32,
    pub memory_bus_bits: u32,
    pub l2_bytes: u32,
    pub total_bytes: u64,
}

pub enum wafer_ctx {}

pub type wafer_halo_fn = extern "C" fn(*mut c_void, *mut c_void, *mut c_void, *mut c_void, *mut c_void, usize, *mut c_void) -> c_int;
pub type wafer_allreduce_fn = extern "C" fn(*mut c_void, *mut c_void, usize, *mut c_void) -> c_int;

extern "C" {
    pub fn wafer_abi_version() -> c_int;
    pub fn wafer_last_error() -> *const c_char;
    pub fn wafer_ctx_create(p: *const wafer_params, out: *mut *mut wafer_ctx) -> c_int;
    pub fn wafer_ctx_destroy(ctx: *mut wafer_ctx) -> c_int;
    pub fn wafer_synchronize(ctx: *mut wafer_ctx) -> c_int;
    pub fn wafer_set_potential_builtin(ctx: *mut wafer_ctx, potential: c_int) -> c_int;
    pub fn wafer_set_potential_host(ctx: *mut wafer_ctx, v: *const f64, potsub_kind: c_int, potsub_scalar: f64, potsub: *const f64) -> c_int;
    pub fn wafer_set_initial_condition(ctx: *mut wafer_ctx, ic: c_int, seed: u64) -> c_int;
    pub fn wafer_upload_phi(ctx: *mut wafer_ctx, phi: *const f64) -> c_int;
    pub fn wafer_download_phi(ctx: *mut wafer_ctx, phi: *mut f64) -> c_int;
    pub fn wafer_download_phi_owned(ctx: *mut wafer_ctx, out: *mut f64) -> c_int;
    pub fn wafer_upload_phi_resampled(ctx: *mut wafer_ctx, src: *const f64, sx: u32, sy: u32, sz: u32, basis: *const u32) -> c_int;
    pub fn wafer_evolve(ctx: *mut wafer_ctx, wnum: u32, n_steps: u64) -> c_int;
    pub fn wafer_observables(ctx: *mut wafer_ctx, out: *mut wafer_observables_t) -> c_int;
    pub fn wafer_norm2(ctx: *mut wafer_ctx, out: *mut f64) -> c_int;
    pub fn wafer_normalise(ctx: *mut wafer_ctx, norm2: f64) -> c_int;
    pub fn wafer_orthogonalise(ctx: *mut wafer_ctx, wnum: u32) -> c_int;
    pub fn wafer_set_potsub(ctx: *mut wafer_ctx, kind: c_int, scalar: f64, potsub: *const f64) -> c_int;
    pub fn wafer_set_potsub_resampled(ctx: *mut wafer_ctx, src: *const f64, sx: u32, sy: u32, sz: u32) -> c_int;
    pub fn wafer_symmetrise(ctx: *mut wafer_ctx, constraint: c_int) -> c_int;
    pub fn wafer_push_state(ctx: *mut wafer_ctx) -> c_int;
    pub fn wafer_load_state(ctx: *mut wafer_ctx, idx: u32, state: *const f64) -> c_int;
    pub fn wafer_clone_state_to_phi(ctx: *mut wafer_ctx, idx: u32) -> c_int;
    pub fn wafer_set_comm_hooks(ctx: *mut wafer_ctx, halo: wafer_halo_fn, allreduce: wafer_allreduce_fn, user: *mut c_void) -> c_int;
    /// mode 0: exchange after the pass; 1: boundary planes first on a second stream; 2 (default): one launch per
    /// three-step pass, the slab as two halves marched outwards (include/wafer_hip.h)
    pub fn wafer_set_overlap(ctx: *mut wafer_ctx, mode: c_int) -> c_int;
    pub fn wafer_set_stream(ctx: *mut wafer_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn wafer_get_slab_info(ctx: *mut wafer_ctx, out: *mut wafer_slab_info) -> c_int;
    pub fn wafer_get_device_info(ctx: *mut wafer_ctx, out: *mut wafer_device_info) -> c_int;
    pub fn wafer_set_potential_resampled(ctx: *mut wafer_ctx, src: *const f64, sx: u32, sy: u32, sz: u32, basis: *const u32) -> c_int;
    pub fn wafer_download_array(ctx: *mut wafer_ctx, array_id: c_int, out: *mut f64) -> c_int;
    pub fn wafer_get_potsub(ctx: *mut wafer_ctx, kind: *mut c_int, scalar: *mut f64) -> c_int;
    pub fn wafer_download_state(ctx: *mut wafer_ctx, idx: u32, state: *mut f64) -> c_int;
    pub fn wafer_num_states(ctx: *mut wafer_ctx, out: *mut u32) -> c_int;
    pub fn wafer_clear_states(ctx: *mut wafer_ctx) -> c_int;
    pub fn wafer_solve_state(
        ctx: *mut wafer_ctx, wnum: u32, tolerance: f64, screen_update: u64, has_max_steps: c_int, max_steps: u64,
        records: *mut wafer_block_record, max_records: usize, n_records: *mut usize, final_out: *mut wafer_observables_output,
    ) -> c_int;
    pub fn wafer_last_evolve_ms(ctx: *mut wafer_ctx, ms: *mut f32, steps: *mut u64) -> c_int;
    pub fn wafer_stencil_kernel_name(ctx: *mut wafer_ctx) -> *const c_char;
    pub fn wafer_stencil_kernel_instance(ctx: *mut wafer_ctx) -> *const c_char;
    pub fn wafer_stencil_steps_per_launch(ctx: *mut wafer_ctx) -> c_int;
    pub fn wafer_set_stencil_variant(ctx: *mut wafer_ctx, variant: c_int) -> c_int;
    pub fn wafer_set_halo_cycle(ctx: *mut wafer_ctx, passes: c_int) -> c_int;
    pub fn wafer_diag_copy_bw(ctx: *mut wafer_ctx, iters: c_int, unroll: c_int, blocks_per_cu: c_int, gbps: *mut f64) -> c_int;
    pub fn wafer_diag_checksum(ctx: *mut wafer_ctx, z_begin: u32, z_count: u32, out: *mut u64) -> c_int;
    /// which kernel a pass of `wafer_evolve(ctx, wnum, .)` launches, as one line of `key=value` pairs
    pub fn wafer_diag_dispatch(ctx: *mut wafer_ctx, wnum: u32, buf: *mut c_char, n: usize) -> c_int;
    /// padded planes [zp_begin, zp_begin + zp_count) of V / a / b (0 / 1 / 2) or phi (WAFER_ARRAY_PHI = 4), `[px][py][zp_count]`
    pub fn wafer_diag_download_window(ctx: *mut wafer_ctx, array_id: c_int, zp_begin: u32, zp_count: u32, out: *mut f64) -> c_int;
    pub fn wafer_diag_x2_passes(ctx: *mut wafer_ctx, out: *mut u64) -> c_int;
    pub fn wafer_peer_export(ctx: *mut wafer_ctx, out: *mut wafer_peer_info) -> c_int;
    pub fn wafer_peer_connect(ctx: *mut wafer_ctx, lower: *const wafer_peer_info, upper: *const wafer_peer_info) -> c_int;
    pub fn wafer_peer_disconnect(ctx: *mut wafer_ctx) -> c_int;
    pub fn wafer_diag_div_check(ctx: *mut wafer_ctx, den: f64, seed: u64, n_operands: u64, lo_exp: c_int, hi_exp: c_int, mismatches: *mut u64) -> c_int;
    pub fn wafer_div_plan(den: f64, out: *mut wafer_div_plan_t, candidates: *mut f64, cap: usize, n_written: *mut usize) -> c_int;
    pub fn wafer_get_div_plan(ctx: *mut wafer_ctx, out: *mut wafer_div_plan_t) -> c_int;
    pub fn wafer_div_plan_f32(den: f32, out: *mut wafer_div_plan_f32_t) -> c_int;
    pub fn wafer_diag_div_planned_f32(ctx: *mut wafer_ctx, plan: *const wafer_div_plan_f32_t, lo_exp: c_int, hi_exp: c_int, mismatches: *mut u64) -> c_int;
    pub fn wafer_diag_div_planned(ctx: *mut wafer_ctx, plan: *const wafer_div_plan_t, seed: u64, n_random: u64, lo_exp: c_int, hi_exp: c_int,
                                  operands: *const f64, n_operands: usize, mismatches_random: *mut u64, mismatches_operands: *mut u64) -> c_int;
}

/// `Err(message)` for any non-zero status; a Wafer integration maps it to an `ErrorKind`.
pub fn check(rc: c_int) -> Result<(), String> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(wafer_last_error()) }.to_string_lossy().into_owned();
    Err(format!("wafer_hip error {}: {}", rc, msg))
}

/// Device-resident solver state; method names follow `grid.rs`.
pub struct Engine {
    ctx: *mut wafer_ctx,
}

impl Engine {
    pub fn new(mut p: wafer_params) -> Result<Engine, String> {
        p.struct_size = std::mem::size_of::<wafer_params>() as u32;
        let mut ctx: *mut wafer_ctx = std::ptr::null_mut();
        check(unsafe { wafer_ctx_create(&p, &mut ctx) })?;
        Ok(Engine { ctx })
    }
    /// grid.rs:544-687
    pub fn evolve(&mut self, wnum: u8, steps: u64) -> Result<(), String> {
        check(unsafe { wafer_evolve(self.ctx, wnum as u32, steps) })
    }
    /// grid.rs:303-445
    pub fn compute_observables(&mut self) -> Result<wafer_observables_t, String> {
        let mut o = wafer_observables_t::default();
        check(unsafe { wafer_observables(self.ctx, &mut o) })?;
        Ok(o)
    }
    /// grid.rs:465-468
    pub fn normalise_wavefunction(&mut self, norm2: f64) -> Result<(), String> {
        check(unsafe { wafer_normalise(self.ctx, norm2) })
    }
    /// grid.rs:477-492
    pub fn orthogonalise_wavefunction(&mut self, wnum: u8) -> Result<(), String> {
        check(unsafe { wafer_orthogonalise(self.ctx, wnum as u32) })
    }
    /// `phi` is ndarray's standard-layout `Array3<R64>` storage, `(nx+bb, ny+bb, nz+bb)`.
    pub fn upload_phi(&mut self, phi: &[f64]) -> Result<(), String> {
        check(unsafe { wafer_upload_phi(self.ctx, phi.as_ptr()) })
    }
    pub fn download_phi(&mut self, phi: &mut [f64]) -> Result<(), String> {
        check(unsafe { wafer_download_phi(self.ctx, phi.as_mut_ptr()) })
    }
    /// One state of grid::solve (grid.rs:50-246): the evolve / observables / convergence loop on the
    /// device; `Ok(rows, summary)` when converged, the engine's MaxStep error otherwise.
    pub fn solve_state(
        &mut self, wnum: u8, tolerance: f64, screen_update: u64, max_steps: Option<u64>,
    ) -> Result<(Vec<wafer_block_record>, wafer_observables_output), String> {
        let mut rows = vec![wafer_block_record::default(); 4096];
        let mut n: usize = 0;
        let mut out = wafer_observables_output::default();
        check(unsafe {
            wafer_solve_state(
                self.ctx, wnum as u32, tolerance, screen_update, max_steps.is_some() as c_int, max_steps.unwrap_or(0),
                rows.as_mut_ptr(), rows.len(), &mut n, &mut out,
            )
        })?;
        rows.truncate(n);
        Ok((rows, out))
    }
    /// grid.rs:241
    pub fn push_state(&mut self) -> Result<(), String> {
        check(unsafe { wafer_push_state(self.ctx) })
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { wafer_ctx_destroy(self.ctx) };
    }
}

/// `libwafer_rccl.so` (include/wafer_rccl.h): the halo / all-reduce hooks of a multi-GPU run served by RCCL's C API, for a
/// host that does not want to write them again, and the device-side all-reduce of the 1 + k sums
/// (include/wafer_mailbox.h, part of `libwafer_hip.so`).
#[repr(C)]
pub struct wafer_mailbox {
    _private: [u8; 0],
}

#[link(name = "wafer_rccl")]
extern "C" {
    pub fn wafer_rccl_last_error() -> *const c_char;
    pub fn wafer_rccl_unique_id_bytes() -> c_int;
    pub fn wafer_rccl_unique_id(out: *mut c_void) -> c_int;
    pub fn wafer_rccl_attach(
        ctx: *mut wafer_ctx, rank: c_int, world: c_int, unique_id: *const c_void, lower_override: c_int, upper_override: c_int,
        handle_out: *mut *mut c_void,
    ) -> c_int;
    pub fn wafer_rccl_warm_up(handle: *mut c_void, scratch: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn wafer_rccl_use_mailbox(handle: *mut c_void, mailbox: *mut c_void) -> c_int;
    pub fn wafer_rccl_allreduce_now(handle: *mut c_void, dev_ptr: *mut c_void, count: usize, stream: *mut c_void) -> c_int;
    pub fn wafer_rccl_halo_calls(handle: *mut c_void) -> std::os::raw::c_long;
    pub fn wafer_rccl_comm_info(
        handle: *mut c_void, nranks: *mut c_int, rank: *mut c_int, lower: *mut c_int, upper: *mut c_int, version: *mut c_int,
    ) -> c_int;
    pub fn wafer_rccl_detach(ctx: *mut wafer_ctx, handle: *mut c_void) -> c_int;
}

#[link(name = "wafer_hip")]
extern "C" {
    pub fn wafer_mailbox_create(rank: c_int, world: c_int, device: c_int, out: *mut *mut wafer_mailbox) -> c_int;
    pub fn wafer_mailbox_handle(mb: *mut wafer_mailbox, handle_out: *mut c_void) -> c_int;
    pub fn wafer_mailbox_connect(mb: *mut wafer_mailbox, all_handles: *const c_void) -> c_int;
    pub fn wafer_mailbox_allreduce(mailbox: *mut c_void, dev_ptr: *mut c_void, count: usize, hip_stream: *mut c_void) -> c_int;
    pub fn wafer_mailbox_check(mb: *mut wafer_mailbox) -> c_int;
    pub fn wafer_mailbox_destroy(mb: *mut wafer_mailbox) -> c_int;
}
