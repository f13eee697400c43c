//! `speechsauce-amd`: the hot-path API of the `speechsauce` crate on an MI355X.
//!
//! Drop-in by module path: the crate has the reference's public modules (`speechsauce/src/lib.rs:2-6`) -- `config`, `feature`,
//! `functions`, `processing`, `util` -- so with `speechsauce = { package = "speechsauce-amd", path = ".." }` in Cargo.toml an
//! existing `use speechsauce::feature::mfcc;` keeps compiling with no source edit.  Every item also stays at the crate root.
//! Signatures follow the reference one for one:
//!   speechsauce::feature::mfcc(ArrayView1<f32>, &SpeechConfig) -> Array2<f32>          (feature.rs:99)
//!   speechsauce::feature::mfe(ArrayView1<f32>, &SpeechConfig) -> (Array2, Array1)       (feature.rs:200)
//!   speechsauce::feature::mel_spectrogram1 / mel_spectrogram2                           (feature.rs:151,163)
//!   speechsauce::processing::preemphasis(Array1<f32>, isize, f32) -> Array1<f32>        (processing.rs:31)
//!   speechsauce::processing::stack_frames(signal, sample_rate, frame_length, frame_stride, filter, zero_padding)
//!   speechsauce::processing::power_spectrum(frames, fft_points)                         (processing.rs:65,179)
//!   speechsauce::functions::stft1 / stft2 -> Array2 / Array3<Complex32>                 (functions.rs:199,86)
//!   speechsauce::config::{SpeechConfig, SpeechConfigBuilder}                            (config.rs:10-190)
//! The host side here owns what the north-star assigns to Rust: ndarray I/O (contiguity, shapes,
//! allocation of the outputs) and the framing parameters; every numeric step runs in the HIP kernels
//! behind the `extern "C"` layer of include/speechsauce_amd.h.  Where the reference panics these
//! functions panic too, with the library's error text (`expect`-style), so behaviour under bad input is
//! unchanged for existing callers; `try_*` variants return `Result`.
//!
//! `SpeechConfig` is `Clone` like the reference's (config.rs:98): clones share one immutable device handle (`Arc`), which is
//! released when the last clone drops.  Its public fields are the reference's plain-data fields (config.rs:100-126), including
//! `window_size_half`, `frame_size`, `wnorm` and `window`; the reference's five fields that hold third-party plan objects or
//! streaming state (`analysis_mem`, `dct_handler`, `fft_handler`, `fft_forward`, `analysis_scratch`) have no counterpart: the
//! plans live on the device and the STFT starts from zero state per call.
//!
//! This file is shipped uncompiled (the build image has no Rust toolchain): NEVER COMPILED, checked textually by
//! tests/test_binding_mirrors.py.

use ndarray::{Array, Array1, Array2, Array3, ArrayView1, ArrayView2, Dimension};
use num_complex::Complex32;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_long, c_void};
use std::sync::Arc;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct SsParams {
    pub struct_size: u32,
    pub sample_rate: u32,
    pub fft_points: u32,
    pub frame_length: f32,
    pub frame_stride: f32,
    pub num_cepstral: u32,
    pub num_filters: u32,
    pub low_frequency: f32,
    pub high_frequency: f32,
    pub dc_elimination: i32,
    pub framing: i32,
    pub spectrum_exponent: i32,
    pub dct_norm: i32,
    pub dct2_gain: f32,
    pub mfcc_window: i32,
    pub preemph_coef: f32,
    pub preemph_shift: i32,
    /// librosa-compatible variants (0 = reference mode): SS_MEL_*, SS_MEL_NORM_*, SS_PAD_*
    pub mel_scale: i32,
    pub mel_norm: i32,
    pub pad_mode: i32,
}

#[repr(C)]
pub struct SsConfig {
    _private: [u8; 0],
}

extern "C" {
    fn ss_params_default(p: *mut SsParams, sample_rate: u32) -> c_int;
    fn ss_config_create(p: *const SsParams, out: *mut *mut SsConfig) -> c_int;
    fn ss_config_destroy(cfg: *mut SsConfig);
    fn ss_num_frames(p: *const SsParams, n_samples: usize, n_frames: *mut usize) -> c_int;
    fn ss_stft_rows(p: *const SsParams, n_samples: usize, rows: *mut usize, real_rows: *mut usize) -> c_int;
    fn ss_stft_sizes(p: *const SsParams, hop: *mut usize, n_pad: *mut usize, wnorm: *mut f32) -> c_int;  // (stft_sizes below)
    fn ss_vorbis_window(n: usize, w: *mut f32) -> c_int;
    fn ss_mfcc(cfg: *const SsConfig, x: *const f32, n: usize, out: *mut f32) -> c_int;
    fn ss_mfe(cfg: *const SsConfig, x: *const f32, n: usize, feat: *mut f32, energy: *mut f32) -> c_int;
    fn ss_lmfe(cfg: *const SsConfig, x: *const f32, n: usize, feat: *mut f32) -> c_int;
    fn ss_mel_spectrogram(cfg: *const SsConfig, x: *const f32, channels: usize, n: usize, out: *mut f32) -> c_int;
    fn ss_mfcc_batch(cfg: *const SsConfig, x: *const f32, batch: usize, n: usize, ld: usize, out: *mut f32) -> c_int;
    fn ss_mfcc_batch_device(cfg: *const SsConfig, d_x: *const f32, batch: usize, n: usize, ld: usize, d_out: *mut f32,
                            stream: *mut c_void) -> c_int;
    fn ss_mfcc_batches_device(cfg: *const SsConfig, n_batches: usize, d_x: *const *const f32, batch: *const usize, n: usize, ld: usize,
                              d_out: *const *mut f32, stream: *mut c_void) -> c_int;
    fn ss_preemphasis(x: *const f32, n: usize, shift: c_long, cof: f32, y: *mut f32) -> c_int;
    fn ss_frame_sizes(p: *const SsParams, frame_len: *mut usize, frame_step: *mut usize) -> c_int;
    fn ss_stft(cfg: *const SsConfig, x: *const f32, channels: usize, n: usize, out: *mut f32) -> c_int;
    fn ss_stack_frames(cfg: *const SsConfig, x: *const f32, n: usize, frames: *mut f32) -> c_int;
    fn ss_stack_frames_shape(n_samples: usize, sample_rate: u32, frame_length: f32, frame_stride: f32, zero_padding: c_int,
                             num_frames: *mut usize, frame_len: *mut usize) -> c_int;
    fn ss_stack_frames_signal(x: *const f32, n_samples: usize, sample_rate: u32, frame_length: f32, frame_stride: f32,
                              window: *const f32, zero_padding: c_int, frames: *mut f32) -> c_int;
    fn ss_power_spectrum_frames(cfg: *const SsConfig, frames: *const f32, rows: usize, cols: usize, p_out: *mut f32) -> c_int;
    fn ss_power_spectrum(cfg: *const SsConfig, x: *const f32, n: usize, p_out: *mut f32) -> c_int;
    fn ss_config_device_status(cfg: *const SsConfig) -> c_int;
    fn ss_cmvn(vec: *const f32, rows: usize, cols: usize, variance_normalization: c_int, out: *mut f32) -> c_int;
    fn ss_cmvnw(vec: *const f32, rows: usize, cols: usize, win_size: usize, variance_normalization: c_int, out: *mut f32) -> c_int;
    fn ss_derivative_extraction(feat: *const f32, rows: usize, cols: usize, delta_windows: usize, out: *mut f32) -> c_int;
    fn ss_extract_derivative_feature(feat: *const f32, rows: usize, cols: usize, cube: *mut f32) -> c_int;
    fn ss_shard_bounds(n_items: usize, world: c_int, rank: c_int, lo: *mut usize, hi: *mut usize) -> c_int;
    fn ss_all_gather_features(nccl_comm: *mut c_void, d_block: *const f32, elems_per_rank: usize, d_out: *mut f32,
                              stream: *mut c_void) -> c_int;
    fn ss_gather_features(nccl_comm: *mut c_void, d_block: *const f32, elems_per_rank: usize, d_out: *mut f32, root: c_int,
                          rank: c_int, world: c_int, stream: *mut c_void) -> c_int;
    fn ss_last_error_string() -> *const c_char;
}

/// Contiguous clip shard of rank `rank` of `world` (one process or thread per GPU; clips are independent units).
pub fn shard_bounds(n_items: usize, world: usize, rank: usize) -> (usize, usize) {
    let (mut lo, mut hi) = (0usize, 0usize);
    check(unsafe { ss_shard_bounds(n_items, world as c_int, rank as c_int, &mut lo, &mut hi) }).expect("shard_bounds");
    (lo, hi)
}

/// RCCL all-gather of the ranks' feature blocks (device pointers, `comm` = the caller's ncclComm_t): the north-star's
/// "gather over xGMI of the final [n_frames x n_mfcc] blocks".
///
/// # Safety
/// `d_block` / `d_out` must be device buffers of `elems_per_rank` / `world * elems_per_rank` floats on the current device.
pub unsafe fn all_gather_features(comm: *mut c_void, d_block: *const f32, elems_per_rank: usize, d_out: *mut f32,
                                  stream: *mut c_void) -> Result<(), Error> {
    check(ss_all_gather_features(comm, d_block, elems_per_rank, d_out, stream))
}

/// RCCL gather to `root` (the north-star's collective): the root receives `[world x elems_per_rank]` in rank order, one
/// direct xGMI transfer per peer; the other ranks only send and may pass a null `d_out`.
///
/// # Safety
/// As `all_gather_features`; `comm` must come from the RCCL library this crate's C side resolves (see speechsauce_amd.h).
#[allow(clippy::too_many_arguments)]
pub unsafe fn gather_features(comm: *mut c_void, d_block: *const f32, elems_per_rank: usize, d_out: *mut f32, root: usize,
                              rank: usize, world: usize, stream: *mut c_void) -> Result<(), Error> {
    check(ss_gather_features(comm, d_block, elems_per_rank, d_out, root as c_int, rank as c_int, world as c_int, stream))
}

/// SS_ERR_ARG of include/speechsauce_amd.h
pub const SS_ERR_ARG: i32 = 3;

#[derive(Debug)]
pub struct Error {
    pub status: i32,
    pub detail: String,
}

fn check(status: c_int) -> Result<(), Error> {
    if status == 0 {
        return Ok(());
    }
    let detail = unsafe { CStr::from_ptr(ss_last_error_string()) }.to_string_lossy().into_owned();
    Err(Error { status, detail })
}

/// config.rs:10-97
#[derive(Clone)]
pub struct SpeechConfigBuilder {
    p: SsParams,
}
impl Default for SpeechConfigBuilder {
    /// config.rs:10 derives Default (every field zero); `SpeechConfig::builder()` hands that out (config.rs:187-189)
    fn default() -> Self {
        let mut b = SpeechConfigBuilder::new(16000);  // (the switches of ss_params keep their reference-mode defaults)
        b.p.sample_rate = 0;
        b.p.fft_points = 0;
        b.p.frame_length = 0.0;
        b.p.frame_stride = 0.0;
        b.p.num_cepstral = 0;
        b.p.num_filters = 0;
        b.p.high_frequency = 0.0;
        b.p.dc_elimination = 0;
        b
    }
}

impl SpeechConfigBuilder {
    pub fn new(sample_rate: usize) -> Self {
        let mut p = std::mem::MaybeUninit::<SsParams>::uninit();
        unsafe {
            check(ss_params_default(p.as_mut_ptr(), sample_rate as u32)).expect("ss_params_default");
            SpeechConfigBuilder { p: p.assume_init() }
        }
    }
    pub fn high_freq(mut self, v: f32) -> Self { self.p.high_frequency = v; self }
    pub fn low_freq(mut self, v: f32) -> Self { self.p.low_frequency = v; self }
    pub fn dc_elimination(mut self, v: bool) -> Self { self.p.dc_elimination = v as i32; self }
    pub fn num_cepstral(mut self, v: usize) -> Self { self.p.num_cepstral = v as u32; self }
    pub fn frame_stride(mut self, v: f32) -> Self { self.p.frame_stride = v; self }
    pub fn frame_length(mut self, v: f32) -> Self { self.p.frame_length = v; self }
    pub fn fft_points(mut self, v: usize) -> Self { self.p.fft_points = v as u32; self }
    pub fn build(self) -> SpeechConfig { SpeechConfig::from_params(self.p).expect("SpeechConfig::new") }
}

/// The device-side half of a config: the opaque handle of include/speechsauce_amd.h, destroyed with the last clone.
struct Handle(*mut SsConfig);
// the handle is immutable after creation and safe for concurrent calls (speechsauce_amd.h, "Threading")
unsafe impl Send for Handle {}
unsafe impl Sync for Handle {}
impl Drop for Handle {
    fn drop(&mut self) { unsafe { ss_config_destroy(self.0) } }
}

/// config.rs:98-131.  Immutable after creation (no STFT carry-over), hence Send + Sync; `Clone` shares the device handle.
#[derive(Clone)]
pub struct SpeechConfig {
    pub sample_rate: usize,
    pub window_size: usize,
    pub window_size_half: usize,
    pub frame_length: f32,
    pub frame_stride: f32,
    pub num_cepstral: usize,
    pub num_filters: usize,
    pub low_frequency: f32,
    pub high_frequency: f32,
    pub freq_size: usize,
    /// samples per STFT chunk = trunc(frame_length * sample_rate) (config.rs:154), as the library computes it
    pub frame_size: usize,
    pub dc_elimination: bool,
    /// 2 * frame_size / fft_points^2 (config.rs:178)
    pub wnorm: f32,
    /// the Vorbis power-complementary window of `window_size` points (config.rs:151-160)
    pub window: Vec<f32>,
    params: SsParams,
    handle: Arc<Handle>,
}

impl SpeechConfig {
    /// config.rs:140-150
    #[allow(clippy::too_many_arguments)]
    pub fn new(sample_rate: usize, fft_points: usize, frame_length: f32, frame_stride: f32, num_cepstral: usize,
               num_filters: usize, low_frequency: f32, high_frequency: f32, dc_elimination: bool) -> Self {
        let mut b = SpeechConfigBuilder::new(sample_rate).p;
        b.fft_points = fft_points as u32;
        b.frame_length = frame_length;
        b.frame_stride = frame_stride;
        b.num_cepstral = num_cepstral as u32;
        b.num_filters = num_filters as u32;
        b.low_frequency = low_frequency;
        b.high_frequency = high_frequency;
        b.dc_elimination = dc_elimination as i32;
        Self::from_params(b).expect("SpeechConfig::new")
    }
    pub fn from_params(p: SsParams) -> Result<Self, Error> {
        let mut h: *mut SsConfig = std::ptr::null_mut();
        check(unsafe { ss_config_create(&p, &mut h) })?;
        let handle = Arc::new(Handle(h));  // from here on an early return releases the handle
        // config.rs:154 / :178 in the reference's own f32 arithmetic (every config has these two, not only the STFT-capable
        // ones ss_stft_sizes answers for; for those the library's values are the same numbers)
        let frame_size = (p.frame_length * p.sample_rate as f32) as usize;
        let wnorm = 1.0 / ((p.fft_points as usize).pow(2) as f32 / (2 * frame_size) as f32);
        let mut window = vec![0f32; p.fft_points as usize];
        check(unsafe { ss_vorbis_window(window.len(), window.as_mut_ptr()) })?;
        Ok(SpeechConfig {
            sample_rate: p.sample_rate as usize,
            window_size: p.fft_points as usize,
            window_size_half: p.fft_points as usize / 2,
            frame_size,
            wnorm,
            window,
            frame_length: p.frame_length,
            frame_stride: p.frame_stride,
            num_cepstral: p.num_cepstral as usize,
            num_filters: p.num_filters as usize,
            low_frequency: p.low_frequency,
            high_frequency: p.high_frequency,
            freq_size: p.fft_points as usize / 2 + 1,
            dc_elimination: p.dc_elimination != 0,
            params: p,
            handle,
        })
    }
    /// config.rs:187-189: the all-zero builder (set every field before `build()`, as with the reference)
    pub fn builder() -> SpeechConfigBuilder { SpeechConfigBuilder::default() }
}
impl Default for SpeechConfig {
    fn default() -> Self { SpeechConfigBuilder::new(16000).build() }
}
impl SpeechConfig {
    /// the raw handle for the `extern "C"` calls below (valid while any clone lives)
    fn raw(&self) -> *const SsConfig { self.handle.0 }
}

/// The samples of a 1-D view as one contiguous run: borrowed when the view already is one, copied otherwise (the reference accepts
/// any stride, `as_array()`).  `ArrayView::to_slice(&self) -> Option<&'a [A]>` hands out the VIEW's lifetime; `ArrayBase::as_slice`
/// would borrow from the by-value parameter, a local (E0515).
fn contiguous<'a>(signal: ArrayView1<'a, f32>) -> std::borrow::Cow<'a, [f32]> {
    match signal.to_slice() {
        Some(s) => std::borrow::Cow::Borrowed(s),
        None => std::borrow::Cow::Owned(signal.to_vec()),
    }
}

/// feature.rs:99-148
pub fn try_mfcc(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> Result<Array2<f32>, Error> {
    let x = contiguous(signal);
    let mut t = 0usize;
    check(unsafe { ss_num_frames(&cfg.params, x.len(), &mut t) })?;
    let mut out = Array2::<f32>::zeros((t, cfg.num_cepstral));
    check(unsafe { ss_mfcc(cfg.raw(), x.as_ptr(), x.len(), out.as_mut_ptr()) })?;
    Ok(out)
}
pub fn mfcc(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> Array2<f32> { try_mfcc(signal, cfg).expect("mfcc") }

/// feature.rs:200-233
pub fn try_mfe(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> Result<(Array2<f32>, Array1<f32>), Error> {
    let x = contiguous(signal);
    let mut t = 0usize;
    check(unsafe { ss_num_frames(&cfg.params, x.len(), &mut t) })?;
    let mut feat = Array2::<f32>::zeros((t, cfg.num_filters));
    let mut en = Array1::<f32>::zeros(t);
    check(unsafe { ss_mfe(cfg.raw(), x.as_ptr(), x.len(), feat.as_mut_ptr(), en.as_mut_ptr()) })?;
    Ok((feat, en))
}
pub fn mfe(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> (Array2<f32>, Array1<f32>) { try_mfe(signal, cfg).expect("mfe") }

/// feature.rs:242-245 (private there; README.md:14 lists log mel-filterbank energies as a supported feature)
pub fn try_lmfe(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> Result<Array2<f32>, Error> {
    let x = contiguous(signal);
    let mut t = 0usize;
    check(unsafe { ss_num_frames(&cfg.params, x.len(), &mut t) })?;
    let mut feat = Array2::<f32>::zeros((t, cfg.num_filters));
    check(unsafe { ss_lmfe(cfg.raw(), x.as_ptr(), x.len(), feat.as_mut_ptr()) })?;
    Ok(feat)
}
pub fn lmfe(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> Array2<f32> { try_lmfe(signal, cfg).expect("lmfe") }

/// feature.rs:163-174: [channels, samples] -> [channels, n_mels, rows]; rows need contiguous storage like
/// the reference's `as_slice().expect(..)` (functions.rs:104)
pub fn try_mel_spectrogram2(signal: ArrayView2<f32>, cfg: &SpeechConfig) -> Result<Array3<f32>, Error> {
    let owned = signal.as_standard_layout();
    let (ch, n) = owned.dim();
    let (mut rows, mut _real) = (0usize, 0usize);
    check(unsafe { ss_stft_rows(&cfg.params, n, &mut rows, &mut _real) })?;
    let mut out = Array3::<f32>::zeros((ch, cfg.num_filters, rows));
    check(unsafe { ss_mel_spectrogram(cfg.raw(), owned.as_ptr(), ch, n, out.as_mut_ptr()) })?;
    Ok(out)
}
pub fn mel_spectrogram2(signal: ArrayView2<f32>, cfg: &SpeechConfig) -> Array3<f32> {
    try_mel_spectrogram2(signal, cfg).expect("mel_spectrogram2")
}
/// feature.rs:151-162 (one channel)
pub fn mel_spectrogram1(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> Array2<f32> {
    let x = contiguous(signal);
    let v = ArrayView2::from_shape((1, x.len()), &x[..]).expect("shape");  // (&x[..]: a plain &[f32], no deref coercion of &Cow needed)
    let out = mel_spectrogram2(v, cfg);
    let (_, m, r) = out.dim();
    out.into_shape((m, r)).expect("shape")
}

/// Batch form with no counterpart in the reference: [batch, samples] -> [batch, frames, n_cepstral] in one launch.
pub fn try_mfcc_batch(signals: ArrayView2<f32>, cfg: &SpeechConfig) -> Result<Array3<f32>, Error> {
    let owned = signals.as_standard_layout();
    let (b, n) = owned.dim();
    let mut t = 0usize;
    check(unsafe { ss_num_frames(&cfg.params, n, &mut t) })?;
    let mut out = Array3::<f32>::zeros((b, t, cfg.num_cepstral));
    check(unsafe { ss_mfcc_batch(cfg.raw(), owned.as_ptr(), b, n, n, out.as_mut_ptr()) })?;
    Ok(out)
}

/// Device-resident batch: raw device pointers + a hipStream_t, asynchronous.
/// # Safety
/// `d_x` / `d_out` must be device allocations of the right size on the device the config was created on.
pub unsafe fn mfcc_batch_device(cfg: &SpeechConfig, d_x: *const f32, batch: usize, n: usize, ld: usize, d_out: *mut f32,
                                stream: *mut c_void) -> Result<(), Error> {
    check(ss_mfcc_batch_device(cfg.raw(), d_x, batch, n, ld, d_out, stream))
}

/// Several device-resident batches per call (ABI 7): batch `b` is `batch[b]` clips of `n` samples at `d_x[b]` (row stride `ld`), its
/// features go to `d_out[b]`.  One persistent launch for up to 8 batches where the configuration's kernel takes a batch table;
/// bit-identical to `batch.len()` calls of `mfcc_batch_device`.
/// # Safety
/// As `mfcc_batch_device`, for every batch.
pub unsafe fn mfcc_batches_device(cfg: &SpeechConfig, d_x: &[*const f32], batch: &[usize], n: usize, ld: usize, d_out: &[*mut f32],
                                  stream: *mut c_void) -> Result<(), Error> {
    if d_x.len() != batch.len() || d_out.len() != batch.len() {
        return Err(Error { status: SS_ERR_ARG, detail: "d_x, batch and d_out must have one entry per batch".to_string() });
    }
    check(ss_mfcc_batches_device(cfg.raw(), batch.len(), d_x.as_ptr(), batch.as_ptr(), n, ld, d_out.as_ptr(), stream))
}

/// functions.rs:86-123: `[channels, samples]` -> `Array3<Complex32>` `[channels, rows, freq_size]`
pub fn try_stft2(input: ArrayView2<f32>, cfg: &SpeechConfig) -> Result<Array3<Complex32>, Error> {
    let owned = input.as_standard_layout();
    let (ch, n) = owned.dim();
    let (mut rows, mut _real) = (0usize, 0usize);
    check(unsafe { ss_stft_rows(&cfg.params, n, &mut rows, &mut _real) })?;
    let mut out = Array3::<Complex32>::zeros((ch, rows, cfg.freq_size));
    // Complex32 is #[repr(C)] { re: f32, im: f32 }: the interleaved block the ABI writes
    check(unsafe { ss_stft(cfg.raw(), owned.as_ptr(), ch, n, out.as_mut_ptr() as *mut f32) })?;
    Ok(out)
}
pub fn stft2(input: ArrayView2<f32>, cfg: &SpeechConfig) -> Array3<Complex32> { try_stft2(input, cfg).expect("stft2") }

/// functions.rs:199-233 (one channel): `Array2<Complex32>` `[rows, freq_size]`
pub fn stft1(input: ArrayView1<f32>, cfg: &SpeechConfig) -> Array2<Complex32> {
    let x = contiguous(input);
    let v = ArrayView2::from_shape((1, x.len()), &x[..]).expect("shape");
    let out = stft2(v, cfg);
    let (_, r, f) = out.dim();
    out.into_shape((r, f)).expect("shape")
}

/// processing.rs:65-129 with the reference's own signature:
/// `stack_frames(signal, sample_rate, frame_length, frame_stride, filter, zero_padding) -> Array2<f32>` -- a caller of
/// `speechsauce::processing::stack_frames` switches the `use` line and nothing else.  `filter` is called with the frame
/// length and its row 0 multiplies every frame (what `repeat_axis(filt, Axis(0), numframes)` does with the reference's
/// `(1, frame_len)` array, processing.rs:122-126); the framing is the documented contract (frame t = samples
/// `t * step .. t * step + frame_len`), see SURVEY.md Q1 for what the reference's copy loop does instead.
pub fn try_stack_frames(signal: ArrayView1<f32>, sample_rate: usize, frame_length: f32, frame_stride: f32,
                        filter: Option<fn(usize) -> Array2<f32>>, zero_padding: bool) -> Result<Array2<f32>, Error> {
    let x = contiguous(signal);
    let (mut t, mut flen) = (0usize, 0usize);
    if sample_rate > u32::MAX as usize {
        return Err(Error { status: SS_ERR_ARG, detail: "sample_rate does not fit 32 bits".to_string() });
    }
    check(unsafe { ss_stack_frames_shape(x.len(), sample_rate as u32, frame_length, frame_stride, zero_padding as c_int, &mut t, &mut flen) })?;
    let window: Option<Vec<f32>> = match filter {
        None => None,
        Some(f) => {
            let w = f(flen);
            // row 0 of the (1, frame_len) array; a (frame_len, 1) column (feature.rs:176-178 `_f_it`) is read down its column
            let v: Vec<f32> = if w.nrows() == 1 { w.row(0).to_vec() } else { w.column(0).to_vec() };
            if v.len() != flen {  // the try_ form reports, it does not panic
                return Err(Error { status: SS_ERR_ARG, detail: format!("filter({}) gave {} values", flen, v.len()) });
            }
            Some(v)
        }
    };
    let mut out = Array2::<f32>::zeros((t, flen));
    let wptr = window.as_ref().map_or(std::ptr::null(), |w| w.as_ptr());
    check(unsafe { ss_stack_frames_signal(x.as_ptr(), x.len(), sample_rate as u32, frame_length, frame_stride, wptr, zero_padding as c_int, out.as_mut_ptr()) })?;
    Ok(out)
}
pub fn stack_frames(signal: ArrayView1<f32>, sample_rate: usize, frame_length: f32, frame_stride: f32,
                    filter: Option<fn(usize) -> Array2<f32>>, zero_padding: bool) -> Array2<f32> {
    try_stack_frames(signal, sample_rate, frame_length, frame_stride, filter, zero_padding).expect("stack_frames")
}

/// The same stage with the framing taken from a config (its `framing` / `mfcc_window` / `pad_mode` switches apply: literal and
/// centred framing exist only in this form).  An extra beside the reference's signature above.
pub fn try_stack_frames_with(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> Result<Array2<f32>, Error> {
    let x = contiguous(signal);
    let (mut t, mut flen, mut _step) = (0usize, 0usize, 0usize);
    check(unsafe { ss_num_frames(&cfg.params, x.len(), &mut t) })?;
    check(unsafe { ss_frame_sizes(&cfg.params, &mut flen, &mut _step) })?;
    let mut out = Array2::<f32>::zeros((t, flen));
    check(unsafe { ss_stack_frames(cfg.raw(), x.as_ptr(), x.len(), out.as_mut_ptr()) })?;
    Ok(out)
}
pub fn stack_frames_with(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> Array2<f32> { try_stack_frames_with(signal, cfg).expect("stack_frames") }

/// processing.rs:179-181 with the reference's own signature: `power_spectrum(frames: Array2<f32>, fft_points: usize)`.
/// Only `fft_points` matters to this stage; the config it runs on is kept per `fft_points` in a small process-wide cache
/// (the Python front memoises the same way), so repeated calls cost one lookup.
pub fn try_power_spectrum(frames: Array2<f32>, fft_points: usize) -> Result<Array2<f32>, Error> {
    use std::collections::HashMap;
    use std::sync::{Mutex, OnceLock};
    struct Kept(*mut SsConfig);
    unsafe impl Send for Kept {}  // the handle is immutable after creation and thread-safe (speechsauce_amd.h)
    static CACHE: OnceLock<Mutex<HashMap<usize, Kept>>> = OnceLock::new();
    let cache = CACHE.get_or_init(|| Mutex::new(HashMap::new()));
    let handle = {
        let mut map = cache.lock().expect("power_spectrum config cache");
        if let Some(h) = map.get(&fft_points) {
            h.0
        } else {
            // a config that validates for this FFT length: a frame of half its length, a bank that fits its spectrum
            let mut p = SpeechConfigBuilder::new(16000).p;
            p.fft_points = fft_points as u32;
            p.frame_length = fft_points as f32 / 32000.0;
            p.frame_stride = fft_points as f32 / 64000.0;
            p.num_filters = std::cmp::max(1, std::cmp::min(40, fft_points / 8)) as u32;
            p.num_cepstral = std::cmp::min(13, p.num_filters);
            let mut h: *mut SsConfig = std::ptr::null_mut();
            check(unsafe { ss_config_create(&p, &mut h) })?;
            map.insert(fft_points, Kept(h));  // kept for the life of the process
            h
        }
    };
    let x = frames.as_standard_layout();
    let (rows, cols) = x.dim();
    let mut out = Array2::<f32>::zeros((rows, fft_points / 2 + 1));
    check(unsafe { ss_power_spectrum_frames(handle, x.as_ptr(), rows, cols, out.as_mut_ptr()) })?;
    Ok(out)
}
pub fn power_spectrum(frames: Array2<f32>, fft_points: usize) -> Array2<f32> { try_power_spectrum(frames, fft_points).expect("power_spectrum") }

/// The same stage on an existing config (`fft_points` is the config's).  An extra beside the reference's signature above.
pub fn try_power_spectrum_with(frames: Array2<f32>, cfg: &SpeechConfig) -> Result<Array2<f32>, Error> {
    let x = frames.as_standard_layout();
    let (rows, cols) = x.dim();
    let mut out = Array2::<f32>::zeros((rows, cfg.freq_size));
    check(unsafe { ss_power_spectrum_frames(cfg.raw(), x.as_ptr(), rows, cols, out.as_mut_ptr()) })?;
    Ok(out)
}
pub fn power_spectrum_with(frames: Array2<f32>, cfg: &SpeechConfig) -> Array2<f32> { try_power_spectrum_with(frames, cfg).expect("power_spectrum") }

/// stack_frames + power_spectrum fused, as mfe uses them (feature.rs:203-214): `[frames, freq_size]`
pub fn try_power_spectrum_of_signal(signal: ArrayView1<f32>, cfg: &SpeechConfig) -> Result<Array2<f32>, Error> {
    let x = contiguous(signal);
    let mut t = 0usize;
    check(unsafe { ss_num_frames(&cfg.params, x.len(), &mut t) })?;
    let mut out = Array2::<f32>::zeros((t, cfg.freq_size));
    check(unsafe { ss_power_spectrum(cfg.raw(), x.as_ptr(), x.len(), out.as_mut_ptr()) })?;
    Ok(out)
}

/// (hop, n_pad, wnorm) of the STFT path as the library computes them (functions.rs:96-97, config.rs:178); `Err` for a config whose
/// `fft_points < 2 * frame_size` (the reference's `frame_analysis` underflows there, functions.rs:136).
pub fn stft_sizes(cfg: &SpeechConfig) -> Result<(usize, usize, f32), Error> {
    let (mut hop, mut n_pad, mut wnorm) = (0usize, 0usize, 0f32);
    check(unsafe { ss_stft_sizes(&cfg.params, &mut hop, &mut n_pad, &mut wnorm) })?;
    Ok((hop, n_pad, wnorm))
}

/// Status of the asynchronous launches made on `cfg` (`Err` with status 6 after a device-side protocol error; cleared by the call).
pub fn device_status(cfg: &SpeechConfig) -> Result<(), Error> { check(unsafe { ss_config_device_status(cfg.raw()) }) }

/// processing.rs:31-53
pub fn preemphasis(signal: Array1<f32>, shift: isize, cof: f32) -> Array1<f32> {
    let x = signal.as_standard_layout();
    let mut y = Array1::<f32>::zeros(x.len());
    check(unsafe { ss_preemphasis(x.as_ptr(), x.len(), shift as c_long, cof, y.as_mut_ptr()) }).expect("preemphasis");
    y
}

/// processing.rs:265-300
pub fn cmvn(vec: ArrayView2<f32>, variance_normalization: bool) -> Array2<f32> {
    let x = vec.as_standard_layout();
    let (rows, cols) = x.dim();
    let mut out = Array2::<f32>::zeros((rows, cols));
    check(unsafe { ss_cmvn(x.as_ptr(), rows, cols, variance_normalization as c_int, out.as_mut_ptr()) }).expect("cmvn");
    out
}

/// processing.rs:315-371 (panics on an even `win_size`, like the reference's assert)
pub fn cmvnw(vec: Array2<f32>, win_size: usize, variance_normalization: bool) -> Array2<f32> {
    let x = vec.as_standard_layout();
    let (rows, cols) = x.dim();
    let mut out = Array2::<f32>::zeros((rows, cols));
    check(unsafe { ss_cmvnw(x.as_ptr(), rows, cols, win_size, variance_normalization as c_int, out.as_mut_ptr()) })
        .expect("Windows size must be odd!");
    out
}

/// processing.rs:222-254
pub fn derivative_extraction(feat: &Array2<f32>, delta_windows: usize) -> Array2<f32> {
    let x = feat.as_standard_layout();
    let (rows, cols) = x.dim();
    let mut out = Array2::<f32>::zeros((rows, cols));
    check(unsafe { ss_derivative_extraction(x.as_ptr(), rows, cols, delta_windows, out.as_mut_ptr()) }).expect("derivative_extraction");
    out
}

/// feature.rs:253-269: N x M x 3 cube of static, first and second derivative features
pub fn extract_derivative_feature(feature: Array2<f32>) -> Array3<f32> {
    let x = feature.as_standard_layout();
    let (rows, cols) = x.dim();
    let mut cube = Array3::<f32>::zeros((rows, cols, 3));
    check(unsafe { ss_extract_derivative_feature(x.as_ptr(), rows, cols, cube.as_mut_ptr()) }).expect("extract_derivative_feature");
    cube
}

// ---------------------------------------------------------------------------------------------------------------------------
// Scalar helpers the reference exports from `functions` (functions.rs:19-71) and `util` (util.rs:372-381): host arithmetic on a
// handful of values, written here in plain Rust (nothing to launch a kernel for); the filterbank the kernels use is built by the
// library from the same formulas in the same f32 operation order (ss_filterbank).
// ---------------------------------------------------------------------------------------------------------------------------

/// functions.rs:19-21: Hz -> mel, 1127 ln(1 + f / 700)
pub fn frequency_to_mel(f: f32) -> f32 { 1127.0 * (1.0 + f / 700.0).ln() }
/// functions.rs:23-29
pub fn frequency_arr_to_mel<D: Dimension>(freq: Array<f32, D>) -> Array<f32, D> { freq.mapv(frequency_to_mel) }
/// functions.rs:31-34 (the reference's unused type parameter is kept so `mel_to_frequency::<Ix1>(m)` still compiles)
pub fn mel_to_frequency<D>(mel: f32) -> f32 { 700.0 * ((mel / 1127.0).exp() - 1.0) }
/// functions.rs:36-41
pub fn mel_arr_to_frequency<D: Dimension>(mel: Array<f32, D>) -> Array<f32, D> { mel.mapv(|m| 700.0 * ((m / 1127.0).exp() - 1.0)) }
/// functions.rs:43-60: triangular weight over the half-open range [left, right); at x == middle the falling branch wins (= 1)
pub fn triangle(arr: Array1<f32>, left: f32, middle: f32, right: f32) -> Array1<f32> {
    arr.mapv(|x| {
        if !(x >= left && x < right) {
            return 0.0;
        }
        let mut w = 0.0;
        if x <= middle { w = (x - left) / (middle - left); }
        if x >= middle { w = (right - x) / (right - middle); }
        w
    })
}
/// functions.rs:66-71: exact zeros become f32::EPSILON
pub fn zero_handling<D: Dimension>(x: Array<f32, D>) -> Array<f32, D> { x.mapv(|v| if v == 0.0 { f32::EPSILON } else { v }) }

/// util.rs:372-381: natural logarithm of every element, in place.  Two type parameters with the reference's bounds, so that caller
/// code naming `ArrayLog<f32, Ix2>` (or bounding on it) resolves unchanged.
pub trait ArrayLog<A: num_traits::real::Real, I: Dimension> {
    fn log(self) -> Array<A, I>;
}
impl<A: num_traits::real::Real, I: Dimension> ArrayLog<A, I> for Array<A, I> {
    fn log(mut self) -> Array<A, I> {
        self.map_inplace(|n| *n = (*n).ln());
        self
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The reference's module paths (speechsauce/src/lib.rs:2-6).  Re-exports only: one implementation, two spellings.
// ---------------------------------------------------------------------------------------------------------------------------

/// speechsauce::config (config.rs)
pub mod config {
    pub use super::{SpeechConfig, SpeechConfigBuilder};
}
/// speechsauce::feature (feature.rs): mfcc, mfe, mel_spectrogram1 / 2 (+ the private lmfe / extract_derivative_feature, public here)
pub mod feature {
    pub use super::{extract_derivative_feature, lmfe, mel_spectrogram1, mel_spectrogram2, mfcc, mfe};
    pub use super::{mfcc_batch_device, mfcc_batches_device, try_lmfe, try_mel_spectrogram2, try_mfcc, try_mfcc_batch, try_mfe};
}
/// speechsauce::processing (processing.rs)
pub mod processing {
    pub use super::{cmvn, cmvnw, derivative_extraction, power_spectrum, preemphasis, stack_frames};
    pub use super::{power_spectrum_with, stack_frames_with, try_power_spectrum, try_power_spectrum_of_signal, try_power_spectrum_with,
                    try_stack_frames, try_stack_frames_with};
}
/// speechsauce::functions (functions.rs)
pub mod functions {
    pub use super::{frequency_arr_to_mel, frequency_to_mel, mel_arr_to_frequency, mel_to_frequency, stft1, stft2, triangle, try_stft2,
                    zero_handling};
}
/// speechsauce::util: the `ArrayLog` trait of the hot path (util.rs:372-381).  The pad helpers of util.rs are not part of the path
///  and are not provided.
pub mod util {
    pub use super::ArrayLog;
}
/// Not in the reference: the clip-sharding helpers and the RCCL gather of the final blocks.
pub mod distributed {
    pub use super::{all_gather_features, gather_features, shard_bounds};
}
