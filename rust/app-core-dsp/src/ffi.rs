//! Raw bindings of include/audioflow_dsp.h -- the only module of the crate that says `unsafe extern`.
//! Kept in the header's order; every struct is `#[repr(C)]` with the header's field order.
//! NOT COMPILED in this repository's image (no Rust toolchain); tests/test_abi.py checks the same
//! declarations against the built library through ctypes.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub type af_status = i32;
pub const AF_OK: af_status = 0;
pub const AF_ERR_INVALID_ARG: af_status = 1;
pub const AF_ERR_BUFFER_TOO_SMALL: af_status = 2;
pub const AF_ERR_NO_DEVICE: af_status = 3;
pub const AF_ERR_HIP: af_status = 4;
pub const AF_ERR_OUT_OF_MEMORY: af_status = 5;
pub const AF_ERR_UNSUPPORTED: af_status = 6;
pub const AF_PCM_F32: u32 = 0;
pub const AF_PCM_I16: u32 = 1;
pub const AF_OUT_MEL: u32 = 0;
pub const AF_OUT_POWER: u32 = 1;
pub const AF_CH_DOWNMIX: u32 = 0;
pub const AF_CH_SEPARATE: u32 = 1;

macro_rules! opaque { ($($n:ident),*) => { $( #[repr(C)] pub struct $n { _p: [u8; 0] } )* } }
opaque!(af_vad, af_resampler, af_batch_resampler, af_fir, af_features, af_stft, af_mel, af_chain, af_ring, af_stream, af_chain_stream);

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct af_vad_config {
    pub threshold_db: f32, pub smoothing_factor: f32,
    pub silence_timeout_frames: usize, pub min_speech_frames: usize,
}
/// one detector's carried state for `af_vad_detect_batch_device` (vad.rs:63-69)
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct af_vad_lane { pub smoothed_energy: f32, pub silence_frames: u32, pub speech_frames: u32, pub state: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct af_features_config {
    pub sample_rate: u32, pub channels: u32, pub n_fft: u32, pub hop: u32, pub n_mels: u32,
    pub f_min: f32, pub f_max: f32, pub sample_format: u32, pub device: i32,
    pub output: u32, pub log_mel: u32, pub mel_scale: u32, pub mel_norm: u32, pub channel_mode: u32,
}
#[repr(C)] pub struct af_batch_desc { pub cfg: af_features_config, pub n_clips: usize, pub clip_offsets: *const i64 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct af_stft_config { pub channels: u32, pub n_fft: u32, pub hop: u32, pub sample_format: u32, pub device: i32 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct af_mel_config {
    pub sample_rate: u32, pub n_fft: u32, pub n_mels: u32, pub f_min: f32, pub f_max: f32,
    pub mel_scale: u32, pub mel_norm: u32, pub log_mel: u32, pub device: i32,
}
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct af_chain_config {
    pub input_rate: u32, pub channels: u32, pub output_rate: u32,
    pub fir_taps: *const f32, pub n_fir_taps: usize,
    pub hop: u32, pub n_mels: u32, pub f_min: f32, pub f_max: f32, pub device: i32,
    pub sample_format: u32,
}
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct af_wire_config {
    pub input_rate: u32, pub output_rate: u32, pub channels: u32, pub sample_format: u32, pub frame_samples: u32, pub device: i32,
}
#[repr(C)] pub struct af_wire { _private: [u8; 0] }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct af_stream_config { pub features: af_features_config, pub step_frames: usize }

extern "C" {
    pub fn af_last_error() -> *const c_char;
    pub fn af_version() -> *const c_char;
    pub fn af_device_count() -> c_int;
    // capture.rs:30-42
    pub fn af_to_mono_out_len(n_samples: usize, channels: u16) -> usize;
    pub fn af_to_mono(input: *const f32, n_samples: usize, channels: u16, out: *mut f32, out_cap: usize, n_out: *mut usize) -> af_status;
    // network/websocket.rs:244-254
    pub fn af_quantize_pcm16(input: *const f32, n: usize, out: *mut i16) -> af_status;
    pub fn af_base64_len(n_bytes: usize) -> usize;
    pub fn af_pcm16_base64(input: *const f32, n: usize, out: *mut c_char, out_cap: usize, n_out: *mut usize) -> af_status;
    // vad.rs:157-168 (batched) and the level meter of specs/0002-design.md:727-732
    pub fn af_vad_energy(input: *const f32, n_frames: usize, frame_len: usize, energy_out: *mut f32) -> af_status;
    pub fn af_level_meter(input: *const f32, n_windows: usize, window_len: usize, rms: *mut f32, peak: *mut f32, dbfs: *mut f32) -> af_status;
    // vad.rs:78-205
    pub fn af_vad_default_config(cfg: *mut af_vad_config);
    pub fn af_vad_create(cfg: *const af_vad_config, out: *mut *mut af_vad) -> af_status;
    pub fn af_vad_destroy(v: *mut af_vad);
    pub fn af_vad_detect(v: *mut af_vad, frame: *const f32, n: usize, state_out: *mut c_int) -> af_status;
    pub fn af_vad_detect_frames(v: *mut af_vad, input: *const f32, n_frames: usize, frame_len: usize, states_out: *mut c_int) -> af_status;
    pub fn af_vad_reset(v: *mut af_vad);
    // batched detectors, one lane per clip (vad.rs:60-70, :97-154)
    pub fn af_vad_detect_batch_device(cfg: *const af_vad_config, energy_dev: *const f32, n_clips: usize, n_frames: usize,
                                      row_stride: usize, lanes_dev: *mut af_vad_lane, states_dev: *mut u8,
                                      state_stride: usize, stream: *mut c_void) -> af_status;
    pub fn af_vad_detect_batch(cfg: *const af_vad_config, input: *const f32, n_clips: usize, n_frames: usize,
                               frame_len: usize, states_out: *mut u8) -> af_status;
    pub fn af_vad_state(v: *const af_vad) -> c_int;
    pub fn af_vad_energy_db(v: *const af_vad) -> f32;
    pub fn af_vad_speech_frame_count(v: *const af_vad) -> usize;
    // resampler.rs:23-109
    pub fn af_resampler_create(input_rate: u32, output_rate: u32, out: *mut *mut af_resampler) -> af_status;
    pub fn af_resampler_destroy(r: *mut af_resampler);
    pub fn af_resampler_input_rate(r: *const af_resampler) -> u32;
    pub fn af_resampler_output_rate(r: *const af_resampler) -> u32;
    pub fn af_resampler_needs_resampling(r: *const af_resampler) -> c_int;
    pub fn af_resampler_out_len(r: *const af_resampler, n_in: usize) -> usize;
    pub fn af_resampler_process(r: *mut af_resampler, input: *const f32, n_in: usize, out: *mut f32, out_cap: usize, n_out: *mut usize) -> af_status;
    pub fn af_resampler_flush(r: *mut af_resampler, out: *mut f32, out_cap: usize, n_out: *mut usize) -> af_status;
    pub fn af_resampler_reset(r: *mut af_resampler);
    // resampler.rs:114-167
    pub fn af_batch_resampler_create(input_rate: u32, output_rate: u32, out: *mut *mut af_batch_resampler) -> af_status;
    pub fn af_batch_resampler_destroy(r: *mut af_batch_resampler);
    pub fn af_batch_resampler_out_len(r: *const af_batch_resampler, n_in: usize) -> usize;
    pub fn af_batch_resampler_process(r: *mut af_batch_resampler, input: *const f32, n_in: usize, out: *mut f32, out_cap: usize, n_out: *mut usize) -> af_status;
    pub fn af_batch_resampler_flush(r: *mut af_batch_resampler, out: *mut f32, out_cap: usize, n_out: *mut usize) -> af_status;
    // FirFilter (docs/OPS.md)
    pub fn af_fir_create(taps: *const f32, n_taps: usize, out: *mut *mut af_fir) -> af_status;
    pub fn af_fir_destroy(f: *mut af_fir);
    pub fn af_fir_process(f: *mut af_fir, input: *const f32, n: usize, out: *mut f32) -> af_status;
    pub fn af_fir_reset(f: *mut af_fir);
    // FeaturePipeline (docs/OPS.md)
    pub fn af_features_default_config(cfg: *mut af_features_config);
    pub fn af_features_create(cfg: *const af_features_config, out: *mut *mut af_features) -> af_status;
    pub fn af_features_destroy(p: *mut af_features);
    pub fn af_features_n_frames(p: *const af_features, n_pcm_frames: usize) -> usize;
    pub fn af_features_row_len(p: *const af_features) -> usize;
    pub fn af_features_set_batch(p: *mut af_features, clip_offsets: *const i64, n_clips: usize, total_frames_out: *mut usize) -> af_status;
    pub fn af_features_run(p: *mut af_features, pcm: *const c_void, pcm_len: usize, mel: *mut f32, mel_cap: usize) -> af_status;
    pub fn af_features_run_device(p: *mut af_features, pcm_dev: *const c_void, mel_dev: *mut f32, stream: *mut c_void) -> af_status;
    pub fn af_features_mel_weights(p: *const af_features, w: *mut f32, cap: usize) -> af_status;
    /// diagnostics: kernel family (1 general, 2 hop-256 register FFT, 4 matrix-product DFT) and the shape of its plans, info[12]
    pub fn af_features_kernel_describe(p: *const af_features, info: *mut i32) -> c_int;
    pub fn af_pipeline_run(batch: *const af_batch_desc, pcm: *const c_void, pcm_len: usize, out: *mut f32, out_cap: usize, total_rows_out: *mut usize) -> af_status;
    // Stft / MelFilterbank (docs/OPS.md)
    pub fn af_stft_default_config(cfg: *mut af_stft_config);
    pub fn af_stft_create(cfg: *const af_stft_config, out: *mut *mut af_stft) -> af_status;
    pub fn af_stft_destroy(s: *mut af_stft);
    pub fn af_stft_n_frames(s: *const af_stft, n_pcm_frames: usize) -> usize;
    pub fn af_stft_n_bins(s: *const af_stft) -> usize;
    pub fn af_stft_set_batch_uniform(s: *mut af_stft, n_clips: usize, frames_per_clip: usize, total_frames_out: *mut usize) -> af_status;
    pub fn af_stft_power(s: *mut af_stft, pcm: *const c_void, pcm_len: usize, power: *mut f32, power_cap: usize) -> af_status;
    pub fn af_mel_default_config(cfg: *mut af_mel_config);
    pub fn af_mel_create(cfg: *const af_mel_config, out: *mut *mut af_mel) -> af_status;
    pub fn af_mel_destroy(m: *mut af_mel);
    pub fn af_mel_weights(m: *const af_mel, w: *mut f32, cap: usize) -> af_status;
    pub fn af_mel_apply(m: *mut af_mel, power: *const f32, n_frames: usize, mel: *mut f32, mel_cap: usize) -> af_status;
    // chain (BASELINE config 4)
    pub fn af_chain_default_config(cfg: *mut af_chain_config);
    pub fn af_chain_create(cfg: *const af_chain_config, out: *mut *mut af_chain) -> af_status;
    pub fn af_chain_destroy(c: *mut af_chain);
    pub fn af_chain_n_frames(c: *const af_chain, n_pcm_frames: usize) -> usize;
    pub fn af_chain_set_batch_uniform(c: *mut af_chain, n_clips: usize, frames_per_clip: usize, total_frames_out: *mut usize) -> af_status;
    pub fn af_chain_set_batch(c: *mut af_chain, clip_offsets: *const i64, n_clips: usize, total_frames_out: *mut usize) -> af_status;
    pub fn af_chain_run(c: *mut af_chain, pcm: *const c_void, pcm_len: usize, mel: *mut f32, mel_cap: usize) -> af_status;
    pub fn af_chain_run_device(c: *mut af_chain, pcm_dev: *const c_void, mel_dev: *mut f32, stream: *mut c_void) -> af_status;
    pub fn af_chain_set_option(c: *mut af_chain, key: *const c_char, value: c_int) -> af_status;
    pub fn af_features_set_option(p: *mut af_features, key: *const c_char, value: c_int) -> af_status;
    // config 5 on the whole chain: resampler / FIR / STFT history owned by the handle, one hipGraph per step
    pub fn af_chain_stream_create(cfg: *const af_chain_config, step_frames: usize, out: *mut *mut af_chain_stream) -> af_status;
    pub fn af_chain_stream_destroy(s: *mut af_chain_stream);
    pub fn af_chain_stream_step_frames(s: *const af_chain_stream) -> usize;
    pub fn af_chain_stream_rows_per_step(s: *const af_chain_stream) -> usize;
    pub fn af_chain_stream_row_len(s: *const af_chain_stream) -> usize;
    pub fn af_chain_stream_total_rows(s: *const af_chain_stream, n_frames: u64) -> usize;
    pub fn af_chain_stream_flush_rows(s: *const af_chain_stream) -> usize;
    pub fn af_chain_stream_max_rows(s: *const af_chain_stream, n_frames: usize) -> usize;
    pub fn af_chain_stream_push(s: *mut af_chain_stream, pcm: *const c_void, n_frames: usize, rows_out: *mut f32, rows_cap: usize, n_rows: *mut usize) -> af_status;
    pub fn af_chain_stream_flush(s: *mut af_chain_stream, rows_out: *mut f32, rows_cap: usize, n_rows: *mut usize) -> af_status;
    pub fn af_chain_stream_reset(s: *mut af_chain_stream) -> af_status;
    pub fn af_chain_stream_stats(s: *const af_chain_stream, stats: *mut f64) -> af_status;
    pub fn af_chain_stream_first_step_frames(s: *const af_chain_stream) -> usize;
    pub fn af_chain_stream_history_frames(s: *const af_chain_stream) -> usize;
    pub fn af_chain_stream_seek(s: *mut af_chain_stream, step: u64, first_frame: *mut u64) -> af_status;
    // the capture -> wire path in one launch (capture.rs:30-42, resampler.rs:60-93, websocket.rs:244-254)
    pub fn af_wire_default_config(cfg: *mut af_wire_config);
    pub fn af_wire_create(cfg: *const af_wire_config, out: *mut *mut af_wire) -> af_status;
    pub fn af_wire_destroy(w: *mut af_wire);
    pub fn af_wire_out_samples(w: *const af_wire, n_frames: usize) -> usize;
    pub fn af_wire_text_len(w: *const af_wire, n_frames: usize) -> usize;
    pub fn af_wire_message_len(w: *const af_wire) -> usize;
    pub fn af_wire_n_messages(w: *const af_wire, n_frames: usize) -> usize;
    pub fn af_wire_set_batch_uniform(w: *mut af_wire, n_clips: usize, frames_per_clip: usize, total_text: *mut usize, total_samples: *mut usize) -> af_status;
    pub fn af_wire_set_batch(w: *mut af_wire, clip_offsets: *const i64, n_clips: usize, total_text: *mut usize, total_samples: *mut usize) -> af_status;
    pub fn af_wire_text_offsets(w: *const af_wire, text_off: *mut i64, cap: usize) -> af_status;
    pub fn af_wire_total_messages(w: *const af_wire) -> usize;
    pub fn af_wire_run_device(w: *mut af_wire, pcm_dev: *const c_void, text_dev: *mut c_char, pcm16_dev: *mut i16, energy_dev: *mut f32, peak_dev: *mut f32, stream: *mut c_void) -> af_status;
    pub fn af_wire_run(w: *mut af_wire, pcm: *const c_void, pcm_len: usize, text: *mut c_char, text_cap: usize, pcm16: *mut i16, pcm16_cap: usize, energy: *mut f32, peak: *mut f32, level_cap: usize) -> af_status;
    pub fn af_wire_set_option(w: *mut af_wire, key: *const c_char, value: c_int) -> af_status;
    pub fn af_wire_n_launches(w: *const af_wire) -> c_int;
    pub fn af_resample_release_staging();
    // optional cubic mode (what resampler.rs:43-49 configures; docs/OPS.md "Cubic interpolation", parity unpinned)
    pub fn af_resample_cubic_batch(input_rate: u32, output_rate: u32, input: *const f32, n_clips: usize, n_in: usize,
                                   out: *mut f32, out_cap: usize) -> af_status;
    pub fn af_chain_n_launches(c: *const af_chain) -> c_int;
    // RingBuffer, capture.rs:84-161
    pub fn af_ring_create(capacity_samples: usize, out: *mut *mut af_ring) -> af_status;
    pub fn af_ring_destroy(r: *mut af_ring);
    pub fn af_ring_capacity(r: *const af_ring) -> usize;
    pub fn af_ring_write(r: *mut af_ring, data: *const f32, n: usize) -> usize;
    pub fn af_ring_read(r: *mut af_ring, size: usize, out: *mut f32, out_cap: usize, n_read: *mut usize, is_none: *mut c_int) -> af_status;
    pub fn af_ring_available(r: *mut af_ring) -> usize;
    pub fn af_ring_clear(r: *mut af_ring);
    // FeatureStream (BASELINE config 5)
    pub fn af_stream_default_config(cfg: *mut af_stream_config);
    pub fn af_stream_create(cfg: *const af_stream_config, out: *mut *mut af_stream) -> af_status;
    pub fn af_stream_destroy(s: *mut af_stream);
    pub fn af_stream_rows_per_step(s: *const af_stream) -> usize;
    pub fn af_stream_row_len(s: *const af_stream) -> usize;
    pub fn af_stream_total_rows(s: *const af_stream, n_frames: u64) -> usize;
    pub fn af_stream_max_rows(s: *const af_stream, n_frames: usize) -> usize;
    pub fn af_stream_push(s: *mut af_stream, pcm: *const c_void, n_frames: usize, rows_out: *mut f32, rows_cap: usize, n_rows: *mut usize) -> af_status;
    pub fn af_stream_flush(s: *mut af_stream, rows_out: *mut f32, rows_cap: usize, n_rows: *mut usize) -> af_status;
    pub fn af_stream_reset(s: *mut af_stream) -> af_status;
    pub fn af_stream_stats(s: *const af_stream, stats: *mut f64) -> af_status;
}
