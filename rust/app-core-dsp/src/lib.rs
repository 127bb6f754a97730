//! app-core-dsp -- audio-flow's audio hot path on an MI355X, behind the reference's own Rust surface.
//!
//! Every type here keeps the name, signature and error behaviour of its counterpart in
//! forfd8960/audio-flow-rs (`src-tauri/src/modules/audio/{capture,resampler,vad}.rs`,
//! `modules/network/websocket.rs`), cited per item, and forwards to `libaudioflow_dsp.so`
//! (`include/audioflow_dsp.h`).  The build-defined siblings (`FirFilter`, `Stft`, `MelFilterbank`,
//! `FeaturePipeline`, `FeatureStream`, `ChainPipeline`; docs/OPS.md) follow the same style.
//! All `unsafe` lives in this file's thin wrappers over `ffi` (the reference's CLAUDE.md:13 forbids
//! `unsafe` in application code; an FFI shim cannot avoid it, so it is confined here).
//!
//! NOT COMPILED in this repository's image: there is no Rust toolchain.  The
//! C++ twin `include/audioflow.hpp`, which binds the same entry points, is compiled and run on the
//! GPU by `tests/test_gpu_cpp_harness.py`.
pub mod ffi;

use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::ptr;

/// error.rs:94-111.  Library failures surface as `ResamplingFailed(msg)`, exactly like
/// `map_err(|e| AudioError::ResamplingFailed(e.to_string()))` at resampler.rs:49,:85.
#[derive(Debug, PartialEq, Eq, thiserror::Error)]
pub enum AudioError {
    #[error("No input device available")]
    NoDevice,
    #[error("Device configuration failed: {0}")]
    ConfigurationFailed(String),
    #[error("Stream creation failed: {0}")]
    StreamCreationFailed(String),
    #[error("Capture failed: {0}")]
    CaptureFailed(String),
    #[error("Resampling failed: {0}")]
    ResamplingFailed(String),
}

fn check(status: ffi::af_status) -> Result<(), AudioError> {
    if status == ffi::AF_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::af_last_error()) }.to_string_lossy().into_owned();
    Err(AudioError::ResamplingFailed(msg))
}

// ------------------------------------------------------------------------------------------------
/// capture.rs:13-22
#[derive(Debug, Clone)]
pub struct AudioFrame {
    pub samples: Vec<f32>,
    pub sample_rate: u32,
    pub channels: u16,
    pub timestamp_ns: u128,
}

impl AudioFrame {
    /// capture.rs:25-27
    pub fn new(samples: Vec<f32>, sample_rate: u32, channels: u16, timestamp_ns: u128) -> Self {
        Self { samples, sample_rate, channels, timestamp_ns }
    }
    /// capture.rs:30-42 with the device failure reported: `Err(AudioError::ResamplingFailed(msg))` carries the
    /// library's message (no device, HIP error, out of memory).
    pub fn try_to_mono(&self) -> Result<Self, AudioError> {
        if self.channels == 1 {
            return Ok(self.clone());
        }
        let cap = unsafe { ffi::af_to_mono_out_len(self.samples.len(), self.channels) };
        let mut mono = vec![0.0f32; cap];
        let mut n = 0usize;
        check(unsafe {
            ffi::af_to_mono(self.samples.as_ptr(), self.samples.len(), self.channels, mono.as_mut_ptr(), cap, &mut n)
        })?;
        mono.truncate(n);
        Ok(Self { samples: mono, sample_rate: self.sample_rate, channels: 1, timestamp_ns: self.timestamp_ns })
    }
    /// capture.rs:30-42.  The reference's signature is infallible, the device is not: a failure PANICS with the
    /// library's message rather than hand back a frame that still has `channels` channels (a caller that must
    /// not unwind uses `try_to_mono`; the C++ and Python mirrors raise in the same place).
    pub fn to_mono(&self) -> Self {
        match self.try_to_mono() {
            Ok(m) => m,
            Err(e) => panic!("AudioFrame::to_mono: {e:?}"),
        }
    }
}

// ------------------------------------------------------------------------------------------------
/// capture.rs:84-89.  `&self` methods like the reference (interior mutex on the C side).
pub struct RingBuffer { h: *mut ffi::af_ring }
unsafe impl Send for RingBuffer {}
unsafe impl Sync for RingBuffer {}

impl RingBuffer {
    /// capture.rs:93-100
    pub fn new(capacity_samples: usize) -> Self {
        let mut h = ptr::null_mut();
        let _ = unsafe { ffi::af_ring_create(capacity_samples, &mut h) };
        Self { h }
    }
    /// capture.rs:103-122
    pub fn write(&self, data: &[f32]) -> usize {
        unsafe { ffi::af_ring_write(self.h, data.as_ptr(), data.len()) }
    }
    /// capture.rs:125-145
    pub fn read(&self, size: usize) -> Option<Vec<f32>> {
        let cap = size.min(unsafe { ffi::af_ring_capacity(self.h) });
        let mut out = vec![0.0f32; cap];
        let (mut n, mut none) = (0usize, 0 as c_int);
        let st = unsafe { ffi::af_ring_read(self.h, size, out.as_mut_ptr(), cap, &mut n, &mut none) };
        if st != ffi::AF_OK || none != 0 {
            return None;
        }
        out.truncate(n);
        Some(out)
    }
    /// capture.rs:148-152
    pub fn available(&self) -> usize { unsafe { ffi::af_ring_available(self.h) } }
    /// capture.rs:155-160
    pub fn clear(&self) { unsafe { ffi::af_ring_clear(self.h) } }
}
impl Drop for RingBuffer { fn drop(&mut self) { unsafe { ffi::af_ring_destroy(self.h) } } }

// ------------------------------------------------------------------------------------------------
/// resampler.rs:12-21
pub struct AudioResampler { h: *mut ffi::af_resampler }
unsafe impl Send for AudioResampler {}

impl AudioResampler {
    /// resampler.rs:32-57
    pub fn new(input_rate: u32, output_rate: u32) -> Result<Self, AudioError> {
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_resampler_create(input_rate, output_rate, &mut h) })?;
        Ok(Self { h })
    }
    /// resampler.rs:60-62
    pub fn create_48k_to_16k() -> Result<Self, AudioError> { Self::new(48000, 16000) }
    /// resampler.rs:71-93
    pub fn process(&mut self, input: &[f32]) -> Result<Vec<f32>, AudioError> {
        let cap = unsafe { ffi::af_resampler_out_len(self.h, input.len()) };
        let mut out = vec![0.0f32; cap];
        let mut n = 0usize;
        check(unsafe { ffi::af_resampler_process(self.h, input.as_ptr(), input.len(), out.as_mut_ptr(), cap, &mut n) })?;
        out.truncate(n);
        Ok(out)
    }
    /// the polyphase filter's tail (build-defined: rubato exposes no flush)
    pub fn flush(&mut self) -> Result<Vec<f32>, AudioError> {
        let cap = unsafe { ffi::af_resampler_out_len(self.h, 0) } + 64;
        let mut out = vec![0.0f32; cap];
        let mut n = 0usize;
        check(unsafe { ffi::af_resampler_flush(self.h, out.as_mut_ptr(), cap, &mut n) })?;
        out.truncate(n);
        Ok(out)
    }
    /// resampler.rs:96-98
    pub fn input_rate(&self) -> u32 { unsafe { ffi::af_resampler_input_rate(self.h) } }
    /// resampler.rs:101-103
    pub fn output_rate(&self) -> u32 { unsafe { ffi::af_resampler_output_rate(self.h) } }
    /// resampler.rs:106-108
    pub fn needs_resampling(&self) -> bool { unsafe { ffi::af_resampler_needs_resampling(self.h) != 0 } }
}
impl Drop for AudioResampler { fn drop(&mut self) { unsafe { ffi::af_resampler_destroy(self.h) } } }
/// resampler.rs:169-178 (`expect` there; the same panic here if no device is present)
impl Default for AudioResampler {
    fn default() -> Self { Self::create_48k_to_16k().expect("Failed to create default resampler") }
}

/// resampler.rs:114-117
pub struct BatchResampler { h: *mut ffi::af_batch_resampler }
unsafe impl Send for BatchResampler {}

impl BatchResampler {
    /// resampler.rs:121-129
    pub fn new(input_rate: u32, output_rate: u32) -> Result<Self, AudioError> {
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_batch_resampler_create(input_rate, output_rate, &mut h) })?;
        Ok(Self { h })
    }
    /// resampler.rs:132-147: whole 128-frame chunks are resampled, the rest is carried
    pub fn process(&mut self, input: &[f32]) -> Result<Vec<f32>, AudioError> {
        let cap = unsafe { ffi::af_batch_resampler_out_len(self.h, input.len()) };
        let mut out = vec![0.0f32; cap];
        let mut n = 0usize;
        check(unsafe { ffi::af_batch_resampler_process(self.h, input.as_ptr(), input.len(), out.as_mut_ptr(), cap, &mut n) })?;
        out.truncate(n);
        Ok(out)
    }
    /// resampler.rs:150-166: the remainder, zero-padded to one chunk
    pub fn flush(&mut self) -> Result<Vec<f32>, AudioError> {
        let cap = unsafe { ffi::af_batch_resampler_out_len(self.h, 128) };
        let mut out = vec![0.0f32; cap];
        let mut n = 0usize;
        check(unsafe { ffi::af_batch_resampler_flush(self.h, out.as_mut_ptr(), cap, &mut n) })?;
        out.truncate(n);
        Ok(out)
    }
}
impl Drop for BatchResampler { fn drop(&mut self) { unsafe { ffi::af_batch_resampler_destroy(self.h) } } }

// ------------------------------------------------------------------------------------------------
/// vad.rs:8-18
#[derive(Debug, Clone, Copy, PartialEq, Eq, Default)]
pub enum VadLevel { #[default] Aggressive, Balanced, Relaxed }

/// vad.rs:21-32
#[derive(Debug, Clone, Copy)]
pub struct VadConfig {
    pub threshold_db: f32,
    pub smoothing_factor: f32,
    pub silence_timeout_frames: usize,
    pub min_speech_frames: usize,
}
/// vad.rs:34-43
impl Default for VadConfig {
    fn default() -> Self {
        Self { threshold_db: -50.0, smoothing_factor: 0.3, silence_timeout_frames: 15, min_speech_frames: 3 }
    }
}

/// vad.rs:46-54
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum VadState { Silence, Speech, Ending }

fn vad_state(v: c_int) -> VadState {
    match v { 1 => VadState::Speech, 2 => VadState::Ending, _ => VadState::Silence }
}

/// vad.rs:59-70: energy on the device (mean square, vad.rs:157-168), EMA / dBFS / state machine
/// (vad.rs:97-151) on the host inside the library.
pub struct VoiceActivityDetector { h: *mut ffi::af_vad }
unsafe impl Send for VoiceActivityDetector {}

impl VoiceActivityDetector {
    /// vad.rs:80-88
    pub fn new(config: VadConfig) -> Self {
        let c = ffi::af_vad_config {
            threshold_db: config.threshold_db,
            smoothing_factor: config.smoothing_factor,
            silence_timeout_frames: config.silence_timeout_frames,
            min_speech_frames: config.min_speech_frames,
        };
        let mut h = ptr::null_mut();
        let _ = unsafe { ffi::af_vad_create(&c, &mut h) };
        Self { h }
    }
    /// vad.rs:97-154 (a device failure leaves the state unchanged and reports it)
    pub fn detect(&mut self, frame: &[f32]) -> VadState {
        let mut s: c_int = 0;
        let st = unsafe { ffi::af_vad_detect(self.h, frame.as_ptr(), frame.len(), &mut s) };
        if st != ffi::AF_OK { return self.state(); }
        vad_state(s)
    }
    /// a run of equally long frames in one device pass (build-defined batch form of `detect`)
    pub fn detect_frames(&mut self, input: &[f32], frame_len: usize) -> Result<Vec<VadState>, AudioError> {
        let n = if frame_len == 0 { 0 } else { input.len() / frame_len };
        let mut s = vec![0 as c_int; n];
        check(unsafe { ffi::af_vad_detect_frames(self.h, input.as_ptr(), n, frame_len, s.as_mut_ptr()) })?;
        Ok(s.into_iter().map(vad_state).collect())
    }
    /// A batch of independent detectors, one per clip (vad.rs:60-70: each owns its state), scanned on the device:
    /// `clips` = n_clips rows of n_frames * frame_len samples; returns n_clips rows of n_frames states, each row
    /// equal to `VoiceActivityDetector::new(config)` fed that row frame by frame (vad.rs:97-154).
    pub fn detect_batch(config: VadConfig, clips: &[f32], n_clips: usize, frame_len: usize)
        -> Result<Vec<Vec<VadState>>, AudioError> {
        let c = ffi::af_vad_config {
            threshold_db: config.threshold_db,
            smoothing_factor: config.smoothing_factor,
            silence_timeout_frames: config.silence_timeout_frames,
            min_speech_frames: config.min_speech_frames,
        };
        let n_frames = if n_clips > 0 && frame_len > 0 { clips.len() / n_clips / frame_len } else { 0 };
        let mut s = vec![0u8; n_clips * n_frames];
        check(unsafe { ffi::af_vad_detect_batch(&c, clips.as_ptr(), n_clips, n_frames, frame_len, s.as_mut_ptr()) })?;
        Ok(s.chunks(n_frames.max(1)).take(n_clips).map(|r| r.iter().map(|&v| vad_state(v as c_int)).collect()).collect())
    }
    /// vad.rs:179-184
    pub fn reset(&mut self) { unsafe { ffi::af_vad_reset(self.h) } }
    /// vad.rs:187-189
    pub fn state(&self) -> VadState { vad_state(unsafe { ffi::af_vad_state(self.h) }) }
    /// vad.rs:192-194
    pub fn energy_db(&self) -> f32 { unsafe { ffi::af_vad_energy_db(self.h) } }
    /// vad.rs:197-199
    pub fn is_speaking(&self) -> bool { matches!(self.state(), VadState::Speech) }
    /// vad.rs:202-204
    pub fn speech_frame_count(&self) -> usize { unsafe { ffi::af_vad_speech_frame_count(self.h) } }
}
impl Drop for VoiceActivityDetector { fn drop(&mut self) { unsafe { ffi::af_vad_destroy(self.h) } } }
/// vad.rs:72-76
impl Default for VoiceActivityDetector { fn default() -> Self { Self::new(VadConfig::default()) } }

// ------------------------------------------------------------------------------------------------
/// network/websocket.rs:246-251 / :340-345: `(x.clamp(-1, 1) * 32767.0) as i16`, little-endian
pub fn quantize_pcm16(audio_data: &[f32]) -> Result<Vec<i16>, AudioError> {
    let mut out = vec![0i16; audio_data.len()];
    check(unsafe { ffi::af_quantize_pcm16(audio_data.as_ptr(), audio_data.len(), out.as_mut_ptr()) })?;
    Ok(out)
}

/// network/websocket.rs:254: the base64 text of the quantised chunk (`audio_base_64`)
pub fn pcm16_base64(audio_data: &[f32]) -> Result<String, AudioError> {
    let cap = unsafe { ffi::af_base64_len(2 * audio_data.len()) };
    let mut out = vec![0u8; cap];
    let mut n = 0usize;
    check(unsafe { ffi::af_pcm16_base64(audio_data.as_ptr(), audio_data.len(), out.as_mut_ptr() as *mut _, cap, &mut n) })?;
    out.truncate(n);
    String::from_utf8(out).map_err(|e| AudioError::ResamplingFailed(e.to_string()))
}

/// `AudioLevel {rms, peak, dbfs}` of specs/0002-design.md:727-732, one per window
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct AudioLevel { pub rms: f32, pub peak: f32, pub dbfs: f32 }

pub fn level_meter(samples: &[f32], window_len: usize) -> Result<Vec<AudioLevel>, AudioError> {
    let n = if window_len == 0 { 0 } else { samples.len() / window_len };
    let (mut r, mut p, mut d) = (vec![0.0f32; n], vec![0.0f32; n], vec![0.0f32; n]);
    check(unsafe { ffi::af_level_meter(samples.as_ptr(), n, window_len, r.as_mut_ptr(), p.as_mut_ptr(), d.as_mut_ptr()) })?;
    Ok((0..n).map(|i| AudioLevel { rms: r[i], peak: p[i], dbfs: d[i] }).collect())
}

// ------------------------------------------------------------------------------------------------
/// Causal FIR with carried history (docs/OPS.md "FIR"; the reference only has the
/// `noise_suppression` / `auto_gain` flags, config/manager.rs:27-32)
pub struct FirFilter { h: *mut ffi::af_fir }
unsafe impl Send for FirFilter {}

impl FirFilter {
    pub fn new(taps: &[f32]) -> Result<Self, AudioError> {
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_fir_create(taps.as_ptr(), taps.len(), &mut h) })?;
        Ok(Self { h })
    }
    pub fn process(&mut self, input: &[f32]) -> Result<Vec<f32>, AudioError> {
        let mut out = vec![0.0f32; input.len()];
        check(unsafe { ffi::af_fir_process(self.h, input.as_ptr(), input.len(), out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn reset(&mut self) { unsafe { ffi::af_fir_reset(self.h) } }
}
impl Drop for FirFilter { fn drop(&mut self) { unsafe { ffi::af_fir_destroy(self.h) } } }

/// Power spectrogram rows `[frame][n_fft / 2 + 1]` (docs/OPS.md "STFT"; BASELINE config 1)
pub struct Stft { h: *mut ffi::af_stft }
unsafe impl Send for Stft {}

impl Stft {
    pub fn new(n_fft: u32, hop: u32) -> Result<Self, AudioError> {
        let mut cfg = ffi::af_stft_config { channels: 1, n_fft, hop, sample_format: ffi::AF_PCM_F32, device: -1 };
        unsafe { ffi::af_stft_default_config(&mut cfg) };
        cfg.n_fft = n_fft;
        cfg.hop = hop;
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_stft_create(&cfg, &mut h) })?;
        Ok(Self { h })
    }
    pub fn n_frames(&self, n_samples: usize) -> usize { unsafe { ffi::af_stft_n_frames(self.h, n_samples) } }
    pub fn n_bins(&self) -> usize { unsafe { ffi::af_stft_n_bins(self.h) } }
    pub fn power(&mut self, samples: &[f32]) -> Result<Vec<f32>, AudioError> {
        let mut total = 0usize;
        check(unsafe { ffi::af_stft_set_batch_uniform(self.h, 1, samples.len(), &mut total) })?;
        let mut out = vec![0.0f32; total * self.n_bins()];
        check(unsafe { ffi::af_stft_power(self.h, samples.as_ptr() as *const _, samples.len(), out.as_mut_ptr(), out.len()) })?;
        Ok(out)
    }
}
impl Drop for Stft { fn drop(&mut self) { unsafe { ffi::af_stft_destroy(self.h) } } }

/// Triangular mel filterbank and its stand-alone contraction (docs/OPS.md "Mel filterbank")
pub struct MelFilterbank { h: *mut ffi::af_mel, n_mels: usize, n_bins: usize }
unsafe impl Send for MelFilterbank {}

impl MelFilterbank {
    pub fn new(sample_rate: u32, n_fft: u32, n_mels: u32) -> Result<Self, AudioError> {
        let mut cfg = ffi::af_mel_config { sample_rate, n_fft, n_mels, f_min: 0.0, f_max: 0.0, mel_scale: 0, mel_norm: 0, log_mel: 0, device: -1 };
        unsafe { ffi::af_mel_default_config(&mut cfg) };
        cfg.sample_rate = sample_rate;
        cfg.n_fft = n_fft;
        cfg.n_mels = n_mels;
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_mel_create(&cfg, &mut h) })?;
        Ok(Self { h, n_mels: n_mels as usize, n_bins: n_fft as usize / 2 + 1 })
    }
    pub fn weights(&self) -> Result<Vec<f32>, AudioError> {
        let mut w = vec![0.0f32; self.n_mels * self.n_bins];
        check(unsafe { ffi::af_mel_weights(self.h, w.as_mut_ptr(), w.len()) })?;
        Ok(w)
    }
    pub fn apply(&mut self, power: &[f32]) -> Result<Vec<f32>, AudioError> {
        let n_frames = power.len() / self.n_bins;
        let mut mel = vec![0.0f32; n_frames * self.n_mels];
        check(unsafe { ffi::af_mel_apply(self.h, power.as_ptr(), n_frames, mel.as_mut_ptr(), mel.len()) })?;
        Ok(mel)
    }
}
impl Drop for MelFilterbank { fn drop(&mut self) { unsafe { ffi::af_mel_destroy(self.h) } } }

/// One clip's mel spectrogram, rows `[frame][n_mels]`
#[derive(Debug, Clone, PartialEq)]
pub struct MelSpectrogram { pub data: Vec<f32>, pub n_frames: usize, pub n_mels: usize, pub sample_rate: u32, pub hop: u32 }

/// to_mono -> Hann STFT -> |X|^2 -> mel, fused in one kernel (docs/OPS.md)
pub struct FeaturePipeline { h: *mut ffi::af_features, cfg: ffi::af_features_config }
unsafe impl Send for FeaturePipeline {}

impl FeaturePipeline {
    pub fn new(sample_rate: u32, channels: u32, hop: u32, n_mels: u32) -> Result<Self, AudioError> {
        let mut cfg = unsafe { std::mem::zeroed::<ffi::af_features_config>() };
        unsafe { ffi::af_features_default_config(&mut cfg) };
        cfg.sample_rate = sample_rate;
        cfg.channels = channels;
        cfg.hop = hop;
        cfg.n_mels = n_mels;
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_features_create(&cfg, &mut h) })?;
        Ok(Self { h, cfg })
    }
    pub fn n_frames(&self, n_pcm_frames: usize) -> usize { unsafe { ffi::af_features_n_frames(self.h, n_pcm_frames) } }
    /// `FeaturePipeline::process_batch(&[AudioFrame]) -> Result<Vec<MelSpectrogram>, AudioError>`
    pub fn process_batch(&mut self, frames: &[AudioFrame]) -> Result<Vec<MelSpectrogram>, AudioError> {
        let ch = self.cfg.channels as usize;
        let mut offs = vec![0i64; frames.len() + 1];
        let mut pcm: Vec<f32> = Vec::new();
        for (i, f) in frames.iter().enumerate() {
            if f.channels as u32 != self.cfg.channels || f.sample_rate != self.cfg.sample_rate {
                return Err(AudioError::ResamplingFailed("frame format does not match the pipeline".into()));
            }
            let nf = f.samples.len() / ch;
            offs[i + 1] = offs[i] + nf as i64;
            pcm.extend_from_slice(&f.samples[..nf * ch]);
        }
        let mut total = 0usize;
        check(unsafe { ffi::af_features_set_batch(self.h, offs.as_ptr(), frames.len(), &mut total) })?;
        let row = unsafe { ffi::af_features_row_len(self.h) };
        let mut mel = vec![0.0f32; total * row];
        check(unsafe { ffi::af_features_run(self.h, pcm.as_ptr() as *const _, pcm.len(), mel.as_mut_ptr(), mel.len()) })?;
        let per_clip = if self.cfg.channel_mode == ffi::AF_CH_SEPARATE { 2 } else { 1 };
        let (mut out, mut r0) = (Vec::with_capacity(frames.len()), 0usize);
        for i in 0..frames.len() {
            let nf = per_clip * self.n_frames((offs[i + 1] - offs[i]) as usize);
            out.push(MelSpectrogram { data: mel[r0 * row..(r0 + nf) * row].to_vec(), n_frames: nf, n_mels: row,
                                      sample_rate: self.cfg.sample_rate, hop: self.cfg.hop });
            r0 += nf;
        }
        Ok(out)
    }
}
impl Drop for FeaturePipeline { fn drop(&mut self) { unsafe { ffi::af_features_destroy(self.h) } } }

/// The fused path over an unbounded stream in fixed, hipGraph-replayed steps (BASELINE config 5):
/// what the capture consumer loop of specs/0002-design.md:1102-1125 would feed from the ring buffer.
pub struct FeatureStream { h: *mut ffi::af_stream, channels: usize }
unsafe impl Send for FeatureStream {}

impl FeatureStream {
    pub fn new(step_frames: usize) -> Result<Self, AudioError> {
        let mut cfg = unsafe { std::mem::zeroed::<ffi::af_stream_config>() };
        unsafe { ffi::af_stream_default_config(&mut cfg) };
        cfg.step_frames = step_frames;
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_stream_create(&cfg, &mut h) })?;
        Ok(Self { h, channels: cfg.features.channels as usize })
    }
    fn rows(&self, n: usize) -> Vec<f32> { vec![0.0f32; n * unsafe { ffi::af_stream_row_len(self.h) }] }
    /// rows (`[n][n_mels]`, flattened) of every step this push completed
    pub fn push(&mut self, pcm: &[f32]) -> Result<Vec<f32>, AudioError> {
        let n_frames = pcm.len() / self.channels;
        let mut out = self.rows(unsafe { ffi::af_stream_max_rows(self.h, n_frames) });
        let mut n = 0usize;
        check(unsafe { ffi::af_stream_push(self.h, pcm.as_ptr() as *const _, n_frames, out.as_mut_ptr(), out.len(), &mut n) })?;
        out.truncate(n * unsafe { ffi::af_stream_row_len(self.h) });
        Ok(out)
    }
    /// end of stream: the rows of the last, partial step
    pub fn flush(&mut self) -> Result<Vec<f32>, AudioError> {
        let mut out = self.rows(unsafe { ffi::af_stream_rows_per_step(self.h) });
        let mut n = 0usize;
        check(unsafe { ffi::af_stream_flush(self.h, out.as_mut_ptr(), out.len(), &mut n) })?;
        out.truncate(n * unsafe { ffi::af_stream_row_len(self.h) });
        Ok(out)
    }
    pub fn reset(&mut self) -> Result<(), AudioError> { check(unsafe { ffi::af_stream_reset(self.h) }) }
}
impl Drop for FeatureStream { fn drop(&mut self) { unsafe { ffi::af_stream_destroy(self.h) } } }

/// PCM sample types the chain takes: f32 in [-1, 1] (capture.rs:13-22) or the wire format's i16
/// (websocket.rs:246-251, scaled by 1/32768).
pub trait PcmSample: Copy { const FORMAT: u32; }
impl PcmSample for f32 { const FORMAT: u32 = ffi::AF_PCM_F32; }
impl PcmSample for i16 { const FORMAT: u32 = ffi::AF_PCM_I16; }

fn chain_config(input_rate: u32, channels: u32, output_rate: u32, fir_taps: &[f32], format: u32) -> ffi::af_chain_config {
    let mut cfg = unsafe { std::mem::zeroed::<ffi::af_chain_config>() };
    unsafe { ffi::af_chain_default_config(&mut cfg) };
    cfg.input_rate = input_rate;
    cfg.channels = channels;
    cfg.output_rate = output_rate;
    cfg.fir_taps = if fir_taps.is_empty() { ptr::null() } else { fir_taps.as_ptr() };
    cfg.n_fir_taps = fir_taps.len();
    cfg.sample_format = format;
    cfg
}

/// BASELINE config 4: to_mono -> polyphase resample -> FIR -> STFT -> mel on a batch of clips, equal-length or
/// ragged (`AudioFrame` is variable-length, capture.rs:13-28)
pub struct ChainPipeline<T: PcmSample = f32> { h: *mut ffi::af_chain, n_mels: usize, _t: std::marker::PhantomData<T> }
unsafe impl<T: PcmSample> Send for ChainPipeline<T> {}

impl<T: PcmSample> ChainPipeline<T> {
    pub fn new(input_rate: u32, channels: u32, output_rate: u32, fir_taps: &[f32]) -> Result<Self, AudioError> {
        let cfg = chain_config(input_rate, channels, output_rate, fir_taps, T::FORMAT);
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_chain_create(&cfg, &mut h) })?;
        Ok(Self { h, n_mels: cfg.n_mels as usize, _t: std::marker::PhantomData })
    }
    pub fn run(&mut self, pcm: &[T], n_clips: usize, frames_per_clip: usize) -> Result<Vec<f32>, AudioError> {
        let mut total = 0usize;
        check(unsafe { ffi::af_chain_set_batch_uniform(self.h, n_clips, frames_per_clip, &mut total) })?;
        self.run_described(pcm, total)
    }
    /// ragged batch: clip i = PCM frames [clip_offsets[i], clip_offsets[i + 1]) of `pcm`
    pub fn run_ragged(&mut self, pcm: &[T], clip_offsets: &[i64]) -> Result<Vec<f32>, AudioError> {
        let mut total = 0usize;
        let n = clip_offsets.len().saturating_sub(1);
        check(unsafe { ffi::af_chain_set_batch(self.h, clip_offsets.as_ptr(), n, &mut total) })?;
        self.run_described(pcm, total)
    }
    fn run_described(&mut self, pcm: &[T], total: usize) -> Result<Vec<f32>, AudioError> {
        let mut mel = vec![0.0f32; total * self.n_mels];
        check(unsafe { ffi::af_chain_run(self.h, pcm.as_ptr() as *const c_void, pcm.len(), mel.as_mut_ptr(), mel.len()) })?;
        Ok(mel)
    }
    /// kernels one pass enqueues for the batch described last (2 with the fused resampler + FIR stage)
    pub fn n_launches(&self) -> usize { unsafe { ffi::af_chain_n_launches(self.h) as usize } }
}
impl<T: PcmSample> Drop for ChainPipeline<T> { fn drop(&mut self) { unsafe { ffi::af_chain_destroy(self.h) } } }

/// The reference's capture -> wire path on a batch of clips in ONE launch: interleaved PCM -> `AudioFrame::to_mono`
/// (capture.rs:30-42) -> `AudioResampler::create_48k_to_16k` / `process` (resampler.rs:60-93) ->
/// `(x.clamp(-1.0, 1.0) * 32767.0) as i16`, little-endian (websocket.rs:246-251) -> `STANDARD.encode` (websocket.rs:254).
/// `frame_samples == 0`: one padded base64 string per clip; `320`: one per 20 ms send batch
/// (specs/0002-design.md:1543-1546; a trailing partial batch is not sent, as `AudioBatcher` keeps it buffered).
pub struct WireEncoder<T: PcmSample = f32> {
    h: *mut ffi::af_wire,
    /// of the last run, per message (messages only): mean-square energy (vad.rs:157-168) and peak (events/mod.rs:73-75)
    pub energy: Vec<f32>,
    pub peak: Vec<f32>,
    _t: std::marker::PhantomData<T>,
}
unsafe impl<T: PcmSample> Send for WireEncoder<T> {}

impl<T: PcmSample> WireEncoder<T> {
    pub fn new(input_rate: u32, output_rate: u32, channels: u32, frame_samples: u32) -> Result<Self, AudioError> {
        let mut cfg = ffi::af_wire_config { input_rate: 0, output_rate: 0, channels: 0, sample_format: 0, frame_samples: 0, device: -1 };
        unsafe { ffi::af_wire_default_config(&mut cfg) };
        cfg.input_rate = input_rate;
        cfg.output_rate = output_rate;
        cfg.channels = channels;
        cfg.sample_format = T::FORMAT;
        cfg.frame_samples = frame_samples;
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_wire_create(&cfg, &mut h) })?;
        Ok(Self { h, energy: Vec::new(), peak: Vec::new(), _t: std::marker::PhantomData })
    }
    /// `AudioResampler::create_48k_to_16k` (resampler.rs:60-62) + the 20 ms batch of specs/0002-design.md:1543-1546
    pub fn create_48k_to_16k_messages(channels: u32) -> Result<Self, AudioError> { Self::new(48000, 16000, channels, 320) }
    pub fn out_samples(&self, n_frames: usize) -> usize { unsafe { ffi::af_wire_out_samples(self.h, n_frames) } }
    pub fn message_len(&self) -> usize { unsafe { ffi::af_wire_message_len(self.h) } }
    pub fn n_messages(&self, n_frames: usize) -> usize { unsafe { ffi::af_wire_n_messages(self.h, n_frames) } }
    /// kernels one pass enqueues (1 with the fused launch)
    pub fn n_launches(&self) -> usize { unsafe { ffi::af_wire_n_launches(self.h) as usize } }
    /// equal-length clips back to back in `pcm`; returns, per clip, its base64 payloads (one per message, or the one string)
    pub fn run(&mut self, pcm: &[T], n_clips: usize, frames_per_clip: usize) -> Result<Vec<Vec<String>>, AudioError> {
        let (mut total, mut samples) = (0usize, 0usize);
        check(unsafe { ffi::af_wire_set_batch_uniform(self.h, n_clips, frames_per_clip, &mut total, &mut samples) })?;
        self.run_described(pcm, n_clips, total)
    }
    /// ragged batch: clip i = PCM frames [clip_offsets[i], clip_offsets[i + 1]) of `pcm`
    pub fn run_ragged(&mut self, pcm: &[T], clip_offsets: &[i64]) -> Result<Vec<Vec<String>>, AudioError> {
        let (mut total, mut samples) = (0usize, 0usize);
        let n = clip_offsets.len().saturating_sub(1);
        check(unsafe { ffi::af_wire_set_batch(self.h, clip_offsets.as_ptr(), n, &mut total, &mut samples) })?;
        self.run_described(pcm, n, total)
    }
    fn run_described(&mut self, pcm: &[T], n_clips: usize, total: usize) -> Result<Vec<Vec<String>>, AudioError> {
        let mut text = vec![0u8; total];
        // per message: `VoiceActivityDetector::calculate_energy` of the message (vad.rs:157-168) and max |x|
        let nm = unsafe { ffi::af_wire_total_messages(self.h) };
        self.energy = vec![0.0f32; nm];
        self.peak = vec![0.0f32; nm];
        let (ep, pp) = if nm > 0 { (self.energy.as_mut_ptr(), self.peak.as_mut_ptr()) } else { (ptr::null_mut(), ptr::null_mut()) };
        check(unsafe { ffi::af_wire_run(self.h, pcm.as_ptr() as *const c_void, pcm.len(), text.as_mut_ptr() as *mut _, text.len(), ptr::null_mut(), 0, ep, pp, nm) })?;
        let mut off = vec![0i64; n_clips + 1];
        check(unsafe { ffi::af_wire_text_offsets(self.h, off.as_mut_ptr(), off.len()) })?;
        let ml = self.message_len();
        let mut out = Vec::with_capacity(n_clips);
        for i in 0..n_clips {
            let row = &text[off[i] as usize..off[i + 1] as usize];
            // base64 is ASCII: from_utf8_lossy never replaces anything
            let cut = if ml > 0 { ml } else { row.len().max(1) };
            out.push(row.chunks(cut).map(|m| String::from_utf8_lossy(m).into_owned()).collect());
        }
        Ok(out)
    }
}
impl<T: PcmSample> Drop for WireEncoder<T> { fn drop(&mut self) { unsafe { ffi::af_wire_destroy(self.h) } } }

/// BASELINE config 5 on the whole chain: the handle owns the resampler history, the FIR's K - 1 samples and the
/// STFT's n_fft - hop carry, and replays one captured hipGraph per step; `push` returns the rows of every step it
/// completed, `flush` the rest, and all rows in order equal `ChainPipeline::run` on the whole stream bit for bit.
/// Reference analogue of the carry / flush: `BatchResampler::{process, flush}`, resampler.rs:132-166.
pub struct ChainStream<T: PcmSample = f32> { h: *mut ffi::af_chain_stream, channels: usize, _t: std::marker::PhantomData<T> }
unsafe impl<T: PcmSample> Send for ChainStream<T> {}

impl<T: PcmSample> ChainStream<T> {
    pub fn new(step_frames: usize, input_rate: u32, channels: u32, output_rate: u32, fir_taps: &[f32]) -> Result<Self, AudioError> {
        let cfg = chain_config(input_rate, channels, output_rate, fir_taps, T::FORMAT);
        let mut h = ptr::null_mut();
        check(unsafe { ffi::af_chain_stream_create(&cfg, step_frames, &mut h) })?;
        Ok(Self { h, channels: channels as usize, _t: std::marker::PhantomData })
    }
    /// input frames per replayed step (the multiple of the step quantum nearest to the request)
    pub fn step_frames(&self) -> usize { unsafe { ffi::af_chain_stream_step_frames(self.h) } }
    pub fn row_len(&self) -> usize { unsafe { ffi::af_chain_stream_row_len(self.h) } }
    pub fn total_rows(&self, n_frames: u64) -> usize { unsafe { ffi::af_chain_stream_total_rows(self.h, n_frames) } }
    pub fn push(&mut self, pcm: &[T]) -> Result<Vec<f32>, AudioError> {
        let n_frames = pcm.len() / self.channels;
        let mut rows = vec![0.0f32; unsafe { ffi::af_chain_stream_max_rows(self.h, n_frames) } * self.row_len()];
        let mut n = 0usize;
        check(unsafe { ffi::af_chain_stream_push(self.h, pcm.as_ptr() as *const c_void, n_frames, rows.as_mut_ptr(), rows.len(), &mut n) })?;
        rows.truncate(n * self.row_len());
        Ok(rows)
    }
    pub fn flush(&mut self) -> Result<Vec<f32>, AudioError> {
        // (the flush can owe one row more than a step: af_chain_stream_flush_rows is the bound)
        let mut rows = vec![0.0f32; unsafe { ffi::af_chain_stream_flush_rows(self.h) }.max(1) * self.row_len()];
        let mut n = 0usize;
        check(unsafe { ffi::af_chain_stream_flush(self.h, rows.as_mut_ptr(), rows.len(), &mut n) })?;
        rows.truncate(n * self.row_len());
        Ok(rows)
    }
    pub fn reset(&mut self) -> Result<(), AudioError> { check(unsafe { ffi::af_chain_stream_reset(self.h) }) }
    /// Frames step 0 consumes / frames of history every later step reads in front of its own.
    pub fn first_step_frames(&self) -> usize { unsafe { ffi::af_chain_stream_first_step_frames(self.h) } }
    pub fn history_frames(&self) -> usize { unsafe { ffi::af_chain_stream_history_frames(self.h) } }
    /// Time sharding: continue as if `step` steps had run; returns the frame the next push must start at.
    pub fn seek(&mut self, step: u64) -> Result<u64, AudioError> {
        let mut first = 0u64;
        check(unsafe { ffi::af_chain_stream_seek(self.h, step, &mut first) })?;
        Ok(first)
    }
}
impl<T: PcmSample> Drop for ChainStream<T> { fn drop(&mut self) { unsafe { ffi::af_chain_stream_destroy(self.h) } } }

// ------------------------------------------------------------------------------------------------
// The reference's own unit tests, to be run on a machine with the library and an MI355X
// (`AUDIOFLOW_DSP_LIB_DIR=... cargo test`).  Their C++ twins run in tests/cpp/harness.cpp.
#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn test_audio_frame_to_mono_stereo() {                       // capture.rs:385-400
        let frame = AudioFrame::new(vec![0.5, 0.25, -0.5, -0.25], 16000, 2, 1000);
        let mono = frame.to_mono();
        assert_eq!(mono.channels, 1);
        assert_eq!(mono.samples.len(), 2);
        assert!((mono.samples[0] - 0.375).abs() < 0.001);
        assert!((mono.samples[1] - (-0.375)).abs() < 0.001);
    }

    #[test]
    fn test_stft_speech_frames() {                               // build-defined (docs/OPS.md "Frame lengths"): Stft::new(400, 160)
        let mut st = Stft::new(400, 160).unwrap();               // 25 ms / 10 ms at 16 kHz (websocket.rs:274 names the format)
        let x: Vec<f32> = (0..16000).map(|i| 0.5 * (2.0 * std::f32::consts::PI * 1000.0 * i as f32 / 16000.0).sin()).collect();
        assert_eq!(st.n_frames(x.len()), 1 + (16000 - 400) / 160);
        assert_eq!(st.n_bins(), 201);
        let p = st.power(&x).unwrap();
        let arg = (1..201).max_by(|&a, &b| p[a].partial_cmp(&p[b]).unwrap()).unwrap();
        assert_eq!(arg, 25);                                     // 1 kHz at 16 kHz / 400
        let law = (0.5f32 * 400.0 / 4.0).powi(2);                // a bin-centred tone of amplitude A: |X|^2 = (A N / 4)^2
        assert!((p[25] / law - 1.0).abs() < 1e-3);
    }

    #[test]
    fn test_no_resample_needed() {                               // resampler.rs:185-190
        let mut resampler = AudioResampler::new(16000, 16000).unwrap();
        let input = vec![0.1, 0.2, 0.3, 0.4];
        assert_eq!(resampler.process(&input).unwrap(), input);
    }

    #[test]
    fn test_resample_rates() {                                   // resampler.rs:193-198
        let resampler = AudioResampler::new(48000, 16000).unwrap();
        assert_eq!(resampler.input_rate(), 48000);
        assert_eq!(resampler.output_rate(), 16000);
        assert!(resampler.needs_resampling());
    }

    #[test]
    fn test_ring_buffer_wrap_around() {                          // capture.rs:459-471
        let buffer = RingBuffer::new(10);
        buffer.write(&[1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]);
        buffer.read(5);
        buffer.write(&[8.0, 9.0, 10.0]);
        assert_eq!(buffer.read(5).unwrap(), vec![6.0, 7.0, 8.0, 9.0, 10.0]);
    }

    #[test]
    fn test_ring_buffer_overflow() {                             // capture.rs:473-485
        let buffer = RingBuffer::new(10);
        assert_eq!(buffer.write(&vec![1.0; 15]), 9);
        assert_eq!(buffer.available(), 9);
    }

    #[test]
    fn test_vad_speech_detection() {                             // vad.rs:225-236
        let mut vad = VoiceActivityDetector::new(VadConfig { threshold_db: -50.0, smoothing_factor: 0.3, ..Default::default() });
        assert_eq!(vad.detect(&vec![0.5; 480]), VadState::Speech);
    }
}
