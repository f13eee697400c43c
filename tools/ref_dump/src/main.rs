//! Writes raw little-endian f32 files (+ a manifest) with inputs and outputs of the reference crate for the
//! BASELINE configurations.  Every array is row-major.  `pack.py` turns the directory into
//! tests/golden/reference_v1.npz, which tests/test_reference_vectors.py picks up.
//!
//! What each vector pins:
//!   dct2_*          ndrustfft::nddct2 on known rows: the DCT-II gain `g` (feature.rs:120-123)
//!   pspec_*         processing::power_spectrum on two hand-made frames: rustfft's forward convention + the 1/N scale
//!   mfcc1_* / mfe1_* feature::mfcc / mfe on clips that give exactly ONE frame (`full`: all 40 cepstra, dc_elimination = false --
//!                   pins the `[0, 0]`-only scaling of column 0); (stack_frames copies the signal correctly
//!                   only then, processing.rs:110-120): the whole chain filterbank -> ln -> DCT -> scaling on real data
//!   mfcc_* / mfe_*  the same on 1 s clips: the literal behaviour (all frames zero for more than two frames)
//!   stft_* / mel_*  functions::stft2 and feature::mel_spectrogram2 on a two-channel clip with a FRESH SpeechConfig
//!                   (realfft's convention, Vorbis window, wnorm, carry-over between channels)
use ndarray::{Array1, Array2, Axis};
use ndrustfft::{nddct2, DctHandler};
use speechsauce::config::SpeechConfig;
use speechsauce::{feature, functions, processing};
use std::fs::File;
use std::io::Write;

/// The same generator as tools/ref_dump/pack.py `lcg_signal`: x[i] = ((state >> 8) / 2^24 - 0.5) * 2 * amp with
/// state <- state * 1664525 + 1013904223 (mod 2^32), starting from `seed`.
fn lcg_signal(seed: u32, n: usize, amp: f32) -> Array1<f32> {
    let mut s = seed;
    Array1::from_shape_fn(n, |_| {
        s = s.wrapping_mul(1664525).wrapping_add(1013904223);
        (((s >> 8) as f32) / 16777216.0 - 0.5) * 2.0 * amp
    })
}

fn dump(dir: &str, name: &str, shape: &[usize], data: &[f32], manifest: &mut Vec<String>) {
    let mut f = File::create(format!("{}/{}.f32", dir, name)).expect("create");
    for v in data {
        f.write_all(&v.to_le_bytes()).expect("write");
    }
    let dims: Vec<String> = shape.iter().map(|d| d.to_string()).collect();
    manifest.push(format!("{} {}", name, dims.join(",")));
}

fn cfg(which: &str) -> SpeechConfig {
    match which {
        // SpeechConfig::new(sample_rate, fft_points, frame_length, frame_stride, num_cepstral, num_filters, low, high, dc_elimination)
        "cfg1" => SpeechConfig::new(16000, 512, 0.02, 0.01, 13, 40, 0.0, 8000.0, true),
        "cfg3" => SpeechConfig::new(16000, 2048, 0.032, 0.032, 13, 128, 0.0, 8000.0, true),
        "cfg5" => SpeechConfig::new(44100, 4096, 4096.0 / 44100.0, 1024.0 / 44100.0, 40, 256, 0.0, 22050.0, true),
        // num_cepstral == num_filters and no dc elimination: column 0 keeps its DCT value, so the `[0, 0]`-only 1/sqrt(4n)
        // scaling of feature.rs:126-131 shows in the output
        "full" => SpeechConfig::new(16000, 512, 0.02, 0.01, 40, 40, 0.0, 8000.0, false),
        _ => panic!("unknown config"),
    }
}

fn main() {
    let dir = std::env::args().nth(1).unwrap_or_else(|| "ref_dump_out".to_string());
    std::fs::create_dir_all(&dir).expect("mkdir");
    let mut man: Vec<String> = Vec::new();

    // ---- DCT-II gain ----
    for (name, n) in [("dct2_ramp8", 8usize), ("dct2_ramp40", 40usize)] {
        let mut x = Array2::<f32>::zeros((1, n));
        for i in 0..n {
            x[[0, i]] = (i + 1) as f32;
        }
        let mut y = Array2::<f32>::zeros((1, n));
        let mut h: DctHandler<f32> = DctHandler::new(n);
        nddct2(&x, &mut y, &mut h, 1);
        dump(&dir, &format!("{}_in", name), &[1, n], x.as_slice().unwrap(), &mut man);
        dump(&dir, &format!("{}_out", name), &[1, n], y.as_slice().unwrap(), &mut man);
    }

    // ---- power_spectrum on hand-made frames (processing.rs:179-181) ----
    {
        let mut frames = Array2::<f32>::zeros((2, 320));
        let a = lcg_signal(11, 320, 0.1);
        for i in 0..320 {
            frames[[0, i]] = a[i];
        }
        frames[[1, 1]] = 1.0; // an impulse at n = 1
        dump(&dir, "pspec_in", &[2, 320], frames.as_slice().unwrap(), &mut man);
        let p = processing::power_spectrum(frames, 512);
        dump(&dir, "pspec_out", &[p.len_of(Axis(0)), p.len_of(Axis(1))], p.as_standard_layout().as_slice().unwrap(), &mut man);
    }

    // ---- single-frame clips: the whole MFCC chain on real data ----
    for (which, n) in [("cfg1", 500usize), ("cfg5", 5200usize), ("full", 500usize)] {
        let c = cfg(which);
        let x = lcg_signal(21, n, 0.1);
        dump(&dir, &format!("mfcc1_{}_in", which), &[n], x.as_slice().unwrap(), &mut man);
        let m = feature::mfcc(x.view(), &c);
        dump(&dir, &format!("mfcc1_{}_out", which), &[m.len_of(Axis(0)), m.len_of(Axis(1))], m.as_standard_layout().as_slice().unwrap(), &mut man);
        let (f, e) = feature::mfe(x.view(), &c);
        dump(&dir, &format!("mfe1_{}_feat", which), &[f.len_of(Axis(0)), f.len_of(Axis(1))], f.as_standard_layout().as_slice().unwrap(), &mut man);
        dump(&dir, &format!("mfe1_{}_energy", which), &[e.len()], e.as_standard_layout().as_slice().unwrap(), &mut man);
    }

    // ---- 1 s clips: literal behaviour of the frame path ----
    for (which, n) in [("cfg1", 16000usize), ("cfg5", 44100usize)] {
        let c = cfg(which);
        let x = lcg_signal(31, n, 0.1);
        dump(&dir, &format!("mfcc_{}_in", which), &[n], x.as_slice().unwrap(), &mut man);
        let m = feature::mfcc(x.view(), &c);
        dump(&dir, &format!("mfcc_{}_out", which), &[m.len_of(Axis(0)), m.len_of(Axis(1))], m.as_standard_layout().as_slice().unwrap(), &mut man);
    }

    // ---- STFT path, two channels, fresh configs (the config carries state between calls: functions.rs:89-121) ----
    {
        let n = 16000usize;
        let mut x = Array2::<f32>::zeros((2, n));
        let a = lcg_signal(41, n, 0.1);
        let b = lcg_signal(42, n, 0.1);
        for i in 0..n {
            x[[0, i]] = a[i];
            x[[1, i]] = b[i];
        }
        dump(&dir, "stft_cfg3_in", &[2, n], x.as_slice().unwrap(), &mut man);
        let c = cfg("cfg3");
        let s = functions::stft2(x.view(), &c);
        let sh = s.shape().to_vec();
        let flat: Vec<f32> = s.as_standard_layout().iter().flat_map(|z| vec![z.re, z.im]).collect();
        dump(&dir, "stft_cfg3_out", &[sh[0], sh[1], sh[2], 2], &flat, &mut man);
        let c2 = cfg("cfg3");
        let m = feature::mel_spectrogram2(x.view(), &c2);
        let sh = m.shape().to_vec();
        dump(&dir, "mel_cfg3_out", &[sh[0], sh[1], sh[2]], m.as_standard_layout().as_slice().unwrap(), &mut man);
    }

    let mut f = File::create(format!("{}/manifest.txt", dir)).expect("manifest");
    for line in man {
        writeln!(f, "{}", line).expect("write");
    }
    println!("wrote {}", dir);
}
