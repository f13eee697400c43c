// Internal declarations shared by the host-side table builder (ss_host.cpp) and the HIP side
// (ss_kernels.hip, ss_api.hip).  Not part of the public ABI.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "speechsauce_amd.h"

namespace ss {

constexpr float kEpsF32 = 1.1920929e-7f;  // f32::EPSILON, functions.rs:70

struct Derived {
    // MFCC path (processing.rs:77-78)
    uint32_t flen = 0, step = 0;
    // STFT path (config.rs:154,178; functions.rs:96); stft_ok=false when fft_points < 2*hop
    bool stft_ok = false;
    uint32_t hop = 0, n_pad = 0;
    float wnorm = 0.f;
    uint32_t n_fft = 0, n_bins = 0, log2c = 0;  // c = n_fft/2 complex points
    // fft_points that are not a power of two: chirp-z (Bluestein) transform through a complex FFT of blu_len = 2^log2c
    // points, blu_len >= n_fft + n_fft/2 (bins 0..n_fft/2 of an n_fft-sample frame)
    bool bluestein = false;
    uint32_t blu_len = 0;
};

// Sparse triangular bank: filter m covers bins [start[m], start[m]+len[m]) with weights
// w[off[m] .. off[m]+len[m]).  Built from the dense bank so the numbers are the reference's.
struct SparseBank {
    std::vector<int32_t> start, len, off;
    std::vector<float> w;
    int32_t max_len = 0;
    int32_t last_bin = 0;  // one past the highest bin with a non-zero weight
};

struct HostTables {
    ss_params params{};
    Derived d{};
    std::vector<float> fb_dense;      // [M x F]
    std::vector<int32_t> fb_idx;      // [M+2]
    SparseBank bank;
    std::vector<float> window_mfcc;   // [flen] or empty (rect)
    std::vector<float> window_stft;   // [n_fft] Vorbis
    std::vector<float> tw_c;          // interleaved re,im: exp(-2*pi*i*t/C), t in [0,C)
    std::vector<float> tw_n;          // interleaved re,im: exp(-2*pi*i*k/N), k in [0,C/2]
    std::vector<float> dct;           // [num_cepstral x M] cos(pi*k*(2m+1)/(2M))
    // chirp-z tables (Derived::bluestein): c[n] = exp(-i pi n^2 / N), n < N; FFT_L of the wrapped conjugate chirp
    std::vector<float> blu_c, blu_b;
};

void set_error(const std::string &msg);
int fail(int status, const std::string &msg);

int validate(const ss_params &p);
int derive(const ss_params &p, Derived &d);
int num_frames(const ss_params &p, size_t n, size_t &t);
int stft_rows(const ss_params &p, size_t n, size_t &rows, size_t &real_rows);
int build_filterbank(const ss_params &p, std::vector<float> &fb, std::vector<int32_t> &idx);
void sparsify(const std::vector<float> &fb, size_t M, size_t F, SparseBank &out);
void vorbis_window(size_t n, float *w);
void hann_window(size_t n, float *w);
int build_tables(const ss_params &p, HostTables &t);

// Table block of the fft_points = 512 MFCC kernel (ss_mfcc512.hip), float offsets; global layout == LDS layout.
namespace fast512_layout {
constexpr int kTw2 = 0;                  // [8][16] float4: (W^(j(2p+1)), W^(j(2p+2))), W = exp(-2 pi i / 256); last .zw unused
constexpr int kTwn = kTw2 + 8 * 64;      // [8][16] float2
constexpr int kCos = kTwn + 8 * 32;      // [16][52]
constexpr int kStart = kCos + 16 * 52;   // [3][16] int32
constexpr int kFilt = kStart + 48;       // [3][16] int32: filter index of (slot, lane), -1 if none (mfe output order)
constexpr int kCosH = kFilt + 48;        // [16][20]: row c: cos(pi c (2m+1) / 80), m < 20, natural order (40 filters: symmetric DCT)
constexpr int kMelW = kCosH + 16 * 20;   // [16][pitch]
}  // namespace fast512_layout

// Filters sorted by tap count and dealt to 3 slots x 16 lanes so the lock-step tap loops are short;
// the log-mel row and the cosine table use the same (slot, lane) order.
struct Fast512Tables {
    bool ok = false;
    std::vector<float> tab;
    int32_t q4[3] = {0, 0, 0};  // taps / 4 per slot
    int32_t wpitch = 0;
    bool fullp = false;         // the bank reaches above bin 128 (librosa-style banks): the kernel keeps all 257 P bins
    bool paired = false;        // 40 filters: (slot, lane) cells laid out for the in-register symmetric DCT (build_fast512)
    bool tight = false;         // paired, and every filter of slot 1 / slot 2 lies inside the slot's first 6 / 2 taps (the kernel reads no more)
    int32_t win_floats = 0;     // frame window appended behind the mel rows (kMelW + 16 * wpitch): 512 floats (zero beyond flen), 0 = rectangular
};
void build_fast512(const HostTables &t, Fast512Tables &f);


// Table block of the fft_points = 2048 mel-spectrogram kernel (ss_mel2048.hip), float offsets.
namespace mel2048_layout {
constexpr int kTw2Pitch = 68;            // 16 float4 + 1 float4 pad
constexpr int kTw2 = 0;                  // [32 lanes][kTw2Pitch]: lane j, entry p: float4 (W^(j(2p+1)), W^(j(2p+2))), W = exp(-2 pi i / 1024)
// The untangle twiddles and the window are stored per lane, so that a lane fetches two of its values per ds_read_b128 (half
// the LDS instructions of the [r][lane] layout); row pitches are odd in 16-byte units: the lanes of a read spread over all banks
constexpr int kTwnPitch = 36;            // 16 float2 + 1 float4 pad
constexpr int kWinPitch = 68;            // 32 float2 + 1 float4 pad
constexpr int kTwn = kTw2 + 32 * kTw2Pitch + 4;  // [32 lanes][kTwnPitch]: lane j, entry r: exp(-2 pi i (j + 32 r) / 2048)
constexpr int kWin = kTwn + 32 * kTwnPitch;  // [32 lanes][kWinPitch]: lane j, entry e: Vorbis window pair (w[2n], w[2n+1]), n = j + 32 e
constexpr int kStart = kWin + 32 * kWinPitch;  // [4][32] int32: first P bin of the filter owned by (slot, lane)
constexpr int kFilt = kStart + 128;      // [4][32] int32: filter index of (slot, lane), -1 if none
constexpr int kMelW = kFilt + 128;       // [32][pitch]
constexpr int kPRow = 520;               // floats per P row: bins 0..512 + zero pad bins
}  // namespace mel2048_layout

struct Mel2048Tables {
    bool ok = false;
    bool fullp = false;      // the bank reaches past bin 512: P rows of all 1025 bins
    bool stft_only = false;  // the bank does not fit the kernel's mel stage: the block serves the stft build only
    std::vector<float> tab;
    int32_t q4[4] = {0, 0, 0, 0};
    int32_t wpitch = 0;
};
void build_mel2048(const HostTables &t, Mel2048Tables &f);

// Table block of the wide-bank fft_points = 512 MFCC / mfe kernel (ss_mfcc512w.hip), float offsets.
namespace mfcc512w_layout {
constexpr int kTw2 = 0;                  // [8][16] float4: (W^(j(2p+1)), W^(j(2p+2))), W = exp(-2 pi i / 256); last .zw unused
constexpr int kTwn = kTw2 + 8 * 64;      // [8][16] float2: exp(-2 pi i (j + 16 r) / 512)
constexpr int kCos = kTwn + 8 * 32;      // [32][84]: row c: cos(pi c (2m+1) / 2M) in (slot, lane) order, zero where no filter
constexpr int kCosPitch = 84;
constexpr int kStart = kCos + 32 * kCosPitch;  // [5][16] int32: first P bin of the filter owned by (slot, lane)
constexpr int kFilt = kStart + 80;       // [5][16] int32: filter index of (slot, lane), -1 if none
constexpr int kMelW = kFilt + 80;        // [16][pitch]; an optional frame window [512] follows
}  // namespace mfcc512w_layout

struct Mfcc512wTables {
    bool ok = false;
    bool windowed = false;
    std::vector<float> tab;
    int32_t q4[5] = {0, 0, 0, 0, 0};
    int32_t wpitch = 0;
};
void build_mfcc512w(const HostTables &t, Mfcc512wTables &f);

// Table block of the fft_points = 512 mel-spectrogram kernel (ss_mel512.hip), float offsets; global layout == LDS layout.
namespace mel512_layout {
constexpr int kTw2 = 0;                  // [8][16] float4: (W^(j(2p+1)), W^(j(2p+2))), W = exp(-2 pi i / 256); last .zw unused
constexpr int kTwn = kTw2 + 8 * 64;      // [8][16] float2: exp(-2 pi i (j + 16 r) / 512)
constexpr int kWin = kTwn + 8 * 32;      // [256] float2: Vorbis window pairs (w[2n], w[2n+1])
constexpr int kStart = kWin + 512;       // [5][16] int32: first P bin of the filter owned by (slot, lane)
constexpr int kFilt = kStart + 80;       // [5][16] int32: filter index of (slot, lane), -1 if none
constexpr int kMelW = kFilt + 80;        // [16][pitch]
}  // namespace mel512_layout

struct Mel512Tables {
    bool ok = false;
    bool stft_only = false;  // the bank does not fit the kernel's mel stage: the block serves the stft build only
    bool fullp = false;
    std::vector<float> tab;
    int32_t q4[5] = {0, 0, 0, 0, 0};
    int32_t wpitch = 0;
};
void build_mel512(const HostTables &t, Mel512Tables &f);

// Table block of the fft_points = 256 MFCC / mfe kernel (ss_mfcc256.hip), float offsets; global layout == LDS layout.
namespace mfcc256_layout {
constexpr int kTw2 = 0;                  // [8][16] float4: (W^(j(2p+1)), W^(j(2p+2))), W = exp(-2 pi i / 256); last .zw unused
constexpr int kCos = kTw2 + 8 * 64;      // [32][52]: row c: cos(pi c (2m+1) / 2M) in (slot, lane) order, zero where no filter
constexpr int kStart = kCos + 32 * 52;   // [3][16] int32: first P bin of the filter owned by (slot, lane)
constexpr int kFilt = kStart + 48;       // [3][16] int32: filter index of (slot, lane), -1 if none
constexpr int kMelW = kFilt + 48;        // [16][pitch]; an optional frame window [256] follows
}  // namespace mfcc256_layout

struct Mfcc256Tables {
    bool ok = false;
    bool windowed = false;
    std::vector<float> tab;
    int32_t q4[3] = {0, 0, 0};  // taps / 4 per slot
    int32_t wpitch = 0;
};
void build_mfcc256(const HostTables &t, Mfcc256Tables &f);

// Table block of the fft_points = 1024 MFCC / mfe kernel (ss_mfcc1024.hip), float offsets.  Reader lane jj = k1 + 16 h.
namespace mfcc1024_layout {
constexpr int kT1 = 0;                   // [8][16] float4: (W^(k1(2p+1)), W^(k1(2p+2))), W = exp(-2 pi i / 256)
constexpr int kT2 = kT1 + 8 * 64;        // [8][32] float2: exp(-2 pi i (k1 + 16 (i + 8 h)) / 512)
constexpr int kTwn = kT2 + 8 * 64;       // [8][32] float2: exp(-2 pi i (k1 + 16 i + 128 h) / 1024)
constexpr int kWin = kTwn + 8 * 64;      // [512] float2: frame window pairs (zero beyond flen); unused without a window
constexpr int kStart = kWin + 1024;      // [4][32] int32: first P bin (multiple of 4) of the filter owned by (slot, lane)
constexpr int kFilt = kStart + 128;      // [4][32] int32: filter index of (slot, lane), -1 if none
constexpr int kCos = kFilt + 128;        // [64][68]: row c: cos(pi c (2m+1) / 2M), m < (M+1)/2, zero padded
constexpr int kCosPitch = 68;
constexpr int kMelW = kCos + 64 * kCosPitch;  // [32][pitch]
}  // namespace mfcc1024_layout

struct Mfcc1024Tables {
    bool ok = false;
    bool stft_only = false;  // mel-spectrogram block whose bank does not fit the mel stage: stft build only
    bool windowed = false;
    bool fullp = false;  // the bank reaches past bin 256: LIB builds
    std::vector<float> tab;
    int32_t q4[4] = {0, 0, 0, 0};
    int32_t wpitch = 0;
};
void build_mfcc1024(const HostTables &t, Mfcc1024Tables &f);
void build_mel1024(const HostTables &t, Mfcc1024Tables &f);  // the mel-spectrogram kernel's block: Vorbis window, no cosines

// Table block of the fft_points = 2048 MFCC / mfe kernel (ss_mfcc2048.hip), float offsets; global layout == LDS layout.
namespace mfcc2048_layout {
constexpr int kTw2 = 0;                  // [16][32] float4: (W^(j(2p+1)), W^(j(2p+2))), W = exp(-2 pi i / 1024)
constexpr int kTwn = kTw2 + 16 * 128;    // [16][32] float2: exp(-2 pi i (j + 32 r) / 2048)
constexpr int kWin = kTwn + 16 * 64;     // [1024] float2: frame window pairs (zero beyond flen); unused without a window
constexpr int kStart = kWin + 2048;      // [4][32] int32: first P bin (multiple of 4) of the filter owned by (slot, lane)
constexpr int kFilt = kStart + 128;      // [4][32] int32: filter index of (slot, lane), -1 if none
constexpr int kCos = kFilt + 128;        // [64][68]: row c: cos(pi c (2m+1) / 2M), m < (M+1)/2 (the other half by symmetry), zero padded
constexpr int kCosPitch = 68;
constexpr int kMelW = kCos + 64 * kCosPitch;  // [32][pitch]
}  // namespace mfcc2048_layout

struct Mfcc2048Tables {
    bool ok = false;
    bool windowed = false;
    bool fullp = false;  // the bank reaches past bin 512 (librosa-style banks up to fs/2): LIB builds
    std::vector<float> tab;
    int32_t q4[4] = {0, 0, 0, 0};
    int32_t wpitch = 0;
};
void build_mfcc2048(const HostTables &t, Mfcc2048Tables &f);

// Table block of the fft_points = 4096 MFCC kernel (ss_mfcc4096.hip), float offsets.  Reader lane L' = k1 + 32 a.
namespace mfcc4096_layout {
constexpr int kT1 = 0;                    // [16][32] float4: (W^(k1(2p+1)), W^(k1(2p+2))), W = exp(-2 pi i / 1024)
constexpr int kT2 = kT1 + 16 * 128;       // [16][64] float2: exp(-2 pi i (k1 + 32 (i + 16 h)) / 2048), lane = k1 + 32 h
constexpr int kTwn = kT2 + 16 * 128;      // [16][64] float2: exp(-2 pi i (k1 + 32 i + 512 h) / 4096)
constexpr int kStart = kTwn + 16 * 128;   // [4][64] int32: (slot, lane) -> first P bin of its filter (low 16 bits) | filter index << 16 (-1: none)
constexpr int kCos = kStart + 256;        // [n_ceps][132]: cos(pi c (2m+1) / 2M), m < 128 (the other half by symmetry)
constexpr int kCosPitch = 132;
constexpr int kCosLanePitch = 68;         // dct_fold2 layout: [64 lanes][68], the 64 cosines of the lane's share of its coefficient
constexpr int kPRow = 1032;               // floats per P row: bins 0..1024 + zero pad bins
// melw [64][pitch] follows the cosine block: offset kCos + cos_floats
}  // namespace mfcc4096_layout

struct Mfcc4096Tables {
    bool ok = false;
    bool stft_only = false;  // mel-spectrogram block whose bank does not fit the mel stage: stft build only
    bool dct_fold2 = false;  // the cosine block holds one 64-term row per lane ([64][kCosLanePitch]) instead of [n_ceps][kCosPitch]
    int32_t cos_floats = 0;  // floats of the cosine block (the mel rows follow it)
    std::vector<float> tab;
    int32_t q4[4] = {0, 0, 0, 0};
    int32_t wpitch = 0;
};
void build_mfcc4096(const HostTables &t, Mfcc4096Tables &f);
void build_mel4096(const HostTables &t, Mfcc4096Tables &f);  // the mel-spectrogram kernel's block: no cosines, Vorbis window appended

}  // namespace ss
