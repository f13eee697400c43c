// C ABI entry points that touch the device (include/speechsauce_amd.h).  Host-pointer variants
// stage through device buffers; device-pointer variants only enqueue kernels on the caller's
// stream.  There is no CPU compute path here: without a HIP device these return SS_ERR_HIP.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ss_device.h"
#include "ss_internal.h"

struct ss_config {
    ss::HostTables host;
    int device = -1;
    int num_cus = 256;
    // device tables
    float *d_window_mfcc = nullptr;
    float *d_window_stft = nullptr;
    float2 *d_tw_c = nullptr;
    float2 *d_tw_n = nullptr;
    float2 *d_blu_c = nullptr;  // chirp-z tables (fft_points not a power of two)
    float2 *d_blu_b = nullptr;
    int32_t *d_f_start = nullptr, *d_f_len = nullptr, *d_f_off = nullptr;
    float *d_f_w = nullptr;
    float *d_dct = nullptr;
    // fft_points = 512 MFCC kernel tables (ss_mfcc512.hip)
    ss::Fast512Tables fast;
    float *d_fast_tab = nullptr;
    // fft_points = 2048 mel-spectrogram kernel tables (ss_mel2048.hip)
    ss::Mel2048Tables mel2048;
    float *d_mel2048_tab = nullptr;
    // fft_points = 4096 MFCC kernel tables (ss_mfcc4096.hip)
    ss::Mfcc4096Tables mfcc4096;
    float *d_mfcc4096_tab = nullptr;
    // fft_points = 2048 MFCC kernel tables (ss_mfcc2048.hip)
    ss::Mfcc2048Tables mfcc2048;
    float *d_mfcc2048_tab = nullptr;
    // fft_points = 1024 MFCC kernel tables (ss_mfcc1024.hip)
    ss::Mfcc1024Tables mfcc1024;
    float *d_mfcc1024_tab = nullptr;
    // fft_points = 256 MFCC kernel tables (ss_mfcc256.hip)
    ss::Mfcc256Tables mfcc256;
    float *d_mfcc256_tab = nullptr;
    // wide-bank fft_points = 512 MFCC kernel tables (ss_mfcc512w.hip)
    ss::Mfcc512wTables mfcc512w;
    float *d_mfcc512w_tab = nullptr;
    // fft_points = 512 mel-spectrogram kernel tables (ss_mel512.hip)
    ss::Mel512Tables mel512;
    float *d_mel512_tab = nullptr;
    // fft_points = 1024 mel-spectrogram kernel tables (ss_mel_c512 in ss_mfcc1024.hip)
    ss::Mfcc1024Tables mel1024;
    float *d_mel1024_tab = nullptr;
    // fft_points = 4096 mel-spectrogram kernel tables (ss_mel_c2048 in ss_mfcc4096.hip)
    ss::Mfcc4096Tables mel4096;
    float *d_mel4096_tab = nullptr;
    // Host-pointer entry points (ss_mfcc_batch, ...): two private streams and two sets of device buffers, kept with the
    // config so that a call costs no hipMalloc / hipFree and never touches the null stream.  One host call at a time per
    // config (the mutex); calls on different configs, and device-pointer calls, run concurrently.
    struct HostPipe {
        std::mutex mu;
        hipStream_t stream[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        void *d_in[2] = {nullptr, nullptr}, *d_out0[2] = {nullptr, nullptr}, *d_out1[2] = {nullptr, nullptr};
        size_t cap_in = 0, cap_out0 = 0, cap_out1 = 0;
        // small calls (one utterance): pinned, device-mapped staging the kernel reads and writes over PCIe itself -- no
        // copy commands on the stream.  [0] input, [1] / [2] outputs; host address and the device's view of it
        void *h_small[3] = {nullptr, nullptr, nullptr}, *d_small[3] = {nullptr, nullptr, nullptr};
        size_t cap_small[3] = {0, 0, 0};
    };
    mutable HostPipe pipe;
    // Device error word: one pinned, device-mapped word a kernel sets when it detects a condition that must not pass for a
    // result (today: a tile hand-off of ss_mel_c1024<tile> that never came).  The host reads it without a copy; a non-zero
    // word turns into SS_ERR_DEVICE at the next launch or synchronisation point on this config (pending_device_error).
    unsigned *h_err = nullptr, *d_err = nullptr;
    mutable unsigned tile_spin_limit = 1u << 24;  // what the word behind the 2048-point mel table block holds (test aid toggles it)
    mutable unsigned tile_spin_stage[4] = {0, 0, 0, 0};
};

#if SS_LAB
// Process-wide test aids of include/speechsauce_amd_debug.h: LAB BUILDS ONLY.  The product library's kernel selection is a pure
// function of the configuration and the call (ss_device.h has the constant forms of these accessors).
namespace {
extern std::atomic<int> g_force_generic, g_mel_tile_off;
extern std::atomic<unsigned> g_tile_fault;
}  // namespace
namespace ss {
bool dbg_force_generic() { return g_force_generic.load(std::memory_order_relaxed) != 0; }
bool dbg_mel_tile_off() { return g_mel_tile_off.load(std::memory_order_relaxed) == 1; }
int dbg_mel_build() { return g_mel_tile_off.load(std::memory_order_relaxed); }
// a forced fault: no polling at all -- the first hand-off that is not there at once counts as lost
unsigned dbg_tile_spin_limit() { return g_tile_fault.load(std::memory_order_relaxed) ? 0u : (1u << 24); }
}  // namespace ss
#endif

namespace {

thread_local const char *g_last_kernel = "";
// per-wave stamps of the 512-point MFCC kernel: the buffer of the CURRENT call on this thread (ss_mfcc_shader_clock sets it
// around its own launches; nothing process-wide in the product build)
thread_local unsigned long long *g_call_stamps = nullptr;
// the same for the kernels that write two words per wave (cycles lived, 100 MHz ticks lived): the twelve-wave builds of the
// 4096-point MFCC and the 2048-point mel kernel (the *_timed_region diagnostics set it around their own launches)
thread_local unsigned long long *g_call_stamps2 = nullptr;
#if SS_LAB
std::atomic<unsigned long long *> g_stamp_buffer{nullptr};  // ss_debug_stamp_buffer
// test aids of include/speechsauce_amd_debug.h (process-wide)
std::atomic<int> g_force_generic{0}, g_mel_tile_off{0};
std::atomic<unsigned> g_tile_fault{0};
#endif

int hip_fail(hipError_t e, const char *what)
{
    return ss::fail(SS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define SS_HIP(call)                                  \
    do {                                              \
        hipError_t e_ = (call);                       \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

template <typename T>
int upload(T **dst, const void *src, size_t bytes)
{
    *dst = nullptr;
    if (bytes == 0) return SS_OK;
    SS_HIP(hipMalloc(reinterpret_cast<void **>(dst), bytes));
    SS_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return SS_OK;
}

struct DeviceBuf {
    void *p = nullptr;
    ~DeviceBuf()
    {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes)
    {
        SS_HIP(hipMalloc(&p, bytes ? bytes : 16));
        return SS_OK;
    }
    template <typename T>
    T *as() { return static_cast<T *>(p); }
};

int check_device(const ss_config *cfg);

// SS_ERR_DEVICE if a kernel of an earlier launch on this config has set the device error word (and clears it).
int pending_device_error(const ss_config *cfg)
{
    if (!cfg->h_err || __atomic_load_n(cfg->h_err, __ATOMIC_ACQUIRE) == 0) return SS_OK;
    const unsigned w = __atomic_exchange_n(cfg->h_err, 0u, __ATOMIC_ACQ_REL);
    if (w == 0) return SS_OK;
    return ss::fail(SS_ERR_DEVICE, "a kernel launched on this config reported a device-side protocol error (error word " + std::to_string(w) +
                                       "): the results of that launch are incomplete");
}

void fill_common(const ss_config *cfg, ss::FrontArgs &a)
{
    const ss::HostTables &h = cfg->host;
    a.tw_c = cfg->d_tw_c;
    a.tw_n = cfg->d_tw_n;
    a.blu_c = cfg->d_blu_c;
    a.blu_b = cfg->d_blu_b;
    a.blu_n = h.d.bluestein ? h.params.fft_points : 0u;
    a.f_start = cfg->d_f_start;
    a.f_len = cfg->d_f_len;
    a.f_off = cfg->d_f_off;
    a.f_w = cfg->d_f_w;
    a.n_filters = h.params.num_filters;
    a.dct = cfg->d_dct;
    a.n_ceps = h.params.num_cepstral;
    a.dc_elimination = h.params.dc_elimination;
    a.spectrum_exponent = h.params.spectrum_exponent;
}

// MFCC-path launch (OUT_MFCC / OUT_MFE / OUT_POWER).
// rows_are_frames: d_x is a frames matrix [batch x n] (row stride ld) -- every row is one frame of n <= fft_points samples,
// no window, no pre-emphasis (processing::power_spectrum(frames, fft_points), processing.rs:179-181).
// `multi` (ss_mfcc_batches_device): several independent batches of the same clip shape for ONE launch; d_x / batch / out0 then
// describe the first of them.  Only the kernel builds that take a batch table serve it: for every other configuration NOTHING is
// launched and kNoMultiBuild comes back (the caller then launches batch by batch).
struct MultiBatches {
    int n;
    const float *const *x;
    float *const *out;
    const size_t *clips;
};
constexpr int kNoMultiBuild = -1;

int launch_frames(const ss_config *cfg, int out_kind, const float *d_x, size_t batch, size_t n, size_t ld,
                  float *out0, float *out1, hipStream_t stream, bool rows_are_frames = false, const MultiBatches *multi = nullptr)
{
    if (!cfg) return ss::fail(SS_ERR_ARG, "null config");
    if (batch == 0) return SS_OK;  // an empty batch has no buffers (a zero-row tensor's data pointer is null)
    if (!d_x || !out0) return ss::fail(SS_ERR_ARG, "null buffer");
    if (ld < n) return ss::fail(SS_ERR_ARG, "leading dimension smaller than n_samples");
    if (n > 0x7fffffffull || batch > 0x7fffffffull) return ss::fail(SS_ERR_ARG, "clip too long / batch too large");
    {
        // the tables live on the device the config was created on: a launch from another current device would hand this
        // device pointers into that one's memory
        const int drc = check_device(cfg);
        if (drc) return drc;
        const int erc = pending_device_error(cfg);  // no more work behind a launch that reported a device-side error
        if (erc) return erc;
    }
    const ss::HostTables &h = cfg->host;
    size_t T = 0;
    int rc = SS_OK;
    if (rows_are_frames) {
        if (n == 0 || n > h.params.fft_points) return ss::fail(SS_ERR_ARG, "frame length must be in [1, fft_points]");
        T = 1;
    } else {
        rc = ss::num_frames(h.params, n, T);
    }
    if (rc) return rc;
    ss::FrontArgs a{};
    fill_common(cfg, a);
    a.x = d_x;
    a.ld = ld;
    a.n_samples = static_cast<uint32_t>(n);
    a.batch = static_cast<uint32_t>(batch);
    a.flen = rows_are_frames ? static_cast<uint32_t>(n) : h.d.flen;
    a.step = rows_are_frames ? static_cast<uint32_t>(n) : h.d.step;
    a.n_frames = static_cast<uint32_t>(T);
    // processing.rs:110-120 as written: nothing is copied for > 2 frames, x[0..flen] into every row otherwise
    if (rows_are_frames) a.frame_mode = ss::FRAME_NORMAL;
    else if (h.params.framing == SS_FRAMING_LITERAL) a.frame_mode = T > 2 ? ss::FRAME_ZERO : ss::FRAME_FIRST;
    else if (h.params.framing == SS_FRAMING_CENTER) a.frame_mode = ss::FRAME_CENTER;  // librosa center=True (generic kernel only)
    else if (h.params.framing == SS_FRAMING_PADDED) a.frame_mode = ss::FRAME_PADDED;  // zero_padding = true (generic kernel only)
    else a.frame_mode = ss::FRAME_NORMAL;
    a.pad_reflect = h.params.pad_mode == SS_PAD_REFLECT;
    a.preemph = rows_are_frames ? 0.0f : h.params.preemph_coef;
    a.preemph_shift = static_cast<uint32_t>(h.params.preemph_shift > 0 ? h.params.preemph_shift : 1);
    a.window = rows_are_frames ? nullptr : cfg->d_window_mfcc;
    a.scale = 1.0f / static_cast<float>(h.params.fft_points);  // processing.rs:180
    // feature.rs:126-131 (n = T * M as f32) or scipy ortho over the axis length
    const float g = h.params.dct2_gain;
    const float M = static_cast<float>(h.params.num_filters);
    if (h.params.dct_norm == SS_DCT_ORTHO) {
        a.dct_scale_k = g * (1.0f / sqrtf(2.0f * M));
        a.dct_scale_0 = a.dct_scale_00 = g * (1.0f / sqrtf(4.0f * M));
    } else {
        const float nn = static_cast<float>(T * h.params.num_filters);
        a.dct_scale_k = g * (1.0f / sqrtf(2.0f * nn));
        a.dct_scale_0 = g;
        a.dct_scale_00 = g * (1.0f / sqrtf(4.0f * nn));
    }
    a.out_kind = out_kind;
    a.out0 = out0;
    a.out1 = out1;
    ss::LaunchInfo info{};
    // fft_points = 512 MFCC: the specialised wave-private kernel (its builds: default bank / run-time bank, window,
    // pre-emphasis, mfe and power outputs, librosa variants)
    const bool force_generic = ss::dbg_force_generic();
    // the mfe-output build of that kernel exists for the default bank shape only
    const bool mfe_shape = a.flen == 320 && a.spectrum_exponent != 2 && cfg->fast.q4[0] == 4 && cfg->fast.q4[1] == 2 &&
                           cfg->fast.q4[2] == 1 && a.n_filters <= 40;
    const bool front = a.preemph != 0.0f || a.window != nullptr;  // optional window / fused pre-emphasis: default-bank build only
    // librosa-compatible variants: centred frames and banks over the whole spectrum have MFCC builds of their own (optional
    // window, no fused pre-emphasis)
    const bool centre = a.frame_mode == ss::FRAME_CENTER;
    const bool lib_variant = centre || cfg->fast.fullp;
    const bool lib_ok = out_kind == ss::OUT_MFCC && a.preemph == 0.0f;
    // Sample pairs load as 8-byte words at dword alignment (gfx950 global loads need no more: tools/unaligned_probe.py), and an
    // odd frame length ends in a half pair that one lane loads as a single float -- so hops, leading dimensions, base offsets
    // and frame lengths of either parity reach the dedicated kernels.
    const bool fast_ok = !force_generic && cfg->fast.ok &&
                         (out_kind == ss::OUT_MFCC || (out_kind == ss::OUT_MFE && mfe_shape) || (out_kind == ss::OUT_POWER && mfe_shape && !front)) &&
                         (lib_variant ? lib_ok : (!front || mfe_shape)) && (a.frame_mode == ss::FRAME_NORMAL || centre);
    const bool fits32 = static_cast<unsigned long long>(batch) * T < 0xffffffffull;
#if SS_LAB
    static const char *dbg_path = std::getenv("SS_DEBUG_TIMES");  // diagnostic only (lab build): per-wave realtime stamps of ONE launch
    static bool dbg_done = false;
#endif
    if (fast_ok && fits32) {
        ss::Fast512Args f{};
        f.x = d_x;
        f.ld = ld;
        f.n_samples = a.n_samples;
        f.batch = a.batch;
        f.flen = a.flen;
        f.step = a.step;
        f.n_frames = a.n_frames;
        f.scale = a.scale;
        f.spectrum_exponent = a.spectrum_exponent;
        f.tab = cfg->d_fast_tab;
        f.mel_wpitch = cfg->fast.wpitch;
        for (int s = 0; s < 3; ++s) f.mel_q4[s] = cfg->fast.q4[s];
        f.n_filters = a.n_filters;
        f.n_ceps = a.n_ceps;
        f.dct_scale_k = a.dct_scale_k;
        f.dct_scale_0 = a.dct_scale_0;
        f.dct_scale_00 = a.dct_scale_00;
        f.dc_elimination = a.dc_elimination;
        f.out = out0;
        f.out_energy = out1;
        f.out_mfe = out_kind == ss::OUT_MFE ? 1 : (out_kind == ss::OUT_POWER ? 2 : 0);
        f.win_floats = a.window ? cfg->fast.win_floats : 0;
        f.preemph = a.preemph;
        f.preemph_shift = a.preemph_shift;
        f.center = centre;
        f.pad_reflect = a.pad_reflect;
        f.fullp = cfg->fast.fullp;
        f.paired = cfg->fast.paired ? (cfg->fast.tight ? 2 : 1) : 0;
#if SS_LAB
        if (dbg_path && !dbg_done && !f.out_mfe && !multi) {
            dbg_done = true;
            const size_t nwaves = static_cast<size_t>(cfg->num_cus) * 16;
            DeviceBuf db;
            int rc2 = db.alloc(nwaves * 6 * sizeof(unsigned long long));
            if (rc2) return rc2;
            DeviceBuf flush;  // larger than the Infinity Cache: the stamped launch reads its samples from HBM like a bench step
            rc2 = flush.alloc(512ull << 20);
            if (rc2) return rc2;
            // the last repetition has warm code / tables and cold samples; SS_DEBUG_TIMES_REPS=<n> stamps n launches and
            // writes <file>.<k> for each from the third on (are the same workgroups late every time?)
            const char *reps_env = std::getenv("SS_DEBUG_TIMES_REPS");
            const int reps = reps_env && std::atoi(reps_env) > 3 ? std::atoi(reps_env) : 3;
            std::vector<unsigned long long> hb(nwaves * 6);
            for (int rep = 0; rep < reps; ++rep) {
                SS_HIP(hipMemsetAsync(flush.p, rep, 512ull << 20, stream));
                SS_HIP(hipMemsetAsync(db.p, 0, nwaves * 6 * sizeof(unsigned long long), stream));
                f.dbg = db.as<unsigned long long>();
                hipError_t e2 = ss::launch_mfcc_c256(f, stream, cfg->num_cus, &info);
                if (e2 != hipSuccess) return hip_fail(e2, "launch_mfcc_c256");
                SS_HIP(hipStreamSynchronize(stream));
                if (rep < 2) continue;
                SS_HIP(hipMemcpy(hb.data(), db.p, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                const std::string path = rep == reps - 1 ? std::string(dbg_path) : std::string(dbg_path) + "." + std::to_string(rep);
                if (FILE *fp = std::fopen(path.c_str(), "w")) {
                    for (size_t w = 0; w < nwaves; ++w)
                        if (hb[6 * w + 2])
                            std::fprintf(fp, "%zu %llu %llu %llu %llu %llu %llu %llu\n", w, hb[6 * w], hb[6 * w + 1], hb[6 * w + 2],
                                         hb[6 * w + 3] >> 32, hb[6 * w + 3] & 0xffffffffull, hb[6 * w + 4], hb[6 * w + 5]);
                    std::fclose(fp);
                }
            }
            f.dbg = nullptr;
        }
#endif
        f.dbg = g_call_stamps;
#if SS_LAB
        if (!f.dbg) f.dbg = g_stamp_buffer.load(std::memory_order_relaxed);
#endif
        if (multi) {
            const hipError_t em = ss::launch_mfcc_c256_multi(f, multi->n, multi->x, multi->out, multi->clips, stream, cfg->num_cus, &info);
            if (em == hipErrorInvalidValue) return kNoMultiBuild;  // (before the launch: this shape has no batch-table build)
            if (em != hipSuccess) return hip_fail(em, "launch_mfcc_c256_multi");
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
        const hipError_t e = ss::launch_mfcc_c256(f, stream, cfg->num_cus, &info);
        if (e == hipSuccess) {
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
        // hipErrorInvalidValue before the launch: the configuration does not fit this kernel (LDS budget) -> next candidate
        if (e != hipErrorInvalidValue) return hip_fail(e, "launch_mfcc_c256");
    }
    // fft_points = 4096 MFCC: the twelve-wave default-shape build takes a batch table too
    auto fill4096 = [&](ss::Mfcc4096Args &f) {
        f.preemph = a.preemph;
        f.preemph_shift = a.preemph_shift;
        f.x = d_x;
        f.ld = ld;
        f.n_samples = a.n_samples;
        f.batch = a.batch;
        f.flen = a.flen;
        f.step = a.step;
        f.n_frames = a.n_frames;
        f.scale = a.scale;
        f.spectrum_exponent = a.spectrum_exponent;
        f.tab = cfg->d_mfcc4096_tab;
        f.mel_wpitch = cfg->mfcc4096.wpitch;
        for (int s = 0; s < 4; ++s) f.mel_q4[s] = cfg->mfcc4096.q4[s];
        f.cos_floats = cfg->mfcc4096.cos_floats;
        f.dct_fold2 = cfg->mfcc4096.dct_fold2 ? 1 : 0;
        f.n_filters = a.n_filters;
        f.n_ceps = a.n_ceps;
        f.dct_scale_k = a.dct_scale_k;
        f.dct_scale_0 = a.dct_scale_0;
        f.dct_scale_00 = a.dct_scale_00;
        f.dc_elimination = a.dc_elimination;
        f.out = out0;
        f.out_energy = out1;
        f.out_mfe = out_kind == ss::OUT_MFE;
        f.window = a.window;
        f.stamps = g_call_stamps2;
    };
    if (multi) {
        if (!force_generic && cfg->mfcc4096.ok && out_kind == ss::OUT_MFCC && a.frame_mode == ss::FRAME_NORMAL) {
            ss::Mfcc4096Args f{};
            fill4096(f);
            const hipError_t em = ss::launch_mfcc_c2048_multi(f, multi->n, multi->x, multi->out, multi->clips, stream, cfg->num_cus, &info);
            if (em == hipSuccess) {
                g_last_kernel = info.kernel_name;
                return SS_OK;
            }
            if (em != hipErrorInvalidValue) return hip_fail(em, "launch_mfcc_c2048_multi");
        }
        return kNoMultiBuild;  // the other kernels take one batch per launch
    }
    // fft_points = 512 MFCC / mfe with more than 48 filters or 16 cepstra, and the output / window / framing combinations the
    // headline kernel has no build for (ss_mfcc512w.hip): optional frame window, centred frames, fused pre-emphasis
    if (!force_generic && cfg->mfcc512w.ok && static_cast<unsigned long long>(batch) * T + 4 < 0x7fffffffull &&
        (out_kind == ss::OUT_MFCC || out_kind == ss::OUT_MFE) && (a.frame_mode == ss::FRAME_NORMAL || centre)) {
        ss::Mfcc256Args f{};
        f.center = centre;
        f.pad_reflect = a.pad_reflect;
        f.preemph = a.preemph;
        f.preemph_shift = a.preemph_shift;
        f.x = d_x;
        f.ld = ld;
        f.n_samples = a.n_samples;
        f.batch = a.batch;
        f.flen = a.flen;
        f.step = a.step;
        f.n_frames = a.n_frames;
        f.scale = a.scale;
        f.spectrum_exponent = a.spectrum_exponent;
        f.tab = cfg->d_mfcc512w_tab;
        f.mel_wpitch = cfg->mfcc512w.wpitch;
        for (int s = 0; s < 5; ++s) f.mel_q4[s] = cfg->mfcc512w.q4[s];
        f.n_filters = a.n_filters;
        f.n_ceps = a.n_ceps;
        f.dct_scale_k = a.dct_scale_k;
        f.dct_scale_0 = a.dct_scale_0;
        f.dct_scale_00 = a.dct_scale_00;
        f.dc_elimination = a.dc_elimination;
        f.windowed = cfg->mfcc512w.windowed;
        f.out_mfe = out_kind == ss::OUT_MFE;
        f.out = out0;
        f.out_energy = out1;
        const hipError_t e5 = ss::launch_mfcc_c256w(f, stream, cfg->num_cus, &info);
        if (e5 == hipSuccess) {
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
        // hipErrorInvalidValue before the launch: the configuration does not fit this kernel (LDS budget) -> next candidate
        if (e5 != hipErrorInvalidValue) return hip_fail(e5, "launch_mfcc_c256w");
    }
    // fft_points = 256 MFCC / mfe: two frames per complex transform (ss_mfcc256.hip); optional frame window, fused pre-emphasis
    if (!force_generic && cfg->mfcc256.ok && static_cast<unsigned long long>(batch) * T + 8 < 0x7fffffffull &&
        (out_kind == ss::OUT_MFCC || out_kind == ss::OUT_MFE) && a.frame_mode == ss::FRAME_NORMAL && a.flen <= 256) {
        ss::Mfcc256Args f{};
        f.preemph = a.preemph;
        f.preemph_shift = a.preemph_shift;
        f.x = d_x;
        f.ld = ld;
        f.n_samples = a.n_samples;
        f.batch = a.batch;
        f.flen = a.flen;
        f.step = a.step;
        f.n_frames = a.n_frames;
        f.scale = a.scale;
        f.spectrum_exponent = a.spectrum_exponent;
        f.tab = cfg->d_mfcc256_tab;
        f.mel_wpitch = cfg->mfcc256.wpitch;
        for (int s = 0; s < 3; ++s) f.mel_q4[s] = cfg->mfcc256.q4[s];
        f.n_filters = a.n_filters;
        f.n_ceps = a.n_ceps;
        f.dct_scale_k = a.dct_scale_k;
        f.dct_scale_0 = a.dct_scale_0;
        f.dct_scale_00 = a.dct_scale_00;
        f.dc_elimination = a.dc_elimination;
        f.windowed = cfg->mfcc256.windowed;
        f.out_mfe = out_kind == ss::OUT_MFE;
        f.out = out0;
        f.out_energy = out1;
        const hipError_t e3 = ss::launch_mfcc_c256x2(f, stream, cfg->num_cus, &info);
        if (e3 == hipSuccess) {
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
        // hipErrorInvalidValue before the launch: the configuration does not fit this kernel (LDS budget) -> next candidate
        if (e3 != hipErrorInvalidValue) return hip_fail(e3, "launch_mfcc_c256x2");
    }
    // fft_points = 2048 / 1024 MFCC / mfe: two frames per wave (ss_mfcc2048.hip, ss_mfcc1024.hip), optional frame window
    // (both have librosa-compatible builds: centred frames, banks up to fs/2)
    if (!force_generic && (cfg->mfcc2048.ok || cfg->mfcc1024.ok) && fits32 && (out_kind == ss::OUT_MFCC || out_kind == ss::OUT_MFE) &&
        (a.frame_mode == ss::FRAME_NORMAL || centre)) {
        ss::Mfcc2048Args f{};
        f.preemph = a.preemph;
        f.preemph_shift = a.preemph_shift;
        f.x = d_x;
        f.ld = ld;
        f.n_samples = a.n_samples;
        f.batch = a.batch;
        f.flen = a.flen;
        f.step = a.step;
        f.n_frames = a.n_frames;
        f.scale = a.scale;
        f.spectrum_exponent = a.spectrum_exponent;
        const bool k2048 = cfg->mfcc2048.ok;
        f.tab = k2048 ? cfg->d_mfcc2048_tab : cfg->d_mfcc1024_tab;
        f.mel_wpitch = k2048 ? cfg->mfcc2048.wpitch : cfg->mfcc1024.wpitch;
        for (int s = 0; s < 4; ++s) f.mel_q4[s] = k2048 ? cfg->mfcc2048.q4[s] : cfg->mfcc1024.q4[s];
        f.n_filters = a.n_filters;
        f.n_ceps = a.n_ceps;
        f.dct_scale_k = a.dct_scale_k;
        f.dct_scale_0 = a.dct_scale_0;
        f.dct_scale_00 = a.dct_scale_00;
        f.dc_elimination = a.dc_elimination;
        f.windowed = k2048 ? cfg->mfcc2048.windowed : cfg->mfcc1024.windowed;
        f.out_mfe = out_kind == ss::OUT_MFE;
        f.center = centre;
        f.pad_reflect = a.pad_reflect;
        f.fullp = k2048 ? cfg->mfcc2048.fullp : cfg->mfcc1024.fullp;
        f.out = out0;
        f.out_energy = out1;
        const hipError_t e2 = k2048 ? ss::launch_mfcc_c1024(f, stream, cfg->num_cus, &info) : ss::launch_mfcc_c512(f, stream, cfg->num_cus, &info);
        if (e2 == hipSuccess) {
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
        // hipErrorInvalidValue before the launch: the configuration does not fit this kernel (LDS budget) -> next candidate
        if (e2 != hipErrorInvalidValue) return hip_fail(e2, k2048 ? "launch_mfcc_c1024" : "launch_mfcc_c512");
    }
    // fft_points = 4096 MFCC / mfe (up to 256 filters): the one-frame-per-wave kernel
    if (!force_generic && cfg->mfcc4096.ok && fits32 && (out_kind == ss::OUT_MFCC || out_kind == ss::OUT_MFE) && a.frame_mode == ss::FRAME_NORMAL) {
        ss::Mfcc4096Args f{};
        fill4096(f);
#if SS_LAB
        static const char *rows_path = std::getenv("SS_DEBUG_ROWS");  // diagnostic only (lab build): frame 0's P row and ln(mel) row
#else
        constexpr const char *rows_path = nullptr;
#endif
        constexpr size_t kDbgFloats = 131072;  // >= 1028 + 256 + 4 * 4096 (stage dumps) and >= waves x 16 x 2 (SS_PROF5 phase sums)
        if (rows_path && hipMalloc(reinterpret_cast<void **>(&f.dbg), kDbgFloats * sizeof(float)) == hipSuccess)
            (void)hipMemsetAsync(f.dbg, 0, kDbgFloats * sizeof(float), stream);
        const hipError_t e4 = ss::launch_mfcc_c2048(f, stream, cfg->num_cus, &info);
        if (e4 != hipSuccess && f.dbg) (void)hipFree(f.dbg);
        // hipErrorInvalidValue before the launch: the configuration does not fit this kernel (LDS budget) -> generic kernel
        if (e4 != hipSuccess && e4 != hipErrorInvalidValue) return hip_fail(e4, "launch_mfcc_c2048");
        if (e4 == hipSuccess && f.dbg) {
            std::vector<float> rows(kDbgFloats);
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(rows.data(), f.dbg, rows.size() * sizeof(float), hipMemcpyDeviceToHost);
            (void)hipFree(f.dbg);
            if (FILE *fp = std::fopen(rows_path, "wb")) {
                std::fwrite(rows.data(), sizeof(float), rows.size(), fp);
                std::fclose(fp);
            }
        }
        if (e4 == hipSuccess) {
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
    }
    hipError_t e = ss::launch_front_generic(a, h.d.log2c, stream, cfg->num_cus, &info);
    if (e != hipSuccess) return hip_fail(e, "launch_front_generic");
    g_last_kernel = info.kernel_name;
    return SS_OK;
}

// STFT-path launch (OUT_MEL / OUT_STFT).
int launch_stft(const ss_config *cfg, int out_kind, const float *d_x, size_t channels, size_t n, size_t ld,
                float *out0, hipStream_t stream, const MultiBatches *multi = nullptr)
{
    if (!cfg) return ss::fail(SS_ERR_ARG, "null config");
    if (channels == 0) return SS_OK;  // nothing to do, and no buffers to check
    if (!d_x || !out0) return ss::fail(SS_ERR_ARG, "null buffer");
    if (ld < n) return ss::fail(SS_ERR_ARG, "leading dimension smaller than n_samples");
    if (n == 0 || n > 0x7fffffffull || channels > 0x7fffffffull) return ss::fail(SS_ERR_ARG, "bad clip length / channel count");
    {
        const int drc = check_device(cfg);  // see launch_frames
        if (drc) return drc;
        const int erc = pending_device_error(cfg);
        if (erc) return erc;
    }
    const ss::HostTables &h = cfg->host;
    size_t R = 0, Rreal = 0;
    int rc = ss::stft_rows(h.params, n, R, Rreal);
    if (rc) return rc;
    ss::FrontArgs a{};
    fill_common(cfg, a);
    a.x = d_x;
    a.ld = ld;
    a.n_samples = static_cast<uint32_t>(n);
    a.batch = static_cast<uint32_t>(channels);
    a.hop = h.d.hop;
    a.n_pad = h.d.n_pad;
    a.rows = static_cast<uint32_t>(R);
    a.real_rows = static_cast<uint32_t>(Rreal);
    a.window = cfg->d_window_stft;
    a.scale = h.d.wnorm;
    a.out_kind = out_kind;
    a.out0 = out0;
    ss::LaunchInfo info{};
    // fft_points = 2048 mel spectrogram: the wave-private kernel when its layout assumptions hold
    const bool force_generic = ss::dbg_force_generic();
    const bool want_stft = out_kind == ss::OUT_STFT;  // the stft builds do not use the bank (stft_only table blocks)
    if (!force_generic && (out_kind == ss::OUT_MEL || want_stft) && (cfg->mel2048.ok || (want_stft && cfg->mel2048.stft_only)) &&
        static_cast<unsigned long long>(a.rows + a.n_pad + 1) * a.hop < 0x7fffffffull) {
        ss::Mel2048Args m{};
        m.x = d_x;
        m.ld = ld;
        m.n_samples = a.n_samples;
        m.batch = a.batch;
        m.hop = a.hop;
        m.n_pad = a.n_pad;
        m.rows = a.rows;
        m.real_rows = a.real_rows;
        m.scale = a.scale;
        m.tab = cfg->d_mel2048_tab;
        m.fullp = cfg->mel2048.fullp;
        m.mel_wpitch = cfg->mel2048.wpitch;
        for (int s = 0; s < 4; ++s) m.mel_q4[s] = cfg->mel2048.q4[s];
        m.n_filters = a.n_filters;
        m.out = out0;
        m.out_stft = out_kind == ss::OUT_STFT;
        m.ctl = cfg->d_err;
        m.stamps = g_call_stamps2;
        {
            // the poll bound lives behind the table block in device memory; it changes only when the test aid is toggled
            const unsigned lim = ss::dbg_tile_spin_limit();
            if (lim != cfg->tile_spin_limit && cfg->d_mel2048_tab) {
                unsigned *h = static_cast<unsigned *>(static_cast<void *>(cfg->tile_spin_stage));
                SS_HIP(hipStreamSynchronize(stream));  // (debug toggle only) the staging word may still be in flight
                *h = lim;
                SS_HIP(hipMemcpyAsync(cfg->d_mel2048_tab + ss::mel2048_layout::kMelW + 32 * cfg->mel2048.wpitch, h, sizeof(unsigned),
                                      hipMemcpyHostToDevice, stream));
                cfg->tile_spin_limit = lim;
            }
        }
        if (multi) {
            // several blocks for one launch (ss_mel_spectrogram_batches_device): the twelve-wave mel build takes a batch table
            const hipError_t em = ss::launch_mel_c1024_multi(m, multi->n, multi->x, multi->out, multi->clips, stream, cfg->num_cus, &info);
            if (em == hipErrorInvalidValue) return kNoMultiBuild;
            if (em != hipSuccess) return hip_fail(em, "launch_mel_c1024_multi");
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
        const hipError_t e = ss::launch_mel_c1024(m, stream, cfg->num_cus, &info);
        if (e == hipSuccess) {
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
        // hipErrorInvalidValue before the launch: the configuration does not fit this kernel (LDS budget) -> next candidate
        if (e != hipErrorInvalidValue) return hip_fail(e, "launch_mel_c1024");
    }
    if (multi) return kNoMultiBuild;  // the other STFT-path kernels take one block per launch
    // fft_points = 512 mel spectrogram: four rows per wave (ss_mel512.hip), same layout assumptions
    if (!force_generic && (out_kind == ss::OUT_MEL || want_stft) && (cfg->mel512.ok || (want_stft && cfg->mel512.stft_only)) &&
        static_cast<unsigned long long>(a.rows + a.n_pad + 1) * a.hop < 0x7fffffffull) {
        ss::Mel512Args m{};
        m.x = d_x;
        m.ld = ld;
        m.n_samples = a.n_samples;
        m.batch = a.batch;
        m.hop = a.hop;
        m.n_pad = a.n_pad;
        m.rows = a.rows;
        m.real_rows = a.real_rows;
        m.scale = a.scale;
        m.tab = cfg->d_mel512_tab;
        m.mel_wpitch = cfg->mel512.wpitch;
        for (int s = 0; s < 5; ++s) m.mel_q4[s] = cfg->mel512.q4[s];
        m.fullp = cfg->mel512.fullp;
        m.n_filters = a.n_filters;
        m.out = out0;
        m.out_stft = out_kind == ss::OUT_STFT;
        const hipError_t e = ss::launch_mel_c256(m, stream, cfg->num_cus, &info);
        if (e == hipSuccess) {
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
        // hipErrorInvalidValue before the launch: the configuration does not fit this kernel (LDS budget) -> next candidate
        if (e != hipErrorInvalidValue) return hip_fail(e, "launch_mel_c256");
    }
    // fft_points = 1024 / 4096 mel spectrogram: two rows / one row per wave (ss_mel_c512 in ss_mfcc1024.hip, ss_mel_c2048 in
    // ss_mfcc4096.hip), same layout assumptions
    const bool use1024 = cfg->mel1024.ok || (want_stft && cfg->mel1024.stft_only), use4096 = cfg->mel4096.ok || (want_stft && cfg->mel4096.stft_only);
    if (!force_generic && (out_kind == ss::OUT_MEL || want_stft) && (use1024 || use4096) &&
        static_cast<unsigned long long>(a.rows + a.n_pad + 1) * a.hop < 0x7fffffffull) {
        ss::Mel2048Args m{};
        m.x = d_x;
        m.ld = ld;
        m.n_samples = a.n_samples;
        m.batch = a.batch;
        m.hop = a.hop;
        m.n_pad = a.n_pad;
        m.rows = a.rows;
        m.real_rows = a.real_rows;
        m.scale = a.scale;
        const bool k1024 = use1024;
        m.tab = k1024 ? cfg->d_mel1024_tab : cfg->d_mel4096_tab;
        m.mel_wpitch = k1024 ? cfg->mel1024.wpitch : cfg->mel4096.wpitch;
        for (int s = 0; s < 4; ++s) m.mel_q4[s] = k1024 ? cfg->mel1024.q4[s] : cfg->mel4096.q4[s];
        m.fullp = k1024 && cfg->mel1024.fullp;
        m.n_filters = a.n_filters;
        m.out = out0;
        m.out_stft = out_kind == ss::OUT_STFT;
        const hipError_t e = k1024 ? ss::launch_mel_c512(m, stream, cfg->num_cus, &info) : ss::launch_mel_c2048(m, stream, cfg->num_cus, &info);
        if (e == hipSuccess) {
            g_last_kernel = info.kernel_name;
            return SS_OK;
        }
        // hipErrorInvalidValue before the launch: the configuration does not fit this kernel (LDS budget) -> next candidate
        if (e != hipErrorInvalidValue) return hip_fail(e, k1024 ? "launch_mel_c512" : "launch_mel_c2048");
    }
    hipError_t e = ss::launch_front_generic(a, h.d.log2c, stream, cfg->num_cus, &info);
    if (e != hipSuccess) return hip_fail(e, "launch_front_generic");
    g_last_kernel = info.kernel_name;
    return SS_OK;
}

// stack_frames (processing.rs:65-129): frames[clip][t][i] = x[clip][t * step + i] (* window[i]); one thread per element,
// neighbouring threads on neighbouring samples of a frame.  frame_mode as in the fused kernels' loaders: contract framing,
// zero_padding = true (zeros past the signal), the literal exact_chunks copy (all-zero rows for > 2 frames, x[0 .. flen & ~1]
// in every row otherwise), librosa's centred frames.
__global__ __launch_bounds__(256) void ss_stack_frames_kernel(const float *__restrict__ x, unsigned long long ld, unsigned n_samples, unsigned flen,
                                                             unsigned step, unsigned n_frames, int frame_mode, int pad_reflect,
                                                             const float *__restrict__ window, float *__restrict__ frames, unsigned long long total)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const unsigned i = static_cast<unsigned>(g % flen);
    const unsigned long long row = g / flen;
    const unsigned t = static_cast<unsigned>(row % n_frames);
    const float *xc = x + (row / n_frames) * ld;
    float v = 0.0f;
    if (frame_mode == ss::FRAME_NORMAL) {
        v = xc[t * step + i];
    } else if (frame_mode == ss::FRAME_PADDED) {
        const unsigned idx = t * step + i;
        v = idx < n_samples ? xc[idx] : 0.0f;
    } else if (frame_mode == ss::FRAME_FIRST) {
        v = i < (flen & ~1u) ? xc[i] : 0.0f;
    } else if (frame_mode == ss::FRAME_CENTER) {
        long long pos = static_cast<long long>(t) * step + i - flen / 2;
        const long long ns = n_samples;
        bool inside = true;
        if (pos < 0 || pos >= ns) {
            if (pad_reflect) pos = pos < 0 ? -pos : 2 * (ns - 1) - pos;
            else inside = false;
        }
        v = inside ? xc[pos] : 0.0f;
    }  // FRAME_ZERO: zeros
    if (window) v *= window[i];
    frames[g] = v;
}

// The same for the common case -- contract framing, frame length, step and leading dimension multiples of four samples, 16-byte
// aligned buffers: a wave copies rows as float4s (one scalar division per row instead of two 64-bit ones per element).
__global__ __launch_bounds__(64) void ss_stack_frames_rows4(const float *__restrict__ x, unsigned long long ld, unsigned flen4, unsigned step,
                                                            unsigned n_frames, const float *__restrict__ window, float *__restrict__ frames,
                                                            unsigned long long rows, unsigned rows_per_block)
{
    const unsigned long long r0 = static_cast<unsigned long long>(blockIdx.x) * rows_per_block;
    for (unsigned k = 0; k < rows_per_block && r0 + k < rows; ++k) {
        const unsigned long long row = r0 + k;
        const unsigned long long clip = row / n_frames;
        const unsigned t = static_cast<unsigned>(row - clip * n_frames);
        const float4 *src = reinterpret_cast<const float4 *>(x + clip * ld + static_cast<unsigned long long>(t) * step);
        float4 *dst = reinterpret_cast<float4 *>(frames) + row * flen4;
        for (unsigned i = threadIdx.x; i < flen4; i += 64) {
            float4 v = src[i];
            if (window) {
                const float4 w = reinterpret_cast<const float4 *>(window)[i];
                v = make_float4(v.x * w.x, v.y * w.y, v.z * w.z, v.w * w.w);
            }
            dst[i] = v;
        }
    }
}

int check_device(const ss_config *cfg)
{
    int dev = -1;
    SS_HIP(hipGetDevice(&dev));
    if (dev != cfg->device) return ss::fail(SS_ERR_ARG, "config was created on a different HIP device than the current one");
    return SS_OK;
}

// ---- host-pointer pipeline ---------------------------------------------------------------------
// The reference's callers hand over host arrays (feature.rs:99 takes an ArrayView1, py lib.rs:167-177 a numpy array).
// `units` rows of `n` samples (row stride `ld`) are cut into chunks of a few MB; chunk k uses buffer set k % 2 on its own
// stream: H2D copy, kernel launch, D2H copy, all asynchronous, so the copy of chunk k + 1 runs while chunk k computes and
// returns (PCIe is full duplex).  Pinned caller memory (hipHostMalloc / hipHostRegister / torch pin_memory) is DMA'd directly;
// for pageable memory the runtime stages the copy and hipMemcpyAsync returns once the staging is done.
// launch(d_x, units_in_chunk, d_out0, d_out1, stream) issues the kernels for one chunk.
template <typename Launch>
int host_pipeline(const ss_config *cfg, const float *x, size_t units, size_t n, size_t ld, float *out0, size_t out0_per_unit,
                  float *out1, size_t out1_per_unit, Launch launch)
{
    ss_config::HostPipe &hp = cfg->pipe;
    std::lock_guard<std::mutex> lock(hp.mu);
#if SS_LAB
    static const size_t chunk_bytes = [] {
        const char *e = std::getenv("SS_HOST_CHUNK_MB");  // A/B knob (lab build)
        const long mb = e ? std::atol(e) : 16;
        return static_cast<size_t>(mb > 0 ? mb : 16) << 20;
    }();
#else
    constexpr size_t chunk_bytes = size_t(16) << 20;
#endif
    for (int b = 0; b < 2; ++b) {
        if (!hp.stream[b]) {
            SS_HIP(hipStreamCreateWithFlags(&hp.stream[b], hipStreamNonBlocking));
            SS_HIP(hipEventCreateWithFlags(&hp.done[b], hipEventDisableTiming));
        }
    }
    // Small calls -- a single utterance, the reference's mfcc(signal) -- are latency-bound: two copy commands and their
    // completion signals cost more than moving the bytes.  Up to SS_HOST_SMALL_KB (default 1024; measured crossover between
    // 1 and 2 MB of samples, tools/latency.py) of input and of output
    // the samples are copied into pinned, device-mapped memory by the CPU, the kernel reads them and writes its results
    // through the mapping, and the only stream operations are the launch and one synchronise.
#if SS_LAB
    static const size_t small_bytes = [] {
        const char *e = std::getenv("SS_HOST_SMALL_KB");  // A/B knob (lab build)
        const long kb = e ? std::atol(e) : 1024;
        return static_cast<size_t>(kb > 0 ? kb : 0) << 10;
    }();
#else
    constexpr size_t small_bytes = size_t(1024) << 10;
#endif
    const size_t in_all = ((units - 1) * ld + n) * sizeof(float), o0_all = units * out0_per_unit * sizeof(float),
                 o1_all = out1 ? units * out1_per_unit * sizeof(float) : 0;
    if (units > 0 && in_all <= small_bytes && o0_all <= small_bytes && o1_all <= small_bytes) {
        const size_t need[3] = {in_all, o0_all, o1_all};
        for (int i = 0; i < 3; ++i) {
            if (need[i] <= hp.cap_small[i]) continue;
            SS_HIP(hipStreamSynchronize(hp.stream[0]));
            if (hp.h_small[i]) (void)hipHostFree(hp.h_small[i]);
            hp.h_small[i] = hp.d_small[i] = nullptr;
            hp.cap_small[i] = 0;
            const size_t cap = std::max<size_t>(need[i], 64 << 10);
            SS_HIP(hipHostMalloc(&hp.h_small[i], cap, hipHostMallocMapped));
            SS_HIP(hipHostGetDevicePointer(&hp.d_small[i], hp.h_small[i], 0));
            hp.cap_small[i] = cap;
        }
        std::memcpy(hp.h_small[0], x, in_all);
        int rc = launch(static_cast<const float *>(hp.d_small[0]), units, static_cast<float *>(hp.d_small[1]), static_cast<float *>(hp.d_small[2]), hp.stream[0]);
        const hipError_t e = hipStreamSynchronize(hp.stream[0]);
        if (e != hipSuccess && rc == SS_OK) rc = hip_fail(e, "host call (mapped staging)");
        if (rc == SS_OK) rc = pending_device_error(cfg);
        if (rc == SS_OK) {
            std::memcpy(out0, hp.h_small[1], o0_all);
            if (out1) std::memcpy(out1, hp.h_small[2], o1_all);
        }
        return rc;
    }
    size_t cu = chunk_bytes / (ld * sizeof(float));
    if (cu == 0) cu = 1;
    if (cu > units) cu = units;
    const size_t in_cap = ((cu - 1) * ld + n) * sizeof(float), o0_cap = cu * out0_per_unit * sizeof(float), o1_cap = cu * out1_per_unit * sizeof(float);
    auto grow = [&](void *(&buf)[2], size_t &cap, size_t need) -> int {
        if (need <= cap) return SS_OK;
        // the capacity is void until BOTH new buffers exist: a failed hipMalloc must not leave a non-zero cap beside a
        // freed pointer (every later call with need <= cap would launch on a null buffer)
        cap = 0;
        for (int b = 0; b < 2; ++b) {
            SS_HIP(hipStreamSynchronize(hp.stream[b]));
            if (buf[b]) (void)hipFree(buf[b]);
            buf[b] = nullptr;
        }
        for (int b = 0; b < 2; ++b) {
            const hipError_t e = hipMalloc(&buf[b], need);
            if (e != hipSuccess) {
                buf[b] = nullptr;
                if (buf[0]) (void)hipFree(buf[0]);
                buf[0] = buf[1] = nullptr;
                return hip_fail(e, "hipMalloc (host pipeline buffers)");
            }
        }
        cap = need;
        return SS_OK;
    };
    int rc;
    if ((rc = grow(hp.d_in, hp.cap_in, in_cap)) || (rc = grow(hp.d_out0, hp.cap_out0, o0_cap))) return rc;
    if (out1 && (rc = grow(hp.d_out1, hp.cap_out1, o1_cap))) return rc;
    size_t k = 0;
    for (size_t u0 = 0; u0 < units; u0 += cu, ++k) {
        const int b = static_cast<int>(k & 1);
        const size_t c = std::min(cu, units - u0);
        hipStream_t st = hp.stream[b];
        // the stream orders this chunk behind the previous use of the same buffer set.  A failure only breaks out of the
        // loop: earlier chunks may still have copies in flight that touch the caller's x / out buffers, so both streams are
        // synchronised below before this function returns, whatever happened.
        hipError_t e = hipMemcpyAsync(hp.d_in[b], x + u0 * ld, ((c - 1) * ld + n) * sizeof(float), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpyAsync (H2D)"); break; }
        rc = launch(static_cast<const float *>(hp.d_in[b]), c, static_cast<float *>(hp.d_out0[b]), static_cast<float *>(hp.d_out1[b]), st);
        if (rc) break;
        e = hipMemcpyAsync(out0 + u0 * out0_per_unit, hp.d_out0[b], c * out0_per_unit * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && out1) e = hipMemcpyAsync(out1 + u0 * out1_per_unit, hp.d_out1[b], c * out1_per_unit * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpyAsync (D2H)"); break; }
    }
    for (int b = 0; b < 2; ++b) {
        const hipError_t e = hipStreamSynchronize(hp.stream[b]);
        if (e != hipSuccess && rc == SS_OK) rc = hip_fail(e, "host pipeline");
    }
    if (rc == SS_OK) rc = pending_device_error(cfg);
    return rc;
}

}  // namespace

extern "C" {

int ss_device_count(int *count)
{
    if (!count) return ss::fail(SS_ERR_ARG, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return hip_fail(e, "hipGetDeviceCount");
    }
    *count = n;
    return SS_OK;
}

int ss_set_device(int device)
{
    SS_HIP(hipSetDevice(device));
    return SS_OK;
}

int ss_config_create(const ss_params *p, ss_config **out)
{
    if (!p || !out) return ss::fail(SS_ERR_ARG, "null argument");
    *out = nullptr;
    std::unique_ptr<ss_config> cfg(new ss_config());
    int rc = ss::build_tables(*p, cfg->host);
    if (rc) return rc;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return ss::fail(SS_ERR_HIP, "no usable HIP device: the speechsauce_amd hot path has no CPU fallback");
    SS_HIP(hipGetDevice(&cfg->device));
    hipDeviceProp_t prop;
    SS_HIP(hipGetDeviceProperties(&prop, cfg->device));
    cfg->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const ss::HostTables &h = cfg->host;
    ss_config *c = cfg.get();
#define SS_UP(field, vec)                                                              \
    do {                                                                               \
        rc = upload(&c->field, (vec).data(), (vec).size() * sizeof((vec)[0]));         \
        if (rc) { ss_config_destroy(cfg.release()); return rc; }                       \
    } while (0)
    SS_UP(d_window_mfcc, h.window_mfcc);
    SS_UP(d_window_stft, h.window_stft);
    SS_UP(d_tw_c, h.tw_c);
    SS_UP(d_tw_n, h.tw_n);
    if (h.d.bluestein) {
        SS_UP(d_blu_c, h.blu_c);
        SS_UP(d_blu_b, h.blu_b);
    }
    SS_UP(d_f_start, h.bank.start);
    SS_UP(d_f_len, h.bank.len);
    SS_UP(d_f_off, h.bank.off);
    SS_UP(d_f_w, h.bank.w);
    SS_UP(d_dct, h.dct);
    ss::build_fast512(h, c->fast);
    if (c->fast.ok) {
        SS_UP(d_fast_tab, c->fast.tab);
    }
    ss::build_mfcc4096(h, c->mfcc4096);
    if (c->mfcc4096.ok) SS_UP(d_mfcc4096_tab, c->mfcc4096.tab);
    ss::build_mfcc2048(h, c->mfcc2048);
    if (c->mfcc2048.ok) SS_UP(d_mfcc2048_tab, c->mfcc2048.tab);
    ss::build_mfcc1024(h, c->mfcc1024);
    if (c->mfcc1024.ok) SS_UP(d_mfcc1024_tab, c->mfcc1024.tab);
    ss::build_mfcc256(h, c->mfcc256);
    if (c->mfcc256.ok) SS_UP(d_mfcc256_tab, c->mfcc256.tab);
    // the wide-bank build also serves what the headline kernel's specialised builds leave out at 512 points (mfe / window /
    // centred frames at other than the default frame shape)
    ss::build_mfcc512w(h, c->mfcc512w);
    if (c->mfcc512w.ok) SS_UP(d_mfcc512w_tab, c->mfcc512w.tab);
    ss::build_mel512(h, c->mel512);
    if (c->mel512.ok || c->mel512.stft_only) SS_UP(d_mel512_tab, c->mel512.tab);
    ss::build_mel1024(h, c->mel1024);
    if (c->mel1024.ok || c->mel1024.stft_only) SS_UP(d_mel1024_tab, c->mel1024.tab);
    ss::build_mel4096(h, c->mel4096);
    if (c->mel4096.ok || c->mel4096.stft_only) SS_UP(d_mel4096_tab, c->mel4096.tab);
    ss::build_mel2048(h, c->mel2048);
    if (c->mel2048.ok || c->mel2048.stft_only) SS_UP(d_mel2048_tab, c->mel2048.tab);
#undef SS_UP
    {
        // device error word (see ss_config): pinned and device-mapped, so that a kernel's store is visible to the host without
        // a copy and costs nothing unless it happens
        void *hp = nullptr, *dp = nullptr;
        hipError_t e2 = hipHostMalloc(&hp, 64, hipHostMallocMapped);
        if (e2 == hipSuccess) e2 = hipHostGetDevicePointer(&dp, hp, 0);
        if (e2 != hipSuccess) {
            if (hp) (void)hipHostFree(hp);
            ss_config_destroy(cfg.release());
            return hip_fail(e2, "hipHostMalloc (device error word)");
        }
        std::memset(hp, 0, 64);
        c->h_err = static_cast<unsigned *>(hp);
        c->d_err = static_cast<unsigned *>(dp);
    }
    *out = cfg.release();
    return SS_OK;
}

void ss_config_destroy(ss_config *cfg)
{
    if (!cfg) return;
    void *ptrs[] = {cfg->d_window_mfcc, cfg->d_window_stft, cfg->d_tw_c, cfg->d_tw_n, cfg->d_blu_c, cfg->d_blu_b, cfg->d_f_start,
                    cfg->d_f_len,       cfg->d_f_off,       cfg->d_f_w,  cfg->d_dct,
                    cfg->d_fast_tab,    cfg->d_mel2048_tab, cfg->d_mfcc4096_tab, cfg->d_mfcc2048_tab, cfg->d_mfcc1024_tab, cfg->d_mfcc256_tab, cfg->d_mfcc512w_tab, cfg->d_mel512_tab, cfg->d_mel1024_tab, cfg->d_mel4096_tab};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (int b = 0; b < 2; ++b) {
        if (cfg->pipe.stream[b]) (void)hipStreamSynchronize(cfg->pipe.stream[b]);
        for (void *p : {cfg->pipe.d_in[b], cfg->pipe.d_out0[b], cfg->pipe.d_out1[b]})
            if (p) (void)hipFree(p);
        if (cfg->pipe.done[b]) (void)hipEventDestroy(cfg->pipe.done[b]);
        if (cfg->pipe.stream[b]) (void)hipStreamDestroy(cfg->pipe.stream[b]);
    }
    for (void *p : cfg->pipe.h_small)
        if (p) (void)hipHostFree(p);
    if (cfg->h_err) (void)hipHostFree(cfg->h_err);
    delete cfg;
}

int ss_config_params(const ss_config *cfg, ss_params *out)
{
    if (!cfg || !out) return ss::fail(SS_ERR_ARG, "null argument");
    *out = cfg->host.params;
    return SS_OK;
}

int ss_config_device_status(const ss_config *cfg)
{
    if (!cfg) return ss::fail(SS_ERR_ARG, "null config");
    return pending_device_error(cfg);
}

// ---- device-pointer variants ---------------------------------------------------------------

int ss_mfcc_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld,
                         float *d_out, void *stream)
{
    return launch_frames(cfg, ss::OUT_MFCC, d_x, batch, n_samples, ld, d_out, nullptr, static_cast<hipStream_t>(stream));
}

// Several independent batches of equal-length clips per call.  Where the configuration's kernel takes a batch table (the 512-point
// MFCC kernel's default build: ss_mfcc512.hip, MULTI) up to ss::kMaxLaunchBatches batches share ONE launch -- the persistent
// workgroups' work range is the concatenation of the batches' frame quads, so a launch's start-up and its one-unit tail are paid
// once per call instead of once per batch; every other configuration is served batch by batch on the same stream.  Either way the
// results are those of n_batches separate ss_mfcc_batch_device calls, bit for bit.
int ss_mfcc_batches_device(const ss_config *cfg, size_t n_batches, const float *const *d_x, const size_t *batch, size_t n_samples,
                           size_t ld, float *const *d_out, void *stream)
{
    if (!cfg) return ss::fail(SS_ERR_ARG, "null config");
    if (n_batches == 0) return SS_OK;
    if (!d_x || !batch || !d_out) return ss::fail(SS_ERR_ARG, "null batch table");
    if (ld < n_samples) return ss::fail(SS_ERR_ARG, "leading dimension smaller than n_samples");
    {
        size_t T = 0;  // the clip shape is checked before anything is launched (every batch has the same one)
        const int rc = ss::num_frames(cfg->host.params, n_samples, T);
        if (rc) return rc;
    }
    std::vector<const float *> xs;
    std::vector<float *> outs;
    std::vector<size_t> clips;
    try {  // (nothing unwinds across the boundary: a failed allocation of the three small tables is an error code)
        xs.reserve(n_batches);
        outs.reserve(n_batches);
        clips.reserve(n_batches);
    } catch (const std::exception &) {
        return ss::fail(SS_ERR_ARG, "n_batches too large for this host's memory");
    }
    for (size_t b = 0; b < n_batches; ++b) {
        if (batch[b] == 0) continue;  // an empty batch has no buffers
        if (!d_x[b] || !d_out[b]) return ss::fail(SS_ERR_ARG, "null buffer in batch " + std::to_string(b));
        if (batch[b] > 0x7fffffffull) return ss::fail(SS_ERR_ARG, "batch too large");
        xs.push_back(d_x[b]);
        outs.push_back(d_out[b]);
        clips.push_back(batch[b]);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (size_t g0 = 0; g0 < xs.size(); g0 += ss::kMaxLaunchBatches) {
        const size_t gn = std::min<size_t>(ss::kMaxLaunchBatches, xs.size() - g0);
        int rc = kNoMultiBuild;
        if (gn > 1) {
            const MultiBatches mb{static_cast<int>(gn), xs.data() + g0, outs.data() + g0, clips.data() + g0};
            rc = launch_frames(cfg, ss::OUT_MFCC, xs[g0], clips[g0], n_samples, ld, outs[g0], nullptr, st, false, &mb);
        }
        if (rc == kNoMultiBuild) {
            rc = SS_OK;
            for (size_t b = g0; b < g0 + gn && rc == SS_OK; ++b)
                rc = launch_frames(cfg, ss::OUT_MFCC, xs[b], clips[b], n_samples, ld, outs[b], nullptr, st);
        }
        if (rc) return rc;
    }
    return SS_OK;
}

int ss_mfe_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld,
                        float *d_feat, float *d_energy, void *stream)
{
    if (!d_energy) return ss::fail(SS_ERR_ARG, "null buffer");
    return launch_frames(cfg, ss::OUT_MFE, d_x, batch, n_samples, ld, d_feat, d_energy, static_cast<hipStream_t>(stream));
}

// lmfe (feature.rs:242-245): ln of mfe's zero-handled filterbank energies.  The frame energies mfe also returns are
// dropped, as in the reference; a caller that has room for them passes d_energy, otherwise a stream-ordered temporary is used.
int ss_lmfe_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld,
                         float *d_feat, float *d_energy, void *stream)
{
    if (!cfg) return ss::fail(SS_ERR_ARG, "null config");
    if (batch == 0) return SS_OK;
    size_t T = 0;
    int rc = ss::num_frames(cfg->host.params, n_samples, T);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *tmp = nullptr;
    if (!d_energy) {
        SS_HIP(hipMallocAsync(reinterpret_cast<void **>(&tmp), batch * T * sizeof(float), st));
        d_energy = tmp;
    }
    rc = launch_frames(cfg, ss::OUT_MFE, d_x, batch, n_samples, ld, d_feat, d_energy, st);
    if (rc == SS_OK) rc = ss_ln_device(d_feat, batch * T * cfg->host.params.num_filters, stream);
    if (tmp) {
        const hipError_t e = hipFreeAsync(tmp, st);
        if (e != hipSuccess && rc == SS_OK) rc = hip_fail(e, "hipFreeAsync");
    }
    return rc;
}

int ss_lmfe_batch(const ss_config *cfg, const float *x, size_t batch, size_t n_samples, size_t ld, float *feat)
{
    if (!cfg || !x || !feat) return ss::fail(SS_ERR_ARG, "null argument");
    if (ld < n_samples) return ss::fail(SS_ERR_ARG, "leading dimension smaller than n_samples");
    size_t T = 0;
    int rc = ss::num_frames(cfg->host.params, n_samples, T);
    if (rc) return rc;
    if (batch == 0) return SS_OK;
    rc = check_device(cfg);
    if (rc) return rc;
    return host_pipeline(cfg, x, batch, n_samples, ld, feat, T * cfg->host.params.num_filters, nullptr, 0,
                         [&](const float *d_x, size_t c, float *d_o0, float *, hipStream_t st) {
                             return ss_lmfe_batch_device(cfg, d_x, c, n_samples, ld, d_o0, nullptr, st);
                         });
}

int ss_lmfe(const ss_config *cfg, const float *x, size_t n_samples, float *feat)
{
    return ss_lmfe_batch(cfg, x, 1, n_samples, n_samples, feat);
}

int ss_shard_bounds(size_t n_items, int world, int rank, size_t *lo, size_t *hi)
{
    if (!lo || !hi || world <= 0 || rank < 0 || rank >= world) return ss::fail(SS_ERR_ARG, "bad world / rank");
    // contiguous block partition: ranks [0, n % world) get one extra item (speechsauce_amd.distributed.shard_bounds)
    const size_t w = static_cast<size_t>(world), r = static_cast<size_t>(rank), base = n_items / w, extra = n_items % w;
    *lo = r * base + std::min(r, extra);
    *hi = *lo + base + (r < extra ? 1 : 0);
    return SS_OK;
}

}  // extern "C" (reopened below, after the RCCL resolver)

// ---- RCCL, resolved at run time (nothing is linked at build time) ---------------------------------------------------------
// The communicator is the caller's, so the functions must come from the RCCL copy that created it.  Order: a library named
// with ss_rccl_library; a copy that is ALREADY mapped into the process (dlsym(RTLD_DEFAULT), then RTLD_NOLOAD on the usual
// sonames -- a Python process that imported torch has torch/lib/librccl.so mapped, and torch.distributed's communicators
// come from it); only then a fresh dlopen of librccl.so.1 / librccl.so.
namespace {

struct Rccl {
    using all_gather_fn = int (*)(const void *, void *, size_t, int, void *, hipStream_t);
    using send_fn = int (*)(const void *, size_t, int, int, void *, hipStream_t);
    using recv_fn = int (*)(void *, size_t, int, int, void *, hipStream_t);
    using group_fn = int (*)();
    all_gather_fn all_gather = nullptr;
    send_fn send = nullptr;
    recv_fn recv = nullptr;
    group_fn group_start = nullptr, group_end = nullptr;
    std::string origin;
    bool ok() const { return all_gather && send && recv && group_start && group_end; }
};

std::mutex g_rccl_mu;
std::string g_rccl_path;       // ss_rccl_library
std::unique_ptr<Rccl> g_rccl;  // set once a resolution has succeeded
std::string g_rccl_error;      // why the last attempt failed
bool g_rccl_absent = false;    // the auto-discovery found nothing: not repeated per call (ss_rccl_library resets it)

bool rccl_from(void *handle, const char *origin, Rccl &r)
{
    r.all_gather = reinterpret_cast<Rccl::all_gather_fn>(dlsym(handle, "ncclAllGather"));
    r.send = reinterpret_cast<Rccl::send_fn>(dlsym(handle, "ncclSend"));
    r.recv = reinterpret_cast<Rccl::recv_fn>(dlsym(handle, "ncclRecv"));
    r.group_start = reinterpret_cast<Rccl::group_fn>(dlsym(handle, "ncclGroupStart"));
    r.group_end = reinterpret_cast<Rccl::group_fn>(dlsym(handle, "ncclGroupEnd"));
    r.origin = origin;
    return r.ok();
}

// (why: the reason of a failed resolution, copied under the lock)
const Rccl *rccl(std::string *why = nullptr)
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    struct Report {
        std::string *out;
        ~Report() { if (out) *out = g_rccl_error; }
    } report{why};
    if (g_rccl) {  // only a successful resolution is kept
        g_rccl_error.clear();  // (a stale reason of an earlier, corrected attempt must not reach `why` on success)
        return g_rccl.get();
    }
    std::unique_ptr<Rccl> r(new Rccl());
    auto keep = [&]() {
        g_rccl = std::move(r);
        g_rccl_error.clear();
        return g_rccl.get();
    };
    // a handle THIS function opened and that did not resolve is closed again (one reference would leak per call)
    auto from_opened = [&](void *h, const char *origin) {
        if (!h) return false;
        if (rccl_from(h, origin, *r)) return true;
        (void)dlclose(h);
        return false;
    };
    if (!g_rccl_path.empty()) {
        void *h = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);
        const char *de = h ? nullptr : dlerror();
        if (from_opened(h, g_rccl_path.c_str())) return keep();
        // an explicit path that does not load is an error, not a reason to guess; nothing is cached, so a corrected
        // ss_rccl_library call is accepted and tried again
        g_rccl_error = g_rccl_path + (h ? ": ncclAllGather / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd not all found" : std::string(": ") + (de ? de : "dlopen failed"));
        return nullptr;
    }
    g_rccl_error = "no mapped copy, librccl.so.1 / librccl.so not loadable";
    if (g_rccl_absent) return nullptr;  // looked already: the lookups, the /proc scan and four dlopen attempts are not repeated per call
    if (rccl_from(RTLD_DEFAULT, "already in the global symbol scope", *r)) return keep();
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        void *h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (from_opened(h, "already mapped (RTLD_NOLOAD)")) return keep();
    }
    {
        // a copy mapped under another name (torch/lib/librccl.so is loaded by path): look through the process's objects
        struct Ctx { std::string path; } ctx;
        if (FILE *fp = std::fopen("/proc/self/maps", "r")) {
            char line[1024];
            while (std::fgets(line, sizeof line, fp)) {
                const char *sl = std::strchr(line, '/');
                if (sl && std::strstr(sl, "librccl.so")) {
                    ctx.path.assign(sl, std::strcspn(sl, "\n"));
                    break;
                }
            }
            std::fclose(fp);
        }
        if (!ctx.path.empty()) {
            void *h = dlopen(ctx.path.c_str(), RTLD_NOW | RTLD_NOLOAD);
            if (from_opened(h, ctx.path.c_str())) return keep();
        }
    }
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (from_opened(h, name)) return keep();
    }
    g_rccl_absent = true;  // (a later ss_rccl_library call, e.g. after the caller has loaded RCCL, asks again)
    return nullptr;
}

int rccl_fail(const char *what, int rc)
{
    return ss::fail(SS_ERR_HIP, std::string(what) + " failed with ncclResult_t " + std::to_string(rc));
}

constexpr int kNcclFloat32 = 7;  // rccl.h: ncclFloat32

}  // namespace

extern "C" {

int ss_rccl_library(const char *path)
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl) return ss::fail(SS_ERR_ARG, "RCCL was already resolved (" + g_rccl->origin + "): ss_rccl_library must precede the first collective");
    g_rccl_path = path ? path : "";
    g_rccl_absent = false;
    return SS_OK;
}

int ss_all_gather_features(void *nccl_comm, const float *d_block, size_t elems_per_rank, float *d_out, void *stream)
{
    if (!nccl_comm || !d_block || !d_out) return ss::fail(SS_ERR_ARG, "null argument");
    if (elems_per_rank == 0) return SS_OK;
    std::string why;
    const Rccl *r = rccl(&why);
    if (!r) return ss::fail(SS_ERR_UNSUPPORTED, "RCCL could not be resolved (" + why + ")");
    // ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream)
    const int rc = r->all_gather(d_block, d_out, elems_per_rank, kNcclFloat32, nccl_comm, static_cast<hipStream_t>(stream));
    if (rc != 0) return rccl_fail("ncclAllGather", rc);
    return SS_OK;
}

int ss_gather_features(void *nccl_comm, const float *d_block, size_t elems_per_rank, float *d_out, int root, int rank, int world,
                       void *stream)
{
    if (!nccl_comm || !d_block) return ss::fail(SS_ERR_ARG, "null argument");
    if (world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world) return ss::fail(SS_ERR_ARG, "bad world / rank / root");
    if (rank == root && !d_out) return ss::fail(SS_ERR_ARG, "the root needs an output buffer");
    if (elems_per_rank == 0) return SS_OK;
    std::string why;
    const Rccl *r = rccl(&why);
    if (!r) return ss::fail(SS_ERR_UNSUPPORTED, "RCCL could not be resolved (" + why + ")");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (rank != root) {
        const int rc = r->send(d_block, elems_per_rank, kNcclFloat32, root, nccl_comm, st);
        return rc ? rccl_fail("ncclSend", rc) : SS_OK;
    }
    // root: its own block is a device copy; one receive per peer inside a group, so that all of them progress together
    // (every peer has a direct xGMI link to the root)
    SS_HIP(hipMemcpyAsync(d_out + static_cast<size_t>(root) * elems_per_rank, d_block, elems_per_rank * sizeof(float), hipMemcpyDeviceToDevice, st));
    int rc = r->group_start();
    if (rc) return rccl_fail("ncclGroupStart", rc);
    int first_err = 0;
    for (int p = 0; p < world; ++p) {
        if (p == root) continue;
        const int e = r->recv(d_out + static_cast<size_t>(p) * elems_per_rank, elems_per_rank, kNcclFloat32, p, nccl_comm, st);
        if (e && !first_err) first_err = e;
    }
    rc = r->group_end();  // always closed, also after a failed ncclRecv
    if (first_err) return rccl_fail("ncclRecv", first_err);
    if (rc) return rccl_fail("ncclGroupEnd", rc);
    return SS_OK;
}

int ss_power_spectrum_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples,
                                   size_t ld, float *d_P, void *stream)
{
    return launch_frames(cfg, ss::OUT_POWER, d_x, batch, n_samples, ld, d_P, nullptr, static_cast<hipStream_t>(stream));
}

int ss_power_spectrum_frames_device(const ss_config *cfg, const float *d_frames, size_t rows, size_t cols, size_t ld, float *d_P,
                                    void *stream)
{
    return launch_frames(cfg, ss::OUT_POWER, d_frames, rows, cols, ld, d_P, nullptr, static_cast<hipStream_t>(stream), true);
}

// Host-pointer forms of the stage outputs the reference exposes as pub fns (processing.rs:179-181, functions.rs:86-123,
// :199-233, processing.rs:65-129): same chunked two-stream pipeline / mapped small-call staging as ss_mfcc_batch.
int ss_power_spectrum_batch(const ss_config *cfg, const float *x, size_t batch, size_t n_samples, size_t ld, float *P)
{
    if (!cfg || !x || !P) return ss::fail(SS_ERR_ARG, "null argument");
    if (ld < n_samples) return ss::fail(SS_ERR_ARG, "leading dimension smaller than n_samples");
    size_t T = 0;
    int rc = ss::num_frames(cfg->host.params, n_samples, T);
    if (rc) return rc;
    if (batch == 0) return SS_OK;
    rc = check_device(cfg);
    if (rc) return rc;
    return host_pipeline(cfg, x, batch, n_samples, ld, P, T * (cfg->host.params.fft_points / 2 + 1), nullptr, 0,
                         [&](const float *d_x, size_t c, float *d_o0, float *, hipStream_t st) {
                             return ss_power_spectrum_batch_device(cfg, d_x, c, n_samples, ld, d_o0, st);
                         });
}

int ss_power_spectrum(const ss_config *cfg, const float *x, size_t n_samples, float *P)
{
    return ss_power_spectrum_batch(cfg, x, 1, n_samples, n_samples, P);
}

int ss_power_spectrum_frames(const ss_config *cfg, const float *frames, size_t rows, size_t cols, float *P)
{
    if (!cfg || !frames || !P) return ss::fail(SS_ERR_ARG, "null argument");
    if (cols == 0 || cols > cfg->host.params.fft_points) return ss::fail(SS_ERR_ARG, "frame length must be in [1, fft_points]");
    if (rows == 0) return SS_OK;
    const int rc = check_device(cfg);
    if (rc) return rc;
    return host_pipeline(cfg, frames, rows, cols, cols, P, cfg->host.params.fft_points / 2 + 1, nullptr, 0,
                         [&](const float *d_x, size_t c, float *d_o0, float *, hipStream_t st) {
                             return ss_power_spectrum_frames_device(cfg, d_x, c, cols, cols, d_o0, st);
                         });
}

}  // extern "C"

namespace {

// The framing kernels for `batch` clips of n_samples (row stride ld): frames [batch x T x flen]; d_window: flen floats or null.
int launch_stack_frames(const float *d_x, size_t batch, size_t n_samples, size_t ld, uint32_t flen, uint32_t step, size_t T, int mode,
                        int pad_reflect, const float *d_window, float *d_frames, hipStream_t stream)
{
    if (n_samples > 0x7fffffffull || static_cast<unsigned long long>(T) * step + flen > 0xffffffffull) return ss::fail(SS_ERR_ARG, "clip too long");
    const unsigned long long total = static_cast<unsigned long long>(batch) * T * flen;
    const bool aligned4 = mode == ss::FRAME_NORMAL && flen % 4 == 0 && step % 4 == 0 && ld % 4 == 0 &&
                          reinterpret_cast<uintptr_t>(d_x) % 16 == 0 && reinterpret_cast<uintptr_t>(d_frames) % 16 == 0 &&
                          (!d_window || reinterpret_cast<uintptr_t>(d_window) % 16 == 0);
    if (aligned4) {
        const unsigned long long rows = static_cast<unsigned long long>(batch) * T;
        const unsigned rpb = 4;
        const unsigned long long nb = (rows + rpb - 1) / rpb;
        if (nb > 0x7fffffffull) return ss::fail(SS_ERR_ARG, "batch too large");
        hipLaunchKernelGGL(ss_stack_frames_rows4, dim3(static_cast<unsigned>(nb)), dim3(64), 0, stream, d_x, static_cast<unsigned long long>(ld),
                           flen / 4, step, static_cast<unsigned>(T), d_window, d_frames, rows, rpb);
        const hipError_t e4 = hipGetLastError();
        if (e4 != hipSuccess) return hip_fail(e4, "ss_stack_frames_rows4");
        g_last_kernel = "ss_stack_frames_rows4";
        return SS_OK;
    }
    const unsigned long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffull) return ss::fail(SS_ERR_ARG, "batch too large");
    hipLaunchKernelGGL(ss_stack_frames_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, d_x, static_cast<unsigned long long>(ld),
                       static_cast<unsigned>(n_samples), flen, step, static_cast<unsigned>(T), mode, pad_reflect, d_window, d_frames, total);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "ss_stack_frames_kernel");
    g_last_kernel = "ss_stack_frames_kernel";
    return SS_OK;
}

// processing.rs:77-78, :91-92, :101: frame sizes and count from the reference's own loose arguments (no SpeechConfig, no FFT length)
int frames_shape(size_t n_samples, uint32_t sample_rate, float frame_length, float frame_stride, int zero_padding, size_t &T, uint32_t &flen,
                 uint32_t &step)
{
    ss_params p{};
    int rc = ss_params_default(&p, sample_rate ? sample_rate : 1u);
    if (rc) return rc;
    if (sample_rate == 0) return ss::fail(SS_ERR_BAD_CONFIG, "sample_rate must be > 0");
    p.frame_length = frame_length;
    p.frame_stride = frame_stride;
    p.framing = zero_padding ? SS_FRAMING_PADDED : SS_FRAMING_CONTRACT;
    ss::Derived d;
    rc = ss::derive(p, d);
    if (rc) return rc;
    rc = ss::num_frames(p, n_samples, T);
    if (rc) return rc;
    flen = d.flen;
    step = d.step;
    return SS_OK;
}

}  // namespace

extern "C" {

int ss_stack_frames_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld, float *d_frames,
                           void *stream)
{
    if (!cfg) return ss::fail(SS_ERR_ARG, "null config");
    if (ld < n_samples) return ss::fail(SS_ERR_ARG, "leading dimension smaller than n_samples");
    const ss::HostTables &h = cfg->host;
    size_t T = 0;
    int rc = ss::num_frames(h.params, n_samples, T);
    if (rc) return rc;
    if (batch == 0) return SS_OK;
    if (!d_x || !d_frames) return ss::fail(SS_ERR_ARG, "null buffer");
    rc = check_device(cfg);
    if (rc) return rc;
    int mode = ss::FRAME_NORMAL;
    if (h.params.framing == SS_FRAMING_LITERAL) mode = T > 2 ? ss::FRAME_ZERO : ss::FRAME_FIRST;
    else if (h.params.framing == SS_FRAMING_CENTER) mode = ss::FRAME_CENTER;
    else if (h.params.framing == SS_FRAMING_PADDED) mode = ss::FRAME_PADDED;
    return launch_stack_frames(d_x, batch, n_samples, ld, h.d.flen, h.d.step, T, mode, h.params.pad_mode == SS_PAD_REFLECT ? 1 : 0,
                               cfg->d_window_mfcc, d_frames, static_cast<hipStream_t>(stream));
}

// ---- stack_frames with the reference's own argument list (processing.rs:65-76): no SpeechConfig, no FFT length ----

int ss_stack_frames_shape(size_t n_samples, uint32_t sample_rate, float frame_length, float frame_stride, int zero_padding,
                          size_t *num_frames, size_t *frame_len)
{
    if (!num_frames || !frame_len) return ss::fail(SS_ERR_ARG, "null argument");
    size_t T = 0;
    uint32_t flen = 0, step = 0;
    const int rc = frames_shape(n_samples, sample_rate, frame_length, frame_stride, zero_padding, T, flen, step);
    if (rc) return rc;
    *num_frames = T;
    *frame_len = flen;
    return SS_OK;
}

int ss_stack_frames_signal_device(const float *d_x, size_t n_samples, uint32_t sample_rate, float frame_length, float frame_stride,
                                  const float *d_window, int zero_padding, float *d_frames, void *stream)
{
    size_t T = 0;
    uint32_t flen = 0, step = 0;
    const int rc = frames_shape(n_samples, sample_rate, frame_length, frame_stride, zero_padding, T, flen, step);
    if (rc) return rc;
    if (!d_x || !d_frames) return ss::fail(SS_ERR_ARG, "null buffer");
    return launch_stack_frames(d_x, 1, n_samples, n_samples, flen, step, T, zero_padding ? ss::FRAME_PADDED : ss::FRAME_NORMAL, 0, d_window,
                               d_frames, static_cast<hipStream_t>(stream));
}

int ss_stack_frames_signal(const float *x, size_t n_samples, uint32_t sample_rate, float frame_length, float frame_stride,
                           const float *window, int zero_padding, float *frames)
{
    size_t T = 0;
    uint32_t flen = 0, step = 0;
    int rc = frames_shape(n_samples, sample_rate, frame_length, frame_stride, zero_padding, T, flen, step);
    if (rc) return rc;
    if (!x || !frames) return ss::fail(SS_ERR_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return ss::fail(SS_ERR_HIP, "no usable HIP device: the speechsauce_amd hot path has no CPU fallback");
    // a cold path (no config, so no staging pipeline to reuse): plain allocations and synchronous copies
    DeviceBuf dx, dw, df;
    const size_t out_bytes = T * static_cast<size_t>(flen) * sizeof(float);
    if ((rc = dx.alloc(n_samples * sizeof(float))) || (rc = df.alloc(out_bytes))) return rc;
    if (window && (rc = dw.alloc(flen * sizeof(float)))) return rc;
    SS_HIP(hipMemcpy(dx.p, x, n_samples * sizeof(float), hipMemcpyHostToDevice));
    if (window) SS_HIP(hipMemcpy(dw.p, window, flen * sizeof(float), hipMemcpyHostToDevice));
    rc = ss_stack_frames_signal_device(dx.as<float>(), n_samples, sample_rate, frame_length, frame_stride, window ? dw.as<float>() : nullptr,
                                       zero_padding, df.as<float>(), nullptr);
    if (rc) return rc;
    SS_HIP(hipMemcpy(frames, df.p, out_bytes, hipMemcpyDeviceToHost));
    return SS_OK;
}

int ss_stack_frames(const ss_config *cfg, const float *x, size_t n_samples, float *frames)
{
    if (!cfg || !x || !frames) return ss::fail(SS_ERR_ARG, "null argument");
    size_t T = 0;
    int rc = ss::num_frames(cfg->host.params, n_samples, T);
    if (rc) return rc;
    rc = check_device(cfg);
    if (rc) return rc;
    return host_pipeline(cfg, x, 1, n_samples, n_samples, frames, T * cfg->host.d.flen, nullptr, 0,
                         [&](const float *d_x, size_t c, float *d_o0, float *, hipStream_t st) {
                             return ss_stack_frames_device(cfg, d_x, c, n_samples, n_samples, d_o0, st);
                         });
}

int ss_stft(const ss_config *cfg, const float *x, size_t channels, size_t n_samples, float *out)
{
    if (!cfg || !x || !out) return ss::fail(SS_ERR_ARG, "null argument");
    size_t R = 0, Rreal = 0;
    int rc = ss::stft_rows(cfg->host.params, n_samples, R, Rreal);
    if (rc) return rc;
    if (channels == 0) return SS_OK;
    if (n_samples == 0) return ss::fail(SS_ERR_ARG, "empty signal");
    rc = check_device(cfg);
    if (rc) return rc;
    return host_pipeline(cfg, x, channels, n_samples, n_samples, out, R * (cfg->host.params.fft_points / 2 + 1) * 2, nullptr, 0,
                         [&](const float *d_x, size_t c, float *d_o0, float *, hipStream_t st) {
                             return ss_stft_device(cfg, d_x, c, n_samples, n_samples, d_o0, st);
                         });
}

int ss_mel_spectrogram_device(const ss_config *cfg, const float *d_x, size_t channels, size_t n_samples,
                              size_t ld, float *d_out, void *stream)
{
    return launch_stft(cfg, ss::OUT_MEL, d_x, channels, n_samples, ld, d_out, static_cast<hipStream_t>(stream));
}

// The mel-spectrogram form of ss_mfcc_batches_device: n_batches independent [channels[b] x n_samples] blocks, each with its own
// output block.  Where the configuration runs on the twelve-wave 2048-point mel build and every block is large enough to select
// that build on its own, up to ss::kMaxLaunchBatches blocks share ONE launch; otherwise block by block on `stream`.  The results
// are those of separate ss_mel_spectrogram_device calls, bit for bit.
int ss_mel_spectrogram_batches_device(const ss_config *cfg, size_t n_batches, const float *const *d_x, const size_t *channels,
                                      size_t n_samples, size_t ld, float *const *d_out, void *stream)
{
    if (!cfg) return ss::fail(SS_ERR_ARG, "null config");
    if (n_batches == 0) return SS_OK;
    if (!d_x || !channels || !d_out) return ss::fail(SS_ERR_ARG, "null batch table");
    if (ld < n_samples) return ss::fail(SS_ERR_ARG, "leading dimension smaller than n_samples");
    std::vector<const float *> xs;
    std::vector<float *> outs;
    std::vector<size_t> chans;
    try {
        xs.reserve(n_batches);
        outs.reserve(n_batches);
        chans.reserve(n_batches);
    } catch (const std::exception &) {
        return ss::fail(SS_ERR_ARG, "n_batches too large for this host's memory");
    }
    for (size_t b = 0; b < n_batches; ++b) {
        if (channels[b] == 0) continue;  // an empty block has no buffers
        if (!d_x[b] || !d_out[b]) return ss::fail(SS_ERR_ARG, "null buffer in batch " + std::to_string(b));
        if (channels[b] > 0x7fffffffull) return ss::fail(SS_ERR_ARG, "bad clip length / channel count");
        xs.push_back(d_x[b]);
        outs.push_back(d_out[b]);
        chans.push_back(channels[b]);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (size_t g0 = 0; g0 < xs.size(); g0 += ss::kMaxLaunchBatches) {
        const size_t gn = std::min<size_t>(ss::kMaxLaunchBatches, xs.size() - g0);
        int rc = kNoMultiBuild;
        if (gn > 1) {
            const MultiBatches mb{static_cast<int>(gn), xs.data() + g0, outs.data() + g0, chans.data() + g0};
            rc = launch_stft(cfg, ss::OUT_MEL, xs[g0], chans[g0], n_samples, ld, outs[g0], st, &mb);
        }
        if (rc == kNoMultiBuild) {
            rc = SS_OK;
            for (size_t b = g0; b < g0 + gn && rc == SS_OK; ++b) rc = launch_stft(cfg, ss::OUT_MEL, xs[b], chans[b], n_samples, ld, outs[b], st);
        }
        if (rc) return rc;
    }
    return SS_OK;
}

int ss_stft_device(const ss_config *cfg, const float *d_x, size_t channels, size_t n_samples, size_t ld,
                   float *d_out, void *stream)
{
    return launch_stft(cfg, ss::OUT_STFT, d_x, channels, n_samples, ld, d_out, static_cast<hipStream_t>(stream));
}

int ss_preemphasis_device(const float *d_x, size_t n_samples, long shift, float cof, float *d_y, void *stream)
{
    if (!d_x || !d_y) return ss::fail(SS_ERR_ARG, "null buffer");
    // slices s![..shift] / s![-shift..] panic for shift <= 0 or shift > len (processing.rs:43-50)
    if (n_samples == 0 || shift <= 0 || static_cast<size_t>(shift) > n_samples) return ss::fail(SS_ERR_ARG, "shift must be in [1, n_samples]");
    hipError_t e = ss::launch_preemphasis(d_x, d_y, n_samples, static_cast<size_t>(shift) % n_samples, cof,
                                          static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "launch_preemphasis");
    g_last_kernel = "ss_preemphasis_kernel";
    return SS_OK;
}

// ---- host-pointer variants -------------------------------------------------------------------

int ss_mfcc_batch(const ss_config *cfg, const float *x, size_t batch, size_t n_samples, size_t ld, float *out)
{
    if (!cfg || !x || !out) return ss::fail(SS_ERR_ARG, "null argument");
    if (ld < n_samples) return ss::fail(SS_ERR_ARG, "leading dimension smaller than n_samples");
    size_t T = 0;
    int rc = ss::num_frames(cfg->host.params, n_samples, T);
    if (rc) return rc;
    if (batch == 0) return SS_OK;
    rc = check_device(cfg);
    if (rc) return rc;
    return host_pipeline(cfg, x, batch, n_samples, ld, out, T * cfg->host.params.num_cepstral, nullptr, 0,
                         [&](const float *d_x, size_t c, float *d_o0, float *, hipStream_t st) {
                             return ss_mfcc_batch_device(cfg, d_x, c, n_samples, ld, d_o0, st);
                         });
}

int ss_mfcc(const ss_config *cfg, const float *x, size_t n_samples, float *out)
{
    return ss_mfcc_batch(cfg, x, 1, n_samples, n_samples, out);
}

int ss_mfe_batch(const ss_config *cfg, const float *x, size_t batch, size_t n_samples, size_t ld, float *feat, float *energy)
{
    if (!cfg || !x || !feat || !energy) return ss::fail(SS_ERR_ARG, "null argument");
    if (ld < n_samples) return ss::fail(SS_ERR_ARG, "leading dimension smaller than n_samples");
    size_t T = 0;
    int rc = ss::num_frames(cfg->host.params, n_samples, T);
    if (rc) return rc;
    if (batch == 0) return SS_OK;
    rc = check_device(cfg);
    if (rc) return rc;
    return host_pipeline(cfg, x, batch, n_samples, ld, feat, T * cfg->host.params.num_filters, energy, T,
                         [&](const float *d_x, size_t c, float *d_o0, float *d_o1, hipStream_t st) {
                             return ss_mfe_batch_device(cfg, d_x, c, n_samples, ld, d_o0, d_o1, st);
                         });
}

int ss_mfe(const ss_config *cfg, const float *x, size_t n_samples, float *feat, float *energy)
{
    return ss_mfe_batch(cfg, x, 1, n_samples, n_samples, feat, energy);
}

int ss_mel_spectrogram(const ss_config *cfg, const float *x, size_t channels, size_t n_samples, float *out)
{
    if (!cfg || !x || !out) return ss::fail(SS_ERR_ARG, "null argument");
    size_t R = 0, Rreal = 0;
    int rc = ss::stft_rows(cfg->host.params, n_samples, R, Rreal);
    if (rc) return rc;
    if (channels == 0) return SS_OK;
    if (n_samples == 0) return ss::fail(SS_ERR_ARG, "empty signal");
    rc = check_device(cfg);
    if (rc) return rc;
    return host_pipeline(cfg, x, channels, n_samples, n_samples, out, cfg->host.params.num_filters * R, nullptr, 0,
                         [&](const float *d_x, size_t c, float *d_o0, float *, hipStream_t st) {
                             return ss_mel_spectrogram_device(cfg, d_x, c, n_samples, n_samples, d_o0, st);
                         });
}

int ss_preemphasis(const float *x, size_t n_samples, long shift, float cof, float *y)
{
    if (!x || !y) return ss::fail(SS_ERR_ARG, "null argument");
    if (n_samples == 0 || shift <= 0 || static_cast<size_t>(shift) > n_samples) return ss::fail(SS_ERR_ARG, "shift must be in [1, n_samples]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return ss::fail(SS_ERR_HIP, "no usable HIP device: the speechsauce_amd hot path has no CPU fallback");
    DeviceBuf dx, dy;
    int rc;
    if ((rc = dx.alloc(n_samples * sizeof(float))) || (rc = dy.alloc(n_samples * sizeof(float)))) return rc;
    SS_HIP(hipMemcpy(dx.p, x, n_samples * sizeof(float), hipMemcpyHostToDevice));
    rc = ss_preemphasis_device(dx.as<float>(), n_samples, shift, cof, dy.as<float>(), nullptr);
    if (rc) return rc;
    SS_HIP(hipMemcpy(y, dy.p, n_samples * sizeof(float), hipMemcpyDeviceToHost));
    return SS_OK;
}

// ---- diagnostics ---------------------------------------------------------------------------

const char *ss_last_kernel_name(void) { return g_last_kernel; }

// ---- test aids (include/speechsauce_amd_debug.h): exported by the LAB library only ----
#if SS_LAB
int ss_debug_force_generic(int on)
{
    g_force_generic.store(on ? 1 : 0, std::memory_order_relaxed);
    return SS_OK;
}

int ss_debug_mel_tile(int mode)
{
    // stored: 0 automatic, 1 eight waves + direct stores, 2 eight-wave builds only (tile when the batch allows), 3 twelve waves
    if (mode < 0 || mode > 3) return ss::fail(SS_ERR_ARG, "ss_debug_mel_tile: mode must be 0 .. 3");
    g_mel_tile_off.store(mode == 0 ? 1 : (mode == 1 ? 0 : mode), std::memory_order_relaxed);
    return SS_OK;
}

int ss_debug_tile_fault(int on)
{
    g_tile_fault.store(on ? 1u : 0u, std::memory_order_relaxed);
    return SS_OK;
}

int ss_debug_stamp_buffer(unsigned long long *d_stamps)
{
    g_stamp_buffer.store(d_stamps, std::memory_order_relaxed);
    return SS_OK;
}

int ss_debug_poison_lds(void *stream)
{
    int dev = 0, cus = 0;
    SS_HIP(hipGetDevice(&dev));
    SS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipError_t e = ss::launch_poison_lds(static_cast<hipStream_t>(stream), cus);
    if (e != hipSuccess) return hip_fail(e, "launch_poison_lds");
    return SS_OK;
}
#endif  // SS_LAB

// Shader clock the part held during launches of the MFCC batch kernel: `launches` launches on `stream`, each with the kernel's
// per-wave stamps switched on (a wave writes its lifetime in shader cycles, s_memtime, and on the constant 100 MHz clock,
// s_memrealtime, once, as it ends), into a buffer this call owns; the mean of cycles / lifetime over the waves of the last
// launch.  A per-call diagnostic: nothing process-wide is touched, launches of other threads are not affected.  The stamps
// exist in the 512-point kernel only: SS_ERR_UNSUPPORTED for configurations that run on another kernel.
int ss_mfcc_shader_clock(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld, float *d_out,
                         void *stream, int launches, float *ghz)
{
    if (!cfg || !ghz || launches <= 0) return ss::fail(SS_ERR_ARG, "bad clock request");
    *ghz = 0.f;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nwaves = static_cast<size_t>(cfg->num_cus) * 16, words = nwaves * 6;
    DeviceBuf db;
    int rc = db.alloc(words * sizeof(unsigned long long));
    if (rc) return rc;
    SS_HIP(hipMemsetAsync(db.p, 0, words * sizeof(unsigned long long), s));
    g_call_stamps = db.as<unsigned long long>();
    for (int i = 0; i < launches && rc == SS_OK; ++i) rc = ss_mfcc_batch_device(cfg, d_x, batch, n_samples, ld, d_out, stream);
    g_call_stamps = nullptr;
    const bool stamped = std::strncmp(g_last_kernel, "ss_mfcc_c256<", 13) == 0;
    const hipError_t es = hipStreamSynchronize(s);
    if (rc) return rc;
    if (es != hipSuccess) return hip_fail(es, "hipStreamSynchronize");
    if (!stamped) return ss::fail(SS_ERR_UNSUPPORTED, "the shader-clock stamps exist in the 512-point MFCC kernel only");
    std::vector<unsigned long long> w(words);
    SS_HIP(hipMemcpy(w.data(), db.p, words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0.0;
    size_t n = 0;
    for (size_t k = 0; k < nwaves; ++k) {
        const unsigned long long t0 = w[6 * k], t1 = w[6 * k + 2], cyc = w[6 * k + 5];
        if (t1 <= t0 || (cyc >> 40) != 1) continue;  // waves that did not run / the two table waves of a workgroup (they report something else)
        sum += static_cast<double>(cyc & ((1ull << 40) - 1)) / (static_cast<double>(t1 - t0) * 10.0);  // cycles per ns
        ++n;
    }
    if (n == 0) return ss::fail(SS_ERR_DEVICE, "no wave reported its lifetime");
    *ghz = static_cast<float>(sum / static_cast<double>(n));
    return SS_OK;
}

extern "C++" {  // (a template: C++ linkage inside the extern "C" block)
namespace {
// Sums of the per-wave stamps of a timed region's launches: sums[0] += shader cycles lived, sums[1] += 100 MHz ticks lived,
// sums[2] += waves counted.  `slots` launches, each with `slot_words` words of records for `nwaves` waves.  Format 0 (512-point
// MFCC kernel): six words per wave -- start / end on the 100 MHz clock in words 0 / 2, cycles lived flagged 1 in bits 40..41 of
// word 5 (the two table waves of a workgroup report something else there and are skipped).  Format 1 (twelve-wave 4096-point MFCC
// and 2048-point mel builds): two words per wave -- cycles lived, ticks lived.  Records of waves that never ran are zero.
static __global__ __launch_bounds__(256) void stamp_sums_kernel(const unsigned long long *w, unsigned long long slots, unsigned long long slot_words,
                                                                unsigned long long nwaves, int format, unsigned long long *sums)
{
    unsigned long long cyc = 0, ticks = 0, cnt = 0;
    const unsigned long long n = slots * nwaves;
    for (unsigned long long k = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; k < n;
         k += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
        const unsigned long long *r = w + (k / nwaves) * slot_words + (k % nwaves) * (format == 0 ? 6ull : 2ull);
        if (format == 0) {
            const unsigned long long t0 = r[0], t1 = r[2], c = r[5];
            if (t1 <= t0 || (c >> 40) != 1) continue;
            cyc += c & ((1ull << 40) - 1);
            ticks += t1 - t0;
        } else {
            if (r[1] == 0) continue;
            cyc += r[0];
            ticks += r[1];
        }
        ++cnt;
    }
    for (int m = 32; m > 0; m >>= 1) {
        cyc += __shfl_xor(cyc, m, 64);
        ticks += __shfl_xor(ticks, m, 64);
        cnt += __shfl_xor(cnt, m, 64);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        atomicAdd(&sums[0], cyc);
        atomicAdd(&sums[1], ticks);
        atomicAdd(&sums[2], cnt);
    }
}

// The timed region behind ss_mfcc_timed_region / ss_mel_spectrogram_timed_region: `launches` launches through `launch(i)` on
// `stream` between two HIP events, the per-wave stamps of the last `stamped` of them kept (each launch in a slot of its own of a
// buffer this call owns) and summed on the device.
template <typename Launch>
int timed_region(const ss_config *cfg, void *stream, int launches, int stamped, float *avg_ms, float *ghz, float *wall_ms, Launch launch)
{
    *avg_ms = 0.f;
    *ghz = 0.f;
    if (wall_ms) *wall_ms = 0.f;
    if (stamped > launches) stamped = launches;
    if (stamped > 4096) stamped = 4096;  // 196 KB of wave records per stamped launch
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nwaves = static_cast<size_t>(cfg->num_cus) * 16, words = nwaves * 6;
    DeviceBuf db;
    int rc = db.alloc(std::max<size_t>(1, static_cast<size_t>(stamped)) * words * sizeof(unsigned long long));
    if (rc) return rc;
    if (stamped) SS_HIP(hipMemsetAsync(db.p, 0, static_cast<size_t>(stamped) * words * sizeof(unsigned long long), s));
    hipEvent_t e0, e1;
    SS_HIP(hipEventCreate(&e0));
    {
        const hipError_t ec = hipEventCreate(&e1);
        if (ec != hipSuccess) {
            (void)hipEventDestroy(e0);
            return hip_fail(ec, "hipEventCreate");
        }
    }
    {
        const hipError_t es = hipStreamSynchronize(s);  // the region starts on an idle stream (the memset above is not part of it)
        if (es != hipSuccess) {
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return hip_fail(es, "hipStreamSynchronize");
        }
    }
    const auto w0 = std::chrono::steady_clock::now();
    (void)hipEventRecord(e0, s);
    int format = -1;  // which record format the stamped launches wrote (-2: a kernel without stamps, or not always the same one)
    for (int i = 0; i < launches && rc == SS_OK; ++i) {
        const int slot = i - (launches - stamped);
        g_call_stamps = g_call_stamps2 = slot >= 0 ? db.as<unsigned long long>() + static_cast<size_t>(slot) * words : nullptr;
        rc = launch(i);
        if (slot >= 0) {
            const int f = std::strncmp(g_last_kernel, "ss_mfcc_c256<", 13) == 0 ? 0
                          : (std::strncmp(g_last_kernel, "ss_mfcc_c2048<exact,mel8321,w12>", 32) == 0 || std::strncmp(g_last_kernel, "ss_mel_c1024<w12", 16) == 0) ? 1 : -2;
            format = (format == -1 || format == f) ? f : -2;
        }
    }
    g_call_stamps = g_call_stamps2 = nullptr;
    (void)hipEventRecord(e1, s);
    hipError_t e = hipSuccess;
    while ((e = hipEventQuery(e1)) == hipErrorNotReady) {  // polled: a blocking wait's wake-up latency is several launches long
    }
    if (wall_ms) *wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(e, "event timing");
    *avg_ms = ms / static_cast<float>(launches);
    if (stamped == 0) return SS_OK;
    if (format < 0) return ss::fail(SS_ERR_UNSUPPORTED, "the wave stamps exist in the 512-point MFCC kernel and in the twelve-wave builds of the 4096-point MFCC and 2048-point mel kernels only");
    // the records are summed on the device (1000 stamped launches are 196 MB of them): three words come back
    DeviceBuf sums;
    rc = sums.alloc(3 * sizeof(unsigned long long));
    if (rc) return rc;
    SS_HIP(hipMemsetAsync(sums.p, 0, 3 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(stamp_sums_kernel, dim3(1024), dim3(256), 0, s, db.as<unsigned long long>(), static_cast<unsigned long long>(stamped),
                       static_cast<unsigned long long>(words), static_cast<unsigned long long>(nwaves), format, sums.as<unsigned long long>());
    SS_HIP(hipGetLastError());
    unsigned long long hs[3] = {0, 0, 0};
    SS_HIP(hipMemcpyAsync(hs, sums.p, sizeof hs, hipMemcpyDeviceToHost, s));
    SS_HIP(hipStreamSynchronize(s));
    if (hs[1] == 0 || hs[2] == 0) return ss::fail(SS_ERR_DEVICE, "no wave reported its lifetime");
    *ghz = static_cast<float>(static_cast<double>(hs[0]) / (static_cast<double>(hs[1]) * 10.0));  // cycles per ns
    return SS_OK;
}
}  // namespace
}  // extern "C++"

// A timed region whose duration AND shader clock come from the same launches: `launches` launches of the MFCC batch kernel on
// `stream`, launch i reading d_x[i % n_x] and writing d_out[i % n_out] (a ring of inputs larger than the Infinity Cache keeps the
// samples coming from HBM), HIP events on `stream` around all of them, and the per-wave stamps of the LAST `stamped` launches kept,
// each launch in a slot of its own of a buffer this call owns.  *avg_ms = region / launches; *ghz = sum of the waves' shader
// cycles / sum of their lifetimes on the 100 MHz clock over every stamped launch.
int ss_mfcc_timed_region(const ss_config *cfg, const float *const *d_x, size_t n_x, size_t batch, size_t n_samples, size_t ld,
                         float *const *d_out, size_t n_out, void *stream, int launches, int stamped, float *avg_ms, float *ghz,
                         float *wall_ms)
{
    if (!cfg || !d_x || !d_out || n_x == 0 || n_out == 0 || launches <= 0 || stamped < 0 || !avg_ms || !ghz)
        return ss::fail(SS_ERR_ARG, "bad timed-region request");
    return timed_region(cfg, stream, launches, stamped, avg_ms, ghz, wall_ms, [&](int i) {
        return ss_mfcc_batch_device(cfg, d_x[static_cast<size_t>(i) % n_x], batch, n_samples, ld, d_out[static_cast<size_t>(i) % n_out], stream);
    });
}

// the same for the mel-spectrogram path (2048-point kernel, twelve-wave builds)
int ss_mel_spectrogram_timed_region(const ss_config *cfg, const float *const *d_x, size_t n_x, size_t channels, size_t n_samples, size_t ld,
                                    float *const *d_out, size_t n_out, void *stream, int launches, int stamped, float *avg_ms, float *ghz,
                                    float *wall_ms)
{
    if (!cfg || !d_x || !d_out || n_x == 0 || n_out == 0 || launches <= 0 || stamped < 0 || !avg_ms || !ghz)
        return ss::fail(SS_ERR_ARG, "bad timed-region request");
    return timed_region(cfg, stream, launches, stamped, avg_ms, ghz, wall_ms, [&](int i) {
        return ss_mel_spectrogram_device(cfg, d_x[static_cast<size_t>(i) % n_x], channels, n_samples, ld, d_out[static_cast<size_t>(i) % n_out], stream);
    });
}

namespace {
// One wave: shader cycles (s_memtime) against the constant 100 MHz counter (s_memrealtime) over about `ticks` of the latter,
// asleep in between (s_sleep: no issue slots, no memory traffic -- the workload beside it is not disturbed).  A lead-in of a
// tenth of the interval (at most 200 us) is slept through first and not counted: a probe launched just ahead of the load it is
// meant to watch does not average the idle clock of those first microseconds in.
static __global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *out, unsigned ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned lead = ticks / 10u < 20000u ? ticks / 10u : 20000u;
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tl = t;
    while (t - tl < lead) {
        __builtin_amdgcn_s_sleep(64);
        t = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    t = t0;
    while (t - t0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        t = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[0] = c1 - c0;
    out[1] = t1 - t0;
}
}  // namespace

int ss_shader_clock_probe_async(void *stream, uint32_t micros, unsigned long long *d_words)
{
    if (!d_words || micros < 10 || micros > 1000000) return ss::fail(SS_ERR_ARG, "bad clock probe request");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), d_words, micros * 100u);
    SS_HIP(hipGetLastError());
    return SS_OK;
}

int ss_shader_clock_probe(void *stream, uint32_t micros, float *ghz)
{
    if (!ghz || micros < 10 || micros > 1000000) return ss::fail(SS_ERR_ARG, "bad clock probe request");
    *ghz = 0.f;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceBuf db;
    int rc = db.alloc(2 * sizeof(unsigned long long));
    if (rc) return rc;
    SS_HIP(hipMemsetAsync(db.p, 0, 2 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, db.as<unsigned long long>(), micros * 100u);
    SS_HIP(hipGetLastError());
    unsigned long long w[2] = {0, 0};
    SS_HIP(hipMemcpyAsync(w, db.p, sizeof w, hipMemcpyDeviceToHost, s));
    SS_HIP(hipStreamSynchronize(s));
    if (w[1] == 0) return ss::fail(SS_ERR_DEVICE, "the clock probe reported nothing");
    *ghz = static_cast<float>(static_cast<double>(w[0]) / (static_cast<double>(w[1]) * 10.0));  // cycles per ns
    return SS_OK;
}

int ss_time_mfcc_batch_device(const ss_config *cfg, const float *d_x, size_t batch, size_t n_samples, size_t ld,
                              float *d_out, void *stream, int iters, float *avg_ms)
{
    if (!avg_ms || iters <= 0) return ss::fail(SS_ERR_ARG, "bad timing request");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    SS_HIP(hipEventCreate(&e0));
    {
        const hipError_t ec = hipEventCreate(&e1);
        if (ec != hipSuccess) {
            (void)hipEventDestroy(e0);
            return hip_fail(ec, "hipEventCreate");
        }
    }
    int rc = ss_mfcc_batch_device(cfg, d_x, batch, n_samples, ld, d_out, stream);  // warm-up
    if (rc == SS_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && rc == SS_OK; ++i) rc = ss_mfcc_batch_device(cfg, d_x, batch, n_samples, ld, d_out, stream);
        (void)hipEventRecord(e1, s);
        hipError_t e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = hip_fail(e, "event timing");
        *avg_ms = ms / static_cast<float>(iters);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int ss_time_mel_spectrogram_device(const ss_config *cfg, const float *d_x, size_t channels, size_t n_samples,
                                   size_t ld, float *d_out, void *stream, int iters, float *avg_ms)
{
    if (!avg_ms || iters <= 0) return ss::fail(SS_ERR_ARG, "bad timing request");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    SS_HIP(hipEventCreate(&e0));
    {
        const hipError_t ec = hipEventCreate(&e1);
        if (ec != hipSuccess) {
            (void)hipEventDestroy(e0);
            return hip_fail(ec, "hipEventCreate");
        }
    }
    int rc = ss_mel_spectrogram_device(cfg, d_x, channels, n_samples, ld, d_out, stream);
    if (rc == SS_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && rc == SS_OK; ++i) rc = ss_mel_spectrogram_device(cfg, d_x, channels, n_samples, ld, d_out, stream);
        (void)hipEventRecord(e1, s);
        hipError_t e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = hip_fail(e, "event timing");
        *avg_ms = ms / static_cast<float>(iters);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

}  // extern "C"
