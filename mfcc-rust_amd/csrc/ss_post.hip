// Post-processing on the feature matrix: cmvn, cmvnw, derivative_extraction, extract_derivative_feature.
// Reference: speechsauce/src/processing.rs:222-254 (derivative_extraction), :265-300 (cmvn), :315-371 (cmvnw),
// feature.rs:253-269 (extract_derivative_feature); the pads are np.pad 'edge' / 'symmetric' as util.rs:108-124 quotes.
//
// The data is the [clips x rows x cols] feature block the hot path just wrote (5 MB for 1024 one-second clips), so these
// are small HBM/L2-bound kernels with neighbouring threads on neighbouring addresses (columns fastest) and f64 accumulators
// for the statistics (full rate on gfx950; it keeps the result within an ulp of the f64 oracle for 6 000-row matrices too):
// cmvn sums chunks of rows per thread and normalises per element (two launches, no atomics: bit-reproducible); cmvnw slides
// its window sums over a chunk of rows per thread (O(rows + win) loads per column chunk instead of O(rows * win)).
#include "ss_internal.h"
#include "speechsauce_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>

namespace ss {

namespace {

constexpr double kEps30 = 9.313225746154785e-10;  // 2f32.powf(-30.), processing.rs:266, :324

// Walker over np.pad(..., 'symmetric') of an axis of length n (reflection that repeats the edge sample, period 2n):
// position p of the padded axis maps to idx; step() moves to p + 1 without a division.
struct SymWalk {
    unsigned idx;
    int dir;
    unsigned n;
    __device__ SymWalk(long long p, unsigned n_) : n(n_)
    {
        const long long period = 2ll * n_;
        long long m = p % period;
        if (m < 0) m += period;
        dir = m < n_ ? 1 : -1;
        idx = static_cast<unsigned>(m < n_ ? m : period - 1 - m);
    }
    __device__ void step()
    {
        if (dir > 0) {
            if (idx + 1 == n) dir = -1;  // the edge sample repeats
            else ++idx;
        } else {
            if (idx == 0) dir = 1;
            else --idx;
        }
    }
};

// ---- element-wise logarithms: lmfe's ln (feature.rs:242-245 -> util.rs:372-381) and librosa's power_to_db ----
__global__ __launch_bounds__(256) void ss_ln_kernel(float *__restrict__ x, unsigned long long n)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g < n) x[g] = logf(x[g]);
}

// floats as ordered integers, so that atomicMax on an int finds the largest float (negative values included)
__device__ __forceinline__ int float_key(float v)
{
    const int b = __float_as_int(v);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float key_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// 10 log10(max(amin, S)) - 10 log10(max(amin, ref)); the block maxima go to one word for the top_db clamp
__global__ __launch_bounds__(256) void ss_power_to_db_kernel(const float *__restrict__ s, float *__restrict__ out, unsigned long long n, float amin,
                                                            float ref_db, int *__restrict__ max_key)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    float db = -INFINITY;
    if (g < n) {
        db = 10.0f * log10f(fmaxf(amin, s[g])) - ref_db;
        out[g] = db;
    }
    if (max_key) {
        for (int m = 1; m < 64; m <<= 1) db = fmaxf(db, __shfl_xor(db, m, 64));
        if ((threadIdx.x & 63) == 0 && db > -INFINITY) atomicMax(max_key, float_key(db));
    }
}

__global__ __launch_bounds__(256) void ss_db_floor_kernel(float *__restrict__ out, unsigned long long n, float top_db, const int *__restrict__ max_key)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g < n) out[g] = fmaxf(out[g], key_float(*max_key) - top_db);
}

// ---- cmvn: column statistics in two kernels, no atomics (bit-reproducible) ----
// partial sums of x and x^2 over a chunk of rows; thread = (clip, chunk, column), columns fastest
__global__ __launch_bounds__(256) void ss_cmvn_partial_kernel(const float *__restrict__ x, double *__restrict__ part, unsigned long long total,
                                                             unsigned rows, unsigned cols, unsigned chunks, unsigned rpc)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const unsigned c = static_cast<unsigned>(g % cols);
    const unsigned long long cc = g / cols;
    const unsigned chunk = static_cast<unsigned>(cc % chunks);
    const unsigned long long clip = cc / chunks;
    const unsigned r0 = chunk * rpc, r1 = min(rows, r0 + rpc);
    const float *src = x + clip * rows * cols + c;
    double s1 = 0.0, s2 = 0.0;
    for (unsigned r = r0; r < r1; ++r) {
        const double v = static_cast<double>(src[static_cast<size_t>(r) * cols]);
        s1 += v;
        s2 += v * v;
    }
    part[2 * g] = s1;
    part[2 * g + 1] = s2;
}

// thread = output element: mean (and population std) of its column from the chunk partials, then (x - mean) / (std + 2^-30)
__global__ __launch_bounds__(256) void ss_cmvn_apply_kernel(const float *__restrict__ x, const double *__restrict__ part, float *__restrict__ out,
                                                           unsigned long long total, unsigned rows, unsigned cols, unsigned chunks, int variance)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const unsigned long long per_clip = static_cast<unsigned long long>(rows) * cols;
    const unsigned long long clip = g / per_clip;
    const unsigned c = static_cast<unsigned>((g - clip * per_clip) % cols);
    const double *pp = part + 2 * (clip * chunks * cols + c);
    double s1 = 0.0, s2 = 0.0;
    for (unsigned k = 0; k < chunks; ++k) {
        s1 += pp[2 * static_cast<size_t>(k) * cols];
        s2 += pp[2 * static_cast<size_t>(k) * cols + 1];
    }
    const double mean = s1 / rows;
    double inv = 1.0;
    if (variance) {
        const double var = fmax(s2 / rows - mean * mean, 0.0);  // std_axis(Axis(0), 0.) of the mean-subtracted column (processing.rs:283)
        inv = 1.0 / (sqrt(var) + kEps30);
    }
    out[g] = static_cast<float>((static_cast<double>(x[g]) - mean) * inv);
}

// ---- cmvnw: sliding window sums over the symmetric-padded rows; thread = (clip, chunk of rows, column) ----
// pass 1: ms[i] = x[i] - mean of the win rows centred on i
__global__ __launch_bounds__(256) void ss_cmvnw_mean_kernel(const float *__restrict__ x, float *__restrict__ ms, unsigned long long total,
                                                            unsigned rows, unsigned cols, unsigned win, unsigned chunks, unsigned rpc)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const unsigned c = static_cast<unsigned>(g % cols);
    const unsigned long long cc = g / cols;
    const unsigned chunk = static_cast<unsigned>(cc % chunks);
    const unsigned long long clip = cc / chunks;
    const unsigned i0 = chunk * rpc, i1 = min(rows, i0 + rpc);
    const float *src = x + clip * rows * cols + c;
    float *dst = ms + clip * rows * cols + c;
    const long long pad = (win - 1) / 2;
    SymWalk head(static_cast<long long>(i0) - pad, rows), tail = head;  // tail: oldest row of the window, head: next row to enter
    double s = 0.0;
#pragma unroll 4
    for (unsigned w = 0; w < win; ++w) {
        s += static_cast<double>(src[static_cast<size_t>(head.idx) * cols]);
        head.step();
    }
    for (unsigned i = i0; i < i1; ++i) {
        dst[static_cast<size_t>(i) * cols] = static_cast<float>(static_cast<double>(src[static_cast<size_t>(i) * cols]) - s / win);
        s += static_cast<double>(src[static_cast<size_t>(head.idx) * cols]) - static_cast<double>(src[static_cast<size_t>(tail.idx) * cols]);
        head.step();
        tail.step();
    }
}

// pass 2 (variance_normalization): ms / (population std of ms over the same symmetric window + 2^-30)
__global__ __launch_bounds__(256) void ss_cmvnw_var_kernel(const float *__restrict__ ms, float *__restrict__ out, unsigned long long total,
                                                           unsigned rows, unsigned cols, unsigned win, unsigned chunks, unsigned rpc)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const unsigned c = static_cast<unsigned>(g % cols);
    const unsigned long long cc = g / cols;
    const unsigned chunk = static_cast<unsigned>(cc % chunks);
    const unsigned long long clip = cc / chunks;
    const unsigned i0 = chunk * rpc, i1 = min(rows, i0 + rpc);
    const float *src = ms + clip * rows * cols + c;
    float *dst = out + clip * rows * cols + c;
    const long long pad = (win - 1) / 2;
    SymWalk head(static_cast<long long>(i0) - pad, rows), tail = head;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
    for (unsigned w = 0; w < win; ++w) {
        const double v = static_cast<double>(src[static_cast<size_t>(head.idx) * cols]);
        s1 += v;
        s2 += v * v;
        head.step();
    }
    for (unsigned i = i0; i < i1; ++i) {
        const double m = s1 / win;
        const double var = fmax(s2 / win - m * m, 0.0);
        dst[static_cast<size_t>(i) * cols] = static_cast<float>(static_cast<double>(src[static_cast<size_t>(i) * cols]) / (sqrt(var) + kEps30));
        const double vin = static_cast<double>(src[static_cast<size_t>(head.idx) * cols]);
        const double vout = static_cast<double>(src[static_cast<size_t>(tail.idx) * cols]);
        s1 += vin - vout;
        s2 += vin * vin - vout * vout;
        head.step();
        tail.step();
    }
}

// derivative along the FEATURE axis with edge clamping, literal reference arithmetic: sum_R (R f[c+R] - f[c-R]) / sum_R 2R^2
__device__ __forceinline__ float deriv_at(const float *row, unsigned cols, unsigned c, unsigned dw, float inv_scale)
{
    float acc = 0.f;
    for (unsigned R = 1; R <= dw; ++R) {
        const unsigned hi = c + R < cols ? c + R : cols - 1;
        const unsigned lo = c >= R ? c - R : 0;
        acc += row[hi] * static_cast<float>(R) - row[lo];
    }
    return acc * inv_scale;
}

__global__ __launch_bounds__(256) void ss_derivative_kernel(const float *__restrict__ x, float *__restrict__ out, unsigned long long total,
                                                            unsigned cols, unsigned dw, float inv_scale)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const unsigned long long r = g / cols;
    const unsigned c = static_cast<unsigned>(g - r * cols);
    out[g] = deriv_at(x + r * cols, cols, c, dw, inv_scale);
}

// cube[r][c][0..2] = (f, d1, d2), d1 = derivative(f, 2), d2 = derivative(d1, 2); d1 is recomputed for the 5 clamped neighbours
__global__ __launch_bounds__(256) void ss_derivative_cube_kernel(const float *__restrict__ x, float *__restrict__ cube, unsigned long long total,
                                                                 unsigned cols)
{
    const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const unsigned long long r = g / cols;
    const unsigned c = static_cast<unsigned>(g - r * cols);
    const float *row = x + r * cols;
    constexpr float inv10 = 1.0f / 10.0f;  // sum_{R=1,2} 2 R^2
    float acc = 0.f;
    for (unsigned R = 1; R <= 2; ++R) {
        const unsigned hi = c + R < cols ? c + R : cols - 1;
        const unsigned lo = c >= R ? c - R : 0;
        acc += deriv_at(row, cols, hi, 2, inv10) * static_cast<float>(R) - deriv_at(row, cols, lo, 2, inv10);
    }
    cube[3 * g] = row[c];
    cube[3 * g + 1] = deriv_at(row, cols, c, 2, inv10);
    cube[3 * g + 2] = acc * inv10;
}

int hip_err(hipError_t e, const char *what) { return fail(SS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }

unsigned blocks_for(unsigned long long n) { return static_cast<unsigned>((n + 255) / 256); }

int check_shape(const void *a, const void *b, size_t batch, size_t rows, size_t cols)
{
    if (!a || !b) return fail(SS_ERR_ARG, "null buffer");
    if (rows == 0 || cols == 0) return fail(SS_ERR_ARG, "empty feature matrix");
    if (rows >= (1ull << 31) || cols >= (1ull << 31) || batch * rows * cols / 256 >= (1ull << 31)) return fail(SS_ERR_ARG, "feature block too large");
    return SS_OK;
}

// host-pointer wrapper: upload, run the device entry point, download
template <typename F>
int via_device(const float *in, size_t n_in, float *out, size_t n_out, F &&run)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SS_ERR_HIP, "no usable HIP device: the speechsauce_amd path has no CPU fallback");
    float *d_in = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_in), n_in * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_out), n_out * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_in, in, n_in * sizeof(float), hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? run(d_in, d_out) : hip_err(e, "host staging");
    if (rc == SS_OK) {
        e = hipMemcpy(out, d_out, n_out * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_err(e, "hipMemcpy D2H");
    }
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // namespace

}  // namespace ss

extern "C" {

int ss_cmvn_batch_device(const float *d_vec, size_t batch, size_t rows, size_t cols, int variance_normalization, float *d_out,
                         void *stream)
{
    int rc = ss::check_shape(d_vec, d_out, batch, rows, cols);
    if (rc) return rc;
    if (batch == 0) return SS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // rows are summed in chunks (one thread per clip, chunk and column), then every element is normalised
    const unsigned chunks = static_cast<unsigned>(std::min<size_t>(64, (rows + 31) / 32));
    const unsigned rpc = static_cast<unsigned>((rows + chunks - 1) / chunks);
    const unsigned long long np = static_cast<unsigned long long>(batch) * chunks * cols;
    const unsigned long long n = static_cast<unsigned long long>(batch) * rows * cols;
    double *d_part = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&d_part), np * 2 * sizeof(double), s);
    if (e != hipSuccess) return ss::hip_err(e, "hipMallocAsync");
    hipLaunchKernelGGL(ss::ss_cmvn_partial_kernel, dim3(ss::blocks_for(np)), dim3(256), 0, s, d_vec, d_part, np, static_cast<unsigned>(rows),
                       static_cast<unsigned>(cols), chunks, rpc);
    hipLaunchKernelGGL(ss::ss_cmvn_apply_kernel, dim3(ss::blocks_for(n)), dim3(256), 0, s, d_vec, d_part, d_out, n, static_cast<unsigned>(rows),
                       static_cast<unsigned>(cols), chunks, variance_normalization);
    e = hipFreeAsync(d_part, s);
    if (e != hipSuccess) return ss::hip_err(e, "hipFreeAsync");
    e = hipGetLastError();
    return e == hipSuccess ? SS_OK : ss::hip_err(e, "ss_cmvn kernels");
}

int ss_cmvnw_batch_device(const float *d_vec, size_t batch, size_t rows, size_t cols, size_t win_size, int variance_normalization,
                          float *d_out, void *stream)
{
    int rc = ss::check_shape(d_vec, d_out, batch, rows, cols);
    if (rc) return rc;
    if (win_size % 2 != 1) return ss::fail(SS_ERR_BAD_CONFIG, "Windows size must be odd!");  // assert, processing.rs:327
    if (win_size >= (1ull << 31)) return ss::fail(SS_ERR_ARG, "window too large");
    if (batch == 0) return SS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned r = static_cast<unsigned>(rows), c = static_cast<unsigned>(cols), w = static_cast<unsigned>(win_size);
    // a thread slides the window over a chunk of rows: chunks long enough to amortise the first window sum
    const unsigned rpc = static_cast<unsigned>(std::min<size_t>(256, std::max<size_t>(8, (win_size + 15) / 16)));
    const unsigned chunks = (r + rpc - 1) / rpc;
    const unsigned long long nt = static_cast<unsigned long long>(batch) * chunks * cols;
    const unsigned long long n = static_cast<unsigned long long>(batch) * rows * cols;
    if (!variance_normalization) {
        hipLaunchKernelGGL(ss::ss_cmvnw_mean_kernel, dim3(ss::blocks_for(nt)), dim3(256), 0, s, d_vec, d_out, nt, r, c, w, chunks, rpc);
    } else {
        // the second pass reads its neighbours' mean-subtracted values: they go through a stream-ordered scratch block
        float *d_ms = nullptr;
        hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&d_ms), n * sizeof(float), s);
        if (e != hipSuccess) return ss::hip_err(e, "hipMallocAsync");
        hipLaunchKernelGGL(ss::ss_cmvnw_mean_kernel, dim3(ss::blocks_for(nt)), dim3(256), 0, s, d_vec, d_ms, nt, r, c, w, chunks, rpc);
        hipLaunchKernelGGL(ss::ss_cmvnw_var_kernel, dim3(ss::blocks_for(nt)), dim3(256), 0, s, d_ms, d_out, nt, r, c, w, chunks, rpc);
        e = hipFreeAsync(d_ms, s);
        if (e != hipSuccess) return ss::hip_err(e, "hipFreeAsync");
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SS_OK : ss::hip_err(e, "ss_cmvnw kernels");
}

int ss_derivative_extraction_device(const float *d_feat, size_t rows, size_t cols, size_t delta_windows, float *d_out, void *stream)
{
    int rc = ss::check_shape(d_feat, d_out, 1, rows, cols);
    if (rc) return rc;
    if (delta_windows == 0 || delta_windows >= (1u << 20)) return ss::fail(SS_ERR_ARG, "delta_windows must be >= 1");  // scale = 0 in the reference
    double scale = 0.0;
    for (size_t R = 1; R <= delta_windows; ++R) scale += 2.0 * static_cast<double>(R) * static_cast<double>(R);
    const unsigned long long n = static_cast<unsigned long long>(rows) * cols;
    hipLaunchKernelGGL(ss::ss_derivative_kernel, dim3(ss::blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), d_feat, d_out, n,
                       static_cast<unsigned>(cols), static_cast<unsigned>(delta_windows), static_cast<float>(1.0 / scale));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SS_OK : ss::hip_err(e, "ss_derivative_kernel");
}

int ss_extract_derivative_feature_device(const float *d_feat, size_t rows, size_t cols, float *d_cube, void *stream)
{
    int rc = ss::check_shape(d_feat, d_cube, 1, rows, cols);
    if (rc) return rc;
    const unsigned long long n = static_cast<unsigned long long>(rows) * cols;
    hipLaunchKernelGGL(ss::ss_derivative_cube_kernel, dim3(ss::blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), d_feat, d_cube,
                       n, static_cast<unsigned>(cols));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SS_OK : ss::hip_err(e, "ss_derivative_cube_kernel");
}

// ---- host-pointer variants (synchronous) ----

int ss_cmvn(const float *vec, size_t rows, size_t cols, int variance_normalization, float *out)
{
    int rc = ss::check_shape(vec, out, 1, rows, cols);
    if (rc) return rc;
    return ss::via_device(vec, rows * cols, out, rows * cols, [&](const float *di, float *dout) {
        int r = ss_cmvn_batch_device(di, 1, rows, cols, variance_normalization, dout, nullptr);
        if (r == SS_OK && hipDeviceSynchronize() != hipSuccess) r = ss::fail(SS_ERR_HIP, "ss_cmvn: device error");
        return r;
    });
}

int ss_cmvnw(const float *vec, size_t rows, size_t cols, size_t win_size, int variance_normalization, float *out)
{
    int rc = ss::check_shape(vec, out, 1, rows, cols);
    if (rc) return rc;
    if (win_size % 2 != 1) return ss::fail(SS_ERR_BAD_CONFIG, "Windows size must be odd!");
    return ss::via_device(vec, rows * cols, out, rows * cols, [&](const float *di, float *dout) {
        int r = ss_cmvnw_batch_device(di, 1, rows, cols, win_size, variance_normalization, dout, nullptr);
        if (r == SS_OK && hipDeviceSynchronize() != hipSuccess) r = ss::fail(SS_ERR_HIP, "ss_cmvnw: device error");
        return r;
    });
}

int ss_derivative_extraction(const float *feat, size_t rows, size_t cols, size_t delta_windows, float *out)
{
    int rc = ss::check_shape(feat, out, 1, rows, cols);
    if (rc) return rc;
    if (delta_windows == 0) return ss::fail(SS_ERR_ARG, "delta_windows must be >= 1");
    return ss::via_device(feat, rows * cols, out, rows * cols, [&](const float *di, float *dout) {
        int r = ss_derivative_extraction_device(di, rows, cols, delta_windows, dout, nullptr);
        if (r == SS_OK && hipDeviceSynchronize() != hipSuccess) r = ss::fail(SS_ERR_HIP, "ss_derivative_extraction: device error");
        return r;
    });
}

int ss_extract_derivative_feature(const float *feat, size_t rows, size_t cols, float *cube)
{
    int rc = ss::check_shape(feat, cube, 1, rows, cols);
    if (rc) return rc;
    return ss::via_device(feat, rows * cols, cube, 3 * rows * cols, [&](const float *di, float *dout) {
        int r = ss_extract_derivative_feature_device(di, rows, cols, dout, nullptr);
        if (r == SS_OK && hipDeviceSynchronize() != hipSuccess) r = ss::fail(SS_ERR_HIP, "ss_extract_derivative_feature: device error");
        return r;
    });
}

int ss_ln_device(float *d_x, size_t n, void *stream)
{
    if (n == 0) return SS_OK;
    if (!d_x) return ss::fail(SS_ERR_ARG, "null buffer");
    hipLaunchKernelGGL(ss::ss_ln_kernel, dim3(ss::blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), d_x, static_cast<unsigned long long>(n));
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SS_OK : ss::hip_err(e, "ss_ln_kernel");
}

int ss_power_to_db_device(const float *d_s, size_t n, float ref, float amin, float top_db, float *d_out, void *stream)
{
    if (n == 0) return SS_OK;
    if (!d_s || !d_out) return ss::fail(SS_ERR_ARG, "null buffer");
    if (!(amin > 0.0f)) return ss::fail(SS_ERR_ARG, "amin must be strictly positive");  // librosa.power_to_db raises the same
    if (ref != ref) return ss::fail(SS_ERR_ARG, "ref must not be NaN");  // |ref| is used, as in librosa (np.abs(ref))
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float ref_db = 10.0f * std::log10(std::max(amin, std::fabs(ref)));
    int *d_max = nullptr;
    if (top_db >= 0.0f) {
        hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&d_max), sizeof(int), st);
        if (e != hipSuccess) return ss::hip_err(e, "hipMallocAsync");
        e = hipMemsetAsync(d_max, 0x80, sizeof(int), st);  // 0x80808080: below the key of every finite float
        if (e != hipSuccess) return ss::hip_err(e, "hipMemsetAsync");
    }
    hipLaunchKernelGGL(ss::ss_power_to_db_kernel, dim3(ss::blocks_for(n)), dim3(256), 0, st, d_s, d_out, static_cast<unsigned long long>(n), amin,
                       ref_db, d_max);
    if (d_max) {
        hipLaunchKernelGGL(ss::ss_db_floor_kernel, dim3(ss::blocks_for(n)), dim3(256), 0, st, d_out, static_cast<unsigned long long>(n), top_db, d_max);
        const hipError_t e = hipFreeAsync(d_max, st);
        if (e != hipSuccess) return ss::hip_err(e, "hipFreeAsync");
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SS_OK : ss::hip_err(e, "ss_power_to_db kernels");
}

int ss_power_to_db(const float *s, size_t n, float ref, float amin, float top_db, float *out)
{
    if (n == 0) return SS_OK;
    if (!s || !out) return ss::fail(SS_ERR_ARG, "null buffer");
    return ss::via_device(s, n, out, n, [&](const float *di, float *dout) {
        int r = ss_power_to_db_device(di, n, ref, amin, top_db, dout, nullptr);
        if (r == SS_OK && hipDeviceSynchronize() != hipSuccess) r = ss::fail(SS_ERR_HIP, "ss_power_to_db: device error");
        return r;
    });
}

}  // extern "C"
