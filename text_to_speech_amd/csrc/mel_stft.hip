// mel_stft.hip -- TacotronSTFT mel spectrogram on gfx950.
//
// Replaces /root/reference/utils/audio/stft.py:242-274 (STFT.transform: reflect pad, windowed-DFT conv1d, magnitude)
// and :306-314 (TacotronSTFT.mel_spectrogram: mag @ mel_basis^T, log(max(., 1e-5))).  The reference computes the DFT
// as a dense conv1d against a [1024, 1, 1026] basis; so does this file: frames are overlapping rows (stride 256) of
// the reflect-padded signal, fed to the fp32 MFMA GEMM without materialising them.
#include "engine.h"
#include "gemm_f32.h"

#include <cmath>

using namespace ttsgemm;

namespace {

constexpr int FL = 1024, HOP = 256, CUT = 513, NMEL = 80;
constexpr int NB = 1056;      // 2 * 513 = 1026 basis rows padded to a multiple of 32
constexpr int MAGK = 544;     // 513 padded to a multiple of 32

// y[b][p] = x[b][reflect(p - 512)], rows of y are NPS floats apart (NPS = N + 1024 rounded up to 4 for 16-B rows)
__global__ void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int B, int NPS) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * NPS) return;
    const int b = (int)(idx / NPS), p = (int)(idx % NPS);
    float v = 0.f;
    if (p < N + FL) {
        int s = p - FL / 2;
        if (s < 0) s = -s;                   // numpy/keras 'reflect' (edge sample not repeated)
        if (s >= N) s = 2 * (N - 1) - s;
        v = x[(long long)b * N + s];
    }
    y[idx] = v;
}

// mag[f][c] = sqrt(re^2 + im^2), c < 513; zero in the K padding
__global__ void magnitude_kernel(const float* __restrict__ ft, float* __restrict__ mag, long long rows) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * MAGK) return;
    const long long r = idx / MAGK;
    const int c = (int)(idx % MAGK);
    float v = 0.f;
    if (c < CUT) {
        const float re = ft[r * NB + c], im = ft[r * NB + CUT + c];
        v = sqrtf(re * re + im * im);
    }
    mag[idx] = v;
}

__global__ void log_clamp_kernel(float* __restrict__ m, long long n, float clip) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) m[idx] = logf(fmaxf(m[idx], clip));
}

double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

}  // namespace

void melstft_free(tts_hip_engine* e) {
    for (void* p : e->stft.allocs) (void)hipFree(p);
    e->stft.allocs.clear();
    e->stft.frames.release();
    e->stft.mag.release();
    e->stft.io_in.release();
    e->stft.io_out.release();
    e->stft.ready = false;
}

int melstft_finalize(tts_hip_engine* e) {
    melstft_free(e);
    // windowed DFT rows (stft.py:211-236): periodic Hann, rows 0..512 real, 513..1025 imaginary
    std::vector<float> basis((size_t)NB * FL, 0.f);
    for (int r = 0; r < CUT; ++r)
        for (int n = 0; n < FL; ++n) {
            const double win = 0.5 - 0.5 * std::cos(2.0 * M_PI * n / FL);
            const int kn = (int)(((long long)r * n) % FL);          // exact phase reduction
            const double ang = 2.0 * M_PI * kn / FL;
            basis[(size_t)r * FL + n] = (float)((double)(float)std::cos(ang) * win);
            basis[(size_t)(CUT + r) * FL + n] = (float)((double)(float)(-std::sin(ang)) * win);
        }
    int rc = upload(e, basis.data(), basis.size(), &e->stft.basis_Bt, e->stft.allocs);
    if (rc) return rc;
    // Slaney mel filterbank (librosa.filters.mel defaults; stft.py:65-72), sr 22050, fmin 0, fmax 8000
    const double sr = 22050.0, fmin = 0.0, fmax = 8000.0;
    std::vector<double> mel_f(NMEL + 2);
    const double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
    for (int i = 0; i < NMEL + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (NMEL + 1));
    std::vector<float> mb((size_t)NMEL * MAGK, 0.f);
    for (int i = 0; i < NMEL; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int c = 0; c < CUT; ++c) {
            const double fr = (sr / 2.0) * c / (CUT - 1);
            const double lower = (fr - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - fr) / (mel_f[i + 2] - mel_f[i + 1]);
            const double w = std::fmax(0.0, std::fmin(lower, upper));
            mb[(size_t)i * MAGK + c] = (float)(w * enorm);
        }
    }
    rc = upload(e, mb.data(), mb.size(), &e->stft.mel_Bt, e->stft.allocs);
    if (rc) return rc;
    e->stft.ready = true;
    return TTS_HIP_OK;
}

int melstft_run(tts_hip_engine* e, const float* d_audio, int B, int N, float* d_mel) {
    MelStftDev& s = e->stft;
    const int NP = (N + FL + 3) / 4 * 4;      // padded row stride (16-B aligned rows for the float4 operand loads)
    const int F = N / HOP + 1;
    hipStream_t st = e->stream;
    HIPCHK(e, s.frames.ensure(((size_t)B * NP + 64) * 4 + (size_t)B * F * NB * 4));
    HIPCHK(e, s.mag.ensure((size_t)B * F * MAGK * 4));
    float* padded = s.frames.f();
    float* ft = padded + (((size_t)B * NP + 63) / 64) * 64;
    {
        const long long n = (long long)B * NP;
        hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_audio, padded, N,
                           B, NP);
        HIPCHK(e, hipGetLastError());
    }
    {   // ft[b][f][r] = sum_n padded[b][f*256 + n] * basis[r][n]
        GemmArgs g{};
        g.M = F;
        g.N = NB;
        g.L = F;
        g.nseg = 1;
        g.seg[0] = ASeg{padded, HOP, 0, FL, FL};
        g.strideAz = NP;
        g.Bt = s.basis_Bt;
        g.ldb = FL;
        g.mode = EPI_LINEAR;
        g.split = NB;
        g.out0 = ft;
        g.ld0 = NB;
        g.strideOutZ = (long long)F * NB;
        HIPCHK(e, gemm_small(g, B, st));
    }
    {
        const long long n = (long long)B * F * MAGK;
        hipLaunchKernelGGL(magnitude_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ft, s.mag.f(),
                           (long long)B * F);
        HIPCHK(e, hipGetLastError());
    }
    {   // mel[m][j] = sum_c mag[m][c] * mel_basis[j][c]
        GemmArgs g{};
        g.M = B * F;
        g.N = NMEL;
        g.L = B * F;
        g.nseg = 1;
        g.seg[0] = ASeg{s.mag.f(), MAGK, 0, MAGK, MAGK};
        g.Bt = s.mel_Bt;
        g.ldb = MAGK;
        g.mode = EPI_LINEAR;
        g.split = NMEL;
        g.out0 = d_mel;
        g.ld0 = NMEL;
        HIPCHK(e, gemm_small(g, 1, st));
    }
    {
        const long long n = (long long)B * F * NMEL;
        hipLaunchKernelGGL(log_clamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_mel, n, 1e-5f);
        HIPCHK(e, hipGetLastError());
    }
    return TTS_HIP_OK;
}
