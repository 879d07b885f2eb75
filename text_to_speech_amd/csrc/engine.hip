// engine.hip -- C ABI entry points: handle lifetime, weight intake, host/device staging, timing hooks.
#include "engine.h"
#include "ttsw_host.h"

#include <cstdarg>
#include <cstring>
#include <exception>

int set_err(const tts_hip_engine* e, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    return code;
}

const HostTensor* find_tensor(const tts_hip_engine* e, const std::string& name) {
    auto it = e->host.find(name);
    return it == e->host.end() ? nullptr : &it->second;
}

int dev_alloc(tts_hip_engine* e, size_t n_floats, float** dst, std::vector<void*>& allocs, bool zero) {
    void* p = nullptr;
    if (hipError_t err = hipMalloc(&p, n_floats * sizeof(float)); err != hipSuccess) {
        if (err == hipErrorOutOfMemory) (void)hipGetLastError();
        return set_err(e, err == hipErrorOutOfMemory ? TTS_HIP_ENOMEM : TTS_HIP_EHIP, "hipMalloc(%zu bytes) -> %s",
                       n_floats * sizeof(float), hipGetErrorString(err));
    }
    allocs.push_back(p);
    if (zero) HIPCHK(e, hipMemsetAsync(p, 0, n_floats * sizeof(float), e->stream));
    *dst = (float*)p;
    return TTS_HIP_OK;
}

int upload(tts_hip_engine* e, const float* src, size_t n, float** dst, std::vector<void*>& allocs) {
    int rc = dev_alloc(e, n, dst, allocs, false);
    if (rc) return rc;
    HIPCHK(e, hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return TTS_HIP_OK;
}

// ------------------------------------------------------------------------------------------- timing hooks
void timing_begin(tts_hip_engine* e, int kind) {
    if (!e->timing) return;
    TimedLaunch t;
    if (!e->ev_pool.empty()) {
        t = e->ev_pool.back();
        e->ev_pool.pop_back();
    } else {
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return;
    }
    t.kind = kind;
    (void)hipEventRecord(t.a, e->stream);
    e->timed.push_back(t);
}
void timing_end(tts_hip_engine* e) {
    if (!e->timing || e->timed.empty()) return;
    (void)hipEventRecord(e->timed.back().b, e->stream);
}
void timing_collect(tts_hip_engine* e) {
    if (e->timed.empty()) return;
    (void)hipStreamSynchronize(e->stream);
    for (auto& t : e->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess && t.kind >= 0 && t.kind < 4) {
            e->time_sum_us[t.kind] += 1e3 * ms;
            e->time_cnt[t.kind] += 1;
        }
        e->ev_pool.push_back(t);
    }
    e->timed.clear();
}

// ------------------------------------------------------------------------------------------- device-side sampling
// The reference draws WaveGlow's noise and the prenet dropout inside `infer`, on the device
// (/root/reference/architectures/waveglow_arch.py:272-274,299-302, tacotron2_arch.py:197-201).  Generator used here
// (documented so that a caller can reproduce a stream; restated in oracle/philox_ref.py):
//   Philox4x32-10 (Salmon et al., SC'11): key = (seed lo, seed hi), counter = (c lo, c hi, 0, 0) with c = offset + i / 4;
//   element i takes word i % 4 of block c.  A word x becomes u = ((x >> 8) + 0.5) * 2^-24 in (0, 1).
//   normals: words (0, 1) and (2, 3) of a block feed one Box-Muller pair each:
//     r = sqrt(-2 ln u_a), (z_a, z_b) = (r cos(2 pi u_b), r sin(2 pi u_b));
//   prenet masks (keep probability 0.5, scale 2, tacotron2_arch.py:188-203): 2.0 if the word's top bit is set else 0.0.
namespace {

struct Philox4 { uint32_t x[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(uint64_t ctr, uint64_t seed) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}
__device__ __forceinline__ float unit_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-8f; }

// kind 0: standard normals, kind 1: prenet dropout masks.  One thread per Philox block (4 outputs).
__global__ void philox_fill_kernel(float* __restrict__ out, long long n, uint64_t seed, uint64_t offset, int kind) {
    const long long blk = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk * 4 >= n) return;
    const Philox4 w = philox4x32_10(offset + (uint64_t)blk, seed);
    float v[4];
    if (kind == 0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float r = sqrtf(-2.0f * logf(unit_open(w.x[2 * p])));
            float sn, cs;
            sincosf(6.283185307179586f * unit_open(w.x[2 * p + 1]), &sn, &cs);
            v[2 * p] = r * cs;
            v[2 * p + 1] = r * sn;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (w.x[k] >> 31) ? 2.0f : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (blk * 4 + k < n) out[blk * 4 + k] = v[k];
}

}  // namespace

int philox_fill(tts_hip_engine* e, float* out, long long n, uint64_t seed, uint64_t offset, int kind, hipStream_t st) {
    if (n <= 0) return TTS_HIP_OK;
    const long long blocks = (n + 3) / 4;
    hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, st, out, n, seed, offset, kind);
    HIPCHK(e, hipGetLastError());
    return TTS_HIP_OK;
}

// Box probe (bench.py): what the fp32 matrix pipe of THIS device sustains right now on a bare v_mfma_f32_32x32x2_f32 loop with
// the register traffic of the WN GEMMs (8 independent accumulators per wave, 2 waves per SIMD, every CU; scripts/micro/
// mfma_f32_rate.cpp is the stand-alone version: 155.3 TFLOP/s at 2.398 GHz on the round-3 boxes), and the shader clock it
// holds meanwhile.  Boxes of one pool differ by up to ~9 %: the probe puts a run's numbers in context.
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void mfma_probe_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            unsigned long long* __restrict__ clk, int iters) {
    probe_f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(threadIdx.x * 16 + i) & 4095];
        b[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + i) & 7], b[(k + 3 * i) & 7], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;                                // shader clocks
        clk[1] = r1 - r0;                                // 100 MHz ticks
    }
}

// ------------------------------------------------------------------------------------------- C ABI
extern "C" {

int tts_hip_abi_version(void) { return 11; }

int tts_hip_create(int device, tts_hip_engine** out) {
    if (!out) return TTS_HIP_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TTS_HIP_EHIP;
    if (device < 0 || device >= n) return TTS_HIP_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return TTS_HIP_EHIP;
    tts_hip_engine* e = new (std::nothrow) tts_hip_engine();
    if (!e) return TTS_HIP_ENOMEM;
    e->device = device;
    (void)hipDeviceGetAttribute(&e->n_cu, hipDeviceAttributeMultiprocessorCount, device);
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
        delete e;
        return TTS_HIP_EHIP;
    }
    *out = e;
    return TTS_HIP_OK;
}

int tts_hip_destroy(tts_hip_engine* e) {
    if (!e) return TTS_HIP_EINVAL;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    timing_collect(e);
    for (auto& t : e->ev_pool) {
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
    }
    waveglow_free(e);
    tacotron2_free(e);
    melstft_free(e);
    (void)hipStreamDestroy(e->stream);
    delete e;
    return TTS_HIP_OK;
}

const char* tts_hip_last_error(const tts_hip_engine* e) { return e ? e->err.c_str() : "null engine"; }

int tts_hip_set_tensor(tts_hip_engine* e, const char* name, const float* data, const int64_t* dims, int ndim) {
    if (!e || !name || !data || !dims || ndim <= 0 || ndim > 8) return set_err(e, TTS_HIP_EINVAL, "set_tensor: bad argument");
    try {
        HostTensor t;
        t.dims.assign(dims, dims + ndim);
        const size_t n = checked_numel(t.dims);
        if (!n) return set_err(e, TTS_HIP_EINVAL, "set_tensor(%s): non-positive or oversized dim", name);
        t.data.assign(data, data + n);
        e->host[name] = std::move(t);
    } catch (const std::exception& ex) {                       // bad_alloc / length_error must not cross the C boundary
        return set_err(e, TTS_HIP_ENOMEM, "set_tensor(%s): %s", name, ex.what());
    }
    return TTS_HIP_OK;
}

int tts_hip_load_weights(tts_hip_engine* e, const char* path) {
    if (!e || !path) return TTS_HIP_EINVAL;
    std::string err;
    const int rc = parse_ttsw(path, &e->host, &err);
    if (rc) return set_err(e, rc, "%s", err.c_str());
    return TTS_HIP_OK;
}

int tts_hip_check_weights_file(const char* path, char* errbuf, int errbuf_len) {
    if (!path) return TTS_HIP_EINVAL;
    std::string err;
    const int rc = parse_ttsw(path, nullptr, &err);
    if (errbuf && errbuf_len > 0) snprintf(errbuf, (size_t)errbuf_len, "%s", err.c_str());
    return rc;
}

static int finalize_impl(tts_hip_engine* e);

int tts_hip_finalize(tts_hip_engine* e) {
    if (!e) return TTS_HIP_EINVAL;
    try {
        return finalize_impl(e);
    } catch (const std::exception& ex) {
        return set_err(e, TTS_HIP_ENOMEM, "finalize: %s", ex.what());
    }
}

static int finalize_impl(tts_hip_engine* e) {
    HIPCHK(e, hipSetDevice(e->device));
    int rc;
    bool has_wg = false, has_taco = false;
    for (auto& kv : e->host) {
        if (kv.first.rfind("waveglow/", 0) == 0) has_wg = true;
        if (kv.first.rfind("tacotron2/", 0) == 0) has_taco = true;
    }
    if (has_wg) {
        if ((rc = waveglow_finalize(e))) return rc;
        // the packed device copies are all that is needed from here on: drop 1 GB of host staging
        for (auto it = e->host.begin(); it != e->host.end();)
            it = (it->first.rfind("waveglow/", 0) == 0) ? e->host.erase(it) : std::next(it);
    }
    if (has_taco) {
        if ((rc = tacotron2_finalize(e))) return rc;
    }
    if (!e->stft.ready) {
        if ((rc = melstft_finalize(e))) return rc;
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return TTS_HIP_OK;
}

int tts_hip_has_model(const tts_hip_engine* e, const char* model) {
    if (!e || !model) return 0;
    if (!strcmp(model, "waveglow")) return e->wg.ready;
    if (!strcmp(model, "tacotron2")) return e->taco.ready;
    if (!strcmp(model, "mel_stft")) return e->stft.ready;
    return 0;
}

static int waveglow_infer_impl(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma,
                               float* audio, int mem, int precision) {
    if (!e) return TTS_HIP_EINVAL;
    if (!e->wg.ready) return set_err(e, TTS_HIP_ENOTREADY, "waveglow weights not finalized");
    if (!mel || !audio || B <= 0 || T <= 0) return set_err(e, TTS_HIP_EINVAL, "waveglow_infer: bad argument");
    if ((long long)B * T * 32 > (1ll << 30)) return set_err(e, TTS_HIP_EINVAL, "waveglow_infer: B*T too large");
    HIPCHK(e, hipSetDevice(e->device));
    const size_t n_mel = (size_t)B * T * 80, n_z = (size_t)B * T * 32 * 8, n_out = (size_t)B * T * 256;
    const float* d_mel = mel;
    const float* d_z = z;
    float* d_out = audio;
    if (mem == TTS_HIP_MEM_HOST) {
        HIPCHK(e, e->wg.io_mel.ensure(n_mel * 4));
        HIPCHK(e, e->wg.io_out.ensure(n_out * 4));
        HIPCHK(e, hipMemcpyAsync(e->wg.io_mel.p, mel, n_mel * 4, hipMemcpyHostToDevice, e->stream));
        d_mel = e->wg.io_mel.f();
        d_out = e->wg.io_out.f();
        if (z) {
            HIPCHK(e, e->wg.io_z.ensure(n_z * 4));
            HIPCHK(e, hipMemcpyAsync(e->wg.io_z.p, z, n_z * 4, hipMemcpyHostToDevice, e->stream));
            d_z = e->wg.io_z.f();
        }
    } else if (mem != TTS_HIP_MEM_DEVICE) {
        return set_err(e, TTS_HIP_EINVAL, "waveglow_infer: bad mem kind %d", mem);
    }
    // One run addresses its activations with 31-bit byte offsets (<= ~32 k frames).  Utterances are independent, so a
    // larger batch is processed in slices of whole utterances; a single utterance above the limit is refused (the Python
    // wrapper's windowed inference, models/tts/waveglow.py:114-142, is the reference's own answer to long mels).
    constexpr int kMaxFramesPerRun = 31744;
    if (T > kMaxFramesPerRun)
        return set_err(e, TTS_HIP_EINVAL, "waveglow_infer: T = %d frames exceeds one run's limit (%d); use windowed inference",
                       T, kMaxFramesPerRun);
    const int chunkB = kMaxFramesPerRun / T;
    for (int b0 = 0; b0 < B; b0 += chunkB) {
        const int nb = B - b0 < chunkB ? B - b0 : chunkB;
        int rc = waveglow_run(e, d_mel + (size_t)b0 * T * 80, nb, T, d_z ? d_z + (size_t)b0 * T * 32 * 8 : nullptr, sigma,
                              d_out + (size_t)b0 * T * 256, precision);
        if (rc) return rc;
    }
    if (mem == TTS_HIP_MEM_HOST)
        HIPCHK(e, hipMemcpyAsync(audio, d_out, n_out * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return TTS_HIP_OK;
}

int tts_hip_waveglow_infer(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma,
                           float* audio, int mem) {
    return waveglow_infer_impl(e, mel, B, T, z, sigma, audio, mem, 0);
}

int tts_hip_waveglow_infer_f16(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma,
                               float* audio, int mem) {
    return waveglow_infer_impl(e, mel, B, T, z, sigma, audio, mem, 1);
}

int tts_hip_waveglow_infer_f16x3(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma,
                                 float* audio, int mem) {
    return waveglow_infer_impl(e, mel, B, T, z, sigma, audio, mem, 2);
}

// Test hook: the gated activations of one WN layer (before the res/skip and `end` convolutions), natural position order.
int tts_hip_waveglow_probe_acts(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma, int flow,
                                int layer, float* acts, int mem) {
    if (!e) return TTS_HIP_EINVAL;
    if (!e->wg.ready) return set_err(e, TTS_HIP_ENOTREADY, "waveglow weights not finalized");
    if (!mel || !acts || B <= 0 || T <= 0 || flow < 0 || flow > 11 || layer < 0 || layer > 7 || (long long)B * T > 31744)
        return set_err(e, TTS_HIP_EINVAL, "waveglow_probe_acts: bad argument");
    if (mem != TTS_HIP_MEM_HOST && mem != TTS_HIP_MEM_DEVICE) return set_err(e, TTS_HIP_EINVAL, "waveglow_probe_acts: bad mem kind %d", mem);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t n_mel = (size_t)B * T * 80, n_z = (size_t)B * T * 32 * 8, n_acts = (size_t)B * T * 32 * 512;
    const float* d_mel = mel;
    const float* d_z = z;
    struct Scratch : DevBuf {                                   // (DevBuf has no destructor: the engine's buffers live with the handle)
        ~Scratch() { release(); }
    } tmp;
    if (mem == TTS_HIP_MEM_HOST) {
        HIPCHK(e, e->wg.io_mel.ensure(n_mel * 4));
        HIPCHK(e, hipMemcpyAsync(e->wg.io_mel.p, mel, n_mel * 4, hipMemcpyHostToDevice, e->stream));
        d_mel = e->wg.io_mel.f();
        if (z) {
            HIPCHK(e, e->wg.io_z.ensure(n_z * 4));
            HIPCHK(e, hipMemcpyAsync(e->wg.io_z.p, z, n_z * 4, hipMemcpyHostToDevice, e->stream));
            d_z = e->wg.io_z.f();
        }
        HIPCHK(e, tmp.ensure(n_acts * 4));
    }
    HIPCHK(e, e->wg.io_out.ensure((size_t)B * T * 256 * 4));      // the run's audio argument (not reached before the stop)
    e->wg.probe_flow = flow;
    e->wg.probe_layer = layer;
    e->wg.probe_out = mem == TTS_HIP_MEM_HOST ? tmp.f() : acts;
    int rc = waveglow_run(e, d_mel, B, T, d_z, sigma, e->wg.io_out.f(), 0);
    e->wg.probe_out = nullptr;
    e->wg.probe_flow = e->wg.probe_layer = -1;
    hipError_t herr = hipSuccess;
    if (!rc && mem == TTS_HIP_MEM_HOST) herr = hipMemcpyAsync(acts, tmp.p, n_acts * 4, hipMemcpyDeviceToHost, e->stream);
    const hipError_t serr = hipStreamSynchronize(e->stream);
    if (rc) return rc;
    HIPCHK(e, herr);
    HIPCHK(e, serr);
    return TTS_HIP_OK;
}

int tts_hip_random_fill(tts_hip_engine* e, int kind, uint64_t seed, uint64_t offset, float* out, int64_t n, void* stream) {
    if (!e) return TTS_HIP_EINVAL;
    if (!out || n < 0 || (kind != TTS_HIP_RANDOM_NORMAL && kind != TTS_HIP_RANDOM_PRENET_MASK))
        return set_err(e, TTS_HIP_EINVAL, "random_fill: bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    return philox_fill(e, out, (long long)n, seed, offset, kind, stream ? (hipStream_t)stream : e->stream);
}

// WaveGlow.infer with the noise drawn on the device (the reference's default: z = None, deterministic = False).
int tts_hip_waveglow_infer_seeded(tts_hip_engine* e, const float* mel, int B, int T, uint64_t seed, uint64_t offset,
                                  float sigma, float* audio, int precision, int mem) {
    if (!e) return TTS_HIP_EINVAL;
    if (precision < 0 || precision > 2) return set_err(e, TTS_HIP_EINVAL, "waveglow_infer_seeded: precision must be 0 (f32), 1 (f16) or 2 (f16x3)");
    if (B <= 0 || T <= 0 || (long long)B * T * 32 > (1ll << 30)) return set_err(e, TTS_HIP_EINVAL, "waveglow_infer_seeded: bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    const size_t n_z = (size_t)B * T * 32 * 8;
    HIPCHK(e, e->wg.io_zgen.ensure(n_z * 4));
    int rc = philox_fill(e, e->wg.io_zgen.f(), (long long)n_z, seed, offset, TTS_HIP_RANDOM_NORMAL, e->stream);
    if (rc) return rc;
    // the generated noise is a device buffer whatever `mem` says about mel / audio
    if (mem == TTS_HIP_MEM_DEVICE) return waveglow_infer_impl(e, mel, B, T, e->wg.io_zgen.f(), sigma, audio, mem, precision);
    if (mem != TTS_HIP_MEM_HOST) return set_err(e, TTS_HIP_EINVAL, "waveglow_infer_seeded: bad mem kind %d", mem);
    const size_t n_mel = (size_t)B * T * 80, n_out = (size_t)B * T * 256;
    HIPCHK(e, e->wg.io_mel.ensure(n_mel * 4));
    HIPCHK(e, e->wg.io_out.ensure(n_out * 4));
    HIPCHK(e, hipMemcpyAsync(e->wg.io_mel.p, mel, n_mel * 4, hipMemcpyHostToDevice, e->stream));
    rc = waveglow_infer_impl(e, e->wg.io_mel.f(), B, T, e->wg.io_zgen.f(), sigma, e->wg.io_out.f(), TTS_HIP_MEM_DEVICE, precision);
    if (rc) return rc;
    HIPCHK(e, hipMemcpyAsync(audio, e->wg.io_out.p, n_out * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return TTS_HIP_OK;
}

// Device-pointer variants on a caller stream: enqueue and return (no synchronization).  Same arithmetic as the calls above.
int tts_hip_waveglow_infer_async(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma,
                                 float* audio, int precision, void* stream) {
    if (!e) return TTS_HIP_EINVAL;
    if (precision < 0 || precision > 2) return set_err(e, TTS_HIP_EINVAL, "waveglow_infer_async: precision must be 0 (f32), 1 (f16) or 2 (f16x3)");
    if (!e->wg.ready) return set_err(e, TTS_HIP_ENOTREADY, "waveglow weights not finalized");
    if (!mel || !audio || B <= 0 || T <= 0) return set_err(e, TTS_HIP_EINVAL, "waveglow_infer_async: bad argument");
    constexpr int kMaxFramesPerRun = 31744;
    if (T > kMaxFramesPerRun)
        return set_err(e, TTS_HIP_EINVAL, "waveglow_infer_async: T = %d frames exceeds one run's limit (%d); use windowed inference",
                       T, kMaxFramesPerRun);
    HIPCHK(e, hipSetDevice(e->device));
    StreamScope scope(e, stream);
    const int chunkB = kMaxFramesPerRun / T;
    for (int b0 = 0; b0 < B; b0 += chunkB) {
        const int nb = B - b0 < chunkB ? B - b0 : chunkB;
        int rc = waveglow_run(e, mel + (size_t)b0 * T * 80, nb, T, z ? z + (size_t)b0 * T * 32 * 8 : nullptr, sigma,
                              audio + (size_t)b0 * T * 256, precision);
        if (rc) return rc;
    }
    return TTS_HIP_OK;
}

int tts_hip_mel_stft_async(tts_hip_engine* e, const float* audio, int B, int N, float* mel, void* stream) {
    if (!e) return TTS_HIP_EINVAL;
    if (!e->stft.ready) return set_err(e, TTS_HIP_ENOTREADY, "mel_stft not finalized");
    if (!audio || !mel || B <= 0 || N < 1024) return set_err(e, TTS_HIP_EINVAL, "mel_stft_async: bad argument (N >= 1024)");
    HIPCHK(e, hipSetDevice(e->device));
    StreamScope scope(e, stream);
    return melstft_run(e, audio, B, N, mel);
}

int tts_hip_mel_stft(tts_hip_engine* e, const float* audio, int B, int N, float* mel, int mem) {
    if (!e) return TTS_HIP_EINVAL;
    if (!e->stft.ready) return set_err(e, TTS_HIP_ENOTREADY, "mel_stft not finalized");
    if (!audio || !mel || B <= 0 || N < 1024) return set_err(e, TTS_HIP_EINVAL, "mel_stft: bad argument (N >= 1024)");
    HIPCHK(e, hipSetDevice(e->device));
    const size_t n_in = (size_t)B * N, n_out = (size_t)B * (N / 256 + 1) * 80;
    const float* d_in = audio;
    float* d_out = mel;
    if (mem == TTS_HIP_MEM_HOST) {
        HIPCHK(e, e->stft.io_in.ensure(n_in * 4));
        HIPCHK(e, e->stft.io_out.ensure(n_out * 4));
        HIPCHK(e, hipMemcpyAsync(e->stft.io_in.p, audio, n_in * 4, hipMemcpyHostToDevice, e->stream));
        d_in = e->stft.io_in.f();
        d_out = e->stft.io_out.f();
    } else if (mem != TTS_HIP_MEM_DEVICE) {
        return set_err(e, TTS_HIP_EINVAL, "mel_stft: bad mem kind %d", mem);
    }
    int rc = melstft_run(e, d_in, B, N, d_out);
    if (rc) return rc;
    if (mem == TTS_HIP_MEM_HOST) HIPCHK(e, hipMemcpyAsync(mel, d_out, n_out * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return TTS_HIP_OK;
}

int tts_hip_kernel_timing(tts_hip_engine* e, int enable) {
    if (!e) return TTS_HIP_EINVAL;
    timing_collect(e);
    e->timing = enable != 0;
    for (int i = 0; i < 4; ++i) {
        e->time_sum_us[i] = 0;
        e->time_cnt[i] = 0;
    }
    return TTS_HIP_OK;
}

int tts_hip_kernel_time_us(tts_hip_engine* e, int kind, double* avg_us, int64_t* launches) {
    if (!e || kind < 0 || kind >= 4) return TTS_HIP_EINVAL;
    timing_collect(e);
    if (avg_us) *avg_us = e->time_cnt[kind] ? e->time_sum_us[kind] / (double)e->time_cnt[kind] : 0.0;
    if (launches) *launches = e->time_cnt[kind];
    return TTS_HIP_OK;
}

int tts_hip_set_decoder_mode(tts_hip_engine* e, int mode) {
    if (!e || mode < 0 || mode > 3) return set_err(e, TTS_HIP_EINVAL, "set_decoder_mode: mode must be 0 (graph), 1 (persistent), 2 (fused) or 3 (auto)");
    e->taco.persist_mode = mode;
    return TTS_HIP_OK;
}

int tts_hip_last_decoder_mode(const tts_hip_engine* e) { return e ? e->taco.last_path : -1; }

int tts_hip_set_waveglow_form(tts_hip_engine* e, int form) {
    if (!e || form < 0 || form > 3)
        return set_err(e, TTS_HIP_EINVAL, "set_waveglow_form: form must be 0 (direct), 1 (Winograd when the call shape allows it) or a measurement form (2, 3)");
    e->wg.form_mode = form;
    return TTS_HIP_OK;
}

int tts_hip_last_waveglow_form(const tts_hip_engine* e) { return e ? e->wg.last_form : -1; }

int tts_hip_probe_mfma_f32(tts_hip_engine* e, double* tflops, double* shader_clock_ghz) {
    if (!e) return TTS_HIP_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    const int ncu = e->n_cu > 0 ? e->n_cu : 256, blocks = 2 * ncu, iters = 6000;
    DevBuf in, out, clk;
    struct Free {
        DevBuf &a, &b, &c;
        ~Free() { a.release(); b.release(); c.release(); }
    } guard{in, out, clk};
    HIPCHK(e, in.ensure(4096 * 4));
    HIPCHK(e, out.ensure((size_t)blocks * 256 * 4));
    HIPCHK(e, clk.ensure(16));
    std::vector<float> h(4096);
    uint32_t st = 12345u;
    for (auto& v : h) {                                  // N(0, 1)-like operands: the clock the part holds depends on the data
        float u = 0.f;
        for (int i = 0; i < 12; ++i) {
            st = st * 1664525u + 1013904223u;
            u += (float)(st >> 8) * (1.0f / 16777216.0f);
        }
        v = u - 6.f;
    }
    HIPCHK(e, hipMemcpyAsync(in.p, h.data(), 4096 * 4, hipMemcpyHostToDevice, e->stream));
    hipEvent_t e0, e1;
    HIPCHK(e, hipEventCreate(&e0));
    HIPCHK(e, hipEventCreate(&e1));
    double best = 0.0, ghz = 0.0;
    hipError_t err = hipSuccess;
    for (int rep = 0; rep < 3 && err == hipSuccess; ++rep) {
        (void)hipEventRecord(e0, e->stream);
        hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, e->stream, in.f(), out.f(), (unsigned long long*)clk.p, iters);
        (void)hipEventRecord(e1, e->stream);
        err = hipEventSynchronize(e1);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2] = {0, 1};
        if (err == hipSuccess) err = hipMemcpy(c, clk.p, 16, hipMemcpyDeviceToHost);
        const double flop = (double)blocks * 4 * iters * 64.0 * 4096.0;      // 4 waves x iters x 64 MFMAs x (32 x 32 x 2 x 2) FLOP
        if (err == hipSuccess && ms > 0.f && flop / ms / 1e9 > best) {
            best = flop / ms / 1e9;
            ghz = c[1] ? (double)c[0] / ((double)c[1] * 10.0) : 0.0;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(e, err);
    if (tflops) *tflops = best;
    if (shader_clock_ghz) *shader_clock_ghz = ghz;
    return TTS_HIP_OK;
}

int tts_hip_synchronize(tts_hip_engine* e) {
    if (!e) return TTS_HIP_EINVAL;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return TTS_HIP_OK;
}

}  // extern "C"
