/*
 * tts_hip.h -- C ABI of the MI355X (gfx950) Tacotron2 + WaveGlow + mel-STFT inference engine.
 *
 * This is the drop-in boundary for the reference's runtime seam: `BaseModel(runtime=...)`
 * (/root/reference/models/interfaces/base_model.py:139-209) hands `compiled_infer` calls
 * (base_model.py:367-375) to a `Runtime` object (utils/keras/runtimes/runtime.py:19-41) registered in
 * `_runtimes` (utils/keras/runtimes/__init__.py:39-45).  The Python class `text_to_speech_amd.runtime.HipRuntime`
 * is that object; it binds the functions below with ctypes.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success, a negative TTS_HIP_E* code otherwise; `tts_hip_last_error` gives the text;
 *   - all tensors are dense, row-major, float32 (tokens/lengths int32), channels-last like the reference ([B, T, C]);
 *   - `mem` says where caller buffers live: TTS_HIP_MEM_HOST (pageable/pinned host memory) or TTS_HIP_MEM_DEVICE
 *     (pointers into the engine's GPU, e.g. torch tensors' data_ptr()); the engine never keeps caller pointers;
 *   - one HIP stream per handle; calls on one handle are serialised by the caller (the reference calls
 *     `infer` sequentially from one thread, base_model.py:702); different handles are independent;
 *   - no global state besides the handle.
 */
#ifndef TTS_HIP_H_
#define TTS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tts_hip_engine tts_hip_engine;

enum {
    TTS_HIP_OK = 0,
    TTS_HIP_EINVAL = -1,   /* bad argument / shape */
    TTS_HIP_ENOTREADY = -2,/* weights missing or not finalized */
    TTS_HIP_EHIP = -3,     /* HIP runtime error (text in last_error) */
    TTS_HIP_EIO = -4,      /* weight file error */
    TTS_HIP_ENOMEM = -5
};

enum { TTS_HIP_MEM_HOST = 0, TTS_HIP_MEM_DEVICE = 1 };

/* ---- lifetime ---------------------------------------------------------------------------------------------------
 * Replaces Runtime.load_engine(path) / Runtime.__init__ (runtimes/runtime.py:22-29).                               */
int tts_hip_create(int device, tts_hip_engine** out);
int tts_hip_destroy(tts_hip_engine* e);
const char* tts_hip_last_error(const tts_hip_engine* e);
/* ABI version of this header (bumped on any signature change). */
int tts_hip_abi_version(void);

/* ---- weights ----------------------------------------------------------------------------------------------------
 * Replaces BaseModel._restore_model -> CheckpointManager.load (base_model.py:760-783,
 * custom_train_objects/checkpoint_manager.py:169-215).  Tensor names and Keras layouts: text_to_speech_amd/weights.py.
 * `tts_hip_set_tensor` copies `data` (host float32) into the engine; `tts_hip_load_weights` reads a TTSW file and
 * calls it per tensor.  `tts_hip_finalize` builds the derived device buffers (transposed / permuted kernels, folded
 * batch-norm, inverted 1x1 matrices -- the analogue of WaveGlow.set_weights -> build_inverse,
 * waveglow_arch.py:308-310) for every model whose tensors are complete.  `speaker_embedding_dim` is 0 or 256.        */
int tts_hip_set_tensor(tts_hip_engine* e, const char* name, const float* data, const int64_t* dims, int ndim);
int tts_hip_load_weights(tts_hip_engine* e, const char* ttsw_path);
int tts_hip_finalize(tts_hip_engine* e);
/* Validates the container structure of a TTSW file (magic, version, entry table, dims, payload ranges against the file
 * size) without an engine or a GPU; 0 if `tts_hip_load_weights` would accept it, else TTS_HIP_EIO / TTS_HIP_ENOMEM with the
 * reason in `errbuf` (may be NULL).  The reference's loader trusts its checkpoint files
 * (custom_train_objects/checkpoint_manager.py:169-215); a C loader cannot.                                           */
int tts_hip_check_weights_file(const char* ttsw_path, char* errbuf, int errbuf_len);
/* 1 if the model ("waveglow" | "tacotron2" | "mel_stft") is ready to run, else 0. */
int tts_hip_has_model(const tts_hip_engine* e, const char* model);

/* ---- WaveGlow.infer  (architectures/waveglow_arch.py:244-306; called at models/tts/waveglow.py:82,96,104,112,128)
 * mel   [B, T, 80]
 * z     NULL (=> zeros: the reference's deterministic=True) or [B, T*32, 8] noise, consumed in the reference's order
 *       (channels 0..3 initial audio, 4..5 early output after flow 8, 6..7 after flow 4)
 * audio [B, T*256] out                                                                                              */
int tts_hip_waveglow_infer(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma,
                           float* audio, int mem);
/* Same contract with fp16 GEMM operands (the reference's Keras mixed_float16 policy, utils/keras/gpu.py:32-34; BASELINE
 * configs 3 and 5): activations, mel and weights are fp16 in HBM, accumulation and epilogue math are fp32, the residual
 * stream and the flow state keep fp32 master copies; inputs / outputs stay float32.  The fp16 operands are derived from
 * the finalized fp32 weights on first use.                                                                          */
int tts_hip_waveglow_infer_f16(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma,
                               float* audio, int mem);

/* Same contract in split-fp16 arithmetic: every GEMM operand (activations, mel, weights) is held as two fp16 planes
 * hi = fp16(v), lo = fp16(v - hi) (~22 significant bits) and a product is the three MFMAs hi*hi + hi*lo + lo*hi
 * accumulated in fp32 -- the "3x" emulation of fp32 GEMM on half-precision matrix cores: fp32-class results (waveform RMS
 * error ~1e-6 against the fp32 oracle) at 3/16 of the fp32 MFMA cost.  Everything outside the GEMM operands is fp32.      */
int tts_hip_waveglow_infer_f16x3(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma,
                                 float* audio, int mem);

/* ---- device-side sampling
 * The reference draws WaveGlow's noise and the prenet dropout inside `infer`, on the device
 * (architectures/waveglow_arch.py:272-274,299-302: `keras.random.normal`; tacotron2_arch.py:197-201: dropout p = 0.5).
 * `tts_hip_random_fill` writes n floats to a DEVICE buffer `out` on `stream` (NULL = the handle's stream) without
 * synchronizing; kind TTS_HIP_RANDOM_NORMAL: N(0, 1); kind TTS_HIP_RANDOM_PRENET_MASK: 2.0 with probability 0.5 else 0.0
 * (the multiplicative masks `tts_hip_tacotron2_infer` takes).  Generator: Philox4x32-10, key = seed, block counter =
 * offset + i / 4, element i = word i % 4; normals by Box-Muller on word pairs (see csrc/engine.hip; restated in
 * oracle/philox_ref.py).  The same (seed, offset) always gives the same values; consecutive calls should advance `offset`
 * by ceil(n / 4).
 * `tts_hip_waveglow_infer_seeded` = WaveGlow.infer(mel, z=None, deterministic=False): z [B, T*32, 8] is generated on the
 * device from (seed, offset) and never crosses PCIe.  precision: 0 f32, 1 f16, 2 f16x3.                               */
enum { TTS_HIP_RANDOM_NORMAL = 0, TTS_HIP_RANDOM_PRENET_MASK = 1 };
int tts_hip_random_fill(tts_hip_engine* e, int kind, uint64_t seed, uint64_t offset, float* out, int64_t n, void* stream);
int tts_hip_waveglow_infer_seeded(tts_hip_engine* e, const float* mel, int B, int T, uint64_t seed, uint64_t offset,
                                  float sigma, float* audio, int precision, int mem);

/* ---- Tacotron2.infer  (architectures/tacotron2_arch.py:866-925; called at models/tts/tacotron2.py:162)
 * tokens        int32 [B, Tin], 0 = pad
 * speaker       NULL or [B, speaker_embedding_dim]
 * max_len       decoder steps allocated (the caller resolves the reference's float `max_length`, :886-892)
 * early_stop    1: stop when every row has fired its stop token (:625-627); 0: run max_len steps
 * prenet_masks  NULL (=> deterministic prenet) or [B, max_len, 2, 256] multiplicative dropout masks
 * win_len/win_offset  attention window (:630-638); win_len <= 0 disables it
 * outputs (any may be NULL): mel [B, max_len, 80], decoder_output [B, max_len, 80], stop_tokens [B, max_len],
 *                            attention [B, max_len, Tin], lengths int32 [B]; *steps_run = loop iterations executed    */
int tts_hip_tacotron2_infer(tts_hip_engine* e, const int32_t* tokens, int B, int Tin, const float* speaker,
                            int max_len, int early_stop, const float* prenet_masks, int win_len, int win_offset,
                            float* mel, float* decoder_output, float* stop_tokens, float* attention,
                            int32_t* lengths, int32_t* steps_run, int mem);

/* Same contract with the two decoder LSTM weight matrices (98 % of the bytes a decoder step streams) held in fp16
 * (BASELINE configs 3 and 5: "fp16 weights, fp32 accumulate"); inputs, recurrent state, accumulation, attention, prenet,
 * projections, encoder and postnet stay float32.  The fp16 copies are derived from the finalized weights on first use. */
int tts_hip_tacotron2_infer_f16(tts_hip_engine* e, const int32_t* tokens, int B, int Tin, const float* speaker,
                                int max_len, int early_stop, const float* prenet_masks, int win_len, int win_offset,
                                float* mel, float* decoder_output, float* stop_tokens, float* attention,
                                int32_t* lengths, int32_t* steps_run, int mem);

/* ---- stream-ordered variants ("... , hipStream_t" entry points) ------------------------------------
 * The calls above run on the handle's own stream and return when it has drained.  These take a caller `stream`
 * (a hipStream_t passed as void*; NULL = the handle's stream), only accept device pointers, enqueue their work and return
 * WITHOUT synchronizing, so a caller can queue transfers, several calls and its own kernels back to back.  A handle still
 * has one workspace per model: two calls on the same handle must be ordered (same stream, or an event between streams).
 * precision: 0 = f32, 1 = f16 operands, 2 = f16x3 (WaveGlow); 0 = f32, 1 = fp16 LSTM weights (Tacotron2).
 *
 * Tacotron2 in two calls -- Tacotron2Encoder (tacotron2_arch.py:235-333, once per batch) and the decoder loop + postnet
 * (:609-749, :915-917):  `encode` is asynchronous and returns an opaque encoded batch (its own device buffer; free it with
 * tts_hip_encoded_free; several may be alive, e.g. the next sentence's encoder running ahead); `decode` may be called any
 * number of times on it (the retry loop of models/tts/tacotron2.py:160-179 re-runs only the decoder with new dropout
 * masks) and synchronizes `stream` before it returns, because the loop's exit is data dependent and `steps_run` is a host
 * value.  tts_hip_tacotron2_infer == encode + decode.                                                                  */
typedef struct tts_hip_encoded tts_hip_encoded;
int tts_hip_waveglow_infer_async(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma,
                                 float* audio, int precision, void* stream);
int tts_hip_mel_stft_async(tts_hip_engine* e, const float* audio, int B, int N, float* mel, void* stream);
int tts_hip_tacotron2_encode(tts_hip_engine* e, const int32_t* tokens, int B, int Tin, const float* speaker, int mem,
                             void* stream, tts_hip_encoded** out);
int tts_hip_tacotron2_decode(tts_hip_engine* e, const tts_hip_encoded* encoded, int max_len, int early_stop,
                             const float* prenet_masks, int win_len, int win_offset, int precision, float* mel,
                             float* decoder_output, float* stop_tokens, float* attention, int32_t* lengths,
                             int32_t* steps_run, int mem, void* stream);
/* `decode` with the prenet dropout masks drawn on the device from (seed, offset) (tts_hip_random_fill, kind
 * TTS_HIP_RANDOM_PRENET_MASK, n = B * max_len * 512): the reference's default inference path keeps this dropout on
 * (tacotron2_arch.py:197-201), and a retry only needs another offset.                                                  */
int tts_hip_tacotron2_decode_seeded(tts_hip_engine* e, const tts_hip_encoded* encoded, int max_len, int early_stop,
                                    uint64_t seed, uint64_t offset, int win_len, int win_offset, int precision,
                                    float* mel, float* decoder_output, float* stop_tokens, float* attention,
                                    int32_t* lengths, int32_t* steps_run, int mem, void* stream);
/* Runs the encoder for another token batch INTO an existing encoded batch (its device buffer is reused and only grows):
 * what a caller that synthesizes sentence after sentence wants -- no hipMalloc / hipFree per sentence, and the decoder's
 * cached step graphs (keyed by the buffer) survive from one sentence to the next.  Asynchronous like `encode`.        */
int tts_hip_tacotron2_reencode(tts_hip_engine* e, tts_hip_encoded* encoded, const int32_t* tokens, int B, int Tin,
                               const float* speaker, int mem, void* stream);
int tts_hip_encoded_free(tts_hip_engine* e, tts_hip_encoded* encoded);

/* How the autoregressive loop (tacotron2_arch.py:710-735, K.while_loop) is executed.  mode 1: one persistent,
 * weight-stationary persistent kernel for the whole loop when the call shape allows it (batch <= 4, B * Tin small enough
 * for LDS, a device with >= 256 CUs that can host the whole grid); mode 2: the fused two-kernel step (batch <= 8, at most
 * 256 tokens; csrc/taco_fused.hip); mode 3 (default): persistent for 1 - 2 rows, fused for 3 - 8, whichever applies
 * otherwise; mode 0 -- and the fallback of every other mode -- one hipGraph of 7 kernels per decoder step.  All give the
 * same results up to fp32 re-association.                                                                              */
int tts_hip_set_decoder_mode(tts_hip_engine* e, int mode);
/* Which one the last tts_hip_tacotron2_infer* call on this handle used: 2 fused step, 1 persistent kernel, 0 per-step graph
 * (also after a fallback), -1 before the first call.                                                                          */
int tts_hip_last_decoder_mode(const tts_hip_engine* e);

/* How the fp32 WaveGlow path evaluates the k = 3 dilated convolution of WN layers 1 .. 7 (waveglow_arch.py:117-127).
 * form 1 (default): Winograd minimal filtering along the tap axis for calls of 144 frames or more (any utterance length;
 * csrc/wn_wino.hip) -- F(4,3), six products per four outputs: K per output ~800 + 320 instead of 1536 + 320, fp32 operands
 * and accumulators, ONE kernel per layer (input transform in the operand reads, the six products as accumulator sets of one
 * block, output transform + gate in the epilogue); results within fp32 rounding of the direct form (6.0e-7 vs 5.0e-7
 * waveform RMS error against the oracle);
 * form 0: always the direct form;
 * forms 2, 3 (measurement only, same results as form 1): the three-pass form of round 3 (pre-pass, per-product GEMM, combine
 * pass) and the fused GEMM behind the pre-pass.                                                                         */
int tts_hip_set_waveglow_form(tts_hip_engine* e, int form);
/* Which one the last tts_hip_waveglow_infer* call on this handle used: 1 Winograd, 0 direct, -1 before the first call.  */
int tts_hip_last_waveglow_form(const tts_hip_engine* e);

/* Test hook (used by tests/ only; no effect on later calls): runs the fp32 path of tts_hip_waveglow_infer -- in the form
 * selected by tts_hip_set_waveglow_form -- up to WN layer `layer` (0 .. 7) of flow `flow` (flows run 11 .. 0) and copies that
 * layer's gated activations tanh(.) * sigmoid(.) (waveglow_arch.py:19-24,117-127), i.e. the values BEFORE the res/skip and
 * `end` convolutions attenuate an error, to `acts` [B, T*32, 512] in the reference's position order.  B*T <= 31744.       */
int tts_hip_waveglow_probe_acts(tts_hip_engine* e, const float* mel, int B, int T, const float* z, float sigma, int flow,
                                int layer, float* acts, int mem);

/* ---- TacotronSTFT.mel_spectrogram  (utils/audio/stft.py:242-274,306-314)
 * audio [B, N] (N >= 1024) -> mel [B, N/256 + 1, 80]                                                                */
int tts_hip_mel_stft(tts_hip_engine* e, const float* audio, int B, int N, float* mel, int mem);

/* ---- measurement hooks (used by bench.py; no effect on results) -------------------------------------------------
 * Average duration in microseconds of the dominant kernel's launches (HIP events on the engine's stream) since the
 * last reset, and how many launches were timed.  kind: 0 = WaveGlow WN in-layer GEMM (layers 1..7 of a flow: K = 2176),
 * 1 = WN residual GEMM, 2 = Tacotron2 decoder step, 3 = first WN layer of a flow (start conv composed: K = 688).  Timing is off unless enabled (events perturb nothing but cost a few us each).        */
int tts_hip_kernel_timing(tts_hip_engine* e, int enable);
/* Box probe: TFLOP/s a bare v_mfma_f32_32x32x2_f32 loop sustains on this device right now (every CU, 2 waves per SIMD, ~20 ms)
 * and the shader clock (GHz) it holds meanwhile -- what the fp32 MFMA roofline of THIS box is; boxes of one pool differ.    */
int tts_hip_probe_mfma_f32(tts_hip_engine* e, double* tflops, double* shader_clock_ghz);
int tts_hip_kernel_time_us(tts_hip_engine* e, int kind, double* avg_us, int64_t* launches);
int tts_hip_synchronize(tts_hip_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* TTS_HIP_H_ */
