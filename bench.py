#!/usr/bin/env python3
"""Headline benchmark: WaveGlow vocoding of precomputed 80x800 mels, batch 8, fp32, per MI355X (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Both forms work for any N: under torchrun (WORLD_SIZE set) the process is one rank; without it `--gpus N > 1` makes this
process a launcher that starts N rank processes itself (one per GPU, rendezvous on 127.0.0.1) BEFORE anything touches a
GPU, forwards rank 0's JSON line and exits with the worst child status.

A "step" is one pass of the hot path (tts_hip_waveglow_infer through the C ABI) over one batch of synthetic mels that is
already resident in HBM.  With N > 1 every rank vocodes its own batch of 8 utterances (utterances are independent:
no data-path collective, weak scaling); the timed region is bracketed by barrier + synchronize and the MAX over ranks
is reported.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- dominant kernel = the WN in-layer implicit GEMM (K = 3*512 + 640, N = 1024).  `achieved` = algorithmic
                  FLOPs per launch / average launch duration measured with HIP events on the engine's stream.
  cpu_baseline -- CPU restatements of the reference algorithm (the numpy oracle and the torch.nn.functional one -- NOT the
                  reference's TF2 path, which cannot run here) timed on bounded samples on rank 0 at N = 1: WaveGlow with
                  all host threads (`value`) and with one thread, Tacotron2 decode at batch 1 and 8.
  extra        -- the other numbers BASELINE.json's metric names (batch-1 WaveGlow, Tacotron2 mel-frames/s at batch 1 / 8,
                  fp16 modes, the configs[2] pipeline) with their own roofline fractions.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md "Peak FP32 (matrix)"
FP16_MFMA_PEAK_TFLOPS = 2500.0         # dense fp16 / bf16 matrix peak
HBM_PEAK_TBS = 8.0                     # HBM3E
BATCH, FRAMES = 8, 800                 # BASELINE.json configs[1]
SAMPLE_RATE = 22050
EXCHANGE_HOP_US_MICROBENCH = 1.2       # constant (microbenchmark result, scripts/micro/xcd_exchange.cpp), used by one derived key
DECODER_STEP_BYTES_F32 = 72.73e6       # all decoder-step weights once (enc 512; SURVEY.md section 8d / BASELINE.md section 2)
DECODER_STEP_BYTES_F16W = 72.73e6 - 0.5 * (29.36e6 + 41.94e6)   # the two LSTM matrices in fp16, the rest fp32


K_EXECUTED = 3 * 512 + 4 * 80      # dilated k3 conv taps + conditioning folded onto 4 mel frames
K_REFERENCE = 3 * 512 + 640        # the reference formulation: taps + 640-channel upsampled spectrogram


def wino_in_layer_flops(B: int, T: int) -> float:
    """FLOPs the Winograd form of the in-layer GEMM EXECUTES per launch, averaged over the 7 launches of a flow (one fused
    kernel per layer, csrc/wn_wino.hip, F(4,3)): six products on M / 4 group rows -- K = 512 taps + the product's conditioning
    chunks: 208 columns for products 0, 3, 4, 5 and 224 for products 1, 2 -- for the dilations 2, 4, 8 (groups of phases) and
    32, 64, 128 (groups of frames, group rows per phase padded to the 64-row tile); dilation 16 (two phases x two frames):
    four K = 512 + 320 products and two K = 512 products."""
    BT = B * T
    PR = (BT + 255) // 256 * 256                         # frame rows per phase block (256-row tiles)
    PRq = (B * ((T + 15) // 16 * 4) + 63) // 64 * 64     # group rows per phase, padded to the fused kernel's 64-row tile
    PRm = (B * ((T + 1) // 2) + 63) // 64 * 64
    k6 = 4 * (512 + 208) + 2 * (512 + 224)               # K summed over the six products
    phases = (8 * PR) * k6
    frames = (32 * PRq) * k6
    mixed = (16 * PRm) * (4 * (512 + 320) + 2 * 512)
    return 2.0 * 1024 * (3 * phases + 3 * frames + mixed) / 7.0


def wn_in_layer_flops(M: int, k: int = K_EXECUTED) -> float:
    """FLOPs of one WN in-layer GEMM launch over M positions (N = 1024 gate pre-activations)."""
    return 2.0 * M * k * 1024


def pmc_traffic_bytes(B, T):
    """HBM bytes per launch of the WN in-layer GEMM from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of this command, summarised in profiles/): 2 * FETCH_SIZE + WRITE_SIZE KB (gfx950 reports half the
    bytes of 16-B/lane reads -- MI355X_MICROARCH.md HBM section).  Only valid for the profiled workload (B 8, T 800)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_hbm_traffic_latest.json')
    if (B, T) != (BATCH, FRAMES) or not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        k = d['wn_in_layer']
        return (2.0 * k['FETCH_SIZE_KB_mean'] + k['WRITE_SIZE_KB_mean']) * 1024.0
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------- CPU baseline
def _host_threads():
    """(threads this process may run on, os.cpu_count() of the box)."""
    box = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:                                   # pragma: no cover
        usable = box
    return max(1, usable), box


def cpu_baseline(wg_weights, cfg, frames: int):
    """CPU legs (the oracle is only the thing being TIMED here, as the reported baseline -- never the product path).

    WaveGlow: `frames` mel frames at batch 1 through (a) the numpy oracle and (b) the independent torch.nn.functional
    restatement (oneDNN / MKL: the kernel class Keras-on-TF would call), both with every usable host thread, plus (c) a
    single-thread run of the faster one on frames / 10.  Tacotron2: 200 decoder steps (encoder and postnet included) at
    batch 1 and 8 through both restatements.  `value` = the best all-thread WaveGlow leg."""
    import torch
    from oracle import tacotron2_ref, torch_ref, waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import Tacotron2Config
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                   # pragma: no cover
        threadpool_limits = None
    usable, box = _host_threads()
    threads = min(usable, 64)
    legs = {}

    def wg_inputs(T):
        mel = np.random.default_rng(7).uniform(-11.5, 1.2, (1, T, 80)).astype(np.float32)
        z = np.random.default_rng(11).standard_normal((1, T * 32, 8)).astype(np.float32)
        return mel, z

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    def with_threads(n, fn):
        torch.set_num_threads(n)
        if threadpool_limits is not None:
            with threadpool_limits(limits=n):
                return timed(fn)
        return timed(fn)

    def median3(n_threads, fn):
        """one untimed warm-up run (allocator, thread pools, oneDNN primitive caches), then the median of three timed runs"""
        with_threads(n_threads, fn)
        runs = sorted(with_threads(n_threads, fn) for _ in range(3))
        return runs[1], runs

    mel, z = wg_inputs(frames)
    with torch.no_grad():
        np_frames = max(8, frames // 6)                 # the numpy leg is ~6x slower than the torch one: a sixth of the frames
        mel_n, z_n = wg_inputs(np_frames)
        dt_np_s, runs_np = median3(threads, lambda: waveglow_ref.infer(mel_n, wg_weights, cfg, z=z_n))
        dt_np = dt_np_s * frames / np_frames
        dt_th, runs_th = median3(threads, lambda: torch_ref.torch_waveglow(mel, wg_weights, cfg, z))
        legs['waveglow_numpy_all_threads'] = {'samples_per_s': frames * 256 / dt_np, 'threads': threads, 'frames': np_frames,
                                              'seconds': sum(runs_np), 'runs_s': runs_np}
        legs['waveglow_torch_all_threads'] = {'samples_per_s': frames * 256 / dt_th, 'threads': threads, 'frames': frames,
                                              'seconds': sum(runs_th), 'runs_s': runs_th}
        small = max(8, frames // 20)
        mel1, z1 = wg_inputs(small)
        use_torch = dt_th <= dt_np
        dt_1, runs_1 = median3(1, (lambda: torch_ref.torch_waveglow(mel1, wg_weights, cfg, z1)) if use_torch
                               else (lambda: waveglow_ref.infer(mel1, wg_weights, cfg, z=z1)))
        legs['waveglow_single_thread'] = {'samples_per_s': small * 256 / dt_1, 'threads': 1, 'frames': small,
                                          'seconds': sum(runs_1), 'runs_s': runs_1, 'impl': 'torch' if use_torch else 'numpy'}
        # Tacotron2: 100-token utterances padded to 128, 100 decoder steps, deterministic prenet
        tcfg = Tacotron2Config()
        tw = weights.synth_tacotron2(tcfg, seed=1234)
        steps = 100
        taco_threads = min(threads, 16)                 # GEMV-sized work: more threads only add synchronisation
        for B in (1, 8):
            tok = np.zeros((B, 128), np.int32)
            tok[:, :100] = np.random.default_rng(5).integers(1, 148, (B, 100))
            dt_n, runs_n = median3(taco_threads, lambda: tacotron2_ref.infer(tok, tw, tcfg, max_length=steps, early_stopping=False))
            dt_t, runs_t = median3(taco_threads, lambda: torch_ref.torch_tacotron2(tok, tw, tcfg, None, steps, None))
            legs[f'tacotron2_batch{B}'] = {'mel_frames_per_s_numpy': B * steps / dt_n, 'mel_frames_per_s_torch': B * steps / dt_t,
                                           'threads': taco_threads, 'decoder_steps': steps, 'seconds': sum(runs_n) + sum(runs_t)}
    torch.set_num_threads(threads)
    best = max(legs['waveglow_numpy_all_threads']['samples_per_s'], legs['waveglow_torch_all_threads']['samples_per_s'])
    which = 'torch.nn.functional (oneDNN/MKL)' if use_torch else 'numpy oracle (OpenBLAS)'
    total = sum(v['seconds'] for v in legs.values())
    return {
        'value': best, 'unit': 'audio samples/s', 'cores': threads, 'kind': 'port',
        'host_cpu_count': box, 'usable_cpus': usable,
        'sample': f'{which} CPU restatement, WaveGlow batch 1 x {frames} frames with {threads} threads (box: os.cpu_count() = '
                  f'{box}, {usable} usable by this process; thread count capped at 64: more only adds synchronisation at this '
                  f'size); every leg = one warm-up run + the median of three, {total:.1f} s of timed CPU work in total; a '
                  f'stand-in for the reference TF2 CPU path, which cannot be imported here',
        'legs': legs,
    }


# ------------------------------------------------------------------------------------------------ secondary metrics
def secondary_metrics(eng, dev, rank):
    """The other numbers BASELINE.json's metric names (batch 1 WaveGlow; Tacotron2 mel-frames/s at batch 1 and 8),
    measured after the headline region on the same engine.  Tacotron2: 100-token utterances padded to 128, 800 decoder
    steps with early stopping off, deterministic prenet, seeded synthetic weights; encoder + postnet included."""
    import torch
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import Tacotron2Config
    from text_to_speech_amd.engine import KERNEL_WN_IN
    out = {}
    mel1 = torch.from_numpy(np.random.default_rng(3).uniform(-11.5, 1.2, (1, FRAMES, 80)).astype(np.float32)).to(dev)
    z1 = torch.from_numpy(np.random.default_rng(4).standard_normal((1, FRAMES * 32, 8)).astype(np.float32)).to(dev)
    eng.waveglow_infer(mel1, z=z1)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.waveglow_infer(mel1, z=z1)
    dt = (time.perf_counter() - t0) / 3
    out['waveglow_batch1_samples_per_s'] = FRAMES * 256 / dt
    # fp16-operand mode on the headline shape (BASELINE.json configs 3 / 5 run the vocoder in fp16); NOT the headline
    # value, which stays exact fp32.  fp16 operands, fp32 accumulate: see DESIGN.md "fp16 mode".  Its waveform RMS error
    # against the fp32 oracle is 2e-4 (above the fp32 tolerance of 1e-4): stated next to every fp16 number.
    mel8 = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (BATCH, FRAMES, 80)).astype(np.float32)).to(dev)
    z8 = torch.from_numpy(np.random.default_rng(2).standard_normal((BATCH, FRAMES * 32, 8)).astype(np.float32)).to(dev)
    M = BATCH * FRAMES * 32
    for prec, mult in (('f16', 1), ('f16x3', 3)):
        # split-fp16 (f16x3): (hi, lo) fp16 operand planes, three MFMAs per product, fp32 accumulate -- fp32-class accuracy
        # (5e-7 waveform RMS error against the oracle, like the exact path) on the half-precision matrix cores.
        eng.waveglow_infer(mel8, z=z8, precision=prec)
        t0 = time.perf_counter()
        for _ in range(3):
            eng.waveglow_infer(mel8, z=z8, precision=prec)
        dt = (time.perf_counter() - t0) / 3
        out[f'waveglow_batch8_{prec}_samples_per_s'] = BATCH * FRAMES * 256 / dt
        out[f'waveglow_batch8_{prec}_ms_per_step'] = dt * 1e3
        eng.kernel_timing(True)                          # HIP events on the engine stream (perturbs the step: own run)
        eng.waveglow_infer(mel8, z=z8, precision=prec)
        us, n = eng.kernel_time_us(KERNEL_WN_IN)
        eng.kernel_timing(False)
        if n:
            tf = mult * wn_in_layer_flops(M) / (us * 1e-6) / 1e12
            out[f'waveglow_{prec}_wn_in_layer_us'] = us
            out[f'waveglow_{prec}_wn_in_layer_mfma_frac'] = tf / FP16_MFMA_PEAK_TFLOPS
            # SURVEY 8(d) asks for both fractions on the fp16 sweep: algorithmic bytes of one launch (x in + activations out
            # as fp16 planes, mel, weights) over the same launch time, against 8 TB/s.  MFMA-bound by construction.
            planes = 2 if prec == 'f16x3' else 1
            nbytes = planes * (M * (512 + 512) + BATCH * FRAMES * 80 + 1024 * (1536 + 32 * 320)) * 2.0
            out[f'waveglow_{prec}_wn_in_layer_hbm_frac'] = nbytes / (us * 1e-6) / 8.0e12
    # Drift monitor, measured here: waveform RMS difference of the two half-precision modes from the exact fp32 HIP path on one
    # 8 x 64-frame batch (HIP vs HIP -- the parity tests, not this, compare each mode with the oracle: fp32 and f16x3 hold
    # the 1e-4 tolerance, f16 does not and says so everywhere it is quoted).
    mel_s, z_s = mel8[:, :64].contiguous(), z8[:, :64 * 32].contiguous()
    exact = eng.waveglow_infer(mel_s, z=z_s, precision='f32')
    for prec in ('f16', 'f16x3'):
        d = eng.waveglow_infer(mel_s, z=z_s, precision=prec) - exact
        out[f'waveglow_{prec}_rms_diff_vs_fp32_hip_path'] = float(torch.sqrt(torch.mean(d * d)))
    del mel8, z8
    eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234))
    eng.finalize()
    # Three ways to run the loop: 'persistent' = one weight-stationary kernel for the whole utterance
    # (no weight streaming at all, the step is bound by six CU-to-CU exchange hops; taken for 1 - 2 rows), 'fused' = two kernels
    # per step with in-kernel exchanges, every LSTM weight streamed once per step beside the dependency chain (3 - 8 rows),
    # 'graph' = 7 kernels per step in a hipGraph (any shape; the fallback).  The default ('auto') is what the plain keys
    # report; `_graph` keys time the 7-kernel path.
    for B in (1, 2, 4, 8):
        tok = np.zeros((B, 128), np.int32)
        tok[:, :100] = np.random.default_rng(5).integers(1, 148, (B, 100))
        tok_d = torch.from_numpy(tok).to(dev)
        for mode in ('auto', 'graph'):
            eng.set_decoder_mode(mode)
            for prec, tag, nbytes in (('f32', '', DECODER_STEP_BYTES_F32), ('f16', '_f16w', DECODER_STEP_BYTES_F16W)):
                if B in (2, 4) and prec == 'f16':
                    continue
                eng.tacotron2_infer(tok_d, max_len=64, early_stopping=False, want_attention=False, precision=prec)
                ran = eng.last_decoder_mode
                key = f'tacotron2_batch{B}{tag}' + ('' if mode == 'auto' else '_graph')
                # the call's fixed part (encoder, postnet, transfers) is separated from the per-step cost with two lengths
                reps = 3
                times = {}
                for n_steps in (FRAMES // 2, FRAMES):
                    runs = []
                    for _ in range(reps):
                        t0 = time.perf_counter()
                        eng.tacotron2_infer(tok_d, max_len=n_steps, early_stopping=False, want_attention=False, precision=prec)
                        runs.append(time.perf_counter() - t0)
                    times[n_steps] = sorted(runs)[reps // 2]     # median of three: one disturbed call does not move the figure
                dt = times[FRAMES]
                step_us = 1e6 * (times[FRAMES] - times[FRAMES // 2]) / (FRAMES - FRAMES // 2)
                out[f'{key}_mel_frames_per_s'] = B * FRAMES / dt
                out[f'{key}_us_per_decoder_step'] = 1e6 * dt / FRAMES          # whole call / steps
                out[f'{key}_us_per_decoder_step_marginal'] = step_us            # loop only
                out[f'{key}_decoder_path'] = ran
                if ran in ('graph', 'fused'):
                    # streaming roofline: all step weights once per step from HBM
                    out[f'{key}_decoder_hbm_frac'] = nbytes / (step_us * 1e-6) / (HBM_PEAK_TBS * 1e12)
                else:
                    # weight-stationary: nothing is streamed.  A builder-defined bound, NOT a hardware peak: 6 exchange hops x the
                    # 1.2 us a single tagged CU-to-CU hop took in scripts/micro/xcd_exchange.cpp on an idle chip.
                    out[f'{key}_exchange_floor_frac_builder_bound'] = 6 * EXCHANGE_HOP_US_MICROBENCH / step_us
    eng.set_decoder_mode('auto')
    # BASELINE.json configs[2] shape: full text -> audio pipeline, batch 8, mixed token counts 50..200 padded to 256,
    # mel kept on the GPU between the two models, fp16 modes of both models (decoder LSTM weights fp16; WaveGlow GEMM
    # operands fp16; fp32 accumulation everywhere).
    # Synthetic weights never fire the stop token, so every row decodes max_len = 800 frames (fixed-length timing run).
    from text_to_speech_amd.pipeline import TTSPipeline
    lens = [50, 70, 90, 110, 130, 150, 170, 200]
    tok = np.zeros((8, 256), np.int32)
    for i, n in enumerate(lens):
        tok[i, :n] = np.random.default_rng(6 + i).integers(1, 148, n)
    pipe = TTSPipeline(eng, seed=0, vocoder_precision='f16', synthesizer_precision='f16')
    pipe.synthesize_tokens(tok, max_length=64, deterministic=True, early_stopping=False)
    t0 = time.perf_counter()
    audio, n_frames, _ = pipe.synthesize_tokens(tok, max_length=FRAMES, deterministic=True, early_stopping=False)
    dt = time.perf_counter() - t0
    secs = sum(len(a) for a in audio) / SAMPLE_RATE
    out['pipeline_batch8_f16_audio_seconds_per_s'] = secs / dt
    out['pipeline_batch8_f16_ms'] = dt * 1e3
    out.update(config5_streaming(eng))
    # mel-STFT (rows C1 / C2 of SURVEY 8a): 8 x 204 800 samples, DFT as a windowed real / imaginary basis product + mel + log.
    # Executed work as written by the reference: 2 * 1024 * 1026 + 2 * 513 * 80 FLOP per frame.
    wav = torch.from_numpy(np.random.default_rng(9).uniform(-0.5, 0.5, (BATCH, FRAMES * 256)).astype(np.float32)).to(dev)
    eng.mel_stft(wav)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.mel_stft(wav)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    n_fr = BATCH * (FRAMES + 1)
    out['mel_stft_batch8_ms'] = dt * 1e3
    out['mel_stft_audio_seconds_per_s'] = BATCH * FRAMES * 256 / SAMPLE_RATE / dt
    out['mel_stft_tflops_as_written'] = n_fr * (2.0 * 1024 * 1026 + 2.0 * 513 * 80) / dt / 1e12
    return out


def config5_streaming(eng):
    """BASELINE.json configs[4]: `stream()`-style long-form synthesis, 64 sentences one after the other (token counts cycling
    50 .. 200, the decoder on its default machine at batch 1, both models in their fp16 modes, noise and dropout off so that the
    two runs do the same work): sequential, and with Tacotron2(n + 1) overlapped with WaveGlow(n) on a second engine handle
    (`predict(overlap=True)`).  Synthetic weights never fire the stop token: every sentence decodes 4 frames per token."""
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import Tacotron2Config, WaveGlowConfig
    from text_to_speech_amd.engine import HipEngine
    from text_to_speech_amd.runtime import HipRuntime
    from text_to_speech_amd.tacotron2 import Tacotron2
    from text_to_speech_amd.waveglow import WaveGlow
    out = {}
    eng2 = HipEngine(eng.device)                                         # the overlapped run's vocoder lane (`eng` holds both models)
    try:
        eng2.load_state(weights.synth_waveglow(WaveGlowConfig(), seed=1234))
        eng2.finalize()
        model = Tacotron2(HipRuntime('t', model='tacotron2', engine=eng, seed=0, synthesizer_precision='f16'))
        voc_same = WaveGlow(HipRuntime('w', model='waveglow', engine=eng, seed=0, vocoder_precision='f16'))
        voc_own = WaveGlow(HipRuntime('w2', model='waveglow', engine=eng2, seed=0, vocoder_precision='f16'))
        rng = np.random.default_rng(0)
        letters = np.array(list('abcdefghijklmnopqrstuvwxyz     '))
        lens = [50, 70, 90, 110, 130, 150, 170, 200]
        texts = [''.join(rng.choice(letters, lens[i % 8])).strip() + f' {i}.' for i in range(64)]
        kw = dict(max_length=4., deterministic=True, save=False, return_results=False)
        model.predict(texts[:2], vocoder=voc_same, **kw)
        model.predict(texts[:2], vocoder=voc_own, overlap=True, **kw)
        for name, voc, ov in (('sequential', voc_same, False), ('overlapped', voc_own, True)):
            secs = []
            t0 = time.perf_counter()
            model.predict(texts, vocoder=voc, overlap=ov, callbacks=[lambda time, **_: secs.append(time)], **kw)
            dt = time.perf_counter() - t0
            out[f'config5_stream64_f16_{name}_ms'] = dt * 1e3
            out[f'config5_stream64_f16_{name}_x_realtime'] = sum(secs) / dt
        out['config5_stream64_audio_seconds'] = float(sum(secs))
        out['config5_stream64_decoder_path'] = eng.last_decoder_mode
    except Exception as exc:                                         # a secondary figure never costs the line
        out['config5_stream64_error'] = f'{type(exc).__name__}: {exc}'
    finally:
        eng2.close()
    return out


# ------------------------------------------------------------------------------------- BASELINE config 4 (sharded job)
CONFIG4_UTTERANCES, CONFIG4_FRAMES, CONFIG4_TIN = 32, 400, 256
CONFIG4_TIMEOUT_S = 240                     # watchdog of the config-4 job (a normal run takes a few seconds)


def config4_job(eng, dev, rank, world, reps=2):
    """BASELINE.json configs[3]: 32 SV2TTS utterances (256-d speaker embeddings, enc 768; token counts 50 .. 200 padded to
    256) live on rank 0; `distributed.synthesize_sharded` scatters tokens + embeddings over the ranks (longest first, round
    robin), every rank runs text -> mel -> audio on its share (`TTSPipeline.shard_fn`, mel kept on the GPU, noise and
    dropout off so that the job is the same on every run), and the waveforms are gathered back on rank 0 -- the two
    collectives the north star names, over RCCL.  Fixed total work (strong scaling): 32 x 400 frames = 148.6 s of audio.
    Synthetic weights never fire the stop token, so every row decodes all 400 frames.  Needs an initialised process group
    (world 1 included, so that N = 1 runs the same code)."""
    import torch
    import torch.distributed as dist
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import Tacotron2Config
    from text_to_speech_amd.distributed import synthesize_sharded
    from text_to_speech_amd.pipeline import TTSPipeline
    eng.load_state(weights.synth_tacotron2(Tacotron2Config(speaker_embedding_dim=256), seed=1234))
    eng.finalize()
    eng.set_decoder_mode('auto')
    N, T, Tin = CONFIG4_UTTERANCES, CONFIG4_FRAMES, CONFIG4_TIN
    tok = spk = None
    if rank == 0:
        rng = np.random.default_rng(21)
        lens = rng.integers(50, 201, N)
        tok = np.zeros((N, Tin), np.int32)
        for i, n in enumerate(lens):
            tok[i, :n] = rng.integers(1, 70, n)            # the French symbol table has 70 entries
        spk = rng.standard_normal((N, 256)).astype(np.float32)
        spk /= np.linalg.norm(spk, axis=1, keepdims=True)
    pipe = TTSPipeline(eng, seed=0)
    fn = pipe.shard_fn(max_length=T, deterministic=True, early_stopping=False)

    def run():
        return synthesize_sharded(tok, fn, speaker=spk, src=0, device=dev)

    run()                                                  # warm-up: graph capture, workspace growth, RCCL channels
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        audios = run()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item()) / reps
    paths = [None] * world
    dist.all_gather_object(paths, eng.last_decoder_mode)
    if rank != 0:
        return None
    samples = sum(len(a) for a in audios)
    assert samples == N * T * 256 and all(np.isfinite(a).all() for a in audios)
    per_rank = (N + world - 1) // world
    return {
        'workload': f'{N} SV2TTS utterances (enc 768, 50-200 tokens padded to {Tin}), {T} frames each, text -> audio, fp32; '
                    f'tokens + speaker embeddings scattered from rank 0, waveforms gathered on rank 0 (BASELINE.json configs[3])',
        'pipeline_samples_per_s': samples / dt, 'x_realtime': samples / dt / SAMPLE_RATE, 'ms_per_job': dt * 1e3,
        'scaling': 'strong', 'world_size': world, 'utterances_per_rank': per_rank, 'backend': 'nccl (RCCL)',
        'decoder_path_per_rank': paths,
        'bytes_scattered': int(world * per_rank * (Tin * 4 + 256 * 4) + N * 4 + 3 * 8),
        'bytes_gathered': int(world * per_rank * (T * 256 * 4 + 8)),
        'collectives_per_job': 'broadcast(meta) + broadcast(lengths) + scatter(tokens) + scatter(speaker) + all_gather(counts) + gather(waveforms)',
    }


# ------------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without torchrun: start the N ranks ourselves.  This process never imports torch or
    touches a GPU (a process that has initialised the GPU must not exec / be replaced, and needs none here): it only sets
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* for each child, lets rank 0 write the JSON line to our stdout and returns
    the worst exit status."""
    env0 = dict(os.environ)
    env0.setdefault('MASTER_ADDR', '127.0.0.1')
    env0.setdefault('MASTER_PORT', str(_free_port()))
    env0.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env0['WORLD_SIZE'] = str(n)
    env0['LOCAL_WORLD_SIZE'] = str(n)
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    deadline = None
    while procs:
        for p in list(procs):
            rc = p.poll()
            if rc is None:
                continue
            procs.remove(p)
            if rc != 0:
                worst = worst or rc
                if deadline is None:                      # a rank died: the others would wait at a barrier forever
                    deadline = time.time() + 20
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()                                  # exact PIDs we started
            for p in procs:
                p.wait()
            procs = []
        time.sleep(0.05)
    return worst


def headline_result(args, world, B, T, dt, avg_us, launches, distributed, form='direct', probe=None):
    """The JSON line without its secondary parts (CPU baseline, config-4 job, extras), from the timed region's numbers."""
    samples = world * B * T * 256 * args.steps
    M = B * T * 32
    roofline = None
    if launches and args.precision != 'f32':
        # fp16 matrix pipe (2.5 PFLOP/s dense): the split mode issues three MFMAs per product
        mfma_flops = wn_in_layer_flops(M) * (3 if args.precision == 'f16x3' else 1)
        achieved = mfma_flops / (avg_us * 1e-6) / 1e12
        roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': FP16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': achieved / FP16_MFMA_PEAK_TFLOPS,
                    'flops_per_launch': mfma_flops, 'traffic': None, 'launches_timed': launches,
                    'avg_launch_us': avg_us,
                    'kernel': 'WN in-layer implicit GEMM, fp16 MFMA (' + args.precision + ')'}
    elif launches:
        # `achieved` prices the FLOPs the timed kernel EXECUTES: the Winograd form ~K = 1120 per output (wino_in_layer_flops),
        # the direct form K = 1536 + 320
        wino = form == 'winograd'
        flops = wino_in_layer_flops(B, T) if wino else wn_in_layer_flops(M)
        achieved = flops / (avg_us * 1e-6) / 1e12
        if wino:
            kernel = ('wino4_fused2_kernel (csrc/wn_wino.hip): WN in-layer GEMM of layers 1-7 in its Winograd F(4,3) form, ONE launch per '
                      'layer -- the six input tiles of a K chunk by LDS-DMA, input transform in the operand reads, six products (K = 512 + '
                      '208 / 224 conditioning columns; dilation 16: 512 + 320 / 512) as accumulator sets of a 64 x 128 block tile, output '
                      'transform + bias + gate in the epilogue; average over the 7 launches of a flow')
            # per launch: the residual stream in, the gated activations out, the six tap-combination planes and the conditioning
            # planes of the layer's group kind (phase groups: [8][6][1024][224]), the mel planes
            alg_bytes = (M * 512 + M * 512 + 6 * 1024 * 512 + 8 * 6 * 1024 * 224 + 6 * B * T * 224) * 4.0
        else:
            kernel = ('gemm_f32_kernel<4,1,2,4,16,2,TAG_WN_IN=1,3,PIPE_DMA> (WN in-layer implicit GEMM, layers 1-7 of each flow; '
                      'K = 1536 taps + 320 folded conditioning)')
            alg_bytes = (M * (512 + 512) + B * T * 80 + 1024 * (1536 + 32 * 320)) * 4.0
        roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': achieved / FP32_MFMA_PEAK_TFLOPS,
                    'flops_per_launch': flops,
                    # the same launch counted in the direct (K = 1856) and in the reference's (K = 2176) formulation: one launch =
                    # one layer in both forms (round 3's figures divided one layer's FLOPs by the average of 8 launches for 7 layers)
                    'direct_formulation_tflops': wn_in_layer_flops(M, K_EXECUTED) / (avg_us * 1e-6) / 1e12,
                    'reference_formulation_tflops': wn_in_layer_flops(M, K_REFERENCE) / (avg_us * 1e-6) / 1e12,
                    'traffic': pmc_traffic_bytes(B, T),
                    'traffic_unit': 'bytes/launch (rocprofv3 PMC, profiles/pmc_hbm_traffic_latest.json)',
                    'algorithmic_bytes': alg_bytes, 'in_layer_form': form,
                    'kernel': kernel, 'launches_timed': launches, 'avg_launch_us': avg_us}
        # north_star also asks for the fraction of the HBM roofline on the WN sweep: algorithmic bytes (and, when the PMC passes of
        # this workload exist, the measured L2-miss traffic) of one launch over its time, against 8 TB/s.  MFMA-bound by construction.
        roofline['hbm_frac_algorithmic'] = alg_bytes / (avg_us * 1e-6) / (HBM_PEAK_TBS * 1e12)
        if roofline['traffic']:
            roofline['hbm_frac_traffic'] = roofline['traffic'] / (avg_us * 1e-6) / (HBM_PEAK_TBS * 1e12)
        if probe:
            # boxes of the pool differ by up to ~9 % in what their matrix pipe sustains (a bare MFMA loop measured right after the
            # timed region: 155 TFLOP/s at 2.40 GHz on most boxes, ~142 at ~2.2 GHz on some): the same kernel against THIS box
            roofline['box_probe_fp32_mfma_tflops'] = probe[0]
            roofline['box_probe_shader_clock_ghz'] = probe[1]
            roofline['frac_of_box_probe'] = achieved / probe[0] if probe[0] > 0 else None
    return {
        'metric': 'audio samples/sec (22.05 kHz WaveGlow vocoding, fp32)' if args.precision == 'f32' else
                  f'audio samples/sec (22.05 kHz WaveGlow vocoding, {args.precision})',
        'value': samples / dt, 'unit': 'audio samples/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': f'WaveGlow-only vocoding of precomputed 80x{T} mel, batch {B} per GPU, fp32 '
                               f'(BASELINE.json configs[1])', 'batch_per_gpu': B, 'mel_frames': T,
                   'audio_seconds_per_step': world * B * T * 256 / SAMPLE_RATE, 'sharding': 'utterances/GPU',
                   'world_size': world, 'backend': 'nccl (RCCL)' if distributed else 'single process',
                   'weights': 'seeded synthetic (rng 1234)',
                   'arithmetic': 'fp32 operands, fp32 MFMA accumulate; dilated convolutions of WN layers 1-7 in their '
                                 + ('Winograd F(4,3) form along the tap axis (csrc/wn_wino.hip, one fused kernel per layer; 6.0e-7 '
                                    'waveform RMS error against the oracle, the direct form 4.96e-7: tests/test_waveglow_gpu.py)'
                                    if form == 'winograd'
                                    else 'direct three-tap form')},
        'x_realtime': samples / dt / SAMPLE_RATE,
        'roofline': roofline, 'cpu_baseline': None, 'config4_sharded_job': None, 'extra': None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--frames', type=int, default=FRAMES)
    ap.add_argument('--cpu-frames', type=int, default=160, help='mel frames of the CPU-baseline WaveGlow sample (0 = skip)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the secondary (untimed-region) metrics')
    ap.add_argument('--no-config4', action='store_true', help='skip the BASELINE config-4 scatter / synthesize / gather job')
    ap.add_argument('--precision', default='f32', choices=('f32', 'f16x3', 'f16'),
                    help="arithmetic of the timed path: f32 = exact fp32 MFMA (default, the contract's config); "
                         "f16x3 = split fp16, fp32-class accuracy; f16 = fp16 operands")
    ap.add_argument('--dry-run', action='store_true',
                    help='rehearse the multi-rank plumbing on the CPU (gloo, no GPU, no HIP library): launcher, rendezvous, '
                         'barriers, MAX-over-ranks timing and the JSON line, with a sleep as the step')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE {world}')
    # The process group is always created (world 1 included): the config-4 job below runs its scatter / gather through RCCL
    # at every N, so that the N = 1 line and the N = 1 point of a scaling run are the same code.
    distributed = True
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()) if world == 1 else '29533')
        # RCCL prints a version banner to STDOUT when its first communicator is created; the contract is ONE JSON line on
        # stdout, so stdout points at stderr until the communicator exists (first collective)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
            assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from text_to_speech_amd import weights
    from text_to_speech_amd.config import WaveGlowConfig
    from text_to_speech_amd.engine import HipEngine, KERNEL_WN_IN

    cfg = WaveGlowConfig()
    w = weights.synth_waveglow(cfg, seed=1234)
    eng = HipEngine(local_rank)
    eng.load_state(w)
    eng.finalize()

    B, T = args.batch, args.frames
    dev = torch.device('cuda', local_rank)
    # per-rank shard of the job: B utterances (seeded per rank), resident in HBM before the timed region
    mel = torch.from_numpy(np.random.default_rng(7 + rank).uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)).to(dev)
    z = torch.from_numpy(np.random.default_rng(11 + rank).standard_normal((B, T * 32, 8)).astype(np.float32)).to(dev)
    torch.cuda.synchronize()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.waveglow_infer(mel, z=z, sigma=1.0, precision=args.precision)
    if not args.no_kernel_timing:
        eng.kernel_timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = eng.waveglow_infer(mel, z=z, sigma=1.0, precision=args.precision)   # returns after the engine stream has drained
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert bool(torch.isfinite(out).all())

    avg_us, launches = (0.0, 0) if args.no_kernel_timing else eng.kernel_time_us(KERNEL_WN_IN)
    eng.kernel_timing(False)
    probe = None
    if rank == 0 and args.precision == 'f32':
        try:
            probe = eng.probe_mfma_f32()              # right after the timed region: the box in the state the steps left it in
        except Exception:
            probe = None
    form = eng.last_waveglow_form                      # 'winograd' or 'direct': which in-layer GEMM the timed steps ran
    # BASELINE config 4: the scatter / synthesize / gather job, on every rank, at every N.  It must never cost the headline
    # line: an exception is reported inside the line, and a job that does not come back (a rank lost inside a collective
    # leaves the others waiting) is cut off by a watchdog thread that lets rank 0 print the line without it.
    config4 = None
    watchdog = None
    headline = {'ready': False}

    def give_up():
        if rank == 0 and headline['ready']:
            headline['result']['config4_sharded_job'] = {'error': f'no result after {CONFIG4_TIMEOUT_S} s (cut off by the watchdog)'}
            print(json.dumps(headline['result']), flush=True)
        # a rank stuck in a GPU collective is a FAILED run: the line (when there is one) is out, the status says so
        # (launch_ranks returns the worst child status), and nothing is retried in this process
        sys.stderr.write(f'[bench rank {rank}] config-4 job hung: no result after {CONFIG4_TIMEOUT_S} s; exiting with status 3\n')
        sys.stderr.flush()
        os._exit(3)
    if not args.no_config4:
        import threading
        if rank == 0:
            headline['result'] = headline_result(args, world, B, T, dt, avg_us, launches, distributed, form, probe)
            headline['ready'] = True
        watchdog = threading.Timer(CONFIG4_TIMEOUT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            config4 = config4_job(eng, dev, rank, world)
        except Exception as exc:                                 # never lose the headline line to the secondary job
            config4 = {'error': f'{type(exc).__name__}: {exc}'} if rank == 0 else None
        watchdog.cancel()
    # secondary metrics and the CPU leg only at N = 1 (the other ranks would idle at the final barrier)
    extra = secondary_metrics(eng, dev, rank) if (rank == 0 and world == 1 and not args.no_extra) else None
    if rank == 0:
        result = headline_result(args, world, B, T, dt, avg_us, launches, distributed, form, probe)
        if args.cpu_frames > 0 and world == 1:
            result['cpu_baseline'] = cpu_baseline(w, cfg, args.cpu_frames)
        result['config4_sharded_job'] = config4
        result['extra'] = extra
        print(json.dumps(result), flush=True)
    if distributed:
        # the line is out: a rank that is gone must not keep the others (and the driver) at this barrier
        import threading
        def stuck_at_exit():
            sys.stderr.write(f'[bench rank {rank}] final barrier not reached by every rank within 60 s; exiting with status 4\n')
            sys.stderr.flush()
            os._exit(4)
        bye = threading.Timer(60, stuck_at_exit)
        bye.daemon = True
        bye.start()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()
    eng.close()


def dry_run(args, world, rank):
    """CPU rehearsal of the rank plumbing (tests/test_bench_launcher.py): gloo instead of RCCL, a sleep instead of the HIP
    step.  Prints a line of the same shape marked "dry_run": true -- never a measurement."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus

    def barrier():
        if world > 1:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))                     # rank-dependent: MAX over ranks must pick the slowest
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        B, T = args.batch, args.frames
        samples = world * B * T * 256 * args.steps
        print(json.dumps({'metric': 'dry run (no GPU work)', 'dry_run': True, 'value': samples / dt, 'unit': 'audio samples/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision,
                          'data': 'synthetic', 'config': {'workload': 'dry run', 'world_size': world, 'backend': 'gloo'},
                          'roofline': None, 'cpu_baseline': None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
