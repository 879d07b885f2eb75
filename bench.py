#!/usr/bin/env python3
"""Headline benchmark: WaveGlow vocoding of precomputed 80x800 mels, batch 8, fp32, per MI355X (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (tts_hip_waveglow_infer through the C ABI) over one batch of synthetic mels that is
already resident in HBM.  With N > 1 every rank vocodes its own batch of 8 utterances (utterances are independent:
no data-path collective, weak scaling); the timed region is bracketed by barrier + synchronize and the MAX over ranks
is reported.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- dominant kernel = the WN in-layer implicit GEMM (K = 3*512 + 640, N = 1024).  `achieved` = algorithmic
                  FLOPs per launch / average launch duration measured with HIP events on the engine's stream.
  cpu_baseline -- the numpy oracle (a CPU port of the reference algorithm, NOT the reference's TF2 path, which cannot run
                  here) timed on a bounded sample on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md "Peak FP32 (matrix)"
BATCH, FRAMES = 8, 800                 # BASELINE.json configs[1]
SAMPLE_RATE = 22050


K_EXECUTED = 3 * 512 + 4 * 80      # dilated k3 conv taps + conditioning folded onto 4 mel frames
K_REFERENCE = 3 * 512 + 640        # the reference formulation: taps + 640-channel upsampled spectrogram


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


def cpu_baseline(wg_weights, cfg, frames: int, threads: int):
    """Times the numpy oracle on `frames` mel frames, batch 1 (oracle = checker; here only as the reported CPU leg)."""
    from oracle import waveglow_ref
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                   # pragma: no cover
        threadpool_limits = None
    mel = np.random.default_rng(7).uniform(-11.5, 1.2, (1, frames, 80)).astype(np.float32)
    z = np.random.default_rng(11).standard_normal((1, frames * 32, 8)).astype(np.float32)

    def run():
        t0 = time.perf_counter()
        waveglow_ref.infer(mel, wg_weights, cfg, z=z)
        return time.perf_counter() - t0

    if threadpool_limits is not None:
        with threadpool_limits(limits=threads):
            dt = run()
    else:
        dt = run()
    return {
        'value': frames * 256 / dt, 'unit': 'audio samples/s', 'cores': threads, 'kind': 'port',
        'sample': f'numpy oracle (OpenBLAS, {threads} threads), WaveGlow batch 1 x {frames} frames, one run of '
                  f'{dt:.1f} s; stand-in for the reference TF2 CPU path, which cannot be imported here',
    }


def secondary_metrics(eng, dev, rank):
    """The other numbers BASELINE.json's metric names (batch 1 WaveGlow; Tacotron2 mel-frames/s at batch 1 and 8),
    measured after the headline region on the same engine.  Tacotron2: 100-token utterances padded to 128, 800 decoder
    steps with early stopping off, deterministic prenet, seeded synthetic weights; encoder + postnet included."""
    import torch
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import Tacotron2Config
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
    # value, which stays exact fp32.  fp16 operands, fp32 accumulate: see DESIGN.md "fp16 mode".
    mel8 = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (BATCH, FRAMES, 80)).astype(np.float32)).to(dev)
    z8 = torch.from_numpy(np.random.default_rng(2).standard_normal((BATCH, FRAMES * 32, 8)).astype(np.float32)).to(dev)
    eng.waveglow_infer(mel8, z=z8, precision='f16')
    t0 = time.perf_counter()
    for _ in range(3):
        eng.waveglow_infer(mel8, z=z8, precision='f16')
    dt = (time.perf_counter() - t0) / 3
    out['waveglow_batch8_f16_samples_per_s'] = BATCH * FRAMES * 256 / dt
    out['waveglow_batch8_f16_ms_per_step'] = dt * 1e3
    # split-fp16 mode (f16x3): (hi, lo) fp16 operand planes, three MFMAs per product, fp32 accumulate -- fp32-class
    # accuracy (5e-7 waveform RMS error against the oracle, like the exact path) on the half-precision matrix cores.
    # Reported here, not as the headline: the headline `value` stays the exact fp32 MFMA path.
    eng.waveglow_infer(mel8, z=z8, precision='f16x3')
    t0 = time.perf_counter()
    for _ in range(3):
        eng.waveglow_infer(mel8, z=z8, precision='f16x3')
    dt = (time.perf_counter() - t0) / 3
    out['waveglow_batch8_f16x3_samples_per_s'] = BATCH * FRAMES * 256 / dt
    out['waveglow_batch8_f16x3_ms_per_step'] = dt * 1e3
    del mel8, z8
    eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234))
    eng.finalize()
    for B in (1, 8):
        tok = np.zeros((B, 128), np.int32)
        tok[:, :100] = np.random.default_rng(5).integers(1, 148, (B, 100))
        tok_d = torch.from_numpy(tok).to(dev)
        eng.tacotron2_infer(tok_d, max_len=64, early_stopping=False, want_attention=False)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            eng.tacotron2_infer(tok_d, max_len=FRAMES, early_stopping=False, want_attention=False)
        dt = (time.perf_counter() - t0) / reps
        out[f'tacotron2_batch{B}_mel_frames_per_s'] = B * FRAMES / dt
        out[f'tacotron2_batch{B}_us_per_decoder_step'] = 1e6 * dt / FRAMES
        eng.tacotron2_infer(tok_d, max_len=64, early_stopping=False, want_attention=False, precision='f16')
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.tacotron2_infer(tok_d, max_len=FRAMES, early_stopping=False, want_attention=False, precision='f16')
        dt = (time.perf_counter() - t0) / reps
        out[f'tacotron2_batch{B}_f16w_mel_frames_per_s'] = B * FRAMES / dt
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
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--frames', type=int, default=FRAMES)
    ap.add_argument('--cpu-frames', type=int, default=480, help='mel frames of the CPU-baseline sample (0 = skip)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the secondary (untimed-region) metrics')
    ap.add_argument('--precision', default='f32', choices=('f32', 'f16x3', 'f16'),
                    help="arithmetic of the timed path: f32 = exact fp32 MFMA (default, the contract's config); "
                         "f16x3 = split fp16, fp32-class accuracy; f16 = fp16 operands")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N')
    distributed = world > 1 or os.environ.get('TTS_BENCH_FORCE_DIST') == '1'   # (env: exercise the RCCL path at N = 1)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

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
    # secondary metrics and the CPU leg only at N = 1 (the other ranks would idle at the final barrier)
    extra = secondary_metrics(eng, dev, rank) if (rank == 0 and world == 1 and not args.no_extra) else None
    samples = world * B * T * 256 * args.steps
    result = None
    if rank == 0:
        M = B * T * 32
        roofline = None
        if launches and args.precision != 'f32':
            # fp16 matrix pipe (2.5 PFLOP/s dense): the split mode issues three MFMAs per product
            mfma_flops = wn_in_layer_flops(M) * (3 if args.precision == 'f16x3' else 1)
            achieved = mfma_flops / (avg_us * 1e-6) / 1e12
            roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': achieved / 2500.0,
                        'flops_per_launch': mfma_flops, 'traffic': None, 'launches_timed': launches,
                        'avg_launch_us': avg_us,
                        'kernel': 'WN in-layer implicit GEMM, fp16 MFMA (' + args.precision + ')'}
        elif launches:
            achieved = wn_in_layer_flops(M) / (avg_us * 1e-6) / 1e12
            roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': achieved / FP32_MFMA_PEAK_TFLOPS,
                        'flops_per_launch': wn_in_layer_flops(M),
                        'reference_formulation_tflops': wn_in_layer_flops(M, K_REFERENCE) / (avg_us * 1e-6) / 1e12,
                        'traffic': pmc_traffic_bytes(B, T),
                        'traffic_unit': 'bytes/launch (rocprofv3 PMC, profiles/pmc_hbm_traffic_latest.json)',
                        'algorithmic_bytes': (M * (512 + 512) + B * T * 80 + 1024 * (1536 + 32 * 320)) * 4.0,
                        'kernel': 'gemm_f32_kernel<4,1,2,4,16,2,TAG_WN_IN=1,3,PIPE_DMA> (WN in-layer implicit GEMM, layers 1-7 of each flow; K = 1536 taps + 320 folded conditioning)', 'launches_timed': launches,
                        'avg_launch_us': avg_us}
        cpu = None
        if args.cpu_frames > 0 and world == 1:
            threads = min(os.cpu_count() or 1, 16)
            cpu = cpu_baseline(w, cfg, args.cpu_frames, threads)
        result = {
            'metric': 'audio samples/sec (22.05 kHz WaveGlow vocoding, fp32)' if args.precision == 'f32' else
                      f'audio samples/sec (22.05 kHz WaveGlow vocoding, {args.precision})',
            'value': samples / dt, 'unit': 'audio samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': f'WaveGlow-only vocoding of precomputed 80x{T} mel, batch {B} per GPU, fp32 '
                                   f'(BASELINE.json configs[1])', 'batch_per_gpu': B, 'mel_frames': T,
                       'audio_seconds_per_step': world * B * T * 256 / SAMPLE_RATE, 'sharding': 'utterances/GPU',
                       'weights': 'seeded synthetic (rng 1234)'},
            'x_realtime': samples / dt / SAMPLE_RATE,
            'roofline': roofline, 'cpu_baseline': cpu, 'extra': extra,
        }
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
