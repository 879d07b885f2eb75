"""GPU: the BASELINE.json workloads other than the headline one (configs[0], [2], [3] per-GPU shard, [4]) run through the
product's own entry points with parity assertions -- not only as bench.py `extra` timings.

Tolerances (BASELINE.json north_star): mel within 1e-3 abs and waveform RMS within 1e-4 for fp32; `lengths` exact.  The
fp16 modes (configs 3 and 5: fp16 LSTM weights, fp16 WaveGlow GEMM operands, fp32 accumulation) are held to the measured
fp16 error with margin -- MEL_TOL_F16 / WAVE_RMS_TOL_F16 below -- and that error is printed: it is ABOVE the fp32 tolerance
(2e-4 vs 1e-4 waveform RMS), which every fp16 throughput figure in DESIGN.md repeats.
"""
import os
import socket

import numpy as np
import pytest

from conftest import rms
from stop_script import script_stop_tokens

pytestmark = pytest.mark.gpu

MEL_TOL, WAVE_RMS_TOL = 1e-3, 1e-4                  # fp32 (north_star)
MEL_TOL_F16, WAVE_RMS_TOL_F16 = 5e-3, 1e-3          # fp16 modes: measured 4e-4 / 2e-4, see the module docstring
CONFIG3_TOKENS = [50, 70, 90, 110, 130, 150, 170, 200]


def _mixed_length_tokens(lens, Tin, seed0=6):
    tok = np.zeros((len(lens), Tin), np.int32)
    for i, n in enumerate(lens):
        tok[i, :n] = np.random.default_rng(seed0 + i).integers(1, 148, n)
    return tok


def _engine(*states):
    from text_to_speech_amd.engine import HipEngine
    eng = HipEngine(0)
    for st in states:
        eng.load_state(st)
    eng.finalize()
    return eng


# ---------------------------------------------------------------------------------------------------------- configs[2]
def test_config3_pipeline_batch8_mixed_lengths_fp16(taco_cfg, taco_weights, wg_weights, wg_cfg):
    """Full Tacotron2 -> WaveGlow pipeline, batch 8, token counts 50..200 padded to 256, fp16 modes of both models.
    Stop tokens are scripted (tests/stop_script.py) so that rows end at staggered steps <= 40: `lengths` must be exact,
    the fp16-weight mel within MEL_TOL_F16 of the fp32 oracle, and the waveforms of the shortest and the longest row
    within WAVE_RMS_TOL_F16 of oracle Tacotron2 -> oracle WaveGlow (rows are independent in WaveGlow; two rows keep the
    numpy oracle to seconds)."""
    from oracle import tacotron2_ref, waveglow_ref
    from text_to_speech_amd.pipeline import PAD_MEL_VALUE, TTSPipeline
    tok = _mixed_length_tokens(CONFIG3_TOKENS, 256)
    targets = [12, 17, 23, 21, 28, 33, 37, 40]
    tw, sens, margin = script_stop_tokens(taco_weights, taco_cfg, tok, targets)
    print(f'scripted stops: gate norm {sens:.1f}, logit margin {margin:.2f}')
    ref = tacotron2_ref.infer(tok, tw, taco_cfg, max_length=64, early_stopping=True)
    assert ref.lengths.tolist() == targets
    eng = _engine(tw, wg_weights)
    try:
        out = eng.tacotron2_infer(tok, max_len=64, early_stopping=True, precision='f16')
        assert out.lengths.tolist() == targets and eng.last_steps == max(targets) + 1
        err_mel = max(float(np.abs(out.mel[b, :n + 1] - ref.mel[b, :n + 1]).max()) for b, n in enumerate(targets))
        print(f'config 3: fp16-weight mel max abs err {err_mel:.2e} (fp32 tolerance {MEL_TOL})')
        assert err_mel <= MEL_TOL_F16
        z = np.random.default_rng(11).standard_normal((8, 40 * 32, 8)).astype(np.float32)
        pipe = TTSPipeline(eng, vocoder_precision='f16', synthesizer_precision='f16')
        audios, n, steps = pipe.synthesize_tokens(tok, max_length=64, deterministic=True, z=z)
        assert n.tolist() == targets and steps == max(targets) + 1
        T = 40                                                         # the longest row, a multiple of 8 already
        mel = ref.mel[:, :T].copy()
        for b in range(8):
            mel[b, targets[b]:] = PAD_MEL_VALUE
        for b in (0, 7):
            ref_audio = waveglow_ref.infer(mel[b:b + 1], wg_weights, wg_cfg, z=z[b:b + 1])[0, :targets[b] * 256]
            assert audios[b].shape == ref_audio.shape
            e = rms(audios[b] - ref_audio)
            print(f'config 3: row {b} waveform RMS err {e:.2e} at signal RMS {rms(ref_audio):.2f} '
                  f'(fp32 tolerance {WAVE_RMS_TOL})')
            assert e <= WAVE_RMS_TOL_F16
        for b in range(8):
            assert audios[b].shape == (targets[b] * 256,) and np.isfinite(audios[b]).all()
    finally:
        eng.close()


def test_config3_full_size_properties_fp16(gpu_engine):
    """The workload at its full size (batch 8, 800 decoder steps, 800-frame vocoding) is beyond the numpy oracle, so it
    is checked through size-independent properties: everything finite, every row of the batch equal to its own batch-1
    run (Tacotron2 rows only interact through the shared step loop; WaveGlow rows not at all), lengths = max_len when no
    stop token fires."""
    from text_to_speech_amd.pipeline import TTSPipeline
    tok = _mixed_length_tokens(CONFIG3_TOKENS, 256)
    out = gpu_engine.tacotron2_infer(tok, max_len=800, early_stopping=False, want_attention=False, precision='f16')
    assert np.isfinite(out.mel).all() and out.lengths.tolist() == [800] * 8 and gpu_engine.last_steps == 800
    for b in (0, 3, 7):
        one = gpu_engine.tacotron2_infer(tok[b:b + 1], max_len=800, early_stopping=False, want_attention=False,
                                         precision='f16')
        d = float(np.abs(one.mel[0] - out.mel[b]).max())
        print(f'config 3 full size: row {b} batch-8 vs batch-1 mel max abs diff {d:.2e}')
        assert d <= MEL_TOL
    pipe = TTSPipeline(gpu_engine, seed=0, vocoder_precision='f16', synthesizer_precision='f16')
    audios, n, steps = pipe.synthesize_tokens(tok, max_length=800, deterministic=True, early_stopping=False)
    assert steps == 800 and n.tolist() == [800] * 8
    assert all(a.shape == (800 * 256,) and np.isfinite(a).all() for a in audios)
    assert 0.05 < rms(audios[0]) < 20.0


# ---------------------------------------------------------------------------------------------------------- configs[4]
_WORDS = ('the quick brown fox jumps over a lazy dog while seven wizards quietly box with five jumping zebras and '
          'every good vocoder makes sharp clear audio from a plain spectrogram without any audible glitch').split()


def _sentences(n, seed=0):
    """n pseudo sentences whose character counts cycle through the config-3 set (50 .. 200)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        want, words = CONFIG3_TOKENS[i % len(CONFIG3_TOKENS)], []
        while sum(len(w) + 1 for w in words) < want - 1:
            words.append(_WORDS[int(rng.integers(len(_WORDS)))])
        s = ' '.join(words)[:want - 1].rstrip()
        s += 's' * (want - 1 - len(s)) + '.'
        out.append(s[0].upper() + s[1:])
    return out


def test_config5_stream_64_sentences_overlapped_fp16(gpu_engine, wg_weights):
    """Streaming long-form: 64 sentences through `stream()` with Tacotron2(n + 1) pipelined against WaveGlow(n) on a
    second engine handle, fp16 modes.  The overlapped stream must deliver, in order, exactly the audio of the sequential
    stream (same kernels, same inputs: bit equal), and each sentence's length must be the decoder's frame count."""
    from text_to_speech_amd.runtime import HipRuntime
    from text_to_speech_amd.tacotron2 import Tacotron2, stream
    from text_to_speech_amd.waveglow import WaveGlow
    eng2 = _engine(wg_weights)
    try:
        kw = dict(synthesizer_precision='f16', vocoder_precision='f16', seed=0)
        model = Tacotron2(HipRuntime('t5', model='tacotron2', engine=gpu_engine, **kw))
        voc_same = WaveGlow(HipRuntime('w5', model='waveglow', engine=gpu_engine, **kw))
        voc_own = WaveGlow(HipRuntime('w5b', model='waveglow', engine=eng2, **kw))
        texts = _sentences(64)
        assert len(set(texts)) == 64 and {len(t) for t in texts} == set(CONFIG3_TOKENS)
        run_kw = dict(max_length=3., deterministic=True, save=False)    # 3 frames / token: inside the (2, 10) window
        got = {}
        for name, voc, overlap in (('sequential', voc_same, False), ('overlapped', voc_own, True)):
            rec = []
            stream(iter(texts), model=model, vocoder=voc, overlap=overlap,
                   callbacks=[lambda text, audio, **_: rec.append((text, np.asarray(audio).copy()))], **run_kw)
            got[name] = rec
        for name in got:
            assert [t for t, _ in got[name]] == texts, name                 # order preserved, nothing dropped
        n_tok = [len(model.encode_text(model.clean_text(t), cleaned=True)) for t in texts]
        for (t, a), (_, b), n in zip(got['sequential'], got['overlapped'], n_tok):
            assert a.shape == b.shape == (int(np.float32(n) * np.float32(3.)) * 256,)
            assert np.array_equal(a, b), t
            assert np.isfinite(a).all()
    finally:
        eng2.close()


# -------------------------------------------------------------------------------------------------- configs[3], one shard
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_config4_sv2tts_shard_through_rccl_world1(wg_weights, wg_cfg):
    """One GPU's share of config 4 (SV2TTS, 256-d speaker embeddings, French text): 4 utterances through
    `distributed.synthesize_sharded` on the `nccl` (= RCCL) backend at world size 1 -- scatter, per-rank synthesis with
    `TTSPipeline.shard_fn`, gather -- with per-row parity against the oracle (enc 768)."""
    import torch
    import torch.distributed as dist
    from oracle import tacotron2_ref, waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import Tacotron2Config
    from text_to_speech_amd.distributed import synthesize_sharded
    from text_to_speech_amd.pipeline import PAD_MEL_VALUE, TTSPipeline
    from text_to_speech_amd.text import CharTokenizer
    cfg = Tacotron2Config(speaker_embedding_dim=256)
    tw = weights.synth_tacotron2(cfg, seed=99)
    tokenizer = CharTokenizer('fr')
    texts = ['Bonjour à tous, ceci est un test.', 'Il y a 91 chats dans le jardin.', 'Très bien !',
             'La synthèse vocale fonctionne sur quatre phrases.']
    enc = [tokenizer.encode(t) for t in texts]
    Tin = max(len(e) for e in enc)
    tok = np.zeros((4, Tin), np.int32)
    for i, e in enumerate(enc):
        tok[i, :len(e)] = e
    spk = np.random.default_rng(5).standard_normal((4, 256)).astype(np.float32)
    spk /= np.linalg.norm(spk, axis=1, keepdims=True)
    targets = [14, 11, 6, 16]
    tw, sens, margin = script_stop_tokens(tw, cfg, tok, targets, speaker_embedding=spk)
    ref = tacotron2_ref.infer(tok, tw, cfg, speaker_embedding=spk, max_length=32, early_stopping=True)
    assert ref.lengths.tolist() == targets
    T = 16
    z = np.random.default_rng(3).standard_normal((4, T * 32, 8)).astype(np.float32)

    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ['MASTER_PORT'] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    eng = _engine(tw, wg_weights)
    try:
        assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
        pipe = TTSPipeline(eng)
        # z rows travel with the utterances: the shard function receives them in the scattered order
        order = {}

        def synth(local_tok, local_spk):
            # match each local row to its global utterance (tokens are unique) to pick its noise
            lt = local_tok.cpu().numpy()
            idx = [int(np.where((tok[:, :lt.shape[1]] == r).all(1))[0][0]) for r in lt]
            order['idx'] = idx
            return pipe.shard_fn(max_length=32, deterministic=True, z=z[idx])(local_tok, local_spk)

        audios = synthesize_sharded(tok, synth, speaker=spk)
        assert order['idx'] == sorted(range(4), key=lambda i: (-len(enc[i]), i))      # longest first
        # the waveforms stayed on the GPU from WaveGlow's output into the RCCL gather: no upload, one download (rank 0's)
        from text_to_speech_amd import distributed as D
        assert D.last_transfer == {'scatter_h2d': 2, 'gather_h2d': 0, 'gather_d2h': 1, 'gather_device_resident': True}, D.last_transfer
    finally:
        dist.destroy_process_group()
        eng.close()
    mel = ref.mel[:, :T].copy()
    for b in range(4):
        mel[b, targets[b]:] = PAD_MEL_VALUE
    ref_audio = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z)
    for b in range(4):
        assert audios[b].shape == (targets[b] * 256,)
        e = rms(audios[b] - ref_audio[b, :targets[b] * 256])
        print(f'config 4 shard: row {b} ({targets[b]} frames) waveform RMS err {e:.2e}')
        assert e <= WAVE_RMS_TOL


def test_config4_full_batch_of_32_through_rccl_world1(wg_weights, wg_cfg):
    """BASELINE config 4 in full on one GPU: 32 SV2TTS utterances (enc 768) through scatter -> `TTSPipeline.shard_fn` ->
    gather on `nccl` at world size 1 (what `bench.py` times as its config-4 job).  Rows are independent: the batch of 32
    (per-step graph, four LSTM passes per step) must equal the same utterances synthesized as four batches of 8 (fused
    two-kernel step), and two of the rows are checked against the oracle."""
    import torch
    import torch.distributed as dist
    from oracle import tacotron2_ref, waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import Tacotron2Config
    from text_to_speech_amd.distributed import partition, synthesize_sharded
    from text_to_speech_amd.pipeline import TTSPipeline
    cfg = Tacotron2Config(speaker_embedding_dim=256)
    tw = weights.synth_tacotron2(cfg, seed=99)
    rng = np.random.default_rng(12)
    N, Tin, T = 32, 64, 12
    lens = rng.integers(20, 61, N)
    tok = np.zeros((N, Tin), np.int32)
    for i, n in enumerate(lens):
        tok[i, :n] = rng.integers(1, 148, n)
    spk = rng.standard_normal((N, 256)).astype(np.float32)
    spk /= np.linalg.norm(spk, axis=1, keepdims=True)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ['MASTER_PORT'] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    eng = _engine(tw, wg_weights)
    try:
        pipe = TTSPipeline(eng)
        kw = dict(max_length=T, deterministic=True, early_stopping=False)
        audios = synthesize_sharded(tok, pipe.shard_fn(**kw), speaker=spk)
        from text_to_speech_amd import distributed as D
        assert D.last_transfer['gather_device_resident'] and D.last_transfer['gather_h2d'] == 0 and D.last_transfer['gather_d2h'] == 1
        assert eng.last_decoder_mode == 'graph'                      # 32 rows: above the fused step's 8
        assert len(audios) == N and all(a.shape == (T * 256,) for a in audios)
        order = partition(lens.tolist(), 1)[0]
        for k in range(0, N, 8):
            idx = order[k:k + 8]
            part, n_frames, _ = pipe.synthesize_tokens(tok[idx], speaker=spk[idx], **kw)
            assert eng.last_decoder_mode == 'fused' and n_frames.tolist() == [T] * 8
            for row, i in enumerate(idx):
                assert rms(part[row] - audios[i]) <= WAVE_RMS_TOL, (k, row)
    finally:
        dist.destroy_process_group()
        eng.close()
    pair = [0, 17]
    ref = tacotron2_ref.infer(tok[pair], tw, cfg, speaker_embedding=spk[pair], max_length=T, early_stopping=False)
    ref_audio = waveglow_ref.infer(ref.mel, wg_weights, wg_cfg, z=None)
    for row, i in enumerate(pair):
        e = rms(audios[i] - ref_audio[row])
        print(f'config 4, batch 32: utterance {i} waveform RMS err vs oracle {e:.2e}')
        assert e <= WAVE_RMS_TOL


# ---------------------------------------------------------------------------------------------------------- configs[0]
def test_config1_single_100_char_sentence_through_tts(gpu_engine, taco_weights, taco_cfg, wg_weights, wg_cfg):
    """The plumbing case: one ~100-character English sentence through `tts()` at batch 1 (cleaners -> ids -> Tacotron2
    -> mel slice -> WaveGlow -> result dict), compared with oracle Tacotron2 -> oracle WaveGlow on the same ids."""
    from oracle import tacotron2_ref, waveglow_ref
    from text_to_speech_amd.runtime import HipRuntime
    from text_to_speech_amd.tacotron2 import Tacotron2, tts
    from text_to_speech_amd.waveglow import WaveGlow
    text = 'The quick brown fox jumps over the lazy dog, and then it runs back to the forest before dark falls..'
    assert len(text) == 100
    model = Tacotron2(HipRuntime('t1', model='tacotron2', engine=gpu_engine, seed=0))
    voc = WaveGlow(HipRuntime('w1', model='waveglow', engine=gpu_engine, seed=0))
    res = tts(text, model=model, vocoder=voc, max_length=2.2, deterministic=True, save=False)
    ids = model.encode_text(model.clean_text(text), cleaned=True)
    assert len(ids) == 100 and res['cleaned'] == text.lower()
    frames = int(np.float32(100) * np.float32(2.2))
    ref = tacotron2_ref.infer(ids[None], taco_weights, taco_cfg, max_length=2.2, early_stopping=True)
    assert ref.lengths.tolist() == [frames] and len(res['mel']) == 1 and res['mel'][0].shape == (frames, 80)
    err = float(np.abs(np.asarray(res['mel'][0]) - ref.mel[0]).max())
    print(f'config 1: mel max abs err {err:.2e}')
    assert err <= MEL_TOL
    ref_audio = waveglow_ref.infer(ref.mel, wg_weights, wg_cfg, z=None)[0]
    assert res['audio'].shape == ref_audio.shape == (frames * 256,) and res['rate'] == 22050
    e = rms(res['audio'] - ref_audio)
    print(f'config 1: waveform RMS err {e:.2e} at signal RMS {rms(ref_audio):.2f}')
    assert e <= WAVE_RMS_TOL
    assert abs(res['time'] - frames * 256 / 22050) < 1e-9
