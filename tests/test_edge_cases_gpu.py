"""GPU: edge cases of the HIP path (smallest / ragged sizes, error behaviour, device-tensor boundary, pipeline)."""
import ctypes

import numpy as np
import pytest

from conftest import rms

pytestmark = pytest.mark.gpu


def test_waveglow_single_frame_and_odd_sizes(gpu_engine, wg_weights, wg_cfg):
    """T = 1 (32 positions: every block is mostly tail) and B*L not a multiple of the 256-row tile."""
    from oracle import waveglow_ref
    for B, T in ((1, 1), (5, 3)):
        mel = np.random.default_rng(T).uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)
        z = np.random.default_rng(B).standard_normal((B, T * 32, 8)).astype(np.float32)
        ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z)
        out = gpu_engine.waveglow_infer(mel, z=z)
        assert out.shape == (B, T * 256) and rms(out - ref) <= 1e-4


def test_waveglow_batch_rows_are_independent(gpu_engine):
    """Dilated taps must not leak across utterance boundaries inside the flattened [B*L] position axis."""
    rng = np.random.default_rng(0)
    mel = rng.uniform(-11.5, 1.2, (3, 9, 80)).astype(np.float32)
    z = rng.standard_normal((3, 9 * 32, 8)).astype(np.float32)
    full = gpu_engine.waveglow_infer(mel, z=z)
    for b in range(3):
        single = gpu_engine.waveglow_infer(mel[b:b + 1], z=z[b:b + 1])
        np.testing.assert_allclose(single[0], full[b], atol=2e-6)


def test_waveglow_device_tensors_match_host_path(gpu_engine):
    import torch
    rng = np.random.default_rng(1)
    mel = rng.uniform(-11.5, 1.2, (2, 7, 80)).astype(np.float32)
    z = rng.standard_normal((2, 7 * 32, 8)).astype(np.float32)
    host = gpu_engine.waveglow_infer(mel, z=z)
    dev = gpu_engine.waveglow_infer(torch.from_numpy(mel).cuda(), z=torch.from_numpy(z).cuda())
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), host)       # same kernels: bit-identical


def test_waveglow_is_deterministic_run_to_run(gpu_engine):
    rng = np.random.default_rng(2)
    mel = rng.uniform(-11.5, 1.2, (1, 16, 80)).astype(np.float32)
    z = rng.standard_normal((1, 16 * 32, 8)).astype(np.float32)
    a = gpu_engine.waveglow_infer(mel, z=z)
    b = gpu_engine.waveglow_infer(mel, z=z)
    assert np.array_equal(a, b)


def test_tacotron2_minimal_sizes(gpu_engine, taco_weights, taco_cfg):
    from oracle import tacotron2_ref
    tok = np.array([[7]], np.int32)                                       # one token, one decoder step
    ref = tacotron2_ref.infer(tok, taco_weights, taco_cfg, max_length=1, early_stopping=False)
    out = gpu_engine.tacotron2_infer(tok, max_len=1, early_stopping=False)
    assert np.abs(out.mel - ref.mel).max() <= 1e-3 and np.array_equal(out.lengths, ref.lengths)
    tok = np.random.default_rng(3).integers(1, 148, (2, 3)).astype(np.int32)
    ref = tacotron2_ref.infer(tok, taco_weights, taco_cfg, max_length=33, early_stopping=False)   # 1 step into chunk 2
    out = gpu_engine.tacotron2_infer(tok, max_len=33, early_stopping=False)
    assert gpu_engine.last_steps == 33
    assert np.abs(out.mel - ref.mel).max() <= 1e-3


def test_tacotron2_float_max_length_through_runtime(gpu_engine, taco_weights, taco_cfg):
    """HipRuntime resolves max_length=10. like Tacotron2.infer (tacotron2_arch.py:886-892) and returns the namedtuple."""
    from oracle import tacotron2_ref
    from text_to_speech_amd.runtime import HipRuntime
    rt = HipRuntime('test', model='tacotron2', engine=gpu_engine)
    tok = np.zeros((1, 64), np.int32)
    tok[0, :5] = [3, 9, 27, 81, 100]
    out = rt(tok, max_length=10., deterministic=True, early_stopping=False, padding_multiple=64)
    ref = tacotron2_ref.infer(tok, taco_weights, taco_cfg, max_length=10., early_stopping=False)
    assert out.mel.shape == ref.mel.shape == (1, 50, 80)
    assert out._fields == ('decoder_output', 'mel', 'stop_tokens', 'attention_weights', 'lengths')
    assert np.abs(out.mel - ref.mel).max() <= 1e-3


def test_c_abi_argument_errors(gpu_engine):
    lib, h = gpu_engine._lib, gpu_engine._h
    buf = (ctypes.c_float * 16)()
    assert lib.tts_hip_waveglow_infer(h, None, 1, 1, None, 1.0, buf, 0) == -1
    assert b'bad argument' in lib.tts_hip_last_error(h)
    assert lib.tts_hip_waveglow_infer(h, buf, 0, 1, None, 1.0, buf, 0) == -1
    assert lib.tts_hip_waveglow_infer(h, buf, 1, 1, None, 1.0, buf, 7) == -1          # bad mem kind
    assert lib.tts_hip_mel_stft(h, buf, 1, 16, buf, 0) == -1                            # N < 1024
    with pytest.raises(ValueError):
        gpu_engine.waveglow_infer(np.zeros((1, 4, 79), np.float32))
    with pytest.raises(ValueError):
        gpu_engine.waveglow_infer(np.zeros((1, 4, 80), np.float32), z=np.zeros((1, 4, 8), np.float32))
    with pytest.raises(ValueError):
        gpu_engine.tacotron2_infer(np.zeros((1, 4), np.int32), max_len=8, prenet_masks=np.ones((1, 7, 2, 256), np.float32))


def test_engine_without_weights_reports_not_ready():
    from text_to_speech_amd.engine import HipEngine
    from text_to_speech_amd import HipLibraryError
    eng = HipEngine(0)
    try:
        eng.finalize()                                                     # nothing loaded: only mel-STFT becomes ready
        assert eng.has_model('mel_stft') and not eng.has_model('waveglow') and not eng.has_model('tacotron2')
        with pytest.raises(HipLibraryError, match='not finalized'):
            eng.waveglow_infer(np.zeros((1, 2, 80), np.float32))
        eng.set_tensor('waveglow/upsample/kernel', np.zeros((1024, 80, 80), np.float32))
        with pytest.raises(HipLibraryError, match='missing tensor'):
            eng.finalize()
    finally:
        eng.close()


def test_wrong_shape_tensors_are_rejected_at_finalize(taco_weights):
    """Every Tacotron2 tensor is indexed with fixed extents after finalize: a right name with a wrong shape must be an
    EINVAL at finalize (with the tensor named), not a host over-read or an undersized device buffer."""
    from text_to_speech_amd.engine import HipEngine
    from text_to_speech_amd import HipLibraryError
    bad = {
        'tacotron2/encoder/bi_lstm/forward/kernel': (512, 512),
        'tacotron2/decoder/attention_rnn/kernel': (768, 2048),
        'tacotron2/decoder/decoder_rnn/cell_0/recurrent_kernel': (512, 4096),
        'tacotron2/decoder/lsa/location_conv/kernel': (31, 2, 16),
        'tacotron2/decoder/gate_output/kernel': (1024, 1),
        'tacotron2/decoder/prenet/layer_1/kernel': (256, 128),
        'tacotron2/postnet/norm_3/moving_variance': (80,),
    }
    for name, shape in bad.items():
        assert tuple(taco_weights[name].shape) != shape
        eng = HipEngine(0)
        try:
            eng.load_state(taco_weights)
            eng.set_tensor(name, np.zeros(shape, np.float32))
            with pytest.raises(HipLibraryError) as ei:
                eng.finalize()
            tail = name.split('/', 1)[1].rsplit('/', 1)[0]
            assert tail in str(ei.value) or 'unsupported' in str(ei.value), str(ei.value)
            assert not eng.has_model('tacotron2')
        finally:
            eng.close()


def test_ttsw_file_roundtrip_through_c_loader(tmp_path, gpu_engine, taco_weights, taco_cfg):
    """tts_hip_load_weights reads the same TTSW file weights.save_ttsw writes."""
    from text_to_speech_amd import weights
    from text_to_speech_amd.engine import HipEngine
    p = tmp_path / 'taco.ttsw'
    weights.save_ttsw(p, taco_weights)
    eng = HipEngine(0)
    try:
        eng.load_weights(str(p))
        eng.finalize()
        tok = np.random.default_rng(4).integers(1, 148, (1, 9)).astype(np.int32)
        a = eng.tacotron2_infer(tok, max_len=6, early_stopping=False)
        b = gpu_engine.tacotron2_infer(tok, max_len=6, early_stopping=False)
        assert np.array_equal(a.mel, b.mel)
    finally:
        eng.close()


def test_pipeline_batch_mixed_lengths(taco_cfg, wg_weights, wg_cfg):
    """Config-3 shape: mixed-length batch, mel stays on the GPU; result equals oracle Tacotron2 -> padded oracle WaveGlow."""
    from oracle import tacotron2_ref, waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.engine import HipEngine
    from text_to_speech_amd.pipeline import TTSPipeline, PAD_MEL_VALUE
    tw = weights.synth_tacotron2(taco_cfg, seed=1234, gate_bias=-6.55)
    tw['tacotron2/decoder/gate_output/kernel'] = tw['tacotron2/decoder/gate_output/kernel'] * 10
    rng = np.random.default_rng(2)
    tok = rng.integers(1, 148, (4, 30)).astype(np.int32)
    for b, n in enumerate([30, 25, 18, 12]):
        tok[b, n:] = 0
    eng = HipEngine(0)
    try:
        eng.load_state(tw)
        eng.load_state(wg_weights)
        eng.finalize()
        z = rng.standard_normal((4, 40 * 32, 8)).astype(np.float32)
        audios, n, steps = TTSPipeline(eng).synthesize_tokens(tok, max_length=100, deterministic=True, z=z)
    finally:
        eng.close()
    ref = tacotron2_ref.infer(tok, tw, taco_cfg, max_length=100, early_stopping=True)
    assert n.tolist() == ref.lengths.tolist() == [7, 3, 5, 37] and steps == 38
    T = 40                                                                  # 37 rounded up to a multiple of 8
    mel = ref.mel[:, :T].copy()
    for b in range(4):
        mel[b, ref.lengths[b]:] = PAD_MEL_VALUE
    ref_audio = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z)
    for b in range(4):
        assert audios[b].shape == (ref.lengths[b] * 256,)
        assert rms(audios[b] - ref_audio[b, :ref.lengths[b] * 256]) <= 1e-4


def test_tts_and_stream_facade_on_the_engine(gpu_engine):
    """models.tts.tts()/stream() call shapes (models/tts/__init__.py:62-101) on the real HIP engine.

    Synthetic weights never fire the stop token, so every part runs to max_length = 6 x tokens (ratio 6 is inside the
    reference's (2, 10) acceptance window: no retry), then is vocoded and concatenated."""
    import queue
    from text_to_speech_amd.runtime import HipRuntime
    from text_to_speech_amd.tacotron2 import Tacotron2, tts, stream
    from text_to_speech_amd.waveglow import WaveGlow
    model = Tacotron2(HipRuntime('t', model='tacotron2', engine=gpu_engine, seed=0))
    voc = WaveGlow(HipRuntime('w', model='waveglow', engine=gpu_engine, seed=0))
    res = tts('Hello there. General test!', model=model, vocoder=voc, max_length=6., max_text_length=-2, save=False)
    assert res['splitted'] == ['hello there. ', 'general test!']          # the space behind a terminator is a token too
    frames = [m.shape[0] for m in res['mel']]
    assert frames == [6 * 13, 6 * 13]
    assert res['audio'].shape == (sum(frames) * 256,) and np.isfinite(res['audio']).all() and res['rate'] == 22050
    # windowed vocoding of a long mel through the same engine: seamless length, finite
    long_audio = voc.infer(np.concatenate(res['mel'], 0), win_len=64, hop_len=-16)
    assert long_audio.shape == (sum(frames) * 256,) and np.isfinite(long_audio).all()
    q = queue.Queue()
    for s in ('First sentence.', 'Second one.', None):
        q.put(s)
    got = []
    stream(q, model=model, vocoder=voc, max_length=6., save=False,
           callbacks=[lambda text, audio, **_: got.append((text, len(audio)))])
    assert [g[0] for g in got] == ['First sentence.', 'Second one.']
    assert got[0][1] == 6 * 15 * 256 and got[1][1] == 6 * 11 * 256


def test_waveglow_large_batch_is_sliced_by_utterance(gpu_engine):
    """B*T above one run's 31-bit addressing limit (~31.7 k frames): the library runs slices of whole utterances.
    5 x 8000 frames -> slices of 3 + 2; an utterance of the second slice must equal its own batch-1 run."""
    import torch
    g = torch.Generator(device='cuda').manual_seed(5)
    mel = torch.rand((5, 8000, 80), device='cuda', generator=g) * 12.7 - 11.5
    z = torch.randn((5, 8000 * 32, 8), device='cuda', generator=g)
    full = gpu_engine.waveglow_infer(mel, z=z)
    assert full.shape == (5, 8000 * 256) and bool(torch.isfinite(full).all())
    single = gpu_engine.waveglow_infer(mel[4:5].contiguous(), z=z[4:5].contiguous())
    assert float((single[0] - full[4]).double().pow(2).mean().sqrt()) <= 5e-6
    del full, single, mel, z
    torch.cuda.empty_cache()
    with pytest.raises(Exception, match='windowed inference'):
        gpu_engine.waveglow_infer(np.zeros((1, 31745, 80), np.float32))


def test_waveglow_exact_halo_tiling_matches_single_run(gpu_engine):
    """WaveGlow.infer_exact: time tiles with a 100-frame halo reproduce the one-run samples (receptive field of 12 flows =
    3060 groups = 95.6 frames + upsampling window); the reference's win_len/hop_len windowing is only approximate."""
    from text_to_speech_amd.runtime import HipRuntime
    from text_to_speech_amd.waveglow import WaveGlow
    voc = WaveGlow(HipRuntime('w', model='waveglow', engine=gpu_engine, seed=0))
    rng = np.random.default_rng(9)
    mel = rng.uniform(-11.5, 1.2, (1, 460, 80)).astype(np.float32)
    z = rng.standard_normal((1, 460 * 32, 8)).astype(np.float32)
    from text_to_speech_amd.waveglow import infer_tiled
    try:
        # direct form: a tile computes its core samples with the same arithmetic as the one-run call (only the tile kernels --
        # 128 / 256 rows -- may differ)
        gpu_engine.set_waveglow_form('direct')
        full = gpu_engine.waveglow_infer(mel, z=z)
        tiled = voc.infer_exact(mel, tile_frames=130, z=z)
        assert tiled.shape == full.shape == (1, 460 * 256)
        d = np.abs(tiled - full).max()
        print('exact tiling max abs diff, direct form', d)
        assert d <= 2e-6
        # context does matter: a 2-frame halo is shorter than even one flow's reach (255 groups = 8 frames)
        short = infer_tiled(voc.compiled_infer, mel, z=z, tile_frames=130, halo=2)
        assert np.abs(short - full).max() > 1e-4
    finally:
        gpu_engine.set_waveglow_form('winograd')
    # default form: the one-run call (460 frames: 256-row tiles) takes the Winograd form, the 330-frame tiles the direct one, and
    # a Winograd output's rounding depends on its slot in its group of four: equal within fp32 rounding, not bit for bit
    full_w = gpu_engine.waveglow_infer(mel, z=z)
    assert gpu_engine.last_waveglow_form == 'winograd'
    d_w = np.abs(voc.infer_exact(mel, tile_frames=130, z=z) - full_w).max()
    print('exact tiling max abs diff, default form', d_w)
    assert d_w <= 2e-5 and np.abs(full_w - full).max() <= 2e-5


def test_stream_overlap_two_engines_same_audio(gpu_engine, wg_weights, taco_weights):
    """stream(overlap=True) with the vocoder on its own engine handle (second HIP stream on the same GPU) gives exactly the
    audio of the sequential path: the two halves only exchange host arrays, and each engine is deterministic."""
    from text_to_speech_amd.engine import HipEngine
    from text_to_speech_amd.runtime import HipRuntime
    from text_to_speech_amd.tacotron2 import Tacotron2
    from text_to_speech_amd.waveglow import WaveGlow
    eng2 = HipEngine(0)
    eng2.load_state(wg_weights)
    eng2.finalize()
    model = Tacotron2(HipRuntime('t', model='tacotron2', engine=gpu_engine, seed=0))
    voc_same = WaveGlow(HipRuntime('w', model='waveglow', engine=gpu_engine, seed=0))
    voc_own = WaveGlow(HipRuntime('w2', model='waveglow', engine=eng2, seed=0))
    texts = ['First sentence of the stream.', 'A second, slightly longer sentence follows.', 'Third.', 'And the last one.']
    kw = dict(max_length=5., deterministic=True, save=False)
    seq = model.predict(texts, vocoder=voc_same, **kw)
    ovl = model.predict(texts, vocoder=voc_own, overlap=True, **kw)
    assert [r['text'] for r in ovl] == texts
    for a, b in zip(seq, ovl):
        assert a['audio'].shape == b['audio'].shape and np.array_equal(a['audio'], b['audio'])
    eng2.close()
