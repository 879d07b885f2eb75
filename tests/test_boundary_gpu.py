"""GPU: the stream-ordered half of the C ABI: tts_hip_tacotron2_encode / _decode, the *_async calls
on a caller's stream, replay of cached decoder graphs, and the encoder reuse of the reference's retry loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tokens(B, Tin, lens, seed=0):
    rng = np.random.default_rng(seed)
    tok = rng.integers(1, 148, (B, Tin)).astype(np.int32)
    for b, n in enumerate(lens):
        tok[b, n:] = 0
    return tok


@pytest.mark.parametrize('mode', ['persistent', 'fused', 'graph'])
def test_encode_then_decode_equals_infer_and_can_be_repeated(gpu_engine, taco_weights, taco_cfg, mode):
    from oracle import tacotron2_ref
    gpu_engine.set_decoder_mode(mode)
    try:
        tok = _tokens(2, 33, [33, 21], seed=3)
        rng = np.random.default_rng(1)
        m1 = (rng.random((2, 24, 2, 256)) >= 0.5).astype(np.float32) * 2.0
        m2 = (rng.random((2, 40, 2, 256)) >= 0.5).astype(np.float32) * 2.0
        whole = gpu_engine.tacotron2_infer(tok, max_len=24, early_stopping=False, prenet_masks=m1)
        enc = gpu_engine.tacotron2_encode(tok)
        a = gpu_engine.tacotron2_decode(enc, max_len=24, early_stopping=False, prenet_masks=m1)
        assert gpu_engine.last_decoder_mode == mode
        for k in ('mel', 'decoder_output', 'stop_tokens', 'attention_weights', 'lengths'):
            assert np.array_equal(getattr(a, k), getattr(whole, k)), k          # same kernels, same inputs
        # a second decode of the same encoded batch: other masks, other length (the retry loop of tacotron2.py:160-179)
        b = gpu_engine.tacotron2_decode(enc, max_len=40, early_stopping=False, prenet_masks=m2)
        ref = tacotron2_ref.infer(tok, taco_weights, taco_cfg, max_length=40, early_stopping=False, prenet_masks=m2)
        assert np.abs(b.mel - ref.mel).max() <= 1e-3 and np.array_equal(b.lengths, ref.lengths)
        # and the first one again: a cached graph of the 256-step bucket is replayed with different masks / max_len
        c = gpu_engine.tacotron2_decode(enc, max_len=24, early_stopping=False, prenet_masks=m1)
        assert np.array_equal(c.mel, a.mel)
        enc.close()
        with pytest.raises(ValueError, match='freed'):
            gpu_engine.tacotron2_decode(enc, max_len=4)
    finally:
        gpu_engine.set_decoder_mode('auto')


def test_cached_decoder_graphs_survive_shape_changes(gpu_engine, taco_weights, taco_cfg):
    """Per-step graph path: calls alternate between shapes (each has its own cached executable graph), a larger call grows
    the workspace (which drops every cached graph), and every result still matches the oracle."""
    from oracle import tacotron2_ref
    gpu_engine.set_decoder_mode('graph')
    try:
        cases = [(1, 20, 40), (3, 26, 37), (1, 20, 33), (3, 26, 37), (9, 12, 70), (1, 20, 40)]
        for i, (B, Tin, T) in enumerate(cases):
            tok = _tokens(B, Tin, [Tin - b for b in range(B)], seed=B)
            ref = tacotron2_ref.infer(tok, taco_weights, taco_cfg, max_length=T, early_stopping=False)
            out = gpu_engine.tacotron2_infer(tok, max_len=T, early_stopping=False)
            assert gpu_engine.last_decoder_mode == 'graph' and gpu_engine.last_steps == T
            assert np.abs(out.mel - ref.mel).max() <= 1e-3, (i, B, Tin, T)
            assert np.abs(out.attention_weights - ref.attention_weights).max() <= 1e-3
    finally:
        gpu_engine.set_decoder_mode('auto')


def test_async_calls_on_a_torch_stream_match_the_blocking_calls(gpu_engine):
    import torch
    rng = np.random.default_rng(0)
    mel = torch.from_numpy(rng.uniform(-11.5, 1.2, (2, 12, 80)).astype(np.float32)).cuda()
    z = torch.from_numpy(rng.standard_normal((2, 12 * 32, 8)).astype(np.float32)).cuda()
    wav = torch.from_numpy(rng.uniform(-1, 1, (2, 5000)).astype(np.float32)).cuda()
    ref_audio = {p: gpu_engine.waveglow_infer(mel, z=z, precision=p) for p in ('f32', 'f16', 'f16x3')}
    ref_mel = gpu_engine.mel_stft(wav)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        mel_s = mel * 1.0                                       # produced on `s`: the engine call must be ordered after it
        outs = {p: gpu_engine.waveglow_infer(mel_s, z=z, precision=p, stream=s) for p in ('f32', 'f16', 'f16x3')}
        m = gpu_engine.mel_stft(wav, stream=s)
        doubled = outs['f32'] * 2.0                              # consumed on `s` without any host synchronization
    s.synchronize()
    for p in outs:
        assert torch.equal(outs[p], ref_audio[p]), p
    assert torch.equal(m, ref_mel) and torch.equal(doubled, ref_audio['f32'] * 2.0)
    with pytest.raises(ValueError, match='device tensors'):
        gpu_engine.waveglow_infer(mel.cpu().numpy(), stream=s)


def test_runtime_retry_reuses_the_encoder(gpu_engine, taco_weights, taco_cfg):
    """The reference's retry loop calls compiled_infer again with the same tokens (fresh dropout): the runtime keeps the
    encoded batch and only re-runs the decoder; a different sentence encodes again."""
    from oracle import tacotron2_ref
    from text_to_speech_amd.runtime import HipRuntime
    rt = HipRuntime('retry', model='tacotron2', engine=gpu_engine, seed=0)
    tok = _tokens(1, 30, [30], seed=8)
    a = rt(tok, max_length=12, early_stopping=False, seed=1)
    b = rt(tok, max_length=12, early_stopping=False, seed=2)                     # retry: new masks, same tokens
    assert rt.encoder_reuses == 1 and not np.array_equal(a.mel, b.mel)
    c = rt(tok, max_length=12, early_stopping=False, deterministic=True)
    ref = tacotron2_ref.infer(tok, taco_weights, taco_cfg, max_length=12, early_stopping=False)
    assert rt.encoder_reuses == 2 and np.abs(c.mel - ref.mel).max() <= 1e-3
    rt(_tokens(1, 30, [30], seed=9), max_length=12, early_stopping=False, deterministic=True)
    assert rt.encoder_reuses == 2


def test_two_handles_decoding_at_once_share_the_gpu_correctly(gpu_engine, taco_weights):
    """The persistent decoder needs every CU: when two handles decode at the same time (two threads, two HIP streams), the
    second grid waits at its start-up rendezvous until the first has finished -- or gives up cleanly and runs the per-step
    graph.  Either way both results must equal the ones computed alone."""
    import threading
    from text_to_speech_amd.engine import HipEngine
    eng2 = HipEngine(0)
    eng2.load_state(taco_weights)
    eng2.finalize()
    try:
        toks = [_tokens(1, 60, [60], seed=21), _tokens(2, 45, [45, 30], seed=22)]
        engines = [gpu_engine, eng2]
        alone = [e.tacotron2_infer(t, max_len=300, early_stopping=False, want_attention=False) for e, t in zip(engines, toks)]
        for _ in range(3):
            res, modes = [None, None], [None, None]

            def run(i):
                res[i] = engines[i].tacotron2_infer(toks[i], max_len=300, early_stopping=False, want_attention=False)
                modes[i] = engines[i].last_decoder_mode

            th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            print('concurrent decoder paths:', modes)
            for i in range(2):
                # a fallback to the per-step graph differs by fp32 re-association only
                assert np.abs(res[i].mel - alone[i].mel).max() <= 1e-4, (i, modes)
                assert np.array_equal(res[i].lengths, alone[i].lengths)
    finally:
        eng2.close()


def test_stream_calls_order_their_conversions_on_that_stream(gpu_engine):
    """`stream=` with inputs that need a conversion (fp16, non-contiguous, short audio to pad) produced on torch's CURRENT
    stream: the conversion temporaries and the outputs must be ordered on `stream`, and must not be handed back to the
    caching allocator while the engine kernels still read them (round-2 advisor finding).  Right after every call the default
    stream allocates and scribbles over same-sized tensors, which is what recycled the temporaries before the fix."""
    import torch
    rng = np.random.default_rng(3)
    B, T = 2, 24
    mel32 = torch.from_numpy(rng.uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)).cuda()
    z32 = torch.from_numpy(rng.standard_normal((B, T * 32, 8)).astype(np.float32)).cuda()
    wav = torch.from_numpy(rng.uniform(-1, 1, (3, 900)).astype(np.float32)).cuda()
    ref_audio = gpu_engine.waveglow_infer(mel32.half().float(), z=z32.half().float())
    ref_mel = gpu_engine.mel_stft(wav.half().float())
    s = torch.cuda.Stream()
    for _ in range(6):
        big = torch.randn(4096, 4096, device='cuda')
        big = big @ big                                           # keeps the default stream busy while we enqueue
        mel_nc = (mel32.transpose(1, 2).contiguous() + 0 * big[0, 0]).half().transpose(1, 2)    # fp16, non-contiguous, late
        assert not mel_nc.is_contiguous()
        wav16 = (wav + 0 * big[0, 0]).half()
        out_a = gpu_engine.waveglow_infer(mel_nc, z=z32.half(), stream=s)
        out_m = gpu_engine.mel_stft(wav16, stream=s)
        junk = [torch.full_like(mel32, float('nan')), torch.full_like(z32, float('nan')),
                torch.full((3, 1024), float('nan'), device='cuda')]
        del junk, mel_nc, wav16
        s.synchronize()
        assert torch.equal(out_a, ref_audio) and torch.equal(out_m, ref_mel)
    # seeded noise on a stream: same values as the blocking seeded call
    a = gpu_engine.waveglow_infer(mel32, seed=3, offset=9, stream=s)
    s.synchronize()
    assert torch.equal(a, gpu_engine.waveglow_infer(mel32, seed=3, offset=9))


def test_prenet_masks_drawn_on_the_device(gpu_engine, taco_weights, taco_cfg):
    """`mask_seed` = (seed, offset): the decoder's dropout masks come from the engine's documented Philox stream; the result
    equals the explicit-mask call with the restated stream, and (through it) the oracle.  Also through the runtime."""
    from oracle import philox_ref, tacotron2_ref
    from text_to_speech_amd.runtime import HipRuntime
    tok = _tokens(3, 22, [22, 15, 9], seed=8)
    T = 20
    masks = philox_ref.prenet_masks(3 * T * 512, 41, 6).reshape(3, T, 2, 256)
    ref = tacotron2_ref.infer(tok, taco_weights, taco_cfg, max_length=T, early_stopping=False, prenet_masks=masks)
    for mode in ('fused', 'graph'):
        gpu_engine.set_decoder_mode(mode)
        try:
            enc = gpu_engine.tacotron2_encode(tok)
            a = gpu_engine.tacotron2_decode(enc, max_len=T, early_stopping=False, mask_seed=(41, 6))
            b = gpu_engine.tacotron2_decode(enc, max_len=T, early_stopping=False, prenet_masks=masks)
            assert np.array_equal(a.mel, b.mel) and np.abs(a.mel - ref.mel).max() <= 1e-3
            enc.close()
        finally:
            gpu_engine.set_decoder_mode('auto')
    rt = HipRuntime('unused', engine=gpu_engine, model='tacotron2', seed=41)
    first = rt(tok, max_length=T, early_stopping=False)                    # offset 0 of seed 41
    from text_to_speech_amd.runtime import MASK_STREAM                     # the runtime's dropout key: seed ^ purpose constant
    m0 = philox_ref.prenet_masks(3 * T * 512, 41 ^ MASK_STREAM, 0).reshape(3, T, 2, 256)
    ref0 = tacotron2_ref.infer(tok, taco_weights, taco_cfg, max_length=T, early_stopping=False, prenet_masks=m0)
    assert np.abs(first.mel - ref0.mel).max() <= 1e-3
    second = rt(tok, max_length=T, early_stopping=False)                   # a retry: same encoder output, other masks
    assert rt.encoder_reuses >= 1 and not np.array_equal(first.mel, second.mel)
    assert np.array_equal(rt(tok, max_length=T, early_stopping=False, seed=41).mel, first.mel)


def test_runtime_reuses_one_encoded_batch_handle_across_sentences(gpu_engine, taco_weights, taco_cfg):
    """Sentence after sentence through `HipRuntime`: the encoder writes into the SAME encoded-batch buffer
    (tts_hip_tacotron2_reencode), so the per-step graphs cached for a shape are replayed by the next sentence of that shape
    instead of being dropped with a freed buffer (round-2 advisor finding), and every result still matches the oracle."""
    from oracle import tacotron2_ref
    from text_to_speech_amd.runtime import HipRuntime
    gpu_engine.set_decoder_mode('graph')
    try:
        rt = HipRuntime('unused', engine=gpu_engine, model='tacotron2')
        handles = set()
        for seed, (B, Tin) in enumerate([(1, 30), (1, 30), (2, 30), (1, 30), (1, 44)]):
            tok = _tokens(B, Tin, [Tin - 3 * b for b in range(B)], seed=100 + seed)
            out = rt(tok, max_length=20, early_stopping=False, deterministic=True)
            ref = tacotron2_ref.infer(tok, taco_weights, taco_cfg, max_length=20, early_stopping=False)
            assert np.abs(out.mel - ref.mel).max() <= 1e-3, (seed, B, Tin)
            handles.add(rt._encoded[1].handle.value)
        assert len(handles) == 1 and rt.encoder_reuses == 0
    finally:
        gpu_engine.set_decoder_mode('auto')


def test_runtime_never_reuses_the_encoder_for_other_device_tokens(gpu_engine, taco_weights, taco_cfg):
    """Round-3 advisor finding: the encoder-reuse key of DEVICE inputs was (address, version, shape).  A caller that builds a
    fresh int64 token tensor per sentence gets the freed tensor's address back from torch's caching allocator -- same key,
    other tokens -- and the decoder ran on the previous sentence's encoder output.  The key compares contents now: two
    different sentences of one shape, each created, used and freed back to back, must each match the oracle; the same
    sentence again (the reference's retry, models/tts/tacotron2.py:160-179) must still reuse the encoder, also from a
    re-created tensor and after an in-place change of the caller's tensor has been undone."""
    import torch
    from oracle import tacotron2_ref
    from text_to_speech_amd.runtime import HipRuntime
    rt = HipRuntime('unused', engine=gpu_engine, model='tacotron2')
    toks = [_tokens(1, 40, [40], seed=200 + i) for i in range(3)]
    refs = [tacotron2_ref.infer(t, taco_weights, taco_cfg, max_length=16, early_stopping=False) for t in toks]
    ptrs = set()
    for t, ref in zip(toks, refs):
        dev_tok = torch.from_numpy(t.astype(np.int64)).cuda()           # torch's default integer type: a new tensor per sentence
        ptrs.add(dev_tok.data_ptr())
        out = rt(dev_tok, max_length=16, early_stopping=False, deterministic=True)
        assert np.abs(out.mel.cpu().numpy() - ref.mel).max() <= 1e-3
        del dev_tok, out
    print(f'{len(ptrs)} distinct token addresses over 3 sentences')     # (1 when the allocator recycles the block)
    assert rt.encoder_reuses == 0
    again = torch.from_numpy(toks[2].astype(np.int64)).cuda()           # same sentence, new tensor: the encoder is reused
    out = rt(again, max_length=16, early_stopping=False, deterministic=True)
    assert rt.encoder_reuses == 1 and np.abs(out.mel.cpu().numpy() - refs[2].mel).max() <= 1e-3
    again[0, 3] = (int(again[0, 3]) % 147) + 1                          # changed in place: another sentence at the same address
    changed = toks[2].copy()
    changed[0, 3] = int(again[0, 3])
    out = rt(again, max_length=16, early_stopping=False, deterministic=True)
    ref_c = tacotron2_ref.infer(changed, taco_weights, taco_cfg, max_length=16, early_stopping=False)
    assert rt.encoder_reuses == 1 and np.abs(out.mel.cpu().numpy() - ref_c.mel).max() <= 1e-3


def test_two_handles_on_the_fused_step_take_turns(gpu_engine, taco_weights):
    """The fused two-kernel step needs every CU at once, like the persistent kernel: two handles of one process decoding at the
    same time are serialised by a per-device lock inside the library (two half-resident grids would wait for each other until
    their bounded waits give up and both calls fall back).  Both calls must stay on the fused step and equal their solo runs."""
    import threading
    from text_to_speech_amd.engine import HipEngine
    eng2 = HipEngine(0)
    eng2.load_state(taco_weights)
    eng2.finalize()
    try:
        engines = [gpu_engine, eng2]
        for e in engines:
            e.set_decoder_mode('fused')
        toks = [_tokens(4, 60, [60, 50, 33, 60], seed=31), _tokens(6, 45, [45, 30, 45, 12, 40, 45], seed=32)]
        alone = [e.tacotron2_infer(t, max_len=200, early_stopping=False, want_attention=False) for e, t in zip(engines, toks)]
        for _ in range(3):
            res, modes = [None, None], [None, None]

            def run(i):
                res[i] = engines[i].tacotron2_infer(toks[i], max_len=200, early_stopping=False, want_attention=False)
                modes[i] = engines[i].last_decoder_mode

            th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert modes == ['fused', 'fused'], modes
            for i in range(2):
                assert np.array_equal(res[i].mel, alone[i].mel) and np.array_equal(res[i].lengths, alone[i].lengths)
    finally:
        gpu_engine.set_decoder_mode('auto')
        eng2.close()
