"""CPU, world_size 2 (gloo): utterance scatter / gather keeps order, balances lengths, handles empty shards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_synth(tok, spk):
    """audio for an utterance with n tokens = n*7 samples of value (first token id) (+ speaker[0])."""
    tok = tok.cpu().numpy()
    n = (tok != 0).sum(1)
    S = max(1, int(n.max()) * 7)
    audio = np.zeros((tok.shape[0], S), np.float32)
    for i in range(tok.shape[0]):
        audio[i, :n[i] * 7] = tok[i, 0] + (0 if spk is None else float(spk[i, 0]))
    return audio, n * 7


def _fake_synth_tensors(tok, spk):
    """The same audio as torch tensors on the collective's device (what `TTSPipeline.shard_fn` returns on a GPU): they go into
    the gather as they are."""
    audio, counts = _fake_synth(tok, spk)
    return torch.from_numpy(audio), torch.from_numpy(np.asarray(counts, dtype=np.int64))


def _worker(rank, world, port, n_utt, with_spk, ret, tensors=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from text_to_speech_amd.distributed import synthesize_sharded
    rng = np.random.default_rng(0)
    lens = rng.integers(3, 20, n_utt)
    tok = np.zeros((n_utt, 24), np.int32)
    for i, n in enumerate(lens):
        tok[i, :n] = rng.integers(1, 148, n)
    spk = rng.standard_normal((n_utt, 4)).astype(np.float32) if with_spk else None
    out = synthesize_sharded(tok if rank == 0 else None, _fake_synth_tensors if tensors else _fake_synth,
                             speaker=spk if rank == 0 else None)
    if rank == 0:
        ok = len(out) == n_utt
        for i in range(n_utt):
            expect = tok[i, 0] + (spk[i, 0] if with_spk else 0)
            ok &= out[i].shape == (lens[i] * 7,) and np.allclose(out[i], expect)
        ret.put(bool(ok))
    else:
        assert out is None
    dist.destroy_process_group()


@pytest.mark.parametrize('n_utt,with_spk,tensors', [(7, False, False), (8, True, False), (1, False, False), (7, True, True), (1, False, True)])
def test_scatter_gather_roundtrip_world2(n_utt, with_spk, tensors):
    ctx = mp.get_context('spawn')
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_utt, with_spk, ret, tensors)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get() is True


def test_partition_balances_decoder_steps():
    from text_to_speech_amd.distributed import partition
    lens = [50, 70, 90, 110, 130, 150, 170, 200] * 4              # BASELINE config 4: 32 utterances over 8 GPUs
    parts = partition(lens, 8)
    assert sorted(i for p in parts for i in p) == list(range(32)) and all(len(p) == 4 for p in parts)
    loads = [sum(lens[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 150                         # vs 600 for a contiguous split
    assert partition([5, 5, 5], 2) == [[0, 2], [1]]


# ---- one long utterance over two ranks: exact halo tiling ---------------------------------------------------------------
def _fir_vocoder(mel, z=None):
    """Finite-receptive-field stand-in for WaveGlow: sample n of frame t = sum_{|k| <= 60} c_k * mel[t + k, n % 80] (zero
    outside the sequence) + 0.5 * z[...]; radius 60 frames < the 100-frame halo, so tiling must be exact."""
    mel = np.asarray(mel, np.float64)
    B, T, _ = mel.shape
    k = np.arange(-60, 61)
    c = np.cos(k * 0.37) / (1.0 + np.abs(k))
    acc = np.zeros((B, T, 80))
    for kk, ck in zip(k, c):
        lo, hi = max(0, -kk), min(T, T - kk)
        acc[:, lo:hi] += ck * mel[:, lo + kk:hi + kk]
    audio = np.tile(acc, (1, 1, 4))[:, :, :256].reshape(B, T * 256)
    if z is not None:
        audio = audio + 0.5 * np.asarray(z, np.float64).reshape(B, T * 256)
    return audio.astype(np.float32)


def test_tile_plan_and_exact_tiling_on_cpu():
    from text_to_speech_amd.waveglow import HALO_FRAMES, infer_tiled, tile_plan
    assert HALO_FRAMES * 32 >= 12 * 255 + 3 * 32                  # receptive field of 12 flows + upsampling window
    assert tile_plan(250, 100) == [(0, 100, 0, 200), (100, 200, 0, 250), (200, 250, 100, 250)]
    rng = np.random.default_rng(0)
    mel = rng.standard_normal((2, 333, 80)).astype(np.float32)
    z = rng.standard_normal((2, 333 * 32, 8)).astype(np.float32)
    full = _fir_vocoder(mel, z)
    tiled = infer_tiled(_fir_vocoder, mel, z=z, tile_frames=70)
    assert tiled.shape == full.shape and np.array_equal(tiled, full)
    # a halo shorter than the receptive field is NOT exact (the test would be vacuous otherwise)
    assert not np.array_equal(infer_tiled(_fir_vocoder, mel, z=z, tile_frames=70, halo=20), full)


def _long_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from text_to_speech_amd.distributed import vocode_long_sharded
    rng = np.random.default_rng(1)
    mel = rng.standard_normal((1, 517, 80)).astype(np.float32)
    z = rng.standard_normal((1, 517 * 32, 8)).astype(np.float32)
    out = vocode_long_sharded(mel if rank == 0 else None, _fir_vocoder, z=z if rank == 0 else None, tile_frames=128)
    if rank == 0:
        ret.put(bool(out.shape == (517 * 256,) and np.array_equal(out, _fir_vocoder(mel, z)[0])))
    else:
        assert out is None
    dist.destroy_process_group()


def test_long_utterance_sharded_over_two_ranks_is_exact():
    ctx = mp.get_context('spawn')
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_long_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get() is True
