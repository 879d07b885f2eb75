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


def _worker(rank, world, port, n_utt, with_spk, ret):
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
    out = synthesize_sharded(tok if rank == 0 else None, _fake_synth, speaker=spk if rank == 0 else None)
    if rank == 0:
        ok = len(out) == n_utt
        for i in range(n_utt):
            expect = tok[i, 0] + (spk[i, 0] if with_spk else 0)
            ok &= out[i].shape == (lens[i] * 7,) and np.allclose(out[i], expect)
        ret.put(bool(ok))
    else:
        assert out is None
    dist.destroy_process_group()


@pytest.mark.parametrize('n_utt,with_spk', [(7, False), (8, True), (1, False)])
def test_scatter_gather_roundtrip_world2(n_utt, with_spk):
    ctx = mp.get_context('spawn')
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_utt, with_spk, ret)) for r in range(2)]
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
