"""Utterance sharding over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL).

The reference has no multi-device path at all (it loops over sentences at batch 1,
models/tts/tacotron2.py:154); utterances are independent, every rank holds a full weight replica, so the only
communication is: scatter the token batch (+ lengths, + speaker embeddings) from rank 0, gather the per-utterance sample
counts and the padded waveforms back to rank 0.  No all-reduce is on the data path.

Balancing: utterances are sorted longest-first and dealt round-robin, so every rank gets a similar number of
autoregressive decoder steps.

Works unchanged on the gloo backend with CPU tensors (that is what the CPU tests run, world_size 2).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


# What the last scatter / gather on this rank did (the GPU test asserts the device-resident path through these):
# 'scatter_h2d' / 'gather_h2d' / 'gather_d2h' count host<->device copies of the token / waveform payloads made by this module.
last_transfer = {'scatter_h2d': 0, 'gather_h2d': 0, 'gather_d2h': 0, 'gather_device_resident': False}


def partition(lengths, world: int):
    """Longest-first round-robin: returns `world` lists of utterance indices (stable for equal lengths)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return [order[r::world] for r in range(world)]


def _device(backend_device=None):
    if backend_device is not None:
        return backend_device
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def scatter_tokens(tokens, speaker=None, src: int = 0, device=None):
    """Rank `src` passes int32 tokens [N, Tin] (0 = pad) (+ optional speaker [N, E]); every rank receives its shard.

    Returns (local_tokens [n_r, Tin] int32 tensor, local_speaker or None, indices) where `indices` are the global
    utterance ids of the local rows (all ranks know the full assignment).
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    device = _device(device)
    meta = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == src:
        tokens = torch.as_tensor(np.asarray(tokens), dtype=torch.int32)
        meta = torch.tensor([tokens.shape[0], tokens.shape[1], 0 if speaker is None else np.asarray(speaker).shape[1]],
                            dtype=torch.int64, device=device)
    dist.broadcast(meta, src)
    N, Tin, E = (int(v) for v in meta.tolist())
    lens = torch.zeros(N, dtype=torch.int32, device=device)
    if rank == src:
        lens = (tokens != 0).sum(dim=1).to(torch.int32).to(device)
    dist.broadcast(lens, src)
    parts = partition(lens.tolist(), world)
    n_max = max(1, max(len(p) for p in parts))
    # equal-sized padded shards so that one scatter moves everything
    recv = torch.zeros((n_max, Tin), dtype=torch.int32, device=device)
    recv_spk = torch.zeros((n_max, E), dtype=torch.float32, device=device) if E else None
    last_transfer['scatter_h2d'] = 0
    if rank == src:
        # the padded shards of all ranks are ONE [world, n_max, ...] tensor, gathered by row index on `device` after a single
        # upload of the batch (the chunk list passed to scatter = its views); row N of the uploaded batch is the all-zero pad row
        idx = torch.full((world, n_max), N, dtype=torch.int64)
        for r, p in enumerate(parts):
            if p:
                idx[r, :len(p)] = torch.as_tensor(p, dtype=torch.int64)
        idx = idx.to(device)
        tok_d = torch.cat([tokens, torch.zeros((1, Tin), dtype=torch.int32)]).to(device)
        last_transfer['scatter_h2d'] += 1 if device.type == 'cuda' else 0
        chunks = list(tok_d[idx.reshape(-1)].reshape(world, n_max, Tin).unbind(0))
        dist.scatter(recv, chunks, src=src)
        if E:
            spk_t = torch.as_tensor(np.asarray(speaker), dtype=torch.float32)
            spk_d = torch.cat([spk_t, torch.zeros((1, E), dtype=torch.float32)]).to(device)
            last_transfer['scatter_h2d'] += 1 if device.type == 'cuda' else 0
            dist.scatter(recv_spk, list(spk_d[idx.reshape(-1)].reshape(world, n_max, E).unbind(0)), src=src)
    else:
        dist.scatter(recv, None, src=src)
        if E:
            dist.scatter(recv_spk, None, src=src)
    mine = parts[rank]
    return recv[:len(mine)], (recv_spk[:len(mine)] if E else None), parts


def gather_audio(local_audio, local_counts, parts, dst: int = 0, device=None):
    """Every rank passes its waveforms [n_r, S_r] (float32, padded) and sample counts [n_r]; rank `dst` gets a list of
    N numpy arrays in the original utterance order, the others get None.  Tensors that already live on `device` (what
    `TTSPipeline.shard_fn` returns) go into the collective as they are; host arrays (numpy, or CPU tensors under gloo) are
    uploaded first.  The waveforms cross to the host once: rank `dst`'s copy of the gathered block."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = _device(device)
    n_max = max(1, max(len(p) for p in parts))
    n_mine = len(parts[rank])
    on_dev = lambda t: torch.is_tensor(t) and t.device == device
    resident = device.type == 'cuda' and on_dev(local_audio) and on_dev(local_counts)
    last_transfer.update(gather_h2d=0, gather_d2h=0, gather_device_resident=bool(resident))
    counts = torch.zeros(n_max, dtype=torch.int64, device=device)
    if n_mine:
        if not on_dev(local_counts):
            last_transfer['gather_h2d'] += 1 if device.type == 'cuda' else 0
            local_counts = torch.as_tensor(np.asarray(local_counts.cpu() if torch.is_tensor(local_counts) else local_counts),
                                           dtype=torch.int64).to(device)
        counts[:n_mine] = local_counts.to(torch.int64)
    all_counts = torch.zeros((world, n_max), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_counts, counts) if dist.get_backend() == 'nccl' else _all_gather_rows(all_counts, counts)
    counts_h = all_counts.cpu()                                   # [world, n_max] sample counts: every rank sizes its buffer
    s_max = max(1, int(counts_h.max()))
    buf = torch.zeros((n_max, s_max), dtype=torch.float32, device=device)
    if n_mine:
        if not on_dev(local_audio):
            last_transfer['gather_h2d'] += 1 if device.type == 'cuda' else 0
            local_audio = torch.as_tensor(np.asarray(local_audio.cpu() if torch.is_tensor(local_audio) else local_audio),
                                          dtype=torch.float32).to(device)
        w = min(s_max, int(local_audio.shape[1]))
        buf[:local_audio.shape[0], :w] = local_audio[:, :w]
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst)
    if rank != dst:
        return None
    g = torch.stack(gathered).cpu().numpy()                       # the job's one device-to-host waveform copy
    last_transfer['gather_d2h'] += 1 if device.type == 'cuda' else 0
    c = counts_h.numpy()
    N = sum(len(p) for p in parts)
    out = [None] * N
    for r, p in enumerate(parts):
        for row, idx in enumerate(p):
            out[idx] = g[r, row, :int(c[r, row])].copy()
    return out


def _all_gather_rows(dst, row):
    """all_gather of one row per rank into `dst` [world, n] (gloo has no all_gather_into_tensor)."""
    rows = [torch.zeros_like(row) for _ in range(dst.shape[0])]
    dist.all_gather(rows, row)
    for r, t in enumerate(rows):
        dst[r] = t


def synthesize_sharded(tokens, synth_fn, speaker=None, src: int = 0, device=None):
    """scatter -> `synth_fn(local_tokens, local_speaker) -> (audio [n, S], counts [n])` on every rank -> gather."""
    local_tok, local_spk, parts = scatter_tokens(tokens, speaker, src=src, device=device)
    if local_tok.shape[0]:
        audio, counts = synth_fn(local_tok, local_spk)             # numpy arrays or tensors (device tensors stay on the device)
    else:
        audio, counts = np.zeros((0, 1), np.float32), np.zeros((0,), np.int64)
    return gather_audio(audio, counts, parts, dst=src, device=device)


def vocode_long_sharded(mel, vocode_fn, z=None, tile_frames=2048, halo=None, src: int = 0, device=None):
    """One long utterance over all ranks (SURVEY.md section 8e "within one long utterance"): rank `src` passes
    mel [1, T, 80] (+ z [1, T*32, 8]); it is broadcast, every rank vocodes the time tiles r, r + world, ... with an exact
    receptive-field halo (text_to_speech_amd.waveglow.infer_tiled) and rank `src` receives the [T*256] waveform (others
    None).  `vocode_fn(mel, z=None) -> audio [1, T*256]`.  Communication: one broadcast in, one gather out."""
    from .waveglow import HALO_FRAMES, infer_tiled, tile_plan
    world, rank = dist.get_world_size(), dist.get_rank()
    device = _device(device)
    halo = HALO_FRAMES if halo is None else halo
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        mel = torch.as_tensor(np.asarray(mel), dtype=torch.float32)
        if mel.dim() == 2:
            mel = mel[None]
        meta = torch.tensor([mel.shape[1], 0 if z is None else 1], dtype=torch.int64, device=device)
    dist.broadcast(meta, src)
    T, has_z = int(meta[0]), bool(meta[1])
    mel_d = mel.to(device) if rank == src else torch.zeros((1, T, 80), dtype=torch.float32, device=device)
    dist.broadcast(mel_d, src)
    z_d = None
    if has_z:
        z_d = (torch.as_tensor(np.asarray(z), dtype=torch.float32).to(device) if rank == src
               else torch.zeros((1, T * 32, 8), dtype=torch.float32, device=device))
        dist.broadcast(z_d, src)
    plan = tile_plan(T, tile_frames, halo)
    mine = list(range(rank, len(plan), world))
    on_gpu = device.type == 'cuda'
    m_in = mel_d if on_gpu else mel_d.numpy()
    z_in = None if z_d is None else (z_d if on_gpu else z_d.numpy())
    pieces = infer_tiled(vocode_fn, m_in, z=z_in, tile_frames=tile_frames, halo=halo, tiles=mine)
    # equal-sized padded contributions -> one gather
    n_max = max(1, (len(plan) + world - 1) // world)
    buf = torch.zeros((n_max, tile_frames * 256), dtype=torch.float32, device=device)
    for row, (start, stop, a) in enumerate(pieces):
        buf[row, :(stop - start) * 256] = torch.as_tensor(a, dtype=torch.float32).reshape(-1).to(device)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == src else None
    dist.gather(buf, gathered, dst=src)
    if rank != src:
        return None
    out = np.zeros((T * 256,), np.float32)
    for r in range(world):
        g = gathered[r].cpu().numpy()
        for row, i in enumerate(range(r, len(plan), world)):
            start, stop, _, _ = plan[i]
            out[start * 256:stop * 256] = g[row, :(stop - start) * 256]
    return out
