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
    if rank == src:
        chunks, spk_chunks = [], []
        spk_t = torch.as_tensor(np.asarray(speaker), dtype=torch.float32) if E else None
        for p in parts:
            c = torch.zeros((n_max, Tin), dtype=torch.int32)
            if p:
                c[:len(p)] = tokens[p]
            chunks.append(c.to(device))
            if E:
                s = torch.zeros((n_max, E), dtype=torch.float32)
                if p:
                    s[:len(p)] = spk_t[p]
                spk_chunks.append(s.to(device))
        dist.scatter(recv, chunks, src=src)
        if E:
            dist.scatter(recv_spk, spk_chunks, src=src)
    else:
        dist.scatter(recv, None, src=src)
        if E:
            dist.scatter(recv_spk, None, src=src)
    mine = parts[rank]
    return recv[:len(mine)], (recv_spk[:len(mine)] if E else None), parts


def gather_audio(local_audio, local_counts, parts, dst: int = 0, device=None):
    """Every rank passes its waveforms [n_r, S_r] (float32, padded) and sample counts [n_r]; rank `dst` gets a list of
    N numpy arrays in the original utterance order, the others get None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = _device(device)
    n_max = max(1, max(len(p) for p in parts))
    counts = torch.zeros(n_max, dtype=torch.int64, device=device)
    if len(parts[rank]):
        counts[:len(parts[rank])] = torch.as_tensor(np.asarray(local_counts), dtype=torch.int64).to(device)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    s_max = max(1, int(max(int(c.max()) for c in all_counts)))
    buf = torch.zeros((n_max, s_max), dtype=torch.float32, device=device)
    if len(parts[rank]):
        la = torch.as_tensor(local_audio, dtype=torch.float32).to(device)
        buf[:la.shape[0], :min(s_max, la.shape[1])] = la[:, :s_max]
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst)
    if rank != dst:
        return None
    N = sum(len(p) for p in parts)
    out = [None] * N
    for r, p in enumerate(parts):
        g = gathered[r].cpu().numpy()
        c = all_counts[r].cpu().numpy()
        for row, idx in enumerate(p):
            out[idx] = g[row, :int(c[row])].copy()
    return out


def synthesize_sharded(tokens, synth_fn, speaker=None, src: int = 0, device=None):
    """scatter -> `synth_fn(local_tokens, local_speaker) -> (audio [n, S], counts [n])` on every rank -> gather."""
    local_tok, local_spk, parts = scatter_tokens(tokens, speaker, src=src, device=device)
    if local_tok.shape[0]:
        audio, counts = synth_fn(local_tok, local_spk)
    else:
        audio, counts = np.zeros((0, 1), np.float32), np.zeros((0,), np.int64)
    return gather_audio(audio, counts, parts, dst=src, device=device)
