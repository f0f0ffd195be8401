"""Phrase-aware collate of the XE training batch, vectorised (NumPy), and a synthetic caption sampler.

The reference builds ``extend_phrase_syn_seq``, ``extend_phrase_seq`` and ``extend_phrase_seq_mask`` with
O(B*P*L) Python loops inside ``collate_func`` (captioning/data/dataloader.py:343-428).  They sit directly upstream
of the XE step, so here they are index arithmetic over whole [N, S] arrays:

  * token t of a caption belongs to phrase ``pid[t]`` = number of phrase ends <= t (a cumulative-sum compare);
  * the SA decoder input of a phrase is the PREVIOUS phrase squeezed or stretched to the current length
    (dataloader.py:396-412): position k of the current phrase reads source position
        prev - cur + k                         if cur <= prev   (the last `cur` tokens)
        k // times                             if k <  pre_less * times
        pre_less + (k - pre_less*times) // (times+1)   otherwise,
    with times = cur // prev and pre_less = prev - cur % prev;
  * the SA self-attention mask row r allows the keys up to the end of r's own phrase (dataloader.py:414) -- a key
    prefix, which is what the attention kernels consume.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def phrase_collate(labels: np.ndarray, phrase_len: np.ndarray, phrase_syn_real: np.ndarray, *, pad_idx=0, bos_idx=1, eos_idx=2,
                   len_idx=3) -> Dict[str, np.ndarray]:
    """labels int64 [N, S+2] ([BOS] at position 0, tokens from position 1, [EOS] at S+1 as the loader frames them,
    dataloader.py:295-300); phrase_len / phrase_syn_real int64 [N, S]
    (real phrases first, zeros after).  Returns the loader's phrase tensors, un-grouped ([N, ...])."""
    labels = np.asarray(labels, np.int64)
    plen = np.asarray(phrase_len, np.int64)
    psyn = np.asarray(phrase_syn_real, np.int64)
    N, L = labels.shape
    S = L - 2
    if plen.shape != (N, S) or psyn.shape != (N, S):
        raise ValueError("phrase_len / phrase_syn_real must be [N, S]")
    P = (plen > 0).sum(1)                                   # real phrases per caption
    if ((plen > 0) != (np.arange(S)[None] < P[:, None])).any():
        raise ValueError("phrase lengths must be a prefix of positive entries")
    ntok = plen.sum(1)
    if (ntok > S).any():
        raise ValueError("caption longer than seq_length")
    rows = np.arange(N)[:, None]

    phrase_length = np.zeros((N, L), np.int64)
    phrase_length[:, 0] = 1
    phrase_length[:, 1:S + 1] = plen
    phrase_syn = np.zeros((N, L), np.int64)
    phrase_syn[:, 0] = bos_idx
    phrase_syn[:, 1:S + 1] = psyn
    phrase_syn[np.arange(N), P + 1] = eos_idx

    ends = plen.cumsum(1)                                   # [N, S] end (exclusive) of phrase j in token coordinates
    t = np.arange(S)[None, :]
    valid = t < ntok[:, None]
    pid = (ends[:, None, :] <= t[:, :, None]).sum(2)        # [N, S] phrase index (0-based) of token t
    pid = np.minimum(pid, S - 1)
    ext_syn = np.zeros((N, L), np.int64)
    ext_syn[:, 0] = len_idx
    ext_syn[:, 1:S + 1] = np.where(valid, psyn[rows, pid], 0)

    start = ends - plen                                      # start of phrase j (token coordinates)
    cur = plen[rows, pid]
    k = t - start[rows, pid]
    # previous phrase in LABEL coordinates: phrase 0 is the single position 0, real phrase j starts at 1 + start[j]
    first = pid == 0
    prev = np.where(first, 1, plen[rows, np.maximum(pid - 1, 0)])
    prev_start = np.where(first, 0, 1 + start[rows, np.maximum(pid - 1, 0)])
    cur_s, prev_s = np.maximum(cur, 1), np.maximum(prev, 1)
    times = cur_s // prev_s
    pre_less = prev_s - cur_s % prev_s
    stretched = np.where(k < pre_less * times, k // np.maximum(times, 1),
                         pre_less + (k - pre_less * times) // (times + 1))
    src = np.where(cur <= prev, prev - cur + k, stretched)
    ext_seq = np.where(valid, labels[rows, np.clip(prev_start + src, 0, L - 1)], 0)

    klen = np.where(valid, ends[rows, pid], ntok[:, None])   # keys each SA row may see
    ext_mask = np.arange(S)[None, None, :] < klen[:, :, None]
    return dict(labels=labels, phrase_num=P + 1, phrase_length=phrase_length, phrase_syn=phrase_syn,
                extend_phrase_syn_seq=ext_syn, extend_phrase_seq=ext_seq, extend_phrase_seq_mask=ext_mask)


def synthetic_captions(cfg, n_captions: int, seed: int = 0):
    """Random captions for benchmarks and tests: 2..6 phrases of 1..3 tokens, syntactic labels in 4..6, token ids
    above the special/label range, framed by [BOS] / [EOS].  Returns (labels [N, S+2], phrase_len [N, S], phrase_syn_real [N, S])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    S, L = cfg.seq_length, cfg.seq_length + 2
    labels = np.zeros((n_captions, L), np.int64)
    labels[:, 0], labels[:, L - 1] = cfg.bos_idx, cfg.eos_idx             # as the loader frames a caption (dataloader.py:295-300)
    plen = np.zeros((n_captions, S), np.int64)
    psyn = np.zeros((n_captions, S), np.int64)
    for n in range(n_captions):
        lens = rng.integers(1, 4, int(rng.integers(2, 7)))
        while lens.sum() > S:
            lens = lens[:-1]
        P = len(lens)
        plen[n, :P] = lens
        psyn[n, :P] = rng.integers(4, 7, P)
        ntok = int(lens.sum())
        labels[n, 1:1 + ntok] = rng.integers(7, cfg.tgt_vocab, ntok)
    return labels, plen, psyn


def synthetic_training_batch(cfg, n_img: int, seq_per_img: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """The loader's batch layout ([n_img, seq_per_img, ...]) for synthetic captions."""
    labels, plen, psyn = synthetic_captions(cfg, n_img * seq_per_img, seed)
    b = phrase_collate(labels, plen, psyn, pad_idx=cfg.pad_idx, bos_idx=cfg.bos_idx, eos_idx=cfg.eos_idx, len_idx=cfg.len_idx)
    return {k: v.reshape(n_img, seq_per_img, *v.shape[1:]) for k, v in b.items()}


def max_tokens(batch) -> int:
    """Length of the longest caption of the batch (host value): the training forward computes that many decoder positions."""
    return int((np.asarray(batch["phrase_length"]).sum(-1) - 1).max())


def max_phrase_num(batch) -> int:
    """max over the batch of the loader's phrase_num (host value; lets the training forward skip a device->host read)."""
    return int(np.asarray(batch["phrase_num"]).max())
