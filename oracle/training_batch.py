"""Synthetic XE training batch in the layout the reference's collate builds.  TEST INFRASTRUCTURE.

Restates captioning/data/dataloader.py:343-428 (the phrase-aware part of collate_func) for captions
drawn at random: per caption 2..6 phrases of 1..3 tokens, labels [N, S+2] with
[BOS] at position 0, the tokens from position 1 and [EOS] at position S+1 (dataloader.py:295-300).

PINNED (round 5): oracle/make_golden.py calls the reference's own ``Dataset.collate_func`` (dataloader.py:231-452) on sampled captions
that take both branches of the previous-phrase copy, asserts ``collate_loops`` reproduces every tensor it returns and stores them as
tests/golden/tiny_collate.npz; tests/test_oracle_golden.py re-checks this file against that fixture.
"""
from __future__ import annotations

import numpy as np


def make_training_batch(cfg, n_img: int, seq_per_img: int, seed: int = 0):
    rng = np.random.Generator(np.random.PCG64(seed))
    S, L = cfg.seq_length, cfg.seq_length + 2
    N = n_img * seq_per_img
    labels = np.zeros((N, L), np.int64)
    labels[:, 0], labels[:, L - 1] = cfg.bos_idx, cfg.eos_idx     # tmp_label[:, 0] = bos_idx; tmp_label[:, seq_length + 1] = eos_idx
    phrase_num = np.zeros(N, np.int64)                    # real phrases; data['phrase_num'] = this + 1
    plen = np.zeros((N, S), np.int64)
    psyn = np.zeros((N, S), np.int64)
    for n in range(N):
        P = int(rng.integers(2, 7))
        lens = rng.integers(1, 4, P)
        while lens.sum() > S:
            lens = lens[:-1]
        P = len(lens)
        phrase_num[n] = P
        plen[n, :P] = lens
        psyn[n, :P] = rng.integers(4, 7, P)
        ntok = int(lens.sum())
        labels[n, 1:1 + ntok] = rng.integers(7, cfg.tgt_vocab, ntok)     # ids above the special / label range
    b = collate_loops(cfg, labels, phrase_num, plen, psyn)
    sh = lambda a: a.reshape(n_img, seq_per_img, *a.shape[1:])
    return {k: sh(v) for k, v in b.items()}


def collate_loops(cfg, labels, phrase_num, plen, psyn):
    """dataloader.py:343-428 with its loops kept as loops (the checker for boficap_amd.collate.phrase_collate)."""
    N, L = labels.shape
    S = L - 2
    data_phrase_num = phrase_num + 1
    ext_syn = np.zeros((N, L), np.int64)
    ext_syn[:, 0] = cfg.len_idx
    ext_seq = np.zeros((N, S), np.int64)
    ext_mask = np.zeros((N, S, S), bool)
    phrase_length = np.zeros((N, L), np.int64)
    phrase_length[:, 0] = 1
    phrase_syn = np.zeros((N, L), np.int64)
    phrase_syn[:, 0] = cfg.bos_idx
    for ix in range(N):
        P = phrase_num[ix]
        phrase_length[ix, 1:P + 1] = plen[ix, :P]
        phrase_syn[ix, 1:P + 1] = psyn[ix, :P]
        phrase_syn[ix, P + 1] = cfg.eos_idx
        syn_last = 1
        for j in range(P):
            ext_syn[ix, syn_last:syn_last + plen[ix, j]] = psyn[ix, j]
            syn_last += plen[ix, j]
        seq_last = 0
        phrase_last = 0
        for j in range(1, data_phrase_num[ix]):
            cur, prev = phrase_length[ix, j], phrase_length[ix, j - 1]
            if cur <= prev:
                pre_pad = prev - cur
                ext_seq[ix, phrase_last:phrase_last + cur] = labels[ix, seq_last + pre_pad:seq_last + pre_pad + cur]
            else:
                pre_less = prev - (cur % prev)
                times = cur // prev
                copied = 0
                for k in range(prev):
                    n_rep = times if k < pre_less else times + 1
                    ext_seq[ix, phrase_last + copied:phrase_last + copied + n_rep] = labels[ix, seq_last + k]
                    copied += n_rep
            ext_mask[ix, phrase_last:, :phrase_last + cur] = True
            seq_last += prev
            phrase_last += cur
    return dict(labels=labels, phrase_num=data_phrase_num, phrase_length=phrase_length, phrase_syn=phrase_syn,
                extend_phrase_syn_seq=ext_syn, extend_phrase_seq=ext_seq, extend_phrase_seq_mask=ext_mask)
