"""The label file of the reference's preprocessing (scripts/prepro_labels_stanford.py:389-399) -> training batches.

``<output_h5>_label.h5`` holds, for M captions of N images (all uint32):
    labels [M, max_length]            token ids, 0-padded
    label_start_ix / label_end_ix [N] 1-based caption range of every image
    label_length [M]                  caption lengths
    phrase_num [M]                    phrases per caption
    phrase_length [M, max_length]     tokens per phrase
    phrase_label [M, max_length]      syntactic label per phrase (VP 4 / NP 5 / CP 6, prepro_labels_stanford.py:49-51)
``LabelStore`` reads those arrays (from an open h5py file, a path when h5py is installed, or any mapping of arrays with those
names) and restates the loader's caption sampling (captioning/data/dataloader.py:186-229) and the label framing of its collate
(:295-300); the phrase-aware part of the collate is ``boficap_amd.collate.phrase_collate`` (vectorised restatement of :343-428).
Region features come from elsewhere (lmdb / npz directories in the reference): this class handles the label side only.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from .collate import phrase_collate

def collate_regions(att_list: Sequence[np.ndarray]):
    """The region side of the loader's collate (captioning/data/dataloader.py:326-338): images' region features [r_i, F] stacked into a zero-padded
    float32 [B, max r_i, F] and a float32 mask [B, max r_i] of ones over the real regions -- ``None`` when every image has the same count, as the
    loader returns it (the model then builds an all-ones mask itself, TransformerModel.py:1684-1686).  Pinned by tests/golden/tiny_collate.npz."""
    R = max(a.shape[0] for a in att_list)
    feats = np.zeros((len(att_list), R, att_list[0].shape[1]), np.float32)
    masks = np.zeros((len(att_list), R), np.float32)
    for i, a in enumerate(att_list):
        feats[i, :a.shape[0]] = a
        masks[i, :a.shape[0]] = 1
    return feats, (None if masks.sum() == masks.size else masks)


FIELDS = ("labels", "label_start_ix", "label_end_ix", "label_length", "phrase_num", "phrase_length", "phrase_label")


class LabelStore:
    def __init__(self, source, *, pad_idx=0, bos_idx=1, eos_idx=2, len_idx=3):
        if isinstance(source, str):
            try:
                import h5py
            except ImportError as e:
                raise ImportError("reading a label .h5 needs h5py; pass an open file or a mapping of arrays instead") from e
            with h5py.File(source, "r") as f:
                arrays = {k: f[k][:] for k in FIELDS if k in f}
        else:
            arrays = {k: np.asarray(source[k][:] if hasattr(source[k], "shape") else source[k]) for k in FIELDS if k in source}
        missing = [k for k in FIELDS if k not in arrays and k != "label_length"]
        if missing:
            raise KeyError(f"label file lacks {missing} (schema: scripts/prepro_labels_stanford.py:389-399)")
        self.label = arrays["labels"].astype(np.int64)
        self.label_start_ix = arrays["label_start_ix"].astype(np.int64)
        self.label_end_ix = arrays["label_end_ix"].astype(np.int64)
        self.phrase_num = arrays["phrase_num"].astype(np.int64)
        self.phrase_length = arrays["phrase_length"].astype(np.int64)
        self.phrase_syn = arrays["phrase_label"].astype(np.int64)
        self.seq_length = int(self.label.shape[1])                 # dataloader.py:136
        self.num_images = int(self.label_start_ix.shape[0])
        self.pad_idx, self.bos_idx, self.eos_idx, self.len_idx = pad_idx, bos_idx, eos_idx, len_idx
        M = self.label.shape[0]
        if not (self.phrase_num.shape == (M,) and self.phrase_length.shape[0] == M and self.phrase_syn.shape[0] == M):
            raise ValueError("phrase arrays and labels disagree on the number of captions")

    def captions(self, ix: int, seq_per_img: int, rng: np.random.Generator):
        """get_captions_and_phrase (dataloader.py:202-229): ``seq_per_img`` captions of image ``ix`` -- a random run of
        consecutive ones, or draws with replacement when the image has fewer."""
        ix1, ix2 = int(self.label_start_ix[ix]) - 1, int(self.label_end_ix[ix]) - 1       # 1-based in the file
        ncap = ix2 - ix1 + 1
        assert ncap > 0, "an image does not have any label"
        if ncap < seq_per_img:
            rows = rng.integers(ix1, ix2 + 1, seq_per_img)
        else:
            first = int(rng.integers(ix1, ix2 - seq_per_img + 2))
            rows = np.arange(first, first + seq_per_img)
        S = self.seq_length
        return self.label[rows, :S], self.phrase_num[rows], self.phrase_length[rows, :S], self.phrase_syn[rows, :S]

    def gts(self, ix: int) -> np.ndarray:
        """All ground-truth captions of an image (dataloader.py:304), for the reward scorer."""
        return self.label[int(self.label_start_ix[ix]) - 1: int(self.label_end_ix[ix])]

    def batch(self, image_ixs: Sequence[int], seq_per_img: int, rng: Optional[np.random.Generator] = None) -> Dict[str, np.ndarray]:
        """The label side of one training batch in the loader's layout ([B, seq_per_img, ...], dataloader.py:231-452):
        labels framed by [BOS] / [EOS] (:295-300), phrase_num (+1), phrase_length, phrase_syn, extend_phrase_syn_seq,
        extend_phrase_seq, extend_phrase_seq_mask, plus ``gts`` (list of arrays)."""
        rng = rng if rng is not None else np.random.default_rng()
        S = self.seq_length
        seqs, pns, pls, pss, gts = [], [], [], [], []
        for ix in image_ixs:
            seq, pn, pl, ps = self.captions(ix, seq_per_img, rng)
            seqs.append(seq); pns.append(pn); pls.append(pl); pss.append(ps)
            gts.append(self.gts(ix))
        seq = np.vstack(seqs)
        pn = np.concatenate(pns)
        pl, ps = np.vstack(pls), np.vstack(pss)
        keep = np.arange(S)[None, :] < pn[:, None]                  # entries past phrase_num are not read by the collate (:369-371)
        pl, ps = np.where(keep, pl, 0), np.where(keep, ps, 0)
        labels = np.zeros((seq.shape[0], S + 2), np.int64)
        labels[:, 1:S + 1] = seq
        labels[:, 0] = self.bos_idx
        labels[:, S + 1] = self.eos_idx
        b = phrase_collate(labels, pl, ps, pad_idx=self.pad_idx, bos_idx=self.bos_idx, eos_idx=self.eos_idx, len_idx=self.len_idx)
        B = len(image_ixs)
        out = {k: v.reshape(B, seq_per_img, *v.shape[1:]) for k, v in b.items()}
        out["gts"] = gts
        return out
