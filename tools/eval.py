#!/usr/bin/env python3
"""Entry point mirroring the reference's tools/eval.py (/root/reference/tools/eval.py:24-44,123) for the
path this repository implements: greedy NAIC bound+fill decoding of precomputed region features.

    python tools/eval.py --model model.pth [--infos_path infos.pkl] --inference_mode NAIC \\
        [--input_att_npy feats.npy | --synthetic 64] [--batch_size 64] [--dtype bf16|f32] [--dump_json out.json]

`--model` is a state_dict written by the reference (311 entries) or by this repository.  Data loading
(lmdb/h5), language evaluation (coco-caption) and beam search are outside the scope of this build
(SURVEY.md §2): features come from a .npy of shape [N, R, 2048] or are synthetic.
"""
import argparse
import json
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="", help="path to model.pth (state_dict); empty = seeded synthetic weights")
    ap.add_argument("--infos_path", default="", help="infos_*.pkl of the reference (opt namespace + vocab)")
    ap.add_argument("--inference_mode", default="NAIC", choices=["NAIC", "SAIC"])
    ap.add_argument("--input_att_npy", default="")
    ap.add_argument("--synthetic", type=int, default=64)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--dump_json", default="")
    ap.add_argument("--input_label_npz", default="", help="label arrays (schema of scripts/prepro_labels_stanford.py:393-399, as .npz, or .h5 with h5py): "
                    "also report the validation loss of eval_split (eval_utils.py:440-453); image i of the features = image i of the file")
    ap.add_argument("--seq_per_img", type=int, default=5)
    args = ap.parse_args()

    import captioning.models as models
    from boficap_amd import weights as W
    from boficap_amd.config import FULL

    vocab = None
    if args.infos_path:
        with open(args.infos_path, "rb") as f:
            infos = pickle.load(f, encoding="latin1")
        opt, vocab = infos["opt"], infos["vocab"]
        opt.vocab = vocab
    else:
        opt = FULL.to_opt()
    opt.bofi_compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    opt.bofi_max_batch = args.batch_size
    model = models.setup(opt)
    if args.model:
        model.load_state_dict(torch.load(args.model, map_location="cpu"), strict=True)
    else:
        model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(model.cfg, 0).items()}, strict=True)
    model.cuda().eval()

    feats = np.load(args.input_att_npy) if args.input_att_npy else W.synthetic_att_feats(args.synthetic, 36, model.cfg.att_feat_size, seed=1235)
    store = None
    if args.input_label_npz:
        from boficap_amd.data import LabelStore
        from boficap_amd.loss_wrapper import LanguageModelCriterion_UIC
        c = model.cfg
        src = args.input_label_npz if args.input_label_npz.endswith((".h5", ".hdf5")) else dict(np.load(args.input_label_npz))
        store = LabelStore(src, pad_idx=c.pad_idx, bos_idx=c.bos_idx, eos_idx=c.eos_idx, len_idx=c.len_idx)
        crit, rng, loss_sum, loss_evals = LanguageModelCriterion_UIC(), np.random.default_rng(0), 0.0, 0
    results, seconds = [], 0.0
    with torch.no_grad():
        for i in range(0, len(feats), args.batch_size):
            att = torch.from_numpy(np.ascontiguousarray(feats[i:i + args.batch_size])).cuda()
            fc = torch.zeros(att.size(0), 0, device="cuda")
            if store is not None:                                # the loss of eval_split (verbose_loss), eval_utils.py:440-453
                hb = store.batch(range(i, i + att.size(0)), args.seq_per_img, rng)
                hb.pop("gts", None)
                b = {k: torch.from_numpy(v).cuda() for k, v in hb.items()}
                outs = model(fc, att.float(), b["labels"], None, b["phrase_num"], b["phrase_length"], b["phrase_syn"],
                             b["extend_phrase_syn_seq"], b["extend_phrase_seq"], b["extend_phrase_seq_mask"])
                loss_sum += float(crit(*outs, b["phrase_num"], b["phrase_length"], b["phrase_syn"], b["labels"])[0])
                loss_evals += 1
            seq, lp, pn, pl, ps, t = model(fc, att, None, opt={"train_mode": args.inference_mode, "sample_method": "greedy", "sample_n": 1}, mode="sample")
            seconds += t
            # per-image entropy / perplexity as eval_utils.py:463-464, from the fused row reductions (bofi_vocab_stats)
            ent, ppl = model.engine().entropy_perplexity({"seq": seq, "seq_logprob": lp})
            for k in range(att.size(0)):
                ids = [int(v) for v in seq[k].tolist() if v > 0]
                entry = {"image_id": i + k, "seq": ids, "phrase_num": int(pn[k]), "phrase_length": [int(v) for v in pl[k].tolist() if v > 0],
                         "entropy": float(ent[k]), "perplexity": float(ppl[k])}
                if vocab:
                    entry["caption"] = " ".join(vocab.get(str(v), "UNK") for v in ids if v > 6)
                results.append(entry)
    print(f"decoded {len(results)} images in {seconds:.4f} s ({len(results) / max(seconds, 1e-9):.1f} images/s incl. host sync)")
    if store is not None:
        print(f"validation loss {loss_sum / max(1, loss_evals):.4f} over {loss_evals} batches (LanguageModelCriterion_UIC)")
    if args.dump_json:
        with open(args.dump_json, "w") as f:
            json.dump(results, f)


if __name__ == "__main__":
    main()
