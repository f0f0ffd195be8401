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
# (before the HIP runtime starts: the pipelined decode keeps 4 launch streams and a copy stream on hardware queues of their own; the runtime's default is 4 queues)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
    ap.add_argument("--pipeline", type=int, default=1, help="1 (default, NAIC): the batches go through TransformerModel.decode_many -- 4 launches in flight, 16 batches per "
                    "launch, features copied from pinned host memory ahead of the launches; 0: one synchronised mode='sample' call per batch as the reference's eval loop "
                    "(eval_utils.py:456-460)")
    ap.add_argument("--batches_per_launch", type=int, default=16)
    ap.add_argument("--in_flight", type=int, default=4, help="launch streams (this script starts the runtime with 8 hardware queues: launch streams and the copy stream get one each)")
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

    if args.input_att_npy:
        feats = np.load(args.input_att_npy)
    else:                                                        # (2 048 distinct synthetic images, repeated: the generator is a CPU loop)
        uniq = W.synthetic_att_feats(min(args.synthetic, 2048), 36, model.cfg.att_feat_size, seed=1235)
        feats = uniq if args.synthetic <= 2048 else np.concatenate([uniq] * (-(-args.synthetic // 2048)))[:args.synthetic]
    store = None
    if args.input_label_npz:
        from boficap_amd.data import LabelStore
        from boficap_amd.loss_wrapper import LanguageModelCriterion_UIC
        c = model.cfg
        src = args.input_label_npz if args.input_label_npz.endswith((".h5", ".hdf5")) else dict(np.load(args.input_label_npz))
        store = LabelStore(src, pad_idx=c.pad_idx, bos_idx=c.bos_idx, eos_idx=c.eos_idx, len_idx=c.len_idx)
        crit, rng, loss_sum, loss_evals = LanguageModelCriterion_UIC(), np.random.default_rng(0), 0.0, 0
    results, seconds = [], 0.0

    def entry_of(i, k, seq, pn, pl, ent, ppl):
        ids = [int(v) for v in seq[k].tolist() if v > 0]
        entry = {"image_id": i + k, "seq": ids, "phrase_num": int(pn[k]), "phrase_length": [int(v) for v in pl[k].tolist() if v > 0],
                 "entropy": float(ent[k]), "perplexity": float(ppl[k])}
        if vocab:
            entry["caption"] = " ".join(vocab.get(str(v), "UNK") for v in ids if v > 6)
        return entry

    with torch.no_grad():
        if store is not None:                                    # the loss of eval_split (verbose_loss), eval_utils.py:440-453: a pass of its own
            for i in range(0, len(feats), args.batch_size):
                att = torch.from_numpy(np.ascontiguousarray(feats[i:i + args.batch_size])).cuda()
                fc = torch.zeros(att.size(0), 0, device="cuda")
                hb = store.batch(range(i, i + att.size(0)), args.seq_per_img, rng)
                hb.pop("gts", None)
                b = {k: torch.from_numpy(v).cuda() for k, v in hb.items()}
                outs = model(fc, att.float(), b["labels"], None, b["phrase_num"], b["phrase_length"], b["phrase_syn"],
                             b["extend_phrase_syn_seq"], b["extend_phrase_seq"], b["extend_phrase_seq_mask"])
                loss_sum += float(crit(*outs, b["phrase_num"], b["phrase_length"], b["phrase_syn"], b["labels"])[0])
                loss_evals += 1
        if args.pipeline and args.inference_mode == "NAIC":
            # the features as a loader of half-precision feature files hands them over: compute dtype, pinned host memory (staged once, outside the clock --
            # float32 features would cross PCIe at twice the bytes: ~190 k images/s at 55 GB/s)
            host = torch.from_numpy(np.ascontiguousarray(feats))
            if args.dtype == "bf16":
                host = host.to(torch.bfloat16)
            host = host.pin_memory()
            batches = [host[i:i + args.batch_size] for i in range(0, host.size(0), args.batch_size)]
            for _ in model.decode_many(batches[:2 * args.in_flight * args.batches_per_launch], batches_per_launch=args.batches_per_launch, in_flight=args.in_flight):
                pass                                             # graph captures and stream choice, outside the clock
            torch.cuda.synchronize()
            import time
            t0, got = time.time(), []
            for r in model.decode_many(batches, batches_per_launch=args.batches_per_launch, in_flight=args.in_flight):
                got.append(r)                                    # host tensors of one batch: ids, slot layout, entropy, perplexity
            seconds = time.time() - t0
            i = 0
            for r in got:                                        # (the JSON entries are built outside the clock: ~20 us of Python per image)
                n = r["seq"].size(0)
                results.extend(entry_of(i, k, r["seq"], r["phrase_num"], r["phrase_length"], r["entropy"], r["perplexity"]) for k in range(n))
                i += n
            how = f"pipelined: {args.in_flight} launches in flight, {args.batches_per_launch} batches of {args.batch_size} per launch, features from pinned host memory, host results included"
        else:
            for i in range(0, len(feats), args.batch_size):
                att = torch.from_numpy(np.ascontiguousarray(feats[i:i + args.batch_size])).cuda()
                fc = torch.zeros(att.size(0), 0, device="cuda")
                seq, lp, pn, pl, ps, t = model(fc, att, None, opt={"train_mode": args.inference_mode, "sample_method": "greedy", "sample_n": 1}, mode="sample")
                seconds += t
                # per-image entropy / perplexity as eval_utils.py:463-464, from the fused row reductions (bofi_vocab_stats)
                ent, ppl = model.engine().entropy_perplexity({"seq": seq, "seq_logprob": lp})
                results.extend(entry_of(i, k, seq, pn, pl, ent, ppl) for k in range(att.size(0)))
            how = "one synchronised mode='sample' call per batch (the reference's eval loop), decode time only"
    print(f"decoded {len(results)} images in {seconds:.4f} s ({len(results) / max(seconds, 1e-9):.1f} images/s; {how})")
    if store is not None:
        print(f"validation loss {loss_sum / max(1, loss_evals):.4f} over {loss_evals} batches (LanguageModelCriterion_UIC)")
    if args.dump_json:
        with open(args.dump_json, "w") as f:
            json.dump(results, f)


if __name__ == "__main__":
    main()
