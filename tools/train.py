#!/usr/bin/env python3
"""Entry point mirroring the reference's tools/train.py for the XE phase of a UIC model
(/root/reference/tools/train.py:97-101 model + DataParallel, :150-170 schedules, :198-229 the step, :304-360 checkpoints).

    python tools/train.py [--cfg configs/uic_sd.yml] [--id run] [--checkpoint_path DIR] [--max_iters 100]
                          [--batch_size 10] [--seq_per_img 5] [--dtype bf16|f32] [--glancing_token 1] [--tiny]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train.py ...     # data parallel

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment) instead of ``nn.DataParallel``: every rank
steps on its own shard of images and the gradients meet in one RCCL all-reduce over the flat gradient bucket
(boficap_amd/trainer.py).  Data loading (lmdb / h5), language evaluation and the self-critical phase are outside this
build: batches are synthetic captions in the loader's layout (boficap_amd/collate.py), or -- with
``--input_label_h5`` -- real captions from the preprocessing's label file (boficap_amd/data.py; region features stay synthetic,
the feature directories being outside this build).  ``--cfg`` reads the reference's yml files including their ``_BASE_``
inheritance (captioning/utils/config.py:35-95).  ``--self_critical_after N`` switches to the self-critical step (loss_wrapper.py:181-230,
``structure_loss_weight: 1``) from iteration N on, scored by ``boficap_amd.loss_wrapper``'s installed scorer or, without one, by
token overlap with the ground-truth captions (the reference's CIDEr-D scorer is an external package).
Checkpoints are the reference's files (captioning/utils/misc.py:87-102): ``model.pth`` (311-entry state_dict), ``optimizer.pth``
(NoamOpt / torch Adam layout), ``infos_<id>.pkl``, ``histories_<id>.pkl``; ``--start_from`` resumes from either code base's directory.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="", help="yml with the reference's option names (configs/uic_sd.yml)")
    ap.add_argument("--id", default="bofi")
    ap.add_argument("--checkpoint_path", default="")
    ap.add_argument("--start_from", default="", help="directory holding model.pth / optimizer.pth to resume from")
    ap.add_argument("--max_iters", type=int, default=50)
    ap.add_argument("--batch_size", type=int, default=None, help="images per rank and step (uic_sd.yml: 10)")
    ap.add_argument("--seq_per_img", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"], help="GEMM operand dtype (fp32 accumulation and master weights either way)")
    ap.add_argument("--glancing_token", type=int, default=0, help="1: GLAT with the reference's unmasked-rate schedule start value")
    ap.add_argument("--unmasked_rate_start", type=float, default=0.5)
    ap.add_argument("--losses_log_every", type=int, default=10)
    ap.add_argument("--save_checkpoint_every", type=int, default=0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--tiny", action="store_true", help="the small test configuration instead of the 512-d model")
    ap.add_argument("--no_graph", action="store_true", help="run the step eagerly instead of replaying one hipGraph per batch signature")
    ap.add_argument("--input_label_h5", default="", help="label file of scripts/prepro_labels_stanford.py (captions and phrase cuts); needs h5py")
    ap.add_argument("--self_critical_after", type=int, default=-1, help="iteration from which the self-critical step replaces the XE step (-1: never)")
    ap.add_argument("--train_sample_n", type=int, default=5)
    ap.add_argument("--scheduled_sampling_start", type=int, default=None, help="epoch from which ss_prob rises (opts.py:153-160; -1: never, the shipped configs)")
    ap.add_argument("--drop_worst_after", type=int, default=None, help="epoch from which a step keeps the best (1 - drop_worst_rate) captions (opts.py:165-168, "
                    "tools/train.py:186-189, 216-220; -1: never, the shipped configs)")
    ap.add_argument("--iters_per_epoch", type=int, default=1000, help="the synthetic stream has no epochs of its own: iterations that count as one")
    args = ap.parse_args()

    import captioning.models as models
    from boficap_amd import checkpoint as ck, dp, weights as W
    from boficap_amd.collate import synthetic_training_batch
    from boficap_amd.config import FULL, TINY
    from boficap_amd.trainer import XETrainer

    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)                                   # before the process group: RCCL binds to the current device
    rank, local_rank, world = dp.init_from_env("nccl")
    opt = (TINY if args.tiny else FULL).to_opt()
    if args.cfg:
        for k, v in ck.load_yaml_with_base(args.cfg).items():
            setattr(opt, k, v)
    for k in ("batch_size", "seq_per_img"):
        if getattr(args, k) is not None:
            setattr(opt, k, getattr(args, k))
    opt.batch_size, opt.seq_per_img = getattr(opt, "batch_size", 10), getattr(opt, "seq_per_img", 5)
    # the dropout streams of the ranks differ (nn.DataParallel replicas draw from per-device generators); the weights' seed does not
    opt.seed = args.seed + 1000003 * rank
    opt.id, opt.checkpoint_path = args.id, args.checkpoint_path
    opt.bofi_train_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if not hasattr(opt, "vocab"):
        opt.vocab = {str(i): f"w{i}" for i in range(1, getattr(opt, "vocab_size", 9487) + 1)}
    torch.manual_seed(args.seed)                                 # same initial weights on every rank (train.py:29-37)
    model = models.setup(opt)
    if args.start_from:
        model.load_state_dict(torch.load(os.path.join(args.start_from, "model.pth"), map_location="cpu"), strict=True)
    model.to(dev).train()
    trainer = XETrainer(model, opt, graph=not args.no_graph)
    infos, histories = ck.new_infos(opt.vocab), {}           # incl. 'loader_state_dict': the reference's resume path indexes it (train.py:125)
    if args.start_from:
        if os.path.exists(os.path.join(args.start_from, "optimizer.pth")):
            trainer.load_state_dict(torch.load(os.path.join(args.start_from, "optimizer.pth"), map_location="cpu", weights_only=False))
        saved, histories = ck.load_infos(args.start_from, args.id)
        if saved:
            if "opt" in saved:
                ck.check_resume_opts(saved["opt"], opt)            # tools/train.py:66-68
            infos.update({k: saved[k] for k in ("iter", "epoch", "best_val_score") if k in saved})
    for k in ("loss_history", "lr_history", "ss_prob_history", "val_result_history"):
        histories.setdefault(k, {})
    cfg = model.cfg
    store = None
    if args.input_label_h5:
        from boficap_amd.data import LabelStore
        store = LabelStore(args.input_label_h5, pad_idx=cfg.pad_idx, bos_idx=cfg.bos_idx, eos_idx=cfg.eos_idx, len_idx=cfg.len_idx)
    if rank == 0:
        print(f"{type(model).__name__}: {trainer.bucket.numel} parameters in one bucket, {world} rank(s), "
              f"{opt.batch_size} images x {opt.seq_per_img} captions per rank and step, GEMM operands {args.dtype}", flush=True)

    import numpy as np
    from boficap_amd import loss_wrapper as LW
    t0, first = time.time(), trainer._step
    for it in range(first, first + args.max_iters):
        # a different shard per rank and step (the loader's role, dataloader.py:550-556)
        seed = (it * world + rank) * 7919 + args.seed
        gts = None
        if store is None:
            host_batch = synthetic_training_batch(cfg, opt.batch_size, opt.seq_per_img, seed=seed)
        else:
            rng = np.random.default_rng(seed)
            host_batch = store.batch(rng.integers(0, store.num_images, opt.batch_size), opt.seq_per_img, rng)
            gts = host_batch.pop("gts")
        ss_start = args.scheduled_sampling_start if args.scheduled_sampling_start is not None else getattr(opt, "scheduled_sampling_start", -1)
        epoch = it // max(1, args.iters_per_epoch)
        infos["epoch"] = epoch
        if ss_start >= 0 and epoch >= ss_start:                 # scheduled sampling probability (tools/train.py:159-162)
            frac = (epoch - ss_start) // getattr(opt, "scheduled_sampling_increase_every", 5) + 1
            opt.ss_prob = min(getattr(opt, "scheduled_sampling_increase_prob", 0.05) * frac, getattr(opt, "scheduled_sampling_max_prob", 0.25))
            model.ss_prob = opt.ss_prob
        if 0 <= args.self_critical_after <= it:                 # struc_flag (tools/train.py:181-185)
            att = torch.from_numpy(W.synthetic_att_feats(opt.batch_size, 36, cfg.att_feat_size, seed=seed)).to(dev)
            refs = gts if gts is not None else [host_batch["labels"][b, :, 1:-1] for b in range(opt.batch_size)]
            n = args.train_sample_n

            def score(seq, refs=refs, n=n):
                if LW._SCORER["fn"] is not None:
                    return LW._SCORER["fn"](refs, seq)
                out = torch.zeros(seq.shape[0])                  # stand-in: best token overlap with one of the image's captions
                for i, s_ in enumerate(seq.numpy()):
                    toks = set(int(t) for t in s_ if t > 0)
                    out[i] = max((len(toks & set(int(t) for t in r if t > 0)) / max(1, len(toks)) for r in np.asarray(refs[i // n])), default=0.0)
                return out
            loss, rs, rn = trainer.rl_step(att, None, score, sample_n=n)
            if (it + 1) % args.losses_log_every == 0 or it == first:
                if rank == 0:
                    print(f"iter {it + 1} lr {trainer.rate():.3e} struc_loss {float(loss):.4f} reward SAIC {float(rs):.4f} NAIC {float(rn):.4f} "
                          f"{(time.time() - t0) / (it + 1 - first):.3f} s/it", flush=True)
                histories["loss_history"][it + 1] = float(loss)
            infos["iter"] = it + 1
            continue
        batch = {k: torch.from_numpy(v).to(dev) for k, v in host_batch.items()}
        batch["max_phrase_num"] = int(host_batch["phrase_num"].max())
        batch["max_tokens"] = int((host_batch["phrase_length"].sum(-1) - 1).max())
        batch["att_feats"] = torch.from_numpy(W.synthetic_att_feats(opt.batch_size, 36, cfg.att_feat_size, seed=seed)).to(dev)
        batch = trainer.add_token_rows(batch, host_batch)
        glat_p = args.unmasked_rate_start if args.glancing_token else -1.0          # train.py:165-170
        dw_after = args.drop_worst_after if args.drop_worst_after is not None else getattr(opt, "drop_worst_after", -1)
        drop_worst = dw_after != -1 and epoch >= dw_after                         # train.py:186-189
        try:
            loss, parts = trainer.step(batch, glat_p, drop_worst)
        except FloatingPointError as e:                          # scheduled sampling on a model whose SA bounding step opens no phrase for
            if not model.ss_prob > 0:                            # some caption: the reference crashes there (TransformerModel.py:2103-2105);
                raise                                            # this batch is stepped teacher-forced instead
            if rank == 0:
                print(f"iter {it + 1}: {e}; teacher-forced step for this batch", flush=True)
            keep, model.ss_prob = model.ss_prob, 0.0
            loss, parts = trainer.step(batch, glat_p, drop_worst)
            model.ss_prob = keep
        if (it + 1) % args.losses_log_every == 0 or it == 0:
            mean_loss = dp.reduce_scalar(float(loss), "sum", device=dev) / world
            if rank == 0:
                names = ("sa_len", "sa_tok", "sa_syn", "na_len", "na_tok", "na_syn")
                detail = " ".join(f"{n}={float(p):.3f}" for n, p in zip(names, parts))
                print(f"iter {it + 1} lr {trainer.rate():.3e} train_loss {mean_loss:.4f} ({detail}) "
                      f"{(time.time() - t0) / (it + 1 - first):.3f} s/it", flush=True)
            histories["loss_history"][it + 1] = mean_loss         # tools/train.py:262-266
            histories["lr_history"][it + 1] = trainer.rate()
            histories["ss_prob_history"][it + 1] = model.ss_prob
        infos["iter"] = it + 1                                   # tools/train.py:292-294
        if args.checkpoint_path and args.save_checkpoint_every and (it + 1) % args.save_checkpoint_every == 0 and rank == 0:
            infos["opt"] = ck.resume_opt(opt)
            ck.save_checkpoint(opt, model, infos, trainer, histories)
    if args.checkpoint_path and rank == 0:
        infos["opt"] = ck.resume_opt(opt)                     # plain values, every option of the reference's resume check present
        ck.save_checkpoint(opt, model, infos, trainer, histories)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
