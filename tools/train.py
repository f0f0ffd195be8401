#!/usr/bin/env python3
"""Entry point mirroring the reference's tools/train.py for the XE phase of a UIC model
(/root/reference/tools/train.py:97-101 model + DataParallel, :150-170 schedules, :198-229 the step, :304-360 checkpoints).

    python tools/train.py [--cfg configs/uic_sd.yml] [--id run] [--checkpoint_path DIR] [--max_iters 100]
                          [--batch_size 10] [--seq_per_img 5] [--dtype bf16|f32] [--glancing_token 1] [--tiny]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train.py ...     # data parallel

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment) instead of ``nn.DataParallel``: every rank
steps on its own shard of images and the gradients meet in one RCCL all-reduce over the flat gradient bucket
(boficap_amd/trainer.py).  Data loading (lmdb / h5), language evaluation and the self-critical phase are outside this
build: batches are synthetic captions in the loader's layout (boficap_amd/collate.py), which is what
the training-path benchmark uses as well.  ``--cfg`` reads the keys of the reference's yml that this path uses.
Checkpoints are the reference's files: ``model.pth`` (311-entry state_dict) and ``optimizer.pth``.
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
    args = ap.parse_args()

    import captioning.models as models
    from boficap_amd import dp, weights as W
    from boficap_amd.collate import synthetic_training_batch
    from boficap_amd.config import FULL, TINY
    from boficap_amd.trainer import XETrainer

    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)                                   # before the process group: RCCL binds to the current device
    rank, local_rank, world = dp.init_from_env("nccl")
    opt = (TINY if args.tiny else FULL).to_opt()
    if args.cfg:
        import yaml
        with open(args.cfg) as f:
            for k, v in (yaml.safe_load(f) or {}).items():
                setattr(opt, k, v)
    for k in ("batch_size", "seq_per_img"):
        if getattr(args, k) is not None:
            setattr(opt, k, getattr(args, k))
    opt.batch_size, opt.seq_per_img = getattr(opt, "batch_size", 10), getattr(opt, "seq_per_img", 5)
    opt.seed = args.seed
    opt.bofi_train_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if not hasattr(opt, "vocab"):
        opt.vocab = {str(i): f"w{i}" for i in range(1, getattr(opt, "vocab_size", 9487) + 1)}
    torch.manual_seed(args.seed)                                 # same initial weights on every rank (train.py:29-37)
    model = models.setup(opt)
    if args.start_from:
        model.load_state_dict(torch.load(os.path.join(args.start_from, "model.pth"), map_location="cpu"), strict=True)
    model.to(dev).train()
    trainer = XETrainer(model, opt, graph=not args.no_graph)
    if args.start_from and os.path.exists(os.path.join(args.start_from, "optimizer.pth")):
        trainer.load_state_dict(torch.load(os.path.join(args.start_from, "optimizer.pth"), map_location="cpu"))
    cfg = model.cfg
    if rank == 0:
        print(f"{type(model).__name__}: {trainer.bucket.numel} parameters in one bucket, {world} rank(s), "
              f"{opt.batch_size} images x {opt.seq_per_img} captions per rank and step, GEMM operands {args.dtype}", flush=True)

    t0, first = time.time(), trainer._step
    for it in range(first, first + args.max_iters):
        # a different synthetic shard per rank and step (the loader's role, dataloader.py:550-556)
        seed = (it * world + rank) * 7919 + args.seed
        host_batch = synthetic_training_batch(cfg, opt.batch_size, opt.seq_per_img, seed=seed)
        batch = {k: torch.from_numpy(v).to(dev) for k, v in host_batch.items()}
        batch["max_phrase_num"] = int(host_batch["phrase_num"].max())
        batch["max_tokens"] = int((host_batch["phrase_length"].sum(-1) - 1).max())
        batch["att_feats"] = torch.from_numpy(W.synthetic_att_feats(opt.batch_size, 36, cfg.att_feat_size, seed=seed)).to(dev)
        batch = trainer.add_token_rows(batch, host_batch)
        glat_p = args.unmasked_rate_start if args.glancing_token else -1.0          # train.py:165-170
        loss, parts = trainer.step(batch, glat_p)
        if (it + 1) % args.losses_log_every == 0 or it == 0:
            mean_loss = dp.reduce_scalar(float(loss), "sum", device=dev) / world
            if rank == 0:
                names = ("sa_len", "sa_tok", "sa_syn", "na_len", "na_tok", "na_syn")
                detail = " ".join(f"{n}={float(p):.3f}" for n, p in zip(names, parts))
                print(f"iter {it + 1} lr {trainer.rate():.3e} train_loss {mean_loss:.4f} ({detail}) "
                      f"{(time.time() - t0) / (it + 1 - first):.3f} s/it", flush=True)
        if args.checkpoint_path and args.save_checkpoint_every and (it + 1) % args.save_checkpoint_every == 0 and rank == 0:
            save(model, trainer, args.checkpoint_path)
    if args.checkpoint_path and rank == 0:
        save(model, trainer, args.checkpoint_path)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def save(model, trainer, path):
    """model.pth + optimizer.pth, captioning/utils/misc.py:87-102."""
    os.makedirs(path, exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(path, "model.pth"))
    torch.save(trainer.state_dict(), os.path.join(path, "optimizer.pth"))
    print(f"checkpoint saved to {path}", flush=True)


if __name__ == "__main__":
    main()
