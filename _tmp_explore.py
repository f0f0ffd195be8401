import sys; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import torch, numpy as np, boficap_oracle as O
from boficap_amd import weights as W
from boficap_amd.config import FULL, TINY
from boficap_amd.engine import BofiEngine
torch.set_num_threads(16)
for name, cfg in (("tiny", TINY), ("full", FULL)):
    sd = W.make_state_dict(cfg, 0)
    w = O.as_torch(sd)
    B = 32
    att_np = W.synthetic_att_feats(B, 36, cfg.att_feat_size, seed=99)
    att = torch.from_numpy(att_np)
    oseq, olp, opn, opl, ops, _ = O.sample_naic(w, cfg, att, fix_q1=True)
    omem, osm = O.memory_of(w, cfg, att)
    for dt in (torch.float32, torch.bfloat16):
        eng = BofiEngine(cfg, dt, max_batch=B, max_regions=36); eng.load_state_dict(sd)
        r = eng.decode_naic(att.cuda(), strict_q1=False, want_memory=True)
        torch.cuda.synchronize()
        same = (r["phrase_length"].cpu() == opl).all(1) & (r["phrase_syn"].cpu() == ops).all(1)
        lp = r["seq_logprob"].cpu()
        d_ = (lp[same] - olp[same]); err = d_.nan_to_num().abs().max().item(); rms = d_.nan_to_num().pow(2).mean().sqrt().item()
        merr = (r["memory"].cpu() - omem).abs().max().item()
        ids = (r["seq"].cpu()[same] == oseq[same]).float().mean().item()
        print(name, dt, "layout agree %d/%d" % (int(same.sum()), B), "max logp err %.4g" % err, "memory err %.4g" % merr, "id agreement %.4f" % ids, "rms %.4g" % rms, "logit std %.3f" % O.logit(w, O.decode_na(w, cfg, omem, torch.zeros(B, cfg.seq_length, dtype=torch.long)+4, osm, torch.ones(B, cfg.seq_length, cfg.seq_length, dtype=torch.bool))).std().item())
