import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import boficap_oracle as O
from boficap_amd import weights as W
from boficap_amd.config import TINY as cfg
from boficap_amd.engine import BofiEngine
from conftest import load_golden
sd = W.make_state_dict(cfg, seed=0, gen_scale=1.0)
lut = sd["model.tgt_embed.lut.weight"].copy(); lut[cfg.len_idx] = sd["model.syn_embed.lut.weight"][cfg.len_idx]; sd["model.tgt_embed.lut.weight"] = lut
g = load_golden("tiny_mix")
att = torch.from_numpy(g["att_feats"][g["naic_phrase_num"] > 0])
w = O.as_torch(sd)
seq, lp, pn, pl, ps, _ = O.sample_saic(w, cfg, att)
print("oracle pn", pn.tolist()); print("oracle seq0", seq[0].tolist()); print("oracle pl0", pl[0].tolist())
eng = BofiEngine(cfg, torch.float32, max_batch=64, max_regions=36); eng.load_state_dict(sd)
r = eng.decode_saic(att.cuda())
print("engine pn", r["phrase_num"].tolist()); print("engine seq0", r["seq"][0].tolist()); print("engine pl0", r["phrase_length"][0].tolist())
print("seq equal", torch.equal(r["seq"].cpu(), seq), "pl equal", torch.equal(r["phrase_length"].cpu(), pl))
d = (r["seq_logprob"].cpu() - lp)
print("max logprob diff", float(d.nan_to_num().abs().max()))
