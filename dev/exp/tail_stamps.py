import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BOFI_DBG_PART"] = "1"; os.environ["BOFI_TAIL_DBG"] = "16"
import torch
from boficap_amd import hip as H, weights as W
from boficap_amd.config import FULL as cfg
from boficap_amd.engine import BofiEngine
lib = H.lib()
sd = W.make_state_dict(cfg, seed=0)
eng = BofiEngine(cfg, torch.bfloat16, max_batch=64, max_regions=36); eng.load_state_dict(sd)
att = torch.from_numpy(W.synthetic_att_feats(64, 36, cfg.att_feat_size, seed=1235)).cuda().to(torch.bfloat16)
for rep in range(3):
    r = eng.decode_naic(att, graph=(rep > 0))
torch.cuda.synchronize()
stride = 8 * 200 + cfg.d_model
t = torch.empty(64, stride, device="cuda")
H.check(lib.bofi_engine_debug_copy(eng._h, b"dbg_part", H.ptr(t), t.numel() * 4, H.stream_ptr()))
torch.cuda.synchronize()
st = t[:, :11].cpu()
names = ["entry", "w+y loads issued", "y arrived+mean", "xs ready", "hidden done", "hidden sync", "logits", "serial+sync", "scores+sync", "table acc+sync", "end"]
# the last tail launch that wrote stamps: images unfinished at the last active iteration have full stamps
full = st[st[:, 10] > 0]
print("images with a full set of stamps:", full.shape[0], "(100 MHz ticks: s_memtime counts at a constant 100 MHz)")
med = full.median(0).values
for i, n in enumerate(names):
    print(f"{n:22s} {float(med[i]) / 100:7.2f} us   (+{float(med[i] - med[i - 1]) / 100 if i else 0:5.2f})")
