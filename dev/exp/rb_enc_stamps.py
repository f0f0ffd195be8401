"""In-kernel timeline of rb_encoder_kernel (workgroup 0, s_memtime stamps = shader clock, BOFI_RB_DBG=16): per layer the phases
[block + statistics | q, k, v, scores, context | output projection | feed-forward].  python dev/exp/rb_enc_stamps.py [B]"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BOFI_RB_DBG"] = "16"
from boficap_amd import hip as H
L = H.lib()
L.bofi_rb_stamps.restype = C.c_int; L.bofi_rb_stamps.argtypes = [C.c_void_p]
d, dff, dev, R, nl = 512, 2048, "cuda", 36, 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def pack(w):
    out = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
    H.check(L.bofi_pack_frag(H.ptr(w), H.ptr(out), w.shape[0], w.shape[1], H.stream_ptr()))
    return out
keep, arrs = [], {k: [] for k in ("wqkv", "cqkv", "csqkv", "wo", "bo", "w1", "c1", "cs1", "w2", "b2")}
for _ in range(nl):
    wqkv = (torch.randn(3 * d, d, device=dev) / math.sqrt(d)).bfloat16(); wo = (torch.randn(d, d, device=dev) / math.sqrt(d)).bfloat16()
    w1 = (torch.randn(dff, d, device=dev) / math.sqrt(d)).bfloat16(); w2 = (torch.randn(d, dff, device=dev) / math.sqrt(dff)).bfloat16()
    t = {"wqkv": pack(wqkv), "cqkv": torch.randn(3 * d, device=dev) * 0.1, "csqkv": wqkv.float().sum(1), "wo": pack(wo), "bo": torch.randn(d, device=dev) * 0.1,
         "w1": pack(w1), "c1": torch.randn(dff, device=dev) * 0.1, "cs1": w1.float().sum(1), "w2": pack(w2), "b2": torch.randn(d, device=dev) * 0.1}
    keep.append(t)
    for k, v in t.items():
        arrs[k].append(H.ptr(v))
ptrs = {k: (C.c_void_p * nl)(*v) for k, v in arrs.items()}
x, y = torch.randn(B * R, d, device=dev), torch.empty(B * R, d, device=dev)
klen = torch.full((B,), R, dtype=torch.int32, device=dev)
run = lambda: H.check(L.bofi_encoder_block(H.ptr(x), H.ptr(y), H.ptr(klen), B, R, nl, ptrs["wqkv"], ptrs["cqkv"], ptrs["csqkv"], ptrs["wo"], ptrs["bo"],
                                           ptrs["w1"], ptrs["c1"], ptrs["cs1"], ptrs["w2"], ptrs["b2"], dff, H.stream_ptr()))
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
H.check(L.bofi_rb_stamps(buf))
for w in (0, 3, 7):
    s = [buf[w * 16 + k] for k in range(13)]
    t0 = s[0]
    print(f"B {B} wavefront {w}: stats {s[1] - t0} | context in block {s[2] - t0} | output projection {s[3] - t0} | layer 0 done {s[4] - t0} | layer 1 done {s[5] - t0} | exit {s[12] - t0}"
          f"   (slots 1-3 hold the LAST layer's stamps: relative to layer 0's end: {s[1] - s[4]} {s[2] - s[4]} {s[3] - s[4]})")
