"""In-kernel stamps (BOFI_RB_DBG=16, workgroup 0, s_memtime ticks) and time per launch of the attention sublayer kernels at the benchmark shapes:
encoder self-attention (36 x 36, 16 wavefronts), filling-pass self-attention (20 x 20) and cross-attention (20 x 36) in their in-flight forms (16 wavefronts).
    python dev/exp/attn_stamps.py [images]"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 320
d = 512
from boficap_amd import hip as H
L = H.lib()
L.bofi_rb_stamps.restype = C.c_int; L.bofi_rb_stamps.argtypes = [C.c_void_p]
g = torch.Generator().manual_seed(1)
bf = lambda t: t.to(torch.bfloat16).cuda()
def pack(w):
    out = torch.empty(w.numel(), dtype=torch.bfloat16, device="cuda")
    H.check(L.bofi_pack_frag(H.ptr(w), H.ptr(out), w.shape[0], w.shape[1], H.stream_ptr()))
    return out
wo = pack(bf(torch.randn(d, d, generator=g) / math.sqrt(d)))
bo = torch.randn(d, generator=g).cuda() * 0.1
names = ["entry", "constants + barrier", "attention", "barrier", "block written + barrier", "W_o segment", "barrier", "closing stores"]
for what, Lq, Lk, W in (("encoder self-attention", 36, 36, 16), ("filling self-attention", 20, 20, 16), ("filling cross-attention", 20, 36, 16), ("filling self-attention", 20, 20, 8),
                        ("filling cross-attention", 20, 36, 8)):
    os.environ["BOFI_RB_ATTN_W"] = str(W); os.environ.pop("BOFI_RB_DBG", None); L.bofi_reload_env()
    if Lq == Lk:
        qkv = bf(torch.randn(B * Lq, 3 * d, generator=g)); q, k, v, ldq, ldk = qkv, qkv[:, d:], qkv[:, 2 * d:], 3 * d, 3 * d
    else:
        q = bf(torch.randn(B * Lq, d, generator=g)); kv = bf(torch.randn(B * Lk, 14 * d, generator=g)); k, v, ldq, ldk = kv[:, 2 * d:], kv[:, 3 * d:], d, 14 * d
    x = torch.randn(B * Lq, d, generator=g).cuda()
    def run():
        H.check(L.bofi_attn_block(H.ptr(q), ldq, H.ptr(k), ldk, H.ptr(v), ldk, B, Lq, Lk, None, 0, 0, 0, 0, H.ptr(wo), H.ptr(bo), H.ptr(x), d, H.ptr(x), d, None, None, H.stream_ptr()))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    os.environ["BOFI_RB_DBG"] = "16"
    run(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 256)()
    H.check(L.bofi_rb_stamps(buf))
    t = [buf[i] for i in range(8)]
    print(f"{what} {Lq} x {Lk}, {W} wavefronts, {B} images: {us:.1f} us per launch; wavefront 0 of workgroup 0: " +
          "  ".join(f"{names[i]} +{t[i] - t[i - 1]}" for i in range(1, 8)) + f"   total {t[7] - t[0]} ticks")
