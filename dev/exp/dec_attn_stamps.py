"""The fused decoder-layer attention kernel (rb_dec_attn_kernel) alone at the benchmark shape: time per launch against the three launches it replaces
(rb_attn + rb_gemm<false, 6> + rb_attn), and the in-kernel stamps of workgroup 0 (BOFI_RB_DBG=16; s_memtime ticks):
    python dev/exp/dec_attn_stamps.py [images] [regions]"""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 320
R = int(sys.argv[2]) if len(sys.argv) > 2 else 36
S, d = 20, 512
from boficap_amd import hip as H
L = H.lib()
L.bofi_rb_stamps.restype = C.c_int; L.bofi_rb_stamps.argtypes = [C.c_void_p]
g = torch.Generator().manual_seed(1)
bf = lambda t: t.to(torch.bfloat16).cuda()
def pack(w):
    out = torch.empty(w.numel(), dtype=torch.bfloat16, device="cuda")
    H.check(L.bofi_pack_frag(H.ptr(w), H.ptr(out), w.shape[0], w.shape[1], H.stream_ptr()))
    return out
qkv, kv = bf(torch.randn(B * S, 3 * d, generator=g)), bf(torch.randn(B * R, 14 * d, generator=g))
wo1, wo2, wq = (pack(bf(torch.randn(d, d, generator=g) / math.sqrt(d))) for _ in range(3))
bo1, bo2, cq, csq = (torch.randn(d, generator=g).cuda() * 0.1 for _ in range(4))
last = torch.randint(2, S + 2, (B,), generator=g).int().cuda()
alen = torch.full((B,), R, dtype=torch.int32).cuda()
x0 = torch.randn(B * S, d, generator=g).cuda()
qs = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
sp = H.stream_ptr
def fused(x):
    H.check(L.bofi_decoder_attn_block(H.ptr(qkv), 3 * d, B, S, H.ptr(last), -1, 64, H.ptr(wo1), H.ptr(bo1), H.ptr(kv[:, 2 * d:]), H.ptr(kv[:, 3 * d:]), 14 * d, R, H.ptr(alen),
                                      H.ptr(wq), H.ptr(cq), H.ptr(csq), H.ptr(wo2), H.ptr(bo2), H.ptr(x), d, None, None, sp()))
def three(x):
    H.check(L.bofi_attn_block(H.ptr(qkv), 3 * d, H.ptr(qkv[:, d:]), 3 * d, H.ptr(qkv[:, 2 * d:]), 3 * d, B, S, S, H.ptr(last), 1, 0, -1, 64, H.ptr(wo1), H.ptr(bo1), H.ptr(x), d,
                              H.ptr(x), d, None, None, sp()))
    H.check(L.bofi_linear_block(H.ptr(x), d, H.ptr(wq), H.ptr(cq), H.ptr(csq), H.ptr(qs), d, 0, B * S, d, 0, sp()))
    H.check(L.bofi_attn_block(H.ptr(qs), d, H.ptr(kv[:, 2 * d:]), 14 * d, H.ptr(kv[:, 3 * d:]), 14 * d, B, S, R, H.ptr(alen), 1, 0, 0, 0, H.ptr(wo2), H.ptr(bo2), H.ptr(x), d,
                              H.ptr(x), d, None, None, sp()))
def timed(fn, n=50):
    x = x0.clone()
    for _ in range(5):
        fn(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"{B} images x {S} rows, {R} regions: one launch {timed(fused):.1f} us, three launches {timed(three):.1f} us (back to back on one stream, eager)")
os.environ["BOFI_RB_DBG"] = "16"
x = x0.clone(); fused(x); torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
H.check(L.bofi_rb_stamps(buf))
names = ["entry", "self-attention + block", "W_o + residual", "y1, statistics, W_q', fold", "cross-attention", "W_o'", "stores"]
for w in (0, 7):
    t = [buf[w * 16 + i] for i in range(7)]
    print(f"wavefront {w}: " + "  ".join(f"{names[i]} +{t[i] - t[i - 1]}" for i in range(1, 7)) + f"   total {t[6] - t[0]} ticks")
