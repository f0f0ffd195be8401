"""Does a HIP stream created with a CU mask (hipExtStreamCreateWithCUMask) confine kernels -- launched directly and replayed from a
captured graph -- to its CUs?  Times the persistent GEMM (one workgroup per CU, 240 workgroups) on an unmasked stream and on streams
masked to 128 CUs (bits 0..127 / even bits / the first 4 of every 8), eager and as a graph replay."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from boficap_amd import hip as H
lib = H.lib()
hiprt = ctypes.CDLL("libamdhip64.so")
hiprt.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hiprt.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(((1 << b) if (w * 32 + b) in bits else 0) for b in range(32)) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hiprt.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


M, N, K = 11520, 2048, 512
x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
masks = {"none": None, "bits 0..127": set(range(128)), "even bits": set(range(0, 256, 2)), "first 4 of every 8": {i for i in range(256) if i % 8 < 4},
         "first 16 of every 32": {i for i in range(256) if i % 32 < 16}}
for name, bits in masks.items():
    st = torch.cuda.Stream() if bits is None else masked_stream(bits)
    with torch.cuda.stream(st):
        def run():
            H.check(lib.bofi_linear_fused(H.ptr(x), K, H.ptr(w), H.ptr(bias), None, N, H.ptr(y), H.DT_BF16, N, None, N, None, None, 0, None, M, N, K, 0, H.stream_ptr()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            run()
        e1.record(st); torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) * 1e3 / 20
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                run()
        g.replay(); torch.cuda.synchronize()
        e0.record(st); g.replay(); e1.record(st); torch.cuda.synchronize()
        graph = e0.elapsed_time(e1) * 1e3 / 20
    print(f"mask {name:22s}: eager {eager:6.1f} us per GEMM, graph replay {graph:6.1f} us", flush=True)
