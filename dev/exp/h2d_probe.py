"""Host -> device copy rate from pinned memory (what a pipelined eval loop can feed): python dev/exp/h2d_probe.py"""
import time, torch
dev = torch.device("cuda:0")
for mb in (9.4, 47, 188):
    n = int(mb * 1e6) // 2
    h = torch.empty(n, dtype=torch.bfloat16).pin_memory()
    d = torch.empty(n, dtype=torch.bfloat16, device=dev)
    for nstreams in (1, 2, 4):
        streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
        hs = [torch.empty(n, dtype=torch.bfloat16).pin_memory() for _ in range(nstreams)]
        ds = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(nstreams)]
        for _ in range(2):
            for s, a, b in zip(streams, hs, ds):
                with torch.cuda.stream(s):
                    b.copy_(a, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            for s, a, b in zip(streams, hs, ds):
                with torch.cuda.stream(s):
                    b.copy_(a, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{mb:6.1f} MB x {nstreams} stream(s): {reps * nstreams * n * 2 / dt / 1e9:6.1f} GB/s")
# device -> host of a launch's small outputs
o = torch.empty(320 * 20, dtype=torch.int64, device=dev); oh = torch.empty(320 * 20, dtype=torch.int64).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    oh.copy_(o, non_blocking=True)
torch.cuda.synchronize(); print(f"D2H of 51 KB: {(time.perf_counter() - t0) * 10:.3f} ms each")
# pageable -> pinned staging on the host (one core)
import numpy as np
a = np.random.rand(64 * 36 * 2048).astype(np.float32); p = torch.empty(a.size, dtype=torch.float32).pin_memory()
t0 = time.perf_counter()
for _ in range(5):
    p.numpy()[:] = a
print(f"host memcpy pageable float32 -> pinned: {5 * a.nbytes / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
t0 = time.perf_counter()
for _ in range(3):
    b = torch.from_numpy(a).to(torch.bfloat16)
print(f"host float32 -> bf16 conversion (torch, {torch.get_num_threads()} threads): {3 * a.nbytes / (time.perf_counter() - t0) / 1e9:.1f} GB/s of float32 input")
