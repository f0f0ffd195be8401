import sys, os
sys.path.insert(0, ".")
import torch
n = int(os.environ.get("DUMMY_STREAMS", "0"))
keep = [torch.cuda.Stream() for _ in range(n)]
for s in keep:
    with torch.cuda.stream(s):
        torch.zeros(1, device="cuda")
import bench
sys.argv = ["bench.py", "--mode", "rl", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"]
bench.main()
