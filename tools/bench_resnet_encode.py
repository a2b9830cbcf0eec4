"""ResNetAE.encode on 2,000 crops (eval mode): python tools/bench_resnet_encode.py [1 = cudnn.benchmark].  Run under rocprofv3
--kernel-trace --stats to see the split (before the BatchNorm folding: 90 % MIOpenBatchNormFwdInferSpatialEst)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from batch3dmot_amd import encoders
dev = torch.device("cuda:0"); torch.manual_seed(0)
m = encoders.ResNetAE().to(dev).eval()
x = torch.rand(2000, 3, 32, 32, device=dev)
bench = len(sys.argv) > 1
torch.backends.cudnn.benchmark = bench
with torch.no_grad():
    for _ in range(5): m.encode(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.encode(x)
    torch.cuda.synchronize(); print("benchmark" if bench else "default", round(1e3 * (time.perf_counter() - t0) / 20, 3), "ms")
