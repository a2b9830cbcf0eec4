"""PointNet / RadarNet forward_feat in eval mode: the HIP point-feature kernel against the PyTorch-ROCm path.
python tools/bench_encoders.py [clouds]     (3,000 detections x 70 % with LiDAR = 2,100 clouds by default)"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from batch3dmot_amd import encoders

b = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
dev = torch.device("cuda:0")
torch.manual_seed(5621)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
for name, mod, c, p, nb in (("pointnet", encoders.PointNetClassifier(k=7), 3, 128, b), ("radarnet", encoders.RadarNetClassifier(k=7), 4, 64, max(1, b * 25 // 70))):
    mod = mod.to(dev).eval()
    x = torch.randn(nb, c, p, device=dev)
    stacks = 2 if name == "pointnet" else 1                       # STN + trunk
    mac = nb * p * (c * 64 + 64 * 128 + 128 * 1024) * stacks
    with torch.no_grad():
        for flag in (True, False):
            for m in mod.modules():
                m.use_hip = flag
            ms = timed(lambda: mod.forward_feat(x))
            res[f"{name}_{'hip' if flag else 'torch'}"] = {"clouds": nb, "ms": round(ms, 3), "conv_stack_tflops_if_all_time": round(2 * mac / ms / 1e9, 1)}
print(json.dumps(res))
