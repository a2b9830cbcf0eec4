"""Which PyTorch-ROCm ops (and how many device copies / fills) one eager training step of the benchmark workload enqueues beside
the library's own launches: the launch-diet worklist.  python tools/profile_aten_ops.py"""
import argparse, collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
args = argparse.Namespace(no_dead_knn=False, encode_ahead=False, force_collective=False)
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
wl = bench.Workload("clr", dev, 0, 1, args)
for i in range(3):
    wl.step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    wl.pre(3)
    wl.step(3)
    torch.cuda.synchronize()
ops = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.cpu_parent is None or (ev.name.startswith("aten::") and not str(getattr(ev.cpu_parent, "name", "")).startswith("aten::")):
        shapes = str(ev.input_shapes)[:60]
        stack = [s for s in (ev.stack or []) if "batch3dmot_amd" in s or "bench.py" in s]
        ops[(ev.name, shapes, stack[0][-70:] if stack else "")] += 1
for (name, shapes, where), n in sorted(ops.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} {name:28s} {shapes:60s} {where}")
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type is not None and "cuda" in str(ev.device_type).lower():
        kern[ev.name[:60]] += 1
print("--- device activities")
for k, n in kern.most_common(60):
    if "b3d" not in k: print(f"{n:3d} {k}")
