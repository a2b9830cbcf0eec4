"""Device time of one average-precision call at the training-batch size (b3d_average_precision: direct form)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from batch3dmot_amd import metrics, synth
dev = torch.device("cuda:0")
b = synth.make_batch(2, 1500, 15000, first_graph_idx=0, modalities=False).to(dev)
e = b.edge_index.size(1)
s = torch.rand(e, device=dev)
print("edges", e, "positives", int(b.y.sum()))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        metrics._run(s, b.y, b.edge_classes, 7)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print(f"average precision (overall + 7 classes): {1e6 * dt:.1f} us per call (host-enqueue bound if > device time)")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    metrics._run(s, b.y, b.edge_classes, 7)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        for _ in range(20):
            metrics._run(s, b.y, b.edge_classes, 7)
    torch.cuda.synchronize()
    ev0.record(); g.replay(); ev1.record(); torch.cuda.synchronize()
    print(f"replayed x20: {ev0.elapsed_time(ev1) * 1e3 / 20:.1f} us per call (device)")
