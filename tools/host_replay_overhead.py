"""Where the host time of a REPLAYED training step goes (bench.py's timed region): the eager prologue (modality masks + row
compaction, whose counts are read back) and the hipGraph launch, separately and together.  No device sync inside the loops."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
args = argparse.Namespace(no_dead_knn=False, encode_ahead=False, force_collective=False)
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
wl = bench.Workload("clr", dev, 0, 1, args)
for i in range(5):
    wl.step(i)
torch.cuda.synchronize()
graphs, _ = bench.capture(wl, False)
n = len(wl.pool)


def timed(name, fn):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        fn(i)
    enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"{name:60s} host enqueue {1e3 * enq / a.steps:7.3f} ms/step   wall {1e3 * tot / a.steps:7.3f} ms/step")


nodes = None
try:
    nodes = "n/a"
except Exception:
    pass
timed("graph replay only (rows of the captured batch reused)", lambda i: graphs[i % n].replay())
timed("eager prologue only (masks + compaction + count read-back)", lambda i: wl.pre(i))
timed("prologue + replay (the bench's timed step)", lambda i: (wl.pre(i), graphs[i % n].replay()))
ms = wl.model.mask_stream
wl.model.mask_stream = None
timed("prologue on the launch stream + replay", lambda i: (wl.pre(i), graphs[i % n].replay()))
wl.model.mask_stream = ms
