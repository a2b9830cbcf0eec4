"""Experiment: the frozen encoders as a FREE-RUNNING second pipeline instead of a branch of the step's graph.

Shipped form (bench.py, train_step.EncodeAhead inside the captured step): graph k = step(batch k) with the encoders of batch k + 1
as a parallel branch -- forked and joined inside the graph, so no encoder work crosses a step boundary.
Here: two graphs per pool batch, `enc[k]` (the three encoders of batch k into static buffers) and `step[k]` (the training step on
those buffers), replayed on two streams that only meet in events: step k waits for enc k; enc k (next time round the pool) waits for
the step that last read its buffers.  The encoder passes run in the same order, so the state is the same.

    python tools/free_pipeline.py [--steps 40] [--lead 1]
"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--lead", type=int, default=1, help="how many batches the encoder pipeline runs ahead of the step pipeline")
ap.add_argument("--pre", action="store_true", help="with the eager prologue (masks + compaction + count read-back) in both forms")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
args = argparse.Namespace(no_dead_knn=False, encode_ahead=True, force_collective=False)
wl = bench.Workload("clr", dev, 0, 1, args)
n = len(wl.pool)
for i in range(4):
    wl.step(i)
torch.cuda.synchronize()


def timed(name, fn, steps):
    for i in range(2 * n):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(2 * n + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:60s} {1e3 * dt:7.4f} ms/step", flush=True)
    return dt


# ---- shipped form ----
graphs, _ = bench.capture(wl, False)


def shipped(i):
    if a.pre:
        wl.pre(i)
    graphs[i % n].replay()


# ---- free-running form: separate graphs ----
A = torch.cuda.current_stream(dev)
B = torch.cuda.Stream(dev)
cap = torch.cuda.Stream(dev)
enc_g, step_g = [], []
torch.cuda.synchronize()
cap.wait_stream(A)
for k in range(n):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        wl.ahead.launch(wl.pool[k], rows=wl.rows_static[k], static=wl.enc_static[k])
        wl.ahead.take(wl.pool[k])
    enc_g.append(g)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        wl.cap_ret[k] = wl._run(k, {"encoded": wl.enc_static[k]})
    step_g.append(g)
A.wait_stream(cap)
torch.cuda.synchronize()
enc_done = [torch.cuda.Event() for _ in range(n)]
step_done = [torch.cuda.Event() for _ in range(n)]
state = {"enc_next": 0}


def enqueue_enc(j):
    """Encoder pass number j (batch j % n) on stream B."""
    k = j % n
    with torch.cuda.stream(B):
        B.wait_event(step_done[k])                 # the step that last read enc_static[k] (recorded or never: a no-op)
        if a.pre:
            li, ri = wl.model.modality_rows(wl.pool[k])        # masks + compaction of this batch, on the mask stream, joined into B
            sl, sr = wl.rows_static[k]
            sl.copy_(li); sr.copy_(ri)
        enc_g[k].replay()
        enc_done[k].record(B)


def free(i):
    # keep the encoder pipeline `lead` batches ahead of the step about to be launched
    while state["enc_next"] <= i + a.lead:
        enqueue_enc(state["enc_next"])
        state["enc_next"] += 1
    k = i % n
    A.wait_event(enc_done[k])
    step_g[k].replay()
    step_done[k].record(A)


for r in range(2):
    timed("shipped: encoders as a branch of the step's graph", shipped, a.steps)
    state["enc_next"] = 0
    # restart the free pipeline at a pool boundary: timed() calls fn with i = 0 .. first
    timed(f"free-running encoder pipeline, lead {a.lead}", free, a.steps)
    state["enc_next"] = 0
