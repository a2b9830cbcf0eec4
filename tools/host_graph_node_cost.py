"""What one hipGraph node costs the HOST on this image, apart from queue back-pressure: graphs of N tiny kernels (one stream, and
three branches forked and joined every 12 nodes like the step's side streams) replayed without device syncs, and the training
step's own graph replayed for 40 and 240 steps (a host that is only waiting for queue space leads the GPU by a fixed number of
packets, whatever the loop length; a host that is the slower side never leads)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=180)
ap.add_argument("--no-step", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())


def timed(name, fn, steps):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    each = []
    t0 = time.perf_counter()
    for i in range(steps):
        t1 = time.perf_counter()
        fn(i)
        each.append(time.perf_counter() - t1)
    enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    first, last = sorted(each[:8])[4], sorted(each[-8:])[4]
    print(f"{name:58s} steps {steps:4d}  host {1e3 * enq / steps:7.3f} ms/replay (median of the first 8: {1e3 * first:6.3f}, of the last 8: "
          f"{1e3 * last:6.3f})  wall {1e3 * tot / steps:7.3f} ms/replay  host lead at the end {1e3 * (tot - enq):7.2f} ms", flush=True)


def tiny_graph(n, branches):
    x = [torch.zeros(256, device=dev) for _ in range(max(1, branches))]
    side = [torch.cuda.Stream() for _ in range(max(0, branches - 1))]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        cur = torch.cuda.current_stream()
        k = 0
        while k < n:
            if branches <= 1:
                x[0].add_(1.0); k += 1
                continue
            for s in side:
                s.wait_stream(cur)
            for j in range(4):
                x[0].add_(1.0); k += 1
            for b, s in enumerate(side):
                with torch.cuda.stream(s):
                    for j in range(4):
                        x[b + 1].add_(1.0); k += 1
            for s in side:
                cur.wait_stream(s)
    return g, x


for br in (1, 3):
    g, keep = tiny_graph(a.nodes, br)
    for steps in (40, 400):
        timed(f"{a.nodes} tiny kernels, {br} branch(es)", lambda i: g.replay(), steps)
    del g

if not a.no_step:
    import bench
    for ahead in (False, True):
        args = argparse.Namespace(no_dead_knn=False, encode_ahead=ahead, force_collective=False)
        wl = bench.Workload("clr", dev, 0, 1, args)
        for i in range(5):
            wl.step(i)
        torch.cuda.synchronize()
        graphs, _ = bench.capture(wl, False)
        n = len(wl.pool)
        for steps in (40, 240):
            timed(f"training step graph, encode_ahead={ahead}", lambda i: graphs[i % n].replay(), steps)
        del graphs, wl
