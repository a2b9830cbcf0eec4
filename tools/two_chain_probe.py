"""Probe (round 6): does the GPU run TWO half-size message-passing chains side by side faster than one full-size chain?
The forward / backward of a step is a strict chain edge(l) -> node(l) -> edge(l + 1) ... whose node kernels are 188 workgroups on
256 CUs; the two graphs of a batch are independent, so graph A's node phase could run under graph B's edge phase.  This tool times
  (a) the shipped step on a 2-graph batch (encoder outputs precomputed), replayed from its hipGraph, against
  (b) two model replicas stepping ONE graph each from two hipGraphs on two streams (optionally offset),
same total edges per step.  Timing only (the replicas do not share weights).  python tools/two_chain_probe.py [--steps 40]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--no-dead-knn", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
args = argparse.Namespace(no_dead_knn=a.no_dead_knn, encode_ahead=False, force_collective=False)


def timed(fn, steps):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


full = bench.Workload("clr", dev, 0, 1, args, encoders="precomputed", graphs=2)
for i in range(3):
    full.step(i)
torch.cuda.synchronize()
gf, _ = bench.capture(full, False)
t_full = timed(lambda i: gf[i % 4].replay(), a.steps)
e_full = sum(full.edges) / len(full.edges)
print(f"one chain, 2 graphs / step: {t_full:.3f} ms / step ({e_full:.0f} edges)", flush=True)

halves = [bench.Workload("clr", dev, r, 1, args, encoders="precomputed", graphs=1) for r in (0, 1)]
caps = []
for h in halves:
    for i in range(3):
        h.step(i)
    torch.cuda.synchronize()
    caps.append(bench.capture(h, False)[0])
e_half = sum(sum(h.edges) / len(h.edges) for h in halves)
t_one = timed(lambda i: caps[0][i % 4].replay(), a.steps)
print(f"one half-size chain alone: {t_one:.3f} ms / step", flush=True)
s = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
for offset_us in (0, 40, 100):
    def both(i):
        for r in (0, 1):
            with torch.cuda.stream(s[r]):
                if i == 0 and r == 1 and offset_us:
                    torch.cuda._sleep(int(offset_us * 2100))
                caps[r][i % 4].replay()
    for st in s:
        st.wait_stream(torch.cuda.current_stream())
    t_two = timed(both, a.steps)
    print(f"two half-size chains on two streams, offset {offset_us:3d} us: {t_two:.3f} ms per pair of steps ({e_half:.0f} edges)  "
          f"-> {t_full / t_two:.3f} x the one-chain rate at equal edges", flush=True)
