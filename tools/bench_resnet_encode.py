"""ResNetAE.encode on N crops: the HIP phase kernels against the PyTorch / MIOpen modules, train (frozen) and eval mode.
usage: python tools/bench_resnet_encode.py [N]"""
import copy
import sys
import time

import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from batch3dmot_amd import encoders  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = encoders.ResNetAE().to(dev)
    for p in m.parameters():
        p.requires_grad = False
    ref = copy.deepcopy(m)
    for mod in ref.modules():
        mod.use_hip = False
    x = torch.rand(n, 3, 32, 32, device=dev)
    for mode in ("train", "eval"):
        m.train(mode == "train")
        ref.train(mode == "train")
        with torch.no_grad():
            a, b = m.encode(x), ref.encode(x)
            err = float((a - b).abs().max() / b.abs().max())
            t_hip = timed(lambda: m.encode(x))
            t_ref = timed(lambda: ref.encode(x))
        print(f"{mode}: N={n} hip {t_hip:.3f} ms, pytorch {t_ref:.3f} ms, rel err {err:.2e}")


if __name__ == "__main__":
    main()
