import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from batch3dmot_amd import synth
from oracle import ref_torch
th = int(sys.argv[1]); torch.set_num_threads(th)
b = synth.make_batch(2, 1500, 15000)
torch.manual_seed(5621)
m = ref_torch.PoseGNN(run_dead_knn=True)
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
for i in range(5):
    t0 = time.perf_counter()
    ref_torch.train_step(m, b, opt, 2, "cb", True)
    print(th, "threads step", i, f"{time.perf_counter()-t0:.3f}s", flush=True)
