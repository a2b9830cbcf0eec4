import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from batch3dmot_amd import synth
from batch3dmot_amd.pose_gnn import PoseGNN
dev = torch.device("cuda:0"); torch.manual_seed(5621)
m = PoseGNN().to(dev); m.run_dead_knn = False; m.eval()
big = synth.make_batch(2, 1500, 15000).to(dev)
with torch.no_grad():
    for it in range(20):
        out, _ = m(big)
torch.cuda.synchronize()
