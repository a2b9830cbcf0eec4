"""Debug aid: split-graph data-parallel step under torch.distributed (gloo) with every rank on GPU 0."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from batch3dmot_amd import synth
from batch3dmot_amd.dist import FlatGradSync
from batch3dmot_amd.pose_gnn import PoseGNN
from batch3dmot_amd.train_step import forward_backward, make_optimizer, train_step
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
backend = os.environ.get("B3D_DEBUG_BACKEND", "gloo")
if world > 1 or backend == "nccl":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
def say(*a):
    torch.cuda.synchronize(); print(f"[rank {rank}]", *a, flush=True)
torch.manual_seed(5621)
model = PoseGNN().to(dev).train()
opt = make_optimizer(model, capturable=True)
sync = FlatGradSync(model.parameters(), flat=opt) if (world > 1 and "nosyncobj" not in mode) else None
pool = [synth.make_batch(2, 1500, 15000, first_graph_idx=rank * 1000 + 2 * i).to(dev) for i in range(2)]
def fresh(b):
    if hasattr(b, "_b3d_graph"): del b._b3d_graph
for i in range(3):
    b = pool[i % 2]; fresh(b)
    train_step(model, b, opt, batch_size=2, loss_kind="cb", logits=True, grad_sync=sync if "nowarmsync" not in mode else None)
say("warm-up done")
if world > 1 and "nobarrier" not in mode:
    dist.barrier()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
graphs = []
for b in pool:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        fresh(b); forward_backward(model, b, opt, batch_size=2, loss_kind="cb", logits=True)
    graphs.append(g)
g_opt = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_opt, stream=s, capture_error_mode="thread_local"):
    opt.step()
torch.cuda.current_stream().wait_stream(s)
say("captured")
for k in range(4):
    graphs[k % 2].replay(); say("replayed fwd/bwd", k)
    if sync is not None and "noreplaysync" not in mode:
        sync.sync(force=True); say("synced", k)
    elif backend == "nccl" and world == 1:
        dist.all_reduce(opt.flat_grad); say("nccl all_reduce (1 rank)", k)     # an eager RCCL call between the replays
    g_opt.replay(); say("replayed adam", k)
say("OK")
