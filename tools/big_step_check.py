import sys, torch
sys.path.insert(0, '/root/repo')
from batch3dmot_amd import encoders, synth
from batch3dmot_amd.clr_att_gnn import GNN
from batch3dmot_amd.train_step import make_optimizer, train_step
dev = torch.device("cuda:0")
torch.manual_seed(0)
for graphs, n, e in ((8, 1500, 15550), (1, 37, None), (3, 5000, 52000)):
    b = synth.make_batch(graphs, n, e, first_graph_idx=4000, modalities=True).to(dev) if e else synth.make_graph(n, None, k=4, graph_idx=4001, modalities=True).to(dev)
    losses = []
    for rep in range(2):
        torch.manual_seed(1)
        m = GNN(encoders.ResNetAE(), encoders.PointNetClassifier(k=7), encoders.RadarNetClassifier(k=7)).to(dev)
        from oracle.seeded import seeded_fill_
        seeded_fill_(m, 77)
        m.train()
        opt = make_optimizer(m, lr=1e-3)
        torch.manual_seed(2)
        out = []
        for it in range(2):
            loss, _, _ = train_step(m, b, opt, batch_size=graphs, loss_kind="cb", logits=False)
            out.append(float(loss))
        torch.cuda.synchronize()
        losses.append(out)
    print(graphs, b.pose_feats.size(0), b.edge_index.size(1), losses, "reproducible" if losses[0] == losses[1] else "DIFFERENT", flush=True)
