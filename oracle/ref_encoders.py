"""CPU restatement of the reference's frozen sensor encoders -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this file.  It imports nothing from
batch3dmot_amd: until round 4 the oracle's camera+LiDAR+radar model (oracle/ref_torch.GNN) was handed the PRODUCT's encoder
classes with their HIP paths switched off, i.e. one file was both the thing tested and the thing it was tested against.

Plain PyTorch, operation for operation in the reference's order (no BatchNorm folding, no fused kernels):

  * ResNetAE.encode             /root/reference/batch_3dmot/models/resnet_fully_conv.py:42-82 (ResidualBlock, downsample),
                                :84-140 (layers), :155-161 (encode)
  * PointNetClassifier.forward_feat   .../models/pointnet.py:9-57 (STN3d), :111-157 (PointNetFeat, global feature, no
                                feature transform), :168-179, :188-192 (fc1-bn1-relu, fc2-dropout-bn2-relu)
  * RadarNetClassifier.forward_feat   .../models/radarnet.py:9-36 (RadarNetFeat), :40-50, :60-64

Parameter and buffer names are the reference's (a reference state_dict loads with strict=True -- checked against the reference
modules themselves by oracle/make_golden.py, whose g2 / g9 fixtures pin the numbers), including the layers the GNN path never
evaluates (ResNetAE.bn, fc_encoder, fc_decoder, conv_decoder; the classifiers' fc3).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _relu_bn(bn, y):
    return F.relu(bn(y))


def _global_max(x):
    """[B, 1024, P] -> [B, 1024]: maximum over the points (pointnet.py:44-45, :155-156; radarnet.py:31-33)."""
    return x.max(dim=2, keepdim=True)[0].view(-1, 1024)


class _PointStack(nn.Module):
    """conv1-bn1, conv2-bn2, conv3-bn3 of kernel size 1 over [B, C, P] (the 'shared MLP' 64-128-1024)."""

    def __init__(self, cin):
        super().__init__()
        for i, (a, b) in enumerate(((cin, 64), (64, 128), (128, 1024)), start=1):
            setattr(self, f"conv{i}", nn.Conv1d(a, b, 1))
        for i, c in enumerate((64, 128, 1024), start=1):
            setattr(self, f"bn{i}", nn.BatchNorm1d(c))

    def stack(self, x, relu_last):
        x = _relu_bn(self.bn1, self.conv1(x))
        x = _relu_bn(self.bn2, self.conv2(x))
        x = self.bn3(self.conv3(x))
        return _global_max(F.relu(x) if relu_last else x)


class STN3d(_PointStack):
    """Input transform: pointnet.py:9-57.  The ReLU sits on all three stack layers here (:41-43)."""

    def __init__(self):
        super().__init__(3)
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, 9)
        self.bn4, self.bn5 = nn.BatchNorm1d(512), nn.BatchNorm1d(256)

    def forward(self, x):
        g = self.stack(x, relu_last=True)
        g = _relu_bn(self.bn4, self.fc1(g))                    # :47
        g = _relu_bn(self.bn5, self.fc2(g))                    # :48
        g = self.fc3(g)                                        # :49
        eye = torch.eye(3, dtype=g.dtype, device=g.device).reshape(1, 9)
        return (g + eye).view(-1, 3, 3)                        # :52-56


class PointNetFeat(_PointStack):
    """pointnet.py:111-157 with global_feat=True, feature_transform=False: the last stack layer has no ReLU (:154)."""

    def __init__(self):
        super().__init__(3)
        self.stn = STN3d()

    def forward(self, x):
        trans = self.stn(x)                                                # :131
        x = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)           # :132-136
        return self.stack(x, relu_last=False)


class RadarNetFeat(_PointStack):
    """radarnet.py:9-36: four input channels, no input transform."""

    def __init__(self):
        super().__init__(4)

    def forward(self, x):
        return self.stack(x, relu_last=False)


class _Classifier(nn.Module):
    """fc1-bn1-relu, fc2-dropout-bn2-relu on the 1024-wide global feature (pointnet.py:188-192, radarnet.py:60-64); fc3 belongs
    to forward(), which the GNN never calls."""

    def __init__(self, feat, k):
        super().__init__()
        self.feat = feat
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, k)
        self.dropout = nn.Dropout(p=0.3)
        self.bn1, self.bn2 = nn.BatchNorm1d(512), nn.BatchNorm1d(256)

    def forward_feat(self, x):
        g = self.feat(x)
        g = _relu_bn(self.bn1, self.fc1(g))
        return _relu_bn(self.bn2, self.dropout(self.fc2(g)))


class PointNetClassifier(_Classifier):
    def __init__(self, k=7, feature_transform=False):
        if feature_transform:
            raise NotImplementedError("feature_transform=True is not on the GNN path (clr_att_gnn.py builds it with the default)")
        super().__init__(PointNetFeat(), k)


class RadarNetClassifier(_Classifier):
    def __init__(self, k=2, feature_transform=False):
        super().__init__(RadarNetFeat(), k)


class ResidualBlock(nn.Module):
    """resnet_fully_conv.py:42-73.  BOTH convolutions take `stride` and padding 1 (:53-55)."""

    def __init__(self, cin, cout, kernel, stride, down):
        super().__init__()
        self.downsample = down
        self.conv1 = nn.Conv2d(cin, cout, kernel, stride, padding=1)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, kernel, stride, padding=1)
        self.bn2 = nn.BatchNorm2d(cout)

    def forward(self, x):
        skip = self.downsample(x) if self.downsample is not None else x       # :62-64
        y = F.relu(self.bn1(self.conv1(x)))                                   # :66-67
        y = self.bn2(self.conv2(y))                                           # :68
        return F.relu(y + skip)                                               # :70-71


def _downsample(cin, cout, kernel, stride):
    """resnet_fully_conv.py:76-81: an unpadded convolution and its BatchNorm."""
    return nn.Sequential(nn.Conv2d(cin, cout, kernel, stride), nn.BatchNorm2d(cout))


def _fc_pairs(widths):
    mods = []
    for a, b in zip(widths[:-1], widths[1:]):
        mods += [nn.Linear(a, b), nn.BatchNorm1d(b, momentum=0.01), nn.ReLU()]
    return nn.Sequential(*mods)


class ResNetAE(nn.Module):
    """resnet_fully_conv.py:84-161.  encode() is the GNN's image embedding: [N, 3, 32, 32] -> [N, 96]."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 12, kernel_size=4, stride=2, padding=1)      # :90
        self.bn = nn.BatchNorm2d(12)                                          # :91 (declared, never applied by encode)
        self.res_block1 = ResidualBlock(12, 24, 4, 2, _downsample(12, 24, 5, 3))      # :102
        self.res_block2 = ResidualBlock(24, 48, 3, 1, _downsample(24, 48, 1, 1))      # :103
        self.res_block3 = ResidualBlock(48, 96, 3, 2, _downsample(48, 96, 3, 2))      # :104
        # the autoencoder's remaining halves (:108-140): state_dict entries only
        self.fc_encoder = _fc_pairs([192, 128, 64])
        self.fc_decoder = _fc_pairs([64, 128, 192])
        chans = [96, 72, 48, 24, 12, 3]
        dec = []
        for i, (a, b) in enumerate(zip(chans[:-1], chans[1:])):
            dec += [nn.ConvTranspose2d(a, b, 4, stride=2, padding=1), nn.Sigmoid() if i == len(chans) - 2 else nn.ReLU()]
        self.conv_decoder = nn.Sequential(*dec)

    def encode(self, x):
        y = self.res_block3(self.res_block2(self.res_block1(self.conv(x))))   # :156-159
        return y.view(y.size(0), -1)                                          # :160
