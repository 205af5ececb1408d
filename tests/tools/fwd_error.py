"""Diagnostic (GPU box): per-stage forward error of the HIP ResNet vs float64, next to the CPU fp32 oracle's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import stil_oracle as O
from oracle.make_golden import randomize_state
from stil_tta_amd.modules import ResNet, _conv_bn
from stil_tta_amd import ops

torch.manual_seed(0)
hp = O.default_hparams()
sd = randomize_state(O.init_state(hp, seed=3), seed=4)
x = torch.rand(16, 3, 128, 128)
p = "model.encoder_imaging."

def cpu_stages(dtype):
    s = {k[len(p):]: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items() if k.startswith(p)}
    outs = []
    h = F.conv2d(x.to(dtype), s["conv1.weight"], stride=2, padding=3)
    outs.append(("conv1", h))
    h = F.relu(F.batch_norm(h, None, None, s["bn1.weight"], s["bn1.bias"], training=True, eps=1e-5))
    outs.append(("bn1relu", h))
    h = F.max_pool2d(h, 3, 2, 1)
    for li, nblk in enumerate([3, 4, 6, 3], start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 1) else 1
            q = f"layer{li}.{bi}."
            idn = h
            o = F.relu(F.batch_norm(F.conv2d(h, s[q + "conv1.weight"]), None, None, s[q + "bn1.weight"], s[q + "bn1.bias"], training=True, eps=1e-5))
            o = F.relu(F.batch_norm(F.conv2d(o, s[q + "conv2.weight"], stride=stride, padding=1), None, None, s[q + "bn2.weight"], s[q + "bn2.bias"], training=True, eps=1e-5))
            o = F.batch_norm(F.conv2d(o, s[q + "conv3.weight"]), None, None, s[q + "bn3.weight"], s[q + "bn3.bias"], training=True, eps=1e-5)
            if q + "downsample.0.weight" in s:
                idn = F.batch_norm(F.conv2d(h, s[q + "downsample.0.weight"], stride=stride), None, None, s[q + "downsample.1.weight"], s[q + "downsample.1.bias"], training=True, eps=1e-5)
            h = F.relu(o + idn)
        outs.append((f"layer{li}", h))
    return outs

c64, c32 = cpu_stages(torch.float64), cpu_stages(torch.float32)
net = ResNet("resnet50")
net.load_state_dict({k[len(p):]: v for k, v in sd.items() if k.startswith(p)})
net.cuda().train()
with torch.no_grad():
    col, meta = ops.im2col_stem(x.cuda(), 7, 2, 3)
    wpad = ops.pad_stem_weight(net.conv1.weight, meta[3])
    y0 = ops.gemm_nt(col, wpad, col.shape[0], 64, meta[3]).view(16, 64, 64, 64)
    g = [("conv1", y0)]
    h = _conv_bn(col, net.conv1, net.bn1, True, True, stem=(*meta, wpad))
    g.append(("bn1relu", h))
    h = ops.MaxPoolFn.apply(h)
    for li, layer in enumerate((net.layer1, net.layer2, net.layer3, net.layer4), start=1):
        for blk in layer:
            h = blk.run(h, True)
        g.append((f"layer{li}", h))
rel = lambda a, b: float((a.double() - b).norm() / b.norm())
for (n, a64), (_, a32), (_, ag) in zip(c64, c32, g):
    ag = ag.cpu().permute(0, 3, 1, 2)
    print(f"{n:8s} relL2 vs fp64:  cpu32 {rel(a32, a64):.2e}   hip {rel(ag, a64):.2e}")
