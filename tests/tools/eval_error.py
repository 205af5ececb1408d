"""Diagnostic (GPU box): (1) error of one long-K convolution against float64, HIP vs ATen-CPU fp32; (2) per-stage error of
the EVAL-mode (teacher) ResNet-50 trunk at the baseline shape, HIP vs the CPU fp32 oracle, both against float64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import stil_oracle as O
from oracle.make_golden import randomize_state
from stil_tta_amd.modules import ResNet, _conv_bn
from stil_tta_amd import ops

rel = lambda a, b: float((a.double() - b).norm() / b.norm())
torch.manual_seed(0)
for (cin, cout, k, hw, pos) in ((512, 512, 3, 14, True), (512, 512, 3, 14, False), (2048, 512, 1, 7, True), (64, 64, 3, 56, True)):
    x = torch.rand(8, cin, hw, hw) if pos else torch.randn(8, cin, hw, hw)
    w = torch.randn(cout, cin, k, k) * (2.0 / (cout * k * k)) ** 0.5
    y64 = F.conv2d(x.double(), w.double(), padding=k // 2)
    y32 = F.conv2d(x, w, padding=k // 2)
    xd = x.cuda().permute(0, 2, 3, 1).contiguous()
    wd = w.cuda()
    if k == 1:
        wf = wd.reshape(cout, cin)
    else:
        wf = torch.empty((cout, k * k * cin), device="cuda")
        from stil_tta_amd._lib import lib
        lib().conv_weight_layout(ops._p(wd), ops._p(wf), None, cout, cin, k, k, ops._stream())
    M = 8 * hw * hw
    yg = ops.gemm_nt(xd, wf, M, cout, k * k * cin, geom=(hw, hw, cin, hw, hw, k, k, 1, k // 2, 0))
    yg = yg.view(8, hw, hw, cout).permute(0, 3, 1, 2).cpu()
    print(f"conv Cin={cin} k={k} K={cin*k*k} positive_input={pos}: relL2 vs fp64  cpu32 {rel(y32, y64):.2e}  hip {rel(yg, y64):.2e}")

hp = O.default_hparams(img_size=224)
sd = randomize_state(O.init_state(hp, seed=21), seed=22)
x = torch.rand(32, 3, 224, 224)
p = "model.encoder_imaging."


def cpu_stages(dtype):
    s = {k_[len(p):]: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k_, v in sd.items() if k_.startswith(p)}
    bn = lambda h, q: F.batch_norm(h, s[q + ".running_mean"], s[q + ".running_var"], s[q + ".weight"], s[q + ".bias"], training=False, eps=1e-5)
    outs = []
    h = F.relu(bn(F.conv2d(x.to(dtype), s["conv1.weight"], stride=2, padding=3), "bn1"))
    outs.append(("bn1relu", h))
    h = F.max_pool2d(h, 3, 2, 1)
    for li, nblk in enumerate([3, 4, 6, 3], start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 1) else 1
            q = f"layer{li}.{bi}."
            idn = h
            o = F.relu(bn(F.conv2d(h, s[q + "conv1.weight"]), q + "bn1"))
            o = F.relu(bn(F.conv2d(o, s[q + "conv2.weight"], stride=stride, padding=1), q + "bn2"))
            o = bn(F.conv2d(o, s[q + "conv3.weight"]), q + "bn3")
            if q + "downsample.0.weight" in s:
                idn = bn(F.conv2d(h, s[q + "downsample.0.weight"], stride=stride), q + "downsample.1")
            h = F.relu(o + idn)
            outs.append((f"layer{li}.{bi}", h))
    return outs


c64, c32 = cpu_stages(torch.float64), cpu_stages(torch.float32)
net = ResNet("resnet50")
net.load_state_dict({k_[len(p):]: v for k_, v in sd.items() if k_.startswith(p)})
net.cuda().eval()
with torch.no_grad():
    col, meta = ops.im2col_stem(x.cuda(), 7, 2, 3)
    wpad = ops.pad_stem_weight(net.conv1.weight, meta[3])
    h = _conv_bn(col, net.conv1, net.bn1, True, False, stem=(*meta, wpad))
    g = [("bn1relu", h)]
    h = ops.MaxPoolFn.apply(h)
    for li, layer in enumerate((net.layer1, net.layer2, net.layer3, net.layer4), start=1):
        for bi, blk in enumerate(layer):
            h = blk.run(h, False)
            g.append((f"layer{li}.{bi}", h))
for (n, a64), (_, a32), (_, ag) in zip(c64, c32, g):
    ag = ag.cpu().permute(0, 3, 1, 2)
    print(f"{n:10s} absmax {float(a64.abs().max()):9.3e}  relL2 vs fp64:  cpu32 {rel(a32, a64):.2e}   hip {rel(ag, a64):.2e}")
