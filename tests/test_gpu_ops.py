"""GPU parity of every HIP operator (through the C ABI) against plain fp32 PyTorch on the CPU
(the same ATen ops the oracle / the reference use).  Tolerance: fp32 reassociation noise,
|a-b| <= tol * (1 + |b| + max|b|) with tol = 2e-5 unless stated (north_star asks 1e-4)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


def close(a, b, tol=TOL, name=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = float(b.abs().max()) if b.numel() else 0.0
    err = (a - b).abs()
    ok = bool((err <= tol * (1.0 + b.abs() + scale)).all())
    assert ok, f"{name}: max err {float(err.max()):.3e} (scale {scale:.3e})"


@pytest.fixture(scope="module")
def ops():
    from stil_tta_amd import ops as o
    return o


def dev(t):
    return t.cuda().contiguous()


def nhwc(t):  # NCHW cpu -> NHWC cuda
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):  # NHWC cuda -> NCHW cpu
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (37, 7, 286), (300, 70, 48), (513, 257, 512), (1, 5, 20), (4100, 64, 64), (5000, 200, 32)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_nt_plain(ops, M, N, K, act):
    g = torch.Generator().manual_seed(M * 7 + N)
    A, W, b, R = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref_pre = 0.5 * (A @ W.t()) + b + R
    ref = ref_pre if act == 0 else (F.relu(ref_pre) if act == 1 else F.gelu(ref_pre))
    pre = torch.empty(M, N, device="cuda")
    out = ops.gemm_nt(dev(A), dev(W), M, N, K, bias=dev(b), resid=dev(R), pre=pre, act=act, alpha=0.5)
    close(out, ref, name="gemm")
    close(pre, ref_pre, name="pre")


@pytest.mark.parametrize("M,N,K,conv", [(1024, 256, 2048, False), (640, 128, 64, False), (4096, 2048, 512, False), (1024, 128, 9 * 128, True)])
def test_gemm_nt_split_precision_mode_is_as_close_to_float64_as_the_fp32_chain(ops, M, N, K, conv):
    """The OPT-IN split-precision mode (`tune` + 100000, STIL_PRECISION=bf16x3; csrc/gemm.hip B3): three bf16 terms per operand, six
    bf16 MFMAs per term pair, fp32 accumulation, two-level sums.  Against a float64 product its relative L2 distance must be within
    1.25x the fp32-exact kernel's on the same operands (measured: at or below it, profiles/r05_bf16x3_lab.txt), through the same
    epilogue (bias + residual + ReLU), for a plain product, a short one, a wide one under the column-panel tile order and an
    implicit-GEMM 3x3 convolution; and a launch that does not qualify (K % 32 != 0) silently runs the fp32-exact kernel."""
    g = torch.Generator().manual_seed(M + K)
    W, b = torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    if conv:
        C = K // 9
        A4 = torch.randn(M // 64, 8, 8, C, generator=g)
        A, geom = dev(A4), (8, 8, C, 8, 8, 3, 3, 1, 1, 0)
        cols = F.unfold(A4.permute(0, 3, 1, 2).double(), 3, padding=1)                     # [n, C*9, 64], K index (c, tap)
        A64 = cols.view(M // 64, C, 9, 64).permute(0, 3, 2, 1).reshape(M, K)               # rows (n, oy, ox), K index (tap, c)
    else:
        A, geom, A64 = dev(torch.randn(M, K, generator=g)), None, None
        A64 = A.cpu().double()
    ref = F.relu(A64 @ W.double().t() + b.double() + R.double())
    rel = lambda x: float((x.cpu().double() - ref).norm() / ref.norm())
    t0 = ops.TUNE["gemm"]
    try:
        ops.TUNE["gemm"] = t0 % ops.B3_FLAG
        e32 = rel(ops.gemm_nt(A, dev(W), M, N, K, geom=geom, bias=dev(b), resid=dev(R), act=1))
        ops.TUNE["gemm"] = t0 % ops.B3_FLAG + ops.B3_FLAG
        out3 = ops.gemm_nt(A, dev(W), M, N, K, geom=geom, bias=dev(b), resid=dev(R), act=1)
        e3 = rel(out3)
        again = ops.gemm_nt(A, dev(W), M, N, K, geom=geom, bias=dev(b), resid=dev(R), act=1)
        # a product the mode cannot take (K = 48: not a whole 32-deep k-tile) falls back to the fp32-exact kernel
        A2, W2 = dev(torch.randn(300, 48, generator=g)), dev(torch.randn(70, 48, generator=g))
        fb = ops.gemm_nt(A2, W2, 300, 70, 48)
        ops.TUNE["gemm"] = t0 % ops.B3_FLAG
        fb32 = ops.gemm_nt(A2, W2, 300, 70, 48)
    finally:
        ops.TUNE["gemm"] = t0
    assert torch.equal(out3, again), "the split-precision product is not bit-identical on repetition"
    assert torch.equal(fb, fb32), "a launch that does not qualify must run the fp32-exact kernel"
    assert e3 <= 1.25 * e32 + 1e-8 and e3 < 5e-7, f"split precision {e3:.2e} from float64, fp32-exact chain {e32:.2e}"


def test_gemm_nt_wide_epilogue_is_the_scalar_one_bit_for_bit(ops):
    """64x64 tiles stage their accumulators through LDS so that residual loads, `pre` stores and output stores are 16 bytes
    per lane (csrc/gemm.hip); `tune + 10000` keeps the one-dword-per-lane epilogue.  Same arithmetic per element: the two
    must agree exactly, with every epilogue feature on (alpha, eval-BN affine, bias, residual, pre, ReLU / GELU, conv gather,
    ragged M / N edges, per-tile statistics)."""
    g = torch.Generator().manual_seed(11)
    for M, N, K, act in [(4100, 64, 64, 1), (5000, 200, 96, 2), (130, 260, 512, 0), (64 * 37 + 5, 128, 1024, 1)]:
        A, W = dev(torch.randn(M, K, generator=g)), dev(torch.randn(N, K, generator=g))
        bias, sub, scale, shift = (dev(torch.randn(N, generator=g)) for _ in range(4))
        R = dev(torch.randn(M, N, generator=g))
        outs = []
        for tune in (0, 10000):
            ops.TUNE["gemm"] = tune
            pre = torch.empty(M, N, device="cuda")
            ts = torch.empty(2 * ((M + 63) // 64), N, device="cuda")
            o = ops.gemm_nt(A, W, M, N, K, bias=bias, sub=sub, scale=scale, shift=shift, resid=R, pre=pre, act=act, alpha=0.7)
            raw = ops.gemm_nt(A, W, M, N, K, colstats=ts)
            outs.append((o.clone(), pre.clone(), raw.clone(), ts.clone()))
        ops.TUNE["gemm"] = 0
        for u, v, n in zip(outs[0], outs[1], ("out", "pre", "raw", "tile statistics")):
            assert torch.equal(u, v), f"{n} differs between the 16-byte and the scalar epilogue at {(M, N, K)}"
    # 32-deep k-tiles (automatic for plain products with K >= 256) against 16-deep ones: same k order, same two-level sums
    for M, N, K in [(300, 128, 256), (4100, 64, 1024), (130, 260, 2048)]:
        A, W = dev(torch.randn(M, K, generator=g)), dev(torch.randn(N, K, generator=g))
        outs = []
        for tune in (100, 200):
            ops.TUNE["gemm"] = tune
            outs.append(ops.gemm_nt(A, W, M, N, K).clone())
        ops.TUNE["gemm"] = 0
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], ops.gemm_nt(A, W, M, N, K)), (M, N, K)
    # conv gather + strided output map (phase-decomposed dgrad)
    x = dev(torch.randn(3, 9, 9, 32, generator=g)); w = dev(torch.randn(64, 9 * 32, generator=g))
    outs = []
    for tune in (0, 10000):
        ops.TUNE["gemm"] = tune
        outs.append(ops.gemm_nt(x, w, 3 * 81, 64, 9 * 32, geom=(9, 9, 32, 9, 9, 3, 3, 1, 1, 0)).clone())
    ops.TUNE["gemm"] = 0
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K", [(1280, 256, 64), (512, 384, 1024), (2048, 2048, 512), (384, 128, 96)])
def test_gemm_nt_lean_128x128_kernel_equals_the_64x64_path(ops, M, N, K):
    """gemm_nt_big_kernel (`tune` 44: the 128x128 block tile of the plain products as a lean kernel of its own) against the 64x64
    path on the same operands: the PRODUCT bit for bit through every epilogue form (same k order, same two-level sums; alpha,
    eval-BatchNorm affine from the running variance, bias, residual, stored pre-activation, ReLU / GELU, ReLU mask; the
    operand-staging BatchNorm; the column-panel tile order of the wide case), the 64-row tile statistics (forward mean / M2, backward
    sums) to rounding -- a wave row of the block sums its 64 rows in registers where the 64x64 kernel combines two waves through
    LDS -- and against float64.  (384, 128, 96): fewer than 256 tiles of 64x64, so `tune` 44 falls back to the 64x64 kernel itself.)"""
    g = torch.Generator().manual_seed(M + N + K)
    A, W = dev(torch.randn(M, K, generator=g)), dev(torch.randn(N, K, generator=g) * 0.1)
    bias, sub, scale, shift = (dev(torch.randn(N, generator=g)) for _ in range(4))
    var = dev(torch.rand(N, generator=g) + 0.5)
    R, mask = dev(torch.randn(M, N, generator=g)), dev(torch.randn(M, N, generator=g))
    y = torch.randn(M, N, generator=g) * 2 + 0.3
    mean, rstd = y.mean(0), 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
    gamma, beta = 0.5 + torch.rand(N, generator=g), 0.2 * torch.randn(N, generator=g)
    stats = dev(torch.stack([mean, rstd, gamma * rstd, beta]).contiguous())
    yd = dev(y)
    nt = M // 64
    # operand-staging BatchNorm: A is the raw output of a producing layer with its statistics block [4][K]
    a_stats = dev(torch.stack([torch.randn(K, generator=g) * 0.1, torch.rand(K, generator=g) + 0.5, torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1]).contiguous())
    outs = {}
    t0 = ops.TUNE["gemm"]
    try:
        for tune in (11, 44):
            ops.TUNE["gemm"] = tune
            pre = torch.empty(M, N, device="cuda")
            o1 = ops.gemm_nt(A, W, M, N, K, bias=bias, sub=sub, scale=scale, scale_var=var, var_eps=1e-5, shift=shift, resid=R, pre=pre, act=2, alpha=0.7)
            o2 = ops.gemm_nt(A, W, M, N, K, resid=R, act=1)
            ts = torch.zeros(2 * nt, N, device="cuda")
            raw = ops.gemm_nt(A, W, M, N, K, colstats=ts)
            part = torch.zeros(2 * nt, N, device="cuda")
            gmask = ops.gemm_nt(A, W, M, N, K, resid=R, relu_mask=mask, bstats=(yd, stats, part, 0, 0))
            part2 = torch.zeros(2 * nt, N, device="cuda")
            g2 = ops.gemm_nt(A, W, M, N, K, bstats=(yd, stats, part2, 2, 0))
            ts_bn = torch.zeros(2 * nt, N, device="cuda")
            bna = ops.gemm_nt(A, W, M, N, K, a_bn=a_stats, colstats=ts_bn) if K % 32 == 0 and K % 16 == 0 else raw
            outs[tune] = [t.clone() for t in (o1, pre, o2, raw, gmask, g2, bna, ts, part, part2, ts_bn)]
    finally:
        ops.TUNE["gemm"] = t0
    names = ("affine+bias+resid+gelu", "pre", "resid+relu", "raw", "resid+mask", "plain gradient", "operand-staging BN")
    for i, nm in enumerate(names):
        assert torch.equal(outs[11][i], outs[44][i]), f"{nm}: the 128x128 kernel's product differs from the 64x64 path at {(M, N, K)}"
    for i, nm in ((7, "forward tile statistics"), (8, "backward sums, mask carried"), (9, "backward sums, mask recomputed"), (10, "forward tile statistics under operand-staging BN")):
        a, b = outs[11][i].double().cpu(), outs[44][i].double().cpu()
        sc = float(a.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * sc, f"{nm}: {float((a - b).abs().max()):.3e} of {sc:.3e} at {(M, N, K)}"
    # and the statistics against float64 on the 128x128 path's own product
    raw64 = outs[44][3].double().cpu()
    got = outs[44][7].double().cpu().view(nt, 2, N)
    for t in (0, nt // 2, nt - 1):
        rows = raw64[64 * t: 64 * t + 64]
        assert float((got[t, 0] - rows.mean(0)).abs().max()) <= 1e-5 * (1 + float(rows.abs().max()))
        assert float((got[t, 1] - ((rows - rows.mean(0)) ** 2).sum(0)).abs().max()) <= 1e-4 * (1 + float((rows ** 2).sum(0).max()))
    close(outs[44][2], F.relu(A.cpu() @ W.cpu().t() + R.cpu()), name="128x128 vs ATen")


CONVS = [  # Cin, Cout, k, stride, pad, H
    (64, 64, 1, 1, 0, 14), (64, 128, 1, 2, 0, 14), (32, 64, 3, 1, 1, 9), (64, 48, 3, 2, 1, 14), (16, 32, 3, 2, 1, 7),
    (128, 256, 3, 1, 1, 4),
]


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H", CONVS)
def test_conv_fwd_dgrad_wgrad(ops, Cin, Cout, k, stride, pad, H):
    from stil_tta_amd._lib import lib
    from stil_tta_amd.ops import _p, _stream
    g = torch.Generator().manual_seed(Cin + Cout + k)
    Nb, W_ = 3, H + 1
    x = torch.randn(Nb, Cin, H, W_, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g, requires_grad=True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    OH, OW = y.shape[2], y.shape[3]
    xd, wd_ = nhwc(x.detach()), dev(w.detach())
    wf = torch.empty(Cout, k * k * Cin, device="cuda")
    wdg = torch.empty(Cin, k * k * Cout, device="cuda")
    lib().conv_weight_layout(_p(wd_), _p(wf), _p(wdg), Cout, Cin, k, k, _stream())
    M = Nb * OH * OW
    geom = (H, W_, Cin, OH, OW, k, k, stride, pad, 0)
    yo = ops.gemm_nt(xd, wf, M, Cout, k * k * Cin, geom=geom)
    close(nchw(yo.view(Nb, OH, OW, Cout)), y, name="conv fwd")
    gyd = nhwc(gy).view(M, Cout)
    dx = ops.gemm_nt(gyd, wdg, Nb * H * W_, Cin, k * k * Cout, geom=(OH, OW, Cout, H, W_, k, k, stride, pad, 1))
    close(nchw(dx.view(Nb, H, W_, Cin)), x.grad, name="conv dgrad")
    dw = torch.full((Cout, Cin, k, k), 7.0, device="cuda")
    ops.wgrad_tn(gyd, xd, dw, M, Cout, k * k * Cin, geom=geom[:9], accumulate=0)
    close(dw, w.grad, name="conv wgrad")
    ops.wgrad_tn(gyd, xd, dw, M, Cout, k * k * Cin, geom=geom[:9], accumulate=1)
    close(dw, 2 * w.grad, name="conv wgrad accumulate")


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H,W", [(32, 64, 3, 2, 1, 14, 14), (64, 128, 1, 2, 0, 14, 10), (16, 32, 3, 2, 1, 7, 9), (32, 32, 3, 2, 1, 8, 6)])
def test_strided_dgrad_phase_decomposition(ops, Cin, Cout, k, stride, pad, H, W):
    g = torch.Generator().manual_seed(H * W)
    Nb = 3
    x = torch.randn(Nb, Cin, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    OH, OW = y.shape[2], y.shape[3]
    dx = ops.strided_dgrad(nhwc(gy).view(-1, Cout), dev(w), Nb, H, W, Cin, OH, OW, Cout, k, stride, pad)
    close(nchw(dx), x.grad, name="strided dgrad")


@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H", [(64, 64, 3, 1, 1, 8), (64, 128, 1, 2, 0, 8), (128, 256, 1, 1, 0, 6)])
def test_conv_bn_act_train_and_eval(ops, Cin, Cout, k, stride, pad, H, relu, res):
    g = torch.Generator().manual_seed(3)
    Nb = 4
    x = torch.randn(Nb, Cin, H, H, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.1).requires_grad_()
    gam = (0.5 + torch.rand(Cout, generator=g)).requires_grad_()
    bet = (0.1 * torch.randn(Cout, generator=g)).requires_grad_()
    rm, rv = 0.1 * torch.randn(Cout, generator=g), 0.5 + torch.rand(Cout, generator=g)
    OH = (H + 2 * pad - k) // stride + 1
    r = torch.randn(Nb, Cout, OH, OH, generator=g, requires_grad=True) if res else None
    rm_c, rv_c = rm.clone(), rv.clone()
    y = F.batch_norm(F.conv2d(x, w, stride=stride, padding=pad), rm_c, rv_c, gam, bet, training=True, momentum=0.1, eps=1e-5)
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = nhwc(x.detach()).requires_grad_()
    wd, gd, bd = dev(w.detach()).requires_grad_(), dev(gam.detach()).requires_grad_(), dev(bet.detach()).requires_grad_()
    rmd, rvd, nbt = dev(rm), dev(rv), torch.zeros((), dtype=torch.long, device="cuda")
    rd = nhwc(r.detach()).view(-1, Cout).requires_grad_() if res else None
    z = ops.ConvBnActFn.apply(xd, wd, gd, bd, rmd, rvd, nbt, rd, k, stride, pad, relu, None)
    close(nchw(z), y, name="z")
    close(rmd, rm_c, name="running_mean")
    close(rvd, rv_c, name="running_var")
    assert int(nbt) == 1
    z.backward(nhwc(gy))
    close(nchw(xd.grad), x.grad, tol=5e-5, name="dx")
    close(wd.grad, w.grad, tol=5e-5, name="dw")
    close(gd.grad, gam.grad, tol=5e-5, name="dgamma")
    close(bd.grad, bet.grad, tol=5e-5, name="dbeta")
    if res:
        close(nchw(rd.grad.view(Nb, OH, OH, Cout)), r.grad, name="dres")
    # eval mode (teacher): BN folded into the conv epilogue
    ye = F.batch_norm(F.conv2d(x.detach(), w.detach(), stride=stride, padding=pad), rm, rv, gam.detach(), bet.detach(), training=False, eps=1e-5)
    if res:
        ye = ye + r.detach()
    if relu:
        ye = F.relu(ye)
    ze = ops.conv_bn_eval(xd.detach(), wd.detach(), gd.detach(), bd.detach(), dev(rm), dev(rv), rd.detach() if res else None, k, stride, pad, relu)
    close(nchw(ze), ye, name="eval")


@pytest.mark.parametrize("C1,C2,k2,stride2,H,res2", [(64, 64, 3, 1, 9, False), (64, 128, 3, 2, 10, False), (128, 64, 1, 1, 6, True), (64, 64, 3, 1, 7, True),
                                                    (512, 128, 1, 1, 5, False)])
def test_deferred_batchnorm_is_the_materialised_path_bit_for_bit(ops, C1, C2, k2, stride2, H, res2):
    """models/resnets.py:112-132 inner layers: conv1+bn1+relu -> conv2+bn2(+residual)+relu.  With `defer` the first node
    returns its RAW conv output and its statistics; the second applies bn1 + relu while it stages its operand (forward GEMM
    a_bn, weight-gradient GEMM x_bn) and z1 never exists.  Same arithmetic on the same values: outputs, running buffers
    and every gradient must equal the materialised chain exactly; and the chain still matches ATen."""
    g = torch.Generator().manual_seed(C1 + k2)
    Nb, Cin = 3, 32
    x = torch.randn(Nb, Cin, H, H, generator=g)
    w1 = torch.randn(C1, Cin, 1, 1, generator=g) * 0.2
    w2 = torch.randn(C2, C1, k2, k2, generator=g) * 0.1
    g1, b1 = 0.5 + torch.rand(C1, generator=g), 0.3 * torch.randn(C1, generator=g)
    g2, b2 = 0.5 + torch.rand(C2, generator=g), 0.1 * torch.randn(C2, generator=g)
    pad2 = k2 // 2
    OH = (H + 2 * pad2 - k2) // stride2 + 1
    r = torch.randn(Nb, C2, OH, OH, generator=g) if res2 else None
    gy = torch.randn(Nb, C2, OH, OH, generator=g)

    def run(defer):
        leaves = [nhwc(x).requires_grad_()] + [dev(t).requires_grad_() for t in (w1, g1, b1, w2, g2, b2)]
        xd, w1d, g1d, b1d, w2d, g2d, b2d = leaves
        bufs = [torch.zeros(C1, device="cuda"), torch.ones(C1, device="cuda"), torch.zeros((), dtype=torch.long, device="cuda"),
                torch.zeros(C2, device="cuda"), torch.ones(C2, device="cuda"), torch.zeros((), dtype=torch.long, device="cuda")]
        rd = nhwc(r).view(-1, C2).requires_grad_() if res2 else None
        if defer:
            h, st = ops.ConvBnActFn.apply(xd, w1d, g1d, b1d, bufs[0], bufs[1], bufs[2], None, 1, 1, 0, True, None, False, True)
            assert st.shape == (4, C1) and not st.requires_grad
        else:
            h, st = ops.ConvBnActFn.apply(xd, w1d, g1d, b1d, bufs[0], bufs[1], bufs[2], None, 1, 1, 0, True, None), None
        z = ops.ConvBnActFn.apply(h, w2d, g2d, b2d, bufs[3], bufs[4], bufs[5], rd, k2, stride2, pad2, True, None, False, False, st)
        z.backward(nhwc(gy))
        torch.cuda.synchronize()
        return [z.detach()] + [t.grad for t in leaves] + ([rd.grad] if res2 else []) + bufs

    a, b = run(False), run(True)
    names = ["z", "dx", "dw1", "dgamma1", "dbeta1", "dw2", "dgamma2", "dbeta2"] + (["dres"] if res2 else []) + ["rm1", "rv1", "nbt1", "rm2", "rv2", "nbt2"]
    for n, u, v in zip(names, a, b):
        assert torch.equal(u, v), f"{n}: deferred BatchNorm differs from the materialised path (max |d| = {float((u - v).abs().max()):.3e})"
    # ... and the chain is the reference's arithmetic (ATen-CPU fp32)
    xr, w1r, g1r, b1r, w2r, g2r, b2r = [t.clone().requires_grad_() for t in (x, w1, g1, b1, w2, g2, b2)]
    hr = F.relu(F.batch_norm(F.conv2d(xr, w1r), torch.zeros(C1), torch.ones(C1), g1r, b1r, training=True, eps=1e-5))
    yr = F.batch_norm(F.conv2d(hr, w2r, stride=stride2, padding=pad2), torch.zeros(C2), torch.ones(C2), g2r, b2r, training=True, eps=1e-5)
    yr = F.relu(yr + r if res2 else yr)
    yr.backward(gy)
    close(nchw(b[0]), yr, name="z vs ATen")
    close(nchw(b[1]), xr.grad, tol=5e-5, name="dx vs ATen")
    close(b[2], w1r.grad, tol=5e-5, name="dw1 vs ATen")
    close(b[5], w2r.grad, tol=5e-5, name="dw2 vs ATen")
    close(b[3], g1r.grad, tol=5e-5, name="dgamma1 vs ATen")


@pytest.mark.parametrize("arch", ["resnet50", "resnet18"])
def test_premasked_residual_gradient_and_deferred_bn_equal_the_plain_blocks(ops, arch):
    """models/resnets.py:112-132 over a stack of residual blocks (layer1 + layer2 of the trunk: identity and downsample
    shortcuts, stride-2 stage boundary).  Production path: inner BatchNorms in the consumer's operand staging, and every block
    output's gradient leaving the next block's input-gradient GEMM already multiplied by the ReLU mask (stil_gemm_nt
    relu_mask).  Against the same blocks with both switched off: outputs, BN buffers and every parameter gradient bit for bit."""
    from stil_tta_amd.modules import ResNet
    res = []
    for fused, bstat in ((True, False), (False, False), (True, True)):
        ops._BN_DEFER, ops._PREMASK, ops._BN_BWD_EPILOGUE = fused, fused, bstat
        try:
            torch.manual_seed(5)
            net = ResNet(arch).cuda()
            blocks = list(net.layer1) + list(net.layer2)
            x = torch.randn(4, 12, 12, 64, generator=torch.Generator().manual_seed(1)).cuda().requires_grad_()
            h = x
            for blk in blocks:
                h = blk.run(h, True)
            gy = torch.randn(h.shape, generator=torch.Generator().manual_seed(2)).cuda()
            h.backward(gy)
            torch.cuda.synchronize()
            ps = [p for blk in blocks for p in blk.parameters()]
            bufs = [b for blk in blocks for b in blk.buffers()]
            res.append([h.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in ps] + [b.clone() for b in bufs])
        finally:
            ops._BN_DEFER, ops._PREMASK, ops._BN_BWD_EPILOGUE = True, True, True
    assert len(res[0]) == len(res[1]) == len(res[2]) > 20
    for i, (u, v) in enumerate(zip(res[0], res[1])):
        assert torch.equal(u, v), f"tensor {i}: fused blocks differ from the plain ones (max |d| = {float((u.float() - v.float()).abs().max()):.3e})"
    # ... and with the BatchNorm-backward sums left by the input-gradient GEMMs' epilogues (stil_gemm_nt bstats: per-tile fp32 sums
    # combined in double, instead of a reduction pass of its own): the forward and the buffers are untouched, the gradients agree to
    # rounding (the sums are added in another order -- and more accurately)
    nfw = 2 + len(ps)
    assert torch.equal(res[2][0], res[1][0]) and all(torch.equal(u, v) for u, v in zip(res[2][nfw:], res[1][nfw:]))
    worst = 0.0
    for i, (u, v) in enumerate(zip(res[2][1:nfw], res[1][1:nfw])):
        d = float((u - v).abs().max()) / (1e-30 + float(v.abs().max()))
        worst = max(worst, d)
        assert d <= 2e-5, f"gradient {i}: epilogue BatchNorm-backward sums differ from the pass by {d:.2e} of the tensor's scale"
    print(f"epilogue BN-backward sums vs the reduction pass: worst |d| / max|g| over {nfw - 1} gradients = {worst:.2e}")


@pytest.mark.parametrize("M,N,K,mode,resid", [(64 * 5 + 17, 64, 96, 2, False), (1000, 128, 64, 0, True), (130, 256, 512, 2, True), (64, 64, 32, 0, False),
                                              (64 * 257 + 3, 64, 64, 2, False)])
def test_gemm_epilogue_batchnorm_backward_sums(ops, M, N, K, mode, resid):
    """stil_gemm_nt `bstats`: the per-tile sums of g' and g' * xhat the epilogue leaves (g' = the stored gradient, masked by the
    recomputed ReLU sign for mode 2) against float64 sums over the same rows; stil_bn_train_bwd_tiles on them against
    stil_bn_train_bwd's own reduction pass (dx, dgamma, dbeta)."""
    from stil_tta_amd._lib import lib
    from stil_tta_amd.ops import _p, _stream
    g = torch.Generator().manual_seed(M + N)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
    y = torch.randn(M, N, generator=g) * 2 + 0.3
    R = torch.randn(M, N, generator=g) if resid else None
    zmask = torch.randn(M, N, generator=g) if mode == 0 else None
    mean, var = y.mean(0), y.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma, beta = 0.5 + torch.rand(N, generator=g), 0.2 * torch.randn(N, generator=g)
    stats = torch.stack([mean, rstd, gamma * rstd, beta]).contiguous()
    nt = (M + 63) // 64
    part = torch.full((2 * (nt + 3), N), float("nan"), device="cuda")
    yd, sd_ = dev(y), dev(stats)
    out = ops.gemm_nt(dev(A), dev(W), M, N, K, resid=None if R is None else dev(R), relu_mask=None if zmask is None else dev(zmask),
                      bstats=(yd, sd_, part, mode, 3))
    plain = ops.gemm_nt(dev(A), dev(W), M, N, K, resid=None if R is None else dev(R), relu_mask=None if zmask is None else dev(zmask))
    assert torch.equal(out, plain) and bool(torch.isnan(part[:6]).all())        # same stored gradient; tiles before tile0 untouched
    gd = out.cpu().double()
    if mode == 2:
        gd = gd * (((y - mean) * (gamma * rstd) + beta) > 0).double()
    xhat = ((y - mean) * rstd).double()
    got = part[6:].cpu().double().view(nt, 2, N)
    for t in range(nt):
        rows = slice(64 * t, min(M, 64 * t + 64))
        s1, s2 = gd[rows].sum(0), (gd[rows] * xhat[rows]).sum(0)
        sc = float((gd[rows].abs()).sum(0).max()) + 1.0
        assert float((got[t, 0] - s1).abs().max()) <= 1e-5 * sc and float((got[t, 1] - s2).abs().max()) <= 2e-5 * sc * (1 + float(xhat.abs().max())), t
    # the two BatchNorm backward entry points on the same gradient
    L = lib()
    gam = dev(gamma)
    outs = []
    for tiles in (False, True):
        dx, coef = torch.empty(M, N, device="cuda"), torch.empty(3, N, device="cuda")
        dga, dbe = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        if tiles:
            nb = L.bn_bwd_tiles_workspace_bytes(nt, N)
            ws = torch.empty(nb + 16, dtype=torch.uint8, device="cuda")
            L.bn_train_bwd_tiles(_p(out), None, _p(yd), _p(gam), _p(sd_), _p(part[6:]), nt, _p(dx), _p(dga), _p(dbe), _p(coef), M, N, mode, 0, _p(ws), nb, _stream())
        else:
            if N % 64:
                continue
            nb = L.bn_workspace_bytes(M, N)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            L.bn_train_bwd(_p(out), None, _p(yd), _p(gam), _p(sd_), _p(dx), None, _p(dga), _p(dbe), _p(coef), M, N, mode, 0, _p(ws), nb, _stream())
        outs.append((dx, dga, dbe))
    for u, v in zip(*outs):
        close(u, v, tol=1e-5, name="bn_train_bwd_tiles vs bn_train_bwd")
    close(outs[1][2], gd.sum(0), tol=1e-5, name="dbeta vs float64")
    close(outs[1][1], (gd * xhat).sum(0), tol=1e-5, name="dgamma vs float64")


def test_stem_and_maxpool(ops):
    g = torch.Generator().manual_seed(5)
    Nb, H = 3, 40
    x = torch.rand(Nb, 3, H, H, generator=g)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).requires_grad_()
    gam, bet = (0.5 + torch.rand(64, generator=g)).requires_grad_(), (0.1 * torch.randn(64, generator=g)).requires_grad_()
    y = F.relu(F.batch_norm(F.conv2d(x, w, stride=2, padding=3), None, None, gam, bet, training=True, eps=1e-5))
    yp = F.max_pool2d(y, 3, 2, 1)
    gy = torch.randn(yp.shape, generator=g)
    yp.backward(gy)
    col, meta = ops.im2col_stem(dev(x), 7, 2, 3)
    wd, gd, bd = dev(w.detach()).requires_grad_(), dev(gam.detach()).requires_grad_(), dev(bet.detach()).requires_grad_()
    wp = ops.pad_stem_weight(wd, meta[3])
    z = ops.ConvBnActFn.apply(col, wd, gd, bd, None, None, None, None, 7, 2, 3, True, (*meta, wp))
    close(nchw(z), y, name="stem")
    zp = ops.MaxPoolFn.apply(z)
    close(nchw(zp), yp, name="maxpool")
    zp.backward(nhwc(gy))
    close(wd.grad, w.grad, tol=5e-5, name="stem dw")
    close(gd.grad, gam.grad, tol=5e-5, name="stem dgamma")


def test_maxpool_ties_after_relu(ops):
    x = torch.zeros(1, 64, 6, 6)
    x[0, :, 2, 3] = 1.0
    x.requires_grad_()
    y = F.max_pool2d(x, 3, 2, 1)
    gy = torch.arange(y.numel(), dtype=torch.float32).reshape(y.shape)
    y.backward(gy)
    xd = nhwc(x.detach()).requires_grad_()
    z = ops.MaxPoolFn.apply(xd)
    z.backward(nhwc(gy))
    close(nchw(z), y)
    close(nchw(xd.grad), x.grad, name="tie routing")


@pytest.mark.parametrize("rows,D", [(33, 512), (7, 32), (130, 96)])
def test_layernorm(ops, rows, D):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g, requires_grad=True)
    w, b = (0.5 + torch.rand(D, generator=g)).requires_grad_(), torch.randn(D, generator=g).requires_grad_()
    y = F.layer_norm(x, (D,), w, b, eps=1e-5)
    gy = torch.randn(rows, D, generator=g)
    y.backward(gy)
    xd, wd, bd = dev(x.detach()).requires_grad_(), dev(w.detach()).requires_grad_(), dev(b.detach()).requires_grad_()
    z = ops.layernorm(xd, wd, bd)
    z.backward(dev(gy))
    close(z, y); close(xd.grad, x.grad, name="dx"); close(wd.grad, w.grad, name="dgamma"); close(bd.grad, b.grad, name="dbeta")


def _ref_attn(qkv, H, win, mask, p):
    B, T, _ = qkv.shape
    d = qkv.shape[-1] // (3 * H)
    t = qkv.reshape(B, T, 3, H, d).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    qo, Sq, ko, Sk = win
    a = ((q[:, :, qo:qo + Sq] @ k[:, :, ko:ko + Sk].transpose(-2, -1)) * d ** -0.5).softmax(-1)
    if mask is not None:
        a = a * mask.float() / (1 - p)
    return (a @ v[:, :, ko:ko + Sk]).transpose(1, 2).reshape(B, Sq, H * d)


@pytest.mark.parametrize("B,T,H,d,wins,use_mask", [
    (3, 18, 8, 64, [(0, 18, 0, 18)], False), (2, 65, 8, 64, [(0, 65, 0, 65)], False),
    (2, 12, 4, 128, [(1, 4, 1, 4), (5, 7, 5, 7), (0, 1, 0, 12)], True), (2, 114, 4, 128, [(1, 49, 1, 49), (50, 64, 50, 64), (0, 1, 0, 114)], True),
    (2, 9, 4, 16, [(0, 9, 0, 9)], False),
    (2, 10, 2, 12, [(0, 10, 0, 10)], True),          # head dim not a multiple of 16: the VALU kernels
    (2, 96, 2, 64, [(0, 96, 0, 96)], True),          # a 6 x 6 tile grid of the MFMA kernels
    # MFMA backward with head dims whose LDS regions are sized by the SECOND occupant (round-2 advisor finding: V with stride
    # d + 4 > padT(d) for d = 16, 80; Q with stride padT(d) > pad(d) for d = 32, 48, 96): SAINT's column attention is (65, 8, 16)
    (2, 65, 8, 16, [(0, 65, 0, 65)], False), (2, 65, 8, 16, [(0, 65, 0, 65)], True), (2, 70, 4, 32, [(0, 70, 0, 70)], True),
    (2, 80, 2, 48, [(0, 80, 0, 80)], False), (2, 50, 2, 80, [(0, 50, 0, 50)], True), (2, 64, 2, 96, [(0, 64, 0, 64)], True),
    (2, 114, 4, 32, [(1, 49, 1, 49), (50, 64, 50, 64), (0, 1, 0, 114)], True)])
def test_attention(ops, B, T, H, d, wins, use_mask):
    """softmax(q k^T) (dropout) v, forward and backward (models/Transformer.py:63-88, disentangle_transformer.py:49-94).
    Head dims that are multiples of 16 run on the matrix pipe (attn_*_mfma_kernel, v_mfma_f32_16x16x4_f32)."""
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(B, T, 3 * H * d, generator=g, requires_grad=True)
    masks = [(torch.rand(B, H, w[1], w[3], generator=g) >= 0.1) for w in wins] if use_mask else None
    out = torch.zeros(B, T, H * d)
    parts = []
    for i, w in enumerate(wins):
        parts.append((w, _ref_attn(qkv, H, w, masks[i] if masks else None, 0.1)))
    out = torch.cat([p for _, p in sorted(parts, key=lambda x: x[0][0])], dim=1) if sum(w[1] for w in wins) == T else None
    gy = torch.randn(B, T, H * d, generator=g)
    loss = sum((p * gy[:, w[0]:w[0] + w[1]]).sum() for w, p in parts)
    loss.backward()
    qd = dev(qkv.detach()).requires_grad_()
    md = [dev(m.to(torch.uint8)) for m in masks] if masks else None
    o = ops.attention(qd, H, wins, md, 0.1 if masks else 0.0)
    for w, p in parts:
        close(o[:, w[0]:w[0] + w[1]], p, name=f"attn out {w}")
    o.backward(dev(gy))
    close(qd.grad, qkv.grad, tol=5e-5, name="dqkv")


@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_fn(ops, act):
    g = torch.Generator().manual_seed(act)
    x = torch.randn(6, 11, 48, generator=g, requires_grad=True)
    w, b = torch.randn(70, 48, generator=g).requires_grad_(), torch.randn(70, generator=g).requires_grad_()
    y = F.linear(x, w, b)
    y = y if act == 0 else (F.relu(y) if act == 1 else F.gelu(y))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, wd, bd = dev(x.detach()).requires_grad_(), dev(w.detach()).requires_grad_(), dev(b.detach()).requires_grad_()
    z = ops.linear(xd, wd, bd, act)
    z.backward(dev(gy))
    close(z, y); close(xd.grad, x.grad, name="dx"); close(wd.grad, w.grad, name="dw"); close(bd.grad, b.grad, name="db")
    if act == 0:   # the residual sum of a transformer block in the GEMM epilogue: bit for bit the separate add, identity gradient
        r = torch.randn(6, 11, 70, generator=g)
        rd = dev(r).requires_grad_()
        xd2, wd2, bd2 = dev(x.detach()).requires_grad_(), dev(w.detach()).requires_grad_(), dev(b.detach()).requires_grad_()
        z2 = ops.linear(xd2, wd2, bd2, 0, resid=rd)
        z2.backward(dev(gy))
        assert torch.equal(z2.detach(), (ops.linear(xd, wd, bd, 0).detach() + rd.detach())) and torch.equal(rd.grad, dev(gy))
        assert torch.equal(xd2.grad, xd.grad) and torch.equal(wd2.grad, wd.grad) and torch.equal(bd2.grad, bd.grad)


@pytest.mark.parametrize("cat,ncon", [([3, 4, 5, 2, 6], 6), ([], 5), ([4, 4], 0)])
def test_tab_embed(ops, cat, ncon):
    from stil_tta_amd.modules import TabularTransformerEncoder
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(len(cat))
    B, D = 9, 64
    enc = TabularTransformerEncoder(SimpleNamespace(tabular_embedding_dim=D, tabular_transformer_num_layers=0), cat, [1] * ncon)
    cols = [torch.randint(0, c, (B, 1), generator=g).float() for c in cat] + [torch.randn(B, ncon, generator=g)]
    x = torch.cat(cols, dim=1)
    # CPU reference (models/Transformer.py:240-256)
    parts = []
    if cat:
        parts.append(F.embedding(x[:, :len(cat)].long() + enc.cat_offsets.long(), enc.cat_embedding.weight))
    if ncon:
        parts.append(F.linear(x[:, len(cat):].unsqueeze(-1), enc.con_proj.weight, enc.con_proj.bias))
    h = torch.cat([enc.cls_token.expand(B, -1, -1)] + parts, dim=1) + enc.column_embedding.weight.unsqueeze(0)
    gy = torch.randn(h.shape, generator=g)
    h.backward(gy)
    ref = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
    enc.zero_grad()
    enc.cuda()
    hd = ops.TabEmbedFn.apply(dev(x), enc.cat_embedding.weight if cat else None, enc.con_proj.weight if ncon else None,
                              enc.con_proj.bias if ncon else None, enc.cls_token, enc.column_embedding.weight, enc.cat_offsets,
                              enc.emb_rowcol, len(cat))
    close(hd, h, name="embed")
    hd.backward(dev(gy))
    for n, p in enc.named_parameters():
        if n in ref:
            close(p.grad, ref[n], name="grad " + n)


def test_small_losses(ops):
    g = torch.Generator().manual_seed(0)
    R, K, D = 13, 286, 128
    z = (3 * torch.randn(R, K, generator=g)).requires_grad_()
    y = torch.randint(0, K, (R,), generator=g)
    q = torch.softmax(torch.randn(R, K, generator=g), 1)
    w = (torch.rand(R, generator=g) > 0.4).float()
    l1 = F.cross_entropy(z, y)
    l2 = (F.cross_entropy(z, q, reduction="none") * w).mean()
    (2 * l1 + 3 * l2).backward()
    zd = dev(z.detach()).requires_grad_()
    a = ops.CEHardFn.apply(zd, dev(y))
    b = ops.CESoftFn.apply(zd, dev(q), dev(w))
    (2 * a + 3 * b).backward()
    close(a, l1, name="ce"); close(b, l2, name="soft ce"); close(zd.grad, z.grad, name="dlogits")
    # normalize + token mean
    x = torch.randn(R, 7, D, generator=g, requires_grad=True)
    o = F.normalize(x.mean(1))
    go = torch.randn(R, D, generator=g)
    o.backward(go)
    xd = dev(x.detach()).requires_grad_()
    od = ops.l2norm(ops.tokmean(xd))
    od.backward(dev(go))
    close(od, o); close(xd.grad, x.grad, name="d normalize/mean")


@pytest.mark.parametrize("B", [16, 100, 256])
def test_clip_and_club(ops, B):
    g = torch.Generator().manual_seed(B)
    D = 128
    f0, f1 = torch.randn(B, D, generator=g).requires_grad_(), torch.randn(B, D, generator=g).requires_grad_()
    n0, n1 = F.normalize(f0), F.normalize(f1)
    logits = n0 @ n1.t() / 0.1
    lab = torch.arange(B)
    ref = 0.3 * F.cross_entropy(logits, lab) + 0.7 * F.cross_entropy(logits.t(), lab)
    ref.backward()
    a, b = dev(f0.detach()).requires_grad_(), dev(f1.detach()).requires_grad_()
    l, Z = ops.clip_loss(a, b, 0.1, 0.3)
    l.backward()
    close(l, ref, name="clip"); close(Z, logits, name="logits"); close(a.grad, f0.grad, name="df0"); close(b.grad, f1.grad, name="df1")
    # CLUB: materialised [B,B,D] form of club.py:107-121 vs the closed form
    mu, y = torch.randn(B, 64, generator=g).requires_grad_(), (0.5 + torch.randn(B, 64, generator=g)).requires_grad_()
    pos = (-(mu - y) ** 2 / 2.0).sum(-1)
    neg = (-((y.unsqueeze(0) - mu.unsqueeze(1)) ** 2).mean(dim=1) / 2.0).sum(-1)
    club = (pos - neg).mean()
    est = ((mu - y) ** 2).sum(1).mean(0)
    (1.5 * club + 0.5 * est).backward()
    md, yd = dev(mu.detach()).requires_grad_(), dev(y.detach()).requires_grad_()
    c, e = ops.ClubFn.apply(md, yd)
    (1.5 * c + 0.5 * e).backward()
    close(c, club, name="club"); close(e, est, name="est"); close(md.grad, mu.grad, name="dmu"); close(yd.grad, y.grad, name="dy")


def test_cgpl_pgls_and_prototypes(ops):
    from stil_tta_amd._lib import lib
    from stil_tta_amd.ops import _p, _stream
    from oracle import stil_oracle as O
    g = torch.Generator().manual_seed(1)
    Bu, Bl, K, Dp = 40, 6, 11, 128
    zm = torch.randn(Bu, K, generator=g)
    zi = zm + 0.8 * torch.randn(Bu, K, generator=g)
    zt = zm + 0.8 * torch.randn(Bu, K, generator=g)
    feat = F.normalize(torch.randn(Bl + Bu, Dp, generator=g))
    protos = F.normalize(torch.randn(K, Dp, generator=g))
    mr = torch.rand(Bu, generator=g) >= 0.5
    r, T, th = 0.9, 0.1, 0.45
    a, b, d = zm.argmax(1), zi.argmax(1), zt.argmax(1)
    c1 = (a == b) & (a == d); c2i = (a == b) & (a != d); c2t = (a == d) & (a != b); c3 = ~(c1 | c2i | c2t)
    assert c1.any() and c2i.any() and c2t.any() and c3.any()
    q0 = (c1[:, None] * ((zm + zi + zt) / 3.0).softmax(1) + c2i[:, None] * ((zm + zi) / 2.0).softmax(1)
          + c2t[:, None] * ((zm + zt) / 2.0).softmax(1) + c3[:, None] * zm.softmax(1))
    tp = torch.softmax(feat[Bl:] @ protos.t() / T, 1)
    pl_ref = r * q0 + (1 - r) * tp
    pred_ref = r * zm.softmax(1) + (1 - r) * tp
    mask1 = pred_ref.max(1)[0].ge(th)
    assert mask1.any() and (~mask1).any()
    for use_pseudo in (True, False):
        pl, po, pred, flags, hard, w3 = ops.cgpl_pgls(dev(zm), dev(zi), dev(zt), dev(feat[Bl:]), dev(protos), dev(mr.to(torch.uint8)), r, T, th,
                                                      use_pseudo, want_orig=True)
        close(pl, pl_ref, name="pseudo_label"); close(po, q0, name="pseudo_label_orig")
        close(pred, pred_ref if use_pseudo else torch.zeros_like(pred_ref), name="prediction")
        cs = flags[:, 0].cpu()
        assert torch.equal(cs == 1, c1) and torch.equal(cs == 2, c2i) and torch.equal(cs == 3, c2t) and torch.equal(cs == 4, c3)
        assert torch.equal(flags[:, 1].cpu().bool(), mask1)
        close(w3[0], (mask1 & c1).float()); close(w3[1], (mask1 & (c1 | c2t | (c3 & mr))).float()); close(w3[2], (mask1 & (c1 | c2i | (c3 & ~mr))).float())
        # prototype loss + accumulation against the oracle
        y_l = torch.randint(0, K, (Bl,), generator=g)
        label_all = torch.cat((F.one_hot(y_l, K).float(), pred_ref if use_pseudo else torch.zeros_like(pred_ref)))
        fm = F.normalize(torch.randn(Bl + Bu, Dp, generator=g)).requires_grad_()
        ref = O.prototype_loss(label_all, protos, fm, T, th)
        ref.backward()
        hard_all = torch.cat((dev(y_l).to(torch.int32), hard))
        conf = torch.cat((torch.ones(Bl, dtype=torch.uint8, device="cuda"), flags[:, 2].contiguous()))
        fd = dev(fm.detach()).requires_grad_()
        l = ops.ProtoLossFn.apply(fd, dev(protos), hard_all, conf, T)
        l.backward()
        close(l, ref, name="proto loss"); close(fd.grad, fm.grad, name="proto dfeat")
        ls, lc = O.cal_prototypes(label_all[:Bl], feat[:Bl], th)
        us, uc = O.cal_prototypes(label_all[Bl:], feat[Bl:], th)
        out = torch.empty(K, Dp + 1, device="cuda")
        lib().proto_accum(_p(dev(feat)), _p(hard_all), _p(conf), _p(out), Bl + Bu, Bl, K, Dp, 2.0, _stream())
        close(out[:, :Dp], ls / 2.0 + us, name="class_sum"); close(out[:, Dp:], lc / 2.0 + uc, name="class_count")


def test_cgpl_top1_is_argmax_of_softmax_with_first_index_ties(ops):
    """STiLModel.py:262-263 takes torch.argmax(torch.softmax(logits)): two DISTINCT logits whose probabilities round to
    the same fp32 value tie and the first index wins, where argmax(logits) would pick the larger logit.  Crafted rows:
    the maximum sits at column 5 and column 2 is one ulp below it (exp(-7.5e-9) == 1.0f), for each head in turn."""
    import numpy as np
    K, Dp = 9, 16
    g = torch.Generator().manual_seed(3)
    hi = np.float32(0.1)
    lo = np.nextafter(hi, np.float32(0.0), dtype=np.float32)
    assert lo < hi

    def row(tie):
        z = -1.0 - torch.rand(K, generator=g)
        z[5] = float(hi)
        z[2] = float(lo) if tie else -2.0
        return z

    # row u: which heads carry the near-tie (m, i, t)
    pattern = [(0, 0, 0), (1, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1)]
    zm = torch.stack([row(p_[0]) for p_ in pattern]); zi = torch.stack([row(p_[1]) for p_ in pattern]); zt = torch.stack([row(p_[2]) for p_ in pattern])
    Bu = len(pattern)
    a, b, d = zm.softmax(1).argmax(1), zi.softmax(1).argmax(1), zt.softmax(1).argmax(1)   # the reference's expression, ATen CPU
    assert a.tolist() == [2 if p_[0] else 5 for p_ in pattern], "the crafted probabilities must tie on the CPU too"
    assert zm.argmax(1).tolist() == [5] * Bu                                               # ... where argmax(logits) sees no tie
    c1 = (a == b) & (a == d); c2i = (a == b) & (a != d); c2t = (a == d) & (a != b); c3 = ~(c1 | c2i | c2t)
    feat = F.normalize(torch.randn(Bu, Dp, generator=g)); protos = F.normalize(torch.randn(K, Dp, generator=g))
    mr = torch.zeros(Bu, dtype=torch.uint8)
    _, _, _, flags, _, _ = ops.cgpl_pgls(dev(zm), dev(zi), dev(zt), dev(feat), dev(protos), dev(mr), 0.9, 0.1, 0.5, True)
    cs = flags[:, 0].cpu()
    want = (1 * c1 + 2 * c2i + 3 * c2t + 4 * c3).to(torch.uint8)
    assert torch.equal(cs, want), (cs.tolist(), want.tolist())


def test_ema_and_adam_slabs():
    from stil_tta_amd._lib import lib
    from stil_tta_amd.ops import _p, _stream
    from oracle import stil_oracle as O
    g = torch.Generator().manual_seed(2)
    n = 4096 * 3
    e, v = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = e.clone().mul_(0.996).add_((1.0 - 0.996) * v)
    ed, vd = dev(e), dev(v)
    lib().ema_update(_p(ed), _p(vd), n, 0.996, _stream())
    assert torch.equal(ed.cpu(), ref), "EMA must be bit-exact"
    # Adam: 3 tensors of 1024-aligned slots, the middle one inactive (grad None in torch)
    sd = {"a": torch.randn(1000, generator=g), "b": torch.randn(700, generator=g), "c": torch.randn(2048, generator=g)}
    offs = {"a": 0, "b": 1024, "c": 2048}
    total = 4096
    P, G = torch.zeros(total), torch.zeros(total)
    for k, t in sd.items():
        P[offs[k]:offs[k] + t.numel()] = t
    c2t = torch.tensor([0, 1, 2, 2], dtype=torch.int32)
    Pd, Md, Vd = dev(P), torch.zeros(total, device="cuda"), torch.zeros(total, device="cuda")
    steps = torch.zeros(3, dtype=torch.int32, device="cuda")
    active = dev(torch.tensor([1, 0, 1], dtype=torch.uint8))
    opt = {}
    for step in (1, 2, 3):
        grads = {"a": torch.randn(1000, generator=g), "c": torch.randn(2048, generator=g) * 1e-3}
        G.zero_()
        for k, t in grads.items():
            G[offs[k]:offs[k] + t.numel()] = t
        O.adam_step(sd, grads, opt, step, 1e-3, 0.01)
        Gd, c2td = dev(G), dev(c2t)  # keep the device buffers alive until the launch is enqueued
        lib().adam_step(_p(Pd), _p(Gd), _p(Md), _p(Vd), _p(c2td), _p(steps), _p(active), 3, total, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0, _stream())
    for k, t in sd.items():
        close(Pd[offs[k]:offs[k] + t.numel()], t, tol=2e-6, name="adam " + k)
    assert steps.cpu().tolist() == [3, 0, 3]


def test_rng_mask_rate_and_determinism(ops):
    a = ops.rng_mask((1 << 20,), 0.1, 7, 0, "cuda")
    b = ops.rng_mask((1 << 20,), 0.1, 7, 0, "cuda")
    c = ops.rng_mask((1 << 20,), 0.1, 7, 1 << 20, "cuda")
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.float().mean()) - 0.9) < 2e-3
    from stil_tta_amd._lib import lib
    from stil_tta_amd.ops import _p, _stream
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    d0 = ops.rng_mask((1 << 16,), 0.1, 7, 0, "cuda", step)
    lib().counter_inc(_p(step), _stream())
    d1 = ops.rng_mask((1 << 16,), 0.1, 7, 0, "cuda", step)
    assert torch.equal(d0, a[: 1 << 16]) and not torch.equal(d0, d1) and int(step) == 1


def test_saint_pieces(ops):
    """SAINT embedding + per-column simple_MLP, GEGLU, row softmax, P@V against the oracle's restatement."""
    from oracle import stil_oracle as O
    from stil_tta_amd.saint import SaintBackbone
    g = torch.Generator().manual_seed(3)
    hp = O.default_hparams(img_size=64, num_classes=5, field_lengths=[3, 4, 1, 5, 1, 1, 2], model="resnet18", embedding_dim=512,
                           tabular_encoder="saint")
    B = 12
    sd = O.init_state(hp, 1)
    bb = SaintBackbone(hp, hp.field_lengths)
    bb.load_state_dict({k[6:]: v for k, v in sd.items() if k.startswith("model.")})
    x = O.synthetic_batch(hp, B)["u"][1][1]
    x = torch.cat([x, O.synthetic_batch(hp, B, seed=5)["u"][1][1]])[:B]
    masks = {"ff_col": torch.rand(B, 8, 128, generator=g) >= 0.8, "ff_row": torch.rand(1, B, 128 * 8, generator=g) >= 0.8}
    keys = [k for k in O.trainable_keys(sd) if k.startswith("model.encoder_tabular.") or k == "model.cls_token"]
    for k in keys:
        sd[k].requires_grad_(True)
    ref = O.saint_tabular_forward(sd, "model.", x, hp, masks)
    gy = torch.randn(ref.shape, generator=g)
    gr = torch.autograd.grad(ref, [sd[k] for k in keys], gy, allow_unused=True)
    bb.cuda()
    dm = {"ff_col": dev(masks["ff_col"].to(torch.uint8)), "ff_row": dev(masks["ff_row"].reshape(B, -1).to(torch.uint8))}
    out = bb.forward_tabular(dev(x), dm)
    close(out, ref, name="saint tokens")
    out.backward(dev(gy))
    params = dict(bb.named_parameters())
    n_checked = 0
    for k, gref in zip(keys, gr):
        p = params[k[len("model."):]]
        if gref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        close(p.grad, gref, tol=5e-5, name="grad " + k)
        n_checked += 1
    assert n_checked >= 25


def test_geglu_rowsoftmax_matmul_nn(ops):
    g = torch.Generator().manual_seed(4)
    h = torch.randn(7, 9, 64, generator=g, requires_grad=True)
    a, gt = h.chunk(2, -1)
    y = a * F.gelu(gt)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    hd = dev(h.detach()).requires_grad_()
    z = ops.GegluFn.apply(hd)
    z.backward(dev(gy))
    close(z, y); close(hd.grad, h.grad, name="geglu grad")
    s = (3 * torch.randn(37, 300, generator=g)).requires_grad_()
    v = torch.randn(300, 64, generator=g, requires_grad=True)
    o = s.softmax(-1) @ v
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    sd_, vd = dev(s.detach()).requires_grad_(), dev(v.detach()).requires_grad_()
    od = ops.MatmulNNFn.apply(ops.RowSoftmaxFn.apply(sd_), vd)
    od.backward(dev(go))
    close(od, o); close(sd_.grad, s.grad, name="dlogits"); close(vd.grad, v.grad, name="dv")


@pytest.mark.parametrize("M,C,tile_note", [(200, 64, "ragged last tile"), (64 * 37 + 5, 128, "many tiles"), (9000, 256, "one split, one launch"), (40, 64, "single partial tile"),
                                           (64 * 300 + 7, 64, "two splits: stage 1 + stage 2 launches")])
def test_bn_statistics_two_pass_and_tile_paths_agree_with_float64(ops, M, C, tile_note):
    """Training-mode BN statistics three ways: stil_bn_train_fwd (pilot-shifted two-pass), the per-tile Welford partials a
    GEMM epilogue writes (stil_gemm_nt colstats -> stil_bn_train_fwd_tiles), and float64 -- on data whose mean dwarfs its
    spread (|mean| / std = 50), where a naive E[x^2] - mean^2 in fp32 loses every digit."""
    from stil_tta_amd._lib import lib
    L = lib()
    g = torch.Generator().manual_seed(M)
    K = 32
    A = torch.randn(M, K, generator=g)
    W = torch.randn(C, K, generator=g) * 0.05
    A[:, 0] = 50.0 / 0.05  # a constant input column: every output channel gets mean ~ 50 * W[c, 0] / 0.05, spread ~ 0.28
    y64 = A.double() @ W.double().t()
    mean64, var64 = y64.mean(0), y64.var(0, unbiased=False)
    gam, bet = torch.ones(C), torch.zeros(C)
    Ad, Wd, gd, bd = dev(A), dev(W), dev(gam), dev(bet)   # keep the device tensors alive across the raw C-ABI calls
    outs = {}
    for mode in ("tiles", "two_pass"):
        rm, rv, nbt = dev(torch.zeros(C)), dev(torch.ones(C)), torch.zeros((), dtype=torch.long, device="cuda")
        stats = torch.empty(4, C, device="cuda"); z = torch.empty(M, C, device="cuda")
        if mode == "tiles":
            T = L.gemm_nt_tile_rows(M, C, 0)
            ts = torch.empty(2 * ((M + T - 1) // T), C, device="cuda")
            y = ops.gemm_nt(Ad, Wd, M, C, K, colstats=ts)
            nb = L.bn_tiles_workspace_bytes(M, C, T)
            ws = torch.empty(max(nb, 8), dtype=torch.uint8, device="cuda")
            L.bn_train_fwd_tiles(y.data_ptr(), ts.data_ptr(), T, gd.data_ptr(), bd.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(),
                                 None, None, z.data_ptr(), stats.data_ptr(), M, C, 0, 1e-5, 0.1, ws.data_ptr(), nb, None)
        else:
            y = ops.gemm_nt(Ad, Wd, M, C, K)
            nb = L.bn_workspace_bytes(M, C)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            L.bn_train_fwd(y.data_ptr(), gd.data_ptr(), bd.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), None, z.data_ptr(),
                           stats.data_ptr(), M, C, 0, 1e-5, 0.1, ws.data_ptr(), nb, None)
        torch.cuda.synchronize()
        outs[mode] = (stats.cpu(), z.cpu(), rm.cpu(), rv.cpu(), int(nbt))
    for mode, (stats, z, rm, rv, nbt) in outs.items():
        assert nbt == 1
        assert float((stats[0].double() - mean64).abs().max() / mean64.abs().max()) < 5e-7, (mode, "mean")
        rel = ((1.0 / stats[1].double() ** 2 - 1e-5) - var64).abs() / var64
        assert float(rel.max()) < 2e-3, (mode, "variance", float(rel.max()))  # fp32 rounding of y itself: ulp(50) / 0.28
        assert float((rm.double() - 0.1 * mean64).abs().max()) < 1e-4 * float(mean64.abs().max()), (mode, "running_mean")
        zref = ((y64 - mean64) / (var64 + 1e-5).sqrt()).float()
        assert float((z - zref).abs().max()) < 2e-2, (mode, "z")
    assert float((outs["tiles"][1] - outs["two_pass"][1]).abs().max()) < 2e-3


def test_onehot_argmax_first_maximum_and_threshold(ops):
    from stil_tta_amd._lib import lib
    p = torch.tensor([[0.2, 0.5, 0.5, 0.1], [0.9, 0.05, 0.03, 0.02], [0.25, 0.25, 0.25, 0.25]], device="cuda")
    oh = torch.empty_like(p); mask = torch.empty(3, device="cuda"); idx = torch.empty(3, dtype=torch.int32, device="cuda")
    lib().onehot_argmax(p.data_ptr(), 3, 4, 0.5, oh.data_ptr(), mask.data_ptr(), idx.data_ptr(), None)
    assert idx.tolist() == [1, 0, 0] and mask.tolist() == [1.0, 1.0, 0.0]
    assert torch.equal(oh.cpu(), torch.nn.functional.one_hot(torch.tensor([1, 0, 0]), 4).float())


def test_weight_layout_plan_equals_the_per_call_layouts_bit_for_bit(ops):
    """layouts.WeightLayouts (ONE launch for every conv / Linear weight of a model, csrc/layout.hip) against the per-call
    kernels it replaces on the step path (stil_conv_weight_layout / _phase / stil_transpose): identical bytes for every view,
    stale plans are never used, and a whole ResNet-18 STiL step runs bit-identically with and without the plan."""
    import torch.nn as nn
    from stil_tta_amd._lib import lib
    from stil_tta_amd.layouts import WeightLayouts, phase_specs
    from stil_tta_amd.ops import _p, _stream, cached_layout
    torch.manual_seed(3)
    net = nn.Sequential(nn.Conv2d(16, 32, 3, padding=1, bias=False), nn.Conv2d(32, 64, 3, stride=2, padding=1, bias=False), nn.Conv2d(64, 24, 1, bias=False),
                        nn.Conv2d(24, 40, 1, stride=2, bias=False), nn.Conv2d(3, 8, 7, stride=2, padding=3, bias=False), nn.Linear(70, 33), nn.Linear(64, 128, bias=False))
    total = sum((p.numel() + 1023) // 1024 * 1024 for p in net.parameters())
    slab = torch.zeros(total, device="cuda")
    o = 0
    for p in net.parameters():
        v = slab[o:o + p.numel()].view(p.shape)
        v.copy_(p.data)
        p.data = v
        o += (p.numel() + 1023) // 1024 * 1024
    plan = WeightLayouts(slab, [net], True)
    assert plan.n_jobs == 2 + (1 + 4) + 1 + 1 + 2 and not hasattr(net[4].weight, "_stil_wf")   # the 3-channel stem is not planned
    assert cached_layout(net[0].weight, "_stil_wf") is None                                      # not refreshed yet: stale
    plan.refresh()
    L = lib()
    for m in net:
        w = m.weight
        if isinstance(m, nn.Linear):
            assert torch.equal(cached_layout(w, "_stil_wd"), ops.transpose(w.detach()))
            continue
        Cout, Cin, k, _ = w.shape
        if Cin % 4:
            continue
        s_, pad = m.stride[0], m.padding[0]
        wf, wd = torch.empty(Cout, k * k * Cin, device="cuda"), torch.empty(Cin, k * k * Cout, device="cuda")
        L.conv_weight_layout(_p(w), _p(wf), _p(wd), Cout, Cin, k, k, _stream())
        if k > 1:
            assert torch.equal(cached_layout(w, "_stil_wf"), wf)
        if s_ == 1:
            assert torch.equal(cached_layout(w, "_stil_wd"), wd)
        else:
            specs = phase_specs(k, s_, pad)
            assert len(specs) == (4 if k == 3 else 1)
            for (py, px, ky0, kx0, KHs, KWs) in specs:
                ws = torch.empty(Cin, KHs * KWs * Cout, device="cuda")
                L.conv_weight_layout_phase(_p(w), _p(ws), Cout, Cin, k, k, s_, ky0, kx0, KHs, KWs, _stream())
                assert torch.equal(cached_layout(w, "_stil_wphase", (py, px)), ws), (k, py, px)
    plan.invalidate()
    assert cached_layout(net[0].weight, "_stil_wd") is None
    # freshness is tied to the DATA (round-4 advisor finding): a writer that goes around invalidate() -- a collective or a
    # checkpoint restore copying straight into the slab, an in-place update of one parameter -- makes the views stale too
    plan.refresh()
    assert cached_layout(net[0].weight, "_stil_wd") is not None
    slab[:16].add_(1.0)                                   # e.g. comm.broadcast_state -> dist.broadcast(flat.params)
    assert cached_layout(net[0].weight, "_stil_wd") is None and cached_layout(net[2].weight, "_stil_wd") is None
    plan.refresh()
    assert torch.equal(cached_layout(net[0].weight, "_stil_wf")[0, :16], net[0].weight.detach().permute(0, 2, 3, 1).reshape(32, -1)[0, :16])
    with torch.no_grad():
        net[2].weight.mul_(2.0)                           # one parameter written through PyTorch
    assert cached_layout(net[2].weight, "_stil_wd") is None and cached_layout(net[0].weight, "_stil_wd") is not None
    plan.refresh()
    assert torch.equal(cached_layout(net[2].weight, "_stil_wd"), ops.transpose(net[2].weight.detach().reshape(24, 64)))
    # a refresh recorded inside a hipGraph capture is not usable by eager callers afterwards (its event was never recorded eagerly)
    plan.captured = True
    assert cached_layout(net[0].weight, "_stil_wd") is None
    plan.captured = False
    assert cached_layout(net[0].weight, "_stil_wd") is not None
    plan.invalidate()
    # the whole step with and without the plan
    import os
    from stil_tta_amd import STiLModel
    from stil_tta_amd.driver import synthetic_batch, train_step
    from stil_tta_amd.flat import StilAdam
    fl = [3, 4] + [1] * 3
    outs = []
    for flag in ("1", "0"):
        os.environ["STIL_LAYOUT_PLAN"] = flag
        try:
            torch.manual_seed(0)
            m = STiLModel(dict(model="resnet18", embedding_dim=512, field_lengths=fl, num_classes=5, start_epoch=0, batch_size=16, th1=0.3, mi_dropout=False))
            m.setup_device("cuda"); m.train(); m.current_epoch = 1
            m.prototypes.copy_(F.normalize(torch.randn(5, 128, generator=torch.Generator().manual_seed(1))).cuda())
            opt = StilAdam(m.flat, lr=1e-3)
            b = synthetic_batch(fl, 5, 16, 64, seed=3, device="cuda")
            mr = (torch.arange(14) % 2 == 0).cuda()
            for _ in range(2):
                loss = train_step(m, opt, b, mask_random=mr)
            torch.cuda.synchronize()
            assert (m.flat._plans is not None) == (flag == "1")
            outs.append((float(loss), m.flat.params.clone(), m.flat.grads.clone(), m.flat.ema.clone()))
        finally:
            os.environ.pop("STIL_LAYOUT_PLAN")
    assert outs[0][0] == outs[1][0] and all(torch.equal(a, b_) for a, b_ in zip(outs[0][1:], outs[1][1:]))


@pytest.mark.parametrize("M,N,K,conv", [(256, 512, 2048, False), (100, 130, 1000, False), (1024, 256, 1024, False), (64, 64, 4608, False), (128, 256, 576, True),
                                        (256, 286, 1536, False)])
def test_gemm_split_k_is_the_unsplit_product_and_deterministic(ops, M, N, K, conv):
    """stil_gemm_nt split-K (grids below one 64x64 workgroup per CU): a tile's slices write their accumulators to slabs, the slice
    that draws the last arrival ticket adds them in slice order and runs the epilogue.  Against the unsplit launch (same epilogue:
    bias + residual + ReLU, per-tile statistics) to rounding, bit-identical on repetition with the SAME workspace (the tickets are
    left zero), and unchanged when two streams run split products at the same time on their own workspaces."""
    from stil_tta_amd._lib import lib
    assert lib().gemm_nt_split_workspace_bytes(M, N, K, 0) > 0, "the case is meant to be split"
    g = torch.Generator().manual_seed(M + K)
    W, b, R = dev(torch.randn(N, K, generator=g) * 0.05), dev(torch.randn(N, generator=g)), dev(torch.randn(M, N, generator=g))
    if conv:
        C = K // 9
        A, geom = dev(torch.randn(M // 64, 8, 8, C, generator=g)), (8, 8, C, 8, 8, 3, 3, 1, 1, 0)
    else:
        A, geom = dev(torch.randn(M, K, generator=g)), None
    nt = (M + 63) // 64

    default_split = ops._SPLITK

    def run(split):
        ops._SPLITK = split
        try:
            ts = torch.zeros(2 * nt, N, device="cuda")
            raw = ops.gemm_nt(A, W, M, N, K, geom=geom, colstats=ts if N % 4 == 0 else None)
            out = ops.gemm_nt(A, W, M, N, K, geom=geom, bias=b, resid=R, act=1)
            torch.cuda.synchronize()
            return raw, out, ts
        finally:
            ops._SPLITK = default_split
    ref = run(False)
    got = [run(True) for _ in range(3)]
    ops._SPLITK = True       # (restored at the end of the test)
    for u, v, nm in zip(got[0], ref, ("raw", "epilogue", "tile statistics")):
        close(u, v, tol=1e-5, name=f"split-K {nm} vs unsplit")
    for rep in got[1:]:
        assert all(torch.equal(u, v) for u, v in zip(rep, got[0])), "split-K is not bit-identical on repetition"
    if not conv:   # and against ATen
        close(got[0][1], F.relu(A.cpu() @ W.cpu().t() + b.cpu() + R.cpu()), name="split-K vs ATen")
    # the workspace is shared by products of other shapes (more tiles, fewer tiles) on the same stream: their slabs must never
    # land where this product keeps its arrival tickets
    for (m2, n2, k2) in ((832, 256, 512), (64, 128, 1024)):
        ops.gemm_nt(torch.randn(m2, k2, device="cuda"), torch.randn(n2, k2, device="cuda"), m2, n2, k2)
        again = ops.gemm_nt(A, W, M, N, K, geom=geom, bias=b, resid=R, act=1)
        assert torch.equal(again, got[0][1]), f"split-K result changed after a ({m2}, {n2}, {k2}) product used the same workspace"
    # two streams, each with its own workspace, at the same time
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    A2 = A * 0.5
    outs = {}
    torch.cuda.synchronize()
    for it in range(8):
        for st, a_, key in ((s1, A, "a"), (s2, A2, "b")):
            with torch.cuda.stream(st):
                outs[key] = ops.gemm_nt(a_, W, M, N, K, geom=geom, bias=b, resid=R, act=1)
    torch.cuda.synchronize()
    try:
        assert torch.equal(outs["a"], got[0][1]), "split-K result changed when another stream ran a split product beside it"
        close(outs["b"], ops.gemm_nt(A2, W, M, N, K, geom=geom, bias=b, resid=R, act=1), tol=0.0, name="second stream")
    finally:
        ops._SPLITK = default_split


def test_deferred_gradient_reductions_equal_the_immediate_ones_bit_for_bit(ops, monkeypatch):
    """ops._DeferredReduce (graph-replayed small batches): weight-gradient / bias-sum partials stay in the arena and ONE multi-job
    launch (stil_reduce_jobs) finishes them at join_side() -- the same bits as stil_wgrad_tn / stil_colsum, for convolution layouts
    (taps > 1), truncated rows (Kdst < K: the padded stem), unaligned shapes (scalar body), two contributions to one slot (flushed in
    order), more jobs than one launch carries (48) and an arena too small for the step (flush + regrow in the middle)."""
    monkeypatch.setattr(ops._defer, "mode", "auto")   # whatever STIL_REDUCE_DEFER says: deferred only inside `deferring()` here
    g = torch.Generator().manual_seed(5)
    cases = []    # (dY, X, M, N, K, kwargs, dW shape)
    for (Cin, Cout, k, H) in ((32, 64, 3, 9), (64, 48, 1, 14), (16, 32, 3, 7), (128, 256, 1, 4)):
        Nb = 3
        x = torch.randn(Nb, H, H, Cin, generator=g).cuda()
        M = Nb * H * H
        dy = torch.randn(M, Cout, generator=g).cuda()
        kw = dict(geom=(H, H, Cin, H, H, k, k, 1, k // 2)) if k > 1 else {}
        cases.append((dy, x.view(M, Cin) if k == 1 else x, M, Cout, k * k * Cin, kw, (Cout, Cin, k, k) if k > 1 else (Cout, Cin)))
    xs = torch.randn(700, 160, generator=g).cuda(); dys = torch.randn(700, 64, generator=g).cuda()
    cases.append((dys, xs, 700, 64, 160, dict(Kdst=147), (64, 147)))                       # padded stem: only 147 of 160 columns exist
    xu = torch.randn(300, 37, generator=g).cuda(); dyu = torch.randn(300, 10, generator=g).cuda()
    cases.append((dyu, xu, 300, 10, 37, {}, (10, 37)))                                     # K % 4 != 0: scalar body
    cases = cases * 9                                                                      # 54 weight jobs + 54 bias jobs > 2 x 48
    bias_in = [c[0] for c in cases]

    def run(deferred):
        outs = [torch.full(c[6], 0.5, device="cuda") for c in cases]
        bouts = [torch.full((c[3],), -0.25, device="cuda") for c in cases]
        twice = torch.zeros(cases[0][6], device="cuda")
        for c, o, b, bi in zip(cases, outs, bouts, bias_in):
            ops.wgrad_tn(c[0], c[1], o, c[2], c[3], c[4], accumulate=1, slot=True, **c[5])
            ops.colsum(bi, b, c[2], c[3], accumulate=1, scale=0.5, slot=True)
        c = cases[0]
        for _ in range(3):      # three contributions to ONE destination
            ops.wgrad_tn(c[0], c[1], twice, c[2], c[3], c[4], accumulate=1, slot=True, **c[5])
        if deferred:
            assert any(d["jobs"] for d in ops._defer.st.values()), "nothing was deferred"
        ops.join_side()
        torch.cuda.synchronize()
        return outs + bouts + [twice]

    ref = run(False)
    with ops.deferring():
        got = run(True)
        for d in ops._defer.st.values():    # an arena far too small: every allocation past it flushes the pending jobs first
            d["size"] = 1 << 16
        ops._defer.high.clear()
        ops._defer.ws = type(ops._defer.ws)()
        got_small = run(True)
    assert not any(d["jobs"] for d in ops._defer.st.values())
    for a, b, c in zip(ref, got, got_small):
        assert torch.equal(a, b) and torch.equal(a, c)
    w = torch.zeros(4, 4, device="cuda")
    ops.wgrad_tn(cases[0][0], cases[0][1], torch.zeros(cases[0][6], device="cuda"), *cases[0][2:5], accumulate=0, slot=True, **cases[0][5])
    assert not any(d["jobs"] for d in ops._defer.st.values()), "outside `deferring()` nothing is deferred"
