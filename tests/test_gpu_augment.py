"""Device input pipeline (stil_tta_amd/augment.py, csrc/augment.hip).
Tabular corruption is pinned to golden vectors recorded from the REFERENCE's own `corrupt`
(tests/golden/tab_corrupt.npz, oracle/make_golden_data.py): bit-exact with the reference's draws injected.
The image transforms are checked against PyTorch-CPU restatements of torchvision's tensor formulas (torchvision and
albumentations are absent offline: unpinned)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "tab_corrupt.npz")


def test_tab_corrupt_matches_reference_golden_bit_for_bit():
    from stil_tta_amd.augment import TabularCorruptor
    fx = np.load(GOLD)
    table = torch.tensor(fx["table"], dtype=torch.float32)
    for tag, c in (("c030", 0.3), ("c000", 0.0), ("c100", 1.0), ("c005", 0.05)):
        cor = TabularCorruptor(table, c, "cuda")
        assert cor.k == fx[tag + "_idx"].shape[1]
        clean = table[torch.arange(12) % len(table)].cuda()
        got = cor(clean, draws=(torch.tensor(fx[tag + "_idx"]), torch.tensor(fx[tag + "_pos"])))
        assert torch.equal(got.cpu(), torch.tensor(fx[tag + "_out"], dtype=torch.float32)), tag   # bit-exact (values are copied)
        assert torch.equal(clean.cpu(), table[torch.arange(12) % len(table)])                       # the clean view is untouched


def test_tab_corrupt_device_draws_are_valid_and_uniform():
    from stil_tta_amd.augment import TabularCorruptor
    g = torch.Generator().manual_seed(0)
    table = torch.randn(500, 64, generator=g)
    cor = TabularCorruptor(table, 0.3, "cuda", seed=7)
    B = 4096
    idx, pos = cor.draw(B)
    idx, pos = idx.cpu().long(), pos.cpu().long()
    assert idx.shape == (B, 19) and int(idx.min()) >= 0 and int(idx.max()) < 64 and int(pos.min()) >= 0 and int(pos.max()) < 500
    assert all(len(set(r.tolist())) == 19 for r in idx[:512]), "columns of one row must be distinct (random.sample)"
    col_freq = torch.bincount(idx.flatten(), minlength=64).float() / (B * 19)
    assert float((col_freq - 1 / 64).abs().max()) < 0.15 / 64 * 4          # every column is picked about equally often
    row_freq = torch.bincount(pos.flatten(), minlength=500).float() / (B * 19)
    assert float((row_freq - 1 / 500).abs().max()) < 1 / 500
    idx2, _ = cor.draw(B)
    assert not torch.equal(idx2.cpu().long(), idx), "successive batches must draw fresh values"
    # end to end: exactly k entries of every row come from the marginal of their own column, the rest are untouched
    clean = table[:B % 500 + 300].cuda()
    out = cor(clean).cpu()
    changed = (out != clean.cpu())
    assert int(changed.sum(1).max()) <= 19
    cols = changed.nonzero()
    for r, c in cols[:200].tolist():
        assert bool((table[:, c] == out[r, c]).any())


def _ref_resize(src_f, box, P, flip):
    """crop -> F.interpolate(bilinear, align_corners=False) -> flip: torchvision's tensor resized_crop + hflip."""
    t, l, h, w = [int(v) for v in box]
    crop = src_f[:, t:t + h, l:l + w].unsqueeze(0)
    out = F.interpolate(crop, size=(P, P), mode="bilinear", align_corners=False)[0]
    return out.flip(-1) if flip else out


def test_resize_crop_flip_matches_interpolate():
    from stil_tta_amd.augment import resize_crop, rrc_boxes
    g = torch.Generator().manual_seed(1)
    B, H, W, P = 6, 97, 83, 64
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    boxes = rrc_boxes(H, W, B, rng=np.random.default_rng(3))
    boxes[0] = [0, 0, H, W]           # the plain Resize of default_transform
    boxes[1] = [5, 7, 1, 1]           # degenerate one-pixel crop
    boxes[2] = [0, 0, 32, 32]         # upsampling
    flip = np.array([0, 1, 0, 1, 1, 0], dtype=np.uint8)
    got = resize_crop(u8.cuda(), boxes, P, flip).cpu()
    src_f = u8.permute(0, 3, 1, 2).float() / 255.0
    for b in range(B):
        ref = _ref_resize(src_f[b], boxes[b], P, bool(flip[b]))
        assert float((got[b] - ref).abs().max()) <= 2e-6, b
    # float CHW source, no flip
    fsrc = torch.rand(B, 3, H, W, generator=g)
    got = resize_crop(fsrc.cuda(), boxes, P).cpu()
    for b in range(B):
        assert float((got[b] - _ref_resize(fsrc[b], boxes[b], P, False)).abs().max()) <= 2e-6, b
    with pytest.raises(ValueError):
        resize_crop(u8.cuda(), np.array([[0, 0, H + 1, W]] * B, dtype=np.int32), P)


def test_colour_jitter_matches_torchvision_float_formulas():
    from stil_tta_amd.augment import resize_crop
    g = torch.Generator().manual_seed(2)
    B, H, W, P = 4, 40, 48, 40
    src = torch.rand(B, 3, H, W, generator=g)
    boxes = np.array([[0, 0, H, W], [3, 4, 30, 30], [0, 8, 40, 40], [10, 0, 20, 48]], dtype=np.int32)
    jit = np.array([[1.3, 0.6, 1.5, 0.0], [0.4, 1.7, 0.3, 1.0], [1.0, 1.0, 1.0, 0.0], [1.8, 0.2, 1.0, 0.0]], dtype=np.float32)
    got = resize_crop(src.cuda(), boxes, P, None, jit).cpu()

    def gray(x):
        return (0.2989 * x[0] + 0.587 * x[1] + 0.114 * x[2]).unsqueeze(0)

    for b in range(B):
        br, ct, sa, gr = [float(v) for v in jit[b]]
        mean = gray((src[b] * br).clamp(0, 1)).mean()          # adjust_contrast's constant: the jittered SOURCE image
        x = _ref_resize(src[b], boxes[b], P, False)
        x = (x * br).clamp(0, 1)
        x = (ct * x + (1 - ct) * mean).clamp(0, 1)
        x = (sa * x + (1 - sa) * gray(x)).clamp(0, 1)
        if gr:
            x = gray(x).expand(3, -1, -1)
        assert float((got[b] - x).abs().max()) <= 5e-6, b
    assert torch.equal(got[2], resize_crop(src.cuda(), boxes, P).cpu()[2])   # identity factors change nothing


def test_gaussian_blur_matches_torchvision_formula():
    from stil_tta_amd.augment import gaussian_blur
    g = torch.Generator().manual_seed(5)
    B, H, W, k = 4, 50, 37, 29
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    sigma = torch.tensor([0.1, 2.0, 0.0, 0.9])
    got = gaussian_blur(u8.cuda(), sigma, k).cpu()
    src = u8.permute(0, 3, 1, 2).float() / 255.0
    for b in range(B):
        if float(sigma[b]) <= 0:   # copied (x * (1/255) on the device vs x / 255 here: one ulp)
            assert float((got[b] - src[b]).abs().max()) <= 1.2e-7
            continue
        x = torch.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k)
        w1 = torch.exp(-0.5 * (x / sigma[b]) ** 2)
        w1 = w1 / w1.sum()
        pad = F.pad(src[b:b + 1], (k // 2, k // 2, k // 2, k // 2), mode="reflect")
        ref = F.conv2d(pad, (w1[:, None] * w1[None, :]).expand(3, 1, k, k).contiguous(), groups=3)[0]
        assert float((got[b] - ref).abs().max()) <= 3e-6, b
    f32 = torch.rand(2, 3, 40, 40, generator=g)
    assert float((gaussian_blur(f32.cuda(), [1.3, 0.0], 9).cpu()[1] - f32[1]).abs().max()) == 0.0


def test_batch_builder_feeds_the_training_step():
    """ContrastiveBatchBuilder yields the reference's part tuple; a step on its output runs and is finite."""
    from stil_tta_amd import STiLModel
    from stil_tta_amd.augment import ContrastiveBatchBuilder
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    g = torch.Generator().manual_seed(4)
    N, fl = 48, [3, 4] + [1] * 3
    imgs = torch.randint(0, 256, (N, 80, 72, 3), generator=g, dtype=torch.uint8)
    table = torch.cat([torch.randint(0, 3, (N, 1), generator=g).float(), torch.randint(0, 4, (N, 1), generator=g).float(), torch.randn(N, 3, generator=g)], 1)
    labels = torch.randint(0, 5, (N,), generator=g)
    lab = ContrastiveBatchBuilder(imgs[:8], table[:8], labels[:8], 64, "dvm", 0.3, 0.95, labelled=True)
    unl = ContrastiveBatchBuilder(imgs[8:], table[8:], labels[8:], 64, "dvm", 0.3, 0.95, labelled=False)
    bl, bu = lab(torch.tensor([0, 5])), unl(torch.randperm(40, generator=g)[:14])
    im, tab, y, orig, ident = bu
    assert im[1].shape == (14, 3, 64, 64) and orig.shape == (14, 3, 64, 64) and tab[0].shape == tab[1].shape == (14, 5)
    assert y.dtype == torch.int64 and ident.dtype == torch.bool and not bool(ident.any()) and bool(bl[4].all())
    assert float(im[1].min()) >= 0.0 and float(im[1].max()) <= 1.0
    assert int((tab[0] != tab[1]).sum(1).max()) <= 1          # int(5 * 0.3) = 1 column per row
    assert bool((tab[1][:, :2] == tab[1][:, :2].round()).all())  # categorical columns stay valid codes (marginal values)
    torch.manual_seed(0)
    m = STiLModel(dict(model="resnet18", embedding_dim=512, field_lengths=fl, num_classes=5, start_epoch=0, batch_size=16, th1=0.3, img_size=64))
    m.setup_device("cuda"); m.train(); m.current_epoch = 1
    m.prototypes.copy_(F.normalize(torch.randn(5, 128, generator=g)).cuda())
    loss = train_step(m, StilAdam(m.flat, lr=1e-3), {"l": bl, "u": bu})
    assert bool(torch.isfinite(loss))


def _reflect101(i, n):
    period = 2 * n - 2
    i = i % period
    return torch.where(i < n, i, period - i)


def test_rotate_matches_bilinear_reflect101_restatement():
    """stil_aug_rotate against a float64 restatement of A.Rotate's geometry (cv2.getRotationMatrix2D about ((W-1)/2, (H-1)/2),
    bilinear, BORDER_REFLECT_101); 0 degrees copies, 90 / 180 degrees on a square image are exact permutations."""
    from stil_tta_amd.augment import rotate
    g = torch.Generator().manual_seed(8)
    img = torch.randint(0, 256, (4, 37, 53, 3), generator=g, dtype=torch.uint8)
    ang = torch.tensor([0.0, 17.5, -44.0, 180.0])
    got = rotate(img.cuda(), ang).cpu().double()
    src = img.permute(0, 3, 1, 2).double() / 255.0
    H, W = 37, 53
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    for b in range(4):
        th = float(ang[b]) * np.pi / 180.0
        ca, sa = np.cos(th), np.sin(th)
        cx, cy = (W - 1) / 2, (H - 1) / 2
        fx = ca * (xs - cx) - sa * (ys - cy) + cx
        fy = sa * (xs - cx) + ca * (ys - cy) + cy
        x0, y0 = torch.floor(fx), torch.floor(fy)
        lx, ly = fx - x0, fy - y0
        X0, X1 = _reflect101(x0.long(), W), _reflect101(x0.long() + 1, W)
        Y0, Y1 = _reflect101(y0.long(), H), _reflect101(y0.long() + 1, H)
        s = src[b]
        ref = (s[:, Y0, X0] * (1 - lx) + s[:, Y0, X1] * lx) * (1 - ly) + (s[:, Y1, X0] * (1 - lx) + s[:, Y1, X1] * lx) * ly
        tol = 1e-6 if b == 0 else 2e-4           # float32 sin / cos move the sample position by up to ~1e-5 pixels
        assert float((got[b] - ref).abs().max()) <= tol, (b, float((got[b] - ref).abs().max()))
    assert float((got[0] - src[0]).abs().max()) <= 1e-7
    sq = torch.rand(1, 3, 16, 16, generator=g)
    r90 = rotate(sq.cuda(), [90.0]).cpu()
    assert float((r90[0] - torch.rot90(sq[0], 1, dims=(1, 2))).abs().max()) <= 2e-6      # counter-clockwise


def test_hue_matches_torchvision_float_formulas():
    """stil_aug_hue against a restatement of torchvision's _rgb2hsv / _hsv2rgb / adjust_hue (float tensors); hue 0 is the
    identity, grey pixels do not move, and the grey conversion comes after the hue shift."""
    from stil_tta_amd.augment import adjust_hue_
    g = torch.Generator().manual_seed(9)
    img = torch.rand(3, 3, 20, 24, generator=g)
    img[0, :, :4] = img[0, :1, :4]            # some exactly grey pixels
    hue = torch.tensor([0.07, -0.1, 0.0])
    gray = torch.tensor([0.0, 1.0, 0.0])
    got = adjust_hue_(img.clone().cuda(), hue, gray).cpu()

    def tv_hue(x, hf):
        r, gg, b = x.unbind(0)
        maxc, minc = x.max(0).values, x.min(0).values
        eqc = maxc == minc
        cr = maxc - minc
        ones = torch.ones_like(maxc)
        s = cr / torch.where(eqc, ones, maxc)
        crd = torch.where(eqc, ones, cr)
        rc, gc, bc = (maxc - r) / crd, (maxc - gg) / crd, (maxc - b) / crd
        hr = (maxc == r) * (bc - gc)
        hg = ((maxc == gg) & (maxc != r)) * (2.0 + rc - bc)
        hb = ((maxc != gg) & (maxc != r)) * (4.0 + gc - rc)
        h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
        h = (h + hf) % 1.0
        i = torch.floor(h * 6.0)
        f = h * 6.0 - i
        i = i.to(torch.int32) % 6
        v = maxc
        p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
        q = torch.clamp(v * (1.0 - f * s), 0.0, 1.0)
        t = torch.clamp(v * (1.0 - (1.0 - f) * s), 0.0, 1.0)
        a1 = torch.stack((v, q, p, p, t, v)); a2 = torch.stack((t, v, v, q, p, p)); a3 = torch.stack((p, p, t, v, v, q))
        sel = torch.nn.functional.one_hot(i.long(), 6).permute(2, 0, 1).to(x.dtype)
        return torch.stack(((a1 * sel).sum(0), (a2 * sel).sum(0), (a3 * sel).sum(0)))

    ref0 = tv_hue(img[0], 0.07)
    ref1 = tv_hue(img[1], -0.1)
    ref1 = (0.2989 * ref1[0] + 0.587 * ref1[1] + 0.114 * ref1[2]).expand(3, -1, -1)
    assert float((got[0] - ref0).abs().max()) <= 3e-6 and float((got[1] - ref1).abs().max()) <= 3e-6
    assert float((got[2] - img[2]).abs().max()) == 0.0
    assert float((got[0, :, :4] - img[0, :, :4]).abs().max()) <= 1e-6


def test_match_batch_builders_feed_a_comatch_step():
    """EvalTrainBatchBuilder + StrongWeakBatchBuilder emit the batches of trainers/evaluate.py:50-83; a CoMatch step on them
    (cardiac transform family: rotation, no grayscale) runs and is finite."""
    from stil_tta_amd import CoMatch
    from stil_tta_amd.augment import EvalTrainBatchBuilder, StrongWeakBatchBuilder
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    g = torch.Generator().manual_seed(4)
    N, fl = 48, [3, 4] + [1] * 8
    imgs = torch.rand(N, 3, 80, 72, generator=g)          # cardiac images are float [0,1] (convert_to_ts_01)
    table = torch.cat([torch.randint(0, 3, (N, 1), generator=g).float(), torch.randint(0, 4, (N, 1), generator=g).float(), torch.randn(N, 8, generator=g)], 1)
    labels = torch.randint(0, 2, (N,), generator=g)
    lab = EvalTrainBatchBuilder(imgs[:8], table[:8], labels[:8], 64, "CAD", 0.3, 0.8)
    unl = StrongWeakBatchBuilder(imgs[8:], table[8:], labels[8:], 64, "CAD", 0.3, two_strong=True)
    (x_l, t_l), y_l, idx = lab(torch.tensor([0, 5]))
    views, y_u = unl(torch.randperm(40, generator=g)[:14])
    assert x_l.shape == (2, 3, 64, 64) and t_l.shape == (2, 10) and idx.tolist() == [0, 5] and len(views) == 3
    w, s0, s1 = views
    assert w[0].shape == s0[0].shape == s1[0].shape == (14, 3, 64, 64) and w[1].shape == (14, 10)
    clean = table[8:].cuda()
    assert all(float(v[0].min()) >= 0.0 and float(v[0].max()) <= 1.0 + 1e-6 for v in views)   # the post-crop blur may round 1 ulp above 1
    assert not torch.equal(s0[0], s1[0])                                           # two independent strong views
    torch.manual_seed(0)
    m = CoMatch(dict(model="resnet18", embedding_dim=512, field_lengths=fl, num_classes=2, start_epoch=0, batch_size=16, img_size=64, K=40,
                     co_threshold=0.5, contrast_th=0.5))
    m.setup_device("cuda"); m.train(); m.current_epoch = 1
    loss = train_step(m, StilAdam(m.flat, lr=1e-3), {"l": ((x_l, t_l), y_l, idx), "u": (views, y_u)})
    assert bool(torch.isfinite(loss)) and int(m.model.queue_ptr_s) == 14
