"""GPU parity of the whole STiL training step (zero_grad -> training_step -> backward -> Adam) through the
C ABI: (1) against the golden vectors recorded from the REAL reference (tests/golden/*.npz), (2) against the
CPU oracle on the same seeded inputs at the reference-native DVM shape, (3) size-independent properties at the
BASELINE.json bench shape.  Tolerance 1e-4 relative-to-scale (north_star: 1e-4 fp32)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# north_star: 1e-4 fp32.  The bar here is tighter than that, set from the measured margin (worst scaled error over all
# cases is printed by every test): |a - b| <= TOL * (1 + |b| + max|b|), which implies |dloss| <= 1e-4 * (1 + |loss|).
TOL = 3e-5
NORTH_STAR = 1e-4
_worst = {}


def _close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max() if b.size else 0.0
    err = np.abs(a - b)
    if err.size:
        _worst["scaled"] = max(_worst.get("scaled", 0.0), float((err / (1.0 + np.abs(b) + scale)).max()))
    return bool(np.all(err <= tol * (1.0 + np.abs(b) + scale))), float(err.max()) if err.size else 0.0


def _scaled(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if not b.size:
        return 0.0
    return float((np.abs(a - b) / (1.0 + np.abs(b) + np.abs(b).max())).max())


def _fwd_check(bad, key, gpu, ref32, ref64, report):
    """Forward parity of one quantity.  d = scaled error against the reference's (oracle's) CPU fp32 value.
    Pass when d <= TOL (3e-5, tighter than the 1e-4 north star).  A quantity that two correct fp32 evaluations cannot
    pin that tightly (the crafted, batch-centred teacher logits: large cancelling terms behind 50 eval-mode layers)
    must instead (i) stay inside the north-star 1e-4 and (ii) be as close to the float64 oracle as the reference's own
    fp32 value is: e_gpu <= 3 * e_ref + 1e-5.
    One fixture (dvm_r50_b32_224, y_hat_m_e) holds a quantity on which the REFERENCE's own fp32 value is 1.2e-4 from float64,
    i.e. further than the north star: no implementation can be pinned to 1e-4 of a value that is itself 1.2e-4 off (the device
    sat at 0.98e-4 of it with one rounding of the small products and at 1.04e-4 with another, while 6.3-6.6e-5 from float64 both
    times).  For such a quantity (e_ref > 1e-4) the bar is the exact value: the device must be AT LEAST as close to float64 as
    the reference is (e_gpu <= e_ref) and within the two distances of the reference (d <= 1e-4 + e_ref)."""
    gpu = np.asarray(gpu, dtype=np.float64)
    d = _scaled(gpu, ref32)
    _worst["scaled"] = max(_worst.get("scaled", 0.0), d)
    if d <= TOL:
        return
    e_ref, e_gpu = _scaled(ref32, ref64), _scaled(gpu, ref64)
    report.append((key, f"d={d:.2e}", f"e_gpu64={e_gpu:.2e}", f"e_ref64={e_ref:.2e}"))
    if d <= NORTH_STAR and e_gpu <= 3 * e_ref + 1e-5:
        return
    if e_ref > NORTH_STAR and e_gpu <= e_ref and d <= NORTH_STAR + e_ref:
        return
    bad.append((key, d, e_gpu, e_ref))


class _trace_decisions:
    """Keep the device's ReLU outputs / max-pool winners of the student pass (ops._trace) for the duration of a step."""

    def __enter__(self):
        from stil_tta_amd import ops
        ops._trace = {"relu": {}, "pool": {}}
        return ops._trace

    def __exit__(self, *a):
        from stil_tta_amd import ops
        ops._trace = None


def _device_decisions(m, trace):
    """ops._trace -> (relu, pool) in the oracle's tags and layouts (oracle/stil_oracle.py: force_decisions)."""
    names = {id(p): n for n, p in m.named_parameters() if not n.startswith("ema.")}
    relu = {}
    for pid, z in trace["relu"].items():
        tag = names[pid][: -len(".weight")]
        if not tag.startswith("model."):
            tag = "model.:" + tag                      # heads outside the backbone (projectors, CLUB estimators)
        mask = z.detach() > 0
        if z.ndim == 4:
            mask = mask.permute(0, 3, 1, 2)            # NHWC -> NCHW
        relu[tag] = mask.cpu().contiguous()
    pool = {}
    if "maxpool" in trace["pool"]:
        idx, H, W = trace["pool"]["maxpool"]           # [N, OH, OW, C] uint8: tap = ky*3 + kx of the winner
        t = idx.cpu().long()
        N, OH, OW, C = t.shape
        oy = torch.arange(OH).view(1, OH, 1, 1)
        ox = torch.arange(OW).view(1, 1, OW, 1)
        flat = (oy * 2 - 1 + t // 3) * W + (ox * 2 - 1 + t % 3)
        pool["model.encoder_imaging.maxpool"] = flat.permute(0, 3, 1, 2).contiguous()
    return relu, pool


def _check_flips(flips):
    """A unit may sit on the other side of its kink in float64 only if it is a genuine near-tie: its |pre-activation|
    (for the max-pool: the gap between the two candidates) must be inside the north-star forward tolerance of that tensor
    (an intermediate BatchNorm output over 16 x 2 x 2 samples is itself only good to ~2e-5 of its range in fp32)."""
    for t, (n, mag, scale) in flips.items():
        assert mag <= NORTH_STAR * (1.0 + scale), f"decision {t}: {n} units differ from float64 with |pre-activation| up to {mag:.2e} (tensor max {scale:.2e})"


def _grad_errors(params, g64, e32_of):
    """relative L2 error of every device gradient against the float64 oracle evaluated on the device's own
    decisions, judged against the reference's own fp32-vs-fp64 distance e32 OF THAT TENSOR, one realisation:
    err <= 3 * e32 + 1e-4 for EVERY tensor.  (Round 4 had widened the yardstick to per-module / two realisations because
    projector_imaging.bias sat at 2e-4 against ATen's 5e-5; round 5 found the operator -- a log-sum-exp error common to all rows
    of the CLIP softmax, amplified by the batch size in that one gradient, csrc/loss.hip -- fixed it, and the bound is per tensor
    again: DESIGN.md section 2.)"""
    e32s = {k: e32_of(k) for k, g in g64.items() if g is not None}
    bad, ratios, named = [], [], []
    for k, g in g64.items():
        p = params[k]
        if g is None:
            assert not p._stil_touched, k
            continue
        e32 = e32s[k]
        err = float((p._gslot.cpu().double() - g).norm() / (g.norm() + 1e-30))
        ratios.append(err / (3 * e32 + 1e-4))
        named.append((ratios[-1], k, f"{err:.2e}", f"e32 {e32:.2e}"))
        if err > 3 * e32 + 1e-4:
            bad.append(("grad " + k, err, e32))
    print("closest to their bound:", sorted(named, reverse=True)[:4])
    return bad, ratios


def _make_model(hp, sd):
    from stil_tta_amd import STiLModel
    d = dict(vars(hp))
    d["mi_dropout"] = False
    m = STiLModel(d)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m.setup_device("cuda")
    m.train()
    return m


def _to_dev(batch):
    out = {}
    for k in ("l", "u"):
        im, tab, y, orig, ident = batch[k]
        out[k] = ([im[0].cuda(), im[1].cuda()], [tab[0].cuda(), tab[1].cuda()], y.cuda(), orig.cuda(), ident.cuda())
    return out


def _named_params(m):
    return {n: p for n, p in m.named_parameters() if not n.startswith("ema.")}


def _check_flags(last, o, B_u):
    f = last["flags"].cpu()
    cs = f[:, 0]
    assert torch.equal(cs == 1, o["case1"]) and torch.equal(cs == 2, o["case2_i"])
    assert torch.equal(cs == 3, o["case2_t"]) and torch.equal(cs == 4, o["case3"])
    assert torch.equal(f[:, 1].bool(), o["mask1"])


from oracle.make_golden import CASES, SCALARS, build_case, run_oracle64  # noqa: E402
from oracle import stil_oracle as O  # noqa: E402

FWD_KEYS = ["y_hat_m", "y_hat_i", "y_hat_t", "x_si_enhance", "x_si", "x_ai", "x_st_enhance", "x_st", "x_at", "x_c", "feat_m", "feat_i",
            "feat_t", "y_hat_m_e", "y_hat_i_e", "y_hat_t_e", "feat_m_e", "pseudo_label_orig", "pseudo_label", "prediction",
            "class_sum", "class_count"]


@pytest.mark.parametrize("name", list(CASES))
def test_training_step_matches_reference_golden(name):
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    hp, sd, batch, epoch, mask_random, mi_masks = build_case(name)
    m = _make_model(hp, sd)
    m.current_epoch = epoch
    opt = StilAdam(m.flat, lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)
    with _trace_decisions() as trace:
        train_step(m, opt, _to_dev(batch), mask_random=mask_random, mi_masks=mi_masks)
        torch.cuda.synchronize()
        decisions = _device_decisions(m, trace)
    last = m.last
    bad, report = [], []
    o64 = run_oracle64(hp, sd, batch, epoch, mask_random, mi_masks, decisions=decisions)   # float64, the device's decisions
    for k in SCALARS + FWD_KEYS:
        _fwd_check(bad, k, last[k].detach().cpu().numpy(), fx["out_" + k], o64[k].numpy(), report)
    f = last["flags"].cpu().numpy()
    for cid, key in ((1, "case1"), (2, "case2_i"), (3, "case2_t"), (4, "case3")):
        assert np.array_equal(f[:, 0] == cid, fx["out_" + key]), key
    assert np.array_equal(f[:, 1].astype(bool), fx["out_mask1"])
    params = _named_params(m)
    # Gradients.  Yardstick: deep train-mode-BN backward amplifies fp32 rounding, so the fixture stores, per tensor,
    # gerr32 = relL2(REFERENCE fp32 gradient, float64 gradient); the device must be as close to float64 as the
    # reference's own CPU fp32 path is: err <= 3 * gerr32 + 1e-4 (1e-4 = north-star tolerance) for EVERY tensor.
    # The float64 oracle is evaluated on the DEVICE's piecewise-linear decisions (ReLU signs, max-pool winners, exported
    # through ops._trace): a pre-activation within rounding of 0 may land on the other side of the kink in another
    # fp32 (or the fp64) evaluation, which moves every gradient upstream of it by O(1) of that unit's share although
    # both runs are right; with the decisions pinned the comparison is between two evaluations of ONE smooth function.
    flips = {t: v for t, v in o64["flips"].items() if v[0]}
    _check_flips(flips)
    gbad, ratios = _grad_errors(params, o64["grads"], lambda k: float(fx["gerr32_" + k]))
    bad += gbad
    for key in fx.files:
        if key.startswith("gnorm_") and "g64norm_" + key[6:] not in fx.files:
            assert not params[key[6:]]._stil_touched, key  # grad None in the reference
        elif key.startswith("ssum_"):
            v = m.state_dict()[key[5:]].double()
            ref_abs = float(fx["sabs_" + key[5:]])
            if abs(float(v.sum()) - float(fx[key])) > 5e-5 * (1.0 + ref_abs):
                bad.append((key, float(v.sum()), float(fx[key])))
    print(f"[{name}] worst scaled forward error so far {_worst.get('scaled', 0.0):.2e}; beyond {TOL:g}: {report}")
    print(f"[{name}] gradient error / (3*gerr32 + 1e-4): median {np.median(ratios):.3f}, p90 {np.percentile(ratios, 90):.3f}, "
          f"max {np.max(ratios):.3f}; units deciding differently in float64: "
          f"{ {t: v[0] for t, v in flips.items()} }")
    assert not bad, f"{len(bad)} mismatches, first: {bad[:10]}"
    # inference hooks (STiLModel.py:424-474, 517-533), eval-mode BatchNorm, on the REFERENCE's post-step weights:
    # the first Adam step moves every weight by ~+-lr whatever the size of its gradient (m/sqrt(v) ~ sign g), so
    # noise-level gradients (exact zeros on one fp32 path, 1e-12 on another) put two correct paths 2*lr apart per
    # weight; the oracle's post-step state equals the reference's (same ATen kernels) and is loaded here.
    hp_o, sd_o, batch_o, epoch_o, mr_o, mm_o = build_case(name)
    O.full_step(sd_o, {}, 1, batch_o, hp_o, epoch_o, mr_o, mm_o)
    m.load_state_dict(sd_o)
    vx = [torch.cat((batch["l"][0][1], batch["u"][0][1])).cuda(), torch.cat((batch["l"][1][1], batch["u"][1][1])).cuda()]
    vy = torch.cat((batch["l"][2], batch["u"][2])).cuda()
    vloss = m.validation_step((vx, vy), 0)
    probs = m.test_step((vx, vy), 0)
    logged = m.logged if hasattr(m, "logged") else {}
    for key, val in (("val_loss", vloss), ("val_loss_ce", logged.get("multimodal.val.CEloss")), ("val_loss_itc", logged.get("multimodal.val.ITCloss")),
                     ("test_probs", probs)):
        ok, err = _close(val.detach().cpu().numpy(), fx["out_" + key], 2e-4)
        assert ok, (key, err)


@pytest.mark.parametrize("name", ["dvm_r50_pseudo", "cardiac_r50", "dvm_saint"])
def test_training_step_with_split_k_matches_reference_golden(name):
    """Split-K of the small NT products (ops._SPLITK; automatic since round 5 -- every golden test above runs with it) switched
    OFF: the unsplit form of the same step still meets the forward quantities of three reference goldens at the north-star 1e-4
    with exact CGPL decisions, and the split form is bit-identical on repetition (arrival tickets, slice-order sums:
    deterministic)."""
    from stil_tta_amd import ops
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    outs = []
    default_split = ops._SPLITK
    try:
        for rep in range(3):
            ops._SPLITK = rep < 2      # two split steps (repeatability), one unsplit
            hp, sd, batch, epoch, mask_random, mi_masks = build_case(name)
            m = _make_model(hp, sd)
            m.current_epoch = epoch
            opt = StilAdam(m.flat, lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)
            train_step(m, opt, _to_dev(batch), mask_random=mask_random, mi_masks=mi_masks)
            torch.cuda.synchronize()
            outs.append((m.flat.params.clone(), m.flat.grads.clone(), {k: m.last[k].detach().clone() for k in SCALARS + FWD_KEYS}, m.last["flags"].cpu().numpy()))
    finally:
        ops._SPLITK = default_split
    bad = []
    for which in (0, 2):
        last, f = outs[which][2], outs[which][3]
        for k in SCALARS + FWD_KEYS:
            d = _scaled(last[k].cpu().numpy(), fx["out_" + k])
            if d > NORTH_STAR:
                bad.append((which, k, d))
        for cid, key in ((1, "case1"), (2, "case2_i"), (3, "case2_t"), (4, "case3")):
            assert np.array_equal(f[:, 0] == cid, fx["out_" + key]), key
        assert np.array_equal(f[:, 1].astype(bool), fx["out_mask1"])
    assert not bad, bad
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "split-K step is not bit-identical on repetition"


def test_two_steps_match_oracle_dvm_native_shape():
    """DVM-native config (128 px, 4 cat + 13 con, K = 286), B = 16, two consecutive steps incl. Adam + EMA +
    BN running stats + prototype commit, against the CPU oracle on identical seeded inputs."""
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    from oracle.make_golden import randomize_state, make_mi_masks
    hp = O.default_hparams(batch_size=16, start_epoch=0, th1=0.02)
    sd = randomize_state(O.init_state(hp, seed=3), seed=4)
    g = torch.Generator().manual_seed(7)
    sd["prototypes"] = torch.nn.functional.normalize(torch.randn(hp.num_classes, hp.projection_dim, generator=g))
    m = _make_model(hp, {k: v.clone() for k, v in sd.items()})
    m.current_epoch = 1
    opt = StilAdam(m.flat, lr=hp.lr_eval)
    oopt = {}
    for step in (1, 2):
        batch = O.synthetic_batch(hp, 16, seed=100 + step)
        mr = torch.rand(14, generator=g).ge(0.5)
        mm = {0: make_mi_masks(16, 16, 17, 512, 4, 0.1, seed=step)}
        sd_before = {k: v.clone() for k, v in sd.items()}
        o = O.full_step(sd, oopt, step, batch, hp, 1, mr, mm)
        with _trace_decisions() as trace:
            train_step(m, opt, _to_dev(batch), mask_random=mr, mi_masks=mm)
            torch.cuda.synchronize()
            decisions = _device_decisions(m, trace)
        o64 = run_oracle64(hp, sd_before, batch, 1, mr, mm, decisions=decisions)  # float64 on the device's decisions
        o64free = run_oracle64(hp, sd_before, batch, 1, mr, mm)                    # float64, own decisions: the yardstick
        bad, report = [], []
        for k in SCALARS + FWD_KEYS:
            _fwd_check(bad, k, m.last[k].detach().cpu().numpy(), o[k].numpy(), o64[k].numpy(), report)
        _check_flags(m.last, o, 14)
        for nm, ref in (("threshold1_ratio", o["mask1"]), ("case1_ratio", o["case1"]), ("case2_i_ratio", o["case2_i"]),
                        ("case2_t_ratio", o["case2_t"]), ("case3_ratio", o["case3"])):   # logged like STiLModel.py:307-311
            assert abs(float(m.logged["multimodal.train." + nm]) - float(ref.float().mean())) < 1e-6, nm
        params = _named_params(m)

        def e32_of(k):  # the CPU fp32 oracle's own distance from float64 (both on their own decisions)
            g = o64free["grads"][k]
            return float((o["grads"][k].double() - g).norm() / (g.norm() + 1e-30))

        gbad, ratios = _grad_errors(params, o64["grads"], e32_of)   # every tensor: err <= 3 * e32 + 1e-4
        bad += [(step,) + b for b in gbad]
        print(f"step {step}: worst scaled forward error so far {_worst.get('scaled', 0.0):.2e}; beyond {TOL:g}: {report}")
        print(f"step {step}: gradient error / (3*e32 + 1e-4): median {np.median(ratios):.3f}, p90 {np.percentile(ratios, 90):.3f}, max {np.max(ratios):.3f}")
        msd = m.state_dict()
        tr = set(O.trainable_keys(sd))
        for k, v in sd.items():
            if k in tr:
                if float((msd[k].cpu() - v).abs().max()) > 2.2 * hp.lr_eval * step:
                    bad.append((step, "adam " + k))
            else:
                ok, err = _close(msd[k].cpu().double().numpy(), v.double().numpy(), 5e-5)
                if not ok:
                    bad.append((step, "state " + k, err))
        assert not bad, f"{len(bad)} mismatches, first: {bad[:10]}"
        # keep both sides on identical parameters so step 2 tests the step, not Adam's noise amplification
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
    # epoch end (STiLModel.py:408-415): prototypes <- class sums / counts; needs a confident sample of every class, which
    # 2 x 16 samples over 286 classes cannot give: seed the accumulators of the missing classes identically on both sides
    g2 = torch.Generator().manual_seed(99)
    add_sum = torch.randn(hp.num_classes, hp.projection_dim, generator=g2)
    missing = (sd["prototypes_count_sum"] < 1).float()
    sd["prototypes_sum"] += add_sum * missing
    sd["prototypes_count_sum"] += 2.0 * missing
    m.prototypes_sum.copy_(sd["prototypes_sum"].cuda()); m.prototypes_count_sum.copy_(sd["prototypes_count_sum"].cuda())
    O.training_epoch_end(sd)
    m.training_epoch_end()
    ok, err = _close(m.prototypes.cpu().numpy(), sd["prototypes"].numpy(), 2e-5)
    assert ok, ("prototypes after epoch end", err)
    assert float(m.prototypes_sum.abs().sum()) == 0.0 and float(m.prototypes_count_sum.abs().sum()) == 0.0
    m.prototypes_count_sum[3] = 0.0  # a class without a confident sample must trip the reference's assert (STiLModel.py:412)
    m.prototypes_count_sum[:3] = 1.0; m.prototypes_count_sum[4:] = 1.0
    with pytest.raises(AssertionError):
        m.training_epoch_end()


def test_baseline_shape_step_matches_oracle():
    """BASELINE.json configs[0] / [1] shape -- ResNet-50, 224 px, 16 categorical (cardinality 8) + 48 continuous columns,
    K = 286, B = 32 (4 labelled + 28 unlabelled), epoch > start_epoch so every loss term is live, injected mask_random
    and MI-layer dropout masks -- one full step (training_step + backward + Adam + EMA + prototype accumulation,
    STiLModel.py:228-386) through the HIP path against oracle.full_step on the host cores: every forward quantity of
    the golden test's key list, exact CGPL case ids / mask1, BN buffers + EMA teacher + prototype accumulators, and every
    gradient tensor against the float64 oracle on the device's decisions."""
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    from oracle.make_golden import randomize_state, make_mi_masks, craft_heads
    fl = [8] * 16 + [1] * 48
    B, K = 32, 286
    hp = O.default_hparams(img_size=224, field_lengths=fl, num_classes=K, batch_size=B, start_epoch=0, th1=0.5)
    sd = randomize_state(O.init_state(hp, seed=21), seed=22)
    g = torch.Generator().manual_seed(23)
    sd["prototypes"] = torch.nn.functional.normalize(torch.randn(K, hp.projection_dim, generator=g))
    batch = O.synthetic_batch(hp, B, seed=2022)
    B_u = len(batch["u"][2])
    mr = torch.rand(B_u, generator=g).ge(0.5)
    mm = {0: make_mi_masks(B, 49, len(fl), 512, 4, hp.mi_drop, seed=5)}
    # couple the three classifiers (imaging / tabular heads reuse the e_si / e_st slices of the multimodal head, whose
    # e_st slice is boosted: token means over 64 columns vary little between samples) so that the CGPL cases are mixed
    for pre_ in ("model.", "ema."):
        sd[pre_ + "classifier_multimodal.weight"][:, 512:1024] *= 0.3
        sd[pre_ + "classifier_multimodal.weight"][:, 1024:] *= 28.0
    sd = craft_heads(sd, batch, hp, 1, mr, scale=6.0, fi=1.0, ft=1.0)
    # th1 in the widest gap of the confidence ranking (prediction does not depend on th1): a mixed mask1 that no
    # rounding difference can flip
    with torch.no_grad():
        pre = O.training_step({k: v.clone() for k, v in sd.items()}, batch, hp, 1, mr, None)
    conf = pre["prediction"].max(1)[0].sort()[0]
    lo = B_u // 4
    gaps = conf[lo + 1: B_u - lo + 1] - conf[lo: B_u - lo]
    j = int(gaps.argmax()) + lo
    hp.th1 = float((conf[j] + conf[j + 1]) / 2)
    assert float(gaps.max()) > 1e-4
    sd0 = {k: v.clone() for k, v in sd.items()}

    m = _make_model(hp, {k: v.clone() for k, v in sd0.items()})
    m.current_epoch = 1
    with _trace_decisions() as trace:
        train_step(m, StilAdam(m.flat, lr=hp.lr_eval), _to_dev(batch), mask_random=mr, mi_masks=mm)
        torch.cuda.synchronize()
        decisions = _device_decisions(m, trace)
    o = O.full_step(sd, {}, 1, batch, hp, 1, mr, mm)
    assert 0 < int(o["mask1"].sum()) < B_u
    assert sum(int(o[c].sum()) > 0 for c in ("case1", "case2_i", "case2_t", "case3")) >= 3, "CGPL cases must be mixed"
    bad, report = [], []
    o64 = run_oracle64(hp, sd0, batch, 1, mr, mm, decisions=decisions)
    for k in SCALARS + FWD_KEYS:
        _fwd_check(bad, k, m.last[k].detach().cpu().numpy(), o[k].numpy(), o64[k].numpy(), report)
    _check_flags(m.last, o, B_u)
    msd = m.state_dict()
    tr = set(O.trainable_keys(sd))
    for k, v in sd.items():
        if k in tr:
            if float((msd[k].cpu() - v).abs().max()) > 2.2 * hp.lr_eval:
                bad.append(("adam " + k,))
        else:  # BN running statistics, the EMA teacher, prototype accumulators
            ok, err = _close(msd[k].cpu().double().numpy(), v.double().numpy(), 5e-5)
            if not ok:
                bad.append(("state " + k, err))
    # gradients: float64 on the device's decisions; yardstick = the CPU fp32 oracle on the same decisions, per tensor
    with O.force_decisions(*decisions):
        o32 = O.full_step({k: v.clone() for k, v in sd0.items()}, {}, 1, batch, hp, 1, mr, mm)
    flips = {t: v for t, v in o64["flips"].items() if v[0]}
    _check_flips(flips)

    def e32_of(k):
        g64 = o64["grads"][k]
        return float((o32["grads"][k].double() - g64).norm() / (g64.norm() + 1e-30))

    gbad, ratios = _grad_errors(_named_params(m), o64["grads"], e32_of)
    bad += gbad
    print(f"baseline shape: worst scaled forward error so far {_worst.get('scaled', 0.0):.2e}; beyond {TOL:g}: {report}")
    print(f"baseline shape: gradient error / (3*e32 + 1e-4): median {np.median(ratios):.3f}, p90 {np.percentile(ratios, 90):.3f}, "
          f"max {np.max(ratios):.3f}; th1 = {hp.th1:.4f}, mask1 {int(o['mask1'].sum())}/{B_u}, cases "
          f"{[int(o[c].sum()) for c in ('case1', 'case2_i', 'case2_t', 'case3')]}; float64 decides differently on { {t: v[0] for t, v in flips.items()} }")
    assert not bad, f"{len(bad)} mismatches, first: {bad[:10]}"


_CASE_CACHE = {}


def _full_size_case(key, over, B, seeds, boost=(0.3, 28.0), scale=6.0, min_gap=2e-4, min_cases=3):
    """A bench-shape configuration with crafted heads (mixed CGPL cases, th1 in the widest gap of the confidence ranking so that
    no rounding can flip mask1): -> hp, sd0 (the state the device gets), batch, mask_random, MI masks, o (oracle.training_step
    on the host cores under no_grad).  Two oracle passes; cached per key (the forward and the backward test of one
    configuration share them).  The head coupling / centring is oracle.make_golden.craft_heads' with its oracle pass shared."""
    if key in _CASE_CACHE:
        return _CASE_CACHE[key]
    from oracle.make_golden import randomize_state, make_mi_masks
    hp = O.default_hparams(batch_size=B, start_epoch=0, th1=0.5, **over)
    K = hp.num_classes
    sd = randomize_state(O.init_state(hp, seed=seeds[0]), seed=seeds[1])
    g = torch.Generator().manual_seed(seeds[2])
    sd["prototypes"] = torch.nn.functional.normalize(torch.randn(K, hp.projection_dim, generator=g))
    batch = O.synthetic_batch(hp, B, seed=2022)
    B_u = len(batch["u"][2])
    mr = torch.rand(B_u, generator=g).ge(0.5)
    Ni, Nt = (hp.img_size // 32) ** 2, len(hp.field_lengths)
    mm = {0: make_mi_masks(B, Ni, Nt, hp.multimodal_embedding_dim, 4, hp.mi_drop, seed=seeds[3])}
    if hp.tabular_encoder == "saint":  # FF dropout p = 0.8 of the SAINT column / row feed-forwards (oracle.make_golden.build_case)
        g2 = torch.Generator().manual_seed(seeds[3] + 1)
        nf = Nt + 1
        mm["saint"] = {"ff_col": torch.rand(B, nf, 4 * O.SAINT_DIM, generator=g2) >= hp.saint_ff_drop,
                       "ff_row": torch.rand(1, B, 4 * O.SAINT_DIM * nf, generator=g2) >= hp.saint_ff_drop}
    C = hp.multimodal_embedding_dim
    for pre_ in ("model.", "ema."):
        sd[pre_ + "classifier_multimodal.weight"][:, C:2 * C] *= boost[0]
        sd[pre_ + "classifier_multimodal.weight"][:, 2 * C:] *= boost[1]
        Wm = sd[pre_ + "classifier_multimodal.weight"]
        sd[pre_ + "classifier_imaging.weight"][:, :C] = Wm[:, :C]
        sd[pre_ + "classifier_tabular.weight"][:, :C] = Wm[:, 2 * C:]
        sd[pre_ + "classifier_imaging.weight"][:, C:] *= 0.2
        sd[pre_ + "classifier_tabular.weight"][:, C:] *= 0.2
    with torch.no_grad():
        o1 = O.training_step({k: v.clone() for k, v in sd.items()}, batch, hp, 1, mr, mm)
    for nm, key_ in (("classifier_multimodal", "y_hat_m_e"), ("classifier_imaging", "y_hat_i_e"), ("classifier_tabular", "y_hat_t_e")):
        mean = o1[key_].mean(0)
        for pre_ in ("model.", "ema."):
            sd[pre_ + nm + ".weight"] *= scale
            sd[pre_ + nm + ".bias"] = (sd[pre_ + nm + ".bias"] - mean) * scale
    # th1 in the widest gap of the confidence ranking.  Under the crafted heads the teacher's multimodal logits are
    # (y_hat_m_e - mean) * scale and the prototype similarities are unchanged, so `prediction` follows from pass 1
    # (STiLModel.py:276-296); the second oracle pass below re-derives it and is what the device is compared with.
    zm = (o1["y_hat_m_e"] - o1["y_hat_m_e"].mean(0)) * scale
    tp = torch.softmax(o1["feat_m_e"][B - B_u:] @ sd["prototypes"].t() / hp.temperature, dim=1)
    pred = hp.rate_pseudo * torch.softmax(zm[B - B_u:], dim=1) + (1 - hp.rate_pseudo) * tp
    conf = pred.max(1)[0].sort()[0]
    lo = B_u // 4
    gaps = conf[lo + 1: B_u - lo + 1] - conf[lo: B_u - lo]
    j = int(gaps.argmax()) + lo
    hp.th1 = float((conf[j] + conf[j + 1]) / 2)
    assert float(gaps.max()) > min_gap, f"no rounding-proof th1 on this batch (widest gap {float(gaps.max()):.2e})"
    sd0 = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        o = O.training_step(sd, batch, hp, 1, mr, mm)
    assert float((o["prediction"] - pred).abs().max()) < 1e-3
    assert 0 < int(o["mask1"].sum()) < B_u
    assert sum(int(o[c].sum()) > 0 for c in ("case1", "case2_i", "case2_t", "case3")) >= min_cases, \
        f"CGPL cases must be mixed: {[int(o[c].sum()) for c in ('case1', 'case2_i', 'case2_t', 'case3')]}"
    _CASE_CACHE[key] = (hp, sd0, batch, mr, mm, o)
    return _CASE_CACHE[key]


def _forward_vs_oracle(label, hp, sd0, batch, mr, mm, o):
    """One no_grad training_step through the HIP path against the oracle's forward quantities (north-star 1e-4) and EXACT CGPL
    case ids / mask1."""
    B_u = len(batch["u"][2])
    m = _make_model(hp, {k: v.clone() for k, v in sd0.items()})
    m.current_epoch = 1
    with torch.no_grad():
        m.training_step(_to_dev(batch), 0, mask_random=mr, mi_masks=mm)
    torch.cuda.synchronize()
    bad, beyond = [], []
    for k in SCALARS + FWD_KEYS:
        d = _scaled(m.last[k].detach().cpu().numpy(), o[k].numpy())
        _worst["scaled"] = max(_worst.get("scaled", 0.0), d)
        if d > TOL:
            beyond.append((k, f"{d:.2e}"))
        if d > NORTH_STAR:
            bad.append((k, d))
    _check_flags(m.last, o, B_u)
    print(f"{label}: th1 = {hp.th1:.4f}, mask1 {int(o['mask1'].sum())}/{B_u}, cases "
          f"{[int(o[c].sum()) for c in ('case1', 'case2_i', 'case2_t', 'case3')]}; beyond {TOL:g} (bar {NORTH_STAR:g}): {beyond}")
    assert not bad, f"{label}: {len(bad)} forward quantities beyond the north-star 1e-4: {bad[:10]}"
    del m
    torch.cuda.empty_cache()


CONFIGS1 = dict(img_size=224, field_lengths=[8] * 16 + [1] * 48, num_classes=286)
# configs[3] (config_dvm_STiL_SAINT.yaml) and configs[4] (config_cardiac_STiL.yaml deltas, as bench.py --variant cardiac) at the
# shapes the round-4 profiles measured them on (profiles/r04z_{saint,cardiac64,cardiac16}_*)
CONFIGS3 = dict(img_size=224, field_lengths=[8] * 16 + [1] * 48, num_classes=286, tabular_encoder="saint")
CONFIGS4 = dict(img_size=128, field_lengths=[4] * 26 + [1] * 49, num_classes=2, target="CAD", rate_pseudo=0.95, ema_momentum=0.4, beta=1.0,
                gamma=1.0, lr_eval=1e-3)


def test_configs1_full_size_forward_matches_oracle():
    """BASELINE.json configs[1] at FULL size -- ResNet-50, 224 px, 16 categorical + 48 continuous columns, K = 286, B = 256
    (32 labelled + 224 unlabelled), pseudo-label phase, injected mask_random and MI-layer dropout masks: every forward
    quantity of STiLModel.training_step (:228-345: logits, features, teacher outputs, CGPL / PGLS pseudo-labels, all 11 loss
    terms, class sums) through the HIP path against oracle.training_step on the host cores (no_grad: two oracle passes,
    the first one also crafts the heads), and EXACT CGPL case ids / mask1.  Backward at this size: the next test."""
    _forward_vs_oracle("configs[1] full size", *_full_size_case("configs1_b256", CONFIGS1, 256, (31, 32, 33, 6)))


def test_configs1_full_size_backward_matches_oracle():
    """configs[1] at FULL size, the WHOLE optimisation step (STiLModel.py:228-386 + backward + Adam :557-570): B = 256 takes code
    paths B = 32 never does (weight-gradient M-splits, > 256-tile BatchNorm finalisation in two launches, more XCD-remapped
    tiles).  One device step (decisions exported), one fp32 oracle.full_step on the device's ReLU / max-pool decisions (about two
    minutes on the box's 16 cores), then: EVERY gradient tensor relL2(device, fp32 oracle) <= 4 e32 + 1e-4, where e32 is the
    REFERENCE's own fp32-vs-float64 distance of that tensor at this architecture (fixture dvm_r50_b32_224) -- the
    B = 32 test's bar (3 e32 + 1e-4 against float64) plus the fp32 oracle's own e32; BatchNorm running statistics, the EMA
    teacher and the prototype accumulators at 5e-5; |delta Adam| <= 2.2 lr."""
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    hp, sd0, batch, mr, mm, o_fwd = _full_size_case("configs1_b256", CONFIGS1, 256, (31, 32, 33, 6))
    fx = np.load(os.path.join(GOLD, "dvm_r50_b32_224.npz"))
    m = _make_model(hp, {k: v.clone() for k, v in sd0.items()})
    m.current_epoch = 1
    with _trace_decisions() as trace:
        train_step(m, StilAdam(m.flat, lr=hp.lr_eval), _to_dev(batch), mask_random=mr, mi_masks=mm)
        torch.cuda.synchronize()
        decisions = _device_decisions(m, trace)
    params = _named_params(m)
    gdev = {k: params[k]._gslot.detach().cpu().double() for k in params}
    touched = {k: bool(params[k]._stil_touched) for k in params}
    msd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    B_u = len(batch["u"][2])
    flags = m.last["flags"].cpu()
    del m, trace, params
    torch.cuda.empty_cache()
    sd = {k: v.clone() for k, v in sd0.items()}
    with O.force_decisions(*decisions) as d:
        o = O.full_step(sd, {}, 1, batch, hp, 1, mr, mm)
    flips = {t: v for t, v in d.get("flips", {}).items() if v[0]}
    _check_flips(flips)
    cs = flags[:, 0]
    assert torch.equal(cs == 1, o["case1"]) and torch.equal(cs == 2, o["case2_i"]) and torch.equal(cs == 3, o["case2_t"]) and torch.equal(cs == 4, o["case3"])
    assert torch.equal(flags[:, 1].bool(), o["mask1"])
    bad, named = [], []
    for k, g in o["grads"].items():
        if g is None:
            assert not touched[k], k
            continue
        e32 = float(fx["gerr32_" + k])
        err = float((gdev[k] - g.double()).norm() / (g.double().norm() + 1e-30))
        named.append((err / (4 * e32 + 1e-4), k, f"{err:.2e}", f"e32 {e32:.2e}"))
        if err > 4 * e32 + 1e-4:
            bad.append(("grad " + k, err, e32))
    ratios = [r[0] for r in named]
    tr = set(O.trainable_keys(sd))
    for k, v in sd.items():
        if k in tr:
            if float((msd[k] - v).abs().max()) > 2.2 * hp.lr_eval:
                bad.append(("adam " + k,))
        else:  # BN running statistics, the EMA teacher, prototype accumulators
            ok, err = _close(msd[k].double().numpy(), v.double().numpy(), 5e-5)
            if not ok:
                bad.append(("state " + k, err))
    print(f"configs[1] full size, backward: gradient error / (4*e32 + 1e-4): median {np.median(ratios):.3f}, p90 {np.percentile(ratios, 90):.3f}, "
          f"max {np.max(ratios):.3f}; closest to their bound: {sorted(named, reverse=True)[:4]}; the fp32 oracle decides differently on "
          f"{ {t: v[0] for t, v in flips.items()} }")
    assert not bad, f"{len(bad)} mismatches, first: {bad[:10]}"


def test_configs3_saint_bench_shape_forward_matches_oracle():
    """BASELINE.json configs[3] (config_dvm_STiL_SAINT: SAINT tabular encoder) at the shape bench.py --variant saint measures --
    B = 256, 224 px, 16 categorical + 48 continuous columns, K = 286: the intersample attention runs over the 256 rows of
    65 x 32 features (SAINT/model_util.py:111-129, STiLModel_SAINT_backbone.py:159-184), the golden case only over 16 rows of
    8 x 32.  Forward of the whole step against the oracle (no_grad), exact CGPL ids / mask1."""
    _forward_vs_oracle("configs[3] SAINT, B=256", *_full_size_case("configs3_b256", CONFIGS3, 256, (41, 42, 43, 7)))


@pytest.mark.parametrize("B", [64, 16])
def test_configs4_cardiac_bench_shape_forward_matches_oracle(B):
    """BASELINE.json configs[4] (config_cardiac_STiL.yaml: K = 2, SimCLR projection heads, rate_pseudo 0.95, EMA momentum 0.4) at
    the shapes the cardiac benches run -- 128 px, 26 categorical + 49 continuous columns, B = 64 (one GPU) and B = 16 (the share
    of one of the 4 GPUs the config names): forward of the whole step against the oracle (no_grad), exact CGPL ids / mask1.
    With two classes `case3` (image and table agree against the multimodal head) is rare: two mixed cases are asked for."""
    _forward_vs_oracle(f"configs[4] cardiac, B={B}", *_full_size_case(f"configs4_b{B}", CONFIGS4, B, (51 + B, 52, 53, 8), min_cases=2, min_gap=1e-4))


@pytest.mark.parametrize("B", [64, 16])
def test_configs4_cardiac_bench_shape_backward_matches_oracle(B):
    """configs[4] at its bench shapes, the WHOLE optimisation step (backward + Adam): 128 px, 26 + 49 columns, K = 2, SimCLR heads,
    B = 64 and the 16-sample share of one GPU -- the regime where the small NT products run split over K (automatic since round 5)
    and the step is launch-bound.  Same protocol as the BASELINE-shape test: every gradient tensor against the float64 oracle on
    the device's decisions, yardstick = the fp32 oracle's own distance from float64 on the same decisions, per tensor
    (err <= 3 e32 + 1e-4); BatchNorm statistics / EMA teacher / prototype sums at 5e-5; |delta Adam| <= 2.2 lr."""
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    hp, sd0, batch, mr, mm, o_fwd = _full_size_case(f"configs4_b{B}", CONFIGS4, B, (51 + B, 52, 53, 8), min_cases=2, min_gap=1e-4)
    m = _make_model(hp, {k: v.clone() for k, v in sd0.items()})
    m.current_epoch = 1
    with _trace_decisions() as trace:
        train_step(m, StilAdam(m.flat, lr=hp.lr_eval), _to_dev(batch), mask_random=mr, mi_masks=mm)
        torch.cuda.synchronize()
        decisions = _device_decisions(m, trace)
    o64 = run_oracle64(hp, sd0, batch, 1, mr, mm, decisions=decisions)
    sd = {k: v.clone() for k, v in sd0.items()}
    with O.force_decisions(*decisions):
        o32 = O.full_step(sd, {}, 1, batch, hp, 1, mr, mm)
    flips = {t: v for t, v in o64["flips"].items() if v[0]}
    _check_flips(flips)
    _check_flags(m.last, o32, len(batch["u"][2]))

    def e32_of(k):
        g64 = o64["grads"][k]
        return float((o32["grads"][k].double() - g64).norm() / (g64.norm() + 1e-30))

    bad, ratios = _grad_errors(_named_params(m), o64["grads"], e32_of)
    msd = m.state_dict()
    tr = set(O.trainable_keys(sd))
    for k, v in sd.items():
        if k in tr:
            if float((msd[k].cpu() - v).abs().max()) > 2.2 * hp.lr_eval:
                bad.append(("adam " + k,))
        else:
            ok, err = _close(msd[k].cpu().double().numpy(), v.double().numpy(), 5e-5)
            if not ok:
                bad.append(("state " + k, err))
    print(f"configs[4] cardiac B={B}, backward: gradient error / (3*e32 + 1e-4): median {np.median(ratios):.3f}, p90 {np.percentile(ratios, 90):.3f}, "
          f"max {np.max(ratios):.3f}; float64 decides differently on {sum(v[0] for v in flips.values())} units")
    assert not bad, f"{len(bad)} mismatches, first: {bad[:10]}"


def test_five_step_trajectory_matches_oracle():
    """FIVE consecutive optimisation steps (zero_grad -> training_step -> backward -> Adam, STiLModel.py:228-386, :557-570) of a
    ResNet-18 case in the pseudo-label phase against the trajectory the REAL reference took on the same five seeded batches
    and mask_random draws (tests/golden/traj_r18.npz, written by oracle/make_golden_traj.py): every loss term of every step,
    the logged mask / case ratios exactly, and the final state (strided sample + sum of every tensor).  Single-step parity
    cannot see what is carried ACROSS steps (Adam moments and bias correction, EMA teacher, BatchNorm running statistics,
    prototype sums); this can.  Yardstick: Adam's update is ~lr * sign(g) whatever the gradient's size, so two correct fp32
    evaluations separate along the trajectory (an element whose gradient changes sign under rounding lands 2 lr away); the
    fixture carries, for every quantity, how far the float64 oracle is from the reference and how far the fp32 oracle moves
    under one-ulp perturbations of the initial weights (~1e-4 of the loss within five steps).  The device must stay within
    three times that spread: |gpu - ref| <= 3 max(|f64 - ref|, spread) + TOL (1 + |ref|)."""
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    from oracle import make_golden_traj as T
    fx = np.load(os.path.join(GOLD, "traj_r18.npz"))
    hp, sd, batches, masks = T.build()
    assert int(fx["steps"]) == T.STEPS and int(fx["B"]) == T.B
    m = _make_model(hp, sd)
    m.current_epoch = T.EPOCH
    opt = StilAdam(m.flat, lr=hp.lr_eval, weight_decay=hp.weight_decay_eval)
    names = {"loss": "loss", "loss_ce": "loss_ce", "loss_itc": "loss_itc", "loss_club_i": "loss_club_i", "loss_club_i_est": "loss_club_i_est",
             "loss_club_t": "loss_club_t", "loss_club_t_est": "loss_club_t_est", "loss_m_u": "loss_m_u", "loss_i_u": "loss_i_u", "loss_t_u": "loss_t_u"}
    bad, worst = [], 0.0
    for s_ in range(T.STEPS):
        train_step(m, opt, _to_dev(batches[s_]), mask_random=masks[s_])
        torch.cuda.synchronize()
        for k, lk in names.items():
            got, ref, f64 = float(m.last[lk].detach()), float(fx["ref_" + k][s_]), float(fx["o64_" + k][s_])
            bound = 3 * max(abs(f64 - ref), float(fx["sens_" + k][s_])) + TOL * (1 + abs(ref))
            worst = max(worst, abs(got - ref) / bound)
            if abs(got - ref) > bound:
                bad.append((s_, k, got, ref, f64))
        for nm, key in (("threshold1_ratio", "mask1_ratio"), ("case1_ratio", "case1_ratio"), ("case3_ratio", "case3_ratio")):
            if abs(float(m.logged["multimodal.train." + nm]) - float(fx["ref_" + key][s_])) > 1e-6:   # integer decisions: exact
                bad.append((s_, nm, float(m.logged["multimodal.train." + nm]), float(fx["ref_" + key][s_])))
    msd = m.state_dict()
    n_s = int(fx["sample"])
    ratios, sums = [], []
    for key in fx.files:
        if not key.startswith("state/"):
            continue
        k = key[6:]
        v = msd[k].detach().cpu()
        if not v.is_floating_point():
            assert np.array_equal(v.numpy(), fx[key]), k
            continue
        f = v.reshape(-1)
        got = f[::max(1, f.numel() // n_s)][:n_s].double().numpy()
        ref = fx[key].astype(np.float64)
        err = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))
        bound = 3 * max(float(fx["dist64/" + k]), float(fx["distp/" + k])) + 1e-4
        ratios.append(err / bound)
        if err > bound:
            bad.append(("state " + k, err, float(fx["dist64/" + k]), float(fx["distp/" + k])))
        dsum = abs(float(v.double().sum()) - float(fx["sum/" + k]))
        sbound = 3 * max(float(fx["dsum64/" + k]), float(fx["dsump/" + k])) + 1e-4 * (1.0 + abs(float(fx["sum/" + k])) + float(fx["norm/" + k]))
        sums.append(dsum / sbound)
        if dsum > sbound:
            bad.append(("sum " + k, dsum, float(fx["dsum64/" + k]), float(fx["dsump/" + k])))
    print(f"trajectory: worst loss-term error / bound {worst:.3f}; final state (sampled relL2) error / bound: median {np.median(ratios):.3f}, "
          f"p90 {np.percentile(ratios, 90):.3f}, max {np.max(ratios):.3f}; checksum error / bound: max {np.max(sums):.3f}")
    assert not bad, f"{len(bad)} mismatches, first: {bad[:12]}"


@pytest.mark.parametrize("B", [32, 256])
def test_bench_shape_properties(B):
    """BASELINE configs[1] shape (224 px, 64 columns, K = 286) at B = 32 and at the bench's B = 256: size-independent
    properties."""
    from stil_tta_amd import STiLModel
    from stil_tta_amd.driver import train_step, synthetic_batch
    from stil_tta_amd.flat import StilAdam
    torch.manual_seed(0)
    fl = [8] * 16 + [1] * 48
    m = STiLModel(dict(field_lengths=fl, num_classes=286, start_epoch=0, batch_size=B, th1=0.0))
    m.setup_device("cuda")
    m.train()
    m.current_epoch = 1
    m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(286, 128, device="cuda")))
    opt = StilAdam(m.flat, lr=1e-4)
    batch = synthetic_batch(fl, 286, B, 224, device="cuda")
    ema0 = m.flat.ema.clone()
    p0 = m.flat.params.clone()
    loss = train_step(m, opt, batch)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss))
    L = m.last
    for k in ("feat_m", "feat_i", "feat_t", "feat_m_e"):  # unit rows
        assert float((L[k].norm(dim=1) - 1).abs().max()) < 1e-5, k
    for k in ("pseudo_label", "pseudo_label_orig", "prediction"):  # rows are distributions
        assert float((L[k].sum(1) - 1).abs().max()) < 1e-5, k
    f = L["flags"].cpu()
    assert int(((f[:, 0] >= 1) & (f[:, 0] <= 4)).sum()) == B - B // 8  # the four cases partition the unlabelled rows
    # th1 = 0: every row is confident -> counts sum to the batch; class sums add up to the feature sum
    assert abs(float(L["class_count"].sum()) - float(B)) < 1e-4 * B
    assert float((L["class_sum"].sum(0) - L["feat_m_e"].sum(0)).abs().max()) < 1e-4
    # EMA: e' = m e + (1-m) p (student BEFORE Adam), exact
    n = m.flat.n_backbone_params
    ref = ema0[:n].mul(0.996).add((1.0 - 0.996) * p0[:n])
    assert torch.equal(m.flat.ema[:n], ref)
    # Adam moved every touched parameter by at most lr (first step: |m/sqrt(v)| <= 1)
    d = (m.flat.params - p0).abs()
    assert float(d[: n].max()) <= 1.001e-4 and float(d.max()) > 0
    # total loss is the weighted sum of its logged parts (STiLModel.py:345)
    hp = m.hp
    tot = hp.alpha * L["loss_ce"] + hp.beta * L["loss_itc"] + hp.gamma * (L["loss_club_i"] + L["loss_club_i_est"] + L["loss_club_t"] + L["loss_club_t_est"]) \
        + hp.rate_pt * L["loss_pt"] + hp.rate_uce * (L["loss_m_u"] + L["loss_i_u"] + L["loss_t_u"])
    assert abs(float(tot) - float(L["loss"])) < 1e-4 * (1 + abs(float(tot)))
    # determinism: an identical second model/step gives bit-identical loss and gradients
    g1 = m.flat.grads.clone()
    torch.manual_seed(0)
    m2 = STiLModel(dict(field_lengths=fl, num_classes=286, start_epoch=0, batch_size=B, th1=0.0))
    m2.setup_device("cuda"); m2.train(); m2.current_epoch = 1
    m2.flat.params.copy_(p0); m2.flat.copy_student_to_teacher(); m2.flat.ema.copy_(ema0)
    m2.prototypes.copy_(m.prototypes)
    loss2 = train_step(m2, StilAdam(m2.flat, lr=1e-4), batch)
    assert torch.equal(loss, loss2) and torch.equal(g1, m2.flat.grads)


def test_mislabelled_parts_are_rejected():
    """STiLModel.py:237-238: batch['l'] must be all labelled, batch['u'] all unlabelled."""
    from stil_tta_amd import STiLModel
    from stil_tta_amd.driver import synthetic_batch
    fl = [3, 4] + [1] * 3
    m = STiLModel(dict(model="resnet18", embedding_dim=512, field_lengths=fl, num_classes=5, batch_size=16))
    m.setup_device("cuda"); m.train()
    b = synthetic_batch(fl, 5, 16, 64, seed=1, device="cuda")
    im, tab, y, orig, ident = b["u"]
    bad = dict(b, u=(im, tab, y, orig, torch.ones_like(ident)))
    with pytest.raises(AssertionError):
        m.training_step(bad, 0)


def test_empty_and_ragged_inputs_fail_loudly():
    from stil_tta_amd import ops
    x = torch.randn(4, 8)
    with pytest.raises(RuntimeError):  # CPU tensors are refused: no fallback
        ops.linear(x, torch.randn(3, 8), None)
    with pytest.raises(RuntimeError):  # non-contiguous
        ops.L2NormFn.apply(torch.randn(8, 4, device="cuda").t())
    with pytest.raises(RuntimeError):  # conv with Cin not a multiple of 16
        ops.gemm_nt(torch.randn(9, 3, device="cuda"), torch.randn(4, 27, device="cuda"), 9, 4, 27, geom=(3, 3, 3, 3, 3, 3, 3, 1, 1, 0))


@pytest.mark.parametrize("two_streams", [True, False])
def test_graphed_step_replays_the_eager_step_bit_for_bit(two_streams):
    """hipGraph capture of the whole step (driver.GraphedTrainStep): same kernels, same operands -> identical bits,
    including fresh dropout masks per replay (device-side RNG step counter).  two_streams: the capture keeps the side
    stream's forks and joins (teacher beside student, weight gradients beside the input-gradient chain) as graph
    dependencies; False: every captured launch inline on the capturing stream (STIL_GRAPH_SIDE=0)."""
    from stil_tta_amd import STiLModel, ops
    from stil_tta_amd.driver import train_step, synthetic_batch, GraphedTrainStep
    ops._side.in_capture = two_streams
    from stil_tta_amd.flat import StilAdam
    fl = [3, 4] + [1] * 3

    def make():
        torch.manual_seed(0)
        m = STiLModel(dict(model="resnet18", embedding_dim=512, field_lengths=fl, num_classes=5, start_epoch=0, batch_size=16, th1=0.3))
        m.setup_device("cuda"); m.train(); m.current_epoch = 1
        m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(5, 128, generator=torch.Generator().manual_seed(1))).cuda())
        return m, StilAdam(m.flat, lr=1e-3)

    batches = [synthetic_batch(fl, 5, 16, 64, seed=s, device="cuda") for s in range(6)]
    me, oe = make()
    mg, og = make()
    gs = GraphedTrainStep(mg, og, batches[0], warmup=2)  # capture (incl. its warm-up steps) leaves the state untouched
    masks_seen = []
    for b in batches:
        le = train_step(me, oe, b)
        lg = gs(b)
        torch.cuda.synchronize()
        assert torch.equal(le, lg), (float(le), float(lg))
        masks_seen.append(mg.last["mask_random"].clone())
    assert torch.equal(me.flat.params, mg.flat.params) and torch.equal(me.flat.ema, mg.flat.ema)
    assert torch.equal(me.prototypes_sum, mg.prototypes_sum)
    assert any(not torch.equal(masks_seen[0], m_) for m_ in masks_seen[1:]), "replays must draw fresh random masks"
    ops._side.in_capture = False


def test_side_stream_pipeline_is_bit_exact():
    """Teacher beside the student (layer-wise BN hand-over) + weight gradients on the side stream + the student's tabular encoder
    (forward and, through autograd, backward) on the branch stream produce exactly the bits of the single-stream order "student,
    momentum_update_ema, teacher, backward" (same kernels, same operands); also with the (opt-in) branch stream switched off, the default."""
    from stil_tta_amd import STiLModel, ops
    from stil_tta_amd.driver import synthetic_batch, train_step
    from stil_tta_amd.flat import StilAdam
    fl = [3, 4] + [1] * 3
    outs = []
    for enabled, branch in ((False, True), (True, True), (True, False)):
        ops._side.enabled, ops._side.use_branch = enabled, branch
        ops._side.branch.clear()
        try:
            torch.manual_seed(0)
            m = STiLModel(dict(model="resnet18", embedding_dim=512, field_lengths=fl, num_classes=5, start_epoch=0, batch_size=16, th1=0.3))
            m.setup_device("cuda"); m.train(); m.current_epoch = 1
            m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(5, 128, generator=torch.Generator().manual_seed(1))).cuda())
            opt = StilAdam(m.flat, lr=1e-3)
            batch = synthetic_batch(fl, 5, 16, 64, seed=3, device="cuda")
            losses = [float(train_step(m, opt, batch)) for _ in range(3)]
            torch.cuda.synchronize()
            outs.append((losses, m.flat.params.clone(), m.flat.ema.clone(), m.flat.grads.clone(), m.last["y_hat_m_e"].clone(), m.prototypes_sum.clone()))
            assert bool(ops._side.branch) == (enabled and branch), "the branch stream runs exactly when the side stream does and it is on"
        finally:
            ops._side.enabled, ops._side.use_branch = True, False
    a = outs[0]
    for b in outs[1:]:
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert torch.equal(x, y)


@pytest.mark.parametrize("label,over,B", [
    ("all_continuous", dict(field_lengths=[1] * 5), 8),
    ("all_categorical", dict(field_lengths=[3, 4, 5]), 8),
    ("single_column", dict(field_lengths=[1]), 8),
    ("img96_batch6", dict(field_lengths=[3, 1, 1], img_size=96), 6),          # B_l = 1, B_u = 5: ragged tiles everywhere
    ("img100_not_multiple_of_32", dict(field_lengths=[3, 1, 1], img_size=100), 8),
    ("three_classes_more_samples_than_tile", dict(field_lengths=[4, 1], num_classes=3), 72),  # M crosses a 64-row tile
    ("no_ema_teacher", dict(field_lengths=[3, 1, 1], use_ema=False), 8),                          # STiLModel.py:254-257: teacher = student
    ("distribution_alignment_on", dict(field_lengths=[3, 1, 1], DA=True), 8),
    ("repeat_ratio_3", dict(field_lengths=[3, 1, 1], repeat_ratio=3.0), 16),                      # trainers/evaluate.py:83 -> STiLModel.py:222-224
])
def test_edge_layouts_match_oracle(label, over, B):
    """Column layouts / image sizes / batch sizes the reference's modules accept (empty categorical or continuous part,
    one column, maps that are not multiples of the tile sizes): one full step against the CPU oracle."""
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    from oracle.make_golden import randomize_state, make_mi_masks
    base = dict(model="resnet18", embedding_dim=512, img_size=64, num_classes=5, batch_size=B, start_epoch=0, th1=0.05)
    base.update(over)
    hp = O.default_hparams(**base)
    sd = randomize_state(O.init_state(hp, seed=11), seed=12)
    if not hp.use_ema:  # the reference module then has no `ema` child (STiLModel.py:83-91)
        sd = {k: v for k, v in sd.items() if not k.startswith("ema.")}
    g = torch.Generator().manual_seed(13)
    sd["prototypes"] = torch.nn.functional.normalize(torch.randn(hp.num_classes, hp.projection_dim, generator=g))
    m = _make_model(hp, {k: v.clone() for k, v in sd.items()})
    m.current_epoch = 1
    batch = O.synthetic_batch(hp, B, seed=21)
    B_u = len(batch["u"][2])
    mr = torch.rand(B_u, generator=g).ge(0.5)
    s = hp.img_size
    for _ in range(5):
        s = (s + 1) // 2
    mm = {0: make_mi_masks(B, s * s, len(hp.field_lengths), 512, 4, 0.1, seed=5)}
    o = O.full_step(sd, {}, 1, batch, hp, 1, mr, mm)
    train_step(m, StilAdam(m.flat, lr=hp.lr_eval), _to_dev(batch), mask_random=mr, mi_masks=mm)
    torch.cuda.synchronize()
    bad = []
    for k in SCALARS + ["y_hat_m", "y_hat_i", "y_hat_t", "y_hat_m_e", "feat_m", "feat_i", "feat_t", "pseudo_label", "prediction", "class_sum", "class_count"]:
        ok, err = _close(m.last[k].detach().cpu().numpy(), o[k].numpy())
        if not ok:
            bad.append((k, err))
    _check_flags(m.last, o, B_u)
    msd = m.state_dict()
    tr = set(O.trainable_keys(sd))
    for k, v in sd.items():
        if k not in tr:
            ok, err = _close(msd[k].cpu().double().numpy(), v.double().numpy(), 5e-5)
            if not ok:
                bad.append(("state " + k, err))
    assert not bad, f"{label}: {len(bad)} mismatches, first: {bad[:8]}"


def test_frozen_encoders_are_not_updated():
    """finetune_strategy == 'frozen' (STiLModel_backbone.py:78-84): the encoders take no gradient and Adam leaves them alone,
    while their BatchNorm layers still run in training mode (batch statistics, running buffers move) as in the reference."""
    from stil_tta_amd import STiLModel
    from stil_tta_amd.driver import synthetic_batch, train_step
    from stil_tta_amd.flat import StilAdam
    fl = [3, 4] + [1] * 3
    torch.manual_seed(0)
    m = STiLModel(dict(model="resnet18", embedding_dim=512, field_lengths=fl, num_classes=5, start_epoch=0, batch_size=16, th1=0.3))
    for mod in (m.model.encoder_imaging, m.model.encoder_tabular):
        for p in mod.parameters():
            p.requires_grad = False
    m.setup_device("cuda"); m.train(); m.current_epoch = 1
    m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(5, 128, generator=torch.Generator().manual_seed(1))).cuda())
    before = {k: v.clone() for k, v in m.state_dict().items()}
    train_step(m, StilAdam(m.flat, lr=1e-2), synthetic_batch(fl, 5, 16, 64, seed=3, device="cuda"))
    torch.cuda.synchronize()
    after = m.state_dict()
    moved_head = False
    for k, v in after.items():
        if k.startswith("model.encoder_"):
            if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
                continue
            assert torch.equal(v, before[k]), f"frozen parameter {k} changed"
        elif k.startswith("model.projection_si") and k.endswith("weight"):
            moved_head = moved_head or not torch.equal(v, before[k])
    assert moved_head
    assert not torch.equal(after["model.encoder_imaging.bn1.running_mean"], before["model.encoder_imaging.bn1.running_mean"])
    assert all(not p._stil_touched for p in m.model.encoder_imaging.parameters())
