"""GPU parity of the CoMatch / SimMatch baselines (stil_tta_amd/match.py) against the golden vectors recorded from the REAL
reference (models/MatchModel/{CoMatch,SimMatch}.py, oracle/make_golden_match.py): logits, pseudo-labels, confidence masks,
the pseudo-label graph and the similarity matrix (CoMatch), losses, every gradient tensor against the float64 oracle
evaluated on the device's own ReLU / max-pool decisions (as test_gpu_step.py), queues / banks / BN buffers after the step."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

from oracle import make_golden_match as GX  # noqa: E402
from oracle import match_oracle as XO  # noqa: E402
from oracle import stil_oracle as O  # noqa: E402
from test_gpu_step import _check_flips, _close, _grad_errors, _trace_decisions  # noqa: E402


def _dev(x):
    if isinstance(x, (tuple, list)):
        return type(x)(_dev(t) for t in x)
    return x.cuda()


def _decisions(m, trace, student_prefix, img_prefix):
    """ops._trace -> (relu, pool) in the oracle's tags (reference module names) and layouts."""
    names = {id(p): n for n, p in m.named_parameters()}
    relu = {}
    for pid, z in trace["relu"].items():
        tag = names[pid][: -len(".weight")]
        if not tag.startswith(student_prefix):
            continue                                   # the momentum copy's ReLUs are not gradient decisions
        mask = z.detach() > 0
        if z.ndim == 4:
            mask = mask.permute(0, 3, 1, 2)            # NHWC -> NCHW
        relu[tag] = mask.cpu().contiguous()
    pool = {}
    if "maxpool" in trace["pool"]:
        idx, H, W = trace["pool"]["maxpool"]
        t = idx.cpu().long()
        N, OH, OW, C = t.shape
        oy = torch.arange(OH).view(1, OH, 1, 1)
        ox = torch.arange(OW).view(1, 1, OW, 1)
        flat = (oy * 2 - 1 + t // 3) * W + (ox * 2 - 1 + t % 3)
        pool[img_prefix + "maxpool"] = flat.permute(0, 3, 1, 2).contiguous()
    return relu, pool


@pytest.mark.parametrize("name", list(GX.CASES))
def test_match_training_step_matches_reference_golden(name):
    import stil_tta_amd
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    kind, hp, sd, batch, epoch, aux = GX.build_case(name)
    for nm in ("co_threshold", "contrast_th", "sim_threshold"):   # data-dependent thresholds of the generating machine
        setattr(hp, nm, float(fx["meta_" + nm]))
    if kind == "freematch":
        aux["time_p"] = torch.tensor(fx["meta_time_p"])
    m = {"comatch": stil_tta_amd.CoMatch, "simmatch": stil_tta_amd.SimMatch, "freematch": stil_tta_amd.FreeMatch}[kind](dict(vars(hp)))
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m.setup_device("cuda"); m.train(); m.current_epoch = epoch
    if kind == "comatch":
        m.model.hist_prob = [t.cuda() for t in aux.get("hist_prob", [])]
    if kind == "freematch":
        m.model.p_model.copy_(aux["p_model"]); m.model.label_hist.copy_(aux["label_hist"]); m.model.time_p.copy_(aux["time_p"].reshape(1))
    dbatch = {"l": tuple(_dev(t) for t in batch["l"]), "u": (_dev(batch["u"][0]), batch["u"][1].cuda())}
    with _trace_decisions() as tr:
        train_step(m, StilAdam(m.flat, lr=hp.lr_eval), dbatch)
        torch.cuda.synchronize()
        stu = GX.STUDENT[kind]
        img = stu + ("encoder_imaging." if hp.eval_datatype == "imaging_and_tabular" else "backbone.")
        relu, pool = _decisions(m, tr, stu, img)
    bad = []
    scalars, tensors = GX.OUT[kind]
    L = dict(m.last)
    if kind == "comatch":
        L["sim"] = torch.exp(L["sim_logits"].detach())
    for k in scalars + tensors:
        ok, err = _close(L[k].detach().cpu().numpy().reshape(fx["out_" + k].shape), fx["out_" + k])
        if not ok:
            bad.append((k, err))
    assert np.array_equal(L["mask"].cpu().numpy() > 0.5, fx["out_mask"] > 0.5), "confidence mask"
    # gradients: float64 oracle on the device's decisions, every tensor <= 3 * e32 + 1e-4
    d64 = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t  # noqa: E731
    cv = lambda x: tuple(d64(t) for t in x) if isinstance(x, (tuple, list)) else d64(x)  # noqa: E731
    sd64 = {k: d64(v.clone()) for k, v in sd.items()}
    b64 = {"l": (cv(batch["l"][0]), batch["l"][1], batch["l"][2]), "u": ([cv(v) for v in batch["u"][0]], batch["u"][1])}
    with O.force_decisions(relu, pool) as dec:
        o64 = XO.full_step(kind, sd64, {}, 1, b64, hp, epoch, aux=GX.clone_aux(aux, torch.float64))
    _check_flips(dec.get("flips", {}))
    params = dict(m.named_parameters())
    gbad, ratios = _grad_errors(params, o64["grads"], lambda k: float(fx["gerr32_" + k]) if ("gerr32_" + k) in fx.files else 0.0)
    bad += gbad
    # state after the step: queues / banks in full, everything else by checksum
    msd = m.state_dict()
    for key in fx.files:
        if key.startswith("state_"):
            got, ref = msd[key[6:]].cpu().numpy(), fx[key]
            if ref.dtype == np.int64:
                assert np.array_equal(got, ref), key
            else:
                ok, err = _close(got, ref, 5e-5)
                if not ok:
                    bad.append((key, err))
        elif key.startswith("ssum_"):
            v = msd[key[5:]].double()
            if abs(float(v.sum()) - float(fx[key])) > 5e-5 * (1.0 + float(fx["sabs_" + key[5:]])):
                bad.append(("state " + key[5:], float(v.sum()), float(fx[key])))
    if kind == "comatch":
        hist = m.model.hist_prob
        assert len(hist) == int(fx["hist_len"])
        ok, err = _close(hist[-1].cpu().numpy(), fx["hist_last"])
        if not ok:
            bad.append(("hist_prob", err))
    print(f"[{name}] worst gradient error / bound: {max(ratios):.3f}")
    assert not bad, f"{len(bad)} mismatches, first: {bad[:8]}"
    # validation hook on the post-step weights of the REFERENCE (Adam noise excluded): load the oracle's post-step state
    sd_o = {k: v.clone() for k, v in sd.items()}
    XO.full_step(kind, sd_o, {}, 1, batch, hp, epoch, aux=GX.clone_aux(aux))
    m.load_state_dict(sd_o)
    m.eval()
    ok, err = _close(m.validation_step((_dev(batch["l"][0]), batch["l"][1].cuda())).cpu().numpy(), fx["out_val_loss"])
    assert ok, ("val_loss", err)


def test_contrast_graph_and_unfold_kernels_against_torch():
    """stil_contrast_graph (value + dS) and stil_simmatch_unfold against the reference formulas in float64 on random inputs,
    including a row whose only edge is its diagonal and ragged sizes."""
    from stil_tta_amd import ops
    g = torch.Generator().manual_seed(3)
    for R, N, K in ((14, 54, 5), (3, 3, 2), (37, 2597, 286)):
        S = (torch.randn(R, N, generator=g) * 3).clamp(-10, 10)
        Q = torch.rand(R, N, generator=g)
        Q[torch.arange(R), torch.arange(R)] = 1.0
        Q[0, 1:] = 0.0
        th = 0.7
        Sd = S.cuda().requires_grad_(True)
        loss = ops.ContrastGraphFn.apply(Sd, Q.cuda(), th)
        loss.backward()
        S64 = S.double().requires_grad_(True)
        sim, Q64 = torch.exp(S64), Q.double()
        pm = Q64 >= th
        Qm = Q64 * pm
        Qm = Qm / Qm.sum(1, keepdim=True)
        ref = (-(torch.log((sim * pm) / sim.sum(1, keepdim=True) + 1e-7) * pm * Qm).sum(1)).mean()
        ref.backward()
        assert abs(float(loss) - float(ref)) <= 2e-5 * (1 + abs(float(ref)))
        assert float((Sd.grad.cpu().double() - S64.grad).abs().max()) <= 2e-6 * (1 + float(S64.grad.abs().max()))
        tpo = torch.softmax(torch.randn(R, N, generator=g) * 2, dim=1)
        p = torch.softmax(torch.randn(R, K, generator=g), dim=1)
        labels = torch.randint(0, K, (N,), generator=g)
        for c in (0.9, 1.0):
            t_d, ps_d = ops.simmatch_unfold(tpo.cuda(), p.cuda(), labels.cuda(), c)
            t = tpo.double() * p.double().gather(1, labels.expand(R, -1))
            t = t / t.sum(1, keepdim=True)
            agg = torch.zeros(R, K, dtype=torch.float64).scatter_add(1, labels.expand(R, -1), tpo.double())
            ps = p.double() * c + agg * (1 - c) if c < 1 else p.double()
            assert float((t_d.cpu().double() - t).abs().max()) <= 1e-6
            assert float((ps_d.cpu().double() - ps).abs().max()) <= 1e-6


# ---------------------------------------------------------------------------------------------------------------- two ranks
def _dp_worker(rank, world, port, outdir):
    """concat_all_gather / all_reduce paths of comatch_model.py:114-118,265-270 and simmatch_model.py:137-156 on two ranks."""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      STIL_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import stil_tta_amd
    from stil_tta_amd.driver import init_distributed, train_step
    from stil_tta_amd.flat import StilAdam
    init_distributed()
    out = {}
    for kind, cls in (("comatch", stil_tta_amd.CoMatch), ("simmatch", stil_tta_amd.SimMatch)):
        hp = XO.default_hparams(**dict(GX.R18, K=40, contrast_th=0.3, co_threshold=0.3, sim_threshold=0.3))
        torch.manual_seed(rank)          # differently initialised ranks: the first step broadcasts rank 0's state
        m = cls(dict(vars(hp)))
        m.setup_device("cuda"); m.train(); m.current_epoch = 2
        batch = XO.synthetic_batch(hp, 16, seed=70 + rank, views=3 if kind == "comatch" else 2)
        if kind == "simmatch":           # disjoint bank slots per rank
            batch["l"] = (batch["l"][0], batch["l"][1], torch.tensor([3 + rank, 20 + rank]))
        dbatch = {"l": tuple(_dev(t) for t in batch["l"]), "u": (_dev(batch["u"][0]), batch["u"][1].cuda())}
        train_step(m, StilAdam(m.flat, lr=1e-3), dbatch)
        torch.cuda.synchronize()
        sd = {k: v.cpu() for k, v in m.state_dict().items() if "encoder" not in k and ".main." not in k and ".ema." not in k}
        out[kind] = dict(state=sd, params=m.flat.params.cpu(), probs=m.last["probs" if kind == "comatch" else "prob_ku_orig"].cpu(),
                         nb=(m.flat.n_backbone_params, m.flat.n_backbone_state))
        if kind == "comatch":
            out[kind]["hist"] = m.model.hist_prob[-1].cpu()
    torch.save(out, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_match_baselines_two_ranks_share_queues_and_banks():
    import tempfile
    from test_gpu_dp import _run_pair
    with tempfile.TemporaryDirectory() as td:
        r0, r1 = _run_pair(_dp_worker, td)
    for kind in ("comatch", "simmatch"):
        a, b = r0[kind], r1[kind]
        n = a["nb"][0]
        assert torch.equal(a["params"][:n], b["params"][:n]), kind + ": parameters diverged"
        for k in a["state"]:
            assert torch.equal(a["state"][k], b["state"][k]), (kind, k)       # every rank holds the gathered queue / bank
    co = r0["comatch"]["state"]
    assert int(co["model.queue_ptr_s"]) == 28 and int(co["model.queue_ptr_w"]) == 32      # 2 x 14 strong / 2 x 16 weak features enqueued
    assert float(co["model.probs_xu"][:, :32].sum(0).sub(1).abs().max()) < 1e-5 and float(co["model.probs_xu"][:, 32:].abs().max()) == 0.0
    assert torch.equal(r0["comatch"]["hist"], r1["comatch"]["hist"])                     # all-reduced batch mean of the weak probabilities
    si = r0["simmatch"]["state"]
    g = torch.Generator().manual_seed(0)
    assert float(si["model.bank"][:, [3, 4, 20, 21]].norm(dim=0).sub(1).abs().max()) < 1e-5
    assert not torch.equal(si["model.bank"][:, 3], si["model.bank"][:, 4])
    assert int(si["model.DA_ptr"]) == 1


def test_freematch_kernels_against_torch():
    """stil_freematch_update / stil_freematch_entropy against the reference formulas (freematch_model.py:132-168,
    freematch_utils.py:18-47) in float64: more rows than one workgroup has threads, K = 286, an empty mask, a full mask."""
    from stil_tta_amd import ops
    g = torch.Generator().manual_seed(5)
    for R, K in ((14, 5), (300, 286), (33, 2)):
        probs = torch.softmax(torch.randn(R, K, generator=g) * 3, dim=1)
        p_model = torch.softmax(torch.randn(K, generator=g), dim=0)
        label_hist = torch.softmax(torch.randn(K, generator=g), dim=0)
        time_p = torch.rand(1, generator=g) * 0.5 + 0.3
        pm_d, lh_d, tp_d = p_model.cuda(), label_hist.cuda(), time_p.cuda()
        mask, onehot, idx = ops.freematch_update(probs.cuda(), pm_d, lh_d, tp_d, 0.999)
        P = probs.double()
        mp_, mi_ = P.max(dim=-1)
        tp = time_p.double() * 0.999 + 0.001 * mp_.mean()
        pm = p_model.double() * 0.999 + 0.001 * P.mean(0)
        hist = torch.bincount(mi_, minlength=K).double()
        lh = label_hist.double() * 0.999 + 0.001 * hist / hist.sum()
        ref_mask = mp_ >= tp * (pm / pm.max())[mi_]
        assert torch.equal(idx.cpu().long(), mi_) and torch.equal(onehot.cpu().argmax(1), mi_) and float(onehot.sum()) == R
        assert float((pm_d.cpu().double() - pm).abs().max()) < 1e-7 and float((lh_d.cpu().double() - lh).abs().max()) < 1e-7
        assert abs(float(tp_d) - float(tp)) < 1e-7
        margin = (mp_ - tp * (pm / pm.max())[mi_]).abs() > 1e-6            # rows not within rounding of the threshold
        assert torch.equal((mask.cpu() > 0.5)[margin], ref_mask[margin])
        z = torch.randn(R, K, generator=g) * 2
        for which in ("mixed", "empty", "full"):
            m = {"mixed": (torch.rand(R, generator=g) < 0.5).float(), "empty": torch.zeros(R), "full": torch.ones(R)}[which]
            zd = z.cuda().requires_grad_(True)
            loss = ops.FreeMatchEntropyFn.apply(zd, m.cuda(), pm_d, lh_d)
            loss.backward()
            if float(m.sum()) == 0:
                assert float(loss) == 0.0 and float(zd.grad.abs().max()) == 0.0
                continue
            z64 = z.double().requires_grad_(True)
            ref = XO.freematch_entropy_loss(m, z64, pm_d.cpu().double(), lh_d.cpu().double())
            ref.backward()
            assert abs(float(loss) - float(ref)) <= 2e-5 * (1 + abs(float(ref))), (R, K, which, float(loss), float(ref))
            assert float((zd.grad.cpu().double() - z64.grad).abs().max()) <= 2e-5 * (1e-3 + float(z64.grad.abs().max())), (R, K, which)


def test_frozen_encoders_keep_their_weights_and_the_heads_still_train():
    """finetune_strategy == 'frozen' (multimodal_backbone.py:70-76): the TIP encoders receive no gradient and do not move; the
    projections / head / classifier do."""
    import stil_tta_amd
    from stil_tta_amd.driver import train_step
    from stil_tta_amd.flat import StilAdam
    hp = XO.default_hparams(**dict(GX.R18, K=40, contrast_th=0.3, co_threshold=0.3))
    torch.manual_seed(0)
    m = stil_tta_amd.CoMatch(dict(vars(hp)))
    for enc in (m.model.encoder.encoder_imaging, m.model.encoder.encoder_tabular):
        for p in enc.parameters():
            p.requires_grad = False
    m.setup_device("cuda"); m.train(); m.current_epoch = 2
    before = {k: v.clone() for k, v in m.state_dict().items() if k.startswith("model.encoder.")}
    batch = XO.synthetic_batch(hp, 16, seed=5, views=3)
    dbatch = {"l": tuple(_dev(t) for t in batch["l"]), "u": (_dev(batch["u"][0]), batch["u"][1].cuda())}
    loss = train_step(m, StilAdam(m.flat, lr=1e-2), dbatch)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss))
    after = m.state_dict()
    moved = {k for k, v in before.items() if v.is_floating_point() and not torch.equal(v, after[k])}
    assert not any(("encoder_imaging" in k or "encoder_tabular" in k) and not ("running_" in k) for k in moved), sorted(moved)[:5]
    assert {"model.encoder.head.2.weight", "model.encoder.classifier_multimodal.weight", "model.encoder.image_proj.weight"} <= moved
