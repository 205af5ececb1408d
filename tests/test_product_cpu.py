"""CPU-side checks of the product: C-ABI library loads and exports every symbol include/stil_hip.h declares,
module tree == reference state_dict layout, host logic (sharding, schedule, synthetic data), the N>1 path over
gloo (world_size 2), and the "no silent fallback" rule."""
import ctypes
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import stil_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G
    G.build()
    from stil_tta_amd._lib import LIB_PATH, parse_header
    protos = parse_header()
    assert len(protos) >= 48
    dll = ctypes.CDLL(LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/stil_hip.h but not exported"
    dll.stil_version.restype = ctypes.c_int
    assert dll.stil_version() >= 100


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    from stil_tta_amd._lib import lib
    L = lib()
    with pytest.raises(RuntimeError, match="null pointer"):
        L.gemm_nt(None, None, None, 4, 4, 4, 4, 4, 4, 1, 1, 4, 1, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 1, None, None, None, None, None, 0, None, 0, 1.0, None, 0, None)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        L.ema_update(ctypes.c_void_p(16), ctypes.c_void_p(32), 6, 0.9, None)
    assert L.wgrad_workspace_bytes(1 << 20, 64, 576, 0) > 0
    assert L.gemm_nt_variant(50176, 1024, 0) == 21 and L.gemm_nt_variant(256, 286, 0) == 11 and L.gemm_nt_variant(802816, 64, 0) == 21 and L.gemm_nt_variant(256, 286, 22) == 22


def test_ops_refuse_cpu_tensors_no_fallback():
    from stil_tta_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.randn(4, 8), torch.randn(3, 8), None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.l2norm(torch.randn(4, 8))


def test_product_does_not_import_the_oracle():
    import subprocess
    code = "import sys; sys.path.insert(0, %r); import stil_tta_amd, stil_tta_amd.driver; " \
           "assert not any(m.startswith('oracle') for m in sys.modules), 'product imported the oracle'" % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)
    for f in os.listdir(os.path.join(ROOT, "stil_tta_amd")):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "stil_tta_amd", f)).read().replace("oracle/make_golden.py", "").replace("Oracle-format", "").replace("oracle layout", "")


@pytest.mark.parametrize("over", [dict(), dict(target="CAD", num_classes=2, field_lengths=[4] * 26 + [1] * 49),
                                  dict(model="resnet18", embedding_dim=512)])
def test_state_dict_layout_equals_reference(over):
    from stil_tta_amd import STiLModel
    hp = O.default_hparams(**over)
    sd = O.init_state(hp)
    m = STiLModel(dict(vars(hp)))
    msd = m.state_dict()
    assert list(msd.keys()) == list(sd.keys())
    for k, v in sd.items():
        assert tuple(msd[k].shape) == tuple(v.shape) and msd[k].dtype == v.dtype, k
    m.load_state_dict(sd)  # a reference-layout checkpoint loads strictly


def test_host_logic_matches_oracle():
    from stil_tta_amd.driver import anneal_lambda, shard_batch, synthetic_batch
    f = anneal_lambda(10, 500)
    for e in (0, 1, 5, 9, 10, 11, 250, 499):
        assert abs(f(e) * 1e-4 - O.anneal_lr(e, 1e-4, 10, 500)) < 1e-12
    hp = O.default_hparams()
    a = O.synthetic_batch(hp, 32, seed=2022)
    b = synthetic_batch(hp.field_lengths, hp.num_classes, 32, hp.img_size, seed=2022)
    for k in ("l", "u"):
        assert torch.equal(a[k][0][1], b[k][0][1]) and torch.equal(a[k][1][1], b[k][1][1]) and torch.equal(a[k][2], b[k][2])
    assert len(b["l"][2]) == 4 and len(b["u"][2]) == 28 and bool(b["l"][4].all()) and not bool(b["u"][4].any())
    s0, s1 = shard_batch(b, 0, 2), shard_batch(b, 1, 2)
    assert len(s0["l"][2]) == 2 and len(s0["u"][2]) == 14
    assert torch.equal(torch.cat((s0["u"][1][1], s1["u"][1][1])), b["u"][1][1])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from stil_tta_amd.driver import allreduce_flat, init_distributed, world_size
    init_distributed(backend="gloo")
    assert world_size() == world
    g = torch.Generator().manual_seed(rank)
    slab = torch.randn(5000, generator=g)
    mine = slab.clone()
    scale = allreduce_flat(slab, bucket_elems=2048)  # 3 buckets
    ref = sum(torch.randn(5000, generator=torch.Generator().manual_seed(r)) for r in range(world))
    ok = bool(torch.allclose(slab, ref, atol=1e-6)) and scale == 1.0 / world
    # fused [K, Dp+1] prototype exchange (STiLModel.py:378-379 as ONE collective)
    cs = torch.full((7, 129), float(rank + 1))
    dist.all_reduce(cs)
    ok = ok and bool((cs == sum(range(1, world + 1))).all()) and not torch.equal(mine, slab)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


# ---------------------------------------------------------------- fit-loop host logic (stil_tta_amd/fit.py)
def test_fit_host_helpers():
    from stil_tta_amd import fit as F
    assert F.split_batch_size(512, 7) == (64, 448) and F.split_batch_size(64, 7) == (8, 56)  # trainers/evaluate.py:84-85
    assert F.repeat_ratio(90000, 1000, 7) == 11 and F.repeat_ratio(100, 1000, 7) == 1       # trainers/evaluate.py:83

    class Loader:  # reshuffles on every restart, like DataLoader(shuffle=True)
        def __init__(self, n):
            self.n, self.starts = n, 0

        def __len__(self):
            return self.n

        def __iter__(self):
            self.starts += 1
            return iter([(self.starts, i) for i in range(self.n)])

    l, u = Loader(2), Loader(5)
    seen = list(F.max_size_cycle({"l": l, "u": u}))
    assert len(seen) == 5 and [b["u"][1] for b in seen] == [0, 1, 2, 3, 4]
    assert [b["l"] for b in seen] == [(1, 0), (1, 1), (2, 0), (2, 1), (3, 0)]  # the short loader is restarted, not padded

    es = F.EarlyStopping(min_delta=1e-4, patience=3)
    assert [es.should_stop(v) for v in (0.5, 0.50005, 0.4, 0.6, 0.6, 0.6, 0.60009)] == [False, False, False, False, False, False, True]
    bc = F.BestCheckpoint("eval.val.acc", "/tmp/x", "checkpoint_best_acc")
    assert bc.path == "/tmp/x/checkpoint_best_acc.ckpt"
    assert [bc.improved(v, e) for e, v in enumerate((0.1, 0.1, 0.3, 0.2))] == [True, False, True, False] and bc.best_epoch == 2


def test_checkpoint_loading_tip_and_saint(tmp_path):
    """hparams.checkpoint (TIP: encoder_imaging.* [+ encoder_tabular.* for the Transformer backbone], frozen / trainable) and
    hparams.checkpoint_SAINT (a bare SAINT state_dict) -- STiLModel_backbone.py:69-90, STiLModel_SAINT_backbone.py:68-90,144-146."""
    import torch
    from stil_tta_amd import STiLModel
    fl = [3, 4, 1, 1, 1]
    base = dict(model="resnet18", embedding_dim=512, field_lengths=fl, num_classes=5, batch_size=8)
    torch.manual_seed(1)
    src = STiLModel(dict(base))
    tip = {"state_dict": {k[len("model."):]: (v.clone() + 0.5 if v.is_floating_point() else v.clone() + 3) for k, v in src.state_dict().items()
                          if k.startswith("model.encoder_imaging.") or k.startswith("model.encoder_tabular.")}, "hyper_parameters": {}}
    tip["state_dict"]["encoder_tabular.projection_head.weight"] = torch.zeros(1)  # ignored keys (STiLModel_backbone.py:111)
    pt = tmp_path / "tip.ckpt"
    torch.save(tip, pt)
    torch.manual_seed(2)
    m = STiLModel(dict(base, checkpoint=str(pt), finetune_strategy="frozen"))
    sd = m.state_dict()
    for k, v in tip["state_dict"].items():
        if "projection_head" not in k:
            assert torch.equal(sd["model." + k], v) and torch.equal(sd["ema." + k], v), k
    assert not any(p.requires_grad for p in m.model.encoder_imaging.parameters()) and not any(p.requires_grad for p in m.model.encoder_tabular.parameters())
    assert all(p.requires_grad for p in m.model.projection_si.parameters())
    # SAINT: image encoder from TIP, tabular encoder from checkpoint_SAINT
    torch.manual_seed(3)
    s0 = STiLModel(dict(base, tabular_encoder="saint"))
    saint_sd = {k: (v.clone() + 1 if v.is_floating_point() else v.clone()) for k, v in s0.model.encoder_tabular.state_dict().items()}
    ps = tmp_path / "saint.pth"
    torch.save(saint_sd, ps)
    torch.manual_seed(4)
    s1 = STiLModel(dict(base, tabular_encoder="saint", checkpoint=str(pt), checkpoint_SAINT=str(ps)))
    got = s1.state_dict()
    for k, v in saint_sd.items():
        assert torch.equal(got["model.encoder_tabular." + k], v) and torch.equal(got["ema.encoder_tabular." + k], v), k
    assert torch.equal(got["model.encoder_imaging.conv1.weight"], tip["state_dict"]["encoder_imaging.conv1.weight"])
