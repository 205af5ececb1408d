"""CPU oracle for the STiL semi-supervised training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``stil_tta_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

It is a plain-PyTorch fp32 restatement (functional, operating on a flat
``state_dict``-style mapping whose keys equal the reference's) of

    models/Disentangle/STiLModel.py:228-386        training_step
    models/Disentangle/utils/STiLModel_backbone.py  DisCoAttentionBackbone
    models/resnets.py                               ResNet / Bottleneck
    models/Transformer.py                           TabularTransformerEncoder
    models/Disentangle/utils/disentangle_transformer.py  MITransformerLayer
    models/Disentangle/utils/club.py                CLUBMean
    utils/clip_loss.py, utils/prototype_loss.py

Pinning: ``oracle/make_golden.py`` imports the real reference (with stubs for
absent non-arithmetic third-party packages) in the build container, checks this
restatement against it and writes ``tests/golden/*.npz``.  The reference has no
tests or golden vectors of its own, so those generated
fixtures are the pin.  One third-party piece of arithmetic is *unpinned*:
``lightly==1.2.22``'s ``SimCLRProjectionHead`` (absent offline) is restated as
Linear -> ReLU -> Linear.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

RESNET_LAYERS = {"resnet50": ("bottleneck", [3, 4, 6, 3]), "resnet18": ("basic", [2, 2, 2, 2])}


# --------------------------------------------------------------------------
# hyper-parameters (configs/config_dvm_STiL.yaml + configs/models/resnet50.yaml)
# --------------------------------------------------------------------------
def default_hparams(**over) -> SimpleNamespace:
    hp = dict(
        model="resnet50", embedding_dim=2048, img_size=128,
        num_classes=286, target="dvm",
        field_lengths=[10, 20, 30, 40] + [1] * 13,  # list version of field_lengths_tabular (.pt)
        tabular_embedding_dim=512, tabular_transformer_num_layers=4,
        embedding_dropout=0.0, drop_rate=0.0,
        multimodal_embedding_dim=512, multimodal_transformer_num_layers=1,
        projection_dim=128, temperature=0.1, lambda_0=0.5,
        alpha=0.2, beta=3.0, gamma=0.5, rate_pt=1.0, rate_uce=0.2,
        th1=0.9, th2=0.95, th_contrast=0.8, start_epoch=35, rate_pseudo=0.9,
        use_ema=True, eman=True, ema_momentum=0.996, DA=False,
        repeat_ratio=1.0, batch_size=32, lr_eval=1e-4, weight_decay_eval=0.0,
        mi_drop=0.1,  # attn_drop = proj_drop = drop_path of MITransformerLayer (backbone.py:60)
        tabular_encoder="transformer",  # "saint" = STiLModel_SAINT.py / STiLModel_SAINT_backbone.py variant
        saint_ff_drop=0.8,  # ff_dropout of the SAINT RowColTransformer (STiLModel_SAINT_backbone.py:120-122)
        scheduler="anneal", warmup_epochs=10, max_epochs=500,
    )
    hp.update(over)
    return SimpleNamespace(**hp)


def split_field_lengths(field_lengths: List[int]):
    """STiLModel_backbone.py:96-105: cardinality 1 == continuous column."""
    cat = [int(x) for x in field_lengths if int(x) != 1]
    con = [int(x) for x in field_lengths if int(x) == 1]
    return cat, con


# --------------------------------------------------------------------------
# parameter construction (same names/shapes/init as the reference; Appendix A)
# --------------------------------------------------------------------------
def _linear_init(sd, name, out_f, in_f, gen, bias=True):
    # torch.nn.Linear default init (kaiming_uniform a=sqrt(5))
    bound = 1.0 / math.sqrt(in_f)
    sd[name + ".weight"] = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    if bias:
        sd[name + ".bias"] = (torch.rand(out_f, generator=gen) * 2 - 1) * bound


def _bn_init(sd, name, c):
    sd[name + ".weight"] = torch.ones(c)
    sd[name + ".bias"] = torch.zeros(c)
    sd[name + ".running_mean"] = torch.zeros(c)
    sd[name + ".running_var"] = torch.ones(c)
    sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _conv_init(sd, name, cout, cin, k, gen):
    # kaiming_normal_(fan_out, relu)  models/resnets.py:190-192
    std = math.sqrt(2.0 / (cout * k * k))
    sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=gen) * std


def resnet_spec(model: str):
    """Yields the conv/bn structure of models/resnets.py for `model`."""
    kind, layers = RESNET_LAYERS[model]
    return kind, layers


SAINT_DIM = 32      # STiLModel_SAINT_backbone.py:110-121: colrow attention -> depth 1, 4 heads, embedding 32
SAINT_HEADS = 4


def init_saint_state(hp, gen) -> Dict[str, Tensor]:
    """SAINT(...) state_dict in the reference's key order (SAINT/Tabular_Encoder.py:24-146, model_util.py:90-110).
    Only embeds, pos_encodings, simple_MLP and transformer.layers are used by STiL's forward; the rest is carried
    (EMA'd, never trained) exactly like the reference does."""
    cat, con = split_field_lengths(hp.field_lengths)
    ncat, ncon, d = len(cat), len(con), SAINT_DIM
    nfeats = ncat + ncon + 1
    total_tokens = sum(cat) + 1
    sd: Dict[str, Tensor] = {}
    sd["categories_offset"] = torch.tensor([0, 1] + cat).cumsum(0)[:-1]
    sd["cat_mask_offset"] = torch.tensor([0, 2] + [2] * ncat).cumsum(0)[:-1]
    sd["con_mask_offset"] = F.pad(torch.full((ncon,), 2.0).to(torch.int8), (1, 0), value=0).cumsum(0)[:-1]
    sd["norm.weight"] = torch.ones(ncon); sd["norm.bias"] = torch.zeros(ncon)

    def emb(name, n, dim):
        sd[name + ".weight"] = torch.randn(n, dim, generator=gen)

    def smlp(name, dims):
        _linear_init(sd, name + ".layers.0", dims[1], dims[0], gen)
        _linear_init(sd, name + ".layers.2", dims[2], dims[1], gen)

    for i in range(ncon):
        smlp(f"simple_MLP.{i}", [1, 100, d])
    emb("transformer.embeds", total_tokens, d)
    for li, (dim, dh) in enumerate(((d, 16), (d, None), (d * nfeats, 64), (d * nfeats, None))):
        q = f"transformer.layers.0.{li}."
        sd[q + "norm.weight"] = torch.ones(dim); sd[q + "norm.bias"] = torch.zeros(dim)
        if dh is not None:
            _linear_init(sd, q + "fn.fn.to_qkv", 3 * SAINT_HEADS * dh, dim, gen, bias=False)
            _linear_init(sd, q + "fn.fn.to_out", dim, SAINT_HEADS * dh, gen)
        else:
            _linear_init(sd, q + "fn.fn.net.0", dim * 8, dim, gen)
            _linear_init(sd, q + "fn.fn.net.3", dim, dim * 4, gen)
    emb("transformer.mask_embed", nfeats, d)
    input_size = d * ncat + d * ncon
    l = input_size // 8
    dims = [input_size, l * 4, l * 2, 1]
    for i in range(3):
        _linear_init(sd, f"mlp.mlp.{i}", dims[i + 1], dims[i], gen)
    emb("embeds", total_tokens, d)
    emb("mask_embeds_cat", ncat * 2 + 2, d)
    emb("mask_embeds_cont", ncon * 2, d)
    emb("single_mask", 2, d)
    emb("pos_encodings", ncat + ncon, d)
    for i in range(ncat):
        smlp(f"mlp1.layers.{i}", [d, 5 * d, cat[i]])
    for i in range(ncon):
        smlp(f"mlp2.layers.{i}", [d, 5 * d, 1])
    smlp("mlpfory", [d, 1000, hp.num_classes])
    smlp("pt_mlp", [d * nfeats, 6 * d * nfeats // 5, d * nfeats // 2])
    smlp("pt_mlp2", [d * nfeats, 6 * d * nfeats // 5, d * nfeats // 2])
    return sd


def _saint_attention(x, wqkv, wo, bo, heads):
    """SAINT/model_util.py:79-87 (the constructed dropout is never applied)."""
    b, n, _ = x.shape
    q, k, v = F.linear(x, wqkv).chunk(3, dim=-1)
    dh = q.shape[-1] // heads
    q, k, v = (t.reshape(b, n, heads, dh).permute(0, 2, 1, 3) for t in (q, k, v))
    attn = ((q @ k.transpose(-2, -1)) * dh ** -0.5).softmax(dim=-1)
    out = (attn @ v).permute(0, 2, 1, 3).reshape(b, n, heads * dh)
    return F.linear(out, wo, bo)


def saint_tabular_forward(sd, pb, x_t, hp, masks=None):
    """DisCoAttentionBackbone.forward_tabular (STiLModel_SAINT_backbone.py:159-184) + RowColTransformer.forward
    'colrow' (SAINT/model_util.py:111-122).  pb = backbone prefix ("model." / "ema.").
    masks (train-mode FF dropout p = 0.8, injected): {"ff_col": [B,nfeats,4*d], "ff_row": [1,B,4*d*nfeats]} or None."""
    p = pb + "encoder_tabular."
    cat_cols = [i for i, c in enumerate(hp.field_lengths) if int(c) != 1]
    con_cols = [i for i, c in enumerate(hp.field_lengths) if int(c) == 1]
    B = x_t.shape[0]
    cls = sd[pb + "cls_token"].expand(B, -1)
    x_categ = torch.cat((cls, x_t[:, cat_cols]), dim=1).long() + sd[p + "categories_offset"]
    x_categ_enc = F.embedding(x_categ, sd[p + "embeds.weight"])
    conts = []
    for j, col in enumerate(con_cols):
        h = F.relu(F.linear(x_t[:, col].reshape(B, 1), sd[p + f"simple_MLP.{j}.layers.0.weight"], sd[p + f"simple_MLP.{j}.layers.0.bias"]))
        conts.append(F.linear(h, sd[p + f"simple_MLP.{j}.layers.2.weight"], sd[p + f"simple_MLP.{j}.layers.2.bias"]))
    x_categ_enc = x_categ_enc + sd[p + "pos_encodings.weight"][: x_categ.shape[1]].unsqueeze(0)
    x = torch.cat([x_categ_enc] + ([torch.stack(conts, dim=1)] if conts else []), dim=1)  # [B, nfeats, d]
    n = x.shape[1]
    q = p + "transformer.layers.0."
    pdrop = hp.saint_ff_drop

    def prenorm_res(x, li, fn):  # PreNorm(dim, Residual(fn)): fn(norm(x)) + norm(x)
        xn = F.layer_norm(x, (x.shape[-1],), sd[q + f"{li}.norm.weight"], sd[q + f"{li}.norm.bias"], eps=1e-5)
        return fn(xn) + xn

    def ff(li, mask):
        def f(xn):
            h = F.linear(xn, sd[q + f"{li}.fn.fn.net.0.weight"], sd[q + f"{li}.fn.fn.net.0.bias"])
            a, gates = h.chunk(2, dim=-1)
            h = _drop(a * F.gelu(gates), mask, pdrop)
            return F.linear(h, sd[q + f"{li}.fn.fn.net.3.weight"], sd[q + f"{li}.fn.fn.net.3.bias"])
        return f

    mk = (lambda k: None) if masks is None else (lambda k: masks.get(k))
    x = prenorm_res(x, 0, lambda xn: _saint_attention(xn, sd[q + "0.fn.fn.to_qkv.weight"], sd[q + "0.fn.fn.to_out.weight"], sd[q + "0.fn.fn.to_out.bias"], SAINT_HEADS))
    x = prenorm_res(x, 1, ff(1, mk("ff_col")))
    x = x.reshape(1, B, n * SAINT_DIM)  # row (inter-sample) attention over the batch
    x = prenorm_res(x, 2, lambda xn: _saint_attention(xn, sd[q + "2.fn.fn.to_qkv.weight"], sd[q + "2.fn.fn.to_out.weight"], sd[q + "2.fn.fn.to_out.bias"], SAINT_HEADS))
    x = prenorm_res(x, 3, ff(3, mk("ff_row")))
    return x.reshape(B, n, SAINT_DIM)


def init_backbone_state(hp, gen) -> Dict[str, Tensor]:
    """DisCoAttentionBackbone parameters + buffers (STiLModel_backbone.py:45-68)."""
    sd: Dict[str, Tensor] = {}
    kind, layers = resnet_spec(hp.model)
    exp = 4 if kind == "bottleneck" else 1
    p = "encoder_imaging."
    _conv_init(sd, p + "conv1", 64, 3, 7, gen)
    _bn_init(sd, p + "bn1", 64)
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], layers), start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 1) else 1
            q = f"{p}layer{li}.{bi}."
            if kind == "bottleneck":
                _conv_init(sd, q + "conv1", planes, inplanes, 1, gen); _bn_init(sd, q + "bn1", planes)
                _conv_init(sd, q + "conv2", planes, planes, 3, gen); _bn_init(sd, q + "bn2", planes)
                _conv_init(sd, q + "conv3", planes * 4, planes, 1, gen); _bn_init(sd, q + "bn3", planes * 4)
            else:
                _conv_init(sd, q + "conv1", planes, inplanes, 3, gen); _bn_init(sd, q + "bn1", planes)
                _conv_init(sd, q + "conv2", planes, planes, 3, gen); _bn_init(sd, q + "bn2", planes)
            if bi == 0 and (stride != 1 or inplanes != planes * exp):
                _conv_init(sd, q + "downsample.0", planes * exp, inplanes, 1, gen)
                _bn_init(sd, q + "downsample.1", planes * exp)
            inplanes = planes * exp
    # tabular encoder
    cat, con = split_field_lengths(hp.field_lengths)
    p = "encoder_tabular."
    if getattr(hp, "tabular_encoder", "transformer") == "saint":
        # backbone-level nn.Parameter cls_token (STiLModel_SAINT_backbone.py:138) precedes the submodules in state_dict order
        sd = {"cls_token": torch.zeros(1, 1), **sd}
        sd.update({p + k: v for k, v in init_saint_state(hp, gen).items()})
        D = SAINT_DIM
    else:  # models/Transformer.py:193-238
        D = hp.tabular_embedding_dim
        sd[p + "cls_token"] = torch.nn.init.trunc_normal_(torch.zeros(1, 1, D), std=0.02, generator=gen)
        sd[p + "mask_special_token"] = torch.nn.init.trunc_normal_(torch.zeros(1, 1, D), std=0.02, generator=gen)
        sd[p + "cat_embedding.weight"] = torch.randn(max(sum(cat), 1) if cat else 0, D, generator=gen) * 0.02
        sd[p + "con_proj.weight"] = torch.randn(D, 1, generator=gen) * 0.02
        sd[p + "con_proj.bias"] = torch.zeros(D)
        sd[p + "column_embedding.weight"] = torch.randn(len(cat) + len(con) + 1, D, generator=gen) * 0.02
        sd[p + "norm.weight"] = torch.ones(D); sd[p + "norm.bias"] = torch.zeros(D)
        for i in range(hp.tabular_transformer_num_layers):
            q = f"{p}transformer_blocks.{i}."
            sd[q + "norm1.weight"] = torch.ones(D); sd[q + "norm1.bias"] = torch.zeros(D)
            sd[q + "attn.qkv.weight"] = torch.randn(3 * D, D, generator=gen) * 0.02
            sd[q + "attn.proj.weight"] = torch.randn(D, D, generator=gen) * 0.02
            sd[q + "attn.proj.bias"] = torch.zeros(D)
            sd[q + "norm2.weight"] = torch.ones(D); sd[q + "norm2.bias"] = torch.zeros(D)
            sd[q + "mlp.fc1.weight"] = torch.randn(4 * D, D, generator=gen) * 0.02
            sd[q + "mlp.fc1.bias"] = torch.zeros(4 * D)
            sd[q + "mlp.fc2.weight"] = torch.randn(D, 4 * D, generator=gen) * 0.02
            sd[q + "mlp.fc2.bias"] = torch.zeros(D)
    C = hp.multimodal_embedding_dim
    for nm, din, dh in (("projection_si", hp.embedding_dim, C), ("projection_ai", hp.embedding_dim, C),
                        ("projection_st", D, D), ("projection_at", D, D)):
        _linear_init(sd, f"{nm}.model.0", dh, din, gen)
        _linear_init(sd, f"{nm}.model.2", C, dh, gen)
    _linear_init(sd, "reduce", C, 2 * C, gen)
    for i in range(hp.multimodal_transformer_num_layers):
        q = f"transformer.{i}."
        sd[q + "norm1.weight"] = torch.ones(C); sd[q + "norm1.bias"] = torch.zeros(C)
        _linear_init(sd, q + "attn.qkv", 3 * C, C, gen)
        _linear_init(sd, q + "attn.proj", C, C, gen)
        sd[q + "norm2.weight"] = torch.ones(C); sd[q + "norm2.bias"] = torch.zeros(C)
        _linear_init(sd, q + "mlp.fc1", C, C, gen)
        _linear_init(sd, q + "mlp.fc2", C, C, gen)
    _linear_init(sd, "classifier_multimodal", hp.num_classes, 3 * C, gen)
    _linear_init(sd, "classifier_imaging", hp.num_classes, 2 * C, gen)
    _linear_init(sd, "classifier_tabular", hp.num_classes, 2 * C, gen)
    return sd


def init_state(hp, seed: int = 0) -> Dict[str, Tensor]:
    """Full STiLModel state_dict layout."""
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    K, Dp, C = hp.num_classes, hp.projection_dim, hp.multimodal_embedding_dim
    sd["prototypes"] = torch.zeros(K, Dp)
    sd["prototypes_sum"] = torch.zeros(K, Dp)
    sd["prototypes_count_sum"] = torch.zeros(K, 1)
    if hp.DA:  # STiLModel.py:100-103
        sd["DA_queue"] = torch.zeros(256, K)
        sd["DA_ptr"] = torch.zeros(1, dtype=torch.long)
    bb = init_backbone_state(hp, gen)
    for k, v in bb.items():
        sd["model." + k] = v
    # SimCLRProjectionHead(in, hidden, out) restated as Linear-ReLU-Linear under `.layers` (unpinned, see header)
    _linear_init(sd, "projector_multimodal.layers.0", 3 * C, 3 * C, gen)
    _linear_init(sd, "projector_multimodal.layers.2", Dp, 3 * C, gen)
    if hp.target == "dvm":
        _linear_init(sd, "projector_imaging", Dp, C, gen)
        _linear_init(sd, "projector_tabular", Dp, C, gen)
    else:
        for nm in ("projector_imaging", "projector_tabular"):
            _linear_init(sd, nm + ".layers.0", C, C, gen)
            _linear_init(sd, nm + ".layers.2", Dp, C, gen)
    for nm in ("CLUB_imaging", "CLUB_tabular"):
        _linear_init(sd, nm + ".p_mu.0", 512, C, gen)
        _linear_init(sd, nm + ".p_mu.2", C, 512, gen)
    # EMA teacher = copy of the student (STiLModel.py:88-91)
    for k, v in bb.items():
        sd["ema." + k] = v.clone()
    return sd


def trainable_keys(sd: Dict[str, Tensor]) -> List[str]:
    """Parameters Adam updates (STiLModel.py:563-570): everything except ema.*, buffers."""
    out = []
    for k, v in sd.items():
        if k.startswith("ema.") or k.startswith("prototypes") or k.startswith("DA_"):
            continue
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            continue
        if not v.is_floating_point():  # SAINT's int64 *_offset buffers
            continue
        out.append(k)
    return out


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def _bn(sd, name, x, train: bool):
    """nn.BatchNorm2d, eps 1e-5, momentum 0.1; updates running stats in place in train mode."""
    if train:
        sd[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=train, momentum=0.1, eps=1e-5)


# Piecewise-linear decisions of the STUDENT pass (ReLU signs, max-pool winners).  Two fp32 evaluations of the same
# network can land on different sides of a kink when a pre-activation sits within rounding of 0, and every gradient
# upstream of that unit then differs by O(1) of its contribution although both runs are right.  The gradient parity
# tests therefore run the float64 oracle ON THE DEVICE'S DECISIONS: `record_decisions()` collects the oracle's own,
# `force_decisions(d)` replays the ones exported by the HIP path (tags = reference module names; missing tags fall
# back to the oracle's own decision).
_DEC: Optional[dict] = None


class record_decisions:
    def __enter__(self):
        global _DEC
        self.prev, _DEC = _DEC, {"mode": "record", "relu": {}, "pool": {}}
        return _DEC

    def __exit__(self, *a):
        global _DEC
        _DEC = self.prev


class force_decisions(record_decisions):
    def __init__(self, relu: Dict[str, Tensor], pool: Dict[str, Tensor]):
        self.d = {"mode": "force", "relu": relu, "pool": pool}

    def __enter__(self):
        global _DEC
        self.prev, _DEC = _DEC, self.d
        return _DEC


def _relu(x: Tensor, tag: str) -> Tensor:
    if _DEC is None or not tag.startswith("model."):
        return F.relu(x)
    if _DEC["mode"] == "force" and tag in _DEC["relu"]:
        m = _DEC["relu"][tag]
        assert m.shape == x.shape, (tag, m.shape, x.shape)
        dis = (x.detach() > 0) != m  # units on which this evaluation would have decided differently
        _DEC.setdefault("flips", {})[tag] = (int(dis.sum()), float(x.detach()[dis].abs().max()) if bool(dis.any()) else 0.0,
                                              float(x.detach().abs().max()))
        return x * m.to(x.dtype)
    if _DEC["mode"] == "record":
        _DEC["relu"][tag] = (x.detach() > 0)
    return F.relu(x)


def _max_pool(x: Tensor, tag: str) -> Tensor:
    """nn.MaxPool2d(3, 2, 1) (models/resnets.py:252); decisions = flat input index iy*W+ix of every window's winner."""
    if _DEC is None or not tag.startswith("model."):
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    if _DEC["mode"] == "force" and tag in _DEC["pool"]:
        idx = _DEC["pool"][tag]
        y = x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
        own = F.max_pool2d(x.detach(), kernel_size=3, stride=2, padding=1)
        _DEC.setdefault("flips", {})[tag] = (int((own != y.detach()).sum()), float((own - y.detach()).abs().max()), float(own.abs().max()))
        return y
    y, idx = F.max_pool2d(x, kernel_size=3, stride=2, padding=1, return_indices=True)
    if _DEC["mode"] == "record":
        _DEC["pool"][tag] = idx
    return y


def resnet_forward(sd, p, x, model: str, train: bool) -> Tensor:
    """models/resnets.py:248-260 with return_all_feature_maps=True; returns the last map."""
    kind, layers = resnet_spec(model)
    x = F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)
    x = _relu(_bn(sd, p + "bn1", x, train), p + "bn1")
    x = _max_pool(x, p + "maxpool")
    for li, nblk in enumerate(layers, start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 1) else 1
            q = f"{p}layer{li}.{bi}."
            identity = x
            if kind == "bottleneck":  # models/resnets.py:112-132
                out = _relu(_bn(sd, q + "bn1", F.conv2d(x, sd[q + "conv1.weight"]), train), q + "bn1")
                out = _relu(_bn(sd, q + "bn2", F.conv2d(out, sd[q + "conv2.weight"], stride=stride, padding=1), train), q + "bn2")
                out = _bn(sd, q + "bn3", F.conv2d(out, sd[q + "conv3.weight"]), train)
                last = q + "bn3"
            else:  # models/resnets.py:71-88
                out = _relu(_bn(sd, q + "bn1", F.conv2d(x, sd[q + "conv1.weight"], stride=stride, padding=1), train), q + "bn1")
                out = _bn(sd, q + "bn2", F.conv2d(out, sd[q + "conv2.weight"], padding=1), train)
                last = q + "bn2"
            if (q + "downsample.0.weight") in sd:
                identity = _bn(sd, q + "downsample.1", F.conv2d(x, sd[q + "downsample.0.weight"], stride=stride), train)
            x = _relu(out + identity, last)  # the block's closing ReLU is tagged by its last BatchNorm
    return x


def _attention(x, qkv_w, qkv_b, proj_w, proj_b, heads):
    """models/Transformer.py:63-88 (mask=None, dropouts p=0)."""
    B, N, C = x.shape
    qkv = F.linear(x, qkv_w, qkv_b).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * ((C // heads) ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, proj_w, proj_b)


def tabular_forward(sd, p, x, hp) -> Tensor:
    """models/Transformer.py:240-278 TabularTransformerEncoder (mask=None, mask_special=None)."""
    cat, con = split_field_lengths(hp.field_lengths)
    ncat = len(cat)
    D = hp.tabular_embedding_dim
    parts = []
    if ncat:
        offs = torch.tensor([0] + cat[:-1]).cumsum(0)
        parts.append(F.embedding(x[:, :ncat].long() + offs, sd[p + "cat_embedding.weight"]))
    if len(con):
        parts.append(F.linear(x[:, ncat:].unsqueeze(-1), sd[p + "con_proj.weight"], sd[p + "con_proj.bias"]))
    h = torch.cat(parts, dim=1)
    h = torch.cat([sd[p + "cls_token"].expand(h.shape[0], -1, -1), h], dim=1)
    h = h + sd[p + "column_embedding.weight"].unsqueeze(0)
    h = F.layer_norm(h, (D,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-5)
    for i in range(hp.tabular_transformer_num_layers):  # Block, pre-LN (Transformer.py:165-174)
        q = f"{p}transformer_blocks.{i}."
        a = F.layer_norm(h, (D,), sd[q + "norm1.weight"], sd[q + "norm1.bias"], eps=1e-5)
        h = h + _attention(a, sd[q + "attn.qkv.weight"], None, sd[q + "attn.proj.weight"], sd[q + "attn.proj.bias"], 8)
        m = F.layer_norm(h, (D,), sd[q + "norm2.weight"], sd[q + "norm2.bias"], eps=1e-5)
        m = F.gelu(F.linear(m, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]))
        h = h + F.linear(m, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"])
    return h


def _mlp2(sd, p, x):
    """STiLModel_backbone.py:19-32  Linear -> ReLU -> Linear."""
    return F.linear(_relu(F.linear(x, sd[p + "model.0.weight"], sd[p + "model.0.bias"]), p + "model.0"),
                    sd[p + "model.2.weight"], sd[p + "model.2.bias"])


def _drop(x, keep: Optional[Tensor], p: float):
    """nn.Dropout in train mode with an injected keep-mask (1=keep); None -> identity (p = 0)."""
    if keep is None:
        return x
    return x * keep.to(x.dtype) / (1.0 - p)


def _drop_path(x, keep: Optional[Tensor], p: float):
    """disentangle_transformer.py:108-123: x.div(keep_prob) * per-sample mask."""
    if keep is None:
        return x
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    return x.div(1.0 - p) * keep.to(x.dtype).reshape(shape)


def mi_layer_forward(sd, q, xi, xt, xc, masks: Optional[Dict[str, Tensor]], p: float):
    """MITransformerLayer.forward (disentangle_transformer.py:151-169), 4 heads, qkv bias.

    `masks` (train-mode stochasticity, injected): keys attn_{i,t,c} [B,H,Nq,Nk],
    proj_{i,t,c} [B,N,C], dp1_{i,t,c} [B], fc1_{i,t,c} / fc2_{i,t,c} [B,N,C], dp2_{i,t,c} [B].
    None -> all dropouts are identity (p = 0).
    """
    C = xi.shape[-1]
    H = 4
    d = C // H
    mk = (lambda k: None) if masks is None else (lambda k: masks.get(k))

    def ln1(x):
        return F.layer_norm(x, (C,), sd[q + "norm1.weight"], sd[q + "norm1.bias"], eps=1e-5)

    def ln2(x):
        return F.layer_norm(x, (C,), sd[q + "norm2.weight"], sd[q + "norm2.bias"], eps=1e-5)

    def qkv(x):
        B, N, _ = x.shape
        t = F.linear(x, sd[q + "attn.qkv.weight"], sd[q + "attn.qkv.bias"]).reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
        return t[0], t[1], t[2]

    qi, ki, vi = qkv(ln1(xi))
    qt, kt, vt = qkv(ln1(xt))
    qc, kc, vc = qkv(ln1(xc))
    scale = d ** -0.5
    ai = ((qi @ ki.transpose(-2, -1)) * scale).softmax(-1)
    at = ((qt @ kt.transpose(-2, -1)) * scale).softmax(-1)
    ac = ((qc @ torch.cat((kc, ki, kt), dim=2).transpose(-2, -1)) * scale).softmax(-1)
    ai, at, ac = _drop(ai, mk("attn_i"), p), _drop(at, mk("attn_t"), p), _drop(ac, mk("attn_c"), p)
    B = xi.shape[0]
    oi = (ai @ vi).transpose(1, 2).reshape(B, -1, C)
    ot = (at @ vt).transpose(1, 2).reshape(B, -1, C)
    oc = (ac @ torch.cat((vc, vi, vt), dim=2)).transpose(1, 2).reshape(B, -1, C)

    def proj(x, key):
        return _drop(F.linear(x, sd[q + "attn.proj.weight"], sd[q + "attn.proj.bias"]), mk(key), p)

    xi = xi + _drop_path(proj(oi, "proj_i"), mk("dp1_i"), p)
    xt = xt + _drop_path(proj(ot, "proj_t"), mk("dp1_t"), p)
    xc = xc + _drop_path(proj(oc, "proj_c"), mk("dp1_c"), p)

    def mlp(x, s):
        h = F.gelu(F.linear(x, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]))
        h = _drop(h, mk("fc1_" + s), p)
        h = F.linear(h, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"])
        return _drop(h, mk("fc2_" + s), p)

    xi = xi + _drop_path(mlp(ln2(xi), "i"), mk("dp2_i"), p)
    xt = xt + _drop_path(mlp(ln2(xt), "t"), mk("dp2_t"), p)
    xc = xc + _drop_path(mlp(ln2(xc), "c"), mk("dp2_c"), p)
    return xi, xt, xc


def backbone_forward_all(sd, p, x_img, x_tab, hp, train: bool, masks=None):
    """DisCoAttentionBackbone.forward_all (STiLModel_backbone.py:150-156)."""
    f = resnet_forward(sd, p + "encoder_imaging.", x_img, hp.model, train)  # [B,C,H,W]
    B, Cc, Hh, Ww = f.shape
    x_i = f.reshape(B, Cc, Hh * Ww).permute(0, 2, 1)
    if getattr(hp, "tabular_encoder", "transformer") == "saint":
        sm = None if (masks is None or not train) else masks.get("saint")
        x_t = saint_tabular_forward(sd, p, x_tab, hp, sm)
    else:
        x_t = tabular_forward(sd, p + "encoder_tabular.", x_tab, hp)
    x_si = _mlp2(sd, p + "projection_si.", x_i)
    x_ai = _mlp2(sd, p + "projection_ai.", x_i.mean(dim=1))
    x_st = _mlp2(sd, p + "projection_st.", x_t[:, 1:, :])
    x_at = _mlp2(sd, p + "projection_at.", x_t[:, 0, :])
    x_c = F.linear(torch.cat([x_ai, x_at], dim=1), sd[p + "reduce.weight"], sd[p + "reduce.bias"]).unsqueeze(1)
    e_si, e_st, e_c = x_si, x_st, x_c
    for i in range(hp.multimodal_transformer_num_layers):
        mk = None if masks is None else masks.get(i)
        e_si, e_st, e_c = mi_layer_forward(sd, f"{p}transformer.{i}.", e_si, e_st, e_c, mk if train else None, hp.mi_drop)
    e_si, e_st, e_c = e_si.mean(1), e_st.mean(1), e_c.mean(1)
    out_m = F.linear(torch.cat([e_si, e_c, e_st], 1), sd[p + "classifier_multimodal.weight"], sd[p + "classifier_multimodal.bias"])
    out_i = F.linear(torch.cat([e_si, x_ai], 1), sd[p + "classifier_imaging.weight"], sd[p + "classifier_imaging.bias"])
    out_t = F.linear(torch.cat([e_st, x_at], 1), sd[p + "classifier_tabular.weight"], sd[p + "classifier_tabular.bias"])
    return out_m, out_i, out_t, e_si, x_si.mean(1), x_ai, e_st, x_st.mean(1), x_at, e_c


def _head(sd, p, x, student: bool = True):
    """projection heads (STiLModel.py:56-63): nn.Linear or SimCLR head (Linear-ReLU-Linear).
    student=False: the teacher's pass through the same head (no gradient, its ReLU is not a tracked decision)."""
    if (p + "weight") in sd:
        return F.linear(x, sd[p + "weight"], sd[p + "bias"])
    h = _relu(F.linear(x, sd[p + "layers.0.weight"], sd[p + "layers.0.bias"]), ("model.:" if student else "teacher.:") + p + "layers.0")
    return F.linear(h, sd[p + "layers.2.weight"], sd[p + "layers.2.bias"])


def clip_loss(f0, f1, T, lam0):
    """utils/clip_loss.py:27-39."""
    f0 = F.normalize(f0, dim=1)
    f1 = F.normalize(f1, dim=1)
    logits = f0 @ f1.t() / T
    labels = torch.arange(len(f0))
    return lam0 * F.cross_entropy(logits, labels) + (1 - lam0) * F.cross_entropy(logits.t(), labels), logits


def club_forward(sd, p, x, y):
    """CLUBMean.forward (club.py:107-121) + learning_loss (club.py:125-130)."""
    mu = F.linear(_relu(F.linear(x, sd[p + "p_mu.0.weight"], sd[p + "p_mu.0.bias"]), "model.:" + p + "p_mu.0"), sd[p + "p_mu.2.weight"], sd[p + "p_mu.2.bias"])
    positive = (-(mu - y) ** 2 / 2.0).sum(-1)
    negative = (-((y.unsqueeze(0) - mu.unsqueeze(1)) ** 2).mean(dim=1) / 2.0).sum(-1)
    club = (positive - negative).mean()
    est = -((-(mu - y) ** 2).sum(dim=1).mean(dim=0))
    return club, est


def prototype_loss(label, prototypes, feat, T, th):
    """utils/prototype_loss.py:24-40."""
    sim = torch.softmax(feat @ prototypes.t() / T, dim=1)
    log_sim = torch.log(sim + 1e-7)
    max_prob, max_id = torch.max(label, dim=1)
    conf = max_prob.ge(th)
    hard = torch.zeros_like(label)
    hard[torch.arange(len(label)), max_id] = 1
    loss = -torch.sum(log_sim * hard, dim=1)
    return (loss * conf).mean()


def cal_prototypes(label, feat, th):
    """STiLModel.py:199-214."""
    max_prob, max_id = torch.max(label, dim=1)
    conf = max_prob.ge(th)
    hard = torch.zeros_like(label)
    hard[torch.arange(len(label)), max_id] = 1
    hard, feat = hard[conf], feat[conf]
    return hard.t() @ feat, hard.sum(dim=0, keepdim=True).t()


def ema_update(sd, m: float, eman: bool = True):
    """momentum_update_ema (STiLModel.py:154-168)."""
    with torch.no_grad():
        for k in list(sd.keys()):
            if not k.startswith("model."):
                continue
            ke = "ema." + k[len("model."):]
            v = sd[k]
            is_buf = k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")
            if is_buf and not eman:
                continue
            if k.endswith("num_batches_tracked") or "offset" in k:  # STiLModel.py:162 / STiLModel_SAINT.py:161
                sd[ke].copy_(v)
            else:
                sd[ke].mul_(m).add_((1.0 - m) * v.detach())


# --------------------------------------------------------------------------
# the training step
# --------------------------------------------------------------------------
def training_step(sd: Dict[str, Tensor], batch, hp, current_epoch: int,
                  mask_random: Optional[Tensor] = None, mi_masks=None) -> Dict[str, Tensor]:
    """STiLModel.training_step (STiLModel.py:228-386).

    `sd` is mutated the way the reference mutates module state: BN running stats
    (student), EMA teacher weights, prototype accumulators.  `mask_random`
    (STiLModel.py:299) and the MI-layer dropout masks are injected so CPU and GPU
    see the same randomness.  Returns every intermediate the parity tests check.
    """
    im_l, tab_l, y_l = batch["l"][0][1], batch["l"][1][1], batch["l"][2]
    im_u, tab_u, y_u = batch["u"][0][1], batch["u"][1][1], batch["u"][2]
    assert bool(batch["l"][4].all()) and not bool(batch["u"][4].any())
    B_l, B_u = len(y_l), len(y_u)
    K, T, th, r = hp.num_classes, hp.temperature, hp.th1, hp.rate_pseudo
    x_img, x_tab = torch.cat((im_l, im_u)), torch.cat((tab_l, tab_u))

    s_out = backbone_forward_all(sd, "model.", x_img, x_tab, hp, train=True, masks=mi_masks)
    y_m, y_i, y_t, si_e, si_m, ai, st_e, st_m, at, xc = s_out
    feat_m = F.normalize(_head(sd, "projector_multimodal.", torch.cat((si_e, xc, st_e), dim=1)))
    feat_i = F.normalize(_head(sd, "projector_imaging.", ai))
    feat_t = F.normalize(_head(sd, "projector_tabular.", at))

    with torch.no_grad():
        if hp.use_ema:
            ema_update(sd, hp.ema_momentum, hp.eman)
            t_out = backbone_forward_all(sd, "ema.", x_img, x_tab, hp, train=False)
            feat_m_e = F.normalize(_head(sd, "projector_multimodal.", torch.cat((t_out[3], t_out[9], t_out[6]), dim=1), student=False))
            ym_e, yi_e, yt_e = t_out[0], t_out[1], t_out[2]
        else:
            t_out = tuple(t.detach() for t in s_out)
            ym_e, yi_e, yt_e, feat_m_e = y_m.detach(), y_i.detach(), y_t.detach(), feat_m.detach()
        feat_m_ue = feat_m_e[B_l:]
        ym_u, yi_u, yt_u = ym_e[B_l:], yi_e[B_l:], yt_e[B_l:]
        a, b, d = ym_u.softmax(1).argmax(1), yi_u.softmax(1).argmax(1), yt_u.softmax(1).argmax(1)
        case1 = (a == b) & (a == d)
        case2_i = (a == b) & (a != d)
        case2_t = (a == d) & (a != b)
        case3 = ~(case1 | case2_i | case2_t)
        q0 = (case1[:, None] * ((ym_u + yi_u + yt_u) / 3.0).softmax(1) + case2_i[:, None] * ((ym_u + yi_u) / 2.0).softmax(1)
              + case2_t[:, None] * ((ym_u + yt_u) / 2.0).softmax(1) + case3[:, None] * ym_u.softmax(1))
        prediction = ym_u.softmax(1)
        if hp.DA:  # distribution_alignment (STiLModel.py:171-180), single process: all_reduce is the identity
            ptr = int(sd["DA_ptr"])
            sd["DA_queue"][ptr] = prediction.mean(0)
            sd["DA_ptr"][0] = (ptr + 1) % sd["DA_queue"].shape[0]
            prediction = prediction / sd["DA_queue"].mean(0)
            prediction = prediction / prediction.sum(dim=1, keepdim=True)

    ce = F.cross_entropy
    loss_ce = ce(y_m[:B_l], y_l) + ce(y_i[:B_l], y_l) + ce(y_t[:B_l], y_l)
    prototypes = sd["prototypes"].clone()
    with torch.no_grad():
        tp = torch.softmax(feat_m_ue @ prototypes.t() / T, dim=1)
        pseudo_label = r * q0 + (1 - r) * tp
        prediction = r * prediction + (1 - r) * tp
        mask1 = prediction.max(dim=1)[0].ge(th)
        if mask_random is None:
            mask_random = torch.rand(B_u).ge(0.5)
    loss_m_u = (ce(y_m[B_l:], pseudo_label, reduction="none") * mask1 * case1).mean()
    loss_i_u = (ce(y_i[B_l:], pseudo_label, reduction="none") * mask1 * (case1 + case2_t + case3 * mask_random)).mean()
    loss_t_u = (ce(y_t[B_l:], pseudo_label, reduction="none") * mask1 * (case1 + case2_i + case3 * (~mask_random))).mean()

    if not current_epoch > hp.start_epoch:
        prediction = torch.zeros_like(prediction)
    pseudo_label_all = torch.cat((F.one_hot(y_l, K).float(), prediction), dim=0)
    loss_itc, itc_logits = clip_loss(feat_i, feat_t, T, hp.lambda_0)
    club_i, est_i = club_forward(sd, "CLUB_imaging.", si_m, ai)
    club_t, est_t = club_forward(sd, "CLUB_tabular.", st_m, at)
    loss_pt = prototype_loss(pseudo_label_all, prototypes, feat_m, T, th)
    loss = hp.alpha * loss_ce + hp.beta * loss_itc + hp.gamma * (club_i + est_i + club_t + est_t)
    if current_epoch > hp.start_epoch:
        loss = loss + hp.rate_pt * loss_pt + hp.rate_uce * (loss_m_u + loss_i_u + loss_t_u)

    with torch.no_grad():
        ls, lc = cal_prototypes(pseudo_label_all[:B_l], feat_m_e[:B_l], th)
        us, uc = cal_prototypes(pseudo_label_all[B_l:], feat_m_e[B_l:], th)
        class_sum = ls / hp.repeat_ratio + us
        class_count = lc / hp.repeat_ratio + uc
        sd["prototypes_sum"].add_(class_sum)
        sd["prototypes_count_sum"].add_(class_count)

    return dict(
        loss=loss, loss_ce=loss_ce, loss_itc=loss_itc, loss_club_i=club_i, loss_club_i_est=est_i,
        loss_club_t=club_t, loss_club_t_est=est_t, loss_pt=loss_pt, loss_m_u=loss_m_u, loss_i_u=loss_i_u,
        loss_t_u=loss_t_u,
        y_hat_m=y_m, y_hat_i=y_i, y_hat_t=y_t, x_si_enhance=si_e, x_si=si_m, x_ai=ai, x_st_enhance=st_e,
        x_st=st_m, x_at=at, x_c=xc, feat_m=feat_m, feat_i=feat_i, feat_t=feat_t,
        y_hat_m_e=ym_e, y_hat_i_e=yi_e, y_hat_t_e=yt_e, feat_m_e=feat_m_e,
        pseudo_label_orig=q0, pseudo_label=pseudo_label, prediction=prediction,
        case1=case1, case2_i=case2_i, case2_t=case2_t, case3=case3, mask1=mask1, mask_random=mask_random,
        class_sum=class_sum, class_count=class_count, itc_logits=itc_logits,
    )


def validation_step(sd, x_img, x_tab, y, hp):
    """STiLModel.validation_step (STiLModel.py:424-474), module in eval mode (Lightning's validation loop)."""
    with torch.no_grad():
        o = backbone_forward_all(sd, "model.", x_img, x_tab, hp, train=False)
        y_hat, y_i, y_t, si_e, si_m, ai, st_e, st_m, at, xc = o
        feat_i = F.normalize(_head(sd, "projector_imaging.", ai))
        feat_t = F.normalize(_head(sd, "projector_tabular.", at))
        loss_itc, logits = clip_loss(feat_i, feat_t, hp.temperature, hp.lambda_0)
        ci, ei = club_forward(sd, "CLUB_imaging.", si_m, ai)
        ct, et = club_forward(sd, "CLUB_tabular.", st_m, at)
        loss_ce = F.cross_entropy(y_hat, y)
        loss = hp.alpha * loss_ce + hp.beta * loss_itc + hp.gamma * (ci + ei + ct + et)
        return dict(loss=loss, loss_ce=loss_ce, loss_itc=loss_itc, y_hat=torch.softmax(y_hat, 1), y_i_hat=torch.softmax(y_i, 1),
                    y_t_hat=torch.softmax(y_t, 1), itc_logits=logits)


def test_step(sd, x_img, x_tab, hp):
    """STiLModel.test_step (STiLModel.py:517-530): softmax of the multimodal logits (column 1 for binary tasks)."""
    with torch.no_grad():
        y_hat = backbone_forward_all(sd, "model.", x_img, x_tab, hp, train=False)[0]
        p = torch.softmax(y_hat, dim=1)
        return p[:, 1] if hp.num_classes == 2 else p


def training_epoch_end(sd):
    """STiLModel.py:408-415: commit prototypes, zero accumulators."""
    with torch.no_grad():
        assert bool((sd["prototypes_count_sum"] >= 1).all()), "a class received no confident sample this epoch"
        sd["prototypes"].copy_(sd["prototypes_sum"] / sd["prototypes_count_sum"])
        sd["prototypes_sum"].zero_()
        sd["prototypes_count_sum"].zero_()


# --------------------------------------------------------------------------
# optimizer (torch.optim.Adam semantics, STiLModel.py:563-570) + full step driver
# --------------------------------------------------------------------------
def adam_step(sd, grads: Dict[str, Tensor], opt: Dict[str, Dict[str, Tensor]], step: int, lr: float,
              wd: float = 0.0, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            if wd != 0.0:
                g = g + wd * sd[k]
            st = opt.setdefault(k, dict(m=torch.zeros_like(sd[k]), v=torch.zeros_like(sd[k])))
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** step
            bc2 = 1 - b2 ** step
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
            sd[k].addcdiv_(st["m"], denom, value=-lr / bc1)


def anneal_lr(epoch: int, base_lr: float, warmup_epochs: int, max_epochs: int, warmup_start_lr: float = 0.0, eta_min: float = 0.0):
    """pl_bolts LinearWarmupCosineAnnealingLR closed form (unpinned: package absent offline)."""
    if epoch < warmup_epochs:
        return warmup_start_lr + epoch * (base_lr - warmup_start_lr) / max(1, warmup_epochs - 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * (epoch - warmup_epochs) / (max_epochs - warmup_epochs)))


def full_step(sd, opt, step_idx, batch, hp, current_epoch, mask_random=None, mi_masks=None, lr=None):
    """zero_grad -> training_step -> backward -> Adam (Lightning automatic optimisation)."""
    keys = trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    out = training_step(sd, batch, hp, current_epoch, mask_random, mi_masks)
    gl = torch.autograd.grad(out["loss"], [sd[k] for k in keys], allow_unused=True)
    for k in keys:
        sd[k].requires_grad_(False)
    grads = dict(zip(keys, gl))
    adam_step(sd, grads, opt, step_idx, hp.lr_eval if lr is None else lr, hp.weight_decay_eval, eps=getattr(hp, "adam_eps", 1e-8))
    out = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    out["grads"] = grads
    return out


# --------------------------------------------------------------------------
# synthetic batch in the reference's batch layout
# --------------------------------------------------------------------------
def field_order_permutation(field_lengths):
    """Index list that scatters [categorical..., continuous...] generated columns to their positions in field order."""
    cat_pos = [i for i, c in enumerate(field_lengths) if int(c) != 1]
    con_pos = [i for i, c in enumerate(field_lengths) if int(c) == 1]
    perm = [0] * len(field_lengths)
    for j, pos in enumerate(cat_pos + con_pos):
        perm[pos] = j
    return perm


def synthetic_batch(hp, B: int, seed: int = 2022, img_size: Optional[int] = None):
    g = torch.Generator().manual_seed(seed)
    P = img_size or hp.img_size
    cat, con = split_field_lengths(hp.field_lengths)
    B_l = max(B // 8, 1)
    img = torch.rand(B, 3, P, P, generator=g)
    cols = [torch.randint(0, c, (B, 1), generator=g).float() for c in cat]
    cols.append(torch.randn(B, len(con), generator=g))
    tab = torch.cat(cols, dim=1)
    tab = tab[:, field_order_permutation(hp.field_lengths)]  # identity for categorical-first column orders
    y = torch.randint(0, hp.num_classes, (B,), generator=g)

    def part(sl, lab):
        n = sl.stop - sl.start
        return ([torch.zeros(n), img[sl]], [tab[sl], tab[sl]], y[sl], img[sl], torch.full((n,), lab, dtype=torch.bool))

    return {"l": part(slice(0, B_l), True), "u": part(slice(B_l, B), False)}
