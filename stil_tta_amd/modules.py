"""Module tree of the STiL backbone with the reference's parameter names.

torch.nn containers (nn.Linear, nn.Conv2d, nn.BatchNorm2d, nn.LayerNorm, nn.Embedding) are used only as
PARAMETER HOLDERS -- identical names, shapes and default initialisation to the reference, so released
checkpoints load with load_state_dict -- their own forward() is never called: every forward below goes
through the HIP operators in ops.py.  Activations are NHWC.

Mirrors  models/resnets.py (ResNet/Bottleneck/BasicBlock), models/Transformer.py
(TabularTransformerEncoder/Block/Attention/Mlp), models/Disentangle/utils/disentangle_transformer.py
(MITransformerLayer/MIAttention) and models/Disentangle/utils/STiLModel_backbone.py (DisCoAttentionBackbone).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops


# ------------------------------------------------------------------------------------------ ResNet
class TeacherPipe:
    """Layer-wise hand-over between the student (main stream) and the EMA teacher (side stream).  Every training-mode
    conv+BN publishes an event after its running statistics are updated; the k-th eval-mode conv+BN first waits for the
    k-th event, then (eman) averages its own running_mean / running_var from the student's -- the per-layer slice of
    momentum_update_ema (STiLModel.py:154-168) -- and only then reads them.  ResNet.run_pair issues the two networks
    block by block, so the teacher trails the student by one block on the device as well as on the host."""

    def __init__(self, flat, momentum: float, eman: bool):
        self.flat, self.momentum, self.eman = flat, float(momentum), eman
        self.events = []
        self.i = 0
        self.start = torch.cuda.Event()

    def published(self):
        ev = torch.cuda.Event()
        ev.record(ops.current_stream_obj())
        self.events.append(ev)

    def before_teacher_bn(self, bn: nn.BatchNorm2d):
        ops.current_stream_obj().wait_event(self.events[self.i])
        self.i += 1
        if self.eman:
            f = self.flat
            off = (bn.running_mean.data_ptr() - f.ema.data_ptr()) // 4
            n = (bn.running_var.data_ptr() - bn.running_mean.data_ptr()) // 4  # = padded length of one buffer
            f.ema_update_range(off, 2 * n, self.momentum)


_PIPE: Optional[TeacherPipe] = None


def set_teacher_pipe(pipe: Optional[TeacherPipe]):
    global _PIPE
    _PIPE = pipe


def _conv_bn(x, conv: nn.Conv2d, bn: nn.BatchNorm2d, relu: bool, train: bool, resid=None, stem=None, passthrough=False, defer=False,
             xstats=None, premask_in=None, premask_out=None, rstats=None, bstat_send=None, bstat_recv=None):
    """passthrough (first conv of a residual block): -> (out, alias of x) so the identity branch's gradient is folded
    into this conv's input-gradient GEMM (ops.ConvBnActFn).
    defer (training, inner layers of a block): -> (..., stats) with out = the RAW conv output; the next conv gets them as
    `xstats` and applies this layer's BatchNorm + ReLU while it stages its operand (the student's bn1 / bn2 of
    models/resnets.py:112-132 fused into conv2 / conv3)."""
    k, stride, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    pipe = _PIPE
    if train:
        out = ops.ConvBnActFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                    bn.num_batches_tracked, resid, k, stride, pad, relu, stem, passthrough, defer, xstats,
                                    premask_in, premask_out, rstats, bstat_send, bstat_recv)
        if pipe is not None:
            pipe.published()
        return out
    assert not defer and xstats is None and rstats is None
    if pipe is not None:
        pipe.before_teacher_bn(bn)
    out = ops.conv_bn_eval(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, resid, k, stride, pad,
                           relu, stem)
    return (out, x) if passthrough else out


class Bottleneck(nn.Module):  # models/resnets.py:91-132
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def run(self, x, train):
        d1 = train and ops.can_defer_bn(self.conv1.out_channels)   # bn1 + relu applied inside conv2's operand staging
        d2 = train and ops.can_defer_bn(self.conv2.out_channels)   # bn2 + relu applied inside conv3's operand staging
        st1 = st2 = None
        pm = train and ops._PREMASK
        cin = getattr(x, "_stil_premask", None) if pm else None   # x is the previous block's relu output: hand its gradient back masked
        cout = {} if pm else None
        # BatchNorm-backward sums ride in the consumer's input-gradient GEMM (ops.ConvBnActFn bstat_send / bstat_recv): one cell
        # per conv+BN layer with a single consumer conv -- bn1 -> conv2, bn2 -> conv3, bn3 (block output) -> the next block's conv1
        bin_ = getattr(x, "_stil_bstat", None) if train else None
        b1, b2, b3 = ({}, {}, {}) if train else (None, None, None)
        if d1:
            out, identity, st1 = _conv_bn(x, self.conv1, self.bn1, True, train, passthrough=True, defer=True, premask_in=cin, bstat_send=bin_, bstat_recv=b1)
        else:
            out, identity = _conv_bn(x, self.conv1, self.bn1, True, train, passthrough=True, premask_in=cin, bstat_send=bin_)
        if d2:
            out, st2 = _conv_bn(out, self.conv2, self.bn2, True, train, defer=True, xstats=st1, bstat_send=b1, bstat_recv=b2)
        else:
            out = _conv_bn(out, self.conv2, self.bn2, True, train, xstats=st1, bstat_send=b1)
        rst = None
        if self.downsample is not None:
            if train and ops.can_defer_bn(self.downsample[0].out_channels):   # the shortcut's BatchNorm is applied inside bn3's pass
                identity, rst = _conv_bn(identity, self.downsample[0], self.downsample[1], False, train, defer=True)
            else:
                identity = _conv_bn(identity, self.downsample[0], self.downsample[1], False, train)
        Nb, H, W, C = identity.shape
        out = _conv_bn(out, self.conv3, self.bn3, True, train, resid=identity.reshape(Nb * H * W, C), xstats=st2, premask_out=cout, rstats=rst,
                       bstat_send=b2, bstat_recv=b3)
        if cout is not None:
            out._stil_premask = cout   # the next block's first conv (the only consumer of `out`) picks it up
            out._stil_bstat = b3
        return out


class BasicBlock(nn.Module):  # models/resnets.py:50-88
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def run(self, x, train):
        st1 = None
        pm = train and ops._PREMASK
        cin = getattr(x, "_stil_premask", None) if pm else None
        cout = {} if pm else None
        bin_ = getattr(x, "_stil_bstat", None) if train else None
        b1, b2 = ({}, {}) if train else (None, None)
        if train and ops.can_defer_bn(self.conv1.out_channels):     # bn1 + relu applied inside conv2's operand staging
            out, identity, st1 = _conv_bn(x, self.conv1, self.bn1, True, train, passthrough=True, defer=True, premask_in=cin, bstat_send=bin_, bstat_recv=b1)
        else:
            out, identity = _conv_bn(x, self.conv1, self.bn1, True, train, passthrough=True, premask_in=cin, bstat_send=bin_)
        rst = None
        if self.downsample is not None:
            if train and ops.can_defer_bn(self.downsample[0].out_channels):
                identity, rst = _conv_bn(identity, self.downsample[0], self.downsample[1], False, train, defer=True)
            else:
                identity = _conv_bn(identity, self.downsample[0], self.downsample[1], False, train)
        Nb, H, W, C = identity.shape
        out = _conv_bn(out, self.conv2, self.bn2, True, train, resid=identity.reshape(Nb * H * W, C), xstats=st1, premask_out=cout, rstats=rst,
                       bstat_send=b1, bstat_recv=b2)
        if cout is not None:
            out._stil_premask = cout
            out._stil_bstat = b2
        return out


class ResNet(nn.Module):
    """models/resnets.py:135-269 with fc = Identity (models/self_supervised.py:14); returns the last feature map
    as tokens [B, H*W, C] (STiLModel_backbone.py:121-124)."""

    CFG = {"resnet18": (BasicBlock, [2, 2, 2, 2]), "resnet34": (BasicBlock, [3, 4, 6, 3]),
           "resnet50": (Bottleneck, [3, 4, 6, 3]), "resnet101": (Bottleneck, [3, 4, 23, 3]),
           "resnet152": (Bottleneck, [3, 8, 36, 3])}

    def __init__(self, name: str):
        super().__init__()
        block, layers = self.CFG[name]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.out_dim = 512 * block.expansion
        for m in self.modules():  # models/resnets.py:190-195
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _stem(self, x_nchw, train: bool, cache: Optional[dict]):
        # stem: im2col is shared between the student and the teacher pass (same input batch)
        if cache is not None and "stem_col" in cache:
            col, meta = cache["stem_col"]
        else:
            col, meta = ops.im2col_stem(x_nchw.contiguous(), 7, 2, 3)
            if cache is not None:
                cache["stem_col"] = (col, meta)
        wpad = ops.pad_stem_weight(self.conv1.weight, meta[3])
        x = _conv_bn(col, self.conv1, self.bn1, True, train, stem=(*meta, wpad))
        return ops.MaxPoolFn.apply(x)

    def _blocks(self):
        return [blk for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for blk in layer]

    def run(self, x_nchw, train: bool, cache: Optional[dict] = None):
        x = self._stem(x_nchw, train, cache)
        for blk in self._blocks():
            x = blk.run(x, train)
        Nb, H, W, C = x.shape
        return x.reshape(Nb, H * W, C)

    def run_pair(self, teacher: "ResNet", x_nchw, cache: dict, side: torch.cuda.Stream):
        """Student (self, training mode, current stream) and EMA teacher (eval mode, no grad, `side`) issued block by
        block; needs an active TeacherPipe (set_teacher_pipe) for the per-BN hand-over.  -> (student tokens, teacher tokens)"""
        def on_side(fn, *a):
            with torch.no_grad(), torch.cuda.stream(side):
                return fn(*a)
        xs = self._stem(x_nchw, True, cache)
        xt = on_side(teacher._stem, x_nchw, False, cache)
        for bs, bt in zip(self._blocks(), teacher._blocks()):
            xs = bs.run(xs, True)
            xt = on_side(bt.run, xt, False)
        Nb, H, W, C = xs.shape
        return xs.reshape(Nb, H * W, C), on_side(lambda t: t.reshape(Nb, H * W, C), xt)


# ------------------------------------------------------------------------------------------ tabular transformer
class Mlp(nn.Module):  # models/Transformer.py:17-33
    def __init__(self, in_f, hid, out_f=None):
        super().__init__()
        self.fc1 = nn.Linear(in_f, hid)
        self.fc2 = nn.Linear(hid, out_f or in_f)


class Attention(nn.Module):  # models/Transformer.py:36-92
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):  # models/Transformer.py:145-174 (pre-LN, dropouts p = 0, DropPath = Identity)
    def __init__(self, dim, num_heads=8, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def run(self, x):
        B, T, D = x.shape
        a = ops.layernorm(x, self.norm1.weight, self.norm1.bias)
        qkv = ops.linear(a, self.attn.qkv.weight, self.attn.qkv.bias)
        o = ops.attention(qkv, self.attn.num_heads, [(0, T, 0, T)])
        # proj + residual and fc2 + residual: the sums ride in the GEMM epilogue (round 4; STIL_LINEAR_RESID=0: a pass of their own)
        if ops._LINEAR_RESID:
            x = ops.linear(o, self.attn.proj.weight, self.attn.proj.bias, resid=x.contiguous())
        else:
            x = ops.drop_add(ops.linear(o, self.attn.proj.weight, self.attn.proj.bias), resid=x, rowlen=D)
        m = ops.layernorm(x, self.norm2.weight, self.norm2.bias)
        m = ops.linear(m, self.mlp.fc1.weight, self.mlp.fc1.bias, act=2)
        if ops._LINEAR_RESID:
            return ops.linear(m, self.mlp.fc2.weight, self.mlp.fc2.bias, resid=x)
        return ops.drop_add(ops.linear(m, self.mlp.fc2.weight, self.mlp.fc2.bias), resid=x, rowlen=D)


class TabularTransformerEncoder(nn.Module):  # models/Transformer.py:186-278
    def __init__(self, hp, cat_lengths: List[int], con_lengths: List[int]):
        super().__init__()
        D = hp.tabular_embedding_dim
        self.num_cat, self.num_con = len(cat_lengths), len(con_lengths)
        self.cat_embedding = nn.Embedding(sum(cat_lengths), D)
        self.con_proj = nn.Linear(1, D)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.mask_special_token = nn.Parameter(torch.zeros(1, 1, D))  # unused by STiL (no grad), kept for the state_dict
        self.column_embedding = nn.Embedding(self.num_cat + self.num_con + 1, D)
        self.norm = nn.LayerNorm(D)
        self.transformer_blocks = nn.ModuleList([Block(D) for _ in range(hp.tabular_transformer_num_layers)])
        offs = torch.tensor([0] + list(cat_lengths[:-1]), dtype=torch.int64).cumsum(0).to(torch.int32)
        rowcol = torch.cat([torch.full((c,), j, dtype=torch.int32) for j, c in enumerate(cat_lengths)]) if cat_lengths \
            else torch.zeros(0, dtype=torch.int32)
        self.register_buffer("cat_offsets", offs, persistent=False)
        self.register_buffer("emb_rowcol", rowcol, persistent=False)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        nn.init.trunc_normal_(self.mask_special_token, std=0.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):  # models/Transformer.py:231-238
        if isinstance(m, (nn.Linear, nn.Embedding)):
            m.weight.data.normal_(mean=0.0, std=0.02)
        elif isinstance(m, nn.LayerNorm):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)
        if isinstance(m, nn.Linear) and m.bias is not None:
            m.bias.data.zero_()

    def run(self, x):
        h = ops.TabEmbedFn.apply(x.contiguous(), self.cat_embedding.weight if self.num_cat else None,
                                 self.con_proj.weight if self.num_con else None, self.con_proj.bias if self.num_con else None,
                                 self.cls_token, self.column_embedding.weight, self.cat_offsets, self.emb_rowcol, self.num_cat)
        h = ops.layernorm(h, self.norm.weight, self.norm.bias)
        for blk in self.transformer_blocks:
            h = blk.run(h)
        return h


# ------------------------------------------------------------------------------------------ MI transformer layer
class MIAttention(nn.Module):  # disentangle_transformer.py:29-47
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class MITransformerLayer(nn.Module):
    """disentangle_transformer.py:125-169.  The three streams share all weights, so they are processed as ONE
    token buffer [B, 1+Ni+Nt, C] ordered (shared, image, tabular): LN/qkv/proj/MLP are single GEMMs over all
    rows, and the three attentions are windows of that buffer (the shared token's keys are exactly
    cat(k_global, k_histology, k_pathways), disentangle_transformer.py:69)."""

    def __init__(self, dim, num_heads=4, p=0.1):
        super().__init__()
        self.p = p
        self.norm1 = nn.LayerNorm(dim)
        self.attn = MIAttention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, dim, dim)

    def run(self, X, Ni, Nt, masks: Optional[Dict[str, torch.Tensor]]):
        B, T, C = X.shape
        p = self.p
        sc = 1.0 / (1.0 - p) if masks is not None else 1.0
        mk = (lambda k: None) if masks is None else (lambda k: masks[k])
        a = ops.layernorm(X, self.norm1.weight, self.norm1.bias)
        qkv = ops.linear(a, self.attn.qkv.weight, self.attn.qkv.bias)
        windows = [(1, Ni, 1, Ni), (1 + Ni, Nt, 1 + Ni, Nt), (0, 1, 0, T)]
        am = None if masks is None else [masks["attn_i"], masks["attn_t"], masks["attn_c"]]
        o = ops.attention(qkv, self.attn.num_heads, windows, am, p if masks is not None else 0.0)
        pr = ops.linear(o, self.attn.proj.weight, self.attn.proj.bias)
        pr = ops.drop_add(pr, emask=mk("proj"), rowlen=C, scale=sc)                       # proj_drop
        X = ops.drop_add(pr, resid=X, rmask=mk("dp1"), rowlen=C, scale=sc)                # drop_path + residual
        m = ops.layernorm(X, self.norm2.weight, self.norm2.bias)
        m = ops.linear(m, self.mlp.fc1.weight, self.mlp.fc1.bias, act=2)
        m = ops.drop_add(m, emask=mk("fc1"), rowlen=C, scale=sc)
        m = ops.linear(m, self.mlp.fc2.weight, self.mlp.fc2.bias)
        m = ops.drop_add(m, emask=mk("fc2"), rowlen=C, scale=sc)
        return ops.drop_add(m, resid=X, rmask=mk("dp2"), rowlen=C, scale=sc)


def fuse_mi_masks(m: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """Oracle-format per-stream keep masks (oracle/make_golden.py:make_mi_masks) -> fused token-buffer layout."""
    u8 = lambda t: t.to(device=device, dtype=torch.uint8).contiguous()
    Ni, Nt = m["proj_i"].shape[1], m["proj_t"].shape[1]
    out = {"attn_i": u8(m["attn_i"]), "attn_t": u8(m["attn_t"]), "attn_c": u8(m["attn_c"])}
    for k in ("proj", "fc1", "fc2"):
        out[k] = u8(torch.cat([m[k + "_c"], m[k + "_i"], m[k + "_t"]], dim=1))
    for k in ("dp1", "dp2"):
        out[k] = u8(torch.cat([m[k + "_c"][:, None], m[k + "_i"][:, None].expand(-1, Ni), m[k + "_t"][:, None].expand(-1, Nt)], dim=1))
    return out


_stream_idx_cache = {}


def random_mi_masks(B, Ni, Nt, C, H, p, seed, offset, device, step=None) -> Dict[str, torch.Tensor]:
    """Train-mode stochasticity of the MI layer drawn on the device (counter-based hash RNG)."""
    T = 1 + Ni + Nt
    o = [offset]

    def draw(*shape):
        n = 1
        for s in shape:
            n *= s
        t = ops.rng_mask(shape, p, seed, o[0], device, step)
        o[0] += n
        return t

    out = {"attn_i": draw(B, H, Ni, Ni), "attn_t": draw(B, H, Nt, Nt), "attn_c": draw(B, H, 1, T),
           "proj": draw(B, T, C), "fc1": draw(B, T, C), "fc2": draw(B, T, C)}
    idx = _stream_idx_cache.get((Ni, Nt, str(device)))
    if idx is None:  # built once: an H2D copy per step could not be captured into a hipGraph
        idx = torch.cat([torch.zeros(1, dtype=torch.long), torch.ones(Ni, dtype=torch.long), torch.full((Nt,), 2, dtype=torch.long)]).to(device)
        _stream_idx_cache[(Ni, Nt, str(device))] = idx
    for k in ("dp1", "dp2"):
        out[k] = draw(B, 3)[:, idx].contiguous()
    return out


# ------------------------------------------------------------------------------------------ backbone
class MLP(nn.Module):  # STiLModel_backbone.py:19-32
    def __init__(self, in_dim, hidden_dim, out_dim):
        super().__init__()
        self.model = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.ReLU(inplace=True), nn.Linear(hidden_dim, out_dim))

    def run(self, x):
        h = ops.linear(x, self.model[0].weight, self.model[0].bias, act=1)
        return ops.linear(h, self.model[2].weight, self.model[2].bias)


def split_field_lengths(field_lengths):
    cat = [int(v) for v in field_lengths if int(v) != 1]
    con = [int(v) for v in field_lengths if int(v) == 1]
    return cat, con


class DisCoAttentionBackbone(nn.Module):
    """STiLModel_backbone.py:35-165."""

    def __init__(self, hp, field_lengths):
        super().__init__()
        self.encoder_imaging = ResNet(hp.model)
        cat, con = split_field_lengths(field_lengths)
        self.encoder_tabular = TabularTransformerEncoder(hp, cat, con)
        pooled, C, Dt = hp.embedding_dim, hp.multimodal_embedding_dim, hp.tabular_embedding_dim
        self.projection_si = MLP(pooled, C, C)
        self.projection_ai = MLP(pooled, C, C)
        self.projection_st = MLP(Dt, Dt, C)
        self.projection_at = MLP(Dt, Dt, C)
        self.reduce = nn.Linear(C * 2, C)
        self.transformer = nn.ModuleList([MITransformerLayer(C, 4, 0.1) for _ in range(hp.multimodal_transformer_num_layers)])
        self.classifier_multimodal = nn.Linear(C * 3, hp.num_classes)
        self.classifier_imaging = nn.Linear(C * 2, hp.num_classes)
        self.classifier_tabular = nn.Linear(C * 2, hp.num_classes)

    def tabular_tokens(self, x_tab, train: bool, mi_masks=None):
        """the tabular encoder alone -> [B, Nt+1, Dt] (STiLModel.training_step issues it on its own stream, beside the image encoder)"""
        return self.encoder_tabular.run(x_tab)

    def forward_all(self, x, train: Optional[bool] = None, mi_masks=None, cache=None, x_i=None, x_t=None):
        """-> (out_m, out_i, out_t, x_si_enhance, mean(x_si), x_ai, x_st_enhance, mean(x_st), x_at, x_c);
        x_i / x_t: image / tabular tokens when the caller already ran that encoder (ResNet.run_pair, tabular_tokens)."""
        train = self.training if train is None else train
        x_img, x_tab = x[0], x[1]
        if x_i is None:
            x_i = self.encoder_imaging.run(x_img, train, cache)       # [B, Ni, pooled]
        if x_t is None:
            x_t = self.tabular_tokens(x_tab, train, mi_masks)         # [B, Nt+1, Dt]
        x_si = self.projection_si.run(x_i)
        x_ai = self.projection_ai.run(ops.tokmean(x_i))
        x_st = self.projection_st.run(x_t[:, 1:, :].contiguous())
        x_at = self.projection_at.run(x_t[:, 0, :].contiguous())
        x_c = ops.linear(torch.cat([x_ai, x_at], dim=1), self.reduce.weight, self.reduce.bias)
        Ni, Nt = x_si.shape[1], x_st.shape[1]
        X = torch.cat([x_c.unsqueeze(1), x_si, x_st], dim=1)
        for li, layer in enumerate(self.transformer):
            mk = None if (mi_masks is None or not train) else mi_masks[li]
            X = layer.run(X, Ni, Nt, mk)
        e_c = X[:, 0, :].contiguous()
        e_si = ops.tokmean(X[:, 1:1 + Ni, :])
        e_st = ops.tokmean(X[:, 1 + Ni:, :])
        out_m = ops.linear(torch.cat([e_si, e_c, e_st], dim=1), self.classifier_multimodal.weight, self.classifier_multimodal.bias)
        out_i = ops.linear(torch.cat([e_si, x_ai], dim=1), self.classifier_imaging.weight, self.classifier_imaging.bias)
        out_t = ops.linear(torch.cat([e_st, x_at], dim=1), self.classifier_tabular.weight, self.classifier_tabular.bias)
        return out_m, out_i, out_t, e_si, ops.tokmean(x_si), x_ai, e_st, ops.tokmean(x_st), x_at, e_c

    def forward(self, x, train: Optional[bool] = None):
        """STiLModel_backbone.py:159-165 (8-tuple used by test_step)."""
        o = self.forward_all(x, train)
        return o[0], o[1], o[2], o[3], o[5], o[6], o[8], o[9]
