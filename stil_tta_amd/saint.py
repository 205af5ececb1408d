"""SAINT tabular encoder variant (config_dvm_STiL_SAINT): parameter holders with the reference's state_dict keys
(models/Disentangle/utils/SAINT/Tabular_Encoder.py:24-146, SAINT/model_util.py:25-188) and the HIP forward of
DisCoAttentionBackbone.forward_tabular / RowColTransformer 'colrow' (STiLModel_SAINT_backbone.py:94-184).

Only embeds, pos_encodings, simple_MLP and transformer.layers take part in STiL's forward; the remaining SAINT heads
(mlp, mlp1, mlp2, mlpfory, pt_mlp*, mask embeddings) are carried in the state_dict / EMA exactly as the reference does.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .modules import MLP, MITransformerLayer, ResNet

SAINT_DIM, SAINT_HEADS = 32, 4


class simple_MLP(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(dims[0], dims[1]), nn.ReLU(), nn.Linear(dims[1], dims[2]))


class sep_MLP(nn.Module):
    def __init__(self, dim, len_feats, categories):
        super().__init__()
        self.layers = nn.ModuleList([simple_MLP([dim, 5 * dim, int(categories[i])]) for i in range(len_feats)])


class _MLP(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.mlp = nn.Sequential(*[nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])


class _Attention(nn.Module):
    def __init__(self, dim, heads, dim_head):
        super().__init__()
        self.heads, self.dim_head = heads, dim_head
        self.to_qkv = nn.Linear(dim, heads * dim_head * 3, bias=False)
        self.to_out = nn.Linear(heads * dim_head, dim)


class _FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), nn.Identity(), nn.Identity(), nn.Linear(dim * mult, dim))


class _Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class RowColTransformer(nn.Module):
    def __init__(self, num_tokens, dim, nfeats, heads, dim_head):
        super().__init__()
        self.embeds = nn.Embedding(num_tokens, dim)
        self.layers = nn.ModuleList([nn.ModuleList([
            _PreNorm(dim, _Residual(_Attention(dim, heads, dim_head))),
            _PreNorm(dim, _Residual(_FeedForward(dim))),
            _PreNorm(dim * nfeats, _Residual(_Attention(dim * nfeats, heads, 64))),
            _PreNorm(dim * nfeats, _Residual(_FeedForward(dim * nfeats))),
        ])])
        self.mask_embed = nn.Embedding(nfeats, dim)


class SAINT(nn.Module):
    def __init__(self, categories: List[int], num_continuous: int, num_classes: int):
        super().__init__()
        d, ncat, ncon = SAINT_DIM, len(categories), num_continuous
        nfeats = ncat + ncon + 1
        total_tokens = sum(categories) + 1
        self.register_buffer("categories_offset", torch.tensor([0, 1] + list(categories)).cumsum(0)[:-1])
        self.norm = nn.LayerNorm(ncon)
        self.simple_MLP = nn.ModuleList([simple_MLP([1, 100, d]) for _ in range(ncon)])
        self.transformer = RowColTransformer(total_tokens, d, nfeats, SAINT_HEADS, 16)
        input_size = d * ncat + d * ncon
        l = input_size // 8
        self.mlp = _MLP([input_size, l * 4, l * 2, 1])
        self.embeds = nn.Embedding(total_tokens, d)
        self.register_buffer("cat_mask_offset", torch.tensor([0, 2] + [2] * ncat).cumsum(0)[:-1])
        self.register_buffer("con_mask_offset", F.pad(torch.full((ncon,), 2.0).to(torch.int8), (1, 0), value=0).cumsum(0)[:-1])
        self.mask_embeds_cat = nn.Embedding(ncat * 2 + 2, d)
        self.mask_embeds_cont = nn.Embedding(ncon * 2, d)
        self.single_mask = nn.Embedding(2, d)
        self.pos_encodings = nn.Embedding(ncat + ncon, d)
        self.mlp1 = sep_MLP(d, ncat, categories)
        self.mlp2 = sep_MLP(d, ncon, [1] * ncon)
        self.mlpfory = simple_MLP([d, 1000, num_classes])
        self.pt_mlp = simple_MLP([d * nfeats, 6 * d * nfeats // 5, d * nfeats // 2])
        self.pt_mlp2 = simple_MLP([d * nfeats, 6 * d * nfeats // 5, d * nfeats // 2])


def _prenorm_res(x, pn: _PreNorm, fn):
    xn = ops.layernorm(x, pn.norm.weight, pn.norm.bias)
    return ops.drop_add(fn(xn), resid=xn, rowlen=xn.shape[-1])  # PreNorm(Residual(fn)): fn(norm(x)) + norm(x)


def _ff(ffm: _FeedForward, xn, mask, p):
    h = ops.linear(xn, ffm.net[0].weight, ffm.net[0].bias)
    h = ops.GegluFn.apply(h)
    if mask is not None:
        h = ops.drop_add(h, emask=mask, rowlen=h.shape[-1], scale=1.0 / (1.0 - p))
    return ops.linear(h, ffm.net[3].weight, ffm.net[3].bias)


def register_saint_meta(self, cats):
    """Column metadata of the fused embedding kernel as non-persistent buffers (absent from the state_dict)."""
    rowcol = torch.cat([torch.zeros(1, dtype=torch.int32)] + [torch.full((c,), j + 1, dtype=torch.int32) for j, c in enumerate(cats)])
    self.register_buffer("_cat_cols", torch.tensor(self.cat_cols, dtype=torch.int32), persistent=False)
    self.register_buffer("_con_cols", torch.tensor(self.con_cols, dtype=torch.int32), persistent=False)
    self.register_buffer("_rowcol", rowcol, persistent=False)
    self.ff_drop = 0.8
    self._ncat = len(cats)


def saint_forward_tabular(self, x_t, masks=None):
    """forward_tabular of STiLModel_SAINT_backbone.py:159-184 / Multimodal_model_SAINT.py:160-185 for a module that holds
    `encoder_tabular` (SAINT), `cls_token` and the column metadata of register_saint_meta.  The categorical offsets are read
    from the module's LIVE `categories_offset` buffer (a loaded checkpoint, or the teacher copy that CoTraining_SAINT's EMA
    alters, may differ from the constructor's values)."""
    enc = self.encoder_tabular
    meta = dict(ncat=self._ncat, ncon=len(self.con_cols), hid=100, cat_cols=self._cat_cols, con_cols=self._con_cols,
                offs=enc.categories_offset.to(torch.int32), rowcol=self._rowcol)
    mlp_params = []
    for m in enc.simple_MLP:
        mlp_params += [m.layers[0].weight, m.layers[0].bias, m.layers[2].weight, m.layers[2].bias]
    x = ops.SaintEmbedColMlpFn.apply(x_t.contiguous(), enc.embeds.weight, enc.pos_encodings.weight, meta, *mlp_params)
    B, n, d = x.shape
    a1, f1, a2, f2 = enc.transformer.layers[0]
    mk = (lambda k: None) if masks is None else (lambda k: masks[k])

    def col_attn(xn):
        at = a1.fn.fn
        qkv = ops.linear(xn, at.to_qkv.weight, None)
        o = ops.attention(qkv, at.heads, [(0, n, 0, n)])
        return ops.linear(o, at.to_out.weight, at.to_out.bias)

    def row_attn(xn):  # inter-sample attention: sequence = the batch, 4 heads x 64 (SAINT/model_util.py:79-87,117-119)
        at = a2.fn.fn
        qkv = ops.linear(xn, at.to_qkv.weight, None)  # [B, 3*256]
        inner = at.heads * at.dim_head
        outs = []
        for h in range(at.heads):
            q = qkv[:, h * 64:(h + 1) * 64].contiguous()
            k = qkv[:, inner + h * 64: inner + (h + 1) * 64].contiguous()
            v = qkv[:, 2 * inner + h * 64: 2 * inner + (h + 1) * 64].contiguous()
            P = ops.RowSoftmaxFn.apply(ops.MatmulNTFn.apply(q, k, at.dim_head ** -0.5))
            outs.append(ops.MatmulNNFn.apply(P, v))
        return ops.linear(torch.cat(outs, dim=1), at.to_out.weight, at.to_out.bias)

    x = _prenorm_res(x, a1, col_attn)
    x = _prenorm_res(x, f1, lambda xn: _ff(f1.fn.fn, xn, mk("ff_col"), self.ff_drop))
    xr = x.reshape(B, n * d)
    xr = _prenorm_res(xr, a2, row_attn)
    xr = _prenorm_res(xr, f2, lambda xn: _ff(f2.fn.fn, xn, mk("ff_row"), self.ff_drop))
    return xr.reshape(B, n, d)


class SaintBackbone(nn.Module):
    """DisCoAttentionBackbone of STiLModel_SAINT_backbone.py:37-234 (same heads as the base backbone)."""

    def __init__(self, hp, field_lengths):
        super().__init__()
        self.encoder_imaging = ResNet(hp.model)
        self.cat_cols = [i for i, c in enumerate(field_lengths) if int(c) != 1]
        self.con_cols = [i for i, c in enumerate(field_lengths) if int(c) == 1]
        cats = [int(field_lengths[i]) for i in self.cat_cols]
        self.encoder_tabular = SAINT(cats, len(self.con_cols), hp.num_classes)
        self.cls_token = nn.Parameter(torch.zeros(1, 1))
        pooled, C, Dt = hp.embedding_dim, hp.multimodal_embedding_dim, SAINT_DIM
        self.projection_si = MLP(pooled, C, C)
        self.projection_ai = MLP(pooled, C, C)
        self.projection_st = MLP(Dt, Dt, C)
        self.projection_at = MLP(Dt, Dt, C)
        self.reduce = nn.Linear(C * 2, C)
        self.transformer = nn.ModuleList([MITransformerLayer(C, 4, 0.1) for _ in range(hp.multimodal_transformer_num_layers)])
        self.classifier_multimodal = nn.Linear(C * 3, hp.num_classes)
        self.classifier_imaging = nn.Linear(C * 2, hp.num_classes)
        self.classifier_tabular = nn.Linear(C * 2, hp.num_classes)
        register_saint_meta(self, cats)

    def forward_tabular(self, x_t, masks=None):
        return saint_forward_tabular(self, x_t, masks)

    def tabular_tokens(self, x_tab, train: bool, mi_masks=None):
        sm = None if (mi_masks is None or not train) else mi_masks.get("saint")
        return self.forward_tabular(x_tab, sm)

    def forward_all(self, x, train: Optional[bool] = None, mi_masks=None, cache=None, x_i=None, x_t=None):
        train = self.training if train is None else train
        x_img, x_tab = x[0], x[1]
        if x_i is None:
            x_i = self.encoder_imaging.run(x_img, train, cache)
        if x_t is None:
            x_t = self.tabular_tokens(x_tab, train, mi_masks)
        x_si = self.projection_si.run(x_i)
        x_ai = self.projection_ai.run(ops.tokmean(x_i))
        x_st = self.projection_st.run(x_t[:, 1:, :].contiguous())
        x_at = self.projection_at.run(x_t[:, 0, :].contiguous())
        x_c = ops.linear(torch.cat([x_ai, x_at], dim=1), self.reduce.weight, self.reduce.bias)
        Ni, Nt = x_si.shape[1], x_st.shape[1]
        X = torch.cat([x_c.unsqueeze(1), x_si, x_st], dim=1)
        for li, layer in enumerate(self.transformer):
            mk = None if (mi_masks is None or not train) else mi_masks.get(li)
            X = layer.run(X, Ni, Nt, mk)
        e_c = X[:, 0, :].contiguous()
        e_si = ops.tokmean(X[:, 1:1 + Ni, :])
        e_st = ops.tokmean(X[:, 1 + Ni:, :])
        out_m = ops.linear(torch.cat([e_si, e_c, e_st], dim=1), self.classifier_multimodal.weight, self.classifier_multimodal.bias)
        out_i = ops.linear(torch.cat([e_si, x_ai], dim=1), self.classifier_imaging.weight, self.classifier_imaging.bias)
        out_t = ops.linear(torch.cat([e_st, x_at], dim=1), self.classifier_tabular.weight, self.classifier_tabular.bias)
        return out_m, out_i, out_t, e_si, ops.tokmean(x_si), x_ai, e_st, ops.tokmean(x_st), x_at, e_c

    def forward(self, x, train: Optional[bool] = None):
        o = self.forward_all(x, train)
        return o[0], o[1], o[2], o[3], o[5], o[6], o[8], o[9]
