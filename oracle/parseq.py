"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference PARSeq text recogniser as functions of a state dict, plain
PyTorch fp32 (paths relative to /root/reference/src/yomitoku).  It follows the reference
literally - no K/V caching, every AR step re-projects the whole context and memory
(models/parseq.py:204-250) - so that the HIP path, which caches, is checked against the
as-written arithmetic.

Third-party arithmetic not in the reference tree: timm 1.0.27 `VisionTransformer` / `PatchEmbed`
(uv.lock:2418), restated from its published definition: conv patchify (kernel = stride = patch),
learned absolute position embedding, pre-LN blocks (LayerNorm eps 1e-6, qkv bias, SDPA with scale
hd^-0.5, exact-erf GELU MLP), final LayerNorm, no class token.  `nn.MultiheadAttention` of the
decoder is torch itself (F.multi_head_attention_forward).

Pinning: oracle/pin_against_reference.py runs the reference's own `PARSeq` class (decoder, AR
loop, refinement, repetition stop unmodified; timm replaced by oracle/_refstubs._TimmViT) and
compares with `parseq_forward`; vectors in tests/golden/.  The ViT is *parity unpinned* vs timm.
"""

from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F


def make_cfg(patch=(4, 8), enc_dim=192, enc_heads=6, enc_depth=12, enc_mlp=4, dec_dim=192, dec_heads=6, dec_mlp=4,
             num_tokens=7121, max_label_length=100, refine_iters=1, img_size=(32, 800), decode_ar=1,
             repetition_stop=True, rep_period_max=8, rep_min_run_p1=8, rep_min_repeats=3):
    return SimpleNamespace(**locals())


PRESETS = {
    # configs/cfg_text_recognizer_parseq_tiny_dynw_v4.py:34-74
    "parseq-tiny-dynw-v4": dict(patch=(4, 8), enc_dim=192, enc_heads=6, dec_dim=192, dec_heads=6, num_tokens=7121),
    # configs/cfg_text_recognizer_parseq.py:14-53 (open-beta)
    "parseq": dict(patch=(8, 8), enc_dim=512, enc_heads=8, dec_dim=512, dec_heads=8, num_tokens=7312),
    # configs/cfg_text_recognizer_parseq_large_v4_1.py
    "parseq-large-v4_1": dict(patch=(8, 8), enc_dim=768, enc_heads=8, dec_dim=768, dec_heads=8, num_tokens=7121),
}


def _ln(sd, x, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def vit_encode(sd, cfg, x, prefix="encoder."):
    """models/layers/parseq_transformer.py:206-234 (Encoder.forward / forward_features_dynamic):
    both branches add the position rows of the columns present, so one code path serves."""
    D, H = cfg.enc_dim, cfg.enc_heads
    ph, pw = cfg.patch
    x = F.conv2d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"], stride=(ph, pw))
    B, _, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    full_gh, full_gw = cfg.img_size[0] // ph, cfg.img_size[1] // pw
    pos = sd[prefix + "pos_embed"].reshape(1, full_gh, full_gw, D)[:, :gh, :gw].reshape(1, gh * gw, D)
    x = x + pos
    hd = D // H
    for i in range(cfg.enc_depth):
        p = f"{prefix}blocks.{i}."
        y = _ln(sd, x, p + "norm1", 1e-6)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(B, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        a = a.transpose(1, 2).reshape(B, -1, D)
        x = x + F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        y = _ln(sd, x, p + "norm2", 1e-6)
        y = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        x = x + F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return _ln(sd, x, prefix + "norm", 1e-6)


def _mha(sd, prefix, heads, q, k, v, attn_mask=None, key_padding_mask=None):
    # nn.MultiheadAttention(batch_first=True), eval, need_weights default (parseq_transformer.py:83-92)
    out, _ = F.multi_head_attention_forward(
        q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), q.shape[-1], heads,
        sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"], None, None, False, 0.0,
        sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"], training=False,
        key_padding_mask=key_padding_mask, need_weights=True, attn_mask=attn_mask,
    )
    return out.transpose(0, 1)


def decode(sd, cfg, tgt, memory, tgt_mask=None, tgt_padding_mask=None, tgt_query=None, tgt_query_mask=None):
    """models/parseq.py:133-157 (PARSeq.decode) + parseq_transformer.py:101-169 for depth 1:
    only the query stream runs in the last (= only) layer (update_content=False)."""
    D = cfg.dec_dim
    N, L = tgt.shape
    emb = sd["text_embed.embedding.weight"]
    scale = math.sqrt(D)
    null_ctx = scale * emb[tgt[:, :1]]
    tgt_emb = sd["pos_queries"][:, : L - 1] + scale * emb[tgt[:, 1:]]
    content = torch.cat([null_ctx, tgt_emb], dim=1)
    if tgt_query is None:
        tgt_query = sd["pos_queries"][:, :L].expand(N, -1, -1)
    p = "decoder.layers.0."
    query = tgt_query
    qn = _ln(sd, query, p + "norm_q", 1e-5)
    cn = _ln(sd, content, p + "norm_c", 1e-5)
    query = query + _mha(sd, p + "self_attn.", cfg.dec_heads, qn, cn, cn, tgt_query_mask, tgt_padding_mask)
    query = query + _mha(sd, p + "cross_attn.", cfg.dec_heads, _ln(sd, query, p + "norm1", 1e-5), memory, memory)
    y = F.gelu(F.linear(_ln(sd, query, p + "norm2", 1e-5), sd[p + "linear1.weight"], sd[p + "linear1.bias"]))
    query = query + F.linear(y, sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return _ln(sd, query, "decoder.norm", 1e-5)


def detect_repeat_onset(seq, period_max, min_run_p1, min_repeats):
    """models/parseq.py:108-128."""
    n = len(seq)
    for p in range(1, period_max + 1):
        if n < 2 * p:
            continue
        unit = seq[n - p : n]
        k, t = 1, n - p
        while t - p >= 0 and seq[t - p : t] == unit:
            k += 1
            t -= p
        if k >= (min_run_p1 if p == 1 else min_repeats):
            return t, p
    return None


@torch.inference_mode()
def parseq_forward(sd, cfg, images, return_steps=False):
    """models/parseq.py:159-311 (testing path: max_length None, export_onnx False)."""
    eos_id, bos_id, pad_id = 0, cfg.num_tokens - 2, cfg.num_tokens - 1  # parseq_tokenizer.py:96-103
    bs = images.shape[0]
    num_steps = cfg.max_label_length + 1
    memory = vit_encode(sd, cfg, images)
    pos_queries = sd["pos_queries"][:, :num_steps].expand(bs, -1, -1)
    tgt_mask = torch.triu(torch.ones((num_steps, num_steps), dtype=torch.bool), 1)
    query_mask = tgt_mask.clone()
    head = lambda t: F.linear(t, sd["head.weight"], sd["head.bias"])  # noqa: E731
    rep_on = bool(cfg.repetition_stop)
    rep_cut = [None] * bs
    rep_done = [False] * bs
    tgt_in = torch.full((bs, num_steps), pad_id, dtype=torch.long)
    tgt_in[:, 0] = bos_id
    logits = []
    for i in range(num_steps):
        j = i + 1
        out = decode(sd, cfg, tgt_in[:, :j], memory, tgt_mask[:j, :j], tgt_query=pos_queries[:, i:j],
                     tgt_query_mask=query_mask[i:j, :j])
        p_i = head(out)
        logits.append(p_i)
        if j < num_steps:
            tgt_in[:, j] = p_i.squeeze().argmax(-1)
            if rep_on:
                for b in range(bs):
                    if rep_done[b] or int(tgt_in[b, j]) == eos_id:
                        continue
                    hit = detect_repeat_onset(tgt_in[b, 1 : j + 1].tolist(), cfg.rep_period_max, cfg.rep_min_run_p1,
                                              cfg.rep_min_repeats)
                    if hit is not None:
                        rep_cut[b] = hit[0] + hit[1]
                        rep_done[b] = True
                        tgt_in[b, j] = eos_id
            if (tgt_in == eos_id).any(dim=-1).all():
                break
    logits = torch.cat(logits, dim=1)
    steps = logits.shape[1]
    if cfg.refine_iters:
        # quirk Q1: integer (not boolean) indexing clears rows 0 and 1 only
        query_mask[torch.triu(torch.ones(num_steps, num_steps, dtype=torch.int64), 2)] = 0
        bos = torch.full((bs, 1), bos_id, dtype=torch.long)
        for _ in range(cfg.refine_iters):
            tgt_in = torch.cat([bos, logits[:, :-1].argmax(-1)], dim=1)
            pad_mask = (tgt_in == eos_id).int().cumsum(-1) > 0
            out = decode(sd, cfg, tgt_in, memory, tgt_mask, pad_mask, pos_queries, query_mask[:, : tgt_in.shape[1]])
            logits = head(out)
    if rep_on:
        for b, cut in enumerate(rep_cut):
            if cut is not None and cut < logits.shape[1]:
                logits[b, cut, :] = -30.0
                logits[b, cut, eos_id] = 30.0
    return (logits, steps) if return_steps else logits


def tokenizer_decode(probs, eos_id=0):
    """postprocessor/parseq_tokenizer.py:64-88,117-126 on softmaxed logits -> (id lists, scores)."""
    ids_all, scores = [], []
    for dist in probs:
        p, ids = dist.max(-1)
        ids = ids.tolist()
        try:
            e = ids.index(eos_id)
        except ValueError:
            e = len(ids)
        ids_all.append(ids[:e])
        scores.append(float(p[: e + 1].cpu().numpy().prod()))
    return ids_all, scores
