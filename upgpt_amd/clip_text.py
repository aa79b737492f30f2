"""CLIP text transformer of the conditioning stage on the HIP kernels (first half).

The reference's `FrozenCLIPEmbedder` (ldm/modules/encoders/modules.py:137-162) is Hugging Face's
`CLIPTextModel("openai/clip-vit-large-patch14")` (transformers 4.19.2 in its environment.yaml — a third-party
dependency, not vendored): token + position embeddings -> 12 pre-LN layers (causal self-attention, 12 heads of 64;
MLP 768 -> 3072 -> 768 with quick_gelu) -> final LayerNorm; the embedder returns `last_hidden_state` [B, 77, 768].

* `CLIPTextTransformer` holds the weights under the checkpoint's key names (`text_model.embeddings.token_embedding.
  weight`, `text_model.encoder.layers.N.self_attn.q_proj.weight`, ...), so the reference's
  `cond_stage_model.transformer.*` entries load with `load_state_dict`.
* Compute: `upk_embed_tokens_f16`, LayerNorm folded into the fused q|k|v projection and into fc1
  (`upk_conv_desc.ln_colsum`), `upk_attention_causal_f16`, `UPK_F_QUICKGELU` epilogue, residual epilogues,
  `upk_layernorm_f16` for the final norm.  One program per batch size; no torch op computes anything.
* `FrozenCLIPTextEmbedder` (modules.py:164-198) is the OTHER text encoder of the reference — OpenAI's `clip` package
  model (`clip.model.CLIP.encode_text`: the same 12-layer tower under the package's own key names, then ln_final on
  the end-of-text row and `@ text_projection` -> [n, 768]); `InferenceModel.mix_style` (ldm/data/generate_utils.py:
  170-189) uses it to replace style-image embeddings by text.  Same program with the `OPENAI_NAMES` table plus
  `upk_gather_rows_f16` + a projection GEMM.
* `FrozenCLIPEmbedder` mirrors the reference class.  Its tokenizer (vocab / merges files of the hub model) is not
  available offline: `encode(text)` needs `transformers.CLIPTokenizer` files on disk, `encode_tokens(ids)` takes the
  int token ids directly (the oracle boundary for this stage).
"""
import os

import torch
from torch import nn

from . import _lib as L
from .engine import Act, Emitter, Packer, Program, _rup
from .packing import SharedPacks
from .params import ParamTree, weights_fingerprint

# openai/clip-vit-large-patch14 text tower (config.json of the hub model; CLIPTextConfig)
CLIP_L14_TEXT = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                     num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5)


def text_param_shapes(cfg):
    d, f = cfg["hidden_size"], cfg["intermediate_size"]
    s = {"text_model.embeddings.token_embedding.weight": (cfg["vocab_size"], d),
         "text_model.embeddings.position_embedding.weight": (cfg["max_position_embeddings"], d)}
    for i in range(cfg["num_hidden_layers"]):
        p = "text_model.encoder.layers.%d." % i
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + "self_attn.%s.weight" % proj] = (d, d)
            s[p + "self_attn.%s.bias" % proj] = (d,)
        for ln in ("layer_norm1", "layer_norm2"):
            s[p + ln + ".weight"] = (d,)
            s[p + ln + ".bias"] = (d,)
        s[p + "mlp.fc1.weight"], s[p + "mlp.fc1.bias"] = (f, d), (f,)
        s[p + "mlp.fc2.weight"], s[p + "mlp.fc2.bias"] = (d, f), (d,)
    s["text_model.final_layer_norm.weight"] = (d,)
    s["text_model.final_layer_norm.bias"] = (d,)
    return s


# weight names of the two layouts the tower is stored under
HF_NAMES = dict(tok="text_model.embeddings.token_embedding.weight", pos="text_model.embeddings.position_embedding.weight",
                layer="text_model.encoder.layers.%d.", qkv=["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"],
                ln1="layer_norm1", out="self_attn.out_proj", ln2="layer_norm2", fc1="mlp.fc1", fc2="mlp.fc2",
                final="text_model.final_layer_norm", proj=None)
OPENAI_NAMES = dict(tok="token_embedding.weight", pos="positional_embedding", layer="transformer.resblocks.%d.",
                    qkv="attn.in_proj", ln1="ln_1", out="attn.out_proj", ln2="ln_2", fc1="mlp.c_fc", fc2="mlp.c_proj",
                    final="ln_final", proj="text_projection")


def openai_text_param_shapes(cfg):
    """Text half of `clip.model.CLIP` (ViT-L/14: transformer_width 768, 12 layers, 12 heads, embed_dim 768)."""
    d, f = cfg["hidden_size"], cfg["intermediate_size"]
    s = {"token_embedding.weight": (cfg["vocab_size"], d), "positional_embedding": (cfg["max_position_embeddings"], d),
         "ln_final.weight": (d,), "ln_final.bias": (d,), "text_projection": (d, cfg.get("projection_dim", d))}
    for i in range(cfg["num_hidden_layers"]):
        b = "transformer.resblocks.%d." % i
        s[b + "attn.in_proj_weight"], s[b + "attn.in_proj_bias"] = (3 * d, d), (3 * d,)
        s[b + "attn.out_proj.weight"], s[b + "attn.out_proj.bias"] = (d, d), (d,)
        s[b + "ln_1.weight"], s[b + "ln_1.bias"] = (d,), (d,)
        s[b + "mlp.c_fc.weight"], s[b + "mlp.c_fc.bias"] = (f, d), (f,)
        s[b + "mlp.c_proj.weight"], s[b + "mlp.c_proj.bias"] = (d, f), (d,)
        s[b + "ln_2.weight"], s[b + "ln_2.bias"] = (d,), (d,)
    return s


class _TextPlan(Emitter):
    """Launch program of the text tower for one batch size."""

    def __init__(self, ctx, cfg, get, B, names=HF_NAMES, shared=None):
        super().__init__(ctx)
        self.cfg, self.B, self.pooled = cfg, B, names["proj"] is not None
        nm, raw = names, get
        if isinstance(nm["qkv"], str):  # OpenAI layout: packer-style "<name>.weight / .bias" on top of in_proj_weight
            def get(n):
                if n.endswith("in_proj.weight") or n.endswith("in_proj.bias"):
                    return raw(n.replace("in_proj.", "in_proj_"))
                if n == "proj_t.weight":
                    return raw(nm["proj"]).t().contiguous()
                return raw(n)
        d, f = cfg["hidden_size"], cfg["intermediate_size"]
        S, heads = cfg["max_position_embeddings"], cfg["num_attention_heads"]
        dh = d // heads
        if dh not in (32, 64, 128) or d % 32 or f % 32:
            raise NotImplementedError("CLIP text tower with head dim %d / width %d" % (dh, d))
        eps = float(cfg["layer_norm_eps"])
        pk = Packer(ctx, get)
        if shared is not None:  # (one packed tower for every lane's plan)
            pk = shared.wrap(pk)
        M = B * S
        self.ids = self.alloc(M, dtype=torch.int32)
        self.tok = get(nm["tok"]).half().contiguous()
        self.pos = get(nm["pos"]).half().contiguous()
        x = Act(self.alloc(M, d), B, S, 1, d)
        P = self.prog = Program(ctx)
        fn_e, h, chk = self.lib.upk_embed_tokens_f16, self.hctx, self._chk
        ae = (self.ids.data_ptr(), self.tok.data_ptr(), self.pos.data_ptr(), M, S, d, cfg["vocab_size"], x.t.data_ptr(), x.ld)
        P.add(lambda s: chk(fn_e(h, *ae, s)), self.ids, self.tok, self.pos, x, cls="other")
        vt_ld = _rup(S, 32)
        fn_a = self.lib.upk_attention_causal_f16
        for i in range(cfg["num_hidden_layers"]):
            p = nm["layer"] % i
            # first LayerNorm folded into the fused q|k|v projection; V leaves the GEMM transposed for the attention kernel
            qkv = [p + q for q in nm["qkv"]] if isinstance(nm["qkv"], list) else p + nm["qkv"]
            wqkv = pk.pack(qkv, n_out=2 * d, ln=p + nm["ln1"])
            qk = Act(self.alloc(M, 2 * d), B, S, 1, 2 * d)
            vt = self.alloc(B, heads, dh, vt_ld, zero=True)
            self.conv(P, x, wqkv, out=qk, ln_eps=eps,
                      vt=dict(t=vt, heads=heads, dhead=dh, ld=vt_ld, tokens=S, **{"from": 2 * d}))
            att = Act(self.alloc(M, d), B, S, 1, d)
            aa = (qk.t.data_ptr(), 2 * d, S * 2 * d, qk.t[:, d:].data_ptr(), 2 * d, S * 2 * d, vt.data_ptr(), vt_ld,
                  att.t.data_ptr(), d, S * d, B, heads, S, dh, float(dh ** -0.5))
            P.add(lambda s, aa=aa: chk(fn_a(h, *aa, s)), qk, vt, att, cls="attention")
            x = self.conv(P, att, pk.pack(p + nm["out"]), residual=x)
            # second LayerNorm folded into the first MLP GEMM; quick_gelu in its epilogue; the second adds the residual
            hmid = self.conv(P, x, pk.pack(p + nm["fc1"], ln=p + nm["ln2"]), ln_eps=eps, flags=L.F_QUICKGELU)
            x = self.conv(P, hmid, pk.pack(p + nm["fc2"]), residual=x)
        g, b_ = pk.vec(nm["final"] + ".weight"), pk.vec(nm["final"] + ".bias")
        fn_l = self.lib.upk_layernorm_f16
        if self.pooled:
            # clip.model.CLIP.encode_text: ln_final, the end-of-text row of each sequence, @ text_projection
            # (LayerNorm is per row, so the rows are gathered first)
            self.eot = self.alloc(B, dtype=torch.int32)
            rows = Act(self.alloc(B, d), B, 1, 1, d)
            fn_g = self.lib.upk_gather_rows_f16
            ag = (x.t.data_ptr(), x.ld, self.eot.data_ptr(), B, M, d, rows.t.data_ptr(), rows.ld)
            P.add(lambda s: chk(fn_g(h, *ag, s)), x, self.eot, rows, cls="other")
            normed = Act(self.alloc(B, d), B, 1, 1, d)
            al = (rows.t.data_ptr(), rows.ld, B, d, g.data_ptr(), b_.data_ptr(), eps, normed.t.data_ptr(), normed.ld)
            P.add(lambda s: chk(fn_l(h, *al, s)), rows, g, b_, normed, cls="layernorm")
            self.out_f32 = self.alloc(B, get("proj_t.weight").shape[0], dtype=torch.float32)
            self.conv(P, normed, pk.pack("proj_t", bias=False), out_f32=self.out_f32)
        else:
            self.out = Act(self.alloc(M, d), B, S, 1, d)
            al = (x.t.data_ptr(), x.ld, M, d, g.data_ptr(), b_.data_ptr(), eps, self.out.t.data_ptr(), self.out.ld)
            P.add(lambda s: chk(fn_l(h, *al, s)), x, g, b_, self.out, cls="layernorm")
        self.apply_tuning(tune_missing=os.environ.get("UPGPT_AUTOTUNE", "0") == "1")

    def run(self, ids):
        self.check(ids)
        self.upload(ids)
        return self.execute()

    def check(self, ids):
        if tuple(ids.shape) != (self.B, self.cfg["max_position_embeddings"]):
            raise ValueError("token ids must be [%d, %d], got %s" % (self.B, self.cfg["max_position_embeddings"],
                                                                     tuple(ids.shape)))
        if int(ids.min()) < 0 or int(ids.max()) >= self.cfg["vocab_size"]:
            raise ValueError("token id outside [0, %d)" % self.cfg["vocab_size"])

    def upload(self, ids):
        """The host -> device part (token ids, end-of-text rows): what _lib.host_io brackets; shapes checked by run()."""
        self.ids.copy_(ids.reshape(-1).to(self.dev, torch.int32))
        if self.pooled:
            # end-of-text = the highest token id of each sequence (clip/model.py encode_text: text.argmax(dim=-1))
            S = self.cfg["max_position_embeddings"]
            eot = ids.argmax(dim=-1).to(torch.int64).cpu() + torch.arange(self.B, dtype=torch.int64) * S
            self.eot.copy_(eot.to(torch.int32))

    def execute(self):
        if self.pooled:
            self.prog.run()
            return self.out_f32.clone()
        self.prog.run()
        return self.out.t.float().view(self.B, self.cfg["max_position_embeddings"], -1)


class _TextTower(ParamTree):
    """Weights of one storage layout + the per-batch-size launch programs."""
    NAMES = HF_NAMES

    def __init__(self, shapes, config):
        super().__init__()
        self.config = config
        self.add_params(shapes)
        self._plans = {}
        self._fp = None

    def _run(self, input_ids):
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("upgpt_amd.%s computes only through the HIP kernels on an MI355X: move it to 'cuda' first. "
                               "There is no CPU fallback." % type(self).__name__)
        from ._lib import PLAN_LOCK, concurrency, current_lane, get_context, host_io
        with PLAN_LOCK:  # (execution lanes: a plan — buffers and packed weights — per (batch, tuning table, lane))
            fp = weights_fingerprint(self)
            if fp != self._fp:
                self._plans, self._fp, self._packs = {}, fp, SharedPacks()
            B = int(input_ids.shape[0])
            key = (B, concurrency() > 1, current_lane())
            plan = self._plans.get(key)
            if plan is None:
                mine = [k for k in self._plans if k[-1] == key[-1]]
                if len(mine) >= 4:
                    self._plans.pop(mine[0])
                params = dict(self.named_parameters())
                with torch.cuda.device(p.device), host_io():
                    plan = self._plans[key] = _TextPlan(get_context(p.device), self.config, lambda n: params[n].data, B,
                                                        names=self.NAMES, shared=self._packs)
                    if self._packs.take_fresh():
                        torch.cuda.current_stream(p.device).synchronize()
        from ._lib import host_io
        with torch.cuda.device(p.device):
            plan.check(input_ids)
            with host_io():  # (token ids come from the host; the tower's eager launches stay under the lock too: clip_image.py)
                plan.upload(input_ids)
                return plan.execute()


class CLIPTextTransformer(_TextTower):
    """Weights + forward of the CLIP text tower; `forward(input_ids)` returns an object with `.last_hidden_state`
    like transformers' CLIPTextModel (that is all the reference reads, modules.py:156-158)."""

    def __init__(self, **config):
        cfg = dict(CLIP_L14_TEXT, **config)
        super().__init__(text_param_shapes(cfg), cfg)

    class Output:
        def __init__(self, last_hidden_state):
            self.last_hidden_state = last_hidden_state

    def forward(self, input_ids):
        return CLIPTextTransformer.Output(self._run(input_ids))


class CLIPTextTower(_TextTower):
    """Text half of OpenAI's `clip.model.CLIP` under the package's key names; `encode_text(tokens)` -> [n, 768]
    (clip/model.py CLIP.encode_text; the image half lives in upgpt_amd/clip_image.py)."""
    NAMES = OPENAI_NAMES

    def __init__(self, **config):
        cfg = dict(CLIP_L14_TEXT, **config)
        super().__init__(openai_text_param_shapes(cfg), cfg)

    def encode_text(self, tokens):
        return self._run(tokens)

    forward = encode_text


class FrozenCLIPEmbedder(nn.Module):
    """Drop-in for ldm.modules.encoders.modules.FrozenCLIPEmbedder (modules.py:137-162): `transformer` is the
    HIP-kernel text tower (state-dict keys `transformer.text_model.*` as in the reference's checkpoints)."""

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, tokenizer=None, **config):
        super().__init__()
        self.version, self.device, self.max_length = version, device, max_length
        self.transformer = CLIPTextTransformer(**config)
        self.tokenizer = tokenizer
        self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    def _tokenizer(self):
        if self.tokenizer is None:
            try:
                from transformers import CLIPTokenizer
                tok = CLIPTokenizer.from_pretrained(self.version, local_files_only=True)
                if len(tok) < self.transformer.config["vocab_size"] - 2:
                    # (transformers 5.x hands back an EMPTY tokenizer instead of failing when the files are missing)
                    raise FileNotFoundError("tokenizer of %r has %d entries" % (self.version, len(tok)))
                self.tokenizer = tok
            except Exception as e:  # no hub access: the vocabulary files have to be on disk
                raise RuntimeError(
                    "FrozenCLIPEmbedder needs the CLIP tokenizer files of %r on disk (no network): pass tokenizer=..., "
                    "call encode_tokens(input_ids) with [B, %d] token ids, or feed precomputed embeddings through "
                    "DummyModel as the reference's InferenceModel does" % (self.version, self.max_length)) from e
        return self.tokenizer

    def encode_tokens(self, input_ids):
        return self.transformer(input_ids=input_ids).last_hidden_state

    def forward(self, text):
        enc = self._tokenizer()(text, truncation=True, max_length=self.max_length, return_length=True,
                                return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return self.encode_tokens(enc["input_ids"])

    def encode(self, text):
        return self(text)


class FrozenCLIPTextEmbedder(nn.Module):
    """Drop-in for ldm.modules.encoders.modules.FrozenCLIPTextEmbedder (modules.py:164-198): `model` is the text half of
    the `clip` package's CLIP (state-dict keys `model.token_embedding.weight`, `model.transformer.resblocks.N...`,
    `model.ln_final.*`, `model.text_projection` as `clip.load` produces them).  `forward(texts)`: `texts` is a list whose
    items are each a string or a list of strings; every item yields [n, 768] and the results are stacked.

    `clip.tokenize` needs the package's BPE vocabulary, which is not available offline: pass `tokenizer=` (a callable
    strings -> int tensor [n, 77]) or call `encode_tokens(ids)`.  The optional L2 normalisation of the [n, 768] result is
    host-side post-processing (one torch expression on a few KB)."""

    def __init__(self, version="ViT-L/14", device="cuda", max_length=77, n_repeat=1, normalize=True, tokenizer=None,
                 **config):
        super().__init__()
        if version != "ViT-L/14" and not config:
            raise NotImplementedError("only the ViT-L/14 text tower of the reference is described here; pass the tower "
                                      "dimensions as keyword arguments for another one")
        self.model = CLIPTextTower(max_position_embeddings=max_length, **config)
        self.device, self.max_length, self.n_repeat, self.normalize = device, max_length, n_repeat, normalize
        self.tokenizer = tokenizer
        self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for param in self.parameters():
            param.requires_grad = False

    def _tokenize(self, text):
        if self.tokenizer is None:
            try:
                import clip  # the OpenAI package, if the user has it
                self.tokenizer = clip.tokenize
            except Exception as e:
                raise RuntimeError("FrozenCLIPTextEmbedder needs the `clip` package's tokenizer (BPE vocabulary, not "
                                   "available offline): pass tokenizer=..., or call encode_tokens(input_ids) with [n, %d] "
                                   "token ids" % self.max_length) from e
        return self.tokenizer(text)

    @torch.no_grad()
    def encode_tokens(self, tokens):
        z = self.model.encode_text(tokens)
        if self.normalize:
            z = z / torch.linalg.norm(z, dim=1, keepdim=True)
        return z

    @torch.no_grad()
    def forward(self, texts):
        return torch.stack([self.encode_tokens(self._tokenize(text)) for text in texts])

    def encode(self, text):
        z = self(text)
        if z.ndim == 2:
            z = z[:, None, :].repeat(1, self.n_repeat, 1)
        return z
