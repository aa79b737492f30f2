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
* `FrozenCLIPEmbedder` mirrors the reference class.  Its tokenizer (vocab / merges files of the hub model) is not
  available offline: `encode(text)` needs `transformers.CLIPTokenizer` files on disk, `encode_tokens(ids)` takes the
  int token ids directly (the oracle boundary for this stage).
"""
import os

import torch
from torch import nn

from . import _lib as L
from .engine import Act, Emitter, Packer, Program, _rup
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


class _TextPlan(Emitter):
    """Launch program of the text tower for one batch size."""

    def __init__(self, ctx, cfg, get, B):
        super().__init__(ctx)
        self.cfg, self.B = cfg, B
        d, f = cfg["hidden_size"], cfg["intermediate_size"]
        S, heads = cfg["max_position_embeddings"], cfg["num_attention_heads"]
        dh = d // heads
        if dh not in (32, 64, 128) or d % 32 or f % 32:
            raise NotImplementedError("CLIP text tower with head dim %d / width %d" % (dh, d))
        eps = float(cfg["layer_norm_eps"])
        pk = Packer(ctx, get)
        M = B * S
        self.ids = self.alloc(M, dtype=torch.int32)
        self.tok = get("text_model.embeddings.token_embedding.weight").half().contiguous()
        self.pos = get("text_model.embeddings.position_embedding.weight").half().contiguous()
        x = Act(self.alloc(M, d), B, S, 1, d)
        P = self.prog = Program(ctx)
        fn_e, h, chk = self.lib.upk_embed_tokens_f16, self.hctx, self._chk
        ae = (self.ids.data_ptr(), self.tok.data_ptr(), self.pos.data_ptr(), M, S, d, cfg["vocab_size"], x.t.data_ptr(), x.ld)
        P.add(lambda s: chk(fn_e(h, *ae, s)), self.ids, self.tok, self.pos, x, cls="other")
        vt_ld = _rup(S, 32)
        fn_a = self.lib.upk_attention_causal_f16
        for i in range(cfg["num_hidden_layers"]):
            p = "text_model.encoder.layers.%d." % i
            # layer_norm1 folded into the fused q|k|v projection; V leaves the GEMM transposed for the attention kernel
            wqkv = pk.pack([p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"], n_out=2 * d,
                           ln=p + "layer_norm1")
            qk = Act(self.alloc(M, 2 * d), B, S, 1, 2 * d)
            vt = self.alloc(B, heads, dh, vt_ld, zero=True)
            self.conv(P, x, wqkv, out=qk, ln_eps=eps,
                      vt=dict(t=vt, heads=heads, dhead=dh, ld=vt_ld, tokens=S, **{"from": 2 * d}))
            att = Act(self.alloc(M, d), B, S, 1, d)
            aa = (qk.t.data_ptr(), 2 * d, S * 2 * d, qk.t[:, d:].data_ptr(), 2 * d, S * 2 * d, vt.data_ptr(), vt_ld,
                  att.t.data_ptr(), d, S * d, B, heads, S, dh, float(dh ** -0.5))
            P.add(lambda s, aa=aa: chk(fn_a(h, *aa, s)), qk, vt, att, cls="attention")
            x = self.conv(P, att, pk.pack(p + "self_attn.out_proj"), residual=x)
            # layer_norm2 folded into fc1; quick_gelu in its epilogue; fc2 adds the residual
            hmid = self.conv(P, x, pk.pack(p + "mlp.fc1", ln=p + "layer_norm2"), ln_eps=eps, flags=L.F_QUICKGELU)
            x = self.conv(P, hmid, pk.pack(p + "mlp.fc2"), residual=x)
        self.out = Act(self.alloc(M, d), B, S, 1, d)
        g, b_ = pk.vec("text_model.final_layer_norm.weight"), pk.vec("text_model.final_layer_norm.bias")
        fn_l = self.lib.upk_layernorm_f16
        al = (x.t.data_ptr(), x.ld, M, d, g.data_ptr(), b_.data_ptr(), eps, self.out.t.data_ptr(), self.out.ld)
        P.add(lambda s: chk(fn_l(h, *al, s)), x, g, b_, self.out, cls="layernorm")
        self.apply_tuning(tune_missing=os.environ.get("UPGPT_AUTOTUNE", "0") == "1")

    def run(self, ids):
        if tuple(ids.shape) != (self.B, self.cfg["max_position_embeddings"]):
            raise ValueError("token ids must be [%d, %d], got %s" % (self.B, self.cfg["max_position_embeddings"],
                                                                     tuple(ids.shape)))
        if int(ids.min()) < 0 or int(ids.max()) >= self.cfg["vocab_size"]:
            raise ValueError("token id outside [0, %d)" % self.cfg["vocab_size"])
        self.ids.copy_(ids.reshape(-1).to(self.dev, torch.int32))
        self.prog.run()
        return self.out.t.float().view(self.B, self.cfg["max_position_embeddings"], -1)


class CLIPTextTransformer(ParamTree):
    """Weights + forward of the CLIP text tower; `forward(input_ids)` returns an object with `.last_hidden_state`
    like transformers' CLIPTextModel (that is all the reference reads, modules.py:156-158)."""

    def __init__(self, **config):
        super().__init__()
        self.config = dict(CLIP_L14_TEXT, **config)
        self.add_params(text_param_shapes(self.config))
        self._plans = {}
        self._fp = None

    class Output:
        def __init__(self, last_hidden_state):
            self.last_hidden_state = last_hidden_state

    def forward(self, input_ids):
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("upgpt_amd.CLIPTextTransformer computes only through the HIP kernels on an MI355X: move it "
                               "to 'cuda' first. There is no CPU fallback.")
        from ._lib import get_context
        fp = weights_fingerprint(self)
        if fp != self._fp:
            self._plans, self._fp = {}, fp
        B = int(input_ids.shape[0])
        plan = self._plans.get(B)
        if plan is None:
            if len(self._plans) >= 4:
                self._plans.pop(next(iter(self._plans)))
            params = dict(self.named_parameters())
            with torch.cuda.device(p.device):
                plan = self._plans[B] = _TextPlan(get_context(p.device), self.config, lambda n: params[n].data, B)
        with torch.cuda.device(p.device):
            return CLIPTextTransformer.Output(plan.run(input_ids))


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
