"""oracle/ — TEST INFRASTRUCTURE, NOT PRODUCT.

A CPU (PyTorch fp32 / numpy float64) restatement of the reference algorithm for the
UPGPT denoising hot path: UNetModel.forward x DDIMSampler loop -> VAE decode
(SURVEY.md §8a).  Every function cites the reference file:line it follows.

Who may import this package: tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg — and there only as the checker / the timed CPU baseline, never as
the thing shipped.  The product (upgpt_amd/) never imports it and has no CPU fallback.

Parity pinning: the reference has NO tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
imported in the build container from /root/reference (pure Python) by
tests/golden/make_goldens.py, and committed as fixtures under tests/golden/.
tests/test_oracle_golden.py checks the oracle against every one of them.
Third-party arithmetic on the path is PyTorch's own (F.conv2d, F.group_norm,
F.layer_norm, F.gelu, softmax) — semantics fixed by torch.

The CLIP encoders that produce the conditioning tensor (next row §8f-1, built in
upgpt_amd/clip_{text,image}.py) live in third-party packages the reference does not
vendor: transformers' CLIPTextModel (4.19.2 pinned by its environment.yaml) and OpenAI's
`clip` package.  There is no restatement of them here: the checker for those stages is
the transformers implementation itself, run on CPU fp32 by
tests/golden/make_clip_{text,image,textproj}_golden.py in the build container (transformers
5.15.0; CLIPVisionModelWithProjection is the same VisionTransformer as the clip
package's), with recipe weights and seeded inputs; only ids / seeds + outputs are
committed (tests/golden/clip_{text,image,textproj}.npz).  Their tokenizer / crop pre-processing
and trained weights are not available offline: parity is pinned on the towers from
token ids / pre-processed crops to embeddings.
"""
