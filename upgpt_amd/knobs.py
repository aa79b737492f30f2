"""Environment switches of the lowering (read at import; tests and tuning scripts flip the attributes of THIS module)."""
import os

# GroupNorm statistics of the VAE decoder's tensors as conv by-products (gn_stats_cap + upk_groupnorm_finalize_f32):
# measured neutral (8.29 vs 8.29 ms per decode) — the statistics pass runs at 5.6 TB/s since round 2, the channel
# partials cost the 200-us convs 2-3 % and a 32-block fold per apply workgroup more than the pass it replaces — off
VAE_GN_BYPRODUCT = os.environ.get("UPGPT_VAE_GN_BYPRODUCT", "0")  # "0" | "1" | "auto" (only tensors of >= VAE_GN_MINM rows)
VAE_GN_MINM = int(os.environ.get("UPGPT_VAE_GN_MINM", "131072"))


def vae_gn_byproduct(M):
    """Whether the producer conv of an M-row VAE tensor leaves the GroupNorm statistics (round 6: "auto" arms only the two
    highest-resolution levels, where the statistics pass it replaces costs 13-51 us against 2-3 % of a 200-us conv)."""
    return VAE_GN_BYPRODUCT == "1" or (VAE_GN_BYPRODUCT == "auto" and M >= VAE_GN_MINM)
UPS_PHASES = os.environ.get("UPGPT_UPS_PHASES", "1") == "1"
LN_ROWS = os.environ.get("UPGPT_LN_ROWS", "1") == "1"
QPROJ_FUSE = os.environ.get("UPGPT_QPROJ_FUSE", "1") == "1"
GN_REDUCE_APPLY = os.environ.get("UPGPT_GN_REDUCE_APPLY", "1") == "1"
# fused feed-forward tail (csrc/mlp.hip: GEGLU -> ff.net.2 o proj_out with the hidden activation in LDS): "auto" = where
# M / rows-per-workgroup covers the chip (the 32x32 level at B = 8), "0" off, "1" wherever the kernel takes the shape
MLP_FUSE = os.environ.get("UPGPT_MLP_FUSE", "auto")
MLP_ROWS = int(os.environ.get("UPGPT_MLP_ROWS", "0"))  # rows per workgroup (32 / 64; 0 = by M)
# fused cross-attention half of a transformer block (csrc/xblock.hip: attn1.to_out -> norm2 -> to_q -> attention over the
# context -> attn2.to_out, one launch instead of three / four): "auto", "0" off, "1" wherever the kernel takes the shape
XBLOCK = os.environ.get("UPGPT_XBLOCK", "auto")
XB_ROWS = int(os.environ.get("UPGPT_XB_ROWS", "0"))  # rows per workgroup (16 / 32; 0 = by M)
# fused head of a SpatialTransformer (csrc/xblock.hip hblock_kernel: proj_in -> norm1 -> q | k | v, one launch instead of two)
HBLOCK = os.environ.get("UPGPT_HBLOCK", "auto")
HBLOCK_GN = os.environ.get("UPGPT_HBLOCK_GN", "1") == "1"  # SpatialTransformer.norm applied on the tile inside that launch
# next-weight prefetch (include/upk.h pf_next, Emitter.link_weight_prefetch): "auto" = while one batch has the chip to itself
# (replayed forward 2.83 -> 2.76 ms, serial bench 53.28 -> 53.61 images/s in same-box pairs; with four forwards in flight
# 1.461 -> 1.471 ms per forward: off there), "0" off, "1" always
WEIGHT_PREFETCH = os.environ.get("UPGPT_WEIGHT_PREFETCH", "auto")
WEIGHT_PREFETCH_AHEAD = int(os.environ.get("UPGPT_WEIGHT_PREFETCH_AHEAD", "1"))  # which following launch's weight: 1 = the next
WEIGHT_PREFETCH_MAX = int(os.environ.get("UPGPT_WEIGHT_PREFETCH_MAX", str(32 << 20)))  # bytes of the next weight at most
LN_LAUNCH_US = 5.0  # what a separate LayerNorm launch costs inside the replayed forward (3.8 us of kernel + its boundary: DESIGN.md 11i / 11j)
