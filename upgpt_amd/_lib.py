"""ctypes binding of libupk.so (include/upk.h).

The product path has NO CPU fallback: if the HIP library is missing, was not
built for gfx950, or no MI355X is visible, the calls raise.  PyTorch is used
only for device memory, streams and (elsewhere) torch.distributed.
"""
import contextlib
import ctypes as C
import os
import threading

import torch
from ._check import require

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UPK_LIB") or os.path.join(_HERE, "libupk.so")  # (UPK_LIB: dev builds)

# every symbol include/upk.h declares (tests check the library exports all of them)
SYMBOLS = [
    "upk_version", "upk_create", "upk_destroy", "upk_last_error", "upk_set_workspace", "upk_num_cus",
    "upk_pack_weight_f16", "upk_packed_weight_bytes", "upk_conv2d_nhwc_f16", "upk_gemm_f16",
    "upk_conv_autotune", "upk_conv_override", "upk_conv_num_configs", "upk_conv_config_name", "upk_conv_gn_fused",
    "upk_conv_ln_rows",
    "upk_geglu_mlp_f16", "upk_geglu_mlp_supported", "upk_cross_block_f16", "upk_cross_block_supported", "upk_head_block_f16", "upk_head_block_supported",
    "upk_attention_f16", "upk_groupnorm_nhwc_f16", "upk_groupnorm_stats_nhwc_f16", "upk_groupnorm_chunks", "upk_groupnorm_apply_nhwc_f16", "upk_groupnorm_finalize_f32", "upk_groupnorm_ws_bytes",
    "upk_layernorm_f16", "upk_timestep_embed_f16",
    "upk_nchw_f32_to_nhwc_f16", "upk_nhwc_f16_to_nchw_f32", "upk_f32_to_f16", "upk_ddim_step_f32",
    "upk_ddim_step_cfg_f32", "upk_plms_step_f32", "upk_attention_causal_f16", "upk_attention_qproj_f16", "upk_embed_tokens_f16",
    "upk_patchify_nchw_f32_f16", "upk_vit_assemble_f16", "upk_gather_rows_f16",
    "upk_advance_step", "upk_step_autoadvance", "upk_kernel_launches", "upk_graph_begin", "upk_graph_end", "upk_graph_launch", "upk_graph_destroy",
    "upk_prof_enable", "upk_prof_collect",
    "upk_stream_create_cumask", "upk_stream_destroy", "upk_probe_placement", "upk_probe_clock",
]

F_SILU, F_GEGLU, F_OUT_F32, F_OUT_NCHW_F32, F_UPSAMPLE2X, F_PAD_ASYM = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
F_QUICKGELU = 0x40
NUM_CLASSES = 5
CLASS_NAMES = ["igemm", "attention", "groupnorm", "layernorm", "other"]

E_NAMES = {0: "UPK_OK", -1: "UPK_EINVAL", -2: "UPK_ESHAPE", -3: "UPK_EWORKSPACE", -4: "UPK_EHIP", -5: "UPK_ENODEV"}


class UpkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (E_NAMES.get(code, str(code)), msg))
        self.code = code


class ConvDesc(C.Structure):
    """Mirror of struct upk_conv_desc (include/upk.h)."""
    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p),
        ("c1", C.c_int32), ("c2", C.c_int32), ("ld1", C.c_int32), ("ld2", C.c_int32),
        ("batch", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32),
        ("w_packed", C.c_void_p), ("n_out", C.c_int32), ("n_pad", C.c_int32),
        ("bias", C.c_void_p), ("residual", C.c_void_p), ("ld_res", C.c_int32),
        ("rowvec", C.c_void_p), ("rv_batch_stride", C.c_int32), ("rv_step_stride", C.c_int32),
        ("step", C.c_void_p), ("y", C.c_void_p), ("ldy", C.c_int32),
        ("vt", C.c_void_p), ("vt_from", C.c_int32), ("vt_heads", C.c_int32), ("vt_dhead", C.c_int32),
        ("vt_ld", C.c_int32), ("vt_tokens", C.c_int32), ("flags", C.c_int32),
        ("tune_cfg", C.c_int32), ("tune_splitk", C.c_int32),
        ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float), ("ln_dim", C.c_int32),
        ("gn_stats_ws", C.c_void_p), ("gn_groups", C.c_int32),
        ("x3", C.c_void_p), ("x4", C.c_void_p),
        ("c3", C.c_int32), ("c4", C.c_int32), ("ld3", C.c_int32), ("ld4", C.c_int32),
        ("gno_gamma", C.c_void_p), ("gno_beta", C.c_void_p), ("gno_y", C.c_void_p), ("gno_eps", C.c_float),
        ("gno_silu", C.c_int32), ("gno_ld", C.c_int32), ("gno_skip_y", C.c_int32),
        ("ln_rows_out", C.c_void_p), ("ln_rows_in", C.c_void_p), ("ln_rows_slots", C.c_int32),
        ("w_phase", C.c_void_p), ("gn_stats_cap", C.c_int32),
        ("pf_next", C.c_void_p), ("pf_bytes", C.c_int64),
    ]


class MlpDesc(C.Structure):
    """Mirror of struct upk_mlp_desc (include/upk.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32), ("m", C.c_int32), ("c", C.c_int32), ("inner", C.c_int32),
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("u1", C.c_void_p), ("ln_eps", C.c_float), ("ln_dim", C.c_int32),
        ("w2", C.c_void_p), ("b2", C.c_void_p), ("n_out", C.c_int32), ("n_pad", C.c_int32),
        ("residual", C.c_void_p), ("ld_res", C.c_int32), ("y", C.c_void_p), ("ldy", C.c_int32),
        ("gn_stats_ws", C.c_void_p), ("hw", C.c_int32), ("rows_per_wg", C.c_int32),
    ]


class XblockDesc(C.Structure):
    """Mirror of struct upk_xblock_desc (include/upk.h)."""
    _fields_ = [
        ("a1", C.c_void_p), ("lda", C.c_int32), ("m", C.c_int32), ("c", C.c_int32), ("heads", C.c_int32), ("d", C.c_int32),
        ("t0", C.c_void_p), ("ld_t0", C.c_int32),
        ("w_out1", C.c_void_p), ("w_q", C.c_void_p), ("w_out2", C.c_void_p), ("vec", C.c_void_p),
        ("ln_eps", C.c_float), ("ln_dim", C.c_int32),
        ("k_ctx", C.c_void_p), ("ldk", C.c_int32), ("n_kv", C.c_int32),
        ("vt_ctx", C.c_void_p), ("vt_ld", C.c_int32), ("scale", C.c_float),
        ("y", C.c_void_p), ("ldy", C.c_int32), ("hw", C.c_int32), ("rows_per_wg", C.c_int32),
    ]


class HblockDesc(C.Structure):
    """Mirror of struct upk_hblock_desc (include/upk.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int32), ("m", C.c_int32), ("c", C.c_int32), ("heads", C.c_int32), ("d", C.c_int32),
        ("w_in", C.c_void_p), ("w_qkv", C.c_void_p), ("vec", C.c_void_p), ("ln_eps", C.c_float), ("ln_dim", C.c_int32),
        ("t0", C.c_void_p), ("ld_t0", C.c_int32), ("qk", C.c_void_p), ("ld_qk", C.c_int32),
        ("vt", C.c_void_p), ("vt_ld", C.c_int32), ("hw", C.c_int32), ("rows_per_wg", C.c_int32),
        ("gn_part", C.c_void_p), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p),
        ("gn_nblk", C.c_int32), ("gn_ld", C.c_int32), ("gn_groups", C.c_int32), ("gn_eps", C.c_float),
    ]


_lib = None
_lib_lock = threading.Lock()


def load_library(path=None):
    """dlopen libupk.so and declare prototypes. Raises if it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise RuntimeError(
                "libupk.so not found at %s: build it with `python -m upgpt_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
        lib = C.CDLL(p)
        vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float
        protos = {
            "upk_version": (C.c_int, []),
            "upk_create": (C.c_int, [C.POINTER(vp), i32]),
            "upk_destroy": (C.c_int, [vp]),
            "upk_last_error": (C.c_char_p, [vp]),
            "upk_set_workspace": (C.c_int, [vp, vp, C.c_size_t]),
            "upk_num_cus": (C.c_int, [vp]),
            "upk_pack_weight_f16": (C.c_int, [vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, vp]),
            "upk_packed_weight_bytes": (C.c_size_t, [i32, i32, i32, i32]),
            "upk_conv2d_nhwc_f16": (C.c_int, [vp, C.POINTER(ConvDesc), vp]),
            "upk_gemm_f16": (C.c_int, [vp, vp, i32, i32, i32, vp, i32, i32, vp, vp, i32, vp, i32, i32, vp]),
            "upk_conv_autotune": (C.c_int, [vp, C.POINTER(ConvDesc), vp, i32, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(C.c_float), C.POINTER(C.c_float)]),
            "upk_conv_override": (C.c_int, [vp, i32, i32]),
            "upk_conv_num_configs": (C.c_int, []),
            "upk_conv_config_name": (C.c_char_p, [i32]),
            "upk_conv_ln_rows": (C.c_int, [vp, C.POINTER(ConvDesc), C.POINTER(C.c_int)]),
            "upk_geglu_mlp_f16": (C.c_int, [vp, C.POINTER(MlpDesc), vp]),
            "upk_geglu_mlp_supported": (C.c_int, [vp, C.POINTER(MlpDesc)]),
            "upk_cross_block_f16": (C.c_int, [vp, C.POINTER(XblockDesc), vp]),
            "upk_cross_block_supported": (C.c_int, [vp, C.POINTER(XblockDesc)]),
            "upk_head_block_f16": (C.c_int, [vp, C.POINTER(HblockDesc), vp]),
            "upk_head_block_supported": (C.c_int, [vp, C.POINTER(HblockDesc)]),
            "upk_attention_f16": (C.c_int, [vp, vp, i32, i64, vp, i32, i64, vp, i32, vp, i32, i64,
                                            i32, i32, i32, i32, i32, f32, vp]),
            "upk_attention_causal_f16": (C.c_int, [vp, vp, i32, i64, vp, i32, i64, vp, i32, vp, i32, i64,
                                                   i32, i32, i32, i32, f32, vp]),
            "upk_attention_qproj_f16": (C.c_int, [vp, vp, i32, i64, i32, i32, f32, vp, vp, vp, vp, i32, i64, vp, i32, vp,
                                                  i32, i64, i32, i32, i32, i32, i32, f32, vp]),
            "upk_embed_tokens_f16": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
            "upk_gather_rows_f16": (C.c_int, [vp, vp, i32, vp, i32, i32, i32, vp, i32, vp]),
            "upk_patchify_nchw_f32_f16": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
            "upk_vit_assemble_f16": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, i32, vp, i32, vp]),
            "upk_groupnorm_nhwc_f16": (C.c_int, [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp,
                                                 f32, i32, vp, i32, vp, vp]),
            "upk_groupnorm_stats_nhwc_f16": (C.c_int, [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp]),
            "upk_groupnorm_chunks": (C.c_int, [i32]),
            "upk_groupnorm_apply_nhwc_f16": (C.c_int, [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp,
                                                       f32, i32, vp, i32, vp, i32, i32, i32, vp, i32, i32, vp]),
            "upk_conv_gn_fused": (C.c_int, [vp, C.POINTER(ConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
            "upk_groupnorm_finalize_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
            "upk_groupnorm_ws_bytes": (C.c_size_t, [i32, i32]),
            "upk_layernorm_f16": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, f32, vp, i32, vp]),
            "upk_timestep_embed_f16": (C.c_int, [vp, vp, i32, i32, f32, vp, i32, vp]),
            "upk_nchw_f32_to_nhwc_f16": (C.c_int, [vp, vp, i32, i32, i32, vp, i32, i32, i32, f32, vp]),
            "upk_nhwc_f16_to_nchw_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, vp, vp]),
            "upk_f32_to_f16": (C.c_int, [vp, vp, i32, i32, vp, i32, vp]),
            "upk_ddim_step_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
            "upk_ddim_step_cfg_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
            "upk_plms_step_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
            "upk_advance_step": (C.c_int, [vp, vp, vp]),
            "upk_step_autoadvance": (C.c_int, [vp, vp]),
            "upk_kernel_launches": (C.c_longlong, [vp, i32]),
            "upk_graph_begin": (C.c_int, [vp, vp]),
            "upk_graph_end": (C.c_int, [vp, vp, C.POINTER(vp)]),
            "upk_graph_launch": (C.c_int, [vp, vp, vp]),
            "upk_graph_destroy": (C.c_int, [vp, vp]),
            "upk_prof_enable": (C.c_int, [vp, i32]),
            "upk_prof_collect": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
            "upk_stream_create_cumask": (C.c_int, [vp, C.POINTER(C.c_uint32), i32, C.POINTER(vp)]),
            "upk_stream_destroy": (C.c_int, [vp, vp]),
            "upk_probe_placement": (C.c_int, [vp, vp, i32, i32, vp]),
            "upk_probe_clock": (C.c_int, [vp, vp, i64, vp]),
        }
        require(sorted(protos) == sorted(SYMBOLS), "ctypes prototypes and SYMBOLS differ", RuntimeError)
        for name, (res, args) in protos.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        if path is None:
            _lib = lib
        return lib


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


class Context:
    """One upk_ctx bound to one HIP device + the stream launches go to."""

    def __init__(self, device=None, workspace_bytes=256 << 20):
        if not torch.cuda.is_available():
            raise RuntimeError("upgpt_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                               "and there is no CPU fallback for the HIP path")
        self.lib = load_library()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   (device.index if isinstance(device, torch.device) else int(device)))
        h = C.c_void_p()
        rc = self.lib.upk_create(C.byref(h), self.device.index)
        if rc != 0:
            raise UpkError(rc, "upk_create(device=%d) failed (need a gfx950 device)" % self.device.index)
        self.h = h
        self.stream = None  # None => torch current stream at call time
        self.workspace = torch.empty(workspace_bytes, dtype=torch.uint8, device=self.device)
        self._chk(self.lib.upk_set_workspace(self.h, self.workspace.data_ptr(), workspace_bytes))
        self.num_cus = self.lib.upk_num_cus(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.upk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers
    def _chk(self, rc):
        if rc != 0:
            raise UpkError(rc, (self.lib.upk_last_error(self.h) or b"").decode())

    def _s(self):
        s = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        return s.cuda_stream

    # -- raw launchers (tensor arguments are torch CUDA tensors or None)
    def conv(self, desc):
        self._chk(self.lib.upk_conv2d_nhwc_f16(self.h, C.byref(desc), self._s()))

    def gemm(self, a, lda, m, k, w, n_out, n_pad, bias, res, ld_res, y, ldy, flags=0):
        self._chk(self.lib.upk_gemm_f16(self.h, _ptr(a), lda, m, k, _ptr(w), n_out, n_pad, _ptr(bias), _ptr(res),
                                        ld_res, _ptr(y), ldy, flags, self._s()))

    def conv_autotune(self, desc, reps=5):
        """-> (cfg, splitk, best_us, default_us); synchronises the stream."""
        cfg, sk, bu, du = C.c_int(), C.c_int(), C.c_float(), C.c_float()
        self._chk(self.lib.upk_conv_autotune(self.h, C.byref(desc), self._s(), reps, C.byref(cfg), C.byref(sk),
                                             C.byref(bu), C.byref(du)))
        return cfg.value, sk.value, bu.value, du.value

    def conv_override(self, cfg=-1, splitk=0):
        self._chk(self.lib.upk_conv_override(self.h, cfg, splitk))

    def attention(self, q, ldq, qbs, k, ldk, kbs, vt, vt_ld, out, ldo, obs, batch, heads, n_q, n_kv, d, scale):
        self._chk(self.lib.upk_attention_f16(self.h, _ptr(q), ldq, qbs, _ptr(k), ldk, kbs, _ptr(vt), vt_ld,
                                             _ptr(out), ldo, obs, batch, heads, n_q, n_kv, d, scale, self._s()))

    def attention_causal(self, q, ldq, qbs, k, ldk, kbs, vt, vt_ld, out, ldo, obs, batch, heads, n, d, scale):
        self._chk(self.lib.upk_attention_causal_f16(self.h, _ptr(q), ldq, qbs, _ptr(k), ldk, kbs, _ptr(vt), vt_ld,
                                                    _ptr(out), ldo, obs, batch, heads, n, d, scale, self._s()))

    def embed_tokens(self, ids, tok, pos, rows, seq, dim, vocab, out, ld):
        self._chk(self.lib.upk_embed_tokens_f16(self.h, _ptr(ids), _ptr(tok), _ptr(pos), rows, seq, dim, vocab, _ptr(out),
                                                ld, self._s()))

    def gather_rows(self, x, ldx, idx, n, n_src, dim, y, ldy):
        self._chk(self.lib.upk_gather_rows_f16(self.h, _ptr(x), ldx, _ptr(idx), n, n_src, dim, _ptr(y), ldy, self._s()))

    def groupnorm(self, x1, c1, ld1, x2, c2, ld2, batch, hw, groups, gamma, beta, eps, silu, y, ldy, ws):
        self._chk(self.lib.upk_groupnorm_nhwc_f16(self.h, _ptr(x1), c1, ld1, _ptr(x2), c2, ld2, batch, hw, groups,
                                                  _ptr(gamma), _ptr(beta), eps, int(silu), _ptr(y), ldy, _ptr(ws),
                                                  self._s()))

    def conv_gn_fused(self, desc):
        """(mode, nblk) of the GroupNorm by-product the launch of `desc` will leave (include/upk.h)."""
        f, n = C.c_int(0), C.c_int(0)
        self._chk(self.lib.upk_conv_gn_fused(self.h, C.byref(desc), C.byref(f), C.byref(n)))
        return f.value, n.value

    def gn_stats_floats(self, batch, n_pad, cap=32):
        """Size (floats) of a upk_conv_desc.gn_stats_ws buffer (cap = upk_conv_desc.gn_stats_cap row blocks)."""
        return batch * max(32, cap) * 2 * max(n_pad, 32)

    def groupnorm_ws_bytes(self, batch, hw):
        return self.lib.upk_groupnorm_ws_bytes(batch, hw)

    def layernorm(self, x, ldx, rows, d, gamma, beta, eps, y, ldy):
        self._chk(self.lib.upk_layernorm_f16(self.h, _ptr(x), ldx, rows, d, _ptr(gamma), _ptr(beta), eps,
                                             _ptr(y), ldy, self._s()))

    def timestep_embed(self, t, n, dim, max_period, out, ld_out):
        self._chk(self.lib.upk_timestep_embed_f16(self.h, _ptr(t), n, dim, max_period, _ptr(out), ld_out, self._s()))

    def nchw_to_nhwc(self, x, batch, c, hw, y, ldy, c_off=0, zero_pad_to=0, scale=1.0):
        self._chk(self.lib.upk_nchw_f32_to_nhwc_f16(self.h, _ptr(x), batch, c, hw, _ptr(y), ldy, c_off, zero_pad_to,
                                                    scale, self._s()))

    def nhwc_to_nchw(self, x, ldx, batch, c, hw, y):
        self._chk(self.lib.upk_nhwc_f16_to_nchw_f32(self.h, _ptr(x), ldx, batch, c, hw, _ptr(y), self._s()))

    def f32_to_f16(self, x, rows, cols, y, ldy):
        self._chk(self.lib.upk_f32_to_f16(self.h, _ptr(x), rows, cols, _ptr(y), ldy, self._s()))

    def ddim_step(self, x, eps, coefs, noise, step, pred_x0, xin, ld_xin, batch, c, hw):
        self._chk(self.lib.upk_ddim_step_f32(self.h, _ptr(x), _ptr(eps), _ptr(coefs), _ptr(noise), _ptr(step),
                                             _ptr(pred_x0), _ptr(xin), ld_xin, batch, c, hw, self._s()))

    def ddim_step_cfg(self, x, eps2, coefs, noise, step, pred_x0, xin, ld_xin, batch, c, hw, scale):
        self._chk(self.lib.upk_ddim_step_cfg_f32(self.h, _ptr(x), _ptr(eps2), _ptr(coefs), _ptr(noise), _ptr(step),
                                                 _ptr(pred_x0), _ptr(xin), ld_xin, batch, c, hw, float(scale), self._s()))

    def advance_step(self, step):
        self._chk(self.lib.upk_advance_step(self.h, _ptr(step), self._s()))

    def pack_weight(self, w, row_map=None, n_rows=None, col_map=None, cin_packed=None):
        """fp32 [O,I,kh,kw] or [O,I] CUDA tensor -> packed fp16 tensor, (n_pad)."""
        require(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous(), "pack_weight: w must be a contiguous fp32 CUDA tensor", TypeError)
        if w.dim() == 2:
            cout, cin = w.shape
            kh = kw = 1
        else:
            cout, cin, kh, kw = w.shape
        if n_rows is None:
            n_rows = cout if row_map is None else int(row_map.numel())
        if cin_packed is None:
            cin_packed = ((cin if col_map is None else int(col_map.numel())) + 31) // 32 * 32
        n_pad = (n_rows + 15) // 16 * 16
        out = torch.empty(self.lib.upk_packed_weight_bytes(n_rows, cin_packed, kh, kw) // 2, dtype=torch.float16,
                          device=w.device)
        for m in (row_map, col_map):
            require(m is None or (m.is_cuda and m.dtype == torch.int32 and m.is_contiguous()), "pack_weight: row_map / col_map must be contiguous int32 CUDA tensors", TypeError)
        self._chk(self.lib.upk_pack_weight_f16(self.h, w.data_ptr(), cout, cin, kh, kw, _ptr(row_map), n_rows,
                                               _ptr(col_map), cin_packed, out.data_ptr(), self._s()))
        return out, n_pad

    # -- graphs
    def graph_begin(self):
        self._chk(self.lib.upk_graph_begin(self.h, self._s()))

    def graph_end(self):
        g = C.c_void_p()
        self._chk(self.lib.upk_graph_end(self.h, self._s(), C.byref(g)))
        return g

    def graph_launch(self, g):
        self._chk(self.lib.upk_graph_launch(self.h, g, self._s()))

    def graph_destroy(self, g):
        self.lib.upk_graph_destroy(self.h, g)

    # -- profiling
    def prof_enable(self, on=True):
        self._chk(self.lib.upk_prof_enable(self.h, int(on)))

    def prof_collect(self):
        ms = (C.c_double * NUM_CLASSES)()
        n = (C.c_longlong * NUM_CLASSES)()
        self._chk(self.lib.upk_prof_collect(self.h, ms, n))
        return {CLASS_NAMES[i]: (ms[i], n[i]) for i in range(NUM_CLASSES)}


_ctxs = {}
_ctx_lock = threading.Lock()
PLAN_LOCK = threading.RLock()  # plan / packed-weight caches are built under it (lanes share the model objects)
_lane = threading.local()


# ---- how many batches share the chip: scoped, not a process-wide mode switch
# Two notions that used to be one global:
#  * concurrency() — THREAD-scoped: the number of batches in flight the calling thread builds plans for.  It is n inside a
#    lane of a LanePool(n) (LanePool.lane(i), what the pool's own threads run in) or inside shared_chip(n), else 1.  Plan keys
#    carry `concurrency() > 1`, i.e. the identity of the tuning table the plan was built from (tuning.TUNE_CACHE_LANES vs the
#    latency table), so a server that mixes pooled and un-pooled callers gives each the plans of ITS table, whatever
#    was active when somebody else built theirs.
#  * pools_in_flight() — process-wide: the largest live LanePool (registered at construction, dropped by close()).  Only
#    host_io() looks at it: a capture on one thread and a pageable upload on another collide whichever table they use.
_pools = {}
_pools_lock = threading.Lock()


def concurrency():
    """Batches in flight the CALLING THREAD builds plans for (1 outside LanePool.lane(i) / shared_chip(n)): plans built
    at concurrency > 1 take the throughput-tuned launch choices (tuning.TUNE_CACHE_LANES)."""
    return getattr(_lane, "conc", 1)


def set_concurrency(n):
    """Thread-scoped setter (scripts that build lane plans from their main thread); prefer `with shared_chip(n):`."""
    _lane.conc = max(1, int(n))


@contextlib.contextmanager
def shared_chip(n):
    """Plans the calling thread builds inside this block are tuned for `n` batches sharing the chip."""
    prev = concurrency()
    set_concurrency(n)
    try:
        yield
    finally:
        set_concurrency(prev)


def pools_in_flight():
    """Lanes of the largest live LanePool of the process (1 without one)."""
    with _pools_lock:
        return max(_pools.values(), default=1)


def _register_pool(pool, n):
    with _pools_lock:
        _pools[id(pool)] = int(n)


def _unregister_pool(pool):
    with _pools_lock:
        _pools.pop(id(pool), None)


_HOST_IO_LOCK = threading.RLock()
_HOST_IO_OFF = os.environ.get("UPGPT_HOST_IO_LOCK", "1") == "0"  # (dev: A/B of the lock's cost)


@contextlib.contextmanager
def host_io():
    """Brackets (a) every graph capture and (b) the host sections of a lane that upload from pageable host memory
    (schedule tables, token ids, x_T given on the host): with several lanes the runtime rejects such an upload issued
    while ANOTHER thread captures ("operation not permitted when stream is capturing", also in thread-local capture
    mode), so the two never overlap.  The lane's own stream is drained BEFORE the lock is taken — a blocking upload waits
    for the stream's earlier work (the lane's previous batch, hundreds of ms in flight) and must not do that inside the
    lock.  A no-op with one batch in flight: the serial path is unchanged."""
    if (concurrency() <= 1 and pools_in_flight() <= 1) or _HOST_IO_OFF:
        yield
        return
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()
    with _HOST_IO_LOCK:
        yield


def current_lane():
    """Index of the execution lane of the calling thread (0 unless inside `lane(i)`)."""
    return getattr(_lane, "i", 0)


@contextlib.contextmanager
def lane(i, stream=None, concurrency=None):
    """Everything the calling thread launches inside this block belongs to execution lane `i`: its own upk_ctx (own
    split-K workspace), its own activation buffers / plans / captured graphs (the plan caches of UNetModel and
    AutoencoderKL are keyed by lane) and, when `stream` is given, that HIP stream.  The packed weights are shared.
    Two lanes never touch the same scratch memory, so two batches may be in flight on the device at once — the
    kernels of one lane's latency chain fill the launch boundaries and prologues of the other's.
    Lane 0 is the default lane: code that never enters a lane behaves exactly as before."""
    require(int(i) >= 0, "lane index must be >= 0", ValueError)
    prev, prev_c = getattr(_lane, "i", 0), getattr(_lane, "conc", 1)
    _lane.i = int(i)
    if concurrency is not None:  # (LanePool.lane: plans built in here are tuned for that many batches in flight)
        _lane.conc = max(1, int(concurrency))
    try:
        if stream is not None:
            with torch.cuda.stream(stream):
                yield
        else:
            yield
    finally:
        _lane.i, _lane.conc = prev, prev_c


def get_context(device=None, lane=None):
    idx = torch.cuda.current_device() if device is None else (
        device.index if isinstance(device, torch.device) else int(device))
    if idx is None:
        idx = torch.cuda.current_device()
    key = (idx, current_lane() if lane is None else int(lane))
    if key not in _ctxs:
        with _ctx_lock:
            if key not in _ctxs:
                _ctxs[key] = Context(idx)
    return _ctxs[key]
