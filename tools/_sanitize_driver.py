"""Child of tools/sanitize_host.py: runs under LD_PRELOAD = the ASan runtime against the host-only, stub-runtime build of libwf_hip
(WF_LIB).  numpy + ctypes only (no torch: this process must not load a real HIP runtime).  See sanitize_host.py for what is checked."""
import ctypes
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CT = {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t, "int64_t": ctypes.c_int64,
      "uint64_t": ctypes.c_uint64}


def protos():
    src = open(os.path.join(ROOT, "include", "wf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#[^\n]*", " ", src, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w \t\*]*?)\b(wf_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        rt = (ctypes.c_char_p if "char" in ret else ctypes.c_void_p) if "*" in ret else CT[ret.replace("const", "").strip()]
        at = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                at.append(ctypes.c_void_p if "*" in a else CT[re.sub(r"\bconst\b", "", a).split()[0]])
        out[name] = (rt, at)
    return out


P = protos()
dll = ctypes.CDLL(os.environ["WF_LIB"])
for n, (rt, at) in P.items():
    f = getattr(dll, n)
    f.restype, f.argtypes = rt, at
launches = ctypes.c_ulonglong.in_dll(dll, "wf_stub_launches")
empty = ctypes.c_ulonglong.in_dll(dll, "wf_stub_empty_grids")
KEEP = []


def buf(nbytes, fill=0):
    """An exactly-sized host buffer (np.uint8): ASan's red zones sit right behind it."""
    a = np.full(int(nbytes), fill, dtype=np.uint8)
    KEEP.append(a)
    return a.ctypes.data


def f32(n, v=0.0):
    a = np.full(int(n), v, dtype=np.float32)
    KEEP.append(a)
    return a.ctypes.data


def i32(vals):
    a = np.asarray(vals, dtype=np.int32)
    KEEP.append(a)
    return a.ctypes.data


def ok(name, *args):
    before = launches.value
    rc = getattr(dll, name)(*args)
    assert rc == 0, (name, rc, dll.wf_last_error())
    assert launches.value > before, f"{name}: returned success without launching anything"
    assert empty.value == 0, f"{name}: a launch with an empty grid / block or more LDS than a CU has"


def bad(name, *args):
    before = launches.value
    rc = getattr(dll, name)(*args)
    assert rc != 0, f"{name}: accepted bad arguments"
    assert launches.value == before, f"{name}: launched a kernel although it rejected its arguments"
    assert dll.wf_last_error(), name


if os.environ.get("WF_SANITIZE_SELFTEST"):
    # prove the harness bites: a crack-fill workspace 200 bytes short of what wf_crack_fill_workspace_bytes reports MUST trip ASan inside the
    # stub's host-side memset (the process then exits with ASan's exit code, never reaching the line below)
    nv, Hh, Ww = 3, 40, 56
    dll.wf_crack_fill(buf(nv * Hh * Ww * 3), buf(nv * Hh * Ww, 1), f32(nv * Hh * Ww, 2.0), buf(nv * Hh * Ww * 3), buf(nv * Hh * Ww), f32(nv * Hh * Ww),
                      nv, Hh, Ww, 4, 3, 5, buf(dll.wf_crack_fill_workspace_bytes(nv, Hh, Ww) - 200), None)
    print("selftest: the undersized workspace was NOT detected")
    sys.exit(0)

# ---- (a) every entry point with all-null pointers and zero sizes: an error (or a no-op for the element-wise n == 0 case), never a crash ----
nul = 0
for name, (rt, at) in P.items():
    if name in ("wf_version", "wf_last_error", "wf_attn_debug_body_counter"):
        continue
    args = [None if t is ctypes.c_void_p else t(0) for t in at]
    before = launches.value
    rc = getattr(dll, name)(*args)
    assert launches.value == before, f"{name}: launched with null pointers"
    nul += 1
assert dll.wf_device_info(0, None, None, None, 0) != 0       # the stub reports no device: surfaced as an error

# ---- (b) valid calls, buffers sized exactly ----
BF, F = 1, 0
n = 4096 + 37
ok("wf_cfg_combine", buf(2 * n), buf(2 * n), buf(2 * n), BF, 4.0, n, None)
ok("wf_x0_from_v", f32(n), F, buf(2 * n), BF, f32(n), 0.5, n, None)
ok("wf_unipc_update", f32(n), F, f32(n), F, f32(n), F, f32(n), 1.0, 0.5, 0.25, 0.5, n, None)
ok("wf_add_noise", f32(n), F, f32(n), F, f32(n), 0.4, 0.6, n, None)
ok("wf_latent_affine", f32(16 * 100), F, f32(16 * 100), F, f32(16), f32(16, 1.0), 0, 1, 16, 100, None)
ok("wf_blend_pixels", f32(3 * 500), f32(500), f32(3 * 500), f32(3 * 500), 1, 3, 500, None)
ok("wf_postprocess_video", f32(3 * 2 * 8 * 8), f32(3 * 2 * 8 * 8), 3, 2, 8, 8, None)
ok("wf_cast", f32(n), F, buf(2 * n), BF, n, None)
ok("wf_channel_swap", f32(16 * 50), F, f32(16 * 50), F, i32([1, 5, 9]), 3, 1, 16, 50, None)
bad("wf_channel_swap", f32(16 * 50), F, f32(16 * 50), F, i32([1, 16]), 2, 1, 16, 50, None)          # channel index out of range
ok("wf_resize_bilinear2d", f32(2 * 8 * 8), f32(2 * 12 * 10), 2, 8, 8, 12, 10, None)
ok("wf_resize_nearest2d", f32(2 * 8 * 8), f32(2 * 12 * 10), 2, 8, 8, 12, 10, None)
ok("wf_soften_mask", f32(2 * 20 * 24), f32(2 * 20 * 24), 2, 20, 24, 15, 2, None)
bad("wf_soften_mask", f32(4), f32(4), 1, 2, 2, 15, 9, None)                                              # unknown decay type
ok("wf_dsg", buf(2 * n), buf(2 * n), buf(2 * n), BF, 4.0, n, f32(dll.wf_dsg_workspace_floats()), None)
ok("wf_cfg_zero", f32(n), f32(n), f32(n), 4.0, 1, n, f32(dll.wf_dsg_workspace_floats()), None)
ok("wf_temporal_diff", f32(16 * 5 * 60), F, f32(16 * 4 * 60), 16, 5, 60, None)
ok("wf_flow_metrics", f32(16 * 4 * 2 * 60), f32(16 * 4 * 2 * 60), f32(16), 16, 4, 2, 2, 60, f32(dll.wf_flow_metrics_workspace_floats(16)), None)
for (C, T, h, w) in ((16, 5, 60, 104), (3, 2, 17, 33), (16, 21, 90, 160)):
    ok("wf_farneback_flows", f32(C * T * h * w), F, f32(C * (T - 1) * 2 * h * w), C, T, h, w, 0, buf(dll.wf_farneback_workspace_bytes(C, T, h, w)), None)
bad("wf_farneback_flows", f32(8), F, f32(8), 1, 1, 2, 4, 0, buf(64), None)                               # a single frame has no flow
# GEMMs: every tile-selection branch (320-wide, 256-wide, ragged fallback), every epilogue
for (M, N, K, epi) in ((512, 5120, 5120, 0), (4095, 15360, 5120, 0), (300, 13824, 5120, 1), (257, 64, 5120, 2), (1, 30720, 5120, 2),
                       (640, 5120, 13824, 3), (32760, 5120, 144, 2)):
    out_b = 4 if epi in (2, 3, 4) else 2
    ok("wf_gemm_bf16", buf(2 * M * K), buf(2 * N * K), f32(N), buf(out_b * M * N), f32(N) if epi == 3 else None, M, N, K, K, K, N, epi, None)
bad("wf_gemm_bf16", buf(64), buf(64), None, buf(64), None, 4, 4, 4, 2, 4, 4, 0, None)                    # ldx < K
bad("wf_gemm_bf16", buf(64), buf(64), None, buf(64), None, 4, 4, 4, 4, 4, 4, 9, None)                    # unknown epilogue
ok("wf_gemm_bf16_batched", buf(2 * 8 * 770 * 128), buf(2 * 8 * 776 * 128), buf(2 * 8 * 770 * 776), 8, 770, 776, 128, 128, 128, 776, 770 * 128,
   776 * 128, 770 * 776, 0, None)
ok("wf_gemm_f16_batched", buf(2 * 3 * 130 * 64), buf(2 * 3 * 72 * 64), buf(4 * 3 * 130 * 72), 3, 130, 72, 64, 64, 64, 72, 130 * 64, 72 * 64, 130 * 72, 2, None)
bad("wf_gemm_f16_batched", buf(64), buf(64), buf(64), 1, 4, 4, 8, 8, 8, 4, 0, 0, 0, 3, None)             # epilogue 3 is not built for the batched form
# attention: dense, split (workspace), sparse
H, Lq, Lkp = 8, 1000, 1024
Q, K, V, O = buf(2 * H * Lq * 128), buf(2 * H * Lkp * 128), buf(2 * H * Lkp * 128), buf(2 * Lq * H * 128)
ok("wf_attn_fwd", Q, K, V, O, H, Lq, Lkp, 1000, Lkp, 0, H * 128, 0.0884, 0, None, 0, 0, None, 0, None)
ok("wf_attn_fwd", Q, K, V, O, H, Lq, Lkp, 1000, 512, 0, H * 128, 0.0, 1, f32(2 * H), 2, 0, f32(H), 1, None)    # two gathered shards, pre-scaled Q
ws = buf(dll.wf_attn_split_workspace_bytes(H, Lq, 2))
ok("wf_attn_fwd_split", Q, K, V, O, H, Lq, Lkp, 1000, Lkp, 0, H * 128, 0.0, 0, 2, ws, f32(H), 1, 0, f32(H), 1, None)
bad("wf_attn_fwd", Q, K, V, O, H, Lq, 1000, 1000, 1000, 0, H * 128, 0.0884, 0, None, 0, 0, None, 0, None)      # Lkp not a multiple of 64
bad("wf_attn_fwd_split", Q, K, V, O, H, Lq, Lkp, 1000, Lkp, 0, H * 128, 0.0, 0, 2, None, None, 0, 0, None, 0, None)   # split without workspace
bad("wf_attn_fwd", Q + 2, K, V, O, H, Lq, Lkp, 1000, Lkp, 0, H * 128, 0.0884, 0, None, 0, 0, None, 0, None)    # misaligned Q
# packed exchange slots (parallel.KVExchange): 4 sources x [K | V^T | bounds], 256 keys each; own-first part launches + merge
seg, P4 = 256, 4
slot = 2 * (2 * H * seg * 128) + 256                      # bytes
X = buf(P4 * slot)
Kx, Vx, KMx = X, X + 2 * H * seg * 128, X + 4 * H * seg * 128
ok("wf_attn_fwd", Q, Kx, Vx, O, H, Lq, P4 * seg, 1000, seg, slot, H * 128, 0.0, 0, KMx, P4, slot // 4, f32(H), 1, None)
bad("wf_attn_fwd", Q, Kx, Vx, O, H, Lq, P4 * seg, 1000, seg, H * seg * 256 - 16, H * 128, 0.0, 0, None, 0, 0, None, 0, None)   # a slot shorter than its K shard
bad("wf_attn_fwd", Q, Kx, Vx, O, H, Lq, P4 * seg, 1000, seg, slot, H * 128, 0.0, 0, KMx, P4, H - 1, f32(H), 1, None)  # bound vectors overlap
ws6 = buf(dll.wf_attn_split_workspace_bytes(H, Lq, 4))
ok("wf_attn_fwd_part", Q, Kx, Vx, H, Lq, P4 * seg, 1000, seg, slot, 4, 8, 0, 0, 1, 0, 4, ws6, None, 0, KMx + slot, 1, slot // 4, f32(H), 1, None)     # own segment 1
ok("wf_attn_fwd_part", Q, Kx, Vx, H, Lq, P4 * seg, 1000, seg, slot, 0, 4, 8, 16, 3, 1, 4, ws6, None, 0, KMx, P4, slot // 4, f32(H), 1, None)          # peers: before + after as one sequence in 3 splits
ok("wf_attn_merge", O, H, Lq, H * 128, 0, 4, ws6, None)
bad("wf_attn_fwd_part", Q, Kx, Vx, H, Lq, P4 * seg, 1000, seg, slot, 0, 4, 2, 16, 3, 1, 4, ws6, None, 0, None, 0, 0, None, 0, None)                  # second window overlaps the first
bad("wf_attn_fwd_part", Q, Kx, Vx, H, Lq, P4 * seg, 1000, seg, slot, 0, 4, 8, 16, 3, 1, 4, ws6, O, H * 128, None, 0, 0, None, 0, None)             # a merging launch must be the last slot, one split
bad("wf_attn_fwd_part", Q, Kx, Vx, H, Lq, P4 * seg, 1000, seg, slot, 0, 4, 8, 16, 3, 2, 4, ws6, None, 0, None, 0, 0, None, 0, None)                  # slots 2..4 of 4
bad("wf_attn_fwd_part", Q, Kx, Vx, H, Lq, P4 * seg, 1000, seg, slot, 16, 20, 0, 0, 1, 0, 4, ws6, None, 0, None, 0, 0, None, 0, None)                 # window behind the last valid tile
bad("wf_attn_fwd_part", Q, Kx, Vx, H, Lq, P4 * seg, 1000, seg, slot, 0, 3, 8, 16, 1, 1, 4, ws6, None, 0, None, 0, 0, None, 0, None)                   # the hole must be whole segments (4 tiles each)
bad("wf_attn_merge", O, H, Lq, H * 128, 0, 13, ws6, None)                                                                                # more than 12 slots
ok("wf_head_max_norm2", K, H, 1000, Lkp, f32(H), None)
ok("wf_attn_cross2_fwd", Q, buf(2 * H * 832 * 128), buf(2 * H * 832 * 128), O, H, Lq, 320, 257, 512, 512, H * 128, 0.0884, None)
bad("wf_attn_cross2_fwd", Q, K, V, O, H, Lq, 320, 200, 512, 512, H * 128, 0.0884, None)                  # context 1 leaves a whole tile empty
bad("wf_attn_cross2_fwd", Q, K, V, O, H, Lq, 320, 257, 512, 512, H * 128, 0.0, None)                     # the fused kernel applies the scale itself
nq, nk, nsel = 7, 9, 3
sc = buf(2 * 2 * nq * 16)
mx = min(2 * nsel, nk)
lists, counts, mask = buf(4 * 2 * 4 * mx), buf(4 * 2 * 4), buf(4 * 2 * nq * 1)
ok("wf_bsa_topk_lists", sc, 16, 2, nq, nk, nsel, 128, nk, lists, counts, mx, mask, None)
bad("wf_bsa_topk_lists", sc, 16, 2, nq, nk, 0, 128, nk, lists, counts, mx, mask, None)                    # nothing selected
lists_c = buf(4 * 2 * 4 * nk)
ok("wf_bsa_cdf_lists", sc, 16, 2, nq, nk, 0.5, 2, 128, nk, lists_c, counts, nk, mask, buf(4 * 2 * nq), None)
bad("wf_bsa_cdf_lists", sc, 16, 2, nq, nk, 0.5, 2, 128, nk, lists_c, counts, nk - 1, mask, buf(4 * 2 * nq), None)   # a row may take every block
bad("wf_bsa_cdf_lists", sc, 16, 2, nq, nk, -0.1, 2, 128, nk, lists_c, counts, nk, mask, buf(4 * 2 * nq), None)      # negative threshold
ok("wf_attn_bsa_fwd", buf(2 * 2 * 896 * 128), buf(2 * 2 * 1152 * 128), buf(2 * 2 * 1152 * 128), buf(2 * 896 * 256), 2, 896, 1152, 1152, 256,
   0.0884, lists, counts, mx, 128, None)
bad("wf_attn_bsa_fwd", Q, K, V, O, 2, 900, 1152, 1152, 256, 0.0884, lists, counts, mx, 128, None)        # Lq not whole blocks
# DiT row-wise ops
L, C = 300, 5120
ok("wf_ln_modulate", f32(L * C), f32(C), f32(C), buf(2 * L * C), BF, L, C, 1e-6, 1, None)
ok("wf_rmsnorm_heads", buf(2 * L * 3 * C), 3 * C, f32(C), f32(L * 64), f32(L * 64), buf(2 * 40 * 320 * 128), L, 320, C, 1e-6, 0.1275, None)
ok("wf_rmsnorm_heads_bound", buf(2 * L * 3 * C), 3 * C, f32(C), f32(L * 64), f32(L * 64), buf(2 * 40 * 320 * 128), L, 320, C, 1e-6, 0.1275,
   f32(dll.wf_rmsnorm_heads_bound_ws_floats(L, C)), f32(40), None)
bad("wf_rmsnorm_heads_bound", buf(64), 8, f32(8), None, None, buf(64), 4, 4, 0, 1e-6, 1.0, f32(8), f32(8), None)   # C = 0: no heads to divide by
ok("wf_v_transpose", buf(2 * L * 3 * C), 3 * C, buf(2 * 40 * 320 * 128), L, 320, 40, None)
ok("wf_v_transpose_seg", buf(2 * L * 3 * C), 3 * C, buf(2 * 40 * 13 * 128 * 64), L, 320, 40, 13, None)
bad("wf_v_transpose_seg", buf(2 * L * 3 * C), 3 * C, buf(64), L, 320, 40, 4, None)                        # head stride shorter than the segment
bad("wf_v_transpose", buf(64), 8, buf(64), 4, 100, 1, None)                                               # Lp not a multiple of 64
ok("wf_patchify", buf(2 * 36 * 3 * 8 * 12), buf(2 * 3 * 4 * 6 * 144), 36, 3, 8, 12, None)
ok("wf_unpatchify", f32(3 * 4 * 6 * 64), f32(16 * 3 * 8 * 12), 16, 3, 8, 12, None)
bad("wf_patchify", buf(64), buf(64), 1, 1, 3, 4, None)                                                    # odd latent height
ok("wf_act", f32(n), F, None, 0, buf(2 * n), BF, 0, n, None)
bad("wf_act", f32(4), F, None, 0, f32(4), F, 7, 4, None)
# LongCat row-wise ops
Lc, Cc, tpf = 24, 4096, 8
ok("wf_lc_ln_modulate", buf(2 * Lc * Cc), f32(3 * Cc), f32(3 * Cc), Cc, tpf, 0, None, 1, buf(2 * Lc * Cc), Lc, Cc, 1e-6, None)
ok("wf_lc_gate_residual", buf(2 * Lc * Cc), buf(2 * Lc * Cc), Cc, f32(3 * Cc), Cc, tpf, 0, None, Lc, Cc, None)
ok("wf_lc_norm_heads", buf(2 * Lc * 3 * Cc), 3 * Cc, f32(128), f32(Lc * 64), f32(Lc * 64), buf(2 * 32 * 64 * 128), Lc, 64, 32, 1e-6, 1.0, None)
ok("wf_lc_swiglu", buf(2 * Lc * 2 * 11008), 2 * 11008, buf(2 * Lc * 11008), Lc, 11008, None)
ok("wf_lc_mean_pool_blocks", buf(2 * 4 * 256 * 128), buf(2 * 4 * 2 * 128), 4, 256, 128, None)
ok("wf_gather_rows_bf16", buf(2 * 10 * 64), 64, i32([3, 1, 9, 0]), buf(2 * 4 * 64), 64, 4, 64, None)
ok("wf_refine_upsample_u8", buf(5 * 16 * 24 * 3), f32(3 * 9 * 32 * 48), 5, 16, 24, 9, 32, 48, None)
# VAE ops
T_, H_, W_, Ci, Co = 3, 16, 20, 96, 96
zp = buf(4096)
ok("wf_conv3d_cl", buf(2 * T_ * H_ * W_ * Ci), buf(2 * Co * 27 * Ci), f32(Co), None, f32(T_ * H_ * W_ * Co), None, T_, H_, W_, Ci, T_, H_, W_, Co,
   3, 3, 3, 1, 1, 2, 1, 1, 0, 0, zp, None)
ok("wf_conv3d_pack333", buf(2 * Co * 27 * Ci), buf(2 * 27 * (Ci // 16) * Co * 16), Co, Ci, None)
ok("wf_conv3d_333", buf(2 * T_ * H_ * W_ * Ci), buf(2 * 27 * (Ci // 16) * Co * 16), f32(Co), None, f32(T_ * H_ * W_ * Co), None, T_, H_, W_, Ci, H_, Co, 1,
   zp, 4096, 0, Ci, None)
bad("wf_conv3d_333", buf(64), buf(64), None, None, f32(16), None, 1, 4, 4, 24, 4, 96, 1, zp, 4096, 0, 24, None)
bad("wf_conv3d_333", buf(2 * T_ * H_ * W_ * Ci), buf(2 * 27 * (Ci // 16) * Co * 16), f32(Co), None, f32(T_ * H_ * W_ * Co), None, T_, H_, W_, Ci, H_, Co, 1,
    zp, dll.wf_conv3d_333_zero_page_bytes(W_, Ci, 1) - 1, 1, Ci, None)                                   # zero page shorter than the largest slice offset   # Cin not a multiple of 32
ok("wf_conv3d_small", f32(T_ * H_ * W_ * 16), F, f32(16 * 32), f32(32), f32(T_ * H_ * W_ * 32), None, T_, H_, W_, 16, T_, H_, W_, 32, 1, 1, 1, 1, 1, 0, 0,
   0.0, None)
npix = T_ * H_ * W_
ok("wf_rms_silu_cl", f32(npix * Ci), f32(Ci), buf(2 * npix * Ci), None, npix, Ci, 1, None)
ok("wf_rms_silu_cl_x3", f32(npix * Ci), f32(Ci), buf(2 * npix * 3 * Ci), npix, Ci, 1, None)
ok("wf_rms_silu_cl_blocked", f32(npix * Ci), f32(Ci), buf(2 * npix * 2 * Ci), npix, Ci, 1, W_, 1, 0, None)
ok("wf_split_bf16x3", f32(npix * Ci), Ci, buf(2 * npix * 3 * Ci), 3 * Ci, npix, Ci, 0, None)
# the fp16 operand formats (round 4): the same host paths with the element type flag set
ok("wf_conv3d_cl_f16", buf(2 * T_ * H_ * W_ * Ci), buf(2 * Co * 27 * Ci), f32(Co), None, f32(T_ * H_ * W_ * Co), None, T_, H_, W_, Ci, T_, H_, W_, Co,
   3, 3, 3, 1, 1, 2, 1, 1, 0, 0, zp, 2.0 ** -14, None)
bad("wf_conv3d_cl_f16", buf(2 * T_ * H_ * W_ * Ci), buf(2 * Co * 27 * Ci), f32(Co), None, f32(T_ * H_ * W_ * Co), None, T_, H_, W_, Ci, T_, H_, W_, Co,
    3, 3, 3, 1, 1, 2, 1, 1, 0, 0, zp, 0.0, None)                                                          # acc_scale must be positive
ok("wf_conv3d_cl_scatter_f16", buf(2 * T_ * H_ * W_ * Ci), buf(2 * Co * 4 * Ci), f32(Co), None, f32(T_ * 2 * H_ * 2 * W_ * Co), None, T_, H_, W_, Ci, T_, H_, W_, Co,
   1, 2, 2, 1, 1, 0, 1, 1, zp, 2 * H_, 2 * W_, 2, 0, 2, 0, 1.0, None)
ok("wf_conv3d_333_f16", buf(2 * T_ * H_ * W_ * Ci), buf(2 * 27 * (Ci // 16) * Co * 16), f32(Co), None, f32(T_ * H_ * W_ * Co), None, T_, H_, W_, Ci, H_, Co, 1,
   zp, 4096, 0, Ci, 2.0 ** -17, None)
ok("wf_rms_silu_cl_x3_f16", f32(npix * Ci), f32(Ci), buf(2 * npix * 3 * Ci), npix, Ci, 1, None)
ok("wf_rms_silu_cl_blocked_f16", f32(npix * Ci), f32(Ci), buf(2 * npix * 2 * Ci), npix, Ci, 1, W_, 1, 0, None)
ok("wf_split_f16x3", f32(npix * Ci), Ci, buf(2 * npix * 3 * Ci), 3 * Ci, npix, Ci, 0, None)
# round 5: the asynchronous range flag (a 4-byte copy into caller memory, no launch), the calibration stream, the stream-ordered delay
flag = np.full(1, 7, dtype=np.int32)
KEEP.append(flag)
assert dll.wf_f16_overflow_flag_async(flag.ctypes.data, None) == 0 and int(flag[0]) == 0, "wf_f16_overflow_flag_async"
bad("wf_f16_overflow_flag_async", None, None)
flop = ctypes.c_double(0.0)
ok("wf_calib_mfma", buf(1 << 20), f32(4), 10, ctypes.addressof(flop), None)
assert flop.value == 256.0 * 4 * 10 * 16 * 2 * 32 * 32 * 16, flop.value
ok("wf_calib_mfma", buf(1 << 20), f32(4), 1, None, None)
bad("wf_calib_mfma", None, f32(4), 10, None, None)
bad("wf_calib_mfma", buf(1 << 20), f32(4), 0, None, None)
ok("wf_delay_us", 5.0, None)
before_ = launches.value
assert dll.wf_delay_us(0.0, None) == 0 and launches.value == before_, "a zero delay launches nothing"
bad("wf_delay_us", -1.0, None)
bad("wf_delay_us", 1.0e7, None)
for (M, N, K, epi) in ((4095, 1152, 1152, 2), (300, 384, 1152, 4), (2048, 2560, 384, 0)):
    ok("wf_gemm_f16", buf(2 * M * K), buf(2 * N * K), f32(N), buf((2 if epi == 0 else 4) * M * N), M, N, K, K, K, N, epi, 0.25, None)
bad("wf_gemm_f16", buf(64), buf(64), None, buf(64), 4, 4, 8, 8, 8, 4, 3, 1.0, None)                           # the gated-residual epilogue is not built for fp16 operands
ok("wf_softmax_rows", f32(40 * 48), 48, buf(2 * 40 * 48), 48, 40, 48, 0.5, None)
ok("wf_softmax_rows_f32", f32(40 * 48), 48, f32(40 * 48), 48, 40, 48, 0.5, None)
ok("wf_transpose_bf16", buf(2 * 40 * 48), 48, buf(2 * 48 * 40), 40, 40, 48, None)
ok("wf_transpose_f32", f32(40 * 48), 48, f32(48 * 40), 40, 40, 48, None)
ok("wf_ncthw_to_cl", f32(3 * npix), f32(npix * 32), None, 3, 32, npix, None)
ok("wf_cl_to_ncthw", f32(npix * 32), f32(3 * npix), 3, 32, npix, 1.0, None)
# stage-1 kernels
Hh, Ww, nv = 40, 56, 3
geo = np.zeros(30, dtype=np.float64); cams = np.zeros(nv * 12, dtype=np.float64); KEEP += [geo, cams]
ok("wf_warp_splat", f32(Hh * Ww * 3), f32(Hh * Ww, 2.0), geo.ctypes.data, cams.ctypes.data, buf(nv * Hh * Ww * 3), buf(nv * Hh * Ww), f32(nv * Hh * Ww),
   buf(8 * nv * Hh * Ww), nv, Hh, Ww, None)
ok("wf_crack_fill", buf(nv * Hh * Ww * 3), buf(nv * Hh * Ww, 1), f32(nv * Hh * Ww, 2.0), buf(nv * Hh * Ww * 3), buf(nv * Hh * Ww), f32(nv * Hh * Ww), nv, Hh, Ww,
   4, 3, 5, buf(dll.wf_crack_fill_workspace_bytes(nv, Hh, Ww)), None)
bad("wf_crack_fill", buf(64), buf(64), f32(16), buf(64), buf(64), f32(16), 1, 4, 4, 4, 3, 9, buf(4096), None)          # more segments than the kernel keeps
ok("wf_fill_small_cracks", buf(Hh * Ww * 3), buf(Hh * Ww, 1), f32(Hh * Ww, 2.0), 1, buf(Hh * Ww * 3), buf(Hh * Ww), Hh, Ww, 0.1, 5, 3,
   buf(dll.wf_fill_small_cracks_workspace_bytes(Hh, Ww)), None)
bad("wf_fill_small_cracks", buf(Hh * Ww * 3), buf(Hh * Ww, 1), None, 1, buf(Hh * Ww * 3), buf(Hh * Ww), Hh, Ww, 0.1, 5, 3, buf(64), None)  # depth step without a depth map
cam = np.zeros(16, dtype=np.float32); KEEP.append(cam)
ok("wf_points_render", f32(500 * 3), f32(500 * 3), None, 500, 3, cam.ctypes.data, Hh, Ww, 0.005, 1, f32(Hh * Ww * 3), buf(Hh * Ww),
   buf(dll.wf_points_render_workspace_bytes(Hh, Ww)), None)
ok("wf_depth_edge_mask", f32(Hh * Ww, 1.0), Hh, Ww, 0.1, 3, 0.3, 2, buf(Hh * Ww), buf(dll.wf_depth_edge_mask_workspace_bytes(Hh, Ww)), None)

print(f"sanitize driver: {nul} entry points survived null arguments; {launches.value} stubbed launches, {empty.value} bad launch geometries")
sys.exit(0)
