"""Weight packing and tensor layouts of the BigVGAN kernels: every function here turns checkpoint tensors (or test tensors)
into the device layouts that `include/flowhigh_hip.h` documents.  Pure torch-CPU arithmetic (float64 where a transform is
applied): no device, no library.  (Split out of vocoder.py in round 5; `flowhigh_amd.vocoder` re-exports every name.)

Reference sites: /root/reference/src/flowhigh/models/bigvgan/models.py:36-72 (AMPBlock convs), :141-146 (ConvTranspose1d),
:196-204 (remove_weight_norm).
"""
import torch

def fold_weight_norm(sd):
    """weight_g / weight_v -> weight (remove_weight_norm, bigvgan/models.py:196-204;
    norm over all dims but 0, for Conv1d and ConvTranspose1d alike)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("weight_g"):
            base = k[:-len("weight_g")]
            vv = sd[base + "weight_v"]
            norm = vv.flatten(1).norm(dim=1).view(-1, *([1] * (vv.ndim - 1)))
            out[base + "weight"] = v * vv / norm
        elif k.endswith("weight_v"):
            continue
        else:
            out[k] = v
    return out


def pick_ck(*cins):
    """Channel chunk of the K loop: 16 when every segment allows it, else 8."""
    return 16 if all(c % 16 == 0 for c in cins) else 8


def pack_conv_weight(w, cout_pad, ck=8):
    """Conv1d weight [co, ci, k] -> [ci/ck, k, cout_pad, ck] (zero padded rows)."""
    co, ci, k = w.shape
    if ci % ck:
        raise ValueError(f"input channels {ci} must be a multiple of {ck}")
    p = torch.zeros(ci // ck, k, cout_pad, ck, dtype=torch.float32)
    p[:, :, :co, :] = w.float().reshape(co, ci // ck, ck, k).permute(1, 3, 0, 2)
    return p.contiguous()


def transposed_conv_phases(k, u):
    """ConvTranspose1d(k, stride u, padding (k-u)//2) as u output phases (models.py:141-146 builds it for
    ANY (u, k)):
    out[co, u*n + r] = sum over taps j with (r + p - j) % u == 0 of x[ci, n + (r + p - j)//u] w[ci, co, j].
    With k - u odd the output has u * L + 1 samples (transposed_conv_extra): phase 0 then has L + 1 positions, the
    others L; the tap lists are the same formula."""
    p = (k - u) // 2
    phases = []
    for r in range(u):
        taps = [(j, (r + p - j) // u) for j in range(k) if (r + p - j) % u == 0]
        phases.append(taps)
    return phases


def transposed_conv_extra(k, u):
    """Samples a ConvTranspose1d(k, u, padding (k - u) // 2) returns beyond u * L: (L - 1) u - 2 ((k - u) // 2) + k - u L."""
    if k < u:
        raise NotImplementedError(f"upsample kernel {k} shorter than its stride {u}")
    return (k - u) % 2


# Winograd F(4,3) weight transform G (6 x 3); interpolation points 0, +-1, +-2, inf
_WINO_G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
           [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]


def pack_wino_weight(w, cout_pad):
    """Conv1d weight [co, ci, k] -> transformed [ci/16, G, 6, cout_pad, 16], G = ceil(k/3):
    u[c, g, xi, co, :] = sum_j G[xi][j] w[co, 16c:16c+16, 3g + j] (float64 on the host, taps past k = 0)."""
    co, ci, k = w.shape
    if ci % 16:
        raise ValueError(f"input channels {ci} must be a multiple of 16")
    ng = -(-k // 3)
    wp = torch.zeros(co, ci, 3 * ng, dtype=torch.float64)
    wp[:, :, :k] = w.double()
    gm = torch.tensor(_WINO_G, dtype=torch.float64)
    u = torch.einsum("xj,ocgj->gxoc", gm, wp.view(co, ci, ng, 3))                 # [G, 6, co, ci]
    u = u.reshape(ng, 6, co, ci // 16, 16).permute(3, 0, 1, 2, 4)                # [ci/16, G, 6, co, 16]
    p = torch.zeros(ci // 16, ng, 6, cout_pad, 16, dtype=torch.float32)
    p[:, :, :, :co, :] = u.float()
    return p.contiguous()


# Winograd F(5,4) weight transform G (8 x 4); points 0, 1, -1, 2, -2, 1/2, -1/2, inf with the scaling of
# tests/tools/winograd_numerics.py: toom_cook (B^T then has the small constants of conv_wino54.hip: kB8Coef)
_WINO54_G = [[-1, 0, 0, 0], [-2 / 9, -2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9, 2 / 9],
             [1 / 90, 1 / 45, 2 / 45, 4 / 45], [1 / 90, -1 / 45, 2 / 45, -4 / 45],
             [32 / 45, 16 / 45, 8 / 45, 4 / 45], [32 / 45, -16 / 45, 8 / 45, -4 / 45], [0, 0, 0, 1]]


def pack_wino54_weight(w, cout_pad):
    """Conv1d weight [co, ci, k] -> transformed [ci/16, G, 8, cout_pad, 16], G = ceil(k/4):
    u[c, g, xi, co, :] = sum_j G8[xi][j] w[co, 16c:16c+16, 4g + j] (float64 on the host, taps past k = 0)."""
    co, ci, k = w.shape
    if ci % 16:
        raise ValueError(f"input channels {ci} must be a multiple of 16")
    ng = -(-k // 4)
    wp = torch.zeros(co, ci, 4 * ng, dtype=torch.float64)
    wp[:, :, :k] = w.double()
    gm = torch.tensor(_WINO54_G, dtype=torch.float64)
    u = torch.einsum("xj,ocgj->gxoc", gm, wp.view(co, ci, ng, 4))                 # [G, 8, co, ci]
    u = u.reshape(ng, 8, co, ci // 16, 16).permute(3, 0, 1, 2, 4)                # [ci/16, G, 8, co, 16]
    p = torch.zeros(ci // 16, ng, 8, cout_pad, 16, dtype=torch.float32)
    p[:, :, :, :co, :] = u.float()
    return p.contiguous()


def pack_amp_weight(w, channels=None):
    """Conv1d weight [co, ci, k] (co, ci <= channels <= 48, channels % 8 == 0) -> the narrow-stage kernel's transformed weights
    (amp_fused.hip, flowhigh_hip.h: fh_amp_seg.u): per (8-channel chunk, group of 4 taps) one stage of 1024 ceil(C / 16) floats
    = blockA [8 points][64 lanes][4] (row tiles 0, 1) | blockB [8 points][64 lanes][2] (the last row tile of an odd count);
    lane l = 16 kq + r holds U[g][xi][16 m + r][8 chunk + 2 kq + s], U = G8 w in float64 as pack_wino54_weight."""
    co, ci, k = w.shape
    c = max(co, ci) if channels is None else channels
    if c % 8 or c > 48 or co > c or ci > c:
        raise ValueError(f"narrow-stage conv: {co} x {ci} channels do not fit {c} (a multiple of 8, <= 48)")
    ng, ma, nch = -(-k // 4), -(-c // 16), c // 8
    wp = torch.zeros(16 * ma, c, 4 * ng, dtype=torch.float64)
    wp[:co, :ci, :k] = w.double()
    gm = torch.tensor(_WINO54_G, dtype=torch.float64)
    u = torch.einsum("xj,ocgj->gxoc", gm, wp.view(16 * ma, c, ng, 4))            # [G, 8, 16 ma, c]
    # -> [chunk, g, xi, kq, r, m, s]
    u = u.view(ng, 8, ma, 16, nch, 4, 2).permute(4, 0, 1, 5, 3, 2, 6).float()    # [chunk, g, xi, kq, r, m, s]
    parts = []
    if ma >= 2:
        parts.append(u[..., :2, :].reshape(nch, ng, 8 * 64 * 4))
    if ma % 2:
        parts.append(u[..., ma - 1, :].reshape(nch, ng, 8 * 64 * 2))
    return torch.cat(parts, dim=-1).contiguous()


def pack_gemm_bf_weight(w):
    """Linear weight [n_pad, K] (n_pad % 64 == 0, K % 32 == 0; rows past N zero) -> the bf16 x 6 GEMM's weights (gemm_bf.hip,
    fh_gemm_bf16x6_f32): float32 container of bf16 bit patterns [n_pad / 64][K / 32][piece h, m, l][k-octet][64 rows][8 bf16]."""
    n_pad, k = w.shape
    if n_pad % 64 or k % 32:
        raise ValueError(f"bf16 x 6 GEMM weight: [{n_pad}, {k}] (rows a multiple of 64, K of 32)")
    w = w.float()
    h = w.to(torch.bfloat16)
    r = w - h.float()
    m = r.to(torch.bfloat16)
    l = (r - m.float()).to(torch.bfloat16)
    p = torch.stack([h, m, l], dim=0).view(3, n_pad // 64, 64, k // 32, 4, 8)           # [piece, granule, row, stage, octet, e]
    return p.permute(1, 3, 0, 4, 2, 5).contiguous().view(torch.int16).reshape(-1).view(torch.float32)


def narrow_slabs(channels):
    """[(first octet, octets)] of the slabs the bf16 x 6 narrow-stage kernel cuts `channels` into (narrow_bf.hip: at most three
    octets = 24 channels per slab, balanced: 24 -> (0, 3); 48 -> (0, 3), (3, 3); 32 -> (0, 2), (2, 2); 40 -> (0, 3), (3, 2))."""
    octets = channels // 8
    n = -(-octets // 3)
    out, ob = [], 0
    for i in range(n):
        og = octets // n + (1 if i < octets % n else 0)
        out.append((ob, og))
        ob += og
    return out


def pack_narrow_bf_weight(w, channels=None):
    """Conv1d weight [co, ci, k] (co, ci <= channels <= 48, channels % 8 == 0, k <= 11) -> the bf16 x 6 narrow-stage kernel's
    weights (narrow_bf.hip, flowhigh_hip.h: fh_narrow_conv_bf16x6_f32): float32 container of bf16 bit patterns,
    per slab (narrow_slabs) ceil(k og / 4) k-blocks x [N tile][piece h, m, l][64 lanes][8 bf16]; lane l of k-block kb holds
    w[16 n + (l & 15), 8 (ob + o) + 0 .. 7, tap] for the pair q = 4 kb + (l >> 4) = tap og + o, zero past k og / past the channels."""
    co, ci, k = w.shape
    c = max(co, ci) if channels is None else channels
    if c % 8 or c > 48 or co > c or ci > c or k > 11:
        raise ValueError(f"narrow-stage conv: {co} x {ci} channels, {k} taps do not fit {c} channels (a multiple of 8, <= 48) / 11 taps")
    ma = -(-c // 16)
    wp = torch.zeros(16 * ma, c, k, dtype=torch.float32)
    wp[:co, :ci] = w.float()
    parts = []
    for ob, og in narrow_slabs(c):
        nb = -(-(k * og) // 4)
        # [tap, o, e] -> q = tap og + o, padded to 4 nb pairs
        blk = wp[:, 8 * ob:8 * (ob + og)].reshape(16 * ma, og, 8, k).permute(0, 3, 1, 2).reshape(16 * ma, k * og, 8)
        full = torch.zeros(16 * ma, 4 * nb, 8, dtype=torch.float32)
        full[:, :k * og] = blk
        # [na, n, kb, lg, e] -> [kb, na, lane = 16 lg + n, e]
        full = full.view(ma, 16, nb, 4, 8).permute(2, 0, 3, 1, 4).reshape(nb, ma, 64, 8)
        h = full.to(torch.bfloat16)
        r = full - h.float()
        m = r.to(torch.bfloat16)
        l = (r - m.float()).to(torch.bfloat16)
        parts.append(torch.stack([h, m, l], dim=2).contiguous().view(torch.int16).reshape(-1))      # [kb, na, piece, lane, 8]
    return torch.cat(parts).contiguous().view(torch.float32)


def split_bf3(u):
    """fp32 tensor [..., 16] -> int16 tensor [..., 3, 16] of bf16 bit patterns: x = h + m + l with h = bf16(x),
    m = bf16(x - h), l = bf16(x - h - m) (round to nearest even; the subtractions are exact in fp32)."""
    u = u.float()
    h = u.to(torch.bfloat16)
    r = u - h.float()
    m = r.to(torch.bfloat16)
    l = (r - m.float()).to(torch.bfloat16)
    return torch.stack([h, m, l], dim=-2).contiguous().view(torch.int16)


def pack_wino_weight_any(w, cout_pad, bf):
    """pack_wino_weight, in the three-piece bf16 form when bf."""
    u = pack_wino_weight(w, cout_pad)
    return split_bf3(u) if bf else u


def pack_wino54_weight_any(w, cout_pad, bf):
    """pack_wino54_weight, in the three-piece bf16 form when bf ([ci/16, G, 8, cout_pad, 3, 16] bf16 bit patterns)."""
    u = pack_wino54_weight(w, cout_pad)
    return split_bf3(u) if bf else u


def wino_phase_weight(wt, taps):
    """ConvTranspose1d weight [cin, cout, k] + the taps [(j, offset)] of one output phase (transposed_conv_phases)
    -> (Conv1d-style weight [cout, cin, k_r] with taps ordered by input offset, center = -smallest offset)."""
    taps = sorted(taps, key=lambda t: t[1])
    offs = [o for _, o in taps]
    if offs != list(range(offs[0], offs[0] + len(offs))):
        raise NotImplementedError(f"phase offsets {offs} are not contiguous")
    w = torch.stack([wt[:, :, j] for j, _ in taps], dim=-1).permute(1, 0, 2).contiguous()
    return w, -offs[0]


def phase_len(length, d):
    """Per-phase row length of the phase-major layout (fh_phase_len)."""
    return ((length + d - 1) // d + 3) & ~3


def to_phase_major(x, d):
    """[B, C, L] -> [B, C, d * phase_len]: x[..., p + d u] at [..., p * lp + u] (host helper for tests / tools)."""
    B, C, L = x.shape
    lp = phase_len(L, d)
    out = torch.zeros(B, C, d, lp, dtype=x.dtype, device=x.device)
    for p_ in range(d):
        v = x[..., p_::d]
        out[:, :, p_, :v.shape[-1]] = v
    return out.reshape(B, C, d * lp)


def from_phase_major(xp, d, length):
    B, C, _ = xp.shape
    lp = phase_len(length, d)
    v = xp.reshape(B, C, d, lp)
    out = torch.empty(B, C, length, dtype=xp.dtype, device=xp.device)
    for p_ in range(d):
        n = len(range(p_, length, d))
        out[..., p_::d] = v[:, :, p_, :n]
    return out
