"""FLowHigh vector-field network (transformer backbone) on the HIP kernels.

Mirrors /root/reference/src/flowhigh/models/flow.py:185-261 (`FLowHigh.forward` at inference:
cond_drop_prob = 0, no masks, architecture = 'transformer'), transformer.py:167-234,
attend.py:153-189 and pos_emb.py.  The reference's tensor-to-string INFO logging and NaN
`.any()` host syncs (flow.py:256-267) are deliberately not reproduced.

One vector-field evaluation = 21 kernel launches for depth 2 (the reference issues 1 882 aten
calls).  Fusions:
  * cat(x, cond) @ W_embed^T  ->  x @ W_x^T + (cond @ W_c^T + b), the second term computed once
    per generate() because cond does not change across ODE evaluations (flow.py:234-239);
  * bias / residual / ODE axpy ("y + dt * f") in GEMM epilogues; GEGLU in the FF1 epilogue;
  * all eight adaptive-norm gamma/beta projections of a forward in one GEMV (the time embedding
    is identical for every clip, flow.py:208-211).
"""
import torch

from . import hip
from .tables import rotary_tables

FH = "flowhigh."


def _pad_rows(w, mult=128):
    n = w.shape[0]
    n_pad = (n + mult - 1) // mult * mult
    if n_pad == n:
        return w.contiguous()
    out = torch.zeros((n_pad,) + tuple(w.shape[1:]), dtype=w.dtype)
    out[:n] = w
    return out


def pack_geglu(w, b):
    """FF1 weight [2*inner, K], bias [2*inner] -> packed blocks of 64 rows = 32 value + 32 gate
    (transformer.py:92-95: x, gate = chunk(2)).  Returns (W' [nblk*64, K], b' [nblk*64], inner_pad)."""
    two_inner, k = w.shape
    inner = two_inner // 2
    nblk = (inner + 31) // 32
    wp = torch.zeros(nblk, 2, 32, k, dtype=torch.float32)
    bp = torch.zeros(nblk, 2, 32, dtype=torch.float32)
    val, gate = w[:inner], w[inner:]
    vpad = torch.zeros(nblk * 32, k)
    gpad = torch.zeros(nblk * 32, k)
    vpad[:inner], gpad[:inner] = val, gate
    wp[:, 0], wp[:, 1] = vpad.view(nblk, 32, k), gpad.view(nblk, 32, k)
    bv = torch.zeros(nblk * 32)
    bg = torch.zeros(nblk * 32)
    bv[:inner], bg[:inner] = b[:inner], b[inner:]
    bp[:, 0], bp[:, 1] = bv.view(nblk, 32), bg.view(nblk, 32)
    return _pad_rows(wp.view(nblk * 64, k)), bp.view(-1).contiguous(), nblk * 32


class FlowNet:
    def __init__(self, sd, device, depth=2, heads=16, dim_head=64, store=None, bf=False):
        if dim_head != 64:
            raise NotImplementedError("dim_head must be 64")
        self.device = hip.norm_device(device)
        dev = self.device
        # (device tensors come through the weight store: packed from `sd`, or taken from an uploaded weight blob: weights.py)
        from .weights import WeightStore
        W = store if store is not None else WeightStore(dev)
        g = lambda name: sd[FH + name].detach().float().cpu()
        self.depth, self.heads = depth, heads
        # bf: the linears run in the bf16 x 6 form (gemm_bf.hip: six bf16 MFMAs per product over exact three-piece splits,
        # fp32-grade; the model's conv_form = 'bf16x6'): their weights are stored split (6 bytes per weight)
        self.bf = bool(bf)
        if getattr(W, "form", None) is not None and (W.form == "bf16x6") != self.bf:
            # (a blob holds the linears in ONE form: asking it for the other would fail on the first missing key)
            raise ValueError(f"the weight blob was packed for conv_form={W.form!r}: its linears are "
                             f"{'split into bf16 pieces' if W.form == 'bf16x6' else 'fp32'}, this model asks for the other form")
        if self.bf:
            from .packing import pack_gemm_bf_weight
            dev_w = lambda key, make: W.dev(key + ".bf3", lambda: pack_gemm_bf_weight(make()))
        else:
            dev_w = W.dev
        w_embed = lambda: g("to_embed.weight")
        dims = W.host("f.dims", lambda: torch.tensor([w_embed().shape[0], w_embed().shape[1] // 2,
                                                      g("conv_embed.dw_conv1d.0.weight").shape[-1]], dtype=torch.int32)).tolist()
        self.dim, self.dim_in, self.dw_k = dims
        if self.dim % 256 or self.dim_in % 32 or heads * dim_head != self.dim:
            raise NotImplementedError("unsupported transformer dims")
        self.w_x = dev_w("f.w_x", lambda: _pad_rows(w_embed()[:, :self.dim_in]))
        self.w_c = dev_w("f.w_c", lambda: _pad_rows(w_embed()[:, self.dim_in:]))
        self.b_embed = W.dev("f.b_embed", lambda: g("to_embed.bias"))
        self.null_cond = W.dev("f.null_cond", lambda: g("null_cond").reshape(1, -1))
        self._e_null = None
        self.dw_w = W.dev("f.dw_w", lambda: g("conv_embed.dw_conv1d.0.weight").reshape(self.dim, self.dw_k).t())   # [ksz, dim], tap-major
        self.dw_b = W.dev("f.dw_b", lambda: g("conv_embed.dw_conv1d.0.bias"))
        self.sinu_w = W.dev("f.sinu_w", lambda: g("sinu_pos_emb.0.weights"))
        self.t_w = W.dev("f.t_w", lambda: g("sinu_pos_emb.1.weight"))
        self.t_b = W.dev("f.t_b", lambda: g("sinu_pos_emb.1.bias"))
        self.layers = []
        for layer in range(depth):
            p = f"transformer.layers.{layer}."
            ff = {}

            def ff1(p=p, ff=ff):
                if not ff:
                    ff["w1"], ff["b1"], ff["inner_pad"] = pack_geglu(g(p + "5.0.weight"), g(p + "5.0.bias"))
                return ff

            def w2_padded(p=p):
                w2 = g(p + "5.3.weight")
                w2p = torch.zeros(w2.shape[0], ff1()["inner_pad"])
                w2p[:, :w2.shape[1]] = w2
                return _pad_rows(w2p)
            k = f"f.layer{layer}."
            self.layers.append(dict(
                gq=W.dev(k + "gq", lambda: g(p + "3.q_norm.gamma").reshape(heads, 64)),
                gk=W.dev(k + "gk", lambda: g(p + "3.k_norm.gamma").reshape(heads, 64)),
                w_qkv=dev_w(k + "w_qkv", lambda: _pad_rows(g(p + "3.to_qkv.weight"))),
                w_out=dev_w(k + "w_out", lambda: _pad_rows(g(p + "3.to_out.weight"))),
                w1=dev_w(k + "w1", lambda: ff1()["w1"]), b1=W.dev(k + "b1", lambda: ff1()["b1"]),
                w2=dev_w(k + "w2", w2_padded), b2=W.dev(k + "b2", lambda: g(p + "5.3.bias")),
                inner_pad=int(W.host(k + "inner_pad", lambda: torch.tensor([ff1()["inner_pad"]], dtype=torch.int32))[0])))

        def gamma_beta(which_tensor):
            out = []
            for layer in range(depth):
                for nidx in ("2", "4"):
                    for which in ("to_gamma", "to_beta"):
                        out.append(g(f"transformer.layers.{layer}.{nidx}.{which}.{which_tensor}"))
            return torch.cat(out, 0)
        self.gb_w = W.dev("f.gb_w", lambda: gamma_beta("weight"))      # [depth*4*dim, dim]
        self.gb_b = W.dev("f.gb_b", lambda: gamma_beta("bias"))
        self.final_gamma = W.dev("f.final_gamma", lambda: g("transformer.final_norm.gamma"))
        self.w_pred = dev_w("f.w_pred", lambda: _pad_rows(g("to_pred.weight")))
        self.inv_freq = W.host("f.inv_freq", lambda: g("transformer.rotary_emb.inv_freq"))
        self._ws = hip.ShapeCache()

    def gemm(self, A, W, C_out, M, N, K, **kw):
        return hip.gemm(A, W, C_out, M, N, K, bf=self.bf, **kw)

    @hip.on_device
    def workspace(self, batch, n):
        key = (batch, n)
        if key in self._ws:
            return self._ws[key]
        dev, M, D = self.device, batch * n, self.dim
        f32 = dict(dtype=torch.float32, device=dev)
        cos_t, sin_t = rotary_tables(self.inv_freq, n)
        ws = dict(
            e_cond=torch.empty(M, D, **f32), h=torch.empty(M, D, **f32), h2=torch.empty(M, D, **f32),
            a=torch.empty(M, D, **f32), att=torch.empty(M, D, **f32), qkv=torch.empty(M, 3 * D, **f32),
            g=torch.empty(M, max(l["inner_pad"] for l in self.layers), **f32),
            four=torch.empty(D, **f32), temb=torch.empty(D, **f32),
            gb=torch.empty(self.depth * 4 * D, **f32), cos=cos_t.to(dev), sin=sin_t.to(dev))
        self._ws[key] = ws
        return ws

    @hip.on_device
    def ragged_workspace(self, frames):
        """Workspace for a ragged batch: clips of `frames` frames each packed back to back (M = sum rows, no padding).
        Carries the device segment table [n_seg][2] = (first row, rows) the seg kernels take."""
        frames = tuple(int(n) for n in frames)
        key = ("ragged",) + frames
        M, max_n = sum(frames), max(frames)
        if key in self._ws:
            small = self._ws[key]
        else:
            cos_t, sin_t = rotary_tables(self.inv_freq, max_n)      # positions restart in every clip
            starts = [0]
            for n in frames[:-1]:
                starts.append(starts[-1] + n)
            small = dict(cos=cos_t.to(self.device), sin=sin_t.to(self.device),
                         seg=torch.tensor([[s_, n] for s_, n in zip(starts, frames)], dtype=torch.int32).to(self.device),
                         frames=frames, rows=M, max_n=max_n)
            self._ws[key] = small           # (accounted with what it owns: the tables)
        # The row buffers are the ones of ONE clip of >= M frames, looked up on every call and never held by the
        # cached entry (an entry that kept them would pin workspace(1, M) for every mix of lengths, outside the byte
        # bound of the cache); M is rounded up so that mixes of similar total length share them.
        ws = dict(self.workspace(1, -(-M // 256) * 256))
        ws.update(small)
        return ws

    @hip.on_device
    def set_cond(self, cond, batch, n, ragged=None):
        """cond [B*n, dim_in] (log-mel of the low-res clip): e_cond = cond @ W_c^T + b."""
        ws = ragged if ragged is not None else self.workspace(batch, n)
        M = ragged["rows"] if ragged is not None else batch * n
        self.gemm(cond, self.w_c, ws["e_cond"], M, self.dim, self.dim_in, bias=self.b_embed)

    @hip.on_device
    def forward(self, x, t, out, batch, n, alpha=1.0, res=None, null_cond=False, ragged=None):
        """out = alpha * v(x, t) + res  with v the vector field; x/out/res [B*n, dim_in].
        null_cond=True evaluates v with every frame's condition replaced by `null_cond`
        (the cond_drop_prob = 1 pass of forward_with_cond_scale, flow.py:174-178,222-230).
        ragged: a ragged_workspace -- the rows are clips of different lengths packed back to back; everything
        row-wise runs as for one long clip, the three operators that look across rows (ConvPositionEmbed, rotary
        positions, attention) take the segment table: the reference's mask paths (transformer.py:35-44,
        attend.py:127-128) without the padding rows."""
        L, st = hip.lib(), hip.stream()
        ws = ragged if ragged is not None else self.workspace(batch, n)
        M, D = (ragged["rows"] if ragged is not None else batch * n), self.dim
        seg = ragged["seg"].data_ptr() if ragged is not None else None
        n_seg, max_n = (len(ragged["frames"]), ragged["max_n"]) if ragged is not None else (0, 0)
        h, h2, a, att, qkv = ws["h"], ws["h2"], ws["a"], ws["att"], ws["qkv"]
        if null_cond:
            if self._e_null is None:            # null_cond @ W_c^T + b: one row, broadcast with ldr = 0
                self._e_null = torch.empty(1, D, dtype=torch.float32, device=self.device)
                self.gemm(self.null_cond, self.w_c, self._e_null, 1, D, self.dim_in, bias=self.b_embed)
            self.gemm(x, self.w_x, h, M, D, self.dim_in, R=self._e_null, ldr=0)
        else:
            self.gemm(x, self.w_x, h, M, D, self.dim_in, R=ws["e_cond"])
        if seg is None:
            hip.check(L.fh_dwconv_gelu_res_f32(h.data_ptr(), self.dw_w.data_ptr(), self.dw_b.data_ptr(),
                                               h2.data_ptr(), batch, n, D, self.dw_k, st), "fh_dwconv_gelu_res_f32")
        else:
            hip.check(L.fh_dwconv_gelu_res_seg_f32(h.data_ptr(), self.dw_w.data_ptr(), self.dw_b.data_ptr(), h2.data_ptr(),
                                                   seg, n_seg, max_n, D, self.dw_k, st), "fh_dwconv_gelu_res_seg_f32")
        hip.check(L.fh_time_fourier_f32(self.sinu_w.data_ptr(), float(t), ws["four"].data_ptr(), D // 2, st),
                  "fh_time_fourier_f32")
        hip.check(L.fh_gemv_f32(self.t_w.data_ptr(), ws["four"].data_ptr(), self.t_b.data_ptr(),
                                ws["temb"].data_ptr(), D, D, 1, st), "fh_gemv_f32")
        hip.check(L.fh_gemv_f32(self.gb_w.data_ptr(), ws["temb"].data_ptr(), self.gb_b.data_ptr(),
                                ws["gb"].data_ptr(), self.gb_w.shape[0], D, 0, st), "fh_gemv_f32")
        gb = ws["gb"]
        cur, other = h2, h
        for li, lay in enumerate(self.layers):
            o = li * 4 * D
            g1, b1, g2, b2 = (gb[o + i * D: o + (i + 1) * D] for i in range(4))
            hip.check(L.fh_rmsnorm_f32(cur.data_ptr(), g1.data_ptr(), b1.data_ptr(), a.data_ptr(), M, D, st),
                      "fh_rmsnorm_f32")
            self.gemm(a, lay["w_qkv"], qkv, M, 3 * D, D)
            if seg is None:
                hip.check(L.fh_qknorm_rope_f32(qkv.data_ptr(), lay["gq"].data_ptr(), lay["gk"].data_ptr(),
                                               ws["cos"].data_ptr(), ws["sin"].data_ptr(), batch, n, self.heads, st),
                          "fh_qknorm_rope_f32")
                hip.check(L.fh_attention_f32(qkv.data_ptr(), att.data_ptr(), batch, n, self.heads, 10.0, st),
                          "fh_attention_f32")
            else:
                hip.check(L.fh_qknorm_rope_seg_f32(qkv.data_ptr(), lay["gq"].data_ptr(), lay["gk"].data_ptr(),
                                                   ws["cos"].data_ptr(), ws["sin"].data_ptr(), seg, n_seg, max_n,
                                                   self.heads, st), "fh_qknorm_rope_seg_f32")
                hip.check(L.fh_attention_seg_f32(qkv.data_ptr(), att.data_ptr(), seg, n_seg, max_n, self.heads, 10.0, st),
                          "fh_attention_seg_f32")
            self.gemm(att, lay["w_out"], other, M, D, D, R=cur)
            cur, other = other, cur
            hip.check(L.fh_rmsnorm_f32(cur.data_ptr(), g2.data_ptr(), b2.data_ptr(), a.data_ptr(), M, D, st),
                      "fh_rmsnorm_f32")
            ip = lay["inner_pad"]
            self.gemm(a, lay["w1"], ws["g"], M, 2 * ip, D, bias=lay["b1"], epilogue=hip.EPI_GEGLU, ldc=ws["g"].shape[1])
            self.gemm(ws["g"], lay["w2"], other, M, D, ip, bias=lay["b2"], R=cur, lda=ws["g"].shape[1])
            cur, other = other, cur
        hip.check(L.fh_rmsnorm_f32(cur.data_ptr(), self.final_gamma.data_ptr(), 0, a.data_ptr(), M, D, st),
                  "fh_rmsnorm_f32")
        self.gemm(a, self.w_pred, out, M, self.dim_in, D, R=res, alpha=alpha)
        return out
