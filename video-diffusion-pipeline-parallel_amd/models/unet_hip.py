"""SVD UNet forward on MI355X, expressed as a sequence of hand-written HIP kernels.

This replaces ``diffusers.UNetSpatioTemporalConditionModel.forward`` under the reference call
``self.unet(sample=..., timestep=..., encoder_hidden_states=..., added_time_ids=...)``
(``/root/reference/src/models/svd_unet.py:389,400,416``).  Nothing here runs a PyTorch operator on
activations: PyTorch allocates buffers and supplies the stream; every arithmetic op is a kernel of
``libsvdpipe_hip.so`` (``include/svdpipe.h``).

Data layout (differs from diffusers on purpose):
  * activations are ONE fp16 token matrix ``[B*F*H*W][C]`` (channels-last, frame-major).  Spatial
    transformer tokens, temporal-conv rows and 3x3-conv pixels are all the same memory, so none of the
    ~30 permute/reshape copies of the NCHW formulation exist; temporal attention / temporal conv address
    frame f of pixel p as row ``f*HW + p``.
  * every weight is fp16 ``[N][K]`` (K contiguous), conv kernels flattened tap-major.

Fusions baked into the kernel calls:
  * conv / linear epilogues add bias, the per-block time-embedding projection, one or two residuals and
    the AlphaBlender mix (``alpha*spatial + (1-alpha)*temporal`` is algebraically folded into the last
    GEMM of the temporal branch), so residual adds and blends never touch HBM separately;
  * GEGLU's ``value * gelu(gate)`` is formed in the projection GEMM's registers (weight rows are
    pre-interleaved), halving that GEMM's output traffic;
  * nearest x2 upsampling is folded into the following 3x3 convolution's gather;
  * Q, K, V projections run as one GEMM with a concatenated weight;
  * cross-attention attends to a SINGLE context token, so softmax == 1 exactly and the block reduces to
    ``to_out(to_v(ctx))`` broadcast over tokens: it is evaluated as two GEMVs per module per step and
    enters the self-attention output projection as an extra bias.  (The reference's Q/K projections for
    these modules cannot influence the result and are not executed; bench.py reports FLOPs accordingly.)
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch

from ..hip import ops
from . import weights as W
from .unet_spec import UNetConfig, up_block_plan


def _version(t):
    """In-place-write counter of a tensor (inference tensors keep none: -1)."""
    try:
        return t._version
    except RuntimeError:
        return -1


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class _Dense:
    """One contraction: packed fp16 weight ``[N][K]`` + fp32 bias, and how its A operand is gathered."""

    def __init__(self, w16, bias32, *, cin, mode=ops.A_LINEAR, n_true=None, geglu=False, colsum=None, ln_eps=None):
        self.w, self.bias, self.cin, self.mode = w16, bias32, cin, mode
        self.n = w16.shape[0]
        self.n_true = n_true if n_true is not None else (self.n // 2 if geglu else self.n)
        self.geglu = geglu
        self.colsum, self.ln_eps = colsum, ln_eps          # set when a LayerNorm is folded into this contraction

    @staticmethod
    def fold_layernorm(w, b, norm_w, norm_b, dev, *, eps, geglu=False):
        """``LN(x) @ W^T + b`` as a contraction on the un-normalised x: ``rstd*(x @ (W*gamma)^T - mean*colsum) + (W @ beta
        + b)`` with ``colsum[n] = sum_k (W*gamma)[n][k]`` taken over the fp16 values the kernel multiplies with."""
        w32, g32, be32 = w.to(dev).float(), norm_w.to(dev).float(), norm_b.to(dev).float()
        bias = w32 @ be32 + (b.to(dev).float() if b is not None else 0.0)
        wg = (w32 * g32[None, :])
        if geglu:
            wg, bias = W.interleave_geglu(wg, bias)
        wg16 = wg.to(torch.float16).contiguous()
        return _Dense(wg16, bias.float().contiguous(), cin=wg16.shape[1], geglu=geglu,
                      colsum=wg16.float().sum(dim=1).contiguous(), ln_eps=eps)

    @staticmethod
    def linear(sd, p, dev, bias=True):
        w = W.pack_linear(sd[p + ".weight"]).to(dev)
        return _Dense(w, _f32(sd[p + ".bias"], dev) if bias else None, cin=w.shape[1])

    @staticmethod
    def conv3x3(sd, p, dev):
        w = sd[p + ".weight"]
        cout, cin = w.shape[:2]
        npad, cpad = W.round_up(cout, 64), W.round_up(cin, 64)
        b = torch.zeros(npad, dtype=torch.float32, device=dev)
        b[:cout] = sd[p + ".bias"].to(dev).float()
        return _Dense(W.pack_conv3x3(w.to(dev), cpad, npad), b, cin=cpad, mode=ops.A_CONV3X3, n_true=cout)

    @staticmethod
    def tconv(sd, p, dev):
        w = sd[p + ".weight"]
        return _Dense(W.pack_tconv3(w.to(dev)), _f32(sd[p + ".bias"], dev), cin=w.shape[1], mode=ops.A_TEMPORAL3)

    @staticmethod
    def geglu_proj(sd, p, dev):
        wi, bi = W.interleave_geglu(sd[p + ".weight"].to(dev), sd[p + ".bias"].to(dev))
        return _Dense(wi, bi, cin=wi.shape[1], geglu=True)


class _Norm:
    def __init__(self, sd, p, dev, eps):
        self.g, self.b, self.eps = _f32(sd[p + ".weight"], dev), _f32(sd[p + ".bias"], dev), eps


@dataclass
class _Run:
    """Per-forward state handed down the module tree."""
    b: int
    f: int
    h: int
    w: int
    temb: torch.Tensor          # fp32 [n_temb_total] : every time_emb_proj(silu(emb)) of the net
    ctx16: torch.Tensor         # fp16 [B][cross_dim]
    gn_ws: torch.Tensor
    sk_ws: torch.Tensor         # fp32 scratch for split-K contractions of the few-row levels (None: no such level)
    frame_ids: torch.Tensor     # fp32 [F] = arange(F)
    cross: dict = None          # width C -> fp32 [modules][B][C]: to_out(to_v(ctx)) + b of every cross-attention of that width
    pos: dict = None            # width C -> fp16 [transformers][F][C]: frame position embeddings

    @property
    def hw(self):
        return self.h * self.w

    @property
    def m(self):
        return self.b * self.f * self.h * self.w


class SVDUNetHIP:
    # fp8 attention pays a quantise pass over q/k/v: below ~1k tokens per frame (the 576- and 144-token levels) the
    # fp16 kernel is faster (tools/bench_attn.py: 46 vs 56 us at 576 tokens, 1907 vs 1580 us at 9216)
    FP8_MIN_SEQ = 1024
    # few-row levels whose long-K contractions may be split over K (csrc/gemm.hip::SPLITK_MAX_ROWS)
    SPLITK_MAX_ROWS = 6144
    # with long_attention, rows this long go through the frozen-reference kernel (csrc/attention_long.hip;
    # tools/bench_attn_long.py: 1.03-1.09x of the ordinary kernel at 9,216 tokens, 0.9-0.97x at 4,096)
    LONG_ATTENTION_MIN_SEQ = 8192

    def __init__(self, cfg: UNetConfig, state_dict: dict, device, *, fp8_attention: bool | None = None,
                 long_attention: bool | None = None):
        """``fp8_attention``: run the spatial self-attention on fp8-e4m3 MFMA (BASELINE config 5: "SVD-XT ... with
        fp8 MFMA attention path"); default off, or ``VDPP_FP8_ATTN=1``.  Everything else stays fp16.
        ``long_attention``: level-0 rows (>= 8,192 tokens) through the frozen-reference kernel; default ON since round 4
        (``False`` or ``VDPP_LONG_ATTN=0`` selects the ordinary kernel for every row).  Same results; it is 5-9 % faster
        while no query meets, far from its own tokens, a key that scores 11 nats above everything near them, and costs at
        most ~1.1-1.25x the ordinary kernel on data where that happens all the time (after 64 flagged waves the remaining
        workgroups hand their blocks to the ordinary kernel at once: csrc/attention_long.hip, DESIGN.md section 3;
        ``tools/long_attn_flags.py`` counts the cases on a given set of weights)."""
        self.cfg = cfg
        self.fp8_attention = (os.environ.get("VDPP_FP8_ATTN") == "1") if fp8_attention is None else bool(fp8_attention)
        self.long_attention = (os.environ.get("VDPP_LONG_ATTN", "1") != "0") if long_attention is None else bool(long_attention)
        # the 16 transformer-entry GroupNorms folded into proj_in where a frame is whole tiles (VDPP_FOLD_GN=0: never)
        self.fold_groupnorm = os.environ.get("VDPP_FOLD_GN", "1") != "0"
        # statistics of a resnet's second norms (norm2 / temporal norm2: inputs without a residual) out of the producing
        # convolution's epilogue where a frame is whole 256-row tiles (VDPP_GN_EPILOGUE=0: always the statistics pass)
        self.gn_from_epilogue = os.environ.get("VDPP_GN_EPILOGUE", "1") != "0"
        # ... also for outputs that residuals are added to (the final tile is rebuilt in LDS and summed there); 0: only
        # the no-residual producers (sums straight from the accumulators)
        self.gn_epilogue_residual = os.environ.get("VDPP_GN_EPILOGUE_RES", "1") != "0"
        # a resnet's 1x1 shortcut convolution as an extra linear tap of its second 3x3 convolution (levels with more rows
        # than the split-K / small-tile routes take: the fused contraction runs on the 256-row ping-pong tiles)
        self.fold_shortcut = os.environ.get("VDPP_FOLD_SHORTCUT", "1") != "0"
        # LayerNorm row statistics from the producing contraction's epilogue also for rows of three / four column tiles
        # (1,280 channels at the 576-token level)
        self.ln_out_wide = os.environ.get("VDPP_LN_OUT_WIDE", "1") != "0"
        self.device = dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("SVDUNetHIP runs on an MI355X HIP device only (no CPU fallback)")
        ops.load()  # fail loudly now if the extension is missing
        sd = state_dict
        self._temb_w, self._temb_b, self._temb_n = [], [], 0
        self._xf_all = []
        boc = list(cfg.block_out_channels)
        g = cfg.norm_groups

        self.conv_in = _Dense.conv3x3(sd, "conv_in", dev)
        self.cin_pad = self.conv_in.cin
        # embedding MLPs: emb = te.l2(silu(te.l1(sin(t)))) + ae.l2(silu(ae.l1(sin(ids))))
        self.te1 = _Dense.linear(sd, "time_embedding.linear_1", dev)
        self.ae1 = _Dense.linear(sd, "add_embedding.linear_1", dev)
        w2 = torch.cat([sd["time_embedding.linear_2.weight"], sd["add_embedding.linear_2.weight"]], dim=1)
        self.emb2_w = W.pack_linear(w2).to(dev)                       # [temb][2*temb]
        self.emb2_b = _f32(sd["time_embedding.linear_2.bias"].float() + sd["add_embedding.linear_2.bias"].float(), dev)

        self.down = []
        ch = boc[0]
        for i, cout in enumerate(boc):
            res, att = [], []
            for j in range(cfg.layers_per_block):
                eps = 1e-6 if cfg.down_has_attn[i] else 1e-5
                res.append(self._resblock(sd, f"down_blocks.{i}.resnets.{j}", ch if j == 0 else cout, cout, eps))
                if cfg.down_has_attn[i]:
                    att.append(self._transformer(sd, f"down_blocks.{i}.attentions.{j}", cout))
            ds = _Dense.conv3x3(sd, f"down_blocks.{i}.downsamplers.0.conv", dev) if i != len(boc) - 1 else None
            self.down.append((res, att, ds))
            ch = cout
        self.mid = (self._resblock(sd, "mid_block.resnets.0", ch, ch, 1e-5),
                    self._transformer(sd, "mid_block.attentions.0", ch),
                    self._resblock(sd, "mid_block.resnets.1", ch, ch, 1e-5))
        self.up = []
        layers = cfg.layers_per_block + 1
        for i, (in_ch, out_ch, prev, attn, _h, ups) in enumerate(up_block_plan(cfg)):
            res, att = [], []
            for j in range(layers):
                skip = in_ch if j == layers - 1 else out_ch
                rin = prev if j == 0 else out_ch
                res.append(dict(self._resblock(sd, f"up_blocks.{i}.resnets.{j}", rin + skip, out_ch, 1e-6), cx=rin))
                if attn:
                    att.append(self._transformer(sd, f"up_blocks.{i}.attentions.{j}", out_ch))
            us = _Dense.conv3x3(sd, f"up_blocks.{i}.upsamplers.0.conv", dev) if ups else None
            self.up.append((res, att, us))
        self.norm_out = _Norm(sd, "conv_norm_out", dev, 1e-5)
        self.conv_out = _Dense.conv3x3(sd, "conv_out", dev)

        # all time_emb_proj layers as one [sum(Cout)][temb] GEMV
        self.temb_w = torch.cat(self._temb_w, dim=0).contiguous()
        self.temb_b = torch.cat(self._temb_b, dim=0).contiguous()
        del self._temb_w, self._temb_b
        self._group_small_gemvs()
        self._gn_ws = None
        self._sk_ws = {}
        self._pos_cache = {}
        self._fp8_ws = {}
        self._long_ws = {}

    def release_stream_state(self) -> None:
        """Drop every per-stream scratch buffer (GroupNorm partials, split-K slabs, fp8 operands).  They are keyed by
        the raw handle of the HIP stream a forward ran on, and a destroyed stream's handle can be handed out again: a
        caller that creates streams per run calls this when it retires them (``StableVideoUNet.clear_conditioning`` and
        ``enable_graphs(False)`` do), otherwise the streams must outlive the model.  The next forward on any stream
        simply allocates again."""
        self._gn_ws = None
        self._sk_ws = {}
        self._fp8_ws = {}
        self._long_ws = {}

    # ------------------------------------------------------------------ weight packing
    def _reg_temb(self, sd, p):
        w = W.pack_linear(sd[p + ".weight"]).to(self.device)
        off = self._temb_n
        self._temb_w.append(w)
        self._temb_b.append(_f32(sd[p + ".bias"], self.device))
        self._temb_n += w.shape[0]
        return off

    def _resblock(self, sd, p, cin, cout, eps):
        dev = self.device
        s, t = p + ".spatial_res_block", p + ".temporal_res_block"
        alpha = float(torch.sigmoid(sd[p + ".time_mixer.mix_factor"].float()).item())
        return dict(
            cin=cin, cout=cout, alpha=alpha,
            n1=_Norm(sd, s + ".norm1", dev, eps), c1=_Dense.conv3x3(sd, s + ".conv1", dev),
            te_s=self._reg_temb(sd, s + ".time_emb_proj"),
            n2=_Norm(sd, s + ".norm2", dev, eps), c2=_Dense.conv3x3(sd, s + ".conv2", dev),
            sc=_Dense.linear(sd, s + ".conv_shortcut", dev) if cin != cout else None,
            c2sc=self._conv_plus_shortcut(sd, s, dev) if cin != cout else None,
            tn1=_Norm(sd, t + ".norm1", dev, eps), tc1=_Dense.tconv(sd, t + ".conv1", dev),
            te_t=self._reg_temb(sd, t + ".time_emb_proj"),
            tn2=_Norm(sd, t + ".norm2", dev, eps), tc2=_Dense.tconv(sd, t + ".conv2", dev),
        )

    @staticmethod
    def _conv_plus_shortcut(sd, s, dev):
        """conv2(h) + conv_shortcut(x) as ONE contraction: weight [N][9*C | Cin] (the shortcut's columns behind the nine
        taps), bias b2 + b_sc.  None where the widths do not fit the extra-tap kernel (Cin a multiple of 64, no padding)."""
        w2, wsc = sd[s + ".conv2.weight"], sd[s + ".conv_shortcut.weight"]
        cout, c = w2.shape[:2]
        cin = wsc.shape[1]
        if cout % 64 or c % 64 or cin % 64 or not (cout % 256 == 0 or cout % 320 == 0):
            return None
        w = torch.cat([W.pack_conv3x3(w2.to(dev), c, cout), W.pack_linear(wsc).to(dev)], dim=1).contiguous()
        b = (sd[s + ".conv2.bias"].float() + sd[s + ".conv_shortcut.bias"].float()).to(dev).contiguous()
        layer = _Dense(w, b, cin=c, mode=ops.A_CONV3X3, n_true=cout)
        layer.cin2 = cin
        return layer

    def _attn(self, sd, p, dev, norm):
        """``norm``: state_dict prefix of the LayerNorm in front of the Q/K/V projections (folded into them)."""
        qkv = torch.cat([sd[p + ".to_q.weight"], sd[p + ".to_k.weight"], sd[p + ".to_v.weight"]], dim=0)
        return dict(qkv=_Dense.fold_layernorm(qkv, None, sd[norm + ".weight"], sd[norm + ".bias"], dev, eps=1e-5),
                    out=_Dense.linear(sd, p + ".to_out.0", dev))

    def _geglu_ln(self, sd, p, dev, norm):
        return _Dense.fold_layernorm(sd[p + ".weight"], sd[p + ".bias"], sd[norm + ".weight"], sd[norm + ".bias"], dev,
                                     eps=1e-5, geglu=True)

    def _xattn(self, sd, p, dev):
        return dict(v=W.pack_linear(sd[p + ".to_v.weight"]).to(dev), o=W.pack_linear(sd[p + ".to_out.0.weight"]).to(dev),
                    ob=_f32(sd[p + ".to_out.0.bias"], dev))

    def _transformer(self, sd, p, c):
        xf = self._transformer_params(sd, p, c)
        self._xf_all.append(xf)
        return xf

    def _group_small_gemvs(self):
        """The 32 single-token cross-attention modules and the 16 frame-position MLPs are M <= 25 GEMVs: stack their
        weights by width so that a forward issues 5 batched launches per width instead of ~110 tiny ones."""
        self._small = {}
        by_c = {}
        for xf in self._xf_all:
            by_c.setdefault(xf["c"], []).append(xf)
        for c, xfs in by_c.items():
            xs = [m for xf in xfs for m in (xf["s_x"], xf["t_x"])]
            sm = dict(G=len(xs), T=len(xfs),
                      Wv=torch.stack([m["v"] for m in xs]).contiguous(), Wo=torch.stack([m["o"] for m in xs]).contiguous(),
                      bo=torch.stack([m["ob"] for m in xs]).contiguous(),
                      W1=torch.stack([xf["pe1"].w for xf in xfs]).contiguous(),
                      b1=torch.stack([xf["pe1"].bias for xf in xfs]).contiguous(),
                      W2=torch.stack([xf["pe2"].w for xf in xfs]).contiguous(),
                      b2=torch.stack([xf["pe2"].bias for xf in xfs]).contiguous())
            for g, m in enumerate(xs):
                m.clear()
                m["grp"] = (c, g)
            for t, xf in enumerate(xfs):
                del xf["pe1"], xf["pe2"]
                xf["pos_idx"] = t
            self._small[c] = sm
        del self._xf_all

    def _small_gemvs(self, r: _Run):
        """Per forward: every cross-attention vector and every frame position embedding (see _group_small_gemvs)."""
        dev = self.device
        cross = r.ctx16.shape[1]
        r.cross, r.pos = {}, {}
        for c, sm in self._small.items():
            g, t = sm["G"], sm["T"]
            v16 = torch.empty((g, r.b, c), dtype=torch.float16, device=dev)
            ops.gemv_batched(r.ctx16, sm["Wv"], None, batch=g, n=c, k=cross, rows=r.b, x_stride=0, y16=v16)
            cv = torch.empty((g, r.b, c), dtype=torch.float32, device=dev)
            ops.gemv_batched(v16, sm["Wo"], sm["bo"], batch=g, n=c, k=c, rows=r.b, y32=cv)
            r.cross[c] = cv
        # The frame position embeddings are a function of the weights and the frame count only (diffusers evaluates
        # time_pos_embed(time_proj(arange(F))) in every forward): evaluated once per frame count, kept.
        pos = self._pos_cache.get(r.f)
        if pos is not None:
            r.pos = pos
            return
        for c, sm in self._small.items():
            t = sm["T"]
            sin = torch.empty((r.f, c), dtype=torch.float16, device=dev)
            ops.sinusoid(r.frame_ids, sin, r.f, c)
            pe_h = torch.empty((t, r.f, 4 * c), dtype=torch.float16, device=dev)
            ops.gemv_batched(sin, sm["W1"], sm["b1"], batch=t, n=4 * c, k=c, rows=r.f, x_stride=0, y16=pe_h, silu_out=True)
            pe = torch.empty((t, r.f, c), dtype=torch.float16, device=dev)
            ops.gemv_batched(pe_h, sm["W2"], sm["b2"], batch=t, n=c, k=4 * c, rows=r.f, y16=pe)
            r.pos[c] = pe
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(dev).synchronize()      # complete before a forward on another stream may read it
            self._pos_cache[r.f] = r.pos

    def _transformer_params(self, sd, p, c):
        dev = self.device
        b, t = p + ".transformer_blocks.0", p + ".temporal_transformer_blocks.0"
        alpha = float(torch.sigmoid(sd[p + ".time_mixer.mix_factor"].float()).item())
        return dict(
            c=c, heads=c // 64, alpha=alpha,
            norm=_Norm(sd, p + ".norm", dev, 1e-6), pin=_Dense.linear(sd, p + ".proj_in", dev),
            pout=_Dense.linear(sd, p + ".proj_out", dev),
            pe1=_Dense.linear(sd, p + ".time_pos_embed.linear_1", dev),
            pe2=_Dense.linear(sd, p + ".time_pos_embed.linear_2", dev),
            # the five LayerNorms of a transformer sit in front of a Q/K/V or GEGLU projection and are folded into it
            s_attn=self._attn(sd, b + ".attn1", dev, b + ".norm1"),
            s_x=self._xattn(sd, b + ".attn2", dev),
            s_ff1=self._geglu_ln(sd, b + ".ff.net.0.proj", dev, b + ".norm3"),
            s_ff2=_Dense.linear(sd, b + ".ff.net.2", dev),
            t_fi1=self._geglu_ln(sd, t + ".ff_in.net.0.proj", dev, t + ".norm_in"),
            t_fi2=_Dense.linear(sd, t + ".ff_in.net.2", dev),
            t_attn=self._attn(sd, t + ".attn1", dev, t + ".norm1"),
            t_x=self._xattn(sd, t + ".attn2", dev),
            t_ff1=self._geglu_ln(sd, t + ".ff.net.0.proj", dev, t + ".norm3"),
            t_ff2=_Dense.linear(sd, t + ".ff.net.2", dev),
        )

    # ------------------------------------------------------------------ kernel helpers
    def _buf(self, rows, c):
        return torch.empty((rows, c), dtype=torch.float16, device=self.device)

    def _gemm(self, r: _Run, layer: _Dense, a, *, m=None, conv=None, **kw):
        m = r.m if m is None else m
        out = kw.pop("out", None)
        # ``ln_next``: the contraction that will consume this output through a folded LayerNorm.  Where one tile spans a
        # whole output row (256 / 320 channels: level 0) the epilogue leaves that LayerNorm's (mean, rstd) beside the
        # output and _ln_stats finds them there instead of reading the tensor again (512 / 640 channels, level 1: two tiles
        # per row, their sums meet in a 16-byte-per-row scratch).
        ln_next = kw.pop("ln_next", None)
        # ``w_groups`` = (w_f [instances][n][c] fp16, bias_f [instances][n] fp32, rows per instance): a GroupNorm folded
        # into this linear layer (ops.groupnorm_fold_linear) -- every instance's rows meet their own scaled weights
        w_groups = kw.pop("w_groups", None)
        weight, bias = layer.w, layer.bias
        if w_groups is not None:
            weight, bias = w_groups[0], None
            kw.update(bias2=w_groups[1], bias2_rows=w_groups[2], w_group_rows=w_groups[2],
                      w_group_stride=w_groups[0].shape[1] * w_groups[0].shape[2])
        st = None
        ws = r.sk_ws if m <= self.SPLITK_MAX_ROWS else None
        # ``gn_next``: this output goes straight into a GroupNorm (no residual in between): where the tiles allow it the
        # epilogue leaves per-tile column sums beside the output and _gn folds them instead of reading the tensor again
        gn_part = None
        if kw.pop("gn_next", False) and self.gn_from_epilogue and m % 256 == 0 and layer.n == layer.n_true \
                and not layer.geglu and (layer.n % 320 == 0 or layer.n % 256 == 0) and r.hw % 256 == 0 \
                and layer.colsum is None and ln_next is None and "euler" not in kw and w_groups is None \
                and (self.gn_epilogue_residual or (kw.get("res1") is None and kw.get("res2") is None)):
            gn_part = torch.empty((m // 256, 2, layer.n, 2), dtype=torch.float32, device=self.device)
            kw.update(gn_part=gn_part)
        st_buf = kw.pop("ln_out_buf", None)             # caller-provided rows of a larger statistics tensor (_ff_pair chunks)
        if ln_next is not None and self._ln_out_ok(layer, m) and "euler" not in kw:
            st = st_buf if st_buf is not None else torch.empty((m, 2), dtype=torch.float32, device=self.device)
            kw.update(ln_out=st, ln_out_eps=ln_next.ln_eps)
            tiles = layer.n_true // (320 if layer.n_true % 320 == 0 else 256)
            ws = torch.empty((m, 2 * tiles), dtype=torch.float32, device=self.device) if tiles > 1 else None
        if out is None:
            out = self._buf(m, layer.n_true)
        elif out.shape != (m, layer.n_true):
            raise RuntimeError(f"destination {tuple(out.shape)} does not match the contraction's output ({m}, {layer.n_true})")
        # operands may be column slices of wider row-major buffers (the halves of a concatenation buffer): row pitches
        # come from the tensors
        for key, ld in (("res1", "ldr1"), ("res2", "ldr2")):
            if kw.get(key) is not None:
                kw.setdefault(ld, kw[key].stride(0))
        n_store = layer.n_true if (layer.n_true != (layer.n // 2 if layer.geglu else layer.n)) else 0
        temporal = (r.f, r.hw) if layer.mode == ops.A_TEMPORAL3 else None
        if layer.colsum is not None and "ln_stats" not in kw:
            raise RuntimeError("this contraction carries a folded LayerNorm: pass ln_stats")
        ops.gemm(a, weight, out, m=m, n=layer.n, cin=layer.cin, mode=layer.mode, conv=conv, temporal=temporal,
                 bias=bias, geglu=layer.geglu, n_store=n_store, ldd=out.stride(0), lda=a.stride(0),
                 ln_colsum=layer.colsum, workspace=ws, **kw)
        if gn_part is not None:
            out._gn_tile_sums = (gn_part, out.data_ptr(), _version(out), tuple(out.shape))
        if st is not None and st_buf is None:
            # valid for exactly this tensor object in exactly this state (checked in _ln_stats): a view, a slice or an
            # in-place write after the contraction silently falls back to the statistics pass
            out._row_ln_stats = (st, ln_next.ln_eps, out.data_ptr(), _version(out), tuple(out.shape))
        return out

    def _ln_out_ok(self, layer: _Dense, m: int) -> bool:
        """Can this contraction's epilogue leave the next LayerNorm's row statistics?  A row = one or two column tiles at
        any row count; three or four (1,024 / 1,280 channels: the 576-token level) only where the large ping-pong tiles
        are what the contraction runs on anyway (more rows than the small-tile / split-K routes take)."""
        if layer.n != layer.n_true or layer.geglu:
            return False
        return layer.n_true in (256, 320, 512, 640) or (self.ln_out_wide and layer.n_true in (768, 960, 1024, 1280)
                                                        and m > self.SPLITK_MAX_ROWS)

    def _ln_stats(self, layer: _Dense, x, **kw):
        """(mean, rstd) per row of x for the LayerNorm folded into ``layer``."""
        have = getattr(x, "_row_ln_stats", None)
        if (have is not None and not kw and have[1] == layer.ln_eps and have[2] == x.data_ptr() and have[3] == _version(x)
                and have[4] == tuple(x.shape)):
            return have[0]                                # left there by the contraction that produced x
        st = torch.empty((x.shape[0], 2), dtype=torch.float32, device=self.device)
        ops.ln_stats(x, st, rows=x.shape[0], c=x.shape[1], eps=layer.ln_eps, **kw)
        return st

    @staticmethod
    def _tile_sums(x, rows):
        """The per-tile column sums the contraction that produced x left beside it (``_gemm(gn_next=True)``), if they are
        for exactly this tensor in this state and an instance of ``rows`` rows is whole 256-row tiles; else None."""
        have = getattr(x, "_gn_tile_sums", None)
        if (have is not None and have[1] == x.data_ptr() and have[2] == _version(x) and have[3] == tuple(x.shape)
                and rows % 256 == 0):
            return have[0]
        return None

    def _gn(self, r: _Run, norm: _Norm, x, *, temporal: bool, silu: bool, concat_sums=None):
        """``concat_sums`` = (sums of the left columns, sums of the right columns): x is a concatenation buffer whose two
        halves were written by two contractions that both left their column sums (an up block's first norm)."""
        c = x.shape[1]
        inst, rows = (r.b, r.f * r.hw) if temporal else (r.b * r.f, r.hw)
        y = self._buf(x.shape[0], c)
        if concat_sums is not None and concat_sums[0] is not None and concat_sums[1] is not None and rows % 256 == 0:
            (pa, _pa, _va, sa), (pb, _pb, _vb, sb) = concat_sums
            if (sa[0] == x.shape[0] == sb[0] and sa[1] + sb[1] == c and x.stride(0) == c and _pa == x.data_ptr()
                    and _pb == x.data_ptr() + 2 * sa[1] and _va == _version(x) == _vb):
                stats = torch.empty((inst, self.cfg.norm_groups, 2), dtype=torch.float32, device=self.device)
                ops.groupnorm_tile_sums(x, pa, norm.g, norm.b, y, instances=inst, rows=rows, c=c, groups=self.cfg.norm_groups,
                                        eps=norm.eps, silu=silu, stats=stats, ldx=c, part_b=pb, c_a=sa[1])
                return y
        part = self._tile_sums(x, rows)
        if part is not None:
            stats = torch.empty((inst, self.cfg.norm_groups, 2), dtype=torch.float32, device=self.device)
            ops.groupnorm_tile_sums(x, part, norm.g, norm.b, y, instances=inst, rows=rows, c=c,
                                    groups=self.cfg.norm_groups, eps=norm.eps, silu=silu, stats=stats, ldx=x.stride(0))
            return y
        ops.groupnorm(x, norm.g, norm.b, y, instances=inst, rows=rows, c=c, groups=self.cfg.norm_groups,
                      eps=norm.eps, silu=silu, ws=r.gn_ws, ldx=x.stride(0))
        return y

    def _ln(self, norm: _Norm, x, **kw):
        y = self._buf(*x.shape)
        ops.layernorm(x, norm.g, norm.b, y, rows=x.shape[0], c=x.shape[1], eps=norm.eps, **kw)
        return y

    def _conv_geom(self, r: _Run, stride=1, ups=0):
        hv, wv = r.h << ups, r.w << ups
        ho, wo = (hv - 1) // stride + 1, (wv - 1) // stride + 1
        return (r.b * r.f, r.h, r.w, ho, wo, stride, ups), ho, wo

    # ------------------------------------------------------------------ blocks
    def _run_resblock(self, r: _Run, p, x, out=None, gn_next=False, concat_sums=None):
        """``out``: where the block's result goes (a half of a concatenation buffer), default a fresh tensor.
        ``gn_next``: the result goes straight into a GroupNorm (a transformer's entry norm, ``conv_norm_out``): ask the last
        contraction for the column sums."""
        geom, _, _ = self._conv_geom(r)
        t = self._gn(r, p["n1"], x, temporal=False, silu=True, concat_sums=concat_sums)
        n1 = p["c1"].n
        t = self._gemm(r, p["c1"], t, conv=geom, bias2=r.temb[p["te_s"]:p["te_s"] + n1], bias2_rows=r.m, gn_next=True)
        t = self._gn(r, p["n2"], t, temporal=False, silu=True)
        if p["sc"] is not None and p["c2sc"] is not None and self.fold_shortcut and r.m > self.SPLITK_MAX_ROWS \
                and r.m % 256 == 0:
            # the shortcut convolution rides in conv2's K loop (sp_gemm_desc.a2): no skip tensor, and no residual in front of
            # the temporal block's first norm (its column sums then come straight from the accumulators)
            s = self._gemm(r, p["c2sc"], t, conv=geom, a2=x, cin2=p["c2sc"].cin2, lda2=x.stride(0), gn_next=True)
        else:
            skip = x if p["sc"] is None else self._gemm(r, p["sc"], x)
            s = self._gemm(r, p["c2"], t, conv=geom, res1=skip, r1scale=1.0, gn_next=True)
        # temporal branch + AlphaBlender: alpha*s + (1-alpha)*(s + conv2(...)) = s + (1-alpha)*conv2(...)
        t = self._gn(r, p["tn1"], s, temporal=True, silu=True)
        t = self._gemm(r, p["tc1"], t, bias2=r.temb[p["te_t"]:p["te_t"] + p["cout"]], bias2_rows=r.m, gn_next=True)
        t = self._gn(r, p["tn2"], t, temporal=True, silu=True)
        return self._gemm(r, p["tc2"], t, oscale=1.0 - p["alpha"], res1=s, r1scale=1.0, out=out, gn_next=gn_next)

    def _cross_vec(self, r: _Run, x):
        """to_out(to_v(ctx)) + b_out for the single context token -> fp32 [B][C] (computed by _small_gemvs)."""
        c, g = x["grp"]
        return r.cross[c][g]

    def _self_attn(self, r: _Run, att, xvec, resid, *, temporal: bool, **epi):
        """LayerNorm (folded into the fused Q/K/V projection) -> self-attention -> output projection + residual."""
        c = att["out"].n
        heads = c // 64
        qkv = self._gemm(r, att["qkv"], resid, ln_stats=self._ln_stats(att["qkv"], resid))
        o = self._buf(r.m, c)
        q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
        if temporal:
            ops.attn_temporal(q, k, v, o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=r.b, frames=r.f, hw=r.hw,
                              heads=heads)
        elif self.fp8_attention and r.hw >= self.FP8_MIN_SEQ:
            need = ops.attn_fp8_ws_bytes(r.b * r.f, r.hw, heads)
            skey = torch.cuda.current_stream(self.device).cuda_stream     # one scratch per HIP stream
            ws = self._fp8_ws.get(skey)
            if ws is None or ws.numel() < need:
                ws = self._fp8_ws[skey] = torch.empty(need, dtype=torch.uint8, device=self.device)
            ops.attn_spatial_fp8(q, k, v, o, ws, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=r.b * r.f, seq=r.hw,
                                 heads=heads)
        elif self.long_attention and r.hw >= self.LONG_ATTENTION_MIN_SEQ and r.hw % 256 == 0:
            need = ops.attn_long_ws_bytes(r.b * r.f, r.hw, heads)
            skey = torch.cuda.current_stream(self.device).cuda_stream     # one set of flag words per HIP stream
            ws = self._long_ws.get(skey)
            if ws is None or ws.numel() < need:
                ws = self._long_ws[skey] = torch.empty(need, dtype=torch.uint8, device=self.device)
            ops.attn_spatial_long(q, k, v, o, ws, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=r.b * r.f, seq=r.hw,
                                  heads=heads)
        else:
            ops.attn_spatial(q, k, v, o, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, batch=r.b * r.f, seq=r.hw,
                             heads=heads)
        # self-attn out-proj + residual, with the (token independent) cross-attention result as extra bias
        return self._gemm(r, att["out"], o, bias2=self._cross_vec(r, xvec), bias2_rows=r.f * r.hw, res1=resid,
                          r1scale=1.0, **epi)

    # GroupNorm -> proj_in fold: the per-frame weight copies must stay small next to the pass they replace
    GN_FOLD_MAX_WEIGHT_BYTES = 64 << 20

    def _proj_in(self, r: _Run, p, x):
        """``proj_in(norm(x))`` of a transformer.  The GroupNorm has no activation and feeds a linear layer, so where a
        frame's rows are whole tiles (the 9,216- and 2,304-token levels) it is folded into per-frame weights
        (``sp_groupnorm_fold_linear_f16``): one statistics pass over x instead of a statistics pass, an apply pass and the
        normalised tensor's round trip through HBM.  Elsewhere: GroupNorm kernel + plain contraction."""
        c, inst = p["c"], r.b * r.f
        if (self.fold_groupnorm and r.hw % 256 == 0 and (c % 256 == 0 or c % 320 == 0)          # whole ping-pong tiles
                and inst * c * c * 2 <= self.GN_FOLD_MAX_WEIGHT_BYTES):
            w_f = torch.empty((inst, c, c), dtype=torch.float16, device=self.device)
            b_f = torch.empty((inst, c), dtype=torch.float32, device=self.device)
            part = self._tile_sums(x, r.hw)
            if part is not None:       # the resnet's last contraction left the column sums: no statistics pass over x
                stats = torch.empty((inst, self.cfg.norm_groups, 2), dtype=torch.float32, device=self.device)
                ops.groupnorm_fold_linear_tile_sums(part, p["norm"].g, p["norm"].b, p["pin"].w, p["pin"].bias, w_f, b_f,
                                                    instances=inst, rows=r.hw, c=c, groups=self.cfg.norm_groups,
                                                    eps=p["norm"].eps, n=c, stats=stats)
            else:
                ops.groupnorm_fold_linear(x, p["norm"].g, p["norm"].b, p["pin"].w, p["pin"].bias, w_f, b_f, instances=inst,
                                          rows=r.hw, c=c, groups=self.cfg.norm_groups, eps=p["norm"].eps, n=c, ws=r.gn_ws,
                                          ldx=x.stride(0))
            return self._gemm(r, p["pin"], x, ln_next=p["s_attn"]["qkv"], w_groups=(w_f, b_f, r.hw))
        t = self._gn(r, p["norm"], x, temporal=False, silu=False)
        return self._gemm(r, p["pin"], t, ln_next=p["s_attn"]["qkv"])

    # GEGLU feed-forward pairs in row chunks (VDPP_FF_CHUNK_MB, default 0 = whole tensor): the hidden activation of a
    # level-0 / level-1 feed-forward (660 / 330 MB for two videos) is written by FF1 and read once by FF2; cut into row
    # chunks whose hidden slice fits the 256 MiB Infinity Cache with room to spare, FF2 finds it there instead of in HBM.
    # Chunks are whole rounds of 256-row FF2 tiles (256 tiles x tiles_n) so that no round of workgroups is cut short.
    FF_CHUNK_BYTES = int(os.environ.get("VDPP_FF_CHUNK_MB", "0")) << 20
    FF_CHUNK_ROUND = None            # tests: rows a chunk is rounded to (default: one full round of FF2 workgroups)

    def _ff_pair(self, r: _Run, ff1: _Dense, ff2: _Dense, x, st, **epi):
        """``ff2(geglu(ff1(LN(x))))`` with ff2's epilogue arguments ``epi`` (residuals are row-sliced along)."""
        m, hid = x.shape[0], ff1.n_true
        budget = self.FF_CHUNK_BYTES
        if not budget or m * hid * 2 <= budget:
            g = self._gemm(r, ff1, x, ln_stats=st)
            return self._gemm(r, ff2, g, **epi)
        tiles_n = max(1, ff2.n // 320 if ff2.n % 320 == 0 else ff2.n // 256)
        rows_round = self.FF_CHUNK_ROUND or 256 * max(1, 256 // tiles_n)   # rows of one full round of FF2 workgroups
        rows = max(rows_round, (budget // (hid * 2)) // rows_round * rows_round)
        out = epi.pop("out", None)
        if out is None:
            out = self._buf(m, ff2.n_true)
        ln_next = epi.pop("ln_next", None)
        st_next = None
        if ln_next is not None and self._ln_out_ok(ff2, m):
            st_next = torch.empty((m, 2), dtype=torch.float32, device=self.device)
        for r0 in range(0, m, rows):
            r1 = min(m, r0 + rows)
            kw = {k: (v[r0:r1] if k in ("res1", "res2") and v is not None else v) for k, v in epi.items()}
            if st_next is not None:
                kw.update(ln_next=ln_next, ln_out_buf=st_next[r0:r1])
            g = self._gemm(r, ff1, x[r0:r1], m=r1 - r0, ln_stats=st[r0:r1])
            self._gemm(r, ff2, g, m=r1 - r0, out=out[r0:r1], **kw)
            del g
        if st_next is not None:
            out._row_ln_stats = (st_next, ln_next.ln_eps, out.data_ptr(), _version(out), tuple(out.shape))
        return out

    def _run_transformer(self, r: _Run, p, x, out=None, gn_next=False):
        c, a = p["c"], p["alpha"]
        hs = self._proj_in(r, p, x)
        # --- spatial block
        hs1 = self._self_attn(r, p["s_attn"], p["s_x"], hs, temporal=False, ln_next=p["s_ff1"])
        hs_s = self._ff_pair(r, p["s_ff1"], p["s_ff2"], hs1, self._ln_stats(p["s_ff1"], hs1), res1=hs1, r1scale=1.0)
        # --- frame positional embedding (B*F rows)
        pe = r.pos[c][p["pos_idx"]]
        if r.b > 1:
            pe = pe.repeat(r.b, 1)
        # --- temporal block on hmix = hs_s + pe[frame]
        hmix = self._buf(r.m, c)
        st = self._ln_stats(p["t_fi1"], hs_s, addvec=pe, addvec_rows=r.hw, sum_out=hmix)   # also writes hmix = hs_s + pe
        ht = self._ff_pair(r, p["t_fi1"], p["t_fi2"], hmix, st, res1=hmix, r1scale=1.0, ln_next=p["t_attn"]["qkv"])
        del hmix, st
        ht1 = self._self_attn(r, p["t_attn"], p["t_x"], ht, temporal=True, ln_next=p["t_ff1"])
        # temporal out = ff(..)+ht1 ; blend = a*hs_s + (1-a)*temporal out   (folded into the epilogue)
        mix = self._ff_pair(r, p["t_ff1"], p["t_ff2"], ht1, self._ln_stats(p["t_ff1"], ht1), oscale=1.0 - a, res1=ht1,
                            r1scale=1.0 - a, res2=hs_s, r2scale=a)
        return self._gemm(r, p["pout"], mix, res1=x, r1scale=1.0, out=out, gn_next=gn_next)

    # ------------------------------------------------------------------ forward
    def forward_rows(self, x_rows, *, b, frames, h, w, t_value, ctx16, added_ids32, euler=None):
        """x_rows: fp16 [B*F*H*W][cin_pad] (see ``sp_pack_input_f16``); ``t_value``: fp32 device tensor [1]
        (continuous timestep); ``ctx16``: fp16 [B][cross_dim]; ``added_ids32``: fp32 device [3].
        Returns eps rows fp16 [B*F*H*W][out_channels].  With ``euler`` (dict: latent, out, sigma, sigma_next and
        optionally eps_uncond + guidance + ld_eps) the last convolution's epilogue applies the guidance mix and the
        Euler update itself (ref svd_unet.py:410-439): ``euler["out"]`` receives the new latent, the
        eps rows are not written and ``None`` is returned."""
        cfg, dev = self.cfg, self.device
        if h % 8 or w % 8:
            raise ValueError("latent height/width must be multiples of 8 (three stride-2 levels)")
        temb_dim, c0 = cfg.time_embed_dim, cfg.block_out_channels[0]
        max_c = 2 * max(cfg.block_out_channels)
        ws_bytes = ops.groupnorm_ws_bytes(b * frames, frames * h * w, max_c, cfg.norm_groups)
        # GroupNorm partial-sum scratch: one per HIP stream, forwards on different streams may overlap in time
        if self._gn_ws is None:
            self._gn_ws = {}
        skey = torch.cuda.current_stream(dev).cuda_stream
        gn_ws = self._gn_ws.get(skey)
        if gn_ws is None or gn_ws.numel() < ws_bytes:
            gn_ws = self._gn_ws[skey] = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

        # split-K scratch for the levels with at most SPLITK_MAX_ROWS rows (up to 8 fp32 slabs of rows x channels), per stream
        sk_bytes = 0
        for lvl, c in enumerate(cfg.block_out_channels):
            rows = b * frames * (h >> min(lvl, 3)) * (w >> min(lvl, 3))
            if rows <= self.SPLITK_MAX_ROWS:
                sk_bytes = max(sk_bytes, 8 * rows * c * 4)
        sk_ws = self._sk_ws.get(skey)
        if sk_bytes and (sk_ws is None or sk_ws.numel() < sk_bytes):
            sk_ws = self._sk_ws[skey] = torch.empty(sk_bytes, dtype=torch.uint8, device=dev)

        # ---- embeddings (M = 1 GEMVs)
        hid = torch.empty((1, 2 * temb_dim), dtype=torch.float16, device=dev)
        tsin = torch.empty((1, c0), dtype=torch.float16, device=dev)
        ops.sinusoid(t_value, tsin, 1, c0)
        ops.gemv(tsin, self.te1.w, self.te1.bias, n=temb_dim, k=c0, y16=hid, ldy=2 * temb_dim, silu_out=True)
        asin = torch.empty((1, cfg.projection_class_embeddings_input_dim), dtype=torch.float16, device=dev)
        ops.sinusoid(added_ids32, asin, 3, cfg.addition_time_embed_dim)
        ops.gemv(asin, self.ae1.w, self.ae1.bias, n=temb_dim, k=asin.shape[1], y16=hid[:, temb_dim:],
                 ldy=2 * temb_dim, silu_out=True)
        emb16 = torch.empty((1, temb_dim), dtype=torch.float16, device=dev)
        ops.gemv(hid, self.emb2_w, self.emb2_b, n=temb_dim, k=2 * temb_dim, y16=emb16)
        temb = torch.empty(self.temb_w.shape[0], dtype=torch.float32, device=dev)
        ops.gemv(emb16, self.temb_w, self.temb_b, n=self.temb_w.shape[0], k=temb_dim, y32=temb, silu_in=True)

        r = _Run(b=b, f=frames, h=h, w=w, temb=temb, ctx16=ctx16, gn_ws=gn_ws, sk_ws=sk_ws,
                 frame_ids=torch.arange(frames, dtype=torch.float32, device=dev))
        self._small_gemvs(r)

        # No torch.cat (diffusers' up blocks concatenate the running tensor with a skip in front of every resnet): the
        # buffer [rows][Cx + Cskip] an up resnet reads is allocated when its skip is PRODUCED in the down path; the skip's
        # producer writes the right-hand columns in place, and the up path's producer of the running tensor (mid block,
        # previous resnet / transformer / upsampler) writes the left-hand columns.  Until then the down path keeps
        # working on the skip through its strided view.
        cx_pop = [p["cx"] for res, _, _ in self.up for p in res]           # in the order the up path consumes skips
        cats = []                                                          # in the order the down path produces them
        cat_sums = []                                                      # per buffer: column sums of its [left, right] halves

        def skip_dest(rows, cskip):
            cx = cx_pop[len(cx_pop) - 1 - len(cats)]
            cats.append(self._buf(rows, cx + cskip))
            cat_sums.append([None, None])
            return cats[-1][:, cx:]

        def note_skip(t):                      # the skip's producer may have left its column sums beside the view it wrote
            cat_sums[-1][1] = getattr(t, "_gn_tile_sums", None)
            return t

        def note_x(t):                         # ... and so may the producer of the running tensor's half
            if cats and t is not None:
                cat_sums[-1][0] = getattr(t, "_gn_tile_sums", None)
            return t

        geom, _, _ = self._conv_geom(r)
        x = note_skip(self._gemm(r, self.conv_in, x_rows, conv=geom, out=skip_dest(r.m, self.conv_in.n_true), gn_next=True))
        for res, att, ds in self.down:
            for j, p in enumerate(res):
                if att:
                    x = self._run_resblock(r, p, x, gn_next=True)          # -> the transformer's entry norm
                    # -> the next resnet's norm1, and (as a skip) the first norm of an up resnet
                    x = note_skip(self._run_transformer(r, att[j], x, out=skip_dest(r.m, p["cout"]), gn_next=True))
                else:
                    x = note_skip(self._run_resblock(r, p, x, out=skip_dest(r.m, p["cout"]), gn_next=True))
            if ds is not None:
                geom, ho, wo = self._conv_geom(r, stride=2)
                m_out = r.b * r.f * ho * wo
                r.h, r.w = ho, wo                    # (the output's level decides whether its frames are whole tiles)
                x = note_skip(self._gemm(r, ds, x, m=m_out, conv=geom, out=skip_dest(m_out, ds.n_true), gn_next=True))
        if len(cats) != len(cx_pop):
            raise RuntimeError("skip bookkeeping out of step with the up blocks")

        def x_dest():                          # left-hand columns of the buffer the next up resnet reads (None: no more)
            return cats[-1][:, :cx_pop[len(cx_pop) - len(cats)]] if cats else None

        x = self._run_resblock(r, self.mid[0], x)
        x = self._run_transformer(r, self.mid[1], x)
        note_x(self._run_resblock(r, self.mid[2], x, out=x_dest(), gn_next=True))
        for res, att, us in self.up:
            for j, p in enumerate(res):
                cat = cats.pop()               # both halves are in place
                sums = cat_sums.pop()          # ... and, where both producers left them, the column sums of both
                last = j == len(res) - 1 and us is not None
                if att:
                    x = self._run_resblock(r, p, cat, gn_next=True, concat_sums=sums)   # -> the transformer's entry norm
                    # -> the left half of the next up resnet's input, or conv_norm_out at the very end
                    x = note_x(self._run_transformer(r, att[j], x, out=None if last else x_dest(), gn_next=not last))
                else:
                    x = note_x(self._run_resblock(r, p, cat, out=None if last else x_dest(), gn_next=not last,
                                                  concat_sums=sums))
                del cat
            if us is not None:
                geom, ho, wo = self._conv_geom(r, ups=1)
                m_up = r.b * r.f * ho * wo
                r.h, r.w = ho, wo                    # (the output's level decides whether its frames are whole tiles)
                x = note_x(self._gemm(r, us, x, m=m_up, conv=geom, out=x_dest(), gn_next=True))
        x = self._gn(r, self.norm_out, x, temporal=False, silu=True)
        geom, _, _ = self._conv_geom(r)
        if euler is not None:
            self._gemm(r, self.conv_out, x, conv=geom, euler=dict(euler, frames=r.f, hw=r.hw))
            return None
        return self._gemm(r, self.conv_out, x, conv=geom)

    # diffusers-style call (sample (B,F,8,H,W)) – used by parity tests and as a drop-in `unet`
    def __call__(self, sample, timestep, encoder_hidden_states, added_time_ids, return_dict=False):
        b, f, c, h, w = sample.shape
        dev = self.device
        rows = torch.zeros((b * f * h * w, self.cin_pad), dtype=torch.float16, device=dev)
        rows[:, :c] = sample.to(dev, torch.float16).permute(0, 1, 3, 4, 2).reshape(-1, c)
        # ONE timestep and ONE (fps-1, motion bucket, noise aug) triple per call: the embedding MLPs run once and their
        # result is shared by every video of the batch (what the reference adapter feeds, svd_unet.py:252-259,389-392:
        # a scalar timestep and `added_time_ids.repeat(batch, 1)`).  Per-video values would be silently ignored -> refuse.
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1)
        if t.numel() not in (1, b) or (t.numel() > 1 and bool((t != t[0]).any())):
            raise ValueError("SVDUNetHIP takes one timestep per call (a scalar, or B equal values)")
        ids = added_time_ids.to(torch.float32).reshape(-1, 3) if added_time_ids.numel() % 3 == 0 else None
        if ids is None or ids.shape[0] not in (1, b):
            raise ValueError(f"added_time_ids must be (B, 3) or (1, 3); got {tuple(added_time_ids.shape)}")
        if ids.shape[0] > 1 and bool((ids != ids[0]).any()):
            raise ValueError("SVDUNetHIP shares the added-time embedding across the batch: the rows of added_time_ids "
                             "differ (run videos with different fps / motion bucket / noise augmentation in separate calls)")
        if encoder_hidden_states.shape[0] != b or encoder_hidden_states[0].numel() != self.cfg.cross_attention_dim:
            raise ValueError(f"encoder_hidden_states must be (B, 1, {self.cfg.cross_attention_dim}) with B = {b}; got "
                             f"{tuple(encoder_hidden_states.shape)}")
        eps = self.forward_rows(rows, b=b, frames=f, h=h, w=w, t_value=t[:1].to(dev),
                                ctx16=encoder_hidden_states.to(dev, torch.float16).reshape(b, -1).contiguous(),
                                added_ids32=ids[0].to(dev).contiguous())
        out = eps.reshape(b, f, h, w, -1).permute(0, 1, 4, 2, 3).contiguous()
        return (out,)
