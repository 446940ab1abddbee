"""LightGlue matcher, native on MI355X (csrc/lightglue.hip).

Drop-in for LightGlue (reference core/modules/matchers/lightglue.py:421-716), inference path:
same `conf` handling (merged over `default_conf`), same parameter tree (`posenc.Wr`,
`transformers.{i}.self_attn|cross_attn.*`, `log_assignment.{i}.*`, `token_confidence.{i}.*`,
optional `input_proj`) and the same output dict.  Training-only members (loss, NLLLoss,
matcher_metrics, token-confidence loss; :17-133, :190-203, :751-800) are out of scope.
Early stopping / point pruning are commented out in the reference (:606-652) and absent here.
"""
import ctypes

import torch
from torch import nn

from ...._native import on_input_device
from .... import _lib
from .... import _native as N
from ._batched import from_feats, materialize_matches, stacked_outputs


class _Conf(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _merge(base, over):
    out = _Conf(base)
    for k in (over.keys() if hasattr(over, "keys") else []):
        out[k] = over[k]
    return out


class LearnableFourierPositionalEncoding(nn.Module):
    def __init__(self, M, dim, F_dim=None, gamma=1.0):
        super().__init__()
        F_dim = F_dim if F_dim is not None else dim
        self.gamma = gamma
        self.Wr = nn.Linear(M, F_dim // 2, bias=False)
        nn.init.normal_(self.Wr.weight.data, mean=0, std=self.gamma ** -2)


def _ffn(d):
    return nn.Sequential(nn.Linear(2 * d, 2 * d), nn.LayerNorm(2 * d, elementwise_affine=True), nn.GELU(), nn.Linear(2 * d, d))


class SelfBlock(nn.Module):
    def __init__(self, embed_dim, num_heads, flash=False, bias=True):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.Wqkv = nn.Linear(embed_dim, 3 * embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.ffn = _ffn(embed_dim)


class CrossBlock(nn.Module):
    def __init__(self, embed_dim, num_heads, flash=False, bias=True):
        super().__init__()
        self.heads = num_heads
        self.to_qk = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.to_v = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.to_out = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.ffn = _ffn(embed_dim)


class TransformerLayer(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.self_attn = SelfBlock(*args, **kwargs)
        self.cross_attn = CrossBlock(*args, **kwargs)


class MatchAssignment(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.matchability = nn.Linear(dim, 1, bias=True)
        self.final_proj = nn.Linear(dim, dim, bias=True)


class TokenConfidence(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.token = nn.Sequential(nn.Linear(dim, 1), nn.Sigmoid())


class LightGlue(nn.Module):
    default_conf = {
        "name": "lightglue", "input_dim": 256, "add_scale_ori": False, "descriptor_dim": 256, "n_layers": 9, "num_heads": 4,
        "flash": False, "mp": False, "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0,
        "checkpointed": False, "weights": "superpoint", "weights_from_version": "v0.1_arxiv",
        "loss": {"gamma": 1.0, "fn": "nll", "nll_balancing": 0.5},
    }

    def __init__(self, conf):
        super().__init__()
        self.conf = conf = _merge(self.default_conf, conf)
        # widths as the reference derives them (lightglue.py:246-248, 456-461): any num_heads dividing descriptor_dim.  The
        # attention kernel is instantiated for 32-, 64- and 128-wide heads (256 = 4 x 64, every EI-Nexus YAML, runs its own
        # instantiation); other widths run the next larger one on zero-padded heads.  What is left out: head widths that are
        # not a multiple of 4 (a head's rows are read 16 bytes at a time) or wider than 256.
        assert conf.descriptor_dim % conf.num_heads == 0
        hd = conf.descriptor_dim // conf.num_heads
        if hd % 4 or hd > 256:
            raise NotImplementedError("einx LightGlue: descriptor_dim // num_heads must be a multiple of 4, at most 256 "
                                      f"(got {conf.descriptor_dim} // {conf.num_heads})")
        if conf.input_dim != conf.descriptor_dim:
            self.input_proj = nn.Linear(conf.input_dim, conf.descriptor_dim, bias=True)
        else:
            self.input_proj = nn.Identity()
        head_dim = conf.descriptor_dim // conf.num_heads
        # add_scale_ori: the reference builds a 4-input encoding (:457-459) but its forward never appends scales / orientations
        # (:540-560 are commented out), so every forward fails in posenc.Wr on the 2-column keypoints; mirrored in forward()
        self.posenc = LearnableFourierPositionalEncoding(2 + 2 * bool(conf.add_scale_ori), head_dim, head_dim)
        h, n, d = conf.num_heads, conf.n_layers, conf.descriptor_dim
        self.transformers = nn.ModuleList([TransformerLayer(d, h, conf.flash) for _ in range(n)])
        self.log_assignment = nn.ModuleList([MatchAssignment(d) for _ in range(n)])
        self.token_confidence = nn.ModuleList([TokenConfidence(d) for _ in range(n - 1)])
        self.want_log_assignment = True
        self.fold_message_projection = True  # inference-time weight folding (see _pack); False = layer by layer as written
        self.merge_qk_v = True  # CrossBlock.to_qk and to_v as one launch over a merged weight image (same results); False = two launches
        self._packed = None
        self._sig = None
        self._sig_tensors = None
        # also reached when a parent module's load_state_dict recurses into this one
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.refresh())

    def _apply(self, fn, *a, **k):
        self._packed = self._sig_tensors = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = self._sig_tensors = None
        return super().load_state_dict(*a, **k)

    def refresh(self):
        """Drop the packed / folded weight images.  REQUIRED after edits through `p.data` (an alias with its own version
        counter: `p._version`, which `_pack` keys on, does not move) or after replacing Parameter objects."""
        self._packed = self._sig_tensors = None

    def _pack(self):
        """ctypes image of the parameter pointers (weights stay in their nn.Parameter storage)."""
        if self._sig_tensors is None:
            self._sig_tensors = list(self.parameters())
        sig = tuple((t.data_ptr(), t._version) for t in self._sig_tensors)
        if sig != self._sig:  # in-place parameter edits and moves drop the folded weight images too
            self._packed, self._sig = None, sig
        if self._packed is not None:
            return self._packed
        c = self.conf
        keep = []

        def p(t):
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != "cuda":
                raise RuntimeError("einx LightGlue: parameters must be contiguous fp32 tensors on a HIP device")
            keep.append(t)
            return t.data_ptr()

        d = c.descriptor_dim

        def fold(ffn0, proj):
            """ffn0(cat[x, proj(ctx)]) == cat[x, ctx] @ [W0a | W0b Wp]^T + (b0 + W0b bp): fold the message
            projection into the FFN's first Linear once, at load time (saves one GEMM per block)."""
            W0, b0 = ffn0.weight.detach(), ffn0.bias.detach()
            W0b = W0[:, d:].contiguous()
            Wf = W0.clone()
            zero = torch.zeros(d, dtype=torch.float32, device=W0.device)
            Wf[:, d:] = N.linear(W0b, proj.weight.detach().t().contiguous(), zero)  # W0b @ Wp
            bf = b0.clone().reshape(-1, 1).contiguous()
            N.linear(W0b, proj.bias.detach().reshape(1, -1).contiguous(), zero[:1], out=bf, accumulate=True)  # b0 + W0b @ bp
            return Wf.contiguous(), bf.reshape(-1).contiguous()

        layers = (_lib.LgLayer * c.n_layers)()
        for i, tl in enumerate(self.transformers):
            s, x, L = tl.self_attn, tl.cross_attn, layers[i]
            L.Wqkv, L.bqkv = p(s.Wqkv.weight), p(s.Wqkv.bias)
            L.Wqk, L.bqk, L.Wv, L.bv = p(x.to_qk.weight), p(x.to_qk.bias), p(x.to_v.weight), p(x.to_v.bias)
            if self.merge_qk_v and c.descriptor_dim == 256 and c.num_heads == 4:  # one launch for to_qk | to_v (copies, like the folds)
                L.Wqk_v = p(torch.cat([x.to_qk.weight.detach(), x.to_v.weight.detach()], 0).contiguous())
                L.bqk_v = p(torch.cat([x.to_qk.bias.detach(), x.to_v.bias.detach()], 0).contiguous())
            else:
                L.Wqk_v, L.bqk_v = None, None
            if self.fold_message_projection:
                sw, sb = fold(s.ffn[0], s.out_proj)
                xw, xb = fold(x.ffn[0], x.to_out)
                L.Wo, L.bo, L.Wco, L.bco = None, None, None, None
                L.sf0_w, L.sf0_b, L.cf0_w, L.cf0_b = p(sw), p(sb), p(xw), p(xb)
            else:
                L.Wo, L.bo, L.Wco, L.bco = p(s.out_proj.weight), p(s.out_proj.bias), p(x.to_out.weight), p(x.to_out.bias)
                L.sf0_w, L.sf0_b, L.cf0_w, L.cf0_b = p(s.ffn[0].weight), p(s.ffn[0].bias), p(x.ffn[0].weight), p(x.ffn[0].bias)
            L.sln_g, L.sln_b = p(s.ffn[1].weight), p(s.ffn[1].bias)
            L.sf3_w, L.sf3_b = p(s.ffn[3].weight), p(s.ffn[3].bias)
            L.cln_g, L.cln_b = p(x.ffn[1].weight), p(x.ffn[1].bias)
            L.cf3_w, L.cf3_b = p(x.ffn[3].weight), p(x.ffn[3].bias)
        w = _lib.LgWeights()
        w.struct_size, w.layer_size = ctypes.sizeof(_lib.LgWeights), ctypes.sizeof(_lib.LgLayer)
        if isinstance(self.input_proj, nn.Linear):
            w.in_w, w.in_b = p(self.input_proj.weight), p(self.input_proj.bias)
        else:
            w.in_w, w.in_b = None, None
        la = self.log_assignment[c.n_layers - 1]
        w.Wr = p(self.posenc.Wr.weight)
        w.proj_w, w.proj_b = p(la.final_proj.weight), p(la.final_proj.bias)
        w.match_w, w.match_b = p(la.matchability.weight), p(la.matchability.bias)
        w.n_layers, w.heads, w.d, w.input_dim = c.n_layers, c.num_heads, c.descriptor_dim, c.input_dim
        w.filter_threshold = float(c.filter_threshold)
        w.layers = ctypes.cast(layers, ctypes.POINTER(_lib.LgLayer))
        self._watch = N.ParamWatch(self._sig_tensors)  # `.data` edits are seen by content at the next forward (round 4)
        self._packed = (w, layers, keep)
        return self._packed

    def _add_scale_ori_error(self, feats0):
        """What the reference's forward raises with add_scale_ori=True (recorded from the reference, tests/golden/lgcfg.json): the
        descriptor-width assert comes first (:562-563), then posenc.Wr -- Linear(4, head_dim/2) -- meets [b, n, 2] keypoints (:565)."""
        pos, desc = feats0["sparse_positions"], feats0["sparse_descriptors"]
        pos = pos if torch.is_tensor(pos) else pos[0][None]
        desc = desc if torch.is_tensor(desc) else desc[0][None]
        if desc.shape[-1] != self.conf.input_dim:
            raise AssertionError("descriptor dimension does not match conf.input_dim")
        rows = pos.shape[0] * pos.shape[1] if pos.dim() == 3 else pos.shape[0]
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({rows}x2 and 4x{self.posenc.Wr.weight.shape[0]})")

    @on_input_device
    def match_batched(self, pb0, pb1, all_layers=False):
        if self.conf.add_scale_ori:
            raise RuntimeError("einx LightGlue: add_scale_ori=True has no working forward in the reference (lightglue.py:540-565)")
        if pb0.desc.shape[-1] != self.conf.input_dim or pb1.desc.shape[-1] != self.conf.input_dim:
            raise AssertionError("descriptor dimension does not match conf.input_dim")
        w = self._pack()[0]
        w.filter_threshold = float(self.conf.filter_threshold)  # read at every call like the reference's filter_matches call (lightglue.py:656)
        stale = self._watch.check()
        r = N.lightglue(w, pb0, pb1, want_la=self.want_log_assignment, want_ref=True, all_layers=all_layers)
        r.stale = stale
        return N.gather_matches(r, pb0.kpts, pb1.kpts, pb0.counts, 2)

    @torch.no_grad()
    @on_input_device
    def forward(self, feats0, feats1):
        """B == 1: the per-pair dict (matched keypoints in pixel coordinates, lightglue.py:689-698).
        B > 1 (stacked [B,n,*] inputs): whole-batch tensors and per-pair matched keypoints in
        *normalised* coordinates, as lightglue.py:675-687 returns them.  In training mode
        (`self.training`) ref_descriptors hold every layer's output [B,L,n,d] (:626-629,709-710);
        forward values only -- autograd / the loss are outside this build."""
        pos0, pos1 = feats0["sparse_positions"], feats1["sparse_positions"]
        if self.conf.add_scale_ori:
            self._add_scale_ori_error(feats0)
        stacked = torch.is_tensor(pos0) and pos0.dim() == 3 and torch.is_tensor(pos1) and pos1.dim() == 3
        if stacked:
            if feats0["sparse_descriptors"].shape[-1] != self.conf.input_dim or feats1["sparse_descriptors"].shape[-1] != self.conf.input_dim:
                raise AssertionError("descriptor dimension does not match conf.input_dim")
            if pos0.shape[0] > 1 and (pos0.numel() == 0 or pos1.numel() == 0):
                f = feats0["sparse_descriptors"]
                B, n, m = pos0.shape[0], pos0.shape[1], pos1.shape[1]
                print("No keypoints found in either image")
                return {"matches0": f.new_full((B, n), -1), "matches1": f.new_full((B, m), -1), "matching_scores0": f.new_zeros((B, n)),
                        "matching_scores1": f.new_zeros((B, m)), "matched_kpts0": [f.new_zeros((n, 3))] * B,
                        "matched_kpts1": [f.new_zeros((m, 3))] * B, "similarity": f.new_zeros((B, n, m)),
                        "log_assignment": f.new_zeros((B, n + 1, m + 1))}
        pb0, pb1 = from_feats(feats0), from_feats(feats1)
        all_layers = bool(self.training)
        r = self.match_batched(pb0, pb1, all_layers=all_layers)
        # match counts + the weight watch in one read-back
        nm_all = torch.cat([r.nmatch, r.stale]).cpu().tolist() if r.stale is not None else r.nmatch.cpu().tolist() + [0]
        if nm_all[-1]:  # a weight was edited through `.data`: rebuild the images, match again
            self.refresh()
            r = self.match_batched(pb0, pb1, all_layers=all_layers)
            nm_all = r.nmatch.cpu().tolist() + [0]
        n = pb0.counts_host or pb0.counts.cpu().tolist()
        m = pb1.counts_host or pb1.counts.cpu().tolist()
        L = self.conf.n_layers
        if pb0.B == 1:
            nm = nm_all[:-1]
            out = {k: v[0] for k, v in materialize_matches(r, n, m, nm, 2).items()}
            if n[0] and m[0]:
                ref0 = r.ref0 if all_layers else r.ref0[:, None]
                ref1 = r.ref1 if all_layers else r.ref1[:, None]
                out["ref_descriptors0"] = ref0[:, :, :n[0]]
                out["ref_descriptors1"] = ref1[:, :, :m[0]]
                out["prune0"] = torch.ones_like(out["matching_scores0"]) * L
                out["prune1"] = torch.ones_like(out["matching_scores1"]) * L
            return out
        if not (stacked and len(set(n)) == 1 and len(set(m)) == 1):
            raise NotImplementedError("einx LightGlue.forward with B > 1 takes stacked [B,n,*] tensors (as the reference does); "
                                      "use Matcher for ragged batches")
        # b > 1: matched keypoints are gathered from the normalised coordinates (reference behaviour)
        k0 = N.normalize_keypoints(pb0.kpts, pb0.image_size, out_cols=3)
        k1 = N.normalize_keypoints(pb1.kpts, pb1.image_size, out_cols=3)
        N.gather_matches(r, k0, k1, pb0.counts, 2)
        out = stacked_outputs(r, r.nmatch.cpu().tolist(), 2)
        out["ref_descriptors0"] = r.ref0 if all_layers else r.ref0[:, None]
        out["ref_descriptors1"] = r.ref1 if all_layers else r.ref1[:, None]
        out["prune0"] = torch.ones_like(r.scores0) * L
        out["prune1"] = torch.ones_like(r.scores1) * L
        return out
