"""Host-side mirror of the reference's PointAggregator (viewmlp, order-2 hybrid path) over libhnr_hip.so.

Same parameter names and shapes as /root/reference/models/aggregators/point_aggregators.py:484-754
(`block1.{0,2}`, `block3.{0,2}`, `alpha_branch.0`, `color_branch.{0,2,4,6}` (constructed, never used --
:542-553 vs :1001-1037), `color_feature_branch.{0,2,4}`, `aux_merge_weight_block.{0,2,4,6}`,
`aux_block_s{1,2,3}.{0,2}`, `color_mixup_block.{0,2,4}`, `color_final_block.0`), so a reference
`*_net_ray_marching.pth` loads with `load_state_dict` unchanged.

Only the configuration family the 19 shipped launch scripts use is implemented; anything else raises
 instead of silently computing something different.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import HnrError
from .linear import PackedLinear, FusedMlp3

_REQUIRED = dict(which_agg_model="viewmlp", agg_distance_kernel="linear", agg_intrp_order=2, agg_dist_pers=20,
                 apply_pnt_mask=1, num_feat_freqs=3, dist_xyz_freq=5, num_viewdir_freqs=4, view_ori=0,
                 point_features_dim=32, shading_feature_num=256, shading_feature_mlp_layer1=2,
                 shading_feature_mlp_layer2=0, shading_feature_mlp_layer3=2, shading_alpha_mlp_layer=1,
                 shading_color_mlp_layer=4, act_type="LeakyReLU", act_super=1, agg_feat_xyz_mode="None",
                 agg_alpha_xyz_mode="None", agg_color_xyz_mode="None", feature_guidance=1, mixup_mode="partial",
                 learn_residuals=1, use_delta_view=1, tradition_attention=0, refine_blend=0, dynamic_weight=0, add_idx=0,
                 separate_color_decoder=0, large_color_final_block=0, use_2D_CNN=0,
                 disable_viewdirs=0, disable_color_feature=0, point_conf_mode="1", point_dir_mode="1", point_color_mode="1")


def check_opt(opt):
    bad = []
    for k, v in _REQUIRED.items():
        got = getattr(opt, k, v)
        if got != v:
            bad.append("%s=%r (supported: %r)" % (k, got, v))
    if float(getattr(opt, "dist_xyz_deno", 0.0)) != 0.0:
        bad.append("dist_xyz_deno=%r (supported: 0)" % opt.dist_xyz_deno)
    if getattr(opt, "agg_weight_norm", 1) <= 0:
        bad.append("agg_weight_norm<=0")
    aw = getattr(opt, "agg_axis_weight", None)
    if aw is not None and not (float(aw[0]) == 1.0 and float(aw[2]) == 1.0):
        bad.append("agg_axis_weight=%r (supported: None or 1 1 1)" % (aw,))
    if bad:
        raise HnrError("PointAggregator: unsupported option(s) for the HIP path: " + "; ".join(bad))


def _xavier_uniform_(m, gain):
    # models/helpers/networks.py:83-122
    if isinstance(m, nn.Conv2d):
        ks = m.kernel_size[0] * m.kernel_size[1]
        std = gain * np.sqrt(2.0 / ((m.in_channels + m.out_channels) * ks))
    else:
        std = gain * np.sqrt(2.0 / (m.in_features + m.out_features))
    m.weight.data.uniform_(-std * np.sqrt(3.0), std * np.sqrt(3.0))


def _init_seq(s):
    # models/helpers/networks.py:163-172
    mods = list(s)
    for a, b in zip(mods[:-1], mods[1:]):
        if isinstance(a, (nn.Linear, nn.Conv2d)):
            gain = nn.init.calculate_gain("leaky_relu", b.negative_slope) if isinstance(b, nn.LeakyReLU) else 1
            _xavier_uniform_(a, gain)
    if isinstance(mods[-1], (nn.Linear, nn.Conv2d)):
        _xavier_uniform_(mods[-1], 1)


def _mlp(dims, act_last=True, final=None):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if act_last or i < len(dims) - 2:
            layers.append(nn.LeakyReLU(inplace=True))
    if final is not None:
        layers.append(final)
    return nn.Sequential(*layers)


def _cnn(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride=2, padding=1), nn.LeakyReLU(inplace=True),
                         nn.Conv2d(cout, cout, 3, stride=1, padding=1), nn.LeakyReLU(inplace=True))


class PointAggregator(nn.Module):
    def __init__(self, opt):
        super().__init__()
        check_opt(opt)
        self.opt = opt
        self.block1 = _mlp([284, 256, 256])
        self.block3 = _mlp([263, 256, 256])
        self.alpha_branch = nn.Sequential(nn.Linear(256, 1))
        self.color_branch = _mlp([280, 128, 128, 128, 3], act_last=False)          # checkpointed, unused (:542-553)
        self.color_feature_branch = _mlp([280, 128, 128, 128])
        self.aux_merge_weight_block = _mlp([176, 64, 64, 64, 1], act_last=False, final=nn.Sigmoid())
        self.aux_block_s1 = _cnn(3, 6)
        self.aux_block_s2 = _cnn(6, 12)
        self.aux_block_s3 = _cnn(12, 24)
        self.color_mixup_block = _mlp([90, 45, 45, 45], act_last=False)
        self.color_final_block = nn.Sequential(nn.Linear(128, 3))
        # blur-kernel predictor of the *_learnable.sh configs (point_aggregators.py:715-750): the aggregator only OWNS these
        # parameters (checkpoint names `learn_blur_kernel_block.*`, `learn_blur_kernel_conv_block.*`) and hands the modules to
        # the training shell as `blur_predictor` (:1339-1344); the shell's learnable_blur_update_output applies them
        # (models/base_rendering_model.py:827-1020, out of scope) -- the hot path itself does not change.
        blur_blocks = []
        self.learn_blur_kernel_block = None
        if getattr(opt, "learnable_blur_kernel", 0):
            ps, ks = int(getattr(opt, "learnable_blur_patch_size", 8)), int(getattr(opt, "learnable_blur_kernel_size", 9))
            n_in, n_out = 2 * ps * ps, ks * ks + (1 if getattr(opt, "learnable_blur_kernel_mode", 4) in (2, 4) else 0)
            if getattr(opt, "learnable_blur_kernel_conv", 0):
                act = lambda: nn.LeakyReLU(inplace=True)
                self.learn_blur_kernel_conv_block = nn.Sequential(nn.Conv2d(2, 4, 3), act(), nn.Conv2d(4, 4, 1), act(), nn.Conv2d(4, 8, 3), act(),
                                                                  nn.Conv2d(8, 8, 1), act())
                blur_blocks.append(self.learn_blur_kernel_conv_block)
                n_in = 8 * (ps - 4) * (ps - 4)
            self.learn_blur_kernel_block = _mlp([n_in, 128, 128, 128, n_out], act_last=False, final=nn.Sigmoid())
            blur_blocks.append(self.learn_blur_kernel_block)
        for m in blur_blocks:
            _init_seq(m)
        for m in (self.block1, self.block3, self.alpha_branch, self.color_branch, self.color_feature_branch,
                  self.aux_merge_weight_block, self.aux_block_s1, self.aux_block_s2, self.aux_block_s3,
                  self.color_mixup_block, self.color_final_block):
            _init_seq(m)
        self._packed = None
        self._packed_key = None

    def blur_predictor(self):
        """What viewmlp returns next to the decoded features (:1339-1344): None, the MLP, or [conv block, MLP]."""
        if self.learn_blur_kernel_block is None:
            return None
        if getattr(self.opt, "learnable_blur_kernel_conv", 0):
            return [self.learn_blur_kernel_conv_block, self.learn_blur_kernel_block]
        return self.learn_blur_kernel_block

    # ------------------------------------------------------------------------------------------
    def packed(self):
        """Weights in the kernels' layouts, re-packed only when a parameter changed."""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is not None and key == self._packed_key:
            return self._packed
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise HnrError("PointAggregator weights must be on the GPU (there is no CPU path)")
        lin = lambda seq, i: PackedLinear(seq[i].weight, seq[i].bias)
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        pk = dict(
            b1=[lin(self.block1, 0), lin(self.block1, 2)], b3=[lin(self.block3, 0), lin(self.block3, 2)],
            # block1.0 split by input columns: [emb32 | PE(emb) 192] depend on the point only, [PE(dists) 60] on the pair
            b1_point=PackedLinear(self.block1[0].weight[:, :224].contiguous(), None),
            b1_dist=PackedLinear(self.block1[0].weight[:, 224:].contiguous(), self.block1[0].bias),
            cf=[lin(self.color_feature_branch, i) for i in (0, 2, 4)],
            # aux_merge_weight_block.0 split by input columns [imgfeat45 | colfeat128 | ddir3]: the colour-feature part is the
            # same for the V views of a sample
            mw0_cf=PackedLinear(self.aux_merge_weight_block[0].weight[:, 45:173].contiguous(), self.aux_merge_weight_block[0].bias),
            mw0_fd=PackedLinear(torch.cat([self.aux_merge_weight_block[0].weight[:, :45],
                                           self.aux_merge_weight_block[0].weight[:, 173:176]], dim=1).contiguous(), None),
            mw=[lin(self.aux_merge_weight_block, i) for i in (0, 2, 4)],
            mx=[lin(self.color_mixup_block, i) for i in (0, 2, 4)],
            alpha_w=f32(self.alpha_branch[0].weight).reshape(256), alpha_b=f32(self.alpha_branch[0].bias).reshape(1),
            mw_last_w=f32(self.aux_merge_weight_block[6].weight).reshape(64), mw_last_b=f32(self.aux_merge_weight_block[6].bias).reshape(1),
            fin_w=f32(self.color_final_block[0].weight).reshape(3 * 128), fin_b=f32(self.color_final_block[0].bias).reshape(3),
            conv_w=[f32(m[i].weight) for m in (self.aux_block_s1, self.aux_block_s2, self.aux_block_s3) for i in (0, 2)],
            conv_b=[f32(m[i].bias) for m in (self.aux_block_s1, self.aux_block_s2, self.aux_block_s3) for i in (0, 2)],
            slope=float(self.block1[1].negative_slope),
        )
        self._packed, self._packed_key = pk, key
        self._packed_chain = None
        self._packed_mlp3 = None
        return pk

    def packed_chain(self):
        """block1 / block3 / alpha_branch in the image of the fused per-neighbour chain (hnr_chain_pack; csrc/chain.hip), packed
        with the fp32 images and re-packed when a parameter changes."""
        self.packed()
        if getattr(self, "_packed_chain", None) is None:
            L = _lib.lib()
            dev = next(self.parameters()).device
            buf = torch.empty((int(L.hnr_chain_packed_bytes()),), dtype=torch.uint8, device=dev)
            f32 = lambda t: t.detach().to(torch.float32).contiguous()
            w0 = f32(self.block1[0].weight[:, 224:284])
            args = [f32(self.block1[0].bias), f32(self.block1[2].weight), f32(self.block1[2].bias), f32(self.block3[0].weight),
                    f32(self.block3[0].bias), f32(self.block3[2].weight), f32(self.block3[2].bias),
                    f32(self.alpha_branch[0].weight).reshape(256), f32(self.alpha_branch[0].bias).reshape(1)]
            with torch.cuda.device(dev):
                _lib.check(L.hnr_chain_pack(_lib.ptr(w0), 60, *[_lib.ptr(t) for t in args], _lib.ptr(buf), _lib.stream()), "hnr_chain_pack")
            self._packed_chain = buf
        return self._packed_chain

    def packed_mlp3(self):
        """The per-sample MLPs as fused three-layer launches (hnr_mlp3_forward; csrc/mlp.hip): colour feature, merge weights
        (first layer split: the image-feature / view-direction columns here, the colour-feature columns per sample in `mw0_cf`), mix-up."""
        self.packed()
        if getattr(self, "_packed_mlp3", None) is None:
            cf, mw, mx = self.color_feature_branch, self.aux_merge_weight_block, self.color_mixup_block
            w_fd = torch.cat([mw[0].weight[:, :45], mw[0].weight[:, 173:176]], dim=1).contiguous()
            self._packed_mlp3 = dict(
                # colour feature + on its tail the colour-feature columns of aux_merge_weight_block.0 (with that layer's bias), once per sample
                cf=FusedMlp3([cf[0].weight, cf[2].weight, cf[4].weight, mw[0].weight[:, 45:173].contiguous()],
                             [cf[0].bias, cf[2].bias, cf[4].bias, mw[0].bias], [1, 1, 1, 0]),
                mw=FusedMlp3([w_fd, mw[2].weight, mw[4].weight], [None, mw[2].bias, mw[4].bias], [1, 1, 1]),
                mx=FusedMlp3([mx[0].weight, mx[2].weight, mx[4].weight], [mx[0].bias, mx[2].bias, mx[4].bias], [1, 1, 0]))
        return self._packed_mlp3

    def point_table(self, emb, ids=None, n_ids=None, want_rows=False, out=None):
        """[N,256] = [emb | PE3(emb)] @ block1.0.weight[:, :224]^T -- the point-only part of block1's first layer
        (exact split of the dot product; the bias and the 60 distance columns are added per (sample, neighbour) row).
        ids: int32 list of point ids -> only those rows (training: the points a batch touches)."""
        L = _lib.lib()
        pk = self.packed()
        emb = _lib.require_gpu(emb, "points_embeding", torch.float32)
        n, F = emb.shape
        if ids is not None:
            n = int(n_ids)
        E = torch.empty((n, 224), dtype=torch.float32, device=emb.device)
        with torch.cuda.device(emb.device):
            _lib.check(L.hnr_point_rows(_lib.ptr(emb), _lib.ptr(ids) if ids is not None else None, n, F, _lib.ptr(E), 224,
                                        _lib.stream()), "hnr_point_rows")
        T = pk["b1_point"](E, act=False) if out is None else pk["b1_point"](E, out=out, act=False)      # out: [n, >= 256] rows of a caller-owned table
        return (T, E) if want_rows else T

    def image_features(self, images_nearest):
        """[1,V,H,W,3] -> channels-last feature map [V,H,W,48] (hnr_image_features); once per frame."""
        L = _lib.lib()
        pk = self.packed()
        img = _lib.require_gpu(images_nearest, "images_nearest", torch.float32)
        if img.dim() == 5:
            img = img[0]
        V, H, W, _ = img.shape
        dev = img.device
        fm = torch.empty((V, H, W, 48), dtype=torch.float32, device=dev)
        scratch = torch.empty((max(int(L.hnr_image_features_scratch_elems(V, H, W)), 1),), dtype=torch.float32, device=dev)
        wp = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in pk["conv_w"]])
        bp = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in pk["conv_b"]])
        with torch.cuda.device(dev):
            _lib.check(L.hnr_image_features(_lib.ptr(img), V, H, W, wp, bp, pk["slope"], _lib.ptr(scratch), _lib.ptr(fm),
                                            _lib.stream()), "hnr_image_features")
        return fm
