"""Loads the committed golden fixtures (tests/golden/*.npz|json) -- data only."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_render(tag):
    z = np.load(os.path.join(GOLD, "render_%s.npz" % tag), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["opt"] = json.loads(str(d.pop("opt_json")))
    d["images_nearest"] = d["images_nearest"].astype(np.float32) / np.float32(255)
    d["sd"] = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}
    return d


def load_train(tag):
    """train_<tag>.npz shares its scene and weights with render_<tag>.npz (asserted by make_golden.py); a fixture with
    `shares_inputs_with` (e.g. the use_nearest=0 variant) also takes its ray batch from that training fixture."""
    z = np.load(os.path.join(GOLD, "train_%s.npz" % tag), allow_pickle=False)
    if "shares_inputs_with" in z.files:
        d = load_train(str(z["shares_inputs_with"])[len("train_"):])
        for k in z.files:
            if not k.startswith("grad.") and k not in ("opt_json", "shares_inputs_with", "grad_names", "scene"):
                d[k] = z[k]
        d["opt"] = json.loads(str(z["opt_json"]))
        d["grad"] = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad.")}
        d["grad_names"] = [str(n) for n in z["grad_names"]]
        return d
    # `scene_from` (train_c5_small.npz): scene and weights of another render fixture, its own ray batch and extra inputs
    d = load_render(str(z["scene_from"])[len("render_"):] if "scene_from" in z.files else tag)
    for k in ("blur_kernels", "frame_weight", "patch", "blurred_raycolor"):
        if k in z.files:
            d[k] = z[k]
    for k in ("coarse_raycolor", "conf_coefficient", "full_coarse_raycolor", "q_sample_pidx", "q_sample_loc_w", "q_ray_mask",
              "pix", "raydir", "c2w", "intrinsic", "bg_color", "near_far", "tmid", "gt", "loss", "zero_epsilon", "loss_compute_losses"):
        d[k] = z[k]
    d["opt"] = json.loads(str(z["opt_json"]))
    d["grad"] = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad.")}
    d["grad_names"] = sorted(k for k in d["grad"] if k.startswith("aggregator."))
    for k in ("coarse_point_opacity", "coarse_is_background", "queried_shading", "ray_mask", "weight", "blend_weight",
              "decoded_features", "ray_valid", "full_coarse_point_opacity", "full_coarse_is_background", "full_coarse_mask"):
        d.pop(k, None)
    return d


def torch_inputs(d, device="cpu"):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    c2w = d["c2w"]
    return dict(
        xyz=t(d["xyz"]), emb=t(d["emb"]), conf=t(d["conf"]), pdir=t(d["pdir"]), color=t(d["color"]),
        campos=t(c2w[:3, 3])[None], camrotc2w=t(c2w[:3, :3])[None], raydir=t(d["raydir"])[None],
        bg_color=t(d["bg_color"])[None], c2w_nearest=t(d["c2w_nearest"])[None],
        campos_nearest=t(d["c2w_nearest"][:, :3, 3])[None], intrinsic_nearest=t(d["intrinsic"])[None],
        images_nearest=t(d["images_nearest"])[None])


def blur_learn_case(tag, device="cpu"):
    """One case of tests/golden/blur_learn.npz: (cfg dict, color, gt, upstream, predictor [module or [conv, mlp]], expected dict)."""
    import numpy as np
    import torch
    import torch.nn as nn
    z = np.load(os.path.join(GOLD, "blur_learn.npz"))
    pn, ps, ks, conv, norm, mode, bmode = (int(v) for v in z[tag + "_cfg"])
    act = lambda: nn.LeakyReLU(inplace=True)
    n_in, n_out = 2 * ps * ps, ks * ks + (1 if mode in (2, 4) else 0)
    blocks = []
    if conv:
        blocks.append(nn.Sequential(nn.Conv2d(2, 4, 3), act(), nn.Conv2d(4, 4, 1), act(), nn.Conv2d(4, 8, 3), act(), nn.Conv2d(8, 8, 1), act()))
        n_in = 8 * (ps - 4) * (ps - 4)
    blocks.append(nn.Sequential(nn.Linear(n_in, 128), act(), nn.Linear(128, 128), act(), nn.Linear(128, 128), act(), nn.Linear(128, n_out), nn.Sigmoid()))
    for bi, blk in enumerate(blocks):
        blk.load_state_dict({k: torch.from_numpy(z["%s_w%d.%s" % (tag, bi, k)]) for k in blk.state_dict()})
        blk.to(device)
    t = lambda k: torch.from_numpy(z[tag + "_" + k]).to(device)
    cfg = dict(pn=pn, ps=ps, ks=ks, conv=conv, norm=norm, mode=mode, bmode=bmode)
    grads = {"%d.%s" % (bi, k): z["%s_g%d.%s" % (tag, bi, k)] for bi, blk in enumerate(blocks) for k, _ in blk.named_parameters()}
    return cfg, t("color"), t("gt"), t("upstream"), (blocks if conv else blocks[0]), blocks, dict(out=z[tag + "_out"], grad_color=z[tag + "_grad_color"], grads=grads)


def c1_chair():
    """BASELINE config C1: chair 200x200 camera, 100 k points (seed 0), ONE 32x32 = 1024-ray batch at the image centre,
    weights of render_synth_small.npz; expected outputs from the imported reference (tests/golden/make_golden.py::gen_c1).
    Returns (scene, pix, raydir, state_dict, expected dict)."""
    from hybridneuralrendering_amd import scenes
    sc = scenes.make_scene("chair", 100000, 0)
    sc.opt.agg_axis_weight = None
    px, py = np.meshgrid(np.arange(84, 116), np.arange(84, 116), indexing="ij")
    pix = np.stack([px, py], axis=-1).reshape(-1, 2).astype(np.int32)
    raydir = scenes.camera_rays(pix, sc.intrinsic, sc.c2w)
    z = np.load(os.path.join(GOLD, "render_c1_chair.npz"), allow_pickle=False)
    sd = load_render(str(z["weights_from"])[len("render_"):-len(".npz")])["sd"]
    exp = {k: z[k] for k in ("ray_mask", "full_coarse_raycolor", "full_coarse_point_opacity", "full_coarse_is_background")}
    exp["counts"] = dict(zip([str(k) for k in z["count_keys"]], [int(v) for v in z["counts"]]))
    return sc, pix, raydir, sd, exp
