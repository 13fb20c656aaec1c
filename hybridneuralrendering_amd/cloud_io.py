"""Point-cloud files either side of the path.

 * `.txt` dumps: /root/reference/utils/visualizer.py:29-39 (`save_points`: `np.savetxt(..., delimiter=";")`, one file per cloud,
   `step-{:04d}-{i}.txt`) and :100-120 (`Visualizer.save_neural_points`: xyz alone, xyz + one colour triple * 255, or three clouds for a
   9-channel feature);
 * the pickled surface cloud the object scenes start from: /root/reference/data/load_blender.py:116-132 (`load_blender_cloud`);
 * the feature initialisation of a cloud that does not come from a checkpoint: /root/reference/models/neural_points/neural_points.py:284-308.
Host-side file plumbing (numpy / torch); nothing here is on the hot path."""
import os
import pickle
import random

import numpy as np
import torch

from ._lib import HnrError


def save_points(xyz, dir, total_steps):
    """visualizer.py:29-39."""
    xyz = np.asarray(xyz)
    if xyz.ndim < 3:
        xyz = xyz[None, ...]
    os.makedirs(dir, exist_ok=True)
    paths = []
    for i in range(xyz.shape[0]):
        filename = ("step-{}-{}.txt" if isinstance(total_steps, str) else "step-{:04d}-{}.txt").format(total_steps, i)
        path = os.path.join(dir, filename)
        np.savetxt(path, xyz[i, ...].reshape(-1, xyz.shape[-1]), delimiter=";")
        paths.append(path)
    return paths


def save_neural_points(point_dir, total_steps, xyz, features=None):
    """Visualizer.save_neural_points (:100-116) without the reference-view images: xyz [N,3] tensor / array, features None, [1,N,9]
    (three clouds: xyz + features[..., 3i:3i+3] * 255) or [1,N,>=3] (xyz + features[..., :3] * 255)."""
    to_np = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    if features is None:
        return save_points(to_np(xyz), point_dir, total_steps)
    xyz_t, f = torch.as_tensor(to_np(xyz)), torch.as_tensor(to_np(features))
    if f.shape[-1] == 9:
        return save_points(np.stack([torch.cat([xyz_t, f[0, ..., 3 * i:3 * i + 3] * 255], dim=-1).numpy() for i in range(3)], axis=0), point_dir, total_steps)
    return save_points(torch.cat([xyz_t, f[0, ..., :3] * 255], dim=-1).numpy(), point_dir, total_steps)


def load_points_txt(path):
    """A cloud written by save_points: [N, C] float32 (C = 3 for positions, 6 with a colour triple)."""
    a = np.loadtxt(path, delimiter=";", dtype=np.float64, ndmin=2)
    return a.astype(np.float32)


def load_blender_cloud(point_path, point_num):
    """load_blender.py:116-132: a pickle {"point_xyz": [N,3] (, "point_face_normal": [N,3])}; more points than point_num are drawn down
    with `random.choices` (WITH replacement, the reference's rule)."""
    with open(point_path, "rb") as f:
        info = pickle.load(f)
    xyz = info["point_xyz"]
    nrm = info.get("point_face_normal")
    if point_num < len(xyz):
        inds = np.asarray(random.choices(range(len(xyz)), k=point_num))
        return xyz[inds, :], (nrm[inds, :] if nrm is not None else None)
    return xyz, nrm


def load_cloud(opt):
    """The initial cloud of NeuralPoints.__init__ when the checkpoint holds none (neural_points.py:248-250): `opt.cloud_path` is the
    reference's pickle, or a `.txt` dump of save_points, or a `.npy` array; returns [N,3] float32."""
    path = getattr(opt, "cloud_path", "")
    if not path or not os.path.exists(path):
        raise HnrError("NeuralPoints: no checkpoint cloud and opt.cloud_path=%r does not exist" % (path,))
    if path.endswith(".txt"):
        xyz = load_points_txt(path)[:, :3]
    elif path.endswith(".npy"):
        xyz = np.load(path)[:, :3]
    else:
        xyz, _ = load_blender_cloud(path, int(getattr(opt, "num_point", 8192)))
    return np.ascontiguousarray(xyz, dtype=np.float32)


def positional_encoding(x, num_freqs):
    """models/helpers/networks.py:175-189 with ori=False: per input dim d and frequency f the pair [sin(x_d 2^f), cos(x_d 2^f)]."""
    freqs = (2.0 ** torch.arange(num_freqs, dtype=torch.float32, device=x.device))
    ang = x[..., None] * freqs                                        # [..., D, F]
    return torch.stack([torch.sin(ang), torch.cos(ang)], dim=-1).reshape(*x.shape[:-1], -1)


def init_point_features(point_xyz, num_channels, method, device, feature_dim):
    """neural_points.py:284-308: points_embeding [1,N,C] for a fresh cloud; points_conf = ones [1,N,1]."""
    shape = (1, point_xyz.shape[0], num_channels)
    if method == "rand":
        emb = torch.rand(shape, device=device, dtype=torch.float32) - 0.5
    elif method == "zeros":
        emb = torch.zeros(shape, device=device, dtype=torch.float32)
    elif method == "ones":
        emb = torch.ones(shape, device=device, dtype=torch.float32)
    elif method == "pos":
        if feature_dim > 3:
            emb = positional_encoding(point_xyz.reshape(1, -1, 3), int(feature_dim / 6))
            if int(feature_dim / 6) * 6 < feature_dim:
                emb = torch.cat([emb, torch.rand(shape[:-1] + (feature_dim - emb.shape[-1],), device=device, dtype=torch.float32) - 0.5], dim=-1)
        else:
            emb = point_xyz.reshape(1, -1, 3)
    elif method.startswith("gau"):
        emb = torch.normal(mean=torch.zeros(shape, device=device, dtype=torch.float32), std=float(method.split("_")[1]))
    else:
        raise ValueError(method)
    return emb, torch.ones_like(emb[..., 0:1])
