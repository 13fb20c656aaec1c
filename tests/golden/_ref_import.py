"""Import the reference's hot-path modules on CPU (THIS CONTAINER ONLY: /root/reference is absent on
the GPU box).  Third-party modules the image lacks and that the hot path never executes are
replaced by empty stand-in modules so that `import` succeeds; no reference source is copied."""
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    if REF not in sys.path:
        sys.path.insert(0, REF)

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Any()

        def __getattr__(self, n):
            return _Any()

    pycuda = _stub("pycuda")
    drv = _stub("pycuda.driver", PointerHolderBase=object, init=lambda: None, Device=_Any)
    comp = _stub("pycuda.compiler", SourceModule=_Any)
    import numpy as np
    ga = _stub("pycuda.gpuarray", to_gpu=lambda x: np.asarray(x))
    pycuda.driver, pycuda.compiler, pycuda.gpuarray = drv, comp, ga
    for name in ("cv2", "imageio", "open3d", "pytorch_msssim", "h5py", "plyfile", "imutils", "kornia", "lpips",
                 "skimage", "skimage.metrics", "tensorboardX", "inplace_abn", "torch_scatter"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name, SSIM=_Any, ms_ssim=_Any, ssim=_Any, PlyData=_Any, PlyElement=_Any, InPlaceABN=_Any)
    if "torchvision" not in sys.modules:
        try:
            import torchvision  # noqa: F401
        except Exception:
            tv = _stub("torchvision")
            tv.utils = _stub("torchvision.utils", make_grid=_Any, save_image=_Any)
            tv.transforms = _stub("torchvision.transforms")
    import models.neural_points.query_point_indices_worldcoords as qw
    import models.neural_points.neural_points as npts
    import models.aggregators.point_aggregators as agg
    import models.rendering.diff_ray_marching as drm
    import models.rendering.diff_render_func as drf
    import models.helpers.networks as nets
    import models.neural_points_volumetric_model as vol
    return types.SimpleNamespace(qw=qw, npts=npts, agg=agg, drm=drm, drf=drf, nets=nets, vol=vol)
