"""hybridneuralrendering_amd: MI355X-native hot path of CVMI-Lab/HybridNeuralRendering.

voxel k-NN neural-point query -> point/image feature gather + aggregation MLP -> alpha composite,
as hand-written gfx950 HIP kernels behind a C ABI (include/hnr.h, libhnr_hip.so) with a host-side
mirror of the reference's module surface.  See DESIGN.md and INTEGRATION.md.
"""
from . import scenes  # noqa: F401
from ._lib import HnrError  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # heavier modules are imported lazily so that `import hybridneuralrendering_amd` works without a GPU
    if name in ("lighting_fast_querier", "VoxelGrid", "march_query", "compact_rays", "tmid_table"):
        from . import querier
        return getattr(querier, name)
    raise AttributeError(name)
