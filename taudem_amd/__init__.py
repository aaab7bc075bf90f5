"""taudem_amd - MI355X (gfx950) implementation of TauDEM's D8 / D-infinity flow-direction and
contributing-area hot path (PitRemove -> D8FlowDir -> AreaD8, DinfFlowDir -> AreaDinf -> DinfDecayAccum).

Layers: ``csrc/`` hand-written HIP kernels + the C ABI of ``include/taudem_amd.h`` (libtaudem_amd.so);
``api`` in-memory binding (numpy host buffers or torch CUDA tensors); ``tools`` the reference's
file-level tool functions; ``dist`` row-strip multi-GPU orchestration over torch.distributed (RCCL).
"""
from ._lib import LIB_PATH, TdxError, load  # noqa: F401
from .api import (ANG_NODATA, AREA_NODATA, FEL_NODATA, P_NODATA, SLOPE_NODATA, Context, raster_info, read_raster,  # noqa: F401
                  synth_base_wavelength, write_raster)

__version__ = "0.1.0"
