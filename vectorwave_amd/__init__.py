"""vectorwave_amd: MI355X-native batched MODWT / SWT engine.

Host mirror of the reference's MODWTTransform / MultiLevelMODWTTransform /
VectorWaveSwtAdapter API over libvwgpu.so (hand-written gfx950 HIP kernels
behind the C ABI in include/vwgpu.h).  There is no CPU compute path.
"""
from .api import (BatchSIMDMODWT, BoundaryMode, IllegalStateException, InvalidArgumentException, InvalidSignalException, MODWTResult,
                  InvalidStateException, MODWTStreamingDenoiser, MODWTStreamingTransform, MODWTTransform, MultiLevelMODWTStreamingTransform, NoiseEstimation,
                  SparseSWTResult, StreamingStatistics, SWTResult,
                  MultiLevelMODWTResult, MultiLevelMODWTTransform, MutableMultiLevelMODWTResult, ParallelMultiLevelMODWT,
                  ThresholdMethod, ThresholdType, VectorWaveSwtAdapter, WaveletDenoiser, WaveletOperations,
                  WaveletTransformException)
from .wavelets import Wavelet, available_wavelets, get_wavelet

__all__ = [
    "BoundaryMode", "InvalidArgumentException", "InvalidSignalException", "MODWTResult", "MODWTTransform",
    "MultiLevelMODWTResult", "MultiLevelMODWTTransform", "MutableMultiLevelMODWTResult", "VectorWaveSwtAdapter",
    "WaveletOperations", "WaveletTransformException", "Wavelet", "available_wavelets", "get_wavelet",
    "WaveletDenoiser", "ThresholdMethod", "ThresholdType", "BatchSIMDMODWT", "ParallelMultiLevelMODWT",
    "MODWTStreamingDenoiser", "NoiseEstimation", "IllegalStateException", "MODWTStreamingTransform", "MultiLevelMODWTStreamingTransform",
    "InvalidStateException", "StreamingStatistics", "SWTResult", "SparseSWTResult",
]
