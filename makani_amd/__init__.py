"""makani_amd -- MI355X-native SFNO spectral stack (HIP kernels behind Makani's module API).

Mirrors the reference's interfaces for the hot path only:

* ``makani_amd.sht``                  RealSHT / InverseRealSHT (torch-harmonics seam)
* ``makani_amd.distributed``          DistributedRealSHT / DistributedInverseRealSHT, transposes
* ``makani_amd.spectral_convolution`` SpectralConv / FactorizedSpectralConv
* ``makani_amd.sfnonet``              FourierNeuralOperatorBlock / SphericalFourierNeuralOperatorNet
* ``makani_amd.comm``                 h / w / data process groups over torch.distributed (RCCL)

All arithmetic of the spectral path runs in libmakani_amd.so (include/makani_amd.h).
"""
__version__ = "0.1.0"
