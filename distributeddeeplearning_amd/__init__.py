"""distributeddeeplearning_amd -- MI355X-native distributed ensemble / DCGAN trainer.

Capabilities of bbondd/DistributedDeepLearning (random-architecture ensemble search, parallel
member training, transfer generations, softmax-vote ensembling, Keras-2.2.4 checkpoint layout,
DCGAN) re-designed for gfx950: hand-written HIP/MFMA kernels, flat-buffer RCCL data parallelism,
hipGraph-captured steps.  See SURVEY.md for the capability map.
"""
__version__ = "0.1.0"
