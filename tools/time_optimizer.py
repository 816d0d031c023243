"""Device time of the fused optimizer launch on the DCGAN networks; DDL_OPT_BLOCKS sweeps its grid cap."""
import torch, sys
from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer
from distributeddeeplearning_amd.ops import hip
t = GANTrainer(GANConfig(batch=64), "cuda")
for name, opt in (("G", t.optG), ("D", t.optD)):
    f = lambda: hip.optimizer_step(opt, 1.0)
    print(name, round(hip._time_us(f, reps=6, inner=8), 2), "us")
