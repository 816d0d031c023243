"""Drop-in for the reference's GAN_distributed.py (see cli/gan.py for flags; torchrun = data parallel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.cli.gan import main

if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
