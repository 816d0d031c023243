"""Drop-in for the reference's Test.py / RayTest.py smoke scripts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.cli.selftest import main

if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
