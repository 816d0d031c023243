"""Drop-in for the reference's DDL_Ray3.py: same workflow, preset `ddl_ray3` (see cli/ddl.py for flags)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.cli.ddl import main

if __name__ == "__main__":
    sys.exit(main(["--preset", "ddl_ray3"] + sys.argv[1:]))
