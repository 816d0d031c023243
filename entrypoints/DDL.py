"""Drop-in for the reference's DDL.py: same workflow, preset `ddl` (see cli/ddl.py for flags)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributeddeeplearning_amd.cli.ddl import main

if __name__ == "__main__":
    sys.exit(main(["--preset", "ddl"] + sys.argv[1:]))
