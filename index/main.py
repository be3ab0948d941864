"""`cd index && python main.py ...` -- the reference's entry point (index/main.py, index/run.sh),
forwarded to the MI355X implementation in lc-rec_amd/main.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lcrec_amd.main import main  # noqa: E402

if __name__ == "__main__":
    main()
