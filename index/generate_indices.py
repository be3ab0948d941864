"""`cd index && python generate_indices.py --ckpt_path ... --output_dir ...` -- the reference's
index/generate_indices.py (a script with hard-coded paths), forwarded to lc-rec_amd/generate_indices.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lcrec_amd.generate_indices import main  # noqa: E402

if __name__ == "__main__":
    main()
