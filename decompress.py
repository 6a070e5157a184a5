#!/usr/bin/env python3
"""`python decompress.py --dataset_dir <dir of .c2df> --save_dir ...` -- same CLI as the reference's src/decompress.py."""
import sys
import sgic_amd  # noqa: F401
from sgic_amd.decompress import main
sys.exit(main())
