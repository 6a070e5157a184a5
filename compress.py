#!/usr/bin/env python3
"""`python compress.py --dataset_dir ... --save_dir ...` -- same CLI as the reference's src/compress.py."""
import sys
import sgic_amd  # noqa: F401
from sgic_amd.compress import main
sys.exit(main())
