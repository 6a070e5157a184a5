#!/usr/bin/env python3
"""same CLI as the reference's src/search.py (query-image / query-c2df)"""
import sys
import sgic_amd  # noqa: F401
from sgic_amd.search import main
sys.exit(main())
