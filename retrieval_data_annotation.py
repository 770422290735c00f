#!/usr/bin/env python3
"""Drop-in CLI: ``python retrieval_data_annotation.py <dataset> <timestamp> <threshold>`` (reference
``retrieval_data_annotation.py:109-200``), Jaccard matrices computed by the gfx950 library."""
import sys

from rag4dyg_amd.annotation import main

if __name__ == '__main__':
    main(sys.argv)
