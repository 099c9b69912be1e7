#!/bin/bash
# see tools/phmm_split_probe.py
python tools/phmm_split_probe.py 64 16 8 4 &&
AGX_PHMM_FORCE_C=19 python tools/phmm_split_probe.py 64 16 8 4 &&
AGX_PHMM_FORCE_C=30 python tools/phmm_split_probe.py 64 60 48 36 24 12 &&
AGX_PHMM_FORCE_C=25 python tools/phmm_split_probe.py 64 60 40 20 &&
AGX_PHMM_FORCE_C=29 python tools/phmm_split_probe.py 64 48 &&
AGX_PHMM_FORCE_C=10 python tools/phmm_split_probe.py 4 2
