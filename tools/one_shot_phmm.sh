#!/bin/bash
python tools/one_shot_phmm.py && AGX_PHMM_NO_TRAINS=1 python tools/one_shot_phmm.py && AGX_PHMM_NO_TRAINS=1 AGX_PHMM_NO_ROWS=1 python tools/one_shot_phmm.py && python tools/one_shot_phmm.py
