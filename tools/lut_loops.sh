#!/bin/bash
python tools/lut_loops.py && AGX_PHMM_LUT_ONE_LOOP=1 python tools/lut_loops.py && python tools/lut_loops.py && AGX_PHMM_LUT_ONE_LOOP=1 python tools/lut_loops.py
