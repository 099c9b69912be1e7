#!/bin/bash
# agx_sw_score on config 2, config 4's shard and the full config 4 against the piece size of the one-shot pipeline
# (AGX_SW_PIECE_MB / AGX_SW_PIECE_MIN_PAIRS, tuning build; default 64 MB / 65536 pairs)
for cfg in "64 65536" "32 32768" "16 16384" "8 16384" "4 8192"; do
  set -- $cfg
  echo "== pieces of $1 MB, at least $2 pairs"
  AGX_SW_PIECE_MB=$1 AGX_SW_PIECE_MIN_PAIRS=$2 python tools/one_shot_sw.py 2>/dev/null | grep -v multi
done
