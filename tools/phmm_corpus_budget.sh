#!/bin/bash
AGX_TRACE_POOL= python tools/phmm_corpus_budget.py 2>/dev/null
for b in 12288 16384 24576 30720 40960; do AGX_PHMM_TAB_BUDGET=$b python tools/phmm_corpus_budget.py 2>/dev/null; done
