#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for a in 20000000 50000000 100000000; do ANCHORS=$a VARIANTS="main" bash profiles/ab_variants.sh; done
python profiles/one_chunk.py 2>&1 | grep -v HW_QU | head -1
