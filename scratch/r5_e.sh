#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
timeout 900 python3 -m pytest tests/test_sixel_delta_e.py -x -q -s -p no:cacheprovider 2>&1 | tail -40
