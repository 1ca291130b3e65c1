#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
timeout 900 python3 -m pytest tests/test_twins.py -x -q -p no:cacheprovider -k "${K:-degrades}" 2>&1 | tail -12
