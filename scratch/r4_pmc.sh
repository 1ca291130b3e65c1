#!/bin/bash
# scratch/r4_pmc.sh -- the round's counter evidence (profiles/collect_pmc.sh), then the per-kernel stats of the default line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash profiles/collect_pmc.sh r4 "$@" 2>&1 | cut -c1-700
