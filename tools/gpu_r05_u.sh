#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2 3; do python tools/chain_hammer.py 60 16x900 2>&1 | tail -6; done
