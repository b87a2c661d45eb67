#!/bin/bash
# the round's closing sequence on the final tree: GPU suite (train bands with their report), profiles, bench lines
cd $GRAFT_REPO_ROOT
./tools/r05_gpu38.sh
./tools/r05_gpu28.sh
