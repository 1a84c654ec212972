#!/bin/bash
cd $GRAFT_REPO_ROOT
RUNS=base:exact,base:1024,fa_noload:1024 TAG=r04d bash tools/gpu/tune.sh
bash tools/gpu/pmc_join.sh r04d base:1024
