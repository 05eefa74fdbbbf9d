#!/bin/bash
cd $GRAFT_REPO_ROOT/scripts/microbench && ./dense_bench
