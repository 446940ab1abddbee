#!/bin/bash
# full GPU check of the tree: all gpu tests, then the default bench line
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3_full_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_full_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/r3_full_bench.json 2> gpurun_out/r3_full_bench.err; rc=$?
tail -c 600 gpurun_out/r3_full_bench.json; tail -3 gpurun_out/r3_full_bench.err
exit $rc
